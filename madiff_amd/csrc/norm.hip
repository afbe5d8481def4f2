// Fused GroupNorm(8) + Mish + {timestep-embedding add | residual add}, forward and backward.
//
// Reference semantics: helpers.py:63-71 (GroupNorm over (C/G channels x T) of one sequence, eps 1e-5,
// affine) followed by Mish, then temporal.py:197-200 adds Linear(Mish(t_emb))[:, :, None] (block 0) or
// the residual branch (block 1).  Layout is channels-last [N*T, ld]; one wave owns one (sequence,
// group) slab (T x Cg <= 1024 elements) held entirely in registers: one HBM read, one write.
// Bandwidth-bound; statistics, normalisation and Mish are fp32 whatever the storage type.
#include "common.h"

#define GN_MAXE 16   // elements per lane: slabs up to 64*16 = 1024 elements

template <typename T>
__global__ __launch_bounds__(256) void gn_mish_fwd_kernel(
    const T* __restrict__ y, int ldy, T* __restrict__ out, int ldo, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean_out, float* __restrict__ rstd_out, int N, int Tn, int C,
    int Cpad, int G, float eps, int mode, const float* __restrict__ rowbias, int ldrb, const T* __restrict__ res,
    int ldr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Cg = C / G, E = Tn * Cg;
  const long slabs = (long)N * G;
  for (long s = (long)blockIdx.x * 4 + wave; s < slabs; s += (long)gridDim.x * 4) {
    const int n = (int)(s / G), g = (int)(s - (long)n * G);
    const size_t rbase = (size_t)n * Tn;
    float v[GN_MAXE];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXE; i++) {
      int e = lane + 64 * i;
      v[i] = 0.f;
      if (e < E) {
        int t = e / Cg, c = e - t * Cg;
        v[i] = to_f32(y[(rbase + t) * ldy + g * Cg + c]);
        sum += v[i];
      }
    }
    const float mean = wave_sum(sum) / (float)E;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXE; i++) {
      int e = lane + 64 * i;
      if (e < E) { float d = v[i] - mean; sq += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)E + eps);
    if (lane == 0 && mean_out) { mean_out[s] = mean; rstd_out[s] = rstd; }
#pragma unroll
    for (int i = 0; i < GN_MAXE; i++) {
      int e = lane + 64 * i;
      if (e < E) {
        int t = e / Cg, c = g * Cg + (e - t * Cg);
        float z = (v[i] - mean) * rstd * gamma[c] + beta[c];
        float o = mish_f(z);
        if (mode == 1) o += rowbias[(size_t)n * ldrb + c];
        else if (mode == 2) o += to_f32(res[(rbase + t) * ldr + c]);
        out[(rbase + t) * ldo + c] = from_f32<T>(o);
      }
    }
    if (g == G - 1 && Cpad > C) {   // keep the zero-padding invariant of the activation layout
      for (int e = lane; e < Tn * (Cpad - C); e += 64) {
        int t = e / (Cpad - C), c = C + e - t * (Cpad - C);
        out[(rbase + t) * ldo + c] = from_f32<T>(0.f);
      }
    }
  }
}

// Backward.  grid = (G, NB): a workgroup owns one group g and strides over sequences, so the per-channel
// sums (dgamma, dbeta) are reduced in LDS first and leave as ONE atomic per channel per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void gn_mish_bwd_kernel(
    const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd, T* __restrict__ dy, int ldg,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dbias, float* __restrict__ drowbias, int ldrb, int N, int Tn, int C, int Cpad, int G) {
  extern __shared__ float lds[];   // [4 waves][4][Cg]: dgamma, dbeta, dbias (= sum dy), per-sequence d_rowbias
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Cg = C / G, E = Tn * Cg;
  const int g = blockIdx.x;
  float* my = lds + wave * 4 * Cg;
  for (int c = lane; c < 4 * Cg; c += 64) my[c] = 0.f;
  __syncthreads();
  for (int n = blockIdx.y * 4 + wave; n < N; n += gridDim.y * 4) {
    const size_t rbase = (size_t)n * Tn;
    const float mean = mean_in[(size_t)n * G + g], rstd = rstd_in[(size_t)n * G + g];
    float xh[GN_MAXE], dxh[GN_MAXE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GN_MAXE; i++) {
      int e = lane + 64 * i;
      xh[i] = 0.f; dxh[i] = 0.f;
      if (e < E) {
        int t = e / Cg, cl = e - t * Cg, c = g * Cg + cl;
        float x = (to_f32(y[(rbase + t) * ldy + c]) - mean) * rstd;
        float gm = gamma[c];
        float z = x * gm + beta[c];
        float go = to_f32(dout[(rbase + t) * ldd + c]);
        float dz = go * mish_grad_f(z);
        atomicAdd(&my[cl], dz * x);
        atomicAdd(&my[Cg + cl], dz);
        if (drowbias) atomicAdd(&my[3 * Cg + cl], go);
        xh[i] = x;
        dxh[i] = dz * gm;
        s1 += dxh[i];
        s2 += dxh[i] * x;
      }
    }
    s1 = wave_sum(s1) / (float)E;
    s2 = wave_sum(s2) / (float)E;
#pragma unroll
    for (int i = 0; i < GN_MAXE; i++) {
      int e = lane + 64 * i;
      if (e < E) {
        int t = e / Cg, cl = e - t * Cg, c = g * Cg + cl;
        float d = rstd * (dxh[i] - s1 - xh[i] * s2);
        dy[(rbase + t) * ldg + c] = from_f32<T>(d);
        if (dbias) atomicAdd(&my[2 * Cg + cl], d);
      }
    }
    if (g == G - 1 && Cpad > C) {
      for (int e = lane; e < Tn * (Cpad - C); e += 64) {
        int t = e / (Cpad - C), c = C + e - t * (Cpad - C);
        dy[(rbase + t) * ldg + c] = from_f32<T>(0.f);
      }
    }
    if (drowbias) {   // d temb[n, c] = sum_t dout: owned by this wave, plain store, then reset
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): LDS atomics of this wave are done
      for (int c = lane; c < Cg; c += 64) {
        drowbias[(size_t)n * ldrb + g * Cg + c] = my[3 * Cg + c];
        my[3 * Cg + c] = 0.f;
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * Cg; c += 256) {
    float a = lds[c] + lds[4 * Cg + c] + lds[8 * Cg + c] + lds[12 * Cg + c];
    int q = c / Cg;
    float* dst = q == 0 ? dgamma : (q == 1 ? dbeta : dbias);
    if (dst) atomicAdd(&dst[g * Cg + c - q * Cg], a);
  }
}

// ================================================================================================
// Vectorised fast path (16-byte accesses): used when a group's channels are a multiple of one vector and the
// vectors-per-row count is a power of two <= 64, i.e. every layer of the real model (Cg = 16, 64, 128).
// One wave owns a (sequence, group) slab: <= VMAX vectors per lane, everything in registers.
// Backward: a workgroup owns one group and strides over sequences; per-channel sums (dgamma, dbeta and the conv
// bias gradient sum_m dy) stay in registers across all its sequences and are reduced once at the end.
// ================================================================================================
template <typename T> struct Vec;
template <> struct Vec<bf16_t> { static constexpr int E = 8; };
template <> struct Vec<float>  { static constexpr int E = 4; };

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float (&f)[Vec<T>::E]);
template <> __device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float (&f)[8]) {
  bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; i++) f[i] = (float)v[i];
}
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float (&f)[4]) {
  f32x4 v = *(const f32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; i++) f[i] = v[i];
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// a raw 16-byte vector (loaded early, converted when it is used)
template <typename T> __device__ __forceinline__ void cvt_vec(const u32x4& v, float (&f)[Vec<T>::E]);
template <> __device__ __forceinline__ void cvt_vec<bf16_t>(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = __builtin_bit_cast(float, v[i] << 16);
    f[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void cvt_vec<float>(const u32x4& v, float (&f)[4]) {
  const f32x4 w = __builtin_bit_cast(f32x4, v);     // whole-vector cast (an element-wise cast of v[i] lost lanes 2, 3)
#pragma unroll
  for (int i = 0; i < 4; i++) f[i] = w[i];
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float (&f)[Vec<T>::E]);
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = (bf16_t)f[i];
  *(bf16x8*)p = v;
}
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float (&f)[4]) {
  *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]};
}

#define GN_VMAX 4

template <typename T>
__global__ __launch_bounds__(256) void gn_mish_fwd_vec_kernel(
    const T* __restrict__ y, int ldy, T* __restrict__ out, int ldo, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean_out, float* __restrict__ rstd_out, int N, int Tn, int C,
    int G, float eps, int mode, const float* __restrict__ rowbias, int ldrb, const T* __restrict__ res, int ldr,
    int vpr_log2) {
  constexpr int E = Vec<T>::E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Cg = C / G, VPR = 1 << vpr_log2, NV = Tn * VPR;
  const float inv_e = 1.f / (float)(Tn * Cg);
  const long slabs = (long)N * G;
  for (long s = (long)blockIdx.x * 4 + wave; s < slabs; s += (long)gridDim.x * 4) {
    const int n = (int)(s / G), g = (int)(s - (long)n * G);
    const size_t rbase = (size_t)n * Tn;
    const int cv = lane & (VPR - 1);
    const int c0 = g * Cg + cv * E;
    float v[GN_VMAX][E];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < GN_VMAX; i++) {
      int vi = lane + 64 * i;
      if (vi < NV) {
        load_vec<T>(y + (rbase + (vi >> vpr_log2)) * ldy + c0, v[i]);
#pragma unroll
        for (int e = 0; e < E; e++) sum += v[i][e];
      }
    }
    const float mean = wave_sum(sum) * inv_e;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < GN_VMAX; i++)
      if (lane + 64 * i < NV) {
#pragma unroll
        for (int e = 0; e < E; e++) { float d = v[i][e] - mean; sq += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(sq) * inv_e + eps);
    if (lane == 0 && mean_out) { mean_out[s] = mean; rstd_out[s] = rstd; }
    float gm[E], bt[E], rb[E];                 // z = v * gm + bt with the normalisation folded in
#pragma unroll
    for (int e = 0; e < E; e++) {
      gm[e] = rstd * gamma[c0 + e];
      bt[e] = beta[c0 + e] - mean * gm[e];
      rb[e] = mode == 1 ? rowbias[(size_t)n * ldrb + c0 + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < GN_VMAX; i++) {
      int vi = lane + 64 * i;
      if (vi < NV) {
        const size_t row = rbase + (vi >> vpr_log2);
        float o[E];
        if (mode == 2) load_vec<T>(res + row * ldr + c0, o);
#pragma unroll
        for (int e = 0; e < E; e++) {
          float z = v[i][e] * gm[e] + bt[e];
          o[e] = mish_f(z) + (mode == 2 ? o[e] : rb[e]);
        }
        store_vec<T>(out + row * ldo + c0, o);
      }
    }
  }
}

// grid = (G, NB).  Per-channel accumulators: lane owns channels c0..c0+E-1 for all its sequences.
template <typename T>
__global__ __launch_bounds__(256) void gn_mish_bwd_vec_kernel(
    const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd, T* __restrict__ dy, int ldg,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dbias, float* __restrict__ drowbias, int ldrb, int N, int Tn, int C, int G, int vpr_log2) {
  constexpr int E = Vec<T>::E;
  __shared__ float red[4][3][128 * 2];     // [wave][dgamma|dbeta|dbias][channels of one group] (Cg <= 256)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Cg = C / G, VPR = 1 << vpr_log2, NV = Tn * VPR;
  const float inv_e = 1.f / (float)(Tn * Cg);
  const int g = blockIdx.x;
  const int cv = lane & (VPR - 1);
  const int c0 = g * Cg + cv * E;
  float gm[E], bt[E], a_dg[E], a_db[E], a_dbias[E];
#pragma unroll
  for (int e = 0; e < E; e++) { gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e]; a_dg[e] = a_db[e] = a_dbias[e] = 0.f; }
  for (int n = blockIdx.y * 4 + wave; n < N; n += gridDim.y * 4) {
    const size_t rbase = (size_t)n * Tn;
    const float mean = mean_in[(size_t)n * G + g], rstd = rstd_in[(size_t)n * G + g];
    float xh[GN_VMAX][E], dxh[GN_VMAX][E], rbs[E];
#pragma unroll
    for (int e = 0; e < E; e++) rbs[e] = 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GN_VMAX; i++) {
      int vi = lane + 64 * i;
      if (vi < NV) {
        const size_t row = rbase + (vi >> vpr_log2);
        float yv[E], go[E];
        load_vec<T>(y + row * ldy + c0, yv);
        load_vec<T>(dout + row * ldd + c0, go);
#pragma unroll
        for (int e = 0; e < E; e++) {
          float x = (yv[e] - mean) * rstd;
          float z = x * gm[e] + bt[e];
          float dz = go[e] * mish_grad_f(z);
          a_dg[e] += dz * x;
          a_db[e] += dz;
          rbs[e] += go[e];
          xh[i][e] = x;
          dxh[i][e] = dz * gm[e];
          s1 += dxh[i][e];
          s2 += dxh[i][e] * x;
        }
      }
    }
    s1 = wave_sum(s1) * inv_e;
    s2 = wave_sum(s2) * inv_e;
#pragma unroll
    for (int i = 0; i < GN_VMAX; i++) {
      int vi = lane + 64 * i;
      if (vi < NV) {
        const size_t row = rbase + (vi >> vpr_log2);
        float o[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
          o[e] = rstd * (dxh[i][e] - s1 - xh[i][e] * s2);
          a_dbias[e] += o[e];
        }
        store_vec<T>(dy + row * ldg + c0, o);
      }
    }
    if (drowbias) {   // d temb[n, c] = sum_t dout: reduce over the lanes that share this channel vector
#pragma unroll
      for (int e = 0; e < E; e++) {
        float r = rbs[e];
        for (int o = VPR; o < 64; o <<= 1) r += __shfl_xor(r, o, 64);
        rbs[e] = r;
      }
      if (lane < VPR) {
#pragma unroll
        for (int e = 0; e < E; e++) drowbias[(size_t)n * ldrb + c0 + e] = rbs[e];
      }
    }
  }
  // per-channel reduction: lanes sharing cv, then the 4 waves, then one atomic per channel per workgroup
#pragma unroll
  for (int e = 0; e < E; e++) {
    float a = a_dg[e], b = a_db[e], c = a_dbias[e];
    for (int o = VPR; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
    if (lane < VPR) { red[wave][0][cv * E + e] = a; red[wave][1][cv * E + e] = b; red[wave][2][cv * E + e] = c; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * Cg; i += 256) {
    int q = i / Cg, c = i - q * Cg;
    float a = red[0][q][c] + red[1][q][c] + red[2][q][c] + red[3][q][c];
    float* dst = q == 0 ? dgamma : (q == 1 ? dbeta : dbias);
    if (dst) atomicAdd(&dst[g * Cg + c], a);
  }
}

// ================================================================================================
// Row-block path (the one every layer of the real model takes).  A wave owns, for one sequence, ALL T rows of a
// channel range of CWV 16-byte vectors (a power of two, >= one group): lane = (row r0 = lane / CWV, vector lane % CWV),
// iteration i covers row r0 + i * (64 / CWV), VPL = ceil(T / (64 / CWV)) iterations.  Every lane keeps the same
// channel vector for all its rows, adjacent lanes read adjacent 16-byte vectors (full-row coalescing), and a wave
// covers 1..8 groups, so small slabs (16 channels x 12 steps) no longer leave most lanes idle.  Group statistics are
// butterfly sums over the lane bits that stay inside a group (low log2(Cg/E) bits of the vector index + the row bits).
// ================================================================================================
template <int MASKBITS>
__device__ __forceinline__ float group_sum(float v, int lowbits, int cwv_log2) {
  // lanes of one group differ in bits [0, lowbits) and [cwv_log2, 6)
  for (int b = 0; b < lowbits; b++) v += __shfl_xor(v, 1 << b, 64);
  for (int b = cwv_log2; b < 6; b++) v += __shfl_xor(v, 1 << b, 64);
  return v;
}

template <typename T, int VPL>
__global__ __launch_bounds__(256) void gn_mish_fwd_rb_kernel(
    const T* __restrict__ y, int ldy, T* __restrict__ out, int ldo, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ mean_out, float* __restrict__ rstd_out, int N, int Tn, int C,
    int G, float eps, int mode, const float* __restrict__ rowbias, int ldrb, const T* __restrict__ res, int ldr,
    int cwv_log2, int gv_log2) {                 // CWV = 1 << cwv_log2 vectors per wave row, 1 << gv_log2 vectors per group
  constexpr int E = Vec<T>::E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CWV = 1 << cwv_log2, RPI = 64 >> cwv_log2;
  const int ranges = (C / E) >> cwv_log2;        // channel ranges per sequence
  const int Cg = C / G;
  const float inv_e = 1.f / (float)(Tn * Cg);
  const int cv = lane & (CWV - 1), r0 = lane >> cwv_log2;
  const long items = (long)N * ranges;
  for (long it = (long)blockIdx.x * 4 + wave; it < items; it += (long)gridDim.x * 4) {
    const int n = (int)(it / ranges), rg = (int)(it - (long)n * ranges);
    const int c0 = ((rg << cwv_log2) + cv) * E;
    const size_t rbase = (size_t)n * Tn;
    float v[VPL][E], rs[VPL][E];
    float sum = 0.f;
    // every global load of the item is issued up front (one exposed memory latency per item instead of three)
    float gm[E], bt[E], rb[E];
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int r = r0 + i * RPI;
      if (r < Tn) load_vec<T>(y + (rbase + r) * ldy + c0, v[i]);
    }
    if (mode == 2) {
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int r = r0 + i * RPI;
        if (r < Tn) load_vec<T>(res + (rbase + r) * ldr + c0, rs[i]);
      }
    }
#pragma unroll
    for (int e = 0; e < E; e++) {
      gm[e] = gamma[c0 + e];
      bt[e] = beta[c0 + e];
      rb[e] = mode == 1 ? rowbias[(size_t)n * ldrb + c0 + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < VPL; i++)
      if (r0 + i * RPI < Tn) {
#pragma unroll
        for (int e = 0; e < E; e++) sum += v[i][e];
      }
    const float mean = group_sum<0>(sum, gv_log2, cwv_log2) * inv_e;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++)
      if (r0 + i * RPI < Tn) {
#pragma unroll
        for (int e = 0; e < E; e++) { float d = v[i][e] - mean; sq += d * d; }
      }
    const float rstd = rsqrtf(group_sum<0>(sq, gv_log2, cwv_log2) * inv_e + eps);
    if (mean_out && r0 == 0 && (cv & ((1 << gv_log2) - 1)) == 0) {
      const int g = c0 / Cg;
      mean_out[(size_t)n * G + g] = mean;
      rstd_out[(size_t)n * G + g] = rstd;
    }
#pragma unroll
    for (int e = 0; e < E; e++) {              // z = v * gm + bt with the normalisation folded in
      gm[e] *= rstd;
      bt[e] -= mean * gm[e];
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int r = r0 + i * RPI;
      if (r < Tn) {
        float o[E];
#pragma unroll
        for (int e = 0; e < E; e++) o[e] = mish_f(v[i][e] * gm[e] + bt[e]) + (mode == 2 ? rs[i][e] : rb[e]);
        store_vec<T>(out + (rbase + r) * ldo + c0, o);
      }
    }
  }
}

// grid = (ranges, NB): a workgroup owns one channel range and strides over sequences; every lane keeps its channel
// vector, so dgamma / dbeta / dbias partial sums stay in registers until the end (then lanes -> waves -> one atomic per
// channel per workgroup).
template <typename T, int VPL>
__global__ __launch_bounds__(256) void gn_mish_bwd_rb_kernel(
    const T* __restrict__ y, int ldy, const T* __restrict__ dout, int ldd, T* __restrict__ dy, int ldg,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean_in,
    const float* __restrict__ rstd_in, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dbias, float* __restrict__ drowbias, int ldrb, int N, int Tn, int C, int G, int cwv_log2,
    int gv_log2) {
  constexpr int E = Vec<T>::E;
  __shared__ float red[4][3][64 * 8];          // [wave][dgamma|dbeta|dbias][channels of the range] (CWV * E <= 512)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int CWV = 1 << cwv_log2, RPI = 64 >> cwv_log2;
  const int Cg = C / G;
  const float inv_e = 1.f / (float)(Tn * Cg);
  const int cv = lane & (CWV - 1), r0 = lane >> cwv_log2;
  const int c0 = (((int)blockIdx.x << cwv_log2) + cv) * E;
  const int g = c0 / Cg;
  float gm[E], bt[E], a_dg[E], a_db[E], a_dbias[E];
#pragma unroll
  for (int e = 0; e < E; e++) { gm[e] = gamma[c0 + e]; bt[e] = beta[c0 + e]; a_dg[e] = a_db[e] = a_dbias[e] = 0.f; }
  // Software pipeline over the wave's sequences: the raw 16-byte vectors (and the statistics) of the NEXT sequence are
  // requested before the current one is worked on, so the loads fly under the ~50 VALU instructions per element.
  const int nstep = gridDim.y * 4;
  int n = blockIdx.y * 4 + wave;
  u32x4 ry[VPL], rd[VPL];
  float mean = 0.f, rstd = 0.f;
  // (loads are unconditional - a row past the sequence re-reads its last row, the step past the last sequence re-reads
  // the current one - so the code is straight-line and the waits can be counted instead of draining everything)
  if (n < N) {
    const size_t rb0 = (size_t)n * Tn;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int r = min(r0 + i * RPI, Tn - 1);
      ry[i] = *(const u32x4*)(y + (rb0 + r) * ldy + c0);
      rd[i] = *(const u32x4*)(dout + (rb0 + r) * ldd + c0);
    }
    mean = mean_in[(size_t)n * G + g];
    rstd = rstd_in[(size_t)n * G + g];
  }
  for (; n < N; n += nstep) {
    const size_t rbase = (size_t)n * Tn;
    const int nn = n + nstep < N ? n + nstep : n;
    u32x4 ny[VPL], nd[VPL];
    const size_t rb1 = (size_t)nn * Tn;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int r = min(r0 + i * RPI, Tn - 1);
      ny[i] = *(const u32x4*)(y + (rb1 + r) * ldy + c0);
      nd[i] = *(const u32x4*)(dout + (rb1 + r) * ldd + c0);
    }
    const float nmean = mean_in[(size_t)nn * G + g];
    const float nrstd = rstd_in[(size_t)nn * G + g];
    float xh[VPL][E], dxh[VPL][E], rbs[E];
#pragma unroll
    for (int e = 0; e < E; e++) rbs[e] = 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int r = r0 + i * RPI;
      if (r < Tn) {
        float yv[E], go[E];
        cvt_vec<T>(ry[i], yv);
        cvt_vec<T>(rd[i], go);
#pragma unroll
        for (int e = 0; e < E; e++) {
          const float x = (yv[e] - mean) * rstd;
          const float dz = go[e] * mish_grad_f(x * gm[e] + bt[e]);
          a_dg[e] += dz * x;
          a_db[e] += dz;
          rbs[e] += go[e];
          xh[i][e] = x;
          dxh[i][e] = dz * gm[e];
          s1 += dxh[i][e];
          s2 += dxh[i][e] * x;
        }
      }
    }
    s1 = group_sum<0>(s1, gv_log2, cwv_log2) * inv_e;
    s2 = group_sum<0>(s2, gv_log2, cwv_log2) * inv_e;
#pragma unroll
    for (int i = 0; i < VPL; i++) {
      const int r = r0 + i * RPI;
      if (r < Tn) {
        float o[E];
#pragma unroll
        for (int e = 0; e < E; e++) {
          o[e] = rstd * (dxh[i][e] - s1 - xh[i][e] * s2);
          a_dbias[e] += o[e];
        }
        store_vec<T>(dy + (rbase + r) * ldg + c0, o);
      }
    }
    if (drowbias) {   // d temb[n, c] = sum_t dout: reduce over the lanes that share this channel vector (the row bits)
#pragma unroll
      for (int e = 0; e < E; e++) {
        float r = rbs[e];
        for (int b = cwv_log2; b < 6; b++) r += __shfl_xor(r, 1 << b, 64);
        rbs[e] = r;
      }
      if (r0 == 0) {
#pragma unroll
        for (int e = 0; e < E; e++) drowbias[(size_t)n * ldrb + c0 + e] = rbs[e];
      }
    }
#pragma unroll
    for (int i = 0; i < VPL; i++) { ry[i] = ny[i]; rd[i] = nd[i]; }
    mean = nmean;
    rstd = nrstd;
  }
#pragma unroll
  for (int e = 0; e < E; e++) {
    float a = a_dg[e], b = a_db[e], c = a_dbias[e];
    for (int bb = cwv_log2; bb < 6; bb++) { a += __shfl_xor(a, 1 << bb, 64); b += __shfl_xor(b, 1 << bb, 64); c += __shfl_xor(c, 1 << bb, 64); }
    if (r0 == 0) { red[wave][0][cv * E + e] = a; red[wave][1][cv * E + e] = b; red[wave][2][cv * E + e] = c; }
  }
  __syncthreads();
  const int CW = CWV * E, cbase = ((int)blockIdx.x << cwv_log2) * E;
  for (int i = threadIdx.x; i < 3 * CW; i += 256) {
    const int q = i / CW, c = i - q * CW;
    const float a = red[0][q][c] + red[1][q][c] + red[2][q][c] + red[3][q][c];
    float* dst = q == 0 ? dgamma : (q == 1 ? dbeta : dbias);
    if (dst) atomicAdd(&dst[cbase + c], a);
  }
}

// Geometry of the row-block path: vectors per wave row (cwv_log2), vectors per group (gv_log2), iterations (vpl).
// Returns 0 when the layer is not covered (the one-slab-per-wave kernels above take it).
static int rb_path(int dtype, int T, int C, int Cpad, int G, int* cwv_log2, int* gv_log2, int* vpl) {
  const int E = dtype == MADIFF_BF16 ? 8 : 4;
  const int Cg = C / G;
  if (Cpad != C || Cg % E) return 0;
  const int gv = Cg / E, rowv = C / E;
  if ((gv & (gv - 1)) || gv > 64 || (rowv & (rowv - 1))) return 0;
  int best = -1, best_vpl = 0, best_waste = 1 << 30;
  for (int l = 0; (1 << l) <= 64 && (1 << l) <= rowv; l++) {
    if ((1 << l) < gv) continue;
    const int rpi = 64 >> l, v = (T + rpi - 1) / rpi;
    if (v > 8) continue;
    const int waste = v * rpi - T;                // idle row slots per wave
    // least padding first, then the iteration count closest to 4 (enough loads in flight, still plenty of waves)
    static int pref = -1;
    if (pref < 0) { const char* e = getenv("MADIFF_GN_VPL"); pref = e ? atoi(e) : 4; }
    const int score = waste * 16 + (v > pref ? v - pref : pref - v);
    if (score < best_waste) { best_waste = score; best = l; best_vpl = v; }
  }
  if (best < 0) return 0;
  int gl = 0;
  while ((1 << gl) < gv) gl++;
  *cwv_log2 = best; *gv_log2 = gl; *vpl = best_vpl;
  return 1;
}

#define GN_RB_SWITCH(VPL_, CALL)  \
  switch (VPL_) {                 \
    case 1: { CALL(1); break; }   \
    case 2: { CALL(2); break; }   \
    case 3: { CALL(3); break; }   \
    case 4: { CALL(4); break; }   \
    case 5: { CALL(5); break; }   \
    case 6: { CALL(6); break; }   \
    case 7: { CALL(7); break; }   \
    default: { CALL(8); break; }  \
  }

static int gn_fwd_rb(int dtype, const void* y, int ldy, void* out, int ldo, const float* gamma, const float* beta, float* mean,
                     float* rstd, int N, int T, int C, int Cpad, int G, float eps, int mode, const float* rowbias, int ldrb,
                     const void* res, int ldr, hipStream_t stream) {
  int cl, gl, vpl;
  const int E = dtype == MADIFF_BF16 ? 8 : 4;
  if (!rb_path(dtype, T, C, Cpad, G, &cl, &gl, &vpl) || ldy % E || ldo % E || (mode == 2 && ldr % E) ||
      ((size_t)y % 16) || ((size_t)out % 16) || (mode == 2 && ((size_t)res % 16)))
    return 1;
  const long items = (long)N * ((C / E) >> cl);
  const int blocks = (int)((items + 3) / 4 < 8192 ? (items + 3) / 4 : 8192);
#define GN_FWD_CALL(V)                                                                                                  \
  if (dtype == MADIFF_BF16)                                                                                             \
    hipLaunchKernelGGL((gn_mish_fwd_rb_kernel<bf16_t, V>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)y, ldy,   \
                       (bf16_t*)out, ldo, gamma, beta, mean, rstd, N, T, C, G, eps, mode, rowbias, ldrb,               \
                       (const bf16_t*)res, ldr, cl, gl);                                                                \
  else                                                                                                                  \
    hipLaunchKernelGGL((gn_mish_fwd_rb_kernel<float, V>), dim3(blocks), dim3(256), 0, stream, (const float*)y, ldy,     \
                       (float*)out, ldo, gamma, beta, mean, rstd, N, T, C, G, eps, mode, rowbias, ldrb,                \
                       (const float*)res, ldr, cl, gl)
  GN_RB_SWITCH(vpl, GN_FWD_CALL)
#undef GN_FWD_CALL
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int gn_bwd_rb(int dtype, const void* y, int ldy, const void* dout, int ldd, void* dy, int ldg, const float* gamma,
                     const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias,
                     float* drowbias, int ldrb, int N, int T, int C, int Cpad, int G, hipStream_t stream) {
  int cl, gl, vpl;
  const int E = dtype == MADIFF_BF16 ? 8 : 4;
  if (!rb_path(dtype, T, C, Cpad, G, &cl, &gl, &vpl) || ldy % E || ldd % E || ldg % E || ((size_t)y % 16) ||
      ((size_t)dout % 16) || ((size_t)dy % 16))
    return 1;
  const int ranges = (C / E) >> cl;
  static int wgs = -1;
  if (wgs < 0) { const char* e = getenv("MADIFF_GN_BWD_WGS"); wgs = e ? atoi(e) : 512; }
  int nb = wgs / ranges;                        // workgroups in flight (each ends with one atomic per channel)
  if (nb > (N + 3) / 4) nb = (N + 3) / 4;
  if (nb < 1) nb = 1;
  dim3 grid(ranges, nb);
#define GN_BWD_CALL(V)                                                                                                  \
  if (dtype == MADIFF_BF16)                                                                                             \
    hipLaunchKernelGGL((gn_mish_bwd_rb_kernel<bf16_t, V>), grid, dim3(256), 0, stream, (const bf16_t*)y, ldy,           \
                       (const bf16_t*)dout, ldd, (bf16_t*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias,      \
                       drowbias, ldrb, N, T, C, G, cl, gl);                                                             \
  else                                                                                                                  \
    hipLaunchKernelGGL((gn_mish_bwd_rb_kernel<float, V>), grid, dim3(256), 0, stream, (const float*)y, ldy,             \
                       (const float*)dout, ldd, (float*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias,        \
                       drowbias, ldrb, N, T, C, G, cl, gl)
  GN_RB_SWITCH(vpl, GN_BWD_CALL)
#undef GN_BWD_CALL
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int vec_path(int dtype, int T, int C, int Cpad, int G, int* vpr_log2) {
  const int E = dtype == MADIFF_BF16 ? 8 : 4;
  const int Cg = C / G;
  if (Cpad != C || Cg % E || Cg > 256) return 0;
  int vpr = Cg / E;
  if (vpr & (vpr - 1) || vpr > 64 || T * vpr > 64 * GN_VMAX) return 0;
  int l = 0;
  while ((1 << l) < vpr) l++;
  *vpr_log2 = l;
  return 1;
}

extern "C" __attribute__((visibility("hidden"))) int madiff_gn_fwd_vec(int dtype, const void* y, int ldy, void* out, int ldo,
    const float* gamma, const float* beta, float* mean, float* rstd, int N, int T, int C, int Cpad, int G, float eps, int mode,
    const float* rowbias, int ldrb, const void* res, int ldr, hipStream_t stream) {
  int l;
  if (!vec_path(dtype, T, C, Cpad, G, &l) || ldy % 8 || ldo % 8 || (mode == 2 && ldr % 8)) return 1;
  long slabs = (long)N * G;
  int blocks = (int)((slabs + 3) / 4 < 16384 ? (slabs + 3) / 4 : 16384);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((gn_mish_fwd_vec_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)y, ldy, (bf16_t*)out,
                       ldo, gamma, beta, mean, rstd, N, T, C, G, eps, mode, rowbias, ldrb, (const bf16_t*)res, ldr, l);
  else
    hipLaunchKernelGGL((gn_mish_fwd_vec_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)y, ldy, (float*)out,
                       ldo, gamma, beta, mean, rstd, N, T, C, G, eps, mode, rowbias, ldrb, (const float*)res, ldr, l);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" __attribute__((visibility("hidden"))) int madiff_gn_bwd_vec(int dtype, const void* y, int ldy, const void* dout, int ldd,
    void* dy, int ldg, const float* gamma, const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta,
    float* dbias, float* drowbias, int ldrb, int N, int T, int C, int Cpad, int G, hipStream_t stream) {
  int l;
  if (!vec_path(dtype, T, C, Cpad, G, &l) || ldy % 8 || ldd % 8 || ldg % 8) return 1;
  int nb = (N + 3) / 4;
  if (nb > 64) nb = 64;
  dim3 grid(G, nb);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((gn_mish_bwd_vec_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)y, ldy, (const bf16_t*)dout, ldd,
                       (bf16_t*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias, drowbias, ldrb, N, T, C, G, l);
  else
    hipLaunchKernelGGL((gn_mish_bwd_vec_kernel<float>), grid, dim3(256), 0, stream, (const float*)y, ldy, (const float*)dout, ldd,
                       (float*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias, drowbias, ldrb, N, T, C, G, l);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int madiff_gn_mish_fwd(int dtype, const void* y, int ldy, void* out, int ldo, const float* gamma,
                                  const float* beta, float* mean, float* rstd, int N, int T, int C, int Cpad, int G,
                                  float eps, int mode, const float* rowbias, int ldrb, const void* res, int ldr,
                                  hipStream_t stream) {
  if (N <= 0) return 0;
  if (C <= 0 || G <= 0 || C % G || T <= 0 || (long)T * (C / G) > 64 * GN_MAXE || Cpad < C || ldy < C || ldo < Cpad ||
      (mode == 1 && (!rowbias || ldrb < C)) || (mode == 2 && (!res || ldr < C)) || mode < 0 || mode > 2)
    return -1;
  {
    int rc = gn_fwd_rb(dtype, y, ldy, out, ldo, gamma, beta, mean, rstd, N, T, C, Cpad, G, eps, mode, rowbias, ldrb, res, ldr,
                       stream);
    if (rc <= 0) return rc;      // 0 = launched, <0 = error, 1 = shape not covered by the row-block path
    rc = madiff_gn_fwd_vec(dtype, y, ldy, out, ldo, gamma, beta, mean, rstd, N, T, C, Cpad, G, eps, mode, rowbias, ldrb,
                           res, ldr, stream);
    if (rc <= 0) return rc;      // ... by the one-slab-per-wave vector path
  }
  long slabs = (long)N * G;
  int blocks = (int)((slabs + 3) / 4 < 8192 ? (slabs + 3) / 4 : 8192);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((gn_mish_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)y, ldy,
                       (bf16_t*)out, ldo, gamma, beta, mean, rstd, N, T, C, Cpad, G, eps, mode, rowbias, ldrb,
                       (const bf16_t*)res, ldr);
  else
    hipLaunchKernelGGL((gn_mish_fwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)y, ldy,
                       (float*)out, ldo, gamma, beta, mean, rstd, N, T, C, Cpad, G, eps, mode, rowbias, ldrb,
                       (const float*)res, ldr);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// dgamma / dbeta / dbias ([C], dbias may be null = gradient of the bias of the conv feeding this norm, sum_m dy) are
// ACCUMULATED with atomics: the caller zeroes them or passes live gradient buffers.
extern "C" int madiff_gn_mish_bwd(int dtype, const void* y, int ldy, const void* dout, int ldd, void* dy, int ldg,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  float* dgamma, float* dbeta, float* dbias, float* drowbias, int ldrb, int N,
                                  int T, int C, int Cpad, int G, hipStream_t stream) {
  if (N <= 0) return 0;
  if (C <= 0 || G <= 0 || C % G || T <= 0 || (long)T * (C / G) > 64 * GN_MAXE || Cpad < C || ldy < C || ldd < C ||
      ldg < Cpad || (drowbias && ldrb < C))
    return -1;
  {
    int rc = gn_bwd_rb(dtype, y, ldy, dout, ldd, dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias, drowbias, ldrb, N, T,
                       C, Cpad, G, stream);
    if (rc <= 0) return rc;
    rc = madiff_gn_bwd_vec(dtype, y, ldy, dout, ldd, dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias, drowbias,
                           ldrb, N, T, C, Cpad, G, stream);
    if (rc <= 0) return rc;
  }
  int nb = (N + 3) / 4;
  if (nb > 128) nb = 128;
  size_t shmem = (size_t)4 * 4 * (C / G) * sizeof(float);
  dim3 grid(G, nb);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((gn_mish_bwd_kernel<bf16_t>), grid, dim3(256), shmem, stream, (const bf16_t*)y, ldy,
                       (const bf16_t*)dout, ldd, (bf16_t*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias,
                       drowbias, ldrb, N, T, C, Cpad, G);
  else
    hipLaunchKernelGGL((gn_mish_bwd_kernel<float>), grid, dim3(256), shmem, stream, (const float*)y, ldy,
                       (const float*)dout, ldd, (float*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, dbias,
                       drowbias, ldrb, N, T, C, Cpad, G);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

