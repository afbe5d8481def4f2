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
    float* __restrict__ drowbias, int ldrb, int N, int Tn, int C, int Cpad, int G) {
  extern __shared__ float lds[];   // [4 waves][3][Cg]: dgamma, dbeta, per-sequence d_rowbias
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Cg = C / G, E = Tn * Cg;
  const int g = blockIdx.x;
  float* my = lds + wave * 3 * Cg;
  for (int c = lane; c < 3 * Cg; c += 64) my[c] = 0.f;
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
        if (drowbias) atomicAdd(&my[2 * Cg + cl], go);
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
        int t = e / Cg, c = g * Cg + (e - t * Cg);
        dy[(rbase + t) * ldg + c] = from_f32<T>(rstd * (dxh[i] - s1 - xh[i] * s2));
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
        drowbias[(size_t)n * ldrb + g * Cg + c] = my[2 * Cg + c];
        my[2 * Cg + c] = 0.f;
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * Cg; c += 256) {
    float a = lds[c] + lds[3 * Cg + c] + lds[6 * Cg + c] + lds[9 * Cg + c];
    if (c < Cg) atomicAdd(&dgamma[g * Cg + c], a);
    else atomicAdd(&dbeta[g * Cg + c - Cg], a);
  }
}

extern "C" int madiff_gn_mish_fwd(int dtype, const void* y, int ldy, void* out, int ldo, const float* gamma,
                                  const float* beta, float* mean, float* rstd, int N, int T, int C, int Cpad, int G,
                                  float eps, int mode, const float* rowbias, int ldrb, const void* res, int ldr,
                                  hipStream_t stream) {
  if (N <= 0) return 0;
  if (C <= 0 || G <= 0 || C % G || T <= 0 || (long)T * (C / G) > 64 * GN_MAXE || Cpad < C || ldy < C || ldo < Cpad ||
      (mode == 1 && (!rowbias || ldrb < C)) || (mode == 2 && (!res || ldr < C)) || mode < 0 || mode > 2)
    return -1;
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

// dgamma / dbeta must be zeroed by the caller (they are accumulated with atomics).
extern "C" int madiff_gn_mish_bwd(int dtype, const void* y, int ldy, const void* dout, int ldd, void* dy, int ldg,
                                  const float* gamma, const float* beta, const float* mean, const float* rstd,
                                  float* dgamma, float* dbeta, float* drowbias, int ldrb, int N, int T, int C,
                                  int Cpad, int G, hipStream_t stream) {
  if (N <= 0) return 0;
  if (C <= 0 || G <= 0 || C % G || T <= 0 || (long)T * (C / G) > 64 * GN_MAXE || Cpad < C || ldy < C || ldd < C ||
      ldg < Cpad || (drowbias && ldrb < C))
    return -1;
  int nb = (N + 3) / 4;
  if (nb > 128) nb = 128;
  size_t shmem = (size_t)4 * 3 * (C / G) * sizeof(float);
  dim3 grid(G, nb);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((gn_mish_bwd_kernel<bf16_t>), grid, dim3(256), shmem, stream, (const bf16_t*)y, ldy,
                       (const bf16_t*)dout, ldd, (bf16_t*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, drowbias,
                       ldrb, N, T, C, Cpad, G);
  else
    hipLaunchKernelGGL((gn_mish_bwd_kernel<float>), grid, dim3(256), shmem, stream, (const float*)y, ldy,
                       (const float*)dout, ldd, (float*)dy, ldg, gamma, beta, mean, rstd, dgamma, dbeta, drowbias,
                       ldrb, N, T, C, Cpad, G);
  MADIFF_CHECK_LAUNCH();
  return 0;
}
