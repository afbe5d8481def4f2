// Cross-agent attention core (temporal.py:104-124 of the reference): tokens are the A agents of one batch
// element, features are the (d, t) pairs of a head.  The Q/K/V projections (+ time-MLP row bias) are done
// by gemm_fwd into one fused [M, 2*Cq + C] matrix; this kernel does scores -> softmax -> P.V -> x + gamma*out
// for one batch element per workgroup with the A x A scores living in LDS.  The contraction sizes here
// (A <= 16) are far below an MFMA tile, so this stage is bandwidth/latency-bound by design: it reads
// q, k, v, x once and writes out once.
#include "common.h"
#include <stdlib.h>

#define ATT_MAX_ENT 1024   // H * A * A

template <typename T>
__device__ __forceinline__ void att_scores(const T* __restrict__ qkv, int ldq, int b, int A, int Tn, int H, int Cq,
                                           float scale, float* P) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dqk = Cq / H, L = Tn * dqk, ent = H * A * A;
  for (int e = wave; e < ent; e += 4) {
    int h = e / (A * A), r = e - h * A * A, a1 = r / A, a2 = r - a1 * A;
    const T* q = qkv + ((size_t)(b * A + a1) * Tn) * ldq + h * dqk;
    const T* k = qkv + ((size_t)(b * A + a2) * Tn) * ldq + Cq + h * dqk;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) {
      int t = i / dqk, d = i - t * dqk;
      s += to_f32(q[(size_t)t * ldq + d]) * to_f32(k[(size_t)t * ldq + d]);
    }
    s = wave_sum(s);
    if (lane == 0) P[e] = s * scale;
  }
  __syncthreads();
  for (int row = threadIdx.x; row < H * A; row += 256) {   // softmax over a2
    float* p = P + row * A;
    float mx = p[0];
    for (int j = 1; j < A; j++) mx = fmaxf(mx, p[j]);
    float sum = 0.f;
    for (int j = 0; j < A; j++) { float ex = __expf(p[j] - mx); p[j] = ex; sum += ex; }
    float inv = 1.f / sum;
    for (int j = 0; j < A; j++) p[j] *= inv;
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, int ldq, const T* __restrict__ x,
                                                       int ldx, T* __restrict__ out, int ldo,
                                                       const float* __restrict__ gamma_p, int B, int A, int Tn, int H,
                                                       int Cq, int C, float scale, int residual) {
  __shared__ float P[ATT_MAX_ENT];
  const int dv = C / H;
  const float gamma = residual ? gamma_p[0] : 1.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    att_scores(qkv, ldq, b, A, Tn, H, Cq, scale, P);
    const int total = A * Tn * C;
    for (int i = threadIdx.x; i < total; i += 256) {
      int c = i % C, r = i / C, t = r % Tn, a1 = r / Tn;
      int h = c / dv;
      const float* p = P + (h * A + a1) * A;
      float o = 0.f;
      for (int a2 = 0; a2 < A; a2++)
        o += p[a2] * to_f32(qkv[((size_t)(b * A + a2) * Tn + t) * ldq + 2 * Cq + c]);
      size_t row = (size_t)(b * A + a1) * Tn + t;
      if (residual) o = to_f32(x[row * ldx + c]) + gamma * o;
      out[row * ldo + c] = from_f32<T>(o);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, int ldq, const T* __restrict__ dout,
                                                       int ldd, T* __restrict__ dqkv, int ldg,
                                                       const float* __restrict__ gamma_p, float* __restrict__ dgamma,
                                                       int B, int A, int Tn, int H, int Cq, int C, float scale,
                                                       int residual) {
  __shared__ float P[ATT_MAX_ENT];
  __shared__ float D[ATT_MAX_ENT];   // raw dP, then dS (scaled)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dv = C / H, dqk = Cq / H, ent = H * A * A, Lv = Tn * dv;
  const float gamma = residual ? gamma_p[0] : 1.f;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    att_scores(qkv, ldq, b, A, Tn, H, Cq, scale, P);
    for (int e = wave; e < ent; e += 4) {   // raw dP[h][a1][a2] = sum_{t,d} dO[a1] * v[a2]
      int h = e / (A * A), r = e - h * A * A, a1 = r / A, a2 = r - a1 * A;
      const T* g = dout + ((size_t)(b * A + a1) * Tn) * ldd + h * dv;
      const T* v = qkv + ((size_t)(b * A + a2) * Tn) * ldq + 2 * Cq + h * dv;
      float s = 0.f;
      for (int i = lane; i < Lv; i += 64) {
        int t = i / dv, d = i - t * dv;
        s += to_f32(g[(size_t)t * ldd + d]) * to_f32(v[(size_t)t * ldq + d]);
      }
      s = wave_sum(s);
      if (lane == 0) D[e] = s;
    }
    __syncthreads();
    if (wave == 0) {   // rows (h, a1): dgamma partial and softmax backward
      float dg = 0.f;
      for (int row = lane; row < H * A; row += 64) {
        float* p = P + row * A;
        float* d = D + row * A;
        float dot = 0.f;
        for (int j = 0; j < A; j++) dot += p[j] * d[j];
        dg += dot;
        for (int j = 0; j < A; j++) d[j] = p[j] * gamma * (d[j] - dot) * scale;
      }
      dg = wave_sum(dg);
      if (lane == 0 && residual) atomicAdd(dgamma, dg);
    }
    __syncthreads();
    const int W = 2 * Cq + C;
    const int total = A * Tn * W;
    for (int i = threadIdx.x; i < total; i += 256) {
      int col = i % W, r = i / W, t = r % Tn, a = r / Tn;
      float o = 0.f;
      if (col < Cq) {            // dq[a1 = a] = sum_a2 dS[h][a][a2] k[a2]
        int h = col / dqk;
        const float* d = D + (h * A + a) * A;
        for (int a2 = 0; a2 < A; a2++) o += d[a2] * to_f32(qkv[((size_t)(b * A + a2) * Tn + t) * ldq + Cq + col]);
      } else if (col < 2 * Cq) { // dk[a2 = a] = sum_a1 dS[h][a1][a] q[a1]
        int cq = col - Cq, h = cq / dqk;
        for (int a1 = 0; a1 < A; a1++)
          o += D[(h * A + a1) * A + a] * to_f32(qkv[((size_t)(b * A + a1) * Tn + t) * ldq + cq]);
      } else {                   // dv[a2 = a] = gamma * sum_a1 P[h][a1][a] dO[a1]
        int c = col - 2 * Cq, h = c / dv;
        for (int a1 = 0; a1 < A; a1++)
          o += P[(h * A + a1) * A + a] * to_f32(dout[((size_t)(b * A + a1) * Tn + t) * ldd + c]);
        o *= gamma;
      }
      dqkv[((size_t)(b * A + a) * Tn + t) * ldg + col] = from_f32<T>(o);
    }
    __syncthreads();
  }
}

extern "C" int madiff_attn_fwd_v2(int dtype, const void* qkv, int ldq, const void* x, int ldx, void* out, int ldo, const float* gamma,
                                  int B, int A, int T, int H, int Cq, int C, float scale, int residual, hipStream_t stream);
extern "C" int madiff_attn_bwd_v2(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg,
                                  const float* gamma, float* dgamma, int B, int A, int T, int H, int Cq, int C, float scale,
                                  int residual, float* drb, int ldrb, hipStream_t stream);
extern "C" int madiff_seqsum(int dtype, const void* x, int ld, float* out, int ldo, long N, int T, int C, hipStream_t stream);

static int att3_fwd(const void* qkv, int ldq, const void* x, int ldx, void* out, int ldo, const float* gamma, int B, int A, int T,
                    int H, int Cq, int C, float scale, int residual, hipStream_t stream);
static int att3_bwd(const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg, const float* gamma, float* dgamma,
                    int B, int A, int T, int H, int Cq, int C, float scale, int residual, float* drb, int ldrb,
                    hipStream_t stream);

static int att_check(int B, int A, int T, int H, int Cq, int C) {
  return (B > 0 && A > 0 && T > 0 && H > 0 && Cq > 0 && C > 0 && Cq % H == 0 && C % H == 0 &&
          H * A * A <= ATT_MAX_ENT) ? 0 : -1;
}

extern "C" int madiff_attn_fwd(int dtype, const void* qkv, int ldq, const void* x, int ldx, void* out, int ldo,
                               const float* gamma, int B, int A, int T, int H, int Cq, int C, int residual,
                               hipStream_t stream) {
  if (att_check(B, A, T, H, Cq, C) || ldq < 2 * Cq + C || ldx < C || ldo < C || (residual && !gamma)) return -1;
  float scale = 1.0f / sqrtf((float)(Cq / H) * (float)T);
  if (dtype == MADIFF_BF16) {          // the feature-axis contractions on MFMA (v3), where the shape allows
    int rc = att3_fwd(qkv, ldq, x, ldx, out, ldo, gamma, B, A, T, H, Cq, C, scale, residual, stream);
    if (rc <= 0) return rc;
  }
  {
    int rc = madiff_attn_fwd_v2(dtype, qkv, ldq, x, ldx, out, ldo, gamma, B, A, T, H, Cq, C, scale, residual, stream);
    if (rc <= 0) return rc;
  }
  int blocks = B < 4096 ? B : 4096;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((attn_fwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)qkv, ldq,
                       (const bf16_t*)x, ldx, (bf16_t*)out, ldo, gamma, B, A, T, H, Cq, C, scale, residual);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)qkv, ldq,
                       (const float*)x, ldx, (float*)out, ldo, gamma, B, A, T, H, Cq, C, scale, residual);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_attn_bwd_rb(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg,
                                  const float* gamma, float* dgamma, float* drowbias, int ldrb, int B, int A, int T, int H,
                                  int Cq, int C, int residual, hipStream_t stream);

// dgamma (1 float) must be zeroed by the caller; dqkv gets every column of the fused [q | k | v] gradient.
extern "C" int madiff_attn_bwd(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg,
                               const float* gamma, float* dgamma, int B, int A, int T, int H, int Cq, int C,
                               int residual, hipStream_t stream) {
  return madiff_attn_bwd_rb(dtype, qkv, ldq, dout, ldd, dqkv, ldg, gamma, dgamma, nullptr, 0, B, A, T, H, Cq, C, residual, stream);
}

// madiff_attn_bwd that also returns the gradient of the per-sequence row bias added to the fused [q | k | v] projection
// (the timestep conditioning, temporal.py:100-111): drowbias[(b,a), col] = sum_t dqkv[(b,a,t), col], fp32 [B*A, ldrb],
// every column in [0, 2Cq + C) written.  NULL = not wanted.
extern "C" int madiff_attn_bwd_rb(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg,
                                  const float* gamma, float* dgamma, float* drowbias, int ldrb, int B, int A, int T, int H,
                                  int Cq, int C, int residual, hipStream_t stream) {
  if (att_check(B, A, T, H, Cq, C) || ldq < 2 * Cq + C || ldd < C || ldg < 2 * Cq + C || (residual && (!gamma || !dgamma)) ||
      (drowbias && ldrb < 2 * Cq + C))
    return -1;
  float scale = 1.0f / sqrtf((float)(Cq / H) * (float)T);
  if (dtype == MADIFF_BF16) {
    int rc = att3_bwd(qkv, ldq, dout, ldd, dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual, drowbias, ldrb, stream);
    if (rc <= 0) return rc;
  }
  {
    int rc = madiff_attn_bwd_v2(dtype, qkv, ldq, dout, ldd, dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual,
                                drowbias, ldrb, stream);
    if (rc <= 0) return rc;
    if (drowbias) {      // no room for the sums next to the staged q / k slices: the same kernel without them, then the sums
      rc = madiff_attn_bwd_v2(dtype, qkv, ldq, dout, ldd, dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual, nullptr,
                              0, stream);
      if (rc < 0) return rc;
      if (rc == 0) return madiff_seqsum(dtype, dqkv, ldg, drowbias, ldrb, (long)B * A, T, 2 * Cq + C, stream);
    }
  }
  int blocks = B < 4096 ? B : 4096;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)qkv, ldq,
                       (const bf16_t*)dout, ldd, (bf16_t*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual);
  else
    hipLaunchKernelGGL((attn_bwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)qkv, ldq,
                       (const float*)dout, ldd, (float*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual);
  MADIFF_CHECK_LAUNCH();
  if (drowbias) return madiff_seqsum(dtype, dqkv, ldg, drowbias, ldrb, (long)B * A, T, 2 * Cq + C, stream);
  return 0;
}

// ================================================================================================
// v2: one workgroup per (batch element, head), 16-byte accesses, q/k head slices staged in LDS as fp32.
// Used when A <= 8, d_qk and d_v are multiples of one vector and 2*A*T*d_qk floats fit in LDS; the generic
// kernels above remain the fallback.
// ================================================================================================
template <typename T> struct AVec;
template <> struct AVec<bf16_t> { static constexpr int E = 8; };
template <> struct AVec<float>  { static constexpr int E = 4; };
template <typename T> __device__ __forceinline__ void ald(const T* p, float (&f)[AVec<T>::E]);
template <> __device__ __forceinline__ void ald<bf16_t>(const bf16_t* p, float (&f)[8]) {
  bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; i++) f[i] = (float)v[i];
}
template <> __device__ __forceinline__ void ald<float>(const float* p, float (&f)[4]) {
  f32x4 v = *(const f32x4*)p;
#pragma unroll
  for (int i = 0; i < 4; i++) f[i] = v[i];
}
template <typename T> __device__ __forceinline__ void ast(T* p, const float (&f)[AVec<T>::E]);
template <> __device__ __forceinline__ void ast<bf16_t>(bf16_t* p, const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = (bf16_t)f[i];
  *(bf16x8*)p = v;
}
template <> __device__ __forceinline__ void ast<float>(float* p, const float (&f)[4]) {
  *(f32x4*)p = (f32x4){f[0], f[1], f[2], f[3]};
}

#define ATT2_AMAX 8

// stage q and k head slices: qs[a][t*dqk + d], ks likewise; then P = softmax_a2(q.k * scale) in Ps[a1*A + a2]
template <typename T>
__device__ __forceinline__ void att2_scores(const T* __restrict__ qkv, int ldq, int b, int h, int A, int Tn, int dqk,
                                            int Cq, float scale, float* qs, float* ks, float* Ps) {
  constexpr int E = AVec<T>::E;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int L = Tn * dqk, vpr = dqk / E, nv = A * Tn * vpr;
  for (int i = threadIdx.x; i < nv; i += 256) {
    int cv = i % vpr, r = i / vpr, t = r % Tn, a = r / Tn;
    const T* src = qkv + ((size_t)(b * A + a) * Tn + t) * ldq + h * dqk + cv * E;
    float q[E], k[E];
    ald<T>(src, q);
    ald<T>(src + Cq, k);
#pragma unroll
    for (int e = 0; e < E; e++) { qs[a * L + t * dqk + cv * E + e] = q[e]; ks[a * L + t * dqk + cv * E + e] = k[e]; }
  }
  __syncthreads();
  for (int e = wave; e < A * A; e += 4) {
    int a1 = e / A, a2 = e - a1 * A;
    float s = 0.f;
    for (int i = lane; i < L; i += 64) s += qs[a1 * L + i] * ks[a2 * L + i];
    s = wave_sum(s);
    if (lane == 0) Ps[e] = s * scale;
  }
  __syncthreads();
  if (threadIdx.x < A) {
    float* p = Ps + threadIdx.x * A;
    float mx = p[0];
    for (int j = 1; j < A; j++) mx = fmaxf(mx, p[j]);
    float sum = 0.f;
    for (int j = 0; j < A; j++) { float ex = __expf(p[j] - mx); p[j] = ex; sum += ex; }
    float inv = 1.f / sum;
    for (int j = 0; j < A; j++) p[j] *= inv;
  }
  __syncthreads();
}

// AM: compile-time bound on the number of agents (4 or 8): the per-agent register arrays are sized by it, and the
// backward kernel needs 219 VGPRs at AM = 8 (two workgroups per CU) against half of that at AM = 4
template <typename T, int AM>
__global__ __launch_bounds__(256) void attn_fwd_v2_kernel(const T* __restrict__ qkv, int ldq, const T* __restrict__ x, int ldx,
                                                          T* __restrict__ out, int ldo, const float* __restrict__ gamma_p,
                                                          int B, int A, int Tn, int H, int Cq, int C, float scale, int residual) {
  constexpr int E = AVec<T>::E;
  extern __shared__ float sm[];
  const int dqk = Cq / H, dv = C / H, L = Tn * dqk;
  float* qs = sm;
  float* ks = sm + A * L;
  float* Ps = ks + A * L;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const float gamma = residual ? gamma_p[0] : 1.f;
  att2_scores<T>(qkv, ldq, b, h, A, Tn, dqk, Cq, scale, qs, ks, Ps);
  const int vpr = dv / E, nv = Tn * vpr;
  for (int i = threadIdx.x; i < nv; i += 256) {
    int cv = i % vpr, t = i / vpr;
    const int c = h * dv + cv * E;
    float v[AM][E];
#pragma unroll
    for (int a2 = 0; a2 < AM; a2++)
      if (a2 < A) ald<T>(qkv + ((size_t)(b * A + a2) * Tn + t) * ldq + 2 * Cq + c, v[a2]);
#pragma unroll
    for (int a1 = 0; a1 < AM; a1++) {
      if (a1 >= A) break;
      float o[E];
#pragma unroll
      for (int e = 0; e < E; e++) o[e] = 0.f;
#pragma unroll
      for (int a2 = 0; a2 < AM; a2++)
        if (a2 < A) {
          float p = Ps[a1 * A + a2];
#pragma unroll
          for (int e = 0; e < E; e++) o[e] += p * v[a2][e];
        }
      size_t row = (size_t)(b * A + a1) * Tn + t;
      if (residual) {
        float xv[E];
        ald<T>(x + row * ldx + c, xv);
#pragma unroll
        for (int e = 0; e < E; e++) o[e] = xv[e] + gamma * o[e];
      }
      ast<T>(out + row * ldo + c, o);
    }
  }
}

template <typename T, int AM>
__global__ __launch_bounds__(256) void attn_bwd_v2_kernel(const T* __restrict__ qkv, int ldq, const T* __restrict__ dout, int ldd,
                                                          T* __restrict__ dqkv, int ldg, const float* __restrict__ gamma_p,
                                                          float* __restrict__ dgamma, int B, int A, int Tn, int H, int Cq, int C,
                                                          float scale, int residual, float* __restrict__ drb, int ldrb) {
  constexpr int E = AVec<T>::E;
  extern __shared__ float sm[];
  __shared__ float Dw[4][AM * AM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dqk = Cq / H, dv = C / H, L = Tn * dqk;
  float* qs = sm;
  float* ks = sm + A * L;
  float* Ps = ks + A * L;
  float* Ds = Ps + A * A;
  // drb != null: also the gradient of the per-sequence row bias of the fused projection (the time-MLP outputs added to
  // q, k, v: temporal.py:100-111), d bias[(b,a), col] = sum_t dqkv[(b,a,t), col] -- this workgroup owns head h's columns
  // of its A rows, so the sums are built in LDS ([A][q | k | v of the head]) and stored without atomics in global memory.
  // The sums never touch dqkv again: sum_t dq[a] = sum_o dS[a][o] (sum_t k[o]) and likewise for dk, from the slices already
  // in LDS; sum_t dv[a2] = gamma * sum_a1 P[a1][a2] (sum_t dO[a1]) from a second (cache-hot) read of dO.
  float* Rs = Ds + A * A;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const float gamma = residual ? gamma_p[0] : 1.f;
  att2_scores<T>(qkv, ldq, b, h, A, Tn, dqk, Cq, scale, qs, ks, Ps);
  // pass over the value slice: raw dP[a1][a2] = sum dO[a1].v[a2] (per-thread partials) and dv = gamma * P^T dO
  float dp[AM][AM];
#pragma unroll
  for (int i = 0; i < AM; i++)
#pragma unroll
    for (int j = 0; j < AM; j++) dp[i][j] = 0.f;
  const int vprv = dv / E, nvv = Tn * vprv;
  for (int i = threadIdx.x; i < nvv; i += 256) {
    int cv = i % vprv, t = i / vprv;
    const int c = h * dv + cv * E;
    float v[AM][E], g[AM][E];
#pragma unroll
    for (int a = 0; a < AM; a++)
      if (a < A) {
        size_t row = (size_t)(b * A + a) * Tn + t;
        ald<T>(qkv + row * ldq + 2 * Cq + c, v[a]);
        ald<T>(dout + row * ldd + c, g[a]);
      }
#pragma unroll
    for (int a1 = 0; a1 < AM; a1++)
#pragma unroll
      for (int a2 = 0; a2 < AM; a2++)
        if (a1 < A && a2 < A) {
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < E; e++) s += g[a1][e] * v[a2][e];
          dp[a1][a2] += s;
        }
#pragma unroll
    for (int a2 = 0; a2 < AM; a2++) {
      if (a2 >= A) break;
      float o[E];
#pragma unroll
      for (int e = 0; e < E; e++) o[e] = 0.f;
#pragma unroll
      for (int a1 = 0; a1 < AM; a1++)
        if (a1 < A) {
          float p = Ps[a1 * A + a2];
#pragma unroll
          for (int e = 0; e < E; e++) o[e] += p * g[a1][e];
        }
#pragma unroll
      for (int e = 0; e < E; e++) o[e] *= gamma;
      ast<T>(dqkv + ((size_t)(b * A + a2) * Tn + t) * ldg + 2 * Cq + c, o);
    }
  }
#pragma unroll
  for (int a1 = 0; a1 < AM; a1++)
#pragma unroll
    for (int a2 = 0; a2 < AM; a2++)
      if (a1 < A && a2 < A) {
        float s = wave_sum(dp[a1][a2]);
        if (lane == 0) Dw[wave][a1 * A + a2] = s;
      }
  __syncthreads();
  if (threadIdx.x < A) {       // row a1: dgamma partial and softmax backward -> dS (already times scale)
    const int a1 = threadIdx.x;
    float d[AM], dot = 0.f;
    for (int j = 0; j < A; j++) {
      d[j] = Dw[0][a1 * A + j] + Dw[1][a1 * A + j] + Dw[2][a1 * A + j] + Dw[3][a1 * A + j];
      dot += Ps[a1 * A + j] * d[j];
    }
    for (int j = 0; j < A; j++) Ds[a1 * A + j] = Ps[a1 * A + j] * gamma * (d[j] - dot) * scale;
    Dw[0][a1] = dot;
  }
  __syncthreads();
  if (threadIdx.x == 0 && residual) {
    float dg = 0.f;
    for (int a1 = 0; a1 < A; a1++) dg += Dw[0][a1];
    atomicAdd(dgamma, dg);
  }
  // dq[a1] = sum_a2 dS[a1][a2] k[a2];  dk[a2] = sum_a1 dS[a1][a2] q[a1]   (q, k from LDS)
  const int vpr = dqk / E, nv = A * Tn * vpr;
  for (int i = threadIdx.x; i < nv; i += 256) {
    int cv = i % vpr, r = i / vpr, t = r % Tn, a = r / Tn;
    float oq[E], ok[E];
#pragma unroll
    for (int e = 0; e < E; e++) { oq[e] = 0.f; ok[e] = 0.f; }
    for (int o = 0; o < A; o++) {
      float sq = Ds[a * A + o], sk = Ds[o * A + a];
      const float* kk = ks + o * L + t * dqk + cv * E;
      const float* qq = qs + o * L + t * dqk + cv * E;
#pragma unroll
      for (int e = 0; e < E; e++) { oq[e] += sq * kk[e]; ok[e] += sk * qq[e]; }
    }
    T* dst = dqkv + ((size_t)(b * A + a) * Tn + t) * ldg + h * dqk + cv * E;
    ast<T>(dst, oq);
    ast<T>(dst + Cq, ok);
  }
  if (drb) {
    for (int i = threadIdx.x; i < A * dqk; i += 256) {
      const int a = i / dqk, c = i - a * dqk;
      float sq = 0.f, sk = 0.f;
      for (int o = 0; o < A; o++) {
        float ksum = 0.f, qsum = 0.f;
        for (int t = 0; t < Tn; t++) { ksum += ks[o * L + t * dqk + c]; qsum += qs[o * L + t * dqk + c]; }
        sq += Ds[a * A + o] * ksum;
        sk += Ds[o * A + a] * qsum;
      }
      float* dst = drb + (size_t)(b * A + a) * ldrb + h * dqk + c;
      dst[0] = sq;
      dst[Cq] = sk;
    }
    for (int i = threadIdx.x; i < A * dv; i += 256) {
      const int a1 = i / dv, c = i - a1 * dv;
      const T* src = dout + (size_t)(b * A + a1) * Tn * ldd + h * dv + c;
      float g = 0.f;
      for (int t = 0; t < Tn; t++) g += (float)src[(size_t)t * ldd];
      Rs[i] = g;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < A * dv; i += 256) {
      const int a2 = i / dv, c = i - a2 * dv;
      float v = 0.f;
      for (int a1 = 0; a1 < A; a1++) v += Ps[a1 * A + a2] * Rs[a1 * dv + c];
      drb[(size_t)(b * A + a2) * ldrb + 2 * Cq + h * dv + c] = gamma * v;
    }
  }
}

static size_t att2_lds(int A, int T, int H, int Cq, int C = 0) {   // C > 0: + the row-bias gradient sums of the backward kernel
  return ((size_t)2 * A * T * (Cq / H) + 2 * A * A + (C > 0 ? (size_t)A * (C / H) : 0)) * sizeof(float);
}

static int att2_ok(int dtype, int A, int T, int H, int Cq, int C, int ld1, int ld2, int ld3) {
  const int E = dtype == MADIFF_BF16 ? 8 : 4;
  return A <= ATT2_AMAX && (Cq / H) % E == 0 && (C / H) % E == 0 && Cq % E == 0 && att2_lds(A, T, H, Cq) <= 60 * 1024 &&
         ld1 % E == 0 && ld2 % E == 0 && ld3 % E == 0;
}

extern "C" __attribute__((visibility("hidden"))) int madiff_attn_fwd_v2(int dtype, const void* qkv, int ldq, const void* x, int ldx,
    void* out, int ldo, const float* gamma, int B, int A, int T, int H, int Cq, int C, float scale, int residual, hipStream_t stream) {
  if (!att2_ok(dtype, A, T, H, Cq, C, ldq, ldx, ldo)) return 1;
  size_t sh = att2_lds(A, T, H, Cq);
#define ATT2_FWD(T_, AM_)                                                                                                  \
  hipLaunchKernelGGL((attn_fwd_v2_kernel<T_, AM_>), dim3(B * H), dim3(256), sh, stream, (const T_*)qkv, ldq, (const T_*)x, ldx, \
                     (T_*)out, ldo, gamma, B, A, T, H, Cq, C, scale, residual)
  if (dtype == MADIFF_BF16) { if (A <= 4) ATT2_FWD(bf16_t, 4); else ATT2_FWD(bf16_t, 8); }
  else { if (A <= 4) ATT2_FWD(float, 4); else ATT2_FWD(float, 8); }
#undef ATT2_FWD
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" __attribute__((visibility("hidden"))) int madiff_attn_bwd_v2(int dtype, const void* qkv, int ldq, const void* dout, int ldd,
    void* dqkv, int ldg, const float* gamma, float* dgamma, int B, int A, int T, int H, int Cq, int C, float scale, int residual,
    float* drb, int ldrb, hipStream_t stream) {
  if (!att2_ok(dtype, A, T, H, Cq, C, ldq, ldd, ldg)) return 1;
  size_t sh = att2_lds(A, T, H, Cq, drb ? C : 0);
  if (sh > 60 * 1024) return 1;
#define ATT2_BWD(T_, AM_)                                                                                                  \
  hipLaunchKernelGGL((attn_bwd_v2_kernel<T_, AM_>), dim3(B * H), dim3(256), sh, stream, (const T_*)qkv, ldq, (const T_*)dout, \
                     ldd, (T_*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual, drb, ldrb)
  if (dtype == MADIFF_BF16) { if (A <= 4) ATT2_BWD(bf16_t, 4); else ATT2_BWD(bf16_t, 8); }
  else { if (A <= 4) ATT2_BWD(float, 4); else ATT2_BWD(float, 8); }
#undef ATT2_BWD
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ================================================================================================
// v3 (bf16 build): the contractions over the (d, t) feature axis on MFMA.
//
// Reference: temporal.py:104-122.  Per (batch element, head) the layer is two kinds of products: the A x A score-shaped
// ones that contract over the FEATURES of a head (S = Q K^T over T*d_qk features, and in backward dP = dO V^T over T*d_v),
// and the mixes of A rows with A x A weights (P V, P^T dO, dS K, dS^T Q: K dimension = A <= 8).  The first kind is where the
// MACs per loaded byte are and it maps onto v_mfma_f32_16x16x32_bf16 with no LDS staging at all: 16 / AP batch elements
// (AP = A padded to 4 or 8) are packed into the 16 rows of one MFMA tile, packed row rho = (b', a); a lane's A- and
// B-operand fragment is 8 consecutive features of ITS packed row, i.e. one 16-byte global load each, straight into the
// MFMA operand registers.  The 16 x 16 result holds the A x A blocks of the packed batch elements on its diagonal (the
// off-diagonal cross-batch products are ignored: with A = 4 that is 4x redundant MFMA work on a kernel that is bound by
// its memory traffic).  The K-steps of a head are dealt out over the 8 waves of the workgroup and the partial tiles meet
// in LDS.  The second kind (K = A) stays on the VALU: an MFMA would multiply zeros for >= half of its K.
//
// Against v2 (one workgroup per (b, h), q/k staged in LDS as fp32, scores by wave_sum loops) this form has every global
// load of a phase in flight at once and 4 (A <= 4) or 2 (A <= 8) batch elements per workgroup.  The row-bias gradient
// (sum over t of dqkv, temporal.py:100-111's time conditioning) is formed from sums over t of the INPUT rows, mixed with
// P / dS at the end (LDS float atomics on the outputs were tried first: 46 us of a 64 us launch).
// Needs d_qk % 32 == 0 and d_v % 32 == 0 (dim 128: 64 / 256 and 32 / 128); other shapes and the fp32 build use v2.
// ================================================================================================
template <int AP>
__device__ __forceinline__ void att3_softmax_row(const float* s, int A, float* p) {
  float mx = s[0];
#pragma unroll
  for (int j = 1; j < AP; j++) if (j < A) mx = fmaxf(mx, s[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < AP; j++) if (j < A) { p[j] = __expf(s[j] - mx); sum += p[j]; }
  const float inv = 1.f / sum;
#pragma unroll
  for (int j = 0; j < AP; j++) p[j] = j < A ? p[j] * inv : 0.f;
}

// acc += X[rho1] . Y[rho2] over the wave's share of a head slice that is `dh` features wide (dh % 32 == 0).  The order of the
// contraction index is free as long as both operands use the same one, so a 64-byte K-step is NOT "lane group fq takes
// bytes [16 fq, 16 fq + 16)": the slice of one time step is cut into 128-byte units (64 features = two MFMAs; a last half
// unit when dh % 64 == 32), lane group fq loads the 32 contiguous bytes [32 fq, 32 fq + 32) of a unit and feeds its two
// halves to the two MFMAs -- a wave then reads whole 128-byte lines of 16 rows per unit.  Units are dealt out to the waves
// in contiguous runs.  xrow / yrow = this lane's packed row (t = 0) at the start of the head slice.
template <int NW>
__device__ __forceinline__ f32x4 att3_rowdots(const bf16_t* __restrict__ xrow, int ldx, const bf16_t* __restrict__ yrow, int ldy,
                                              int Tn, int dh, int wave, int fq) {
  const int upt = (dh + 63) >> 6, nu = Tn * upt;            // units per time step, units in all
  const bool half_tail = (dh & 63) != 0;                    // the last unit of a time step holds 32 features only
  const int per = (nu + NW - 1) / NW, u_lo = wave * per, u_hi = min(nu, u_lo + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int u0 = u_lo; u0 < u_hi; u0 += 2) {                 // two units per round: eight 16-byte loads in flight per lane
    bf16x8 a[2][2], b[2][2];
    bool two[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int u = min(u0 + k, u_hi - 1);
      const int t = u / upt, j = u - t * upt;
      two[k] = !(half_tail && j == upt - 1);
      // full unit: bytes [32 fq, 32 fq + 32) of the 128-byte unit; half unit (64 bytes): [16 fq, 16 fq + 16) only
      const int off = j * 64 + (two[k] ? fq * 16 : fq * 8);
      const bf16_t* px = xrow + (size_t)t * ldx + off;
      const bf16_t* py = yrow + (size_t)t * ldy + off;
      a[k][0] = *(const bf16x8*)px;
      b[k][0] = *(const bf16x8*)py;
      if (two[k]) { a[k][1] = *(const bf16x8*)(px + 8); b[k][1] = *(const bf16x8*)(py + 8); }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      if (u0 + k >= u_hi) continue;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k][0], b[k][0], acc, 0, 0, 0);
      if (two[k]) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k][1], b[k][1], acc, 0, 0, 0);
    }
  }
  return acc;
}

template <int AP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, int ldq,
                                                                const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ out,
                                                                int ldo, const float* __restrict__ gamma_p, int B, int A, int Tn,
                                                                int H, int Cq, int C, float scale, int residual) {
  constexpr int G = 16 / AP, NT = NW * 64;
  __shared__ float Sp[NW][256];
  __shared__ float Ps[16 * AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int dqk = Cq / H, dv = C / H;
  const int bg = blockIdx.x / H, h = blockIdx.x - bg * H, b0 = bg * G;
  const float gamma = residual ? gamma_p[0] : 1.f;
  // ---- first vector of the P.V pass: its V rows are requested before the score phase (nothing below depends on them)
  const int vpr = dv >> 3, nv = G * Tn * vpr;
  bf16x8 vfirst[AP];
  {
    const int cv = tid % vpr, r = tid / vpr, t = r % Tn, pb = r / Tn;
    if (tid < nv && b0 + pb < B) {
#pragma unroll
      for (int a2 = 0; a2 < AP; a2++)
        if (a2 < A) vfirst[a2] = *(const bf16x8*)(qkv + ((size_t)((b0 + pb) * A + a2) * Tn + t) * ldq + 2 * Cq + h * dv + cv * 8);
    }
  }
  // ---- scores on MFMA: packed row fr = (b', a); rows beyond the problem read a valid row and are never used
  {
    const int pb = fr / AP, pa = fr - pb * AP;
    const int rb = (pa < A && b0 + pb < B) ? (b0 + pb) * A + pa : b0 * A;
    const bf16_t* qrow = qkv + (size_t)rb * Tn * ldq + h * dqk;
    const f32x4 acc = att3_rowdots<NW>(qrow, ldq, qrow + Cq, ldq, Tn, dqk, wave, fq);
#pragma unroll
    for (int r = 0; r < 4; r++) Sp[wave][(fq * 4 + r) * 16 + fr] = acc[r];
  }
  __syncthreads();
  if (tid < 16) {
    const int pb = tid / AP, a1 = tid - pb * AP;
    float s[AP], p[AP];
#pragma unroll
    for (int j = 0; j < AP; j++) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) v += Sp[w][tid * 16 + pb * AP + j];
      s[j] = v * scale;
    }
    att3_softmax_row<AP>(s, A, p);
#pragma unroll
    for (int j = 0; j < AP; j++) Ps[tid * AP + j] = a1 < A ? p[j] : 0.f;
  }
  __syncthreads();
  // ---- out[a1] = x[a1] + gamma * sum_a2 P[a1][a2] v[a2] on 8-feature vectors
  for (int i = tid; i < nv; i += NT) {
    const int cv = i % vpr, r = i / vpr, t = r % Tn, pb = r / Tn;
    const int b = b0 + pb;
    if (b >= B) continue;
    const int c = h * dv + cv * 8;
    bf16x8 v[AP], xv[AP];
#pragma unroll
    for (int a = 0; a < AP; a++)
      if (a < A) {
        const size_t row = (size_t)(b * A + a) * Tn + t;
        v[a] = i == tid ? vfirst[a] : *(const bf16x8*)(qkv + row * ldq + 2 * Cq + c);
        if (residual) xv[a] = *(const bf16x8*)(x + row * ldx + c);
      }
#pragma unroll
    for (int a1 = 0; a1 < AP; a1++) {
      if (a1 >= A) continue;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = 0.f;
#pragma unroll
      for (int a2 = 0; a2 < AP; a2++)
        if (a2 < A) {
          const float p = Ps[(pb * AP + a1) * AP + a2];
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] += p * (float)v[a2][e];
        }
      bf16x8 ov;
#pragma unroll
      for (int e = 0; e < 8; e++) ov[e] = (bf16_t)(residual ? (float)xv[a1][e] + gamma * o[e] : o[e]);
      *(bf16x8*)(out + ((size_t)(b * A + a1) * Tn + t) * ldo + c) = ov;
    }
  }
}

template <int AP, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, int ldq,
                                                                const bf16_t* __restrict__ dout, int ldd,
                                                                bf16_t* __restrict__ dqkv, int ldg, const float* __restrict__ gamma_p,
                                                                float* __restrict__ dgamma, int B, int A, int Tn, int H, int Cq,
                                                                int C, float scale, int residual, float* __restrict__ drb, int ldrb) {
  constexpr int G = 16 / AP, NT = NW * 64;
  __shared__ float Sp[NW][256];
  __shared__ float Dp[NW][256];
  __shared__ float Ps[16 * AP];
  __shared__ float Ds[16 * AP];
  extern __shared__ float Rs[];            // [16][W] sums over t of the dqkv tile (only with drb)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int dqk = Cq / H, dv = C / H, W = 2 * dqk + dv;
  const int bg = blockIdx.x / H, h = blockIdx.x - bg * H, b0 = bg * G;
  const float gamma = residual ? gamma_p[0] : 1.f;
  // Row-bias gradient (drb[(b,a), col] = sum_t dqkv[(b,a,t), col]) without touching dqkv again and without atomics: the
  // sums over t commute with the K = A mixes, sum_t dq[a1] = sum_a2 dS[a1][a2] (sum_t k[a2]) and likewise for dk and dv,
  // so the workgroup first sums its rows of q, k and dO over t into Rs = [q sums | k sums | dO sums] per packed row (loads
  // that overlap the MFMA phase below; the second read of these rows comes from cache) and mixes the sums at the end.
  if (drb) {
    const int w8 = W >> 3;
    for (int i = tid; i < 16 * w8; i += NT) {
      const int rho = i / w8, c = (i - rho * w8) * 8;
      const int pb = rho / AP, pa = rho - pb * AP;
      float sum[8];
#pragma unroll
      for (int e = 0; e < 8; e++) sum[e] = 0.f;
      if (pa < A && b0 + pb < B) {
        const size_t row0 = (size_t)((b0 + pb) * A + pa) * Tn;
        const bf16_t* src;
        int ld;
        if (c < 2 * dqk) { src = qkv + row0 * ldq + (c < dqk ? h * dqk + c : Cq + h * dqk + (c - dqk)); ld = ldq; }
        else { src = dout + row0 * ldd + h * dv + (c - 2 * dqk); ld = ldd; }
        for (int t = 0; t < Tn; t++) {
          const bf16x8 v = *(const bf16x8*)(src + (size_t)t * ld);
#pragma unroll
          for (int e = 0; e < 8; e++) sum[e] += (float)v[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) Rs[rho * W + c + e] = sum[e];
    }
  }
  // ---- S = Q K^T and raw dP = dO V^T on MFMA (packed rows, operands straight from global memory)
  {
    const int pb = fr / AP, pa = fr - pb * AP;
    const int rb = (pa < A && b0 + pb < B) ? (b0 + pb) * A + pa : b0 * A;
    const bf16_t* qrow = qkv + (size_t)rb * Tn * ldq + h * dqk;
    const bf16_t* vrow = qkv + (size_t)rb * Tn * ldq + 2 * Cq + h * dv;
    const bf16_t* grow = dout + (size_t)rb * Tn * ldd + h * dv;
    const f32x4 accd = att3_rowdots<NW>(grow, ldd, vrow, ldq, Tn, dv, wave, fq);
    const f32x4 accs = att3_rowdots<NW>(qrow, ldq, qrow + Cq, ldq, Tn, dqk, wave, fq);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      Sp[wave][(fq * 4 + r) * 16 + fr] = accs[r];
      Dp[wave][(fq * 4 + r) * 16 + fr] = accd[r];
    }
  }
  __syncthreads();
  if (tid < 16) {       // packed row (b', a1): softmax, its backward -> dS (times scale), and the row's share of dgamma
    const int pb = tid / AP, a1 = tid - pb * AP;
    const bool valid = a1 < A && b0 + pb < B;
    float s[AP], p[AP], d[AP];
#pragma unroll
    for (int j = 0; j < AP; j++) {
      float v = 0.f, u = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) { v += Sp[w][tid * 16 + pb * AP + j]; u += Dp[w][tid * 16 + pb * AP + j]; }
      s[j] = v * scale;
      d[j] = u;
    }
    att3_softmax_row<AP>(s, A, p);
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < AP; j++) if (j < A) dot += p[j] * d[j];
#pragma unroll
    for (int j = 0; j < AP; j++) {
      Ps[tid * AP + j] = valid ? p[j] : 0.f;
      Ds[tid * AP + j] = (valid && j < A) ? p[j] * gamma * (d[j] - dot) * scale : 0.f;
    }
    if (!valid) dot = 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    if (tid == 0 && residual) atomicAdd(dgamma, dot);
  }
  __syncthreads();
  // ---- the K = A mixes on 8-feature vectors, one index space over both kinds so every thread has work:
  //   [0, nvv): dv[a2] = gamma * sum_a1 P[a1][a2] dO[a1]        [nvv, nvv + nvq): dq[a1] = sum_a2 dS[a1][a2] k[a2],
  //                                                                                dk[a2] = sum_a1 dS[a1][a2] q[a1]
  const int vprv = dv >> 3, nvv = G * Tn * vprv, vprq = dqk >> 3, nvq = G * Tn * vprq;
  for (int i = tid; i < nvv + nvq; i += NT) {
    if (i < nvv) {
      // (cv fastest, then the batch element, then t: the lanes of a wave then add into different Rs entries)
      const int cv = i % vprv, r = i / vprv, pb = r % G, t = r / G;
      const int b = b0 + pb;
      if (b >= B) continue;
      const int c = h * dv + cv * 8;
      bf16x8 g[AP];
#pragma unroll
      for (int a = 0; a < AP; a++)
        if (a < A) g[a] = *(const bf16x8*)(dout + ((size_t)(b * A + a) * Tn + t) * ldd + c);
#pragma unroll
      for (int a2 = 0; a2 < AP; a2++) {
        if (a2 >= A) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = 0.f;
#pragma unroll
        for (int a1 = 0; a1 < AP; a1++)
          if (a1 < A) {
            const float p = Ps[(pb * AP + a1) * AP + a2];
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] += p * (float)g[a1][e];
          }
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; e++) { o[e] *= gamma; ov[e] = (bf16_t)o[e]; }
        *(bf16x8*)(dqkv + ((size_t)(b * A + a2) * Tn + t) * ldg + 2 * Cq + c) = ov;
      }
    } else {
      const int j = i - nvv;
      const int cv = j % vprq, r = j / vprq, pb = r % G, t = r / G;
      const int b = b0 + pb;
      if (b >= B) continue;
      const int c = h * dqk + cv * 8;
      // two sub-passes (dq from the K rows, then dk from the Q rows): half the live registers of doing both at once
#pragma unroll
      for (int part = 0; part < 2; part++) {
        bf16x8 rw[AP];
#pragma unroll
        for (int a = 0; a < AP; a++)
          if (a < A) rw[a] = *(const bf16x8*)(qkv + ((size_t)(b * A + a) * Tn + t) * ldq + c + (part == 0 ? Cq : 0));
#pragma unroll
        for (int a = 0; a < AP; a++) {
          if (a >= A) continue;
          float ov[8];
#pragma unroll
          for (int e = 0; e < 8; e++) ov[e] = 0.f;
#pragma unroll
          for (int o = 0; o < AP; o++)
            if (o < A) {
              const float w = part == 0 ? Ds[(pb * AP + a) * AP + o] : Ds[(pb * AP + o) * AP + a];
#pragma unroll
              for (int e = 0; e < 8; e++) ov[e] += w * (float)rw[o][e];
            }
          bf16x8 vv;
#pragma unroll
          for (int e = 0; e < 8; e++) vv[e] = (bf16_t)ov[e];
          *(bf16x8*)(dqkv + ((size_t)(b * A + a) * Tn + t) * ldg + c + (part == 0 ? 0 : Cq)) = vv;
        }
      }
    }
  }
  if (drb) {      // (Rs was complete before the first barrier, Ps / Ds before the second)
    for (int i = tid; i < 16 * W; i += NT) {
      const int rho = i / W, c = i - rho * W;
      const int pb = rho / AP, a = rho - pb * AP;
      if (a >= A || b0 + pb >= B) continue;
      float v = 0.f;
      int col;
      if (c < dqk) {                      // sum_t dq[a] = sum_o dS[a][o] ksum[o]
        col = h * dqk + c;
#pragma unroll
        for (int o = 0; o < AP; o++) if (o < A) v += Ds[rho * AP + o] * Rs[(pb * AP + o) * W + dqk + c];
      } else if (c < 2 * dqk) {           // sum_t dk[a] = sum_o dS[o][a] qsum[o]
        const int cc = c - dqk;
        col = Cq + h * dqk + cc;
#pragma unroll
        for (int o = 0; o < AP; o++) if (o < A) v += Ds[(pb * AP + o) * AP + a] * Rs[(pb * AP + o) * W + cc];
      } else {                            // sum_t dv[a] = gamma sum_o P[o][a] gsum[o]
        const int cc = c - 2 * dqk;
        col = 2 * Cq + h * dv + cc;
#pragma unroll
        for (int o = 0; o < AP; o++) if (o < A) v += Ps[(pb * AP + o) * AP + a] * Rs[(pb * AP + o) * W + c];
        v *= gamma;
      }
      drb[(size_t)((b0 + pb) * A + a) * ldrb + col] = v;
    }
  }
}

static int att3_on() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MADIFF_ATTN_MFMA"); v = e ? (atoi(e) != 0) : 1; }
  return v;
}

// 1 when the MFMA form covers the layer (bf16, A <= 8, head slices a multiple of 32 features, 16-byte aligned rows)
extern "C" int madiff_attn_mfma_covers(int dtype, int A, int T, int H, int Cq, int C) {
  if (dtype != MADIFF_BF16 || !att3_on() || att_check(1, A, T, H, Cq, C)) return 0;
  return (A <= 8 && (Cq / H) % 32 == 0 && (C / H) % 32 == 0) ? 1 : 0;
}

static int att3_aligned(const void* p1, int ld1, const void* p2, int ld2, const void* p3, int ld3) {
  return ld1 % 8 == 0 && ld2 % 8 == 0 && ld3 % 8 == 0 && ((size_t)p1 % 16) == 0 && ((size_t)p2 % 16) == 0 && ((size_t)p3 % 16) == 0;
}

static int att3_fwd(const void* qkv, int ldq, const void* x, int ldx, void* out, int ldo, const float* gamma, int B, int A, int T,
                    int H, int Cq, int C, float scale, int residual, hipStream_t stream) {
  if (!madiff_attn_mfma_covers(MADIFF_BF16, A, T, H, Cq, C) || !att3_aligned(qkv, ldq, x, ldx, out, ldo) || Cq % 8) return 1;
  const int AP = A <= 4 ? 4 : 8, G = 16 / AP;
  const dim3 grid(((B + G - 1) / G) * H);
  if (AP == 4)
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<4, 8>), grid, dim3(512), 0, stream, (const bf16_t*)qkv, ldq, (const bf16_t*)x, ldx,
                       (bf16_t*)out, ldo, gamma, B, A, T, H, Cq, C, scale, residual);
  else
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<8, 8>), grid, dim3(512), 0, stream, (const bf16_t*)qkv, ldq, (const bf16_t*)x, ldx,
                       (bf16_t*)out, ldo, gamma, B, A, T, H, Cq, C, scale, residual);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

static int att3_bwd(const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg, const float* gamma, float* dgamma,
                    int B, int A, int T, int H, int Cq, int C, float scale, int residual, float* drb, int ldrb,
                    hipStream_t stream) {
  if (!madiff_attn_mfma_covers(MADIFF_BF16, A, T, H, Cq, C) || !att3_aligned(qkv, ldq, dout, ldd, dqkv, ldg) || Cq % 8) return 1;
  const int AP = A <= 4 ? 4 : 8, G = 16 / AP;
  const size_t sh = drb ? (size_t)16 * (2 * (Cq / H) + C / H) * sizeof(float) : 0;
  if (sh > 40 * 1024) return 1;
  const dim3 grid(((B + G - 1) / G) * H);
  if (AP == 4)
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<4, 8>), grid, dim3(512), sh, stream, (const bf16_t*)qkv, ldq, (const bf16_t*)dout,
                       ldd, (bf16_t*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual, drb, ldrb);
  else
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<8, 8>), grid, dim3(512), sh, stream, (const bf16_t*)qkv, ldq, (const bf16_t*)dout,
                       ldd, (bf16_t*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual, drb, ldrb);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
