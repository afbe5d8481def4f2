// Cross-agent attention core (temporal.py:104-124 of the reference): tokens are the A agents of one batch
// element, features are the (d, t) pairs of a head.  The Q/K/V projections (+ time-MLP row bias) are done
// by gemm_fwd into one fused [M, 2*Cq + C] matrix; this kernel does scores -> softmax -> P.V -> x + gamma*out
// for one batch element per workgroup with the A x A scores living in LDS.  The contraction sizes here
// (A <= 16) are far below an MFMA tile, so this stage is bandwidth/latency-bound by design: it reads
// q, k, v, x once and writes out once.
#include "common.h"

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

static int att_check(int B, int A, int T, int H, int Cq, int C) {
  return (B > 0 && A > 0 && T > 0 && H > 0 && Cq > 0 && C > 0 && Cq % H == 0 && C % H == 0 &&
          H * A * A <= ATT_MAX_ENT) ? 0 : -1;
}

extern "C" int madiff_attn_fwd(int dtype, const void* qkv, int ldq, const void* x, int ldx, void* out, int ldo,
                               const float* gamma, int B, int A, int T, int H, int Cq, int C, int residual,
                               hipStream_t stream) {
  if (att_check(B, A, T, H, Cq, C) || ldq < 2 * Cq + C || ldx < C || ldo < C || (residual && !gamma)) return -1;
  float scale = 1.0f / sqrtf((float)(Cq / H) * (float)T);
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

// dgamma (1 float) must be zeroed by the caller; dqkv gets every column of the fused [q | k | v] gradient.
extern "C" int madiff_attn_bwd(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg,
                               const float* gamma, float* dgamma, int B, int A, int T, int H, int Cq, int C,
                               int residual, hipStream_t stream) {
  if (att_check(B, A, T, H, Cq, C) || ldq < 2 * Cq + C || ldd < C || ldg < 2 * Cq + C || (residual && (!gamma || !dgamma)))
    return -1;
  float scale = 1.0f / sqrtf((float)(Cq / H) * (float)T);
  int blocks = B < 4096 ? B : 4096;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((attn_bwd_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, (const bf16_t*)qkv, ldq,
                       (const bf16_t*)dout, ldd, (bf16_t*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual);
  else
    hipLaunchKernelGGL((attn_bwd_kernel<float>), dim3(blocks), dim3(256), 0, stream, (const float*)qkv, ldq,
                       (const float*)dout, ldd, (float*)dqkv, ldg, gamma, dgamma, B, A, T, H, Cq, C, scale, residual);
  MADIFF_CHECK_LAUNCH();
  return 0;
}
