// Shared device helpers for the madiff_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define MADIFF_F32 0
#define MADIFF_BF16 1

#define MADIFF_CHECK_LAUNCH()                      \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// mish(x) = x * tanh(softplus(x)).  With t = e^x:  tanh(log(1 + t)) = ((1+t)^2 - 1) / ((1+t)^2 + 1) = p / (p + 2),
// p = t (t + 2).  x is clamped at 20 before the exponential (t^2 stays far below fp32 overflow; beyond 20 the ratio is
// 1 to fp32 precision anyway), and for very negative x the ratio degrades gracefully to t.  One v_exp_f32, one
// v_rcp_f32 (1 ulp) and a handful of FMAs: the IEEE division this replaces cost more than everything else together.
__device__ __forceinline__ float mish_f(float x) {
  const float t = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896341f);
  const float p = t * (t + 2.f);
  return x * p * __builtin_amdgcn_rcpf(p + 2.f);
}
// d/dx mish(x) = th + x * sigmoid(x) * (1 - th^2),  th = p / (p + 2),  1 - th = 2 / (p + 2),  sigmoid = t / (1 + t)
__device__ __forceinline__ float mish_grad_f(float x) {
  const float t = __builtin_amdgcn_exp2f(fminf(x, 20.f) * 1.44269504088896341f);
  const float p = t * (t + 2.f);
  const float r = __builtin_amdgcn_rcpf(p + 2.f);
  const float th = p * r;
  const float sig = t * __builtin_amdgcn_rcpf(1.f + t);
  return th + x * sig * (2.f * r) * (1.f + th);
}

// Workgroups are dispatched round-robin over the 8 XCDs (id % 8), each with its own L2.  With the natural (x, y) -> tile map
// every XCD then touches all tiles of one operand; this remap hands each XCD a compact (gx/4) x (gy/2) block of tiles
// instead, so what its L2 fetches from HBM is a quarter of one operand plus half of the other.  Identity when the grid does
// not divide (gx % 4, gy % 2); MADIFF_XCD_REMAP=0 switches it off through the `on` argument.
__device__ __forceinline__ void xcd_remap(int& bx, int& by, int on) {
  // on = 1: columns fastest inside the XCD's block (round 3: the column tiles of a row tile run back to back on one XCD,
  // so the activation tile is fetched once per XCD; +3 % on the sampler, where a CU runs six tiles per launch);
  // on = 2: rows fastest (round 2's order, kept for A/B: MADIFF_XCD_REMAP=2)
  const int gx = gridDim.x, gy = gridDim.y;
  if (!on || (gx & 3) || (gy & 1)) return;
  const int id = bx + gx * by;
  const int xcd = id & 7, local = id >> 3;
  const int bw = gx >> 2, bh = gy >> 1;
  if (on == 2) {
    bx = (xcd & 3) * bw + local % bw;
    by = (xcd >> 2) * bh + local / bw;
  } else {
    bx = (xcd & 3) * bw + local / bh;
    by = (xcd >> 2) * bh + local % bh;
  }
}

// Any grid: workgroups reach the 8 XCDs round-robin in linear order (x fastest), so XCD c runs linear ids c, c + 8, ...
// Give XCD c a CONTIGUOUS run of the tile order instead (a bijection for every grid size): mode 1 = x fastest inside the
// run (all row tiles of a column tile meet in one L2: the weight tile is fetched once), mode 2 = y fastest (all column
// tiles of a row tile meet in one L2: the activation tile is fetched once).
__device__ __forceinline__ void xcd_chunk_remap(int& bx, int& by, int mode) {
  if (!mode) return;
  const int gx = gridDim.x, gy = gridDim.y, n = gx * gy;
  const int id = bx + gx * by;
  const int xcd = id & 7, local = id >> 3;
  const int q = n >> 3, r = n & 7;
  const int L = xcd * q + (xcd < r ? xcd : r) + local;
  if (mode == 1) { bx = L % gx; by = L / gx; }
  else { by = L % gy; bx = L / gy; }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
