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

// mish(x) = x * tanh(softplus(x)); written via e = exp(-|x|) so it never overflows.
// tanh(log(1+e^x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1)
__device__ __forceinline__ float mish_f(float x) {
  float e = __expf(-fabsf(x));
  float num, den;
  if (x >= 0.f) {            // u = e^x = 1/e : ((1+u)^2-1)/((1+u)^2+1) = (1+2e)/(1+2e+2e^2)
    num = 1.f + 2.f * e;
    den = num + 2.f * e * e;
  } else {                   // u = e : (u^2+2u)/(u^2+2u+2)
    num = e * e + 2.f * e;
    den = num + 2.f;
  }
  return x * (num / den);
}
// d/dx mish(x) = tanh(sp) + x * sigmoid(x) * (1 - tanh(sp)^2)
__device__ __forceinline__ float mish_grad_f(float x) {
  float e = __expf(-fabsf(x));
  float num, den, sig;
  if (x >= 0.f) {
    num = 1.f + 2.f * e;
    den = num + 2.f * e * e;
    sig = 1.f / (1.f + e);
  } else {
    num = e * e + 2.f * e;
    den = num + 2.f;
    sig = e / (1.f + e);
  }
  float th = num / den;
  return th + x * sig * (1.f - th * th);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
