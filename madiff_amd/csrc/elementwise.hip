// Bandwidth-bound kernels of the diffusion wrapper: layout pack/unpack (+ fused add_noise + inpainting),
// the fused DDPM/DDIM posterior step (guidance blend + x0-predict + clip + posterior mean + noise +
// inpainting), the masked weighted MSE (forward sums and backward), small activations and reductions.
// Trajectory tensors keep the reference layout [B, T, A, F] in fp32; the denoiser's internal layout is
// [(b a), t, Fpad] in the build's storage type.
#include "common.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

// ---------------------------------------------------------------- pack: [B,T,A,F] f32 -> [(b a) t Fpad]
// optional add_noise (diffusers DDPMScheduler.add_noise, called at diffusion.py:292) and inpainting
// (helpers.py:194-196) fused in.
template <typename T>
__global__ void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                  const long* __restrict__ tsteps, const float* __restrict__ sqrt_ac,
                                  const float* __restrict__ sqrt_1mac, const float* __restrict__ cond,
                                  const uint8_t* __restrict__ mask, T* __restrict__ out, float* __restrict__ x_noisy,
                                  int B, int Tn, int A, int F, int Fpad, int x_ld) {
  const long total = (long)B * A * Tn * Fpad;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int f = e % Fpad;
    long r = e / Fpad;
    int t = r % Tn;
    long ba = r / Tn;
    int a = ba % A;
    int b = ba / A;
    float v = 0.f;
    if (f < F) {
      long src = (((long)b * Tn + t) * A + a) * F + f;
      v = x[(((long)b * Tn + t) * A + a) * x_ld + f];        // x may be a column slice of a wider [B,T,A,x_ld] tensor
      if (noise) {
        long ts = tsteps[b];
        v = sqrt_ac[ts] * v + sqrt_1mac[ts] * noise[src];
      }
      if (mask && mask[src]) v = cond[src];
      if (x_noisy) x_noisy[src] = v;
    }
    out[e] = from_f32<T>(v);
  }
}

// unpack: [(b a) t ld] -> [B,T,A,F] f32
template <typename T>
__global__ void unpack_output_kernel(const T* __restrict__ in, int ld, float* __restrict__ out, int B, int Tn, int A,
                                     int F) {
  const long total = (long)B * Tn * A * F;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int f = e % F;
    long r = e / F;
    int a = r % A;
    long bt = r / A;
    int t = bt % Tn;
    int b = bt / Tn;
    out[e] = to_f32(in[(((long)b * A + a) * Tn + t) * ld + f]);
  }
}

// ---------------------------------------------------------------- posterior step
// coef row i: {sqrt(1-abar_t), 1/sqrt(abar_t), c_x0, c_xt, c_eps, sigma}.  DDPM: c_eps = 0; DDIM (eta=0):
// c_xt = 0, c_eps = sqrt(1-abar_prev), sigma = 0.  x0 is always clipped to [-1, 1] (clip_sample=True).
__global__ void posterior_step_kernel(const float* __restrict__ x, const float* __restrict__ eps_c,
                                      const float* __restrict__ eps_u, float w, const float* __restrict__ z,
                                      const float* __restrict__ cond, const uint8_t* __restrict__ mask,
                                      const float* __restrict__ coef, const int* __restrict__ step_ptr, int step_val,
                                      float* __restrict__ out, long total) {
  const int i = step_ptr ? step_ptr[0] : step_val;
  const float sb = coef[i * 6 + 0], isa = coef[i * 6 + 1], c0 = coef[i * 6 + 2], ct = coef[i * 6 + 3],
              ce = coef[i * 6 + 4], sg = coef[i * 6 + 5];
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    float xv = x[e];
    float ec = eps_c[e];
    float ep = eps_u ? (eps_u[e] + w * (ec - eps_u[e])) : ec;
    float x0 = (xv - sb * ep) * isa;
    x0 = fminf(fmaxf(x0, -1.f), 1.f);
    float v = c0 * x0 + ct * xv + ce * ep;
    if (sg != 0.f) v += sg * z[e];
    if (mask && mask[e]) v = cond[e];   // inpainting of the NEXT iteration / final enforcement
    out[e] = v;
  }
}

// ---- the same step with the noise produced ON THE DEVICE (or read from a per-step table), for the whole-loop graph:
// nothing between two denoising steps needs the host.  z_all != null: explicit draws, row `i` of a [n_steps, total] table
// (parity runs: host-supplied noise).  z_all == null: z ~ N(0, 1) from Philox4x32-10 (Salmon et al., SC'11; the generator
// torch / cuRAND / rocRAND use) keyed by rng[0] (64-bit seed), counter = (element index / 4, step index i, rng[1]) -- one
// Philox block yields four 32-bit words = two Box-Muller pairs = the draws of four consecutive elements, so every
// (seed, step, element) has its own draw whatever the launch geometry.  tests/test_gpu_kernels.py restates the generator
// in numpy and compares word for word.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&r)[4]) {
#pragma unroll
  for (int round = 0; round < 10; round++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1,
                   n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// two standard normals from two 32-bit words: u1 in (0, 1], u2 in [0, 1)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  const float rad = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.28318530717958647692f * u2, &sn, &cs);
  n0 = rad * cs;
  n1 = rad * sn;
}

__global__ void posterior_step_rng_kernel(const float* __restrict__ x, const float* __restrict__ eps_c,
                                          const float* __restrict__ eps_u, float w, const float* __restrict__ z_all,
                                          const unsigned long long* __restrict__ rng, const float* __restrict__ cond,
                                          const uint8_t* __restrict__ mask, const float* __restrict__ coef,
                                          const int* __restrict__ step_ptr, int step_val, float* __restrict__ out, long total) {
  const int i = step_ptr ? step_ptr[0] : step_val;
  const float sb = coef[i * 6 + 0], isa = coef[i * 6 + 1], c0 = coef[i * 6 + 2], ct = coef[i * 6 + 3],
              ce = coef[i * 6 + 4], sg = coef[i * 6 + 5];
  const float* z = z_all ? z_all + (size_t)i * total : nullptr;
  const uint32_t k0 = rng ? (uint32_t)rng[0] : 0u, k1 = rng ? (uint32_t)(rng[0] >> 32) : 0u;
  const uint32_t c3 = rng ? (uint32_t)rng[1] : 0u;
  const long quads = (total + 3) >> 2;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < quads; q += (long)gridDim.x * blockDim.x) {
    float zn[4] = {0.f, 0.f, 0.f, 0.f};
    if (sg != 0.f && !z) {
      uint32_t r[4];
      philox4x32_10((uint32_t)q, (uint32_t)((unsigned long long)q >> 32), (uint32_t)i, c3, k0, k1, r);
      box_muller(r[0], r[1], zn[0], zn[1]);
      box_muller(r[2], r[3], zn[2], zn[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const long e = q * 4 + j;
      if (e >= total) break;
      const float xv = x[e], ec = eps_c[e];
      const float ep = eps_u ? (eps_u[e] + w * (ec - eps_u[e])) : ec;
      float x0 = (xv - sb * ep) * isa;
      x0 = fminf(fmaxf(x0, -1.f), 1.f);
      float v = c0 * x0 + ct * xv + ce * ep;
      if (sg != 0.f) v += sg * (z ? z[e] : zn[j]);
      if (mask && mask[e]) v = cond[e];
      out[e] = v;
    }
  }
}

__global__ void inpaint_kernel(float* __restrict__ x, const float* __restrict__ cond, const uint8_t* __restrict__ mask,
                               long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    if (mask[e]) x[e] = cond[e];
}

// SMAC data encoders (utils/data_encoder.py:11-20: `data[..., 1:, :, :k] = data[..., 0:1, :, :k]`): the first k features of
// every later time step take the value of time step 0.  In place: the entries read (t = 0) are never written.
__global__ void encode_first_k_kernel(float* __restrict__ x, int B, int Tn, int A, int F, int k) {
  const long total = (long)B * (Tn - 1) * A * k;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int f = (int)(e % k);
    long r = e / k;
    const int a = (int)(r % A);
    r /= A;
    const int t = 1 + (int)(r % (Tn - 1));
    const long b = r / (Tn - 1);
    x[((b * Tn + t) * A + a) * F + f] = x[((b * Tn) * A + a) * F + f];
  }
}

// sampler state: state[0] = step index.  fill: ts_out[b] = tsteps[state[0]]; advance: state[0] += 1.
__global__ void sampler_fill_time_kernel(const int* __restrict__ state, const long* __restrict__ tsteps,
                                         long* __restrict__ ts_out, int B) {
  long t = tsteps[state[0]];
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) ts_out[b] = t;
}
__global__ void sampler_advance_kernel(int* state) {
  if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += 1;
}

// ---------------------------------------------------------------- masked weighted MSE (helpers.py:243-250 + diffusion.py:328-330)
// sums[b] = { sum_{t,a,f} (p-q)^2 w m , sum_{t,a} m , sum_{t,a,f} (p-q)^2 w }
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ targ,
                                                      const float* __restrict__ w, const float* __restrict__ mask,
                                                      float* __restrict__ sums, int TA, int F) {
  __shared__ float red[3][4];
  const int b = blockIdx.x;
  const long base = (long)b * TA * F;
  float s = 0.f, u = 0.f, ms = 0.f;
  for (int i = threadIdx.x; i < TA * F; i += 256) {
    float d = pred[base + i] - targ[base + i];
    float l = d * d * w[i];
    u += l;
    s += l * mask[(long)b * TA + i / F];
  }
  for (int i = threadIdx.x; i < TA; i += 256) ms += mask[(long)b * TA + i];
  s = wave_sum(s); u = wave_sum(u); ms = wave_sum(ms);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s; red[1][wave] = ms; red[2][wave] = u; }
  __syncthreads();
  if (threadIdx.x < 3) sums[b * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// d pred = go * 2 (p - q) w m / (B F sum_m[b])
__global__ void mse_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ targ,
                               const float* __restrict__ w, const float* __restrict__ mask,
                               const float* __restrict__ sums, const float* __restrict__ go, float* __restrict__ dpred,
                               int B, int TA, int F) {
  const long total = (long)B * TA * F;
  const float g = go[0] * 2.f / ((float)B * (float)F);
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int b = e / ((long)TA * F);
    int i = e - (long)b * TA * F;
    float m = mask[(long)b * TA + i / F];
    dpred[e] = g * (pred[e] - targ[e]) * w[i] * m / sums[b * 3 + 1];
  }
}

// ---------------------------------------------------------------- sinusoidal timestep embedding (helpers.py:17-29)
template <typename T>
__global__ void sinusoid_kernel(const long* __restrict__ ts, const float* __restrict__ freq, T* __restrict__ out,
                                int ld, int B, int A, int half) {
  const long total = (long)B * A * ld;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int j = e % ld;
    int b = (e / ld) / A;
    float v = 0.f;
    if (j < 2 * half) {
      float arg = (float)ts[b] * freq[j < half ? j : j - half];
      v = j < half ? sinf(arg) : cosf(arg);
    }
    out[e] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------- activations on flat (padded) matrices
template <typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n, int kind) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float v = to_f32(x[e]);
    y[e] = from_f32<T>(kind == 2 ? mish_f(v) : fmaxf(v, 0.f));
  }
}
// kind 2 (mish): ref = pre-activation.  kind 1 (relu): ref = output (or pre-activation; same sign test).
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ ref, const T* __restrict__ dy, T* __restrict__ dx, long n, int kind) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    float v = to_f32(ref[e]), g = to_f32(dy[e]);
    dx[e] = from_f32<T>(kind == 2 ? g * mish_grad_f(v) : (v > 0.f ? g : 0.f));
  }
}

// ---------------------------------------------------------------- reductions for bias / row-bias gradients
// colsum: out[c] += sum_m x[m, c]  (caller zeroes out).  grid = (ceil(C/64), row blocks); block 64x4.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ld, float* __restrict__ out, long M, int C) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ry = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C)
    for (long m = (long)blockIdx.y * 4 + ry; m < M; m += (long)gridDim.y * 4) s += to_f32(x[m * ld + c]);
  red[ry][threadIdx.x & 63] = s;
  __syncthreads();
  if (ry == 0 && c < C) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// vectorised colsum (ld and C multiples of the 16-byte vector): 32 column vectors x 8 rows per block
template <typename T, int E>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, int ld, float* __restrict__ out, long M, int C,
                                                            float* const* __restrict__ tab = nullptr) {
  __shared__ float red[8][32 * E];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = (blockIdx.x * 32 + tx) * E;
  float acc[E];
#pragma unroll
  for (int e = 0; e < E; e++) acc[e] = 0.f;
  if (c0 < C) {
    for (long m = (long)blockIdx.y * 8 + ty; m < M; m += (long)gridDim.y * 8) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 v = *(const bf16x8*)(x + m * ld + c0);
#pragma unroll
        for (int e = 0; e < E; e++) acc[e] += (float)v[e];
      } else {
        f32x4 v = *(const f32x4*)(x + m * ld + c0);
#pragma unroll
        for (int e = 0; e < E; e++) acc[e] += v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < E; e++) red[ty][tx * E + e] = acc[e];
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * E; i += 256) {
    int c = blockIdx.x * 32 * E + i;
    if (c < C) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; r++) s += red[r][i];
      atomicAdd(tab ? tab[c] : &out[c], s);       // tab: one destination address per column (a group of bias gradients)
    }
  }
}

// seqsum: out[n, c] = sum_t x[(n, t), c]
template <typename T>
__global__ void seqsum_kernel(const T* __restrict__ x, int ld, float* __restrict__ out, int ldo, long N, int Tn, int C) {
  const long total = N * C;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long n = e / C;
    int c = e - n * C;
    float s = 0.f;
    for (int t = 0; t < Tn; t++) s += to_f32(x[(n * Tn + t) * ld + c]);
    out[n * ldo + c] = s;
  }
}

// ---------------------------------------------------------------- fused Adam + EMA over one flat tensor
// torch.optim.Adam defaults (training.py:109): no weight decay, no amsgrad.  EMA (training.py:20-39):
// ema = beta*ema + (1-beta)*p when do_ema == 1, ema = p when do_ema == 2 (the "copy until step 2000" phase).
// shadow != null: the bf16 copy of the updated parameter is written alongside (the flat buffer is kept in GEMM layout,
// so this IS the packed forward weight of the next step: no separate pack pass).  zero_grad: g is cleared after use.
__device__ __forceinline__ float adam_elem(float& gr_io, float& m_io, float& v_io, float p_in, float lr, float b1, float b2,
                                          float eps, float bc1, float bc2_sqrt, float grad_scale) {
  const float gr = gr_io * grad_scale;
  const float mm = b1 * m_io + (1.f - b1) * gr;
  const float vv = b2 * v_io + (1.f - b2) * gr * gr;
  m_io = mm;
  v_io = vv;
  const float denom = sqrtf(vv) / bc2_sqrt + eps;
  return p_in - (lr / bc1) * (mm / denom);
}

// VEC: 16-byte accesses on every stream (all pointers 16-byte aligned, n % 4 == 0: the flat buffers and their bucket
// slices are); otherwise one element per access.
template <bool VEC>
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, float* __restrict__ ema, long n, float lr, float b1, float b2,
                                float eps, float bc1, float bc2_sqrt, float grad_scale, int do_ema, float ema_beta,
                                const int* __restrict__ state, bf16_t* __restrict__ shadow, int zero_grad) {
  if (state) {   // device-resident step counter / EMA flag: one captured launch serves every step of a hipGraph
    const double step = (double)(state[0] + 1);
    bc1 = (float)(1.0 - pow((double)b1, step));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, step));
    do_ema = state[1];
  }
  const long stride = (long)gridDim.x * blockDim.x, tid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if constexpr (VEC) {
    typedef __attribute__((ext_vector_type(4))) float f4;
    typedef __attribute__((ext_vector_type(4))) bf16_t b4;
    const long n4 = n >> 2;
    for (long q = tid; q < n4; q += stride) {
      f4 gv = ((const f4*)g)[q], mv = ((const f4*)m)[q], vv = ((const f4*)v)[q], pv = ((const f4*)p)[q];
      f4 ev;
      if (do_ema == 1) ev = ((const f4*)ema)[q];
      if (zero_grad) ((f4*)g)[q] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float gg = gv[e], mm = mv[e], v2 = vv[e];
        pv[e] = adam_elem(gg, mm, v2, pv[e], lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
        mv[e] = mm;
        vv[e] = v2;
      }
      ((f4*)m)[q] = mv;
      ((f4*)v)[q] = vv;
      ((f4*)p)[q] = pv;
      if (shadow) {
        b4 sv;
#pragma unroll
        for (int e = 0; e < 4; e++) sv[e] = (bf16_t)pv[e];
        ((b4*)shadow)[q] = sv;
      }
      if (do_ema == 1) {
#pragma unroll
        for (int e = 0; e < 4; e++) ev[e] = ema_beta * ev[e] + (1.f - ema_beta) * pv[e];
        ((f4*)ema)[q] = ev;
      } else if (do_ema == 2) {
        ((f4*)ema)[q] = pv;
      }
    }
  } else {
    for (long e = tid; e < n; e += stride) {
      float gr = g[e], mm = m[e], vv = v[e];
      if (zero_grad) g[e] = 0.f;
      const float pv = adam_elem(gr, mm, vv, p[e], lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
      m[e] = mm;
      v[e] = vv;
      p[e] = pv;
      if (shadow) shadow[e] = (bf16_t)pv;
      if (do_ema == 1) ema[e] = ema_beta * ema[e] + (1.f - ema_beta) * pv;
      else if (do_ema == 2) ema[e] = pv;
    }
  }
}

static inline int nblocks(long total, int per = 256, int cap = 4096) {
  long b = (total + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

extern "C" int madiff_pack_input(int dtype, const float* x, const float* noise, const long* tsteps, const float* sqrt_ac,
                                 const float* sqrt_1mac, const float* cond, const uint8_t* mask, void* out,
                                 float* x_noisy, int B, int T, int A, int F, int Fpad, int x_ld, hipStream_t stream) {
  if (B <= 0) return 0;
  if (x_ld <= 0) x_ld = F;
  if (T <= 0 || A <= 0 || F <= 0 || Fpad < F || x_ld < F || (noise && (!tsteps || !sqrt_ac || !sqrt_1mac)) || (mask && !cond)) return -1;
  long total = (long)B * A * T * Fpad;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((pack_input_kernel<bf16_t>), dim3(nblocks(total)), dim3(256), 0, stream, x, noise, tsteps, sqrt_ac,
                       sqrt_1mac, cond, mask, (bf16_t*)out, x_noisy, B, T, A, F, Fpad, x_ld);
  else
    hipLaunchKernelGGL((pack_input_kernel<float>), dim3(nblocks(total)), dim3(256), 0, stream, x, noise, tsteps, sqrt_ac,
                       sqrt_1mac, cond, mask, (float*)out, x_noisy, B, T, A, F, Fpad, x_ld);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_unpack_output(int dtype, const void* in, int ld, float* out, int B, int T, int A, int F,
                                    hipStream_t stream) {
  if (B <= 0) return 0;
  if (T <= 0 || A <= 0 || F <= 0 || ld < F) return -1;
  long total = (long)B * T * A * F;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((unpack_output_kernel<bf16_t>), dim3(nblocks(total)), dim3(256), 0, stream, (const bf16_t*)in, ld, out, B, T, A, F);
  else
    hipLaunchKernelGGL((unpack_output_kernel<float>), dim3(nblocks(total)), dim3(256), 0, stream, (const float*)in, ld, out, B, T, A, F);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_posterior_step(const float* x, const float* eps_c, const float* eps_u, float w, const float* z,
                                     const float* cond, const uint8_t* mask, const float* coef, const int* step_ptr,
                                     int step_val, float* out, long total, hipStream_t stream) {
  if (total <= 0) return 0;
  if (!x || !eps_c || !coef || !out || (mask && !cond) || (!step_ptr && step_val < 0)) return -1;
  hipLaunchKernelGGL(posterior_step_kernel, dim3(nblocks(total)), dim3(256), 0, stream, x, eps_c, eps_u, w, z, cond, mask,
                     coef, step_ptr, step_val, out, total);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// The posterior step with device-side noise (see posterior_step_rng_kernel): z_all = [n_steps, total] explicit draws (row =
// the step index) or NULL; rng = device {seed, stream id} (uint64[2]) for the Philox draw when z_all is NULL (one of the two
// must be given unless every sigma of the coefficient table is 0: DDIM).  Replaces diffusers' `randn_tensor` inside
// DDPMScheduler.step (called at diffusion.py:261), so the K-step loop needs no host work between steps.
extern "C" int madiff_posterior_step_rng(const float* x, const float* eps_c, const float* eps_u, float w, const float* z_all,
                                         const unsigned long long* rng, const float* cond, const uint8_t* mask,
                                         const float* coef, const int* step_ptr, int step_val, float* out, long total,
                                         hipStream_t stream) {
  if (total <= 0) return 0;
  if (!x || !eps_c || !coef || !out || (mask && !cond) || (!step_ptr && step_val < 0)) return -1;
  hipLaunchKernelGGL(posterior_step_rng_kernel, dim3(nblocks((total + 3) / 4)), dim3(256), 0, stream, x, eps_c, eps_u, w, z_all,
                     rng, cond, mask, coef, step_ptr, step_val, out, total);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_inpaint(float* x, const float* cond, const uint8_t* mask, long total, hipStream_t stream) {
  if (total <= 0) return 0;
  if (!x || !cond || !mask) return -1;
  hipLaunchKernelGGL(inpaint_kernel, dim3(nblocks(total)), dim3(256), 0, stream, x, cond, mask, total);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_encode_first_k(float* x, int B, int T, int A, int F, int k, hipStream_t stream) {
  if (B <= 0 || T <= 1 || k <= 0) return 0;
  if (!x || A <= 0 || F <= 0 || k > F) return -1;
  hipLaunchKernelGGL(encode_first_k_kernel, dim3(nblocks((long)B * (T - 1) * A * k)), dim3(256), 0, stream, x, B, T, A, F, k);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_sampler_fill_time(const int* state, const long* tsteps, long* ts_out, int B, hipStream_t stream) {
  if (B <= 0 || !state || !tsteps || !ts_out) return -1;
  hipLaunchKernelGGL(sampler_fill_time_kernel, dim3(nblocks(B)), dim3(256), 0, stream, state, tsteps, ts_out, B);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_sampler_advance(int* state, hipStream_t stream) {
  if (!state) return -1;
  hipLaunchKernelGGL(sampler_advance_kernel, dim3(1), dim3(64), 0, stream, state);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_mse_fwd(const float* pred, const float* targ, const float* w, const float* mask, float* sums,
                              int B, int TA, int F, hipStream_t stream) {
  if (B <= 0) return 0;
  if (TA <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(B), dim3(256), 0, stream, pred, targ, w, mask, sums, TA, F);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_mse_bwd(const float* pred, const float* targ, const float* w, const float* mask, const float* sums,
                              const float* go, float* dpred, int B, int TA, int F, hipStream_t stream) {
  if (B <= 0) return 0;
  if (TA <= 0 || F <= 0) return -1;
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(nblocks((long)B * TA * F)), dim3(256), 0, stream, pred, targ, w, mask, sums, go,
                     dpred, B, TA, F);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------- the embedding head in ONE launch (bf16 build, inference)
// temporal.py:231-252 + ma_temporal.py:341-354: t -> SinusoidalPosEmb -> Linear(dim, 4 dim) -> Mish -> Linear(4 dim, dim)   (y = 0)
//                                               r -> Linear(1, dim) -> Mish -> Linear(dim, 4 dim) -> Mish -> Linear(4 dim, dim),
//                                               times the dropout mask / zero under force_dropout                      (y = 1)
// and the concatenation [time | returns] -- ~25 dependent launches of a few microseconds each at the head of every denoiser
// forward (sinusoid, five GEMMs, four Mish, mask, cat, casts), which is what the small-batch sampler (the evaluator's 10
// environments: one graph node costs ~10 us whatever it computes) and every one of the K denoising steps pay first.
// One workgroup = 16 rows (b, a) of one of the two MLPs (blockIdx.y); both wide layers run on v_mfma_f32_16x16x32_bf16 with
// the A operand in LDS (the 16 rows) and the B operand -- 8 consecutive k of weight row n, the GEMM layout [O, pad32(I)] the
// other kernels use -- loaded straight from global memory into the fragment registers.  act = 1 stores mish(value): the
// only consumer is Linear(Mish(t_emb)) (temporal.py:166-170, 60-72).  out2 (may be null) receives the classifier-free
// guidance twin of every row: same time half, returns half forced to zero (diffusion.py:195-202).
__global__ __launch_bounds__(256) void embed_mlp_fwd_kernel(
    const long* __restrict__ ts, const float* __restrict__ freq, const bf16_t* __restrict__ W1, int ld1,
    const float* __restrict__ b1, const bf16_t* __restrict__ W3, int ld3, const float* __restrict__ b3,
    const float* __restrict__ ret, const float* __restrict__ M0, const float* __restrict__ c0, const bf16_t* __restrict__ M2,
    int ldm2, const float* __restrict__ c2, const bf16_t* __restrict__ M4, int ldm4, const float* __restrict__ c4,
    const float* __restrict__ mask, int force_zero, bf16_t* __restrict__ out, int ldo, bf16_t* __restrict__ out2, int B, int A,
    int dim, int act) {
  extern __shared__ __attribute__((aligned(16))) unsigned char esm[];
  const int H1 = 4 * dim, lde = dim + 8, ldh = H1 + 8;
  bf16_t* Ein = (bf16_t*)esm;                       // [16][dim + 8]
  bf16_t* Hs = Ein + 16 * lde;                      // [16][4 dim + 8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fq = lane >> 4;
  const int which = blockIdx.y, N = B * A, r0 = blockIdx.x * 16;
  const bf16_t* Wa = which ? M2 : W1;
  const int lda = which ? ldm2 : ld1;
  const float* ba = which ? c2 : b1;
  const bf16_t* Wb = which ? M4 : W3;
  const int ldb = which ? ldm4 : ld3;
  const float* bb = which ? c4 : b3;
  // ---- inputs of the first wide layer
  const int half = dim >> 1;
  for (int i = tid; i < 16 * dim; i += 256) {
    const int rr = i / dim, j = i - rr * dim;
    const int row = r0 + rr;
    float v = 0.f;
    if (row < N) {
      if (which == 0) {
        const float arg = (float)ts[row / A] * freq[j < half ? j : j - half];
        v = j < half ? sinf(arg) : cosf(arg);
      } else {
        v = mish_f(M0[j] * ret[row] + c0[j]);
      }
    }
    Ein[rr * lde + j] = (bf16_t)v;
  }
  __syncthreads();
  // ---- hidden = mish(in . Wa^T + ba): 16 x 4 dim, 16-unit tiles dealt out to the waves
  const int ks1 = dim >> 5;                         // K-steps of 32 (dim is a multiple of 32, at most 128)
  bf16x8 a1[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (k < ks1) a1[k] = *(const bf16x8*)(Ein + fr * lde + k * 32 + fq * 8);
  for (int nt = wave; nt < (H1 >> 4); nt += 4) {
    const bf16_t* wrow = Wa + (size_t)(nt * 16 + fr) * lda + fq * 8;
    bf16x8 b[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < ks1) b[k] = *(const bf16x8*)(wrow + k * 32);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < ks1) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[k], b[k], acc, 0, 0, 0);
    const float bias = ba[nt * 16 + fr];
#pragma unroll
    for (int r = 0; r < 4; r++) Hs[(fq * 4 + r) * ldh + nt * 16 + fr] = (bf16_t)mish_f(acc[r] + bias);
  }
  __syncthreads();
  // ---- out = hidden . Wb^T + bb: 16 x dim
  const int ks2 = H1 >> 5;
  for (int nt = wave; nt < (dim >> 4); nt += 4) {
    const bf16_t* wrow = Wb + (size_t)(nt * 16 + fr) * ldb + fq * 8;
    const bf16_t* hrow = Hs + fr * ldh + fq * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ks2; k0 += 4) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int k = 0; k < 4; k++) { a[k] = *(const bf16x8*)(hrow + (k0 + k) * 32); b[k] = *(const bf16x8*)(wrow + (k0 + k) * 32); }
#pragma unroll
      for (int k = 0; k < 4; k++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[k], acc, 0, 0, 0);
    }
    const int col = nt * 16 + fr;
    const float bias = bb[col];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = r0 + fq * 4 + r;
      if (row >= N) continue;
      float v = acc[r] + bias;
      if (which) v = force_zero ? 0.f : (mask ? v * mask[row] : v);
      out[(size_t)row * ldo + which * dim + col] = (bf16_t)(act ? mish_f(v) : v);
      if (out2) out2[(size_t)row * ldo + which * dim + col] = (bf16_t)(which ? 0.f : (act ? mish_f(v) : v));
    }
  }
}

// dim: a multiple of 32, <= 128 (4 dim is then a multiple of 128).  ret == NULL: no returns branch (the time half only).
// Weight matrices: bf16 [O, ld] in the GEMM layout (rows 16-byte aligned); biases and the Linear(1, dim) parameters fp32.
extern "C" int madiff_embed_mlp_fwd(const long* ts, const float* freq, const void* W1, int ld1, const float* b1, const void* W3,
                                    int ld3, const float* b3, const float* ret, const float* M0, const float* c0, const void* M2,
                                    int ldm2, const float* c2, const void* M4, int ldm4, const float* c4, const float* mask,
                                    int force_zero, void* out, int ldo, void* out2, int B, int A, int dim, int act,
                                    hipStream_t stream) {
  if (B <= 0 || A <= 0) return 0;
  if (dim <= 0 || dim % 32 || dim > 128) return 1;
  if (!ts || !freq || !W1 || !b1 || !W3 || !b3 || !out || ld1 < dim || ld3 < 4 * dim || ld1 % 8 || ld3 % 8) return -1;
  if (ret && (!M0 || !c0 || !M2 || !c2 || !M4 || !c4 || ldm2 < dim || ldm4 < 4 * dim || ldm2 % 8 || ldm4 % 8)) return -1;
  if (ldo < (ret ? 2 : 1) * dim) return -1;
  if (((size_t)W1 % 16) || ((size_t)W3 % 16) || (ret && (((size_t)M2 % 16) || ((size_t)M4 % 16)))) return 1;
  const size_t sh = (size_t)16 * ((dim + 8) + (4 * dim + 8)) * 2;
  const dim3 grid((B * A + 15) / 16, ret ? 2 : 1);
  hipLaunchKernelGGL(embed_mlp_fwd_kernel, grid, dim3(256), sh, stream, ts, freq, (const bf16_t*)W1, ld1, b1, (const bf16_t*)W3,
                     ld3, b3, ret, M0, c0, (const bf16_t*)M2, ldm2, c2, (const bf16_t*)M4, ldm4, c4, mask, force_zero,
                     (bf16_t*)out, ldo, (bf16_t*)out2, B, A, dim, act);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_sinusoid(int dtype, const long* ts, const float* freq, void* out, int ld, int B, int A, int half,
                               hipStream_t stream) {
  if (B <= 0 || A <= 0 || half <= 0 || ld < 2 * half) return -1;
  long total = (long)B * A * ld;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((sinusoid_kernel<bf16_t>), dim3(nblocks(total)), dim3(256), 0, stream, ts, freq, (bf16_t*)out, ld, B, A, half);
  else
    hipLaunchKernelGGL((sinusoid_kernel<float>), dim3(nblocks(total)), dim3(256), 0, stream, ts, freq, (float*)out, ld, B, A, half);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_act_fwd(int dtype, int kind, const void* x, void* y, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  if (kind != 1 && kind != 2) return -1;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((act_fwd_kernel<bf16_t>), dim3(nblocks(n)), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, kind);
  else
    hipLaunchKernelGGL((act_fwd_kernel<float>), dim3(nblocks(n)), dim3(256), 0, stream, (const float*)x, (float*)y, n, kind);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_act_bwd(int dtype, int kind, const void* ref, const void* dy, void* dx, long n, hipStream_t stream) {
  if (n <= 0) return 0;
  if (kind != 1 && kind != 2) return -1;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), dim3(nblocks(n)), dim3(256), 0, stream, (const bf16_t*)ref, (const bf16_t*)dy, (bf16_t*)dx, n, kind);
  else
    hipLaunchKernelGGL((act_bwd_kernel<float>), dim3(nblocks(n)), dim3(256), 0, stream, (const float*)ref, (const float*)dy, (float*)dx, n, kind);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// Column sums of a GROUP of layers' output gradient, each column added (atomically) to its own destination:
// table[c] = address of the gradient element of column c (device array of C pointers).  Replaces one temporary, one
// reduction and one `grad += slice` pass per member of a q/k/v or time-MLP bundle.  Requires the vector layout
// (ld, C multiples of 8 (bf16) / 4 (fp32), 16-byte aligned x); returns 1 otherwise.
extern "C" int madiff_colsum_scatter(int dtype, const void* x, int ld, float* const* table, long M, int C, hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (ld < C || !table) return -1;
  const int E = dtype == MADIFF_BF16 ? 8 : 4;
  if (ld % E || C % E || (size_t)x % 16) return 1;
  int rbv = (int)((M + 63) / 64);
  if (rbv > 128) rbv = 128;
  dim3 g((C / E + 31) / 32, rbv);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 8>), g, dim3(256), 0, stream, (const bf16_t*)x, ld, nullptr, M, C, table);
  else
    hipLaunchKernelGGL((colsum_vec_kernel<float, 4>), g, dim3(256), 0, stream, (const float*)x, ld, nullptr, M, C, table);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_colsum(int dtype, const void* x, int ld, float* out, long M, int C, hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (ld < C) return -1;
  {
    const int E = dtype == MADIFF_BF16 ? 8 : 4;
    if (ld % E == 0 && C % E == 0 && (size_t)x % 16 == 0) {
      int rbv = (int)((M + 63) / 64);
      if (rbv > 128) rbv = 128;
      if (rbv < 1) rbv = 1;
      dim3 g((C / E + 31) / 32, rbv);
      if (dtype == MADIFF_BF16)
        hipLaunchKernelGGL((colsum_vec_kernel<bf16_t, 8>), g, dim3(256), 0, stream, (const bf16_t*)x, ld, out, M, C);
      else
        hipLaunchKernelGGL((colsum_vec_kernel<float, 4>), g, dim3(256), 0, stream, (const float*)x, ld, out, M, C);
      MADIFF_CHECK_LAUNCH();
      return 0;
    }
  }
  int rb = (int)((M + 255) / 256);
  if (rb > 64) rb = 64;
  if (rb < 1) rb = 1;
  dim3 grid((C + 63) / 64, rb);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, stream, (const bf16_t*)x, ld, out, M, C);
  else
    hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, stream, (const float*)x, ld, out, M, C);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_seqsum(int dtype, const void* x, int ld, float* out, int ldo, long N, int T, int C, hipStream_t stream) {
  if (N <= 0 || C <= 0) return 0;
  if (ld < C || ldo < C || T <= 0) return -1;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((seqsum_kernel<bf16_t>), dim3(nblocks(N * C)), dim3(256), 0, stream, (const bf16_t*)x, ld, out, ldo, N, T, C);
  else
    hipLaunchKernelGGL((seqsum_kernel<float>), dim3(nblocks(N * C)), dim3(256), 0, stream, (const float*)x, ld, out, ldo, N, T, C);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_adam_ema(float* p, float* g, float* m, float* v, float* ema, long n, float lr, float b1,
                               float b2, float eps, float bc1, float bc2s, float grad_scale, int do_ema,
                               float ema_beta, int* state, void* shadow_bf16, int zero_grad, hipStream_t stream) {
  // state == null: bc1 = 1 - b1^step, bc2s = sqrt(1 - b2^step) and do_ema come from the host (computed in double, as
  // torch.optim.Adam does).  state != null (device int[2] = {completed steps, do_ema}): both are derived on the device
  // and state[0] is incremented afterwards, so the launch pair can live inside a replayed hipGraph.
  // zero_grad: bit 0 = clear g after use; bit 1 = do NOT advance state[0] after this launch (the optimizer step is issued
  // as several slices, e.g. one per gradient bucket: only the last call advances).  n == 0 with state: advance only.
  if (n < 0) return -1;
  if (n > 0) {
    if (!p || !g || !m || !v || (!state && (bc1 <= 0.f || bc2s <= 0.f)) || ((do_ema || state) && !ema && do_ema)) return -1;
    const bool vec = n % 4 == 0 && (((size_t)p | (size_t)g | (size_t)m | (size_t)v | (size_t)ema) % 16) == 0 &&
                     ((size_t)shadow_bf16 % 8) == 0;
    if (vec)
      hipLaunchKernelGGL(adam_ema_kernel<true>, dim3(nblocks(n / 4, 256, 8192)), dim3(256), 0, stream, p, g, m, v, ema, n, lr,
                         b1, b2, eps, bc1, bc2s, grad_scale, do_ema, ema_beta, (const int*)state, (bf16_t*)shadow_bf16,
                         zero_grad & 1);
    else
      hipLaunchKernelGGL(adam_ema_kernel<false>, dim3(nblocks(n, 256, 8192)), dim3(256), 0, stream, p, g, m, v, ema, n, lr,
                         b1, b2, eps, bc1, bc2s, grad_scale, do_ema, ema_beta, (const int*)state, (bf16_t*)shadow_bf16,
                         zero_grad & 1);
    MADIFF_CHECK_LAUNCH();
  }
  if (state && !(zero_grad & 2)) {
    hipLaunchKernelGGL(sampler_advance_kernel, dim3(1), dim3(64), 0, stream, state);
    MADIFF_CHECK_LAUNCH();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Graph-external events: a point inside a captured (hipGraph) train step that streams OUTSIDE the graph can wait on.
// The data-parallel step replays backward from a graph and starts each gradient bucket's RCCL all-reduce on a
// communication stream as soon as the bucket's last gradient kernel has run, without putting any collective into the
// graph.  torch's own external events are disabled on ROCm builds, hence these four thin wrappers.
// ------------------------------------------------------------------------------------------------
extern "C" int madiff_event_create(void** event_out) {
  if (!event_out) return -1;
  hipEvent_t e;
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) return (int)rc;
  *event_out = (void*)e;
  return 0;
}
// Timing-enabled twin (per-launch kernel durations INSIDE a replayed graph: an event-record node before and after the
// launch, hipEventElapsedTime after the replay has finished).
extern "C" int madiff_event_create_timed(void** event_out) {
  if (!event_out) return -1;
  hipEvent_t e;
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDefault);
  if (rc != hipSuccess) return (int)rc;
  *event_out = (void*)e;
  return 0;
}
extern "C" int madiff_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return -1;
  return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}
extern "C" int madiff_event_destroy(void* event) { return event ? (int)hipEventDestroy((hipEvent_t)event) : 0; }
// On a capturing stream this adds an event-record NODE to the graph (executed at every replay); on an ordinary stream it
// is a plain hipEventRecord.
extern "C" int madiff_event_record_external(void* event, hipStream_t stream) {
  if (!event) return -1;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  hipError_t rc = hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, &deps, &ndeps);
  if (rc != hipSuccess) return (int)rc;
  if (st != hipStreamCaptureStatusActive) return (int)hipEventRecord((hipEvent_t)event, stream);
  // (hipEventRecordWithFlags(.., hipEventRecordExternal) is rejected by this runtime: add the node by hand and make it
  // the capturing stream's new dependency frontier, which is what the flag stands for)
  hipGraphNode_t node;
  rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)event);
  if (rc != hipSuccess) return (int)rc;
  return (int)hipStreamUpdateCaptureDependencies(stream, &node, 1, hipStreamSetCaptureDependencies);
}
// "waiter waits for everything captured so far on waited", for two streams ALREADY inside the same capture, without
// hipStreamWaitEvent: waited's dependency frontier is added to waiter's by hand.  Reason: on this runtime every
// hipStreamWaitEvent by a non-origin capturing stream appends the waiter to the waited stream's list of parallel capture
// streams, and hip::Stream::EndCapture() recurses over those lists without a visited set: two side streams that have
// waited on EACH OTHER (in any order, at any distance) make hipStreamEndCapture recurse until the stack overflows
// (scripts/dbg/capture_topology.cpp topologies 6/7, profiles/r02_capture_topology.txt).  Returns 1 when the streams are
// not both capturing the same graph (the caller then uses an ordinary event record + wait, which is also how a stream
// JOINS a capture).
extern "C" int madiff_capture_join(hipStream_t waiter, hipStream_t waited) {
  hipStreamCaptureStatus s1 = hipStreamCaptureStatusNone, s2 = hipStreamCaptureStatusNone;
  unsigned long long id1 = 0, id2 = 0;
  hipGraph_t g1 = nullptr, g2 = nullptr;
  const hipGraphNode_t *d1 = nullptr, *d2 = nullptr;
  size_t n1 = 0, n2 = 0;
  hipError_t rc = hipStreamGetCaptureInfo_v2(waiter, &s1, &id1, &g1, &d1, &n1);
  if (rc != hipSuccess) return (int)rc;
  rc = hipStreamGetCaptureInfo_v2(waited, &s2, &id2, &g2, &d2, &n2);
  if (rc != hipSuccess) return (int)rc;
  if (s1 != hipStreamCaptureStatusActive || s2 != hipStreamCaptureStatusActive || id1 != id2 || g1 != g2) return 1;
  if (n2 == 0) return 0;
  {
    // A waiter that already waits for exactly these nodes (same capture) keeps doing so through its own chain of nodes:
    // adding them again only multiplies edges (the optimizer's per-slice joins made every Adam node depend on the same
    // six nodes, and gave one data-gradient node 16 successors).
    static std::mutex mu;
    static std::map<std::pair<hipStream_t, hipStream_t>, std::pair<unsigned long long, std::vector<hipGraphNode_t>>> joined;
    std::lock_guard<std::mutex> lock(mu);
    auto& e = joined[std::make_pair(waiter, waited)];
    if (e.first == id1 && e.second.size() == n2 && std::equal(e.second.begin(), e.second.end(), d2)) return 0;
    e.first = id1;
    e.second.assign(d2, d2 + n2);
  }
  return (int)hipStreamUpdateCaptureDependencies(waiter, const_cast<hipGraphNode_t*>(d2), n2, hipStreamAddCaptureDependencies);
}
// 0 = not capturing, 1 = capturing, 2 = its capture has been invalidated (any further call on it fails), < 0 = -hipError_t
extern "C" int madiff_stream_capture_status(hipStream_t stream) {
  hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
  hipError_t rc = hipStreamIsCapturing(stream, &s);
  if (rc != hipSuccess) { (void)hipGetLastError(); return -(int)rc; }
  return s == hipStreamCaptureStatusActive ? 1 : (s == hipStreamCaptureStatusNone ? 0 : 2);
}
extern "C" int madiff_stream_wait_event(hipStream_t stream, void* event) {
  if (!event) return -1;
  return (int)hipStreamWaitEvent(stream, (hipEvent_t)event, 0);
}

extern "C" int madiff_version(void) { return 1; }
extern "C" const char* madiff_arch(void) { return "gfx950"; }
