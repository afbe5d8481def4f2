// bf16 implicit-GEMM forward/dgrad kernel, tuned structure for gfx950:
//   * 128x128 output tile per 256-thread workgroup, K-step of 64 channels of one tap;
//   * both operands go HBM/L2 -> LDS with global_load_lds_dwordx4 (LDS-DMA, 16 B per lane, no VGPR staging and
//     no ds_write pass); the conv taps, sequence boundaries, M/N edges are handled by pointing the per-lane SOURCE
//     address at a 256-byte zero page, so the instruction stream has no predication;
//   * the LDS image is lane-linear (a DMA requirement: dest = wave base + lane*16), so the bank-conflict-free
//     layout is obtained by XOR-swizzling the 16-byte chunk index on the SOURCE side (chunk ^= row & 7) and
//     applying the same XOR when the MFMA fragments are read with ds_read_b128;
//   * two LDS buffers: the DMA of step s+1 is in flight while step s is multiplied.
// Same GemmFwdParams contract as gemm_fwd_kernel in gemm.hip (which remains the fp32 build's kernel and the
// fallback for Cin % 64 != 0).
#include "common.h"

struct RowMap {
  int T_out, T_in, stride, pad, tr;
};

struct GemmFwdParams {
  const void* X;   int ldx;
  const void* X2;  int ldx2;
  int Csplit;
  int M, Cin, taps;
  RowMap rm;
  const void* W;   int ldw;
  int Nout, Npad;
  const float* bias;
  const float* rowbias; int ldrb;
  void* C;         int ldc;
  int act;
  int permT, permA;
};

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

__device__ __forceinline__ int map_row2(int n, int t, int tap, const RowMap& rm) {
  int ti;
  if (!rm.tr) {
    ti = t * rm.stride + tap - rm.pad;
    if (ti < 0 || ti >= rm.T_in) return -1;
  } else {
    int num = t + rm.pad - tap;
    if (num < 0) return -1;
    ti = num / rm.stride;
    if (ti * rm.stride != num || ti >= rm.T_in) return -1;
  }
  return n * rm.T_in + ti;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <typename OutT>
__global__ __launch_bounds__(256, 2) void gemm_fwd_bf16_kernel(GemmFwdParams p) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2][32768];   // [buf][A tile 16 KB | W tile 16 KB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;

  // DMA slots: a tile is 128 rows x 8 chunks (16 B); one wave instruction fills 8 rows (1 KB).  Wave w issues the
  // instructions for rows [32w, 32w+32) of both operands: 4 + 4 per K-step.
  const int pos = lane & 7;              // chunk position inside the LDS row
  int a_n[4], a_t[4];
  bool a_ok[4];
  const unsigned char* w_src[4];
  int chunk[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    int row = 32 * wave + 8 * j + (lane >> 3);
    chunk[j] = pos ^ (row & 7);          // source-side swizzle
    int m = m0 + row;
    a_ok[j] = m < p.M;
    int n = m / p.rm.T_out;
    a_n[j] = n;
    a_t[j] = m - n * p.rm.T_out;
    int wr = n0 + row;
    w_src[j] = wr < p.Nout ? (const unsigned char*)p.W + ((size_t)wr * p.ldw) * 2 + chunk[j] * 16 : nullptr;
  }
  const int spt = p.Cin >> 6;            // K-steps per tap
  const int nsteps = spt * p.taps;

  // row index per slot for the current tap (-1 = zero row)
  int a_row[4];
  auto tap_rows = [&](int tap) {
#pragma unroll
    for (int j = 0; j < 4; j++) a_row[j] = a_ok[j] ? map_row2(a_n[j], a_t[j], tap, p.rm) : -1;
  };

  auto issue = [&](int buf, int tap, int cstep) {
    const int c0 = cstep << 6;
    const unsigned char* xb = (const unsigned char*)p.X;
    int ld = p.ldx, c = c0;
    if (c0 >= p.Csplit) { xb = (const unsigned char*)p.X2; ld = p.ldx2; c = c0 - p.Csplit; }
    const size_t wk = ((size_t)tap * p.Cin + c0) * 2;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned char* ga = a_row[j] >= 0 ? xb + ((size_t)a_row[j] * ld + c) * 2 + chunk[j] * 16
                                              : g_zero_page + (lane & 7) * 16;
      const unsigned char* gw = w_src[j] ? w_src[j] + wk : g_zero_page + (lane & 7) * 16;
      unsigned char* la = &smem[buf][(4 * wave + j) * 1024];
      unsigned char* lw = &smem[buf][16384 + (4 * wave + j) * 1024];
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gw, (lptr_t)lw, 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  int tap = 0, cstep = 0;
  tap_rows(0);
  issue(0, 0, 0);
  __syncthreads();
  for (int s = 0; s < nsteps; s++) {
    const int buf = s & 1;
    if (s + 1 < nsteps) {
      if (++cstep == spt) { cstep = 0; ++tap; tap_rows(tap); }
      issue(buf ^ 1, tap, cstep);
    }
    const unsigned char* As = &smem[buf][0];
    const unsigned char* Ws = &smem[buf][16384];
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int ra = wm * 64 + i * 16 + fr;
        int rb = wn * 64 + i * 16 + fr;
        a[i] = *(const bf16x8*)(As + ra * 128 + (((fq + 4 * kh) ^ (ra & 7)) << 4));
        b[i] = *(const bf16x8*)(Ws + rb * 128 + (((fq + 4 * kh) ^ (rb & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();     // emits vmcnt(0): the DMA of step s+1 has landed, and every wave is done reading buf
  }

  OutT* C = (OutT*)p.C;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int m = m0 + wm * 64 + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
      int seq = m / p.rm.T_out;
      size_t orow = m;
      if (p.permA > 0) {
        int t = m - seq * p.permT;
        int b_ = seq / p.permA, a_ = seq - b_ * p.permA;
        orow = ((size_t)b_ * p.permT + t) * p.permA + a_;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int n = n0 + wn * 64 + j * 16 + fr;
        if (n >= p.Npad) continue;
        float v = 0.f;
        if (n < p.Nout) {
          v = acc[i][j][r];
          if (p.bias) v += p.bias[n];
          if (p.rowbias) v += p.rowbias[(size_t)seq * p.ldrb + n];
          if (p.act == 1) v = fmaxf(v, 0.f);
          else if (p.act == 2) v = mish_f(v);
        }
        C[orow * p.ldc + n] = from_f32<OutT>(v);
      }
    }
  }
}

// launched from madiff_gemm_fwd (gemm.hip) when dtype == bf16 and Cin % 64 == 0 (and Csplit % 64 == 0)
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_gemm_fwd_bf16(const void* params, int out_f32, int grid_x, int grid_y, hipStream_t stream) {
  GemmFwdParams p = *(const GemmFwdParams*)params;
  dim3 grid(grid_x, grid_y), block(256);
  if (out_f32) hipLaunchKernelGGL((gemm_fwd_bf16_kernel<float>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((gemm_fwd_bf16_kernel<bf16_t>), grid, block, 0, stream, p);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16: dW[o, tap*Cin + c] = sum_m dY[m, o] * X[src(m, tap), c]
// Same LDS-DMA structure.  Both operands arrive with the REDUCTION index (m) as the row, so the MFMA fragments
// (8 consecutive reduction elements per lane) are read with ds_read_b64_tr_b16, the hardware transposing LDS read.
// Tiles are [64 reduction rows][128 columns] bf16 = 256-byte rows; chunk swizzle of a row:
//   ch ^= ((row & 3) << 2) | ((row >> 2) & 3)     (conflict-free for the transposed reads of a 16x16x32 operand)
// applied on the DMA source side and on the read side.
// ------------------------------------------------------------------------------------------------
struct GemmWgradParams {
  const void* dY;  int ldy;  int Ycols;
  const void* X;   int ldx;
  const void* X2;  int ldx2;
  int Csplit;
  int M, Cin, taps;
  RowMap rm;
  float* dW;       int ldw;
  int Nout;
  int ctiles;
  int steps_per_split, splits;
};

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__global__ __launch_bounds__(256, 2) void gemm_wgrad_bf16_kernel(GemmWgradParams p, unsigned magic) {
  constexpr int KR = 64;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2][32768];   // [buf][dY tile 16 KB | X tile 16 KB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * 128;
  const int tap = blockIdx.y / p.ctiles;
  const int c0 = (blockIdx.y - tap * p.ctiles) * 128;
  const int total_steps = (p.M + KR - 1) / KR;
  const int s_begin = blockIdx.z * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  if (s_begin >= s_end) return;

  const unsigned char* xsrc = (const unsigned char*)p.X;
  int ldx = p.ldx, cx = c0, cmax = p.Csplit;
  if (c0 >= p.Csplit) { xsrc = (const unsigned char*)p.X2; ldx = p.ldx2; cx = c0 - p.Csplit; cmax = p.Cin - p.Csplit; }

  // DMA slots: one wave instruction = 4 rows x 16 chunks.  Wave w fills rows [16w, 16w+16) of both tiles.
  const int pos = lane & 15;
  int srow[4], ych[4], xch[4];
  bool yok[4], xok[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    srow[j] = 16 * wave + 4 * j + (lane >> 4);
    int ch = pos ^ wg_swz(srow[j]);
    ych[j] = ch; xch[j] = ch;
    yok[j] = n0 + ch * 8 < p.Ycols;
    xok[j] = cx + ch * 8 < cmax;
  }
  const unsigned char* zero = g_zero_page + (lane & 15) * 16;

  auto issue = [&](int buf, int s) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int m = s * KR + srow[j];
      const unsigned char* gy = zero;
      const unsigned char* gx = zero;
      if (m < p.M) {
        if (yok[j]) gy = (const unsigned char*)p.dY + ((size_t)m * p.ldy + n0 + ych[j] * 8) * 2;
        if (xok[j]) {
          int n = magic ? (int)__umulhi((unsigned)m, magic) : m;   // magic == 0 <=> T_out == 1
          int r = map_row2(n, m - n * p.rm.T_out, tap, p.rm);
          if (r >= 0) gx = xsrc + ((size_t)r * ldx + cx + xch[j] * 8) * 2;
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)gy, (lptr_t)&smem[buf][(4 * wave + j) * 1024], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)gx, (lptr_t)&smem[buf][16384 + (4 * wave + j) * 1024], 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  issue(0, s_begin);
  __syncthreads();
  for (int s = s_begin; s < s_end; s++) {
    const int buf = (s - s_begin) & 1;
    if (s + 1 < s_end) issue(buf ^ 1, s + 1);
    const unsigned char* Ys = &smem[buf][0];
    const unsigned char* Xs = &smem[buf][16384];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      bf16x8 a[4], b[4];
      const int r0 = 32 * kb + 8 * fq + tq, r1 = r0 + 4;
      const int z0 = wg_swz(r0), z1 = wg_swz(r1);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int ca = 8 * wn + 2 * i + (tp >> 1), cb = 8 * wc + 2 * i + (tp >> 1);
        const int o = 8 * (tp & 1);
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ys + r0 * 256 + ((ca ^ z0) << 4) + o));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ys + r1 * 256 + ((ca ^ z1) << 4) + o));
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xs + r0 * 256 + ((cb ^ z0) << 4) + o));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xs + r1 * 256 + ((cb ^ z1) << 4) + o));
        a[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        b[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int n = n0 + wn * 64 + i * 16 + fq * 4 + r;
      if (n >= p.Nout) continue;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        int c = c0 + wc * 64 + j * 16 + fr;
        if (c >= p.Cin) continue;
        float* dst = p.dW + (size_t)n * p.ldw + (size_t)tap * p.Cin + c;
        if (p.splits > 1) atomicAdd(dst, acc[i][j][r]);
        else *dst = acc[i][j][r];
      }
    }
}

// KR = 64 reduction rows per step here (the generic kernel uses 32): the caller sizes steps_per_split for 64.
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_gemm_wgrad_bf16(const void* params, int gx, int gy, int gz,
                                                                                   hipStream_t stream) {
  GemmWgradParams p = *(const GemmWgradParams*)params;
  unsigned d = (unsigned)p.rm.T_out;
  unsigned magic = d == 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d);
  hipLaunchKernelGGL(gemm_wgrad_bf16_kernel, dim3(gx, gy, gz), dim3(256), 0, stream, p, magic);
  MADIFF_CHECK_LAUNCH();
  return 0;
}
