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
#include "gemm_params.h"

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Tile BM x BN (BM in {64, 96, 128}, BN in {64, 128}), 4 waves as 2 x 2, wave tile (BM/2) x (BN/2).
// WT = false: W rows are output columns ([Nout, taps*Cin], K contiguous), fragments by ds_read_b128.
// WT = true : W rows are REDUCTION rows (the forward weight layout seen from the data gradient):
//             B[k = (tap, o), n] = W[o*ldw + tap*wtap + n]; the B tile is [64 k][BN = 128 n] (256-byte rows, chunk swizzle
//             wg_swz) and its fragments are read with the transposing ds_read_b64_tr_b16, exactly as the weight-gradient
//             kernel reads its operands.  No transposed weight copy exists anywhere.
template <typename OutT, int BM, int BN, bool WT>
__global__ __launch_bounds__(256, 2) void gemm_fwd_bf16_kernel(GemmFwdParams p) {
  static_assert(!WT || BN == 128, "the transposed-read image is laid out for 256-byte rows");
  constexpr int MI = BM / 32, NJ = BN / 32;          // 16x16 MFMA blocks per wave (rows, cols) = DMA pieces per wave
  constexpr int A_BYTES = BM * 128, BUF_BYTES = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUF_BYTES];   // [buf][A tile | W tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

  // DMA pieces: one wave instruction moves 1 KB = 8 rows x 8 chunks (16 B) of a 128-byte-row tile.  Wave w issues the
  // pieces for rows [BM/4*w, BM/4*(w+1)) of A and the matching quarter of W.
  const int pos = lane & 7;
  int a_n[MI], a_t[MI], a_chunk[MI];
  bool a_ok[MI];
#pragma unroll
  for (int j = 0; j < MI; j++) {
    int row = (BM / 4) * wave + 8 * j + (lane >> 3);
    a_chunk[j] = pos ^ (row & 7);          // source-side swizzle
    int m = m0 + row;
    a_ok[j] = m < p.M;
    int n = m / p.rm.T_out;
    a_n[j] = n;
    a_t[j] = m - n * p.rm.T_out;
  }
  constexpr int WP = WT ? 4 : NJ;          // W pieces per wave
  const unsigned char* w_src[WP];
#pragma unroll
  for (int j = 0; j < WP; j++) {
    if constexpr (!WT) {
      int row = (BN / 4) * wave + 8 * j + (lane >> 3);
      int wr = n0 + row;
      w_src[j] = wr < p.Nout ? (const unsigned char*)p.W + ((size_t)wr * p.ldw) * 2 + (pos ^ (row & 7)) * 16 : nullptr;
    } else {
      int krow = 16 * wave + 4 * j + (lane >> 4);                  // reduction row inside the 64-row K-step
      int ch = (lane & 15) ^ wg_swz(krow);
      int col = n0 + ch * 8;
      w_src[j] = col < p.wcols ? (const unsigned char*)p.W + ((size_t)krow * p.ldw + col) * 2 : nullptr;
    }
  }
  const int spt = p.Cin >> 6;            // K-steps per tap
  const int nsteps = spt * p.taps;

  int a_row[MI];
  auto tap_rows = [&](int tap) {
#pragma unroll
    for (int j = 0; j < MI; j++) a_row[j] = a_ok[j] ? map_row(a_n[j], a_t[j], tap, p.rm) : -1;
  };

  auto issue = [&](int buf, int tap, int cstep) {
    const int c0 = cstep << 6;
    const unsigned char* xb = (const unsigned char*)p.X;
    int ld = p.ldx, c = c0;
    if (c0 >= p.Csplit) { xb = (const unsigned char*)p.X2; ld = p.ldx2; c = c0 - p.Csplit; }
    unsigned char* base = &smem[buf * BUF_BYTES];
#pragma unroll
    for (int j = 0; j < MI; j++) {
      const unsigned char* ga = a_row[j] >= 0 ? xb + ((size_t)a_row[j] * ld + c) * 2 + a_chunk[j] * 16
                                              : g_zero_page + (lane & 7) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)(base + (MI * wave + j) * 1024), 16, 0, 0);
    }
    const size_t wk = WT ? ((size_t)c0 * p.ldw + (size_t)tap * p.wtap) * 2 : ((size_t)tap * p.Cin + c0) * 2;
#pragma unroll
    for (int j = 0; j < WP; j++) {
      const unsigned char* gw = w_src[j] ? w_src[j] + wk : g_zero_page + (lane & 15) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)gw, (lptr_t)(base + A_BYTES + (WP * wave + j) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
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
    const unsigned char* As = &smem[buf * BUF_BYTES];
    const unsigned char* Ws = As + A_BYTES;
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      bf16x8 a[MI], b[NJ];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        int ra = wm * (BM / 2) + i * 16 + fr;
        a[i] = *(const bf16x8*)(As + ra * 128 + (((fq + 4 * kh) ^ (ra & 7)) << 4));
      }
      if constexpr (!WT) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          int rb = wn * (BN / 2) + j * 16 + fr;
          b[j] = *(const bf16x8*)(Ws + rb * 128 + (((fq + 4 * kh) ^ (rb & 7)) << 4));
        }
      } else {
        const int r0 = 32 * kh + 8 * fq + tq, r1 = r0 + 4;
        const int z0 = wg_swz(r0), z1 = wg_swz(r1);
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int cb = 8 * wn + 2 * j + (tp >> 1);
          const int o = 8 * (tp & 1);
          s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ws + r0 * 256 + ((cb ^ z0) << 4) + o));
          s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ws + r1 * 256 + ((cb ^ z1) << 4) + o));
          b[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();     // emits vmcnt(0): the DMA of step s+1 has landed, and every wave is done reading buf
  }

  OutT* C = (OutT*)p.C;
#pragma unroll
  for (int i = 0; i < MI; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int m = m0 + wm * (BM / 2) + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
      int seq = m / p.rm.T_out;
      size_t orow = m;
      if (p.permA > 0) {
        int t = m - seq * p.permT;
        int b_ = seq / p.permA, a_ = seq - b_ * p.permA;
        orow = ((size_t)b_ * p.permT + t) * p.permA + a_;
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        int n = n0 + wn * (BN / 2) + j * 16 + fr;
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

// Tile shapes the launcher can pick (index = the `tile` argument): the caller chooses per layer shape.
//   0: 128x128   1: 96x128   2: 64x128   3: 128x64   4: 96x64   5: 64x64
template <typename OutT, bool WT>
static int launch_tile(const GemmFwdParams& p, int tile, hipStream_t stream) {
#define MADIFF_TILE(ID, BM_, BN_)                                                                                    \
  case ID: {                                                                                                         \
    dim3 grid((p.M + BM_ - 1) / BM_, (p.Npad + BN_ - 1) / BN_);                                                       \
    hipLaunchKernelGGL((gemm_fwd_bf16_kernel<OutT, BM_, BN_, WT>), grid, dim3(256), 0, stream, p);                    \
    break;                                                                                                           \
  }
  switch (tile) {
    MADIFF_TILE(0, 128, 128)
    MADIFF_TILE(1, 96, 128)
    MADIFF_TILE(2, 64, 128)
    default:
      if constexpr (!WT) {
        switch (tile) {
          MADIFF_TILE(3, 128, 64)
          MADIFF_TILE(4, 96, 64)
          MADIFF_TILE(5, 64, 64)
          default: return -1;
        }
      } else {
        return -1;
      }
  }
#undef MADIFF_TILE
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// launched from madiff_gemm_fwd / madiff_gemm_dgrad_wt (gemm.hip) when dtype == bf16 and Cin % 64 == 0 (and Csplit % 64 == 0)
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_gemm_fwd_bf16(const void* params, int out_f32, int tile,
                                                                                 hipStream_t stream) {
  const GemmFwdParams& p = *(const GemmFwdParams*)params;
  if (p.wt) return out_f32 ? -1 : launch_tile<bf16_t, true>(p, tile, stream);
  return out_f32 ? launch_tile<float, false>(p, tile, stream) : launch_tile<bf16_t, false>(p, tile, stream);
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16: dW[o, tap*Cin + c] = sum_m dY[m, o] * X[src(m, tap), c]
// Same LDS-DMA structure.  Both operands arrive with the REDUCTION index (m) as the row, so the MFMA fragments
// (8 consecutive reduction elements per lane) are read with ds_read_b64_tr_b16, the hardware transposing LDS read.
// Tiles are [64 reduction rows][128 columns] bf16 = 256-byte rows; chunk swizzle of a row:
//   ch ^= ((row & 3) << 2) | ((row >> 2) & 3)     (conflict-free for the transposed reads of a 16x16x32 operand)
// applied on the DMA source side and on the read side.
// ------------------------------------------------------------------------------------------------
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
          int r = map_row(n, m - n * p.rm.T_out, tap, p.rm);
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
        else if (p.accumulate) *dst += acc[i][j][r];
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
