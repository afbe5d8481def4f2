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
//
// Pipeline: D LDS stages.  Step s waits (counted s_waitcnt vmcnt, never 0 in steady state) for its own DMA pieces of
// stage s, passes ONE raw s_barrier (now every wave's pieces of stage s have landed and every wave has finished
// reading stage s-1), re-issues stage s+D-1 into the buffer stage s-1 occupied, and multiplies stage s.  D-1 stages
// of global_load_lds stay in flight across the barrier, which is what hides the L2/HBM latency when a CU holds only
// one or two workgroups (small M).  All LDS lives in one __shared__ array and the loop contains no ordinary global
// load, so hipcc adds no vmcnt(0) of its own.
// Split-K: gridDim.z slices of the K-steps; slices combine with fp32 atomics (OutT = float, C pre-zeroed by the host).
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One LDS-DMA piece (64 lanes x 16 bytes: per-lane global address -> LDS [lds_addr + 16 * lane]).  Issued through inline
// asm on purpose: hipcc models __builtin_amdgcn_global_load_lds as a FLAT access that may touch LDS, and while one is
// outstanding (in these pipelines: always) its waitcnt pass turns every lgkmcnt wait into lgkmcnt(0), so each fragment
// read drained the whole LDS queue.  Hidden in asm, the DMA is ordered by the explicit vmcnt waits and barriers (all
// carry memory clobbers) and the ds_reads get exact counted waits.  lds_addr must be wave-uniform.
__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :
               : "s"(__builtin_amdgcn_readfirstlane(lds_addr)), "v"(gptr)
               : "memory");
}
__device__ __forceinline__ unsigned lds_u32(const void* p) { return (unsigned)(uintptr_t)(lptr_t)p; }

template <typename OutT, int BM, int BN, bool WT, int D>
__global__ __launch_bounds__(256, 2) void gemm_fwd_bf16_kernel(GemmFwdParams p) {
  static_assert(!WT || BN == 128, "the transposed-read image is laid out for 256-byte rows");
  constexpr int MI = BM / 32, NJ = BN / 32;          // 16x16 MFMA blocks per wave (rows, cols) = DMA pieces per wave
  constexpr int A_BYTES = BM * 128, BUF_BYTES = (BM + BN) * 128;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D * BUF_BYTES];   // [stage][A tile | W tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  xcd_chunk_remap(bx, by, p.xcd);
  const int m0 = bx * BM, n0 = by * BN;

  // DMA pieces: one wave instruction moves 1 KB = 8 rows x 8 chunks (16 B) of a 128-byte-row tile.  Wave w issues the
  // pieces for rows [BM/4*w, BM/4*(w+1)) of A and the matching quarter of W.
  const int pos = lane & 7;
  // Parity-major rows (p.par: transposed stride-2 maps -- ConvTranspose1d(k4, s2) forward, the strided convolution's
  // data gradient -- with an even T_out): tile row index m enumerates (parity, sequence, t / 2) instead of (sequence, t).
  // An output step t takes tap k only when (t + pad - k) is even, so a tile whose rows share one parity needs every
  // second tap only; the other taps' K-steps multiplied the zero page and are not run at all.  Sums are unchanged bit
  // for bit (the skipped products were +0).  A tile that straddles the parity boundary runs every tap, as before.
  const int Th = p.rm.T_out >> 1, half = p.M >> 1;
  auto row_nt = [&](int m, int& n, int& t) {
    if (p.par) {
      const int ph = m >= half ? 1 : 0, q = m - ph * half;
      n = q / Th;
      t = 2 * (q - n * Th) + ph;
    } else {
      n = m / p.rm.T_out;
      t = m - n * p.rm.T_out;
    }
  };
  int tap0 = 0, tap_step = 1, ntap = p.taps;
  if (p.par) {
    const int lo = m0 >= half ? 1 : 0, hi = (min(m0 + BM, p.M) - 1) >= half ? 1 : 0;
    if (lo == hi) {
      tap0 = (lo + p.rm.pad) & 1;
      tap_step = 2;
      ntap = (p.taps - tap0 + 1) >> 1;
    }
  }
  int a_n[MI], a_t[MI], a_chunk[MI];
  bool a_ok[MI];
#pragma unroll
  for (int j = 0; j < MI; j++) {
    int row = (BM / 4) * wave + 8 * j + (lane >> 3);
    a_chunk[j] = pos ^ (row & 7);          // source-side swizzle
    int m = m0 + row;
    a_ok[j] = m < p.M;
    row_nt(a_ok[j] ? m : 0, a_n[j], a_t[j]);
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
  int s_lo = 0, nsteps = spt * ntap;
  if (p.ksplit > 1) {                    // this workgroup's slice of the K-steps
    const int per = (nsteps + p.ksplit - 1) / p.ksplit;
    s_lo = blockIdx.z * per;
    nsteps = min(nsteps, s_lo + per) - s_lo;
    if (nsteps <= 0) return;             // uniform for the workgroup
  }

  // Per-lane source pointers of the A pieces for the CURRENT tap (channel 0 of each source; the zero page for rows the tap
  // takes outside the sequence), formed when the tap changes: a K-step then adds the chunk's byte offset only.  (With one
  // workgroup on the CU the loop is bound by instruction issue, and the 64-bit row arithmetic was a third of it.)
  const unsigned char* a_ptr[MI];
  unsigned a_live = 0;                     // bit j: piece j reads a real row
  int ptr_tap = -1, ptr_second = -1;       // what a_ptr currently describes
  auto tap_rows = [&](int tap) { (void)tap; };     // (pointers are formed on demand in issue())
  auto form_ptrs = [&](int tap, int second) {
    a_live = 0;
    const unsigned char* xb = (const unsigned char*)(second ? p.X2 : p.X);
    const int ld = second ? p.ldx2 : p.ldx;
#pragma unroll
    for (int j = 0; j < MI; j++) {
      const int r = a_ok[j] ? map_row(a_n[j], a_t[j], tap, p.rm) : -1;
      a_ptr[j] = r >= 0 ? xb + (size_t)r * ld * 2 + a_chunk[j] * 16 : g_zero_page + (lane & 7) * 16;
      if (r >= 0) a_live |= 1u << j;
    }
    ptr_tap = tap; ptr_second = second;
  };

  auto issue = [&](int buf, int tap, int cstep) {
    const int c0 = cstep << 6;
    const int second = c0 >= p.Csplit ? 1 : 0;
    if (tap != ptr_tap || second != ptr_second) form_ptrs(tap, second);      // workgroup-uniform, once per tap (and split)
    const unsigned cb = (unsigned)(second ? c0 - p.Csplit : c0) * 2;
    unsigned char* base = &smem[buf * BUF_BYTES];
#pragma unroll
    for (int j = 0; j < MI; j++) {
      const unsigned char* ga = a_ptr[j] + ((a_live >> j) & 1 ? cb : 0u);
      dma16(ga, lds_u32(base + (MI * wave + j) * 1024));
    }
    const size_t wk = WT ? ((size_t)c0 * p.ldw + (size_t)tap * p.wtap) * 2 : ((size_t)tap * p.Cin + c0) * 2;
#pragma unroll
    for (int j = 0; j < WP; j++) {
      const unsigned char* gw = w_src[j] ? w_src[j] + wk : g_zero_page + (lane & 15) * 16;
      dma16(gw, lds_u32(base + A_BYTES + (WP * wave + j) * 1024));
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
  constexpr int PIECES = MI + WP;        // DMA instructions per wave per stage
  int tap = s_lo / spt, cstep = s_lo - tap * spt;      // position of the NEXT stage to issue
  tap = tap0 + tap_step * tap;
  tap_rows(tap);
  int issued = 0, ibuf = 0;
  auto issue_next = [&]() {
    issue(ibuf, tap, cstep);
    ++issued;
    if (++ibuf == D) ibuf = 0;
    if (++cstep == spt) { cstep = 0; tap += tap_step; if (issued < nsteps) tap_rows(tap); }
  };
#pragma unroll
  for (int d = 0; d < D - 1; d++)
    if (d < nsteps) issue_next();
  int buf = 0;
  for (int s = 0; s < nsteps; s++) {
    if (s + D - 1 <= nsteps) wait_vmcnt<(D - 2) * PIECES>();   // stage s landed, D-2 younger stages still in flight
    else wait_vmcnt<0>();                                      // tail: fewer stages are outstanding
    __builtin_amdgcn_s_barrier();
    const unsigned char* As = &smem[buf * BUF_BYTES];
    const unsigned char* Ws = As + A_BYTES;
    if (++buf == D) buf = 0;
    auto load_frags = [&](bf16x8 (&a)[MI], bf16x8 (&b)[NJ], const int kh) {
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
    };
    // the fragment reads of this stage go out FIRST, the next stage's DMA (address arithmetic + 4..8 issue slots) is
    // issued under their latency: with one workgroup on the CU (small M) nothing else hides either
    bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
    load_frags(a0, b0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (issued < nsteps) issue_next();                         // into the buffer every wave finished reading in step s-1
    __builtin_amdgcn_sched_barrier(0);
    load_frags(a1, b1, 1);
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[i], b1[j], acc[i][j], 0, 0, 0);
  }

  OutT* C = (OutT*)p.C;
  if (p.vec_epi) {
    // Vector epilogue: each 16-row block of the wave tile goes through a wave-private fp32 LDS tile so that a lane
    // owns 8 consecutive columns of one row: bias / row-bias / activation on vectors, one 16-byte store per lane
    // (bf16) instead of eight 2-byte ones.  The host sets vec_epi only when every pointer / stride is 16-byte friendly.
    constexpr int WN = BN / 2, LDT = WN + 4, VPR = WN / 8, RPP = 64 / VPR;
    __syncthreads();                       // every wave is done with the operand tiles (no DMA is outstanding here)
    float* stage = (float*)smem + wave * (16 * LDT);
#pragma unroll
    for (int i = 0; i < MI; i++) {
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) stage[(fq * 4 + r) * LDT + j * 16 + fr] = acc[i][j][r];
#pragma unroll
      for (int pass = 0; pass < 16 / RPP; pass++) {
        const int row = pass * RPP + lane / VPR, cv = lane % VPR;
        const f32x4 v0 = *(const f32x4*)&stage[row * LDT + cv * 8];
        const f32x4 v1 = *(const f32x4*)&stage[row * LDT + cv * 8 + 4];
        const int m = m0 + wm * (BM / 2) + i * 16 + row;
        const int n = n0 + wn * WN + cv * 8;
        if (m >= p.M || n >= p.Npad) continue;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        int seq, tt;
        row_nt(m, seq, tt);
        const size_t nat = (size_t)seq * p.rm.T_out + tt;      // natural row index (== m unless parity-major)
        size_t orow = nat;
        if (p.permA > 0) {
          int b_ = seq / p.permA, a_ = seq - b_ * p.permA;
          orow = ((size_t)b_ * p.permT + tt) * p.permA + a_;
        }
        if (n + 8 <= p.Nout) {
          if (p.bias) {
            const f32x4 b0 = *(const f32x4*)&p.bias[n], b1 = *(const f32x4*)&p.bias[n + 4];
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
          }
          if (p.rowbias) {
            const float* rb = p.rowbias + (size_t)seq * p.ldrb + n;
            const f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
          }
        } else {                              // the vector straddles / lies beyond the real columns: pad columns are 0
#pragma unroll
          for (int e = 0; e < 8; e++) {
            if (n + e < p.Nout) {
              if (p.bias) v[e] += p.bias[n + e];
              if (p.rowbias) v[e] += p.rowbias[(size_t)seq * p.ldrb + n + e];
            } else {
              v[e] = 0.f;
            }
          }
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = n + e < p.Nout ? mish_f(v[e]) : 0.f;
        }
        if (p.addend) {                       // (launcher: bf16 output, no row permutation; pad columns of the addend are 0)
          const bf16x8 ad = *(const bf16x8*)((const bf16_t*)p.addend + nat * p.lda + n);
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] += (float)ad[e];
        }
        OutT* dst = C + orow * p.ldc + n;
        if constexpr (sizeof(OutT) == 2) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] = (bf16_t)v[e];
          *(bf16x8*)dst = o;
        } else {
          *(f32x4*)dst = (f32x4){v[0], v[1], v[2], v[3]};
          *(f32x4*)(dst + 4) = (f32x4){v[4], v[5], v[6], v[7]};
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int m = m0 + wm * (BM / 2) + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
      int seq, tt;
      row_nt(m, seq, tt);
      size_t orow = (size_t)seq * p.rm.T_out + tt;
      if (p.permA > 0) {
        int b_ = seq / p.permA, a_ = seq - b_ * p.permA;
        orow = ((size_t)b_ * p.permT + tt) * p.permA + a_;
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        int n = n0 + wn * (BN / 2) + j * 16 + fr;
        if (n >= p.Npad) continue;
        float v = 0.f;
        if (n < p.Nout) {
          v = acc[i][j][r];
          if (p.ksplit <= 1 || blockIdx.z == 0) {
            if (p.bias) v += p.bias[n];
            if (p.rowbias) v += p.rowbias[(size_t)seq * p.ldrb + n];
          }
          if (p.act == 1) v = fmaxf(v, 0.f);
          else if (p.act == 2) v = mish_f(v);
        }
        if constexpr (sizeof(OutT) == 4) {
          if (p.ksplit > 1) { if (n < p.Nout) atomicAdd((float*)&C[orow * p.ldc + n], v); continue; }
        }
        C[orow * p.ldc + n] = from_f32<OutT>(v);
      }
    }
  }
}

// Kernel variants the launcher can pick (index = the `tile` argument; the caller chooses per layer shape):
//   id  0..5 : 128x128, 96x128, 64x128, 128x64, 96x64, 64x64 with 2 LDS stages
//   id  6..11: the same shapes with 3 stages
//   id 12, 13: 64x128 and 64x64 with 4 stages
// WT kernels exist for the BN = 128 shapes only.
template <typename OutT, bool WT>
static int launch_tile(const GemmFwdParams& p, int tile, hipStream_t stream) {
#define MADIFF_TILE(ID, BM_, BN_, D_)                                                                                \
  case ID: {                                                                                                         \
    dim3 grid((p.M + BM_ - 1) / BM_, (p.Npad + BN_ - 1) / BN_, p.ksplit > 1 ? p.ksplit : 1);                          \
    hipLaunchKernelGGL((gemm_fwd_bf16_kernel<OutT, BM_, BN_, WT, D_>), grid, dim3(256), 0, stream, p);                \
    break;                                                                                                           \
  }
  switch (tile) {
    MADIFF_TILE(0, 128, 128, 2)
    MADIFF_TILE(1, 96, 128, 2)
    MADIFF_TILE(2, 64, 128, 2)
    MADIFF_TILE(6, 128, 128, 3)
    MADIFF_TILE(7, 96, 128, 3)
    MADIFF_TILE(8, 64, 128, 3)
    MADIFF_TILE(12, 64, 128, 4)
    default:
      if constexpr (!WT) {
        switch (tile) {
          MADIFF_TILE(3, 128, 64, 2)
          MADIFF_TILE(4, 96, 64, 2)
          MADIFF_TILE(5, 64, 64, 2)
          MADIFF_TILE(9, 128, 64, 3)
          MADIFF_TILE(10, 96, 64, 3)
          MADIFF_TILE(11, 64, 64, 3)
          MADIFF_TILE(13, 64, 64, 4)
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
  GemmFwdParams p = *(const GemmFwdParams*)params;
  {   // vector epilogue eligibility: 8-column vectors never straddle Npad, all rows / operands 16-byte aligned
    const int es = out_f32 ? 4 : 2;
    p.vec_epi = p.ksplit <= 1 && p.Npad % 8 == 0 && ((size_t)p.C % 16) == 0 && ((size_t)p.ldc * es) % 16 == 0 &&
                (!p.bias || (size_t)p.bias % 16 == 0) &&
                (!p.rowbias || ((size_t)p.rowbias % 16 == 0 && p.ldrb % 4 == 0));
    if (p.addend && (!p.vec_epi || out_f32 || p.permA > 0)) return 1;      // the addend exists in the vector epilogue only
  }
  {   // XCD-aware tile order (common.h xcd_chunk_remap): keep the LARGER operand's tiles together in one L2
    static int on = -1;
    if (on < 0) { const char* e = getenv("MADIFF_GEMM_XCD"); on = e ? atoi(e) : 1; }
    const double xb = (double)p.M, wb = (double)p.Npad;            // bytes ~ rows x K on both sides
    // (measured: mode 1 for weight-heavy small-M layers looks 10 % faster when one layer is launched back to back -- its
    // weight tiles then stay in "their" L2 from launch to launch -- and is 10 % SLOWER inside the evaluator's real loop,
    // where every layer's weights arrive cold: 184 -> 205 ms per planning step.  So only the activation-heavy case.)
    p.xcd = on == 0 ? 0 : (on == 1 ? (xb >= 4 * wb ? 2 : 0) : on - 1);   // MADIFF_GEMM_XCD=2 / 3 force mode 1 / 2
  }
  {   // parity-major rows for transposed stride-2 maps (see the kernel)
    static int on = -1;
    if (on < 0) { const char* e = getenv("MADIFF_GEMM_PARITY"); on = e ? atoi(e) : 1; }
    p.par = on && p.rm.tr && p.rm.stride == 2 && p.rm.T_out % 2 == 0 && p.M % p.rm.T_out == 0 && p.taps >= 2;
  }
  if (p.wt && out_f32) {     // split-K data gradient: one variant, fp32 atomics
    if (tile != 8) return -1;
    dim3 grid((p.M + 63) / 64, (p.Npad + 127) / 128, p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((gemm_fwd_bf16_kernel<float, 64, 128, true, 3>), grid, dim3(256), 0, stream, p);
    MADIFF_CHECK_LAUNCH();
    return 0;
  }
  if (p.wt) return launch_tile<bf16_t, true>(p, tile, stream);
  return out_f32 ? launch_tile<float, false>(p, tile, stream) : launch_tile<bf16_t, false>(p, tile, stream);
}

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16: dW[o, tap*Cin + c] (+)= sum_m dY[m, o] * X[src(m, tap), c]
// Same LDS-DMA structure.  Both operands arrive with the REDUCTION index (m) as the row, so the MFMA fragments
// (8 consecutive reduction elements per lane) are read with ds_read_b64_tr_b16, the hardware transposing LDS read.
// Output tile BO x BC (each 64 or 128), 4 waves as 2 x 2; operand tiles are [64 reduction rows][BO or BC columns]:
//   128 columns = 256-byte rows of 16 chunks, chunk swizzle  ch ^= ((row & 3) << 2) | ((row >> 2) & 3)
//    64 columns = 128-byte rows of  8 chunks, chunk swizzle  ch ^= (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1
// (both conflict-free for the transposed reads of a 16x16x32 operand: the 8 rows a half-wave touches land on 8
// different 32-byte bank slots), applied on the DMA source side and on the read side.
// ------------------------------------------------------------------------------------------------
template <int COLS> __device__ __forceinline__ int tr_swz(int row) {
  if constexpr (COLS == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
  else return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
}

template <int BO, int BC>
__global__ __launch_bounds__(256, 2) void gemm_wgrad_bf16_kernel(GemmWgradParams p, unsigned magic) {
  constexpr int KR = 64;
  constexpr int MI = BO / 32, NJ = BC / 32;
  constexpr int Y_BYTES = KR * BO * 2, X_BYTES = KR * BC * 2, BUF_BYTES = Y_BYTES + X_BYTES;
  constexpr int YCH = BO / 8, XCH = BC / 8;              // 16-byte chunks per tile row
  constexpr int YP = YCH / 4, XP = XCH / 4;              // DMA pieces (1 KB) per wave: wave w fills rows [16w, 16w+16)
  constexpr int YRP = 64 / YCH, XRP = 64 / XCH;          // rows per piece
  // D LDS stages ([dY tile | X tile] each): three for every tile but 128x128 (two workgroups per CU must still fit).
  // Same pipeline as the all-taps kernel below: counted vmcnt wait for the stage's own pieces, one raw barrier, the
  // stage D-1 ahead re-issued into the buffer every wave has just finished with.  The small layers this kernel serves
  // run a handful of K-steps per workgroup and are bound by the DMA latency of each step, not by its MFMAs.
  constexpr int D = (BO + BC >= 256) ? 2 : 3;
  constexpr int PIECES = YP + XP;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * BO;
  const int tap = blockIdx.y / p.ctiles;
  const int c0 = (blockIdx.y - tap * p.ctiles) * BC;
  const int total_steps = (p.M + KR - 1) / KR;
  const int s_begin = blockIdx.z * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  if (s_begin >= s_end) return;

  const unsigned char* xsrc = (const unsigned char*)p.X;
  int ldx = p.ldx, cx = c0, cmax = p.Csplit;
  if (c0 >= p.Csplit) { xsrc = (const unsigned char*)p.X2; ldx = p.ldx2; cx = c0 - p.Csplit; cmax = p.Cin - p.Csplit; }

  int yrow[YP], ych[YP], xrow[XP], xch[XP];
  bool yok[YP], xok[XP];
#pragma unroll
  for (int j = 0; j < YP; j++) {
    yrow[j] = 16 * wave + YRP * j + lane / YCH;
    ych[j] = (lane % YCH) ^ tr_swz<BO>(yrow[j]);
    yok[j] = n0 + ych[j] * 8 < p.Ycols;
  }
#pragma unroll
  for (int j = 0; j < XP; j++) {
    xrow[j] = 16 * wave + XRP * j + lane / XCH;
    xch[j] = (lane % XCH) ^ tr_swz<BC>(xrow[j]);
    xok[j] = cx + xch[j] * 8 < cmax;
  }
  const unsigned char* zero = g_zero_page + (lane & 15) * 16;

  auto issue = [&](int buf, int s) {
    unsigned char* base = &smem[buf * BUF_BYTES];
#pragma unroll
    for (int j = 0; j < YP; j++) {
      int m = s * KR + yrow[j];
      const unsigned char* gy = zero;
      if (m < p.M && yok[j]) gy = (const unsigned char*)p.dY + ((size_t)m * p.ldy + n0 + ych[j] * 8) * 2;
      dma16(gy, lds_u32(base + (YP * wave + j) * 1024));
    }
#pragma unroll
    for (int j = 0; j < XP; j++) {
      int m = s * KR + xrow[j];
      const unsigned char* gx = zero;
      if (m < p.M && xok[j]) {
        int n = magic ? (int)__umulhi((unsigned)m, magic) : m;   // magic == 0 <=> T_out == 1
        int r = map_row(n, m - n * p.rm.T_out, tap, p.rm);
        if (r >= 0) gx = xsrc + ((size_t)r * ldx + cx + xch[j] * 8) * 2;
      }
      dma16(gx, lds_u32(base + Y_BYTES + (XP * wave + j) * 1024));
    }
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Bias gradient = column sums of dY = dY^T * 1: the waves of the (tap 0, first channel tile) workgroups that hold the
  // first X half run one more MFMA per dY fragment against an all-ones operand (every output column then holds the sum).
  const bool do_bias = (p.dbias || p.btab) && blockIdx.y == 0 && wc == 0;
  f32x4 accb[MI];
#pragma unroll
  for (int i = 0; i < MI; i++) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (bf16_t)1.0f;

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int nsteps = s_end - s_begin;
  int issued = 0, ibuf = 0, buf = 0;
#pragma unroll
  for (int d = 0; d < D - 1; d++)
    if (d < nsteps) { issue(ibuf, s_begin + issued); ++issued; if (++ibuf == D) ibuf = 0; }
  for (int s = 0; s < nsteps; s++) {
    if (s + D - 1 <= nsteps) wait_vmcnt<(D - 2) * PIECES>();   // stage s landed, D-2 younger stages still in flight
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (issued < nsteps) { issue(ibuf, s_begin + issued); ++issued; if (++ibuf == D) ibuf = 0; }
    const unsigned char* Ys = &smem[buf * BUF_BYTES];
    if (++buf == D) buf = 0;
    const unsigned char* Xs = Ys + Y_BYTES;
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      bf16x8 a[MI], b[NJ];
      const int r0 = 32 * kb + 8 * fq + tq, r1 = r0 + 4;
      const int o = 8 * (tp & 1);
#pragma unroll
      for (int i = 0; i < MI; i++) {
        const int ca = (YCH / 2) * wn + 2 * i + (tp >> 1);
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ys + r0 * (BO * 2) + ((ca ^ tr_swz<BO>(r0)) << 4) + o));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ys + r1 * (BO * 2) + ((ca ^ tr_swz<BO>(r1)) << 4) + o));
        a[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int cb = (XCH / 2) * wc + 2 * j + (tp >> 1);
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xs + r0 * (BC * 2) + ((cb ^ tr_swz<BC>(r0)) << 4) + o));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xs + r1 * (BC * 2) + ((cb ^ tr_swz<BC>(r1)) << 4) + o));
        b[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < MI; i++) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    }
  }

  if (do_bias && fr == 0) {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int n = n0 + wn * (BO / 2) + i * 16 + fq * 4 + r;
        if (n < p.Nout) atomicAdd(p.btab ? p.btab[n] : p.dbias + n, accb[i][r]);
      }
  }
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int n = n0 + wn * (BO / 2) + i * 16 + fq * 4 + r;
      if (n >= p.Nout) continue;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        int c = c0 + wc * (BC / 2) + j * 16 + fr;
        if (c >= p.Cin) continue;
        float* dst = p.dW + (size_t)n * p.ldw + (size_t)tap * p.Cin + c;
        if (p.splits > 1) atomicAdd(dst, acc[i][j][r]);
        else if (p.accumulate) *dst += acc[i][j][r];
        else *dst = acc[i][j][r];
      }
    }
}

// tile: 0 = 128x128, 1 = 128x64 (O x C), 2 = 64x128, 3 = 64x64.  KR = 64 reduction rows per step (the generic kernel
// uses 32): the caller sizes steps_per_split for 64 and the grid for the tile.
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_gemm_wgrad_bf16(const void* params, int tile, int gx, int gy,
                                                                                   int gz, hipStream_t stream) {
  GemmWgradParams p = *(const GemmWgradParams*)params;
  unsigned d = (unsigned)p.rm.T_out;
  unsigned magic = d == 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d);
  dim3 grid(gx, gy, gz);
  switch (tile) {
    case 0: hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<128, 128>), grid, dim3(256), 0, stream, p, magic); break;
    case 1: hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<128, 64>), grid, dim3(256), 0, stream, p, magic); break;
    case 2: hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<64, 128>), grid, dim3(256), 0, stream, p, magic); break;
    case 3: hipLaunchKernelGGL((gemm_wgrad_bf16_kernel<64, 64>), grid, dim3(256), 0, stream, p, magic); break;
    default: return -1;
  }
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient of a stride-1 "same" convolution, ALL TAPS IN ONE PASS (bf16):
//   dW[o, k*Cin + c] (+)= sum_{n,t} dY[(n,t), o] * X[(n, t + k - pad), c]      k = 0..TAPS-1, rows outside the sequence = 0
// The per-tap kernel above re-reads the same dY tile and (row-shifted) the same X tile once per tap.  Here a workgroup
// owns a BO x BC tile of (o, c) for every tap: one K-step stages KR = 32*NQ rows = a whole number of sequences of dY
// and X ONCE, and the tap shift happens in LDS: ds_read_b64_tr_b16 takes a row address per lane, so the X fragment of
// tap k is read from rows (m + k - pad), and a lane whose (m, k) pair crosses a sequence boundary points at a
// zero row kept in LDS instead.  No masking VALU, no row arithmetic in the loop (all LDS offsets are loop-invariant
// registers), 1/TAPS of the DMA traffic and TAPS x the MFMAs per staged byte.
// NQ: 32-row chunks per K-step; requires (32*NQ) % T == 0 (T in {2,4,8,16,32} -> NQ 1; T in {3,6,12,24,...} -> NQ 3).
// ------------------------------------------------------------------------------------------------
template <int BO, int BC, int TAPS, int NQ, int D>
__global__ __launch_bounds__(256, (BO >= 128 ? 1 : 2)) void gemm_wgrad_taps_bf16_kernel(GemmWgradParams p) {
  static_assert(BC == 64, "X tile: 128-byte rows");
  constexpr int KR = 32 * NQ;
  constexpr int MI = BO / 32, NJ = BC / 32;
  constexpr int Y_BYTES = KR * BO * 2, X_BYTES = KR * BC * 2;
  constexpr int ZERO_OFF = Y_BYTES + X_BYTES;            // 256 zero bytes behind the tiles of every stage
  constexpr int BUF_BYTES = ZERO_OFF + 1024;             // (stages stay 1 KB aligned for the DMA pieces)
  constexpr int YCH = BO / 8;                            // chunks per dY row (8 or 16)
  constexpr int YRP = 64 / YCH;                          // rows per DMA piece
  constexpr int YP = KR / YRP / 4, XP = KR / 8 / 4;      // pieces per wave per stage
  static_assert((KR / YRP) % 4 == 0 && (KR / 8) % 4 == 0, "pieces must split evenly over 4 waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wc = wave & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  xcd_remap(bx, by, p.xcd);
  const int n0 = bx * BO, c0 = by * BC;
  const int T = p.rm.T_out, pad = p.rm.pad;
  const int total_steps = (p.M + KR - 1) / KR;
  const int s_begin = blockIdx.z * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  if (s_begin >= s_end) return;
  const int nsteps = s_end - s_begin;
  if (tid < 64 * D) ((float*)(smem + (tid >> 6) * BUF_BYTES + ZERO_OFF))[tid & 63] = 0.f;   // 256 zero bytes per stage
  __syncthreads();

  const unsigned char* xsrc = (const unsigned char*)p.X;
  int ldx = p.ldx, cx = c0, cmax = p.Csplit;
  if (c0 >= p.Csplit) { xsrc = (const unsigned char*)p.X2; ldx = p.ldx2; cx = c0 - p.Csplit; cmax = p.Cin - p.Csplit; }

  // DMA pieces: piece q of a tile covers rows [q*RP, (q+1)*RP); wave w issues pieces w*P .. w*P + P-1.  The source row of
  // K-step row r is simply step*KR + r for both operands: the tap shift is applied when the fragments are read.
  int yrow[YP], xrow[XP];
  const unsigned char* ysrc0[YP];
  const unsigned char* xsrc0[XP];
#pragma unroll
  for (int j = 0; j < YP; j++) {
    yrow[j] = (YP * wave + j) * YRP + lane / YCH;
    int ch = (lane % YCH) ^ tr_swz<BO>(yrow[j]);
    ysrc0[j] = n0 + ch * 8 < p.Ycols ? (const unsigned char*)p.dY + ((size_t)yrow[j] * p.ldy + n0 + ch * 8) * 2 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < XP; j++) {
    xrow[j] = (XP * wave + j) * 8 + (lane >> 3);
    int ch = (lane & 7) ^ tr_swz<BC>(xrow[j]);
    xsrc0[j] = cx + ch * 8 < cmax ? xsrc + ((size_t)xrow[j] * ldx + cx + ch * 8) * 2 : nullptr;
  }
  const unsigned char* zero = g_zero_page + (lane & 15) * 16;
  auto issue = [&](int buf, int s) {
    unsigned char* base = &smem[buf * BUF_BYTES];
    const size_t m0 = (size_t)s * KR;
#pragma unroll
    for (int j = 0; j < YP; j++) {
      const unsigned char* g = (ysrc0[j] && m0 + yrow[j] < (size_t)p.M) ? ysrc0[j] + m0 * p.ldy * 2 : zero;
      dma16(g, lds_u32(base + (YP * wave + j) * 1024));
    }
#pragma unroll
    for (int j = 0; j < XP; j++) {
      const unsigned char* g = (xsrc0[j] && m0 + xrow[j] < (size_t)p.M) ? xsrc0[j] + m0 * ldx * 2 : zero;
      dma16(g, lds_u32(base + Y_BYTES + (XP * wave + j) * 1024));
    }
  };

  // loop-invariant, stage-relative LDS offsets of the fragment reads (the MFMA block index i / j is XORed in: it only
  // touches chunk-index bit 1, which the swizzle XORs too, and XOR commutes)
  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  const int o8 = 8 * (tp & 1);
  int yoff[NQ][2], xoff[NQ][2][TAPS];
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int r = 32 * q + 8 * fq + tq + 4 * h;          // row of the K-step (the same for every step)
      yoff[q][h] = r * (BO * 2) + ((((YCH / 2) * wn + (tp >> 1)) ^ tr_swz<BO>(r)) << 4) + o8;
      const int t = r % T;
#pragma unroll
      for (int k = 0; k < TAPS; k++) {
        const int ts = t + k - pad, rs = r + k - pad;       // KR % T == 0: a valid shifted row stays inside the K-step
        // (a lane whose shifted row leaves its sequence reads zeros placed at the SAME bank position the real row would
        // have had: a fixed zero address would collide with the valid lanes of the same half-wave on its banks)
        const int nat = rs * (BC * 2) + (((4 * wc + (tp >> 1)) ^ tr_swz<BC>(rs)) << 4) + o8;
        xoff[q][h][k] = (ts >= 0 && ts < T) ? Y_BYTES + nat : ZERO_OFF + (nat & 255);
      }
    }

  f32x4 acc[TAPS][MI][NJ];
#pragma unroll
  for (int k = 0; k < TAPS; k++)
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[k][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  constexpr int PIECES = YP + XP;
  int issued = 0, ibuf = 0;
#pragma unroll
  for (int d = 0; d < D - 1; d++)
    if (d < nsteps) { issue(ibuf, s_begin + issued); ++issued; if (++ibuf == D) ibuf = 0; }
  int buf = 0;
  for (int s = 0; s < nsteps; s++) {
    if (s + D - 1 <= nsteps) wait_vmcnt<(D - 2) * PIECES>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (issued < nsteps) { issue(ibuf, s_begin + issued); ++issued; if (++ibuf == D) ibuf = 0; }
    const unsigned char* B0 = &smem[buf * BUF_BYTES];
    if (++buf == D) buf = 0;
    // Fragment pipeline inside a stage: the X fragments of tap k+1 (or, after the last tap, the dY and tap-0 fragments of
    // the next 32-row chunk) are read while the MFMAs of tap k run.  sched_barrier(0) keeps hipcc from sinking each
    // read down to its first use, which would expose one LDS latency per group of MI*NJ MFMAs.
    auto load_a = [&](bf16x8 (&a)[MI], const int q) {
#pragma unroll
      for (int i = 0; i < MI; i++) {
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(B0 + (yoff[q][0] ^ (i << 5))));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(B0 + (yoff[q][1] ^ (i << 5))));
        a[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    auto load_b = [&](bf16x8 (&b)[NJ], const int q, const int k) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(B0 + (xoff[q][0][k] ^ (j << 5))));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(B0 + (xoff[q][1][k] ^ (j << 5))));
        b[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    bf16x8 a[2][MI], b[2][NJ];
    load_a(a[0], 0);
    load_b(b[0], 0, 0);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
#pragma unroll
      for (int k = 0; k < TAPS; k++) {
        const int t = q * TAPS + k;                 // position in the stage: fragments alternate between two register sets
        __builtin_amdgcn_sched_barrier(0);
        if (k + 1 < TAPS) {
          load_b(b[(t + 1) & 1], q, k + 1);
        } else if (q + 1 < NQ) {
          load_a(a[(q + 1) & 1], q + 1);
          load_b(b[(t + 1) & 1], q + 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++)
            acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q & 1][i], b[t & 1][j], acc[k][i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

#pragma unroll
  for (int k = 0; k < TAPS; k++)
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int n = n0 + wn * (BO / 2) + i * 16 + fq * 4 + r;
        if (n >= p.Nout) continue;
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          int c = c0 + wc * (BC / 2) + j * 16 + fr;
          if (c >= p.Cin) continue;
          float* dst = p.dW + (size_t)n * p.ldw + (size_t)k * p.Cin + c;
          if (p.splits > 1) atomicAdd(dst, acc[k][i][j][r]);
          else if (p.accumulate) *dst += acc[k][i][j][r];
          else *dst = acc[k][i][j][r];
        }
      }
}

// ------------------------------------------------------------------------------------------------
// All-taps weight gradient, 8-wave form (variants 8, 9): the same 64 x 64 x TAPS output tile per workgroup, but the
// reduction rows of every K-step are split over TWO groups of four waves inside the workgroup (group g takes the 32-row
// chunks [g*NQ, (g+1)*NQ) of a K-step of KR = 64*NQ rows), and the two partial tiles are combined through LDS at the end.
// Why: with one 4-wave workgroup per tile a 1024 x 1024 k5 layer has only 256 tiles, so the 4-wave kernel needs a
// split-K of 2 to give every SIMD two waves -- and then every output element is added with an fp32 atomic twice
// (2 x 21 MB per layer at the ~1.3 TB/s chip-wide atomic rate, issued by all workgroups at the same moment at the end:
// a ~30 us tail on an 85 us kernel).  Here 256 workgroups of 8 waves fill the chip without a split, and the tile is
// added to the live gradient by plain read-modify-write 64-byte row segments (splits == 1) -- no atomics at all.
//
// TM (time-major K order, T = 6 only): the order of the reduction rows inside a K-step is free, so the 192 rows = 32
// sequences x 6 steps are staged as (t, sequence): LDS row = 32 t + s, i.e. 32-row chunk t holds time step t of all 32
// sequences.  Then dW[:, k, :] += sum_t Y_t^T X_(t+k-2) over WHOLE chunks: the X fragment of (chunk t, tap k) is chunk
// t+k-2 unshifted, the pairs whose t+k-2 leaves [0, 6) are not computed at all (24 of 30 remain), and each X chunk is
// read from LDS once per K-step for all the taps it serves (8 fragment sets per wave and K-step instead of 18).  Group 0
// takes t = 0, 1, 2 and X chunks 0..4; group 1 the mirror image (t = 5, 4, 3, chunks 5..1), so both run the same code
// with accumulator a holding tap a (group 0) or tap 4 - a (group 1, swapped back before the exchange).
// ------------------------------------------------------------------------------------------------
template <int TAPS, int NQ, int D, bool TM = false>
__global__ __launch_bounds__(512) void gemm_wgrad_taps8_bf16_kernel(GemmWgradParams p) {
  static_assert(!TM || (TAPS == 5 && NQ == 3), "time-major K order: T = 6, k = 5");
  constexpr int BO = 64, BC = 64, KG = 2;
  constexpr int KR = 32 * NQ * KG;
  constexpr int MI = BO / 32, NJ = BC / 32;
  constexpr int Y_BYTES = KR * BO * 2, X_BYTES = KR * BC * 2;
  constexpr int ZERO_OFF = Y_BYTES + X_BYTES;
  constexpr int BUF_BYTES = ZERO_OFF + 1024;
  constexpr int NW = 4 * KG;
  constexpr int YCH = BO / 8, YRP = 64 / YCH;               // chunks per dY row, rows per 1 KB DMA piece
  constexpr int YP = KR / YRP / NW, XP = KR / 8 / NW;       // pieces per wave per stage
  static_assert((KR / YRP) % NW == 0 && (KR / 8) % NW == 0, "pieces must split evenly over the waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, w4 = wave & 3;
  const int wn = w4 >> 1, wc = w4 & 1;
  int bx = blockIdx.x, by = blockIdx.y;
  xcd_remap(bx, by, p.xcd);
  const int n0 = bx * BO, c0 = by * BC;
  const int T = p.rm.T_out, pad = p.rm.pad;
  const int total_steps = (p.M + KR - 1) / KR;
  const int s_begin = blockIdx.z * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  if (s_begin >= s_end) return;
  const int nsteps = s_end - s_begin;
  if (tid < 64 * D) ((float*)(smem + (tid >> 6) * BUF_BYTES + ZERO_OFF))[tid & 63] = 0.f;   // 256 zero bytes per stage
  __syncthreads();

  const unsigned char* xsrc = (const unsigned char*)p.X;
  int ldx = p.ldx, cx = c0, cmax = p.Csplit;
  if (c0 >= p.Csplit) { xsrc = (const unsigned char*)p.X2; ldx = p.ldx2; cx = c0 - p.Csplit; cmax = p.Cin - p.Csplit; }

  int yrow[YP], xrow[XP];                  // row offset of the piece's lane from the first row of the K-step (global side)
  const unsigned char* ysrc0[YP];
  const unsigned char* xsrc0[XP];
#pragma unroll
  for (int j = 0; j < YP; j++) {
    const int lr = (YP * wave + j) * YRP + lane / YCH;                   // LDS row
    int ch = (lane % YCH) ^ tr_swz<BO>(lr);
    yrow[j] = TM ? (lr & 31) * T + (lr >> 5) : lr;
    ysrc0[j] = n0 + ch * 8 < p.Ycols ? (const unsigned char*)p.dY + ((size_t)yrow[j] * p.ldy + n0 + ch * 8) * 2 : nullptr;
  }
#pragma unroll
  for (int j = 0; j < XP; j++) {
    const int lr = (XP * wave + j) * 8 + (lane >> 3);
    int ch = (lane & 7) ^ tr_swz<BC>(lr);
    xrow[j] = TM ? (lr & 31) * T + (lr >> 5) : lr;
    xsrc0[j] = cx + ch * 8 < cmax ? xsrc + ((size_t)xrow[j] * ldx + cx + ch * 8) * 2 : nullptr;
  }
  const unsigned char* zero = g_zero_page + (lane & 15) * 16;
  auto issue = [&](int buf, int s) {
    unsigned char* base = &smem[buf * BUF_BYTES];
    const size_t m0 = (size_t)s * KR;
#pragma unroll
    for (int j = 0; j < YP; j++) {
      const unsigned char* gsrc = (ysrc0[j] && m0 + yrow[j] < (size_t)p.M) ? ysrc0[j] + m0 * p.ldy * 2 : zero;
      dma16(gsrc, lds_u32(base + (YP * wave + j) * 1024));
    }
#pragma unroll
    for (int j = 0; j < XP; j++) {
      const unsigned char* gsrc = (xsrc0[j] && m0 + xrow[j] < (size_t)p.M) ? xsrc0[j] + m0 * ldx * 2 : zero;
      dma16(gsrc, lds_u32(base + Y_BYTES + (XP * wave + j) * 1024));
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  const int o8 = 8 * (tp & 1);
  // loop-invariant LDS offsets of the fragment reads, two 16-bit offsets (h = 0 | h = 1 << 16) per register: dY offsets are
  // relative to the stage, X offsets to the X tile (the zero block sits right behind it: X_BYTES + 255 < 65536)
  static_assert(Y_BYTES < 65536 && X_BYTES + 256 < 65536, "16-bit fragment offsets");
  unsigned yoff[NQ], xoff[NQ][TAPS];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    unsigned yo[2], xo[2][TAPS];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int cq = TM ? (g ? 2 * NQ - 1 - q : q) : g * NQ + q;   // TM: chunk = time step, group 1 mirrored
      const int r = 32 * cq + 8 * fq + tq + 4 * h;              // this wave group's rows of the K-step
      yo[h] = r * (BO * 2) + ((((YCH / 2) * wn + (tp >> 1)) ^ tr_swz<BO>(r)) << 4) + o8;
      const int t = r % T;
#pragma unroll
      for (int k = 0; k < TAPS; k++) {
        const int ts = t + k - pad, rs = r + k - pad;           // KR % T == 0: a valid shifted row stays inside the K-step
        // (a lane whose shifted row leaves its sequence reads zeros placed at the SAME bank position the real row would
        // have had: a fixed zero address would collide with the valid lanes of the same half-wave on its banks)
        const int nat = rs * (BC * 2) + (((4 * wc + (tp >> 1)) ^ tr_swz<BC>(rs)) << 4) + o8;
        xo[h][k] = (ts >= 0 && ts < T) ? nat : X_BYTES + (nat & 255);
      }
    }
    yoff[q] = yo[0] | (yo[1] << 16);
#pragma unroll
    for (int k = 0; k < TAPS; k++) xoff[q][k] = xo[0][k] | (xo[1][k] << 16);
  }
  if constexpr (TM) {                      // xoff[0][u']: X chunk u = u' (group 0) or 5 - u' (group 1), unshifted
#pragma unroll
    for (int uu = 0; uu < TAPS; uu++) {
      unsigned xo[2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int rs = 32 * (g ? 2 * NQ - 1 - uu : uu) + 8 * fq + tq + 4 * h;
        xo[h] = rs * (BC * 2) + (((4 * wc + (tp >> 1)) ^ tr_swz<BC>(rs)) << 4) + o8;
      }
      xoff[0][uu] = xo[0] | (xo[1] << 16);
    }
  }

  f32x4 acc[TAPS][MI][NJ];
#pragma unroll
  for (int k = 0; k < TAPS; k++)
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[k][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  constexpr int PIECES = YP + XP;
  int issued = 0, ibuf = 0;
#pragma unroll
  for (int d = 0; d < D - 1; d++)
    if (d < nsteps) { issue(ibuf, s_begin + issued); ++issued; if (++ibuf == D) ibuf = 0; }
  int buf = 0;
  for (int s = 0; s < nsteps; s++) {
    if (s + D - 1 <= nsteps) wait_vmcnt<(D - 2) * PIECES>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (issued < nsteps) { issue(ibuf, s_begin + issued); ++issued; if (++ibuf == D) ibuf = 0; }
    const unsigned char* B0 = &smem[buf * BUF_BYTES];
    if (++buf == D) buf = 0;
    const unsigned char* BX = B0 + Y_BYTES;
    auto load_a = [&](bf16x8 (&a)[MI], const int q) {
      unsigned pk = yoff[q];
      asm volatile("" : "+v"(pk));          // keep the unpacked addresses out of the loop-invariant set (registers)
      const unsigned o0 = pk & 0xffffu, o1 = pk >> 16;
#pragma unroll
      for (int i = 0; i < MI; i++) {
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(B0 + (o0 ^ (i << 5))));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(B0 + (o1 ^ (i << 5))));
        a[i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    auto load_b = [&](bf16x8 (&b)[NJ], const int q, const int k) {
      unsigned pk = xoff[q][k];
      asm volatile("" : "+v"(pk));
      const unsigned o0 = pk & 0xffffu, o1 = pk >> 16;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(BX + (o0 ^ (j << 5))));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(BX + (o1 ^ (j << 5))));
        b[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    // fragment pipeline inside a stage, two positions deep: while the 4 MFMAs of position t = (chunk q, tap k) run, the X
    // fragments of position t+2 are being read (three register sets); 4 MFMAs = 64 cycles alone do not cover one LDS
    // round trip under load.  The dY fragments of chunk q+1 are read at the same distance.
    if constexpr (TM) {
      // all three dY chunks of the group stay in registers; X chunks stream through a 3-deep register ring
      bf16x8 ya[NQ][MI], xb[3][NJ];
      load_a(ya[0], 0);
      load_b(xb[0], 0, 0);
      load_a(ya[1], 1);
      load_a(ya[2], 2);
      load_b(xb[1], 0, 1);
#pragma unroll
      for (int uu = 0; uu < TAPS; uu++) {
        __builtin_amdgcn_sched_barrier(0);
        if (uu + 2 < TAPS) load_b(xb[(uu + 2) % 3], 0, uu + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int a = uu - q + 2;                       // accumulator (= tap for group 0, 4 - tap for group 1)
          if (a < 0 || a >= TAPS) continue;
#pragma unroll
          for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++)
              acc[a][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ya[q][i], xb[uu % 3][j], acc[a][i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      continue;
    }
    bf16x8 a[2][MI], b[3][NJ];
    load_a(a[0], 0);
    load_b(b[0], 0, 0);
    load_b(b[1], 0, 1);
#pragma unroll
    for (int t = 0; t < NQ * TAPS; t++) {
      const int q = t / TAPS, k = t % TAPS;
      const int t2 = t + 2, q2 = t2 / TAPS, k2 = t2 % TAPS;
      __builtin_amdgcn_sched_barrier(0);
      if (t2 < NQ * TAPS) {
        if (k2 == 0) load_a(a[q2 & 1], q2);
        load_b(b[t2 % 3], q2, k2);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
          acc[k][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q & 1][i], b[t % 3][j], acc[k][i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- combine the two groups' partial tiles through LDS.  The (tap, i) slices of a round are split between the groups;
  // each group hands the other one the slices it does not own, then adds what it received and updates global memory.
  // Rounds (taps [0, TSPLIT) then [TSPLIT, TAPS)) keep the exchange inside the staging memory.
  if constexpr (TM) {
    if (g) {                               // group 1 accumulated tap 4 - a in accumulator a
#pragma unroll
      for (int k = 0; k < TAPS / 2; k++)
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++) { const f32x4 tmp = acc[k][i][j]; acc[k][i][j] = acc[TAPS - 1 - k][i][j]; acc[TAPS - 1 - k][i][j] = tmp; }
    }
  }
  constexpr int TSPLIT = TAPS;
  static_assert(((TSPLIT * MI + 1) / 2) * 2 * NJ * 4 * 256 * 4 <= D * BUF_BYTES, "the partial-tile exchange reuses the staging memory");
  float* red = (float*)smem;
  const int t4 = tid & 255;
#pragma unroll
  for (int round = 0; round < (TSPLIT < TAPS ? 2 : 1); round++) {
    const int k0 = round == 0 ? 0 : TSPLIT, k1 = round == 0 ? TSPLIT : TAPS;
    const int NS = (k1 - k0) * MI, HALF = (NS + 1) / 2;     // slices 0 .. HALF-1 of the round belong to group 0
    __syncthreads();                                        // operand stages / the previous round's exchange are done with
#pragma unroll
    for (int k = 0; k < TAPS; k++)
#pragma unroll
      for (int i = 0; i < MI; i++) {
        if (k < k0 || k >= k1) continue;
        const int sl = (k - k0) * MI + i, owner = sl < HALF ? 0 : 1;
        if (owner == g) continue;
        const int slot = owner == 0 ? sl : sl - HALF;
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) red[((owner * HALF + slot) * NJ * 4 + j * 4 + r) * 256 + t4] = acc[k][i][j][r];
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TAPS; k++)
#pragma unroll
      for (int i = 0; i < MI; i++) {
        if (k < k0 || k >= k1) continue;
        const int sl = (k - k0) * MI + i, owner = sl < HALF ? 0 : 1;
        if (owner != g) continue;
        const int slot = owner == 0 ? sl : sl - HALF;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int n = n0 + wn * (BO / 2) + i * 16 + fq * 4 + r;
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            const int c = c0 + wc * (BC / 2) + j * 16 + fr;
            const float v = acc[k][i][j][r] + red[((owner * HALF + slot) * NJ * 4 + j * 4 + r) * 256 + t4];
            if (n >= p.Nout || c >= p.Cin) continue;
            float* dst = p.dW + (size_t)n * p.ldw + (size_t)k * p.Cin + c;
            if (p.splits > 1) atomicAdd(dst, v);
            else if (p.accumulate) *dst += v;
            else *dst = v;
          }
        }
      }
  }
}

// variant ids 4..7 of madiff_gemm_wgrad: all-taps kernel, {64x64, 128x64} (O x C) x {2, 3} LDS stages; nq = 32-row
// chunks per K-step (1 or 3), chosen by the caller from the sequence length.  Variants 8, 9: the 8-wave form (two wave
// groups split each K-step of 64*nq rows) with 2 / 3 LDS stages.  (A 128 x 64 tile with 64 x 32 x 5-tap wave tiles was
// built and measured in round 2: 78 us against 72 us on the 1024 x 1024 layer -- not kept.)
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_gemm_wgrad_taps_bf16(const void* params, int variant, int nq,
                                                                                        int gx, int gy, int gz,
                                                                                        hipStream_t stream) {
  const GemmWgradParams& p = *(const GemmWgradParams*)params;
  if (p.taps != 5 || (nq != 1 && nq != 3)) return -1;
  static int tm_on = -1;
  if (tm_on < 0) { const char* e = getenv("MADIFF_WGRAD_TM"); tm_on = e ? atoi(e) : 1; }
  const bool tm = tm_on && p.rm.T_out == 6 && p.rm.pad == 2;     // time-major K order (8-wave form only)
  dim3 grid(gx, gy, gz);
#define MADIFF_WT(V, BO_, D_)                                                                                           \
  case V:                                                                                                               \
    if (nq == 3) hipLaunchKernelGGL((gemm_wgrad_taps_bf16_kernel<BO_, 64, 5, 3, D_>), grid, dim3(256), 0, stream, p);   \
    else hipLaunchKernelGGL((gemm_wgrad_taps_bf16_kernel<BO_, 64, 5, 1, D_ + 1>), grid, dim3(256), 0, stream, p);       \
    break;
#define MADIFF_WT8(V, D_, D1_)                                                                                          \
  case V:                                                                                                               \
    if (nq == 3 && tm) hipLaunchKernelGGL((gemm_wgrad_taps8_bf16_kernel<5, 3, D_, true>), grid, dim3(512), 0, stream, p); \
    else if (nq == 3) hipLaunchKernelGGL((gemm_wgrad_taps8_bf16_kernel<5, 3, D_>), grid, dim3(512), 0, stream, p);      \
    else hipLaunchKernelGGL((gemm_wgrad_taps8_bf16_kernel<5, 1, D1_>), grid, dim3(512), 0, stream, p);                  \
    break;
  switch (variant) {
    MADIFF_WT(4, 64, 2)
    MADIFF_WT(5, 64, 3)
    MADIFF_WT(6, 128, 2)
    MADIFF_WT(7, 128, 3)
    MADIFF_WT8(8, 2, 6)
    MADIFF_WT8(9, 3, 7)
    default: return -1;
  }
#undef MADIFF_WT
#undef MADIFF_WT8
  MADIFF_CHECK_LAUNCH();
  return 0;
}
