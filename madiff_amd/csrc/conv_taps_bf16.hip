// All-taps implicit-GEMM kernel for stride-1 "same" k = 5 convolutions (bf16 storage, fp32 accumulate), forward and
// data gradient.  Reference op: nn.Conv1d(k=5, padding=2) inside Conv1dBlock (helpers.py:50-74), 24 of the 25 conv
// blocks of one denoiser forward, and the same layers' input gradients.
//
// Why a second kernel: gemm_fwd_bf16_kernel treats every (tap, 64-channel chunk) as an independent K-step and stages
// the activation tile again for each of the 5 taps, although the 5 tiles are the SAME rows shifted by tap - pad.  At
// 18 operand bytes per algorithmic kFLOP that kernel is bound by the L2 -> LDS rate (and by the issue cost of its
// LDS-DMA pieces), not by the MFMA pipe.  Here an output tile is a whole number of sequences, the activation chunk
// [BM rows x 64 channels] is staged ONCE per channel chunk and used by all 5 taps: the tap shift is applied when the
// MFMA A-fragments are read (ds_read_b128 takes a row address per lane; a lane whose shifted row leaves its sequence
// reads a zero row kept in LDS).  Per channel chunk the workgroup stages 1 activation tile + 5 weight tiles instead of
// 5 + 5: 6.6 B/kFLOP with the 192-row tile, 11.7 with the 96-row one.
//
//   tile        BM x 128 outputs, BM = 48 * WM rows (WM = 4: 192 rows, 8 waves, one workgroup per CU;
//               WM = 2: 96 rows, 4 waves, two per CU); wave tile 48 x 64 = 3 x 4 MFMA 16x16x32 blocks
//   rows        a tile holds SPT = BM / T whole sequences (rows_used = SPT * T <= BM; unused rows are zero-filled
//               and never stored), so any sequence length T <= BM works (6/12/24 use every row)
//   A buffers   2 x [BM rows][64 ch] (128-byte rows, 16-byte chunk index XORed with row & 7 on the DMA source side
//               and on the read side) + 1 KB of zeros behind each (the zero row)
//   W ring      D x 16 KB (D = 4 / 3): WT = false [128 n][64 k] as A;  WT = true [64 k][128 n] read with ds_read_b64_tr_b16
//               (the data gradient reads the FORWARD weight matrix, no transposed copy)
//   pipeline    step s = (chunk c, tap): counted s_waitcnt vmcnt -> one raw s_barrier -> issue W(s+2) (and A(c+1) at
//               tap 0) -> 24 MFMAs per wave.  Two W stages and one A stage stay in flight across the barrier.
//   epilogue    bias / row-bias / activation on 8-column vectors through a wave-private LDS tile, 16-byte stores.
//   time-major  (tm = 1, launcher: SPT % 16 == 0, i.e. T = 6 on both tiles, T = 12 on the 192-row one) the tile's rows are
//               ordered (t, sequence) instead of (sequence, t): tile row rho = t * SPT + s.  A 16-row A fragment then holds
//               ONE time step of 16 sequences, so a tap whose shifted step leaves the sequence is outside it for the whole
//               fragment and its 4 MFMAs are skipped (wave-uniform branch) instead of multiplying zeros: 6 of the 30
//               (t, tap) pairs at T = 6.  The order exists only inside the tile: the A pieces are gathered with per-lane
//               row addresses anyway, and the epilogues map rho back to the global row.
#include "gemm_params.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ int ct_wg_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
template <int N> __device__ __forceinline__ void ct_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One LDS-DMA piece: 64 lanes x 16 bytes, global (uniform base + 32-bit lane offset) -> LDS [lds_addr + 16 * lane].
// Issued through inline asm on purpose: hipcc models the builtin form as a FLAT access that may touch LDS, and while any
// such access is outstanding (here: always) its waitcnt pass turns EVERY lgkmcnt wait into lgkmcnt(0), i.e. each
// fragment read would drain the whole LDS queue.  Hidden in asm, the DMA is ordered by the explicit vmcnt waits and
// barriers below (all carry memory clobbers) and the ds_reads get exact counted waits.
__device__ __forceinline__ void ct_dma16(const void* gbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :
               : "s"(lds_addr), "v"(voff), "s"(gbase)
               : "memory");
}

// shift(tap) = shift0 + tap * shift_step: forward (tr = 0) -pad + tap, data gradient (tr = 1) pad - tap
// Fused epilogue of the forward kernel: GroupNorm(G) + Mish (+ per-sequence row bias | + residual) of Conv1dBlock
// (helpers.py:63-71, temporal.py:197-200).  The output tile holds whole sequences x whole groups, so the statistics never
// leave the workgroup.
struct GnEpi {
  const float* gamma; const float* beta;      // [C]
  float* mean; float* rstd;                   // [n_seq * G] saved for backward, or null
  bf16_t* out; int ldo;                       // normalised + activated output
  int write_y;                                // 1: also store the pre-norm conv output to p.C (needed by backward)
  int mode;                                   // 0 plain, 1 + rowbias[n, c], 2 + res[(n,t), c]
  const float* rowbias; int ldrb;
  const float* rowvec;                        // mode 1: + rowvec[c] as well (one vector for every sequence), or null
  const bf16_t* res; int ldr;
  int G, gv_log2;                             // groups; log2(16-byte... 8-channel vectors per group)
  float eps;
  // GroupNorm BACKWARD fused into the data-gradient kernel (GNB): the tile of dh = dgrad(conv) holds whole sequences x
  // whole groups of the PRECEDING Conv1dBlock's output, so that block's GroupNorm + Mish backward runs here and the kernel
  // stores dy (gradient of the preceding conv's pre-norm output) instead of dh.
  const bf16_t* by; int ldby;                 // pre-norm output y of the preceding block (saved by its forward)
  const float* bmean; const float* brstd;     // its statistics [n_seq * G]
  float* dgamma; float* dbeta; float* dbias;  // [C], accumulated atomically (dbias = the preceding conv's bias gradient, may be null)
  float* drowb; int lddrb;                    // d rowbias [n_seq, lddrb] = sum_t dh (mode 1: the time-embedding term), or null
  // chained form (CH): dh = dgrad(conv) + gadd is ALSO a gradient somebody else needs (the preceding block's residual
  // branch), so it is stored to dhout as well, and gadd (this block's own residual-branch gradient) joins it first
  const bf16_t* gadd; int ldga;               // addend [(n,t), ldga] or null
  bf16_t* dhout; int lddh;                    // dh [(n,t), lddh]
};

// D = 3 / 4: one barrier per (chunk, tap) step.  D = 5: one barrier per TWO steps (the ring then holds the three stages a
// double-step reads or prefetches from plus the two in flight; needs an even number of channel chunks).
template <int WM, bool WT, int D, bool GN, int VPL = 0, bool GNB = false, bool CH = false>
__global__ __launch_bounds__(WM * 128) void conv_taps_bf16_kernel(GemmFwdParams p, int rows_used, int shift0, int shift_step,
                                                                  GnEpi ge, int tm) {
  static_assert(!(GN && WT), "the GroupNorm epilogue belongs to the forward kernel");
  static_assert(!GNB || (WT && !GN), "the GroupNorm-backward epilogue belongs to the data-gradient kernel");
  static_assert(!CH || GNB, "the chained form is a variant of the GroupNorm-backward epilogue");
  constexpr int TAPS = 5;
  static_assert(D == 3 || D == 4 || D == 5, "W ring depth");
  constexpr int BM = 48 * WM, NW = 2 * WM, NT = NW * 64;
  constexpr int MI = 3, NJ = 4;
  constexpr int A_TILE = BM * 128, A_BYTES = A_TILE + 1024, W_BYTES = 128 * 128;
  constexpr int AP = (BM / 8) / NW;        // A DMA pieces (1 KB = 8 rows x 128 B) per wave = 3
  constexpr int WP = 16 / NW;              // W DMA pieces per wave
  static_assert(AP * NW * 8 == BM && WP * NW == 16, "pieces must split evenly over the waves");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_BYTES + D * W_BYTES];
  unsigned char* const Wbase = smem + 2 * A_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int T = p.rm.T_out;
  int bx = blockIdx.x, by = blockIdx.y;
  xcd_remap(bx, by, p.xcd);
  const int m0 = bx * rows_used, n0 = by * 128;
  // time-major tiles: fragment i of this wave holds time step tf[i] of 16 sequences; loc() maps a tile row of that fragment
  // to its row offset from m0
  const int spt_t = rows_used / T;
  int tf[3];
#pragma unroll
  for (int i = 0; i < 3; i++) tf[i] = tm ? (wm * 48 + i * 16) / spt_t : 0;
  auto loc = [&](int i, int rho) { return tm ? (rho - tf[i] * spt_t) * T + tf[i] : rho; };
  const int cbase = p.cks > 1 ? (int)blockIdx.z * ((p.Cin >> 6) / p.cks) : 0;

  for (int i = tid; i < 512; i += NT) ((float*)(smem + (i >> 8) * A_BYTES + A_TILE))[i & 255] = 0.f;   // zero rows

  // ---- DMA sources.  Rows / columns outside the problem are CLAMPED to a valid one instead of being zero-filled:
  // what they produce lands only in output rows that are never stored or in pad columns the epilogue writes as 0,
  // so every lane's source is (uniform base) + (32-bit lane offset) with no per-piece select.
  unsigned a_off[2][AP];                   // lane offset into X (and into X2, the second source of a skip concat)
#pragma unroll
  for (int j = 0; j < AP; j++) {
    const int row = (AP * wave + j) * 8 + (lane >> 3);
    const unsigned c16 = ((lane & 7) ^ (row & 7)) << 4;
    const int lrow = tm ? (row % spt_t) * T + row / spt_t : row;
    const unsigned grow = row < rows_used ? min(m0 + lrow, p.M - 1) : m0;
    a_off[0][j] = grow * (unsigned)(p.ldx * 2) + c16;
    a_off[1][j] = grow * (unsigned)(p.ldx2 * 2) + c16;
  }
  unsigned w_off[WP];
#pragma unroll
  for (int j = 0; j < WP; j++) {
    if constexpr (!WT) {
      const int row = (WP * wave + j) * 8 + (lane >> 3);
      const int wr = min(n0 + row, p.Nout - 1);
      w_off[j] = (unsigned)wr * (unsigned)(p.ldw * 2) + (((lane & 7) ^ (row & 7)) << 4);
    } else {
      const int krow = (WP * wave + j) * 4 + (lane >> 4);
      const int ch = (lane & 15) ^ ct_wg_swz(krow);
      const int col = min(n0 + ch * 8, p.wcols - 8);
      w_off[j] = ((unsigned)krow * (unsigned)p.ldw + (unsigned)col) * 2;
    }
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)smem;
  auto issueA = [&](int c) {
    int c0 = (c + cbase) << 6;
    const unsigned char* xb = (const unsigned char*)p.X;
    const bool second = c0 >= p.Csplit;
    if (second) { xb = (const unsigned char*)p.X2; c0 -= p.Csplit; }
    xb += c0 * 2;
    const unsigned base = lds0 + (c & 1) * A_BYTES + AP * wave * 1024;
#pragma unroll
    for (int j = 0; j < AP; j++) ct_dma16(xb, second ? a_off[1][j] : a_off[0][j], base + j * 1024);
  };
  auto issueW = [&](int tap, int c, int slot) {
    const unsigned char* wb = (const unsigned char*)p.W +
                              (WT ? ((size_t)((c + cbase) << 6) * p.ldw + (size_t)tap * p.wtap) * 2
                                  : ((size_t)tap * p.Cin + ((c + cbase) << 6)) * 2);
    const unsigned base = lds0 + 2 * A_BYTES + slot * W_BYTES + WP * wave * 1024;
#pragma unroll
    for (int j = 0; j < WP; j++) ct_dma16(wb, w_off[j], base + j * 1024);
  };

  // ---- loop-invariant LDS offsets of the fragments.  A: row shifted by the tap and its swizzled chunk; a lane whose
  // shifted row leaves the sequence reads the zero block at the SAME bank position it would have had (so the access
  // stays conflict-free).  kh = 1 toggles chunk bit 2 = byte offset ^ 64 everywhere.
  const int fr = lane & 15, fq = lane >> 4;
  const int tq = fr >> 2, tp = fr & 3;
  int offA[TAPS][MI];
  unsigned skip = 0;                       // bit (k * MI + i): (tap k, fragment i) lies wholly outside its sequences (tm only)
#pragma unroll
  for (int i = 0; i < MI; i++) {
    const int ra = wm * 48 + i * 16 + fr;
    const int t = tm ? tf[i] : ra % T;
#pragma unroll
    for (int k = 0; k < TAPS; k++) {
      const int sh = shift0 + k * shift_step;
      const int ts = t + sh, rs = ra + (tm ? sh * spt_t : sh);
      const int nat = rs * 128 + ((fq ^ (rs & 7)) << 4);
      offA[k][i] = (ra < rows_used && ts >= 0 && ts < T) ? nat : A_TILE + (nat & 255);
      if (tm && (wm * 48 + i * 16 >= rows_used || tf[i] + sh < 0 || tf[i] + sh >= T)) skip |= 1u << (k * MI + i);
    }
  }
  skip = __builtin_amdgcn_readfirstlane(skip);
  int offB[2];
  if constexpr (!WT) {
    const int rb = wn * 64 + fr;                          // + 16 j rows = + 2048 j bytes
    offB[0] = rb * 128 + ((fq ^ (rb & 7)) << 4);
    offB[1] = 0;
  } else {
    const int r0 = 8 * fq + tq, r1 = r0 + 4;              // + 32 kh rows = + 8192 kh bytes (the swizzle ignores bit 5 of the row)
    const int cb = 8 * wn + (tp >> 1), o = 8 * (tp & 1);  // + 2 j chunks = ^ (j << 5) bytes
    offB[0] = r0 * 256 + ((cb ^ ct_wg_swz(r0)) << 4) + o;
    offB[1] = r1 * 256 + ((cb ^ ct_wg_swz(r1)) << 4) + o;
  }

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto load_frags = [&](bf16x8 (&a)[MI], bf16x8 (&b)[NJ], const unsigned char* As, const unsigned char* Ws, const int (&oa)[MI],
                        const int kh) {
#pragma unroll
    for (int i = 0; i < MI; i++) a[i] = *(const bf16x8*)(As + (oa[i] ^ (kh << 6)));
    if constexpr (!WT) {
      const unsigned char* wp = Ws + (offB[0] ^ (kh << 6));
#pragma unroll
      for (int j = 0; j < NJ; j++) b[j] = *(const bf16x8*)(wp + j * 2048);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ws + kh * 8192 + (offB[0] ^ (j << 5))));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Ws + kh * 8192 + (offB[1] ^ (j << 5))));
        b[j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
  };
  auto mma = [&](const bf16x8 (&a)[MI], const bf16x8 (&b)[NJ], const int tap) {
#pragma unroll
    for (int i = 0; i < MI; i++) {
      if (skip & (1u << (tap * MI + i))) continue;           // wave-uniform: the fragment is all zeros for this tap
#pragma unroll
      for (int j = 0; j < NJ; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- software pipeline.  Invariant at the top of step s = (c, tap): the kh = 0 fragments of step s are (being) loaded
  // into a0/b0, W(s) and A(c) are visible to every wave, W(s+1) .. W(s+D-2) are in flight.
  const int nc = p.cks > 1 ? (p.Cin >> 6) / p.cks : (p.Cin >> 6);      // channel chunks of this workgroup's slice
  bf16x8 a0[MI], b0[NJ], a1[MI], b1[NJ];
  __syncthreads();                         // zero rows written before any fragment read
  if constexpr (D == 5) {
    // ---- one barrier per two steps.  Flat step s = 5 c + tap, W(s) lives in ring slot s % 5, A(c) in buffer c & 1; the
    // loop body covers two chunks = ten steps, so every LDS offset below is a compile-time constant.  At the barrier
    // before even step s: W(s), W(s+1), W(s+2) and the A chunks they (or the kh = 0 prefetch of step s+2) touch are
    // complete (each wave drained its own DMA first: everything outstanding was issued one double-step earlier);
    // then W(s+3), W(s+4) are issued into the slots of W(s-2), W(s-1), A(c1) after the barrier of step 0 and the next
    // pair's A(c0) after the barrier of step 6 (its buffer was last read in step 4).
    const int nsteps = nc * TAPS;
    issueA(0);
    issueW(0, 0, 0);
    issueW(1, 0, 1);
    issueW(2, 0, 2);
    ct_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_frags(a0, b0, smem, Wbase, offA[0], 0);
    for (int c0 = 0; c0 < nc; c0 += 2) {
      const int sbase = c0 * TAPS;
#pragma unroll
      for (int f = 0; f < 2 * TAPS; f++) {
        const int cc = f / TAPS, tap = f % TAPS;            // chunk c0 + cc, compile-time tap
        const int s = sbase + f;
        const bool last = s + 1 == nsteps;
        if ((f & 1) == 0) {
          if (f > 0 || c0 > 0) {                            // (the prologue did this for the very first step)
            ct_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
          }
          if (f == 0 && c0 + 1 < nc) issueA(c0 + 1);
          if (f == 6 && c0 + 2 < nc) issueA(c0 + 2);
#pragma unroll
          for (int q = 3; q <= 4; q++) {
            const int f2 = f + q;                           // flat index inside / beyond this pair
            const int c2 = c0 + f2 / TAPS, t2 = f2 % TAPS;
            if (c2 < nc) issueW(t2, c2, f2 % 5);
          }
        }
        const unsigned char* As = smem + cc * A_BYTES;
        const unsigned char* Ws = Wbase + (f % 5) * W_BYTES;
        __builtin_amdgcn_sched_barrier(0);
        load_frags(a1, b1, As, Ws, offA[tap], 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0, tap);
        __builtin_amdgcn_sched_barrier(0);
        if (!last) {
          const int fn = f + 1;
          const unsigned char* An = smem + ((fn / TAPS) & 1) * A_BYTES;
          load_frags(a0, b0, An, Wbase + (fn % 5) * W_BYTES, offA[fn % TAPS], 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1, tap);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
  issueA(0);
#pragma unroll
  for (int d = 0; d < D - 1; d++) issueW(d, 0, d);
  ct_wait_vmcnt<(D - 2) * WP>();
  __builtin_amdgcn_s_barrier();
  load_frags(a0, b0, smem, Wbase, offA[0], 0);
  int slot = 0;                            // ring slot holding W(s)
  for (int c = 0; c < nc; c++) {
    const unsigned char* As = smem + (c & 1) * A_BYTES;
    const unsigned char* An = smem + ((c + 1) & 1) * A_BYTES;
    const bool lastc = c + 1 == nc;
#pragma unroll
    for (int tap = 0; tap < TAPS; tap++) {
      const bool last = lastc && tap == TAPS - 1;
      int nslot = slot + 1;                // slot of W(s+1)
      if (nslot == D) nslot = 0;
      if (!last) {
        // own pieces of W(s+1) landed.  Outstanding, oldest first: W(s+1), [A(c+1) when it was issued in step s-1],
        // W(s+2) .. W(s+D-2) (those that exist)
        if (tap == 1 && !lastc) ct_wait_vmcnt<(D - 3) * WP + AP>();
        else if (tap == TAPS - 2 && lastc) ct_wait_vmcnt<0>();
        else ct_wait_vmcnt<(D - 3) * WP>();
        // every wave's W(s+1) (and, in order, anything older) is in LDS, and every wave has finished step s-1
        __builtin_amdgcn_s_barrier();
        // A(c+1) goes into the buffer A(c-1) occupied.  Order matters for the counted waits above: with D = 4, W(s+1) was
        // issued two steps ago, so A may precede this step's W; with D = 3, W(s+1) is THIS step's W of the previous
        // step and must be older than the A pieces that are allowed to stay outstanding at tap 1.
        if (D == 4 && tap == 0 && !lastc) issueA(c + 1);
        int tn = tap + D - 1, cn = c;
        if (tn >= TAPS) { tn -= TAPS; cn = c + 1; }
        int islot = slot + D - 1;                          // = the slot W(s-1) occupied
        if (islot >= D) islot -= D;
        if (cn < nc) issueW(tn, cn, islot);
        if (D == 3 && tap == 0 && !lastc) issueA(c + 1);
      }
      const unsigned char* Ws = Wbase + slot * W_BYTES;
      // sched_barrier(0): hipcc otherwise sinks each fragment read down to just before its MFMA, which undoes the
      // prefetch distance; with the fences the reads of one phase fly under the MFMAs of the other
      __builtin_amdgcn_sched_barrier(0);
      load_frags(a1, b1, As, Ws, offA[tap], 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0, tap);
      __builtin_amdgcn_sched_barrier(0);
      if (!last) {                                        // kh = 0 fragments of step s+1: legal before the next barrier
        const unsigned char* Wn = Wbase + nslot * W_BYTES;
        if (tap == TAPS - 1) load_frags(a0, b0, An, Wn, offA[0], 0);
        else load_frags(a0, b0, As, Wn, offA[tap + 1 < TAPS ? tap + 1 : 0], 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1, tap);
      __builtin_amdgcn_sched_barrier(0);
      slot = nslot;
    }
  }
  }

  if (p.cks > 1) {
    // ---- reduction split: partial tiles meet in the workspace, the last slice to arrive finishes the tile.  Every thread
    // stores / reloads its own accumulator registers (16-byte vectors, coalesced), so no layout change is involved; the
    // sum runs over the slices in order, whoever is last, so results do not depend on arrival order.
    __shared__ int ticket;
    const int tile_id = by * gridDim.x + bx;
    constexpr int PART = NT * MI * NJ * 4;                     // floats per partial tile
    float* wsb = p.ws + (size_t)tile_id * p.cks * PART;
    float* mine = wsb + (size_t)blockIdx.z * PART;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) *(f32x4*)(mine + ((i * NJ + j) * NT + tid) * 4) = acc[i][j];
    __threadfence();
    __syncthreads();
    if (tid == 0) ticket = atomicAdd(&p.ws_cnt[tile_id], 1);
    __syncthreads();
    if (ticket != p.cks - 1) return;
    __threadfence();
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.cks; z++) {
      const float* src = wsb + (size_t)z * PART;
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const f32x4 v = __builtin_nontemporal_load((const f32x4*)(src + ((i * NJ + j) * NT + tid) * 4));
          acc[i][j] += v;
        }
    }
    if (tid == 0) p.ws_cnt[tile_id] = 0;                       // ready for the next launch
  }

  if constexpr (GN) {
    // ---- fused GroupNorm + Mish epilogue.  (0) every global operand of the epilogue (row bias / residual of all the
    // sequences this wave will normalise) is requested first, so its latency hides behind (1) accumulators + conv bias
    // -> fp32 LDS image of the whole tile; (2) a wave takes whole sequences: lane = (row r0 = lane / 16, 8-channel
    // vector cv = lane % 16), iteration i covers row r0 + 4 i (VPL = ceil(T / 4) iterations, a template parameter so
    // that all SEQ = 8 / VPL sequences of a wave live in registers at once and their butterflies interleave); group
    // statistics are butterfly sums over the lane bits inside a group (as gn_mish_fwd_rb_kernel); (3) y (pre-norm,
    // for backward) and the activated output leave as 16-byte vectors.
    constexpr int LDG = 132;
    constexpr int SEQ = 8 / VPL;             // sequences per wave (launcher: ceil(spt / NW) <= SEQ)
    static_assert(VPL >= 1 && SEQ >= 1, "VPL in 1..8");
    static_assert(BM * LDG * 4 <= 2 * A_BYTES + D * W_BYTES, "tile image must fit the operand LDS");
    float* tile = (float*)smem;
    const int cv = lane & 15, r0 = lane >> 4, c0 = n0 + cv * 8;
    const int Cg = p.Nout / ge.G;
    const float inv_e = 1.f / (float)(T * Cg);
    const int spt = rows_used / T, n_seq = p.M / T;
    bf16x8 rs[SEQ][VPL];
    float rb[SEQ][8];
    float gmv[8], btv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { gmv[e] = ge.gamma[c0 + e]; btv[e] = ge.beta[c0 + e]; }
#pragma unroll
    for (int sl = 0; sl < SEQ; sl++) {
      const int sq = wave + sl * NW, n = bx * spt + sq;
      const bool live = sq < spt && n < n_seq;
#pragma unroll
      for (int e = 0; e < 8; e++) rb[sl][e] = (live && ge.mode == 1) ? ge.rowbias[(size_t)n * ge.ldrb + c0 + e] : 0.f;
      if (ge.rowvec) {                       // (wave-uniform)
#pragma unroll
        for (int e = 0; e < 8; e++) rb[sl][e] += ge.rowvec[c0 + e];
      }
      if (ge.mode == 2) {
#pragma unroll
        for (int i = 0; i < VPL; i++) {
          const int r = r0 + 4 * i;
          if (live && r < T) rs[sl][i] = *(const bf16x8*)(ge.res + ((size_t)n * T + r) * ge.ldr + c0);
        }
      }
    }
    __syncthreads();                         // every wave is done with the operand tiles; no DMA is outstanding
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int col = wn * 64 + j * 16 + fr;
      const float bv = p.bias ? p.bias[n0 + col] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) tile[loc(i, wm * 48 + i * 16 + fq * 4 + r) * LDG + col] = acc[i][j][r] + bv;
    }
    __syncthreads();
    bf16_t* Y = (bf16_t*)p.C;
    float v[SEQ][VPL][8];
    float sum[SEQ], sqs[SEQ], mean[SEQ], rstd[SEQ];
#pragma unroll
    for (int sl = 0; sl < SEQ; sl++) {
      const int sq = wave + sl * NW, n = bx * spt + sq;
      const bool live = sq < spt && n < n_seq;
      sum[sl] = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int r = r0 + 4 * i;
        if (live && r < T) {
          const f32x4 a = *(const f32x4*)&tile[(sq * T + r) * LDG + cv * 8];
          const f32x4 b = *(const f32x4*)&tile[(sq * T + r) * LDG + cv * 8 + 4];
          v[sl][i][0] = a[0]; v[sl][i][1] = a[1]; v[sl][i][2] = a[2]; v[sl][i][3] = a[3];
          v[sl][i][4] = b[0]; v[sl][i][5] = b[1]; v[sl][i][6] = b[2]; v[sl][i][7] = b[3];
#pragma unroll
          for (int e = 0; e < 8; e++) sum[sl] += v[sl][i][e];
          if (ge.write_y) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (bf16_t)v[sl][i][e];
            *(bf16x8*)(Y + ((size_t)n * T + r) * p.ldc + c0) = o;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) v[sl][i][e] = 0.f;
        }
      }
    }
    // (explicit fmaf below: left to -ffp-contract, the unrolled sequence slots got different contractions, so a
    // sequence's statistics depended by an ulp on its position in the tile)
    // lanes of one group differ in bits [0, gv_log2) (vector inside the group) and bits 4, 5 (row)
    for (int b = 0; b < ge.gv_log2; b++) {
#pragma unroll
      for (int sl = 0; sl < SEQ; sl++) sum[sl] += __shfl_xor(sum[sl], 1 << b, 64);
    }
#pragma unroll
    for (int sl = 0; sl < SEQ; sl++) {
      sum[sl] += __shfl_xor(sum[sl], 16, 64);
      sum[sl] += __shfl_xor(sum[sl], 32, 64);
      mean[sl] = sum[sl] * inv_e;
      const int sq = wave + sl * NW, n = bx * spt + sq;
      const bool live = sq < spt && n < n_seq;
      sqs[sl] = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; i++)
        if (live && r0 + 4 * i < T) {
#pragma unroll
          for (int e = 0; e < 8; e++) { const float d = v[sl][i][e] - mean[sl]; sqs[sl] = __builtin_fmaf(d, d, sqs[sl]); }
        }
    }
    for (int b = 0; b < ge.gv_log2; b++) {
#pragma unroll
      for (int sl = 0; sl < SEQ; sl++) sqs[sl] += __shfl_xor(sqs[sl], 1 << b, 64);
    }
#pragma unroll
    for (int sl = 0; sl < SEQ; sl++) {
      sqs[sl] += __shfl_xor(sqs[sl], 16, 64);
      sqs[sl] += __shfl_xor(sqs[sl], 32, 64);
      rstd[sl] = rsqrtf(sqs[sl] * inv_e + ge.eps);
      const int sq = wave + sl * NW, n = bx * spt + sq;
      const bool live = sq < spt && n < n_seq;
      if (!live) continue;
      if (ge.mean && r0 == 0 && (cv & ((1 << ge.gv_log2) - 1)) == 0) {
        const int g = c0 / Cg;
        ge.mean[(size_t)n * ge.G + g] = mean[sl];
        ge.rstd[(size_t)n * ge.G + g] = rstd[sl];
      }
      float sc[8], sh[8];
#pragma unroll
      for (int e = 0; e < 8; e++) { sc[e] = rstd[sl] * gmv[e]; sh[e] = __builtin_fmaf(-mean[sl], sc[e], btv[e]); }
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int r = r0 + 4 * i;
        if (r < T) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++)
            o[e] = (bf16_t)(mish_f(__builtin_fmaf(v[sl][i][e], sc[e], sh[e])) + (ge.mode == 2 ? (float)rs[sl][i][e] : rb[sl][e]));
          *(bf16x8*)(ge.out + ((size_t)n * T + r) * ge.ldo + c0) = o;
        }
      }
    }
    return;
  }

  if constexpr (GNB) {
    // ---- fused GroupNorm + Mish BACKWARD of the preceding Conv1dBlock (helpers.py:63-71 differentiated; the arithmetic of
    // gn_mish_bwd_rb_kernel).  Same geometry as the forward epilogue: (0) y vectors and statistics of the wave's sequences
    // are requested first; (1) accumulators (= dh) -> fp32 LDS image of the tile; (2) lane = (row r0, 8-channel vector cv),
    // per sequence: x^ = (y - mean) rstd, dz = dh mish'(x^ g + b), group sums s1 = mean(dz g), s2 = mean(dz g x^) by
    // butterflies, dy = rstd (dz g - s1 - x^ s2); (3) per-channel sums (dgamma, dbeta, dbias) stay in registers over the
    // wave's sequences, then lanes -> waves (LDS) -> one atomic per channel per workgroup; d rowbias = sum_t dh per sequence.
    constexpr int LDG = 132;
    constexpr int SEQ = 8 / VPL;
    static_assert(VPL >= 1 && SEQ >= 1, "VPL in 1..8");
    static_assert(BM * LDG * 4 <= 2 * A_BYTES + D * W_BYTES, "tile image must fit the operand LDS");
    float* tile = (float*)smem;
    const int cv = lane & 15, r0 = lane >> 4, c0 = n0 + cv * 8;
    const int Cg = p.Nout / ge.G;
    const float inv_e = 1.f / (float)(T * Cg);
    const int spt = rows_used / T, n_seq = p.M / T;
    const int g = c0 / Cg;
    bf16x8 yv[SEQ][VPL];
    bf16x8 adv[CH ? SEQ : 1][CH ? VPL : 1];
    float mean[SEQ], rstd[SEQ];
    float gmv[8], btv[8], a_dg[8], a_db[8], a_dbias[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { gmv[e] = ge.gamma[c0 + e]; btv[e] = ge.beta[c0 + e]; a_dg[e] = a_db[e] = a_dbias[e] = 0.f; }
#pragma unroll
    for (int sl = 0; sl < SEQ; sl++) {
      const int sq = wave + sl * NW, n = bx * spt + sq;
      const bool live = sq < spt && n < n_seq;
      mean[sl] = live ? ge.bmean[(size_t)n * ge.G + g] : 0.f;
      rstd[sl] = live ? ge.brstd[(size_t)n * ge.G + g] : 0.f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int r = r0 + 4 * i;
        if (live && r < T) yv[sl][i] = *(const bf16x8*)(ge.by + ((size_t)n * T + r) * ge.ldby + c0);
        if constexpr (CH) {
          if (live && r < T && ge.gadd) adv[sl][i] = *(const bf16x8*)(ge.gadd + ((size_t)n * T + r) * ge.ldga + c0);
        }
      }
    }
    __syncthreads();                         // every wave is done with the operand tiles; no DMA is outstanding
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int col = wn * 64 + j * 16 + fr;
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) tile[loc(i, wm * 48 + i * 16 + fq * 4 + r) * LDG + col] = acc[i][j][r];
    }
    __syncthreads();
    bf16_t* DY = (bf16_t*)p.C;
#pragma unroll
    for (int sl = 0; sl < SEQ; sl++) {
      const int sq = wave + sl * NW, n = bx * spt + sq;
      const bool live = sq < spt && n < n_seq;
      float dxh[VPL][8], rbs[8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) rbs[e] = 0.f;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int r = r0 + 4 * i;
        if (live && r < T) {
          const f32x4 a = *(const f32x4*)&tile[(sq * T + r) * LDG + cv * 8];
          const f32x4 b = *(const f32x4*)&tile[(sq * T + r) * LDG + cv * 8 + 4];
          float go[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
          if constexpr (CH) {
            if (ge.gadd) {
#pragma unroll
              for (int e = 0; e < 8; e++) go[e] += (float)adv[sl][i][e];
            }
            bf16x8 oh;
#pragma unroll
            for (int e = 0; e < 8; e++) oh[e] = (bf16_t)go[e];
            *(bf16x8*)(ge.dhout + ((size_t)n * T + r) * ge.lddh + c0) = oh;
          }
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float x = ((float)yv[sl][i][e] - mean[sl]) * rstd[sl];
            const float dz = go[e] * mish_grad_f(__builtin_fmaf(x, gmv[e], btv[e]));
            a_dg[e] = __builtin_fmaf(dz, x, a_dg[e]);
            a_db[e] += dz;
            rbs[e] += go[e];
            const float d = dz * gmv[e];
            dxh[i][e] = d;
            s1 += d;
            s2 = __builtin_fmaf(d, x, s2);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) dxh[i][e] = 0.f;
        }
      }
      // lanes of one group differ in bits [0, gv_log2) (vector inside the group) and bits 4, 5 (row)
      for (int b = 0; b < ge.gv_log2; b++) { s1 += __shfl_xor(s1, 1 << b, 64); s2 += __shfl_xor(s2, 1 << b, 64); }
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      s1 *= inv_e; s2 *= inv_e;
#pragma unroll
      for (int i = 0; i < VPL; i++) {
        const int r = r0 + 4 * i;
        if (live && r < T) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float x = ((float)yv[sl][i][e] - mean[sl]) * rstd[sl];
            const float v = rstd[sl] * (dxh[i][e] - s1 - x * s2);
            a_dbias[e] += v;
            o[e] = (bf16_t)v;
          }
          *(bf16x8*)(DY + ((size_t)n * T + r) * p.ldc + c0) = o;
        }
      }
      if (ge.drowb) {
#pragma unroll
        for (int e = 0; e < 8; e++) { rbs[e] += __shfl_xor(rbs[e], 16, 64); rbs[e] += __shfl_xor(rbs[e], 32, 64); }
        if (live && r0 == 0) {
          float* dst = ge.drowb + (size_t)n * ge.lddrb + c0;
          *(f32x4*)dst = (f32x4){rbs[0], rbs[1], rbs[2], rbs[3]};
          *(f32x4*)(dst + 4) = (f32x4){rbs[4], rbs[5], rbs[6], rbs[7]};
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      a_dg[e] += __shfl_xor(a_dg[e], 16, 64); a_dg[e] += __shfl_xor(a_dg[e], 32, 64);
      a_db[e] += __shfl_xor(a_db[e], 16, 64); a_db[e] += __shfl_xor(a_db[e], 32, 64);
      a_dbias[e] += __shfl_xor(a_dbias[e], 16, 64); a_dbias[e] += __shfl_xor(a_dbias[e], 32, 64);
    }
    __syncthreads();                         // the tile image has been read by every wave: its LDS is free
    float* red = (float*)smem;               // [wave][dgamma | dbeta | dbias][128 channels]
    if (r0 == 0) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        red[(wave * 3 + 0) * 128 + cv * 8 + e] = a_dg[e];
        red[(wave * 3 + 1) * 128 + cv * 8 + e] = a_db[e];
        red[(wave * 3 + 2) * 128 + cv * 8 + e] = a_dbias[e];
      }
    }
    __syncthreads();
    for (int i = tid; i < 3 * 128; i += NT) {
      const int q = i >> 7, c = i & 127;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++) a += red[(w * 3 + q) * 128 + c];
      float* dst = q == 0 ? ge.dgamma : (q == 1 ? ge.dbeta : ge.dbias);
      if (dst) atomicAdd(&dst[n0 + c], a);
    }
    return;
  }

  // ---- epilogue (vector path only; the launcher checks the alignment preconditions)
  bf16_t* C = (bf16_t*)p.C;
  constexpr int WN = 64, LDT = WN + 4, VPR = WN / 8, RPP = 64 / VPR;
  __syncthreads();                         // every wave is done with the operand tiles; no DMA is outstanding
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
      const int lr = wm * 48 + i * 16 + row;
      const int m = m0 + loc(i, lr);
      const int n = n0 + wn * WN + cv * 8;
      if (lr >= rows_used || m >= p.M || n >= p.Npad) continue;
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      const int seq = m / T;
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
      if (p.addend) {
        const bf16x8 ad = *(const bf16x8*)((const bf16_t*)p.addend + (size_t)m * p.lda + n);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += (float)ad[e];
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = (bf16_t)v[e];
      *(bf16x8*)(C + (size_t)m * p.ldc + n) = o;
    }
  }
}

// Reduction split of the all-taps kernel (p.cks, see gemm_params.h): only for layers whose output tiles leave most of the
// chip idle; the slices must hold whole chunks (an even number of them for the two-steps-per-barrier variant) and the
// partial tiles must fit the registered workspace.
// Time-major tile order (see the header): needs fragments that hold one time step, i.e. SPT a multiple of 16.
static int ct_time_major(int spt) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MADIFF_CONV_TM"); on = e ? atoi(e) : 1; }
  return on && spt >= 16 && spt % 16 == 0;
}

static int ct_pick_cks(const GemmFwdParams& p, long tiles, int variant) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MADIFF_CONV_SPLITK"); on = e ? atoi(e) : 1; }
  if (!on || !p.ws || !p.ws_cnt || tiles >= 128 || tiles > p.ws_ncnt) return 1;
  const int nc = p.Cin >> 6, nt = variant == 1 ? 256 : 512;
  int best = 1;
  for (int k = 2; k <= nc / 2; k++) {
    if (nc % k) continue;
    const int per = nc / k;
    if (variant == 2 && (per & 1)) continue;
    if (tiles * k > 512 || tiles * k * nt * 48 > p.ws_floats) break;
    best = k;
  }
  return best;
}

// variant: 0 = 192-row tile (8 waves), 1 = 96-row tile (4 waves).  Returns 1 when the layer is not covered.
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_conv_taps_bf16(const void* params, int out_f32, int variant,
                                                                                  hipStream_t stream) {
  GemmFwdParams p = *(const GemmFwdParams*)params;
  const int BM = variant == 1 ? 96 : 192;
  const int T = p.rm.T_out;
  if (variant == 2 && p.Cin % 128) return 1;      // the two-steps-per-barrier loop covers chunk pairs
  if (out_f32 || p.taps != 5 || p.rm.stride != 1 || p.rm.T_in != T || T > BM || p.permA > 0 || p.ksplit > 1 ||
      p.Cin % 64 || p.Csplit % 64 || p.M % T)
    return 1;
  if (p.rm.pad < 0 || p.rm.pad > 4) return 1;
  // lane offsets are 32-bit: operand extents must stay below 2 GiB
  {
    const size_t ldmax = (size_t)(p.ldx > p.ldx2 ? p.ldx : p.ldx2);
    const size_t wext = p.wt ? (size_t)p.Cin * p.ldw * 2 : (size_t)p.Nout * p.ldw * 2;
    if ((size_t)p.M * ldmax * 2 >= (1ull << 31) || wext >= (1ull << 31)) return 1;
    if (p.wt && p.wcols < 8) return 1;
  }
  const bool vec_ok = p.Npad % 8 == 0 && ((size_t)p.C % 16) == 0 && ((size_t)p.ldc * 2) % 16 == 0 &&
                      (!p.bias || (size_t)p.bias % 16 == 0) && (!p.rowbias || ((size_t)p.rowbias % 16 == 0 && p.ldrb % 4 == 0));
  if (!vec_ok) return 1;
  if (p.addend && (((size_t)p.addend % 16) || p.lda % 8 || p.lda < p.Npad)) return 1;
  const int spt = BM / T, rows_used = spt * T;
  const int n_seq = p.M / T;
  const int shift0 = p.rm.tr ? p.rm.pad : -p.rm.pad, shift_step = p.rm.tr ? -1 : 1;
  dim3 grid((n_seq + spt - 1) / spt, (p.Npad + 127) / 128);
  p.cks = ct_pick_cks(p, (long)grid.x * grid.y, variant);
  grid.z = p.cks;
  GnEpi ge = {};
#define CT_LAUNCH(WM_, WT_, D_) hipLaunchKernelGGL((conv_taps_bf16_kernel<WM_, WT_, D_, false>), grid, dim3(WM_ * 128), 0, stream, p, rows_used, shift0, shift_step, ge, ct_time_major(spt))
  if (variant == 2) {          // 192 rows, 5-deep W ring, one barrier per two (chunk, tap) steps
    if (p.wt) CT_LAUNCH(4, true, 5); else CT_LAUNCH(4, false, 5);
  } else if (variant == 0) {   // 192 rows: one 8-wave workgroup per CU, 4-deep W ring
    if (p.wt) CT_LAUNCH(4, true, 4); else CT_LAUNCH(4, false, 4);
  } else {                     // 96 rows: two 4-wave workgroups per CU (75 KB of LDS each), 3-deep W ring
    if (p.wt) CT_LAUNCH(2, true, 3); else CT_LAUNCH(2, false, 3);
  }
#undef CT_LAUNCH
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// Conv1dBlock forward in one launch: k5 "same" convolution + bias + GroupNorm(G) + Mish (+ rowbias | + residual).
// p: the convolution (p.C / p.ldc = pre-norm output y, written only when write_y).  Returns 1 when the layer is not
// covered (the caller then runs the convolution and madiff_gn_mish_fwd separately).
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_conv_taps_gn_bf16(
    const void* params, int variant, void* out, int ldo, const float* gamma, const float* beta, float* mean, float* rstd, int G,
    float eps, int mode, const float* rowbias, int ldrb, const void* res, int ldr, int write_y, const float* rowvec,
    hipStream_t stream) {
  GemmFwdParams p = *(const GemmFwdParams*)params;
  const int BM = variant == 1 ? 96 : 192;
  if (rowvec && (mode != 1 || ((size_t)rowvec % 16))) return -1;
  const int T = p.rm.T_out;
  if (variant == 2 && p.Cin % 128) return 1;
  if (p.wt || p.taps != 5 || p.rm.stride != 1 || p.rm.tr != 0 || p.rm.pad != 2 || p.rm.T_in != T || T > 32 || p.permA > 0 ||
      p.ksplit > 1 || p.Cin % 64 || p.Csplit % 64 || p.M % T || p.act != 0 || p.rowbias)
    return 1;
  if (G <= 0 || p.Nout % 128 || p.Npad != p.Nout || p.Nout % G) return 1;
  const int Cg = p.Nout / G;
  if (Cg % 8 || 128 % Cg) return 1;
  int gl = 0;
  while ((8 << gl) < Cg) gl++;
  if ((8 << gl) != Cg) return 1;
  if (mode < 0 || mode > 2 || (mode == 1 && (!rowbias || ldrb < p.Nout)) || (mode == 2 && (!res || ldr < p.Nout))) return -1;
  if (!out || !gamma || !beta || ldo < p.Nout || (write_y && (!p.C || p.ldc < p.Nout))) return -1;
  if (((size_t)out % 16) || (ldo % 8) || (write_y && (((size_t)p.C % 16) || p.ldc % 8)) ||
      (mode == 2 && (((size_t)res % 16) || ldr % 8)))
    return 1;
  {
    const size_t ldmax = (size_t)(p.ldx > p.ldx2 ? p.ldx : p.ldx2);
    if ((size_t)p.M * ldmax * 2 >= (1ull << 31) || (size_t)p.Nout * p.ldw * 2 >= (1ull << 31)) return 1;
  }
  GnEpi ge;
  ge.gamma = gamma; ge.beta = beta; ge.mean = mean; ge.rstd = rstd; ge.out = (bf16_t*)out; ge.ldo = ldo;
  ge.write_y = write_y ? 1 : 0; ge.mode = mode; ge.rowbias = rowbias; ge.ldrb = ldrb; ge.res = (const bf16_t*)res; ge.ldr = ldr;
  ge.rowvec = rowvec;
  ge.G = G; ge.gv_log2 = gl; ge.eps = eps;
  const int spt = BM / T, rows_used = spt * T;
  const int n_seq = p.M / T;
  dim3 grid((n_seq + spt - 1) / spt, p.Nout / 128);
  p.cks = ct_pick_cks(p, (long)grid.x * grid.y, variant);
  grid.z = p.cks;
  // epilogue geometry: VPL = ceil(T / 4) row iterations per lane, 8 / VPL sequences per wave must cover the tile's share
  const int vpl = (T + 3) / 4, nw = variant == 1 ? 4 : 8;
  if ((spt + nw - 1) / nw > 8 / vpl) return 1;
#define CT_GN(WM_, D_, V_) hipLaunchKernelGGL((conv_taps_bf16_kernel<WM_, false, D_, true, V_>), grid, dim3(WM_ * 128), 0, stream, p, rows_used, -2, 1, ge, ct_time_major(spt))
#define CT_GN_VPL(WM_, D_)                                    \
  switch (vpl) {                                              \
    case 2: CT_GN(WM_, D_, 2); break;                         \
    case 3: CT_GN(WM_, D_, 3); break;                         \
    case 4: CT_GN(WM_, D_, 4); break;                         \
    case 6: CT_GN(WM_, D_, 6); break;                         \
    case 7: CT_GN(WM_, D_, 7); break;                         \
    default: return 1;                                        \
  }
  if (variant == 2) { CT_GN_VPL(4, 5) }
  else if (variant == 1) { CT_GN_VPL(2, 3) }
  else return 1;               // (the 4-deep-ring form of the 192-row tile has no fused epilogue: use variant 2)
#undef CT_GN_VPL
#undef CT_GN
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// Data gradient of a k5 "same" convolution + the GroupNorm/Mish backward of the Conv1dBlock that produced its input, in one
// launch (conv_taps WT kernel with the GNB epilogue).  p: the data-gradient problem as madiff_gemm_dgrad_wt builds it
// (p.C / p.ldc receive dy of the preceding block instead of dh).  Returns 1 when the layer is not covered.
extern "C" __attribute__((visibility("hidden"))) int madiff_launch_conv_taps_gnb_bf16(
    const void* params, int variant, const void* ypre, int ldyp, const float* gamma, const float* beta, const float* mean,
    const float* rstd, int G, float* dgamma, float* dbeta, float* dbias, float* drowbias, int ldrb, const void* gadd, int ldga,
    void* dhout, int lddh, hipStream_t stream) {
  GemmFwdParams p = *(const GemmFwdParams*)params;
  const int BM = variant == 1 ? 96 : 192;
  const int T = p.rm.T_out;
  if (variant == 2 && p.Cin % 128) return 1;
  if (!p.wt || p.taps != 5 || p.rm.stride != 1 || p.rm.tr != 1 || p.rm.pad != 2 || p.rm.T_in != T || T > 32 || p.permA > 0 ||
      p.ksplit > 1 || p.Cin % 64 || p.M % T || p.act != 0 || p.rowbias || p.bias || p.addend || p.X2)
    return 1;
  if (G <= 0 || p.Nout % 128 || p.Npad != p.Nout || p.Nout % G || p.wcols < p.Nout) return 1;
  const int Cg = p.Nout / G;
  if (Cg % 8 || 128 % Cg) return 1;
  int gl = 0;
  while ((8 << gl) < Cg) gl++;
  if ((8 << gl) != Cg) return 1;
  if (!ypre || !gamma || !beta || !mean || !rstd || !dgamma || !dbeta || !p.C || ldyp < p.Nout || p.ldc < p.Nout ||
      (drowbias && ldrb < p.Nout))
    return -1;
  if (((size_t)ypre % 16) || (ldyp % 8) || ((size_t)p.C % 16) || (p.ldc % 8) || (drowbias && (((size_t)drowbias % 16) || ldrb % 4)))
    return 1;
  if (gadd && !dhout) return -1;                               // the addend only exists in the chained form
  if (dhout && (lddh < p.Nout || (gadd && ldga < p.Nout))) return -1;
  if (dhout && (((size_t)dhout % 16) || lddh % 8 || (gadd && (((size_t)gadd % 16) || ldga % 8)))) return 1;
  {
    const size_t wext = (size_t)p.Cin * p.ldw * 2;
    if ((size_t)p.M * (size_t)p.ldx * 2 >= (1ull << 31) || wext >= (1ull << 31)) return 1;
  }
  GnEpi ge = {};
  ge.gamma = gamma; ge.beta = beta; ge.G = G; ge.gv_log2 = gl;
  ge.by = (const bf16_t*)ypre; ge.ldby = ldyp; ge.bmean = mean; ge.brstd = rstd;
  ge.dgamma = dgamma; ge.dbeta = dbeta; ge.dbias = dbias; ge.drowb = drowbias; ge.lddrb = ldrb;
  ge.gadd = (const bf16_t*)gadd; ge.ldga = ldga; ge.dhout = (bf16_t*)dhout; ge.lddh = lddh;
  const int spt = BM / T, rows_used = spt * T;
  const int n_seq = p.M / T;
  dim3 grid((n_seq + spt - 1) / spt, p.Nout / 128);
  p.cks = ct_pick_cks(p, (long)grid.x * grid.y, variant);
  grid.z = p.cks;
  const int vpl = (T + 3) / 4, nw = variant == 1 ? 4 : 8;
  if ((spt + nw - 1) / nw > 8 / vpl) return 1;
#define CT_GNB(WM_, D_, V_)                                                                                              \
  do {                                                                                                                  \
    if (dhout) hipLaunchKernelGGL((conv_taps_bf16_kernel<WM_, true, D_, false, V_, true, true>), grid, dim3(WM_ * 128), 0, stream, p, rows_used, 2, -1, ge, ct_time_major(spt)); \
    else hipLaunchKernelGGL((conv_taps_bf16_kernel<WM_, true, D_, false, V_, true>), grid, dim3(WM_ * 128), 0, stream, p, rows_used, 2, -1, ge, ct_time_major(spt)); \
  } while (0)
#define CT_GNB_VPL(WM_, D_)                                   \
  switch (vpl) {                                              \
    case 2: CT_GNB(WM_, D_, 2); break;                        \
    case 3: CT_GNB(WM_, D_, 3); break;                        \
    case 4: CT_GNB(WM_, D_, 4); break;                        \
    case 6: CT_GNB(WM_, D_, 6); break;                        \
    case 7: CT_GNB(WM_, D_, 7); break;                        \
    default: return 1;                                        \
  }
  if (variant == 2) { CT_GNB_VPL(4, 5) }
  else if (variant == 1) { CT_GNB_VPL(2, 3) }
  else return 1;
#undef CT_GNB_VPL
#undef CT_GNB
  MADIFF_CHECK_LAUNCH();
  return 0;
}
