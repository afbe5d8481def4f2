// Implicit-GEMM convolution / linear kernels on CDNA4 MFMA (gfx950).
//
// Activations are channels-last matrices [rows = (sequence n, time t), cols = channels] whose column
// count is padded to a multiple of 32 with zeros.  One kernel family covers every dense contraction of
// the denoiser and of the inverse-dynamics MLP:
//
//   gemm_fwd   C[m, o] = sum_{tap, c} X[src(m, tap), c] * W[o, tap*Cin + c]  (+bias +rowbias, act)
//              Conv1d k5 (stride 1), Downsample k3 s2, ConvTranspose k4 s2 (tr=1), 1x1 conv, Linear,
//              and every data-gradient (tr flips the tap direction) are all this kernel with a RowMap.
//   gemm_wgrad dW[o, tap*Cin + c] = sum_m dY[m, o] * X[src(m, tap), c]       (fp32, split over m)
//
// bf16 build: __builtin_amdgcn_mfma_f32_16x16x32_bf16, fp32 accumulate.
// fp32 build: __builtin_amdgcn_mfma_f32_16x16x4f32 (exact fp32 fmaf chain).
// Tiles are 128x128 per 256-thread workgroup (4 waves, 64x64 each), operands staged through LDS with
// register prefetch + double buffering; the conv taps are realised as shifted row gathers while
// staging, so no im2col matrix ever exists in memory.
#include <mutex>
#include "gemm_params.h"
#include <stdlib.h>

template <typename T> struct MfmaTraits;
template <> struct MfmaTraits<bf16_t> { static constexpr int KC = 32; };
template <> struct MfmaTraits<float>  { static constexpr int KC = 16; };

__device__ __forceinline__ void mma_step(const bf16x8 (&a)[4], const bf16x8 (&b)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
}
__device__ __forceinline__ void mma_step(const f32x4 (&a)[4], const f32x4 (&b)[4], f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
}

template <typename T> struct FragT;
template <> struct FragT<bf16_t> { typedef bf16x8 type; };
template <> struct FragT<float>  { typedef f32x4 type; };

#define FWD_ROWB 80  // LDS bytes per tile row: 64 B of k + 16 B pad

template <typename T, typename OutT>
__global__ __launch_bounds__(256) void gemm_fwd_kernel(GemmFwdParams p) {
  constexpr int KC = MfmaTraits<T>::KC;
  typedef typename FragT<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][256 * FWD_ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;

  // staging assignment: 2 A chunks + 2 W chunks (16 B each) per thread per K-step
  int a_n[2], a_t[2], w_row[2];
  bool a_ok[2];
  const int kc = tid & 3;  // 16-byte chunk inside the 64-byte k-slab
#pragma unroll
  for (int j = 0; j < 2; j++) {
    int row = (tid >> 2) + 64 * j;
    int m = m0 + row;
    a_ok[j] = m < p.M;
    int n = m / p.rm.T_out;
    a_n[j] = n;
    a_t[j] = m - n * p.rm.T_out;
    w_row[j] = n0 + row;
  }
  const int steps_per_tap = p.Cin / KC;
  const int nsteps = steps_per_tap * p.taps;

  u32x4 ra[2], rw[2];
  auto load_step = [&](int s) {
    int tap = s / steps_per_tap;
    int c0 = (s - tap * steps_per_tap) * KC;
    const T* src = (const T*)p.X;
    int ld = p.ldx, c = c0;
    if (c0 >= p.Csplit) { src = (const T*)p.X2; ld = p.ldx2; c = c0 - p.Csplit; }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (a_ok[j]) {
        int r = map_row(a_n[j], a_t[j], tap, p.rm);
        if (r >= 0) v = *(const u32x4*)((const unsigned char*)(src + (size_t)r * ld + c) + kc * 16);
      }
      ra[j] = v;
      u32x4 w = {0u, 0u, 0u, 0u};
      if (w_row[j] < p.Nout)
        w = *(const u32x4*)((const unsigned char*)((const T*)p.W + (size_t)w_row[j] * p.ldw + (size_t)s * KC) + kc * 16);
      rw[j] = w;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int row = (tid >> 2) + 64 * j;
      *(u32x4*)(&smem[buf][row * FWD_ROWB + kc * 16]) = ra[j];
      *(u32x4*)(&smem[buf][(128 + row) * FWD_ROWB + kc * 16]) = rw[j];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int s = 0; s < nsteps; s++) {
    const int buf = s & 1;
    if (s + 1 < nsteps) load_step(s + 1);
    frag_t a[4], b[4];
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      a[i] = *(const frag_t*)(&smem[buf][(wm * 64 + i * 16 + fr) * FWD_ROWB + fq * 16]);
      b[i] = *(const frag_t*)(&smem[buf][(128 + wn * 64 + i * 16 + fr) * FWD_ROWB + fq * 16]);
    }
    mma_step(a, b, acc);
    if (s + 1 < nsteps) store_step(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[row = (lane>>4)*4 + r][col = lane&15]
  OutT* C = (OutT*)p.C;
  const int fq = lane >> 4, fr = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      int m = m0 + wm * 64 + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
      int seq = m / p.rm.T_out;
      size_t orow = m;
      if (p.permA > 0) {
        int t = m - seq * p.permT;          // permT == T_out here
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

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[o, tap*Cin + c] = sum_m dY[m, o] * X[src(m, tap), c]
// ------------------------------------------------------------------------------------------------
template <typename T> struct WgradTraits;
template <> struct WgradTraits<bf16_t> { static constexpr int KR = 32, ROWB = 288, CPR = 16, EPC = 8; };
template <> struct WgradTraits<float>  { static constexpr int KR = 16, ROWB = 528, CPR = 32, EPC = 4; };

template <typename T>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(GemmWgradParams p) {
  typedef WgradTraits<T> TR;
  constexpr int KR = TR::KR, ROWB = TR::ROWB, CPR = TR::CPR, EPC = TR::EPC;
  constexpr int TILEB = KR * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2 * TILEB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wc = wave & 1;
  const int n0 = blockIdx.x * 128;
  const int tap = blockIdx.y / p.ctiles;
  const int c0 = (blockIdx.y - tap * p.ctiles) * 128;
  const int total_steps = (p.M + KR - 1) / KR;
  const int s_begin = blockIdx.z * p.steps_per_split;
  int s_end = s_begin + p.steps_per_split;
  if (s_end > total_steps) s_end = total_steps;
  if (s_begin >= s_end) return;   // uniform for the whole workgroup

  // which source holds channels [c0, c0+128)?  (Csplit is a multiple of 128 or equals Cin)
  const T* xsrc = (const T*)p.X;
  int ldx = p.ldx, cx = c0, cmax = p.Csplit;
  if (c0 >= p.Csplit) { xsrc = (const T*)p.X2; ldx = p.ldx2; cx = c0 - p.Csplit; cmax = p.Cin - p.Csplit; }

  u32x4 ry[2], rx[2];
  auto load_step = [&](int s) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int c = tid + 256 * j;
      int row = c / CPR, cc = c - row * CPR;
      int m = s * KR + row;
      u32x4 vy = {0u, 0u, 0u, 0u}, vx = {0u, 0u, 0u, 0u};
      if (m < p.M) {
        int col = n0 + cc * EPC;
        if (col < p.Ycols) vy = *(const u32x4*)((const T*)p.dY + (size_t)m * p.ldy + col);
        int xc = cx + cc * EPC;
        if (xc < cmax) {
          int n = m / p.rm.T_out;
          int r = map_row(n, m - n * p.rm.T_out, tap, p.rm);
          if (r >= 0) vx = *(const u32x4*)(xsrc + (size_t)r * ldx + xc);
        }
      }
      ry[j] = vy;
      rx[j] = vx;
    }
  };
  auto store_step = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int c = tid + 256 * j;
      int row = c / CPR, cc = c - row * CPR;
      *(u32x4*)(&smem[buf][row * ROWB + cc * 16]) = ry[j];
      *(u32x4*)(&smem[buf][TILEB + row * ROWB + cc * 16]) = rx[j];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  load_step(s_begin);
  store_step(0);
  __syncthreads();
  for (int s = s_begin; s < s_end; s++) {
    const int buf = (s - s_begin) & 1;
    if (s + 1 < s_end) load_step(s + 1);
    const unsigned char* Ys = &smem[buf][0];
    const unsigned char* Xs = &smem[buf][TILEB];
    if constexpr (sizeof(T) == 2) {
      // transposed LDS reads: lane (group fq, r = fr) supplies row 8*fq + (r>>2) (+4), cols 4*(r&3)..+3 and
      // receives column r of those 4 rows -> 8 consecutive reduction rows for its own output row/col.
      bf16x8 a[4], b[4];
      const int trow = 8 * fq + (fr >> 2), tcol = 4 * (fr & 3);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        typedef s16x4 __attribute__((address_space(3))) * lds_p;
        const unsigned char* pa = Ys + trow * ROWB + (wn * 64 + i * 16 + tcol) * 2;
        const unsigned char* pb = Xs + trow * ROWB + (wc * 64 + i * 16 + tcol) * 2;
        s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pa));
        s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pa + 4 * ROWB));
        s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pb));
        s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(pb + 4 * ROWB));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        a[i] = __builtin_bit_cast(bf16x8, av);
        b[i] = __builtin_bit_cast(bf16x8, bv);
      }
      mma_step(a, b, acc);
    } else {
      // fp32: MFMA s consumes reduction rows {4*fq + s}; one ds_read_b32 per operand per MFMA.
      f32x4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
          a[i][s4] = *(const float*)(Ys + (4 * fq + s4) * ROWB + (wn * 64 + i * 16 + fr) * 4);
          b[i][s4] = *(const float*)(Xs + (4 * fq + s4) * ROWB + (wc * 64 + i * 16 + fr) * 4);
        }
      mma_step(a, b, acc);
    }
    if (s + 1 < s_end) store_step(buf ^ 1);
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

// ------------------------------------------------------------------------------------------------
// weight packing: parameter layout (fp32) <-> GEMM layout [O, taps, Ipad]
// ------------------------------------------------------------------------------------------------
// dst[(o*taps + tap)*dst_ld + dst_off + i] = src[o*so + i*si + tap*st]; pad columns are never touched (the
// host allocates the packed buffer zero-filled once), so several parameters can be packed side by side.
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int O, int I, int taps, long so,
                                   long si, long st, int dst_ld, int dst_off) {
  long total = (long)O * taps * I;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    int i = e % I;
    long r = e / I;
    int tap = r % taps;
    long o = r / taps;
    dst[(o * taps + tap) * dst_ld + dst_off + i] = from_f32<T>(src[o * so + i * si + tap * st]);
  }
}

// Multi-tensor variant: ONE launch repacks every stale weight of the model (a train step would otherwise issue
// ~140 tiny pack launches).  table: n_desc rows of 12 int64 {src, dst, O, I, taps, so, si, st, dst_ld, dst_off,
// n_elems, first_block}; each block handles PACK_EPB consecutive elements of one descriptor.
#define PACK_EPB 2048
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const long* __restrict__ table, int n_desc) {
  int lo = 0, hi = n_desc - 1;
  const long b = blockIdx.x;
  while (lo < hi) {                       // last descriptor whose first_block <= b
    int mid = (lo + hi + 1) >> 1;
    if (table[mid * 12 + 11] <= b) lo = mid; else hi = mid - 1;
  }
  const long* d = table + lo * 12;
  const float* src = (const float*)d[0];
  T* dst = (T*)d[1];
  const int I = (int)d[3], taps = (int)d[4];
  const long so = d[5], si = d[6], st = d[7], dst_ld = d[8], dst_off = d[9], total = d[10];
  const long e0 = (b - d[11]) * PACK_EPB;
  for (int k = threadIdx.x; k < PACK_EPB; k += 256) {
    long e = e0 + k;
    if (e >= total) break;
    int i = e % I;
    long r = e / I;
    int tap = r % taps;
    long o = r / taps;
    dst[(o * taps + tap) * dst_ld + dst_off + i] = from_f32<T>(src[o * so + i * si + tap * st]);
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int madiff_launch_gemm_fwd_bf16(const void* params, int out_f32, int tile, hipStream_t stream);
extern "C" int madiff_launch_gemm_wgrad_bf16(const void* params, int tile, int gx, int gy, int gz, hipStream_t stream);
extern "C" int madiff_launch_gemm_wgrad_taps_bf16(const void* params, int variant, int nq, int gx, int gy, int gz, hipStream_t stream);
extern "C" int madiff_launch_conv_taps_bf16(const void* params, int out_f32, int variant, hipStream_t stream);
extern "C" int madiff_launch_conv_taps_gn_bf16(const void* params, int variant, void* out, int ldo, const float* gamma,
                                               const float* beta, float* mean, float* rstd, int G, float eps, int mode,
                                               const float* rowbias, int ldrb, const void* res, int ldr, int write_y,
                                               const float* rowvec, hipStream_t stream);

// MADIFF_GEMM_IMPL=0 forces the generic register-staged kernel for bf16 too (A/B timing, debugging)
static int use_dma_kernel() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MADIFF_GEMM_IMPL"); v = (e && e[0] == '0') ? 0 : 1; }
  return v;
}

static int check_rowmap(int T_out, int T_in, int stride, int pad, int tr) {
  return (T_out > 0 && T_in > 0 && stride > 0 && pad >= 0 && (tr == 0 || tr == 1)) ? 0 : -1;
}

// Default tile choice of the LDS-DMA kernel when the caller passes tile < 0 (the Python host autotunes per layer
// shape and passes an explicit tile; this model is the fallback, e.g. under graph capture of an unseen shape).
// Busiest CU holds nb = ceil(tiles / 256) workgroups; a lone workgroup runs at ~60 % of a CU's paired rate; smaller
// tiles re-read more operand bytes per FLOP (eff).
// ids 14, 15: the all-taps k5 kernel (conv_taps_bf16.hip), 192- and 96-row tiles; they cover stride-1 "same" k5 layers
// only and the entry points return 1 ("not covered") for any other layer
static const int kNumTiles = 17;
static const int kFirstTapsTile = 14;     // 14: 192 rows, 15: 96 rows, 16: 192 rows with one barrier per two steps
static const int kTileBM[kNumTiles] = {128, 96, 64, 128, 96, 64, 128, 96, 64, 128, 96, 64, 64, 64, 192, 96, 192};
static const int kTileBN[kNumTiles] = {128, 128, 128, 64, 64, 64, 128, 128, 128, 64, 64, 64, 128, 64, 128, 128, 128};
static const int kTileD[kNumTiles] = {2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 3, 5};
static int pick_tile(int M, int Npad, int wt) {
  static const double eff[6] = {1.0, 0.93, 0.85, 0.85, 0.78, 0.70};
  int best = 0;
  double best_t = 1e300;
  for (int t = 0; t < (wt ? 3 : 6); t++) {
    const int BM = kTileBM[t], BN = kTileBN[t];
    if (BN == 128 && Npad <= 64) continue;
    long tiles = (long)((M + BM - 1) / BM) * ((Npad + BN - 1) / BN);
    long nb = (tiles + 255) / 256;
    double thr = nb >= 2 ? 1.0 : 0.6;
    double cost = (double)nb * BM * BN / (thr * eff[t]);
    if (cost < best_t * 0.999) { best_t = cost; best = t; }
  }
  return best >= 1 ? best + 6 : best;      // three LDS stages for every shape but 128x128 (one workgroup per CU there)
}

// ---- workspace of the all-taps kernel's reduction split (conv_taps_bf16.hip, p.cks): registered once per device by the
// host (madiff_amd/ops.py allocates it on first use).  The partial tiles and tickets of a launch are indexed by tile id only,
// so two launches that may overlap on the GPU must not share them: the registered arena is cut into MADIFF_WS_SLOTS equal
// slots and a slot belongs to ONE issue order -- an eager stream (keyed by its handle), or one stream of one capture (keyed
// by the stream handle AND the capture's unique id, so two graphs never share a slot even when their capture streams reuse
// a handle, and neither do two parallel branches of one capture).  Launches of one stream / one captured branch are
// ordered by construction.  A slot is never handed on; when they are used up, later
// streams / captures simply run unsplit (cks = 1).  The counters reset themselves at the end of every launch.
#define MADIFF_MAX_DEV 64
#define MADIFF_WS_SLOTS 16
struct WsArena {
  float* ws; long slot_floats; int* cnt; int slot_ncnt;
  int used; unsigned long long key[MADIFF_WS_SLOTS], cap[MADIFF_WS_SLOTS];
};
static WsArena g_arena[MADIFF_MAX_DEV];
static std::mutex g_arena_mu;

extern "C" int madiff_set_splitk_workspace(void* ws, long bytes, int* counters, int n_counters) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MADIFF_MAX_DEV) return -1;
  if (ws && (((size_t)ws % 16) || bytes < 0 || !counters || n_counters <= 0)) return -1;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  WsArena& a = g_arena[dev];
  a.ws = (float*)ws; a.cnt = ws ? counters : nullptr; a.used = 0;
  a.slot_floats = ws ? (bytes / 4 / MADIFF_WS_SLOTS) & ~3L : 0;          // slots stay 16-byte aligned
  a.slot_ncnt = ws ? n_counters / MADIFF_WS_SLOTS : 0;
  return 0;
}

// number of slots handed out on the current device so far (tests)
extern "C" int madiff_splitk_slots_used(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MADIFF_MAX_DEV) return -1;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  return g_arena[dev].used;
}

static void attach_ws(GemmFwdParams& p, hipStream_t stream) {
  int dev = -1;
  p.ws = nullptr; p.ws_floats = 0; p.ws_cnt = nullptr; p.ws_ncnt = 0; p.cks = 1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MADIFF_MAX_DEV) return;
  // issue order this launch belongs to: (stream, 0) eagerly, (stream, capture id) while the stream is capturing
  const unsigned long long key = (unsigned long long)(size_t)stream;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  unsigned long long cid = 0;
  if (!(hipStreamGetCaptureInfo(stream, &cs, &cid) == hipSuccess && cs == hipStreamCaptureStatusActive)) cid = 0;
  else cid |= 1ull << 63;
  std::lock_guard<std::mutex> lk(g_arena_mu);
  WsArena& a = g_arena[dev];
  if (!a.ws || a.slot_floats <= 0 || a.slot_ncnt <= 0) return;
  int slot = -1;
  for (int i = 0; i < a.used; i++)
    if (a.key[i] == key && a.cap[i] == cid) { slot = i; break; }
  if (slot < 0) {
    if (a.used >= MADIFF_WS_SLOTS) return;             // every slot has an owner: this issue order runs unsplit
    slot = a.used++;
    a.key[slot] = key; a.cap[slot] = cid;
  }
  p.ws = a.ws + (size_t)slot * a.slot_floats; p.ws_floats = a.slot_floats;
  p.ws_cnt = a.cnt + (size_t)slot * a.slot_ncnt; p.ws_ncnt = a.slot_ncnt;
}

static int xcd_on() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("MADIFF_XCD_REMAP"); v = e ? atoi(e) : 1; }
  return v;
}

extern "C" int madiff_gemm_tile_count(void) { return kNumTiles; }
// rows / columns / LDS stages of kernel variant `tile`; returns 1 when the variant also exists for the transposed-weight
// data-gradient kernel (madiff_gemm_dgrad_wt), 0 when not, -1 for an invalid id
extern "C" int madiff_gemm_tile_info(int tile, int* bm, int* bn, int* stages) {
  if (tile < 0 || tile >= kNumTiles) return -1;
  if (bm) *bm = kTileBM[tile];
  if (bn) *bn = kTileBN[tile];
  if (stages) *stages = kTileD[tile];
  return kTileBN[tile] == 128 ? 1 : 0;
}

extern "C" int madiff_gemm_fwd(int dtype, int out_f32, const void* X, int ldx, const void* X2, int ldx2, int Csplit,
                               int M, int Cin, int taps, int T_out, int T_in, int stride, int pad, int tr,
                               const void* W, int ldw, int Nout, int Npad, const float* bias, const float* rowbias,
                               int ldrb, void* C, int ldc, int act, int permT, int permA, int tile,
                               hipStream_t stream) {
  const int KC = dtype == MADIFF_BF16 ? 32 : 16;
  if (M <= 0 || Nout <= 0) return 0;
  if (check_rowmap(T_out, T_in, stride, pad, tr) || Cin <= 0 || Cin % KC || taps <= 0 || Npad < Nout || ldc < Npad ||
      ldw < taps * Cin || (ldw * (dtype == MADIFF_BF16 ? 2 : 4)) % 16 || ldx < 0 || M % T_out || tile >= kNumTiles)
    return -1;
  if (X2 == nullptr) Csplit = Cin;
  if (Csplit % KC || Csplit > Cin || ldx < Csplit || (X2 && ldx2 < Cin - Csplit)) return -1;
  if ((ldx * (dtype == MADIFF_BF16 ? 2 : 4)) % 16 || (X2 && (ldx2 * (dtype == MADIFF_BF16 ? 2 : 4)) % 16)) return -1;
  if (permA > 0 && (permT != T_out || (M / T_out) % permA)) return -1;
  GemmFwdParams p;
  p.X = X; p.ldx = ldx; p.X2 = X2; p.ldx2 = ldx2; p.Csplit = Csplit;
  p.M = M; p.Cin = Cin; p.taps = taps;
  p.rm = RowMap{T_out, T_in, stride, pad, tr};
  p.W = W; p.ldw = ldw; p.Nout = Nout; p.Npad = Npad; p.bias = bias; p.rowbias = rowbias; p.ldrb = ldrb;
  p.C = C; p.ldc = ldc; p.act = act; p.permT = permT; p.permA = permA;
  p.wt = 0; p.wtap = 0; p.wcols = 0; p.ksplit = 0; p.addend = nullptr; p.lda = 0; p.xcd = xcd_on(); attach_ws(p, stream);
  dim3 grid((M + 127) / 128, (Npad + 127) / 128), block(256);
  if (tile >= kFirstTapsTile) {
    if (dtype != MADIFF_BF16 || !use_dma_kernel()) return 1;
    return madiff_launch_conv_taps_bf16(&p, out_f32, tile - kFirstTapsTile, stream);
  }
  if (dtype == MADIFF_BF16 && Cin % 64 == 0 && Csplit % 64 == 0 && use_dma_kernel())
    return madiff_launch_gemm_fwd_bf16(&p, out_f32, tile >= 0 ? tile : pick_tile(M, Npad, 0), stream);
  if (dtype == MADIFF_BF16) {
    if (out_f32) hipLaunchKernelGGL((gemm_fwd_kernel<bf16_t, float>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, stream, p);
  } else {
    hipLaunchKernelGGL((gemm_fwd_kernel<float, float>), grid, block, 0, stream, p);
  }
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// Conv1dBlock forward (helpers.py:50-74) in ONE launch, bf16 build: Conv1d(k=5, padding=2) + bias -> GroupNorm(G) -> Mish,
// then + rowbias[n, c] (mode 1: the time-embedding projection, temporal.py:197) or + res[(n,t), c] (mode 2: the
// residual branch, temporal.py:200).  Y (pre-norm conv output, bf16 [M, ldy]) is what backward needs; pass Y = NULL
// under no_grad to skip that write.  mean / rstd [M/T * G] may be NULL together with Y.  variant: 0 = 192-row tile,
// 1 = 96-row tile.  Returns 1 when the layer is not covered (caller: madiff_gemm_fwd + madiff_gn_mish_fwd).
// madiff_conv_gn_mish_fwd_rv: the same with rowvec [Nout] fp32 (mode 1 only, 16-byte aligned, or NULL) added next to the
// row bias: + rowbias[n, c] + rowvec[c].  For the sampler, where the time half of Linear(Mish(t_emb)) is ONE vector per
// denoising step (every trajectory of the batch is at the same step) and the returns half is constant over the K steps.
extern "C" int madiff_conv_gn_mish_fwd_rv(const void* X, int ldx, const void* X2, int ldx2, int Csplit, int M, int Cin, int T,
                                          const void* W, int ldw, int Nout, const float* bias, void* Y, int ldy, void* out,
                                          int ldo, const float* gamma, const float* beta, float* mean, float* rstd, int G,
                                          float eps, int mode, const float* rowbias, int ldrb, const void* res, int ldr,
                                          int variant, const float* rowvec, hipStream_t stream);
extern "C" int madiff_conv_gn_mish_fwd(const void* X, int ldx, const void* X2, int ldx2, int Csplit, int M, int Cin, int T,
                                       const void* W, int ldw, int Nout, const float* bias, void* Y, int ldy, void* out,
                                       int ldo, const float* gamma, const float* beta, float* mean, float* rstd, int G,
                                       float eps, int mode, const float* rowbias, int ldrb, const void* res, int ldr,
                                       int variant, hipStream_t stream) {
  return madiff_conv_gn_mish_fwd_rv(X, ldx, X2, ldx2, Csplit, M, Cin, T, W, ldw, Nout, bias, Y, ldy, out, ldo, gamma, beta, mean,
                                    rstd, G, eps, mode, rowbias, ldrb, res, ldr, variant, nullptr, stream);
}
extern "C" int madiff_conv_gn_mish_fwd_rv(const void* X, int ldx, const void* X2, int ldx2, int Csplit, int M, int Cin, int T,
                                          const void* W, int ldw, int Nout, const float* bias, void* Y, int ldy, void* out,
                                          int ldo, const float* gamma, const float* beta, float* mean, float* rstd, int G,
                                          float eps, int mode, const float* rowbias, int ldrb, const void* res, int ldr,
                                          int variant, const float* rowvec, hipStream_t stream) {
  if (M <= 0 || Nout <= 0) return 0;
  if (!use_dma_kernel()) return 1;
  if (T <= 0 || Cin <= 0 || M % T || ldw < 5 * Cin || (ldw * 2) % 16 || (ldx * 2) % 16 || variant < 0 || variant > 2) return -1;
  if (X2 == nullptr) Csplit = Cin;
  if (Csplit > Cin || ldx < Csplit || (X2 && (ldx2 < Cin - Csplit || (ldx2 * 2) % 16))) return -1;
  GemmFwdParams p;
  p.X = X; p.ldx = ldx; p.X2 = X2; p.ldx2 = ldx2; p.Csplit = Csplit;
  p.M = M; p.Cin = Cin; p.taps = 5;
  p.rm = RowMap{T, T, 1, 2, 0};
  p.W = W; p.ldw = ldw; p.Nout = Nout; p.Npad = Nout; p.bias = bias; p.rowbias = nullptr; p.ldrb = 0;
  p.C = Y; p.ldc = ldy; p.act = 0; p.permT = 0; p.permA = 0;
  p.wt = 0; p.wtap = 0; p.wcols = 0; p.ksplit = 0; p.vec_epi = 1; p.addend = nullptr; p.lda = 0; p.xcd = xcd_on(); attach_ws(p, stream);
  return madiff_launch_conv_taps_gn_bf16(&p, variant, out, ldo, gamma, beta, mean, rstd, G, eps, mode, rowbias, ldrb, res,
                                         ldr, Y != nullptr, rowvec, stream);
}

extern "C" int madiff_gemm_dgrad_wt_add(const void* dY, int ldy, int M, int Kred, int taps, int T_out, int T_in, int stride,
                                        int pad, int tr, const void* W, int ldw, int wtap, int wcol0, int Ncols, int Npad,
                                        void* dX, int ldx, int tile, int ksplit, const void* addend, int ld_add,
                                        hipStream_t stream);

// Data gradient of a layer straight from its FORWARD bf16 weight matrix W [Kred = Opad rows, ldw = taps*wtap]:
//   dX[m, n] = sum_{tap, o} dY[src(m, tap), o] * W[o, tap*wtap + wcol0 + n],   n < Ncols
// (gfx950 LDS-DMA kernel only; no transposed weight copy).  Requirements: bf16, Kred % 64 == 0 and every one of
// the Kred rows of W is a real weight row; wcol0 % 8 == 0.  Returns 1 when the shape is not covered (the caller
// then uses madiff_gemm_fwd on a transposed pack).
// ksplit > 1: the reduction is cut into ksplit slices that combine with fp32 atomics; dX is then FP32 [M, ldx]
// (zeroed here first).  For few-row, long-reduction shapes (the time-MLP bundle: 1024 x 256 output, K = 11520).
extern "C" int madiff_gemm_dgrad_wt(const void* dY, int ldy, int M, int Kred, int taps, int T_out, int T_in, int stride,
                                    int pad, int tr, const void* W, int ldw, int wtap, int wcol0, int Ncols, int Npad,
                                    void* dX, int ldx, int tile, int ksplit, hipStream_t stream) {
  return madiff_gemm_dgrad_wt_add(dY, ldy, M, Kred, taps, T_out, T_in, stride, pad, tr, W, ldw, wtap, wcol0, Ncols, Npad, dX,
                                  ldx, tile, ksplit, nullptr, 0, stream);
}

// The same with an addend: dX = (the data gradient) + addend[M, ld_add] (bf16, pad columns zero).  This is how the two
// gradient contributions of a tensor with two consumers (temporal.py:197-200: x feeds blocks[0] and residual_conv) merge
// without a separate add pass.  Returns 1 when the chosen kernel has no vector epilogue for this layer (caller adds).
extern "C" int madiff_gemm_dgrad_wt_add(const void* dY, int ldy, int M, int Kred, int taps, int T_out, int T_in, int stride,
                                        int pad, int tr, const void* W, int ldw, int wtap, int wcol0, int Ncols, int Npad,
                                        void* dX, int ldx, int tile, int ksplit, const void* addend, int ld_add,
                                        hipStream_t stream) {
  if (M <= 0 || Ncols <= 0) return 0;
  if (addend && (ksplit > 1 || ld_add < Npad || ld_add % 8 || ((size_t)addend % 16))) return 1;
  if (!use_dma_kernel() || Kred % 64 || wcol0 % 8 || wtap % 8 || ldw % 8) return 1;
  if (check_rowmap(T_out, T_in, stride, pad, tr) || taps <= 0 || ldy < Kred || ldy % 8 || Npad < Ncols || ldx < Npad ||
      wcol0 < 0 || wcol0 + Ncols > wtap || ldw < taps * wtap || M % T_out || tile >= kNumTiles ||
      (tile >= 0 && kTileBN[tile] != 128))
    return -1;
  GemmFwdParams p;
  p.X = dY; p.ldx = ldy; p.X2 = nullptr; p.ldx2 = 0; p.Csplit = Kred;
  p.M = M; p.Cin = Kred; p.taps = taps;
  p.rm = RowMap{T_out, T_in, stride, pad, tr};
  p.W = (const unsigned char*)W + (size_t)wcol0 * 2; p.ldw = ldw; p.Nout = Ncols; p.Npad = Npad;
  p.bias = nullptr; p.rowbias = nullptr; p.ldrb = 0;
  p.C = dX; p.ldc = ldx; p.act = 0; p.permT = 0; p.permA = 0;
  p.wt = 1; p.wtap = wtap; p.wcols = wtap - wcol0; p.addend = addend; p.lda = ld_add; p.xcd = xcd_on(); attach_ws(p, stream);
  const int nsteps = (Kred / 64) * taps;
  p.ksplit = ksplit > 1 ? (ksplit < nsteps ? ksplit : nsteps) : 0;
  if (p.ksplit > 1) {
    hipError_t e = hipMemsetAsync(dX, 0, (size_t)M * ldx * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
    return madiff_launch_gemm_fwd_bf16(&p, 1, 8, stream);       // 64x128, 3 stages, fp32 atomics
  }
  if (tile >= kFirstTapsTile) return madiff_launch_conv_taps_bf16(&p, 0, tile - kFirstTapsTile, stream);
  return madiff_launch_gemm_fwd_bf16(&p, 0, tile >= 0 ? tile : pick_tile(M, Npad, 1), stream);
}

extern "C" int madiff_colsum(int dtype, const void* x, int ld, float* out, long M, int C, hipStream_t stream);
extern "C" int madiff_colsum_scatter(int dtype, const void* x, int ld, float* const* table, long M, int C, hipStream_t stream);
extern "C" int madiff_launch_conv_taps_gnb_bf16(const void* params, int variant, const void* ypre, int ldyp, const float* gamma,
                                                const float* beta, const float* mean, const float* rstd, int G, float* dgamma,
                                                float* dbeta, float* dbias, float* drowbias, int ldrb, const void* gadd,
                                                int ldga, void* dhout, int lddh, hipStream_t stream);

// Backward of "Conv1dBlock A -> k5 convolution B" in ONE launch (bf16 build): the data gradient of B (as
// madiff_gemm_dgrad_wt: dh = dY (*) W_B, straight from B's forward weight matrix) followed, in the kernel's epilogue, by the
// GroupNorm(G) + Mish backward of block A (helpers.py:63-71 differentiated; what madiff_gn_mish_bwd computes), whose
// pre-norm output Ypre and statistics its forward saved.  dX receives dy_A (gradient of A's pre-norm conv output), dh is
// never stored.  dgamma / dbeta / dbias ([C], dbias = gradient of A's conv bias, may be NULL) are accumulated atomically;
// drowbias ([M/T, ldrb] = sum_t dh, the gradient of the per-sequence term added after A: temporal.py:197) may be NULL.
// In ResidualTemporalBlock: A = blocks[0], B = blocks[1]'s convolution.  variant: 1 = 96-row tile, 2 = 192-row tile.
// Returns 1 when the layer is not covered (caller: madiff_gemm_dgrad_wt + madiff_gn_mish_bwd).
static int dgrad_gn_bwd_impl(const void* dY, int ldy, int M, int Kred, int T, const void* W, int ldw, int wtap, int Ncols,
                             const void* Ypre, int ldyp, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int G, void* dX, int ldx, float* dgamma, float* dbeta, float* dbias,
                             float* drowbias, int ldrb, const void* addend, int ld_add, void* dH, int ldh, int variant,
                             hipStream_t stream);

extern "C" int madiff_conv_dgrad_gn_bwd(const void* dY, int ldy, int M, int Kred, int T, const void* W, int ldw, int wtap,
                                        int Ncols, const void* Ypre, int ldyp, const float* gamma, const float* beta,
                                        const float* mean, const float* rstd, int G, void* dX, int ldx, float* dgamma,
                                        float* dbeta, float* dbias, float* drowbias, int ldrb, int variant,
                                        hipStream_t stream) {
  return dgrad_gn_bwd_impl(dY, ldy, M, Kred, T, W, ldw, wtap, Ncols, Ypre, ldyp, gamma, beta, mean, rstd, G, dX, ldx, dgamma,
                           dbeta, dbias, drowbias, ldrb, nullptr, 0, nullptr, 0, variant, stream);
}

// The same launch where dh is ALSO somebody's gradient: two consecutive ResidualTemporalBlocks (temporal.py:197-200 twice:
// the first block's output feeds the second one only).  B = the second block's blocks[0] convolution, A = the FIRST block's
// blocks[1] (GroupNorm + Mish + residual add).  dh = dY (*) W_B + addend (addend = the second block's residual-branch
// gradient, may be NULL) is the gradient of the first block's output: it is stored to dH (the first block's residual
// branch consumes it) and, in the same epilogue, taken through A's GroupNorm + Mish backward into dX.  Replaces
// madiff_gemm_dgrad_wt_add + madiff_gn_mish_bwd.  Status as madiff_conv_dgrad_gn_bwd.
extern "C" int madiff_conv_dgrad_gn_bwd_chain(const void* dY, int ldy, int M, int Kred, int T, const void* W, int ldw, int wtap,
                                              int Ncols, const void* Ypre, int ldyp, const float* gamma, const float* beta,
                                              const float* mean, const float* rstd, int G, void* dX, int ldx, float* dgamma,
                                              float* dbeta, float* dbias, const void* addend, int ld_add, void* dH, int ldh,
                                              int variant, hipStream_t stream) {
  if (!dH) return -1;
  return dgrad_gn_bwd_impl(dY, ldy, M, Kred, T, W, ldw, wtap, Ncols, Ypre, ldyp, gamma, beta, mean, rstd, G, dX, ldx, dgamma,
                           dbeta, dbias, nullptr, 0, addend, ld_add, dH, ldh, variant, stream);
}

static int dgrad_gn_bwd_impl(const void* dY, int ldy, int M, int Kred, int T, const void* W, int ldw, int wtap, int Ncols,
                             const void* Ypre, int ldyp, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int G, void* dX, int ldx, float* dgamma, float* dbeta, float* dbias,
                             float* drowbias, int ldrb, const void* addend, int ld_add, void* dH, int ldh, int variant,
                             hipStream_t stream) {
  if (M <= 0 || Ncols <= 0) return 0;
  if (!use_dma_kernel() || Kred % 64 || wtap % 8 || ldw % 8) return 1;
  if (T <= 0 || M % T || ldy < Kred || ldy % 8 || ldx < Ncols || Ncols > wtap || ldw < 5 * wtap || variant < 1 || variant > 2)
    return -1;
  GemmFwdParams p;
  p.X = dY; p.ldx = ldy; p.X2 = nullptr; p.ldx2 = 0; p.Csplit = Kred;
  p.M = M; p.Cin = Kred; p.taps = 5;
  p.rm = RowMap{T, T, 1, 2, 1};
  p.W = W; p.ldw = ldw; p.Nout = Ncols; p.Npad = Ncols;
  p.bias = nullptr; p.rowbias = nullptr; p.ldrb = 0;
  p.C = dX; p.ldc = ldx; p.act = 0; p.permT = 0; p.permA = 0;
  p.wt = 1; p.wtap = wtap; p.wcols = wtap; p.ksplit = 0; p.vec_epi = 1; p.addend = nullptr; p.lda = 0; p.xcd = xcd_on(); attach_ws(p, stream);
  return madiff_launch_conv_taps_gnb_bf16(&p, variant, Ypre, ldyp, gamma, beta, mean, rstd, G, dgamma, dbeta, dbias, drowbias,
                                          ldrb, addend, ld_add, dH, ldh, stream);
}

static int wgrad_impl(int dtype, const void* dY, int ldy, int Ycols, const void* X, int ldx, const void* X2, int ldx2,
                      int Csplit, int M, int Cin, int taps, int T_out, int T_in, int stride, int pad, int tr, float* dW,
                      int ldw, int Nout, int accumulate, int tile, int splits, float* dbias, float* const* btab,
                      int* bias_fused, hipStream_t stream);

extern "C" int madiff_gemm_wgrad(int dtype, const void* dY, int ldy, int Ycols, const void* X, int ldx, const void* X2,
                                 int ldx2, int Csplit, int M, int Cin, int taps, int T_out, int T_in, int stride,
                                 int pad, int tr, float* dW, int ldw, int Nout, int accumulate, int tile, int splits,
                                 hipStream_t stream) {
  int fused = 0;
  return wgrad_impl(dtype, dY, ldy, Ycols, X, ldx, X2, ldx2, Csplit, M, Cin, taps, T_out, T_in, stride, pad, tr, dW, ldw, Nout,
                    accumulate, tile, splits, nullptr, nullptr, &fused, stream);
}

// Weight gradient AND bias gradient of a layer (group): dbias[o] += sum_m dY[m, o], or *bias_table[o] += ... when the
// group's biases live at scattered addresses (one pointer per output column, as madiff_colsum_scatter takes).  The per-tap
// LDS-DMA kernel produces the sums itself (one extra MFMA per dY fragment against an all-ones operand, no second pass
// over dY); for every other kernel the column-sum kernel is launched behind the weight gradient.  Status as madiff_gemm_wgrad.
extern "C" int madiff_gemm_wgrad_bias(int dtype, const void* dY, int ldy, int Ycols, const void* X, int ldx, const void* X2,
                                      int ldx2, int Csplit, int M, int Cin, int taps, int T_out, int T_in, int stride,
                                      int pad, int tr, float* dW, int ldw, int Nout, int accumulate, int tile, int splits,
                                      float* dbias, const void* bias_table, hipStream_t stream) {
  if (!dbias == !bias_table) return -1;                       // exactly one destination form
  int fused = 0;
  int rc = wgrad_impl(dtype, dY, ldy, Ycols, X, ldx, X2, ldx2, Csplit, M, Cin, taps, T_out, T_in, stride, pad, tr, dW, ldw,
                      Nout, accumulate, tile, splits, dbias, (float* const*)bias_table, &fused, stream);
  if (rc != 0 || fused || M <= 0 || Nout <= 0) return rc;
  return dbias ? madiff_colsum(dtype, dY, ldy, dbias, M, Nout, stream)
               : madiff_colsum_scatter(dtype, dY, ldy, (float* const*)bias_table, M, Nout, stream);
}

static int wgrad_impl(int dtype, const void* dY, int ldy, int Ycols, const void* X, int ldx, const void* X2, int ldx2,
                      int Csplit, int M, int Cin, int taps, int T_out, int T_in, int stride, int pad, int tr, float* dW,
                      int ldw, int Nout, int accumulate, int tile, int splits, float* dbias, float* const* btab,
                      int* bias_fused, hipStream_t stream) {
  static const int kBO[10] = {128, 128, 64, 64, 64, 64, 128, 128, 64, 64}, kBC[10] = {128, 64, 128, 64, 64, 64, 64, 64, 64, 64};
  const int es = dtype == MADIFF_BF16 ? 2 : 4;
  if (M <= 0 || Nout <= 0) return 0;
  if (check_rowmap(T_out, T_in, stride, pad, tr) || Cin <= 0 || Cin % 32 || taps <= 0 || ldw < taps * Cin ||
      Ycols % 32 || ldy < Ycols || (ldy * es) % 16 || (ldx * es) % 16 || M % T_out || tile >= 10)
    return -1;
  if (X2 == nullptr) Csplit = Cin;
  if (Csplit > Cin || ldx < Csplit || (X2 && (Csplit % 128 || ldx2 < Cin - Csplit || (ldx2 * es) % 16))) return -1;
  GemmWgradParams p;
  p.dY = dY; p.ldy = ldy; p.Ycols = Ycols; p.X = X; p.ldx = ldx; p.X2 = X2; p.ldx2 = ldx2; p.Csplit = Csplit;
  p.M = M; p.Cin = Cin; p.taps = taps; p.rm = RowMap{T_out, T_in, stride, pad, tr};
  p.dW = dW; p.ldw = ldw; p.Nout = Nout; p.accumulate = accumulate ? 1 : 0; p.xcd = xcd_on();
  p.dbias = nullptr; p.btab = nullptr;
  const bool dma = dtype == MADIFF_BF16 && use_dma_kernel() && T_out < 256 && M < (1 << 24);
  // all-taps kernel (variants 4..7): stride-1 "same" k5 convolutions whose sequence length divides 32 or 96 rows
  const int nq = (32 % T_out == 0) ? 1 : ((96 % T_out == 0) ? 3 : 0);
  const bool taps_ok = dma && taps == 5 && stride == 1 && tr == 0 && T_in == T_out && pad == 2 && nq > 0;
  if (tile >= 4 && !taps_ok) return 1;                 // variant not applicable to this layer (caller picks another)
  if (!dma || tile < 0) tile = 0;
  const int BO = kBO[tile], BC = kBC[tile];
  const bool all_taps = tile >= 4;
  p.ctiles = (Cin + BC - 1) / BC;
  const int KR = all_taps ? (tile >= 8 ? 64 : 32) * nq : (dma ? 64 : (dtype == MADIFF_BF16 ? 32 : 16));
  int total_steps = (M + KR - 1) / KR;
  int tiles = ((Nout + BO - 1) / BO) * p.ctiles * (all_taps ? 1 : taps);
  if (splits <= 0) {
    // every extra split costs one more fp32 atomic pass over dW (atomics run at ~1.3 TB/s chip-wide), so only split
    // the reduction as far as needed to put ~2 workgroups on each of the 256 CUs
    static int target = -1;
    if (target < 0) { const char* e = getenv("MADIFF_WGRAD_BLOCKS"); target = e ? atoi(e) : 384; }
    splits = (target + tiles / 2) / tiles;
    if (splits > total_steps / 4) splits = total_steps / 4;
  }
  if (splits > total_steps) splits = total_steps;
  if (splits < 1) splits = 1;
  p.steps_per_split = (total_steps + splits - 1) / splits;
  p.splits = (total_steps + p.steps_per_split - 1) / p.steps_per_split;
  if (p.splits > 1 && !p.accumulate) {
    hipError_t e = hipMemsetAsync(dW, 0, (size_t)Nout * ldw * sizeof(float), stream);
    if (e != hipSuccess) return (int)e;
  }
  if (all_taps)
    return madiff_launch_gemm_wgrad_taps_bf16(&p, tile, nq, (Nout + BO - 1) / BO, p.ctiles, p.splits, stream);
  dim3 grid((Nout + BO - 1) / BO, p.ctiles * taps, p.splits), block(256);
  if (dma) {
    p.dbias = dbias; p.btab = btab; *bias_fused = (dbias || btab) ? 1 : 0;
    return madiff_launch_gemm_wgrad_bf16(&p, tile, grid.x, grid.y, grid.z, stream);
  }
  if (dtype == MADIFF_BF16) hipLaunchKernelGGL((gemm_wgrad_kernel<bf16_t>), grid, block, 0, stream, p);
  else hipLaunchKernelGGL((gemm_wgrad_kernel<float>), grid, block, 0, stream, p);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_pack_weight(int dtype, const float* src, void* dst, int O, int I, int taps, long so, long si,
                                  long st, int dst_ld, int dst_off, hipStream_t stream) {
  if (O <= 0 || I <= 0 || taps <= 0 || dst_off < 0 || dst_ld < dst_off + I) return -1;
  long total = (long)O * taps * I;
  int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(blocks), dim3(256), 0, stream, src, (bf16_t*)dst, O, I, taps, so, si, st, dst_ld, dst_off);
  else
    hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(blocks), dim3(256), 0, stream, src, (float*)dst, O, I, taps, so, si, st, dst_ld, dst_off);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_pack_weights_multi(int dtype, const long* table, int n_desc, int total_blocks, hipStream_t stream) {
  if (n_desc <= 0 || total_blocks <= 0) return 0;
  if (!table) return -1;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((pack_weight_multi_kernel<bf16_t>), dim3(total_blocks), dim3(256), 0, stream, table, n_desc);
  else
    hipLaunchKernelGGL((pack_weight_multi_kernel<float>), dim3(total_blocks), dim3(256), 0, stream, table, n_desc);
  MADIFF_CHECK_LAUNCH();
  return 0;
}
