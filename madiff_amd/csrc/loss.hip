// The training loss of GaussianDiffusion.loss as three launches (two forward, one backward) instead of ~40 tiny ATen
// kernels.  Reference arithmetic:
//   denoiser part   models/helpers.py:243-250 (WeightedStateL2) + models/diffusion.py:328-330 (loss-mask reduction)
//   inverse dynamics models/diffusion.py:369-436 (compute_inv_loss, share_inv / joint_inv; continuous MSE or discrete
//                   cross-entropy over legal actions + inv_acc)
//   total           models/diffusion.py:492   loss = 0.5 * (diffuse_loss + inv_loss)
// Layouts: eps / noise [B,T,A,F] fp32; w [T,A,F]; masks [B,T,A] (fp64 as the dataset delivers them, or fp32);
// x [B,T,A,act+F] fp32 (actions first); legal [B,T,A,nout] fp32 or null; inverse-dynamics predictions pred: element
// (r = (b, t < T-1), a, o) at pred[r*pred_rs + a*pred_as + o] (shared MLP: one row per (r, a); joint MLP: one row per r).
#include <atomic>
#include "common.h"

namespace {

constexpr int INV_BLOCKS = 64;   // workgroups of the inverse-dynamics part (partial sums: INV_BLOCKS x 3 doubles of workspace)

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float mask_at(const void* masks, int f64, long i) {
  return f64 ? (float)((const double*)masks)[i] : ((const float*)masks)[i];
}

struct LossArgs {
  const float* eps; const float* noise; const float* w; const void* masks; int masks_f64;
  const float* pred; long pred_rs, pred_as; const float* x; const float* legal;
  int B, T, A, F, act, nout, discrete;
  int S;   // micro-batches folded along B (B % S == 0): the inverse-dynamics loss is normalised per micro-batch, as
           // gradient accumulation over S calls would (training.py:147-152); 1 = one batch
};

// One (r, a) row of the inverse-dynamics loss.  Returns the row loss (sum of squared errors | cross-entropy) and whether
// the arg-max over the (legal) logits hits the taken action; with GRAD the gradient of the row loss w.r.t. the
// predictions, times `scale`, is stored (illegal logits: 0).
template <bool GRAD>
__device__ __forceinline__ float inv_row(const LossArgs& p, long r, int a, int* hit, float* dpred, float scale) {
  const int b = (int)(r / (p.T - 1)), t = (int)(r - (long)b * (p.T - 1));
  const float* pr = p.pred + r * p.pred_rs + (long)a * p.pred_as;
  const float* xa = p.x + (((long)b * p.T + t) * p.A + a) * (p.act + p.F);
  float* dp = GRAD ? dpred + r * p.pred_rs + (long)a * p.pred_as : nullptr;
  if (!p.discrete) {
    float l = 0.f;
    for (int o = 0; o < p.nout; o++) {
      const float d = pr[o] - xa[o];
      l += d * d;
      if (GRAD) dp[o] = 2.f * d * scale;
    }
    *hit = 0;
    return l;
  }
  const float* lg = p.legal ? p.legal + (((long)b * p.T + t) * p.A + a) * p.nout : nullptr;
  const int target = (int)(long)xa[0];
  float mx = -INFINITY;
  int arg = 0;
  for (int o = 0; o < p.nout; o++) {
    const float v = (lg && lg[o] == 0.f) ? -1e10f : pr[o];
    if (v > mx) { mx = v; arg = o; }          // first maximum, as torch.argmax
  }
  float se = 0.f, vt = 0.f;
  for (int o = 0; o < p.nout; o++) {
    const float v = (lg && lg[o] == 0.f) ? -1e10f : pr[o];
    se += expf(v - mx);
    if (o == target) vt = v;
  }
  if (GRAD) {
    const float inv = 1.f / se;
    for (int o = 0; o < p.nout; o++) {
      const bool illegal = lg && lg[o] == 0.f;
      const float v = illegal ? -1e10f : pr[o];
      const float sm = expf(v - mx) * inv;
      dp[o] = illegal ? 0.f : (sm - (o == target ? 1.f : 0.f)) * scale;
    }
  }
  *hit = arg == target;
  return logf(se) + mx - vt;
}

// grid = B (denoiser part, one sample per workgroup) + INV_BLOCKS (inverse-dynamics rows, grid-stride)
__global__ __launch_bounds__(256) void loss_partials_kernel(LossArgs p, float* __restrict__ mse_sums, double* __restrict__ part) {
  __shared__ double red[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if ((int)blockIdx.x < p.B) {
    const int b = blockIdx.x, TA = p.T * p.A;
    const long base = (long)b * TA * p.F;
    float s = 0.f, u = 0.f, ms = 0.f;
    for (int i = threadIdx.x; i < TA * p.F; i += 256) {
      const float d = p.eps[base + i] - p.noise[base + i];
      const float l = d * d * p.w[i];
      u += l;
      s += l * mask_at(p.masks, p.masks_f64, (long)b * TA + i / p.F);
    }
    for (int i = threadIdx.x; i < TA; i += 256) ms += mask_at(p.masks, p.masks_f64, (long)b * TA + i);
    s = wave_sum(s); u = wave_sum(u); ms = wave_sum(ms);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ms; red[2][wave] = u; }
    __syncthreads();
    if (threadIdx.x < 3)
      mse_sums[b * 3 + threadIdx.x] = (float)(red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
    return;
  }
  const int blk = blockIdx.x - p.B;                        // INV_BLOCKS workgroups per micro-batch
  const int seg = blk / INV_BLOCKS, sub = blk - seg * INV_BLOCKS;
  const long rows = (long)(p.B / p.S) * (p.T - 1) * p.A;   // rows of one micro-batch
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (long e = (long)sub * 256 + threadIdx.x; e < rows; e += (long)INV_BLOCKS * 256) {
    const long eg = (long)seg * rows + e;
    const long r = eg / p.A;
    const int a = (int)(eg - r * p.A);
    const int b = (int)(r / (p.T - 1)), t = (int)(r - (long)b * (p.T - 1));
    const float m = mask_at(p.masks, p.masks_f64, ((long)b * p.T + t + 1) * p.A + a);     // loss_masks[:, 1:]
    int hit;
    const float l = inv_row<false>(p, r, a, &hit, nullptr, 0.f);
    s0 += (double)l * m; s1 += m; s2 += hit ? (double)m : 0.0;
  }
  s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; red[2][wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) part[blk * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// out (double) / out_f (float): {loss, a0_loss, inv_loss, inv_acc, inverse-dynamics mask sum, diffuse_loss}; with S > 1
// micro-batches these six are the means over the micro-batches (entry 4: the sum) and six more per micro-batch follow at
// [6 + 6 s ...] (what each separate loss() call would have returned).
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ mse_sums, const double* __restrict__ part,
                                                          int B, int T, int A, int F, int act, int discrete, int S,
                                                          double* __restrict__ out, float* __restrict__ out_f) {
  __shared__ double red[5][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Bs = B / S;
  double tot[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int sg = 0; sg < S; sg++) {
    double ratio = 0.0, unmasked = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int b = sg * Bs + threadIdx.x; b < (sg + 1) * Bs; b += 256) {
      ratio += (double)mse_sums[b * 3] / (double)mse_sums[b * 3 + 1];
      unmasked += (double)mse_sums[b * 3 + 2];
    }
    for (int i = threadIdx.x; i < INV_BLOCKS; i += 256) {
      const double* q = part + ((long)sg * INV_BLOCKS + i) * 3;
      s0 += q[0]; s1 += q[1]; s2 += q[2];
    }
    ratio = wave_sum_d(ratio); unmasked = wave_sum_d(unmasked); s0 = wave_sum_d(s0); s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    __syncthreads();                           // (the previous micro-batch's sums have been read)
    if (lane == 0) { red[0][wave] = ratio; red[1][wave] = unmasked; red[2][wave] = s0; red[3][wave] = s1; red[4][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double v[5];
      for (int k = 0; k < 5; k++) v[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
      const double diffuse = v[0] / ((double)Bs * F);
      const double a0 = v[1] / ((double)Bs * T * A * F);
      // continuous: mean over (rows, act) of (se * m) / mean over rows of m = S0 / (act * S1); discrete: S0 / S1
      const double inv = discrete ? v[2] / v[3] : v[2] / ((double)act * v[3]);
      const double acc = discrete ? v[4] / v[3] : 0.0;
      const double res[6] = {0.5 * (diffuse + inv), a0, inv, acc, v[3], diffuse};
      for (int k = 0; k < 6; k++) {
        tot[k] += k == 4 ? res[k] : res[k] / S;
        if (S > 1) { out[6 + 6 * sg + k] = res[k]; out_f[6 + 6 * sg + k] = (float)res[k]; }
      }
    }
  }
  if (threadIdx.x == 0)
    for (int k = 0; k < 6; k++) { out[k] = tot[k]; out_f[k] = (float)tot[k]; }
}

// d loss / d eps and d loss / d pred, times the upstream gradient *go.  grid-stride over both ranges in one launch.
__global__ __launch_bounds__(256) void loss_bwd_kernel(LossArgs p, const float* __restrict__ mse_sums, const double* __restrict__ out,
                                                       const void* __restrict__ go, int go_f64, float* __restrict__ d_eps,
                                                       float* __restrict__ d_pred) {
  const float g = 0.5f * (go_f64 ? (float)((const double*)go)[0] : ((const float*)go)[0]);
  const int TA = p.T * p.A;
  const long n_eps = (long)p.B * TA * p.F;
  const float ge = g * 2.f / ((float)p.B * (float)p.F);
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n_eps; e += (long)gridDim.x * 256) {
    const int b = (int)(e / ((long)TA * p.F));
    const int i = (int)(e - (long)b * TA * p.F);
    const float m = mask_at(p.masks, p.masks_f64, (long)b * TA + i / p.F);
    d_eps[e] = ge * (p.eps[e] - p.noise[e]) * p.w[i] * m / mse_sums[b * 3 + 1];
  }
  const long rows = (long)p.B * (p.T - 1) * p.A;
  const int Bs = p.B / p.S;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < rows; e += (long)gridDim.x * 256) {
    const long r = e / p.A;
    const int a = (int)(e - r * p.A);
    const int b = (int)(r / (p.T - 1)), t = (int)(r - (long)b * (p.T - 1));
    const float m = mask_at(p.masks, p.masks_f64, ((long)b * p.T + t + 1) * p.A + a);
    const float s1 = p.S > 1 ? (float)out[6 + 6 * (b / Bs) + 4] : (float)out[4];      // the row's micro-batch normalises it
    const float gi = (p.discrete ? g / s1 : g / ((float)p.act * s1)) / (float)p.S;
    int hit;
    inv_row<true>(p, r, a, &hit, d_pred, gi * m);
    // pad columns behind this agent's block (shared MLP: up to the row's leading dimension; joint MLP: behind the last agent)
    const long lo = (long)a * p.pred_as + p.nout, hi = a == p.A - 1 ? p.pred_rs : (long)(a + 1) * p.pred_as;
    for (long c = lo; c < hi; c++) d_pred[r * p.pred_rs + c] = 0.f;
  }
}

// Inverse-dynamics inputs (o_t, o_{t+1}) for t < T-1 (diffusion.py:381-384) in the GEMM's activation layout.
// joint = 0: row (b, t, a) = [obs(b,t,a) | obs(b,t+1,a)]; joint = 1: row (b, t) = concat over agents of the same pairs.
template <typename Tout>
__global__ void inv_pairs_kernel(const float* __restrict__ x, Tout* __restrict__ out, int ld, int B, int T, int A, int act, int F,
                                 int joint) {
  const long n_rows = (long)B * (T - 1) * (joint ? 1 : A);
  const long total = n_rows * ld;
  const int W = act + F;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % ld);
    const long row = e / ld;
    long bt;
    int a, j;
    bool ok;
    if (joint) { bt = row; a = c / (2 * F); j = c - a * 2 * F; ok = a < A; }
    else { bt = row / A; a = (int)(row - bt * A); j = c; ok = c < 2 * F; }
    float v = 0.f;
    if (ok) {
      const int b = (int)(bt / (T - 1)), t = (int)(bt - (long)b * (T - 1));
      const int tt = j < F ? t : t + 1, f = j < F ? j : j - F;
      v = x[(((long)b * T + tt) * A + a) * W + act + f];
    }
    out[e] = from_f32<Tout>(v);
  }
}

inline int nblk(long total) {
  long n = (total + 255) / 256;
  return (int)(n < 1 ? 1 : (n > 2048 ? 2048 : n));
}

}  // namespace

// Workspace of madiff_loss_fwd: MADIFF_LOSS_WS_DOUBLES doubles (partial sums of the inverse-dynamics part, written by the
// first launch and read by the second).  The _static form keeps it in a file-scope device array instead: see below.
#define MADIFF_LOSS_WS_DOUBLES (INV_BLOCKS * 3)

static int loss_fwd_impl(const LossArgs& p, float* mse_sums, double* ws, double* out, float* out_f, hipStream_t stream) {
  hipLaunchKernelGGL(loss_partials_kernel, dim3(p.B + INV_BLOCKS * p.S), dim3(256), 0, stream, p, mse_sums, ws);
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, stream, mse_sums, ws, p.B, p.T, p.A, p.F, p.act, p.discrete, p.S,
                     out, out_f);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

static int loss_check(const LossArgs& p) {
  if (p.B <= 0 || p.T < 2 || p.A <= 0 || p.F <= 0 || p.act <= 0 || p.nout <= 0 || p.nout > 256) return -1;
  if (p.S < 1 || p.S > 64 || p.B % p.S) return -1;
  if (!p.eps || !p.noise || !p.w || !p.masks || !p.pred || !p.x) return -1;
  if (p.pred_as < p.nout && p.A > 1) return -1;
  if (p.pred_rs < (long)(p.A - 1) * p.pred_as + p.nout) return -1;
  if (p.discrete && p.act != 1) return -1;
  if (!p.discrete && p.nout != p.act) return -1;
  return 0;
}

extern "C" int madiff_loss_ws_doubles(void) { return MADIFF_LOSS_WS_DOUBLES; }

extern "C" int madiff_loss_fwd(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                               const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B, int T,
                               int A, int F, int act, int nout, int discrete, float* mse_sums, double* workspace, double* out,
                               float* out_f, hipStream_t stream) {
  LossArgs p{eps, noise, w, masks, masks_f64, pred, pred_rs, pred_as, x, legal, B, T, A, F, act, nout, discrete, 1};
  if (loss_check(p) || !mse_sums || !workspace || !out || !out_f) return -1;
  return loss_fwd_impl(p, mse_sums, workspace, out, out_f, stream);
}

// ---- the same forward with NO workspace argument and no allocation anywhere: for callers that cannot allocate (code under
// hipGraph capture without a private pool, serving loops and harnesses that forbid hipMalloc).  The scratch -- the partial
// sums between the two launches and the per-sample sums [B,3] that madiff_loss_fwd hands back for its backward -- lives in a
// file-scope device array of fixed size, one copy per GPU, held until the process ends.  Forward only (loss evaluation /
// logging): the per-sample sums stay inside the library.  Every call on a GPU uses that GPU's one array: calls that overlap on
// one GPU (two streams, two threads) race; calls on one stream are ordered.  Nothing in it has to start at zero: both
// launches overwrite every element they later read.
#define MADIFF_LOSS_STATIC_MAX_B 16384
#define MADIFF_LOSS_STATIC_BYTES (MADIFF_LOSS_WS_DOUBLES * 8 + MADIFF_LOSS_STATIC_MAX_B * 3 * 4)
static __device__ __align__(256) unsigned char scratch[MADIFF_LOSS_STATIC_BYTES];   // 1.5 KB + 192 KB per GPU
#define MADIFF_MAX_DEVICES 64
// device address of `scratch` on each GPU (the array's name is a host-side stand-in).  Atomic: concurrent first calls from
// several threads may both resolve the symbol; they store the same value.
static std::atomic<void*> g_scratch_ptr[MADIFF_MAX_DEVICES];

// Resolves this GPU's copy of the scratch array.  Call once per device, with that device current, BEFORE any stream capture
// that contains madiff_loss_fwd_static: resolving a symbol may load the code object / allocate, which is not legal inside a
// capture.  Idempotent, thread-safe.
extern "C" int madiff_loss_static_init(void) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= MADIFF_MAX_DEVICES) return -1;
  if (g_scratch_ptr[dev].load(std::memory_order_acquire)) return 0;
  void* ptr = nullptr;
  e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess) return (int)e;
  if (!ptr) return -1;
  g_scratch_ptr[dev].store(ptr, std::memory_order_release);
  return 0;
}

extern "C" int madiff_loss_fwd_static(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                                      const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B,
                                      int T, int A, int F, int act, int nout, int discrete, double* out, float* out_f,
                                      hipStream_t stream) {
  LossArgs p{eps, noise, w, masks, masks_f64, pred, pred_rs, pred_as, x, legal, B, T, A, F, act, nout, discrete, 1};
  if (loss_check(p) || !out || !out_f) return -1;
  if (B > MADIFF_LOSS_STATIC_MAX_B) return -2;            // beyond the static scratch: use madiff_loss_fwd with a workspace
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= MADIFF_MAX_DEVICES) return -1;
  if (!g_scratch_ptr[dev].load(std::memory_order_acquire)) {
    // first call on this device.  Under capture the symbol must not be resolved here (it may load code / allocate):
    // -3 = "call madiff_loss_static_init() on this device before capturing"
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return -3;
    int rc = madiff_loss_static_init();
    if (rc) return rc;
  }
  unsigned char* base = (unsigned char*)g_scratch_ptr[dev].load(std::memory_order_acquire);
  double* ws = (double*)base;
  float* mse_sums = (float*)(base + MADIFF_LOSS_WS_DOUBLES * 8);
  return loss_fwd_impl(p, mse_sums, ws, out, out_f, stream);
}

extern "C" int madiff_loss_bwd(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                               const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B, int T,
                               int A, int F, int act, int nout, int discrete, const float* mse_sums, const double* out,
                               const void* go, int go_f64, float* d_eps, float* d_pred, hipStream_t stream) {
  LossArgs p{eps, noise, w, masks, masks_f64, pred, pred_rs, pred_as, x, legal, B, T, A, F, act, nout, discrete, 1};
  if (loss_check(p) || !mse_sums || !out || !go || !d_eps || !d_pred) return -1;
  const long n = (long)B * T * A * F;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, p, mse_sums, out, go, go_f64, d_eps, d_pred);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

extern "C" int madiff_inv_pairs(int dtype, const float* x, void* out, int ld, int B, int T, int A, int act, int F, int joint,
                                hipStream_t stream) {
  if (B <= 0) return 0;
  if (!x || !out || T < 2 || A <= 0 || F <= 0 || act < 0 || ld < (joint ? A : 1) * 2 * F) return -1;
  const long total = (long)B * (T - 1) * (joint ? 1 : A) * ld;
  if (dtype == MADIFF_BF16)
    hipLaunchKernelGGL((inv_pairs_kernel<bf16_t>), dim3(nblk(total)), dim3(256), 0, stream, x, (bf16_t*)out, ld, B, T, A, act, F, joint);
  else
    hipLaunchKernelGGL((inv_pairs_kernel<float>), dim3(nblk(total)), dim3(256), 0, stream, x, (float*)out, ld, B, T, A, act, F, joint);
  MADIFF_CHECK_LAUNCH();
  return 0;
}

// ---- S micro-batches of one optimizer step folded along the batch axis (the reference's gradient_accumulate_every,
// training.py:147-152: S separate loss() calls, each scaled by 1/S before backward).  One pass over the S*Bs samples gives
// the same gradient: the denoiser part is normalised per sample, the inverse-dynamics part per micro-batch (its mask mean,
// diffusion.py:407,418), which is what these entry points do.  workspace: segments * madiff_loss_ws_doubles() doubles;
// out / out_f: 6 * (1 + segments) values (layout at loss_finish_kernel); segments = 1 is madiff_loss_fwd / _bwd.
extern "C" int madiff_loss_fwd_seg(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                                   const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B,
                                   int T, int A, int F, int act, int nout, int discrete, int segments, float* mse_sums,
                                   double* workspace, double* out, float* out_f, hipStream_t stream) {
  LossArgs p{eps, noise, w, masks, masks_f64, pred, pred_rs, pred_as, x, legal, B, T, A, F, act, nout, discrete, segments};
  if (loss_check(p) || !mse_sums || !workspace || !out || !out_f) return -1;
  return loss_fwd_impl(p, mse_sums, workspace, out, out_f, stream);
}

extern "C" int madiff_loss_bwd_seg(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                                   const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B,
                                   int T, int A, int F, int act, int nout, int discrete, int segments, const float* mse_sums,
                                   const double* out, const void* go, int go_f64, float* d_eps, float* d_pred,
                                   hipStream_t stream) {
  LossArgs p{eps, noise, w, masks, masks_f64, pred, pred_rs, pred_as, x, legal, B, T, A, F, act, nout, discrete, segments};
  if (loss_check(p) || !mse_sums || !out || !go || !d_eps || !d_pred) return -1;
  const long n = (long)B * T * A * F;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(nblk(n)), dim3(256), 0, stream, p, mse_sums, out, go, go_f64, d_eps, d_pred);
  MADIFF_CHECK_LAUNCH();
  return 0;
}
