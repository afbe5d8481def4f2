// Launch parameter blocks shared by the generic (gemm.hip) and the LDS-DMA bf16 (gemm_bf16.hip) implicit-GEMM kernels.
#pragma once
#include "common.h"

// Output row (n, t) of a layer reads input row (n, t*stride + tap - pad) (tr = 0: Conv1d) or
// (n, (t + pad - tap)/stride) when divisible (tr = 1: ConvTranspose1d and every data gradient).
struct RowMap {
  int T_out, T_in, stride, pad, tr;
};

struct GemmFwdParams {
  const void* X;   int ldx;          // first source  (channels [0, Csplit))
  const void* X2;  int ldx2;         // second source (channels [Csplit, Cin)), may be null
  int Csplit;                        // == Cin when X2 is null
  int M, Cin, taps;                  // M output rows; Cin padded channel count (multiple of KC)
  RowMap rm;
  const void* W;   int ldw;          // [Nout, taps*Cin]   (wt = 1: [Cin rows, ldw], see below)
  int Nout, Npad;                    // real / padded output columns (pad columns are written as 0)
  const float* bias;                 // [Nout] or null
  const float* rowbias; int ldrb;    // [n_seq, ldrb] or null, row = m / T_out
  void* C;         int ldc;
  int act;                           // 0 none, 1 relu, 2 mish
  int permT, permA;                  // if permA > 0: rows (b, a, t) are stored as (b, t, a)
  // wt = 1 (data gradient straight from the FORWARD weight layout, bf16 LDS-DMA kernel only):
  //   B[k = (tap, o), n] = W[o * ldw + tap * wtap + n], n < wcols valid columns
  int wt, wtap, wcols;
  int vec_epi;                       // 1: pointers/strides allow the 16-byte vector epilogue (set by the launcher)
  int ksplit;                        // > 1: gridDim.z slices of the K-steps combine with fp32 atomics (fp32 C, pre-zeroed)
  const void* addend; int lda;       // bf16 [M, lda] added to the result after the activation (vector epilogues only), or null
  int xcd;                           // 1: XCD-aware workgroup -> tile map (common.h xcd_remap), all-taps kernel
  int par;                           // per-tap LDS-DMA kernel, tr = 1 / stride 2: output rows enumerated parity-major (see gemm_bf16.hip)
  // all-taps kernel, reduction split for layers with few output tiles (small batches): gridDim.z = cks slices of the channel
  // chunks; every slice stores its partial tile to ws, the LAST workgroup to arrive (ticket in ws_cnt[tile], reset by it)
  // sums the partials in slice order and runs the epilogue.  ws / ws_cnt: the workspace registered with
  // madiff_set_splitk_workspace (null: no split); cks is chosen by the launcher.
  float* ws; long ws_floats; int* ws_cnt; int ws_ncnt; int cks;
};

struct GemmWgradParams {
  const void* dY;  int ldy;  int Ycols;   // Ycols: padded column count of dY (multiple of 32)
  const void* X;   int ldx;
  const void* X2;  int ldx2;
  int Csplit;
  int M, Cin, taps;
  RowMap rm;
  float* dW;       int ldw;               // [Nout, taps*Cin] fp32
  int Nout;
  int ctiles;                             // ceil(Cin / 128)
  int steps_per_split, splits;            // reduction steps (KR rows each) per z-slice
  int accumulate;                         // 1: dW += result (dW is a live gradient buffer), 0: dW = result
  int xcd;                                // 1: XCD-aware workgroup -> tile map (all-taps kernels)
  // bias gradient fused into the per-tap LDS-DMA kernel: column sums of dY, accumulated atomically into dbias[o] or, for a
  // fused layer group, into *btab[o] (one destination address per output column); both null = not requested
  float* dbias; float* const* btab;
};

__device__ __forceinline__ int map_row(int n, int t, int tap, const RowMap& rm) {
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
