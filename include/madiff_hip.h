/* madiff_hip.h - C ABI of libmadiff_hip.so (gfx950 / MI355X).
 *
 * The drop-in boundary of the MADiff hot path.  The reference (zbzhu99/madiff) is pure Python/PyTorch and has
 * no FFI of its own; what these entry points replace are the stock ATen/cuDNN ops reached from the reference
 * files cited on each declaration (paths relative to the reference repository root).  They are bound with
 * ctypes by madiff_amd/lib.py and called only from the torch.autograd.Functions in madiff_amd/ops.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch allocation), 16-byte aligned;
 *   - dtype: MADIFF_F32 (0) = fp32 storage + exact-fp32 MFMA, MADIFF_BF16 (1) = bf16 storage + bf16 MFMA with
 *     fp32 accumulation; statistics, softmax, losses, the posterior and all parameter gradients are fp32;
 *   - activations are channels-last matrices [rows = (sequence n, time t), cols = channels], leading dimension
 *     `ld*` in elements, column counts padded with zeros to a multiple of 32;
 *   - trajectories keep the reference layout [B, T, A, F] in fp32;
 *   - return value: 0 = ok, -1 = invalid argument (nothing launched), otherwise a hipError_t; never throws;
 *   - `stream` is a hipStream_t; every call is asynchronous on it and none allocates, frees or synchronises,
 *     so a caller may capture any sequence of them into a hipGraph.
 */
#ifndef MADIFF_HIP_H
#define MADIFF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MADIFF_F32 0
#define MADIFF_BF16 1

typedef void* madiff_stream_t; /* hipStream_t */

int madiff_version(void);
const char* madiff_arch(void);

/* Conv1dBlock forward in one launch (bf16 build; helpers.py:50-74 + temporal.py:197-200): Conv1d(k=5, pad=2) + bias ->
 * GroupNorm(G) -> Mish -> (+ rowbias[n, c] (mode 1) | + res[(n,t), c] (mode 2)).  The all-taps convolution kernel's output
 * tile holds whole sequences x whole groups, so the normalisation is its epilogue.  Y = pre-norm output kept for
 * backward (NULL under no_grad, then mean/rstd may be NULL too).  variant 0/1 = 192/96-row tile.  Returns 1 when the
 * layer is not covered (use madiff_gemm_fwd + madiff_gn_mish_fwd). */
int madiff_conv_gn_mish_fwd(const void* X, int ldx, const void* X2, int ldx2, int Csplit, int M, int Cin, int T,
                            const void* W, int ldw, int Nout, const float* bias, void* Y, int ldy, void* out, int ldo,
                            const float* gamma, const float* beta, float* mean, float* rstd, int G, float eps, int mode,
                            const float* rowbias, int ldrb, const void* res, int ldr, int variant, madiff_stream_t stream);
/* The same with a per-channel vector next to the row bias (mode 1: + rowbias[n, c] + rowvec[c]; rowvec fp32 [Nout],
 * 16-byte aligned, or NULL).  Replaces, in the sampling loop only (diffusion.py:214-275 calling temporal.py:231-262,197):
 * every trajectory of the batch is at the same diffusion step there, so the time half of each block's
 * Linear(Mish(t_emb)) is one vector per step and the returns half is constant over the K steps. */
int madiff_conv_gn_mish_fwd_rv(const void* X, int ldx, const void* X2, int ldx2, int Csplit, int M, int Cin, int T,
                               const void* W, int ldw, int Nout, const float* bias, void* Y, int ldy, void* out, int ldo,
                               const float* gamma, const float* beta, float* mean, float* rstd, int G, float eps, int mode,
                               const float* rowbias, int ldrb, const void* res, int ldr, int variant, const float* rowvec,
                               madiff_stream_t stream);

/* Graph-external events (no counterpart in the reference, which has neither graphs nor data parallelism): a record
 * issued on a capturing stream becomes an event-record node of the hipGraph, so a stream outside the graph (the RCCL
 * communication stream of madiff_amd/trainer.py) can wait for "gradient bucket b is complete" inside a replayed
 * backward pass.  Host-side helpers only: they launch nothing. */
int madiff_event_create(void** event_out);
int madiff_event_destroy(void* event);
/* Timing-enabled events for per-launch kernel durations inside a replayed graph (bench.py's roofline object): record one
 * before and one after a launch with madiff_event_record_external while capturing, read the pair after each replay.
 * No counterpart in the reference. */
int madiff_event_create_timed(void** event_out);
int madiff_event_elapsed_ms(void* start, void* stop, float* ms);
int madiff_event_record_external(void* event, madiff_stream_t stream);
int madiff_stream_wait_event(madiff_stream_t stream, void* event);
/* Inside one stream capture: make `waiter` depend on everything captured so far on `waited` by adding waited's dependency
 * frontier to waiter's (hipStreamUpdateCaptureDependencies) instead of an event record + wait.  Needed because two side
 * streams of a capture that have waited on each other's events make hipStreamEndCapture recurse without end on this
 * runtime.  Returns 1 (nothing done) when the two streams are not both capturing the same graph.
 * No counterpart in the reference (no graphs there). */
int madiff_capture_join(madiff_stream_t waiter, madiff_stream_t waited);
/* 0 = `stream` is not capturing, 1 = capturing, 2 = its capture has been invalidated, < 0 = -hipError_t.  No counterpart in
 * the reference (launch plumbing of the captured step): lets the host refuse to wait on an invalidated capture, which
 * this runtime answers by aborting the process. */
int madiff_stream_capture_status(madiff_stream_t stream);

/* Implicit-GEMM convolution / linear:
 *   C[m, o] = act( sum_{tap, c} X[src(m, tap), c] * W[o, tap*Cin + c] + bias[o] + rowbias[m / T_out, o] )
 * src(m = (n, t), tap): tr == 0: t_in = t*stride + tap - pad;  tr == 1: t_in = (t + pad - tap) / stride when divisible;
 * rows outside [0, T_in) contribute zeros.  Channels [0, Csplit) come from X, [Csplit, Cin) from X2 (skip concat).
 * Replaces nn.Conv1d(k=5) of Conv1dBlock (diffuser/models/helpers.py:63-66), Downsample1d/Upsample1d
 * (helpers.py:32-47), the 1x1 residual / Q,K,V convs (models/temporal.py:70-72,183-187), every nn.Linear
 * (temporal.py:74-88,177-181,231-251; models/diffusion.py:111-119) and their data gradients.
 * act: 0 none, 1 relu, 2 mish.  permA > 0 stores rows (b, a, t) as (b, t, a) (the "b a f t -> b t a f" of
 * models/ma_temporal.py:416).  out_f32 != 0 writes fp32 output in the bf16 build.
 * tile: variant (output tile x LDS pipeline depth) of the gfx950 LDS-DMA kernel, 0..madiff_gemm_tile_count()-1, described
 * by madiff_gemm_tile_info; < 0 lets the library choose.  Results do not depend on the variant. */
int madiff_gemm_fwd(int dtype, int out_f32, const void* X, int ldx, const void* X2, int ldx2, int Csplit, int M, int Cin,
                    int taps, int T_out, int T_in, int stride, int pad, int tr, const void* W, int ldw, int Nout, int Npad,
                    const float* bias, const float* rowbias, int ldrb, void* C, int ldc, int act, int permT, int permA,
                    int tile, madiff_stream_t stream);
/* Workspace for the all-taps k5 kernel's reduction split (no counterpart in the reference): with it registered, layers whose
   output tiles would leave most of the chip idle (small batches: batch 32 x accumulate 2 training, the evaluator's 10
   environments) are cut into slices of the input channels; partial tiles meet in `ws` and the last workgroup to arrive
   finishes the tile (bias / GroupNorm epilogues unchanged, summation order fixed).  ws: device memory, 16-byte aligned,
   `bytes` long; counters: n_counters zeroed ints (they reset themselves).  One registration per device (the current one);
   ws = NULL unregisters.  The arena is cut into 16 equal slots (bytes / 16 and n_counters / 16 each) and a slot is owned
   by one issue order: an eager stream, or one stream of one stream capture -- so launches that may overlap on the GPU
   (two streams, two graphs, two branches of a graph) never share partial tiles or tickets.  Slots are not handed on; once
   all are owned, further streams / captures run unsplit.  A launch that faults may leave tickets non-zero: register again.
   madiff_splitk_slots_used(): slots handed out on the current device so far. */
int madiff_set_splitk_workspace(void* ws, long bytes, int* counters, int n_counters);
int madiff_splitk_slots_used(void);
int madiff_gemm_tile_count(void);
/* rows / columns / LDS stages of variant `tile`; returns 1 when madiff_gemm_dgrad_wt offers it too, 0 when not, -1 when
 * the id is invalid. */
int madiff_gemm_tile_info(int tile, int* bm, int* bn, int* stages);

/* Data gradient of the same modules (the dgrad half of autograd for nn.Conv1d / ConvTranspose1d / Linear) computed
 * straight from the FORWARD bf16 weight matrix W [Kred rows = output channels, ldw = taps*wtap]:
 *   dX[m, n] = sum_{tap, o} dY[src(m, tap), o] * W[o, tap*wtap + wcol0 + n],  n < Ncols (pad columns up to Npad are 0).
 * bf16 only; returns 1 (nothing launched) when the shape is not covered (Kred % 64 != 0): the caller then runs
 * madiff_gemm_fwd on a transposed weight pack.  tile: a variant for which madiff_gemm_tile_info returns 1, < 0 =
 * library's choice.  ksplit > 1 cuts the reduction into slices combined with fp32 atomics: dX is then FP32 [M, ldx]
 * (for few-row, long-reduction shapes such as the time-MLP bundle). */
int madiff_gemm_dgrad_wt(const void* dY, int ldy, int M, int Kred, int taps, int T_out, int T_in, int stride, int pad,
                         int tr, const void* W, int ldw, int wtap, int wcol0, int Ncols, int Npad, void* dX, int ldx,
                         int tile, int ksplit, madiff_stream_t stream);

/* madiff_gemm_dgrad_wt with an addend: dX = data gradient + addend (bf16 [M, ld_add], pad columns zero, may be NULL).
   Merges the two gradient contributions of a tensor with two consumers - ResidualTemporalBlock's input feeds blocks[0]
   and residual_conv (reference diffuser/models/temporal.py:197-200; torch autograd adds them in a separate pass).
   Returns 1 when the selected kernel has no vector epilogue for this layer: the caller then adds itself. */
int madiff_gemm_dgrad_wt_add(const void* dY, int ldy, int M, int Kred, int taps, int T_out, int T_in, int stride, int pad,
                             int tr, const void* W, int ldw, int wtap, int wcol0, int Ncols, int Npad, void* dX, int ldx,
                             int tile, int ksplit, const void* addend, int ld_add, madiff_stream_t stream);

/* Backward of "Conv1dBlock A, then a k5 convolution B" in one launch (bf16 build): B's data gradient (dh = dY (*) W_B from
   B's forward weight matrix, as madiff_gemm_dgrad_wt) with block A's GroupNorm + Mish backward as the kernel's epilogue
   (reference diffuser/models/helpers.py:63-71 differentiated; in ResidualTemporalBlock, temporal.py:197-200: A = blocks[0],
   B = blocks[1]'s Conv1d).  Ypre / mean / rstd: what A's forward saved.  dX <- gradient of A's pre-norm conv output; dgamma,
   dbeta, dbias (A's conv bias, may be NULL) accumulated atomically; drowbias [M/T, ldrb] = sum_t dh (gradient of the
   per-sequence time-embedding term) may be NULL.  variant 1 | 2 = 96- | 192-row tile.  Returns 1 when not covered. */
int madiff_conv_dgrad_gn_bwd(const void* dY, int ldy, int M, int Kred, int T, const void* W, int ldw, int wtap, int Ncols,
                             const void* Ypre, int ldyp, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int G, void* dX, int ldx, float* dgamma, float* dbeta, float* dbias,
                             float* drowbias, int ldrb, int variant, madiff_stream_t stream);

/* The same launch for two CONSECUTIVE ResidualTemporalBlocks (reference diffuser/models/temporal.py:197-200, twice: the
   first block's output feeds the second one only): B = the second block's blocks[0] convolution, A = the first block's
   blocks[1] (GroupNorm + Mish, then the residual add).  dh = dY (*) W_B + addend is the gradient of the first block's
   output (addend = the second block's residual-branch gradient, may be NULL); it is stored to dH [M, ldh] because the first
   block's residual branch needs it, and taken through A's GroupNorm + Mish backward into dX in the same epilogue.  Replaces
   madiff_gemm_dgrad_wt_add + madiff_gn_mish_bwd (one launch less per block pair).  Status as madiff_conv_dgrad_gn_bwd. */
int madiff_conv_dgrad_gn_bwd_chain(const void* dY, int ldy, int M, int Kred, int T, const void* W, int ldw, int wtap, int Ncols,
                                   const void* Ypre, int ldyp, const float* gamma, const float* beta, const float* mean,
                                   const float* rstd, int G, void* dX, int ldx, float* dgamma, float* dbeta, float* dbias,
                                   const void* addend, int ld_add, void* dH, int ldh, int variant, madiff_stream_t stream);

/* Weight gradient of the above: dW[o, tap*Cin + c] (+)= sum_m dY[m, o] * X[src(m, tap), c]  (fp32 output).
 * Replaces the wgrad half of autograd for the same modules.  accumulate != 0 adds into dW (a live gradient buffer
 * kept in this GEMM layout by the flat optimizer), 0 overwrites.  tile: variant of the gfx950 LDS-DMA kernel, < 0 =
 * library's choice; 0..3 = one workgroup per (O x C, tap) tile {128x128, 128x64, 64x128, 64x64}; 4..7 = ALL taps of a
 * stride-1 "same" k5 convolution in one pass (operands staged once per K-step, tap shift applied by the transposing LDS
 * reads) with tiles {64x64, 64x64, 128x64, 128x64} and {2, 3, 2, 3} LDS stages -- these return 1 (nothing launched) when
 * the layer is not such a convolution or its sequence length divides neither 32 nor 96; 8, 9 = the same 64x64 all-taps
 * tile on EIGHT waves ({2, 3} LDS stages): the reduction rows of every K-step are split over two wave groups whose partial
 * tiles are combined in LDS, so 256 tiles fill the chip without a reduction split and dW is updated by plain
 * read-modify-write instead of fp32 atomics.  splits: number of slices the
 * reduction over rows is cut into (slices combine with fp32 atomics), <= 0 = library's choice. */
int madiff_gemm_wgrad(int dtype, const void* dY, int ldy, int Ycols, const void* X, int ldx, const void* X2, int ldx2,
                      int Csplit, int M, int Cin, int taps, int T_out, int T_in, int stride, int pad, int tr, float* dW,
                      int ldw, int Nout, int accumulate, int tile, int splits, madiff_stream_t stream);

/* madiff_gemm_wgrad plus the bias gradient of the same layer (group) in the same call: dbias[o] += sum_m dY[m, o], or
   *bias_table[o] += ... (one destination address per output column; exactly one of the two is non-NULL).  Replaces the
   bias half of every Conv1d / Linear backward (ATen's sum over the output gradient).  The per-tap LDS-DMA kernel forms
   the sums itself, without a second pass over dY; otherwise madiff_colsum(_scatter) runs behind the weight gradient. */
int madiff_gemm_wgrad_bias(int dtype, const void* dY, int ldy, int Ycols, const void* X, int ldx, const void* X2, int ldx2,
                           int Csplit, int M, int Cin, int taps, int T_out, int T_in, int stride, int pad, int tr, float* dW,
                           int ldw, int Nout, int accumulate, int tile, int splits, float* dbias, const void* bias_table,
                           madiff_stream_t stream);

/* Parameter layout (fp32, element (o, i, tap) at o*so + i*si + tap*st) <-> GEMM layout [O, taps, Ipad].
 * pack writes dst[(o*taps + tap)*dst_ld + dst_off + i] and leaves pad columns alone (buffer is zero-filled once). */
int madiff_pack_weight(int dtype, const float* src, void* dst, int O, int I, int taps, long so, long si, long st,
                       int dst_ld, int dst_off, madiff_stream_t stream);
/* One launch for many pack jobs.  table (device): n_desc rows of 12 int64 {src, dst, O, I, taps, so, si, st, dst_ld,
 * dst_off, n_elems = O*taps*I, first_block}; row r owns blocks [first_block_r, first_block_{r+1}), 2048 elements each. */
int madiff_pack_weights_multi(int dtype, const long* table, int n_desc, int total_blocks, madiff_stream_t stream);

/* Fused GroupNorm(G) + Mish + {mode 1: + rowbias[n, c] (timestep embedding), mode 2: + res[m, c] (residual)}.
 * Replaces helpers.py:67-71 plus the adds of temporal.py:197-200.  mean/rstd [N, G] are saved for backward. */
int madiff_gn_mish_fwd(int dtype, const void* y, int ldy, void* out, int ldo, const float* gamma, const float* beta,
                       float* mean, float* rstd, int N, int T, int C, int Cpad, int G, float eps, int mode,
                       const float* rowbias, int ldrb, const void* res, int ldr, madiff_stream_t stream);
/* Backward of the above.  dgamma/dbeta/dbias [C] are ACCUMULATED with atomics (caller zeroes them or passes live
 * gradient buffers); dbias (may be null) is the gradient of the bias of the conv feeding the norm (sum_m dy);
 * drowbias [N, ldrb] (may be null) is stored. */
int madiff_gn_mish_bwd(int dtype, const void* y, int ldy, const void* dout, int ldd, void* dy, int ldg, const float* gamma,
                       const float* beta, const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dbias,
                       float* drowbias, int ldrb, int N, int T, int C, int Cpad, int G, madiff_stream_t stream);

/* Cross-agent attention core on the fused projection matrix qkv = [q | k | v] ([M, 2*Cq + C]):
 * softmax_a2(q.k / sqrt(d_qk*T)) . v, then x + gamma*out.  Replaces temporal.py:104-124.
 * bf16 build, A <= 8, head slices a multiple of 32 features (dim 128: d_qk 64 / 32, d_v 256 / 128): the contractions over
 * the (d, t) feature axis -- q.k^T, and dO.v^T in backward -- run on v_mfma_f32_16x16x32_bf16 with 16 / {4, 8} batch
 * elements packed into the 16 rows of a tile, operands loaded straight from global memory in the MFMA fragment layout;
 * madiff_attn_mfma_covers() tells whether a shape takes that form (1) or the VALU kernels (0). */
int madiff_attn_mfma_covers(int dtype, int A, int T, int H, int Cq, int C);
int madiff_attn_fwd(int dtype, const void* qkv, int ldq, const void* x, int ldx, void* out, int ldo, const float* gamma,
                    int B, int A, int T, int H, int Cq, int C, int residual, madiff_stream_t stream);
int madiff_attn_bwd(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg, const float* gamma,
                    float* dgamma, int B, int A, int T, int H, int Cq, int C, int residual, madiff_stream_t stream);

/* madiff_attn_bwd that also yields the gradient of the per-sequence row bias of the fused [q | k | v] projection, i.e. of
   the three timestep MLP outputs TemporalSelfAttention adds to q, k, v (reference diffuser/models/temporal.py:100-111):
   drowbias[(b,a), col] = sum_t dqkv[(b,a,t), col], fp32 [B*A, ldrb], columns [0, 2Cq + C).  drowbias = NULL: madiff_attn_bwd. */
int madiff_attn_bwd_rb(int dtype, const void* qkv, int ldq, const void* dout, int ldd, void* dqkv, int ldg,
                       const float* gamma, float* dgamma, float* drowbias, int ldrb, int B, int A, int T, int H, int Cq,
                       int C, int residual, madiff_stream_t stream);

/* [B,T,A,F] fp32 -> [(b a), t, Fpad] storage type, optionally fused with add_noise (diffusers DDPMScheduler.add_noise,
 * called at models/diffusion.py:292) and inpainting (helpers.py:194-196, diffusion.py:293).  x_noisy (fp32, may be null)
 * receives the noised + inpainted trajectory in the reference layout.  x_ld: feature-row stride of x in elements (x may
 * be the observation columns of the batch's [B,T,A,act+F] tensor, diffusion.py:463: `x[..., self.action_dim:]`, read in
 * place); <= 0 means F. */
int madiff_pack_input(int dtype, const float* x, const float* noise, const long* tsteps, const float* sqrt_ac,
                      const float* sqrt_1mac, const float* cond, const uint8_t* mask, void* out, float* x_noisy, int B,
                      int T, int A, int F, int Fpad, int x_ld, madiff_stream_t stream);
int madiff_unpack_output(int dtype, const void* in, int ld, float* out, int B, int T, int A, int F, madiff_stream_t stream);

/* Fused posterior step: guidance blend eps_u + w (eps_c - eps_u) (diffusion.py:203-205), x0-predict, clip to [-1,1],
 * posterior mean, + sigma*z (diffusers DDPMScheduler.step / DDIMScheduler.step, called at diffusion.py:261) and the
 * inpainting of the next iteration (diffusion.py:251,268).  coef: [n_steps, 6] rows {sqrt(1-abar), 1/sqrt(abar), c_x0,
 * c_xt, c_eps, sigma}; the row index is read from *step_ptr on the device (hipGraph replay) or taken from step_val. */
int madiff_posterior_step(const float* x, const float* eps_c, const float* eps_u, float w, const float* z, const float* cond,
                          const uint8_t* mask, const float* coef, const int* step_ptr, int step_val, float* out, long total,
                          madiff_stream_t stream);
/* madiff_posterior_step with the scheduler's noise taken on the device -- diffusers' `randn_tensor` inside
 * DDPMScheduler.step (diffusion.py:261) -- so that the whole K-step loop (diffusion.py:249-263) is device-driven and can be
 * ONE hipGraph launch.  z_all: explicit draws [n_steps, total] (row = step index; parity runs with host-supplied noise) or
 * NULL; rng: device uint64[2] = {seed, stream id} for z ~ N(0,1) from Philox4x32-10 + Box-Muller, counter = (element
 * index / 4, step index, stream id): every (seed, step, element) has its own draw, independent of the launch geometry. */
int madiff_posterior_step_rng(const float* x, const float* eps_c, const float* eps_u, float w, const float* z_all,
                              const unsigned long long* rng, const float* cond, const uint8_t* mask, const float* coef,
                              const int* step_ptr, int step_val, float* out, long total, madiff_stream_t stream);
int madiff_inpaint(float* x, const float* cond, const uint8_t* mask, long total, madiff_stream_t stream);
/* The SMAC data encoders (utils/data_encoder.py:11-20), applied where the reference calls self.data_encoder
 * (diffusion.py:252,269,294): x[b, t >= 1, a, f < k] = x[b, 0, a, f], in place on a [B,T,A,F] fp32 trajectory. */
int madiff_encode_first_k(float* x, int B, int T, int A, int F, int k, madiff_stream_t stream);
/* Device-side step counter of the captured sampling loop (replaces the host loop variable of diffusion.py:249-255). */
int madiff_sampler_fill_time(const int* state, const long* tsteps, long* ts_out, int B, madiff_stream_t stream);
int madiff_sampler_advance(int* state, madiff_stream_t stream);

/* Masked weighted state loss (helpers.py:243-250, diffusion.py:328-330): per-sample sums
 * {sum (p-q)^2 w m, sum m, sum (p-q)^2 w} and its gradient w.r.t. p. */
int madiff_mse_fwd(const float* pred, const float* targ, const float* w, const float* mask, float* sums, int B, int TA,
                   int F, madiff_stream_t stream);
int madiff_mse_bwd(const float* pred, const float* targ, const float* w, const float* mask, const float* sums,
                   const float* go, float* dpred, int B, int TA, int F, madiff_stream_t stream);

/* The whole training loss of GaussianDiffusion.loss (diffusion.py:438-496) in two launches: the masked weighted state loss
 * (helpers.py:243-250, diffusion.py:328-330), the inverse-dynamics loss of compute_inv_loss (diffusion.py:369-436; shared or
 * joint MLP; continuous MSE, or discrete cross-entropy over the legal actions with inv_acc) and 0.5 * (diffuse + inv)
 * (diffusion.py:492).  masks [B,T,A] fp64 (masks_f64 != 0, as the dataset delivers them) or fp32; x [B,T,A,act+F] (actions
 * first; discrete: column 0 = action id); legal [B,T,A,nout] or NULL; pred = output of the inverse-dynamics MLP, element
 * (r = (b, t < T-1), a, o) at pred[r*pred_rs + a*pred_as + o].  mse_sums [B,3] is kept for backward.  workspace:
 * madiff_loss_ws_doubles() doubles of scratch (partial sums between the two launches).  out / out_f [6] (double / float):
 * {loss, a0_loss, inv_loss, inv_acc, sum of the inverse-dynamics masks, diffuse_loss}. */
int madiff_loss_ws_doubles(void);
int madiff_loss_fwd(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64, const float* pred,
                    long pred_rs, long pred_as, const float* x, const float* legal, int B, int T, int A, int F, int act,
                    int nout, int discrete, float* mse_sums, double* workspace, double* out, float* out_f,
                    madiff_stream_t stream);
/* madiff_loss_fwd for callers that cannot allocate (no counterpart in the reference; for code under hipGraph capture
 * without a memory pool, serving loops and harnesses that forbid hipMalloc): same inputs and outputs, but no workspace and no
 * mse_sums argument -- both live in a fixed device array inside the library (193.5 KB per GPU, held until the process ends),
 * and neither this call nor anything it calls allocates or frees.  Forward only (loss evaluation / logging).  Limit:
 * B <= 16384; beyond it the call launches nothing and returns -2.  Calls that overlap on one GPU share the array and
 * race; calls on one stream are ordered.  madiff_loss_static_init() resolves the array's device address for the current
 * GPU (hipGetSymbolAddress; idempotent, thread-safe); the entry point does the same on its first EAGER call on a device.
 * Resolving a symbol may load code or allocate, which is not legal inside a stream capture: madiff_loss_static_init() MUST
 * have been called once per device before a capture that contains this entry point -- if it was not, the call launches
 * nothing and returns -3. */
int madiff_loss_static_init(void);
int madiff_loss_fwd_static(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                           const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B, int T,
                           int A, int F, int act, int nout, int discrete, double* out, float* out_f, madiff_stream_t stream);
/* Its backward in one launch: d_eps [B,T,A,F] and d_pred (layout of pred, pad columns zeroed) = *go times the gradient of
 * `loss` (go: device scalar, fp64 when go_f64 != 0). */
int madiff_loss_bwd(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64, const float* pred,
                    long pred_rs, long pred_as, const float* x, const float* legal, int B, int T, int A, int F, int act,
                    int nout, int discrete, const float* mse_sums, const double* out, const void* go, int go_f64,
                    float* d_eps, float* d_pred, madiff_stream_t stream);

/* madiff_loss_fwd / _bwd for `segments` micro-batches of one optimizer step folded along the batch axis (B % segments == 0):
   the gradient of (1/S) * sum of the S separate loss() calls of the reference's accumulation loop
   (diffuser/utils/training.py:147-152), in one pass -- the inverse-dynamics part is normalised by each micro-batch's own
   mask mean (diffuser/models/diffusion.py:407,418).  workspace: segments * madiff_loss_ws_doubles() doubles; out / out_f:
   6 * (1 + segments) values = the means over the micro-batches, then the six of every micro-batch. */
int madiff_loss_fwd_seg(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                        const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B, int T, int A,
                        int F, int act, int nout, int discrete, int segments, float* mse_sums, double* workspace, double* out,
                        float* out_f, madiff_stream_t stream);
int madiff_loss_bwd_seg(const float* eps, const float* noise, const float* w, const void* masks, int masks_f64,
                        const float* pred, long pred_rs, long pred_as, const float* x, const float* legal, int B, int T, int A,
                        int F, int act, int nout, int discrete, int segments, const float* mse_sums, const double* out,
                        const void* go, int go_f64, float* d_eps, float* d_pred, madiff_stream_t stream);
/* Inputs of the inverse-dynamics MLP (diffusion.py:381-384): rows [o_t | o_{t+1}] for t < T-1 in the activation layout
 * (storage type, ld columns, pad zeroed).  joint = 0: one row per (b, t, a); joint = 1: one row per (b, t), agents
 * concatenated (joint_inv). */
int madiff_inv_pairs(int dtype, const float* x, void* out, int ld, int B, int T, int A, int act, int F, int joint,
                     madiff_stream_t stream);

/* The embedding head of a denoiser forward in ONE launch (bf16 build, no autograd): SinusoidalPosEmb -> time_mlp
 * (helpers.py:17-29, temporal.py:231-236), returns_mlp and the conditioning-dropout mask / force_dropout zeroing
 * (temporal.py:244-252, ma_temporal.py:341-354), concatenated as [time | returns] per row (b, a).  Replaces ~25 dependent
 * launches (sinusoid, five GEMMs, four Mish, mask, cat, casts) at the head of every one of the K denoising steps.
 * W1 / W3 / M2 / M4: bf16 [O, ld] weight matrices in the GEMM layout; b* / c* / M0 (Linear(1, dim)): fp32 parameters.
 * ret = NULL: no returns branch.  mask: fp32 [B*A] or NULL; force_zero: the unconditional pass.  act = 1 stores
 * mish(value) (the only consumer is Linear(Mish(t_emb)), temporal.py:166-170).  out2 (may be NULL): the classifier-free
 * guidance twin of every row (same time half, returns half zero, diffusion.py:195-202).  dim % 32 == 0, dim <= 128; returns
 * 1 when the shape is not covered (caller: the separate launches). */
int madiff_embed_mlp_fwd(const long* ts, const float* freq, const void* W1, int ld1, const float* b1, const void* W3, int ld3,
                         const float* b3, const float* ret, const float* M0, const float* c0, const void* M2, int ldm2,
                         const float* c2, const void* M4, int ldm4, const float* c4, const float* mask, int force_zero,
                         void* out, int ldo, void* out2, int B, int A, int dim, int act, madiff_stream_t stream);
/* SinusoidalPosEmb (helpers.py:17-29) for timesteps ts[B], repeated over A agents: out [B*A, ld], cols >= 2*half zeroed. */
int madiff_sinusoid(int dtype, const long* ts, const float* freq, void* out, int ld, int B, int A, int half,
                    madiff_stream_t stream);

/* nn.Mish (kind 2) / nn.ReLU (kind 1) on flat padded matrices, forward and backward. */
int madiff_act_fwd(int dtype, int kind, const void* x, void* y, long n, madiff_stream_t stream);
int madiff_act_bwd(int dtype, int kind, const void* ref, const void* dy, void* dx, long n, madiff_stream_t stream);

/* Bias gradients: out[c] += sum_m x[m, c] (caller zeroes) and per-sequence sums out[n, c] = sum_t x[(n,t), c]. */
int madiff_colsum(int dtype, const void* x, int ld, float* out, long M, int C, madiff_stream_t stream);
/* Same sums for a GROUP of layers that share one output-gradient matrix (q/k/v projections, the 21 Linear(Mish(t_emb))
 * of temporal.py:173-176 / :69-80 evaluated as one GEMM): column c is added to *table[c] (device array of C float
 * addresses, one per gradient element of the members' biases).  Returns 1 if the layout is not 16-byte vectorisable. */
int madiff_colsum_scatter(int dtype, const void* x, int ld, float* const* table, long M, int C, madiff_stream_t stream);
int madiff_seqsum(int dtype, const void* x, int ld, float* out, int ldo, long N, int T, int C, madiff_stream_t stream);

/* Fused Adam (torch.optim.Adam defaults, diffuser/utils/training.py:109,154) + EMA (training.py:20-39) over one flat
 * fp32 tensor.  do_ema: 0 none, 1 lerp with ema_beta, 2 copy.  state == NULL: bias corrections and do_ema come from the
 * arguments; state != NULL (device int[2] = {completed steps, do_ema}): they are derived on the device and state[0] is
 * incremented afterwards (hipGraph replay of a whole train step).  shadow_bf16 != NULL: the bf16 rounding of every
 * updated parameter is written to shadow_bf16[e] (the flat buffer is kept in GEMM layout, so that copy is the packed
 * weight operand of the next step).  zero_grad bit 0: clear g after it was consumed (optimizer.zero_grad, :155); bit 1:
 * do not advance state[0] after this launch (the step is issued as several slices, one per gradient bucket, overlapped
 * with backward; only the last call advances).  n == 0 with state != NULL: advance only.  16-byte accesses are used
 * when every pointer is 16-byte aligned and n % 4 == 0. */
int madiff_adam_ema(float* p, float* g, float* m, float* v, float* ema, long n, float lr, float b1, float b2,
                    float eps, float bc1, float bc2_sqrt, float grad_scale, int do_ema, float ema_beta, int* state,
                    void* shadow_bf16, int zero_grad, madiff_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
