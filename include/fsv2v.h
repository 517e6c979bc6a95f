/* fsv2v.h - C ABI of libfsv2v_hip.so: the MI355X (gfx950) kernels behind the few-shot-vid2vid G/D training step.
 *
 * The reference (NVlabs/few-shot-vid2vid) has no FFI for this path - its operator seam is Python name binding
 * (SURVEY.md section 8b): models.networks.base_network.{batch_conv,resample}, normalization.SPADE,
 * architecture.SPADEResnetBlock, generator.{LabelEmbedder,FlowGenerator}, discriminator.NLayerDiscriminator, which in
 * turn call torch ATen (F.conv2d, F.grid_sample, BatchNorm, ...).  Its only native boundary is the pybind pattern of
 * the FlowNet2 extensions (models/networks/flownet2_pytorch/networks/resample2d_package/resample2d_cuda.cc:6-31:
 * plain tensors in, int status out, kernels enqueued on the caller's stream).  This header follows that convention:
 *
 *   - every function returns 0 (FSV_OK) or a negative fsv_status; nothing throws or aborts across the boundary;
 *   - all pointers are device pointers owned by the caller (PyTorch's caching allocator); the library allocates
 *     nothing and keeps no state between calls - no setter-style entry points, nothing "armed" for a later call; scratch
 *     buffers, optional workspaces and optional side outputs are explicit (nullable) arguments of the call that uses them;
 *   - work is only enqueued on `stream` (a hipStream_t); calls are re-entrant and graph-capturable
 *     (only kernels and hipMemsetAsync are issued);
 *   - activations are fp32 NHWC ("channels last"); convolution weights are fp32 OIHW at the boundary and are
 *     re-arranged K-major by fsv_prep_weight.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 */
#ifndef FSV2V_H
#define FSV2V_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fsv_stream_t; /* hipStream_t */

enum fsv_status { FSV_OK = 0, FSV_ERR_BAD_ARG = -1, FSV_ERR_UNSUPPORTED = -2, FSV_ERR_LAUNCH = -3 };
enum fsv_act { FSV_ACT_NONE = 0, FSV_ACT_LRELU = 1 /* leaky_relu(0.2), architecture.py:15-17 */, FSV_ACT_TANH = 2,
               FSV_ACT_SIGMOID = 3, FSV_ACT_RELU = 4 /* VGG19 stack */, FSV_ACT_LRELU01 = 5 /* leaky_relu(0.1), FlowNet2 */,
               /* gather-GEMM epilogue only: out = v * leaky_relu'(res) (res = the activated output of the layer below): the
                * activation-backward pass of a Linear + LeakyReLU chain (generator.py:103-110) folded into the data gradient */
               FSV_ACT_DLRELU = 6 };

/* ---- convolution family (csrc/conv_igemm.hip) -----------------------------------------------------------------
 * Replaces F.conv2d at architecture.py:22-27,60,81-84; generator.py:112-131,479-504,541-572;
 * discriminator.py:67-88; nn.Linear at generator.py:103-110; batch_conv at base_network.py:56-71 and, through
 * autograd, their cudnn backward kernels.
 *
 * out[n, oy*osy+ooy, ox*osx+oox, co] = act(( sum_t sum_ci in[n, oy*sy+ty[t], ox*sx+tx[t], ci] * wt[t*Cin+ci][co]
 *                                            + bias[co] ) * scale) + res[...]
 * taps outside the input read as zero.  wt: K-major [Kpad][ldw] from fsv_prep_weight; per_sample != 0 selects one
 * weight matrix (stride w_bstride) and bias (stride b_bstride) per sample n.  accumulate != 0: `out` was zeroed by
 * the caller, results are added (used for the four parity classes of a stride-2 data gradient).
 * force_tile / force_split: -1 / 0 = automatic (tile ids, pixels x channels: 0 128x128, 1 128x64, 2 128x32, 4 64x64, 9 64x128;
 * 10 / 11 / 12 = 64x128 / 128x128 / 128x64 with the global loads two chunks ahead: what the plan's 9 / 0 / 1 run as unless
 * FSV_CONV_PF2=0).
 * One activation tensor / weight matrix may hold at most 2 GiB (32-bit byte offsets): FSV_ERR_UNSUPPORTED beyond.  wscale: optional device scalar multiplying the accumulator before
 * the bias (the spectral-norm 1/sigma when wt holds un-normalised weights). */
/* split_ws / split_ws_floats (nullable; ordered split-K): when the call's plan splits K, split k stores its partial output into
 * the k-th copy inside split_ws (nsplit x N*outH*outW*Cout floats) and the copies are summed in ascending order: the same bits on
 * every run, no zero fill, no atomics (a finishing pass sums the copies).  A call that does not split, or whose copies do not fit into split_ws_floats,
 * ignores the workspace (and adds atomically into a zeroed output).
 * in_up != 0: nn.Upsample(scale_factor=2) in front of the convolution (generator.py:124,497-504,541-572) folded into the gather -
 * `in` is stored at H / 2 x W / 2 (H, W stay the size the convolution sees: even) and read at (y >> 1, x >> 1); the up-sampled
 * tensor is never written.  Float4-gather MFMA launches only (Cin % 4 == 0, Cout > 4, shared weights, no accumulate):
 * FSV_ERR_UNSUPPORTED otherwise - materialise with fsv_upsample2x_fwd then.  fsv_conv_wgrad takes the same flag. */
int fsv_conv_gather_fwd(const float* in, const float* wt, const float* bias, const float* res, float* out,
                        int N, int H, int W, int Cin, int OH, int OW, int Cout,
                        int ntaps, const int* ty, const int* tx, int sy, int sx,
                        int outH, int outW, int osy, int osx, int ooy, int oox,
                        int ldw, long long w_bstride, long long b_bstride, int per_sample,
                        int act, float scale, int force_tile, int force_split, int accumulate, const float* wscale,
                        float* split_ws, long long split_ws_floats, int in_up, fsv_stream_t stream);

/* ---- grouped launches: up to 64 INDEPENDENT problems in one grid --------------------------------------------------------
 * The reference issues the 16 weight-generator MLPs (generator.py:103-110,245-273: three nn.Linear each, per adaptive level
 * and per SPADE site) and the transposed-convolution data gradients of its stride-2 layers one cudnn call at a time; on this
 * device every such launch lasts as long as its longest workgroup while most of the 256 CUs idle.  A group puts the tiles of
 * all its problems into one grid (longest problems first).  fsv_conv_desc = the arguments of fsv_conv_gather_fwd;
 * accumulate != 0: `out` was zeroed by the caller and the problem may be K-split (atomic adds, no epilogue).  No problem of a
 * group may read what another one writes.  A problem with Cin % 4 != 0 sends the whole group to the scalar-gather kernel. */
typedef struct fsv_conv_desc {
  const float* in; const float* wt; const float* bias; const float* res; float* out; const float* wscale;
  int N, H, W, Cin, OH, OW, Cout, ntaps;
  int ty[16], tx[16];
  int sy, sx, outH, outW, osy, osx, ooy, oox, ldw;
  int per_sample, act, accumulate;
  float scale;
  long long w_bstride, b_bstride;
} fsv_conv_desc;
int fsv_conv_gather_group(const fsv_conv_desc* problems, int n, int force_tile, fsv_stream_t stream);
/* tile id the grouped launch uses for problems of these sizes (Mz = pixels per sample group, nchunks = ceil(taps * Cin / 32)) */
int fsv_conv_group_plan(const int* Mz, const int* Cout, const int* nchunks, const int* nsamp, int n, int* tile_out);
/* grouped weight gradients (arguments of fsv_conv_wgrad): every dwt is a ZEROED [Kpad][ldw] matrix; float4-gather layers only
 * (Cin % 4 == 0), FSV_ERR_UNSUPPORTED with nothing launched otherwise - issue the problems one by one then */
typedef struct fsv_wgrad_desc {
  const float* in; const float* dout; float* dwt;
  int N, H, W, Cin, OH, OW, Cout, ntaps;
  int ty[16], tx[16];
  int sy, sx, ldw, Kpad;
  int per_sample, reserved;
  long long w_bstride;
} fsv_wgrad_desc;
int fsv_conv_wgrad_group(const fsv_wgrad_desc* problems, int n, fsv_stream_t stream);

/* fsv_conv_gather_fwd (dense output, shared weights) that also leaves the per-channel sums of its output for the
 * normalisation that follows - conv -> BatchNorm (architecture.py:57-69, generator.py:479-496) and conv -> InstanceNorm
 * (discriminator.py:67-88) read the convolution's output a second time only to reduce it; here the epilogue adds
 * (sum y, sum y^2) per channel into stats[group][slot][Cout][2] (doubles; zeroed by this call unless stats_prezeroed - a slice of
 * a per-pass arena the caller zeroes once; slot = pixel tile % stats_slots,
 * group = sample block: stats_groups = 1 for BatchNorm, N for InstanceNorm) and fsv_norm_stats_finish turns them into
 * mean / rstd.  *produced = 0: this launch could not (K-split plan, Cin % 4 != 0) - run fsv_norm_stats instead. */
int fsv_conv_gather_fwd_stats(const float* in, const float* wt, const float* bias, const float* res, float* out,
                              int N, int H, int W, int Cin, int OH, int OW, int Cout,
                              int ntaps, const int* ty, const int* tx, int sy, int sx,
                              int ldw, int act, float scale, const float* wscale,
                              double* stats, int stats_groups, int stats_slots, int stats_prezeroed, int* produced,
                              float* split_ws, long long split_ws_floats, int in_up, fsv_stream_t stream);
/* in place: x = act(x + bias[c]) over an NHWC tensor (finishing pass of operators that add several GEMM launches into one
 * output: convolutions with more than 16 taps, transposed convolutions); act codes as in the conv epilogue, 5 = leaky 0.1 */
int fsv_bias_act(float* x, const float* bias, long long total, int C, int act, fsv_stream_t stream);
/* dwt[t*Cin+ci][co] = sum_{n,oy,ox} in[n, oy*sy+ty[t], ox*sx+tx[t], ci] * dout[n, oy, ox, co]  (weight gradient)
 * prezeroed: dwt already holds zeros (a slice of the optimiser's per-pass arena), skip the split-K zero-fill;
 * force_tile: 0 = automatic (1 / 2 / 3 = 64x64 / 128x64 / 64x128 rows x columns, 5 / 6 = 64x64 / 64x128 as one- / two-wave
 * workgroups, 7 / 8 = the same two tiles with double-buffered LDS; for A/B runs) */
int fsv_conv_wgrad(const float* in, const float* dout, float* dwt,
                   int N, int H, int W, int Cin, int OH, int OW, int Cout,
                   int ntaps, const int* ty, const int* tx, int sy, int sx,
                   int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                   int force_tile, int in_up, fsv_stream_t stream);

/* ---- half-precision path (csrc/conv_h.hip): the reference's `--amp O1` arithmetic (options/base_options.py:127,
 * models/models.py:22-26, loss_collector.py:221-224; BASELINE.json configs[4] "fp16 MFMA path") with activations AND weights
 * 16-bit in HBM.  Contract of one convolution: operands are IEEE half in memory (whoever produced them rounded once), products are
 * exact, accumulation is fp32 (v_mfma_f32_32x32x16_f16), (acc * wscale + bias) * scale -> act -> + res in fp32, one rounding at
 * the store when out_h.  Activations NHWC half with Cin % 8 == 0; weights N-MAJOR half wt[z][nrows][Kpad] (K = taps * Cin
 * contiguous, Kpad % 64 == 0, zero padded) as written by fsv_hconv_prep_weight from the K-major fp32 layout of fsv_prep_weight.
 * fsv_hconv_desc = one problem; n == 1: a single launch that may split K (partial sums in fp32: `ws` when the output is half) and
 * may leave normalisation statistics (stats / *produced as fsv_conv_gather_fwd_stats); n > 1: ONE grid over independent
 * problems (fsv_conv_gather_group).  res / out are half when res_h / out_h, else fp32; act FSV_ACT_DLRELU reads res as the aux
 * tensor.  FSV_ERR_UNSUPPORTED (nothing launched) for geometries outside this contract: callers keep those on the fp32 path. */
typedef struct fsv_hconv_desc {
  const void* in; const void* wt; const float* bias; const void* res; void* out; const float* wscale;
  float* ws; double* stats;
  int N, H, W, Cin, OH, OW, Cout, ntaps;
  int ty[16], tx[16];
  int sy, sx, outH, outW, osy, osx, ooy, oox;
  int Kpad, nrows;
  int per_sample, act, accumulate;
  int out_h, res_h;
  int force_tile, force_split;
  int stats_groups, stats_slots, stats_prezeroed;
  float scale;
  long long w_bstride, b_bstride;
} fsv_hconv_desc;
int fsv_hconv_gather(const fsv_hconv_desc* problems, int n, int* produced, fsv_stream_t stream);
/* tile id (0 = 128x128, 1 = 128x64, 2 = 128x32, 4 = 64x64, 9 = 64x128; + 16: two LDS buffers instead of three) and K split the
 * single launch will use; nchunks = ceil(taps * Cin / 64) */
int fsv_hconv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split, int can_split, int* tile_out,
                   int* nsplit_out);
/* weight gradient from half activations `in` (NHWC, Cin % 8 == 0) and half output gradients `dout` ([pixels][Cout], Cout % 8 == 0)
 * into the fp32 K-major matrix dwt[Kpad][ldw] of fsv_conv_wgrad (same arguments); force_tile 1 / 2 / 3 / 4 / 5 = 64x64 / 128x64 /
 * 64x128 / 128x128 / 128x32 */
int fsv_hconv_wgrad(const void* in, const void* dout, float* dwt,
                    int N, int H, int W, int Cin, int OH, int OW, int Cout,
                    int ntaps, const int* ty, const int* tx, int sy, int sx,
                    int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                    int force_tile, fsv_stream_t stream);
/* K-major fp32 wt[z][Kpad32][ldw] -> N-major half wh[z][nrows][Kpad64], table-driven (one launch for a whole layout cache):
 * jobs[j] = {src, dst, Kpad32, ldw, nrows, Kpad64, nbatch, 0} as 64-bit words (device), tmap[b] = (job, 64-k tile, 64-n tile, z) */
int fsv_hconv_prep_weight(const long long* jobs, const int* tmap, int nblocks, fsv_stream_t stream);
/* the same for ONE layout with the geometry in the arguments (no device table: legal inside a graph capture) */
int fsv_hconv_prep_weight_one(const float* src, void* dst, int Kpad32, int ldw, int nrows, int Kpad64, int nbatch,
                              fsv_stream_t stream);
/* dense element conversion, dir 0: fp32 -> half (round to nearest even), 1: half -> fp32 */
int fsv_cast_half(const void* x, void* y, long long n, int dir, fsv_stream_t stream);

/* ---- narrow-operand GEMMs (csrc/conv_np.hip): the reference's `--amp` arithmetic (options/base_options.py:127,
 * models/models.py:22-26 `amp.initialize(..., opt_level=opt.amp, num_losses=2)`; BASELINE.json configs[4]).  Same
 * contracts and arguments as fsv_conv_gather_fwd / fsv_conv_wgrad plus `mode`: 1 = operands rounded to IEEE half while a
 * tile is staged through LDS, 2 = operands split into two bf16 terms (three MFMAs per tile pair); fp32 accumulation,
 * fp32 tensors in HBM.  Only float4-gather layers (Cin % 4 == 0) are implemented: FSV_ERR_UNSUPPORTED otherwise, callers
 * keep those on the fp32 entry points.  force_tile of the weight gradient: 1 / 2 / 3 / 4 = 64x64 / 128x64 / 64x128 /
 * 128x128. */
int fsv_conv_gather_fwd_np(const float* in, const float* wt, const float* bias, const float* res, float* out,
                           int N, int H, int W, int Cin, int OH, int OW, int Cout,
                           int ntaps, const int* ty, const int* tx, int sy, int sx,
                           int outH, int outW, int osy, int osx, int ooy, int oox,
                           int ldw, long long w_bstride, long long b_bstride, int per_sample,
                           int act, float scale, int force_tile, int force_split, int accumulate, const float* wscale,
                           int mode, fsv_stream_t stream);
int fsv_conv_wgrad_np(const float* in, const float* dout, float* dwt,
                      int N, int H, int W, int Cin, int OH, int OW, int Cout,
                      int ntaps, const int* ty, const int* tx, int sy, int sx,
                      int ldw, int Kpad, long long w_bstride, int per_sample, int force_split, int prezeroed,
                      int force_tile, int mode, fsv_stream_t stream);

/* ---- dynamic loss scale of the fp16 mode (csrc/amp.hip) - models/loss_collector.py:221-224 `amp.scale_loss(loss,
 * optimizer, loss_id)`; apex rule: skip the step and halve on inf / nan, double after `window` good steps.
 * scaler = {scale, good_steps, found_inf, window, max_scale, min_scale} on the device (no host read: graph-capturable).
 * fsv_amp_adam is fsv_adam_step with grad * gscale / scale, and does nothing when found_inf is set. */
int fsv_amp_check(const float* grad, long long n, float* scaler, fsv_stream_t stream);
int fsv_amp_adam(float* param, const float* grad, float* m, float* v, float* state, float* scaler, long long n,
                 float beta1, float beta2, float eps, float gscale, fsv_stream_t stream);
int fsv_amp_update(float* scaler, fsv_stream_t stream);

/* OIHW <-> K-major re-arrangement with an optional device scalar multiplier (the spectral-norm 1/sigma).
 * mode 0: wt[j*Cin+ci][co] = s*w[co][ci][kh_j][kw_j]; mode 1 (data gradient): wt[j*Cout+co][ci] = ...;
 * mode 2: inverse of mode 0 (gradients back to OIHW); mode 3: mode 2 accumulating into w. */
int fsv_prep_weight(const float* w, float* wt, const float* scale_ptr, int mode, int nbatch,
                    int Cout, int Cin, int KH, int KW, int ntaps, const int* kh, const int* kw,
                    int Kpad, int ldw, long long w_bstride, long long wt_bstride, fsv_stream_t stream);

/* every parameter weight of an optimiser re-arranged in one launch (un-scaled; used once per optimiser step).  dims[l] =
 * {Cout, Cin_pad, Cin_real, KH, KW, ntaps, Kpad, ldw, mode}; the layouts of one weight are consecutive in the tables and share
 * its source tile: tmap = (first layout of the weight, number of its layouts, 32-co tile, ci tile) quadruples (ci tile = 32
 * channels for KH*KW <= 8, else 16); only the valid region of each (pre-zeroed) layout is written.  mode | 4: the two nibbles of
 * a tap code are MASKS over kh / kw and the written weight is the sum of the selected source taps (the summed-tap layouts of
 * conv3x3(nearest_x2(x)): sub-pixel forward classes, one-launch data gradient) */
int fsv_prep_weight_grouped(const long long* src, const long long* dst, const int* dims, const unsigned long long* taps,
                            const int* tmap, int nblocks, fsv_stream_t stream);
/* grouped fixed-order sums: dst[j] = src[4 j] + src[4 j + 1] + ... (nsrc[j] in 1..4 terms, left to right) over count[j] floats,
 * njobs <= 8, one launch; pointers 16-byte aligned.  Adds the data gradients of the weight-generator MLPs that share their input
 * rows (generator.py:245-273: four FC stacks per level on the same encoded reference) without atomics. */
int fsv_sum_terms(float* const* dst, const float* const* src, const int* nsrc, const long long* count, int njobs,
                  fsv_stream_t stream);
/* Deferred weight-gradient finalisation (csrc/wgrad_finalize.hip): njobs K-major weight gradients -> OIHW, added into
 * their parameter-gradient slices, with torch.nn.utils.spectral_norm's backward correction where sig != 0.
 * ptrs[job][6] = {dwt, K-major W, sink, u, v, sig}; dims[job][8] = {Cout, CinP, CinR, KH, KW, ntaps, ldw, flags};
 * taps[job][2] packed (kh | kw << 4); dots: double[njobs] scratch; tmap_dot (job, 4096-element chunk) pairs over the
 * spectral jobs; tmap_apply (job, 32-co tile, ci tile) triples (ci tile = 32 channels for <= 8 taps, else 16). */
int fsv_wgrad_finalize(const long long* ptrs, const int* dims, const unsigned long long* taps, double* dots, int njobs,
                       const int* tmap_dot, int nblk_dot, const int* tmap_apply, int nblk_apply, fsv_stream_t stream);
/* dst += src over njobs small contiguous fp32 tensors in one launch: table[job][3] = {src, dst, n}, tmap (job, 4096-element
 * chunk) pairs.  Folds the separately accumulated gradients of small parameters into the flat gradient buffer. */
int fsv_gather_add(const long long* table, int njobs, const int* tmap, int nblk, fsv_stream_t stream);
/* n 64-bit words from host memory into a device array, carried in kernel arguments (legal inside a graph capture,
 * no staging buffer); used for the per-pass pointer table of fsv_wgrad_finalize */
int fsv_upload_i64(long long* dst, const long long* host_src, int n, fsv_stream_t stream);
/* tile / split-K plan the launcher will use (exported so host-side profilers label launches consistently) */
int fsv_conv_plan(int Mz, int Cout, int nchunks, int nsamp, int force_tile, int force_split, int* tile_out,
                  int* nsplit_out);
/* 1 when a Cout <= 4 layer of Mz pixels and K = taps * Cin takes the vector-ALU kernels (fsv_conv_thin_*), else 0 - NOT a status;
 * for profiler labels, like fsv_conv_plan */
int fsv_conv_thin_rule(int Mz, int K);

/* One-launch operand preparation of a SPADE (gamma, beta) 1x1 weight pair (normalization.py:37-52): wg / wb [B][C][Ch]
 * (sample strides swg / swb, 0 = shared), bg / bb [B][C] -> wcat_t [B][ceil32(Ch)][2C] (forward operand of map -> [gamma|beta]),
 * wcat_d [B][ceil32(2C)][ceil32(Ch)] (data-gradient operand, may be NULL), bcat [B][2C].  C % 16 == 0. */
int fsv_spade_prep(const float* wg, const float* wb, const float* bg, const float* bb, long long swg, long long swb,
                   long long sbg, long long sbb, float* wcat_t, float* wcat_d, float* bcat, int B, int C, int Ch,
                   fsv_stream_t stream);
/* ---- SPADE (csrc/spade.hip) - replaces SPADE.forward normalization.py:37-52 + actvn architecture.py:95-97 --------
 * h = act( (...((x - mean) * rstd) * (1 + g_0) + b_0 ...) * (1 + g_{n-1}) + b_{n-1} ),  g_k = map_k @ Wg_k + bg_k.
 * One launch; gamma/beta are MFMA accumulators and never reach HBM.
 * up != 0: x is the half-resolution tensor [N][H/2][W/2][C] (H = HW / W) and is read through the nearest x2 up-sampling
 * index - F.interpolate(scale_factor=2) of generator.py:124 folded into its consumer; W is only used then. */
int fsv_spade_mod_fwd(const float* x, const float* mean, const float* rstd, float* h,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act,
                      int W, int up, fsv_stream_t stream);
/* `--amp` forms of fsv_spade_mod_fwd.  flags bit 0: h written as IEEE half (h is only read by half-precision convolutions);
 * bit 2: gamma / beta GEMMs on the f16 matrix instructions - maps IEEE half ([N][HW][Ch], Ch % 8 == 0), wg / wb = gamma rows / beta
 * rows of the N-major half operand of fsv_spade_prep_h (row length ceil32(Ch) halves; ldw unused), w_bstride in halves */
int fsv_spade_mod_fwd_h(const float* x, const float* mean, const float* rstd, void* h,
                        int nmaps, const void* const* maps, const void* const* wg, const void* const* wb,
                        const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                        const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act,
                        int W, int up, int flags, fsv_stream_t stream);
/* wcat_h [B][2C][Kh = ceil32(Ch)] halves, K contiguous, zero padded: rows [0, C) gamma weights, [C, 2C) beta weights */
int fsv_spade_prep_h(const float* wg, const float* wb, long long swg, long long swb, void* wcat_h, int B, int C, int Ch,
                     fsv_stream_t stream);
/* Two norm sites of one SPADEResnetBlock in one launch - bn_0 and bn_s (architecture.py:95-96,103) normalise the same x
 * with the same statistics and read the same maps; only the gamma / beta weights and the activation differ:
 * h0 = act0(SPADE_0(x)), h1 = act1(SPADE_s(x)).  x, the statistics and the label-map tiles are read once, the map tile in LDS
 * feeds four GEMMs.  wg / wb / bg / bb / w_bstride / b_bstride hold 2 * nmaps entries, site 0's first. */
int fsv_spade_mod_fwd2(const float* x, const float* mean, const float* rstd, float* h0, float* h1,
                       int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                       const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                       const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int act0, int act1,
                       int W, int up, fsv_stream_t stream);
/* ---- bn_s modulation fused with conv_s (csrc/spade_conv.hip) - replaces `x_s = self.conv_s(self.bn_s(x, ...))`,
 * architecture.py:103-108 (SPADE.forward normalization.py:37-52 followed by the bias-free spectral-norm 1x1 convolution) ------
 * ONE launch: the gamma / beta GEMMs run with swapped operands (accumulators = [channel][pixel]), the modulated values are the A
 * fragments of a second chain of matrix instructions against the conv_s weight rows kept in registers; the modulated tensor
 * is written only when hs != NULL (a training forward: conv_s' weight gradient reads it).  Operands of the modulation as
 * fsv_spade_mod_fwd (no activation: bn_s has none); ws = K-major forward operand of the 1x1 weight ([>= C rows][ldws],
 * fsv_prep_weight mode 0), wscale = optional device scalar (1 / sigma) on the result; xs [N][HW][Cout].
 * fsv_spade_conv_s_supported: 1 when a kernel exists for (C, Cout, nmaps) - C in {64, 128}, Cout in {32, 64}, 1..3 maps;
 * fsv_spade_conv_s_fwd returns FSV_ERR_UNSUPPORTED otherwise (the caller runs the two launches). */
int fsv_spade_conv_s_supported(int C, int Cout, int nmaps);
int fsv_spade_conv_s_fwd(const float* x, const float* mean, const float* rstd, float* hs, float* xs,
                         int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                         const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                         const long long* b_bstride, int N, int HW, int C, int ldw, long long stat_bstride, int W, int up,
                         const float* ws, int ldws, int Cout, const float* wscale, fsv_stream_t stream);
/* `--amp` form of fsv_spade_conv_s_fwd: f16 GEMMs - maps / wg / wb as fsv_spade_mod_fwd_h with flags bit 2 (IEEE half maps, N-major
 * half gamma | beta operand of fsv_spade_prep_h, Ch % 8 == 0), ws_h = N-major half operand of conv_s ([>= Cout rows][ldws halves],
 * K contiguous: fsv_hconv_prep_weight); the modulated value is rounded to half once (as the two-launch form does at its store),
 * hs_h (optional) receives it as half, xs is fp32 */
int fsv_spade_conv_s_fwd_h(const float* x, const float* mean, const float* rstd, void* hs_h, float* xs,
                           int nmaps, const void* const* maps, const void* const* wg, const void* const* wb,
                           const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                           const long long* b_bstride, int N, int HW, int C, long long stat_bstride, int W, int up,
                           const void* ws_h, int ldws, int Cout, const float* wscale, fsv_stream_t stream);
/* ---- SPADE modulation + activation fused with the 3x3 convolution that consumes it (csrc/spade_conv3.hip) - replaces
 * `dx = self.conv_0(actvn(self.bn_0(x, ...)))` / `self.conv_1(actvn(self.bn_1(dx, ...)))`, architecture.py:96-99 (SPADE.forward
 * normalization.py:37-52, leaky_relu(0.2) architecture.py:111-112, spectral-norm 3x3 convolution with padding 1) ------------
 * ONE launch: a workgroup computes the modulated values of its 8 x 16 output tile's one-pixel halo once into LDS (gamma / beta
 * GEMMs with swapped operands, operands straight from global memory) and runs the convolution from that patch; the modulated
 * tensor is written only when hs != NULL.  Operands of the modulation as fsv_spade_mod_fwd with the image as (H, W); wc = K-major
 * forward operand of the 3x3 weight ([9 C rows (tap, ci)][ldwc], fsv_prep_weight mode 0); out = acc * wscale + bias (+ res),
 * [N][H W][Cout]; stats (optional): fp64 partials [stats_slots][Cout][2] of the output's per-channel sums for the BatchNorm that
 * follows (as fsv_conv_gather_fwd_stats with one group).  fsv_spade_conv3_supported: 1 when a kernel exists - C == 64, Cout in {32, 64}, 1..3 maps;
 * FSV_ERR_UNSUPPORTED otherwise.  Round 6: built and measured against the two-launch form (profiles/r06_notes.md section 8). */
int fsv_spade_conv3_supported(int C, int Cout, int nmaps);
int fsv_spade_conv3_fwd(const float* x, const float* mean, const float* rstd, float* hs, float* out,
                        int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                        const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                        const long long* b_bstride, int N, int H, int W, int C, int ldw, long long stat_bstride, int up, int act,
                        const float* wc, int ldwc, int Cout, const float* bias, const float* res, const float* wscale,
                        double* stats, int stats_slots, int stats_prezeroed, fsv_stream_t stream);
/* backward twin of fsv_spade_mod_fwd: the same operands plus the upstream gradient dh; gamma / beta are recomputed in
 * registers, outputs are dgb[k] = d(gamma | beta) of every map ([P][2C], gamma in columns [0, C)) and dxhat [P][C] (per
 * full-resolution pixel also when up != 0).  act: FSV_ACT_NONE or FSV_ACT_LRELU. */
int fsv_spade_mod_bwd(const float* x, const float* mean, const float* rstd, const float* dh,
                      int nmaps, const float* const* maps, const float* const* wg, const float* const* wb,
                      const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                      const long long* b_bstride, float* const* dgb, float* dxhat, int N, int HW, int C, int ldw,
                      long long stat_bstride, int act, int W, int up, fsv_stream_t stream);
/* general form of fsv_spade_mod_bwd.  flags bit 0: dh is IEEE half; bit 1: the d(gamma|beta) tensors are written as half; bit 2: f16
 * GEMMs (operands as fsv_spade_mod_fwd_h).  dbsum (optional): the bias gradients from this launch - per-channel sums of
 * d(gamma|beta) as doubles at dbsum + z * db_zstride[k] + k * 2C (zeroed by the call; db_zstride[k] = 0: summed over the batch);
 * db_slots > 1 (a power of two): pixel tile t adds into the copy at + (t % db_slots) * db_slot_stride, the caller sums the copies */
int fsv_spade_mod_bwd_h(const float* x, const float* mean, const float* rstd, const void* dh,
                        int nmaps, const void* const* maps, const void* const* wg, const void* const* wb,
                        const float* const* bg, const float* const* bb, const int* ch, const long long* w_bstride,
                        const long long* b_bstride, void* const* dgb, float* dxhat, int N, int HW, int C, int ldw,
                        long long stat_bstride, int act, int W, int up, int flags, double* dbsum, const long long* db_zstride,
                        int db_slots, long long db_slot_stride, fsv_stream_t stream);
/* element-wise part of the backward (general path, C % 16 != 0): from materialised gamma|beta ([P][2C] per map) to d(gamma|beta) and d(xhat) (dxhat is
 * written per full-resolution pixel also when up != 0: summing it over the 2x2 children gives the gradient of the
 * half-resolution normalised tensor) */
int fsv_spade_bwd_elem(const float* x, const float* mean, const float* rstd, const float* dh, const float* h,
                       int nmaps, const float* const* gb, float* const* dgb, float* dxhat,
                       int N, int HW, int C, long long stat_bstride, int act, int W, int up, fsv_stream_t stream);

/* ---- measurement only (csrc/stamp.hip): device-side time stamps, graph-capturable brackets for bench.py's roofline object ----
 * fsv_stamp: *slot = the GPU's constant-rate wall clock when the stream reaches this point; fsv_stamp_rate_khz: its rate. */
int fsv_stamp(unsigned long long* slot, fsv_stream_t stream);
int fsv_stamp_rate_khz(void);

/* ---- normalisation (csrc/norm.hip) - BatchNorm (apex SyncBatchNorm) normalization.py:33,80; InstanceNorm :35,82 ----
 * tensors are [G][P][C]: BatchNorm G=1, P=N*H*W; InstanceNorm G=N, P=H*W.  workspace: fsv_norm_workspace_doubles(). */
int fsv_norm_workspace_doubles(int G, int P, int C);
int fsv_norm_stats(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                   float* run_mean, float* run_var, float momentum, fsv_stream_t stream);
/* second stage alone: mean / rstd (+ running statistics) from partial sums part[g][slot][c][2] left by a producing kernel */
int fsv_norm_stats_finish(const double* part, float* mean, float* rstd, int G, int P, int C, int nslots, float eps,
                          float* run_mean, float* run_var, float momentum, int rep, fsv_stream_t stream);
/* statistics of a tensor that repeats every value of x `rep` times (nearest x2 up-sampling: rep = 4): mean / rstd are those
 * of x, the unbiased running-variance correction counts P * rep values */
int fsv_norm_stats_rep(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                       float* run_mean, float* run_var, float momentum, int rep, fsv_stream_t stream);
/* y_half / dx_half (nullable): half side output, see fsv_act_bwd below */
int fsv_norm_apply(const float* x, const float* mean, const float* rstd, const float* w, const float* b, float* y,
                   int G, int P, int C, int act, void* y_half, fsv_stream_t stream);
int fsv_norm_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                 double* workspace, float* s1, float* s2, float* dx, float* dw, float* db, int G, int P, int C, int act,
                 int fixed_stats, void* dx_half, fsv_stream_t stream);   /* fixed_stats: eval mode, mean / rstd are constants */
int fsv_colsum(const float* x, double* workspace, float* out, int G, int P, int C, int accumulate, fsv_stream_t stream);
/* one-launch forms of the three reductions above: the workgroup that finishes last on a channel slab (a ticket per slab in
 * `counters`) sums that slab's partials and writes the final values, so the separate finalize launch disappears.  counters:
 * at least 64 ints, zero before the first use; every launch leaves them zero again (launches that may run concurrently need
 * different ranges).  counters == NULL, or a tensor above FSV_NORM_FUSE_MAX_MB (default 1 MB; one workgroup per slab reads all
 * partials, which only pays while the reduction is launch-bound): the two-launch path above. */
int fsv_norm_stats_fused(const float* x, double* workspace, float* mean, float* rstd, int G, int P, int C, float eps,
                         float* run_mean, float* run_var, float momentum, int rep, int* counters, fsv_stream_t stream);
int fsv_norm_bwd_fused(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                       double* workspace, float* s1, float* s2, float* dx, float* dw, float* db, int G, int P, int C, int act,
                       int fixed_stats, int* counters, void* dx_half, fsv_stream_t stream);
int fsv_colsum_fused(const float* x, double* workspace, float* out, int G, int P, int C, int accumulate, int* counters,
                     fsv_stream_t stream);
/* cross-replica BatchNorm (opt-in; apex.parallel.SyncBatchNorm of the reference's multi-process path, normalization.py:15,33,80):
 * the device halves on either side of the host's all-reduce.  sums: doubles [2C] = {sum x, sum x^2} resp. {sum d, sum d*xhat};
 * count: values per channel over all ranks.  dw / db of an affine layer are the LOCAL sums (they travel with the gradients). */
int fsv_norm_sums(const float* x, double* workspace, double* sums, int P, int C, fsv_stream_t stream);
int fsv_norm_stats_from_sums(const double* sums, double count, float* mean, float* rstd, int C, float eps, float* run_mean,
                             float* run_var, float momentum, fsv_stream_t stream);
int fsv_norm_bwd_sums(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                      double* workspace, double* sums, int P, int C, int act, fsv_stream_t stream);
int fsv_norm_bwd_apply(const float* dy, const float* y, const float* x, const float* mean, const float* rstd, const float* w,
                       const float* s1, const float* s2, float* dx, int P, int C, int count, int act, void* dx_half,
                       fsv_stream_t stream);
/* the bias gradients of a whole backward pass in two launches: table[job][8] = {src [P][C] rows, dst float[C] (added into),
 * offset of the job's partials in `part` (doubles), P, C, rows_per_blk, nchunks, V | shared << 8}; tmap1 (job, chunk, slab) triples,
 * tmap2 (job, 4-channel block) pairs; fsv_colsum_plan gives {V, TX, nslabs, rows_per_blk, nchunks} for one [P][C] */
int fsv_colsum_plan(int P, int C, int* out);
int fsv_colsum_grouped(const long long* table, int njobs, const int* tmap1, int nblk1, const int* tmap2, int nblk2,
                       double* part, fsv_stream_t stream);

/* ---- flow warp (csrc/warp.hip) - replaces resample/get_grid base_network.py:13-37 (F.grid_sample bilinear, border,
 * align_corners=True); tap indices are bit-identical to ATen's.  strides in elements: (batch, channel, y, x). ------ */
int fsv_warp_fwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, float* out, int* taps,
                 int B, int C, int H, int W, const long long* img_strides, const long long* flow_strides,
                 const long long* out_strides, fsv_stream_t stream);
int fsv_warp_bwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, const float* gout,
                 float* gimg, float* gflow, int B, int C, int H, int W, const long long* img_strides,
                 const long long* flow_strides, const long long* gout_strides, const long long* gimg_strides,
                 const long long* gflow_strides, fsv_stream_t stream);
/* warp + occlusion-mask compositing in one pass (generator.py:214-227, :441-443 on top of resample): writes the warped image
 * AND the composite.  mode 0 (--spade_combine): comp = dense NHWC [B][H*W][C+1] = (warp, mask), the input of the image
 * embedding; mode 1: comp = raw * mask + warp * (1 - mask) with the given strides.  mask: [B,1,H,W], mask_strides = (batch, y,
 * x); C <= 8; tap indices as fsv_warp_fwd.  Backward: g_warp / g_comp are the upstream gradients (either may be NULL), the
 * outputs gimg (scatter-add, zero-initialised, gimg_strides), gflow [B,2,H,W], gmask [B,H,W], graw [B,C,H,W] (dense) are
 * each optional. */
int fsv_warp_compose_fwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, const float* mask,
                         const float* raw, float* warp, float* comp, int mode, int B, int C, int H, int W,
                         const long long* img_strides, const long long* flow_strides, const long long* mask_strides,
                         const long long* raw_strides, const long long* warp_strides, const long long* comp_strides,
                         fsv_stream_t stream);
int fsv_warp_compose_bwd(const float* img, const float* flow, const float* lin_x, const float* lin_y, const float* mask,
                         const float* raw, const float* g_warp, const float* g_comp, float* gimg, float* gflow, float* gmask,
                         float* graw, int mode, int B, int C, int H, int W, const long long* img_strides,
                         const long long* flow_strides, const long long* mask_strides, const long long* raw_strides,
                         const long long* g_warp_strides, const long long* g_comp_strides, const long long* gimg_strides,
                         fsv_stream_t stream);

/* ---- spectral norm (csrc/specnorm.hip) - torch.nn.utils.spectral_norm at architecture.py:60,81-84 etc. ----------- */
/* scratch: fsv_sn_scratch_floats(R, Cc) floats (t[Cc], s[R] and one partial row of W^T u per 64-row slab of W: the slabs are
 * summed in ascending order, no atomics - u, v and sigma are the same bits on every run) */
int fsv_sn_scratch_floats(int R, int Cc);
int fsv_sn_power_iter(const float* W, float* u, float* v, float* scratch, float* sig, int R, int Cc, float eps,
                      int training, fsv_stream_t stream);
/* every spectral-normalised layer of a network in four launches; W/u/v are arrays of device pointers (as 64-bit
 * integers), tmap_* map a flat block index to (layer, tile).  t_off[l]: the layer's t region in `scratch`,
 * cols * (1 + ceil(rows / 64)) floats (t, then the per-slab partials of W^T u); s_off[l]: rows floats */
int fsv_sn_power_iter_batched(const long long* W, const long long* u, const long long* v, const int* rows,
                              const int* cols, const int* t_off, const int* s_off, float* scratch,
                              long long scratch_floats, float* sig, float* snap, const int* u_off, const int* v_off,
                              int nlayers, const int* tmap_t, int nblk_t, const int* tmap_s, int nblk_s, float eps,
                              fsv_stream_t stream);
int fsv_sn_backward(const float* dWsn, const float* W, const float* u, const float* v, const float* sig, double* part,
                    float* dW, int R, int Cc, int accumulate, fsv_stream_t stream);

/* ---- element-wise helpers and the optimiser (csrc/elementwise.hip) ----------------------------------------------
 * nearest x2 up-sampling (generator.py:124, nn.Upsample), activations, Adam (base_model.py:39-48). */
int fsv_upsample2x_fwd(const float* x, float* y, int N, int H, int W, int C, fsv_stream_t stream);
int fsv_upsample2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, fsv_stream_t stream);
int fsv_act_fwd(const float* x, float* y, long long total, int act, fsv_stream_t stream);
/* Half side output of the element-wise producers (`--amp`): fsv_norm_apply (y_half), fsv_norm_bwd / fsv_norm_bwd_fused /
 * fsv_norm_bwd_apply and fsv_act_bwd (dx_half) take a nullable pointer - when set, the call also stores its result as IEEE half
 * there, same element order, one rounding of the fp32 value: the consumer convolution reads that copy instead of converting the
 * fp32 tensor.  Contract: C % 4 == 0 and fewer than 2^31 elements (fsv_act_bwd: total % 4 == 0); FSV_ERR_UNSUPPORTED with nothing
 * launched otherwise. */
int fsv_act_bwd(const float* dy, const float* y, float* dx, long long total, int act, float scale, void* dx_half,
                fsv_stream_t stream);
/* channel concatenation into one NHWC tensor (one call per source) and its gradient slices; occlusion-mask
 * compositing out = a*m + b*(1-m) (generator.py:217,224,441-443,498,563) */
int fsv_cat_put(const float* src, float* out, long long N, int C, long long P, const long long* strides, int Ct, int coff,
                fsv_stream_t stream);
/* out [N][P][Ct] dense NHWC = the C channels of a (batch, channel, pixel)-strided source followed by Ct - C zero channels
 * (inputs whose channel count is not a multiple of 4 are padded once for the float4 gather of the convolutions) */
int fsv_pad_channels(const float* src, float* out, long long N, int C, long long P, const long long* strides, int Ct,
                     fsv_stream_t stream);
/* the same with `out` written as IEEE half ([N][P][Ct] halves; the `--amp` path: the padded tensor is a convolution input).
 * Pixel-contiguous source planes (strides[2] == 1) and Ct % 8 == 0 only: FSV_ERR_UNSUPPORTED otherwise */
int fsv_pad_channels_h(const float* src, void* out, long long N, int C, long long P, const long long* strides, int Ct,
                       fsv_stream_t stream);
int fsv_cat_get(const float* dout, float* dst, long long N, int C, long long P, int Ct, int coff, fsv_stream_t stream);
int fsv_blend_fwd(const float* a, const float* b, const float* m, float* out, int N, int C, long long P,
                  const long long* a_strides, const long long* b_strides, const long long* out_strides, fsv_stream_t stream);
int fsv_blend_bwd(const float* a, const float* b, const float* m, const float* g, float* da, float* db, float* dm, int N, int C,
                  long long P, const long long* a_strides, const long long* b_strides, const long long* g_strides,
                  fsv_stream_t stream);
/* 2x2 stride-2 max pooling of the VGG19 feature stack (models/networks/vgg.py:45-59), NHWC */
int fsv_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, fsv_stream_t stream);
int fsv_maxpool2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, fsv_stream_t stream);
/* nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False) between the discriminators of a multi-scale pyramid
 * (reference models/networks/discriminator.py:28,56), NHWC; output (H - 1) / 2 + 1 by (W - 1) / 2 + 1 */
int fsv_avgpool3s2_fwd(const float* x, float* y, int N, int H, int W, int C, fsv_stream_t stream);
int fsv_avgpool3s2_bwd(const float* dy, float* dx, int N, int H, int W, int C, fsv_stream_t stream);
/* nn.AdaptiveAvgPool2d((OH, OW)) on NHWC tensors, any OH / OW - shrinking or growing (discriminator.py:146,153: AdaptiveDiscriminator.gen_conv_weights);
 * windows [floor(o in / out), ceil((o + 1) in / out)) as in ATen; the backward pass is a gather (no atomics) */
int fsv_adaptive_avgpool_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, fsv_stream_t stream);
int fsv_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH, int OW, fsv_stream_t stream);
/* softmax over the contiguous channel dimension of [rows][C] (nn.Softmax(dim=1) at generator.py:384) */
int fsv_softmax_rows_fwd(const float* x, float* y, long long rows, int C, fsv_stream_t stream);
int fsv_softmax_rows_bwd(const float* dy, const float* y, float* dx, long long rows, int C, fsv_stream_t stream);
/* state = {t, 1-beta1^t, 1-beta2^t, lr} on the device; gscale pre-multiplies the gradient (1/world_size) */
int fsv_adam_step(float* param, const float* grad, float* m, float* v, float* state, long long n, float beta1,
                  float beta2, float eps, float gscale, fsv_stream_t stream);
/* the same step issued in pieces (ranges of the flat buffers): tick != 0 advances the step count / bias corrections in `state`
 * (exactly one piece of a step), the other pieces - ordered behind it by the caller - only read them */
int fsv_adam_step_range(float* param, const float* grad, float* m, float* v, float* state, long long n, float beta1,
                        float beta2, float eps, float gscale, int tick, fsv_stream_t stream);

/* ---- losses, D-input packing, mask pooling (csrc/losses.hip) - models/networks/loss.py:69-83,130-138;
 * models/loss_collector.py:47-58,105-110,180; models/input_process.py:59 -------------------------------------------- */
/* ticket (nullable, fsv_l1_fwd / fsv_hinge_fwd): ONE zeroed int owned by this launch until it completes - the workgroup that takes
 * the last ticket sums the per-workgroup partials in index order (the bits of the two-launch form) and leaves the int at zero;
 * NULL: a second launch finishes the reduction */
int fsv_l1_fwd(const float* a, const float* b, float bconst, const float* m, int N, int C, long long P,
               const long long* a_strides, const long long* b_strides, double* part, float* loss, int* ticket,
               fsv_stream_t stream);
int fsv_l1_bwd(const float* a, const float* b, float bconst, const float* m, int N, int C, long long P,
               const long long* a_strides, const long long* b_strides, const float* gloss, float* da, float* db, float* dm,
               fsv_stream_t stream);
/* weighted sums of one-element loss tensors - `sum(lambda_i * term_i)` of loss_collector.py:60-67,85,161-162,204 and the sum of the
 * means of :218-219 - as one launch each way: out[0] = sum_i weights[i] * terms[i][0] (i ascending, fp32; terms: n <= 32 HOST-side
 * array of device pointers, weights: n host floats - both travel in the kernel argument); dterms[i] = weights[i] * g[0] */
int fsv_wsum_fwd(const float* const* terms, const float* weights, int n, float* out, fsv_stream_t stream);
int fsv_wsum_bwd(const float* weights, int n, const float* g, float* dterms, fsv_stream_t stream);
int fsv_hinge_fwd(const float* x, long long n, float sign, double* part, float* loss, int* ticket, fsv_stream_t stream);
int fsv_hinge_bwd(const float* x, long long n, float sign, const float* gloss, float* dx, fsv_stream_t stream);
int fsv_pack_d_input(const float* ref, const float* lab, const float* fake, const float* real, float* out,
                     int B, int Cr, int Cl, int Ci, long long P, const long long* ref_strides, const long long* lab_strides,
                     const long long* fake_strides, const long long* real_strides, fsv_stream_t stream);
/* the same for one image set only: out [B][P][Cr+Cl+Ci] = [ref | label | img] */
int fsv_pack_d_single(const float* ref, const float* lab, const float* img, float* out, int B, int Cr, int Cl, int Ci,
                      long long P, const long long* ref_strides, const long long* lab_strides, const long long* img_strides,
                      fsv_stream_t stream);
int fsv_unpack_d_grad(const float* dout, float* dfake, int B, int Ci, int Coff, int Ct, long long P, fsv_stream_t stream);
/* `--amp` forms: the packed discriminator input with Cto >= Cr + Cl + Ci channels (zero channels appended = the channel padding of
 * the first discriminator convolution) and, out_half != 0, as IEEE half; halves 2: [fake | real] on the batch axis (out [2B][P][Cto]),
 * 1: fake only (real may be NULL).  fsv_unpack_d_grad_h reads the half data gradient of that convolution. */
int fsv_pack_d_x(const float* ref, const float* lab, const float* fake, const float* real, void* out,
                 int B, int Cr, int Cl, int Ci, long long P, const long long* ref_strides, const long long* lab_strides,
                 const long long* fake_strides, const long long* real_strides, int halves, int Cto, int out_half,
                 fsv_stream_t stream);
int fsv_unpack_d_grad_h(const void* dout, float* dfake, int B, int Ci, int Coff, int Ct, long long P, fsv_stream_t stream);
/* DensePose part-group masks (models/input_process.py:64-94): x = pose channel [B, T, P] (strides sb, st, 1), N = B*T;
 * y[N][ngroups][P] = 1 where (x/2+0.5)*24 is within 0.1 of a member of group g0+g (9 groups; group 8 = face parts 23/24) */
int fsv_part_masks(const float* x, float* y, long long N, long long P, int T, long long sb, long long st, int g0, int ngroups,
                   fsv_stream_t stream);
int fsv_pool15(const float* x, float* y, int N, int H, int W, long long sn, long long sy, long long sx, int mode, float thresh,
               fsv_stream_t stream);

/* ---- face-region discriminator inputs (--add_face_D; models/face_refiner.py:32-39, 56-87), csrc/face.hip ----
 * boxes[n] = {ys, ye, xs, xe} of the face in label map n (pose [N][C][H][W], strides sn / sc, rows contiguous), computed
 * on the device; crop + F.interpolate(nearest, size S) of the last three channels of img for every sample in one launch,
 * and its gradient scattered into a zero-initialised dimg of img's layout. */
int fsv_face_boxes(const float* pose, long long sn, long long sc, int N, int C, int H, int W, int use_openpose,
                   int crop_smaller, int* boxes, fsv_stream_t stream);
int fsv_crop_resize_fwd(const float* img, long long sn, long long sc, long long sy, long long sx, int C, const int* boxes,
                        float* out, int N, int S, fsv_stream_t stream);
int fsv_crop_resize_bwd(const float* dout, const int* boxes, float* dimg, long long sn, long long sc, long long sy,
                        long long sx, int C, int N, int S, fsv_stream_t stream);
/* --refine_face (face_refiner.py:42-54 replace_face_region): out = img outside box n, clamp(bilinear resize of face[n]
 * [3][S][S] to the box size, -1, 1) inside (align_corners False); backward: dimg = dout outside the boxes, dface
 * (zero-initialised) receives the weighted dout where the clamp was inactive */
int fsv_paste_face_fwd(const float* img, const float* face, const int* boxes, float* out, int N, int H, int W, int S,
                       long long isn, long long isc, long long isy, long long isx, fsv_stream_t stream);
int fsv_paste_face_bwd(const float* dout, const float* out, const int* boxes, float* dimg, float* dface, int N, int H, int W,
                       int S, fsv_stream_t stream);

/* ---- the FlowNet2 teacher's three native operators, forward only (csrc/flownet_ops.hip) ----
 * correlation (correlation_cuda_kernel.cu:74-147; kernel_size 1): f1 / f2 NHWC [N][H][W][C] -> out NHWC
 *   [N][OH][OW][D*D], D = 2*(max_disp/stride2)+1, OH = ceil((H + 2*pad - 2*max_disp)/stride1), value = mean_c f1*f2.
 * resample2d (resample2d_kernel.cu:16-64): pixel-unit flow, clamped tap indices; strides = 4 x long long (n, c, y, x).
 * channelnorm (channelnorm_kernel.cu:18-60): out[n][p] = sqrt(sum_c x^2); x strides (sn, sc, sp). */
int fsv_correlation_fwd(const float* f1, const float* f2, float* out, int N, int H, int W, int C, int pad, int kernel_size,
                        int max_disp, int stride1, int stride2, fsv_stream_t stream);
int fsv_resample2d_fwd(const float* img, const float* flow, float* out, int N, int C, int H, int W,
                       const long long* img_strides, const long long* flow_strides, const long long* out_strides,
                       fsv_stream_t stream);
/* F.interpolate(mode='bilinear', align_corners=False) of the FlowNet2 teacher (flownet2_pytorch/models.py:119,137-142: x4 flow
 * up-sampling; models/flownet.py:66-77: resize to a multiple of 64 and back); element strides (n, c, y, x) */
int fsv_bilinear_resize_fwd(const float* in, float* out, int N, int C, int IH, int IW, int OH, int OW,
                            const long long* in_strides, const long long* out_strides, fsv_stream_t stream);
int fsv_channelnorm_fwd(const float* x, float* out, int N, int C, long long HW, long long sn, long long sc, long long sp,
                        fsv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FSV2V_H */
