/* hdrnet_amd.h -- C-ABI of the MI355X-native bilateral-grid slice / slice-apply
 * library (libhdrnet_amd.so, built for gfx950 only).
 *
 * This is the drop-in boundary for the ONE hot path of google/hdrnet: the four
 * TensorFlow custom ops of hdrnet/ops.  Each entry point below replaces the
 * device dispatch of one reference op and is what a binding for that op would
 * call (INTEGRATION.md shows the ctypes / TF-op / torch stubs):
 *
 *   hdrnet_bilateral_slice_apply_f32       <- BilateralSliceApplyOp<GpuDevice>::Compute
 *       hdrnet/ops/bilateral_slice_apply_op.cc:140-235 -> BilateralSliceApplyCudaLauncher
 *       hdrnet/ops/bilateral_slice_apply.cu.cc:368-382
 *   hdrnet_bilateral_slice_apply_grad_f32  <- BilateralSliceApplyGradOp<GpuDevice>::Compute
 *       bilateral_slice_apply_op.cc:249-362 -> BilateralSliceApplyGradCudaLauncher
 *       bilateral_slice_apply.cu.cc:384-417
 *   hdrnet_bilateral_slice_f32             <- BilateralSliceOp<GpuDevice>::Compute
 *       hdrnet/ops/bilateral_slice_op.cc:120-174 -> BilateralSliceCudaLauncher
 *       hdrnet/ops/bilateral_slice.cu.cc:230-244
 *   hdrnet_bilateral_slice_grad_f32        <- BilateralSliceGradOp<GpuDevice>::Compute
 *       bilateral_slice_op.cc:183-256 -> BilateralSliceGradCudaLauncher
 *       bilateral_slice.cu.cc:246-272
 *
 * Conventions (same as the reference ops, bilateral_slice_apply_op.cc:201-227):
 *   - every buffer is a DEVICE pointer to dense, C-contiguous float32 in the
 *     reference's NHWC layouts:
 *         grid  [B][GH][GW][GD][C]      C = Cout*Cj, channel c = i*Cj + j,
 *                                       Cj = Cin + (has_offset ? 1 : 0)
 *         guide [B][H][W]
 *         input [B][H][W][Cin]
 *         out   [B][H][W][Cout]         (slice: [B][H][W][C])
 *   - the caller owns and allocates every buffer, outputs included (the
 *     reference: OpKernelContext::allocate_output); the library keeps no
 *     state and allocates nothing;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the null stream) and the call returns without
 *     synchronising, like the reference launchers on device.stream();
 *   - re-entrant and thread-safe.  Process-wide state: a thread-local error text, the opt-in
 *     last-kernel name (hdrnet_enable_kernel_names; off by default) and read-mostly caches of device
 *     facts (compute-unit count per device ordinal, resident workgroups per kernel) -- nothing a
 *     result depends on;
 *   - never throws, never exits.  Return codes:
 *         HDRNET_OK                0
 *         HDRNET_INVALID_ARGUMENT  1   (TF: errors::InvalidArgument)
 *         HDRNET_RUNTIME_FAILURE   2   (TF: errors::Internal("... kernel failed."))
 *   - in the *_grad entry points a NULL output pointer skips that VJP (the
 *     reference skips outputs whose size() == 0, bilateral_slice_apply.cu.cc:393,401,409).
 *
 * Gradients follow the reference's CPU implementation
 * (bilateral_slice_apply.cc:84-259), which is the semantics its tests pin; the
 * reference CUDA backward kernels carry two indexing bugs.
 */
#ifndef HDRNET_AMD_H_
#define HDRNET_AMD_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDRNET_OK 0
#define HDRNET_INVALID_ARGUMENT 1
#define HDRNET_RUNTIME_FAILURE 2

/* Kernel-selection flags for the *_ex entry points (testing / benchmarking).
 * 0 = automatic (what the plain entry points use). */
#define HDRNET_KERNEL_AUTO 0u
#define HDRNET_KERNEL_GENERIC 1u /* runtime-shape one-thread-per-pixel kernels; \
                                    forward is bit-exact vs the reference CPU op */
#define HDRNET_KERNEL_FAST 2u    /* LDS-staged specialisations; INVALID_ARGUMENT \
                                    if the shape has none */

/* Bits 8..15 of `flags` select a benchmark variant inside the family (0 = library default).
 * Variants exist only in the TOOLS build of the library (libhdrnet_amd_tools.so,
 * include/hdrnet_amd_tools.h); the product library answers HDRNET_INVALID_ARGUMENT to any
 * non-zero variant. */
#define HDRNET_VARIANT(n) (((unsigned)(n) & 0xffu) << 8)

/* ABI version: major*10000 + minor*100 + patch.
 * Changes a C caller can observe:
 *   241  the non-_ex guide-network entry points (..._nnguide_f32, ..._upadd_f32, ..._io) evaluate the EXACT sigmoid
 *        (flags = 0); through 240 they used the hardware exp / rcp form whenever guide_out == NULL.  Pass
 *        HDRNET_GUIDE_SIGMOID_FAST to the _ex twins for the old behaviour (<= 2 ulp of the guide, ~10 % faster).
 *   250  gradients of grids with 9 .. 16 planes (luma_bins = 16) run on the fast pass; their workspace bound is twice
 *        the 8-plane one (..._grad_workspace_bytes).  A frame-sized grid gradient that still falls back to the generic
 *        gather kernel under HDRNET_KERNEL_AUTO prints one line on stderr per process.
 *   251  4 -> 4 with offset (20 grid channels): dgrid runs on the fast pass as two channel windows; the workspace bound of
 *        that shape is no longer 0 (..._grad_workspace_bytes).  dguide of a call with dgrid == NULL changes in its last
 *        bits (contraction of the grid's z difference: closer to the float64 value than before). */
int hdrnet_version(void);

/* Text of the last error raised on the calling thread ("" if none). */
const char* hdrnet_last_error(void);

/* Name of the kernel variant(s) the most recent successful call in this process launched
 * (e.g. "apply_fwd_rows/vec4", "apply_vjp_rows/vec4+grid_grad_mfma"); introspection for
 * tests and benchmarks only (process-wide: autograd calls the VJPs from its own thread). */
const char* hdrnet_last_kernel(void);

/* hdrnet_last_kernel() bookkeeping is OFF by default, so that a launch takes no lock and formats
 * no string; tests and benchmarks switch it on (non-zero) / off here, or with
 * HDRNET_AMD_KERNEL_NAMES=1 in the environment when the library is loaded. */
void hdrnet_enable_kernel_names(int on);

/* BilateralSliceApply forward.
 * out[b,y,x,i] = sum_j trilerp(grid[.., i, j]; x, y, guide[b,y,x]) * (j < Cin ? input[b,y,x,j] : 1)
 * Reference semantics: hdrnet/ops/bilateral_slice_apply.cc:24-82. */
int hdrnet_bilateral_slice_apply_f32(const float* grid, const float* guide,
                                     const float* input, float* out, int B,
                                     int H, int W, int GH, int GW, int GD,
                                     int Cin, int Cout, int has_offset,
                                     void* stream);

int hdrnet_bilateral_slice_apply_f32_ex(const float* grid, const float* guide,
                                        const float* input, float* out, int B,
                                        int H, int W, int GH, int GW, int GD,
                                        int Cin, int Cout, int has_offset,
                                        unsigned flags, void* stream);

/* Row-split forward: ONE frame (or batch of frames) cut into horizontal bands across several GPUs
 * (the optional intra-image split).  The buffers hold rows y0 .. y0 + rows - 1
 * only -- guide [B][rows][W], input [B][rows][W][Cin], out [B][rows][W][Cout] -- of frames that are
 * H_total rows high; the grid is the whole frame's.  The y coordinate keeps the reference's expression
 * on the FRAME height, gyf = (y0 + y + 0.5) * GH / H_total (hdrnet/ops/bilateral_slice_apply.cc:38,42),
 * so the bands of any partition, concatenated, equal the whole-frame call bit for bit (same kernel
 * family).  y0 = 0, rows = H_total is hdrnet_bilateral_slice_apply_f32.  Requires
 * 0 <= y0, y0 + rows <= H_total.  Forward only: the gradients of a band are the whole-frame VJPs of a
 * dout that is zero outside the band. */
int hdrnet_bilateral_slice_apply_rows_f32(const float* grid, const float* guide, const float* input,
                                          float* out, int B, int H_total, int y0, int rows, int W,
                                          int GH, int GW, int GD, int Cin, int Cout, int has_offset,
                                          void* stream);

int hdrnet_bilateral_slice_apply_rows_f32_ex(const float* grid, const float* guide, const float* input,
                                             float* out, int B, int H_total, int y0, int rows, int W,
                                             int GH, int GW, int GD, int Cin, int Cout, int has_offset,
                                             unsigned flags, void* stream);

/* Fused point-wise guide network + BilateralSliceApply forward (inference).
 * guide[b,y,x] = sigmoid(conv2[n] + sum_k conv2[k] * relu(conv1[k][Cin] + sum_j conv1[k][j] * input[b,y,x,j]))
 * is computed in registers and sliced immediately; it is written to `guide_out` [B][H][W] only if
 * that pointer is non-NULL.  The sigmoid is tf.nn.sigmoid's form, expf + an IEEE divide, unless the caller
 * asks for the fast one with HDRNET_GUIDE_SIGMOID_FAST in the `flags` of the ..._ex twin: v_exp_f32 +
 * v_rcp_f32, <= 2 ulp of the guide and ~1e-6 of the output's scale away, 9-11 % faster (these kernels are
 * instruction-bound).  A forward whose guide feeds a backward pass should keep the default: the guide's
 * VJP is steep near bin centres.  (Until round 5 the choice followed `guide_out == NULL`.)
 * This is HDRNetPointwiseNNGuide._guide (hdrnet/models.py:203-210) with
 * batch-norm folded, in the parameter layout hdrnet/bin/freeze_graph.py:170-184 exports
 * (guide_conv1.bin = [n][Cin+1], guide_conv2.bin = [n+1]) -- the fusion the reference's GL
 * renderer performs (benchmark/assets/gpyrnn.frag:42-63, benchmark/src/renderer.cc:119-171).
 * Supported: (Cin, Cout) in {(3,3), (1,1)}, W % 4 == 0, 16-B aligned buffers; otherwise
 * HDRNET_INVALID_ARGUMENT (run the guide network and the plain entry point instead). */
#define HDRNET_GUIDE_SIGMOID_FAST 0x10000u
int hdrnet_bilateral_slice_apply_nnguide_f32(const float* grid, const float* input,
                                             const float* guide_conv1,
                                             const float* guide_conv2, float* out,
                                             float* guide_out, int B, int H, int W,
                                             int GH, int GW, int GD, int Cin, int Cout,
                                             int has_offset, int n_feats, void* stream);
int hdrnet_bilateral_slice_apply_nnguide_f32_ex(const float* grid, const float* input,
                                                const float* guide_conv1,
                                                const float* guide_conv2, float* out,
                                                float* guide_out, int B, int H, int W,
                                                int GH, int GW, int GD, int Cin, int Cout,
                                                int has_offset, int n_feats, unsigned flags,
                                                void* stream);

/* HDRNET_GUIDE_RELU_PRESCALED (flag of the three ..._ex guide-network entry points; Cin = 3): guide_conv1 /
 * guide_conv2 are the arrays hdrnet_guide_nn_prescale_f32 wrote from the exported ones -- per feature k the row
 * {w0, b, w1, w2} * 2^-e_k ([n][4], 16-B aligned) and the mixing weight conv2[k] * 2^e_k ([n+1], 16-B aligned),
 * with 2^e_k >= 2 (|b_k| + x_max * sum_j |w_kj|).  Powers of two commute with every rounding of the evaluation, so
 * the guide is the plain evaluation's bit for bit for every input with |input_j| <= x_max (the hidden activation
 * then never exceeds 2^e_k, and relu(h) 2^-e_k is the [0, 1] clamp modifier of the last multiply-add instead of
 * separate maximum instructions: 128 instead of 208 vector instructions per 256 pixels in these instruction-bound
 * kernels).  An input beyond x_max may saturate a feature at 2^e_k: the caller names the range (x_max; the
 * shipped models pass 65536).  Prepared ONCE per parameter set -- a one-workgroup launch, on `stream`. */
#define HDRNET_GUIDE_RELU_PRESCALED 0x20000u
int hdrnet_guide_nn_prescale_f32(const float* guide_conv1, const float* guide_conv2, int n_feats, int Cin,
                                 float x_max, float* conv1_out, float* conv2_out, void* stream);

/* The standard model's one-pass inference: HDRNetCurves._guide (hdrnet/models.py:145-190) evaluated
 * in registers, then BilateralSliceApply, with the wire-format conversions of
 * hdrnet_bilateral_slice_apply_io -- what the reference's standard GL shader does
 * (benchmark/assets/std.frag:32-53; uniforms loaded by benchmark/src/renderer.cc:197-225).
 *   t_c   = ccm[c][Cin] + sum_j ccm[c][j] * in_j
 *   guide = clip(mix[Cin] + sum_c mix[c] * sum_k slopes[k][c] * relu(t_c - shifts[k][c]), 0, 1)
 * Parameters in the layout hdrnet/bin/freeze_graph.py:107-127 exports: guide_ccm [Cin][Cin+1]
 * (guide_ccm_f32_3x4.bin), guide_shifts / guide_slopes [npts][Cin] (guide_shifts_f32_16x3.bin,
 * guide_slopes_f32_16x3.bin), guide_mix [Cin+1] (guide_mix_matrix_f32_1x4.bin).  `guide_out`
 * [B][H][W] is written only if non-NULL.  Same support as hdrnet_bilateral_slice_apply_io. */
int hdrnet_bilateral_slice_apply_io_curves(const float* grid, const void* input, void* out, int B,
                                           int H, int W, int GH, int GW, int GD, int Cin,
                                           int Cout, int has_offset, int input_dtype,
                                           float input_white_level, int output_dtype,
                                           const float* guide_ccm, const float* guide_shifts,
                                           const float* guide_slopes, const float* guide_mix,
                                           int npts, float* guide_out, void* stream);

/* The same pass with the curves' lookup tables PREPARED once per parameter set (Cin = 3, npts <= 16).
 * hdrnet_curves_guide_prepare_f32 (a SET-UP call: one small launch on `stream`, then it WAITS for the stream to read one word
 * back -- not for the per-frame path, not inside a stream capture) sorts each channel's knots, sums the curve's value and
 * slope at every knot in float64, and cuts the knot range into 64 uniform cells -- per cell the knot inside it (or the last
 * one before it), the value there, the slopes on either side -- into `prepared`, a caller-owned, 16-B aligned buffer of
 * hdrnet_curves_guide_prepared_bytes(Cin) bytes.  A cell holds one knot: `*usable` = 1 if no two knots of a channel share a
 * cell (knots at least 1/63 of their channel's range apart: the reference's equidistant initialisation with room to drift),
 * else 0 -- then do not pass the buffer on.  hdrnet_bilateral_slice_apply_io_curves_prepared takes a USABLE buffer beside
 * the exported arrays: a pixel finds its cell by arithmetic (a monotone fp32 map, so the cell's knot is the only one left to
 * compare with) and evaluates  C + (t >= s ? A_hi : A_lo) (t - s)  from ONE 16-byte table read, where the plain entry point
 * has every workgroup sort the knots itself and every pixel walk a 4-level search tree: 0.86 x (fp32) / 0.75 x (uint8) the
 * time, the same guide to 1e-6 (the anchor knot is at most one cell away).  A buffer that was reported unusable gives a
 * wrong guide (no memory is touched out of bounds).  `prepared` == NULL: exactly hdrnet_bilateral_slice_apply_io_curves. */
size_t hdrnet_curves_guide_prepared_bytes(int Cin);
int hdrnet_curves_guide_prepare_f32(const float* guide_shifts, const float* guide_slopes, int npts, int Cin,
                                    void* prepared, size_t prepared_bytes, int* usable, void* stream);
int hdrnet_bilateral_slice_apply_io_curves_prepared(const float* grid, const void* input, void* out, int B,
                                                    int H, int W, int GH, int GW, int GD, int Cin,
                                                    int Cout, int has_offset, int input_dtype,
                                                    float input_white_level, int output_dtype,
                                                    const float* guide_ccm, const float* guide_shifts,
                                                    const float* guide_slopes, const float* guide_mix,
                                                    int npts, const void* prepared, float* guide_out,
                                                    void* stream);

/* Multi-scale output of HDRNetGaussianPyrNN (hdrnet/models.py:277-289): per pyramid level
 *   out = BilateralSliceApply(grid_level, guide_level, input_level)
 *         + resize_bilinear(coarser level's result -> H x W, align_corners = True)
 * in ONE pass: the up-sampled coarse image and the un-added level output never exist.  The
 * reference's GL renderer folds the levels in one shader too (benchmark/assets/gpyrnn.frag:65-86).
 * `coarse` is [B][Hc][Wc][Cout].  Give either `guide` [B][H][W] or the folded guide network
 * (guide_conv1 [n][Cin+1], guide_conv2 [n+1], evaluated in registers as in ..._nnguide_f32).
 * Supported: Cin = Cout = 3 with offset, W % 4 == 0, 16-B aligned buffers.
 *
 * hdrnet_resize_bilinear_f32: NHWC bilinear resize, align_corners = True -- the
 * tf.image.resize_images call that builds the multi-scale input (hdrnet/models.py:253-266).
 * TensorFlow legacy semantics (tensorflow_gpu==2.12.0, resize_bilinear_op.cc): scale =
 * (in-1)/float(out-1), src = i*scale, lower = floor, upper = min(ceil, in-1), lerp = src - lower. */
int hdrnet_bilateral_slice_apply_upadd_f32(const float* grid, const float* guide,
                                           const float* input, const float* coarse, int Hc,
                                           int Wc, float* out, int B, int H, int W, int GH,
                                           int GW, int GD, int Cin, int Cout, int has_offset,
                                           const float* guide_conv1, const float* guide_conv2,
                                           int n_feats, void* stream);
/* ... with `flags`: HDRNET_GUIDE_SIGMOID_FAST (see hdrnet_bilateral_slice_apply_nnguide_f32), HDRNET_GUIDE_RELU_PRESCALED or 0. */
int hdrnet_bilateral_slice_apply_upadd_f32_ex(const float* grid, const float* guide,
                                              const float* input, const float* coarse, int Hc,
                                              int Wc, float* out, int B, int H, int W, int GH,
                                              int GW, int GD, int Cin, int Cout, int has_offset,
                                              const float* guide_conv1,
                                              const float* guide_conv2, int n_feats,
                                              unsigned flags, void* stream);
int hdrnet_resize_bilinear_f32(const float* in, float* out, int B, int Hin, int Win, int Hout,
                               int Wout, int C, void* stream);

/* Training side of the point-wise guide network (HDRNetPointwiseNNGuide._guide,
 * hdrnet/models.py:203-210; conv + batch norm wrappers hdrnet/layers.py:23-58).  These replace the
 * TensorFlow-generated gradient sub-graph of the two 1x1 convolutions around the hot path; pixels
 * are addressed flat (npx = B*H*W), buffers must be 16-B aligned.
 *
 * hdrnet_pointwise_guide_grad_f32: VJP of
 *   guide = sigmoid(conv2[n] + sum_k conv2[k] * relu(conv1[k][Cin] + sum_j conv1[k][j] * in_j))
 * given `guide` (as written by ..._nnguide_f32's guide_out) and `dguide` (from
 * hdrnet_bilateral_slice_apply_grad_f32): writes dconv1 [n][Cin+1] and dconv2 [n+1]; if `dinput`
 * is non-NULL the guide path's share of the input gradient is added to it
 * (accumulate_dinput != 0: dinput already holds the slice path's share) or stored.
 * Deterministic.  Supported: Cin in {1, 3}, n_feats in {4, 8, 16}; workspace from
 * hdrnet_pointwise_guide_grad_workspace_bytes (0 = unsupported shape).
 *
 * hdrnet_input_moments_f32: sums[j] = sum_px in_j, moments[i][j] = sum_px in_i * in_j
 * (Cin + Cin*Cin floats).  The first convolution is linear, so the training-mode batch-norm
 * statistics of its n-channel output follow from these without materialising that tensor. */
size_t hdrnet_pointwise_guide_grad_workspace_bytes(long long npx, int Cin, int n_feats);
int hdrnet_pointwise_guide_grad_f32(const float* input, const float* guide, const float* dguide,
                                    const float* guide_conv1, const float* guide_conv2,
                                    float* dinput, int accumulate_dinput, float* dconv1,
                                    float* dconv2, long long npx, int Cin, int n_feats,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* VJP of the curves guide (HDRNetCurves._guide, hdrnet/models.py:145-190; formula and parameter
 * layouts at hdrnet_bilateral_slice_apply_io_curves): given dguide it writes dccm [Cin][Cin+1],
 * dshifts / dslopes [npts][Cin], dmix [Cin+1] and adds (accumulate_dinput != 0) or stores the guide
 * path's share of dinput (NULL: skipped).  The clip passes the gradient where the pre-clip value lies
 * in [0, 1].  Deterministic.  Supported: Cin = 3, npts = 16 (the reference hard-codes 16 knots). */
size_t hdrnet_curves_guide_grad_workspace_bytes(long long npx, int Cin, int npts);
int hdrnet_curves_guide_grad_f32(const float* input, const float* dguide, const float* guide_ccm,
                                 const float* guide_shifts, const float* guide_slopes,
                                 const float* guide_mix, float* dinput, int accumulate_dinput,
                                 float* dccm, float* dshifts, float* dslopes, float* dmix,
                                 long long npx, int Cin, int npts, void* workspace,
                                 size_t workspace_bytes, void* stream);
size_t hdrnet_input_moments_workspace_bytes(long long npx, int Cin);
int hdrnet_input_moments_f32(const float* input, long long npx, int Cin, float* sums,
                             float* moments, void* workspace, size_t workspace_bytes,
                             void* stream);

/* BilateralSliceApply forward with the product's wire formats fused in (inference):
 *   input  : HDRNET_F32, or HDRNET_U8 / HDRNET_U16 holding value / input_white_level --
 *            tf.to_float(im) / white_level of hdrnet/data_pipeline.py:202-232 (255, 65535) and
 *            :267-274 (HDR+: 32767);
 *   output : HDRNET_F32, or HDRNET_U8 = (uint8)(255 * clip(out, 0, 1)), truncating --
 *            hdrnet/bin/run.py:95;
 *   guide  : a [B][H][W] float map, or NULL to evaluate the folded point-wise guide network
 *            (guide_conv1 [n][Cin+1], guide_conv2 [n+1], see ..._nnguide_f32) in registers.
 * Supported: Cin = Cout = 3 with offset, W % 4 == 0, 16-B aligned float buffers, 4-B aligned
 * integer buffers; otherwise HDRNET_INVALID_ARGUMENT. */
#define HDRNET_F32 0
#define HDRNET_U8 1
#define HDRNET_U16 2
int hdrnet_bilateral_slice_apply_io(const float* grid, const float* guide, const void* input,
                                    void* out, int B, int H, int W, int GH, int GW, int GD,
                                    int Cin, int Cout, int has_offset, int input_dtype,
                                    float input_white_level, int output_dtype,
                                    const float* guide_conv1, const float* guide_conv2,
                                    int n_feats, float* guide_out, void* stream);
/* ... with `flags`: HDRNET_GUIDE_SIGMOID_FAST, HDRNET_GUIDE_RELU_PRESCALED (guide network only; see ..._nnguide_f32) or 0. */
int hdrnet_bilateral_slice_apply_io_ex(const float* grid, const float* guide, const void* input,
                                       void* out, int B, int H, int W, int GH, int GW, int GD,
                                       int Cin, int Cout, int has_offset, int input_dtype,
                                       float input_white_level, int output_dtype,
                                       const float* guide_conv1, const float* guide_conv2,
                                       int n_feats, float* guide_out, unsigned flags,
                                       void* stream);

/* Scratch (bytes) the grad entry point wants for its deterministic two-stage
 * grid-gradient reduction (partial tiles; no atomics anywhere); 0 is a legal answer
 * (no fast grid-gradient for the shape).  The caller passes a device buffer of at
 * least this size as `workspace` (contents undefined on entry and exit).  The size is
 * the bound over the launch plans of the CURRENT device (rows per task are fitted to
 * its compute-unit count): query it with the device current that the call will run on.
 * With workspace == NULL or too small the grid gradient runs on the generic gather
 * kernel instead (bit-exact to the reference CPU code, two orders of magnitude slower). */
size_t hdrnet_bilateral_slice_apply_grad_workspace_bytes(int B, int H, int W,
                                                         int GH, int GW, int GD,
                                                         int Cin, int Cout,
                                                         int has_offset);

/* BilateralSliceApply VJPs.  dgrid like grid, dguide like guide, dinput like
 * input; any of the three may be NULL (skipped).
 * Reference semantics: bilateral_slice_apply.cc:84-138 (grid), :140-206
 * (guide), :208-259 (input). */
int hdrnet_bilateral_slice_apply_grad_f32(
    const float* grid, const float* guide, const float* input,
    const float* dout, float* dgrid, float* dguide, float* dinput, int B, int H,
    int W, int GH, int GW, int GD, int Cin, int Cout, int has_offset,
    void* workspace, size_t workspace_bytes, void* stream);

int hdrnet_bilateral_slice_apply_grad_f32_ex(
    const float* grid, const float* guide, const float* input,
    const float* dout, float* dgrid, float* dguide, float* dinput, int B, int H,
    int W, int GH, int GW, int GD, int Cin, int Cout, int has_offset,
    void* workspace, size_t workspace_bytes, unsigned flags, void* stream);

/* BilateralSlice forward: out[b,y,x,c] = trilerp(grid[.., c]; x, y, guide).
 * Reference semantics: hdrnet/ops/bilateral_slice.cc:25-70. */
int hdrnet_bilateral_slice_f32(const float* grid, const float* guide, float* out,
                               int B, int H, int W, int GH, int GW, int GD,
                               int C, void* stream);

int hdrnet_bilateral_slice_f32_ex(const float* grid, const float* guide,
                                  float* out, int B, int H, int W, int GH,
                                  int GW, int GD, int C, unsigned flags,
                                  void* stream);

size_t hdrnet_bilateral_slice_grad_workspace_bytes(int B, int H, int W, int GH,
                                                   int GW, int GD, int C);

/* BilateralSlice VJPs (dgrid and/or dguide may be NULL).
 * Reference semantics: bilateral_slice.cc:72-118 (grid), :120-168 (guide). */
int hdrnet_bilateral_slice_grad_f32(const float* grid, const float* guide,
                                    const float* dout, float* dgrid,
                                    float* dguide, int B, int H, int W, int GH,
                                    int GW, int GD, int C, void* workspace,
                                    size_t workspace_bytes, void* stream);

int hdrnet_bilateral_slice_grad_f32_ex(const float* grid, const float* guide,
                                       const float* dout, float* dgrid,
                                       float* dguide, int B, int H, int W,
                                       int GH, int GW, int GD, int C,
                                       void* workspace, size_t workspace_bytes,
                                       unsigned flags, void* stream);

/* The training loop's l2 loss (hdrnet/metrics.py:8-11: reduce_mean(square(target - prediction)); minimised by
 * hdrnet/bin/train.py:95) over n fp32 elements, and its gradient with respect to the prediction:
 *   loss[0] = sum((prediction - target)^2) / n        (one pass over both tensors + a 2048-element reduction)
 *   dprediction = (2 / n) * grad_output[0] * (prediction - target)   (grad_output: a DEVICE scalar, the loss's upstream
 *   gradient -- no host synchronisation).  16-B aligned tensors. */
size_t hdrnet_l2_loss_workspace_bytes(long long n);
int hdrnet_l2_loss_f32(const float* prediction, const float* target, long long n, float* loss, void* workspace,
                       size_t workspace_bytes, void* stream);
int hdrnet_l2_loss_grad_f32(const float* prediction, const float* target, const float* grad_output, long long n,
                            float* dprediction, void* stream);

/* Training-mode fold of the guide network's batch norm into its first layer, from the input's moments
 * (hdrnet_input_moments_f32): the statistics of the first convolution's never-materialised output are
 *   mean_h = w1^T mean_x,  var_h[k] = w1[:,k]^T Cov_x w1[:,k]  (biased),  inv = gamma / sqrt(var_h + eps)
 *   conv1[k] = (w1[:,k] * inv, beta[k] - mean_h[k] * inv)  [n][Cin+1],   conv2 = (w2, b2)  [n+1]
 * -- tf.contrib.layers.batch_norm with is_training=True (hdrnet/layers.py:40-58), folded the way
 * hdrnet/bin/freeze_graph.py:170-184 folds the inference statistics.  w1 is [Cin][n].  running_mean / running_var
 * [n] (both or neither) are moved by `momentum` towards the batch statistics (unbiased variance), num_batches_tracked
 * (may be NULL) is incremented.  float64 arithmetic on the device, one launch; ..._grad_f32 is its VJP with respect
 * to w1, beta, w2, b2 given dconv1 [n][Cin+1] and dconv2 [n+1] (the moments are data).  Cin in {1, 3}. */
int hdrnet_guide_fold_batch_f32(const float* sums, const float* moments, long long npx, const float* w1,
                                const float* gamma, const float* beta, const float* w2, const float* b2,
                                double eps, double momentum, int Cin, int n_feats, float* conv1, float* conv2,
                                float* running_mean, float* running_var, long long* num_batches_tracked,
                                void* stream);
int hdrnet_guide_fold_batch_grad_f32(const float* sums, const float* moments, long long npx, const float* w1,
                                     const float* gamma, const float* beta, double eps, int Cin, int n_feats,
                                     const float* dconv1, const float* dconv2, float* dw1, float* dbeta,
                                     float* dw2, float* db2, void* stream);

/* ---- The low-resolution coefficient network: the caller of the hot path ----
 *
 * HDRNetCurves._coefficients (hdrnet/models.py:62-142; layer wrappers hdrnet/layers.py:25-93) in inference
 * mode: lowres [B][N][N][3] fp32 -> the bilateral grid [B][sb][sb][gd][n_out][n_in] the slice-apply entry points
 * read (the unroll of models.py:134-138: prediction channel (j*n_out + i)*gd + z -> [.., z, i, j]).
 *   splat       n_ds = log2(N / sb) stride-2 3x3 convs, padding SAME, ReLU; layer i has cm * 2^i * gd channels
 *   global      two stride-2 3x3 convs (8*cm*gd channels, ReLU), (h, w, c) flattening, fc 32*cm*gd (ReLU),
 *               fc 16*cm*gd (ReLU), fc 8*cm*gd
 *   local       3x3 conv (ReLU), 3x3 conv (no bias, no activation), both 8*cm*gd channels
 *   fusion      relu(local + global), then the 1x1 prediction conv to gd*n_out*n_in channels
 * Nine launches on `stream` (the convolutions as fp32 matrix-core implicit GEMMs; csrc/coeff_net.hip), no host
 * synchronisation, no allocation, deterministic (no atomics: two calls give identical bits).
 *
 * Parameters (device pointers, fp32, 16-B aligned), batch norm already FOLDED into weight and bias the way
 * hdrnet/bin/freeze_graph.py:170-184 folds the guide's (w * gamma / sqrt(var + eps), beta - mean * that):
 *   convolutions  [Cout][kh][kw][Cin] -- the TensorFlow variable [kh][kw][Cin][Cout] with Cout moved to the front
 *                 (= a torch Conv2d weight in channels_last memory order); bias [Cout], NULL = none
 *   fc layers     [in][out], TensorFlow's own layout; bias [out]
 * n_levels > 1 (HDRNetGaussianPyrNN: n_out = 9, n_levels = 3) writes the output level-major,
 * [n_levels][B][sb][sb][gd][n_out / n_levels][n_in] -- each level's grid contiguous, as the per-level
 * slice-applies of models.py:277-289 (coeffs[:, :, :, :, 3*l : 3*l + 3, :]) need it.
 * Supported: N <= 4096 and sb powers of two, N / sb in [2, 256], cm * gd a multiple of 4 with cm * gd / 4 a power of two,
 * B <= 65535; hdrnet_coefficients_workspace_bytes returns 0 otherwise (run the framework's own graph instead).  The
 * parameter arrays are read by the launches: keep them alive and unchanged until those have run. */
typedef struct hdrnet_coeff_net {
  int net_input_size;     /* N */
  int spatial_bin;        /* sb: the grid is sb x sb cells */
  int luma_bins;          /* gd */
  int channel_multiplier; /* cm */
  int n_out, n_in;        /* 3, 4 (pyramid model: 9, 4) */
  int n_levels;           /* 1 (pyramid model: 3) */
  const float* splat_w[8];
  const float* splat_b[8];
  const float* global_conv_w[2];
  const float* global_conv_b[2];
  const float* fc_w[3];
  const float* fc_b[3];
  const float* local_w[2];
  const float* local_b[2]; /* local_b[1] = NULL: the reference's use_bias=False (models.py:113-115) */
  const float* pred_w;
  const float* pred_b;
  int fc_layout; /* 0: fc weights [in][out] (TensorFlow's); 1: [out][in] (a torch Linear weight, as it is) */
} hdrnet_coeff_net;

size_t hdrnet_coefficients_workspace_bytes(const hdrnet_coeff_net* net, int B);

/* Training side: the VJP of the coefficient network with respect to every weight and bias, for the model WITHOUT batch
 * norm (how the reference's script trains the guide-network model: scripts/ll/train_nn_guide.sh, --nobatch_norm).
 * Forward = hdrnet_coefficients_f32 with the parameters AS TORCH HOLDS THEM -- Conv2d weights in channels_last memory
 * order ([Cout][kh][kw][Cin], the same layout as above) and fc_layout = 1 (Linear weights [out][in]) -- keeping its
 * workspace: `forward_workspace` here is that buffer, untouched since.  `dcoeffs` is [B][sb][sb][gd][n_out][n_in];
 * gradients are written (not accumulated) in the parameters' own layouts; local_b[1] is ignored (no such bias).
 * 15 launches on `stream`, deterministic.  Supported: what the forward supports, n_levels = 1, fc_layout = 1,
 * B <= 8, 8 * cm * gd <= 256; hdrnet_coefficients_grad_workspace_bytes returns 0 otherwise. */
typedef struct hdrnet_coeff_net_grads {
  float* splat_w[8];
  float* splat_b[8];
  float* global_conv_w[2];
  float* global_conv_b[2];
  float* fc_w[3];
  float* fc_b[3];
  float* local_w[2];
  float* local_b[2];
  float* pred_w;
  float* pred_b;
} hdrnet_coeff_net_grads;

size_t hdrnet_coefficients_grad_workspace_bytes(const hdrnet_coeff_net* net, int B);

int hdrnet_coefficients_grad_f32(const float* lowres, const hdrnet_coeff_net* net, const void* forward_workspace,
                                 const float* dcoeffs, const hdrnet_coeff_net_grads* grads, int B, void* workspace,
                                 size_t workspace_bytes, void* stream);

int hdrnet_coefficients_f32(const float* lowres, const hdrnet_coeff_net* net, float* coeffs, int B,
                            void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* HDRNET_AMD_H_ */
