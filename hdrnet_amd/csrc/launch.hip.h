// Argument bundles and launcher prototypes shared by the kernel translation
// units and the C-ABI front-end (capi.hip).  All pointers are device pointers in
// the layouts documented in include/hdrnet_amd.h.
#pragma once

#include <hip/hip_runtime.h>

#include <stddef.h>

#include "../../include/hdrnet_amd.h"

namespace hdrnet_amd {

struct ApplyArgs {
  const float* grid;
  const float* guide;
  const float* input;
  float* out;
  int B, H, W, GH, GW, GD, Cin, Cout, Cj;  // Cj = Cin + has_offset
  bool has_offset;
  int variant;  // 0 = library default; >0 selects a kernel variant (flags bits 8..15)
  // Row-split launch (hdrnet_bilateral_slice_apply_rows_f32): the buffers hold rows y0 .. y0 + H - 1 of a
  // frame that is H_total rows high; gyf = (y0 + y + .5) * GH / H_total (bilateral_slice_apply.cc:38,42).
  // H_total = 0: a whole frame (H_total = H, y0 = 0).
  int y0 = 0, H_total = 0;
  int frame_rows() const { return H_total > 0 ? H_total : H; }
  // fused guide network: v_exp_f32 + v_rcp_f32 sigmoid (HDRNET_GUIDE_SIGMOID_FAST) instead of expf + IEEE divide
  bool fast_sigmoid = false;
  // fused guide network: conv1 / conv2 are the PRESCALED arrays of hdrnet_guide_nn_prescale_f32 (HDRNET_GUIDE_RELU_PRESCALED)
  bool guide_prescaled = false;
};

// Forward with wire-format conversion and / or the fused guide network (apply_fwd_io.hip).
struct ApplyIoArgs {
  const float* grid;
  const float* guide;  // null => guide network (guide_conv1 / guide_conv2)
  const void* input;   // input_dtype: 0 f32, 1 u8, 2 u16; value / white_level
  void* out;           // output_dtype: 0 f32, 1 u8 = (uint8)(255 * clip(v, 0, 1))
  int B, H, W, GH, GW, GD, Cin, Cout;
  bool has_offset;
  int input_dtype, output_dtype;
  float white_level;
  const float* guide_conv1;  // NN: conv1 [n][Cin+1]; curves: ccm [Cin][Cin+1]
  const float* guide_conv2;  // NN: conv2 [n+1];      curves: mix [Cin+1]
  int n_feats;               // NN: features;         curves: knots per channel
  float* guide_out;  // optional
  const float* guide_shifts = nullptr;  // non-null selects the curves guide: [n][Cin]
  const float* guide_slopes = nullptr;  //                                    [n][Cin]
  bool fast_sigmoid = false;            // guide network: as ApplyArgs::fast_sigmoid
  bool guide_prescaled = false;         // guide network: as ApplyArgs::guide_prescaled
  const float* guide_prepared = nullptr;  // curves guide: the cell tables of hdrnet_curves_guide_prepare_f32, or null
};

struct ApplyGradArgs {
  const float* grid;
  const float* guide;
  const float* input;
  const float* dout;
  float* dgrid;   // may be null
  float* dguide;  // may be null
  float* dinput;  // may be null
  int B, H, W, GH, GW, GD, Cin, Cout, Cj;
  bool has_offset;
  void* workspace;
  size_t workspace_bytes;
  int variant = 0;  // tools build A/B: 2 = bf16-split dgrid contraction, 3 = separate (un-fused) kernels
};

// Training side of the point-wise guide network (guide_grad.hip).
struct GuideGradArgs {
  const float* input;   // [npx][Cin]
  const float* guide;   // [npx] the forward's guide (sigmoid output)
  const float* dguide;  // [npx]
  const float* conv1;   // [n][Cin + 1]
  const float* conv2;   // [n + 1]
  float* dinput;        // [npx][Cin] or null
  bool accumulate_dinput;  // true: dinput += guide path's share; false: dinput = it
  float* dconv1;        // [n][Cin + 1]
  float* dconv2;        // [n + 1]
  long long npx;
  int Cin, n_feats;
  void* workspace;
  size_t workspace_bytes;
};

// VJP of the curves guide (guide_grad.hip).  Parameter layouts as exported by the reference.
struct CurvesGradArgs {
  const float* input;   // [npx][3]
  const float* dguide;  // [npx]
  const float *ccm, *shifts, *slopes, *mix;  // [3][4], [16][3], [16][3], [4]
  float* dinput;        // [npx][3] or null
  bool accumulate_dinput;
  float *dccm, *dshifts, *dslopes, *dmix;
  long long npx;
  int Cin, npts;
  void* workspace;
  size_t workspace_bytes;
};

struct SliceArgs {
  const float* grid;
  const float* guide;
  float* out;
  int B, H, W, GH, GW, GD, C;
};

struct SliceGradArgs {
  const float* grid;
  const float* guide;
  const float* dout;
  float* dgrid;   // may be null
  float* dguide;  // may be null
  int B, H, W, GH, GW, GD, C;
  void* workspace;
  size_t workspace_bytes;
  int variant = 0;  // as ApplyGradArgs
};

// The (Cin, Cout, has_offset) shapes every fast BilateralSliceApply path specialises -- ONE table for the
// forward (apply_fwd_seg / apply_fwd_rows), the per-pixel VJPs (apply_vjp_seg / apply_vjp_rows) and the
// grid VJP (grid_grad_mfma: one 16-column MFMA tile for C = Cout * Cj <= 16, fused with the per-pixel VJPs where Cj = 4;
// (4, 4, offset) has C = 20 and runs dgrid as two channel windows beside apply_vjp_seg).  X(CIN, COUT, OFFSET).
#define HDRNET_APPLY_FAST_SHAPES(X) \
  X(3, 3, true) X(3, 3, false) X(3, 4, true) X(1, 1, true) X(1, 1, false) X(1, 3, true) X(4, 4, true) X(4, 4, false)

inline bool apply_fast_shape(int Cin, int Cout, bool has_offset) {
#define HDRNET_SHAPE_EQ(CI, CO, OFF) if (Cin == CI && Cout == CO && has_offset == OFF) return true;
  HDRNET_APPLY_FAST_SHAPES(HDRNET_SHAPE_EQ)
#undef HDRNET_SHAPE_EQ
  return false;
}

// generic_kernels.hip -- any shape, bit-exact vs the reference CPU op.
hipError_t launch_apply_fwd_generic(const ApplyArgs& a, hipStream_t s);
hipError_t launch_apply_grad_generic(const ApplyGradArgs& a, hipStream_t s);
hipError_t launch_slice_fwd_generic(const SliceArgs& a, hipStream_t s);
hipError_t launch_slice_grad_generic(const SliceGradArgs& a, hipStream_t s);

// apply_fwd_rows.hip -- LDS-staged row-segment forward.  `*_supported` says whether a
// specialisation exists for the shape; `name` receives a static string naming
// the variant launched.
bool apply_fwd_rows_supported(const ApplyArgs& a);
hipError_t launch_apply_fwd_rows(const ApplyArgs& a, hipStream_t s, const char** name);
// apply_fwd_variants.hip -- benchmark-only alternatives, reached only with a non-zero variant
// number; hipErrorNotSupported = no such variant for the shape.
hipError_t launch_apply_fwd_variant(const ApplyArgs& a, hipStream_t s, const char** name);
hipError_t launch_apply_fwd_rows_direct_stores(const ApplyArgs& a, hipStream_t s, const char** name, int which);
// apply_fwd_seg.hip -- the product forward for 16-B-aligned, W % 4 == 0 inputs: padded LDS image,
// LDS-DMA nontemporal pixel loads, streaming (write-through / nontemporal) buffer stores.  launch_apply_fwd_rows routes here
// when supported (else the scalar kernel of apply_fwd_rows.hip).
bool apply_fwd_seg_supported(const ApplyArgs& a);
hipError_t launch_apply_fwd_seg(const ApplyArgs& a, hipStream_t s, const char** name);
// the same kernel with the guide network evaluated in registers / the coarser pyramid level added
bool apply_fwd_seg_nnguide_supported(const ApplyArgs& a, const float* guide_out);
hipError_t launch_apply_fwd_seg_nnguide(const ApplyArgs& a, const float* conv1, const float* conv2, int n_feats,
                                        float* guide_out, hipStream_t s, const char** name);
bool apply_fwd_seg_upadd_supported(const ApplyArgs& a, const float* coarse, bool guide_nn);
hipError_t launch_apply_fwd_seg_upadd(const ApplyArgs& a, const float* coarse, int Hc, int Wc, const float* conv1,
                                      const float* conv2, int n_feats, hipStream_t s, const char** name);
#ifdef HDRNET_TOOLS_BUILD
void tools_set_knob(int idx, int value);  // apply_fwd_variants.hip: experiment knobs (include/hdrnet_amd_tools.h)
int tools_knob(int idx);
void grid_grad_set_trace(long long* device_buf);
void coeff_net_set_trace(long long* device_buf);
#endif

// Fused point-wise-NN guide + slice-apply forward (apply_fwd_rows.hip, GUIDE_NN).
bool apply_fwd_nnguide_supported(const ApplyArgs& a, const float* guide_out);
hipError_t launch_apply_fwd_nnguide(const ApplyArgs& a, const float* conv1, const float* conv2,
                                    int n_feats, float* guide_out, hipStream_t s,
                                    const char** name);

// Slice-apply + bilinear (align_corners) up-add of the coarser pyramid level (apply_fwd_rows.hip,
// UPADD); conv1 != null additionally fuses the guide network.
bool apply_fwd_upadd_supported(const ApplyArgs& a, const float* coarse, bool guide_nn);
hipError_t launch_apply_fwd_upadd(const ApplyArgs& a, const float* coarse, int Hc, int Wc,
                                  const float* conv1, const float* conv2, int n_feats,
                                  hipStream_t s, const char** name);

// resize_bilinear.hip -- NHWC bilinear resize, align_corners = true (TF legacy semantics).
hipError_t launch_resize_bilinear(const float* in, float* out, int B, int Hin, int Win, int Hout,
                                  int Wout, int C, hipStream_t s, const char** name);

bool apply_fwd_io_supported(const ApplyIoArgs& a);
hipError_t launch_apply_fwd_io(const ApplyIoArgs& a, hipStream_t s, const char** name);
// the curves guide's uniform cell tables (apply_fwd_io.hip: CurveCells), prepared once per parameter set
size_t curves_guide_prepared_bytes(int Cin);  // 0: no cell tables for this channel count
size_t curves_guide_prepared_ok_offset(int Cin);  // float index of the buffer's `ok` word
hipError_t launch_curves_guide_prepare(const float* shifts, const float* slopes, int npts, int Cin, float* prepared,
                                       hipStream_t s);

// apply_bwd_rows.hip -- LDS-staged per-pixel VJPs: dguide and dinput in one pass
// (BilateralSliceApply), dguide (BilateralSlice).  dgrid is not their business.
bool apply_vjp_rows_supported(const ApplyGradArgs& a);
hipError_t launch_apply_vjp_rows(const ApplyGradArgs& a, hipStream_t s, const char** name);
// the same on the product forward's core (apply_vjp_seg.hip); launch_apply_vjp_rows routes here when it applies
bool apply_vjp_seg_supported(const ApplyGradArgs& a);
hipError_t launch_apply_vjp_seg(const ApplyGradArgs& a, hipStream_t s, const char** name);
bool slice_vjp_rows_supported(const SliceGradArgs& a);
hipError_t launch_slice_vjp_rows(const SliceGradArgs& a, hipStream_t s, const char** name);

// slice_fwd_rows.hip -- LDS-staged BilateralSlice forward with lane-contiguous stores.
bool slice_fwd_rows_supported(const SliceArgs& a);
hipError_t launch_slice_fwd_rows(const SliceArgs& a, hipStream_t s, const char** name);

// grid_grad_mfma.hip -- deterministic two-stage dgrid: per-row-run fp32 MFMA contraction over
// the pixels + fixed-order reduction of partial tiles held in the caller's workspace.
size_t apply_grid_grad_mfma_workspace(int B, int H, int W, int GH, int GW, int GD, int Cin, int Cout,
                                      bool has_offset);
bool apply_grid_grad_mfma_supported(const ApplyGradArgs& a);  // shape AND workspace large enough
hipError_t launch_apply_grid_grad_mfma(const ApplyGradArgs& a, hipStream_t s, const char** name);
// The same pass also producing dguide / dinput (fused backward: pixels read once for all gradients).
bool apply_bwd_fused_supported(const ApplyGradArgs& a);
hipError_t launch_apply_bwd_fused(const ApplyGradArgs& a, hipStream_t s, const char** name);
size_t slice_grid_grad_mfma_workspace(int B, int H, int W, int GH, int GW, int GD, int C);
bool slice_grid_grad_mfma_supported(const SliceGradArgs& a);
hipError_t launch_slice_grid_grad_mfma(const SliceGradArgs& a, hipStream_t s, const char** name);
bool slice_bwd_fused_supported(const SliceGradArgs& a);
hipError_t launch_slice_bwd_fused(const SliceGradArgs& a, hipStream_t s, const char** name);

// guide_grad.hip -- VJP of the folded point-wise guide network; input moments for batch norm.
size_t guide_grad_workspace_bytes(long long npx, int Cin, int n);
bool guide_grad_supported(const GuideGradArgs& a);
hipError_t launch_guide_grad(const GuideGradArgs& a, hipStream_t s, const char** name);
size_t curves_grad_workspace_bytes(long long npx, int Cin, int npts);
bool curves_grad_supported(const CurvesGradArgs& a);
hipError_t launch_curves_grad(const CurvesGradArgs& a, hipStream_t s, const char** name);
size_t input_moments_workspace_bytes(long long npx, int Cin);
hipError_t launch_input_moments(const float* input, long long npx, int Cin, float* sums, float* moments,
                                void* workspace, hipStream_t s, const char** name);

// guide_grad.hip -- training-mode fold of the guide network's batch norm (statistics from the input's moments) and its VJP.
hipError_t launch_guide_nn_prescale(const float* conv1, const float* conv2, int n_feats, float x_max, float* conv1_out,
                                    float* conv2_out, hipStream_t s);
hipError_t launch_guide_fold_batch(const float* sums, const float* moments, long long npx, const float* w1,
                                   const float* gamma, const float* beta, const float* w2, const float* b2, double eps,
                                   double momentum, int Cin, int n, float* conv1, float* conv2, float* running_mean,
                                   float* running_var, long long* num_batches_tracked, hipStream_t s);
hipError_t launch_guide_fold_batch_grad(const float* sums, const float* moments, long long npx, const float* w1,
                                        const float* gamma, const float* beta, double eps, int Cin, int n,
                                        const float* dconv1, const float* dconv2, float* dw1, float* dbeta, float* dw2,
                                        float* db2, hipStream_t s);

// metrics.hip -- hdrnet/metrics.py's l2 loss and its gradient with respect to the prediction.
size_t l2_loss_workspace_bytes(long long n);
hipError_t launch_l2_loss(const float* pred, const float* target, long long n, float* loss, void* workspace,
                          hipStream_t s);
hipError_t launch_l2_loss_grad(const float* pred, const float* target, const float* grad_output, long long n,
                               float* dpred, hipStream_t s);

// coeff_net.hip -- the low-resolution coefficient network (hdrnet/models.py:62-142) as inference kernels.
bool coefficients_supported(const hdrnet_coeff_net& net);
size_t coefficients_workspace_bytes(const hdrnet_coeff_net& net, int B);  // 0: unsupported hyper-parameters
hipError_t launch_coefficients(const float* lowres, const hdrnet_coeff_net& net, float* coeffs, int B, void* workspace,
                               hipStream_t s, const char** name);
// coeff_net_train.hip -- its VJP with respect to the parameters (no batch norm); fwd_ws = the forward's workspace.
size_t coefficients_grad_workspace_bytes(const hdrnet_coeff_net& net, int B);  // 0: not supported
hipError_t launch_coefficients_grad(const float* lowres, const hdrnet_coeff_net& net, const hdrnet_coeff_net_grads& gr,
                                    const float* dcoeffs, int B, const void* fwd_ws, void* workspace, hipStream_t s,
                                    const char** name);

}  // namespace hdrnet_amd
