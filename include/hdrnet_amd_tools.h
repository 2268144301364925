/* Tools build of the C-ABI library: libhdrnet_amd_tools.so.
 *
 * Same sources and the same entry points as libhdrnet_amd.so (include/hdrnet_amd.h), compiled with
 * -DHDRNET_TOOLS_BUILD, plus what only benchmarks and experiments need:
 *
 *   - kernel VARIANTS of BilateralSliceApply forward, selected with HDRNET_VARIANT(n) in the
 *     `flags` of hdrnet_bilateral_slice_apply_f32_ex (tools/ab_bench.py times them interleaved
 *     with the product kernel).  Variants 101-106 are memory skeletons that do NOT compute the op
 *     (their kernel name starts with "ABLATION"); they are the reason this is a separate library.
 *       2        one wavefront per tile            3..6   persistent software-pipelined stream
 *       7        per-lane strided stores           8      round-1 kernel + nontemporal loads
 *       9..11    2 / 3 / 4 quads per thread        19     the round-1 product kernel (apply_fwd_rows)
 *       (20..72, through round 4: the product kernel's own load / store / pixel-phase flavours, the ticketed tail, the
 *                per-workgroup timeline trace.  Removed in round 5 -- apply_fwd_seg.hip holds the product's
 *                configurations only; measurements in profiles/r02 .. r04, code in the history.)
 *       101, 103..106  memory skeletons; 107 an empty kernel with the product's launch geometry,
 *                108 skeleton 106 on flat 1024-pixel tasks
 *   - kernel variants of the GRADIENT entry points (HDRNET_VARIANT(n) in the flags of
 *     hdrnet_bilateral_slice{,_apply}_grad_f32_ex; tools/bwd_ab.py times them interleaved):
 *       2  bf16-split contraction (two v_mfma_f32_16x16x32_bf16 per 16 pixels)   3  un-fused kernels
 *       4..8  ABLATIONS of the fused pass (timing only, results are garbage): 4 pixels loaded once per
 *             wave, 5 no MFMAs, 6 prologue + epilogue only, 7 = 4 + 5, 8 the launch alone; 9 = the product
 *             pass with per-chunk phase stamps of wave 0 ([task][16][5] clock64 values in the buffer
 *             given to hdrnet_tools_set_trace; tools/exp/r02_exp29.py)
 *       11    (dgrid == NULL) the round-1 per-pixel VJP kernel (apply_vjp_rows) instead of apply_vjp_seg
 *       HDRNET_GG_RG (environment): rows per workgroup task
 *   - hdrnet_tools_set_trace: device buffer for the phase trace of gradient variant 9 and the coefficient network's
 *     per-workgroup timeline (tools/coeff_trace.py).
 *
 * Nothing in the product path links or loads this library.
 */
#ifndef HDRNET_AMD_TOOLS_H_
#define HDRNET_AMD_TOOLS_H_

#include "hdrnet_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* device_buf: sized by the tracing tool (tools/coeff_trace.py, tools/exp/r02_exp29.py); NULL disables. */
void hdrnet_tools_set_trace(void* device_buf);

/* Experiment knobs read at launch (knobs 0-2 and 7 drove the product kernel's removed flavours):
 *   5  hdrnet_bilateral_slice_apply_io, uint8 input + guide network: 1 = the hidden layer as bf16-split 4x4x4 matrix
 *      instructions (apply_fwd_io.hip; rejected on time, kept for the record) */
void hdrnet_tools_set_knob(int idx, int value);

/* EXPERIMENT (round 5, csrc/pyramid_onepass.hip): the multi-scale output of HDRNetGaussianPyrNN
 * (hdrnet/models.py:277-289) in ONE pass over the full-resolution frame -- a workgroup owns 4 rows of a `seg`-pixel
 * row segment and re-evaluates the coarse pixels they tap.  grids / inputs / conv1 / conv2 [0] = full resolution,
 * [1] = half, [2] = quarter (inputs as hdrnet_resize_bilinear_f32 makes them; H % 4 == 0, W % 16 == 0).  Timed against
 * the product's per-level chain by tools/pyramid_onepass_bench.py (profiles/r05/pyramid_onepass.md). */
int hdrnet_tools_pyramid_onepass_f32(const float* const grids[3], const float* const inputs[3],
                                     const float* const conv1[3], const float* const conv2[3], int n_feats,
                                     float* out, int B, int H, int W, int GH, int GW, int GD, int seg,
                                     unsigned flags, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* HDRNET_AMD_TOOLS_H_ */
