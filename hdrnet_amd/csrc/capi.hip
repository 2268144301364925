// C-ABI front-end of libhdrnet_amd.so: argument validation (the OP_REQUIRES checks
// of hdrnet/ops/bilateral_slice_apply_op.cc:147-193 and bilateral_slice_op.cc:129-147
// re-expressed as return codes), kernel selection, asynchronous launch on the
// caller's stream.  See include/hdrnet_amd.h for the contract.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "../../include/hdrnet_amd.h"
#include "launch.hip.h"

namespace {

thread_local char g_error[512] = "";

// Introspection only (tests / benchmarks): name of the kernel(s) the most recent successful
// call launched.  OFF by default -- a launch then does no bookkeeping at all (no lock, no
// formatting); hdrnet_enable_kernel_names(1), or HDRNET_AMD_KERNEL_NAMES=1 in the environment at
// load time, switches it on.  Process-wide rather than thread-local because autograd runs
// backward on a thread of its own.  Not used for any decision.
std::atomic<int> g_kernel_names{-1};  // -1: consult the environment on first use
std::mutex g_kernel_mu;
char g_kernel_buf[128] = "";

bool kernel_names_on() {
  int v = g_kernel_names.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("HDRNET_AMD_KERNEL_NAMES");
    v = (e && *e && *e != '0') ? 1 : 0;
    g_kernel_names.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

void set_kernel(const char* a, const char* b = "", const char* c = "") {
  if (!kernel_names_on()) return;
  std::lock_guard<std::mutex> lock(g_kernel_mu);
  snprintf(g_kernel_buf, sizeof(g_kernel_buf), "%s%s%s%s%s", a, (*a && *b) ? "+" : "", b,
           ((*a || *b) && *c) ? "+" : "", c);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(hipError_t e, const char* what) {
  if (e != hipSuccess) {
    // TF: errors::Internal("BilateralSliceApply kernel failed.")
    return fail(HDRNET_RUNTIME_FAILURE, "%s kernel failed: %s", what, hipGetErrorString(e));
  }
  g_error[0] = '\0';
  return HDRNET_OK;
}

bool positive(int v) { return v > 0; }

// A grid gradient on the generic gather kernel (the reference's own design, bilateral_slice_apply.cc:84-138: every grid
// element loops over its +-1-cell pixel window) is ~100x slower than the contraction pass.  HDRNET_KERNEL_AUTO falls
// back to it for shapes the pass has no specialisation for (GD > 16, C > 32, an unlisted channel combination) or
// without a workspace; on a frame-sized call that is a performance cliff worth one line on stderr per process.
constexpr long long kWarnGenericPixels = 65536;
void warn_generic_grid_grad(const char* op, long long npix, int GD, int C, bool have_workspace) {
  static std::atomic<bool> said{false};
  if (npix <= kWarnGenericPixels || said.exchange(true)) return;
  fprintf(stderr, "hdrnet_amd: %s on %lld pixels (GD=%d, C=%d) takes the generic grid-gradient kernel, ~100x slower "
          "than the fast pass (%s)\n", op, npix, GD, C,
          have_workspace ? "no fast specialisation for this shape: needs GD <= 16, C <= 32 and a listed channel combination"
                         : "no workspace passed: see hdrnet_bilateral_slice*_grad_workspace_bytes");
}

// Extents >= 0; a zero-sized batch / image is a legal no-op, a zero-sized grid is not.
int check_common(int B, int H, int W, int GH, int GW, int GD) {
  if (B < 0 || H < 0 || W < 0)
    return fail(HDRNET_INVALID_ARGUMENT, "negative image extent (B=%d, H=%d, W=%d)", B, H, W);
  if (!positive(GH) || !positive(GW) || !positive(GD))
    return fail(HDRNET_INVALID_ARGUMENT, "grid extents must be positive (GH=%d, GW=%d, GD=%d)",
                GH, GW, GD);
  if ((long long)B * H * W > 0x7fffffffLL * 64)
    return fail(HDRNET_INVALID_ARGUMENT, "image too large");
  if ((long long)GH * GW * GD > 0x7fffffffLL / 4096)
    return fail(HDRNET_INVALID_ARGUMENT, "grid too large");
  return HDRNET_OK;
}

// flags: bits 0..7 kernel family (HDRNET_KERNEL_*), bits 8..15 variant inside the family
// (0 = the library's default; used by benchmarks for A/B runs), rest must be zero.
// flags of the guide-network entry points (..._nnguide_f32_ex, ..._upadd_f32_ex, ..._io_ex): the sigmoid choice and
// whether conv1 / conv2 are the prescaled arrays of hdrnet_guide_nn_prescale_f32
int check_guide_flags(unsigned flags) {
  if ((flags & ~(HDRNET_GUIDE_SIGMOID_FAST | HDRNET_GUIDE_RELU_PRESCALED)) != 0)
    return fail(HDRNET_INVALID_ARGUMENT,
                "unknown flags 0x%x (guide-network entry points take HDRNET_GUIDE_SIGMOID_FAST, HDRNET_GUIDE_RELU_PRESCALED)",
                flags);
  return HDRNET_OK;
}

// HDRNET_GUIDE_RELU_PRESCALED: a guide NETWORK of three input channels whose arrays are 16-B aligned ([n][4] rows, s_load_dwordx4)
int check_guide_prescaled(unsigned flags, int Cin, const float* conv1, const float* conv2) {
  if (!(flags & HDRNET_GUIDE_RELU_PRESCALED)) return HDRNET_OK;
  if (!conv1 || !conv2 || Cin != 3)
    return fail(HDRNET_INVALID_ARGUMENT, "HDRNET_GUIDE_RELU_PRESCALED needs a guide network with Cin = 3 (Cin=%d)", Cin);
  if (((uintptr_t)conv1 | (uintptr_t)conv2) & 15u)
    return fail(HDRNET_INVALID_ARGUMENT, "HDRNET_GUIDE_RELU_PRESCALED needs 16-B aligned guide_conv1 / guide_conv2 "
                                         "(the arrays hdrnet_guide_nn_prescale_f32 wrote)");
  return HDRNET_OK;
}

int check_flags(unsigned flags) {
  if ((flags & 0xffu) > HDRNET_KERNEL_FAST || (flags >> 16) != 0)
    return fail(HDRNET_INVALID_ARGUMENT, "unknown flags 0x%x", flags);
#ifndef HDRNET_TOOLS_BUILD
  if ((flags >> 8) != 0)
    return fail(HDRNET_INVALID_ARGUMENT,
                "kernel variants (flags bits 8..15) exist only in the tools build "
                "(libhdrnet_amd_tools.so), flags 0x%x", flags);
#endif
  return HDRNET_OK;
}
unsigned family(unsigned flags) { return flags & 0xffu; }
int variant(unsigned flags) { return (int)((flags >> 8) & 0xffu); }

}  // namespace

extern "C" {

// 0.2.4.1: + hdrnet_guide_nn_prescale_f32 / HDRNET_GUIDE_RELU_PRESCALED, hdrnet_curves_guide_prepare_f32 (with `usable`) /
//          ..._io_curves_prepared; the non-_ex guide-network entry points use the exact sigmoid (flags = 0)
// 0.2.5.0: the gradient entry points take grids of up to 16 planes on the fast pass (the workspace bound grows with it:
//          query ..._grad_workspace_bytes again); one stderr line when a frame-sized dgrid falls back to the generic kernel
// 0.2.5.1: 4 -> 4 with offset (C = 20): dgrid on the contraction pass as two channel windows (the workspace bound doubles
//          for that shape: query again); apply_vjp_seg's dguide in the z-difference form (bits change; closer to float64)
int hdrnet_version(void) { return 251; }

const char* hdrnet_last_error(void) { return g_error; }

const char* hdrnet_last_kernel(void) { return g_kernel_buf; }

void hdrnet_enable_kernel_names(int on) { g_kernel_names.store(on ? 1 : 0, std::memory_order_relaxed); }

#ifdef HDRNET_TOOLS_BUILD
// tools build only (include/hdrnet_amd_tools.h)
void hdrnet_tools_set_knob(int idx, int value) { hdrnet_amd::tools_set_knob(idx, value); }

void hdrnet_tools_set_trace(void* device_buf) {
  hdrnet_amd::grid_grad_set_trace(static_cast<long long*>(device_buf));
  hdrnet_amd::coeff_net_set_trace(static_cast<long long*>(device_buf));
}
#endif

// Forward, whole frames (H_total = rows, y0 = 0) or one row band of every frame.
static int apply_fwd_impl(const float* grid, const float* guide, const float* input, float* out, int B,
                          int H_total, int y0, int H, int W, int GH, int GW, int GD, int Cin, int Cout,
                          int has_offset, unsigned flags, void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_flags(flags)) return rc;
  if (H_total < 0 || y0 < 0 || (long long)y0 + H > H_total)
    return fail(HDRNET_INVALID_ARGUMENT, "row band [%d, %d + %d) outside the frame's %d rows", y0, y0, H, H_total);
  const bool band = y0 != 0 || H != H_total;
  if (band && variant(flags) != 0)
    return fail(HDRNET_INVALID_ARGUMENT, "kernel variants take whole frames");
  if (Cin < 0 || Cout <= 0 || Cin + (has_offset ? 1 : 0) <= 0)
    return fail(HDRNET_INVALID_ARGUMENT,
                "grid should have output_channels * (input_channels%s) channels "
                "(Cin=%d, Cout=%d)", has_offset ? " + 1" : "", Cin, Cout);
  const long long npix = (long long)B * H * W;
  if (npix == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!grid || !guide || !out || (Cin > 0 && !input))
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  ApplyArgs a{grid, guide, input, out, B, H, W, GH, GW, GD, Cin, Cout,
              Cin + (has_offset ? 1 : 0), has_offset != 0, variant(flags)};
  a.y0 = y0;
  a.H_total = H_total;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // a row band runs on the row-segment kernel (apply_fwd_seg.hip) or the generic one; whole frames may
  // also take the scalar row kernel (unaligned buffers, W % 4 != 0)
  const bool fast_ok = band ? apply_fwd_seg_supported(a) : apply_fwd_rows_supported(a);
  if (family(flags) == HDRNET_KERNEL_FAST && !fast_ok)
    return fail(HDRNET_INVALID_ARGUMENT, "no fast BilateralSliceApply variant for this shape");
  if (family(flags) != HDRNET_KERNEL_GENERIC && fast_ok) {
    const char* name = "";
    const hipError_t e = band ? launch_apply_fwd_seg(a, s, &name) : launch_apply_fwd_rows(a, s, &name);
    const int rc = check_launch(e, "BilateralSliceApply");
    if (rc == HDRNET_OK) set_kernel(name);
    return rc;
  }
  const int rc = check_launch(launch_apply_fwd_generic(a, s), "BilateralSliceApply");
  if (rc == HDRNET_OK) set_kernel("apply_fwd_generic");
  return rc;
}

int hdrnet_bilateral_slice_apply_f32_ex(const float* grid, const float* guide,
                                        const float* input, float* out, int B, int H, int W,
                                        int GH, int GW, int GD, int Cin, int Cout,
                                        int has_offset, unsigned flags, void* stream) {
  return apply_fwd_impl(grid, guide, input, out, B, H, 0, H, W, GH, GW, GD, Cin, Cout, has_offset, flags,
                        stream);
}

int hdrnet_bilateral_slice_apply_rows_f32_ex(const float* grid, const float* guide, const float* input,
                                             float* out, int B, int H_total, int y0, int rows, int W,
                                             int GH, int GW, int GD, int Cin, int Cout, int has_offset,
                                             unsigned flags, void* stream) {
  return apply_fwd_impl(grid, guide, input, out, B, H_total, y0, rows, W, GH, GW, GD, Cin, Cout, has_offset,
                        flags, stream);
}

int hdrnet_bilateral_slice_apply_rows_f32(const float* grid, const float* guide, const float* input,
                                          float* out, int B, int H_total, int y0, int rows, int W, int GH,
                                          int GW, int GD, int Cin, int Cout, int has_offset, void* stream) {
  return apply_fwd_impl(grid, guide, input, out, B, H_total, y0, rows, W, GH, GW, GD, Cin, Cout, has_offset,
                        HDRNET_KERNEL_AUTO, stream);
}

int hdrnet_bilateral_slice_apply_f32(const float* grid, const float* guide, const float* input,
                                     float* out, int B, int H, int W, int GH, int GW, int GD,
                                     int Cin, int Cout, int has_offset, void* stream) {
  return hdrnet_bilateral_slice_apply_f32_ex(grid, guide, input, out, B, H, W, GH, GW, GD, Cin,
                                             Cout, has_offset, HDRNET_KERNEL_AUTO, stream);
}

int hdrnet_bilateral_slice_apply_nnguide_f32(const float* grid, const float* input,
                                             const float* guide_conv1, const float* guide_conv2,
                                             float* out, float* guide_out, int B, int H, int W,
                                             int GH, int GW, int GD, int Cin, int Cout,
                                             int has_offset, int n_feats, void* stream) {
  return hdrnet_bilateral_slice_apply_nnguide_f32_ex(grid, input, guide_conv1, guide_conv2, out, guide_out, B, H, W,
                                                     GH, GW, GD, Cin, Cout, has_offset, n_feats, 0u, stream);
}

int hdrnet_bilateral_slice_apply_nnguide_f32_ex(const float* grid, const float* input,
                                                const float* guide_conv1, const float* guide_conv2,
                                                float* out, float* guide_out, int B, int H, int W,
                                                int GH, int GW, int GD, int Cin, int Cout,
                                                int has_offset, int n_feats, unsigned flags, void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_guide_flags(flags)) return rc;
  if (Cin <= 0 || Cout <= 0 || n_feats <= 0 || n_feats > 4096)
    return fail(HDRNET_INVALID_ARGUMENT, "bad channel / feature counts (Cin=%d, Cout=%d, n=%d)", Cin,
                Cout, n_feats);
  if ((long long)B * H * W == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!grid || !input || !out || !guide_conv1 || !guide_conv2)
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (int rc = check_guide_prescaled(flags, Cin, guide_conv1, guide_conv2)) return rc;
  ApplyArgs a{grid, nullptr, input, out, B, H, W, GH, GW, GD, Cin, Cout,
              Cin + (has_offset ? 1 : 0), has_offset != 0, 0};
  a.fast_sigmoid = (flags & HDRNET_GUIDE_SIGMOID_FAST) != 0;
  a.guide_prescaled = (flags & HDRNET_GUIDE_RELU_PRESCALED) != 0;
  if (!apply_fwd_nnguide_supported(a, guide_out))
    return fail(HDRNET_INVALID_ARGUMENT,
                "fused guide + slice-apply needs (Cin, Cout) in {(3,3), (1,1)}, W %% 4 == 0 and 16-B "
                "aligned buffers; run the guide network and hdrnet_bilateral_slice_apply_f32 instead");
  const char* name = "";
  const int rc = check_launch(launch_apply_fwd_nnguide(a, guide_conv1, guide_conv2, n_feats, guide_out,
                                                      static_cast<hipStream_t>(stream), &name),
                              "BilateralSliceApplyNNGuide");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

int hdrnet_bilateral_slice_apply_upadd_f32(const float* grid, const float* guide, const float* input,
                                           const float* coarse, int Hc, int Wc, float* out, int B,
                                           int H, int W, int GH, int GW, int GD, int Cin, int Cout,
                                           int has_offset, const float* guide_conv1,
                                           const float* guide_conv2, int n_feats, void* stream) {
  return hdrnet_bilateral_slice_apply_upadd_f32_ex(grid, guide, input, coarse, Hc, Wc, out, B, H, W, GH, GW, GD, Cin,
                                                   Cout, has_offset, guide_conv1, guide_conv2, n_feats, 0u, stream);
}

int hdrnet_bilateral_slice_apply_upadd_f32_ex(const float* grid, const float* guide, const float* input,
                                              const float* coarse, int Hc, int Wc, float* out, int B,
                                              int H, int W, int GH, int GW, int GD, int Cin, int Cout,
                                              int has_offset, const float* guide_conv1,
                                              const float* guide_conv2, int n_feats, unsigned flags,
                                              void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_guide_flags(flags)) return rc;
  if (Cin <= 0 || Cout <= 0) return fail(HDRNET_INVALID_ARGUMENT, "bad channel counts");
  if (Hc <= 0 || Wc <= 0) return fail(HDRNET_INVALID_ARGUMENT, "bad coarse extents (%d x %d)", Hc, Wc);
  if ((guide != nullptr) == (guide_conv1 != nullptr))
    return fail(HDRNET_INVALID_ARGUMENT, "give either a guide map or the guide network, not both / neither");
  if (guide_conv1 && (!guide_conv2 || n_feats <= 0 || n_feats > 4096))
    return fail(HDRNET_INVALID_ARGUMENT, "guide network needs conv1, conv2 and 0 < n_feats <= 4096");
  if ((long long)B * H * W == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!grid || !input || !out || !coarse) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (int rc = check_guide_prescaled(flags, Cin, guide_conv1, guide_conv2)) return rc;
  ApplyArgs a{grid, guide, input, out, B, H, W, GH, GW, GD, Cin, Cout,
              Cin + (has_offset ? 1 : 0), has_offset != 0, 0};
  a.fast_sigmoid = (flags & HDRNET_GUIDE_SIGMOID_FAST) != 0;
  a.guide_prescaled = (flags & HDRNET_GUIDE_RELU_PRESCALED) != 0;
  if (!apply_fwd_upadd_supported(a, coarse, guide_conv1 != nullptr))
    return fail(HDRNET_INVALID_ARGUMENT,
                "slice-apply + up-add needs Cin = Cout = 3 with offset, W %% 4 == 0 and 16-B aligned "
                "buffers; compose hdrnet_bilateral_slice_apply_f32 and hdrnet_resize_bilinear_f32 instead");
  const char* name = "";
  const int rc = check_launch(launch_apply_fwd_upadd(a, coarse, Hc, Wc, guide_conv1, guide_conv2, n_feats,
                                                    static_cast<hipStream_t>(stream), &name),
                              "BilateralSliceApplyUpAdd");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

int hdrnet_resize_bilinear_f32(const float* in, float* out, int B, int Hin, int Win, int Hout, int Wout,
                               int C, void* stream) {
  using namespace hdrnet_amd;
  if (B < 0 || Hin <= 0 || Win <= 0 || Hout < 0 || Wout < 0 || C <= 0)
    return fail(HDRNET_INVALID_ARGUMENT, "bad extents (B=%d, in %dx%d, out %dx%d, C=%d)", B, Hin, Win, Hout,
                Wout, C);
  if ((long long)B * Hout * Wout == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!in || !out) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  const char* name = "";
  const int rc = check_launch(launch_resize_bilinear(in, out, B, Hin, Win, Hout, Wout, C,
                                                    static_cast<hipStream_t>(stream), &name),
                              "ResizeBilinear");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

size_t hdrnet_pointwise_guide_grad_workspace_bytes(long long npx, int Cin, int n_feats) {
  if (npx <= 0) return 0;
  return hdrnet_amd::guide_grad_workspace_bytes(npx, Cin, n_feats);
}

int hdrnet_pointwise_guide_grad_f32(const float* input, const float* guide, const float* dguide,
                                    const float* guide_conv1, const float* guide_conv2,
                                    float* dinput, int accumulate_dinput, float* dconv1,
                                    float* dconv2, long long npx, int Cin, int n_feats,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  using namespace hdrnet_amd;
  if (npx < 0 || Cin <= 0 || n_feats <= 0)
    return fail(HDRNET_INVALID_ARGUMENT, "bad sizes (npx=%lld, Cin=%d, n=%d)", npx, Cin, n_feats);
  if (!dconv1 || !dconv2 || !guide_conv1 || !guide_conv2)
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (npx == 0) {  // no pixels: zero parameter gradients
    hipError_t e = hipMemsetAsync(dconv1, 0, sizeof(float) * (size_t)n_feats * (Cin + 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(dconv2, 0, sizeof(float) * (size_t)(n_feats + 1), s);
    const int rc = check_launch(e, "PointwiseGuideGrad");
    if (rc == HDRNET_OK) set_kernel("noop");
    return rc;
  }
  if (!input || !guide || !dguide) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  GuideGradArgs a{input, guide, dguide, guide_conv1, guide_conv2, dinput, accumulate_dinput != 0,
                  dconv1, dconv2, npx, Cin, n_feats, workspace, workspace_bytes};
  if (!guide_grad_supported(a))
    return fail(HDRNET_INVALID_ARGUMENT,
                "guide-network gradient needs Cin in {1,3}, n_feats in {4,8,16}, 16-B aligned buffers "
                "and a workspace of hdrnet_pointwise_guide_grad_workspace_bytes()");
  const char* name = "";
  const int rc = check_launch(launch_guide_grad(a, s, &name), "PointwiseGuideGrad");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

size_t hdrnet_curves_guide_grad_workspace_bytes(long long npx, int Cin, int npts) {
  if (npx <= 0) return 0;
  return hdrnet_amd::curves_grad_workspace_bytes(npx, Cin, npts);
}

int hdrnet_curves_guide_grad_f32(const float* input, const float* dguide, const float* guide_ccm,
                                 const float* guide_shifts, const float* guide_slopes,
                                 const float* guide_mix, float* dinput, int accumulate_dinput,
                                 float* dccm, float* dshifts, float* dslopes, float* dmix, long long npx,
                                 int Cin, int npts, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  using namespace hdrnet_amd;
  if (npx < 0 || Cin <= 0 || npts <= 0)
    return fail(HDRNET_INVALID_ARGUMENT, "bad sizes (npx=%lld, Cin=%d, npts=%d)", npx, Cin, npts);
  if (!dccm || !dshifts || !dslopes || !dmix || !guide_ccm || !guide_shifts || !guide_slopes || !guide_mix)
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (npx == 0) {  // no pixels: zero parameter gradients
    hipError_t e = hipMemsetAsync(dccm, 0, sizeof(float) * (size_t)Cin * (Cin + 1), s);
    if (e == hipSuccess) e = hipMemsetAsync(dshifts, 0, sizeof(float) * (size_t)npts * Cin, s);
    if (e == hipSuccess) e = hipMemsetAsync(dslopes, 0, sizeof(float) * (size_t)npts * Cin, s);
    if (e == hipSuccess) e = hipMemsetAsync(dmix, 0, sizeof(float) * (size_t)(Cin + 1), s);
    const int rc = check_launch(e, "CurvesGuideGrad");
    if (rc == HDRNET_OK) set_kernel("noop");
    return rc;
  }
  if (!input || !dguide) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  CurvesGradArgs a{input, dguide, guide_ccm, guide_shifts, guide_slopes, guide_mix, dinput,
                   accumulate_dinput != 0, dccm, dshifts, dslopes, dmix, npx, Cin, npts, workspace,
                   workspace_bytes};
  if (!curves_grad_supported(a))
    return fail(HDRNET_INVALID_ARGUMENT,
                "curves-guide gradient needs Cin = 3, npts = 16 and a workspace of "
                "hdrnet_curves_guide_grad_workspace_bytes()");
  const char* name = "";
  const int rc = check_launch(launch_curves_grad(a, s, &name), "CurvesGuideGrad");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

size_t hdrnet_input_moments_workspace_bytes(long long npx, int Cin) {
  if (npx <= 0) return 0;
  return hdrnet_amd::input_moments_workspace_bytes(npx, Cin);
}

int hdrnet_input_moments_f32(const float* input, long long npx, int Cin, float* sums,
                             float* moments, void* workspace, size_t workspace_bytes,
                             void* stream) {
  using namespace hdrnet_amd;
  if (npx < 0 || (Cin != 1 && Cin != 3))
    return fail(HDRNET_INVALID_ARGUMENT, "input moments need npx >= 0 and Cin in {1,3} (npx=%lld, Cin=%d)",
                npx, Cin);
  if (!sums || !moments) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (npx == 0) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(float) * Cin, s);
    if (e == hipSuccess) e = hipMemsetAsync(moments, 0, sizeof(float) * Cin * Cin, s);
    const int rc = check_launch(e, "InputMoments");
    if (rc == HDRNET_OK) set_kernel("noop");
    return rc;
  }
  const size_t need = input_moments_workspace_bytes(npx, Cin);
  if (!input || ((uintptr_t)input & 15u) || !workspace || workspace_bytes < need)
    return fail(HDRNET_INVALID_ARGUMENT, "input moments need a 16-B aligned input and a workspace of "
                                         "hdrnet_input_moments_workspace_bytes()");
  const char* name = "";
  const int rc = check_launch(launch_input_moments(input, npx, Cin, sums, moments, workspace, s, &name),
                              "InputMoments");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

int hdrnet_guide_nn_prescale_f32(const float* guide_conv1, const float* guide_conv2, int n_feats, int Cin, float x_max,
                                 float* conv1_out, float* conv2_out, void* stream) {
  using namespace hdrnet_amd;
  if (Cin != 3 || n_feats <= 0 || n_feats > 4096)
    return fail(HDRNET_INVALID_ARGUMENT, "guide prescale needs Cin = 3 and 0 < n_feats <= 4096 (Cin=%d, n=%d)", Cin, n_feats);
  if (!(x_max > 0.0f) || !(x_max < 1e30f))
    return fail(HDRNET_INVALID_ARGUMENT, "guide prescale needs a finite positive x_max");
  if (!guide_conv1 || !guide_conv2 || !conv1_out || !conv2_out) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (((uintptr_t)conv1_out | (uintptr_t)conv2_out) & 15u)
    return fail(HDRNET_INVALID_ARGUMENT, "guide prescale needs 16-B aligned output arrays");
  const int rc = check_launch(launch_guide_nn_prescale(guide_conv1, guide_conv2, n_feats, x_max, conv1_out, conv2_out,
                                                       static_cast<hipStream_t>(stream)),
                              "GuideNNPrescale");
  if (rc == HDRNET_OK) set_kernel("guide_nn_prescale");
  return rc;
}

int hdrnet_guide_fold_batch_f32(const float* sums, const float* moments, long long npx, const float* w1,
                                const float* gamma, const float* beta, const float* w2, const float* b2, double eps,
                                double momentum, int Cin, int n_feats, float* conv1, float* conv2,
                                float* running_mean, float* running_var, long long* num_batches_tracked,
                                void* stream) {
  using namespace hdrnet_amd;
  if (npx <= 0 || (Cin != 1 && Cin != 3) || n_feats <= 0)
    return fail(HDRNET_INVALID_ARGUMENT, "guide fold needs npx > 0, Cin in {1,3}, n_feats > 0 (npx=%lld, Cin=%d, n=%d)",
                npx, Cin, n_feats);
  if (!sums || !moments || !w1 || !gamma || !beta || !w2 || !b2 || !conv1 || !conv2 || (!running_mean != !running_var))
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  const int rc = check_launch(launch_guide_fold_batch(sums, moments, npx, w1, gamma, beta, w2, b2, eps, momentum, Cin,
                                                      n_feats, conv1, conv2, running_mean, running_var,
                                                      num_batches_tracked, static_cast<hipStream_t>(stream)),
                              "GuideFoldBatch");
  if (rc == HDRNET_OK) set_kernel("guide_fold_batch");
  return rc;
}

int hdrnet_guide_fold_batch_grad_f32(const float* sums, const float* moments, long long npx, const float* w1,
                                     const float* gamma, const float* beta, double eps, int Cin, int n_feats,
                                     const float* dconv1, const float* dconv2, float* dw1, float* dbeta, float* dw2,
                                     float* db2, void* stream) {
  using namespace hdrnet_amd;
  if (npx <= 0 || (Cin != 1 && Cin != 3) || n_feats <= 0)
    return fail(HDRNET_INVALID_ARGUMENT, "guide fold needs npx > 0, Cin in {1,3}, n_feats > 0 (npx=%lld, Cin=%d, n=%d)",
                npx, Cin, n_feats);
  if (!sums || !moments || !w1 || !gamma || !beta || !dconv1 || !dconv2 || !dw1 || !dbeta || !dw2 || !db2)
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  const int rc = check_launch(launch_guide_fold_batch_grad(sums, moments, npx, w1, gamma, beta, eps, Cin, n_feats,
                                                           dconv1, dconv2, dw1, dbeta, dw2, db2,
                                                           static_cast<hipStream_t>(stream)),
                              "GuideFoldBatchGrad");
  if (rc == HDRNET_OK) set_kernel("guide_fold_batch_grad");
  return rc;
}

size_t hdrnet_l2_loss_workspace_bytes(long long n) { return hdrnet_amd::l2_loss_workspace_bytes(n); }

int hdrnet_l2_loss_f32(const float* prediction, const float* target, long long n, float* loss, void* workspace,
                       size_t workspace_bytes, void* stream) {
  using namespace hdrnet_amd;
  if (n <= 0) return fail(HDRNET_INVALID_ARGUMENT, "l2 loss of an empty tensor (n=%lld)", n);
  if (!prediction || !target || !loss) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if ((((uintptr_t)prediction | (uintptr_t)target) & 15u) || !workspace || workspace_bytes < l2_loss_workspace_bytes(n))
    return fail(HDRNET_INVALID_ARGUMENT, "l2 loss needs 16-B aligned tensors and a workspace of "
                                         "hdrnet_l2_loss_workspace_bytes()");
  const int rc = check_launch(launch_l2_loss(prediction, target, n, loss, workspace, static_cast<hipStream_t>(stream)),
                              "L2Loss");
  if (rc == HDRNET_OK) set_kernel("l2_loss");
  return rc;
}

int hdrnet_l2_loss_grad_f32(const float* prediction, const float* target, const float* grad_output, long long n,
                            float* dprediction, void* stream) {
  using namespace hdrnet_amd;
  if (n <= 0) return fail(HDRNET_INVALID_ARGUMENT, "l2 loss of an empty tensor (n=%lld)", n);
  if (!prediction || !target || !grad_output || !dprediction) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (((uintptr_t)prediction | (uintptr_t)target | (uintptr_t)dprediction) & 15u)
    return fail(HDRNET_INVALID_ARGUMENT, "l2 loss needs 16-B aligned tensors");
  const int rc = check_launch(launch_l2_loss_grad(prediction, target, grad_output, n, dprediction,
                                                  static_cast<hipStream_t>(stream)),
                              "L2LossGrad");
  if (rc == HDRNET_OK) set_kernel("l2_loss_grad");
  return rc;
}

size_t hdrnet_coefficients_workspace_bytes(const hdrnet_coeff_net* net, int B) {
  if (!net || B <= 0) return 0;
  return hdrnet_amd::coefficients_workspace_bytes(*net, B);
}

int hdrnet_coefficients_f32(const float* lowres, const hdrnet_coeff_net* net, float* coeffs, int B,
                            void* workspace, size_t workspace_bytes, void* stream) {
  using namespace hdrnet_amd;
  if (!net) return fail(HDRNET_INVALID_ARGUMENT, "null network description");
  if (B < 0 || B > 65535) return fail(HDRNET_INVALID_ARGUMENT, "batch out of range (B=%d, at most 65535 per call)", B);
  if (!coefficients_supported(*net))
    return fail(HDRNET_INVALID_ARGUMENT,
                "coefficient network: unsupported hyper-parameters (net_input_size=%d, spatial_bin=%d, luma_bins=%d, "
                "channel_multiplier=%d, n_out=%d, n_in=%d, n_levels=%d): sizes must be powers of two and "
                "channel_multiplier * luma_bins / 4 a power of two",
                net->net_input_size, net->spatial_bin, net->luma_bins, net->channel_multiplier, net->n_out,
                net->n_in, net->n_levels);
  if (B == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  int n_ds = 0;
  for (int v = net->net_input_size / net->spatial_bin; v > 1; v >>= 1) ++n_ds;
  bool null_param = !net->pred_w || !net->pred_b || !net->local_w[0] || !net->local_w[1] || !net->local_b[0];
  for (int i = 0; i < n_ds; ++i) null_param = null_param || !net->splat_w[i] || !net->splat_b[i];
  for (int i = 0; i < 2; ++i) null_param = null_param || !net->global_conv_w[i] || !net->global_conv_b[i];
  for (int i = 0; i < 3; ++i) null_param = null_param || !net->fc_w[i] || !net->fc_b[i];
  if (null_param) return fail(HDRNET_INVALID_ARGUMENT, "coefficient network: null parameter");
  if (!lowres || !coeffs) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  const size_t need = coefficients_workspace_bytes(*net, B);
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15u))
    return fail(HDRNET_INVALID_ARGUMENT, "coefficient network needs a 16-B aligned workspace of "
                                         "hdrnet_coefficients_workspace_bytes() = %zu bytes", need);
  const char* name = "";
  const int rc = check_launch(launch_coefficients(lowres, *net, coeffs, B, workspace,
                                                  static_cast<hipStream_t>(stream), &name),
                              "Coefficients");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

size_t hdrnet_coefficients_grad_workspace_bytes(const hdrnet_coeff_net* net, int B) {
  if (!net || B <= 0) return 0;
  return hdrnet_amd::coefficients_grad_workspace_bytes(*net, B);
}

int hdrnet_coefficients_grad_f32(const float* lowres, const hdrnet_coeff_net* net, const void* forward_workspace,
                                 const float* dcoeffs, const hdrnet_coeff_net_grads* grads, int B, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  using namespace hdrnet_amd;
  if (!net || !grads) return fail(HDRNET_INVALID_ARGUMENT, "null network description");
  const size_t need = B > 0 ? coefficients_grad_workspace_bytes(*net, B) : 0;
  if (need == 0)
    return fail(HDRNET_INVALID_ARGUMENT,
                "coefficient network gradient: unsupported (needs the forward's support, n_levels = 1, fc_layout = 1, "
                "1 <= B <= 8, 8 * cm * gd <= 256; got B=%d, n_levels=%d, fc_layout=%d)", B, net->n_levels, net->fc_layout);
  int n_ds = 0;
  for (int v = net->net_input_size / net->spatial_bin; v > 1; v >>= 1) ++n_ds;
  bool null_param = !net->pred_w || !net->pred_b || !net->local_w[0] || !net->local_w[1] || !net->local_b[0] ||
                    !grads->pred_w || !grads->pred_b || !grads->local_w[0] || !grads->local_w[1] || !grads->local_b[0];
  for (int i = 0; i < n_ds; ++i)
    null_param = null_param || !net->splat_w[i] || !net->splat_b[i] || !grads->splat_w[i] || !grads->splat_b[i];
  for (int i = 0; i < 2; ++i)
    null_param = null_param || !net->global_conv_w[i] || !net->global_conv_b[i] || !grads->global_conv_w[i] ||
                 !grads->global_conv_b[i];
  for (int i = 0; i < 3; ++i)
    null_param = null_param || !net->fc_w[i] || !net->fc_b[i] || !grads->fc_w[i] || !grads->fc_b[i];
  if (null_param) return fail(HDRNET_INVALID_ARGUMENT, "coefficient network gradient: null parameter");
  if (!lowres || !forward_workspace || !dcoeffs) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15u))
    return fail(HDRNET_INVALID_ARGUMENT, "coefficient network gradient needs a 16-B aligned workspace of "
                                         "hdrnet_coefficients_grad_workspace_bytes() = %zu bytes", need);
  const char* name = "";
  const int rc = check_launch(launch_coefficients_grad(lowres, *net, *grads, dcoeffs, B, forward_workspace, workspace,
                                                       static_cast<hipStream_t>(stream), &name),
                              "CoefficientsGrad");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

int hdrnet_bilateral_slice_apply_io(const float* grid, const float* guide, const void* input,
                                    void* out, int B, int H, int W, int GH, int GW, int GD, int Cin,
                                    int Cout, int has_offset, int input_dtype,
                                    float input_white_level, int output_dtype,
                                    const float* guide_conv1, const float* guide_conv2, int n_feats,
                                    float* guide_out, void* stream) {
  return hdrnet_bilateral_slice_apply_io_ex(grid, guide, input, out, B, H, W, GH, GW, GD, Cin, Cout, has_offset,
                                            input_dtype, input_white_level, output_dtype, guide_conv1, guide_conv2,
                                            n_feats, guide_out, 0u, stream);
}

int hdrnet_bilateral_slice_apply_io_ex(const float* grid, const float* guide, const void* input,
                                       void* out, int B, int H, int W, int GH, int GW, int GD, int Cin,
                                       int Cout, int has_offset, int input_dtype,
                                       float input_white_level, int output_dtype,
                                       const float* guide_conv1, const float* guide_conv2, int n_feats,
                                       float* guide_out, unsigned flags, void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_guide_flags(flags)) return rc;
  if (Cin <= 0 || Cout <= 0) return fail(HDRNET_INVALID_ARGUMENT, "bad channel counts");
  if (input_dtype < 0 || input_dtype > 2 || output_dtype < 0 || output_dtype > 1)
    return fail(HDRNET_INVALID_ARGUMENT, "unknown dtype code (input %d, output %d)", input_dtype,
                output_dtype);
  if (!(input_white_level > 0.0f))
    return fail(HDRNET_INVALID_ARGUMENT, "input_white_level must be positive");
  if ((long long)B * H * W == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!grid || !input || !out) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (!guide && (!guide_conv1 || !guide_conv2 || n_feats <= 0 || n_feats > 4096))
    return fail(HDRNET_INVALID_ARGUMENT, "either a guide map or the guide network must be given");
  ApplyIoArgs a{grid, guide, input, out, B, H, W, GH, GW, GD, Cin, Cout, has_offset != 0,
                input_dtype, output_dtype, input_white_level, guide_conv1, guide_conv2, n_feats,
                guide_out};
  if (int rc = check_guide_prescaled(flags, guide ? 0 : Cin, guide ? nullptr : guide_conv1, guide ? nullptr : guide_conv2)) return rc;
  a.fast_sigmoid = (flags & HDRNET_GUIDE_SIGMOID_FAST) != 0;
  a.guide_prescaled = (flags & HDRNET_GUIDE_RELU_PRESCALED) != 0;
  if (!apply_fwd_io_supported(a))
    return fail(HDRNET_INVALID_ARGUMENT,
                "the wire-format forward supports Cin = Cout = 3 with offset, W %% 4 == 0, aligned "
                "buffers; convert on the caller's side and use hdrnet_bilateral_slice_apply_f32");
  const char* name = "";
  const int rc = check_launch(launch_apply_fwd_io(a, static_cast<hipStream_t>(stream), &name),
                              "BilateralSliceApplyIO");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

int hdrnet_bilateral_slice_apply_io_curves(const float* grid, const void* input, void* out, int B, int H,
                                           int W, int GH, int GW, int GD, int Cin, int Cout,
                                           int has_offset, int input_dtype, float input_white_level,
                                           int output_dtype, const float* guide_ccm,
                                           const float* guide_shifts, const float* guide_slopes,
                                           const float* guide_mix, int npts, float* guide_out,
                                           void* stream) {
  return hdrnet_bilateral_slice_apply_io_curves_prepared(grid, input, out, B, H, W, GH, GW, GD, Cin, Cout, has_offset,
                                                         input_dtype, input_white_level, output_dtype, guide_ccm,
                                                         guide_shifts, guide_slopes, guide_mix, npts, nullptr, guide_out,
                                                         stream);
}

size_t hdrnet_curves_guide_prepared_bytes(int Cin) { return hdrnet_amd::curves_guide_prepared_bytes(Cin); }

int hdrnet_curves_guide_prepare_f32(const float* guide_shifts, const float* guide_slopes, int npts, int Cin,
                                    void* prepared, size_t prepared_bytes, int* usable, void* stream) {
  using namespace hdrnet_amd;
  const size_t need = curves_guide_prepared_bytes(Cin);
  if (need == 0 || npts <= 0 || npts > 16)
    return fail(HDRNET_INVALID_ARGUMENT, "curves prepare needs Cin = 3 and 1 .. 16 knots per channel (Cin=%d, npts=%d)", Cin,
                npts);
  if (!guide_shifts || !guide_slopes || !prepared) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  if (((uintptr_t)prepared & 15u) || prepared_bytes < need)
    return fail(HDRNET_INVALID_ARGUMENT, "curves prepare needs a 16-B aligned buffer of hdrnet_curves_guide_prepared_bytes()");
  if (!usable) return fail(HDRNET_INVALID_ARGUMENT, "curves prepare: `usable` must point to an int");
  *usable = 0;
  const int rc = check_launch(launch_curves_guide_prepare(guide_shifts, guide_slopes, npts, Cin, static_cast<float*>(prepared),
                                                          static_cast<hipStream_t>(stream)),
                              "CurvesGuidePrepare");
  if (rc != HDRNET_OK) return rc;
  set_kernel("curves_prepare");
  // a SET-UP call, once per parameter set: the table's `ok` word comes back to the host (this waits for `stream`), because
  // which forward kernel a prepared buffer selects is decided on the host
  float ok = 0.0f;
  hipError_t e = hipMemcpyAsync(&ok, static_cast<const float*>(prepared) + curves_guide_prepared_ok_offset(Cin),
                                sizeof(float), hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
  if (e != hipSuccess)
    return fail(HDRNET_RUNTIME_FAILURE, "curves prepare: reading the table's ok word back: %s", hipGetErrorString(e));
  *usable = ok != 0.0f ? 1 : 0;
  return HDRNET_OK;
}

int hdrnet_bilateral_slice_apply_io_curves_prepared(const float* grid, const void* input, void* out, int B, int H,
                                                    int W, int GH, int GW, int GD, int Cin, int Cout,
                                                    int has_offset, int input_dtype, float input_white_level,
                                                    int output_dtype, const float* guide_ccm,
                                                    const float* guide_shifts, const float* guide_slopes,
                                                    const float* guide_mix, int npts, const void* prepared,
                                                    float* guide_out, void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (Cin <= 0 || Cout <= 0) return fail(HDRNET_INVALID_ARGUMENT, "bad channel counts");
  if (input_dtype < 0 || input_dtype > 2 || output_dtype < 0 || output_dtype > 1)
    return fail(HDRNET_INVALID_ARGUMENT, "unknown dtype code (input %d, output %d)", input_dtype,
                output_dtype);
  if (!(input_white_level > 0.0f))
    return fail(HDRNET_INVALID_ARGUMENT, "input_white_level must be positive");
  if (npts <= 0 || npts > 4096) return fail(HDRNET_INVALID_ARGUMENT, "bad number of curve knots (%d)", npts);
  if ((long long)B * H * W == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!grid || !input || !out || !guide_ccm || !guide_shifts || !guide_slopes || !guide_mix)
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  ApplyIoArgs a{grid, nullptr, input, out, B, H, W, GH, GW, GD, Cin, Cout, has_offset != 0,
                input_dtype, output_dtype, input_white_level, guide_ccm, guide_mix, npts,
                guide_out, guide_shifts, guide_slopes};
  if (prepared) {
    if (((uintptr_t)prepared & 15u) || Cin != 3 || npts > 16)
      return fail(HDRNET_INVALID_ARGUMENT, "prepared curves tables need Cin = 3, npts <= 16 and the 16-B aligned buffer "
                                           "hdrnet_curves_guide_prepare_f32 wrote (and reported usable)");
    a.guide_prepared = static_cast<const float*>(prepared);
  }
  if (!apply_fwd_io_supported(a))
    return fail(HDRNET_INVALID_ARGUMENT,
                "the fused curves-guide forward supports Cin = Cout = 3 with offset, W %% 4 == 0, aligned "
                "buffers; evaluate the guide on the caller's side and use hdrnet_bilateral_slice_apply_f32");
  const char* name = "";
  const int rc = check_launch(launch_apply_fwd_io(a, static_cast<hipStream_t>(stream), &name),
                              "BilateralSliceApplyIOCurves");
  if (rc == HDRNET_OK) set_kernel(name);
  return rc;
}

size_t hdrnet_bilateral_slice_apply_grad_workspace_bytes(int B, int H, int W, int GH, int GW,
                                                         int GD, int Cin, int Cout,
                                                         int has_offset) {
  if (B <= 0 || H <= 0 || W <= 0 || GH <= 0 || GW <= 0 || GD <= 0 || Cin < 0 || Cout <= 0) return 0;
  return hdrnet_amd::apply_grid_grad_mfma_workspace(B, H, W, GH, GW, GD, Cin, Cout, has_offset != 0);
}

int hdrnet_bilateral_slice_apply_grad_f32_ex(const float* grid, const float* guide,
                                             const float* input, const float* dout,
                                             float* dgrid, float* dguide, float* dinput, int B,
                                             int H, int W, int GH, int GW, int GD, int Cin,
                                             int Cout, int has_offset, void* workspace,
                                             size_t workspace_bytes, unsigned flags,
                                             void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_flags(flags)) return rc;
  if (Cin < 0 || Cout <= 0 || Cin + (has_offset ? 1 : 0) <= 0)
    return fail(HDRNET_INVALID_ARGUMENT, "bad channel counts (Cin=%d, Cout=%d)", Cin, Cout);
  if (!dgrid && !dguide && !dinput) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Cj = Cin + (has_offset ? 1 : 0);
  const long long npix = (long long)B * H * W;
  if (npix == 0) {
    // Gradients of an empty image: dgrid is all zeros, the others are empty.
    if (dgrid && B > 0) {
      const hipError_t e =
          hipMemsetAsync(dgrid, 0, sizeof(float) * (size_t)B * GH * GW * GD * Cout * Cj, s);
      if (e != hipSuccess) return check_launch(e, "BilateralSliceApplyGrad");
    }
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!guide || !dout || (Cin > 0 && !input) || ((dguide || dinput) && !grid))
    return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  ApplyGradArgs a{grid, guide, input, dout, dgrid, dguide, dinput, B, H, W, GH, GW, GD,
                  Cin, Cout, Cj, has_offset != 0, workspace, workspace_bytes, variant(flags)};
  // All three gradients from ONE pass over the pixels when dgrid and a per-pixel VJP are both
  // wanted (the training case) and the shape has a fused specialisation.
  if (family(flags) != HDRNET_KERNEL_GENERIC && a.variant != 3 && apply_bwd_fused_supported(a)) {
    const char* name = "";
    const int rc = check_launch(launch_apply_bwd_fused(a, s, &name), "BilateralSliceApplyGrad");
    if (rc == HDRNET_OK) set_kernel(name);
    return rc;
  }
  // dguide / dinput: one fused LDS-staged pass when a specialisation exists.
  const bool pix_fast = family(flags) != HDRNET_KERNEL_GENERIC && (dguide || dinput) &&
                        apply_vjp_rows_supported(a);
  if (family(flags) == HDRNET_KERNEL_FAST && (dguide || dinput) && !pix_fast)
    return fail(HDRNET_INVALID_ARGUMENT, "no fast BilateralSliceApplyGrad variant for this shape");
  const char* pix_name = "";
  ApplyGradArgs rest = a;
  if (pix_fast) {
    const int rc = check_launch(launch_apply_vjp_rows(a, s, &pix_name), "BilateralSliceApplyGrad");
    if (rc != HDRNET_OK) return rc;
    rest.dguide = nullptr;
    rest.dinput = nullptr;
  }
  const char* gg_name = "";
  if (dgrid && family(flags) != HDRNET_KERNEL_GENERIC) {
    if (apply_grid_grad_mfma_supported(a)) {
      const int rc = check_launch(launch_apply_grid_grad_mfma(a, s, &gg_name), "BilateralSliceApplyGrad");
      if (rc != HDRNET_OK) return rc;
      rest.dgrid = nullptr;
    } else if (family(flags) == HDRNET_KERNEL_FAST) {
      return fail(HDRNET_INVALID_ARGUMENT,
                  "no fast grid-gradient variant for this shape (or workspace missing / too small)");
    }
  }
  const char* rest_name = "";
  if (rest.dgrid || rest.dguide || rest.dinput) {
    if (rest.dgrid && family(flags) == HDRNET_KERNEL_AUTO)
      warn_generic_grid_grad("BilateralSliceApplyGrad", npix, GD, Cout * Cj, workspace != nullptr);
    const int rc = check_launch(launch_apply_grad_generic(rest, s), "BilateralSliceApplyGrad");
    if (rc != HDRNET_OK) return rc;
    rest_name = "apply_grad_generic";
  }
  set_kernel(pix_name, gg_name, rest_name);
  return HDRNET_OK;
}

int hdrnet_bilateral_slice_apply_grad_f32(const float* grid, const float* guide,
                                          const float* input, const float* dout, float* dgrid,
                                          float* dguide, float* dinput, int B, int H, int W,
                                          int GH, int GW, int GD, int Cin, int Cout,
                                          int has_offset, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return hdrnet_bilateral_slice_apply_grad_f32_ex(grid, guide, input, dout, dgrid, dguide, dinput,
                                                  B, H, W, GH, GW, GD, Cin, Cout, has_offset,
                                                  workspace, workspace_bytes, HDRNET_KERNEL_AUTO,
                                                  stream);
}

int hdrnet_bilateral_slice_f32_ex(const float* grid, const float* guide, float* out, int B, int H,
                                  int W, int GH, int GW, int GD, int C, unsigned flags,
                                  void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_flags(flags)) return rc;
  if (C <= 0) return fail(HDRNET_INVALID_ARGUMENT, "grid_channels must be positive (C=%d)", C);
  if ((long long)B * H * W == 0) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!grid || !guide || !out) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  SliceArgs a{grid, guide, out, B, H, W, GH, GW, GD, C};
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool fast_ok = slice_fwd_rows_supported(a);
  if (family(flags) == HDRNET_KERNEL_FAST && !fast_ok)
    return fail(HDRNET_INVALID_ARGUMENT, "no fast BilateralSlice variant for this shape");
  if (family(flags) != HDRNET_KERNEL_GENERIC && fast_ok) {
    const char* name = "";
    const int rc = check_launch(launch_slice_fwd_rows(a, s, &name), "BilateralSlice");
    if (rc == HDRNET_OK) set_kernel(name);
    return rc;
  }
  const int rc = check_launch(launch_slice_fwd_generic(a, s), "BilateralSlice");
  if (rc == HDRNET_OK) set_kernel("slice_fwd_generic");
  return rc;
}

int hdrnet_bilateral_slice_f32(const float* grid, const float* guide, float* out, int B, int H,
                               int W, int GH, int GW, int GD, int C, void* stream) {
  return hdrnet_bilateral_slice_f32_ex(grid, guide, out, B, H, W, GH, GW, GD, C,
                                       HDRNET_KERNEL_AUTO, stream);
}

size_t hdrnet_bilateral_slice_grad_workspace_bytes(int B, int H, int W, int GH, int GW, int GD,
                                                   int C) {
  if (B <= 0 || H <= 0 || W <= 0 || GH <= 0 || GW <= 0 || GD <= 0 || C <= 0) return 0;
  return hdrnet_amd::slice_grid_grad_mfma_workspace(B, H, W, GH, GW, GD, C);
}

int hdrnet_bilateral_slice_grad_f32_ex(const float* grid, const float* guide, const float* dout,
                                       float* dgrid, float* dguide, int B, int H, int W, int GH,
                                       int GW, int GD, int C, void* workspace,
                                       size_t workspace_bytes, unsigned flags, void* stream) {
  using namespace hdrnet_amd;
  if (int rc = check_common(B, H, W, GH, GW, GD)) return rc;
  if (int rc = check_flags(flags)) return rc;
  if (C <= 0) return fail(HDRNET_INVALID_ARGUMENT, "grid_channels must be positive (C=%d)", C);
  if (!dgrid && !dguide) {
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((long long)B * H * W == 0) {
    if (dgrid && B > 0) {
      const hipError_t e = hipMemsetAsync(dgrid, 0, sizeof(float) * (size_t)B * GH * GW * GD * C, s);
      if (e != hipSuccess) return check_launch(e, "BilateralSliceGrad");
    }
    set_kernel("noop");
    g_error[0] = '\0';
    return HDRNET_OK;
  }
  if (!guide || !dout || (dguide && !grid)) return fail(HDRNET_INVALID_ARGUMENT, "null buffer");
  SliceGradArgs a{grid, guide, dout, dgrid, dguide, B, H, W, GH, GW, GD, C, workspace,
                  workspace_bytes, variant(flags)};
  if (family(flags) != HDRNET_KERNEL_GENERIC && a.variant != 3 && slice_bwd_fused_supported(a)) {
    const char* name = "";
    const int rc = check_launch(launch_slice_bwd_fused(a, s, &name), "BilateralSliceGrad");
    if (rc == HDRNET_OK) set_kernel(name);
    return rc;
  }
  const bool pix_fast =
      family(flags) != HDRNET_KERNEL_GENERIC && dguide && slice_vjp_rows_supported(a);
  if (family(flags) == HDRNET_KERNEL_FAST && dguide && !pix_fast)
    return fail(HDRNET_INVALID_ARGUMENT, "no fast BilateralSliceGrad variant for this shape");
  const char* pix_name = "";
  SliceGradArgs rest = a;
  if (pix_fast) {
    const int rc = check_launch(launch_slice_vjp_rows(a, s, &pix_name), "BilateralSliceGrad");
    if (rc != HDRNET_OK) return rc;
    rest.dguide = nullptr;
  }
  const char* gg_name = "";
  if (dgrid && family(flags) != HDRNET_KERNEL_GENERIC) {
    if (slice_grid_grad_mfma_supported(a)) {
      const int rc = check_launch(launch_slice_grid_grad_mfma(a, s, &gg_name), "BilateralSliceGrad");
      if (rc != HDRNET_OK) return rc;
      rest.dgrid = nullptr;
    } else if (family(flags) == HDRNET_KERNEL_FAST) {
      return fail(HDRNET_INVALID_ARGUMENT,
                  "no fast grid-gradient variant for this shape (or workspace missing / too small)");
    }
  }
  const char* rest_name = "";
  if (rest.dgrid || rest.dguide) {
    if (rest.dgrid && family(flags) == HDRNET_KERNEL_AUTO)
      warn_generic_grid_grad("BilateralSliceGrad", (long long)B * H * W, GD, C, workspace != nullptr);
    const int rc = check_launch(launch_slice_grad_generic(rest, s), "BilateralSliceGrad");
    if (rc != HDRNET_OK) return rc;
    rest_name = "slice_grad_generic";
  }
  set_kernel(pix_name, gg_name, rest_name);
  return HDRNET_OK;
}

int hdrnet_bilateral_slice_grad_f32(const float* grid, const float* guide, const float* dout,
                                    float* dgrid, float* dguide, int B, int H, int W, int GH,
                                    int GW, int GD, int C, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  return hdrnet_bilateral_slice_grad_f32_ex(grid, guide, dout, dgrid, dguide, B, H, W, GH, GW, GD,
                                            C, workspace, workspace_bytes, HDRNET_KERNEL_AUTO,
                                            stream);
}

}  // extern "C"
