// TEST INFRASTRUCTURE ONLY (oracle/). Not part of the shipped product path.
//
// extern "C" harness around the reference's OWN CPU implementation, compiled
// from the sources where they lie under /root/reference (never copied):
//   /root/reference/hdrnet/ops/bilateral_slice_apply.cc  (:24-259)
//   /root/reference/hdrnet/ops/bilateral_slice.cc        (:25-168)
// Built by oracle/Makefile into oracle/_ref/libhdrnet_ref.so.  The views below
// re-create the TF(NHWC) -> nda reinterpretation of the TF op wrappers:
//   bilateral_slice_apply_op.cc:201-227 (fwd), :305-352 (grad)
//   bilateral_slice_op.cc:149-166 (fwd), :214-247 (grad)
#include "bilateral_slice.h"
#include "bilateral_slice_apply.h"

namespace {

using nda::make_array_ref;
using nda::shape_of_rank;

}  // namespace

extern "C" {

// grid [B,GH,GW,GD,Cout*Cj], guide [B,H,W], input [B,H,W,Cin], out [B,H,W,Cout]
void ref_bilateral_slice_apply(const float* grid, const float* guide,
                               const float* input, float* out, int B, int H,
                               int W, int GH, int GW, int GD, int Cin, int Cout,
                               int has_offset) {
  const int Cj = Cin + (has_offset ? 1 : 0);
  auto grid_ref = make_array_ref(grid, shape_of_rank<6>(Cj, Cout, GD, GW, GH, B));
  auto guide_ref = make_array_ref(guide, shape_of_rank<3>(W, H, B));
  auto input_ref = make_array_ref(input, shape_of_rank<4>(Cin, W, H, B));
  auto out_ref = make_array_ref(out, shape_of_rank<4>(Cout, W, H, B));
  hdrnet::BilateralSliceApply(grid_ref, guide_ref, input_ref, out_ref);
}

// Any of dgrid / dguide / dinput may be null => that VJP is skipped.
void ref_bilateral_slice_apply_grad(const float* grid, const float* guide,
                                    const float* input, const float* dout,
                                    float* dgrid, float* dguide, float* dinput,
                                    int B, int H, int W, int GH, int GW, int GD,
                                    int Cin, int Cout, int has_offset) {
  const int Cj = Cin + (has_offset ? 1 : 0);
  auto grid_ref = make_array_ref(grid, shape_of_rank<6>(Cj, Cout, GD, GW, GH, B));
  auto guide_ref = make_array_ref(guide, shape_of_rank<3>(W, H, B));
  auto input_ref = make_array_ref(input, shape_of_rank<4>(Cin, W, H, B));
  auto dout_ref = make_array_ref(dout, shape_of_rank<4>(Cout, W, H, B));
  if (dgrid) {
    auto dgrid_ref =
        make_array_ref(dgrid, shape_of_rank<6>(Cj, Cout, GD, GW, GH, B));
    hdrnet::BilateralSliceApplyGridGrad(guide_ref, input_ref, dout_ref, dgrid_ref);
  }
  if (dguide) {
    auto dguide_ref = make_array_ref(dguide, shape_of_rank<3>(W, H, B));
    hdrnet::BilateralSliceApplyGuideGrad(grid_ref, guide_ref, input_ref, dout_ref,
                                         dguide_ref);
  }
  if (dinput) {
    auto dinput_ref = make_array_ref(dinput, shape_of_rank<4>(Cin, W, H, B));
    hdrnet::BilateralSliceApplyInputGrad(grid_ref, guide_ref, dout_ref, dinput_ref);
  }
}

// grid [B,GH,GW,GD,C], guide [B,H,W], out [B,H,W,C]
void ref_bilateral_slice(const float* grid, const float* guide, float* out, int B,
                         int H, int W, int GH, int GW, int GD, int C) {
  auto grid_ref = make_array_ref(grid, shape_of_rank<5>(C, GD, GW, GH, B));
  auto guide_ref = make_array_ref(guide, shape_of_rank<3>(W, H, B));
  auto out_ref = make_array_ref(out, shape_of_rank<4>(C, W, H, B));
  hdrnet::BilateralSlice(grid_ref, guide_ref, out_ref);
}

void ref_bilateral_slice_grad(const float* grid, const float* guide,
                              const float* dout, float* dgrid, float* dguide,
                              int B, int H, int W, int GH, int GW, int GD, int C) {
  auto grid_ref = make_array_ref(grid, shape_of_rank<5>(C, GD, GW, GH, B));
  auto guide_ref = make_array_ref(guide, shape_of_rank<3>(W, H, B));
  auto dout_ref = make_array_ref(dout, shape_of_rank<4>(C, W, H, B));
  if (dgrid) {
    auto dgrid_ref = make_array_ref(dgrid, shape_of_rank<5>(C, GD, GW, GH, B));
    hdrnet::BilateralSliceGridGrad(guide_ref, dout_ref, dgrid_ref);
  }
  if (dguide) {
    auto dguide_ref = make_array_ref(dguide, shape_of_rank<3>(W, H, B));
    hdrnet::BilateralSliceGuideGrad(grid_ref, guide_ref, dout_ref, dguide_ref);
  }
}

}  // extern "C"
