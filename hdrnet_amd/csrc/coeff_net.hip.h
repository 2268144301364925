// Dimensions and workspace layout of the coefficient network, shared by its inference kernels (coeff_net.hip) and its
// training-side kernels (coeff_net_train.hip): the workspace of a forward pass IS what the backward pass reads.
#pragma once

#include <stddef.h>

#include "../../include/hdrnet_amd.h"

namespace hdrnet_amd {
namespace {

constexpr int kFcChunk = 16;

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

struct NetDims {
  int N, sb, gd, cm, n_ds, feat, gl, pred;  // feat = splat output channels, gl = 8*cm*gd, pred = gd*n_out*n_in
  int gside;                                // side of the global path's last conv
};

bool net_dims(const hdrnet_coeff_net& n, NetDims* d) {
  if (n.net_input_size <= 0 || n.spatial_bin <= 0 || n.luma_bins <= 0 || n.channel_multiplier <= 0) return false;
  if (n.n_out <= 0 || n.n_in <= 0 || n.n_levels <= 0 || n.n_out % n.n_levels != 0) return false;
  if (!pow2(n.net_input_size) || !pow2(n.spatial_bin) || n.spatial_bin > n.net_input_size) return false;
  if (n.net_input_size > 4096) return false;  // tile counts stay below 2^16 (umulhi divisions), grids below 2^31
  d->N = n.net_input_size; d->sb = n.spatial_bin; d->gd = n.luma_bins; d->cm = n.channel_multiplier;
  d->n_ds = 0;
  for (int v = d->N / d->sb; v > 1; v >>= 1) ++d->n_ds;
  if (d->n_ds < 1 || d->n_ds > 8) return false;
  const int base = d->cm * d->gd;  // channels of the first splat layer
  // every staged layer reads whole float4 channel groups, a power of two of them per pixel (<= 64 channels) or whole
  // 64-channel chunks
  if (base % 4 != 0 || !pow2(base / 4)) return false;
  d->feat = base << (d->n_ds - 1);
  d->gl = 8 * base;
  d->pred = d->gd * n.n_out * n.n_in;
  d->gside = (((d->sb + 1) / 2) + 1) / 2;
  if ((long long)d->gside * d->gside * d->gl > (1 << 24)) return false;
  return true;
}

// Workspace layout (floats per image): the activations of every layer + the fc partial sums.
struct NetWorkspace {
  size_t splat[8], local1, local2, g1, g2, fc1, fc2, total;
  int s1, s2;  // fc K-chunks
};

NetWorkspace net_workspace(const NetDims& d) {
  NetWorkspace w{};
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 3) & ~(size_t)3; return o; };
  int side = d.N;
  for (int i = 0; i < d.n_ds; ++i) {
    side /= 2;
    w.splat[i] = take((size_t)side * side * ((d.cm * d.gd) << i));
  }
  w.local1 = take((size_t)d.sb * d.sb * d.gl);
  w.local2 = take((size_t)d.sb * d.sb * d.gl);
  const int g1side = (d.sb + 1) / 2;
  w.g1 = take((size_t)g1side * g1side * d.gl);
  w.g2 = take((size_t)d.gside * d.gside * d.gl);
  const int K1 = d.gside * d.gside * d.gl;
  w.s1 = (K1 + kFcChunk - 1) / kFcChunk;
  w.s2 = (4 * d.gl + kFcChunk - 1) / kFcChunk;
  w.fc1 = take((size_t)w.s1 * 4 * d.gl);
  w.fc2 = take((size_t)w.s2 * 2 * d.gl);
  w.total = off;
  return w;
}


}  // namespace
}  // namespace hdrnet_amd
