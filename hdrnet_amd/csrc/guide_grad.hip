// Training side of the point-wise guide network (HDRNetPointwiseNNGuide._guide,
// hdrnet/models.py:203-210: conv 1x1 Cin -> n with batch norm + ReLU, conv 1x1 n -> 1, sigmoid;
// the conv / batch-norm wrappers are hdrnet/layers.py:23-58).  The forward is fused into the
// slice-apply kernel (apply_fwd_rows.hip, GUIDE_NN) on the batch-norm-folded weights
//
//   guide = sigmoid(conv2[n] + sum_k conv2[k] * relu(conv1[k][Cin] + sum_j conv1[k][j] * in_j))
//
// and this file holds what a training step needs around it:
//
//   guide_nn_grad    the VJP of that expression: given dguide (from the slice-apply VJP) it
//                    produces dconv1 [n][Cin+1], dconv2 [n+1] and adds the guide path's share to
//                    dinput.  One pass over the pixels (40 B / px incl. the dinput read-modify-
//                    write), parameter gradients accumulated in registers by persistent threads
//                    and reduced in a fixed order (deterministic; no atomics).
//   input_moments    sum_px in_j and sum_px in_i * in_j.  The conv is linear, so the batch
//                    statistics batch norm needs in training mode (mean / biased variance of the
//                    n-channel conv1 output over every pixel of the batch) follow from the Cin
//                    means and the Cin x Cin second moments of the INPUT -- the n-channel
//                    full-resolution intermediate never exists, in training either.
//
// In the un-fused graph these are ~20 full-resolution tensor passes, two of them rocBLAS gemv calls
// on an [8.3 M x 16] matrix that take 75 ms of a 95 ms training step at 4 x 1080p.
#include <hip/hip_runtime.h>

#include "launch.hip.h"

namespace hdrnet_amd {
namespace {

constexpr int kThreads = 256;
constexpr int kPx = 4;  // pixels per thread per iteration (float4 guide / dguide, CIN float4 of input)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Adds the block's per-thread accumulators acc[0..NA) in a fixed order (lanes by butterfly,
// then waves 0..3) and writes one row of partial sums per block.
template <int NA>
__device__ __forceinline__ void block_reduce_store(const float (&acc)[NA], float* partial_row) {
  __shared__ float red[kThreads / 64][NA];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const float s = wave_sum(acc[a]);
    if (lane == 0) red[wave][a] = s;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < NA; a += kThreads) {
    float s = red[0][a];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) s += red[w][a];
    partial_row[a] = s;
  }
}

// partial [nb][na] -> sum_b partial[b][a], fixed order, accumulated in double; accumulator a goes to
// the destination segment that contains it (segments are consecutive ranges of a).
struct Segments {
  float* dst[4];
  int n[4];
};

__global__ __launch_bounds__(256) void reduce_partials(const float* __restrict__ partial, int nb, int na,
                                                       Segments seg) {
  __shared__ double red[256];
  const int a = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) s += (double)partial[(size_t)b * na + a];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m >= 1; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int off = a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (off < seg.n[q]) {
        seg.dst[q][off] = (float)red[0];
        break;
      }
      off -= seg.n[q];
    }
  }
}

// ---- VJP of the folded guide network ---------------------------------------------------------
// guide (saved by the forward) gives the sigmoid's derivative without re-running the network's
// second layer: dacc = dguide * g * (1 - g).
// DIN: 0 = the input needs no gradient (a training step: the image is data) -- its 3 FMAs per feature and pixel, a
// fifth of the kernel's arithmetic, are not issued; 1 = store d input; 2 = add to what dinput holds.
template <int CIN, int NF, int DIN>
__global__ __launch_bounds__(kThreads, 2) void guide_nn_grad(
    const float* __restrict__ input, const float* __restrict__ guide, const float* __restrict__ dguide,
    const float* __restrict__ conv1, const float* __restrict__ conv2, float* __restrict__ dinput,
    float* __restrict__ partial, long long npx) {
  constexpr int CJ = CIN + 1;
  constexpr int NA = NF * CJ + NF + 1;
  float acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = 0.0f;

  const long long nquads = (npx + kPx - 1) / kPx;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long q = (long long)blockIdx.x * kThreads + threadIdx.x; q < nquads; q += stride) {
    const long long p = q * kPx;
    float g[kPx], dg[kPx], in[kPx][CIN], din[kPx][CIN];
    if (p + kPx <= npx) {
      const float4 g4 = *reinterpret_cast<const float4*>(guide + p);
      const float4 d4 = *reinterpret_cast<const float4*>(dguide + p);
      g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
      dg[0] = d4.x; dg[1] = d4.y; dg[2] = d4.z; dg[3] = d4.w;
      float4 iv[CIN];
      const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
#pragma unroll
      for (int t = 0; t < CIN; ++t) iv[t] = ip[t];
      const float* inf = reinterpret_cast<const float*>(iv);
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[k][j] = inf[k * CIN + j];
      }
    } else {  // ragged tail: pixels past the end contribute nothing
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        const bool ok = p + k < npx;
        g[k] = ok ? guide[p + k] : 0.0f;
        dg[k] = ok ? dguide[p + k] : 0.0f;
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[k][j] = ok ? input[(p + k) * CIN + j] : 0.0f;
      }
    }
    float da[kPx];
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
      da[k] = dg[k] * g[k] * (1.0f - g[k]);  // d sigmoid
      acc[NF * CJ + NF] += da[k];            // d conv2 bias
#pragma unroll
      for (int j = 0; j < CIN; ++j) din[k][j] = 0.0f;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float w[CJ];
#pragma unroll
      for (int j = 0; j < CJ; ++j) w[j] = conv1[f * CJ + j];  // wave-uniform -> scalar loads
      [[maybe_unused]] const float w2 = conv2[f];
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
        float h = w[CIN];
#pragma unroll
        for (int j = 0; j < CIN; ++j) h = fmaf(w[j], in[k][j], h);
        if constexpr (DIN == 0) {
          // parameters only (a training step): with md = [h > 0] da the three sums are sum md x_j, sum md and sum md h;
          // the factor w2 of d conv1 leaves the loop (applied once below): 10 instead of 12 instructions per feature
          // and pixel on a kernel bound by VALU issue (62 us against a 22-us memory floor at 4 x 1080p)
          const float md = (h > 0.0f) ? da[k] : 0.0f;
#pragma unroll
          for (int j = 0; j < CIN; ++j) acc[f * CJ + j] = fmaf(md, in[k][j], acc[f * CJ + j]);
          acc[f * CJ + CIN] += md;
          acc[NF * CJ + f] = fmaf(md, h, acc[NF * CJ + f]);  // d conv2[f] = sum da relu(h)
        } else {
          acc[NF * CJ + f] = fmaf(da[k], fmaxf(h, 0.0f), acc[NF * CJ + f]);  // d conv2[f]
          const float dh = (h > 0.0f) ? da[k] * w2 : 0.0f;                   // through the ReLU
#pragma unroll
          for (int j = 0; j < CIN; ++j) {
            acc[f * CJ + j] = fmaf(dh, in[k][j], acc[f * CJ + j]);  // d conv1[f][j]
            din[k][j] = fmaf(dh, w[j], din[k][j]);                  // d input_j
          }
          acc[f * CJ + CIN] += dh;  // d conv1 bias
        }
      }
    }
    if constexpr (DIN != 0) {
      constexpr bool ACCUM = DIN == 2;
      if (p + kPx <= npx) {
        float4* op = reinterpret_cast<float4*>(dinput + p * CIN);
        float4 ov[CIN];
        float* of = reinterpret_cast<float*>(ov);
        if constexpr (ACCUM) {
#pragma unroll
          for (int t = 0; t < CIN; ++t) ov[t] = op[t];
        } else {
#pragma unroll
          for (int t = 0; t < CIN * kPx; ++t) of[t] = 0.0f;
        }
#pragma unroll
        for (int k = 0; k < kPx; ++k) {
#pragma unroll
          for (int j = 0; j < CIN; ++j) of[k * CIN + j] += din[k][j];
        }
#pragma unroll
        for (int t = 0; t < CIN; ++t) op[t] = ov[t];
      } else {
#pragma unroll
        for (int k = 0; k < kPx; ++k) {
          if (p + k < npx) {
#pragma unroll
            for (int j = 0; j < CIN; ++j) {
              float* o = dinput + (p + k) * CIN + j;
              *o = (ACCUM ? *o : 0.0f) + din[k][j];
            }
          }
        }
      }
    }
  }
  if constexpr (DIN == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float w2 = conv2[f];
#pragma unroll
      for (int j = 0; j < CJ; ++j) acc[f * CJ + j] *= w2;
    }
  }
  block_reduce_store<NA>(acc, partial + (size_t)blockIdx.x * NA);
}

// ---- VJP of the curves guide (HDRNetCurves._guide, hdrnet/models.py:145-190) -------------------------
//   t_c = ccm[c][3] + sum_j ccm[c][j] in_j;  cv_c = sum_k slopes[k][c] relu(t_c - shifts[k][c]);
//   guide = clip(mix[3] + sum_c mix[c] cv_c, 0, 1)
// 112 parameter gradients (ccm 12, shifts 48, slopes 48, mix 4) are too many register accumulators
// for one pass, so the knots are split: pass FIRST handles knots [0, 8) plus ccm, mix and dinput,
// the second pass knots [8, 16).  Both recompute the (cheap) forward.  The clip passes the gradient
// where 0 <= pre-clip value <= 1 (tf.clip_by_value / torch.clamp).
constexpr int kKnots = 16, kKnotsPerPass = 8;

template <bool FIRST, bool ACCUM>
__global__ __launch_bounds__(kThreads, 2) void curves_guide_grad(
    const float* __restrict__ input, const float* __restrict__ dguide, const float* __restrict__ ccm,
    const float* __restrict__ shifts, const float* __restrict__ slopes, const float* __restrict__ mix,
    float* __restrict__ dinput, float* __restrict__ partial, long long npx) {
  constexpr int CIN = 3;
  constexpr int K0 = FIRST ? 0 : kKnotsPerPass;
  // accumulators: [dshift 8x3][dslope 8x3] (+ FIRST: [dccm 3x4][dmix 4])
  constexpr int NA = 2 * kKnotsPerPass * CIN + (FIRST ? CIN * (CIN + 1) + CIN + 1 : 0);
  constexpr int O_SL = kKnotsPerPass * CIN, O_CCM = 2 * kKnotsPerPass * CIN, O_MIX = O_CCM + CIN * (CIN + 1);
  float acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = 0.0f;

  // One pixel per thread per iteration (a 12-B and a 4-B load, dense across the wave): with the 64
  // accumulators, four pixels unrolled (as in guide_nn_grad) spill.
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long p = (long long)blockIdx.x * kThreads + threadIdx.x; p < npx; p += stride) {
    const float dgp = dguide[p];
    float in[CIN];
#pragma unroll
    for (int j = 0; j < CIN; ++j) in[j] = input[p * CIN + j];
    // forward: t, curve values, pre-clip guide
    float t[CIN], cv[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      float h = ccm[c * (CIN + 1) + CIN];
#pragma unroll
      for (int j = 0; j < CIN; ++j) h = fmaf(ccm[c * (CIN + 1) + j], in[j], h);
      t[c] = h;
      cv[c] = 0.0f;
    }
#pragma unroll
    for (int kk = 0; kk < kKnots; ++kk) {
#pragma unroll
      for (int c = 0; c < CIN; ++c)
        cv[c] = fmaf(slopes[kk * CIN + c], fmaxf(t[c] - shifts[kk * CIN + c], 0.0f), cv[c]);
    }
    float g = mix[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) g = fmaf(mix[c], cv[c], g);
    const float dgk = (g >= 0.0f && g <= 1.0f) ? dgp : 0.0f;  // through the clip
    float dt[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
      const float dcv = dgk * mix[c];
      if constexpr (FIRST) acc[O_MIX + c] = fmaf(dgk, cv[c], acc[O_MIX + c]);
      float dtc = 0.0f;
#pragma unroll
      for (int kk = 0; kk < kKnots; ++kk) {
        if (FIRST || (kk >= K0 && kk < K0 + kKnotsPerPass)) {
          const float a = t[c] - shifts[kk * CIN + c];
          const float w = (a > 0.0f) ? dcv * slopes[kk * CIN + c] : 0.0f;  // d / d t_c through knot kk
          if constexpr (FIRST) dtc += w;
          if (kk >= K0 && kk < K0 + kKnotsPerPass) {
            acc[(kk - K0) * CIN + c] -= w;                                                               // d shifts
            acc[O_SL + (kk - K0) * CIN + c] = fmaf(dcv, fmaxf(a, 0.0f), acc[O_SL + (kk - K0) * CIN + c]);  // d slopes
          }
        }
      }
      dt[c] = dtc;
    }
    if constexpr (FIRST) {
      acc[O_MIX + CIN] += dgk;
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) acc[O_CCM + c * (CIN + 1) + j] = fmaf(dt[c], in[j], acc[O_CCM + c * (CIN + 1) + j]);
        acc[O_CCM + c * (CIN + 1) + CIN] += dt[c];
      }
      if (dinput) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) {
          float v = 0.0f;
#pragma unroll
          for (int c = 0; c < CIN; ++c) v = fmaf(dt[c], ccm[c * (CIN + 1) + j], v);
          float* o = dinput + p * CIN + j;
          *o = (ACCUM ? *o : 0.0f) + v;
        }
      }
    }
  }
  block_reduce_store<NA>(acc, partial + (size_t)blockIdx.x * NA);
}

// ---- first and second moments of the input ------------------------------------------------------
// acc layout: [CIN] sums, then [CIN][CIN] products (full matrix, symmetric).
template <int CIN>
__global__ __launch_bounds__(kThreads) void input_moments(const float* __restrict__ input,
                                                          float* __restrict__ partial, long long npx) {
  constexpr int NA = CIN + CIN * CIN;
  float acc[NA];
#pragma unroll
  for (int a = 0; a < NA; ++a) acc[a] = 0.0f;
  const long long nquads = (npx + kPx - 1) / kPx;
  const long long stride = (long long)gridDim.x * kThreads;
  for (long long q = (long long)blockIdx.x * kThreads + threadIdx.x; q < nquads; q += stride) {
    const long long p = q * kPx;
    float in[kPx][CIN];
    if (p + kPx <= npx) {
      float4 iv[CIN];
      const float4* ip = reinterpret_cast<const float4*>(input + p * CIN);
#pragma unroll
      for (int t = 0; t < CIN; ++t) iv[t] = ip[t];
      const float* inf = reinterpret_cast<const float*>(iv);
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[k][j] = inf[k * CIN + j];
      }
    } else {
#pragma unroll
      for (int k = 0; k < kPx; ++k) {
#pragma unroll
        for (int j = 0; j < CIN; ++j) in[k][j] = (p + k < npx) ? input[(p + k) * CIN + j] : 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < kPx; ++k) {
#pragma unroll
      for (int i = 0; i < CIN; ++i) {
        acc[i] += in[k][i];
#pragma unroll
        for (int j = i; j < CIN; ++j) acc[CIN + i * CIN + j] = fmaf(in[k][i], in[k][j], acc[CIN + i * CIN + j]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CIN; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) acc[CIN + i * CIN + j] = acc[CIN + j * CIN + i];
  }
  block_reduce_store<NA>(acc, partial + (size_t)blockIdx.x * NA);
}

int persistent_blocks(long long npx) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  const long long want = (npx + (long long)kPx * kThreads - 1) / ((long long)kPx * kThreads);
  const long long cap = (long long)cus * 4;
  return (int)(want < cap ? (want > 0 ? want : 1) : cap);
}

bool guide_shape_ok(int Cin, int n) { return (Cin == 3 || Cin == 1) && (n == 16 || n == 8 || n == 4); }

template <int CIN, int NF>
hipError_t launch_grad_t(const GuideGradArgs& a, int nb, hipStream_t s) {
  constexpr int NA = NF * (CIN + 1) + NF + 1;
  float* partial = static_cast<float*>(a.workspace);
  if (!a.dinput)
    guide_nn_grad<CIN, NF, 0><<<nb, kThreads, 0, s>>>(a.input, a.guide, a.dguide, a.conv1, a.conv2, nullptr, partial, a.npx);
  else if (a.accumulate_dinput)
    guide_nn_grad<CIN, NF, 2><<<nb, kThreads, 0, s>>>(a.input, a.guide, a.dguide, a.conv1, a.conv2, a.dinput, partial, a.npx);
  else
    guide_nn_grad<CIN, NF, 1><<<nb, kThreads, 0, s>>>(a.input, a.guide, a.dguide, a.conv1, a.conv2, a.dinput, partial, a.npx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  reduce_partials<<<NA, 256, 0, s>>>(partial, nb, NA,
                                     Segments{{a.dconv1, a.dconv2, nullptr, nullptr}, {NF * (CIN + 1), NF + 1, 0, 0}});
  return hipGetLastError();
}

}  // namespace

size_t guide_grad_workspace_bytes(long long npx, int Cin, int n) {
  if (!guide_shape_ok(Cin, n)) return 0;
  return (size_t)persistent_blocks(npx) * (size_t)(n * (Cin + 1) + n + 1) * sizeof(float);
}

bool guide_grad_supported(const GuideGradArgs& a) {
  const size_t need = guide_grad_workspace_bytes(a.npx, a.Cin, a.n_feats);
  const uintptr_t al = (uintptr_t)a.input | (uintptr_t)a.guide | (uintptr_t)a.dguide | (uintptr_t)a.dinput;
  return need != 0 && a.workspace != nullptr && a.workspace_bytes >= need && (al & 15u) == 0;
}

hipError_t launch_guide_grad(const GuideGradArgs& a, hipStream_t s, const char** name) {
  const int nb = persistent_blocks(a.npx);
  *name = "guide_nn_grad";
#define HDRNET_CASE(CI, NFEATS) \
  if (a.Cin == CI && a.n_feats == NFEATS) return launch_grad_t<CI, NFEATS>(a, nb, s)
  HDRNET_CASE(3, 16);
  HDRNET_CASE(3, 8);
  HDRNET_CASE(3, 4);
  HDRNET_CASE(1, 16);
  HDRNET_CASE(1, 8);
  HDRNET_CASE(1, 4);
#undef HDRNET_CASE
  return hipErrorInvalidValue;
}

// Training-mode fold of the guide network's batch norm into its first layer (hdrnet/layers.py:40-58 with
// is_training=True, folded as hdrnet/bin/freeze_graph.py:170-184 folds the inference statistics): the first
// convolution is linear, so the batch statistics of its never-materialised output follow from the input's moments,
//   mean_h = w1^T mean_x,   var_h[k] = w1[:,k]^T Cov_x w1[:,k]   (biased; Cov_x = moments / N - mean_x mean_x^T)
//   inv = gamma / sqrt(var_h + eps),  conv1[k] = (w1[:,k] * inv, beta[k] - mean_h * inv),  conv2 = (w2, b2)
// and the running statistics move as tf.contrib.layers.batch_norm / nn.BatchNorm1d move them (unbiased variance).
// ~100 numbers: as torch ops this was 33 launches forward and 30 backward of a graph-captured training step
// (profiles/r04/train_step.md); here one thread per feature, float64 arithmetic, one launch each way.
struct FoldArgs {
  const float* sums;     // [Cin]
  const float* moments;  // [Cin][Cin]
  const float* w1;       // [Cin][n]
  const float* gamma;    // [n]
  const float* beta;     // [n]
  const float* w2;       // [n]
  const float* b2;       // [1]
  double npx, eps, momentum;
  int Cin, n;
};

template <int CIN>
__device__ __forceinline__ void fold_stats(const FoldArgs& a, int k, double (&mx)[CIN], double (&cov)[CIN][CIN],
                                           double (&w)[CIN], double& mean_h, double& var_raw) {
#pragma unroll
  for (int i = 0; i < CIN; ++i) mx[i] = (double)a.sums[i] / a.npx;
#pragma unroll
  for (int i = 0; i < CIN; ++i) {
#pragma unroll
    for (int j = 0; j < CIN; ++j) cov[i][j] = (double)a.moments[i * CIN + j] / a.npx - mx[i] * mx[j];
  }
  mean_h = 0.0;
#pragma unroll
  for (int i = 0; i < CIN; ++i) {
    w[i] = (double)a.w1[i * a.n + k];
    mean_h += mx[i] * w[i];
  }
  var_raw = 0.0;
#pragma unroll
  for (int i = 0; i < CIN; ++i) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < CIN; ++j) t += cov[i][j] * w[j];
    var_raw += t * w[i];
  }
}

template <int CIN>
__global__ void guide_fold_batch(const FoldArgs a, float* __restrict__ conv1, float* __restrict__ conv2,
                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                 long long* __restrict__ num_batches_tracked) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) {
    conv2[a.n] = a.b2[0];
    if (num_batches_tracked) num_batches_tracked[0] += 1;
  }
  if (k >= a.n) return;
  double mx[CIN], cov[CIN][CIN], w[CIN], mean_h, var_raw;
  fold_stats<CIN>(a, k, mx, cov, w, mean_h, var_raw);
  const double var_h = var_raw > 0.0 ? var_raw : 0.0;
  if (running_mean) {
    const double unbias = a.npx / (a.npx > 1.0 ? a.npx - 1.0 : 1.0);
    running_mean[k] = (float)((1.0 - a.momentum) * (double)running_mean[k] + a.momentum * (double)(float)mean_h);
    running_var[k] = (float)((1.0 - a.momentum) * (double)running_var[k] + a.momentum * (double)(float)(var_h * unbias));
  }
  const double inv = (double)a.gamma[k] / sqrt(var_h + a.eps);
#pragma unroll
  for (int j = 0; j < CIN; ++j) conv1[k * (CIN + 1) + j] = (float)(w[j] * inv);
  conv1[k * (CIN + 1) + CIN] = (float)((double)a.beta[k] - mean_h * inv);
  conv2[k] = a.w2[k];
}

// VJP of the fold with respect to w1 [Cin][n], beta [n], w2 [n], b2 (the moments are data, gamma is fixed).
template <int CIN>
__global__ void guide_fold_batch_grad(const FoldArgs a, const float* __restrict__ dconv1,
                                      const float* __restrict__ dconv2, float* __restrict__ dw1,
                                      float* __restrict__ dbeta, float* __restrict__ dw2, float* __restrict__ db2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) db2[0] = dconv2[a.n];
  if (k >= a.n) return;
  double mx[CIN], cov[CIN][CIN], w[CIN], mean_h, var_raw;
  fold_stats<CIN>(a, k, mx, cov, w, mean_h, var_raw);
  const double var_h = var_raw > 0.0 ? var_raw : 0.0;
  const double inv = (double)a.gamma[k] / sqrt(var_h + a.eps);
  const double db = (double)dconv1[k * (CIN + 1) + CIN];
  double dinv = -db * mean_h;
#pragma unroll
  for (int j = 0; j < CIN; ++j) dinv += (double)dconv1[k * (CIN + 1) + j] * w[j];
  const double dmean = -db * inv;
  const double dvar = var_raw > 0.0 ? dinv * (-0.5) * inv / (var_h + a.eps) : 0.0;  // clamp_min(0): no gradient below
#pragma unroll
  for (int j = 0; j < CIN; ++j) {
    double cw = 0.0;
#pragma unroll
    for (int i = 0; i < CIN; ++i) cw += cov[j][i] * w[i];
    dw1[j * a.n + k] = (float)((double)dconv1[k * (CIN + 1) + j] * inv + dmean * mx[j] + 2.0 * dvar * cw);
  }
  dbeta[k] = (float)db;
  dw2[k] = dconv2[k];
}

// The PRESCALED form of a folded guide network (rows_common.hip.h: GuideNN::prescaled; Cin = 3): feature k's first-layer
// row, reordered to {w0, b, w1, w2} and multiplied by 2^-e_k, its mixing weight multiplied by 2^e_k, with
//   2^e_k >= 2 (|b_k| + x_max sum_j |w_kj|)          (a factor 2 above the bound: the bound itself is rounded)
// -- exact scalings, so the network's value is unchanged while relu(h) becomes clamp(h 2^-e, 0, 1) for every input with
// |x_j| <= x_max.  e_k is held to [-96, 96]: weights beyond 2^79 are outside what the fused kernels promise.
__global__ void guide_nn_prescale(const float* __restrict__ conv1, const float* __restrict__ conv2, int n, float x_max,
                                  float* __restrict__ conv1_out, float* __restrict__ conv2_out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) conv2_out[n] = conv2[n];
  if (k >= n) return;
  const float w0 = conv1[4 * k + 0], w1 = conv1[4 * k + 1], w2 = conv1[4 * k + 2], b = conv1[4 * k + 3];
  const float bound = fabsf(b) + x_max * (fabsf(w0) + fabsf(w1) + fabsf(w2));
  int e = 0;
  if (bound > 0.0f && bound < __builtin_inff()) (void)frexpf(bound, &e);  // bound = f 2^e, f in [0.5, 1)
  e = min(max(e + 1, -96), 96);
  const float down = ldexpf(1.0f, -e);
  reinterpret_cast<float4*>(conv1_out)[k] = make_float4(w0 * down, b * down, w1 * down, w2 * down);
  conv2_out[k] = ldexpf(conv2[k], e);
}

hipError_t launch_guide_nn_prescale(const float* conv1, const float* conv2, int n_feats, float x_max, float* conv1_out,
                                    float* conv2_out, hipStream_t s) {
  guide_nn_prescale<<<(n_feats + 63) / 64, 64, 0, s>>>(conv1, conv2, n_feats, x_max, conv1_out, conv2_out);
  return hipGetLastError();
}

hipError_t launch_guide_fold_batch(const float* sums, const float* moments, long long npx, const float* w1,
                                   const float* gamma, const float* beta, const float* w2, const float* b2, double eps,
                                   double momentum, int Cin, int n, float* conv1, float* conv2, float* running_mean,
                                   float* running_var, long long* num_batches_tracked, hipStream_t s) {
  const FoldArgs a{sums, moments, w1, gamma, beta, w2, b2, (double)npx, eps, momentum, Cin, n};
  const int nb = (n + 63) / 64;
  if (Cin == 3) guide_fold_batch<3><<<nb, 64, 0, s>>>(a, conv1, conv2, running_mean, running_var, num_batches_tracked);
  else if (Cin == 1) guide_fold_batch<1><<<nb, 64, 0, s>>>(a, conv1, conv2, running_mean, running_var, num_batches_tracked);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_guide_fold_batch_grad(const float* sums, const float* moments, long long npx, const float* w1,
                                        const float* gamma, const float* beta, double eps, int Cin, int n,
                                        const float* dconv1, const float* dconv2, float* dw1, float* dbeta, float* dw2,
                                        float* db2, hipStream_t s) {
  const FoldArgs a{sums, moments, w1, gamma, beta, nullptr, nullptr, (double)npx, eps, 0.0, Cin, n};
  const int nb = (n + 63) / 64;
  if (Cin == 3) guide_fold_batch_grad<3><<<nb, 64, 0, s>>>(a, dconv1, dconv2, dw1, dbeta, dw2, db2);
  else if (Cin == 1) guide_fold_batch_grad<1><<<nb, 64, 0, s>>>(a, dconv1, dconv2, dw1, dbeta, dw2, db2);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

size_t input_moments_workspace_bytes(long long npx, int Cin) {
  if (Cin != 1 && Cin != 3) return 0;
  return (size_t)persistent_blocks(npx) * (size_t)(Cin + Cin * Cin) * sizeof(float);
}

hipError_t launch_input_moments(const float* input, long long npx, int Cin, float* sums, float* moments,
                                void* workspace, hipStream_t s, const char** name) {
  const int nb = persistent_blocks(npx);
  float* partial = static_cast<float*>(workspace);
  *name = "input_moments";
  if (Cin == 3) input_moments<3><<<nb, kThreads, 0, s>>>(input, partial, npx);
  else if (Cin == 1) input_moments<1><<<nb, kThreads, 0, s>>>(input, partial, npx);
  else return hipErrorInvalidValue;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  reduce_partials<<<Cin + Cin * Cin, 256, 0, s>>>(partial, nb, Cin + Cin * Cin,
                                                  Segments{{sums, moments, nullptr, nullptr}, {Cin, Cin * Cin, 0, 0}});
  return hipGetLastError();
}

size_t curves_grad_workspace_bytes(long long npx, int Cin, int npts) {
  if (Cin != 3 || npts != 16) return 0;
  // two passes: 64 and 48 accumulators per workgroup
  return (size_t)persistent_blocks(npx) * (size_t)(64 + 48) * sizeof(float);
}

bool curves_grad_supported(const CurvesGradArgs& a) {
  const size_t need = curves_grad_workspace_bytes(a.npx, a.Cin, a.npts);
  const uintptr_t al = (uintptr_t)a.input | (uintptr_t)a.dguide | (uintptr_t)a.dinput;
  return need != 0 && a.workspace != nullptr && a.workspace_bytes >= need && (al & 3u) == 0;
}

hipError_t launch_curves_grad(const CurvesGradArgs& a, hipStream_t s, const char** name) {
  const int nb = persistent_blocks(a.npx);
  float* pa = static_cast<float*>(a.workspace);
  float* pb = pa + (size_t)nb * 64;
  *name = "curves_guide_grad";
  if (a.accumulate_dinput)
    curves_guide_grad<true, true><<<nb, kThreads, 0, s>>>(a.input, a.dguide, a.ccm, a.shifts, a.slopes, a.mix,
                                                           a.dinput, pa, a.npx);
  else
    curves_guide_grad<true, false><<<nb, kThreads, 0, s>>>(a.input, a.dguide, a.ccm, a.shifts, a.slopes, a.mix,
                                                            a.dinput, pa, a.npx);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  curves_guide_grad<false, false><<<nb, kThreads, 0, s>>>(a.input, a.dguide, a.ccm, a.shifts, a.slopes, a.mix,
                                                           nullptr, pb, a.npx);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  // pass A: [dshift k 0..7][dslope k 0..7][dccm 12][dmix 4]; pass B: [dshift k 8..15][dslope k 8..15]
  reduce_partials<<<64, 256, 0, s>>>(pa, nb, 64, Segments{{a.dshifts, a.dslopes, a.dccm, a.dmix}, {24, 24, 12, 4}});
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  reduce_partials<<<48, 256, 0, s>>>(pb, nb, 48, Segments{{a.dshifts + 24, a.dslopes + 24, nullptr, nullptr}, {24, 24, 0, 0}});
  return hipGetLastError();
}

}  // namespace hdrnet_amd
