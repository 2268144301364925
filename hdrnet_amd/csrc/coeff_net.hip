// The low-resolution coefficient network of HDRNet as inference kernels -- the CALLER of the hot path
// (SURVEY.md section 8f row 1): `HDRNetCurves._coefficients` (hdrnet/models.py:62-142) with the layer
// wrappers of hdrnet/layers.py:25-93, batch norm folded as hdrnet/bin/freeze_graph.py:170-184 folds the guide's.
//
//   splat:      n_ds stride-2 3x3 convs (TF padding SAME, ReLU), 3 -> cm*gd -> ... -> cm*2^(n_ds-1)*gd at sb x sb
//   global:     two stride-2 3x3 convs -> (h, w, c) flattening -> fc 32cmgd -> fc 16cmgd -> fc 8cmgd (no activation)
//   local:      3x3 conv (ReLU) -> 3x3 conv (no bias, no activation)
//   fusion:     relu(local + global broadcast over the cells)
//   prediction: 1x1 conv to gd*n_out*n_in channels, channel (j*n_out + i)*gd + z unrolled to [B][sb][sb][gd][n_out][n_in]
//               = the bilateral grid the slice-apply kernels read (models.py:134-138)
//
// Why this exists: at 4K the slice-apply takes 39-44 us and the stock-op coefficient network ~200 us of a
// graph-captured inference (67 launches of MIOpen / elementwise kernels on a 256 x 256 image: 80 MFLOP, all
// latency).  Here it is 10 launches; what a launch costs is the ~1.5 us boundary between two dependent kernels
// plus ONE round of memory latency, because every layer stages its operands before it computes:
//
//   coeff_conv  workgroup = an 8 x 8 tile of output pixels x 4 output-channel groups (one per wave), lane = pixel.
//               The input tile ((7*stride + k)^2 pixels, <= 32 channels at a time) is staged in LDS with all of a
//               thread's loads in flight at once -- the next channel chunk's loads are issued before the current
//               one is consumed -- at a pixel stride of C + 4 floats (ds_read_b128 conflict-free for stride 1,
//               2-way for stride 2).  Weights are wave-uniform: fetched one float per lane together with the tile
//               and broadcast by v_readlane_b32, one SGPR operand per v_fmac.  Two independent layers (local /
//               global path) share one launch.
//   coeff_fc    K split over workgroups (16 rows of the [in][out] matrix each: the 1-MB fc1 matrix is read by 64
//               CUs at once), the partial sums are reduced in fixed order by the CONSUMER (next fc / prediction
//               layer) while it stages its input -- deterministic, no atomics.
//
// fp32 throughout (v_fmac_f32): the summation order differs from MIOpen's, results agree to ~1e-6 relative
// (tests/test_coeff_net.py compares both with a float64 evaluation).
#include <hip/hip_runtime.h>

#include "../../include/hdrnet_amd.h"
#include "launch.hip.h"

namespace hdrnet_amd {
namespace {

typedef __attribute__((address_space(4))) const float cfloat;  // wave-uniform parameters: s_load

constexpr int kTile = 8;        // output pixels per workgroup edge
constexpr int kChunkC4 = 8;     // float4 channel groups staged at a time (32 channels)
constexpr int kMaxProblems = 2;

struct ConvProblem {
  const float* in;    // [B][Hin][Win][Cin]
  const float* w;     // [Cout][KS][KS][Cin]
  const float* bias;  // [Cout] or null
  float* out;         // [B][Hout][Wout][Cout], or the unrolled grid (unroll != 0)
  int Hin, Win, Hout, Wout, Cin, Cout;
  int stride, pad_top, pad_left, relu;
  int tiles_x, tiles;  // tiles of this problem (tiles_x per row)
  int oc_groups;       // ceil(Cout / (4 * OCT))
  // input transform of the prediction layer: x = relu(in + g[c]), g[c] = gbias[c] + sum_s gpart[b][s][c]
  const float* gpart;
  const float* gbias;
  int gS;
  // unrolled store of the prediction layer: channel o = (j*n_out + i)*gd + z -> [level][B][Hout][Wout][gd][n_out/L][n_in]
  int unroll, gd, n_out, n_in, n_levels;
  long long level_stride;
};

struct ConvBatch {
  ConvProblem p[kMaxProblems];
  int n;
};

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// KS x KS convolution, TF padding SAME, optional bias / ReLU.  CC4 = float4 channel groups per staged chunk
// (min(Cin / 4, 8)); FIRST: Cin == 3 (the RGB input), a pixel is one float4.
//
// A wave's weights -- KS * KS * 4 * CC4 floats per output channel and chunk, at most 288 -- are fetched one float
// per lane (coalesced, in flight together with the input tile) and handed to the arithmetic by v_readlane_b32: the
// weight of every v_fmac is an SGPR that no memory instruction stands behind.  (The first version read them
// through the scalar cache inside the tap loop: 36 dependent s_load round trips per wave, 12 us for a 3x3
// 64 -> 64 layer on a 16 x 16 grid, profiles/r04/coeff_net.md.)
template <int KS, int CC4, int OCT, bool FIRST>
__global__ __launch_bounds__(256) void coeff_conv(const ConvBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  constexpr int KK = KS * KS;
  constexpr int CC = FIRST ? 3 : 4 * CC4;      // weight floats per tap and chunk
  constexpr int NW = KK * CC;                  // ... per output channel and chunk
  constexpr int NWR = (NW + 63) / 64;          // registers holding them, one float per lane
  constexpr int kShift = CC4 == 1 ? 0 : CC4 == 2 ? 1 : CC4 == 4 ? 2 : 3;
  static_assert(CC4 == 1 || CC4 == 2 || CC4 == 4 || CC4 == 8, "power-of-two chunks");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // which problem this workgroup belongs to (uniform)
  int tile = blockIdx.x, pi = 0;
  if (batch.n > 1 && tile >= batch.p[0].tiles) {
    tile -= batch.p[0].tiles;
    pi = 1;
  }
  const ConvProblem& p = batch.p[pi];
  if ((int)blockIdx.y >= p.oc_groups) return;  // the launch grid is the larger of the two problems'
  const int b = blockIdx.z;
  const int S = p.stride;
  const int TI = (kTile - 1) * S + KS;  // input tile edge
  const int tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
  const int oy0 = tyi * kTile, ox0 = txi * kTile;
  const int iy0 = oy0 * S - p.pad_top, ix0 = ox0 * S - p.pad_left;
  const int Cin = p.Cin;
  const int nchunks = FIRST ? 1 : (Cin >> 2) >> kShift;
  constexpr int pstride = FIRST ? 1 : CC4 + 1;  // float4 per staged pixel (+ 1: bank spread)
  const int npix = TI * TI;
  const int nstage = npix << kShift;  // float4 per chunk
  const float* in_b = p.in + (size_t)b * p.Hin * p.Win * Cin;
  float* gl = reinterpret_cast<float*>(lds4 + npix * pstride);  // the transform's g[Cin] behind the tile
  float* red = gl;                                             // ... and its reduction scratch behind that

  if (!FIRST && p.gpart) {  // uniform: g[c] = gbias[c] + sum_s gpart[b][s][c], the partial sums read by all threads at once
    red = gl + Cin;
    constexpr int R = 8;  // reducers per channel
    for (int c0 = 0; c0 < Cin; c0 += 256 / R) {
      const int c = c0 + (tid / R), r = tid % R;
      float g = 0.0f;
      if (c < Cin) {
        const float* gp = p.gpart + (size_t)b * p.gS * Cin + c;
#pragma unroll 4
        for (int s2 = r; s2 < p.gS; s2 += R) g += gp[(size_t)s2 * Cin];
      }
      red[tid] = g;
      __syncthreads();
      if (r == 0 && c < Cin) {
        float acc = p.gbias ? p.gbias[c] : 0.0f;
#pragma unroll
        for (int q = 0; q < R; ++q) acc += red[tid + q];
        gl[c] = acc;
      }
      __syncthreads();
    }
  }

  const int ly = lane >> 3, lx = lane & 7;
  const int ocw = __builtin_amdgcn_readfirstlane((blockIdx.y * 4 + wave) * OCT);
  const bool wave_on = ocw < p.Cout;  // uniform

  // (17 * 17 * 8 + 255) / 256 = 10 float4 per thread at most
  constexpr int kMaxU = ((7 * 2 + KS) * (7 * 2 + KS) * (FIRST ? 1 : CC4) + 255) / 256;
  float4 st[kMaxU];
  float wnext[OCT][NWR];
  auto fetch = [&](int chunk) {
    if (wave_on) {
#pragma unroll
      for (int t = 0; t < OCT; ++t) {
        const float* wo = p.w + (size_t)(ocw + t) * KK * (FIRST ? 3 : Cin) + (FIRST ? 0 : chunk * CC);
#pragma unroll
        for (int r = 0; r < NWR; ++r) {
          const int i = r * 64 + lane;
          const int tap = i / CC, c = i - tap * CC;
          wnext[t][r] = i < NW ? wo[tap * (FIRST ? 3 : Cin) + c] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kMaxU; ++u) {
      const int idx = u * 256 + tid;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < nstage) {
        const int pix = idx >> kShift, c4 = idx - (pix << kShift);
        const int py = pix / TI, px = pix - py * TI;
        const int gy = iy0 + py, gx = ix0 + px;
        if ((unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win) {
          const float* src = in_b + ((size_t)gy * p.Win + gx) * Cin;
          if constexpr (FIRST) {
            v = make_float4(src[0], src[1], src[2], 0.f);
          } else {
            const int c = ((chunk << kShift) + c4) << 2;
            v = *reinterpret_cast<const float4*>(src + c);
            if (p.gpart) {
              const float4 g = *reinterpret_cast<const float4*>(gl + c);
              v = relu4(make_float4(v.x + g.x, v.y + g.y, v.z + g.z, v.w + g.w));
            }
          }
        }
      }
      st[u] = v;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < kMaxU; ++u) {
      const int idx = u * 256 + tid;
      if (idx < nstage) {
        const int pix = idx >> kShift, c4 = idx - (pix << kShift);
        lds4[pix * pstride + c4] = st[u];
      }
    }
  };

  float acc[OCT];
#pragma unroll
  for (int t = 0; t < OCT; ++t) acc[t] = (wave_on && p.bias) ? ((cfloat*)p.bias)[ocw + t] : 0.0f;

  fetch(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();  // everyone is done with the previous chunk's tile
    stash();
    float wcur[OCT][NWR];
#pragma unroll
    for (int t = 0; t < OCT; ++t) {
#pragma unroll
      for (int r = 0; r < NWR; ++r) wcur[t][r] = wnext[t][r];
    }
    __syncthreads();
    if (chunk + 1 < nchunks) fetch(chunk + 1);  // in flight under this chunk's arithmetic
    if (wave_on) {
      const float4* lrow = lds4 + ((ly * S) * TI + lx * S) * pstride;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const float4* lp = lrow + (ky * TI + kx) * pstride;
          const int tap = ky * KS + kx;
          if constexpr (FIRST) {
            const float4 x = lp[0];
#pragma unroll
            for (int t = 0; t < OCT; ++t) {
              const int i = tap * 3;
              acc[t] = __builtin_fmaf(x.x, lane_value(wcur[t][(i + 0) >> 6], (i + 0) & 63), acc[t]);
              acc[t] = __builtin_fmaf(x.y, lane_value(wcur[t][(i + 1) >> 6], (i + 1) & 63), acc[t]);
              acc[t] = __builtin_fmaf(x.z, lane_value(wcur[t][(i + 2) >> 6], (i + 2) & 63), acc[t]);
            }
          } else {
#pragma unroll
            for (int c4 = 0; c4 < CC4; ++c4) {
              const float4 x = lp[c4];
#pragma unroll
              for (int t = 0; t < OCT; ++t) {
                const int i = tap * CC + 4 * c4;
                acc[t] = __builtin_fmaf(x.x, lane_value(wcur[t][(i + 0) >> 6], (i + 0) & 63), acc[t]);
                acc[t] = __builtin_fmaf(x.y, lane_value(wcur[t][(i + 1) >> 6], (i + 1) & 63), acc[t]);
                acc[t] = __builtin_fmaf(x.z, lane_value(wcur[t][(i + 2) >> 6], (i + 2) & 63), acc[t]);
                acc[t] = __builtin_fmaf(x.w, lane_value(wcur[t][(i + 3) >> 6], (i + 3) & 63), acc[t]);
              }
            }
          }
        }
      }
    }
  }

  const int oy = oy0 + ly, ox = ox0 + lx;
  if (!wave_on || oy >= p.Hout || ox >= p.Wout) return;
#pragma unroll
  for (int t = 0; t < OCT; ++t) {
    float v = acc[t];
    if (p.relu) v = fmaxf(v, 0.0f);
    const int o = ocw + t;
    if (p.unroll) {  // uniform
      const int ji = o / p.gd, z = o - ji * p.gd;
      const int j = ji / p.n_out, i = ji - j * p.n_out;
      const int per = p.n_out / p.n_levels;
      const int lvl = i / per, il = i - lvl * per;
      const size_t cell = ((size_t)b * p.Hout + oy) * p.Wout + ox;
      p.out[(size_t)lvl * p.level_stride + ((cell * p.gd + z) * per + il) * p.n_in + j] = v;
    } else {
      p.out[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + o] = v;
    }
  }
}

struct FcParams {
  const float* xpart;  // [B][xS][K] partial sums of the input (xS = 1: the input itself)
  const float* xbias;  // [K] or null: added to the reduced input
  const float* w;      // [K][O]  (TensorFlow's fully_connected layout)
  float* ypart;        // [B][yS][O], yS = gridDim.x
  int xS, xrelu, K, O, kc;
};

// y_part[chunk][o] = sum_{k in chunk} x[k] * w[k][o],  x = act(xbias + sum_s xpart[s])
// The xS partial sums of the chunk's 16 inputs are read by all 256 threads at once (16 reducers per input, then a
// fixed-order sum through LDS): one memory round trip, where a loop over s was xS of them (fc2 behind the 64
// partial sums of fc1: 17 us -> ...).
__global__ __launch_bounds__(256) void coeff_fc(const FcParams p) {
  __shared__ float xs[16];
  __shared__ float red[256];
  const int tid = threadIdx.x, b = blockIdx.z;
  const int k0 = blockIdx.x * p.kc;  // kc == 16
  const int kn = min(p.kc, p.K - k0);
  // the weights first: 16 loads per thread in flight under the reduction
  const int o = blockIdx.y * 256 + tid;
  float wv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) wv[k] = (o < p.O && k < kn) ? p.w[(size_t)(k0 + k) * p.O + o] : 0.0f;
  {
    const int k = tid & 15, r = tid >> 4;  // 16 reducers per input
    float x = 0.0f;
    if (k < kn) {
      const float* xp = p.xpart + (size_t)b * p.xS * p.K + k0 + k;
#pragma unroll 4
      for (int s = r; s < p.xS; s += 16) x += xp[(size_t)s * p.K];
    }
    red[tid] = x;
  }
  __syncthreads();
  if (tid < 16) {
    float x = (tid < kn && p.xbias) ? p.xbias[k0 + tid] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) x += red[r * 16 + tid];
    xs[tid] = p.xrelu ? fmaxf(x, 0.0f) : x;  // inputs beyond the chunk: 0 (their weights are 0 too)
  }
  __syncthreads();
  if (o >= p.O) return;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc = __builtin_fmaf(xs[k], wv[k], acc);
  p.ypart[((size_t)b * gridDim.x + blockIdx.x) * p.O + o] = acc;
}

constexpr int kFcChunk = 16;

inline int same_pad_before(int in, int out, int k, int s) {  // tf padding='SAME'
  const int total = (out - 1) * s + k - in;
  return total > 0 ? total / 2 : 0;
}

ConvProblem conv_problem(const float* in, const float* w, const float* bias, float* out, int Hin, int Win, int Cin,
                         int Cout, int ks, int stride, bool relu, int oct) {
  ConvProblem p{};
  p.in = in; p.w = w; p.bias = bias; p.out = out;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout;
  p.Hout = (Hin + stride - 1) / stride;
  p.Wout = (Win + stride - 1) / stride;
  p.stride = stride;
  p.pad_top = same_pad_before(Hin, p.Hout, ks, stride);
  p.pad_left = same_pad_before(Win, p.Wout, ks, stride);
  p.relu = relu ? 1 : 0;
  p.tiles_x = (p.Wout + kTile - 1) / kTile;
  p.tiles = p.tiles_x * ((p.Hout + kTile - 1) / kTile);
  p.oc_groups = (Cout + 4 * oct - 1) / (4 * oct);
  p.n_levels = 1;
  return p;
}

int chunk_c4(int Cin) { return Cin / 4 < kChunkC4 ? Cin / 4 : kChunkC4; }

size_t conv_lds(int ks, int stride, int Cin, bool first) {
  const int ti = (kTile - 1) * stride + ks;
  return (size_t)ti * ti * (first ? 1 : chunk_c4(Cin) + 1) * 16 + (first ? 0 : (size_t)Cin * 4 + 256 * 4);
}

template <int KS, int CC4, int OCT, bool FIRST>
hipError_t launch_conv_t(const ConvBatch& cb, int B, hipStream_t s) {
  size_t lds = 0;
  unsigned tiles = 0, groups = 0;
  for (int i = 0; i < cb.n; ++i) {
    const size_t l = conv_lds(KS, cb.p[i].stride, cb.p[i].Cin, FIRST);
    lds = l > lds ? l : lds;
    tiles += (unsigned)cb.p[i].tiles;
    groups = (unsigned)cb.p[i].oc_groups > groups ? (unsigned)cb.p[i].oc_groups : groups;
  }
  coeff_conv<KS, CC4, OCT, FIRST><<<dim3(tiles, groups, (unsigned)B), 256, lds, s>>>(cb);
  return hipGetLastError();
}

// every problem of a batch has the same Cin
template <int KS, int OCT>
hipError_t launch_conv(const ConvBatch& cb, int B, hipStream_t s) {
  switch (chunk_c4(cb.p[0].Cin)) {
    case 1: return launch_conv_t<KS, 1, OCT, false>(cb, B, s);
    case 2: return launch_conv_t<KS, 2, OCT, false>(cb, B, s);
    case 4: return launch_conv_t<KS, 4, OCT, false>(cb, B, s);
    case 8: return launch_conv_t<KS, 8, OCT, false>(cb, B, s);
  }
  return hipErrorInvalidValue;
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

struct NetDims {
  int N, sb, gd, cm, n_ds, feat, gl, pred;  // feat = splat output channels, gl = 8*cm*gd, pred = gd*n_out*n_in
  int gside;                                // side of the global path's last conv
};

bool net_dims(const hdrnet_coeff_net& n, NetDims* d) {
  if (n.net_input_size <= 0 || n.spatial_bin <= 0 || n.luma_bins <= 0 || n.channel_multiplier <= 0) return false;
  if (n.n_out <= 0 || n.n_in <= 0 || n.n_levels <= 0 || n.n_out % n.n_levels != 0) return false;
  if (!pow2(n.net_input_size) || !pow2(n.spatial_bin) || n.spatial_bin > n.net_input_size) return false;
  d->N = n.net_input_size; d->sb = n.spatial_bin; d->gd = n.luma_bins; d->cm = n.channel_multiplier;
  d->n_ds = 0;
  for (int v = d->N / d->sb; v > 1; v >>= 1) ++d->n_ds;
  if (d->n_ds < 1 || d->n_ds > 8) return false;
  const int base = d->cm * d->gd;  // channels of the first splat layer
  // every staged layer reads whole float4 channel groups, in power-of-two chunks of at most 32 channels
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
  size_t splat[8], local1, local2, g1, g2, fc1, fc2, fc3, total;
  int s1, s2, s3;  // fc K-chunks
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
  w.s3 = (2 * d.gl + kFcChunk - 1) / kFcChunk;
  w.fc1 = take((size_t)w.s1 * 4 * d.gl);
  w.fc2 = take((size_t)w.s2 * 2 * d.gl);
  w.fc3 = take((size_t)w.s3 * d.gl);
  w.total = off;
  return w;
}

}  // namespace

size_t coefficients_workspace_bytes(const hdrnet_coeff_net& net, int B) {
  NetDims d;
  if (!net_dims(net, &d) || B <= 0) return 0;
  return net_workspace(d).total * sizeof(float) * (size_t)B;
}

bool coefficients_supported(const hdrnet_coeff_net& net) {
  NetDims d;
  return net_dims(net, &d);
}

hipError_t launch_coefficients(const float* lowres, const hdrnet_coeff_net& net, float* coeffs, int B, void* workspace,
                               hipStream_t s, const char** name) {
  NetDims d;
  if (!net_dims(net, &d)) return hipErrorInvalidValue;
  const NetWorkspace ws = net_workspace(d);
  // per-image offsets: every activation buffer holds the whole batch, image-major
  float* base = static_cast<float*>(workspace);
  auto buf = [&](size_t off_floats, size_t per_image) {
    (void)per_image;
    return base + off_floats * (size_t)B;
  };
  *name = "coeff_net";
  hipError_t e = hipSuccess;
  auto one = [&](const ConvProblem& p, int ks, bool first, int oct) -> hipError_t {
    ConvBatch cb{};
    cb.p[0] = p;
    cb.n = 1;
    if (first) return oct == 2 ? launch_conv_t<3, 1, 2, true>(cb, B, s) : launch_conv_t<3, 1, 1, true>(cb, B, s);
    if (ks == 1) return launch_conv<1, 1>(cb, B, s);
    return oct == 2 ? launch_conv<3, 2>(cb, B, s) : launch_conv<3, 1>(cb, B, s);
  };
  // ---- splat
  const float* cur = lowres;
  int side = d.N, cin = 3;
  for (int i = 0; i < d.n_ds; ++i) {
    const int cout = (d.cm * d.gd) << i;
    float* out = buf(ws.splat[i], 0);
    // two output channels per wave while the layer has more waves than the chip has slots for
    const long long waves = (long long)((side / 2 + 7) / 8) * ((side / 2 + 7) / 8) * cout * B;
    const int oct = (cout % 2 == 0 && waves >= 4096) ? 2 : 1;
    const ConvProblem p = conv_problem(cur, net.splat_w[i], net.splat_b[i], out, side, side, cin, cout, 3, 2, true, oct);
    e = one(p, 3, i == 0, oct);
    if (e != hipSuccess) return e;
    cur = out;
    side /= 2;
    cin = cout;
  }
  // ---- local conv1 (stride 1) and global conv1 (stride 2) read the splat features: one launch
  float* l1 = buf(ws.local1, 0);
  float* g1 = buf(ws.g1, 0);
  {
    ConvBatch cb{};
    cb.p[0] = conv_problem(cur, net.local_w[0], net.local_b[0], l1, d.sb, d.sb, d.feat, d.gl, 3, 1, true, 1);
    cb.p[1] = conv_problem(cur, net.global_conv_w[0], net.global_conv_b[0], g1, d.sb, d.sb, d.feat, d.gl, 3, 2, true, 1);
    cb.n = 2;
    e = launch_conv<3, 1>(cb, B, s);
    if (e != hipSuccess) return e;
  }
  // ---- local conv2 (no bias, no activation) and global conv2
  float* l2 = buf(ws.local2, 0);
  float* g2 = buf(ws.g2, 0);
  const int g1side = (d.sb + 1) / 2;
  {
    ConvBatch cb{};
    cb.p[0] = conv_problem(l1, net.local_w[1], net.local_b[1], l2, d.sb, d.sb, d.gl, d.gl, 3, 1, false, 1);
    cb.p[1] = conv_problem(g1, net.global_conv_w[1], net.global_conv_b[1], g2, g1side, g1side, d.gl, d.gl, 3, 2, true, 1);
    cb.n = 2;
    e = launch_conv<3, 1>(cb, B, s);
    if (e != hipSuccess) return e;
  }
  // ---- fully connected layers: K-split partial sums, reduced by the consumer
  const int K1 = d.gside * d.gside * d.gl;
  float* f1 = buf(ws.fc1, 0);
  float* f2 = buf(ws.fc2, 0);
  float* f3 = buf(ws.fc3, 0);
  {
    FcParams p{g2, nullptr, net.fc_w[0], f1, 1, 0, K1, 4 * d.gl, kFcChunk};
    auto fc_grid = [&](int chunks, int O) { return dim3((unsigned)chunks, (unsigned)((O + 255) / 256), (unsigned)B); };
    coeff_fc<<<fc_grid(ws.s1, 4 * d.gl), 256, 0, s>>>(p);
    FcParams q{f1, net.fc_b[0], net.fc_w[1], f2, ws.s1, 1, 4 * d.gl, 2 * d.gl, kFcChunk};
    coeff_fc<<<fc_grid(ws.s2, 2 * d.gl), 256, 0, s>>>(q);
    FcParams r{f2, net.fc_b[1], net.fc_w[2], f3, ws.s2, 1, 2 * d.gl, d.gl, kFcChunk};
    coeff_fc<<<fc_grid(ws.s3, d.gl), 256, 0, s>>>(r);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  // ---- fusion + prediction + unroll
  {
    ConvProblem p = conv_problem(l2, net.pred_w, net.pred_b, coeffs, d.sb, d.sb, d.gl, d.pred, 1, 1, false, 1);
    p.gpart = f3;
    p.gbias = net.fc_b[2];
    p.gS = ws.s3;
    p.unroll = 1;
    p.gd = d.gd;
    p.n_out = net.n_out;
    p.n_in = net.n_in;
    p.n_levels = net.n_levels;
    p.level_stride = (long long)B * d.sb * d.sb * d.gd * (net.n_out / net.n_levels) * net.n_in;
    e = one(p, 1, false, 1);
  }
  return e;
}

}  // namespace hdrnet_amd
