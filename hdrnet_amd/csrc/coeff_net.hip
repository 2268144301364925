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
// latency).  Here it is 9 launches.  The work is latency, not arithmetic or bytes: a launch costs the 1.3-1.9 us
// boundary between two dependent kernels, ~0.6 us until the kernel arguments have arrived, and then one memory
// round trip per DEPENDENT load (profiles/r04/coeff_net.md has the per-workgroup timelines) -- so every kernel
// issues all of its loads before it waits for any, and the arithmetic runs where it is shortest:
//
//   coeff_conv_first   the RGB layer (K = 27) on the VALU: workgroup = 8 x 8 output pixels x 4 output channels (one
//                      per wave), lane = pixel, the wave's 27 weights held one per lane and broadcast by v_readlane.
//   coeff_conv_mfma    every other convolution as an implicit GEMM on the matrix cores, exact fp32
//                      (v_mfma_f32_16x16x4_f32): workgroup = 4 x 4 output pixels (the tile's 16 rows) x 16 output
//                      channels (its 16 columns), the four waves split K = taps x channels and their partial tiles
//                      are summed through LDS in fixed order.  The input tile is staged in LDS; a lane's A operand
//                      is one ds_read_b128 (4 channels of its pixel) and its B operand one 16-byte global load (the
//                      same 4 channels of its output channel's filter, [Cout][kh][kw][Cin] layout) per 16 channels
//                      and tap -- the K order inside a group of 16 channels is permuted (k = 4 q + e over the
//                      four MFMAs e) so that both operands are contiguous.  All of a wave's B loads (<= 9 float4)
//                      are in flight together with the tile.  Two independent layers (local / global path) share
//                      one launch.
//   coeff_fc           fc1, fc2: K split over workgroups (16 rows of the [in][out] matrix each: the 1-MB fc1 matrix is
//                      read by 64 CUs at once); the partial sums are reduced in fixed order by the consumer.  fc3
//                      (32 KB of weights) is evaluated by every workgroup of the prediction layer for itself, its
//                      weights requested before anything else: a launch less.  (fc2 + fc3 in ONE 1024-thread
//                      workgroup per image measured 9 us: a single CU pulls its 224 KB at ~30 GB/s.)
//
// fp32 throughout: the summation order differs from MIOpen's, results agree to ~1e-6 relative
// (tests/test_coeff_net.py compares both with a float64 evaluation).  Deterministic: no atomics.
#include <hip/hip_runtime.h>

#include "../../include/hdrnet_amd.h"
#include "coeff_net.hip.h"
#include "launch.hip.h"

namespace hdrnet_amd {
namespace {

typedef __attribute__((address_space(4))) const float cfloat;  // wave-uniform parameters: s_load
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kTile = 8;        // coeff_conv_first: output pixels per workgroup edge
constexpr int kMTile = 4;       // coeff_conv_mfma: 4 x 4 output pixels = the 16 rows of one MFMA tile
constexpr int kChunkCh = 64;    // coeff_conv_mfma: input channels staged at a time
constexpr int kMaxProblems = 2;

struct ConvGeom {  // one convolution layer; read in ONE batch of wide scalar loads at the top of the kernels
  const float* in;    // [B][Hin][Win][Cin]
  const float* w;     // [Cout][KS][KS][Cin]
  const float* bias;  // [Cout] or null
  float* out;         // [B][Hout][Wout][Cout], or the unrolled grid (coeff_conv_mfma<1, true>)
  int Hin, Win, Hout, Wout, Cin, Cout;
  int stride, pad_top, pad_left, relu;
  int tiles_x, tiles;   // tiles of this problem (tiles_x per row)
  int oc_groups;        // workgroups along the output channels
  unsigned ti_mul;      // ceil(2^32 / input tile edge):  pix / TI  == umulhi(pix, ti_mul)   (pix < 2^16)
  unsigned tx_mul;      // ceil(2^32 / tiles_x):          tile / tiles_x == umulhi(tile, tx_mul)
  int c4shift, nchunks; // coeff_conv_mfma: float4 per staged pixel = 1 << c4shift; 64-channel chunks
  int pad_;
};

// The prediction layer (hdrnet/models.py:103-138): input transform x = relu(in + g[c]) with the global features g,
// whose last fully connected layer every workgroup evaluates for itself,
//   xg[k] = relu(gx_bias[k] + sum_s gx_part[b][s][k]), k < gK  (the previous fc layer's K-split partial sums)
//   g[c]  = gb[c] + sum_k xg[k] * gw[k][c]
// and the unrolled store: channel o = (j*n_out + i)*gd + z -> [level][B][Hout][Wout][gd][n_out / L][n_in].
struct PredExtra {
  const float* gx_part;
  const float* gx_bias;
  const float* gw;
  const float* gb;
  int gxS, gK;
  int gd, n_out, n_in, n_levels;
  long long level_stride;
  int gw_oi;  // gw is [out][in] (a torch Linear weight) instead of TensorFlow's [in][out]
};

constexpr int kStepRow = 12;  // per wave: 9 steps, their count, 2 pad

struct ConvBatch {
  ConvGeom g[kMaxProblems];
  // coeff_conv_mfma, per problem and wave: (LDS float offset of the tap's pixel + 16-channel group) | (weight float
  // offset) << 16 for each of the wave's (tap, group) steps, then their count
  unsigned steps[kMaxProblems][4][kStepRow];
  PredExtra x;
  int n;
  long long* trace;  // tools build: 8 wall-clock stamps (10-ns ticks) per workgroup, or null
};

#ifdef HDRNET_TOOLS_BUILD
#define COEFF_STAMP(slot)                                                                        \
  do {                                                                                           \
    if (batch.trace && tid == 0)                                                                 \
      batch.trace[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (slot)] = \
          (long long)wall_clock64();                                                             \
  } while (0)
#else
#define COEFF_STAMP(slot) \
  do {                    \
  } while (0)
#endif

__device__ __forceinline__ float4 relu4(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}

__device__ __forceinline__ float lane_value(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// The first splat layer: 3x3, stride 2, Cin = 3 (a staged pixel is one float4), TF padding SAME, bias, ReLU.
template <int OCT>
__global__ __launch_bounds__(256) void coeff_conv_first(const ConvBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float4 lds4[];
  constexpr int KS = 3, KK = 9, NW = 27;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvGeom p = batch.g[0];
  COEFF_STAMP(0);
  const int b = blockIdx.z;
  const int S = p.stride;
  const int TI = (kTile - 1) * S + KS;  // input tile edge
  const int tile = blockIdx.x;
  const int tyi = (int)__umulhi((unsigned)tile, p.tx_mul), txi = tile - tyi * p.tiles_x;
  const int oy0 = tyi * kTile, ox0 = txi * kTile;
  const int iy0 = oy0 * S - p.pad_top, ix0 = ox0 * S - p.pad_left;
  const int npix = TI * TI;
  const float* in_b = p.in + (size_t)b * p.Hin * p.Win * 3;
  const int ocw = __builtin_amdgcn_readfirstlane((blockIdx.y * 4 + wave) * OCT);
  const bool wave_on = ocw < p.Cout;  // uniform
  // the wave's weights, one float per lane
  float wreg[OCT];
#pragma unroll
  for (int t = 0; t < OCT; ++t) wreg[t] = (wave_on && lane < NW) ? p.w[(size_t)(ocw + t) * NW + lane] : 0.0f;
  constexpr int kMaxU = (17 * 17 + 255) / 256;
  float4 st[kMaxU];
#pragma unroll
  for (int u = 0; u < kMaxU; ++u) {
    const int pix = u * 256 + tid;
    const int py = (int)__umulhi((unsigned)pix, p.ti_mul), px = pix - py * TI;
    const int gy = iy0 + py, gx = ix0 + px;
    const int gyc = min(max(gy, 0), p.Hin - 1), gxc = min(max(gx, 0), p.Win - 1);
    const float* src = in_b + ((size_t)gyc * p.Win + gxc) * 3;
    const float4 v = make_float4(src[0], src[1], src[2], 0.f);
    st[u] = (gy == gyc && gx == gxc) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < kMaxU; ++u) {
    const int pix = u * 256 + tid;
    if (pix < npix) lds4[pix] = st[u];
  }
  __syncthreads();
  COEFF_STAMP(2);
  const int ly = lane >> 3, lx = lane & 7;
  float acc[OCT];
#pragma unroll
  for (int t = 0; t < OCT; ++t) acc[t] = (wave_on && p.bias) ? ((cfloat*)p.bias)[ocw + t] : 0.0f;
  if (wave_on) {
    const float4* lrow = lds4 + (ly * S) * TI + lx * S;
#pragma unroll
    for (int tap = 0; tap < KK; ++tap) {
      const float4 x = lrow[(tap / KS) * TI + (tap % KS)];
#pragma unroll
      for (int t = 0; t < OCT; ++t) {
        acc[t] = __builtin_fmaf(x.x, lane_value(wreg[t], 3 * tap + 0), acc[t]);
        acc[t] = __builtin_fmaf(x.y, lane_value(wreg[t], 3 * tap + 1), acc[t]);
        acc[t] = __builtin_fmaf(x.z, lane_value(wreg[t], 3 * tap + 2), acc[t]);
      }
    }
  }
  COEFF_STAMP(3);
  const int oy = oy0 + ly, ox = ox0 + lx;
  if (!wave_on || oy >= p.Hout || ox >= p.Wout) return;
#pragma unroll
  for (int t = 0; t < OCT; ++t)
    p.out[(((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + ocw + t] = p.relu ? fmaxf(acc[t], 0.0f) : acc[t];
}

// KS x KS convolution (Cin a multiple of 4) as an implicit GEMM on the fp32 matrix cores; see the header.
// MFMA lane roles (v_mfma_f32_16x16x4_f32): A[i = lane & 15][kk = lane >> 4], B[kk = lane >> 4][j = lane & 15],
// D[i = 4 * (lane >> 4) + r][j = lane & 15] in register r.  Here i = output pixel of the 4 x 4 tile, j = output
// channel, and the k of MFMA e of a 16-channel group is channel 16 * group + 4 * kk + e.
//
// One wave per SIMD, nothing to switch to: what is not a memory round trip is instruction issue, ~5 cycles each,
// and every scalar load the code waits for on its own is a round trip of its own.  The first MFMA version spent
// 2.4 us of a launch in 23 integer divisions and 166 branches around its loads, the second 1.9 us in seven
// dependent rounds of kernel-argument loads (profiles/r04/coeff_net.md).  Hence: both layers' geometry is copied
// out of the kernel arguments at the top (static offsets: wide scalar loads, one round), the waves' step offsets come
// from a host-made table, divisions are umulhi by host constants, loads are clamped + selected instead of branched
// around.  PRED: the prediction layer (transform of the input, unrolled store).
template <int KS, bool PRED>
__global__ __launch_bounds__(256) void coeff_conv_mfma(const ConvBatch batch) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int KK = KS * KS;
  constexpr int kMaxSteps = KK;  // (tap, 16-channel group) steps per wave and chunk: KK * (64 / 16) / 4 waves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // which problem this workgroup belongs to (uniform): both geometries are read, one is kept
  const ConvGeom ga = batch.g[0], gb = batch.g[PRED ? 0 : 1];
  const bool second = !PRED && batch.n > 1 && (int)blockIdx.x >= ga.tiles;
  const int pi = second ? 1 : 0;
  const int tile = second ? (int)blockIdx.x - ga.tiles : (int)blockIdx.x;
#define SEL(f) (second ? gb.f : ga.f)
  const float* const p_in = SEL(in);
  const float* const p_w = SEL(w);
  const float* const p_bias = SEL(bias);
  float* const p_out = SEL(out);
  const int Hin = SEL(Hin), Win = SEL(Win), Hout = SEL(Hout), Wout = SEL(Wout), Cin = SEL(Cin), Cout = SEL(Cout);
  const int S = SEL(stride), pad_top = SEL(pad_top), pad_left = SEL(pad_left), relu = SEL(relu);
  const int tiles_x = SEL(tiles_x), oc_groups = SEL(oc_groups);
  const unsigned ti_mul = SEL(ti_mul), tx_mul = SEL(tx_mul);
  const int c4shift = SEL(c4shift), nchunks = SEL(nchunks);
#undef SEL
  // this wave's steps (one wide scalar load)
  unsigned step[kMaxSteps];
#pragma unroll
  for (int si = 0; si < kMaxSteps; ++si) step[si] = batch.steps[pi][wave][si];
  const int nsw = (int)batch.steps[pi][wave][9];
  COEFF_STAMP(0);
  if ((int)blockIdx.y >= oc_groups) return;  // the launch grid is the larger of the two problems'
  const int b = blockIdx.z;
  const int TI = (kMTile - 1) * S + KS;  // input tile edge
  const int tyi = (int)__umulhi((unsigned)tile, tx_mul), txi = tile - tyi * tiles_x;
  const int oy0 = tyi * kMTile, ox0 = txi * kMTile;
  const int iy0 = oy0 * S - pad_top, ix0 = ox0 * S - pad_left;
  const int cch = 4 << c4shift;         // channels per staged chunk
  const int PS = cch + 4;               // floats per staged pixel (+ 4: bank spread)
  const int npix = TI * TI;
  const float* in_b = p_in + (size_t)b * Hin * Win * Cin;
  float* gl = lds + npix * PS;          // the transform's g[Cin] behind the tile
  float* red = gl + ((Cin + 3) & ~3);   // [4 waves][4][64] partial tiles behind that (the transform: its scratch)
  const int q = lane >> 4, j = lane & 15;
  const int n0 = blockIdx.y * 16;
  const bool qvalid = 4 * q < cch;      // chunks narrower than 16 channels: the upper k rows are zero
  const bool bvalid = qvalid && n0 + j < Cout;

  // (9 * 9 pixels * 16 float4 + 255) / 256 = 6 float4 per thread at most (KS = 1: 4 * 4 * 16 / 256 = 1):
  // thread = (channel group tid & (c4n - 1), pixel (tid >> c4shift) + u * (256 >> c4shift))
  constexpr int kMaxU = (((kMTile - 1) * 2 + KS) * ((kMTile - 1) * 2 + KS) * (kChunkCh / 4) + 255) / 256;
  const int c4 = tid & ((1 << c4shift) - 1), pix0 = tid >> c4shift, pstep = 256 >> c4shift;
  const int nU = ((npix << c4shift) + 255) >> 8;  // uniform: iterations with any work
  float4 st[kMaxU];
  unsigned okbits = 0;  // bit u: this thread's pixel u lies inside the image (else the SAME padding's zero)
  // fetch = address arithmetic + loads, nothing that waits for a load (the padding select and the prediction layer's
  // transform happen when the tile is written to LDS)
  auto fetch_tile = [&](int chunk) {
    const float* src = in_b + chunk * kChunkCh + 4 * c4;
    okbits = 0;
#pragma unroll
    for (int u = 0; u < kMaxU; ++u) {
      if (u < nU) {  // uniform
        const int pix = pix0 + u * pstep;
        const int py = (int)__umulhi((unsigned)pix, ti_mul), px = pix - py * TI;
        const int gy = iy0 + py, gx = ix0 + px;
        const int gyc = min(max(gy, 0), Hin - 1), gxc = min(max(gx, 0), Win - 1);
        st[u] = *reinterpret_cast<const float4*>(src + ((size_t)gyc * Win + gxc) * Cin);
        okbits |= (gy == gyc && gx == gxc && pix < npix) ? (1u << u) : 0u;
      }
    }
  };
  auto stash_tile = [&](int chunk) {
#pragma unroll
    for (int u = 0; u < kMaxU; ++u) {
      const int pix = pix0 + u * pstep;
      if (u < nU && pix < npix) {
        float4 v = st[u];
        if constexpr (PRED) {
          const float4 g = *reinterpret_cast<const float4*>(gl + chunk * kChunkCh + 4 * c4);
          v = relu4(make_float4(v.x + g.x, v.y + g.y, v.z + g.z, v.w + g.w));
        }
        if (!((okbits >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(lds + pix * PS + 4 * c4) = v;
      }
    }
  };
  const float* wl = p_w + (size_t)min(n0 + j, Cout - 1) * KK * Cin + (qvalid ? 4 * q : 0);  // this lane's filter, group q
  auto fetch_w = [&](int chunk, float4 (&dst)[kMaxSteps]) {
#pragma unroll
    for (int si = 0; si < kMaxSteps; ++si) {
      if (si < nsw) {  // uniform
        const float4 v = *reinterpret_cast<const float4*>(wl + (step[si] >> 16) + chunk * kChunkCh);
        dst[si] = bvalid ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  float4 bw[kMaxSteps];
  // Every load of the kernel is requested here, before anything waits: the tile first, then the weights (and, in
  // the prediction layer, fc3's weights and fc2's partial sums below).
  fetch_tile(0);
  fetch_w(0, bw);

  if constexpr (PRED) {  // the global features: the last fully connected layer, see PredExtra
    const PredExtra x = batch.x;
    const int gK = x.gK;
    // Cin and gK are powers of two (net_dims): thread = (output channel c, K part kp) for fc3, (input k, reducer r)
    // for the partial sums
    const int cshift = 31 - __builtin_clz((unsigned)Cin), kshift = 31 - __builtin_clz((unsigned)gK);
    const int parts = Cin < 256 ? 256 >> cshift : 1;  // K parts per output channel
    const int c = tid & (Cin - 1), kp = tid >> cshift;
    const int nk = gK / parts;                         // <= 32 for Cin <= 64
    float w3r[32];
    const bool fast = nk <= 32 && Cin <= 256;
    if (fast) {
#pragma unroll
      for (int i2 = 0; i2 < 32; ++i2)
        w3r[i2] = i2 < nk ? x.gw[x.gw_oi ? (size_t)c * gK + kp * nk + i2 : (size_t)(kp * nk + i2) * Cin + c] : 0.0f;
    }
    float* xg = red;             // [gK]
    float* scratch = red + 512;  // [256]
    if (gK <= 256) {
      const int R = 256 >> kshift, k = tid & (gK - 1), r = tid >> kshift;
      float xs = 0.0f;
      const float* xp = x.gx_part + (size_t)b * x.gxS * gK + k;
#pragma unroll 8
      for (int s2 = r; s2 < x.gxS; s2 += R) xs += xp[(size_t)s2 * gK];
      scratch[tid] = xs;
      __syncthreads();
      if (tid < gK) {
        float x2 = x.gx_bias[tid];
        for (int r2 = 0; r2 < R; ++r2) x2 += scratch[(r2 << kshift) + tid];
        xg[tid] = fmaxf(x2, 0.0f);
      }
    } else {
      for (int k = tid; k < gK; k += 256) {
        float xs = x.gx_bias[k];
        const float* xp = x.gx_part + (size_t)b * x.gxS * gK + k;
        for (int s2 = 0; s2 < x.gxS; ++s2) xs += xp[(size_t)s2 * gK];
        xg[k] = fmaxf(xs, 0.0f);
      }
    }
    __syncthreads();
    if (fast) {
      float acc = 0.0f;
#pragma unroll
      for (int i2 = 0; i2 < 32; ++i2) acc = __builtin_fmaf(i2 < nk ? xg[kp * nk + i2] : 0.0f, w3r[i2], acc);
      scratch[tid] = acc;
      __syncthreads();
      if (tid < Cin) {
        float g = x.gb[tid];
        for (int r2 = 0; r2 < parts; ++r2) g += scratch[(r2 << cshift) + tid];
        gl[tid] = g;
      }
    } else {
      for (int c2 = tid; c2 < Cin; c2 += 256) {
        float g = x.gb[c2];
        for (int k = 0; k < gK; ++k) g = __builtin_fmaf(xg[k], x.gw[x.gw_oi ? (size_t)c2 * gK + k : (size_t)k * Cin + c2], g);
        gl[c2] = g;
      }
    }
    __syncthreads();
    COEFF_STAMP(1);
  }

  v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};  // two accumulators: no dependent-issue stall
  const int ti = lane & 15;
  const float* tl = lds + (((ti >> 2) * S) * TI + (ti & 3) * S) * PS + (qvalid ? 4 * q : 0);  // this lane's pixel, group q
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    if (chunk > 0) __syncthreads();  // everyone is done with the previous chunk's tile
    stash_tile(chunk);
    __syncthreads();
    if (chunk == 0) COEFF_STAMP(2);  // the first chunk has landed
    float4 bwn[kMaxSteps];
    if (chunk + 1 < nchunks) {  // in flight under this chunk's arithmetic
      fetch_w(chunk + 1, bwn);
      fetch_tile(chunk + 1);
    }
#pragma unroll
    for (int si = 0; si < kMaxSteps; ++si) {
      if (si < nsw) {  // uniform
        float4 x = *reinterpret_cast<const float4*>(tl + (step[si] & 0xffffu));
        if (!qvalid) x = make_float4(0.f, 0.f, 0.f, 0.f);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bw[si].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bw[si].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bw[si].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bw[si].w, acc1, 0, 0, 0);
      }
    }
    if (chunk + 1 < nchunks) {
#pragma unroll
      for (int si = 0; si < kMaxSteps; ++si) bw[si] = bwn[si];
    }
  }
  // the four waves' partial tiles, summed in fixed order
  const v4f acc = acc0 + acc1;
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
  __syncthreads();
  COEFF_STAMP(3);
  const int r = wave;  // thread (wave, lane) finishes element (i = 4 * (lane >> 4) + wave, j)
  float v = red[r * 64 + lane];
#pragma unroll
  for (int w = 1; w < 4; ++w) v += red[(w * 4 + r) * 64 + lane];
  const int i = 4 * q + r, o = n0 + j;
  const int oy = oy0 + (i >> 2), ox = ox0 + (i & 3);
  if (o >= Cout || oy >= Hout || ox >= Wout) return;
  if (p_bias) v += p_bias[o];
  if (relu) v = fmaxf(v, 0.0f);
  if constexpr (PRED) {
    const PredExtra x = batch.x;
    const int ji = o / x.gd, z = o - ji * x.gd;
    const int jj = ji / x.n_out, ii = ji - jj * x.n_out;
    const int per = x.n_out / x.n_levels;
    const int lvl = ii / per, il = ii - lvl * per;
    const size_t cell = ((size_t)b * Hout + oy) * Wout + ox;
    p_out[(size_t)lvl * x.level_stride + ((cell * x.gd + z) * per + il) * x.n_in + jj] = v;
  } else {
    p_out[(((size_t)b * Hout + oy) * Wout + ox) * Cout + o] = v;
  }
}

struct FcParams {
  const float* xpart;  // [B][xS][K] partial sums of the input (xS = 1: the input itself)
  const float* xbias;  // [K] or null: added to the reduced input
  const float* w;      // [K][O]  (TensorFlow's fully_connected layout)
  float* ypart;        // [B][yS][O], yS = gridDim.x
  int xS, xrelu, K, O, kc;
  long long* trace;  // tools build
  int w_oi;          // w is [O][K] (a torch Linear weight) instead of [K][O]
};

// y_part[chunk][o] = sum_{k in chunk} x[k] * w[k][o],  x = act(xbias + sum_s xpart[s])
// The xS partial sums of the chunk's 16 inputs are read by all 256 threads at once (16 reducers per input, then a
// fixed-order sum through LDS): one memory round trip, where a loop over s is xS of them (17 us behind 64 partial sums).
__global__ __launch_bounds__(256) void coeff_fc(const FcParams p) {
  __shared__ float xs[16];
  __shared__ float red[256];
  const int tid = threadIdx.x, b = blockIdx.z;
#ifdef HDRNET_TOOLS_BUILD
  const FcParams& batch = p;
#endif
  COEFF_STAMP(0);
  const int k0 = blockIdx.x * p.kc;  // kc == 16
  const int kn = min(p.kc, p.K - k0);
  // the weights first: 16 loads per thread in flight under the reduction
  const int o = blockIdx.y * 256 + tid;
  float wv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k)
    wv[k] = (o < p.O && k < kn) ? p.w[p.w_oi ? (size_t)o * p.K + k0 + k : (size_t)(k0 + k) * p.O + o] : 0.0f;
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
  COEFF_STAMP(2);
  if (o >= p.O) return;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc = __builtin_fmaf(xs[k], wv[k], acc);
  p.ypart[((size_t)b * gridDim.x + blockIdx.x) * p.O + o] = acc;
  COEFF_STAMP(3);
}


inline int same_pad_before(int in, int out, int k, int s) {  // tf padding='SAME'
  const int total = (out - 1) * s + k - in;
  return total > 0 ? total / 2 : 0;
}

unsigned magic32(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }  // umulhi(x, .) == x / d, x < 2^16

// tile: output pixels per workgroup edge; oc_per_wg: output channels per workgroup
ConvGeom conv_geom(const float* in, const float* w, const float* bias, float* out, int Hin, int Win, int Cin,
                   int Cout, int ks, int stride, bool relu, int tile, int oc_per_wg) {
  ConvGeom p{};
  p.in = in; p.w = w; p.bias = bias; p.out = out;
  p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout;
  p.Hout = (Hin + stride - 1) / stride;
  p.Wout = (Win + stride - 1) / stride;
  p.stride = stride;
  p.pad_top = same_pad_before(Hin, p.Hout, ks, stride);
  p.pad_left = same_pad_before(Win, p.Wout, ks, stride);
  p.relu = relu ? 1 : 0;
  p.tiles_x = (p.Wout + tile - 1) / tile;
  p.tiles = p.tiles_x * ((p.Hout + tile - 1) / tile);
  p.oc_groups = (Cout + oc_per_wg - 1) / oc_per_wg;
  p.ti_mul = magic32((tile - 1) * stride + ks);
  p.tx_mul = magic32(p.tiles_x);
  const int cch = Cin < kChunkCh ? Cin : kChunkCh;
  p.c4shift = 0;
  while ((4 << p.c4shift) < cch) ++p.c4shift;
  p.nchunks = Cin > kChunkCh ? Cin / kChunkCh : 1;
  return p;
}

// The step table of problem i of a batch (coeff_conv_mfma): the (tap, 16-channel group) steps of a chunk dealt to the
// four waves in contiguous runs.
void fill_steps(ConvBatch* cb, int i, int ks) {
  const ConvGeom& p = cb->g[i];
  const int ti = (kMTile - 1) * p.stride + ks;
  const int cch = 4 << p.c4shift;
  const int ps = cch + 4, g16 = (cch + 15) / 16;
  const int nsteps = ks * ks * g16;
  for (int wv = 0; wv < 4; ++wv) {
    const int s0 = (wv * nsteps) / 4, s1 = ((wv + 1) * nsteps) / 4;
    for (int k = 0; k < kStepRow; ++k) cb->steps[i][wv][k] = 0;
    for (int s = s0; s < s1; ++s) {
      const int tap = s / g16, grp = s - tap * g16;
      const int ky = tap / ks, kx = tap - ky * ks;
      const unsigned lds_off = (unsigned)((ky * ti + kx) * ps + 16 * grp);
      const unsigned w_off = (unsigned)(tap * p.Cin + 16 * grp);
      cb->steps[i][wv][s - s0] = lds_off | (w_off << 16);
    }
    cb->steps[i][wv][9] = (unsigned)(s1 - s0);
  }
}

#ifdef HDRNET_TOOLS_BUILD
long long* g_coeff_trace = nullptr;  // tools: timeline buffer, kTraceStride stamps per launch
int g_coeff_launch = 0;
constexpr size_t kTraceStride = 8 * 8192;
long long* next_trace() { return g_coeff_trace ? g_coeff_trace + kTraceStride * (size_t)(g_coeff_launch++) : nullptr; }
#else
long long* next_trace() { return nullptr; }
#endif

size_t mfma_lds(int ks, int stride, int Cin) {
  const int ti = (kMTile - 1) * stride + ks;
  const int cch = Cin < kChunkCh ? Cin : kChunkCh;
  return ((size_t)ti * ti * (cch + 4) + ((Cin + 3) & ~3) + 4 * 4 * 64) * sizeof(float);  // tile, g, partial tiles
}

template <int KS, bool PRED>
hipError_t launch_conv_mfma(const ConvBatch& cb_in, int B, hipStream_t s) {
  ConvBatch cb = cb_in;
  cb.trace = next_trace();
  size_t lds = 0;
  unsigned tiles = 0, groups = 0;
  for (int i = 0; i < cb.n; ++i) {
    fill_steps(&cb, i, KS);
    const size_t l = mfma_lds(KS, cb.g[i].stride, cb.g[i].Cin);
    lds = l > lds ? l : lds;
    tiles += (unsigned)cb.g[i].tiles;
    groups = (unsigned)cb.g[i].oc_groups > groups ? (unsigned)cb.g[i].oc_groups : groups;
  }
  coeff_conv_mfma<KS, PRED><<<dim3(tiles, groups, (unsigned)B), 256, lds, s>>>(cb);
  return hipGetLastError();
}

}  // namespace

#ifdef HDRNET_TOOLS_BUILD
void coeff_net_set_trace(long long* device_buf) { g_coeff_trace = device_buf; }
#endif

size_t coefficients_workspace_bytes(const hdrnet_coeff_net& net, int B) {
  NetDims d;
  if (!net_dims(net, &d) || B <= 0 || B > 65535) return 0;  // the batch is the launch grid's z extent
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
  // every activation buffer holds the whole batch, image-major
  float* base = static_cast<float*>(workspace);
  auto buf = [&](size_t off_floats) { return base + off_floats * (size_t)B; };
  *name = "coeff_net";
#ifdef HDRNET_TOOLS_BUILD
  g_coeff_launch = 0;
#endif
  hipError_t e = hipSuccess;
  // ---- splat
  const float* cur = lowres;
  int side = d.N, cin = 3;
  for (int i = 0; i < d.n_ds; ++i) {
    const int cout = (d.cm * d.gd) << i;
    float* out = buf(ws.splat[i]);
    ConvBatch cb{};
    cb.n = 1;
    if (i == 0) {
      // two output channels per wave while the layer has more waves than the chip has slots for
      const long long waves = (long long)((side / 2 + 7) / 8) * ((side / 2 + 7) / 8) * cout * B;
      const int oct = (cout % 2 == 0 && waves >= 4096) ? 2 : 1;
      cb.g[0] = conv_geom(cur, net.splat_w[i], net.splat_b[i], out, side, side, cin, cout, 3, 2, true, kTile, 4 * oct);
      cb.trace = next_trace();
      const size_t lds = (size_t)17 * 17 * 16;
      const dim3 grid((unsigned)cb.g[0].tiles, (unsigned)cb.g[0].oc_groups, (unsigned)B);
      if (oct == 2) coeff_conv_first<2><<<grid, 256, lds, s>>>(cb);
      else coeff_conv_first<1><<<grid, 256, lds, s>>>(cb);
      e = hipGetLastError();
    } else {
      cb.g[0] = conv_geom(cur, net.splat_w[i], net.splat_b[i], out, side, side, cin, cout, 3, 2, true, kMTile, 16);
      e = launch_conv_mfma<3, false>(cb, B, s);
    }
    if (e != hipSuccess) return e;
    cur = out;
    side /= 2;
    cin = cout;
  }
  // ---- local conv1 (stride 1) and global conv1 (stride 2) read the splat features: one launch
  float* l1 = buf(ws.local1);
  float* g1 = buf(ws.g1);
  {
    ConvBatch cb{};
    cb.g[0] = conv_geom(cur, net.local_w[0], net.local_b[0], l1, d.sb, d.sb, d.feat, d.gl, 3, 1, true, kMTile, 16);
    cb.g[1] = conv_geom(cur, net.global_conv_w[0], net.global_conv_b[0], g1, d.sb, d.sb, d.feat, d.gl, 3, 2, true, kMTile, 16);
    cb.n = 2;
    e = launch_conv_mfma<3, false>(cb, B, s);
    if (e != hipSuccess) return e;
  }
  // ---- local conv2 (no bias, no activation) and global conv2
  float* l2 = buf(ws.local2);
  float* g2 = buf(ws.g2);
  const int g1side = (d.sb + 1) / 2;
  {
    ConvBatch cb{};
    cb.g[0] = conv_geom(l1, net.local_w[1], net.local_b[1], l2, d.sb, d.sb, d.gl, d.gl, 3, 1, false, kMTile, 16);
    cb.g[1] = conv_geom(g1, net.global_conv_w[1], net.global_conv_b[1], g2, g1side, g1side, d.gl, d.gl, 3, 2, true, kMTile, 16);
    cb.n = 2;
    e = launch_conv_mfma<3, false>(cb, B, s);
    if (e != hipSuccess) return e;
  }
  // ---- fully connected layers: fc1 and fc2 as K-split launches; fc3 inside the prediction layer's workgroups
  const int K1 = d.gside * d.gside * d.gl;
  float* f1 = buf(ws.fc1);
  float* f2 = buf(ws.fc2);
  {
    auto fc_grid = [&](int chunks, int O) { return dim3((unsigned)chunks, (unsigned)((O + 255) / 256), (unsigned)B); };
    FcParams p{g2, nullptr, net.fc_w[0], f1, 1, 0, K1, 4 * d.gl, kFcChunk, next_trace(), net.fc_layout};
    coeff_fc<<<fc_grid(ws.s1, 4 * d.gl), 256, 0, s>>>(p);
    FcParams q{f1, net.fc_b[0], net.fc_w[1], f2, ws.s1, 1, 4 * d.gl, 2 * d.gl, kFcChunk, next_trace(), net.fc_layout};
    coeff_fc<<<fc_grid(ws.s2, 2 * d.gl), 256, 0, s>>>(q);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  // ---- fc3 + fusion + prediction + unroll
  {
    ConvBatch cb{};
    cb.g[0] = conv_geom(l2, net.pred_w, net.pred_b, coeffs, d.sb, d.sb, d.gl, d.pred, 1, 1, false, kMTile, 16);
    cb.n = 1;
    cb.x.gx_part = f2;
    cb.x.gx_bias = net.fc_b[1];
    cb.x.gxS = ws.s2;
    cb.x.gK = 2 * d.gl;
    cb.x.gw = net.fc_w[2];
    cb.x.gb = net.fc_b[2];
    cb.x.gw_oi = net.fc_layout;
    cb.x.gd = d.gd;
    cb.x.n_out = net.n_out;
    cb.x.n_in = net.n_in;
    cb.x.n_levels = net.n_levels;
    cb.x.level_stride = (long long)B * d.sb * d.sb * d.gd * (net.n_out / net.n_levels) * net.n_in;
    e = launch_conv_mfma<1, true>(cb, B, s);
  }
  return e;
}

}  // namespace hdrnet_amd
