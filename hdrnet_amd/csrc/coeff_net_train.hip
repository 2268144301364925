// Training side of the coefficient network (csrc/coeff_net.hip is its forward): the VJP of
// `HDRNetCurves._coefficients` (hdrnet/models.py:62-142) with respect to every weight and bias, for the model
// WITHOUT batch norm -- how the reference's own script trains the guide-network model
// (scripts/ll/train_nn_guide.sh: --nobatch_norm).  The forward pass is the inference launch sequence; its workspace
// (every layer's activation, the fully connected layers' partial sums) is what this file reads back.
//
// On stock ops the backward of this network is ~75 launches of a graph-captured training step (MIOpen backward-data
// and backward-weights kernels with their companions, ReLU masks, pad slices, bias reductions), ~0.4 ms of launch
// latency around a few tens of microseconds of arithmetic (profiles/r04/train_step.md).  Here it is 18 (a layer's
// backward-weights and backward-data share a launch):
//
//   coeff_recompute   x1 / x2 (the fully connected layers' activated inputs, from their saved partial sums), the
//                     global features g, the fusion relu(local + g); the incoming gradient permuted from the unrolled
//                     grid [.., z, i, j] back to the prediction layer's channel order
//   coeff_conv_dw     backward-weights of a convolution on the fp32 matrix cores: D[oc][ic] per tap, the contraction
//                     runs over the PIXELS (4 per v_mfma_f32_16x16x4_f32).  Both operands are plain 4-byte global
//                     loads in lane order (16 output channels x 4 pixels, 4 pixels x 16 input channels: 64-byte
//                     runs), 40 per 4 x 4 pixel tile and lane, all in flight one tile ahead of the 36 MFMAs that use
//                     them; the ReLU mask of the layer's output and the sum of two consumers' gradients are applied
//                     while loading.  Pixel tiles are dealt to workgroups in chunks; the chunks' partial results are
//                     summed in fixed order by coeff_reduce_parts (one launch for all layers).  The bias gradient is
//                     the same kernel's column sum.
//   coeff_conv_dx     backward-data: a stride-1 convolution of the zero-upsampled (stride 2), masked gradient with the
//                     flipped filter -- the forward's 4 x 4 x 16 MFMA kernel with the upsampling and the mask folded
//                     into the LDS staging and the filter read in place ([Cout][kh][kw][Cin]: four 4-byte loads per
//                     16-channel group instead of one float4; no transposed copy of the weights per step).
//   coeff_fc_bwd      a fully connected layer: dW, db and dx in one launch.
//
// Weights are read and gradients written in the layouts torch holds them in (Conv2d weights in channels_last memory
// order = [Cout][kh][kw][Cin]; Linear weights [out][in]): the training step moves no parameter data.
// Deterministic: fixed-order sums, no atomics.
#include <hip/hip_runtime.h>

#include "coeff_net.hip.h"
#include "launch.hip.h"

namespace hdrnet_amd {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kT = 4;          // pixel tile edge (16 pixels = the 16 rows / 4 K-steps of an MFMA tile)
constexpr int kChunkCh = 64;   // coeff_conv_dx: gradient channels staged at a time
constexpr int kMaxB = 8;       // coeff_fc_bwd keeps one accumulator per image in registers

// x / d == umulhi(x, magic32(d)) for x < 2^16 and d >= 2; d == 1 has no 32-bit magic number (udiv handles it)
unsigned magic32(int d) { return d > 1 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u; }
__device__ __forceinline__ int udiv(int x, unsigned mul, int d) { return d == 1 ? x : (int)__umulhi((unsigned)x, mul); }

// ------------------------------------------------------------------------------------------------ backward-weights

struct DwParams {
  const float* x;      // [B][Hin][Win][Cin]: the layer's forward input
  const float* dy;     // [B][Hout][Wout][Cout]: gradient of the layer's output (before its ReLU mask)
  const float* dy2;    // optional second addend (the output feeds two layers), same shape
  const float* ymask;  // optional: the layer's forward output; the gradient passes where it is > 0
  float* dw_part;      // [nchunks][Cout][KK][Cin]
  float* db_part;      // [nchunks][Cout] or null
  int Hin, Win, Cin, Hout, Wout, Cout, stride, pad_top, pad_left;
  int tiles_x, tiles_per_image, tiles_total, tiles_per_chunk;
  unsigned tx_mul, tpi_mul;
  int ic_blocks;
};

template <int KS>
constexpr int dw_lds_floats() { return 4 * KS * KS * 4 * 64 + 4 * 64; }

template <int KS>
__device__ __forceinline__ void conv_dw_body(const DwParams& p, float* lds, int chunk, int pair) {
  constexpr int KK = KS * KS;
  float* red = lds;                       // [4 waves][KK][4][64]
  float* bred = lds + 4 * KK * 4 * 64;    // [4 waves][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ocb = pair / p.ic_blocks, icb = pair - ocb * p.ic_blocks;
  const int oc0 = ocb * 16, ic0 = icb * 16;
  const int q = lane >> 4, m = lane & 15;  // q: pixel of the K-step (A and B); m: output channel (A) / input channel (B)
  const bool a_ok = oc0 + m < p.Cout, b_ok = ic0 + m < p.Cin;
  const int t_end = min((chunk + 1) * p.tiles_per_chunk, p.tiles_total);

  float a_nxt[4], b_nxt[4][KK];
  const int xrow = p.Win * p.Cin;                       // floats per input row
  const int mo = a_ok ? m : 0, mi = b_ok ? m : 0;
  auto fetch = [&](int t) {  // the 4 + 4 * KK operands of tile t, as loads only
    const int b = udiv(t, p.tpi_mul, p.tiles_per_image);
    const int r = t - b * p.tiles_per_image;
    const int ty = udiv(r, p.tx_mul, p.tiles_x), tx = r - ty * p.tiles_x;
    const int ox = tx * kT + q, oxc = min(ox, p.Wout - 1);
    const int ix0 = ox * p.stride - p.pad_left;
    const size_t yimg = (size_t)b * p.Hout * p.Wout * p.Cout + oc0 + mo;
    const float* dyb = p.dy + yimg;
    const float* dy2b = p.dy2 ? p.dy2 + yimg : nullptr;
    const float* ymb = p.ymask ? p.ymask + yimg : nullptr;
    const float* xb = p.x + (size_t)b * p.Hin * xrow + ic0 + mi;
    bool xok[KS];
    int xoff[KS];
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const int ix = ix0 + kx;
      xok[kx] = (unsigned)ix < (unsigned)p.Win && ox < p.Wout && b_ok;
      xoff[kx] = min(max(ix, 0), p.Win - 1) * p.Cin;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int oy = ty * kT + s;
      const bool in = oy < p.Hout && ox < p.Wout;
      const int off = (min(oy, p.Hout - 1) * p.Wout + oxc) * p.Cout;
      float v = dyb[off];
      if (dy2b) v += dy2b[off];
      if (ymb) v = ymb[off] > 0.0f ? v : 0.0f;
      a_nxt[s] = (in && a_ok) ? v : 0.0f;
      const int iy0 = oy * p.stride - p.pad_top;
#pragma unroll
      for (int ky = 0; ky < KS; ++ky) {
        const int iy = iy0 + ky;
        const bool yok = (unsigned)iy < (unsigned)p.Hin && oy < p.Hout;
        const float* xr = xb + min(max(iy, 0), p.Hin - 1) * xrow;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const float xv = xr[xoff[kx]];
          b_nxt[s][ky * KS + kx] = (yok && xok[kx]) ? xv : 0.0f;
        }
      }
    }
  };

  v4f acc[KK];
#pragma unroll
  for (int tap = 0; tap < KK; ++tap) acc[tap] = v4f{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.0f;
  int t = chunk * p.tiles_per_chunk + wave;
  if (t < t_end) fetch(t);
  for (; t < t_end; t += 4) {
    float a_cur[4], b_cur[4][KK];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a_cur[s] = a_nxt[s];
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) b_cur[s][tap] = b_nxt[s][tap];
    }
    if (t + 4 < t_end) fetch(t + 4);  // in flight under this tile's MFMAs
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bsum += a_cur[s];
#pragma unroll
      for (int tap = 0; tap < KK; ++tap)
        acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], b_cur[s][tap], acc[tap], 0, 0, 0);
    }
  }
  // D[i = oc][j = ic] per tap: lane holds rows 4 * (lane >> 4) + r, column lane & 15
#pragma unroll
  for (int tap = 0; tap < KK; ++tap) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * KK + tap) * 4 + r) * 64 + lane] = acc[tap][r];
  }
  bred[wave * 64 + lane] = bsum;
  __syncthreads();
  {
    const int r = wave;  // thread (wave, lane) finishes row 4 * q + wave, column m of every tap
    const int oc = oc0 + 4 * q + r, ic = ic0 + m;
    if (oc < p.Cout && ic < p.Cin) {
#pragma unroll
      for (int tap = 0; tap < KK; ++tap) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[((w * KK + tap) * 4 + r) * 64 + lane];
        p.dw_part[(((size_t)chunk * p.Cout + oc) * KK + tap) * p.Cin + ic] = v;
      }
    }
  }
  if (p.db_part && icb == 0 && tid < 16 && oc0 + tid < p.Cout) {
    float v = 0.0f;
    for (int w = 0; w < 4; ++w) {
      for (int qq = 0; qq < 4; ++qq) v += bred[w * 64 + qq * 16 + tid];
    }
    p.db_part[(size_t)chunk * p.Cout + oc0 + tid] = v;
  }
}

template <int KS>
__global__ __launch_bounds__(256) void coeff_conv_dw(const DwParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  conv_dw_body<KS>(p, lds, blockIdx.x, blockIdx.y);
}

// Backward-weights of the FIRST splat layer (3x3, stride 2, Cin = 3, Cout <= 9): K = 27 per output channel is no
// matrix-core shape -- on coeff_conv_dw the 16 x 16 tile is 9 % full and its 40 four-byte gathers per tile and lane
// make the launch address-bound (36 us at 4 x 128 x 128 output pixels).  Here thread = (output channel, tap x input
// channel | bias), the 8 x 8 output pixels of a tile and their 17 x 17 x 3 input patch staged in LDS; a thread's 64
// products per tile read two LDS words each.
struct DwFirstParams {
  const float* x;      // [B][Hin][Win][3]
  const float* dy;     // [B][Hout][Wout][Cout]
  const float* ymask;  // the layer's output (ReLU mask)
  float* dw_part;      // [nchunks][Cout][9][3]
  float* db_part;      // [nchunks][Cout]
  int Hin, Win, Hout, Wout, Cout, pad_top, pad_left;
  int tiles_x, tiles_per_image, tiles_total, tiles_per_chunk;
  unsigned tx_mul, tpi_mul, cout_mul;
};

__global__ __launch_bounds__(256) void coeff_conv_dw_first(const DwFirstParams p) {
  constexpr int T8 = 8, TI = 17;
  constexpr int NX = (TI * TI * 3 + 255) / 256, NY = (T8 * T8 * 9 + 255) / 256;
  __shared__ float xs[TI * TI * 3];
  __shared__ float ys[T8 * T8 * 9];
  const int tid = threadIdx.x;
  const int oc = tid / 28, tt = tid - oc * 28;  // tt < 27: (tap, ic); tt == 27: the bias
  const bool on = oc < p.Cout;
  const int tap = tt / 3, ic = tt - tap * 3;
  const int xoff = tt < 27 ? ((tap / 3) * TI + (tap % 3)) * 3 + ic : 0;
  const int chunk = blockIdx.x;
  const int t_end = min((chunk + 1) * p.tiles_per_chunk, p.tiles_total);
  // A tile's input patch and masked gradient travel through registers: the NEXT tile's loads are issued before this
  // tile's products, and the ReLU mask and the gradient are two independent loads (a select, not a dependent load).
  float xr[NX], yr[NY], mr[NY];
  auto load = [&](int t) {
    const int b = udiv(t, p.tpi_mul, p.tiles_per_image);
    const int r = t - b * p.tiles_per_image;
    const int ty = udiv(r, p.tx_mul, p.tiles_x), tx = r - ty * p.tiles_x;
    const int oy0 = ty * T8, ox0 = tx * T8;
    const int iy0 = oy0 * 2 - p.pad_top, ix0 = ox0 * 2 - p.pad_left;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + 256 * j;
      const int pix = i / 3, c = i - pix * 3;
      const int py = pix / TI, px = pix - py * TI;
      const int gy = iy0 + py, gx = ix0 + px;
      const bool ok = i < TI * TI * 3 && (unsigned)gy < (unsigned)p.Hin && (unsigned)gx < (unsigned)p.Win;
      xr[j] = ok ? p.x[(((size_t)b * p.Hin + gy) * p.Win + gx) * 3 + c] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int i = tid + 256 * j;
      const int pix = udiv(i, p.cout_mul, p.Cout), c = i - pix * p.Cout;
      const int oy = oy0 + (pix >> 3), ox = ox0 + (pix & 7);
      const bool ok = pix < T8 * T8 && oy < p.Hout && ox < p.Wout;
      const size_t off = ok ? (((size_t)b * p.Hout + oy) * p.Wout + ox) * p.Cout + c : 0;
      const float m = p.ymask[off], d = p.dy[off];
      mr[j] = ok ? m : 0.0f;
      yr[j] = d;
    }
  };
  float acc = 0.0f;
  int t = chunk * p.tiles_per_chunk;
  if (t < t_end) load(t);
  for (; t < t_end; ++t) {
    __syncthreads();  // the previous tile has been consumed
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + 256 * j;
      if (i < TI * TI * 3) xs[i] = xr[j];
    }
#pragma unroll
    for (int j = 0; j < NY; ++j) {
      const int i = tid + 256 * j;
      const int pix = udiv(i, p.cout_mul, p.Cout), c = i - pix * p.Cout;
      if (pix < T8 * T8) ys[pix * 9 + c] = mr[j] > 0.0f ? yr[j] : 0.0f;
    }
    __syncthreads();
    if (t + 1 < t_end) load(t + 1);
    if (on) {
      if (tt < 27) {
#pragma unroll 8
        for (int pix = 0; pix < T8 * T8; ++pix)
          acc = __builtin_fmaf(ys[pix * 9 + oc], xs[((pix >> 3) * 2 * TI + (pix & 7) * 2) * 3 + xoff], acc);
      } else {
#pragma unroll 8
        for (int pix = 0; pix < T8 * T8; ++pix) acc += ys[pix * 9 + oc];
      }
    }
  }
  if (on) {
    if (tt < 27) p.dw_part[((size_t)chunk * p.Cout + oc) * 27 + tt] = acc;
    else p.db_part[(size_t)chunk * p.Cout + oc] = acc;
  }
}

// Sums the chunks' partial results of every layer in one launch: entry e owns blocks [first[e], first[e + 1]).
constexpr int kMaxParts = 32;
struct ReduceTab {
  const float* src[kMaxParts];
  float* dst[kMaxParts];
  int n[kMaxParts], nsplit[kMaxParts], first[kMaxParts + 1];
  int vec[kMaxParts];  // 4 elements per lane: the length is a multiple of 4 and the partial sums are 16-byte aligned
  int count;
};

// Block = 16 lanes of consecutive elements x 16 groups of the chunks; every load of the block is in flight at once (a loop
// over 512 chunks, one load at a time, was 39 us of the step).  Entries whose length is a multiple of 4 -- every weight
// tensor, most biases -- take 4 elements per lane as one 16-byte load (64 elements per block: a quarter of the ~11 000
// workgroups the launch had, which were what its 12 us consisted of); `first` counts blocks accordingly.
__global__ __launch_bounds__(256) void coeff_reduce_parts(const ReduceTab tab) {
  __shared__ float red[16][68];
  int e = 0;
  while (e + 1 < tab.count && (int)blockIdx.x >= tab.first[e + 1]) ++e;  // uniform
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int n = tab.n[e], ns = tab.nsplit[e];
  const int blk = (int)blockIdx.x - tab.first[e];
  if (!tab.vec[e]) {  // scalar lanes
    const int i = blk * 16 + el;
    float v = 0.0f;
    if (i < n) {
      const float* s = tab.src[e] + i;
      for (int k0 = 0; k0 < ns; k0 += 256) {
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int k = k0 + g + 16 * j;
          t[j] = k < ns ? s[(size_t)k * n] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v += t[j];
      }
    }
    red[g][el] = v;
    __syncthreads();
    if (g == 0 && i < n) {
      float t = 0.0f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[k][el];
      tab.dst[e][i] = t;
    }
    return;
  }
  const int i = (blk * 16 + el) * 4;
  float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (i < n) {
    const float* s = tab.src[e] + i;
    for (int k0 = 0; k0 < ns; k0 += 256) {
      float4 t[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int k = k0 + g + 16 * j;
        t[j] = k < ns ? *reinterpret_cast<const float4*>(s + (size_t)k * n) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
    }
  }
  *reinterpret_cast<float4*>(&red[g][el * 4]) = v;
  __syncthreads();
  if (g < 4 && i < n) {  // lane (el, g): element i + g
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][el * 4 + g];
    tab.dst[e][i + g] = t;
  }
}

// ------------------------------------------------------------------------------------------------- backward-data

struct DxParams {
  const float* dy;     // [B][Hy][Wy][Cy]: gradient of the layer's output
  const float* dy2;    // optional second addend
  const float* ymask;  // optional: the layer's forward output (ReLU mask)
  const float* w;      // the layer's filter, [Cy][KK][Cx]
  float* dx;           // [B][Hx][Wx][Cx]
  int Hy, Wy, Cy, Hx, Wx, Cx;
  int ups, pad_top, pad_left;  // the gradient upsampled by the forward stride; pad' = KS - 1 - forward pad
  int tiles_x, tiles, oc_groups;
  int tpb, tile_blocks;  // a workgroup's run of consecutive tiles (one is a chain of round trips: the next tile's loads fly
                         // under this tile's products); tile_blocks = ceil(tiles / tpb)
  unsigned ti_mul, tx_mul;
  int c4shift, nchunks;
  unsigned lds_off[4][12];  // per wave: LDS float offset of each (tap, 16-channel group) step | group << 24; [9] = count
  unsigned w_off[4][12];    // per wave: filter float offset of the step: (16 g * KK + flipped tap) * Cx
  // the prediction layer only: the tile's column sums of dx where the fusion passed it,
  // colsum_part[b][tile][c] = sum over the tile's pixels of dx[b][px][c] * (xmask[b][px][c] > 0)
  const float* xmask;
  float* colsum_part;
};

// dx[b, i, :] = sum_taps U[b, i - pad' + tap, :] . w[:, KK - 1 - tap, :],  U = the masked gradient, zero-upsampled.
// MFMA roles as in coeff_conv_mfma: rows = the tile's 16 pixels of dx, columns = 16 channels of dx (Cx), the k of
// MFMA e of a 16-channel group of the gradient is channel 16 g + 4 kk + e.
// MULTI: the gradient has more than one chunk of kChunkCh channels (the filter elements change from step to step: a second
// set of 9 float4 in flight; without it the kernel fits a third wave per SIMD).
template <int KS, bool MULTI>
__device__ __forceinline__ void conv_dx_body(const DxParams& p, float* lds, int tile_block, int ocg, int b) {
  constexpr int KK = KS * KS;
  constexpr int kMaxSteps = KK;
  constexpr int TI = (kT - 1) + KS;  // stride 1 over the upsampled gradient
  constexpr int npix = TI * TI;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned lstep[kMaxSteps], wstep[kMaxSteps];
#pragma unroll
  for (int si = 0; si < kMaxSteps; ++si) {
    lstep[si] = p.lds_off[wave][si];
    wstep[si] = p.w_off[wave][si];
  }
  const int nsw = (int)p.lds_off[wave][9];
  const int tile0 = tile_block * p.tpb;
  const int ntile = min(p.tpb, p.tiles - tile0);
  const int Hu = (p.Hy - 1) * p.ups + 1, Wu = (p.Wy - 1) * p.ups + 1;
  const int ushift = p.ups >> 1;  // ups in {1, 2}
  const int Cy = p.Cy, Cx = p.Cx;
  const int cch = 4 << p.c4shift;
  const int PS = cch + 4;
  float* red = lds + npix * PS;  // [4 waves][4][64]
  const int q = lane >> 4, j = lane & 15;
  const int n0 = ocg * 16;
  const bool qvalid = 4 * q < cch;
  const bool bvalid = qvalid && n0 + j < Cx;
  const size_t img = (size_t)b * p.Hy * p.Wy * Cy;

  constexpr int kMaxU = (TI * TI * (kChunkCh / 4) + 255) / 256;
  const int c4 = tid & ((1 << p.c4shift) - 1), pix0 = tid >> p.c4shift, pstep = 256 >> p.c4shift;
  const int nU = ((npix << p.c4shift) + 255) >> 8;
  float4 st[kMaxU], st2[kMaxU], stm[kMaxU];
  unsigned okbits = 0;
  auto fetch_tile = [&](int tile, int chunk) {
    const int tyi = udiv(tile, p.tx_mul, p.tiles_x), txi = tile - tyi * p.tiles_x;
    const int iy0 = tyi * kT - p.pad_top, ix0 = txi * kT - p.pad_left;  // in the upsampled gradient
    okbits = 0;
#pragma unroll
    for (int u = 0; u < kMaxU; ++u) {
      if (u < nU) {  // uniform
        const int pix = pix0 + u * pstep;
        const int py = (int)__umulhi((unsigned)pix, p.ti_mul), px = pix - py * TI;
        const int gy = iy0 + py, gx = ix0 + px;
        const int gyc = min(max(gy, 0), Hu - 1), gxc = min(max(gx, 0), Wu - 1);
        const int c = chunk * kChunkCh + 4 * c4;  // the gradient's channel count need not be a multiple of the chunk
        const bool ok = pix < npix && gy == gyc && gx == gxc && (gy & (p.ups - 1)) == 0 && (gx & (p.ups - 1)) == 0 && c < Cy;
        const size_t off = img + ((size_t)(gyc >> ushift) * p.Wy + (gxc >> ushift)) * Cy + (c < Cy ? c : 0);
        st[u] = *reinterpret_cast<const float4*>(p.dy + off);
        if (p.dy2) st2[u] = *reinterpret_cast<const float4*>(p.dy2 + off);
        if (p.ymask) stm[u] = *reinterpret_cast<const float4*>(p.ymask + off);
        okbits |= ok ? (1u << u) : 0u;
      }
    }
  };
  auto stash_tile = [&]() {
#pragma unroll
    for (int u = 0; u < kMaxU; ++u) {
      const int pix = pix0 + u * pstep;
      if (u < nU && pix < npix) {
        float4 v = st[u];
        if (p.dy2) v = make_float4(v.x + st2[u].x, v.y + st2[u].y, v.z + st2[u].z, v.w + st2[u].w);
        if (p.ymask) {
          v.x = stm[u].x > 0.0f ? v.x : 0.0f;
          v.y = stm[u].y > 0.0f ? v.y : 0.0f;
          v.z = stm[u].z > 0.0f ? v.z : 0.0f;
          v.w = stm[u].w > 0.0f ? v.w : 0.0f;
        }
        if (!((okbits >> u) & 1u)) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(lds + pix * PS + 4 * c4) = v;
      }
    }
  };
  // this lane's filter elements: gradient channels 16 g + 4 q + e (rows of [Cy][KK][Cx]), dx channel n0 + j
  const float* wl = p.w + (size_t)(qvalid ? 4 * q : 0) * KK * Cx + min(n0 + j, Cx - 1);
  const size_t estride = (size_t)KK * Cx;
  auto fetch_w = [&](int chunk, float4 (&dst)[kMaxSteps]) {
#pragma unroll
    for (int si = 0; si < kMaxSteps; ++si) {
      if (si < nsw) {  // uniform
        const int row = chunk * kChunkCh + 16 * (int)(lstep[si] >> 24) + 4 * q;  // first of the lane's 4 filter rows
        const bool rok = bvalid && row < Cy;
        const float* ws = wl + (rok ? wstep[si] + (size_t)chunk * kChunkCh * estride : 0);
        const float4 v = make_float4(ws[0], ws[estride], ws[2 * estride], ws[3 * estride]);
        dst[si] = rok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  float4 bw[kMaxSteps];
  fetch_tile(tile0, 0);
  fetch_w(0, bw);
  v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int ti = lane & 15;
  const float* tl = lds + ((ti >> 2) * TI + (ti & 3)) * PS + (qvalid ? 4 * q : 0);
  int tile = tile0, chunk = 0;
  for (int step = 0, nstep = ntile * p.nchunks; step < nstep; ++step) {  // (tile, chunk) pairs, tile-major
    if (step > 0) __syncthreads();
    stash_tile();
    __syncthreads();
    const bool last_chunk = chunk + 1 == p.nchunks;
    const int ntl = last_chunk ? tile + 1 : tile, nch = last_chunk ? 0 : chunk + 1;
    float4 bwn[MULTI ? kMaxSteps : 1];
    if (step + 1 < nstep) {
      if constexpr (MULTI) fetch_w(nch, bwn);  // (one chunk: the same filter elements serve every tile)
      fetch_tile(ntl, nch);
    }
#pragma unroll
    for (int si = 0; si < kMaxSteps; ++si) {
      if (si < nsw) {  // uniform
        float4 x = *reinterpret_cast<const float4*>(tl + (lstep[si] & 0xffffffu));
        if (!qvalid) x = make_float4(0.f, 0.f, 0.f, 0.f);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, bw[si].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, bw[si].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, bw[si].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, bw[si].w, acc1, 0, 0, 0);
      }
    }
    if constexpr (MULTI) {
      if (step + 1 < nstep) {
#pragma unroll
        for (int si = 0; si < kMaxSteps; ++si) bw[si] = bwn[si];
      }
    }
    if (last_chunk) {  // uniform: the tile is complete
      const v4f acc = acc0 + acc1;
      acc0 = acc1 = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
      __syncthreads();
      const int r = wave;
      float v = red[r * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) v += red[(w * 4 + r) * 64 + lane];
      const int tyi = udiv(tile, p.tx_mul, p.tiles_x), txi = tile - tyi * p.tiles_x;
      const int i = 4 * q + r, o = n0 + j;
      const int oy = tyi * kT + (i >> 2), ox = txi * kT + (i & 3);
      const bool in = o < Cx && oy < p.Hx && ox < p.Wx;
      const size_t off = (((size_t)b * p.Hx + min(oy, p.Hx - 1)) * p.Wx + min(ox, p.Wx - 1)) * Cx + min(o, Cx - 1);
      if (in) p.dx[off] = v;
      if (p.colsum_part) {  // uniform
        const float mv = (in && p.xmask[off] > 0.0f) ? v : 0.0f;
        __syncthreads();  // everyone has read its partial tiles
        red[i * 16 + j] = mv;
        __syncthreads();
        if (tid < 16 && n0 + tid < Cx) {
          float t = 0.0f;
#pragma unroll
          for (int k = 0; k < 16; ++k) t += red[k * 16 + tid];
          p.colsum_part[((size_t)b * p.tiles + tile) * Cx + n0 + tid] = t;
        }
      }
    }
    tile = ntl;
    chunk = nch;
  }
}

template <int KS, bool MULTI>
__global__ __launch_bounds__(256) void coeff_conv_dx(const DxParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  conv_dx_body<KS, MULTI>(p, lds, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Backward-weights and backward-data of ONE layer in one launch (they read the same gradient and do not depend on
// each other): the first `dw_blocks` workgroups are coeff_conv_dw's (chunk, channel-block pair), the rest
// coeff_conv_dx's (tile, channel group, image).  A launch less per layer: ~6 us of ~14.
struct BwdPair {
  DwParams dw;
  DxParams dx;
  int dw_chunks, dw_blocks;   // dw grid: dw_chunks x pairs, flattened
  int dx_tiles, dx_groups;    // dx grid: tiles x groups x B, flattened
  unsigned chunk_mul, tile_mul, tg_mul;  // magic numbers of dw_chunks, dx_tiles, dx_tiles * dx_groups
};

template <int KS, bool MULTI>
__device__ __forceinline__ void bwd_pair_block(const BwdPair& pr, float* lds, int id) {
  if (id < pr.dw_blocks) {  // uniform
    const int pair = udiv(id, pr.chunk_mul, pr.dw_chunks);
    conv_dw_body<KS>(pr.dw, lds, id - pair * pr.dw_chunks, pair);
  } else {
    const int r = id - pr.dw_blocks;
    const int tg = pr.dx_tiles * pr.dx_groups;
    const int b = udiv(r, pr.tg_mul, tg);
    const int r2 = r - b * tg;
    const int ocg = udiv(r2, pr.tile_mul, pr.dx_tiles);
    conv_dx_body<KS, MULTI>(pr.dx, lds, r2 - ocg * pr.dx_tiles, ocg, b);
  }
}

template <int KS, bool MULTI>
__global__ __launch_bounds__(256) void coeff_conv_bwd(const BwdPair pr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  bwd_pair_block<KS, MULTI>(pr, lds, blockIdx.x);
}

// TWO layers' pairs in one launch: the local and the global path of the network do not depend on each other (local2 with
// global conv2, then local1 with global conv1) -- the forward batches them the same way (coeff_net.hip: ConvBatch).
struct BwdTwo {
  BwdPair p[2];
  int first_blocks;
};

template <int KS, bool MULTI>
__global__ __launch_bounds__(256) void coeff_conv_bwd2(const BwdTwo two) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bool second = (int)blockIdx.x >= two.first_blocks;  // uniform
  if (second) bwd_pair_block<KS, MULTI>(two.p[1], lds, (int)blockIdx.x - two.first_blocks);
  else bwd_pair_block<KS, MULTI>(two.p[0], lds, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------- fully connected layers

struct FcBwdParams {
  const float* x;   // [B][K]: the layer's (activated) input
  const float* dy;  // [B][O]
  const float* w;   // [O][K]
  float* dw;        // [O][K]
  float* db;        // [O]
  float* dx;        // [B][K] or null
  int B, K, O, mask_x;  // mask_x: dx passes where x > 0 (the input is a ReLU's output)
};

// Block = 16 inputs k x 16 parts of the outputs; every (o, k) of dW belongs to exactly one thread.  The layers are tiny
// (<= 1 MB of weights) and the kernel is a chain of memory round trips, so every load of a 256-output chunk is issued before
// the first is used: the 16 weights of a thread as predicated loads of a fully unrolled loop (a run-time trip count leaves
// small layers in the compiler's serial remainder loop: one round trip per output), dy staged through the LDS once per
// workgroup instead of B broadcast loads per output, the bias gradient's loads at the top, spread over the workgroups.
__global__ __launch_bounds__(256) void coeff_fc_bwd(const FcBwdParams p) {
  __shared__ float red[kMaxB][16][17];
  __shared__ float dys[kMaxB][256];
  const int tid = threadIdx.x, kl = tid & 15, op = tid >> 4;
  const int k = blockIdx.x * 16 + kl;
  const bool k_ok = k < p.K;
  float xk[kMaxB], dxp[kMaxB], dbv[kMaxB];
  const int ob = blockIdx.x * 256 + tid;  // this thread's bias gradient (one per thread of the first O / 256 workgroups)
#pragma unroll
  for (int b = 0; b < kMaxB; ++b) {
    xk[b] = (b < p.B && k_ok) ? p.x[(size_t)b * p.K + k] : 0.0f;
    dbv[b] = (b < p.B && ob < p.O) ? p.dy[(size_t)b * p.O + ob] : 0.0f;
    dxp[b] = 0.0f;
  }
  for (int o0 = 0; o0 < p.O; o0 += 256) {
    if (o0 > 0) __syncthreads();
#pragma unroll
    for (int b = 0; b < kMaxB; ++b)
      if (b < p.B) dys[b][tid] = (o0 + tid < p.O) ? p.dy[(size_t)b * p.O + o0 + tid] : 0.0f;
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = o0 + op + 16 * i;
      w[i] = (k_ok && o < p.O) ? p.w[(size_t)o * p.K + k] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = o0 + op + 16 * i;
      float dwv = 0.0f;
#pragma unroll
      for (int b = 0; b < kMaxB; ++b) {
        if (b < p.B) {
          const float g = dys[b][op + 16 * i];
          dwv = __builtin_fmaf(g, xk[b], dwv);
          dxp[b] = __builtin_fmaf(g, w[i], dxp[b]);
        }
      }
      if (k_ok && o < p.O) p.dw[(size_t)o * p.K + k] = dwv;
    }
  }
#pragma unroll
  for (int b = 0; b < kMaxB; ++b) red[b][op][kl] = dxp[b];
  __syncthreads();
  if (p.dx && op < p.B && k_ok) {  // thread (kl, op = image)
    float v = 0.0f;
#pragma unroll
    for (int o2 = 0; o2 < 16; ++o2) v += red[op][o2][kl];
    const float xv = p.x[(size_t)op * p.K + k];
    p.dx[(size_t)op * p.K + k] = (p.mask_x && !(xv > 0.0f)) ? 0.0f : v;
  }
  if (ob < p.O) {
    float v = 0.0f;
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) v += dbv[b];  // images beyond B hold zeros: the order of the sum is b = 0, 1, ...
    p.db[ob] = v;
  }
  for (int o = ob + (int)gridDim.x * 256; o < p.O; o += (int)gridDim.x * 256) {  // (never taken: grid >= O / 256)
    float v = 0.0f;
    for (int b = 0; b < p.B; ++b) v += p.dy[(size_t)b * p.O + o];
    p.db[o] = v;
  }
}

// ----------------------------------------------------------------------- what the forward did not keep; permutes

struct RecomputeParams {
  const float* f1part; int s1;  // fc1's partial sums [B][s1][K2]
  const float* f2part; int s2;  // fc2's            [B][s2][K3]
  const float* b1; const float* b2; const float* w3; const float* b3;  // w3 [O3][K3]
  const float* local2;          // [B][P][O3]
  const float* dcoeffs;         // [B][P][gd][n_out][n_in]
  float* x1; float* x2; float* g; float* fusion;  // [B][K2], [B][K3], [B][O3], [B][P][O3]
  float* dyp;                   // [B][P][gd * n_out * n_in] in the prediction layer's channel order
  int K2, K3, O3, P, gd, n_out, n_in;
  unsigned C_mul, gd_mul, nout_mul;  // magic numbers of gd * n_out * n_in, gd, n_out
};

// One workgroup per image (grid.x) and slab of cells (grid.y < slabs); every slab re-derives the (tiny) global features.
// The workgroups y == slabs do nothing but x1 (64 partial sums per element: the longest chain of loads, and nothing in this
// launch needs its result).  A slab is a chain of memory round trips -- x2, g, then the two streaming loops -- so every
// loop issues its loads in batches (predicated, fully unrolled) and the index arithmetic uses host-made magic numbers.
constexpr int kRecomputeSlabs = 32;

__global__ __launch_bounds__(256) void coeff_recompute(const RecomputeParams p) {
  __shared__ float x2s[512];
  __shared__ float gs[256];
  __shared__ float red[256];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int slabs = (int)gridDim.y - 1;
  if ((int)blockIdx.y == slabs) {  // uniform
    for (int k = tid; k < p.K2; k += 256) {
      float v = p.b1[k];
      const float* s = p.f1part + (size_t)b * p.s1 * p.K2 + k;
      for (int i0 = 0; i0 < p.s1; i0 += 16) {
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = i0 + j < p.s1 ? s[(size_t)(i0 + j) * p.K2] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j) v += t[j];
      }
      p.x1[(size_t)b * p.K2 + k] = fmaxf(v, 0.0f);
    }
    return;
  }
  const bool first = blockIdx.y == 0;
  for (int k = tid; k < p.K3; k += 256) {
    float v = p.b2[k];
    const float* s = p.f2part + (size_t)b * p.s2 * p.K3 + k;
    for (int i0 = 0; i0 < p.s2; i0 += 16) {
      float t[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) t[j] = i0 + j < p.s2 ? s[(size_t)(i0 + j) * p.K3] : 0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) v += t[j];
    }
    v = fmaxf(v, 0.0f);
    x2s[k] = v;
    if (first) p.x2[(size_t)b * p.K3 + k] = v;
  }
  __syncthreads();
  // g[c] = b3[c] + sum_k x2[k] w3[c][k]: thread = (channel, K part), the parts summed through LDS
  const int parts = p.O3 <= 256 ? 256 / p.O3 : 1;  // O3 a power of two
  for (int c0 = 0; c0 < p.O3; c0 += 256) {
    const int c = c0 + tid / parts, kp = tid % parts;
    const int nk = p.K3 / parts;
    float v = 0.0f;
    if (c < p.O3) {
      const float* wr = p.w3 + (size_t)c * p.K3 + kp * nk;
      for (int k0 = 0; k0 < nk; k0 += 16) {
        float t[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) t[j] = k0 + j < nk ? wr[k0 + j] : 0.0f;
#pragma unroll
        for (int j = 0; j < 16; ++j)
          if (k0 + j < nk) v = __builtin_fmaf(x2s[kp * nk + k0 + j], t[j], v);
      }
    }
    red[tid] = v;
    __syncthreads();
    if (kp == 0 && c < p.O3) {
      float t = p.b3[c];
      for (int k = 0; k < parts; ++k) t += red[tid + k];
      gs[c] = t;
      if (first) p.g[(size_t)b * p.O3 + c] = t;
    }
    __syncthreads();
  }
  const int per = (p.P + slabs - 1) / slabs;
  const int px0 = blockIdx.y * per, px1 = min(px0 + per, p.P);
  {
    const size_t base = ((size_t)b * p.P + px0) * p.O3;
    const int n = max(px1 - px0, 0) * p.O3;
    for (int i0 = 0; i0 < n; i0 += 4 * 256) {
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 256 + tid;
        t[j] = i < n ? p.local2[base + i] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 256 + tid;
        if (i < n) p.fusion[base + i] = fmaxf(t[j] + gs[i & (p.O3 - 1)], 0.0f);  // px0 * O3 is a multiple of O3
      }
    }
  }
  const int C = p.gd * p.n_out * p.n_in;
  {
    const size_t base = ((size_t)b * p.P + px0) * C;
    const int n = max(px1 - px0, 0) * C;  // < 2^16 (the magic divisions): <= 32 cells x 288 channels per slab
    for (int i0 = 0; i0 < n; i0 += 4 * 256) {
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 256 + tid;
        const int px = udiv(i, p.C_mul, C), o = i - px * C;  // o = (jj * n_out + ii) * gd + z
        const int ji = udiv(o, p.gd_mul, p.gd), z = o - ji * p.gd;
        const int jj = udiv(ji, p.nout_mul, p.n_out), ii = ji - jj * p.n_out;
        t[j] = i < n ? p.dcoeffs[(((size_t)b * p.P + px0 + px) * p.gd + z) * p.n_out * p.n_in + ii * p.n_in + jj] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * 256 + tid;
        if (i < n) p.dyp[base + i] = t[j];
      }
    }
  }
}

// dg[b][c] = the sum of the prediction layer's per-tile column sums (coeff_conv_dx's colsum_part).
__global__ __launch_bounds__(256) void coeff_slab_sum(const float* __restrict__ part, float* __restrict__ dg, int nslab, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = 0.0f;
#pragma unroll 8
    for (int s = 0; s < nslab; ++s) v += part[((size_t)b * nslab + s) * C + c];
    dg[(size_t)b * C + c] = v;
  }
}

// -------------------------------------------------------------------------------------------------------- host

struct Layer {  // one convolution of the network, forward geometry
  const float* x; const float* y; const float* w;
  float* dw; float* db;
  int Hin, Cin, Hout, Cout, ks, stride;
};

int same_pad_before(int in, int out, int k, int s) {
  const int total = (out - 1) * s + k - in;
  return total > 0 ? total / 2 : 0;
}

struct PartPlan {
  int tiles_x, tpi, total, tpc, nchunks;
};

PartPlan part_plan(int B, int Hout, int pairs) {
  PartPlan pl;
  pl.tiles_x = (Hout + kT - 1) / kT;
  pl.tpi = pl.tiles_x * pl.tiles_x;
  pl.total = pl.tpi * B;
  // ~256 workgroups per layer at most (one per CU), at least one tile per wave
  int nchunks = (pl.total + 3) / 4;
  const int cap = 256 / (pairs > 0 ? pairs : 1) > 1 ? 256 / pairs : 1;
  if (nchunks > cap) nchunks = cap;
  pl.tpc = (pl.total + nchunks - 1) / nchunks;
  pl.nchunks = (pl.total + pl.tpc - 1) / pl.tpc;
  return pl;
}

PartPlan part_plan_first(int B, int Hout);
bool first_layer_shape(const Layer& L);

size_t dw_part_floats(int B, const Layer& L) {
  const int pairs = ((L.Cout + 15) / 16) * ((L.Cin + 15) / 16);
  const PartPlan pl = part_plan(B, L.Hout, pairs);
  int nchunks = pl.nchunks;
  if (first_layer_shape(L)) nchunks = nchunks > 256 ? nchunks : 256;  // either kernel's chunk count fits
  return (size_t)nchunks * ((size_t)L.Cout * L.ks * L.ks * L.Cin + L.Cout);
}

bool first_layer_shape(const Layer& L) { return L.Cin == 3 && L.Cout <= 9 && L.ks == 3 && L.stride == 2; }

PartPlan part_plan_first(int B, int Hout) {  // 8 x 8 pixel tiles, one workgroup per chunk, <= 256 chunks
  PartPlan pl;
  pl.tiles_x = (Hout + 7) / 8;
  pl.tpi = pl.tiles_x * pl.tiles_x;
  pl.total = pl.tpi * B;
  int nchunks = pl.total < 256 ? pl.total : 256;
  pl.tpc = (pl.total + nchunks - 1) / nchunks;
  pl.nchunks = (pl.total + pl.tpc - 1) / pl.tpc;
  return pl;
}

DwParams make_dw(const Layer& L, int B, const float* dy, const float* dy2, bool mask, float* part, const PartPlan& pl,
                 int icb) {
  (void)B;
  DwParams p{};
  p.x = L.x; p.dy = dy; p.dy2 = dy2; p.ymask = mask ? L.y : nullptr;
  const size_t nw = (size_t)L.Cout * L.ks * L.ks * L.Cin;
  p.dw_part = part;
  p.db_part = L.db ? part + (size_t)pl.nchunks * nw : nullptr;
  p.Hin = L.Hin; p.Win = L.Hin; p.Cin = L.Cin; p.Hout = L.Hout; p.Wout = L.Hout; p.Cout = L.Cout;
  p.stride = L.stride;
  p.pad_top = p.pad_left = same_pad_before(L.Hin, L.Hout, L.ks, L.stride);
  p.tiles_x = pl.tiles_x; p.tiles_per_image = pl.tpi; p.tiles_total = pl.total; p.tiles_per_chunk = pl.tpc;
  p.tx_mul = magic32(pl.tiles_x); p.tpi_mul = magic32(pl.tpi);
  p.ic_blocks = icb;
  return p;
}

void add_reduce_entry(ReduceTab* tab, const float* src, float* dst, int n, int nchunks) {
  const int e = tab->count++;
  tab->src[e] = src; tab->dst[e] = dst; tab->n[e] = n; tab->nsplit[e] = nchunks;
  tab->vec[e] = (n % 4 == 0 && ((uintptr_t)src & 15u) == 0) ? 1 : 0;
  tab->first[e + 1] = tab->first[e] + (tab->vec[e] ? (n / 4 + 15) / 16 : (n + 15) / 16);
}

void add_reduce(ReduceTab* tab, const DwParams& p, const Layer& L, int nchunks) {
  auto add = [&](const float* src, float* dst, int n) { add_reduce_entry(tab, src, dst, n, nchunks); };
  add(p.dw_part, L.dw, L.Cout * L.ks * L.ks * L.Cin);
  if (L.db) add(p.db_part, L.db, L.Cout);
}

hipError_t launch_dw(const Layer& L, int B, const float* dy, const float* dy2, bool mask, float* part, ReduceTab* tab,
                     hipStream_t s) {
  const int ocb = (L.Cout + 15) / 16, icb = (L.Cin + 15) / 16;
  const bool first = first_layer_shape(L) && !dy2 && mask && L.db;
  const PartPlan pl = first ? part_plan_first(B, L.Hout) : part_plan(B, L.Hout, ocb * icb);
  if (first) {
    DwFirstParams f{};
    const size_t nw1 = (size_t)L.Cout * 27;
    f.x = L.x; f.dy = dy; f.ymask = L.y; f.dw_part = part; f.db_part = part + (size_t)pl.nchunks * nw1;
    f.Hin = f.Win = L.Hin; f.Hout = f.Wout = L.Hout; f.Cout = L.Cout;
    f.pad_top = f.pad_left = same_pad_before(L.Hin, L.Hout, 3, 2);
    f.tiles_x = pl.tiles_x; f.tiles_per_image = pl.tpi; f.tiles_total = pl.total; f.tiles_per_chunk = pl.tpc;
    f.tx_mul = magic32(pl.tiles_x); f.tpi_mul = magic32(pl.tpi); f.cout_mul = magic32(L.Cout);
    coeff_conv_dw_first<<<dim3((unsigned)pl.nchunks), 256, 0, s>>>(f);
    auto add1 = [&](const float* src, float* dst, int n) { add_reduce_entry(tab, src, dst, n, pl.nchunks); };
    add1(f.dw_part, L.dw, (int)nw1);
    add1(f.db_part, L.db, L.Cout);
    return hipGetLastError();
  }
  const DwParams p = make_dw(L, B, dy, dy2, mask, part, pl, icb);
  const dim3 grid((unsigned)pl.nchunks, (unsigned)(ocb * icb));
  if (L.ks == 3) coeff_conv_dw<3><<<grid, 256, dw_lds_floats<3>() * sizeof(float), s>>>(p);
  else coeff_conv_dw<1><<<grid, 256, dw_lds_floats<1>() * sizeof(float), s>>>(p);
  add_reduce(tab, p, L, pl.nchunks);
  return hipGetLastError();
}

struct DxSetup {
  DxParams p;
  size_t lds;
};

DxSetup make_dx(const Layer& L, const float* dy, const float* dy2, bool mask, float* dx, const float* xmask = nullptr,
               float* colsum_part = nullptr) {
  DxSetup su{};
  DxParams& p = su.p;
  p.dy = dy; p.dy2 = dy2; p.ymask = mask ? L.y : nullptr; p.w = L.w; p.dx = dx;
  p.Hy = p.Wy = L.Hout; p.Cy = L.Cout; p.Hx = p.Wx = L.Hin; p.Cx = L.Cin;
  p.ups = L.stride;
  const int pad_f = same_pad_before(L.Hin, L.Hout, L.ks, L.stride);
  p.pad_top = p.pad_left = L.ks - 1 - pad_f;
  p.tiles_x = (L.Hin + kT - 1) / kT;
  p.tiles = p.tiles_x * p.tiles_x;
  p.oc_groups = (L.Cin + 15) / 16;
  p.tpb = 1;
  p.tile_blocks = p.tiles;
  const int ti = kT - 1 + L.ks;
  p.ti_mul = magic32(ti);
  p.tx_mul = magic32(p.tiles_x);
  const int cmax = L.Cout < kChunkCh ? L.Cout : kChunkCh;  // channels of the widest chunk
  p.c4shift = 0;
  while ((4 << p.c4shift) < cmax) ++p.c4shift;  // staged pixel = a power-of-two number of float4 (48 channels: 16)
  p.nchunks = (L.Cout + kChunkCh - 1) / kChunkCh;  // the last chunk may be partial (a 96-channel prediction layer)
  const int ps = (4 << p.c4shift) + 4, g16 = (cmax + 15) / 16, kk = L.ks * L.ks;
  const int nsteps = kk * g16;
  for (int wv = 0; wv < 4; ++wv) {
    const int s0 = (wv * nsteps) / 4, s1 = ((wv + 1) * nsteps) / 4;
    for (int k = 0; k < 12; ++k) p.lds_off[wv][k] = p.w_off[wv][k] = 0;
    for (int st = s0; st < s1; ++st) {
      const int tap = st / g16, grp = st - tap * g16;
      const int ky = tap / L.ks, kx = tap - ky * L.ks;
      p.lds_off[wv][st - s0] = (unsigned)((ky * ti + kx) * ps + 16 * grp) | ((unsigned)grp << 24);
      p.w_off[wv][st - s0] = (unsigned)((16 * grp * kk + (kk - 1 - tap)) * L.Cin);
    }
    p.lds_off[wv][9] = (unsigned)(s1 - s0);
  }
  su.lds = ((size_t)ti * ti * ps + 4 * 4 * 64) * sizeof(float);
  p.xmask = xmask;
  p.colsum_part = colsum_part;
  return su;
}

// More than ~4 workgroups per CU of a few microseconds each is a launch bound by workgroup turnover (the second splat
// layer's backward-data at 4 x 128 x 128: 4096 workgroups, 20 of the pair's 26 us): such layers take runs of tiles.
void plan_dx_tiles(DxParams* p, int B) {
  p->tpb = 1;
  while (p->tpb < 8 && (long long)((p->tiles + p->tpb - 1) / p->tpb) * p->oc_groups * B > 1024) p->tpb *= 2;
  p->tile_blocks = (p->tiles + p->tpb - 1) / p->tpb;
}

hipError_t launch_dx(const Layer& L, int B, const float* dy, const float* dy2, bool mask, float* dx, hipStream_t s) {
  DxSetup su = make_dx(L, dy, dy2, mask, dx);
  plan_dx_tiles(&su.p, B);
  const dim3 grid((unsigned)su.p.tile_blocks, (unsigned)su.p.oc_groups, (unsigned)B);
  const bool multi = su.p.nchunks > 1;
  if (L.ks == 3 && multi) coeff_conv_dx<3, true><<<grid, 256, su.lds, s>>>(su.p);
  else if (L.ks == 3) coeff_conv_dx<3, false><<<grid, 256, su.lds, s>>>(su.p);
  else if (multi) coeff_conv_dx<1, true><<<grid, 256, su.lds, s>>>(su.p);
  else coeff_conv_dx<1, false><<<grid, 256, su.lds, s>>>(su.p);
  return hipGetLastError();
}

// Backward-weights and backward-data of one layer in ONE launch (coeff_conv_bwd).
struct PairSetup {
  BwdPair pr;
  size_t lds;
  unsigned blocks;
  int ks;
  bool multi;
};

PairSetup make_pair(const Layer& L, int B, const float* dy, const float* dy2, bool mask, float* dx, float* part,
                    ReduceTab* tab, const float* xmask = nullptr, float* colsum_part = nullptr) {
  const int ocb = (L.Cout + 15) / 16, icb = (L.Cin + 15) / 16;
  const PartPlan pl = part_plan(B, L.Hout, ocb * icb);
  PairSetup ps{};
  BwdPair& pr = ps.pr;
  pr.dw = make_dw(L, B, dy, dy2, mask, part, pl, icb);
  DxSetup su = make_dx(L, dy, dy2, mask, dx, xmask, colsum_part);
  plan_dx_tiles(&su.p, B);
  pr.dx = su.p;
  pr.dw_chunks = pl.nchunks;
  pr.dw_blocks = pl.nchunks * ocb * icb;
  pr.dx_tiles = su.p.tile_blocks;
  pr.dx_groups = su.p.oc_groups;
  pr.chunk_mul = magic32(pr.dw_chunks);
  pr.tile_mul = magic32(pr.dx_tiles);
  pr.tg_mul = magic32(pr.dx_tiles * pr.dx_groups);
  ps.blocks = (unsigned)pr.dw_blocks + (unsigned)(pr.dx_tiles * pr.dx_groups * B);
  ps.ks = L.ks;
  ps.multi = su.p.nchunks > 1;
  const size_t dwl = (L.ks == 3 ? dw_lds_floats<3>() : dw_lds_floats<1>()) * sizeof(float);
  ps.lds = su.lds > dwl ? su.lds : dwl;
  add_reduce(tab, pr.dw, L, pl.nchunks);
  return ps;
}

hipError_t launch_setup(const PairSetup& ps, hipStream_t s) {
  if (ps.ks == 3) {
    if (ps.multi) coeff_conv_bwd<3, true><<<dim3(ps.blocks), 256, ps.lds, s>>>(ps.pr);
    else coeff_conv_bwd<3, false><<<dim3(ps.blocks), 256, ps.lds, s>>>(ps.pr);
  } else {
    if (ps.multi) coeff_conv_bwd<1, true><<<dim3(ps.blocks), 256, ps.lds, s>>>(ps.pr);
    else coeff_conv_bwd<1, false><<<dim3(ps.blocks), 256, ps.lds, s>>>(ps.pr);
  }
  return hipGetLastError();
}

hipError_t launch_pair(const Layer& L, int B, const float* dy, const float* dy2, bool mask, float* dx, float* part,
                       ReduceTab* tab, hipStream_t s, const float* xmask = nullptr, float* colsum_part = nullptr) {
  return launch_setup(make_pair(L, B, dy, dy2, mask, dx, part, tab, xmask, colsum_part), s);
}

// Two independent 3 x 3 layers' pairs in one launch (coeff_conv_bwd2); one after the other if their shapes differ in kind.
hipError_t launch_two(const PairSetup& a, const PairSetup& b, hipStream_t s) {
  if (a.ks != 3 || b.ks != 3 || a.multi != b.multi) {
    const hipError_t e = launch_setup(a, s);
    return e != hipSuccess ? e : launch_setup(b, s);
  }
  BwdTwo two{};
  two.p[0] = a.pr;
  two.p[1] = b.pr;
  two.first_blocks = (int)a.blocks;
  const size_t lds = a.lds > b.lds ? a.lds : b.lds;
  if (a.multi) coeff_conv_bwd2<3, true><<<dim3(a.blocks + b.blocks), 256, lds, s>>>(two);
  else coeff_conv_bwd2<3, false><<<dim3(a.blocks + b.blocks), 256, lds, s>>>(two);
  return hipGetLastError();
}

struct BwdSpace {  // float offsets into the backward workspace
  size_t x1, x2, g, fusion, dyp, df, dg, dgp, dx2, dx1, dg2, dl1, dg1, ds4a, ds4b, ds[8], parts, total;
};

BwdSpace bwd_space(const NetDims& d, const hdrnet_coeff_net& net, int B) {
  BwdSpace w{};
  size_t off = 0;
  auto take = [&](size_t n) { const size_t o = off; off += (n + 3) & ~(size_t)3; return o; };
  const size_t P = (size_t)d.sb * d.sb;
  w.x1 = take((size_t)B * 4 * d.gl);
  w.x2 = take((size_t)B * 2 * d.gl);
  w.g = take((size_t)B * d.gl);
  w.fusion = take(B * P * d.gl);
  w.dyp = take(B * P * d.pred);
  w.df = take(B * P * d.gl);
  w.dg = take((size_t)B * d.gl);
  w.dgp = take((size_t)B * ((d.sb + kT - 1) / kT) * ((d.sb + kT - 1) / kT) * d.gl);
  w.dx2 = take((size_t)B * 2 * d.gl);
  w.dx1 = take((size_t)B * 4 * d.gl);
  const int g1side = (d.sb + 1) / 2;
  w.dg2 = take((size_t)B * d.gside * d.gside * d.gl);
  w.dl1 = take(B * P * d.gl);
  w.dg1 = take((size_t)B * g1side * g1side * d.gl);
  w.ds4a = take(B * P * d.feat);
  w.ds4b = take(B * P * d.feat);
  int side = d.N;
  for (int i = 0; i + 1 < d.n_ds; ++i) {  // gradients of splat outputs 0 .. n_ds - 2
    side /= 2;
    w.ds[i] = take((size_t)B * side * side * ((d.cm * d.gd) << i));
  }
  // partial sums of every convolution's weight / bias gradient
  size_t parts = 0;
  side = d.N;
  int cin = 3;
  for (int i = 0; i < d.n_ds; ++i) {
    Layer L{nullptr, nullptr, nullptr, nullptr, (float*)1, side, cin, side / 2, (d.cm * d.gd) << i, 3, 2};
    parts += dw_part_floats(B, L);
    side /= 2;
    cin = L.Cout;
  }
  Layer l1{nullptr, nullptr, nullptr, nullptr, (float*)1, d.sb, d.feat, d.sb, d.gl, 3, 1};
  Layer l2{nullptr, nullptr, nullptr, nullptr, nullptr, d.sb, d.gl, d.sb, d.gl, 3, 1};
  Layer c1{nullptr, nullptr, nullptr, nullptr, (float*)1, d.sb, d.feat, g1side, d.gl, 3, 2};
  Layer c2{nullptr, nullptr, nullptr, nullptr, (float*)1, g1side, d.gl, d.gside, d.gl, 3, 2};
  Layer pr{nullptr, nullptr, nullptr, nullptr, (float*)1, d.sb, d.gl, d.sb, d.pred, 1, 1};
  parts += dw_part_floats(B, l1) + dw_part_floats(B, l2) + dw_part_floats(B, c1) + dw_part_floats(B, c2) + dw_part_floats(B, pr);
  (void)net;
  w.parts = take(parts);
  w.total = off;
  return w;
}

bool train_supported(const hdrnet_coeff_net& net, int B, NetDims* d) {
  if (!net_dims(net, d)) return false;
  if (B < 1 || B > kMaxB || net.n_levels != 1 || net.fc_layout != 1) return false;
  if (d->gl > 256) return false;  // coeff_recompute's shared arrays
  return true;
}

}  // namespace

size_t coefficients_grad_workspace_bytes(const hdrnet_coeff_net& net, int B) {
  NetDims d;
  if (!train_supported(net, B, &d)) return 0;
  return bwd_space(d, net, B).total * sizeof(float);
}

// `fwd_ws`: the workspace a forward launch_coefficients() call with the same net and B left behind.
hipError_t launch_coefficients_grad(const float* lowres, const hdrnet_coeff_net& net, const hdrnet_coeff_net_grads& gr,
                                    const float* dcoeffs, int B, const void* fwd_ws, void* workspace, hipStream_t s,
                                    const char** name) {
  NetDims d;
  if (!train_supported(net, B, &d)) return hipErrorInvalidValue;
  *name = "coeff_net_grad";
  const NetWorkspace fw = net_workspace(d);
  const float* fbase = static_cast<const float*>(fwd_ws);
  auto fbuf = [&](size_t off) { return fbase + off * (size_t)B; };
  const BwdSpace bs = bwd_space(d, net, B);
  float* base = static_cast<float*>(workspace);
  auto buf = [&](size_t off) { return base + off; };
  const int P = d.sb * d.sb, g1side = (d.sb + 1) / 2;
  const int K1 = d.gside * d.gside * d.gl;
  hipError_t e;

  // the forward's activations
  const float* S[8];
  for (int i = 0; i < d.n_ds; ++i) S[i] = fbuf(fw.splat[i]);
  const float* L1 = fbuf(fw.local1);
  const float* L2 = fbuf(fw.local2);
  const float* G1 = fbuf(fw.g1);
  const float* G2 = fbuf(fw.g2);

  // ---- what the forward did not keep + the incoming gradient in the prediction layer's channel order
  {
    RecomputeParams p{fbuf(fw.fc1), fw.s1, fbuf(fw.fc2), fw.s2, net.fc_b[0], net.fc_b[1], net.fc_w[2], net.fc_b[2], L2,
                      dcoeffs, buf(bs.x1), buf(bs.x2), buf(bs.g), buf(bs.fusion), buf(bs.dyp),
                      4 * d.gl, 2 * d.gl, d.gl, P, d.gd, net.n_out, net.n_in,
                      magic32(d.gd * net.n_out * net.n_in), magic32(d.gd), magic32(net.n_out)};
    int slabs = kRecomputeSlabs;
    while (slabs > 1 && ((P + slabs - 1) / slabs) * (d.gd * net.n_out * net.n_in) >= 65536) slabs *= 2;  // (never: see the kernel)
    coeff_recompute<<<dim3((unsigned)B, (unsigned)slabs + 1), 256, 0, s>>>(p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  ReduceTab tab{};
  tab.count = 0;
  tab.first[0] = 0;
  float* parts = buf(bs.parts);
  auto dw = [&](const Layer& L, const float* dy, const float* dy2, bool mask) -> hipError_t {
    const hipError_t r = launch_dw(L, B, dy, dy2, mask, parts, &tab, s);
    parts += dw_part_floats(B, L);
    return r;
  };
  auto pair = [&](const Layer& L, const float* dy, const float* dy2, bool mask, float* dx) -> hipError_t {
    const hipError_t r = launch_pair(L, B, dy, dy2, mask, dx, parts, &tab, s);
    parts += dw_part_floats(B, L);
    return r;
  };
  // ---- prediction layer (1x1 on the fusion; its own output has no ReLU)
  const Layer pr{buf(bs.fusion), nullptr, net.pred_w, gr.pred_w, gr.pred_b, d.sb, d.gl, d.sb, d.pred, 1, 1};
  // (its backward-data half also leaves the per-tile column sums of df where the fusion passed it: dg's partial sums)
  const int ntile = ((d.sb + kT - 1) / kT) * ((d.sb + kT - 1) / kT);
  e = launch_pair(pr, B, buf(bs.dyp), nullptr, false, buf(bs.df), parts, &tab, s, buf(bs.fusion), buf(bs.dgp));
  parts += dw_part_floats(B, pr);
  if (e != hipSuccess) return e;
  // ---- fusion = relu(local2 + g): d local2 = df masked (applied by the consumers), dg = its sum over the cells
  // (summed inside fc3's backward instead -- 16 partial sums per element while staging dy -- the launch was 17 us slower:
  // the extra loads sit on every workgroup's critical path; a launch of its own costs 4.9)
  coeff_slab_sum<<<dim3((unsigned)B), 256, 0, s>>>(buf(bs.dgp), buf(bs.dg), ntile, d.gl);
  // ---- fully connected layers
  {
    FcBwdParams f3{buf(bs.x2), buf(bs.dg), net.fc_w[2], gr.fc_w[2], gr.fc_b[2], buf(bs.dx2), B, 2 * d.gl, d.gl, 1};
    coeff_fc_bwd<<<dim3((unsigned)((2 * d.gl + 15) / 16)), 256, 0, s>>>(f3);
    FcBwdParams f2{buf(bs.x1), buf(bs.dx2), net.fc_w[1], gr.fc_w[1], gr.fc_b[1], buf(bs.dx1), B, 4 * d.gl, 2 * d.gl, 1};
    coeff_fc_bwd<<<dim3((unsigned)((4 * d.gl + 15) / 16)), 256, 0, s>>>(f2);
    FcBwdParams f1{G2, buf(bs.dx1), net.fc_w[0], gr.fc_w[0], gr.fc_b[0], buf(bs.dg2), B, K1, 4 * d.gl, 0};
    coeff_fc_bwd<<<dim3((unsigned)((K1 + 15) / 16)), 256, 0, s>>>(f1);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  // ---- the local path (local2: no bias, no ReLU on its own output -- the fusion's mask --, local1) and the global path's
  // convolutions (conv2, conv1) do not depend on each other: local2 + conv2 in one launch, then local1 + conv1
  const float* feat = S[d.n_ds - 1];
  {
    const Layer l2{L1, buf(bs.fusion), net.local_w[1], gr.local_w[1], nullptr, d.sb, d.gl, d.sb, d.gl, 3, 1};
    const Layer c2{G1, G2, net.global_conv_w[1], gr.global_conv_w[1], gr.global_conv_b[1], g1side, d.gl, d.gside, d.gl, 3, 2};
    const PairSetup a = make_pair(l2, B, buf(bs.df), nullptr, true, buf(bs.dl1), parts, &tab);
    parts += dw_part_floats(B, l2);
    const PairSetup b = make_pair(c2, B, buf(bs.dg2), nullptr, true, buf(bs.dg1), parts, &tab);
    parts += dw_part_floats(B, c2);
    if ((e = launch_two(a, b, s)) != hipSuccess) return e;
  }
  {
    const Layer l1{feat, L1, net.local_w[0], gr.local_w[0], gr.local_b[0], d.sb, d.feat, d.sb, d.gl, 3, 1};
    const Layer c1{feat, G1, net.global_conv_w[0], gr.global_conv_w[0], gr.global_conv_b[0], d.sb, d.feat, g1side, d.gl, 3, 2};
    const PairSetup a = make_pair(l1, B, buf(bs.dl1), nullptr, true, buf(bs.ds4a), parts, &tab);
    parts += dw_part_floats(B, l1);
    const PairSetup b = make_pair(c1, B, buf(bs.dg1), nullptr, true, buf(bs.ds4b), parts, &tab);
    parts += dw_part_floats(B, c1);
    if ((e = launch_two(a, b, s)) != hipSuccess) return e;
  }
  // ---- splat, last to first; the last layer's gradient is the sum of the two paths'
  const float* dy = buf(bs.ds4a);
  const float* dy2 = buf(bs.ds4b);
  for (int i = d.n_ds - 1; i >= 0; --i) {
    const int cout = (d.cm * d.gd) << i, cin = i > 0 ? (d.cm * d.gd) << (i - 1) : 3;
    const int hin = d.N >> i;
    const Layer L{i > 0 ? S[i - 1] : lowres, S[i], net.splat_w[i], gr.splat_w[i], gr.splat_b[i], hin, cin, hin / 2, cout, 3, 2};
    if (i > 0) {
      if ((e = pair(L, dy, dy2, true, buf(bs.ds[i - 1]))) != hipSuccess) return e;
      dy = buf(bs.ds[i - 1]);
      dy2 = nullptr;
    } else if ((e = dw(L, dy, dy2, true)) != hipSuccess) {
      return e;
    }
  }
  // ---- the chunks' partial sums of every weight / bias gradient
  coeff_reduce_parts<<<dim3((unsigned)tab.first[tab.count]), 256, 0, s>>>(tab);
  return hipGetLastError();
}

}  // namespace hdrnet_amd
