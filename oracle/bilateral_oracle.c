/* TEST INFRASTRUCTURE ONLY (oracle/). Not part of the shipped product path.
 *
 * CPU restatement (plain C99) of the reference's BilateralSlice /
 * BilateralSliceApply forward and VJPs.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this; the product path
 * (hdrnet_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_pinning.py checks every function
 * here bit-for-bit against oracle/_ref/libhdrnet_ref.so (the reference's own
 * .cc files compiled unchanged, see oracle/Makefile) when that library is
 * present, and against tests/golden/ fixtures generated from it
 * (tests/golden/make_golden.py) everywhere else, plus the reference's
 * known-answer test (hdrnet/test/ops_test.py:61-86).
 *
 * Layouts are the TF/NHWC ones of the op wrappers
 * (hdrnet/ops/bilateral_slice_apply_op.cc:201-227):
 *   grid  [B][GH][GW][GD][Cout][Cj]   (channel c = i*Cj + j)
 *   guide [B][H][W]
 *   input [B][H][W][Cin]
 *   out   [B][H][W][Cout]
 * Arithmetic order follows the reference statement by statement so that a
 * non-contracting build (-ffp-contract=off) is bit-identical to it.
 */
#include <math.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* std::max / std::clamp semantics (not fmaxf: NaN propagation differs). */
#define STD_MAX(a, b) (((a) < (b)) ? (b) : (a))
static inline int clampi(int v, int lo, int hi) {
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}

/* hdrnet/ops/numerics.h:53-57  LerpWeight */
static inline float lerp_weight(float x, float xs) {
  const float dx = x - xs;
  const float abs_dx = fabsf(dx);
  return STD_MAX(1.0f - abs_dx, 0.0f);
}

/* hdrnet/ops/numerics.h:72-80  MirrorBoundary */
static inline int mirror_boundary(int x, int extent) {
  if (x < 0) return -x - 1;
  if (x >= extent) return 2 * extent - 1 - x;
  return x;
}

/* hdrnet/ops/numerics.h:83-85  SmoothedAbs (eps = 1e-8) */
static inline float smoothed_abs(float x) { return sqrtf(x * x + 1.0e-8f); }

/* hdrnet/ops/numerics.h:89-91  SmoothedAbsGrad */
static inline float smoothed_abs_grad(float x) {
  return x / sqrtf(x * x + 1.0e-8f);
}

/* hdrnet/ops/numerics.h:108-113  SmoothedLerpWeight */
static inline float smoothed_lerp_weight(float x, float xs) {
  const float dx = x - xs;
  const float abs_dx = smoothed_abs(dx);
  return STD_MAX(1.0f - abs_dx, 0.0f);
}

/* hdrnet/ops/numerics.h:116-126  SmoothedLerpWeightGrad */
static inline float smoothed_lerp_weight_grad(float x, float xs) {
  const float dx = x - xs;
  const float abs_dx = smoothed_abs(dx);
  if (abs_dx > 1.0f) return 0.0f;
  return smoothed_abs_grad(dx);
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

#define GRID6(b, gy, gx, gz, i, j)                                            \
  grid[((((((size_t)(b) * GH + (gy)) * GW + (gx)) * GD + (gz)) * Cout + (i)) * \
        Cj) + (j)]
#define GRID5(b, gy, gx, gz, c) \
  grid[(((((size_t)(b) * GH + (gy)) * GW + (gx)) * GD + (gz)) * C) + (c)]
#define PIX(b, y, x) (((size_t)(b) * H + (y)) * W + (x))

/* hdrnet/ops/bilateral_slice_apply.cc:24-82  BilateralSliceApply */
/* The forward on rows y0 .. y0 + H - 1 of frames that are H_total rows high: the buffers hold the band
 * only; scale_y (:38) and gyf (:42) are the reference's expressions on the FRAME.  Whole frame:
 * y0 = 0, H_total = H.  (Checker for hdrnet_bilateral_slice_apply_rows_f32; test infrastructure.) */
void oracle_bilateral_slice_apply_rows(const float* grid, const float* guide,
                                       const float* input, float* out, int B, int H_total, int y0,
                                       int H, int W, int GH, int GW, int GD, int Cin,
                                       int Cout, int has_offset) {
  const int Cj = Cin + (has_offset ? 1 : 0);
  const float scale_x = (float)GW / W;       /* :37 */
  const float scale_y = (float)GH / H_total; /* :38 */
  long long by;
#pragma omp parallel for schedule(static)
  for (by = 0; by < (long long)B * H; ++by) {
    const int b = (int)(by / H), y = (int)(by % H);
    for (int x = 0; x < W; ++x) {
      const float gxf = (x + 0.5f) * scale_x;         /* :41 */
      const float gyf = ((y0 + y) + 0.5f) * scale_y;  /* :42 */
      const float gzf = guide[PIX(b, y, x)] * GD;     /* :44 */
      const int gx0 = (int)floorf(gxf - 0.5f);        /* :46 */
      const int gy0 = (int)floorf(gyf - 0.5f);        /* :47 */
      const int gz0 = (int)floorf(gzf - 0.5f);        /* :48 */
      for (int i = 0; i < Cout; ++i) {
        float value = 0.0f;
        for (int j = 0; j < Cj; ++j) {
          float grid_sample = 0.0f;
          for (int gy = gy0; gy < gy0 + 2; ++gy) { /* :54-69 */
            const int gyc = clampi(gy, 0, GH - 1);
            const float wy = lerp_weight(gy + 0.5f, gyf);
            for (int gx = gx0; gx < gx0 + 2; ++gx) {
              const int gxc = clampi(gx, 0, GW - 1);
              const float wx = lerp_weight(gx + 0.5f, gxf);
              for (int gz = gz0; gz < gz0 + 2; ++gz) {
                const int gzc = clampi(gz, 0, GD - 1);
                const float wz = smoothed_lerp_weight(gz + 0.5f, gzf);
                grid_sample += wx * wy * wz * GRID6(b, gyc, gxc, gzc, i, j);
              }
            }
          }
          if (j < Cin) { /* :73-77 */
            value += grid_sample * input[PIX(b, y, x) * Cin + j];
          } else {
            value += grid_sample;
          }
        }
        out[PIX(b, y, x) * Cout + i] = value; /* :80 */
      }
    }
  }
}

void oracle_bilateral_slice_apply(const float* grid, const float* guide,
                                  const float* input, float* out, int B, int H,
                                  int W, int GH, int GW, int GD, int Cin,
                                  int Cout, int has_offset) {
  oracle_bilateral_slice_apply_rows(grid, guide, input, out, B, H, 0, H, W, GH, GW, GD, Cin, Cout,
                                    has_offset);
}

/* hdrnet/ops/bilateral_slice_apply.cc:84-138  BilateralSliceApplyGridGrad */
void oracle_bilateral_slice_apply_grid_grad(const float* guide,
                                            const float* input,
                                            const float* dout, float* dgrid,
                                            int B, int H, int W, int GH, int GW,
                                            int GD, int Cin, int Cout,
                                            int has_offset) {
  const int Cj = Cin + (has_offset ? 1 : 0);
  const float scale_x = (float)W / GW; /* :95 */
  const float scale_y = (float)H / GH; /* :96 */
  long long cell;
#pragma omp parallel for schedule(static)
  for (cell = 0; cell < (long long)B * GH * GW; ++cell) {
    const int gx = (int)(cell % GW);
    const int gy = (int)((cell / GW) % GH);
    const int b = (int)(cell / ((long long)GW * GH));
    const int x0 = (int)floorf(scale_x * (gx + 0.5f - 1.0f)); /* :100 */
    const int x1 = (int)ceilf(scale_x * (gx + 0.5f + 1.0f));  /* :101-102 */
    const int y0 = (int)floorf(scale_y * (gy + 0.5f - 1.0f)); /* :103 */
    const int y1 = (int)ceilf(scale_y * (gy + 0.5f + 1.0f));  /* :104-105 */
    for (int gz = 0; gz < GD; ++gz) {
      for (int i = 0; i < Cout; ++i) {
        for (int j = 0; j < Cj; ++j) {
          float vjp_value = 0.0f;
          for (int y = y0; y < y1; ++y) {
            const int ym = mirror_boundary(y, H);        /* :109 */
            const float gyf = (y + 0.5f) / scale_y;      /* :110 */
            const float wy = lerp_weight(gy + 0.5f, gyf); /* :111 */
            for (int x = x0; x < x1; ++x) {
              const int xm = mirror_boundary(x, W);        /* :115 */
              const float gxf = (x + 0.5f) / scale_x;      /* :116 */
              const float wx = lerp_weight(gx + 0.5f, gxf); /* :117 */
              const float gzf = guide[PIX(b, ym, xm)] * GD; /* :120 */
              float wz = smoothed_lerp_weight(gz + 0.5f, gzf);
              if ((gz == 0 && gzf < 0.5f) ||
                  (gz == GD - 1 && gzf > GD - 0.5f)) { /* :122-125 */
                wz = 1.0f;
              }
              const float input_value =
                  (j < Cin) ? input[PIX(b, ym, xm) * Cin + j] : 1.0f; /* :128 */
              const float grad_value = wx * wy * wz * input_value;    /* :130 */
              vjp_value += grad_value * dout[PIX(b, ym, xm) * Cout + i];
            }
          }
          dgrid[((((((size_t)b * GH + gy) * GW + gx) * GD + gz) * Cout + i) *
                 Cj) + j] = vjp_value; /* :136 */
        }
      }
    }
  }
}

/* hdrnet/ops/bilateral_slice_apply.cc:140-206  BilateralSliceApplyGuideGrad */
void oracle_bilateral_slice_apply_guide_grad(const float* grid,
                                             const float* guide,
                                             const float* input,
                                             const float* dout, float* dguide,
                                             int B, int H, int W, int GH,
                                             int GW, int GD, int Cin, int Cout,
                                             int has_offset) {
  const int Cj = Cin + (has_offset ? 1 : 0);
  const float scale_x = (float)GW / W; /* :154 */
  const float scale_y = (float)GH / H; /* :155 */
  long long by;
#pragma omp parallel for schedule(static)
  for (by = 0; by < (long long)B * H; ++by) {
    const int b = (int)(by / H), y = (int)(by % H);
    for (int x = 0; x < W; ++x) {
      const float gxf = (x + 0.5f) * scale_x;
      const float gyf = (y + 0.5f) * scale_y;
      const float gzf = guide[PIX(b, y, x)] * GD; /* :161 */
      const int gx0 = (int)floorf(gxf - 0.5f);
      const int gy0 = (int)floorf(gyf - 0.5f);
      const int gz0 = (int)floorf(gzf - 0.5f);
      float vjp_value = 0.0f;
      for (int i = 0; i < Cout; ++i) {
        float grad_value = 0.0f;
        for (int j = 0; j < Cj; ++j) {
          float grid_sample = 0.0f;
          for (int gy = gy0; gy < gy0 + 2; ++gy) { /* :175-192 */
            const int gyc = clampi(gy, 0, GH - 1);
            const float wy = lerp_weight(gy + 0.5f, gyf);
            for (int gx = gx0; gx < gx0 + 2; ++gx) {
              const int gxc = clampi(gx, 0, GW - 1);
              const float wx = lerp_weight(gx + 0.5f, gxf);
              for (int gz = gz0; gz < gz0 + 2; ++gz) {
                const int gzc = clampi(gz, 0, GD - 1);
                const float dwz =
                    GD * smoothed_lerp_weight_grad(gz + 0.5f, gzf); /* :186 */
                grid_sample += wx * wy * dwz * GRID6(b, gyc, gxc, gzc, i, j);
              }
            }
          }
          const float input_value =
              (j < Cin) ? input[PIX(b, y, x) * Cin + j] : 1.0f; /* :196 */
          grad_value += grid_sample * input_value;              /* :198 */
        }
        vjp_value += grad_value * dout[PIX(b, y, x) * Cout + i]; /* :201 */
      }
      dguide[PIX(b, y, x)] = vjp_value; /* :204 */
    }
  }
}

/* hdrnet/ops/bilateral_slice_apply.cc:208-259  BilateralSliceApplyInputGrad */
void oracle_bilateral_slice_apply_input_grad(const float* grid,
                                             const float* guide,
                                             const float* dout, float* dinput,
                                             int B, int H, int W, int GH,
                                             int GW, int GD, int Cin, int Cout,
                                             int has_offset) {
  const int Cj = Cin + (has_offset ? 1 : 0);
  const float scale_x = (float)GW / W; /* :219 */
  const float scale_y = (float)GH / H; /* :220 */
  long long by;
#pragma omp parallel for schedule(static)
  for (by = 0; by < (long long)B * H; ++by) {
    const int b = (int)(by / H), y = (int)(by % H);
    for (int x = 0; x < W; ++x) {
      const float gxf = (x + 0.5f) * scale_x;
      const float gyf = (y + 0.5f) * scale_y;
      const float gzf = guide[PIX(b, y, x)] * GD; /* :226 */
      const int gx0 = (int)floorf(gxf - 0.5f);
      const int gy0 = (int)floorf(gyf - 0.5f);
      const int gz0 = (int)floorf(gzf - 0.5f);
      for (int j = 0; j < Cin; ++j) {
        float vjp_value = 0.0f;
        for (int i = 0; i < Cout; ++i) {
          float grad_value = 0.0f;
          for (int gy = gy0; gy < gy0 + 2; ++gy) { /* :236-251 */
            const int gyc = clampi(gy, 0, GH - 1);
            const float wy = lerp_weight(gy + 0.5f, gyf);
            for (int gx = gx0; gx < gx0 + 2; ++gx) {
              const int gxc = clampi(gx, 0, GW - 1);
              const float wx = lerp_weight(gx + 0.5f, gxf);
              for (int gz = gz0; gz < gz0 + 2; ++gz) {
                const int gzc = clampi(gz, 0, GD - 1);
                const float wz = smoothed_lerp_weight(gz + 0.5f, gzf);
                grad_value += wx * wy * wz * GRID6(b, gyc, gxc, gzc, i, j);
              }
            }
          }
          vjp_value += grad_value * dout[PIX(b, y, x) * Cout + i]; /* :254 */
        }
        dinput[PIX(b, y, x) * Cin + j] = vjp_value; /* :257 */
      }
    }
  }
}

/* hdrnet/ops/bilateral_slice.cc:25-70  BilateralSlice */
void oracle_bilateral_slice(const float* grid, const float* guide, float* out,
                            int B, int H, int W, int GH, int GW, int GD,
                            int C) {
  const float scale_x = (float)GW / W; /* :33 */
  const float scale_y = (float)GH / H; /* :34 */
  long long by;
#pragma omp parallel for schedule(static)
  for (by = 0; by < (long long)B * H; ++by) {
    const int b = (int)(by / H), y = (int)(by % H);
    for (int x = 0; x < W; ++x) {
      const float gxf = (x + 0.5f) * scale_x;
      const float gyf = (y + 0.5f) * scale_y;
      const float gzf = guide[PIX(b, y, x)] * GD; /* :42 */
      const int gx0 = (int)floorf(gxf - 0.5f);
      const int gy0 = (int)floorf(gyf - 0.5f);
      const int gz0 = (int)floorf(gzf - 0.5f);
      for (int c = 0; c < C; ++c) {
        float value = 0.0f;
        for (int gy = gy0; gy < gy0 + 2; ++gy) { /* :50-65 */
          const int gyc = clampi(gy, 0, GH - 1);
          const float wy = lerp_weight(gy + 0.5f, gyf);
          for (int gx = gx0; gx < gx0 + 2; ++gx) {
            const int gxc = clampi(gx, 0, GW - 1);
            const float wx = lerp_weight(gx + 0.5f, gxf);
            for (int gz = gz0; gz < gz0 + 2; ++gz) {
              const int gzc = clampi(gz, 0, GD - 1);
              const float wz = smoothed_lerp_weight(gz + 0.5f, gzf);
              value += wx * wy * wz * GRID5(b, gyc, gxc, gzc, c);
            }
          }
        }
        out[PIX(b, y, x) * C + c] = value; /* :68 */
      }
    }
  }
}

/* hdrnet/ops/bilateral_slice.cc:72-118  BilateralSliceGridGrad */
void oracle_bilateral_slice_grid_grad(const float* guide, const float* dout,
                                      float* dgrid, int B, int H, int W, int GH,
                                      int GW, int GD, int C) {
  const float scale_x = (float)W / GW; /* :81 */
  const float scale_y = (float)H / GH; /* :82 */
  long long cell;
#pragma omp parallel for schedule(static)
  for (cell = 0; cell < (long long)B * GH * GW; ++cell) {
    const int gx = (int)(cell % GW);
    const int gy = (int)((cell / GW) % GH);
    const int b = (int)(cell / ((long long)GW * GH));
    const int x0 = (int)floorf(scale_x * (gx + 0.5f - 1.0f)); /* :86 */
    const int x1 = (int)ceilf(scale_x * (gx + 0.5f + 1.0f));
    const int y0 = (int)floorf(scale_y * (gy + 0.5f - 1.0f));
    const int y1 = (int)ceilf(scale_y * (gy + 0.5f + 1.0f));
    for (int gz = 0; gz < GD; ++gz) {
      for (int c = 0; c < C; ++c) {
        float vjp_value = 0.0f;
        for (int y = y0; y < y1; ++y) {
          const int ym = mirror_boundary(y, H);
          const float gyf = (y + 0.5f) / scale_y;
          const float wy = lerp_weight(gy + 0.5f, gyf);
          for (int x = x0; x < x1; ++x) {
            const int xm = mirror_boundary(x, W);
            const float gxf = (x + 0.5f) / scale_x;
            const float wx = lerp_weight(gx + 0.5f, gxf);
            const float gzf = guide[PIX(b, ym, xm)] * GD; /* :105 */
            float wz = smoothed_lerp_weight(gz + 0.5f, gzf);
            if ((gz == 0 && gzf < 0.5f) ||
                (gz == GD - 1 && gzf > GD - 0.5f)) { /* :107-110 */
              wz = 1.0f;
            }
            /* :112 -- note the operand order wz * wx * wy here. */
            vjp_value += wz * wx * wy * dout[PIX(b, ym, xm) * C + c];
          }
        }
        dgrid[(((((size_t)b * GH + gy) * GW + gx) * GD + gz) * C) + c] =
            vjp_value; /* :116 */
      }
    }
  }
}

/* hdrnet/ops/bilateral_slice.cc:120-168  BilateralSliceGuideGrad */
void oracle_bilateral_slice_guide_grad(const float* grid, const float* guide,
                                       const float* dout, float* dguide, int B,
                                       int H, int W, int GH, int GW, int GD,
                                       int C) {
  const float scale_x = (float)GW / W; /* :129 */
  const float scale_y = (float)GH / H; /* :130 */
  long long by;
#pragma omp parallel for schedule(static)
  for (by = 0; by < (long long)B * H; ++by) {
    const int b = (int)(by / H), y = (int)(by % H);
    for (int x = 0; x < W; ++x) {
      const float gxf = (x + 0.5f) * scale_x;
      const float gyf = (y + 0.5f) * scale_y;
      const float gzf = guide[PIX(b, y, x)] * GD; /* :135 */
      const int gx0 = (int)floorf(gxf - 0.5f);
      const int gy0 = (int)floorf(gyf - 0.5f);
      const int gz0 = (int)floorf(gzf - 0.5f);
      float vjp_value = 0.0f;
      for (int c = 0; c < C; ++c) {
        float grid_sample = 0.0f;
        for (int gy = gy0; gy < gy0 + 2; ++gy) { /* :146-162 */
          const int gyc = clampi(gy, 0, GH - 1);
          const float wy = lerp_weight(gy + 0.5f, gyf);
          for (int gx = gx0; gx < gx0 + 2; ++gx) {
            const int gxc = clampi(gx, 0, GW - 1);
            const float wx = lerp_weight(gx + 0.5f, gxf);
            for (int gz = gz0; gz < gz0 + 2; ++gz) {
              const int gzc = clampi(gz, 0, GD - 1);
              const float dwz =
                  GD * smoothed_lerp_weight_grad(gz + 0.5f, gzf); /* :156 */
              grid_sample += wx * wy * dwz * GRID5(b, gyc, gxc, gzc, c);
            }
          }
        }
        vjp_value += grid_sample * dout[PIX(b, y, x) * C + c]; /* :163 */
      }
      dguide[PIX(b, y, x)] = vjp_value; /* :166 */
    }
  }
}
