"""TEST INFRASTRUCTURE ONLY -- a second, independent statement of the two guide functions, held by
the reference itself: its OpenGL fragment shaders.

``oracle.curves_guide`` / ``oracle.pointwise_nn_guide`` restate the TensorFlow graphs
(hdrnet/models.py:145-190, :203-210) on the parameters hdrnet/bin/freeze_graph.py exports.  No
TensorFlow exists in this image, so those restatements cannot be pinned against the graph; but the
reference ships a SECOND implementation of exactly these functions, written against the exported
files: benchmark/assets/std.frag:32-53 (curves guide) and benchmark/assets/gpyrnn.frag:42-63
(point-wise network guide, one per pyramid level), with the uniforms loaded by
benchmark/src/renderer.cc:196-223 and :270-298.  This module transliterates the shaders' guide math
statement by statement -- scalar loops, float32, operating on the FLAT FILE CONTENTS in the order the
renderer uploads them -- so that tests/test_oracle_pinning.py can show the numpy restatements agree
with it on the freeze_graph.py:107-184 layouts.  (Model-graph parity against TensorFlow itself stays
open)

GLSL semantics used: a ``matCxR`` uniform is C columns of R floats, column-major
(glProgramUniformMatrix3x4fv with transpose = GL_FALSE: column c = floats [4c, 4c + 4)); ``vec4 * mat3x4``
is the row-vector product, component c = dot(vec4, column c).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def std_frag_guide(rgb, ccm_file, shifts_file, slopes_file, mix_file):
    """benchmark/assets/std.frag:36-45.  ``rgb``: (3,) one pixel.  Files as float32 arrays exactly as
    renderer.cc:203-223 loads them: guide_ccm_f32_3x4.bin (12), guide_shifts_f32_16x3.bin (48),
    guide_slopes_f32_16x3.bin (48), guide_mix_matrix_f32_1x4.bin (4)."""
    rgba = [F(rgb[0]), F(rgb[1]), F(rgb[2]), F(1.0)]                 # vec4 rgba = vec4(texture(...).xyz, 1.0)
    tmp = [F(0)] * 3
    for c in range(3):                                               # vec3 tmp = rgba * uGuideCcm  (mat3x4)
        acc = F(0)
        for r in range(4):
            acc = F(acc + F(rgba[r] * F(ccm_file[4 * c + r])))
        tmp[c] = acc
    tmp2 = [F(0)] * 3                                                # vec3 tmp2 = vec3(0)
    for i in range(16):                                              # for (int i = 0; i < 16; ++i)
        for c in range(3):                                           #   tmp2 += uGuideSlopes[i] * max(vec3(0), tmp - uGuideShifts[i])
            d = F(tmp[c] - F(shifts_file[3 * i + c]))
            tmp2[c] = F(tmp2[c] + F(F(slopes_file[3 * i + c]) * max(F(0), d)))
    dot = F(0)                                                       # dot(vec4(tmp2, 1.0), uMixMatrix)
    for r, v in enumerate((tmp2[0], tmp2[1], tmp2[2], F(1.0))):
        dot = F(dot + F(v * F(mix_file[r])))
    return F(min(max(dot, F(0)), F(1)))                              # clamp(., 0, 1)


def gpyrnn_frag_guide(rgb, level, conv1_file, conv2_file, literal_bias_index=False):
    """benchmark/assets/gpyrnn.frag:49-63 for pyramid level ``level`` (the shader's index c).
    ``conv1_file``: the renderer's concatenation of guide_level{0,1,2}_conv1.bin (3 x 16 vec4),
    ``conv2_file``: guide_level{0,1,2}_conv2.bin (3 x 17 floats) (renderer.cc:274-295).

    The shader adds ``uGuideConv2[16]`` -- level 0's bias -- at EVERY level (gpyrnn.frag:60); the
    export puts level c's bias at 16 + 17 c.  ``literal_bias_index=True`` reproduces the shader as
    written; the default uses the level's own bias, which is what the TF graph computes (for level 0,
    and for the single-scale HDRNetPointwiseNNGuide export, the two coincide)."""
    c = level
    rgba = [F(rgb[0]), F(rgb[1]), F(rgb[2]), F(1.0)]
    conv1 = [F(0)] * 16
    for i in range(16):                                              # guide_conv1[i + 16 c] = max(dot(rgba[c], uGuideConv1[i + 16 c]), 0)
        acc = F(0)
        for r in range(4):
            acc = F(acc + F(rgba[r] * F(conv1_file[4 * (i + 16 * c) + r])))
        conv1[i] = max(acc, F(0))
    acc = F(0)                                                       # guide_conv2[c] += guide_conv1[i + 16 c] * uGuideConv2[i + 17 c]
    for i in range(16):
        acc = F(acc + F(conv1[i] * F(conv2_file[i + 17 * c])))
    acc = F(acc + F(conv2_file[16 if literal_bias_index else 16 + 17 * c]))
    return F(F(1.0) / F(F(1.0) + F(np.exp(F(-acc)))))                # sigmoid
