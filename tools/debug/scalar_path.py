#!/usr/bin/env python3
"""How much slower are the scalar row kernels (W % 4 != 0 or buffers 4 B off a 16-B boundary)?"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hdrnet_amd import _lib
dev = torch.device("cuda:0")
lib = _lib.load()
stream = torch.cuda.current_stream(dev).cuda_stream
for (H, W, off) in [(2160, 3840, 0), (2160, 3840, 1), (2160, 3838, 0), (2160, 3841, 0)] * 3:
    S = []
    for _ in range(3):
        def mk(n):
            b = torch.rand(n + 4, device=dev)
            return b[off:off + n]
        S.append((torch.rand(1 * 16 * 16 * 8 * 12, device=dev), mk(H * W), mk(H * W * 3), mk(H * W * 3)))
    def call(k):
        g, gu, i, o = S[k % 3]
        rc = lib.hdrnet_bilateral_slice_apply_f32(g.data_ptr(), gu.data_ptr(), i.data_ptr(), o.data_ptr(),
                                                  1, H, W, 16, 16, 8, 3, 3, 1, stream)
        assert rc == 0
    ts = []
    for r in range(4):
        for k in range(300): call(k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for k in range(200): call(k)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / 200)
    print(f"{W}x{H} offset {4*off} B: {lib.hdrnet_last_kernel().decode():28s} {statistics.median(ts):7.2f} us")
