#!/usr/bin/env python3
"""Per-workgroup timeline of the coefficient network's launches (tools build, csrc/coeff_net.hip COEFF_STAMP).

    python tools/coeff_trace.py [--model nn|pyramid] [--batch 1]

Stamps (wall_clock64, 10-ns ticks) per workgroup: 0 = start, 1 = after the prediction layer's reduction of the global
features, 2 = first channel chunk staged in LDS (fc: inputs reduced), 3 = arithmetic done / stores issued.  Printed per
launch, relative to the launch's first workgroup start: when the last workgroup started, the median and the last of
each stamp, and the gap since the previous launch's last stamp.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from hdrnet_amd import _lib, models  # noqa: E402

STRIDE = 8 * 8192


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="nn")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--launches", type=int, default=12)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cls = models.HDRNetGaussianPyrNN if args.model == "pyramid" else models.HDRNetPointwiseNNGuide
    m = cls().to(dev).eval()
    w = m.coefficients.exported()
    lib = _lib.load_tools()
    low = torch.rand(args.batch, 256, 256, 3, device=dev)
    L = w.n_levels
    out = torch.empty((L, args.batch, 16, 16, 8, w.n_out // L, w.n_in), device=dev)
    wbytes = lib.hdrnet_coefficients_workspace_bytes(ctypes.byref(w.net), args.batch)
    ws = torch.empty((wbytes,), dtype=torch.uint8, device=dev)
    trace = torch.zeros((args.launches * STRIDE,), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def run():
        rc = lib.hdrnet_coefficients_f32(low.data_ptr(), ctypes.byref(w.net), out.data_ptr(), args.batch, ws.data_ptr(),
                                         wbytes, stream)
        assert rc == 0, lib.hdrnet_last_error()

    for _ in range(20):
        run()
    torch.cuda.synchronize()
    lib.hdrnet_tools_set_trace(trace.data_ptr())
    for _ in range(3):  # the last of three back-to-back passes is the one read
        trace.zero_()
        run()
    torch.cuda.synchronize()
    lib.hdrnet_tools_set_trace(None)
    t = trace.cpu().reshape(args.launches, 8192, 8)
    prev_end = None
    first = None
    for k in range(args.launches):
        rows = t[k][t[k][:, 0] > 0]
        if rows.numel() == 0:
            continue
        t0 = int(rows[:, 0].min())
        first = t0 if first is None else first
        line = f"launch {k}: {rows.shape[0]:5d} workgroups, begins {(t0 - first) / 100:7.2f} us"
        if prev_end is not None:
            line += f" (gap {(t0 - prev_end) / 100:5.2f})"
        line += f"; last start +{(int(rows[:, 0].max()) - t0) / 100:5.2f}"
        for slot, label in ((1, "reduced"), (2, "staged"), (3, "done")):
            v = rows[:, slot][rows[:, slot] > 0]
            if v.numel():
                line += f"; {label} med +{(int(v.median()) - t0) / 100:5.2f} last +{(int(v.max()) - t0) / 100:5.2f}"
        print(line)
        prev_end = int(rows[:, 1:4].max())
    print(f"whole pass: {(prev_end - first) / 100:.2f} us (first workgroup start -> last stamp)")


if __name__ == "__main__":
    main()
