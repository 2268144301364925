#!/usr/bin/env python3
"""Parity of the round-5 gradient-pass variants (tools build) against the product pass, on the GPU.

    python tools/bwd_variant_check.py [--variants 10,12,13]

Variant 10 (stage 2 folded into stage 1's last arrivers) must give dgrid BIT-IDENTICAL to the two-launch product
pass (same partial tiles, same order of additions), on a fresh workspace full of junk, on a re-used workspace and
when replayed from a hipGraph; variants 12 / 13 (four waves per row) re-associate the wave sums: allclose.
Prints one line per (shape, case, variant); exits non-zero on any mismatch.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from hdrnet_amd import _lib  # noqa: E402

SHAPES = [  # (B, H, W, GH, GW, GD)
    (1, 2160, 3840, 16, 16, 8), (1, 1080, 1920, 16, 16, 8), (2, 300, 500, 16, 16, 8), (1, 97, 131, 5, 7, 3),
    (3, 64, 64, 16, 16, 8), (1, 3000, 4000, 32, 32, 8), (1, 17, 1200, 2, 3, 8), (1, 1200, 36, 9, 2, 5),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="10,12,13")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda:0")
    lib = _lib.load_tools()
    lib.hdrnet_enable_kernel_names(1)
    stream = torch.cuda.current_stream(dev).cuda_stream
    gen = torch.Generator(device=dev).manual_seed(7)
    bad = 0
    for (B, H, W, GH, GW, GD) in SHAPES:
        Cin, Cout, C = 3, 3, 12
        grid = torch.rand((B, GH, GW, GD, C), device=dev, generator=gen)
        guide = torch.rand((B, H, W), device=dev, generator=gen)
        inp = torch.rand((B, H, W, Cin), device=dev, generator=gen)
        dout = torch.randn((B, H, W, Cout), device=dev, generator=gen)
        wsb = lib.hdrnet_bilateral_slice_apply_grad_workspace_bytes(B, H, W, GH, GW, GD, Cin, Cout, 1)

        def run(variant, case, ws, outs=None):
            dg, dgu, di = case
            o = outs or (torch.full_like(grid, float("nan")), torch.full_like(guide, float("nan")),
                         torch.full_like(inp, float("nan")))
            rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
                grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(),
                o[0].data_ptr() if dg else None, o[1].data_ptr() if dgu else None, o[2].data_ptr() if di else None,
                B, H, W, GH, GW, GD, Cin, Cout, 1, ws.data_ptr(), wsb, _lib.KERNEL_AUTO | (variant << 8), stream)
            if rc:
                raise RuntimeError(lib.hdrnet_last_error().decode())
            return o

        def junk_ws():
            return torch.randint(0, 256, (max(wsb, 16),), dtype=torch.uint8, device=dev, generator=gen)

        for case, cname in (((1, 0, 0), "dgrid"), ((1, 1, 0), "dgrid+dguide"), ((1, 1, 1), "all")):
            ref = run(0, case, junk_ws())
            torch.cuda.synchronize()
            kref = lib.hdrnet_last_kernel().decode()
            for v in variants:
                try:
                    ws = junk_ws()
                    got = run(v, case, ws)
                    got2 = run(v, case, ws)  # the same workspace again: the arrival words are back at zero
                    torch.cuda.synchronize()
                    kname = lib.hdrnet_last_kernel().decode()
                    exact = v == 10
                    ok = True
                    msgs = []
                    for nm, a, b2, r in zip(("dgrid", "dguide", "dinput"), got, got2, ref):
                        if not case[("dgrid", "dguide", "dinput").index(nm)]:
                            continue
                        if exact or nm != "dgrid":
                            same = torch.equal(a, r) and torch.equal(b2, r)
                        else:
                            same = torch.allclose(a, r, rtol=1e-5, atol=1e-5 * float(r.abs().max())) and torch.equal(a, b2)
                        ok &= bool(same)
                        msgs.append(f"{nm} max|d|={float((a - r).abs().max()):.2e}")
                    if v == 10 and case == (1, 1, 1):  # replayed from a hipGraph: the same epoch every replay
                        side = torch.cuda.Stream(device=dev)
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            st = torch.cuda.current_stream(dev).cuda_stream
                            outs = (torch.empty_like(grid), torch.empty_like(guide), torch.empty_like(inp))
                            ws_g = junk_ws()
                            g = torch.cuda.CUDAGraph()

                            def launch():
                                rc = lib.hdrnet_bilateral_slice_apply_grad_f32_ex(
                                    grid.data_ptr(), guide.data_ptr(), inp.data_ptr(), dout.data_ptr(),
                                    outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), B, H, W, GH, GW, GD,
                                    Cin, Cout, 1, ws_g.data_ptr(), wsb, _lib.KERNEL_AUTO | (v << 8), st)
                                assert rc == 0, lib.hdrnet_last_error().decode()
                            launch()
                            torch.cuda.synchronize()
                            with torch.cuda.graph(g, stream=side):
                                st = torch.cuda.current_stream(dev).cuda_stream
                                launch()
                            for _ in range(3):
                                outs[0].fill_(float("nan"))
                                g.replay()
                                torch.cuda.synchronize()
                                same = torch.equal(outs[0], ref[0])
                                ok &= bool(same)
                                msgs.append(f"graph replay {'ok' if same else 'MISMATCH'}")
                        torch.cuda.current_stream().wait_stream(side)
                    print(f"{B}x{H}x{W} grid {GH}x{GW}x{GD} {cname:13s} variant {v:3d} ({kname}; ref {kref}): "
                          f"{'OK' if ok else 'FAIL'}  {'; '.join(msgs)}", flush=True)
                    bad += 0 if ok else 1
                except Exception as e:  # noqa: BLE001
                    print(f"{B}x{H}x{W} {cname} variant {v}: ERROR {e}", flush=True)
                    bad += 1
    print("bad =", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
