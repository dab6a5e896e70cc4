"""Roofline probe of the MFMA implicit-GEMM conv kernel on one MI355X.

Times ``atri_tile_gemm_probe`` for a 3x3 (or 1x1) layer at self-play batch size with one
main-loop ingredient removed per variant, so the cost of LDS reads / weight loads / address
math / everything-but-the-loop can be read off the differences.

    python bench/conv_probe.py [--batch 4096] [--cin 128] [--cout 128] [--taps 9] [--rows-per-wg 240]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from alphatriangle_amd.ops import hipbind as hb  # noqa: E402
from alphatriangle_amd.ops.inference import pack_gemm_weights  # noqa: E402

VARIANTS = {0: "full", 1: "no LDS reads", 2: "no weight loads", 4: "no main loop", 8: "no tap address math",
            20: "no loop, no staging", 48: "loop only (no staging, no stores)", 36: "no loop, no stores", 52: "no loop/staging/stores"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--cin", type=int, default=128)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--taps", type=int, default=9)
    ap.add_argument("--rows-per-wg", type=int, default=240)
    ap.add_argument("--kernel", choices=["tile", "rowconv"], default="rowconv")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variants", type=str, default="0,1,2,4,20,36,52")
    args = ap.parse_args()
    dev = torch.device("cuda")
    H, W = 8, 15
    HW = H * W
    rows = args.batch * HW
    x = torch.randn(rows, args.cin, device=dev).to(torch.bfloat16)
    w = torch.randn(args.taps, args.cin, args.cout, device=dev) * 0.05
    packed, nb = pack_gemm_weights(w, dev)
    bias = torch.zeros(nb * 128, device=dev)
    out = torch.empty(rows, args.cout, dtype=torch.bfloat16, device=dev)
    flops = 2.0 * rows * args.cin * args.cout * args.taps
    for dbg in [int(v) for v in args.variants.split(",")]:
        def run() -> None:
            if args.kernel == "rowconv":
                hb.call("atri_rowconv_probe", hb.ptr(x), hb.ptr(packed), hb.ptr(bias), hb.ptr(out), hb.i32(args.batch), hb.i32(H),
                        hb.i32(W), hb.i32(args.cin), hb.i32(args.cout), hb.i32(nb), hb.i32(dbg))
                return
            hb.call("atri_tile_gemm_probe", hb.ptr(x), hb.ptr(packed), hb.ptr(bias), hb.ptr(out), hb.i32(rows),
                    hb.i32(args.rows_per_wg), hb.i32(HW), hb.i32(H), hb.i32(W), hb.i32(args.cin), hb.i32(args.cout),
                    hb.i32(args.taps), hb.i32(nb), hb.i32(dbg))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000.0 / args.iters
        print(f"variant {dbg} ({VARIANTS[dbg]:>20}): {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s-equivalent", flush=True)


if __name__ == "__main__":
    main()
