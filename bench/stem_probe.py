"""Roofline probe of the one-launch conv stem (board plane -> 32 -> 64 -> 128 channels) on one MI355X: times
``atri_rowconv_stem_probe`` with ingredients removed (see the DBG parameter of rowconv_kernel).

    python bench/stem_probe.py [--batch 4096]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from alphatriangle_amd.ops import hipbind as hb  # noqa: E402
from alphatriangle_amd.ops.inference import pack_gemm_weights  # noqa: E402

VARIANTS = {0: "full", 64: "no first-layer evaluation", 4: "no main loops", 32: "no write-back", 36: "no main loops, no write-back",
            100: "zeroing + barriers only"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda")
    H, W = 8, 15
    B = args.batch
    grid = torch.randint(-1, 2, (B, H * W), device=dev).float()
    w_first = (torch.randn(9, 32, device=dev) * 0.3).contiguous()
    b_first = torch.zeros(32, device=dev)
    w1, _ = pack_gemm_weights(torch.randn(9, 32, 64, device=dev) * 0.05, dev)
    w2, _ = pack_gemm_weights(torch.randn(9, 64, 128, device=dev) * 0.05, dev)
    b1 = torch.zeros(128, device=dev)
    b2 = torch.zeros(128, device=dev)
    out = torch.empty(B * H * W, 128, dtype=torch.bfloat16, device=dev)
    for dbg, name in VARIANTS.items():
        def run() -> None:
            hb.call("atri_rowconv_stem_probe", hb.ptr(grid), hb.ptr(w_first), hb.ptr(b_first), hb.ptr(w1), hb.ptr(b1), hb.ptr(w2),
                    hb.ptr(b2), hb.ptr(out), hb.i32(B), hb.i32(H), hb.i32(W), hb.i32(1), hb.i32(dbg))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"variant {dbg:3d} ({name:>32s}): {e0.elapsed_time(e1) / args.iters * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
