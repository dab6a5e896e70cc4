"""Roofline probe of the fused attention-block kernel on one MI355X: phases removed one at a time.

    python bench/attn_probe.py [--batch 4096] [--hw 120]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from alphatriangle_amd.ops import hipbind as hb  # noqa: E402
from alphatriangle_amd.ops.inference import pack_gemm_weights  # noqa: E402

VARIANTS = {0: "full", 1: "no LayerNorm", 2: "no K/V/Q GEMMs", 4: "no attention loop", 8: "no out-proj GEMM",
            15: "staging + epilogue only", 47: "staging only (no stores)", 63: "launch + barriers only"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--hw", type=int, default=120)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda")
    B, HW, D = args.batch, args.hw, 128
    x = torch.randn(B * HW, D, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    wqkv, _ = pack_gemm_weights(torch.randn(1, D, 3 * D, device=dev) * 0.08, dev)
    wo, _ = pack_gemm_weights(torch.randn(1, D, D, device=dev) * 0.08, dev)
    bqkv = torch.zeros(3 * D, device=dev)
    bo = torch.zeros(D, device=dev)
    g = torch.ones(D, device=dev)
    bt = torch.zeros(D, device=dev)
    for dbg, name in VARIANTS.items():
        def run() -> None:
            hb.call("atri_attn_block_probe", hb.ptr(x), hb.ptr(out), hb.ptr(wqkv), hb.ptr(bqkv), hb.ptr(wo), hb.ptr(bo), hb.ptr(g),
                    hb.ptr(bt), hb.i32(B), hb.i32(HW), hb.i32(dbg))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"variant {dbg:2d} ({name:>28}): {e0.elapsed_time(e1) * 1000.0 / args.iters:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
