"""Engine forward of the 12x21 preset with the flat-tile conv kernels and with the generic tile kernel.

    python bench/large_conv_probe.py [--batch 2048]
"""
from __future__ import annotations

import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from alphatriangle_amd.config import EnvConfig, ModelConfig, TrainConfig
from alphatriangle_amd.models.network import NeuralNetwork


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda")
    env, mc = EnvConfig.large(), ModelConfig.large_transformer()
    eng = NeuralNetwork(mc, env, TrainConfig(DEVICE="cuda"), dev, inference_dtype="bf16").engine
    assert eng is not None
    B = args.batch
    grid = torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    for flat in (True, False, True, False):
        eng.use_rowconv = flat
        for _ in range(3):
            eng.forward(grid, other)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.reps):
            eng.forward(grid, other)
        t1.record()
        torch.cuda.synchronize()
        print(f"12x21 bf16 forward, {B} samples, {'flat-tile convs' if flat else 'generic tile kernel'}: "
              f"{t0.elapsed_time(t1) / args.reps * 1e3:9.1f} us", flush=True)
    eng.use_rowconv = True
    phase_clock(eng, grid, other, B)


def phase_clock(eng, grid, other, B: int) -> None:
    """Shader-clock stamps of one workgroup of the fused residual-block kernel (100 MHz ticks)."""
    import ctypes as C

    from alphatriangle_amd.ops import hipbind as hb

    fn = hb.lib().atri_flatconv_phase_clock
    fn.restype = C.c_int
    names = ["staging", "barrier", "conv1 main loop", "barrier", "intermediate -> tile", "barrier", "conv2 main loop",
             "barrier", "output tile", "barrier", "write-back"]
    for block in (B // 2, B - 200):
        clk = torch.zeros(12, dtype=torch.int64, device=grid.device)
        fn(hb.ptr(clk), C.c_int(block))
        try:
            eng.forward(grid, other)
            torch.cuda.synchronize()
        finally:
            fn(hb.VP(0), C.c_int(0))
        c = clk.cpu().tolist()
        print(f"workgroup {block}: total {c[11] - c[0]} ticks")
        for i, n in enumerate(names):
            print(f"  {n:22s} {c[i + 1] - c[i]:7d}")


if __name__ == "__main__":
    main()
