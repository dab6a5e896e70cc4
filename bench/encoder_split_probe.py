"""Time of the fp32-grade engine forward (default preset) and of its two encoder-layer launches, for same-box A/B runs of
kernel variants (`ATRI_HIP_LIB=<variant.so> python bench/encoder_split_probe.py`).

    python bench/encoder_split_probe.py [--batch 4096] [--iters 40] [--repeats 3]
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
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--preset", default="default")
    args = ap.parse_args()
    dev = torch.device("cuda")
    if args.preset == "large":
        env = EnvConfig.large()
        mc = ModelConfig.large_transformer(other_dim=8 * env.NUM_SHAPE_SLOTS + 6)
    else:
        env, mc = EnvConfig(), ModelConfig()
    torch.manual_seed(0)
    eng = NeuralNetwork(mc, env, TrainConfig(DEVICE="cuda"), dev, inference_dtype="fp32").engine
    assert eng is not None and eng.dtype == "fp32"
    eng.reserve(args.batch)
    grid = torch.randint(-1, 2, (args.batch, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(args.batch, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    for _ in range(5):
        out = eng.forward(grid, other)
    torch.cuda.synchronize()
    check = float(out[0].float().abs().sum().item())
    times = []
    for _ in range(args.repeats):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.iters):
            eng.forward(grid, other)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / args.iters * 1e3)
    lib = os.environ.get("ATRI_HIP_LIB", "in-tree")
    print(f"[{os.path.basename(lib)}] forward of {args.batch} samples: " + ", ".join(f"{t:.1f}" for t in times) + f" us (logit checksum {check:.6e})", flush=True)


if __name__ == "__main__":
    main()
