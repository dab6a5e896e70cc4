"""Does the engine forward gain from batches whose activations fit the memory-side cache?

Every layer of the engine reads the previous layer's activation tensor and writes its own.  At 4096 samples one fp32
tensor is 252 MB - about the size of the MI355X memory-side cache - so each layer streams through HBM; in chunks of
2048 or 1024 samples a layer's input should still be cache resident.  This probe times the same 4096 evaluations as one
forward and as 2 / 4 / 8 chunked forwards (the chunks reuse the same activation buffers).

    python bench/chunk_probe.py [--dtype fp32|bf16] [--batch 4096]
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
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    eng = NeuralNetwork(mc, env, TrainConfig(DEVICE="cuda"), dev, inference_dtype=args.dtype).engine
    assert eng is not None
    B = args.batch
    eng.reserve(B)
    grid = torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    for chunks in (1, 2, 4, 8, 1):
        n = B // chunks
        parts = [(grid[i * n:(i + 1) * n].contiguous(), other[i * n:(i + 1) * n].contiguous()) for i in range(chunks)]
        for _ in range(3):
            for g, o in parts:
                eng.forward(g, o)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.reps):
            for g, o in parts:
                eng.forward(g, o)
        t1.record()
        torch.cuda.synchronize()
        print(f"{args.dtype} {B} samples as {chunks} x {n}: {t0.elapsed_time(t1) / args.reps * 1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
