"""Where one workgroup of the fp32-grade encoder-layer kernel spends its cycles.

The kernel stores the shader clock at its phase boundaries for one workgroup (``atri_split_encoder_phase_clock``);
this probe runs the fp32 engine forward a few times and prints the cycle deltas of the last launch.

    python bench/encoder_phase_probe.py [--batch 4096] [--block 2000]
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from alphatriangle_amd.config import EnvConfig, ModelConfig, TrainConfig
from alphatriangle_amd.models.network import NeuralNetwork
from alphatriangle_amd.ops import hipbind as hb

NAMES = {1: "barrier (LayerNorm1 planes)", 17: "residual rows arrive (diagnostic wait)", 18: "store", 19: "LayerNorm1 of the next sample", 2: "K GEMM + image store", 3: "V GEMM", 4: "Q GEMM", 5: "V^T exchange (3 barriers)", 6: "attention",
         7: "barrier", 8: "out-projection + y store", 9: "barrier", 10: "LayerNorm2 + barrier", 11: "FFN1 half 0", 12: "FFN2 half 0",
         13: "FFN1 half 1", 14: "FFN2 half 1"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--block", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    eng = NeuralNetwork(mc, env, TrainConfig(DEVICE="cuda"), dev, inference_dtype="fp32").engine
    assert eng is not None and eng.dtype == "fp32"
    eng.reserve(args.batch)
    grid = torch.randint(-1, 2, (args.batch, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(args.batch, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    clock = torch.zeros(20, dtype=torch.int64, device=dev)
    fn = getattr(hb.lib(), "atri_split_encoder_phase_clock")
    fn.restype = C.c_int
    for _ in range(3):
        eng.forward(grid, other)
    fn(C.c_void_p(clock.data_ptr()), C.c_int(args.block))
    eng.forward(grid, other)
    torch.cuda.synchronize()
    fn(C.c_void_p(0), C.c_int(0))
    c = clock.cpu().tolist()
    marks = sorted((i for i in range(20) if c[i] != 0), key=lambda i: c[i])
    total = c[marks[-1]] - c[marks[0]]
    print(f"workgroup {args.block}: {total} cycles")
    for prev, cur in zip(marks, marks[1:]):
        d = c[cur] - c[prev]
        print(f"  {NAMES.get(cur, str(cur)):32s} {d:8d}  {100.0 * d / total:5.1f} %")


if __name__ == "__main__":
    main()
