"""Time of the fp32-grade GEMM (`ops.hip_learner.mm3`) on the shapes of the encoder's linear layers at batch 256
(30 720 token rows): forward, data gradient and split-K weight gradient, next to the bytes each has to move."""

from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from alphatriangle_amd.ops import hip_learner


def timed(fn, n: int = 50) -> float:
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main() -> None:
    dev = torch.device("cuda")
    rows = 30720
    for name, k, n in (("qkv", 128, 384), ("out-proj", 128, 128), ("ffn1", 128, 256), ("ffn2", 256, 128)):
        x = torch.randn(rows, k, device=dev)
        w = torch.randn(n, k, device=dev)
        g = torch.randn(rows, n, device=dev)
        bias = torch.randn(n, device=dev)
        t_f = timed(lambda: hip_learner.mm3(x, w.t(), bias))
        t_d = timed(lambda: hip_learner.mm3(g, w))
        s = 120
        gs = g.view(s, rows // s, n)
        t_w = timed(lambda: hip_learner.sum_slices(hip_learner.mm3(gs.transpose(1, 2), x.view(s, rows // s, k))))
        mb_f = (rows * k + rows * n) * 4 / 1e6
        print(f"[{name}: {rows} x {k} -> {n}] forward {t_f:.1f} us ({mb_f:.0f} MB, {mb_f / t_f:.2f} TB/s), data gradient {t_d:.1f} us, "
              f"weight gradient (120 slices + sum) {t_w:.1f} us", flush=True)


if __name__ == "__main__":
    main()
