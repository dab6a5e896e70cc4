"""Time of the learner's 3x3 convolution products at the default size (128 -> 128 channels, 256 boards of 8 x 15):
board-per-workgroup kernel against the patch-matrix + GEMM form, forward and weight gradient."""

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
    torch.manual_seed(0)
    for cin, cout, nb, h, w in ((128, 128, 256, 8, 15), (64, 128, 256, 8, 15), (128, 128, 512, 8, 15), (128, 128, 64, 12, 21)):
        x = torch.randn(cin, nb, h, w, device=dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        planes = hip_learner.conv3_pack(wt, False)
        t_direct = timed(lambda: hip_learner.conv3_cm(x, planes, cout))
        t_pack = timed(lambda: hip_learner.conv3_pack(wt, False))
        w2 = wt.reshape(cout, -1)
        t_patch = timed(lambda: hip_learner.patch_matrix_cm(x, 3, 3, 1, 1))
        cols = hip_learner.patch_matrix_cm(x, 3, 3, 1, 1)
        t_gemm = timed(lambda: hip_learner.mm3(w2, cols))
        flops = 2.0 * cout * cin * 9 * nb * h * w * 3
        line = (f"[{cin}->{cout}, {nb} boards {h}x{w}] direct {t_direct:.1f} us ({flops / t_direct / 1e6:.0f} TFLOP/s of bf16 MFMA work), "
                f"pack {t_pack:.1f} us; patch matrix {t_patch:.1f} us + GEMM {t_gemm:.1f} us")
        g = torch.randn(cout, nb, h, w, device=dev)
        if hip_learner.conv3_wgrad_supported(g, x):
            t_wg = timed(lambda: hip_learner.conv3_wgrad_cm(g, x))
            n = nb * h * w
            s = 32
            t_old = timed(lambda: hip_learner.sum_slices(hip_learner.mm3(g.view(cout, s, n // s).transpose(0, 1), cols.view(cin * 9, s, n // s).permute(1, 2, 0))))
            line += f"; weight gradient direct {t_wg:.1f} us, split-K GEMM over the patch matrix {t_old:.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
