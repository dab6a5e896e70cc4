"""Time of the learner's fused attention kernels (`ops.hip_learner.attention_forward` / `attention_backward`) on the default
preset's shape (120 tokens, batch 256, 4 heads) in both row layouts: token-major ``[L, B, 3D]`` and batch-major ``[B, L, 3D]``.
Also checks that the two layouts give the same numbers."""

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
    for seq, batch, heads in ((120, 256, 4), (252, 256, 4)):
        for dtype in (torch.float32, torch.bfloat16):
            d = 32 * heads
            qkv_t = torch.randn(seq, batch, 3 * d, device=dev).to(dtype)
            dctx_t = torch.randn(seq, batch, d, device=dev).to(dtype)
            qkv_b = qkv_t.transpose(0, 1).contiguous()
            dctx_b = dctx_t.transpose(0, 1).contiguous()
            out = {}
            for name, qkv, dctx, bm in (("token-major", qkv_t, dctx_t, False), ("batch-major", qkv_b, dctx_b, True)):
                ctx, lse, seed = hip_learner.attention_forward(qkv, heads, 0.0, bm)
                dqkv = hip_learner.attention_backward(qkv, dctx, lse, seed, heads, 0.0, bm)
                out[name] = (ctx, dqkv)
                t_f = timed(lambda: hip_learner.attention_forward(qkv, heads, 0.0, bm))
                t_b = timed(lambda: hip_learner.attention_backward(qkv, dctx, lse, seed, heads, 0.0, bm))
                print(f"[L={seq} B={batch} H={heads} {str(dtype)[6:]} {name}] forward {t_f:.1f} us, backward {t_b:.1f} us", flush=True)
            e_f = (out["token-major"][0].transpose(0, 1).float() - out["batch-major"][0].float()).abs().max().item()
            e_b = (out["token-major"][1].transpose(0, 1).float() - out["batch-major"][1].float()).abs().max().item()
            print(f"  layouts differ by at most {e_f:.2e} (ctx), {e_b:.2e} (dqkv)", flush=True)


if __name__ == "__main__":
    main()
