"""HBM bandwidth calibration on one MI355X with plain PyTorch ops (copy, fill, read-reduce)."""
import torch

dev = torch.device("cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for mb in (126, 504, 2016):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"{mb:5d} MB copy : {t * 1e6:8.1f} us  {2 * mb / 1024 / t / 1e3:6.2f} TB/s (read+write)")
    t = timeit(lambda: y.fill_(1.0))
    print(f"{mb:5d} MB fill : {t * 1e6:8.1f} us  {mb / 1024 / t / 1e3:6.2f} TB/s (write)")
    xf = x.view(torch.int16)
    t = timeit(lambda: xf.sum(dtype=torch.int32) if False else torch.max(xf))
    print(f"{mb:5d} MB max  : {t * 1e6:8.1f} us  {mb / 1024 / t / 1e3:6.2f} TB/s (read)", flush=True)
