"""Which learner ops put memset nodes into a captured HIP graph?  Captures each candidate (forward + backward at the
learner's shapes) on its own and prints the node census (kernel / memset / other)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from alphatriangle_amd.models.model import EncoderLayer, GemmConv2d, _linear_rows  # noqa: E402
from alphatriangle_amd.ops import hipbind as hb  # noqa: E402

dev = torch.device("cuda")


def census(name, fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g):
        fn()
    c = (C.c_int * 3)()
    hb.call("atri_graph_census", C.c_void_p(g.raw_cuda_graph()), c)
    print(f"{name:44s} kernels {c[0]:4d}  memsets {c[1]:3d}  other {c[2]:3d}", flush=True)


def fb(mod_fn, *inputs):
    def run():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = mod_fn(*inputs)
        y.float().square().mean().backward()
    return run


B = 256
x_tok = torch.randn(120, B, 128, device=dev, requires_grad=True)
lin = torch.nn.Linear(128, 384).to(dev)
census("split-K linear 128->384 (30720 rows)", fb(lambda x: _linear_rows(x, lin.weight, lin.bias), x_tok))
census("F.linear 128->384 (30720 rows)", fb(lambda x: torch.nn.functional.linear(x, lin.weight, lin.bias), x_tok))
layer = EncoderLayer(128, 4, 256, activation="relu", norm_first=True).to(dev).train()
census("EncoderLayer fast path", fb(layer, x_tok))
ln = torch.nn.LayerNorm(128).to(dev)
census("LayerNorm", fb(ln, x_tok))
census("dropout", fb(lambda x: torch.nn.functional.dropout(x, 0.1, True), x_tok))
q = torch.randn(B, 4, 120, 32, device=dev, requires_grad=True)
census("softmax(q k^T) v", fb(lambda t: torch.softmax(t @ t.transpose(-2, -1), -1) @ t, q))
census("flash SDPA", fb(lambda t: torch.nn.functional.scaled_dot_product_attention(t, t, t), q))
x_img = torch.randn(B, 128, 8, 15, device=dev, requires_grad=True)
conv = GemmConv2d(128, 128, 3, 1, 1, bias=False).to(dev)
census("GemmConv2d 128->128", fb(conv, x_img))
bn = torch.nn.BatchNorm2d(128).to(dev).train()
census("BatchNorm2d", fb(bn, x_img))
x_fc = torch.randn(B, 15390, device=dev, requires_grad=True)
fc = torch.nn.Linear(15390, 128).to(dev)
census("Linear 15390->128 (256 rows)", fb(fc, x_fc))
bn1 = torch.nn.BatchNorm1d(128).to(dev).train()
x_h = torch.randn(B, 128, device=dev, requires_grad=True)
census("BatchNorm1d", fb(bn1, x_h))
head = torch.nn.Linear(128, 360).to(dev)
census("Linear 128->360 (256 rows)", fb(head, x_h))
census("cat + permute/flatten", fb(lambda a, b: torch.cat([a.permute(1, 0, 2).flatten(1), b], 1), x_tok, x_fc))
