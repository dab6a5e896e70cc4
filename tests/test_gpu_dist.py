"""RCCL (``nccl`` backend) smoke on the one visible GPU: a single-rank process group runs every collective the
multi-GPU layouts use, on the streams they use them on.  More ranks than GPUs cannot share a card under RCCL; the
multi-rank logic itself is covered by the gloo tests in tests/test_parallel.py."""

from __future__ import annotations

import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import torch, torch.distributed as dist
from alphatriangle_amd.config import EnvConfig, ModelConfig, TrainConfig
from alphatriangle_amd.models.network import NeuralNetwork
from alphatriangle_amd.parallel.dist import init_distributed
from alphatriangle_amd.parallel.weights import WeightWire
from alphatriangle_amd.rl.core.trainer import TensorBatch, Trainer

ctx = init_distributed(0, 1, 0, backend="nccl", init_method="env://", timeout_s=60)
assert ctx.backend == "nccl" and ctx.device.type == "cuda"
dev = ctx.device
ctx.warm_up()
print("init ok", flush=True)

x = torch.arange(1 << 20, dtype=torch.float32, device=dev)
ref = x.clone()
dist.all_reduce(x)                                   # flat-gradient sized all-reduce
ctx.broadcast_weights(x)
assert torch.equal(x, ref)
g = ctx.all_gather_records(x[:4096].view(64, 64))
assert g.shape == (1, 64, 64) and torch.equal(g[0], ref[:4096].view(64, 64))
t = torch.tensor([3.0, 7.0], device=dev)
assert ctx.all_reduce_max_(t).tolist() == [3.0, 7.0]
assert ctx.all_reduce_scalar(2.5, "sum") == 2.5 and ctx.all_reduce_scalar(2.5, "max") == 2.5
assert ctx.agree([1.0, 4.0], "min") == [1.0, 4.0]   # host-side control plane next to the RCCL communicator
assert ctx.control is not None and ctx.control.add("k", 2) == 2 and ctx.control.peek("none", 5) == 5
sub = dist.new_group(ranks=[0])                      # the split layout builds sub-groups
out = [torch.empty(8, device=dev)]
dist.gather(torch.ones(8, device=dev), gather_list=out, dst=0, group=sub)
assert out[0].sum().item() == 8
ctx.barrier()
print("collectives ok", flush=True)

# a data-parallel learner step: all-reduce of the flat gradient buffer between backward and the update,
# issued from a side stream as the overlapped loop does
env, mc = EnvConfig(), ModelConfig()
tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=32, MAX_TRAINING_STEPS=10, COMPILE_MODEL=True)
net = NeuralNetwork(mc, env, tc, dev)
calls = []
def grad_sync(flat):
    calls.append(flat.numel())
    dist.all_reduce(flat)
tr = Trainer(net, tc, env, autocast_bf16=True, grad_sync=grad_sync, world_size=1)
pol = torch.rand(32, env.ACTION_DIM, device=dev)
tb = TensorBatch(torch.randint(-1, 2, (32, 1, env.ROWS, env.COLS), device=dev).float(), torch.rand(32, 30, device=dev),
                 pol / pol.sum(1, keepdim=True), torch.empty(32, device=dev).uniform_(-5, 5), torch.ones(32, device=dev))
before = tr.flat.flat_param.clone()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    info, td = tr.train_step_tensors(tb)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
assert calls == [tr.flat.total] and torch.isfinite(info["total_loss"]) and not torch.equal(before, tr.flat.flat_param)
# the first step captured forward/backward into a HIP graph while the process group (and its watchdog thread) was
# alive; further steps replay it with the eager all-reduce in between
assert tr.graph_report.get("replayed") is True and tr.graph_report["memset_nodes"] == 0, tr.graph_report
for _ in range(3):
    with torch.cuda.stream(side):
        info, td = tr.train_step_tensors(tb)
    torch.cuda.synchronize()
    assert torch.isfinite(info["total_loss"]) and torch.isfinite(tr.flat.flat_param).all()
assert len(calls) == 4
wire = WeightWire(net._orig_model)
buf = wire.pack()
ctx.broadcast_weights(buf)
wire.unpack()
print("learner ok", flush=True)
ctx.shutdown()
print("done", flush=True)
"""


def test_rccl_single_rank_collectives_and_dp_step():
    assert torch.cuda.is_available()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-u", "-c", _SCRIPT], env=env, cwd=root, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0 and "done" in res.stdout, f"stdout:\n{res.stdout}\nstderr:\n{res.stderr[-3000:]}"
