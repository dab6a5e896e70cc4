"""Learner-only throughput: train steps per second on synthetic replay content (one GPU)."""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from alphatriangle_amd.config import EnvConfig, ModelConfig, SystemConfig, TrainConfig
from alphatriangle_amd.models.network import NeuralNetwork
from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
from alphatriangle_amd.rl.core.trainer import Trainer


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no HIP-graph capture of forward/backward")
    args = ap.parse_args()
    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=args.batch, COMPILE_MODEL=not args.eager)
    net = NeuralNetwork(mc, env, tc, dev)
    trainer = Trainer(net, tc, env, autocast_bf16=not args.fp32)
    replay = DeviceReplayBuffer(tc, env, mc, dev, SystemConfig())
    n0 = tc.MIN_BUFFER_SIZE_TO_TRAIN
    pol = torch.rand(n0, env.ACTION_DIM)
    replay.add_arrays(torch.randint(-1, 2, (n0, 120)).float(), torch.rand(n0, 30), pol / pol.sum(1, keepdim=True), torch.empty(n0).uniform_(-5, 5))

    def step(i: int) -> None:
        b = replay.sample(args.batch, current_train_step=i)
        _, td = trainer.train_step_tensors(b)
        replay.update_priorities(b.indices, td)

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    graph_ms = None
    g = trainer._graphed
    if g is not None and g.ok:  # the captured forward + backward alone, replayed back to back
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            g.graph.replay()
        torch.cuda.synchronize()
        graph_ms = (time.perf_counter() - t1) / args.steps * 1e3
    print(json.dumps({"graph_only_ms": graph_ms, "graph_kernel_nodes": trainer.graph_report.get("kernel_nodes"), "train_steps_per_sec": args.steps / dt, "samples_per_sec": args.steps * args.batch / dt, "ms_per_step": dt / args.steps * 1e3,
                      "dtype": "fp32" if args.fp32 else "bf16 autocast"}))


if __name__ == "__main__":
    main()
