"""Does splitting the self-play batch over two HIP streams (independent game sets, own trees, own engine buffers)
raise throughput by letting kernels of the two halves overlap their tails and mix MFMA-bound with VALU-bound phases?

    python bench/two_stream_probe.py [--games 4096] [--sims 400] [--moves 3] [--dtype fp32|bf16]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from alphatriangle_amd.config import SystemConfig, TrainConfig  # noqa: E402
from alphatriangle_amd.models.network import NeuralNetwork  # noqa: E402
from alphatriangle_amd.presets import build_preset  # noqa: E402
from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer  # noqa: E402
from alphatriangle_amd.rl.self_play.gpu_actor import GpuSelfPlayActor  # noqa: E402


def run(n_streams: int, games: int, sims: int, moves: int, dtype: str) -> float:
    dev = torch.device("cuda")
    env_cfg, model_cfg, mcts_cfg, _, _ = build_preset("default")
    mcts_cfg.max_simulations = sims
    train_cfg = TrainConfig(DEVICE="cuda")
    sys_cfg = SystemConfig(GAMES_PER_GPU=games // n_streams, INFERENCE_DTYPE=dtype)
    actors, streams = [], []
    for i in range(n_streams):
        torch.manual_seed(1234)
        net = NeuralNetwork(model_cfg, env_cfg, train_cfg, dev, inference_dtype=dtype)
        replay = DeviceReplayBuffer(train_cfg, env_cfg, model_cfg, dev, sys_cfg)
        actors.append(GpuSelfPlayActor(env_cfg, model_cfg, mcts_cfg, train_cfg, sys_cfg, net, replay, dev, seed=1000 + i))
        streams.append(torch.cuda.Stream(device=dev))  # default priority

    def move() -> None:
        for a, s in zip(actors, streams):
            with torch.cuda.stream(s):
                a.begin_move()
        for a, s in zip(actors, streams):
            with torch.cuda.stream(s):
                a.finish_move()

    move()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(moves):
        move()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / moves
    return games / dt


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--moves", type=int, default=3)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    args = ap.parse_args()
    out = {}
    for n in (1, 2):
        out[f"streams_{n}_moves_per_s"] = round(run(n, args.games, args.sims, args.moves, args.dtype), 1)
        print(json.dumps(out), flush=True)
    out["streams_2_games_x2_moves_per_s"] = round(run(2, 2 * args.games, args.sims, args.moves, args.dtype), 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
