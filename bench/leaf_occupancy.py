"""Share of leaf slots that need a network evaluation, move by move (leaf-compaction go / no-go measurement).

Every simulation wave runs the network over all ``games * leaves_per_tree`` rows; slots whose selection ended in a
terminal node, at the depth cap or on an in-flight leaf need no evaluation.  The tree kernel counts the slots by kind
(per-tree counters, ops/engine.py: census); this script plays ``--moves`` moves and prints the counts per move.

    python bench/leaf_occupancy.py [--games 4096] [--sims 100] [--moves 60] [--leaves 1]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig  # noqa: E402
from alphatriangle_amd.models.network import NeuralNetwork  # noqa: E402
from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer  # noqa: E402
from alphatriangle_amd.rl.self_play.gpu_actor import GpuSelfPlayActor  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--sims", type=int, default=100)
    ap.add_argument("--moves", type=int, default=60)
    ap.add_argument("--leaves", type=int, default=1)
    ap.add_argument("--depth", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=1_000_000, N_STEP_RETURNS=5)
    sc = SystemConfig(GAMES_PER_GPU=args.games, LEAVES_PER_TREE=args.leaves, INFERENCE_DTYPE="bf16")
    torch.manual_seed(0)
    net = NeuralNetwork(mc, env, tc, dev, inference_dtype="bf16")
    replay = DeviceReplayBuffer(tc, env, mc, dev, sc, emit_capacity=args.games * 8)
    actor = GpuSelfPlayActor(env, mc, AlphaTriangleMCTSConfig(max_simulations=args.sims, max_depth=args.depth), tc, sc, net, replay, dev, seed=3)
    actor.per_game_metrics = False
    total = {"none": 0, "expand": 0, "terminal": 0, "depth_cap": 0}
    episodes = 0
    for move in range(args.moves):
        rep = actor.play_move()
        episodes += rep.episodes_finished
        c = rep.slot_census
        for k in total:
            total[k] += c[k]
        n = sum(c.values())
        if move % 5 == 4 or move == args.moves - 1:
            print(f"move {move + 1:3d}: expand {c['expand'] / n:.4f} terminal {c['terminal'] / n:.4f} depth_cap {c['depth_cap'] / n:.4f} "
                  f"none {c['none'] / n:.4f}  episodes finished so far {episodes}", flush=True)
    n = sum(total.values())
    print(json.dumps({"games": args.games, "sims": args.sims, "moves": args.moves, "leaves_per_tree": args.leaves, "episodes_finished": episodes,
                      "slots": n, **{f"share_{k}": round(v / n, 5) for k, v in total.items()}}))


if __name__ == "__main__":
    main()
