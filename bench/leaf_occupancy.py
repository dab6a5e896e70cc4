"""Share of leaf slots that need a network evaluation, move by move (leaf-compaction go / no-go measurement).

Every simulation wave runs the network over all ``games * leaves_per_tree`` rows; slots whose selection ended in a
terminal node, at the depth cap or on an in-flight leaf need no evaluation.  The tree kernel counts the slots by kind
(per-tree counters, ops/engine.py: census); this script plays ``--moves`` moves and prints the counts per move.

    python bench/leaf_occupancy.py [--games 4096] [--sims 100] [--moves 60] [--leaves 1] [--dtype fp32] [--no-compact]

The run is timed per block of five moves, so the same command with and without ``--no-compact`` shows what evaluating
only the live rows (SystemConfig.COMPACT_LEAF_BATCH) buys once the share drops below one.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="bf16")
    ap.add_argument("--no-compact", action="store_true", help="evaluate every row of every wave")
    args = ap.parse_args()
    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=1_000_000, N_STEP_RETURNS=5)
    sc = SystemConfig(GAMES_PER_GPU=args.games, LEAVES_PER_TREE=args.leaves, INFERENCE_DTYPE=args.dtype, COMPACT_LEAF_BATCH=not args.no_compact)
    torch.manual_seed(0)
    net = NeuralNetwork(mc, env, tc, dev, inference_dtype=args.dtype)
    replay = DeviceReplayBuffer(tc, env, mc, dev, sc, emit_capacity=args.games * 8)
    actor = GpuSelfPlayActor(env, mc, AlphaTriangleMCTSConfig(max_simulations=args.sims, max_depth=args.depth), tc, sc, net, replay, dev, seed=3)
    actor.per_game_metrics = False
    total = {"none": 0, "expand": 0, "terminal": 0, "depth_cap": 0}
    episodes = 0
    torch.cuda.synchronize()
    t_block = t_start = time.perf_counter()
    for move in range(args.moves):
        rep = actor.play_move()
        episodes += rep.episodes_finished
        c = rep.slot_census
        for k in total:
            total[k] += c[k]
        n = sum(c.values())
        if move % 5 == 4 or move == args.moves - 1:
            torch.cuda.synchronize()
            now = time.perf_counter()
            print(f"move {move + 1:3d}: expand {c['expand'] / n:.4f} terminal {c['terminal'] / n:.4f} depth_cap {c['depth_cap'] / n:.4f} "
                  f"none {c['none'] / n:.4f}  episodes finished so far {episodes}  {(now - t_block) / 5 * 1e3:7.1f} ms/move", flush=True)
            t_block = now
    n = sum(total.values())
    print(json.dumps({"games": args.games, "sims": args.sims, "moves": args.moves, "leaves_per_tree": args.leaves, "dtype": args.dtype,
                      "compact": actor.compact, "ms_per_move": round((time.perf_counter() - t_start) / args.moves * 1e3, 2), "episodes_finished": episodes,
                      "slots": n, **{f"share_{k}": round(v / n, 5) for k, v in total.items()}}))


if __name__ == "__main__":
    main()
