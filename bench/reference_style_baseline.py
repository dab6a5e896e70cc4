"""Reference-style baseline: the reference's own execution structure, measured in this repo.

One CPU self-play actor: host C++ game engine + host C++ PUCT search with leaf batches of 32,
PyTorch fp32 network evaluated on the actor's device - the structure of the reference's
``SelfPlayWorker`` -> ``trimcts.run_mcts`` -> ``NeuralNetwork.evaluate_batch`` stack
(alphatriangle/rl/self_play/worker.py:273-279, alphatriangle/nn/network.py:243), with the
reference's default pool of ``min(8, cores - 2)`` such actors
(alphatriangle/config/train_config.py:34-38).  The reference publishes no numbers and its
native dependencies (trianglengin / trimcts wheels) are not installable offline, so this is the
comparison point for ``bench.py``'s ``vs_baseline``.

    python bench/reference_style_baseline.py [--moves 6] [--sims 400] [--device cpu] [--write]

``--write`` stores the result in BASELINE.json under ``in_repo_reference_style_baseline``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from alphatriangle_amd import presets  # noqa: E402
from alphatriangle_amd.env.game_state import GameState  # noqa: E402
from alphatriangle_amd.mcts.host_search import run_mcts  # noqa: E402
from alphatriangle_amd.models.network import NeuralNetwork  # noqa: E402
from alphatriangle_amd.utils.helpers import get_device  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="default")
    ap.add_argument("--moves", type=int, default=6)
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--threads", type=int, default=1, help="torch intra-op threads of the one actor (reference: 1 CPU per worker)")
    ap.add_argument("--pool", type=int, default=None, help="actors assumed for the pool figure (default min(8, cores-2))")
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    env_cfg, model_cfg, mcts_cfg, train_cfg, _ = presets.build_preset(args.preset)
    mcts = mcts_cfg.model_copy(update={"max_simulations": args.sims, "mcts_batch_size": 32})
    dev = get_device(args.device)
    net = NeuralNetwork(model_cfg, env_cfg, train_cfg, dev)
    if getattr(net, "engine", None) is not None:
        net.engine = None  # the baseline evaluates with the plain PyTorch module
    net.model.eval()
    game = GameState(env_cfg, initial_seed=1)
    handle = None
    # warm-up move (allocations, first-call overheads)
    visits, handle = run_mcts(game, net, mcts, previous_tree_handle=handle, last_action=-1)
    moves = 0
    t0 = time.perf_counter()
    last = -1
    while moves < args.moves:
        if game.is_over():
            game = GameState(env_cfg, initial_seed=100 + moves)
            handle, last = None, -1
        visits, handle = run_mcts(game, net, mcts, previous_tree_handle=handle, last_action=last)
        if not visits:
            game = GameState(env_cfg, initial_seed=200 + moves)
            handle, last = None, -1
            continue
        last = max(visits, key=visits.get)
        game.step(last)
        moves += 1
    dt = time.perf_counter() - t0
    per_actor = moves / dt
    cores = os.cpu_count() or 8
    pool = args.pool if args.pool is not None else max(1, min(8, cores - 2))
    out = {
        "structure": "1 CPU actor: host C++ PUCT search (leaf batch 32) + PyTorch fp32 network, no GPU-resident tree",
        "preset": args.preset,
        "device": str(dev),
        "sims_per_move": args.sims,
        "moves_timed": moves,
        "moves_per_sec_per_actor": round(per_actor, 4),
        "reference_default_pool": pool,
        "moves_per_sec_400_sims": round(per_actor * pool, 4),
        "note": "pool figure = per-actor rate x min(8, cores-2) actors (perfect scaling assumed, an upper bound)",
    }
    print(json.dumps(out))
    if args.write:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "BASELINE.json")
        with open(path) as fh:
            base = json.load(fh)
        base["in_repo_reference_style_baseline"] = out
        with open(path, "w") as fh:
            json.dump(base, fh, indent=2)
            fh.write("\n")


if __name__ == "__main__":
    main()
