"""Reference-style baseline: the reference's own execution structure, measured in this repo.

A pool of CPU self-play actors, each: host C++ game engine + host C++ PUCT search with leaf batches of 32,
PyTorch fp32 network evaluated on the actor's device - the structure of the reference's
``SelfPlayWorker`` -> ``trimcts.run_mcts`` -> ``NeuralNetwork.evaluate_batch`` stack
(alphatriangle/rl/self_play/worker.py:273-279, alphatriangle/nn/network.py:243), with the
reference's default pool of ``min(8, cores - 2)`` such actors
(alphatriangle/config/train_config.py:34-38).  The reference publishes no numbers and its
native dependencies (trianglengin / trimcts wheels) are not installable offline, so this is the
comparison point for ``bench.py``'s ``vs_baseline``: a builder-measured stand-in, not a number
of the reference's authors.

The pool is real: ``--pool`` processes run at the same time (sharing the one GPU when
``--device cuda``), each plays ``--moves`` timed moves after a warm-up move, and the pool figure
is the sum of the per-actor rates over the common time window - nothing is extrapolated.

    python bench/reference_style_baseline.py [--pool 8] [--moves 100] [--sims 400] [--device cuda] [--write]

``--write`` stores the result in BASELINE.json under ``in_repo_reference_style_baseline``.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _actor(idx: int, args: argparse.Namespace, start, out) -> None:  # noqa: ANN001
    import torch

    from alphatriangle_amd import presets
    from alphatriangle_amd.env.game_state import GameState
    from alphatriangle_amd.mcts.host_search import run_mcts
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.utils.helpers import get_device

    torch.set_num_threads(args.threads)
    env_cfg, model_cfg, mcts_cfg, train_cfg, _ = presets.build_preset(args.preset)
    mcts = mcts_cfg.model_copy(update={"max_simulations": args.sims, "mcts_batch_size": 32})
    dev = get_device(args.device)
    torch.manual_seed(1234)
    net = NeuralNetwork(model_cfg, env_cfg, train_cfg, dev, use_hip_engine=False)  # plain PyTorch module, fp32
    net.model.eval()
    game = GameState(env_cfg, initial_seed=1 + idx)
    handle = None
    visits, handle = run_mcts(game, net, mcts, previous_tree_handle=handle, last_action=-1)  # warm-up move
    start.wait()
    moves, last = 0, -1
    t0 = time.perf_counter()
    while moves < args.moves:
        if game.is_over():
            game = GameState(env_cfg, initial_seed=100 + 1000 * idx + moves)
            handle, last = None, -1
        visits, handle = run_mcts(game, net, mcts, previous_tree_handle=handle, last_action=last)
        if not visits:
            game = GameState(env_cfg, initial_seed=200 + 1000 * idx + moves)
            handle, last = None, -1
            continue
        last = max(visits, key=visits.get)
        game.step(last)
        moves += 1
    out.put((idx, moves, time.perf_counter() - t0))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="default")
    ap.add_argument("--moves", type=int, default=100, help="timed moves per actor")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--device", default="cpu")
    ap.add_argument("--threads", type=int, default=1, help="torch intra-op threads per actor (reference: 1 CPU per worker)")
    ap.add_argument("--pool", type=int, default=None, help="concurrent actors (default min(8, cores-2))")
    ap.add_argument("--write", action="store_true")
    args = ap.parse_args()
    cores = os.cpu_count() or 8
    pool = args.pool if args.pool is not None else max(1, min(8, cores - 2))
    ctx = mp.get_context("spawn")
    start, out = ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=_actor, args=(i, args, start, out)) for i in range(pool)]
    for p in procs:
        p.start()
    time.sleep(0.5)
    start.set()  # actors wait here after their warm-up move, so the timed windows overlap
    results = [out.get(timeout=3600) for _ in procs]
    for p in procs:
        p.join()
    rates = [m / dt for _, m, dt in results]
    total_moves = sum(m for _, m, _ in results)
    res = {
        "what": "the reference's execution structure measured with this repo's host path on the MI355X box: a real pool of "
                f"{pool} concurrent CPU actors, each a host C++ PUCT search (leaf batch 32) calling a PyTorch fp32 network "
                f"on {args.device}; builder-measured, the reference itself publishes no numbers",
        "script": "bench/reference_style_baseline.py",
        "dtype": "fp32",
        "preset": args.preset,
        "device": args.device,
        "sims_per_move": args.sims,
        "pool": pool,
        "moves_timed": total_moves,
        "moves_per_sec_per_actor": [round(r, 3) for r in rates],
        "moves_per_sec_400_sims": round(sum(rates), 3),
    }
    print(json.dumps(res))
    if args.write:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "BASELINE.json")
        with open(path) as fh:
            base = json.load(fh)
        base["in_repo_reference_style_baseline"] = res
        with open(path, "w") as fh:
            json.dump(base, fh, indent=2)
            fh.write("\n")


if __name__ == "__main__":
    main()
