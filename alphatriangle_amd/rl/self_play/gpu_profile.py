"""Per-move GPU timing of the device-resident actor: the ``--profile`` artefact of the GPU path.

The reference profiles worker 0 with cProfile and writes one ``.prof`` per episode into ``<run>/profile_data/``
(alphatriangle/rl/self_play/worker.py:99-104,549-566).  A GPU-resident actor spends its time in kernels, which a host
profiler does not see, so here the move phases are bracketed with HIP events on the streams that run them:

    root_eval   prepare_roots + root forward + init_roots (with Dirichlet noise)
    waves       every simulation wave of the move (tree kernel + network forward, graph replays)
    commit      wait for the overlapped learner + commit_moves + reroot
    learner     the learner steps of this round on their own stream (overlapped with ``waves``)
    move        first event to last event of the move on the self-play stream

Events are recorded while the work is enqueued and read back only inside ``finish_move``, after the synchronisation
that call performs anyway - profiling adds no host synchronisation of its own.  One JSON record per move goes to
``<run>/profile_data/gpu_actor_rank<r>.jsonl``; ``analyze_profiles.py`` summarises the file and the training loop emits
the same numbers as ``Perf/*`` metrics.
"""

from __future__ import annotations

import json
import logging
import time
from pathlib import Path
from typing import Any

logger = logging.getLogger(__name__)

PHASES = ("root_eval", "waves", "commit", "learner", "move")


class MoveProfiler:
    """HIP-event stopwatch of one actor.  ``mark(name)`` records an event on the current (or the given) stream;
    ``finish(extra)`` - called once the move's work is known to be complete - turns the marks into phase times."""

    def __init__(self, device: Any, out_dir: str | Path | None, rank: int = 0):
        import torch

        self._torch = torch
        self.device = device
        self.rank = int(rank)
        self.path: Path | None = None
        if out_dir is not None:
            d = Path(out_dir)
            d.mkdir(parents=True, exist_ok=True)
            self.path = d / f"gpu_actor_rank{self.rank}.jsonl"
        self._events: dict[str, Any] = {}
        self._fh = None
        self.moves = 0
        self.last: dict[str, float] = {}

    def mark(self, name: str, stream: Any = None) -> None:
        ev = self._events.get(name)
        if ev is None:
            ev = self._events[name] = self._torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else self._torch.cuda.current_stream(self.device))

    def _span(self, a: str, b: str) -> float | None:
        ea, eb = self._events.get(a), self._events.get(b)
        if ea is None or eb is None:
            return None
        try:
            return float(ea.elapsed_time(eb))
        except RuntimeError:  # an event of this pair was not recorded in this move
            return None

    def finish(self, extra: dict[str, Any] | None = None) -> dict[str, float]:
        """Phase times in milliseconds of the move whose events have all completed (the caller has synchronised)."""
        spans = {
            "root_eval": self._span("move_start", "root_done"),
            "waves": self._span("root_done", "waves_done"),
            "commit": self._span("waves_done", "move_end"),
            "learner": self._span("learner_start", "learner_end"),
            "move": self._span("move_start", "move_end"),
        }
        rec: dict[str, Any] = {"move": self.moves, "rank": self.rank, "wall_time": round(time.time(), 3)}
        rec.update({f"{k}_ms": round(v, 4) for k, v in spans.items() if v is not None})
        if extra:
            rec.update(extra)
        self.last = {k: v for k, v in rec.items() if k.endswith("_ms")}
        self.moves += 1
        # the learner events belong to one move only: without fresh marks the next move must not report them again
        self._events.pop("learner_start", None)
        self._events.pop("learner_end", None)
        if self.path is not None:
            try:
                if self._fh is None:
                    self._fh = open(self.path, "a", buffering=1)
                self._fh.write(json.dumps(rec) + "\n")
            except OSError as exc:  # pragma: no cover - a full disk must not end the run
                logger.warning("profile record not written: %s", exc)
        return self.last

    def close(self) -> None:
        if self._fh is not None:
            self._fh.close()
            self._fh = None


def summarise(path: str | Path) -> dict[str, Any]:
    """Mean / median / p95 / share of the move per phase over every record of a ``gpu_actor_rank*.jsonl`` file."""
    import statistics

    rows = []
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if line:
                rows.append(json.loads(line))
    out: dict[str, Any] = {"file": str(path), "moves": len(rows), "phases": {}}
    if not rows:
        return out
    move_total = sum(r.get("move_ms", 0.0) for r in rows)
    for phase in PHASES:
        vals = sorted(r[f"{phase}_ms"] for r in rows if f"{phase}_ms" in r)
        if not vals:
            continue
        out["phases"][phase] = {
            "n": len(vals), "mean_ms": statistics.fmean(vals), "median_ms": vals[len(vals) // 2],
            "p95_ms": vals[min(len(vals) - 1, int(0.95 * len(vals)))],
            "share_of_move": (sum(vals) / move_total) if move_total > 0 else None,
        }
    for key in ("games_moved", "simulations", "leaf_share"):
        vals = [r[key] for r in rows if key in r]
        if vals:
            out[key] = statistics.fmean(vals)
    if move_total > 0 and "games_moved" in out:
        out["moves_per_second"] = sum(r.get("games_moved", 0) for r in rows) / (move_total / 1e3)
    return out


def format_summary(s: dict[str, Any]) -> str:
    lines = [f"--- GPU actor profile: {s['file']} ({s['moves']} moves) ---"]
    if s.get("moves_per_second"):
        lines.append(f"self-play rate while moving: {s['moves_per_second']:.1f} game moves/s "
                     f"({s.get('games_moved', 0):.0f} games per move, leaf slots evaluated: {s.get('leaf_share', float('nan')):.3f})")
    lines.append(f"{'phase':<10} {'n':>6} {'mean ms':>10} {'median ms':>10} {'p95 ms':>10} {'share':>7}")
    for phase, p in s["phases"].items():
        share = f"{100 * p['share_of_move']:.1f}%" if p["share_of_move"] is not None else "-"
        lines.append(f"{phase:<10} {p['n']:>6} {p['mean_ms']:>10.3f} {p['median_ms']:>10.3f} {p['p95_ms']:>10.3f} {share:>7}")
    return "\n".join(lines)
