"""N-step return bookkeeping shared by the CPU worker and (as the host twin) the HIP commit kernel.

Target of the experience created at move ``k``:
``G_k = sum_{i<n} gamma^i r_{k+i} + gamma^n V(s_{k+n})`` when the game is still
running ``n`` moves later, else the plain discounted sum to the end of the game
(alphatriangle/rl/self_play/worker.py:410-440,466-485).  Every experience is emitted
exactly once (the reference emits the oldest window entry twice when an episode
ends on a full window).
"""

from __future__ import annotations

from collections import deque
from dataclasses import dataclass
from typing import Any


@dataclass
class _Pending:
    payload: Any
    acc: float = 0.0
    count: int = 0


class NStepAccumulator:
    def __init__(self, n_step: int, gamma: float):
        self.n = int(n_step)
        self.gamma = float(gamma)
        self._pending: deque[_Pending] = deque()

    def push(self, payload: Any) -> None:
        """Register the experience of the move about to be played."""
        self._pending.append(_Pending(payload))

    def add_reward(self, reward: float) -> None:
        """Credit the reward of the move just played to every pending experience."""
        for p in self._pending:
            p.acc += (self.gamma**p.count) * reward
            p.count += 1

    def pop_ready(self, bootstrap_value: float) -> list[tuple[Any, float]]:
        """Experiences whose window is complete, finalised with ``gamma^n * V(current state)``."""
        out: list[tuple[Any, float]] = []
        while self._pending and self._pending[0].count >= self.n:
            p = self._pending.popleft()
            out.append((p.payload, p.acc + (self.gamma**self.n) * bootstrap_value))
        return out

    def flush(self) -> list[tuple[Any, float]]:
        """Episode end: everything still pending gets its truncated discounted sum (no bootstrap)."""
        out = [(p.payload, p.acc) for p in self._pending]
        self._pending.clear()
        return out

    def __len__(self) -> int:
        return len(self._pending)
