"""Self-play manager for the GPU actor behind the same interface the loop uses for CPU workers.

The reference's manager (alphatriangle/training/worker_manager.py:18-238) juggles N
remote workers; here one GPU actor advances thousands of games per call, writes
experiences directly into the HBM replay and reports aggregate counters.
"""

from __future__ import annotations

import logging
from typing import Any

from .components import TrainingComponents

logger = logging.getLogger(__name__)


class GpuActorManager:
    def __init__(self, components: TrainingComponents):
        self.components = components
        self.actor = components.gpu_actor
        self.moves_per_poll = 1
        self._started = False

    def initialize_workers(self) -> None:
        logger.info("GPU actor ready: %d concurrent games.", self.actor.G)

    def submit_initial_tasks(self) -> None:
        self._started = True

    def submit_task(self, worker_idx: int) -> None:  # noqa: ARG002 - the actor is always busy
        return

    def get_completed_tasks(self, timeout: float = 0.0) -> list[tuple[int, Any]]:  # noqa: ARG002
        """Advance every game by one move; the loop interleaves learner steps between calls."""
        out: list[tuple[int, Any]] = []
        for _ in range(self.moves_per_poll):
            out.append((0, self.actor.play_move()))
        return out

    def update_worker_networks(self, global_step: int) -> None:
        self.actor.sync_weights(global_step)

    def get_num_active_workers(self) -> int:
        return 1

    def get_num_pending_tasks(self) -> int:
        return self.actor.G

    def cleanup_actors(self) -> None:
        return
