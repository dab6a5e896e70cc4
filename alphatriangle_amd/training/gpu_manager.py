"""Self-play manager for the GPU actor behind the same interface the loop uses for CPU workers.

The reference's manager (alphatriangle/training/worker_manager.py:18-238) juggles N
remote workers; here one GPU actor advances thousands of games per call, writes
experiences directly into the HBM replay and reports aggregate counters.

One poll = one move of every game plus ``SystemConfig.TRAIN_STEPS_PER_MOVE`` learner steps.
The learner is launch-bound (hundreds of small kernels per step), so with
``SystemConfig.OVERLAP_LEARNER`` its steps are enqueued on a second HIP stream (default
priority) while the search of the move occupies the GPU:

  self-play stream: [weights re-packed] -> search of this move -> (learner done) -> commit into the replay
  learner stream  : (weights re-packed, previous commit done)  -> learner steps of this move

Weight publication requested by the loop during those steps is deferred to the move boundary,
because the search reads the packed inference weights until then.
"""

from __future__ import annotations

import logging
from typing import Any, Callable

import torch

from .components import TrainingComponents

logger = logging.getLogger(__name__)


class GpuActorManager:
    def __init__(self, components: TrainingComponents):
        self.components = components
        self.actor = components.gpu_actor
        self.sys = components.system_config
        self.train_fn: Callable[[int], int] | None = None  # set by the loop: run up to k learner steps, return how many ran
        self.ready_fn: Callable[[], bool] | None = None
        self.last_trained_steps = 0
        self._pending_sync_step: int | None = None
        self._in_move = False
        dev = self.actor.device
        self._overlap = bool(getattr(self.sys, "OVERLAP_LEARNER", True)) and dev.type == "cuda"
        self._train_stream = torch.cuda.Stream(device=dev) if self._overlap else None  # default priority
        self._ev_weights = torch.cuda.Event() if self._overlap else None
        self._ev_trained = torch.cuda.Event() if self._overlap else None

    def initialize_workers(self) -> None:
        logger.info("GPU actor ready: %d concurrent games (learner overlap: %s).", self.actor.G, self._overlap)

    def submit_initial_tasks(self) -> None:
        return

    def submit_task(self, worker_idx: int) -> None:  # noqa: ARG002 - the actor is always busy
        return

    def get_completed_tasks(self, timeout: float = 0.0) -> list[tuple[int, Any]]:  # noqa: ARG002
        """Advance every game by one move and run this move's learner steps (overlapped when enabled)."""
        k = int(self.sys.TRAIN_STEPS_PER_MOVE)
        can_train = self.train_fn is not None and k > 0 and (self.ready_fn is None or self.ready_fn())
        self._in_move = True
        try:
            if can_train and self._overlap:
                dev = self.actor.device
                main = torch.cuda.current_stream(dev)
                self._ev_weights.record(main)
                self.actor.begin_move()
                self._train_stream.wait_event(self._ev_weights)
                with torch.cuda.stream(self._train_stream):
                    self.last_trained_steps = self.train_fn(k)
                    self._ev_trained.record(self._train_stream)
                report = self.actor.finish_move(wait_event=self._ev_trained)
                main.wait_event(self._ev_trained)
            else:
                report = self.actor.play_move()
                self.last_trained_steps = self.train_fn(k) if can_train else 0
        finally:
            self._in_move = False
        if self._pending_sync_step is not None:
            self.actor.sync_weights(self._pending_sync_step)
            self._pending_sync_step = None
        return [(0, report)]

    def update_worker_networks(self, global_step: int) -> None:
        if self._in_move:
            self._pending_sync_step = int(global_step)  # applied at the move boundary
        else:
            self.actor.sync_weights(global_step)

    def get_num_active_workers(self) -> int:
        return 1

    def get_num_pending_tasks(self) -> int:
        return self.actor.G

    def cleanup_actors(self) -> None:
        return
