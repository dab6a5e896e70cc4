"""Self-play source backed by the GPU-resident actor: one scheduler round = one move of every game.

The reference's manager (alphatriangle/training/worker_manager.py:18-238) juggles N remote
workers; here one actor advances thousands of games per round, writes experiences directly
into the HBM replay and reports aggregate counters.

The learner is launch-bound (hundreds of small kernels per step), so with
``SystemConfig.OVERLAP_LEARNER`` the round's learner steps are enqueued on a second HIP stream
(default priority) while the search of the move occupies the GPU:

  self-play stream: [weights re-packed] -> search of this move -> (learner done) -> commit into the replay
  learner stream  : (weights re-packed, previous commit done)  -> learner steps of this round

A weight publication that falls due during those steps is applied at the move boundary,
because the search reads the packed inference weights until then.
"""

from __future__ import annotations

import logging
from typing import Any, Callable

import torch

from .components import TrainingComponents

logger = logging.getLogger(__name__)


class GpuMoveSource:
    def __init__(self, components: TrainingComponents):
        self.components = components
        self.actor = components.gpu_actor
        self.sys = components.system_config
        self.steps_per_round = int(self.sys.TRAIN_STEPS_PER_MOVE)
        self.last_trained_steps = 0
        self._publish_due: int | None = None
        self._in_move = False
        dev = self.actor.device
        self._overlap = bool(getattr(self.sys, "OVERLAP_LEARNER", True)) and dev.type == "cuda"
        self._train_stream = torch.cuda.Stream(device=dev) if self._overlap else None  # default priority
        self._ev_weights = torch.cuda.Event() if self._overlap else None
        self._ev_trained = torch.cuda.Event() if self._overlap else None

    def open(self) -> None:
        logger.info("GPU actor ready: %d concurrent games, %s leaf evaluation (learner overlap: %s).",
                    self.actor.G, self.actor.dtype, self._overlap)

    def play_round(self, quota: int, learn: Callable[[int], int]) -> list[tuple[int, Any]]:
        """Advance every game by one move; run ``learn(quota)`` concurrently with the search when overlap is on."""
        self._in_move = True
        try:
            if quota > 0 and self._overlap:
                main = torch.cuda.current_stream(self.actor.device)
                self._ev_weights.record(main)
                self.actor.begin_move()
                self._train_stream.wait_event(self._ev_weights)
                prof = getattr(self.actor, "profiler", None)
                with torch.cuda.stream(self._train_stream):
                    if prof is not None:
                        prof.mark("learner_start", self._train_stream)
                    self.last_trained_steps = learn(quota)
                    if prof is not None:
                        prof.mark("learner_end", self._train_stream)
                    self._ev_trained.record(self._train_stream)
                report = self.actor.finish_move(wait_event=self._ev_trained)
                main.wait_event(self._ev_trained)
            else:
                report = self.actor.play_move()
                self.last_trained_steps = learn(quota) if quota > 0 else 0
        finally:
            self._in_move = False
        if self._publish_due is not None:
            self.actor.sync_weights(self._publish_due)
            self._publish_due = None
        return [(0, report)]

    def publish(self, step: int) -> None:
        if self._in_move:
            self._publish_due = int(step)  # applied at the move boundary
        else:
            self.actor.sync_weights(step)

    def active(self) -> int:
        return 1

    def pending(self) -> int:
        return self.actor.G

    def close(self) -> None:
        return
