"""Experience replay (uniform or prioritized) holding reference-format experiences.

Behavioural parity: alphatriangle/rl/core/buffer.py:25-203 - max-priority insert,
stratified proportional sampling, importance weights ``(N*P(i))**-beta`` normalised
by the batch maximum, linear beta annealing, ``(|err|+eps)**alpha`` priority updates.
The sum-tree walks are batched native calls instead of per-sample Python loops.
The HBM-resident SoA replay used on GPUs lives in rl/core/device_buffer.py and
converts to / from this list-of-tuples form for checkpoints.
"""

from __future__ import annotations

import logging
import random
from collections import deque

import numpy as np

from ...config.train_config import TrainConfig
from ...utils.sumtree import SumTree
from ...utils.types import Experience, ExperienceBatch, PERBatchSample

logger = logging.getLogger(__name__)


class ExperienceBuffer:
    def __init__(self, config: TrainConfig):
        self.config = config
        self.capacity = config.BUFFER_CAPACITY
        self.min_size_to_train = config.MIN_BUFFER_SIZE_TO_TRAIN
        self.use_per = config.USE_PER
        self._rng = np.random.default_rng(config.RANDOM_SEED + 7919)
        if self.use_per:
            self.tree = SumTree(self.capacity)
            self.per_alpha = config.PER_ALPHA
            self.per_beta_initial = config.PER_BETA_INITIAL
            self.per_beta_final = config.PER_BETA_FINAL
            self.per_beta_anneal_steps = max(1, config.PER_BETA_ANNEAL_STEPS or config.MAX_TRAINING_STEPS or 1)
            self.per_epsilon = config.PER_EPSILON
            logger.info(
                "Experience buffer initialized with PER (alpha=%s, beta_init=%s). Capacity: %d",
                self.per_alpha,
                self.per_beta_initial,
                self.capacity,
            )
        else:
            self.buffer: deque[Experience] = deque(maxlen=self.capacity)
            logger.info("Experience buffer initialized with uniform sampling. Capacity: %d", self.capacity)

    def _get_priority(self, error: float) -> float:
        return float((np.abs(error) + self.per_epsilon) ** self.per_alpha)

    def add(self, experience: Experience) -> None:
        if self.use_per:
            self.tree.add(self.tree.max_priority, experience)
        else:
            self.buffer.append(experience)

    def add_batch(self, experiences: list[Experience]) -> None:
        if self.use_per:
            max_p = self.tree.max_priority
            for exp in experiences:
                self.tree.add(max_p, exp)
        else:
            self.buffer.extend(experiences)

    def _calculate_beta(self, current_step: int) -> float:
        fraction = min(1.0, current_step / self.per_beta_anneal_steps)
        return self.per_beta_initial + fraction * (self.per_beta_final - self.per_beta_initial)

    def sample(self, batch_size: int, current_train_step: int | None = None) -> PERBatchSample | None:
        size = len(self)
        if size < batch_size or size < self.min_size_to_train:
            return None
        if not self.use_per:
            picked = random.sample(list(self.buffer), batch_size)
            return {
                "batch": picked,
                "indices": np.zeros(batch_size, dtype=np.int32),
                "weights": np.ones(batch_size, dtype=np.float32),
            }
        if current_train_step is None:
            raise ValueError("current_train_step is required for PER sampling.")
        beta = self._calculate_beta(current_train_step)
        total = self.tree.total_priority
        segment = total / batch_size
        values = (np.arange(batch_size) + self._rng.random(batch_size)) * segment
        leaf_idx, priorities = self.tree.get_leaf_batch(values)
        data_idx = leaf_idx - (self.capacity - 1)
        batch: ExperienceBatch = []
        keep = np.ones(batch_size, dtype=bool)
        for i in range(batch_size):
            item = self.tree.data[data_idx[i]]
            if not isinstance(item, tuple):
                # empty slot reached through rounding: redraw over the whole mass
                redraw = self._rng.random() * total
                li, pi = self.tree.get_leaf_batch(np.array([redraw]))
                item = self.tree.data[int(li[0]) - (self.capacity - 1)]
                if not isinstance(item, tuple):
                    logger.error("PER resampling failed. Skipping sample %d.", i)
                    keep[i] = False
                    continue
                leaf_idx[i], priorities[i] = li[0], pi[0]
            batch.append(item)
        leaf_idx, priorities = leaf_idx[keep], priorities[keep]
        probs = priorities / total if total > 0 else np.zeros_like(priorities)
        with np.errstate(divide="ignore", invalid="ignore"):
            weights = np.where(probs > 1e-9, (size * probs) ** (-beta), 0.0)
        max_w = float(weights.max()) if weights.size else 0.0
        if max_w > 1e-9:
            weights = weights / max_w
        else:
            logger.warning("Max importance sampling weight is near zero. Using unit weights.")
            weights = np.ones_like(weights)
        return {"batch": batch, "indices": leaf_idx.astype(np.int32), "weights": weights.astype(np.float32)}

    def update_priorities(self, tree_indices: np.ndarray, td_errors: np.ndarray) -> None:
        if not self.use_per:
            return
        if len(tree_indices) != len(td_errors):
            logger.error("Mismatch between tree_indices (%d) and td_errors (%d).", len(tree_indices), len(td_errors))
            return
        if len(tree_indices) == 0:
            return
        errs = np.asarray(td_errors, dtype=np.float64)
        with np.errstate(invalid="ignore", over="ignore"):
            priorities = (np.abs(errs) + self.per_epsilon) ** self.per_alpha
        if not np.all(np.isfinite(priorities)):
            logger.warning("Non-finite priorities calculated. Clamping.")
            priorities = np.nan_to_num(
                priorities, nan=self.per_epsilon, posinf=self.tree.max_priority, neginf=self.per_epsilon
            )
            priorities = np.maximum(priorities, self.per_epsilon)
        idx = np.asarray(tree_indices, dtype=np.int64)
        ok = (idx >= 0) & (idx < len(self.tree.tree))
        if not ok.all():
            logger.error("Invalid tree indices in priority update: %s", idx[~ok][:8])
        self.tree.update_batch(idx[ok], priorities[ok])
        self.tree._max_priority = max(self.tree.max_priority, float(np.max(priorities)))

    # ------------------------------------------------------------------ persistence helpers
    def to_list(self) -> list[Experience]:
        """All stored experiences (oldest first for the uniform buffer)."""
        if self.use_per:
            return [self.tree.data[i] for i in range(self.tree.n_entries) if isinstance(self.tree.data[i], tuple)]
        return list(self.buffer)

    def load_list(self, experiences: list[Experience]) -> None:
        """Replace the content; PER priorities restart at the maximum (they are not persisted)."""
        if self.use_per:
            self.tree = SumTree(self.capacity)
            for exp in experiences[-self.capacity :]:
                self.tree.add(1.0, exp)
        else:
            self.buffer = deque(experiences, maxlen=self.capacity)

    def __len__(self) -> int:
        return self.tree.n_entries if self.use_per else len(self.buffer)

    def is_ready(self) -> bool:
        return len(self) >= self.min_size_to_train
