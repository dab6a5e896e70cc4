"""HBM-resident replay buffer (SoA tensors + GPU sum tree).

Same behaviour as :class:`ExperienceBuffer` (alphatriangle/rl/core/buffer.py:25-203:
ring insert at max priority, stratified proportional sampling, ``(N*P(i))**-beta``
weights normalised by the batch maximum, ``(|err|+eps)**alpha`` priority updates,
linear beta annealing) but nothing leaves the device: the self-play commit kernel
writes finished experiences straight into these tensors, sampling / priority
updates are HIP kernels over a float64 sum tree, and a batch is gathered with
device indexing.  ``to_list`` / ``load_list`` convert to the reference's
list-of-tuples form for checkpoints.
"""

from __future__ import annotations

import logging

import numpy as np
import torch

from ...config.env_config import EnvConfig
from ...config.model_config import ModelConfig
from ...config.system_config import SystemConfig
from ...config.train_config import TrainConfig
from ...ops import hip_learner
from ...stats.serializer import experiences_from_arrays
from ...utils.types import Experience, experience_to_columns
from .trainer import TensorBatch

logger = logging.getLogger(__name__)


class DeviceReplayBuffer:
    def __init__(
        self,
        config: TrainConfig,
        env_config: EnvConfig,
        model_config: ModelConfig,
        device: torch.device,
        system_config: SystemConfig | None = None,
        emit_capacity: int = 1 << 16,
    ):
        self.config = config
        self.device = torch.device(device)
        self.capacity = int(config.BUFFER_CAPACITY)
        self.min_size_to_train = int(config.MIN_BUFFER_SIZE_TO_TRAIN)
        self.use_per = bool(config.USE_PER)
        self.per_alpha = config.PER_ALPHA
        self.per_beta_initial = config.PER_BETA_INITIAL
        self.per_beta_final = config.PER_BETA_FINAL
        self.per_beta_anneal_steps = max(1, config.PER_BETA_ANNEAL_STEPS or config.MAX_TRAINING_STEPS or 1)
        self.per_epsilon = config.PER_EPSILON
        self.cells = env_config.ROWS * env_config.COLS
        self.grid_shape = (model_config.GRID_INPUT_CHANNELS, env_config.ROWS, env_config.COLS)
        self.other_dim = model_config.OTHER_NN_INPUT_FEATURES_DIM
        self.action_dim = env_config.ACTION_DIM
        bytes_per = 4 * (self.cells * model_config.GRID_INPUT_CHANNELS + self.other_dim + self.action_dim + 1) + 16
        budget = (system_config.REPLAY_HBM_BUDGET_GB if system_config else 64.0) * (1 << 30)
        if self.capacity * bytes_per > budget:
            raise ValueError(
                f"replay of {self.capacity} experiences needs {self.capacity * bytes_per / 2**30:.1f} GiB, "
                f"above REPLAY_HBM_BUDGET_GB={budget / 2**30:.0f}"
            )
        dev = self.device
        cin = model_config.GRID_INPUT_CHANNELS
        self.grid = torch.zeros(self.capacity, cin * self.cells, dtype=torch.float32, device=dev)
        self.other = torch.zeros(self.capacity, self.other_dim, dtype=torch.float32, device=dev)
        self.policy = torch.zeros(self.capacity, self.action_dim, dtype=torch.float32, device=dev)
        self.ret = torch.zeros(self.capacity, dtype=torch.float32, device=dev)
        self.head = torch.zeros(1, dtype=torch.int64, device=dev)  # total experiences ever written
        self.emit_capacity = int(emit_capacity)
        self.emitted_idx = torch.zeros(self.emit_capacity, dtype=torch.int32, device=dev)
        self.emitted_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.P = 1
        while self.P < self.capacity:
            self.P *= 2
        self.tree = torch.zeros(2 * self.P, dtype=torch.float64, device=dev) if self.use_per else None
        self.max_prio = torch.ones(1, dtype=torch.float32, device=dev)
        self._written = 0  # host mirror of ``head``
        self._gen = torch.Generator(device=dev)
        self._gen.manual_seed(config.RANDOM_SEED + 7919)
        logger.info(
            "Device replay buffer: capacity %d (%.2f GiB in HBM), PER=%s", self.capacity, self.capacity * bytes_per / 2**30, self.use_per
        )

    # ------------------------------------------------------------------ size
    def __len__(self) -> int:
        return min(self._written, self.capacity)

    def is_ready(self) -> bool:
        return len(self) >= self.min_size_to_train

    def _calculate_beta(self, current_step: int) -> float:
        fraction = min(1.0, current_step / self.per_beta_anneal_steps)
        return self.per_beta_initial + fraction * (self.per_beta_final - self.per_beta_initial)

    # ------------------------------------------------------------------ device-side insertion (commit kernel)
    def begin_commit(self) -> None:
        self.emitted_count.zero_()

    def end_commit(self) -> int:
        """After the commit kernel: give the new leaves the max priority and account for them on the host."""
        n = int(self.emitted_count.item())
        if n > self.emit_capacity:
            logger.error("commit emitted %d experiences, above the index capacity %d; rebuilding the tree", n, self.emit_capacity)
        if n > 0 and self.use_per:
            assert self.tree is not None
            if n <= self.emit_capacity:
                hip_learner.per_set_leaves(self.tree, self.P, self.emitted_idx, self.emitted_count, self.emit_capacity, self.max_prio)
                hip_learner.per_update_paths(self.tree, self.P, self.emitted_idx, self.emitted_count, self.emit_capacity)
            else:
                self.tree[self.P : self.P + min(self._written + n, self.capacity)].clamp_(min=float(self.max_prio.item()))
                hip_learner.per_rebuild(self.tree, self.P)
        self._written += n
        return n

    # ------------------------------------------------------------------ host-side insertion (CPU workers, checkpoints)
    def add(self, experience: Experience) -> None:
        self.add_batch([experience])

    def add_batch(self, experiences: list[Experience]) -> None:
        if not experiences:
            return
        cols = experience_to_columns(experiences, self.action_dim)
        self.add_arrays(*(torch.from_numpy(np.ascontiguousarray(c)) for c in cols))

    def add_arrays(self, grid: torch.Tensor, other: torch.Tensor, policy: torch.Tensor, rets: torch.Tensor, priority: float | None = None) -> None:
        n = int(grid.shape[0])
        if n == 0:
            return
        if n > self.capacity:
            grid, other, policy, rets = grid[-self.capacity :], other[-self.capacity :], policy[-self.capacity :], rets[-self.capacity :]
            n = self.capacity
        dev = self.device
        slots = (torch.arange(n, dtype=torch.int64) + self._written) % self.capacity
        slots_d = slots.to(dev)
        self.grid[slots_d] = grid.to(dev).reshape(n, -1).float()
        self.other[slots_d] = other.to(dev).float()
        self.policy[slots_d] = policy.to(dev).float()
        self.ret[slots_d] = rets.to(dev).float()
        if self.use_per:
            assert self.tree is not None
            idx32 = slots_d.to(torch.int32)
            prio = self.max_prio if priority is None else torch.full((1,), float(priority), dtype=torch.float32, device=dev)
            for s in range(0, n, self.emit_capacity):
                chunk = idx32[s : s + self.emit_capacity].contiguous()
                hip_learner.per_set_leaves(self.tree, self.P, chunk, None, chunk.numel(), prio)
                hip_learner.per_update_paths(self.tree, self.P, chunk, None, chunk.numel())
        self._written += n
        self.head.fill_(self._written)

    # ------------------------------------------------------------------ sampling
    def sample(self, batch_size: int, current_train_step: int | None = None) -> TensorBatch | None:
        size = len(self)
        if size < batch_size or size < self.min_size_to_train:
            return None
        dev = self.device
        if self.use_per:
            if current_train_step is None:
                raise ValueError("current_train_step is required for PER sampling.")
            assert self.tree is not None
            rnd = torch.rand(batch_size, device=dev, generator=self._gen)
            idx = torch.empty(batch_size, dtype=torch.int32, device=dev)
            w = torch.empty(batch_size, dtype=torch.float32, device=dev)
            hip_learner.per_sample(self.tree, self.P, size, batch_size, rnd, self._calculate_beta(current_train_step), idx, w)
        else:
            idx = torch.randint(0, size, (batch_size,), device=dev, generator=self._gen).to(torch.int32)
            w = torch.ones(batch_size, dtype=torch.float32, device=dev)
        sel = idx.long()
        return TensorBatch(
            grid=self.grid.index_select(0, sel).view(batch_size, *self.grid_shape),
            other=self.other.index_select(0, sel),
            policy_target=self.policy.index_select(0, sel),
            returns=self.ret.index_select(0, sel),
            weights=w,
            indices=idx,
        )

    def update_priorities(self, indices, td_errors) -> None:  # noqa: ANN001
        if not self.use_per:
            return
        assert self.tree is not None
        dev = self.device
        idx = indices if isinstance(indices, torch.Tensor) else torch.as_tensor(np.asarray(indices), dtype=torch.int32)
        err = td_errors if isinstance(td_errors, torch.Tensor) else torch.as_tensor(np.asarray(td_errors), dtype=torch.float32)
        if idx.numel() != err.numel():
            logger.error("Mismatch between indices (%d) and td_errors (%d).", idx.numel(), err.numel())
            return
        if idx.numel() == 0:
            return
        idx = idx.to(dev, torch.int32).contiguous()
        err = err.to(dev, torch.float32).contiguous()
        hip_learner.per_priorities(self.tree, self.P, idx, err, self.per_alpha, self.per_epsilon, self.max_prio)
        hip_learner.per_update_paths(self.tree, self.P, idx, None, idx.numel())

    # ------------------------------------------------------------------ persistence
    def arrays(self) -> tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        n = len(self)
        return (
            self.grid[:n].cpu().numpy().reshape(n, *self.grid_shape),
            self.other[:n].cpu().numpy(),
            self.policy[:n].cpu().numpy(),
            self.ret[:n].cpu().numpy(),
        )

    def to_list(self) -> list[Experience]:
        return experiences_from_arrays(*self.arrays())

    def load_list(self, experiences: list[Experience]) -> None:
        self._written = 0
        self.head.zero_()
        if self.tree is not None:
            self.tree.zero_()
        self.max_prio.fill_(1.0)
        self.add_batch(list(experiences)[-self.capacity :])

    def total_priority(self) -> float:
        return float(self.tree[1].item()) if self.tree is not None else float(len(self))
