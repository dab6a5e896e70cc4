"""Learner: one optimisation step on a sampled batch.

Behavioural parity: alphatriangle/rl/core/trainer.py:27-323 (optimizer / scheduler
construction, batch densification, C51 two-hot value cross-entropy, policy
cross-entropy, unweighted entropy bonus, gradient-norm clipping, per-sample value
cross-entropy returned as the PER priority signal).

MI355X path: batches arrive as device tensors straight from the HBM replay
(:class:`TensorBatch`), the loss is one fused HIP kernel, the parameters live in a
flat buffer updated by one fused kernel, and with data-parallel learners a single
RCCL all-reduce of the flat gradient buffer runs between backward and the update.
"""

from __future__ import annotations

import logging
from collections.abc import Callable
from dataclasses import dataclass
from typing import Any

import numpy as np
import torch
from torch.optim.lr_scheduler import CosineAnnealingLR, StepLR, _LRScheduler

from ...config.env_config import EnvConfig
from ...config.train_config import TrainConfig
from ...models.network import NeuralNetwork
from ...ops.losses import alphazero_loss, c51_two_hot
from ...ops.optim import FlatParameterSpace, build_optimizer
from ...utils.types import ExperienceBatch, PERBatchSample

logger = logging.getLogger(__name__)


@dataclass
class TensorBatch:
    """A training batch already resident on the learner device."""

    grid: torch.Tensor           # [B, C, H, W] f32
    other: torch.Tensor          # [B, D] f32
    policy_target: torch.Tensor  # [B, A] f32
    returns: torch.Tensor        # [B] f32
    weights: torch.Tensor        # [B] f32 importance-sampling weights
    indices: torch.Tensor | None = None  # [B] replay slots / tree indices

    def __len__(self) -> int:
        return int(self.grid.shape[0])


class Trainer:
    def __init__(
        self,
        nn_interface: NeuralNetwork,
        train_config: TrainConfig,
        env_config: EnvConfig,
        *,
        fused: bool | None = None,
        autocast_bf16: bool = False,
        grad_sync: Callable[[torch.Tensor], None] | None = None,
        world_size: int = 1,
    ):
        self.nn = nn_interface
        self.model = nn_interface.model
        self.train_config = train_config
        self.env_config = env_config
        self.model_config = nn_interface.model_config
        self.device = nn_interface.device
        self.autocast_bf16 = bool(autocast_bf16 and self.device.type == "cuda")
        self.grad_sync = grad_sync
        self.world_size = int(world_size)
        use_fused = fused if fused is not None else (self.device.type == "cuda" or grad_sync is not None)
        self.optimizer, self.flat = self._create_optimizer(use_fused)
        if self.flat is not None and self.world_size > 1:
            self.optimizer.grad_scale = 1.0 / self.world_size  # type: ignore[attr-defined]
        self.scheduler = self._create_scheduler(self.optimizer)
        self.num_atoms = self.model_config.NUM_VALUE_ATOMS
        self.v_min = self.model_config.VALUE_MIN
        self.v_max = self.model_config.VALUE_MAX
        self.delta_z = (self.v_max - self.v_min) / (self.num_atoms - 1)
        self.action_dim = int(env_config.NUM_SHAPE_SLOTS * env_config.ROWS * env_config.COLS)

    # ------------------------------------------------------------------ construction
    def _create_optimizer(self, fused: bool) -> tuple[torch.optim.Optimizer, FlatParameterSpace | None]:
        logger.info(
            "Creating optimizer: %s, LR: %s, WD: %s%s",
            self.train_config.OPTIMIZER_TYPE,
            self.train_config.LEARNING_RATE,
            self.train_config.WEIGHT_DECAY,
            " (fused flat buffer)" if fused else "",
        )
        return build_optimizer(self.nn._orig_model, self.train_config, fused)

    def _create_scheduler(self, optimizer: torch.optim.Optimizer) -> _LRScheduler | None:
        kind = self.train_config.LR_SCHEDULER_TYPE
        if not kind or str(kind).lower() == "none":
            logger.info("No LR scheduler configured.")
            return None
        if kind == "StepLR":
            return StepLR(
                optimizer,
                step_size=getattr(self.train_config, "LR_SCHEDULER_STEP_SIZE", 100_000),
                gamma=getattr(self.train_config, "LR_SCHEDULER_GAMMA", 0.1),
            )
        if kind == "CosineAnnealingLR":
            t_max = self.train_config.LR_SCHEDULER_T_MAX
            if t_max is None:
                t_max = self.train_config.MAX_TRAINING_STEPS or 1_000_000
                logger.warning("LR_SCHEDULER_T_MAX is None for CosineAnnealingLR. Using %d.", t_max)
            return CosineAnnealingLR(optimizer, T_max=t_max, eta_min=self.train_config.LR_SCHEDULER_ETA_MIN)
        raise ValueError(f"Unsupported scheduler type: {kind}")

    # ------------------------------------------------------------------ batch handling
    def _prepare_batch(self, batch: ExperienceBatch) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """List of reference-format experiences -> ``(grid, other, policy_target, n_step_return)`` on device."""
        n = len(batch)
        policy = np.zeros((n, self.action_dim), dtype=np.float32)
        grids, others, rets = [], [], np.zeros(n, dtype=np.float32)
        for i, (state, target, ret) in enumerate(batch):
            grids.append(state["grid"])
            others.append(state["other_features"])
            rets[i] = ret
            if isinstance(target, np.ndarray):
                policy[i, : min(self.action_dim, target.shape[0])] = target[: self.action_dim]
                continue
            acts = np.fromiter(target.keys(), dtype=np.int64, count=len(target))
            probs = np.fromiter(target.values(), dtype=np.float32, count=len(target))
            ok = (acts >= 0) & (acts < self.action_dim)
            if not ok.all():
                logger.warning("Out-of-bounds actions in policy target map for sample %d.", i)
            policy[i, acts[ok]] = probs[ok]
        grid_t = torch.from_numpy(np.stack(grids)).to(self.device) if n else torch.empty(0, device=self.device)
        other_t = torch.from_numpy(np.stack(others)).to(self.device) if n else torch.empty(0, device=self.device)
        expected = self.model_config.OTHER_NN_INPUT_FEATURES_DIM
        if n > 0 and other_t.shape[1] != expected:
            raise ValueError(f"Unexpected other_features tensor shape: {tuple(other_t.shape)}, expected dim {expected}")
        return grid_t, other_t, torch.from_numpy(policy).to(self.device), torch.from_numpy(rets).to(self.device)

    def _calculate_target_distribution(self, n_step_returns: torch.Tensor) -> torch.Tensor:
        return c51_two_hot(n_step_returns, self.v_min, self.v_max, self.num_atoms)

    # ------------------------------------------------------------------ the step
    def train_step_tensors(self, tb: TensorBatch) -> tuple[dict[str, torch.Tensor], torch.Tensor]:
        """One step on device tensors; returns device scalars and ``td_errors [B]`` without host syncs."""
        self.model.train()
        self.optimizer.zero_grad()
        if self.autocast_bf16:
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                policy_logits, value_logits = self.model(tb.grid, tb.other)
        else:
            policy_logits, value_logits = self.model(tb.grid, tb.other)
        out = alphazero_loss(
            policy_logits,
            value_logits,
            tb.policy_target,
            tb.returns,
            tb.weights,
            v_min=self.v_min,
            v_max=self.v_max,
            policy_weight=self.train_config.POLICY_LOSS_WEIGHT,
            value_weight=self.train_config.VALUE_LOSS_WEIGHT,
            entropy_weight=self.train_config.ENTROPY_BONUS_WEIGHT,
        )
        out.total.backward()
        if self.grad_sync is not None and self.flat is not None:
            self.grad_sync(self.flat.flat_grad)
        clip = self.train_config.GRADIENT_CLIP_VALUE
        if self.flat is None and clip is not None and clip > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        self.optimizer.step()
        if self.scheduler:
            self.scheduler.step()
        info = {
            "total_loss": out.total.detach(),
            "policy_loss": out.policy_loss,
            "value_loss": out.value_loss,
            "entropy": out.entropy,
            "mean_td_error": out.td_errors.abs().mean(),
        }
        return info, out.td_errors

    def train_step(self, per_sample: PERBatchSample | TensorBatch) -> tuple[dict[str, float], np.ndarray] | None:
        """Reference-shaped step: returns ``(loss_info floats, td_errors ndarray)`` or ``None`` on an empty batch."""
        if isinstance(per_sample, TensorBatch):
            if len(per_sample) == 0:
                logger.warning("train_step called with empty batch.")
                return None
            tb = per_sample
        else:
            batch = per_sample["batch"]
            if not batch:
                logger.warning("train_step called with empty batch.")
                return None
            try:
                grid_t, other_t, policy_t, ret_t = self._prepare_batch(batch)
                weights_t = torch.from_numpy(np.asarray(per_sample["weights"], dtype=np.float32)).to(self.device)
            except Exception as exc:
                logger.error("Error preparing batch for training: %s", exc, exc_info=True)
                return None
            tb = TensorBatch(grid_t, other_t, policy_t, ret_t, weights_t)
        info_t, td = self.train_step_tensors(tb)
        td_np = td.detach().float().cpu().numpy()
        info = {k: float(v.item()) if isinstance(v, torch.Tensor) else float(v) for k, v in info_t.items()}
        return info, td_np

    def get_current_lr(self) -> float:
        try:
            return float(self.optimizer.param_groups[0]["lr"])
        except (IndexError, KeyError, AttributeError) as exc:
            logger.warning("Could not retrieve learning rate from optimizer: %s", exc)
            return 0.0

    def load_optimizer_state(self, state_dict: dict[str, Any]) -> None:
        """Load a saved optimizer ``state_dict`` and move its tensors to the learner device."""
        self.optimizer.load_state_dict(state_dict)
        for state in self.optimizer.state.values():
            for k, v in state.items():
                if isinstance(v, torch.Tensor) and k != "step" and v.device != self.device:
                    state[k] = v.to(self.device)
