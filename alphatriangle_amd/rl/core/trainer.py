"""Learner: one optimisation step on a sampled batch.

Behavioural parity: alphatriangle/rl/core/trainer.py:27-323 (optimizer / scheduler
construction, batch densification, C51 two-hot value cross-entropy, policy
cross-entropy, unweighted entropy bonus, gradient-norm clipping, per-sample value
cross-entropy returned as the PER priority signal).

MI355X path: batches arrive as device tensors straight from the HBM replay
(:class:`TensorBatch`), the loss is one fused HIP kernel, the parameters live in a
flat buffer updated by one fused kernel, and with data-parallel learners a single
RCCL all-reduce of the flat gradient buffer runs between backward and the update.
With ``TrainConfig.COMPILE_MODEL`` the zero-grad + forward + loss + backward part of
the step (several hundred launches) is captured once into a HIP graph and replayed
when the graph passes its guards (:class:`GraphedForwardBackward`); the all-reduce
and the optimizer kernels stay eager, so the learning rate and Adam bias corrections
remain ordinary host scalars (6.7 ms eager -> 5.0 ms replayed per step at batch 256
on one MI355X).
"""

from __future__ import annotations

import ctypes as C
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
from ...ops import hip_learner
from ...ops import hipbind as hb
from ...ops.losses import LossOutputs, alphazero_loss, c51_two_hot
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


class GraphedForwardBackward:
    """Zero-grad + forward + loss + backward for one fixed batch shape as a replayable HIP graph.

    The graph reads five static input tensors and leaves the gradients in the flat gradient
    buffer plus five static outputs (total / policy / value loss, entropy, per-sample TD errors).
    Two guards decide whether it is ever replayed.  Structural: library reductions that spread
    over many blocks (torch's column sum for the bias gradients of the encoder's linear layers
    was the case in this network, which is why models/model.py computes them with its own
    kernel) record memset nodes for their semaphores, and such nodes do not replay reliably (a
    NaN in one bias-gradient element on the second new batch was the observed symptom), so a
    graph whose node census shows any memset node is refused outright.
    Numerical: the graph is replayed ``checks`` times, with eager forward/backward passes and
    device synchronisations in between, from the same parameters, BatchNorm buffers and RNG
    state as an eager pass (the graph-safe Philox state makes dropout masks identical), and
    every replay has to reproduce the eager gradients.  A refused graph leaves the trainer eager.
    """

    def __init__(self, trainer: "Trainer", tb: TensorBatch, *, warmup: int = 3, checks: int = 3, tol: float = 2e-2):
        self.trainer = trainer
        self.shape = tuple(tb.grid.shape)
        self.inputs = TensorBatch(tb.grid.clone(), tb.other.clone(), tb.policy_target.clone(), tb.returns.clone(), tb.weights.clone())
        self.outputs: tuple[torch.Tensor, ...] = ()
        self.graph: torch.cuda.CUDAGraph | None = None
        self.max_rel_err = float("nan")
        self.kernel_nodes = self.memset_nodes = self.other_nodes = 0
        model = trainer.model
        flat = trainer.flat
        assert flat is not None
        buffers = [b for b in model.buffers()]
        saved = [b.clone() for b in buffers]

        def restore() -> None:
            for b, s in zip(buffers, saved, strict=True):
                b.copy_(s)

        side = torch.cuda.Stream()  # default priority
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                trainer._forward_backward(self.inputs)
        torch.cuda.current_stream().wait_stream(side)
        restore()
        graph = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(graph):
            out = trainer._forward_backward(self.inputs)
            self.outputs = (out.total.detach(), out.policy_loss, out.value_loss, out.entropy, out.td_errors)
        del out
        restore()
        census = (C.c_int * 3)()
        hb.call("atri_graph_census", C.c_void_p(graph.raw_cuda_graph()), census)
        self.kernel_nodes, self.memset_nodes, self.other_nodes = int(census[0]), int(census[1]), int(census[2])
        if self.memset_nodes > 0:
            logger.warning(
                "Learner HIP graph holds %d memset nodes (next to %d kernel nodes); not replaying it, staying eager.",
                self.memset_nodes, self.kernel_nodes,
            )
            return
        graph.instantiate()
        worst = 0.0
        for _ in range(checks):
            rng = torch.cuda.get_rng_state(tb.grid.device)
            ref = trainer._forward_backward(self.inputs)
            ref_grad = flat.flat_grad.clone()
            ref_td = ref.td_errors.clone()
            torch.cuda.synchronize()
            restore()
            torch.cuda.set_rng_state(rng, tb.grid.device)
            graph.replay()
            torch.cuda.synchronize()
            restore()
            denom = float(torch.linalg.vector_norm(ref_grad)) + 1e-12
            err = float(torch.linalg.vector_norm(flat.flat_grad - ref_grad)) / denom
            err_td = float((self.outputs[4] - ref_td).abs().max()) / (float(ref_td.abs().max()) + 1e-12)
            worst = max(worst, err, err_td)
            if not (worst <= tol):  # also catches NaN
                break
        self.max_rel_err = worst
        if worst <= tol:
            self.graph = graph
        else:
            logger.warning("Learner HIP graph replay differs from the eager pass (rel err %.3g); staying eager.", worst)

    @property
    def ok(self) -> bool:
        return self.graph is not None

    def matches(self, tb: TensorBatch) -> bool:
        return tuple(tb.grid.shape) == self.shape

    def run(self, tb: TensorBatch) -> tuple[torch.Tensor, ...]:
        assert self.graph is not None
        i = self.inputs
        i.grid.copy_(tb.grid)
        i.other.copy_(tb.other)
        i.policy_target.copy_(tb.policy_target)
        i.returns.copy_(tb.returns)
        i.weights.copy_(tb.weights)
        self.graph.replay()
        return tuple(o.clone() for o in self.outputs)  # the static outputs are overwritten by the next replay


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
        use_graph: bool | None = None,
    ):
        self.nn = nn_interface
        self.model = nn_interface.model
        self.train_config = train_config
        self.env_config = env_config
        self.model_config = nn_interface.model_config
        self.device = nn_interface.device
        self.autocast_bf16 = bool(autocast_bf16 and self.device.type == "cuda")
        self.grad_sync = grad_sync
        self.sync_count = 0  # gradient all-reduces entered so far (a failed step is only skipped if it has not entered its own)
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
        # HIP-graph capture of forward+backward: needs the flat (static-address) gradient buffer
        self.use_graph = bool(
            use_graph if use_graph is not None else getattr(train_config, "COMPILE_MODEL", False)
        ) and self.device.type == "cuda" and self.flat is not None
        self._graphed: GraphedForwardBackward | None = None
        self._grad_seed: torch.Tensor | None = None
        self.graph_report: dict[str, Any] = {}  # outcome of the capture attempt (node census, validation error)

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
    def _forward_backward(self, tb: TensorBatch) -> LossOutputs:
        """Zero the gradients, run forward + loss + backward; gradients end up in ``p.grad`` / the flat buffer."""
        if self.flat is not None:
            self.flat.release_grads()  # every gradient is written, not accumulated: no zeroing, no per-parameter adds
        else:
            self.optimizer.zero_grad()
        pooled_seeds = self.device.type == "cuda" and hip_learner.available()
        if pooled_seeds:
            hip_learner.begin_step_seeds(self.device)  # one generator launch for all dropout sites of this pass
        core = getattr(self.nn, "_orig_model", self.model)
        prepacked = pooled_seeds and not self.autocast_bf16 and hasattr(core, "prepack_filters")
        if prepacked:
            core.prepack_filters()  # one launch for the filters of every direct convolution; valid until the weights change
        try:
            try:
                if self.autocast_bf16:
                    if self.flat is not None:
                        self.flat.refresh_shadow(torch.bfloat16)  # one cast of the flat buffer instead of one per weight
                    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                        policy_logits, value_logits = self.model(tb.grid, tb.other)
                else:
                    policy_logits, value_logits = self.model(tb.grid, tb.other)
            finally:
                if pooled_seeds:
                    hip_learner.end_step_seeds(self.device)
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
            if self.device.type == "cuda":
                if self._grad_seed is None:
                    self._grad_seed = torch.ones((), dtype=torch.float32, device=self.device)
                out.total.backward(self._grad_seed.to(out.total.dtype))  # a kept 1.0 instead of a fill launch per step
            else:
                out.total.backward()
        finally:
            if prepacked:
                core.release_packed_filters()
        if self.flat is not None:
            self.flat.gather_grads()
        return out

    def _graph_for(self, tb: TensorBatch) -> GraphedForwardBackward | None:
        """The validated graph for this batch shape, capturing it on first use; ``None`` means run eagerly."""
        if not self.use_graph:
            return None
        if self._graphed is None:
            if len(tb) != self.train_config.BATCH_SIZE:
                return None  # capture only the steady-state batch shape
            try:
                self._graphed = GraphedForwardBackward(self, tb)
            except Exception as exc:  # capture is an optimisation: any failure leaves the eager path
                logger.warning("Learner HIP graph capture failed (%s); staying eager.", exc)
                self.use_graph = False
                return None
            g = self._graphed
            self.graph_report = {
                "replayed": g.ok, "kernel_nodes": g.kernel_nodes, "memset_nodes": g.memset_nodes, "max_rel_err": g.max_rel_err,
            }
            if not g.ok:
                self.use_graph = False
                self._graphed = None
                return None
            logger.info("Learner forward/backward captured in a HIP graph (replay vs eager rel err %.2g).", self._graphed.max_rel_err)
        return self._graphed if self._graphed.matches(tb) else None

    def train_step_tensors(self, tb: TensorBatch) -> tuple[dict[str, torch.Tensor], torch.Tensor]:
        """One step on device tensors; returns device scalars and ``td_errors [B]`` without host syncs."""
        self.model.train()
        graphed = self._graph_for(tb)
        if graphed is not None:
            total, policy_loss, value_loss, entropy, td_errors = graphed.run(tb)
        else:
            out = self._forward_backward(tb)
            total, policy_loss, value_loss, entropy, td_errors = (
                out.total.detach(), out.policy_loss, out.value_loss, out.entropy, out.td_errors,
            )
        if self.grad_sync is not None and self.flat is not None:
            self.grad_sync(self.flat.flat_grad)
            self.sync_count += 1
        clip = self.train_config.GRADIENT_CLIP_VALUE
        if self.flat is None and clip is not None and clip > 0:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
        self.optimizer.step()
        if self.scheduler:
            self.scheduler.step()
        info = {
            "total_loss": total,
            "policy_loss": policy_loss,
            "value_loss": value_loss,
            "entropy": entropy,
            "mean_td_error": td_errors.abs().mean(),
        }
        return info, td_errors

    def skip_step(self) -> bool:
        """This rank's part of a data-parallel step it has no batch for: a zero gradient into the all-reduce, then the
        same optimizer and scheduler step as every other rank (which keeps the replicas identical).

        Peers that did get a batch are already waiting in the all-reduce of this step; a rank that simply returned would
        leave them there until the collective time-out.  Returns False when there is nothing to keep in step with
        (single learner, or no flat gradient buffer)."""
        if self.grad_sync is None or self.flat is None:
            return False
        self.flat.release_grads()
        self.flat.flat_grad.zero_()
        self.grad_sync(self.flat.flat_grad)
        self.sync_count += 1
        self.optimizer.step()
        if self.scheduler:
            self.scheduler.step()
        return True

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
