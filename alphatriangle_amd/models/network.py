"""Inference / weight-management wrapper around :class:`AlphaTriangleNet`.

API parity: alphatriangle/nn/network.py:32-336 (``evaluate_state``,
``evaluate_batch``, ``get_weights``, ``set_weights``, ``NetworkEvaluationError``,
C51 support/expectation).  Additions for this framework:

* array-returning fast paths (``evaluate_features``, ``evaluate_batch_arrays``) so
  the native searches never build per-leaf Python dicts;
* on an MI355X the forward pass of these paths runs on the hand-written HIP
  inference engine (ops/inference.py) fed by BN-folded bf16 weights that are
  re-packed whenever ``set_weights`` / ``sync_inference_weights`` is called.
"""

from __future__ import annotations

import logging
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F

from ..config.env_config import EnvConfig
from ..config.model_config import ModelConfig
from ..config.train_config import TrainConfig
from ..env.game_state import GameState
from ..features.extractor import extract_features_batch, extract_state_features
from ..utils.types import ActionType
from .model import AlphaTriangleNet

logger = logging.getLogger(__name__)


class NetworkEvaluationError(Exception):
    """Raised when inputs or outputs of a network evaluation are not finite / usable."""


class NeuralNetwork:
    def __init__(
        self,
        model_config: ModelConfig,
        env_config: EnvConfig,
        train_config: TrainConfig,
        device: torch.device,
        use_hip_engine: bool | None = None,
        inference_dtype: str = "bf16",
    ):
        self.model_config = model_config
        self.env_config = env_config
        self.train_config = train_config
        self.device = torch.device(device)
        self.action_dim = int(env_config.NUM_SHAPE_SLOTS * env_config.ROWS * env_config.COLS)

        self._orig_model = AlphaTriangleNet(model_config, env_config).to(self.device)
        self._orig_model.eval()
        # No tracing compiler on this platform: `model` is always the eager module;
        # the fast inference path is the HIP engine below.
        self.model: torch.nn.Module = self._orig_model

        self.num_atoms = model_config.NUM_VALUE_ATOMS
        self.v_min = model_config.VALUE_MIN
        self.v_max = model_config.VALUE_MAX
        self.delta_z = (self.v_max - self.v_min) / (self.num_atoms - 1)
        self.support = torch.linspace(self.v_min, self.v_max, self.num_atoms, device=self.device)

        self.engine: Any = None
        # the precision evaluate_* must deliver: an engine of another precision (the bf16 kernels loaded for a configuration
        # that has no fp32-grade ones) is kept for callers that ask for it, but is not what evaluate_features runs
        self.inference_dtype = inference_dtype
        want_engine = use_hip_engine if use_hip_engine is not None else self.device.type == "cuda"
        if want_engine and self.device.type == "cuda":
            from ..ops.inference import InferenceEngine, engine_supports

            ok, why = engine_supports(model_config, env_config, inference_dtype)
            if not ok and inference_dtype != "bf16":
                ok2, _ = engine_supports(model_config, env_config, "bf16")
                if ok2:
                    logger.warning("HIP inference engine has no %s path for this config (%s); its bf16 path is loaded for callers "
                                   "that ask for bf16, evaluate_* keeps to the PyTorch module in fp32.", inference_dtype, why)
                    ok, inference_dtype = True, "bf16"
            if ok:
                self.engine = InferenceEngine(model_config, env_config, self.device, dtype=inference_dtype)
                self.engine.load_from_module(self._orig_model)
                logger.info("HIP inference engine active (BN folded, %s).", self.engine.dtype)
            else:
                logger.warning("HIP inference engine not used for this config: %s", why)

    # ------------------------------------------------------------------ tensors
    def _check_finite(self, grid: torch.Tensor, other: torch.Tensor, what: str) -> None:
        if not torch.all(torch.isfinite(grid)):
            raise NetworkEvaluationError(f"Non-finite values found in input grid_tensor ({what})")
        if not torch.all(torch.isfinite(other)):
            raise NetworkEvaluationError(f"Non-finite values found in input other_features_tensor ({what})")

    def _state_to_tensors(self, state: GameState) -> tuple[torch.Tensor, torch.Tensor]:
        feats = extract_state_features(state, self.model_config)
        grid = torch.from_numpy(np.ascontiguousarray(feats["grid"])).unsqueeze(0).to(self.device)
        other = torch.from_numpy(np.ascontiguousarray(feats["other_features"])).unsqueeze(0).to(self.device)
        self._check_finite(grid, other, f"state {state}")
        return grid, other

    def _batch_states_to_tensors(self, states: list[GameState]) -> tuple[torch.Tensor, torch.Tensor]:
        mc, ec = self.model_config, self.env_config
        if not states:
            return (
                torch.empty((0, mc.GRID_INPUT_CHANNELS, ec.ROWS, ec.COLS), device=self.device),
                torch.empty((0, mc.OTHER_NN_INPUT_FEATURES_DIM), device=self.device),
            )
        feats = [extract_state_features(s, mc) for s in states]
        grid = torch.from_numpy(np.stack([f["grid"] for f in feats])).to(self.device)
        other = torch.from_numpy(np.stack([f["other_features"] for f in feats])).to(self.device)
        self._check_finite(grid, other, "batch")
        return grid, other

    def _logits_to_expected_value(self, value_logits: torch.Tensor) -> torch.Tensor:
        probs = F.softmax(value_logits, dim=1)
        return torch.sum(probs * self.support.expand_as(probs), dim=1, keepdim=True)

    # ------------------------------------------------------------------ fast array paths
    @torch.inference_mode()
    def evaluate_features(self, grid: torch.Tensor, other: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """``(policy_logits [B, A] f32, expected_value [B] f32)`` for feature tensors on ``self.device``."""
        if self.engine is not None and self.engine.dtype == self.inference_dtype:
            return self.engine.forward(grid, other)
        self.model.eval()
        logits, vlogits = self.model(grid.float(), other.float())
        return logits.float(), self._logits_to_expected_value(vlogits.float()).squeeze(1)

    @torch.inference_mode()
    def evaluate_batch_arrays(self, states: list[Any]) -> tuple[np.ndarray, np.ndarray]:
        """Evaluate native or wrapped states; returns ``(logits [n, A], values [n])`` as NumPy arrays."""
        ec = self.env_config
        grid_np, other_np = extract_features_batch(states, self.model_config, ec.ROWS, ec.COLS)
        if not (np.isfinite(grid_np).all() and np.isfinite(other_np).all()):
            raise NetworkEvaluationError("Non-finite values found in batched input features")
        grid = torch.from_numpy(grid_np).to(self.device)
        other = torch.from_numpy(other_np).to(self.device)
        logits, values = self.evaluate_features(grid, other)
        if not (torch.all(torch.isfinite(logits)) and torch.all(torch.isfinite(values))):
            raise NetworkEvaluationError("Non-finite network outputs in batch evaluation")
        return logits.cpu().numpy().astype(np.float32, copy=False), values.cpu().numpy().astype(np.float32, copy=False)

    # ------------------------------------------------------------------ reference-shaped API
    def _finalise_probs(self, probs: np.ndarray, state: GameState, what: str) -> np.ndarray:
        probs = np.maximum(probs, 0)
        total = float(np.sum(probs))
        if abs(total - 1.0) > 1e-5:
            logger.warning("Policy probabilities sum to %.6f for %s. Re-normalizing.", total, what)
            if total <= 1e-9:
                valid = state.valid_actions()
                if not valid:
                    raise NetworkEvaluationError(
                        f"Policy probability sum is near zero ({total}) for {what} with no valid actions."
                    )
                probs = np.zeros_like(probs)
                probs[[a for a in valid if 0 <= a < len(probs)]] = 1.0 / len(valid)
            else:
                probs = probs / total
        return probs

    @torch.inference_mode()
    def evaluate_state(self, state: GameState) -> tuple[dict[int, float], float]:
        """Dense ``{action: prob}`` over all actions plus the scalar expected value."""
        self.model.eval()
        try:
            grid, other = self._state_to_tensors(state)
            policy_logits, value_logits = self.model(grid, other)
            if not torch.all(torch.isfinite(policy_logits)):
                raise NetworkEvaluationError(f"Non-finite policy_logits detected for state {state}")
            if not torch.all(torch.isfinite(value_logits)):
                raise NetworkEvaluationError(f"Non-finite value_logits detected for state {state}")
            probs_t = F.softmax(policy_logits, dim=1)
            if not torch.all(torch.isfinite(probs_t)):
                raise NetworkEvaluationError(f"Non-finite policy probabilities after softmax for state {state}")
            probs = self._finalise_probs(probs_t.squeeze(0).cpu().numpy(), state, f"state {state.current_step}")
            value = float(self._logits_to_expected_value(value_logits).squeeze(0).item())
            policy: dict[ActionType, float] = {i: float(p) for i, p in enumerate(probs)}
            return policy, value
        except NetworkEvaluationError:
            raise
        except Exception as exc:
            logger.error("Exception during single evaluation for state %s: %s", state, exc, exc_info=True)
            raise NetworkEvaluationError(f"Evaluation failed for state {state}: {exc}") from exc

    @torch.inference_mode()
    def evaluate_batch(self, states: list[GameState]) -> list[tuple[dict[int, float], float]]:
        if not states:
            return []
        self.model.eval()
        try:
            grid, other = self._batch_states_to_tensors(states)
            policy_logits, value_logits = self.model(grid, other)
            if not torch.all(torch.isfinite(policy_logits)):
                raise NetworkEvaluationError("Non-finite policy_logits detected in batch evaluation")
            if not torch.all(torch.isfinite(value_logits)):
                raise NetworkEvaluationError("Non-finite value_logits detected in batch evaluation")
            probs_all = F.softmax(policy_logits, dim=1)
            if not torch.all(torch.isfinite(probs_all)):
                raise NetworkEvaluationError("Non-finite policy probabilities after softmax in batch")
            probs_np = probs_all.cpu().numpy()
            values_np = self._logits_to_expected_value(value_logits).squeeze(1).cpu().numpy()
            out: list[tuple[dict[int, float], float]] = []
            for i, state in enumerate(states):
                probs = self._finalise_probs(probs_np[i], state, f"batch sample {i}")
                out.append(({a: float(p) for a, p in enumerate(probs)}, float(values_np[i])))
            return out
        except NetworkEvaluationError:
            raise
        except Exception as exc:
            logger.error("Error during batch evaluation: %s", exc, exc_info=True)
            raise NetworkEvaluationError(f"Batch evaluation failed: {exc}") from exc

    # ------------------------------------------------------------------ weights
    def get_weights(self) -> dict[str, torch.Tensor]:
        """``state_dict`` of the underlying model, on CPU."""
        return {k: v.detach().cpu().clone() for k, v in self._orig_model.state_dict().items()}

    def set_weights(self, weights: dict[str, torch.Tensor]) -> None:
        try:
            # checked before anything is copied: load_state_dict copies the tensors that fit before it reports the others
            own = self._orig_model.state_dict()
            bad = [k for k in own if k not in weights or tuple(weights[k].shape) != tuple(own[k].shape)]
            bad += [k for k in weights if k not in own]
            if bad:
                raise ValueError(f"{len(bad)} parameter(s) differ in name or shape from this model, e.g. {bad[0]}")
            self._orig_model.load_state_dict({k: v.to(self.device) for k, v in weights.items()})
            self.model.eval()
            self.sync_inference_weights()
        except Exception as exc:
            logger.error("Error setting weights on NN instance: %s", exc)
            raise

    def sync_inference_weights(self) -> None:
        """Re-fold BatchNorm and re-pack bf16 weights for the HIP engine after a parameter update."""
        if self.engine is not None:
            self.engine.load_from_module(self._orig_model)
