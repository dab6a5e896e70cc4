"""Checkpoint / replay-buffer (de)serialisation.

Format (alphatriangle/training/loop.py:193-206, alphatriangle/training/runner.py:43-79,
README.md:49,228-230): cloudpickled objects -
checkpoint ``{run_name, global_step, episodes_played, total_simulations_run,
model_state_dict, optimizer_state_dict, model_config_dict, env_config_dict}`` and
buffer ``{buffer_list: list[Experience]}`` with
``Experience = ({"grid": f32[C,H,W], "other_features": f32[D]}, {action: prob}, float)``.
Loading accepts these classes, plain dicts, or any object exposing the same attribute names.
"""

from __future__ import annotations

import logging
from pathlib import Path
from typing import Any

import cloudpickle
import numpy as np
import torch
from pydantic import BaseModel, ConfigDict, Field

from ..utils.types import Experience

logger = logging.getLogger(__name__)


class CheckpointData(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True, protected_namespaces=())

    run_name: str
    global_step: int = Field(..., ge=0)
    episodes_played: int = Field(..., ge=0)
    total_simulations_run: int = Field(..., ge=0)
    model_config_dict: dict[str, Any] = Field(default_factory=dict)
    env_config_dict: dict[str, Any] = Field(default_factory=dict)
    model_state_dict: dict[str, Any] = Field(default_factory=dict)
    optimizer_state_dict: dict[str, Any] = Field(default_factory=dict)
    actor_state: dict[str, Any] = Field(default_factory=dict)
    extra: dict[str, Any] = Field(default_factory=dict)


class BufferData(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)

    buffer_list: list[Any] = Field(default_factory=list)


class LoadedTrainingState(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)

    checkpoint_data: CheckpointData | None = None
    buffer_data: BufferData | None = None


for _m in (CheckpointData, BufferData, LoadedTrainingState):
    _m.model_rebuild(force=True)


def _to_cpu(obj: Any) -> Any:
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def _experience_ok(exp: Any) -> bool:
    if not (isinstance(exp, tuple) and len(exp) == 3):
        return False
    state, policy, value = exp
    if not (isinstance(state, dict) and "grid" in state and "other_features" in state):
        return False
    if not (isinstance(state["grid"], np.ndarray) and isinstance(state["other_features"], np.ndarray)):
        return False
    if not (np.all(np.isfinite(state["grid"])) and np.all(np.isfinite(state["other_features"]))):
        return False
    if not (hasattr(policy, "items") or isinstance(policy, np.ndarray)):
        return False
    try:
        return bool(np.isfinite(float(value)))
    except (TypeError, ValueError):
        return False


class Serializer:
    """CPU-ifies training state and reads / writes the pickle files."""

    def prepare_optimizer_state(self, optimizer: Any) -> dict[str, Any]:
        """Accepts an optimizer object or an already-materialised ``state_dict``; returns a CPU copy."""
        state = optimizer.state_dict() if hasattr(optimizer, "state_dict") else optimizer
        return _to_cpu(state) if isinstance(state, dict) else {}

    def prepare_buffer_data(self, buffer: Any) -> BufferData | None:
        """Accepts a buffer object (``to_list()``) or a list of experiences; drops malformed entries."""
        if buffer is None:
            return None
        if hasattr(buffer, "to_list"):
            items = buffer.to_list()
        elif isinstance(buffer, (list, tuple)):
            items = list(buffer)
        else:
            logger.error("prepare_buffer_data: unsupported buffer type %s", type(buffer).__name__)
            return None
        good = [e for e in items if _experience_ok(e)]
        if len(good) != len(items):
            logger.warning("Dropped %d malformed experiences while preparing the buffer.", len(items) - len(good))
        return BufferData(buffer_list=good)

    # -- files
    def save_checkpoint(self, data: CheckpointData, path: Path) -> None:
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(path.suffix + ".tmp")
        with open(tmp, "wb") as fh:
            cloudpickle.dump(data, fh)
        tmp.replace(path)

    def load_checkpoint(self, path: Path) -> CheckpointData | None:
        try:
            with open(path, "rb") as fh:
                obj = cloudpickle.load(fh)
        except Exception as exc:
            logger.error("Failed to load checkpoint %s: %s", path, exc)
            return None
        return self._coerce_checkpoint(obj)

    def save_buffer(self, data: BufferData, path: Path) -> None:
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(path.suffix + ".tmp")
        with open(tmp, "wb") as fh:
            cloudpickle.dump(data, fh)
        tmp.replace(path)

    def load_buffer(self, path: Path) -> BufferData | None:
        try:
            with open(path, "rb") as fh:
                obj = cloudpickle.load(fh)
        except Exception as exc:
            logger.error("Failed to load buffer %s: %s", path, exc)
            return None
        items = obj.get("buffer_list") if isinstance(obj, dict) else getattr(obj, "buffer_list", None)
        if items is None:
            logger.error("Buffer file %s has no buffer_list.", path)
            return None
        good = [e for e in items if _experience_ok(e)]
        return BufferData(buffer_list=good)

    @staticmethod
    def _coerce_checkpoint(obj: Any) -> CheckpointData | None:
        if isinstance(obj, CheckpointData):
            return obj
        get = (lambda k, d=None: obj.get(k, d)) if isinstance(obj, dict) else (lambda k, d=None: getattr(obj, k, d))
        try:
            return CheckpointData(
                run_name=str(get("run_name", "unknown")),
                global_step=int(get("global_step", 0)),
                episodes_played=int(get("episodes_played", 0)),
                total_simulations_run=int(get("total_simulations_run", 0)),
                model_config_dict=dict(get("model_config_dict", {}) or {}),
                env_config_dict=dict(get("env_config_dict", {}) or {}),
                model_state_dict=dict(get("model_state_dict", {}) or {}),
                optimizer_state_dict=dict(get("optimizer_state_dict", {}) or {}),
                actor_state=dict(get("actor_state", {}) or {}),
            )
        except Exception as exc:
            logger.error("Checkpoint object could not be interpreted: %s", exc)
            return None


def experiences_from_arrays(
    grid: np.ndarray, other: np.ndarray, policy: np.ndarray, returns: np.ndarray
) -> list[Experience]:
    """SoA arrays (as held in the HBM replay) -> list of reference-format experiences."""
    out: list[Experience] = []
    for i in range(grid.shape[0]):
        nz = np.nonzero(policy[i])[0]
        target = {int(a): float(policy[i, a]) for a in nz}
        out.append(({"grid": grid[i].copy(), "other_features": other[i].copy()}, target, float(returns[i])))
    return out
