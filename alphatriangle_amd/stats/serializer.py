"""Checkpoint / replay-buffer (de)serialisation.

Format (alphatriangle/training/loop.py:193-206, alphatriangle/training/runner.py:43-79,
README.md:49,228-230): cloudpickled objects -
checkpoint ``{run_name, global_step, episodes_played, total_simulations_run,
model_state_dict, optimizer_state_dict, model_config_dict, env_config_dict}`` and
buffer ``{buffer_list: list[Experience]}`` with
``Experience = ({"grid": f32[C,H,W], "other_features": f32[D]}, {action: prob}, float)``.

Files are written as PLAIN dictionaries of builtins, tensors and arrays (field names above plus a
``format`` tag): the pickle stream then names no class of this project, so any stack that knows the
field names can read it.  Loading accepts that form, the pydantic objects earlier versions of this
package wrote, and streams that name the reference stack's classes (``trieye.schemas.CheckpointData``
/ ``BufferData`` / ``LoadedTrainingState``, resolved to the classes below by an aliasing unpickler;
alphatriangle/training/runner.py:36-79 is the reference's consumer of these objects).
"""

from __future__ import annotations

import io
import logging
import pickle
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


CHECKPOINT_FORMAT = "alphatriangle-checkpoint/1"
BUFFER_FORMAT = "alphatriangle-buffer/1"

# class paths other stacks (or earlier versions of this one) may have pickled by reference -> the classes above
_CLASS_ALIASES = {
    "CheckpointData": CheckpointData,
    "BufferData": BufferData,
    "LoadedTrainingState": LoadedTrainingState,
}
_ALIAS_MODULE_PREFIXES = ("trieye", "alphatriangle_amd.stats", "alphatriangle.")


class _AliasUnpickler(pickle.Unpickler):
    """``pickle.Unpickler`` that resolves the schema classes of the reference stack to the in-repo ones."""

    def find_class(self, module: str, name: str) -> Any:
        if name in _CLASS_ALIASES and module.startswith(_ALIAS_MODULE_PREFIXES):
            return _CLASS_ALIASES[name]
        return super().find_class(module, name)


def _load_pickle(path: Path) -> Any:
    with open(path, "rb") as fh:
        return _AliasUnpickler(io.BytesIO(fh.read())).load()


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
        payload = {"format": CHECKPOINT_FORMAT, **data.model_dump()}
        with open(tmp, "wb") as fh:
            cloudpickle.dump(payload, fh)
        tmp.replace(path)

    def load_checkpoint(self, path: Path) -> CheckpointData | None:
        try:
            obj = _load_pickle(path)
        except Exception as exc:
            logger.error("Failed to load checkpoint %s: %s", path, exc)
            return None
        return self._coerce_checkpoint(obj)

    def save_buffer(self, data: BufferData, path: Path) -> None:
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(path.suffix + ".tmp")
        with open(tmp, "wb") as fh:
            cloudpickle.dump({"format": BUFFER_FORMAT, "buffer_list": list(data.buffer_list)}, fh)
        tmp.replace(path)

    def load_buffer(self, path: Path) -> BufferData | None:
        try:
            obj = _load_pickle(path)
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
        if isinstance(obj, LoadedTrainingState):
            return obj.checkpoint_data
        inner = obj.get("checkpoint_data") if isinstance(obj, dict) else getattr(obj, "checkpoint_data", None)
        if inner is not None:  # a whole loaded-state object: the checkpoint is one level down
            return Serializer._coerce_checkpoint(inner)
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
