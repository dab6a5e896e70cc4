"""Shared type aliases.

Two forms of the same data live side by side in this framework:

* the per-experience form the reference API speaks (``alphatriangle/utils/types.py:8-56``): one ``StateType`` dict, one
  sparse policy mapping and one n-step return per experience - used by the CPU self-play worker, the host replay
  buffer, checkpoints and every public method that the reference also has;
* the columnar form the GPU path works in: whole batches as dense arrays (``ExperienceColumns``), which is what the
  device tree commits into the HBM replay and what the experience shipper sends between ranks.
"""

from __future__ import annotations

from collections.abc import Mapping
from typing import NamedTuple

import numpy as np
from typing_extensions import TypedDict

# ---------------------------------------------------------------------------------------------- per-experience form
ActionType = int
"""Index into the flat action space (shape slot x row x column)."""

PolicyTargetMapping = Mapping[ActionType, float]
"""Search policy of one state: action -> normalised visit count (only visited actions appear)."""


class StateType(TypedDict):
    """Network input of one state: grid (C, H, W) float32 and other_features (D,) float32."""

    grid: np.ndarray
    other_features: np.ndarray


Experience = tuple[StateType, PolicyTargetMapping, float]
"""(features of s_t, search policy of s_t, n-step return from s_t): one replay entry."""

ExperienceBatch = list[Experience]

PolicyValueOutput = tuple[dict[ActionType, float], float]
"""What ``NeuralNetwork.evaluate_state`` returns: priors over the valid actions and the expected value."""


class PERBatchSample(TypedDict):
    """A sampled batch plus sum-tree indices and importance-sampling weights."""

    batch: ExperienceBatch
    indices: np.ndarray
    weights: np.ndarray


# ---------------------------------------------------------------------------------------------------- columnar form
class ExperienceColumns(NamedTuple):
    """``n`` experiences as four dense arrays (numpy or torch, host or device): grid ``[n, H*W]`` (or ``[n, C, H, W]``),
    other features ``[n, D]``, dense policy target ``[n, action_dim]`` (rows sum to 1) and n-step return ``[n]``.
    ``DeviceReplayBuffer.add_arrays(*columns)`` takes exactly these."""

    grid: object
    other: object
    policy: object
    ret: object


def experience_to_columns(batch: ExperienceBatch, action_dim: int) -> ExperienceColumns:
    """Per-experience form -> columnar form (numpy, float32); the sparse policies become dense rows."""
    n = len(batch)
    if n == 0:
        empty = np.zeros((0,), dtype=np.float32)
        return ExperienceColumns(empty.reshape(0, 0), empty.reshape(0, 0), np.zeros((0, action_dim), np.float32), empty)
    grid = np.stack([np.asarray(s["grid"], dtype=np.float32).reshape(-1) for s, _, _ in batch])
    other = np.stack([np.asarray(s["other_features"], dtype=np.float32) for s, _, _ in batch])
    policy = np.zeros((n, action_dim), dtype=np.float32)
    for i, (_, target, _) in enumerate(batch):
        if isinstance(target, np.ndarray):  # already dense (experiences that came back from the columnar form)
            policy[i, : min(action_dim, target.shape[0])] = target[:action_dim]
            continue
        for action, p in target.items():
            if 0 <= int(action) < action_dim:
                policy[i, int(action)] = float(p)
    ret = np.asarray([r for _, _, r in batch], dtype=np.float32)
    return ExperienceColumns(grid, other, policy, ret)
