"""Shared type aliases (parity: alphatriangle/utils/types.py:8-56)."""

from __future__ import annotations

from collections.abc import Mapping

import numpy as np
from typing_extensions import TypedDict


class StateType(TypedDict):
    """Network input of one state: grid (C, H, W) float32 and other_features (D,) float32."""

    grid: np.ndarray
    other_features: np.ndarray


ActionType = int
PolicyTargetMapping = Mapping[ActionType, float]
Experience = tuple[StateType, PolicyTargetMapping, float]
ExperienceBatch = list[Experience]
PolicyValueOutput = tuple[dict[ActionType, float], float]


class PERBatchSample(TypedDict):
    """A sampled batch plus sum-tree indices and importance-sampling weights."""

    batch: ExperienceBatch
    indices: np.ndarray
    weights: np.ndarray
