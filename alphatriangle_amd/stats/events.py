"""Metric event schema (parity: ``trieye.RawMetricEvent`` as used at
alphatriangle/rl/self_play/worker.py:147-153 and alphatriangle/training/loop.py:103-109)."""

from __future__ import annotations

import math
import time
from typing import Any

from pydantic import BaseModel, Field


class RawMetricEvent(BaseModel):
    name: str
    value: float | int
    global_step: int
    timestamp: float = Field(default_factory=time.time)
    context: dict[str, Any] = Field(default_factory=dict)

    def is_valid(self) -> bool:
        try:
            return math.isfinite(float(self.value))
        except (TypeError, ValueError):
            return False


RawMetricEvent.model_rebuild(force=True)
