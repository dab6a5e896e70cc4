"""Result record of one self-play episode (parity: alphatriangle/rl/types.py:14-89)."""

from __future__ import annotations

import logging
from typing import Any

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, model_validator

logger = logging.getLogger(__name__)


def experience_is_valid(exp: Any) -> bool:
    """Structure + finiteness check of one ``(state, policy_target, n_step_return)`` tuple."""
    if not (isinstance(exp, tuple) and len(exp) == 3):
        return False
    state, policy, value = exp
    if not (isinstance(state, dict) and "grid" in state and "other_features" in state):
        return False
    grid, other = state["grid"], state["other_features"]
    if not (isinstance(grid, np.ndarray) and isinstance(other, np.ndarray)):
        return False
    if not (np.all(np.isfinite(grid)) and np.all(np.isfinite(other))):
        return False
    if not hasattr(policy, "items"):
        return False
    try:
        return bool(np.isfinite(float(value)))
    except (TypeError, ValueError):
        return False


class SelfPlayResult(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)

    episode_experiences: list[Any]
    final_score: float
    episode_steps: int
    trainer_step_at_episode_start: int
    total_simulations: int = Field(..., ge=0)
    avg_root_visits: float = Field(..., ge=0)
    avg_tree_depth: float = Field(..., ge=0)
    context: dict[str, Any] = Field(default_factory=dict)

    @model_validator(mode="after")
    def _drop_malformed(self) -> "SelfPlayResult":
        kept = [e for e in self.episode_experiences if experience_is_valid(e)]
        dropped = len(self.episode_experiences) - len(kept)
        if dropped:
            logger.warning("SelfPlayResult validation: dropped %d malformed experiences.", dropped)
            object.__setattr__(self, "episode_experiences", kept)
        return self


SelfPlayResult.model_rebuild(force=True)
