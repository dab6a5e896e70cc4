"""Visit counts -> action / policy target.

Behavioural parity: alphatriangle/rl/self_play/mcts_helpers.py:19-189.  The
sampling generator is injectable (seeded per worker) instead of a module-level
unseeded one.
"""

from __future__ import annotations

import logging
from collections.abc import Mapping

import numpy as np

from ...utils.types import ActionType, PolicyTargetMapping

logger = logging.getLogger(__name__)
_default_rng = np.random.default_rng()


class PolicyGenerationError(Exception):
    """Raised when no usable action / policy can be derived from the visit counts."""


def select_action_from_visits(
    visit_counts: Mapping[ActionType, int], temperature: float = 1.0, rng: np.random.Generator | None = None
) -> ActionType:
    """Sample an action: argmax (random tie-break) at T=0, else proportional to ``visits**(1/T)``."""
    gen = rng if rng is not None else _default_rng
    if not visit_counts:
        raise PolicyGenerationError("Cannot select action: no visit counts provided.")
    actions = list(visit_counts.keys())
    counts = np.array([visit_counts[a] for a in actions], dtype=np.float64)
    total = counts.sum()
    if total == 0:
        logger.warning("Total visit count is zero; selecting uniformly at random.")
        return int(actions[int(gen.integers(len(actions)))])
    if temperature == 0.0:
        best = np.flatnonzero(counts == counts.max())
        return int(actions[int(gen.choice(best))])
    powered = np.maximum(counts, 1e-9) ** (1.0 / temperature)
    s = powered.sum()
    if s < 1e-9 or not np.isfinite(s):
        raise PolicyGenerationError(f"Could not normalize visit probabilities (sum={s}).")
    probs = powered / s
    if not np.all(np.isfinite(probs)) or np.any(probs < 0):
        raise PolicyGenerationError("Invalid probabilities generated from visit counts.")
    probs = probs / probs.sum()
    try:
        return int(actions[int(gen.choice(len(actions), p=probs))])
    except ValueError as exc:
        raise PolicyGenerationError(f"Error during np.random.choice: {exc}") from exc


def get_policy_target_from_visits(
    visit_counts: Mapping[ActionType, int], action_dim: int, temperature: float = 1.0
) -> PolicyTargetMapping:
    """Dense ``{action: prob}`` over the whole action space from visit counts."""
    target: dict[ActionType, float] = dict.fromkeys(range(action_dim), 0.0)
    if not visit_counts:
        logger.warning("Cannot compute policy target: no visit counts. Returning zero target.")
        return target
    total = sum(visit_counts.values())
    if total == 0:
        logger.warning("Total visits is zero. Returning zero target.")
        return target
    if temperature == 0.0:
        top = max(visit_counts.values())
        best = [a for a, v in visit_counts.items() if v == top]
        share = 1.0 / len(best)
        for a in best:
            if 0 <= a < action_dim:
                target[a] = share
            else:
                logger.warning("Best action %s out of bounds (%d). Skipping.", a, action_dim)
        return target
    actions = list(visit_counts.keys())
    powered = np.maximum(np.array([visit_counts[a] for a in actions], dtype=np.float64), 1e-9) ** (1.0 / temperature)
    s = float(powered.sum())
    if s < 1e-9:
        raise PolicyGenerationError(f"Could not normalize policy target probabilities (sum={s}).")
    probs = powered / s
    if not np.all(np.isfinite(probs)) or np.any(probs < 0):
        raise PolicyGenerationError("Invalid probabilities generated for policy target.")
    probs = probs / probs.sum()
    for a, p in zip(actions, probs, strict=True):
        if 0 <= a < action_dim:
            target[a] = float(p)
        else:
            logger.warning("Policy target: action %s out of bounds (%d). Skipping.", a, action_dim)
    final = sum(target.values())
    if abs(final - 1.0) > 1e-5:
        logger.error("Policy target does not sum to 1.0 (sum=%f).", final)
    return target
