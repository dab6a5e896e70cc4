"""Progress / ETA logging and experience validation (parity: alphatriangle/training/loop_helpers.py:19-146)."""

from __future__ import annotations

import logging
import time
from collections.abc import Callable
from typing import Any

import numpy as np

from ..utils.helpers import format_eta
from ..utils.types import Experience
from .components import TrainingComponents

logger = logging.getLogger(__name__)


class LoopHelpers:
    def __init__(self, components: TrainingComponents, get_loop_state_func: Callable[[], dict[str, Any]]):
        self.components = components
        self.get_loop_state = get_loop_state_func
        self.train_config = components.train_config
        self.trainer = components.trainer
        self.last_rate_calc_time = time.time()
        self.last_rate_calc_step = 0
        self.last_progress_log_step = -1

    def reset_rate_counters(self, global_step: int, _episodes_played: int = 0, _total_simulations: int = 0) -> None:
        self.last_rate_calc_time = time.time()
        self.last_rate_calc_step = global_step

    def log_progress_eta(self, every_steps: int = 100) -> None:
        """Progress line every ``every_steps`` learner steps: steps, episodes, sims, buffer fill, speed, ETA."""
        state = self.get_loop_state()
        step = state["global_step"]
        if step == 0 or step % every_steps != 0 or step == self.last_progress_log_step:
            return
        self.last_progress_log_step = step
        elapsed = time.time() - state["start_time"]
        steps_since = step - state.get("initial_global_step", 0)
        speed = steps_since / elapsed if elapsed > 1 and steps_since > 0 else 0.0
        target = self.train_config.MAX_TRAINING_STEPS
        eta = format_eta((target - step) / speed) if target and speed > 0 else "N/A"
        progress = f"Step: {step}/{target} ({step / target:.1%})" if target else f"Step: {step}"
        sims = state["total_simulations_run"]
        sims_str = f"{sims / 1e6:.2f}M" if sims >= 1e6 else (f"{sims / 1e3:.1f}k" if sims >= 1000 else str(sims))
        cap = state["buffer_capacity"]
        fill = (state["buffer_size"] / cap * 100) if cap > 0 else 0.0
        logger.info(
            "Progress: %s, Episodes: %s, Total Sims: %s, Buffer: %s/%s (%.1f%%), Pending Tasks: %s, Avg Speed: %.2f steps/sec, ETA: %s",
            progress,
            f"{state['episodes_played']:,}",
            sims_str,
            f"{state['buffer_size']:,}",
            f"{cap:,}",
            fill,
            state["num_pending_tasks"],
            speed,
            eta,
        )

    def validate_experiences(self, experiences: list[Experience]) -> tuple[list[Experience], int]:
        """Filter out experiences with a wrong structure or non-finite numbers."""
        valid: list[Experience] = []
        bad = 0
        for i, exp in enumerate(experiences):
            ok = False
            try:
                if isinstance(exp, tuple) and len(exp) == 3:
                    state, policy, value = exp
                    if (
                        isinstance(state, dict)
                        and isinstance(state.get("grid"), np.ndarray)
                        and isinstance(state.get("other_features"), np.ndarray)
                        and isinstance(policy, dict)
                        and isinstance(value, (float, int))
                        and np.all(np.isfinite(state["grid"]))
                        and np.all(np.isfinite(state["other_features"]))
                        and np.isfinite(value)
                    ):
                        ok = True
            except Exception as exc:  # pragma: no cover
                logger.error("Unexpected error validating experience %d: %s", i, exc)
            if ok:
                valid.append(exp)
            else:
                bad += 1
        if bad:
            logger.warning("Filtered out %d invalid experiences.", bad)
        return valid, bad
