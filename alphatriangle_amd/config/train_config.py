"""Training-process configuration.

Fields, defaults and validators follow alphatriangle/config/train_config.py:18-227.
``COMPILE_MODEL`` keeps its name for drop-in compatibility; on this framework it
asks for HIP-graph capture of the learner's forward/backward instead of a tracing
compiler (about 490 launches per step become one replay: 6.7 -> 5.0 ms per step at
batch 256).  A graph is only replayed when it holds no memset nodes and reproduces
eager gradients; otherwise the step stays eager.
"""

from __future__ import annotations

import logging
import time
from typing import Literal

from pydantic import BaseModel, Field, field_validator, model_validator

logger = logging.getLogger(__name__)

_FALLBACK_STEPS = 100_000


class TrainConfig(BaseModel):
    RUN_NAME: str = Field(default_factory=lambda: f"train_{time.strftime('%Y%m%d_%H%M%S')}")
    LOAD_CHECKPOINT_PATH: str | None = Field(default=None)
    LOAD_BUFFER_PATH: str | None = Field(default=None)
    AUTO_RESUME_LATEST: bool = Field(default=True)
    DEVICE: Literal["auto", "cuda", "cpu", "mps"] = Field(default="auto")
    RANDOM_SEED: int = Field(default=42)

    MAX_TRAINING_STEPS: int | None = Field(default=100_000, ge=1)

    NUM_SELF_PLAY_WORKERS: int = Field(
        default=8,
        ge=1,
        description="Requested CPU self-play workers; clamped to the detected core count minus two.",
    )
    WORKER_DEVICE: Literal["auto", "cuda", "cpu", "mps"] = Field(default="cpu")
    BATCH_SIZE: int = Field(default=256, ge=1)
    BUFFER_CAPACITY: int = Field(default=250_000, ge=1)
    MIN_BUFFER_SIZE_TO_TRAIN: int = Field(default=25_000, ge=1)
    WORKER_UPDATE_FREQ_STEPS: int = Field(default=10, ge=1)

    N_STEP_RETURNS: int = Field(default=5, ge=1)
    GAMMA: float = Field(default=0.99, gt=0, le=1.0)

    OPTIMIZER_TYPE: Literal["Adam", "AdamW", "SGD"] = Field(default="AdamW")
    LEARNING_RATE: float = Field(default=2e-4, gt=0)
    WEIGHT_DECAY: float = Field(default=1e-4, ge=0)
    GRADIENT_CLIP_VALUE: float | None = Field(default=1.0)

    LR_SCHEDULER_TYPE: Literal["StepLR", "CosineAnnealingLR"] | None = Field(default="CosineAnnealingLR")
    LR_SCHEDULER_T_MAX: int | None = Field(default=None)
    LR_SCHEDULER_ETA_MIN: float = Field(default=1e-6, ge=0)
    LR_SCHEDULER_STEP_SIZE: int = Field(default=100_000, ge=1)
    LR_SCHEDULER_GAMMA: float = Field(default=0.1, gt=0)

    POLICY_LOSS_WEIGHT: float = Field(default=1.0, ge=0)
    VALUE_LOSS_WEIGHT: float = Field(default=1.0, ge=0)
    ENTROPY_BONUS_WEIGHT: float = Field(default=0.001, ge=0)

    CHECKPOINT_SAVE_FREQ_STEPS: int = Field(default=2500, ge=1)

    USE_PER: bool = Field(default=True)
    PER_ALPHA: float = Field(default=0.6, ge=0)
    PER_BETA_INITIAL: float = Field(default=0.4, ge=0, le=1.0)
    PER_BETA_FINAL: float = Field(default=1.0, ge=0, le=1.0)
    PER_BETA_ANNEAL_STEPS: int | None = Field(default=None)
    PER_EPSILON: float = Field(default=1e-5, gt=0)

    COMPILE_MODEL: bool = Field(
        default=True,
        description="Capture the learner forward/backward into a HIP graph on a GPU (guarded; falls back to eager).",
    )
    PROFILE_WORKERS: bool = Field(default=False, description="cProfile worker 0 and write .prof files.")

    @model_validator(mode="after")
    def _buffer_sizes(self) -> "TrainConfig":
        if self.MIN_BUFFER_SIZE_TO_TRAIN > self.BUFFER_CAPACITY:
            raise ValueError("MIN_BUFFER_SIZE_TO_TRAIN cannot be greater than BUFFER_CAPACITY.")
        if self.BATCH_SIZE > self.BUFFER_CAPACITY:
            raise ValueError("BATCH_SIZE cannot be greater than BUFFER_CAPACITY.")
        return self

    @model_validator(mode="after")
    def _derive_t_max(self) -> "TrainConfig":
        if self.LR_SCHEDULER_TYPE == "CosineAnnealingLR" and self.LR_SCHEDULER_T_MAX is None:
            steps = self.MAX_TRAINING_STEPS
            derived = steps if (steps is not None and steps >= 1) else _FALLBACK_STEPS
            object.__setattr__(self, "LR_SCHEDULER_T_MAX", derived)
        if self.LR_SCHEDULER_T_MAX is not None and self.LR_SCHEDULER_T_MAX <= 0:
            raise ValueError("LR_SCHEDULER_T_MAX must be positive if set.")
        return self

    @model_validator(mode="after")
    def _derive_beta_anneal(self) -> "TrainConfig":
        if self.USE_PER and self.PER_BETA_ANNEAL_STEPS is None:
            steps = self.MAX_TRAINING_STEPS
            derived = steps if (steps is not None and steps >= 1) else _FALLBACK_STEPS
            object.__setattr__(self, "PER_BETA_ANNEAL_STEPS", derived)
        if self.PER_BETA_ANNEAL_STEPS is not None and self.PER_BETA_ANNEAL_STEPS <= 0:
            raise ValueError("PER_BETA_ANNEAL_STEPS must be positive if set.")
        return self

    @field_validator("GRADIENT_CLIP_VALUE")
    @classmethod
    def _clip_positive(cls, v: float | None) -> float | None:
        if v is not None and v <= 0:
            raise ValueError("GRADIENT_CLIP_VALUE must be positive if set.")
        return v

    @model_validator(mode="after")
    def _beta_order(self) -> "TrainConfig":
        if self.PER_BETA_FINAL < self.PER_BETA_INITIAL:
            raise ValueError("PER_BETA_FINAL cannot be less than PER_BETA_INITIAL")
        return self


TrainConfig.model_rebuild(force=True)
