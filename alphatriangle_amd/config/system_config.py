"""Device topology and GPU-engine sizing (no counterpart in the reference, which
runs CPU Ray workers only: alphatriangle/training/worker_manager.py:50-66).

One process per GPU; ranks are split into self-play actors and learners, or run
both roles colocated.  Sizes default to what fits comfortably in 288 GB of HBM.
"""

from __future__ import annotations

from typing import Literal

from pydantic import BaseModel, Field, model_validator


class SystemConfig(BaseModel):
    ENGINE: Literal["auto", "gpu", "cpu"] = Field(
        default="auto", description="gpu = HBM-resident batched tree engine, cpu = host workers."
    )
    ROLE_LAYOUT: Literal["colocated", "split"] = Field(
        default="colocated",
        description="colocated: every rank self-plays and trains (DP all-reduce). split: actor ranks + learner ranks.",
    )
    NUM_LEARNER_RANKS: int = Field(default=1, ge=1, description="Learner ranks when ROLE_LAYOUT == 'split'.")
    GAMES_PER_GPU: int = Field(default=4096, ge=1, description="Concurrent self-play games per actor GPU.")
    LEAVES_PER_TREE: int = Field(default=1, ge=1, le=32, description="Leaves per tree per wave (virtual loss if >1).")
    TREE_REUSE: bool = Field(default=True)
    INFERENCE_DTYPE: Literal["bf16"] = Field(default="bf16")
    LEARNER_AUTOCAST: bool = Field(default=True, description="bf16 autocast for the learner forward/backward.")
    REPLAY_ON_DEVICE: bool = Field(default=True, description="Keep the replay buffer in HBM on GPU runs.")
    REPLAY_HBM_BUDGET_GB: float = Field(default=64.0, gt=0)
    TRAIN_STEPS_PER_MOVE: int = Field(default=1, ge=0, description="Learner steps interleaved per self-play move.")
    TRAIN_STEPS_PER_ROUND: int = Field(
        default=32, ge=0,
        description="Split layout: learner steps per lock-step round (one round = one self-play move on every actor rank).",
    )
    SHIP_CAPACITY: int = Field(default=1 << 15, ge=1, description="Split layout: experience records per actor per round (fixed block).")
    OVERLAP_LEARNER: bool = Field(
        default=True, description="Colocated GPU runs: learner steps on a second HIP stream, concurrent with the search of the move."
    )
    USE_HIP_GRAPHS: bool = Field(default=True)
    WEIGHT_SYNC_ASYNC: bool = Field(default=True)
    DIST_BACKEND: Literal["auto", "nccl", "gloo"] = Field(default="auto", description="nccl = RCCL on ROCm.")
    TEMPERATURE_SWITCH_GAME_STEP: int | None = Field(
        default=None,
        description="Game step after which action selection uses temperature 0.1; None keeps the reference rule "
        "(10% of MAX_TRAINING_STEPS).",
    )

    @model_validator(mode="after")
    def _check(self) -> "SystemConfig":
        if self.REPLAY_HBM_BUDGET_GB > 256:
            raise ValueError("REPLAY_HBM_BUDGET_GB must leave room for trees and activations (<= 256).")
        return self


SystemConfig.model_rebuild(force=True)
