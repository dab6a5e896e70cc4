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
    COMPACT_LEAF_BATCH: bool = Field(
        default=True,
        description="Evaluate only the leaf slots that need the network: their feature rows are written densely (slot scan) and "
        "the engine kernels read the row count on the device (no host sync).  Saves network time once games reach "
        "terminal states (about 10 % of the slots in steady state).",
    )
    TREE_REUSE: bool = Field(default=True)
    INFERENCE_DTYPE: Literal["fp32", "bf16"] = Field(
        default="fp32",
        description="Precision of the leaf evaluations that drive the search. fp32: fp32 activations, every GEMM as three "
        "bf16 MFMA passes over (hi, lo) operand splits with fp32 accumulation (the reference evaluates in fp32); "
        "bf16: bf16 activations and single-pass MFMA (about twice as fast).",
    )
    LEARNER_AUTOCAST: bool | None = Field(
        default=None,
        description="bf16 autocast for the learner forward/backward; None follows INFERENCE_DTYPE (off for fp32).",
    )
    LEARNER_FP32_GEMM: Literal["bf16x3", "library"] = Field(
        default="bf16x3",
        description="fp32 learner steps: 'bf16x3' runs the GEMM-lowered layers on the bf16 matrix cores (three passes over "
        "(hi, lo) operand splits, ~5e-6 relative error per product - the scheme of the fp32-grade inference engine); "
        "'library' keeps hipBLASLt's native-fp32 GEMMs (~7e-7, about 20 % slower per step).",
    )
    LEARNER_PATCH_FREE_CONV: bool = Field(
        default=False,
        description="fp32 'bf16x3' learner steps: generate the 3x3 patches inside the GEMM instead of storing a patch matrix per "
        "layer (9x less activation memory for the backward pass, slower products).",
    )
    REPLAY_HBM_BUDGET_GB: float = Field(default=64.0, gt=0)
    TRAIN_STEPS_PER_MOVE: int = Field(default=1, ge=0, description="Learner steps interleaved per self-play move.")
    TRAIN_STEPS_PER_ROUND: int = Field(
        default=8, ge=1,
        description="Split layout: learner steps between two polls of the control store (shipments, weight requests).",
    )
    SHIP_CAPACITY: int = Field(default=1 << 15, ge=1, description="Split layout: upper bound of experience records in one shipment.")
    MAX_SAMPLE_REUSE: float | None = Field(
        default=None, gt=0,
        description="Split layout: the learner consumes at most this many samples per experience received (replay ratio); None = no limit.",
    )
    ACTOR_SILENCE_S: float = Field(default=300.0, gt=0, description="Split layout: an actor rank that ships nothing for this long is retired.")
    COLLECTIVE_TIMEOUT_S: float = Field(default=120.0, gt=0, description="Time-out of every data-plane collective (RCCL / gloo).")
    CONTROL_TIMEOUT_S: float = Field(default=60.0, gt=0, description="Time-out of the host-side control plane (agreements, drains).")
    TRANSFER_TIMEOUT_S: float = Field(default=60.0, gt=0, description="Split layout: deadline of one posted point-to-point transfer; a pair that misses it loses its communicator, the process carries on.")
    HEARTBEAT_PERIOD_S: float = Field(default=1.0, gt=0, description="Split layout: how often an actor rank renews its heartbeat key in the control store.")
    HEARTBEAT_STALE_S: float = Field(default=45.0, gt=0, description="Split layout: a learner posts no receive for an actor whose heartbeat is older than this, and retires it.")
    OVERLAP_LEARNER: bool = Field(
        default=True, description="Colocated GPU runs: learner steps on a second HIP stream, concurrent with the search of the move."
    )
    USE_HIP_GRAPHS: bool = Field(default=True)
    DIST_BACKEND: Literal["auto", "nccl", "gloo"] = Field(default="auto", description="nccl = RCCL on ROCm.")
    TEMPERATURE_SWITCH_GAME_STEP: int | None = Field(
        default=None,
        description="Game step after which action selection uses temperature 0.1; None keeps the reference rule "
        "(10% of MAX_TRAINING_STEPS).",
    )

    def learner_autocast(self) -> bool:
        return self.INFERENCE_DTYPE == "bf16" if self.LEARNER_AUTOCAST is None else bool(self.LEARNER_AUTOCAST)

    def apply_learner_gemm_mode(self) -> None:
        """Select the fp32 GEMM path of the learner's GEMM-lowered layers for this process (models/model.py)."""
        from ..models import model as model_mod

        model_mod.FP32_GEMM_ON_BF16_CORES = self.LEARNER_FP32_GEMM == "bf16x3"
        model_mod.PATCH_FREE_CONV = bool(self.LEARNER_PATCH_FREE_CONV)

    @model_validator(mode="after")
    def _check(self) -> "SystemConfig":
        if self.REPLAY_HBM_BUDGET_GB > 256:
            raise ValueError("REPLAY_HBM_BUDGET_GB must leave room for trees and activations (<= 256).")
        return self


SystemConfig.model_rebuild(force=True)
