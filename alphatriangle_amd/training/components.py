"""Dependency container handed from setup to the loop (parity: alphatriangle/training/components.py:22-36)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

from ..config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
from ..models.network import NeuralNetwork
from ..rl.core.trainer import Trainer
from ..stats import Serializer, StatsActor, TrieyeConfig


@dataclass
class TrainingComponents:
    nn: NeuralNetwork
    buffer: Any  # ExperienceBuffer (host) or DeviceReplayBuffer (HBM)
    trainer: Trainer | None    # None on pure actor ranks of the split role layout
    trieye_actor: StatsActor
    trieye_config: TrieyeConfig
    serializer: Serializer
    train_config: TrainConfig
    env_config: EnvConfig
    model_config: ModelConfig
    mcts_config: AlphaTriangleMCTSConfig
    profile_workers: bool
    system_config: SystemConfig = field(default_factory=SystemConfig)
    gpu_actor: Any = None      # GpuSelfPlayActor when the HBM-resident engine is active
    dist: Any = None           # parallel.DistContext for multi-GPU runs
    experience_source: Any = None  # split layout, actor ranks: what produces this rank's experience records
