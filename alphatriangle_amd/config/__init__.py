"""Configuration models (parity: alphatriangle/config/__init__.py:10-17, plus SystemConfig)."""

from .app_config import APP_NAME
from .env_config import EnvConfig
from .mcts_config import AlphaTriangleMCTSConfig, SearchConfiguration
from .model_config import ModelConfig
from .system_config import SystemConfig
from .train_config import TrainConfig
from .validation import print_config_info_and_validate

__all__ = [
    "APP_NAME",
    "AlphaTriangleMCTSConfig",
    "EnvConfig",
    "ModelConfig",
    "SearchConfiguration",
    "SystemConfig",
    "TrainConfig",
    "print_config_info_and_validate",
]
