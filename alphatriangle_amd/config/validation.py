"""Instantiate every config to trigger validation and print a summary.

Behavioural parity: alphatriangle/config/validation.py:20-100.
"""

from __future__ import annotations

import logging
from typing import Any

from pydantic import BaseModel, ValidationError

from .env_config import EnvConfig
from .mcts_config import AlphaTriangleMCTSConfig
from .model_config import ModelConfig
from .system_config import SystemConfig
from .train_config import TrainConfig

logger = logging.getLogger(__name__)


def _dump(title: str, cfg: BaseModel) -> None:
    print(f"--- {title} ---")
    for key, value in cfg.model_dump().items():
        print(f"  {key}: {value}")


def print_config_info_and_validate(
    mcts_config_instance: AlphaTriangleMCTSConfig | None = None,
    system_config_instance: SystemConfig | None = None,
) -> dict[str, Any]:
    """Validate all configs, print them, and return the instances."""
    print("-" * 40)
    print("Configuration validation and summary")
    print("-" * 40)
    try:
        env = EnvConfig()
        model = ModelConfig()
        train = TrainConfig()
        mcts = mcts_config_instance if mcts_config_instance is not None else AlphaTriangleMCTSConfig()
        if not isinstance(mcts, AlphaTriangleMCTSConfig):
            raise TypeError(f"expected AlphaTriangleMCTSConfig, got {type(mcts).__name__}")
        AlphaTriangleMCTSConfig.model_validate(mcts.model_dump())
        system = system_config_instance if system_config_instance is not None else SystemConfig()
    except (ValidationError, TypeError) as exc:
        logger.critical("Configuration validation failed: %s", exc)
        raise ValueError("Invalid configuration detected.") from exc
    _dump("Environment", env)
    _dump("Model", model)
    _dump("Training", train)
    _dump("MCTS", mcts)
    _dump("System", system)
    print("-" * 40)
    return {"env": env, "model": model, "train": train, "mcts": mcts, "system": system}
