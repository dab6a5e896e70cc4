"""Log every configuration as run parameters (parity: alphatriangle/training/logging_utils.py:13-35)."""

from __future__ import annotations

import logging

from .components import TrainingComponents

logger = logging.getLogger(__name__)


def log_configs_to_mlflow(components: TrainingComponents) -> None:
    """Writes the config dumps to the MLflow file store and to ``configs.json`` of the run."""
    try:
        params: dict[str, object] = {}
        params.update({f"train_{k}": v for k, v in components.train_config.model_dump().items()})
        params.update({f"model_{k}": v for k, v in components.model_config.model_dump().items()})
        params.update({f"env_{k}": v for k, v in components.env_config.model_dump().items()})
        params.update({f"mcts_{k}": v for k, v in components.mcts_config.model_dump().items()})
        params.update({f"system_{k}": v for k, v in components.system_config.model_dump().items()})
        components.trieye_actor.log_params(params)
        components.trieye_actor.save_run_config(
            {
                "train": components.train_config.model_dump(),
                "model": components.model_config.model_dump(),
                "env": components.env_config.model_dump(),
                "mcts": components.mcts_config.model_dump(),
                "system": components.system_config.model_dump(),
            }
        )
        logger.info("Logged configs as run parameters.")
    except Exception as exc:
        logger.error("Failed to log configs: %s", exc, exc_info=True)
