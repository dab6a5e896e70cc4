"""Training orchestration (parity: alphatriangle/training/__init__.py:17-29)."""

from .components import TrainingComponents
from .logging_utils import log_configs_to_mlflow
from .loop import TrainingLoop
from .loop_helpers import LoopHelpers
from .runner import run_training
from .setup import count_parameters, setup_training_components
from .worker_manager import WorkerManager

__all__ = [
    "LoopHelpers",
    "TrainingComponents",
    "TrainingLoop",
    "WorkerManager",
    "count_parameters",
    "log_configs_to_mlflow",
    "run_training",
    "setup_training_components",
]
