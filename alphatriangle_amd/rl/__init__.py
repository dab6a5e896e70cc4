"""RL components (parity: alphatriangle/rl/__init__.py:11-18)."""

from .core.buffer import ExperienceBuffer
from .core.trainer import TensorBatch, Trainer
from .self_play.worker import SelfPlayWorker
from .types import SelfPlayResult

__all__ = ["ExperienceBuffer", "SelfPlayResult", "SelfPlayWorker", "TensorBatch", "Trainer"]
