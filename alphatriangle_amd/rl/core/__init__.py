from .buffer import ExperienceBuffer
from .trainer import TensorBatch, Trainer

__all__ = ["ExperienceBuffer", "TensorBatch", "Trainer"]
