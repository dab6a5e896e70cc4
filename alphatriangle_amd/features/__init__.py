"""Feature extraction (parity: alphatriangle/features/__init__.py:9-13)."""

from . import grid_features
from .extractor import GameStateFeatures, extract_features_batch, extract_state_features

__all__ = ["GameStateFeatures", "extract_features_batch", "extract_state_features", "grid_features"]
