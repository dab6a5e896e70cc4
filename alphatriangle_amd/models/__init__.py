"""Network definitions and the inference wrapper (parity: alphatriangle/nn/__init__.py)."""

from .model import AlphaTriangleNet, PositionalEncoding, ResidualBlock, conv_block
from .network import NetworkEvaluationError, NeuralNetwork

__all__ = [
    "AlphaTriangleNet",
    "NetworkEvaluationError",
    "NeuralNetwork",
    "PositionalEncoding",
    "ResidualBlock",
    "conv_block",
]
