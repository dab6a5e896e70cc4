"""Alias of :mod:`alphatriangle_amd.models` under the reference's package name (alphatriangle/nn)."""

from ..models import (
    AlphaTriangleNet,
    NetworkEvaluationError,
    NeuralNetwork,
    PositionalEncoding,
    ResidualBlock,
    conv_block,
)
from ..models import model, network

__all__ = [
    "AlphaTriangleNet",
    "NetworkEvaluationError",
    "NeuralNetwork",
    "PositionalEncoding",
    "ResidualBlock",
    "conv_block",
    "model",
    "network",
]
