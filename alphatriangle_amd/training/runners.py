"""Re-export of the training entry point (parity: alphatriangle/training/runners.py:8-12)."""

from .runner import run_training

__all__ = ["run_training"]
