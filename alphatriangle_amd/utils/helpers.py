"""General helpers (parity: alphatriangle/utils/helpers.py:12-108)."""

from __future__ import annotations

import logging
import random

import numpy as np
import torch

logger = logging.getLogger(__name__)


def get_device(device_preference: str = "auto") -> torch.device:
    """Pick the torch device.  An MI355X shows up as ``cuda`` under PyTorch-ROCm."""
    pref = (device_preference or "auto").lower()
    if pref == "cpu":
        return torch.device("cpu")
    has_cuda = torch.cuda.is_available()
    mps_backend = getattr(torch.backends, "mps", None)
    has_mps = bool(mps_backend is not None and mps_backend.is_available() and mps_backend.is_built())
    if pref == "cuda":
        if has_cuda:
            return torch.device("cuda")
        logger.warning("Device 'cuda' requested but no GPU is visible; falling back.")
    if pref == "mps":
        if has_mps:
            return torch.device("mps")
        logger.warning("Device 'mps' requested but not available; falling back.")
    if has_cuda:
        return torch.device("cuda")
    if has_mps:
        return torch.device("mps")
    return torch.device("cpu")


def set_random_seeds(seed: int = 42) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    mps_backend = getattr(torch.backends, "mps", None)
    if mps_backend is not None and mps_backend.is_available() and hasattr(torch, "mps"):
        try:
            torch.mps.manual_seed(seed)
        except Exception:  # pragma: no cover - platform specific
            logger.debug("could not seed MPS", exc_info=True)
    logger.info("Set random seeds to %d", seed)


def format_eta(seconds: float | None) -> str:
    """Human readable time estimate: 'N/A', '>1 month', '2h 03m 04s', '3m 04s' or '4s'."""
    if seconds is None or not np.isfinite(seconds) or seconds < 0:
        return "N/A"
    if seconds > 3600 * 24 * 30:
        return ">1 month"
    total = int(seconds)
    hours, rem = divmod(total, 3600)
    minutes, secs = divmod(rem, 60)
    if hours > 0:
        return f"{hours}h {minutes:02d}m {secs:02d}s"
    if minutes > 0:
        return f"{minutes}m {secs:02d}s"
    return f"{secs}s"


def normalize_color_for_matplotlib(color_tuple_0_255: tuple[int, int, int]) -> tuple[float, float, float]:
    """RGB 0-255 -> 0-1; anything malformed maps to black."""
    if isinstance(color_tuple_0_255, tuple) and len(color_tuple_0_255) == 3:
        r, g, b = (max(0.0, min(255.0, float(c))) / 255.0 for c in color_tuple_0_255)
        return (r, g, b)
    logger.warning("Invalid color %r, returning black", color_tuple_0_255)
    return (0.0, 0.0, 0.0)
