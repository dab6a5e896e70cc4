"""Column statistics of the grid.

NumPy-vectorised twins of the kernels in csrc/core/atri_core.h (``column_stats``);
same definitions as alphatriangle/features/grid_features.py:7-42 (the reference
JIT-compiles them with numba; here the hot path is native code).
"""

from __future__ import annotations

import numpy as np


def get_column_heights(occupied: np.ndarray, death: np.ndarray, rows: int, cols: int) -> np.ndarray:
    """Per column: (row index of the last occupied non-death cell) + 1, or 0."""
    live = np.asarray(occupied, dtype=bool)[:rows, :cols] & ~np.asarray(death, dtype=bool)[:rows, :cols]
    row_idx = np.arange(1, rows + 1, dtype=np.int32)[:, None]
    return (live * row_idx).max(axis=0).astype(np.int32) if rows > 0 else np.zeros(cols, dtype=np.int32)


def count_holes(occupied: np.ndarray, death: np.ndarray, heights: np.ndarray, _rows: int, cols: int) -> int:
    """Empty non-death cells whose row index is below their column's height."""
    occ = np.asarray(occupied, dtype=bool)[:, :cols]
    dead = np.asarray(death, dtype=bool)[:, :cols]
    rows = occ.shape[0]
    below = np.arange(rows, dtype=np.int32)[:, None] < np.asarray(heights, dtype=np.int32)[None, :]
    return int(np.count_nonzero(below & ~occ & ~dead))


def get_bumpiness(heights: np.ndarray) -> float:
    """Sum of absolute height differences between adjacent columns."""
    h = np.asarray(heights, dtype=np.float64)
    return float(np.abs(np.diff(h)).sum()) if h.size > 1 else 0.0
