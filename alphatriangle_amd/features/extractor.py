"""State -> network-input features.

``extract_state_features`` runs the native extractor (csrc/core/atri_core.h:
``grid_value`` / ``other_features_serial``), which is the same code the HIP engine
runs per leaf.  ``GameStateFeatures`` is the readable NumPy implementation of the
same formulas (alphatriangle/features/extractor.py:33-147) and is what the tests
compare the native extractor against.
"""

from __future__ import annotations

import logging
from typing import cast

import numpy as np

from .. import native
from ..config.model_config import ModelConfig
from ..env.game_state import GameState
from ..utils.types import StateType
from . import grid_features

logger = logging.getLogger(__name__)

FEATURES_PER_SHAPE = 7
EXPLICIT_FEATURES_DIM = 6


class GameStateFeatures:
    """NumPy reference implementation of the feature formulas."""

    def __init__(self, game_state: GameState, model_config: ModelConfig):
        self.gs = game_state
        self.env_config = game_state.env_config
        self.model_config = model_config
        data = game_state.get_grid_data_np()
        self.occupied_np = data["occupied"]
        self.death_np = data["death"]
        self.color_id_np = data["color_id"]

    def _get_grid_state(self) -> np.ndarray:
        rows, cols = self.env_config.ROWS, self.env_config.COLS
        grid = np.zeros((self.model_config.GRID_INPUT_CHANNELS, rows, cols), dtype=np.float32)
        grid[0, self.occupied_np & ~self.death_np] = 1.0
        grid[0, self.death_np] = -1.0
        return grid

    def _get_shape_features(self) -> np.ndarray:
        cfg = self.env_config
        mat = np.zeros((cfg.NUM_SHAPE_SLOTS, FEATURES_PER_SHAPE), dtype=np.float32)
        for i, shape in enumerate(self.gs.get_shapes()):
            if not shape or not shape.triangles:
                continue
            n = len(shape.triangles)
            ups = sum(1 for _, _, up in shape.triangles if up)
            min_r, min_c, max_r, max_c = shape.bbox()
            height = max_r - min_r + 1
            width_eff = (max_c - min_c + 1) * 0.75 + 0.25
            mat[i, 0] = np.clip(n / 5.0, 0, 1)
            mat[i, 1] = ups / n
            mat[i, 2] = (n - ups) / n
            mat[i, 3] = np.clip(height / cfg.ROWS, 0, 1)
            mat[i, 4] = np.clip(width_eff / cfg.COLS, 0, 1)
            mat[i, 5] = np.clip(((min_r + max_r) / 2.0) / cfg.ROWS, 0, 1)
            mat[i, 6] = np.clip(((min_c + max_c) / 2.0) / cfg.COLS, 0, 1)
        return mat.flatten()

    def _get_shape_availability(self) -> np.ndarray:
        return np.array([1.0 if s else 0.0 for s in self.gs.get_shapes()], dtype=np.float32)

    def _get_explicit_features(self) -> np.ndarray:
        feats = np.zeros(EXPLICIT_FEATURES_DIM, dtype=np.float32)
        rows, cols = self.env_config.ROWS, self.env_config.COLS
        heights = grid_features.get_column_heights(self.occupied_np, self.death_np, rows, cols)
        holes = grid_features.count_holes(self.occupied_np, self.death_np, heights, rows, cols)
        bump = grid_features.get_bumpiness(heights)
        playable = int(np.sum(~self.death_np))
        feats[0] = np.clip(self.gs.game_score() / 100.0, -5.0, 5.0)
        feats[1] = np.mean(heights) / rows if rows > 0 else 0
        feats[2] = np.max(heights) / rows if rows > 0 else 0
        feats[3] = holes / playable if playable > 0 else 0
        feats[4] = (bump / (cols - 1)) / rows if cols > 1 and rows > 0 else 0
        feats[5] = np.clip(self.gs.current_step / 1000.0, 0, 1)
        return cast("np.ndarray", np.nan_to_num(feats, nan=0.0, posinf=0.0, neginf=0.0))

    def get_combined_other_features(self) -> np.ndarray:
        combined = np.concatenate(
            [self._get_shape_features(), self._get_shape_availability(), self._get_explicit_features()]
        )
        expected = self.model_config.OTHER_NN_INPUT_FEATURES_DIM
        if combined.shape[0] != expected:
            logger.error(
                "other_features dimension mismatch: extracted %d, ModelConfig expects %d. Padding/truncating.",
                combined.shape[0],
                expected,
            )
            if combined.shape[0] < expected:
                combined = np.concatenate([combined, np.zeros(expected - combined.shape[0], dtype=combined.dtype)])
            else:
                combined = combined[:expected]
        if not np.all(np.isfinite(combined)):
            logger.error("Non-finite values in other_features; replacing with 0.")
            combined = np.nan_to_num(combined, nan=0.0, posinf=0.0, neginf=0.0)
        return cast("np.ndarray", combined.astype(np.float32))


_warned_dims: set[tuple[int, int]] = set()


def _warn_dim_mismatch(native_dim: int, expected: int) -> None:
    if native_dim != expected and (native_dim, expected) not in _warned_dims:
        _warned_dims.add((native_dim, expected))
        logger.error(
            "other_features dimension mismatch: engine yields %d, ModelConfig expects %d. Padding/truncating.",
            native_dim,
            expected,
        )


def extract_state_features(game_state: GameState, model_config: ModelConfig) -> StateType:
    """Return ``{"grid": (C,H,W) f32, "other_features": (D,) f32}`` for one state."""
    cfg = game_state.env_config
    _warn_dim_mismatch(8 * cfg.NUM_SHAPE_SLOTS + EXPLICIT_FEATURES_DIM, model_config.OTHER_NN_INPUT_FEATURES_DIM)
    grid, other = native.host_of(game_state.core).extract_features(
        game_state.core, model_config.GRID_INPUT_CHANNELS, model_config.OTHER_NN_INPUT_FEATURES_DIM
    )
    return {"grid": grid, "other_features": other}


def extract_features_batch(
    states: list[GameState], model_config: ModelConfig, rows: int, cols: int
) -> tuple[np.ndarray, np.ndarray]:
    """Batched extraction: ``(grid [n,C,H,W], other [n,D])``."""
    cores = [s.core if isinstance(s, GameState) else s for s in states]
    return (native.host_of(cores[0]) if cores else native.host(rows * cols)).extract_features_batch(
        cores,
        model_config.GRID_INPUT_CHANNELS,
        model_config.OTHER_NN_INPUT_FEATURES_DIM,
        rows,
        cols,
    )
