"""Python face of the native game engine.

Provides the ``GameState`` / ``Shape`` surface that the reference consumes from
the external ``trianglengin`` package (call sites:
alphatriangle/rl/self_play/worker.py:190-379, alphatriangle/features/extractor.py:22-31,58-66,
alphatriangle/nn/network.py:201,289, tests/nn/test_network.py:58,108).  Rules are
specified in docs/GAME_SPEC.md and implemented once in csrc/core/atri_core.h for
both the host and the HIP engine.
"""

from __future__ import annotations

import threading
from dataclasses import dataclass
from typing import Any

import numpy as np

from .. import native
from ..config.env_config import EnvConfig

_geom_cache: dict[str, Any] = {}
_geom_lock = threading.Lock()


def get_geometry(env_config: EnvConfig):
    """Native geometry tables for a config (cached: construction enumerates all placements)."""
    key = env_config.model_dump_json()
    with _geom_lock:
        geom = _geom_cache.get(key)
        if geom is None:
            geom = native.host(env_config.ROWS * env_config.COLS).Geometry(
                rows=env_config.ROWS,
                cols=env_config.COLS,
                playable_range=[tuple(r) for r in env_config.PLAYABLE_RANGE_PER_ROW],
                slots=env_config.NUM_SHAPE_SLOTS,
                min_line_length=env_config.MIN_LINE_LENGTH,
                max_shape_tris=env_config.MAX_SHAPE_TRIANGLES,
                r_place=env_config.REWARD_PER_PLACED_TRIANGLE,
                r_clear=env_config.REWARD_PER_CLEARED_TRIANGLE,
                r_alive=env_config.REWARD_PER_STEP_ALIVE,
                r_over=env_config.PENALTY_GAME_OVER,
                size_weights=list(env_config.SHAPE_SIZE_WEIGHTS),
            )
            _geom_cache[key] = geom
    return geom


_PALETTE: list[tuple[int, int, int]] = [
    (220, 40, 40),
    (60, 60, 220),
    (40, 200, 40),
    (230, 230, 40),
    (255, 165, 0),
    (128, 0, 128),
    (0, 200, 200),
    (200, 100, 180),
]


@dataclass(frozen=True)
class Shape:
    """A placeable piece: triangles are ``(dr, dc, is_up)`` relative to the anchor cell."""

    triangles: tuple[tuple[int, int, bool], ...]
    color: tuple[int, int, int]
    template_id: int = -1

    def bbox(self) -> tuple[int, int, int, int]:
        """(min_r, min_c, max_r, max_c) over the triangles; zeros for an empty shape."""
        if not self.triangles:
            return (0, 0, 0, 0)
        rs = [t[0] for t in self.triangles]
        cs = [t[1] for t in self.triangles]
        return (min(rs), min(cs), max(rs), max(cs))

    def copy(self) -> "Shape":
        return Shape(tuple(self.triangles), self.color, self.template_id)

    def __bool__(self) -> bool:
        return len(self.triangles) > 0


class GameState:
    """One game of the triangle puzzle (thin wrapper over the native state: 64 bytes up to 256 cells, the wide host
    build's 160 bytes above)."""

    def __init__(self, config: EnvConfig | None = None, initial_seed: int = 0, *, _core: Any = None):
        self.env_config: EnvConfig = config if config is not None else EnvConfig()
        self._geom = get_geometry(self.env_config)
        self._core = _core if _core is not None else native.host_of(self._geom).CoreState(self._geom, int(initial_seed) & (2**64 - 1))

    # -- construction helpers
    @classmethod
    def from_core(cls, config: EnvConfig, core: Any) -> "GameState":
        return cls(config, _core=core)

    @property
    def core(self) -> Any:
        return self._core

    @property
    def geometry(self) -> Any:
        return self._geom

    def copy(self) -> "GameState":
        return GameState(self.env_config, _core=self._core.copy())

    def reset(self, seed: int = 0) -> None:
        self._core = native.host_of(self._geom).CoreState(self._geom, int(seed) & (2**64 - 1))

    # -- rules
    def step(self, action: int) -> tuple[float, bool]:
        reward, done = self._core.step(int(action))
        return float(reward), bool(done)

    def valid_actions(self) -> list[int]:
        return self._core.valid_actions()

    def valid_mask(self) -> np.ndarray:
        return self._core.valid_mask()

    def is_over(self) -> bool:
        return bool(self._core.is_over())

    def get_game_over_reason(self) -> str | None:
        return "No valid actions available." if self.is_over() else None

    def game_score(self) -> float:
        return float(self._core.game_score())

    def get_last_cleared_triangles(self) -> int:
        return int(self._core.last_cleared)

    @property
    def current_step(self) -> int:
        return int(self._core.current_step)

    @property
    def triangles_cleared_total(self) -> int:
        return int(self._core.tris_cleared_total)

    # -- observation
    def get_grid_data_np(self) -> dict[str, np.ndarray]:
        return {"occupied": self._core.occupied(), "death": self._core.death(), "color_id": self._core.color_id()}

    def get_shapes(self) -> list[Shape | None]:
        templates = _templates_for(self._geom)
        out: list[Shape | None] = []
        for tid in self._core.shape_ids():
            if tid < 0:
                out.append(None)
            else:
                out.append(Shape(templates[tid], _PALETTE[tid % len(_PALETTE)], tid))
        return out

    # -- action encoding: action = slot * ROWS * COLS + r * COLS + c
    def encode_action(self, slot: int, r: int, c: int) -> int:
        cfg = self.env_config
        return int(slot * cfg.ROWS * cfg.COLS + r * cfg.COLS + c)

    def decode_action(self, action: int) -> tuple[int, int, int]:
        cfg = self.env_config
        cells = cfg.ROWS * cfg.COLS
        slot, pos = divmod(int(action), cells)
        r, c = divmod(pos, cfg.COLS)
        return slot, r, c

    def __repr__(self) -> str:
        return f"GameState(step={self.current_step}, score={self.game_score():.1f}, over={self.is_over()})"


_template_cache: dict[int, list[tuple[tuple[int, int, bool], ...]]] = {}


def _templates_for(geom: Any) -> list[tuple[tuple[int, int, bool], ...]]:
    key = id(geom)
    cached = _template_cache.get(key)
    if cached is None:
        cached = [tuple((int(dr), int(dc), bool(up)) for dr, dc, up in tpl) for tpl in geom.templates()]
        _template_cache[key] = cached
    return cached
