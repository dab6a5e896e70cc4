"""Environment configuration of the triangle puzzle.

The reference re-exports ``EnvConfig`` from the external ``trianglengin`` package
(alphatriangle/config/__init__.py:2,12; fields seen at tests/conftest.py:33-41 and
alphatriangle/nn/model.py:123-125).  This framework owns the engine, so it owns
the config too; field names follow the reference call sites.
"""

from __future__ import annotations

from pydantic import BaseModel, Field, model_validator

_DEFAULT_RANGES: list[tuple[int, int]] = [
    (3, 12),
    (2, 13),
    (1, 14),
    (0, 15),
    (0, 15),
    (1, 14),
    (2, 13),
    (3, 12),
]


class EnvConfig(BaseModel):
    """Grid geometry, shape slots and reward constants."""

    ROWS: int = Field(default=8, gt=0)
    COLS: int = Field(default=15, gt=0)
    PLAYABLE_RANGE_PER_ROW: list[tuple[int, int]] = Field(default_factory=lambda: list(_DEFAULT_RANGES))
    NUM_SHAPE_SLOTS: int = Field(default=3, ge=1, le=4)
    MIN_LINE_LENGTH: int = Field(default=3, ge=1)
    MAX_SHAPE_TRIANGLES: int = Field(default=5, ge=1, le=5)
    SHAPE_SIZE_WEIGHTS: list[float] = Field(default_factory=lambda: [0.10, 0.20, 0.30, 0.25, 0.15])

    REWARD_PER_PLACED_TRIANGLE: float = Field(default=0.01)
    REWARD_PER_CLEARED_TRIANGLE: float = Field(default=0.5)
    REWARD_PER_STEP_ALIVE: float = Field(default=0.005)
    PENALTY_GAME_OVER: float = Field(default=-10.0)

    @model_validator(mode="after")
    def _check_geometry(self) -> "EnvConfig":
        if self.ROWS * self.COLS > 1024:
            # up to 256 cells: the 64-byte state of the host and the device engine; up to 1024: their wide builds (160 bytes)
            raise ValueError("ROWS * COLS must be <= 1024 (bitboard of sixteen 64-bit words in the wide engine builds).")
        if len(self.PLAYABLE_RANGE_PER_ROW) != self.ROWS:
            raise ValueError(
                f"PLAYABLE_RANGE_PER_ROW needs {self.ROWS} entries, got {len(self.PLAYABLE_RANGE_PER_ROW)}."
            )
        for r, (start, end) in enumerate(self.PLAYABLE_RANGE_PER_ROW):
            if not (0 <= start <= end <= self.COLS):
                raise ValueError(f"Row {r}: playable range ({start}, {end}) outside 0..{self.COLS}.")
        if sum(self.SHAPE_SIZE_WEIGHTS[: self.MAX_SHAPE_TRIANGLES]) <= 0:
            raise ValueError("SHAPE_SIZE_WEIGHTS must have positive mass over the allowed sizes.")
        return self

    @property
    def ACTION_DIM(self) -> int:
        return int(self.NUM_SHAPE_SLOTS * self.ROWS * self.COLS)

    @classmethod
    def large(cls) -> "EnvConfig":
        """12 x 21 hexagon-like board used by the large-grid benchmark configuration."""
        rows, cols = 12, 21
        ranges = []
        for r in range(rows):
            inset = max(0, 5 - r) if r < rows // 2 else max(0, r - (rows - 6))
            ranges.append((inset, cols - inset))
        return cls(ROWS=rows, COLS=cols, PLAYABLE_RANGE_PER_ROW=ranges, NUM_SHAPE_SLOTS=3)


EnvConfig.model_rebuild(force=True)
