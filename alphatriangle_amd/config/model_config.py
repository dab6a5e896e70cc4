"""Neural network architecture configuration.

Field names, defaults and validation rules follow
alphatriangle/config/model_config.py:17-157 so that checkpoints and user
overrides carry over unchanged.
"""

from __future__ import annotations

from typing import Literal

from pydantic import BaseModel, Field, model_validator


class ModelConfig(BaseModel):
    """CNN (+ optional Transformer encoder) policy / C51 value network, ~3M parameters by default."""

    GRID_INPUT_CHANNELS: int = Field(default=1, gt=0)

    CONV_FILTERS: list[int] = Field(default=[32, 64, 128])
    CONV_KERNEL_SIZES: list[int | tuple[int, int]] = Field(default=[3, 3, 3])
    CONV_STRIDES: list[int | tuple[int, int]] = Field(default=[1, 1, 1])
    CONV_PADDING: list[int | tuple[int, int] | str] = Field(default=[1, 1, 1])

    NUM_RESIDUAL_BLOCKS: int = Field(default=2, ge=0)
    RESIDUAL_BLOCK_FILTERS: int = Field(default=128, gt=0)

    USE_TRANSFORMER: bool = Field(default=True)
    TRANSFORMER_DIM: int = Field(default=128, gt=0)
    TRANSFORMER_HEADS: int = Field(default=4, gt=0)
    TRANSFORMER_LAYERS: int = Field(default=2, ge=0)
    TRANSFORMER_FC_DIM: int = Field(default=256, gt=0)

    FC_DIMS_SHARED: list[int] = Field(default=[128])
    POLICY_HEAD_DIMS: list[int] = Field(default=[128])

    NUM_VALUE_ATOMS: int = Field(default=51, gt=1)
    VALUE_MIN: float = Field(default=-10.0)
    VALUE_MAX: float = Field(default=10.0)
    VALUE_HEAD_DIMS: list[int] = Field(default=[128])

    ACTIVATION_FUNCTION: Literal["ReLU", "GELU", "SiLU", "Tanh", "Sigmoid"] = Field(default="ReLU")
    USE_BATCH_NORM: bool = Field(default=True)

    # 3 slots * 7 shape features + 3 availability flags + 6 explicit features
    OTHER_NN_INPUT_FEATURES_DIM: int = Field(default=30, gt=0)

    @model_validator(mode="after")
    def _conv_lists_consistent(self) -> "ModelConfig":
        n = len(self.CONV_FILTERS)
        if not (len(self.CONV_KERNEL_SIZES) == n and len(self.CONV_STRIDES) == n and len(self.CONV_PADDING) == n):
            raise ValueError("CONV_FILTERS, CONV_KERNEL_SIZES, CONV_STRIDES and CONV_PADDING must have equal length.")
        return self

    @model_validator(mode="after")
    def _transformer_consistent(self) -> "ModelConfig":
        if self.USE_TRANSFORMER:
            if self.TRANSFORMER_LAYERS < 0:
                raise ValueError("TRANSFORMER_LAYERS cannot be negative.")
            if self.TRANSFORMER_LAYERS > 0:
                if self.TRANSFORMER_DIM <= 0:
                    raise ValueError("TRANSFORMER_DIM must be positive when TRANSFORMER_LAYERS > 0.")
                if self.TRANSFORMER_HEADS <= 0:
                    raise ValueError("TRANSFORMER_HEADS must be positive when TRANSFORMER_LAYERS > 0.")
                if self.TRANSFORMER_DIM % self.TRANSFORMER_HEADS != 0:
                    raise ValueError(
                        f"TRANSFORMER_DIM ({self.TRANSFORMER_DIM}) must be divisible by "
                        f"TRANSFORMER_HEADS ({self.TRANSFORMER_HEADS})."
                    )
                if self.TRANSFORMER_FC_DIM <= 0:
                    raise ValueError("TRANSFORMER_FC_DIM must be positive when TRANSFORMER_LAYERS > 0.")
        return self

    @model_validator(mode="after")
    def _value_range(self) -> "ModelConfig":
        if self.VALUE_MIN >= self.VALUE_MAX:
            raise ValueError("VALUE_MIN must be strictly less than VALUE_MAX.")
        return self

    @classmethod
    def cnn_only(cls) -> "ModelConfig":
        """CNN policy + C51 value network without the Transformer body."""
        return cls(USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0)

    @classmethod
    def large_transformer(cls, other_dim: int = 30) -> "ModelConfig":
        """8-layer Transformer variant used by the large-grid benchmark configuration."""
        return cls(TRANSFORMER_LAYERS=8, OTHER_NN_INPUT_FEATURES_DIM=other_dim)


ModelConfig.model_rebuild(force=True)
