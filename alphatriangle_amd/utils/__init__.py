"""Utilities (parity: alphatriangle/utils/__init__.py:20-38)."""

from .geometry import is_point_in_polygon
from .helpers import format_eta, get_device, normalize_color_for_matplotlib, set_random_seeds
from .sumtree import SumTree
from .types import (
    ActionType,
    Experience,
    ExperienceBatch,
    ExperienceColumns,
    PERBatchSample,
    PolicyTargetMapping,
    PolicyValueOutput,
    StateType,
    experience_to_columns,
)

__all__ = [
    "ActionType",
    "Experience",
    "ExperienceBatch",
    "ExperienceColumns",
    "PERBatchSample",
    "PolicyTargetMapping",
    "PolicyValueOutput",
    "StateType",
    "SumTree",
    "experience_to_columns",
    "format_eta",
    "get_device",
    "is_point_in_polygon",
    "normalize_color_for_matplotlib",
    "set_random_seeds",
]
