"""Game engine package: native rules core plus the Python ``GameState`` / ``Shape`` surface."""

from ..config.env_config import EnvConfig
from .game_state import GameState, Shape, get_geometry

__all__ = ["EnvConfig", "GameState", "Shape", "get_geometry"]
