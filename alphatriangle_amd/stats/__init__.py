"""Stats + persistence service (in-process replacement of the external ``trieye`` Ray actor)."""

from .actor import StatsActor, TrieyeActor
from .config import MetricConfig, PersistenceConfig, StatsConfig, TrieyeConfig
from .events import RawMetricEvent
from .serializer import BufferData, CheckpointData, LoadedTrainingState, Serializer, experiences_from_arrays

__all__ = [
    "BufferData",
    "CheckpointData",
    "LoadedTrainingState",
    "MetricConfig",
    "PersistenceConfig",
    "RawMetricEvent",
    "Serializer",
    "StatsActor",
    "StatsConfig",
    "TrieyeActor",
    "TrieyeConfig",
    "experiences_from_arrays",
]
