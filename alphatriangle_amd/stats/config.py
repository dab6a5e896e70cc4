"""Configuration of the in-process stats + persistence service.

The reference takes ``TrieyeConfig`` / ``PersistenceConfig`` from the external
``trieye`` package (call sites: alphatriangle/cli.py:165-172,223-225,267-268;
alphatriangle/training/loop.py:343-347; alphatriangle/training/runner.py:156).
Field and method names used at those call sites are kept; the directory layout is
the documented one (README.md:63-79): ``.trieye_data/<app>/{mlruns, runs/<run>/
{checkpoints, buffers, logs, tensorboard, profile_data, configs.json}}``.
"""

from __future__ import annotations

import time
from pathlib import Path
from typing import Literal

from pydantic import BaseModel, Field, model_validator

AggregationMethod = Literal["mean", "sum", "latest", "min", "max", "count", "rate"]


class MetricConfig(BaseModel):
    """How one raw event stream becomes a logged metric."""

    name: str
    source_event: str | None = Field(default=None, description="Raw event name; defaults to ``name``.")
    context_key: str | None = Field(default=None, description="Take the value from this context key instead.")
    aggregation: AggregationMethod = "mean"
    log_frequency_steps: int = Field(default=1, ge=0, description="Emit at most once per this many global steps.")
    log_frequency_seconds: float = Field(default=0.0, ge=0)
    log_to: list[Literal["jsonl", "tensorboard", "mlflow", "console"]] = Field(
        default_factory=lambda: ["jsonl", "tensorboard", "mlflow"]
    )

    @property
    def event_name(self) -> str:
        return self.source_event or self.name


def default_metrics() -> list[MetricConfig]:
    """Metric inventory produced by the workers, the loop and the runner."""
    m: list[MetricConfig] = []
    for name in ("Loss/Total", "Loss/Policy", "Loss/Value", "Loss/Entropy", "Loss/Mean_Abs_TD_Error", "LearningRate", "PER/Beta"):
        m.append(MetricConfig(name=name, aggregation="mean", log_frequency_steps=10))
    for name in ("Progress/Episodes_Played", "Progress/Total_Simulations", "Progress/Weight_Updates_Total", "Buffer/Size",
                 "System/Num_Active_Workers", "System/Num_Pending_Tasks"):
        m.append(MetricConfig(name=name, aggregation="latest", log_frequency_steps=10))
    m.append(MetricConfig(name="Rate/Steps_Per_Sec", source_event="step_completed", aggregation="rate", log_frequency_steps=10))
    m.append(MetricConfig(name="Rate/Episodes_Per_Sec", source_event="episode_end", aggregation="rate", log_frequency_steps=10))
    m.append(MetricConfig(name="Rate/Simulations_Per_Sec", source_event="mcts_step", aggregation="rate", log_frequency_steps=10))
    m.append(MetricConfig(name="RL/Score", source_event="episode_end", context_key="score", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="RL/Episode_Length", source_event="episode_end", context_key="length", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="RL/Triangles_Cleared_Episode", source_event="episode_end", context_key="triangles_cleared", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="MCTS/Simulations_Per_Episode", source_event="episode_end", context_key="simulations", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="MCTS/Visits_Per_Step", source_event="mcts_step", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="RL/Step_Reward", source_event="step_reward", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="RL/Triangles_Cleared_Step", source_event="triangles_cleared_step", aggregation="mean", log_frequency_steps=10))
    m.append(MetricConfig(name="RL/Current_Score", source_event="current_score", aggregation="latest", log_frequency_steps=10))
    m.append(MetricConfig(name="System/Training_Status_Code", source_event="training_status_code", aggregation="latest", log_frequency_steps=0))
    m.append(MetricConfig(name="System/Error_Flag", source_event="error_message_flag", aggregation="latest", log_frequency_steps=0))
    return m


class StatsConfig(BaseModel):
    processing_interval_seconds: float = Field(default=1.0, gt=0)
    ring_capacity: int = Field(default=1 << 16, ge=16)
    metrics: list[MetricConfig] = Field(default_factory=default_metrics)
    console_every_steps: int = Field(default=0, ge=0)


class PersistenceConfig(BaseModel):
    ROOT_DATA_DIR: str = Field(default=".trieye_data")
    APP_NAME: str = Field(default="AlphaTriangle")
    RUN_NAME: str = Field(default_factory=lambda: f"run_{time.strftime('%Y%m%d_%H%M%S')}")
    RUNS_DIR_NAME: str = "runs"
    MLFLOW_DIR_NAME: str = "mlruns"
    CHECKPOINT_SAVE_DIR_NAME: str = "checkpoints"
    BUFFER_SAVE_DIR_NAME: str = "buffers"
    LOG_DIR_NAME: str = "logs"
    TENSORBOARD_DIR_NAME: str = "tensorboard"
    PROFILE_DIR_NAME: str = "profile_data"
    LATEST_CHECKPOINT_FILENAME: str = "latest.pkl"
    BEST_CHECKPOINT_FILENAME: str = "best.pkl"
    BUFFER_FILENAME: str = "buffer.pkl"
    CONFIG_FILENAME: str = "configs.json"
    SAVE_BUFFER: bool = True
    BUFFER_SAVE_FREQ_STEPS: int = Field(default=10_000, ge=1)
    AUTO_RESUME_LATEST: bool = True
    LOAD_CHECKPOINT_PATH: str | None = None
    LOAD_BUFFER_PATH: str | None = None
    KEEP_LAST_CHECKPOINTS: int = Field(default=3, ge=1)

    def _app_root(self) -> Path:
        return Path(self.ROOT_DATA_DIR).resolve() / self.APP_NAME

    def get_runs_root_dir(self) -> Path:
        return self._app_root() / self.RUNS_DIR_NAME

    def get_run_base_dir(self, run_name: str | None = None) -> Path:
        return self.get_runs_root_dir() / (run_name or self.RUN_NAME)

    def get_mlflow_abs_path(self) -> Path:
        return self._app_root() / self.MLFLOW_DIR_NAME

    @property
    def MLFLOW_TRACKING_URI(self) -> str:
        return self.get_mlflow_abs_path().as_uri()

    def get_checkpoint_dir(self, run_name: str | None = None) -> Path:
        return self.get_run_base_dir(run_name) / self.CHECKPOINT_SAVE_DIR_NAME

    def get_buffer_dir(self, run_name: str | None = None) -> Path:
        return self.get_run_base_dir(run_name) / self.BUFFER_SAVE_DIR_NAME

    def get_log_dir(self, run_name: str | None = None) -> Path:
        return self.get_run_base_dir(run_name) / self.LOG_DIR_NAME

    def get_tensorboard_log_dir(self, run_name: str | None = None) -> Path:
        return self.get_run_base_dir(run_name) / self.TENSORBOARD_DIR_NAME

    def get_profile_data_dir(self, run_name: str | None = None) -> Path:
        return self.get_run_base_dir(run_name) / self.PROFILE_DIR_NAME


class TrieyeConfig(BaseModel):
    """Run identity + stats + persistence settings (name kept from the reference call sites)."""

    app_name: str = Field(default="AlphaTriangle")
    run_name: str = Field(default_factory=lambda: f"run_{time.strftime('%Y%m%d_%H%M%S')}")
    persistence: PersistenceConfig = Field(default_factory=PersistenceConfig)
    stats: StatsConfig = Field(default_factory=StatsConfig)

    @model_validator(mode="after")
    def _sync(self) -> "TrieyeConfig":
        self.persistence.APP_NAME = self.app_name
        self.persistence.RUN_NAME = self.run_name
        return self


for _m in (MetricConfig, StatsConfig, PersistenceConfig, TrieyeConfig):
    _m.model_rebuild(force=True)
