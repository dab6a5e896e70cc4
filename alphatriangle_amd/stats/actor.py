"""In-process stats + persistence service (replacement of the detached Ray ``TrieyeActor``).

Surface kept from the reference call sites (alphatriangle/training/setup.py:179-199,
alphatriangle/training/loop.py:97-121,193-206,390-391, alphatriangle/training/runner.py:36-38,
287-291): ``log_event``, ``log_batch_events``, ``process_and_log``, ``force_process_and_log``,
``save_training_state``, ``load_initial_state``, ``get_actor_name``,
``get_run_base_dir_str``, ``get_mlflow_run_id``, ``shutdown``.

Design: producers push into a native lock-free ring (never block, never pickle);
one stats thread drains it, aggregates and writes to the sinks; one writer thread
serialises checkpoints so the learner never waits on disk.
"""

from __future__ import annotations

import json
import logging
import queue
import threading
import time
from pathlib import Path
from typing import Any

from .. import native
from .aggregator import StatsAggregator
from .config import TrieyeConfig
from .events import RawMetricEvent
from .serializer import BufferData, CheckpointData, LoadedTrainingState, Serializer
from .sinks import JsonlSink, MetricSink, MlflowFileSink, TensorBoardSink

logger = logging.getLogger(__name__)


class _Interner:
    def __init__(self) -> None:
        self._ids: dict[str, int] = {}
        self._names: list[str] = []
        self._lock = threading.Lock()

    def id(self, name: str) -> int:
        i = self._ids.get(name)
        if i is None:
            with self._lock:
                i = self._ids.get(name)
                if i is None:
                    i = len(self._names)
                    self._names.append(name)
                    self._ids[name] = i
        return i

    def name(self, i: int) -> str:
        return self._names[i]


class StatsActor:
    def __init__(self, config: TrieyeConfig, name: str | None = None, start_threads: bool = True):
        self.config = config
        self.persistence = config.persistence
        self.name = name or f"trieye_actor_{config.run_name}"
        self.serializer = Serializer()
        self.run_dir = self.persistence.get_run_base_dir()
        for d in (
            self.persistence.get_checkpoint_dir(),
            self.persistence.get_buffer_dir(),
            self.persistence.get_log_dir(),
            self.persistence.get_tensorboard_log_dir(),
            self.persistence.get_profile_data_dir(),
        ):
            d.mkdir(parents=True, exist_ok=True)
        self._ring = native.host().MetricRing(config.stats.ring_capacity)
        self._names = _Interner()
        self._agg = StatsAggregator(config.stats)
        self._agg_lock = threading.Lock()
        self.sinks: dict[str, MetricSink] = {
            "jsonl": JsonlSink(self.run_dir / "metrics.jsonl"),
            "tensorboard": TensorBoardSink(self.persistence.get_tensorboard_log_dir()),
        }
        try:
            self.sinks["mlflow"] = MlflowFileSink(self.persistence.get_mlflow_abs_path(), config.app_name, config.run_name)
        except OSError as exc:  # pragma: no cover
            logger.warning("MLflow file sink disabled: %s", exc)
        self._latest_step = 0
        self._stop = threading.Event()
        self._wake = threading.Event()
        self._save_q: queue.Queue[tuple[str, Any] | None] = queue.Queue()
        self._saves_in_flight = 0
        self._saves_lock = threading.Condition()
        self.last_save_error: str | None = None
        self.saved_checkpoints: list[Path] = []
        self._file_handler: logging.Handler | None = None
        self._threads: list[threading.Thread] = []
        if start_threads:
            t1 = threading.Thread(target=self._stats_loop, name="atri-stats", daemon=True)
            t2 = threading.Thread(target=self._writer_loop, name="atri-ckpt-writer", daemon=True)
            t1.start()
            t2.start()
            self._threads = [t1, t2]

    # ------------------------------------------------------------------ identity
    def get_actor_name(self) -> str:
        return self.name

    def get_run_base_dir_str(self) -> str:
        return str(self.run_dir)

    def get_mlflow_run_id(self) -> str | None:
        sink = self.sinks.get("mlflow")
        return getattr(sink, "run_id", None)

    def attach_log_file(self) -> Path:
        """Add a plain file handler under ``logs/`` to the root logger."""
        path = self.persistence.get_log_dir() / f"{self.config.run_name}.log"
        if self._file_handler is None:
            handler = logging.FileHandler(path, mode="a", encoding="utf-8")
            handler.setFormatter(logging.Formatter("%(asctime)s [%(levelname)s] %(name)s: %(message)s"))
            logging.getLogger().addHandler(handler)
            self._file_handler = handler
        return path

    def save_run_config(self, configs: dict[str, Any]) -> Path:
        path = self.run_dir / self.persistence.CONFIG_FILENAME
        path.write_text(json.dumps(configs, indent=2, default=str))
        return path

    def log_params(self, params: dict[str, Any]) -> None:
        for sink in self.sinks.values():
            try:
                sink.log_params(params)
            except Exception as exc:  # pragma: no cover
                logger.warning("log_params failed on %s: %s", sink.name, exc)

    # ------------------------------------------------------------------ metric ingest
    def log_event(self, event: RawMetricEvent) -> bool:
        if not event.is_valid():
            return False
        keys: list[int] = []
        vals: list[float] = []
        for k, v in event.context.items():
            try:
                fv = float(v)
            except (TypeError, ValueError):
                continue
            keys.append(self._names.id(k))
            vals.append(fv)
        return bool(
            self._ring.push(self._names.id(event.name), float(event.value), int(event.global_step), float(event.timestamp), keys, vals)
        )

    def log_scalar(self, name: str, value: float, global_step: int) -> bool:
        """Fast path without building an event object."""
        return bool(self._ring.push(self._names.id(name), float(value), int(global_step), time.time(), [], []))

    def log_array(self, name: str, values: Any, global_step: int) -> int:
        """Thousands of samples of one metric in one call (per-game-step metrics of the GPU actor: one value per game).

        The samples go straight into the aggregator - same aggregation as the same number of ``log_event`` calls,
        without one Python object and one ring slot per sample.  Non-finite samples are dropped.
        """
        import numpy as np

        arr = np.asarray(values, dtype=np.float64).reshape(-1)
        arr = arr[np.isfinite(arr)]
        if arr.size == 0:
            return 0
        with self._agg_lock:
            self._agg.add_many(name, arr.tolist(), int(global_step))
        return int(arr.size)

    def log_batch_events(self, events: list[RawMetricEvent]) -> int:
        return sum(1 for e in events if self.log_event(e))

    def _drain(self) -> int:
        raw = self._ring.drain()
        if not raw:
            return 0
        with self._agg_lock:
            for name_id, value, step, ts, ck, cv in raw:
                ctx = {self._names.name(k): v for k, v in zip(ck, cv, strict=True)}
                self._agg.add(RawMetricEvent(name=self._names.name(name_id), value=value, global_step=step, timestamp=ts, context=ctx))
        return len(raw)

    def _emit(self, global_step: int, force: bool) -> int:
        with self._agg_lock:
            due = self._agg.collect(global_step, force=force)
        now = time.time()
        for metric, value, step in due:
            for target in metric.log_to:
                sink = self.sinks.get(target)
                if sink is not None:
                    try:
                        sink.log_metric(metric.name, value, step, now)
                    except Exception as exc:  # pragma: no cover
                        logger.warning("sink %s failed: %s", target, exc)
        if due:
            for sink in self.sinks.values():
                sink.flush()
        return len(due)

    def process_and_log(self, global_step: int) -> None:
        """Non-blocking: record the step and wake the stats thread."""
        self._latest_step = max(self._latest_step, int(global_step))
        self._wake.set()

    def force_process_and_log(self, global_step: int) -> int:
        """Synchronous drain + flush of every pending metric."""
        self._latest_step = max(self._latest_step, int(global_step))
        self._drain()
        return self._emit(self._latest_step, force=True)

    @property
    def dropped_events(self) -> int:
        return int(self._ring.dropped)

    def _stats_loop(self) -> None:
        interval = self.config.stats.processing_interval_seconds
        while not self._stop.is_set():
            self._wake.wait(timeout=interval)
            self._wake.clear()
            try:
                self._drain()
                self._emit(self._latest_step, force=False)
            except Exception:  # pragma: no cover
                logger.exception("stats thread error")

    # ------------------------------------------------------------------ checkpoints
    def save_training_state(
        self,
        nn_state_dict: dict[str, Any],
        optimizer_state_dict: dict[str, Any],
        buffer_content: Any,
        global_step: int,
        episodes_played: int,
        total_simulations_run: int,
        is_best: bool = False,
        save_buffer: bool = False,
        model_config_dict: dict[str, Any] | None = None,
        env_config_dict: dict[str, Any] | None = None,
        actor_state: dict[str, Any] | None = None,
    ) -> None:
        """Queue a checkpoint (and optionally the buffer) for the writer thread; returns immediately."""
        ckpt = CheckpointData(
            run_name=self.config.run_name,
            global_step=int(global_step),
            episodes_played=int(episodes_played),
            total_simulations_run=int(total_simulations_run),
            model_config_dict=model_config_dict or {},
            env_config_dict=env_config_dict or {},
            model_state_dict=nn_state_dict,
            optimizer_state_dict=optimizer_state_dict,
            actor_state=actor_state or {},
        )
        buf: BufferData | None = None
        if save_buffer and buffer_content is not None:
            buf = buffer_content if isinstance(buffer_content, BufferData) else self.serializer.prepare_buffer_data(buffer_content)
        with self._saves_lock:
            self._saves_in_flight += 1
        self._save_q.put(("save", (ckpt, buf, bool(is_best))))
        if not self._threads:
            self._writer_once()

    def _do_save(self, ckpt: CheckpointData, buf: BufferData | None, is_best: bool) -> None:
        p = self.persistence
        ck_dir = p.get_checkpoint_dir()
        step_path = ck_dir / f"checkpoint_step_{ckpt.global_step}.pkl"
        self.serializer.save_checkpoint(ckpt, step_path)
        self.serializer.save_checkpoint(ckpt, ck_dir / p.LATEST_CHECKPOINT_FILENAME)
        if is_best:
            self.serializer.save_checkpoint(ckpt, ck_dir / p.BEST_CHECKPOINT_FILENAME)
        self.saved_checkpoints.append(step_path)
        while len(self.saved_checkpoints) > p.KEEP_LAST_CHECKPOINTS:
            old = self.saved_checkpoints.pop(0)
            try:
                old.unlink(missing_ok=True)
            except OSError:  # pragma: no cover
                pass
        if buf is not None:
            b_dir = p.get_buffer_dir()
            self.serializer.save_buffer(buf, b_dir / f"buffer_step_{ckpt.global_step}.pkl")
            self.serializer.save_buffer(buf, b_dir / p.BUFFER_FILENAME)
            for old in sorted(b_dir.glob("buffer_step_*.pkl"), key=lambda q: q.stat().st_mtime)[:-1]:
                old.unlink(missing_ok=True)
        logger.info("Saved checkpoint at step %d%s", ckpt.global_step, " (+buffer)" if buf is not None else "")

    def _writer_once(self) -> bool:
        try:
            job = self._save_q.get(timeout=0.2)
        except queue.Empty:
            return False
        if job is None:
            return True
        _, (ckpt, buf, is_best) = job
        try:
            self._do_save(ckpt, buf, is_best)
            self.last_save_error = None
        except Exception as exc:
            self.last_save_error = str(exc)
            logger.exception("checkpoint save failed")
        finally:
            with self._saves_lock:
                self._saves_in_flight -= 1
                self._saves_lock.notify_all()
        return False

    def _writer_loop(self) -> None:
        while not self._stop.is_set() or not self._save_q.empty():
            if self._writer_once():
                break

    def wait_for_saves(self, timeout: float = 60.0) -> bool:
        deadline = time.monotonic() + timeout
        with self._saves_lock:
            while self._saves_in_flight > 0:
                left = deadline - time.monotonic()
                if left <= 0:
                    return False
                self._saves_lock.wait(timeout=left)
        return True

    # ------------------------------------------------------------------ resume
    def _latest_previous_run(self) -> str | None:
        root = self.persistence.get_runs_root_dir()
        if not root.exists():
            return None
        best: tuple[float, str] | None = None
        for run in root.iterdir():
            if not run.is_dir() or run.name == self.config.run_name:
                continue
            latest = run / self.persistence.CHECKPOINT_SAVE_DIR_NAME / self.persistence.LATEST_CHECKPOINT_FILENAME
            if latest.exists():
                mt = latest.stat().st_mtime
                if best is None or mt > best[0]:
                    best = (mt, run.name)
        return best[1] if best else None

    def load_initial_state(self) -> LoadedTrainingState:
        """Explicit paths win; else the current run's latest files; else (auto-resume) the newest other run."""
        p = self.persistence
        ckpt_path: Path | None = None
        buf_path: Path | None = None
        if p.LOAD_CHECKPOINT_PATH:
            ckpt_path = Path(p.LOAD_CHECKPOINT_PATH)
        if p.LOAD_BUFFER_PATH:
            buf_path = Path(p.LOAD_BUFFER_PATH)
        if ckpt_path is None and p.AUTO_RESUME_LATEST:
            own = p.get_checkpoint_dir() / p.LATEST_CHECKPOINT_FILENAME
            if own.exists():
                ckpt_path = own
                cand = p.get_buffer_dir() / p.BUFFER_FILENAME
                buf_path = buf_path or (cand if cand.exists() else None)
            else:
                prev = self._latest_previous_run()
                if prev:
                    ckpt_path = p.get_checkpoint_dir(prev) / p.LATEST_CHECKPOINT_FILENAME
                    cand = p.get_buffer_dir(prev) / p.BUFFER_FILENAME
                    buf_path = buf_path or (cand if cand.exists() else None)
        state = LoadedTrainingState()
        if ckpt_path is not None:
            if ckpt_path.exists():
                state.checkpoint_data = self.serializer.load_checkpoint(ckpt_path)
                if state.checkpoint_data is not None:
                    logger.info("Loaded checkpoint %s (step %d)", ckpt_path, state.checkpoint_data.global_step)
            else:
                logger.warning("Checkpoint path %s does not exist.", ckpt_path)
        if buf_path is not None and p.SAVE_BUFFER:
            if buf_path.exists():
                state.buffer_data = self.serializer.load_buffer(buf_path)
                if state.buffer_data is not None:
                    logger.info("Loaded buffer %s (%d experiences)", buf_path, len(state.buffer_data.buffer_list))
            else:
                logger.warning("Buffer path %s does not exist.", buf_path)
        return state

    # ------------------------------------------------------------------ teardown
    def shutdown(self, timeout: float = 15.0) -> None:
        try:
            self.force_process_and_log(self._latest_step)
        except Exception:  # pragma: no cover
            logger.exception("final metric flush failed")
        self.wait_for_saves(timeout)
        self._stop.set()
        self._wake.set()
        self._save_q.put(None)
        for t in self._threads:
            t.join(timeout=timeout)
        for sink in self.sinks.values():
            try:
                sink.close()
            except Exception:  # pragma: no cover
                pass
        if self._file_handler is not None:
            logging.getLogger().removeHandler(self._file_handler)
            self._file_handler.close()
            self._file_handler = None


TrieyeActor = StatsActor
