"""The training loop: interleaves learner steps with self-play collection.

Observable behaviour follows alphatriangle/training/loop.py:23-420 - counters, the
metric events and their names, weight-update cadence (``WORKER_UPDATE_FREQ_STEPS``),
checkpoint cadence (``CHECKPOINT_SAVE_FREQ_STEPS``, buffer saves at
``BUFFER_SAVE_FREQ_STEPS``), stop handling.  Two self-play back-ends plug in behind
the same manager interface: CPU workers on threads, or the HBM-resident GPU actor
whose experiences land directly in the device replay buffer.
"""

from __future__ import annotations

import logging
import threading
import time
from typing import Any

from ..rl.core.trainer import TensorBatch
from ..rl.types import SelfPlayResult
from ..stats.events import RawMetricEvent
from .components import TrainingComponents
from .loop_helpers import LoopHelpers
from .worker_manager import WorkerManager

logger = logging.getLogger(__name__)

_LOSS_EVENT_NAMES = {
    "total_loss": "Loss/Total",
    "policy_loss": "Loss/Policy",
    "value_loss": "Loss/Value",
    "entropy": "Loss/Entropy",
    "mean_td_error": "Loss/Mean_Abs_TD_Error",
}


class TrainingLoop:
    def __init__(self, components: TrainingComponents):
        self.components = components
        self.train_config = components.train_config
        self.buffer = components.buffer
        self.trainer = components.trainer
        self.trieye_actor = components.trieye_actor
        self.trieye_config = components.trieye_config
        self.serializer = components.serializer

        self.global_step = 0
        self.episodes_played = 0
        self.total_simulations_run = 0
        self.weight_update_count = 0
        self.start_time = time.time()
        self.initial_global_step = 0
        self.stop_requested = threading.Event()
        self.training_complete = False
        self.training_exception: Exception | None = None
        self._buffer_ready_logged = False

        if components.gpu_actor is not None:
            from .gpu_manager import GpuActorManager

            self.worker_manager: Any = GpuActorManager(components)
            self.worker_manager.train_fn = self._train_k_steps
            self.worker_manager.ready_fn = self.buffer.is_ready
        else:
            self.worker_manager = WorkerManager(components)
        self.loop_helpers = LoopHelpers(components, self._get_loop_state)
        logger.info("TrainingLoop initialized (%s self-play).", "GPU" if components.gpu_actor is not None else "CPU")

    # ------------------------------------------------------------------ state
    def _get_loop_state(self) -> dict[str, Any]:
        return {
            "global_step": self.global_step,
            "initial_global_step": self.initial_global_step,
            "episodes_played": self.episodes_played,
            "total_simulations_run": self.total_simulations_run,
            "weight_update_count": self.weight_update_count,
            "buffer_size": len(self.buffer),
            "buffer_capacity": self.buffer.capacity,
            "num_active_workers": self.worker_manager.get_num_active_workers(),
            "num_pending_tasks": self.worker_manager.get_num_pending_tasks(),
            "start_time": self.start_time,
            "num_workers": self.train_config.NUM_SELF_PLAY_WORKERS,
        }

    def set_initial_state(self, global_step: int, episodes_played: int, total_simulations: int) -> None:
        self.global_step = global_step
        self.initial_global_step = global_step
        self.episodes_played = episodes_played
        self.total_simulations_run = total_simulations
        freq = self.train_config.WORKER_UPDATE_FREQ_STEPS
        self.weight_update_count = global_step // freq if freq > 0 else 0
        self.loop_helpers.reset_rate_counters(global_step, episodes_played, total_simulations)
        logger.info(
            "TrainingLoop initial state set: Step=%d, Episodes=%d, Sims=%d, WeightUpdates=%d",
            global_step,
            episodes_played,
            total_simulations,
            self.weight_update_count,
        )

    def initialize_workers(self) -> None:
        self.worker_manager.initialize_workers()

    def request_stop(self) -> None:
        if not self.stop_requested.is_set():
            logger.info("Stop requested for TrainingLoop.")
            self.stop_requested.set()

    # ------------------------------------------------------------------ events
    def _send_event_async(self, name: str, value: float | int, context: dict | None = None) -> None:
        if not self.trieye_actor:
            return
        try:
            self.trieye_actor.log_event(
                RawMetricEvent(name=name, value=value, global_step=self.global_step, timestamp=time.time(), context=context or {})
            )
        except Exception as exc:
            logger.error("Failed to send event '%s': %s", name, exc)

    def _send_batch_events_async(self, events: list[RawMetricEvent]) -> None:
        if self.trieye_actor and events:
            try:
                self.trieye_actor.log_batch_events(events)
            except Exception as exc:
                logger.error("Failed to send batch events: %s", exc)

    # ------------------------------------------------------------------ self-play results
    def _note_buffer_ready(self) -> None:
        size = len(self.buffer)
        if not self._buffer_ready_logged and size >= self.buffer.min_size_to_train:
            logger.info("--- Buffer ready for training (Size: %d/%d) ---", size, self.buffer.min_size_to_train)
            self._buffer_ready_logged = True

    def _process_self_play_result(self, result: SelfPlayResult, worker_id: int) -> None:
        valid, invalid = self.loop_helpers.validate_experiences(result.episode_experiences)
        if invalid > 0:
            logger.warning("Worker %d: %d invalid experiences were filtered out.", worker_id, invalid)
        if not valid:
            logger.error(
                "Worker %d: episode produced NO valid experiences (Steps: %d, Score: %.2f).",
                worker_id,
                result.episode_steps,
                result.final_score,
            )
            return
        try:
            self.buffer.add_batch(valid)
        except Exception as exc:
            logger.error("Error adding batch to buffer from worker %d: %s", worker_id, exc, exc_info=True)
            return
        self._note_buffer_ready()
        self.episodes_played += 1
        self.total_simulations_run += result.total_simulations
        self._send_event_async("Progress/Episodes_Played", self.episodes_played)
        self._send_event_async("Progress/Total_Simulations", self.total_simulations_run)

    def _process_gpu_report(self, report: Any) -> None:
        """The GPU actor already wrote its experiences into the device buffer; account for them."""
        self._note_buffer_ready()
        self.episodes_played += report.episodes_finished
        self.total_simulations_run += report.simulations
        for ctx in report.episode_contexts:
            ctx = dict(ctx)
            ctx["trainer_step"] = report.trainer_step
            self._send_event_async("episode_end", 1.0, ctx)
        self._send_event_async("mcts_step", float(report.simulations), {"moves": report.moves})
        # per-move aggregates over all games (the CPU worker sends these per game step)
        self._send_event_async("step_reward", float(report.mean_step_reward), {"moves": report.moves})
        self._send_event_async("triangles_cleared_step", float(report.triangles_cleared), {"moves": report.moves})
        if report.episode_contexts:
            scores = [c["score"] for c in report.episode_contexts]
            self._send_event_async("current_score", float(sum(scores) / len(scores)), {"episodes": len(scores)})
        if report.episodes_finished:
            self._send_event_async("Progress/Episodes_Played", self.episodes_played)
        self._send_event_async("Progress/Total_Simulations", self.total_simulations_run)

    # ------------------------------------------------------------------ persistence
    def _trigger_save_state(self, is_best: bool = False, save_buffer: bool = False) -> None:
        if not self.trieye_actor:
            logger.error("Cannot save state: stats actor handle is missing.")
            return
        logger.info("Requesting state save at step %d. Best=%s, SaveBuffer=%s", self.global_step, is_best, save_buffer)
        try:
            nn_state = self.components.nn.get_weights()
            opt_state = self.serializer.prepare_optimizer_state(self.trainer.optimizer)
            buffer_data = self.serializer.prepare_buffer_data(self.buffer) if save_buffer else None
            if buffer_data is None and save_buffer:
                logger.error("Failed to prepare buffer data, cannot save buffer.")
                save_buffer = False
            self.trieye_actor.save_training_state(
                nn_state_dict=nn_state,
                optimizer_state_dict=opt_state,
                buffer_content=(buffer_data.buffer_list if buffer_data else []),
                global_step=self.global_step,
                episodes_played=self.episodes_played,
                total_simulations_run=self.total_simulations_run,
                is_best=is_best,
                save_buffer=save_buffer,
                model_config_dict=self.components.model_config.model_dump(),
                env_config_dict=self.components.env_config.model_dump(),
            )
        except Exception as exc:
            logger.error("Failed to trigger save state at step %d: %s", self.global_step, exc, exc_info=True)

    # ------------------------------------------------------------------ learner
    def _run_training_step(self) -> bool:
        if not self.buffer.is_ready():
            return False
        sample = self.buffer.sample(self.train_config.BATCH_SIZE, current_train_step=self.global_step)
        if not sample:
            return False
        result = self.trainer.train_step(sample)
        if not result:
            logger.warning("Training step %d failed.", self.global_step + 1)
            return False
        loss_info, td_errors = result
        first = self.global_step == 0
        self.global_step += 1
        if first:
            logger.info("--- First training step completed (Global Step: %d) ---", self.global_step)
        self._send_event_async("step_completed", 1.0)
        if self.train_config.USE_PER:
            indices = sample.indices if isinstance(sample, TensorBatch) else sample["indices"]
            self.buffer.update_priorities(indices, td_errors)
            self._send_event_async("PER/Beta", self.buffer._calculate_beta(self.global_step))
        events = [
            RawMetricEvent(name=_LOSS_EVENT_NAMES[k], value=v, global_step=self.global_step)
            for k, v in loss_info.items()
            if k in _LOSS_EVENT_NAMES
        ]
        events.append(RawMetricEvent(name="LearningRate", value=self.trainer.get_current_lr(), global_step=self.global_step))
        self._send_batch_events_async(events)
        freq = self.train_config.WORKER_UPDATE_FREQ_STEPS
        if freq > 0 and self.global_step % freq == 0:
            try:
                self.worker_manager.update_worker_networks(self.global_step)
                self.weight_update_count += 1
                self._send_event_async("Progress/Weight_Updates_Total", self.weight_update_count)
            except Exception as exc:
                logger.error("Failed to update worker networks at step %d: %s", self.global_step, exc)
        if self.global_step % 50 == 0:
            logger.info(
                "Step %d: P Loss=%.4f, V Loss=%.4f", self.global_step, loss_info.get("policy_loss", 0.0), loss_info.get("value_loss", 0.0)
            )
        return True

    def _train_k_steps(self, k: int) -> int:
        """Up to ``k`` learner steps (GPU engine: called by the actor manager while the move's search runs)."""
        done = 0
        target = self.train_config.MAX_TRAINING_STEPS
        for _ in range(k):
            if target is not None and self.global_step >= target:
                break
            if not self._run_training_step():
                break
            done += 1
        return done

    def _checkpoint_cadence(self, steps_before: int) -> None:
        """Trigger the periodic saves for every cadence boundary crossed since ``steps_before``."""
        persistence = self.trieye_config.persistence
        ckpt_freq = self.train_config.CHECKPOINT_SAVE_FREQ_STEPS
        if ckpt_freq > 0 and self.global_step // ckpt_freq > steps_before // ckpt_freq:
            self._trigger_save_state(is_best=False, save_buffer=False)
        bfreq = persistence.BUFFER_SAVE_FREQ_STEPS
        if persistence.SAVE_BUFFER and bfreq > 0 and self.global_step // bfreq > steps_before // bfreq:
            self._trigger_save_state(is_best=False, save_buffer=True)

    # ------------------------------------------------------------------ main loop
    def run(self) -> None:
        target = self.train_config.MAX_TRAINING_STEPS
        logger.info("Starting TrainingLoop run... %s", f"Target steps: {target}" if target else "Running indefinitely")
        self.start_time = time.time()
        try:
            self.worker_manager.submit_initial_tasks()
            while not self.stop_requested.is_set():
                if target is not None and self.global_step >= target:
                    logger.info("Reached MAX_TRAINING_STEPS (%d). Stopping loop.", target)
                    self.training_complete = True
                    self.request_stop()
                    break
                trained = False
                gpu_mode = self.components.gpu_actor is not None
                steps_before = self.global_step
                if gpu_mode:
                    pass  # the actor manager runs this move's learner steps (overlapped with the search)
                elif self.buffer.is_ready():
                    trained = self._run_training_step()
                else:
                    if not self._buffer_ready_logged and self.global_step % 100 == 0:
                        logger.debug("Waiting for buffer... Size: %d/%d", len(self.buffer), self.buffer.min_size_to_train)
                    time.sleep(0.01)
                if trained:
                    self._checkpoint_cadence(steps_before)
                if self.stop_requested.is_set():
                    break
                completed = self.worker_manager.get_completed_tasks(0.1 if self.buffer.is_ready() else 0.5)
                for worker_id, item in completed:
                    if isinstance(item, SelfPlayResult):
                        try:
                            self._process_self_play_result(item, worker_id)
                        except Exception as exc:
                            logger.error("Error processing result from worker %d: %s", worker_id, exc, exc_info=True)
                    elif isinstance(item, Exception):
                        logger.error("Worker %d task failed with exception: %s", worker_id, item)
                    elif hasattr(item, "episodes_finished"):
                        self._process_gpu_report(item)
                    else:
                        logger.error("Unexpected item from worker %d: %s", worker_id, type(item))
                    self.worker_manager.submit_task(worker_id)
                if gpu_mode and self.global_step > steps_before:
                    trained = True
                    self._checkpoint_cadence(steps_before)
                if self.stop_requested.is_set():
                    break
                self.loop_helpers.log_progress_eta()
                state = self._get_loop_state()
                self._send_event_async("Buffer/Size", state["buffer_size"])
                self._send_event_async("System/Num_Active_Workers", state["num_active_workers"])
                self._send_event_async("System/Num_Pending_Tasks", state["num_pending_tasks"])
                if self.trieye_actor:
                    self.trieye_actor.process_and_log(self.global_step)
                if not completed and not trained and not self.buffer.is_ready():
                    time.sleep(0.05)
        except KeyboardInterrupt:
            logger.warning("KeyboardInterrupt received in TrainingLoop. Stopping.")
            self.request_stop()
        except Exception as exc:
            logger.critical("Unhandled exception in TrainingLoop: %s", exc, exc_info=True)
            self.training_exception = exc
            self.request_stop()
        finally:
            if self.training_exception or (self.stop_requested.is_set() and not self.training_complete):
                self.training_complete = False
            logger.info(
                "TrainingLoop finished. Complete: %s, Exception: %s", self.training_complete, self.training_exception is not None
            )

    def cleanup_actors(self) -> None:
        self.worker_manager.cleanup_actors()
