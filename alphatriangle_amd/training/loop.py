"""Round scheduler: self-play sources deliver, the learner takes its quota of steps, cadences fire.

Behaviour kept from alphatriangle/training/loop.py:298-416 (what an observer can see): the
counters, every metric name and its cadence, weight publication every
``WORKER_UPDATE_FREQ_STEPS`` learner steps, checkpoints every ``CHECKPOINT_SAVE_FREQ_STEPS``
(buffer saves every ``BUFFER_SAVE_FREQ_STEPS``), stop / completion / failure outcomes.

Structure (this framework's own).  A run is a sequence of ROUNDS, and the pace of a round is
set by the self-play source, not by a polling learner:

  1. negotiate   how many learner steps this round gets.  Single process: a local decision
                 (replay ready?  steps left?).  Data-parallel ranks agree on it through the
                 host-side control plane (MIN over ranks of ready / quota, MAX of stop), so no
                 rank ever enters the gradient all-reduce alone; a rank that stopped answering
                 surfaces as ``PeerLostError`` and the survivors save and leave non-zero.
  2. play        ``source.play_round(quota, learn)``.  The GPU source enqueues the whole search
                 of one move, runs ``learn(quota)`` on a second HIP stream while the search
                 occupies the GPU, then commits the move into the HBM replay.  The thread-worker
                 source hands over every episode that finished since the last round and then runs
                 ``learn(quota)``.
  3. account     deliveries update counters and metrics.

Everything that happens "every N learner steps" is a :class:`StepCadence` fired from the one
place that advances ``global_step``.
"""

from __future__ import annotations

import logging
import threading
import time
from dataclasses import dataclass
from typing import Any, Callable, Protocol

import numpy as np

from ..rl.core.trainer import TensorBatch
from ..rl.types import SelfPlayResult
from ..stats.events import RawMetricEvent
from .components import TrainingComponents
from .loop_helpers import LoopHelpers
from .worker_manager import WorkerManager

logger = logging.getLogger(__name__)

LOSS_METRICS = (
    ("total_loss", "Loss/Total"),
    ("policy_loss", "Loss/Policy"),
    ("value_loss", "Loss/Value"),
    ("entropy", "Loss/Entropy"),
    ("mean_td_error", "Loss/Mean_Abs_TD_Error"),
)


class StepCadence:
    """Calls ``action(step)`` whenever the learner step count reaches a multiple of ``every`` (0 disables it)."""

    def __init__(self, every: int, action: Callable[[int], None], label: str):
        self.every = max(0, int(every or 0))
        self.action = action
        self.label = label

    def tick(self, step: int) -> bool:
        if self.every <= 0 or step <= 0 or step % self.every != 0:
            return False
        try:
            self.action(step)
        except Exception as exc:
            logger.error("%s at step %d failed: %s", self.label, step, exc, exc_info=True)
        return True


@dataclass
class RoundQuota:
    steps: int
    stop: bool = False


class SelfPlaySource(Protocol):
    """What the scheduler needs from a self-play back-end."""

    def open(self) -> None: ...
    def play_round(self, quota: int, learn: Callable[[int], int]) -> list[tuple[int, Any]]: ...
    def publish(self, step: int) -> None: ...
    def active(self) -> int: ...
    def pending(self) -> int: ...
    def close(self) -> None: ...


class ThreadWorkerSource:
    """CPU workers on a thread pool: a round hands over whatever finished, then lets the learner run."""

    steps_per_round = 1

    def __init__(self, manager: Any, buffer: Any):
        self.manager = manager
        self.buffer = buffer

    def open(self) -> None:
        self.manager.submit_initial_tasks()

    def play_round(self, quota: int, learn: Callable[[int], int]) -> list[tuple[int, Any]]:
        # With nothing to learn from yet, waiting for an episode is all there is to do; otherwise only collect.
        done = self.manager.get_completed_tasks(0.0 if quota > 0 else 0.25)
        for worker_id, _ in done:
            self.manager.submit_task(worker_id)
        if quota > 0:
            learn(quota)
        return done

    def publish(self, step: int) -> None:
        self.manager.update_worker_networks(step)

    def active(self) -> int:
        return self.manager.get_num_active_workers()

    def pending(self) -> int:
        return self.manager.get_num_pending_tasks()

    def close(self) -> None:
        self.manager.cleanup_actors()


class TrainingLoop:
    def __init__(self, components: TrainingComponents):
        self.components = components
        self.train_config = components.train_config
        self.buffer = components.buffer
        self.trainer = components.trainer
        self.trieye_actor = components.trieye_actor
        self.trieye_config = components.trieye_config
        self.serializer = components.serializer
        self.dist = components.dist

        self.global_step = 0
        self.episodes_played = 0
        self.total_simulations_run = 0
        self.weight_update_count = 0
        self.rounds = 0
        self.start_time = time.time()
        self.initial_global_step = 0
        self.stop_requested = threading.Event()
        self.training_complete = False
        self.training_exception: Exception | None = None
        self.peer_lost = False
        self.skipped_steps = 0  # data-parallel steps this rank took part in without a batch of its own
        self._announced_ready = False

        self.source: SelfPlaySource = self._build_source()
        persistence = self.trieye_config.persistence
        self.cadences = [
            StepCadence(self.train_config.WORKER_UPDATE_FREQ_STEPS, self._publish_weights, "weight publication"),
            StepCadence(self.train_config.CHECKPOINT_SAVE_FREQ_STEPS, lambda _s: self.save_state(save_buffer=False), "checkpoint"),
            StepCadence(
                persistence.BUFFER_SAVE_FREQ_STEPS if persistence.SAVE_BUFFER else 0,
                lambda _s: self.save_state(save_buffer=True),
                "buffer save",
            ),
            StepCadence(50, self._log_losses, "loss line"),
        ]
        self._last_losses: dict[str, float] = {}
        self.loop_helpers = LoopHelpers(components, self._get_loop_state)
        logger.info("TrainingLoop ready (%s self-play source).", type(self.source).__name__)

    def _build_source(self) -> SelfPlaySource:
        if self.components.gpu_actor is not None:
            from .gpu_manager import GpuMoveSource

            self.worker_manager: Any = None
            return GpuMoveSource(self.components)
        self.worker_manager = WorkerManager(self.components)
        return ThreadWorkerSource(self.worker_manager, self.buffer)

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
            "num_active_workers": self.source.active(),
            "num_pending_tasks": self.source.pending(),
            "start_time": self.start_time,
            "num_workers": self.train_config.NUM_SELF_PLAY_WORKERS,
        }

    def set_initial_state(self, global_step: int, episodes_played: int, total_simulations: int) -> None:
        self.global_step = self.initial_global_step = int(global_step)
        self.episodes_played = int(episodes_played)
        self.total_simulations_run = int(total_simulations)
        freq = self.train_config.WORKER_UPDATE_FREQ_STEPS
        self.weight_update_count = self.global_step // freq if freq > 0 else 0
        self.loop_helpers.reset_rate_counters(global_step, episodes_played, total_simulations)
        logger.info("Resuming at learner step %d (%d episodes, %d simulations, %d weight publications so far).",
                    self.global_step, self.episodes_played, self.total_simulations_run, self.weight_update_count)

    def initialize_workers(self) -> None:
        if self.worker_manager is not None:
            self.worker_manager.initialize_workers()

    def request_stop(self) -> None:
        if not self.stop_requested.is_set():
            logger.info("Stop requested.")
            self.stop_requested.set()

    # ------------------------------------------------------------------ metric plumbing
    def _emit(self, name: str, value: float | int, context: dict | None = None) -> None:
        if not self.trieye_actor:
            return
        try:
            self.trieye_actor.log_event(
                RawMetricEvent(name=name, value=value, global_step=self.global_step, timestamp=time.time(), context=context or {})
            )
        except Exception as exc:
            logger.error("metric '%s' could not be queued: %s", name, exc)

    def _emit_many(self, events: list[RawMetricEvent]) -> None:
        if self.trieye_actor and events:
            try:
                self.trieye_actor.log_batch_events(events)
            except Exception as exc:
                logger.error("metric batch could not be queued: %s", exc)

    def _emit_samples(self, name: str, values: Any) -> None:
        """One metric sample per element (per-game-step metrics of the GPU source, thousands per move)."""
        if self.trieye_actor is None or values is None or len(values) == 0:
            return
        log_array = getattr(self.trieye_actor, "log_array", None)
        if callable(log_array):
            try:
                log_array(name, np.asarray(values, dtype=np.float64), self.global_step)
                return
            except Exception as exc:
                logger.error("metric array '%s' could not be queued: %s", name, exc)
                return
        for v in values:
            self._emit(name, float(v))

    # ------------------------------------------------------------------ deliveries
    def _announce_ready(self) -> None:
        if not self._announced_ready and len(self.buffer) >= self.buffer.min_size_to_train:
            logger.info("Replay holds %d experiences (>= %d): the learner starts.", len(self.buffer), self.buffer.min_size_to_train)
            self._announced_ready = True

    def _take_episode(self, result: SelfPlayResult, worker_id: int) -> None:
        good, bad = self.loop_helpers.validate_experiences(result.episode_experiences)
        if bad:
            logger.warning("worker %d: dropped %d malformed experiences.", worker_id, bad)
        if not good:
            logger.error("worker %d: episode of %d steps (score %.2f) yielded nothing usable.", worker_id, result.episode_steps, result.final_score)
            return
        self.buffer.add_batch(good)
        self._announce_ready()
        self.episodes_played += 1
        self.total_simulations_run += result.total_simulations
        self._emit("Progress/Episodes_Played", self.episodes_played)
        self._emit("Progress/Total_Simulations", self.total_simulations_run)

    def _take_move_report(self, report: Any) -> None:
        """One move of every GPU-resident game: experiences are already in the HBM replay, account for the rest."""
        self._announce_ready()
        self.episodes_played += report.episodes_finished
        self.total_simulations_run += report.simulations
        for ctx in report.episode_contexts:
            self._emit("episode_end", 1.0, {**ctx, "trainer_step": report.trainer_step})
        self._emit("mcts_step", float(report.simulations), {"moves": report.moves})
        per_game = getattr(report, "per_game", None)
        if per_game:  # one sample per game step, like the CPU worker's per-step events
            self._emit_samples("step_reward", per_game.get("step_reward"))
            self._emit_samples("triangles_cleared_step", per_game.get("triangles_cleared_step"))
            self._emit_samples("current_score", per_game.get("current_score"))
        if report.episodes_finished:
            self._emit("Progress/Episodes_Played", self.episodes_played)
        self._emit("Progress/Total_Simulations", self.total_simulations_run)
        for key, ms in (getattr(report, "phase_ms", None) or {}).items():  # --profile: GPU time of the move's phases
            self._emit("Perf/SelfPlay_" + key[:-3].title() + "_ms", float(ms))

    def _account(self, worker_id: int, item: Any) -> None:
        try:
            if isinstance(item, SelfPlayResult):
                self._take_episode(item, worker_id)
            elif isinstance(item, Exception):
                logger.error("worker %d: task failed: %s", worker_id, item)
            elif hasattr(item, "episodes_finished"):
                self._take_move_report(item)
            else:
                logger.error("worker %d delivered an unknown item of type %s", worker_id, type(item).__name__)
        except Exception as exc:
            logger.error("delivery of worker %d could not be processed: %s", worker_id, exc, exc_info=True)

    # ------------------------------------------------------------------ persistence
    def save_state(self, is_best: bool = False, save_buffer: bool = False) -> None:
        """Hand the current model / optimizer / counters (and optionally the replay) to the async checkpoint writer."""
        if not self.trieye_actor:
            logger.error("no stats actor: state at step %d not saved.", self.global_step)
            return
        comps = self.components
        try:
            payload = self.serializer.prepare_buffer_data(self.buffer) if save_buffer else None
            if save_buffer and payload is None:
                logger.error("replay could not be exported; saving the checkpoint without it.")
            self.trieye_actor.save_training_state(
                nn_state_dict=comps.nn.get_weights(),
                optimizer_state_dict=self.serializer.prepare_optimizer_state(self.trainer.optimizer) if self.trainer is not None else {},
                buffer_content=payload.buffer_list if payload else [],
                global_step=self.global_step,
                episodes_played=self.episodes_played,
                total_simulations_run=self.total_simulations_run,
                is_best=is_best,
                save_buffer=bool(save_buffer and payload is not None),
                model_config_dict=comps.model_config.model_dump(),
                env_config_dict=comps.env_config.model_dump(),
                actor_state=comps.gpu_actor.state_dict() if comps.gpu_actor is not None else None,
            )
            logger.info("State save queued at step %d (buffer: %s).", self.global_step, bool(payload))
        except Exception as exc:
            logger.error("state save at step %d failed: %s", self.global_step, exc, exc_info=True)

    # ------------------------------------------------------------------ learner
    def _publish_weights(self, step: int) -> None:
        self.source.publish(step)
        self.weight_update_count += 1
        self._emit("Progress/Weight_Updates_Total", self.weight_update_count)

    def _log_losses(self, step: int) -> None:
        lo = self._last_losses
        logger.info("learner step %d: policy loss %.4f, value loss %.4f, lr %.3g", step, lo.get("policy_loss", 0.0),
                    lo.get("value_loss", 0.0), self.trainer.get_current_lr() if self.trainer is not None else 0.0)

    def _learner_step(self) -> bool:
        """Sample, optimise, re-prioritise, report; advance ``global_step`` and fire the cadences.

        Returns False when no optimizer step was taken.  With data-parallel learners a rank whose own step fails (no
        sample, a batch that cannot be prepared, an exception in forward / backward) still takes part in the step of
        the others with a zero gradient (:meth:`Trainer.skip_step`) and counts it: the ranks agreed on a number of steps
        for this round, and each of those steps is one all-reduce that every rank has to enter."""
        outcome = None
        sample = None
        lockstep = self.dist is not None and self.dist.is_learner and self.dist.num_learners > 1
        syncs_before = int(getattr(self.trainer, "sync_count", 0)) if lockstep else 0
        try:
            sample = self.buffer.sample(self.train_config.BATCH_SIZE, current_train_step=self.global_step)
            if sample:
                outcome = self.trainer.train_step(sample)
        except Exception as exc:
            from ..parallel.dist import PeerLostError

            if not lockstep or isinstance(exc, PeerLostError):
                raise  # a single learner has nobody to keep in step with: the failure ends the run as it always did
            logger.error("learner step %d failed on this rank: %s", self.global_step + 1, exc, exc_info=True)
        if not outcome:
            entered = lockstep and int(getattr(self.trainer, "sync_count", 0)) > syncs_before  # its all-reduce already happened
            if not lockstep or (not entered and not self.trainer.skip_step()):
                if sample:
                    logger.warning("learner step %d produced no result.", self.global_step + 1)
                return False
            # kept in step with the other learners on their gradients alone
            self.skipped_steps += 1
            self.global_step += 1
            logger.warning("learner step %d: no batch on this rank, stepped on the other ranks' gradients.", self.global_step)
            self._emit("Learner/Skipped_Steps", self.skipped_steps)
            for cadence in self.cadences:
                cadence.tick(self.global_step)
            return True
        losses, td_errors = outcome
        self.global_step += 1
        if self.global_step == self.initial_global_step + 1:
            logger.info("First learner step of this run done (step %d).", self.global_step)
        self._emit("step_completed", 1.0)
        if self.train_config.USE_PER:
            self.buffer.update_priorities(sample.indices if isinstance(sample, TensorBatch) else sample["indices"], td_errors)
            self._emit("PER/Beta", self.buffer._calculate_beta(self.global_step))
        events = [RawMetricEvent(name=metric, value=losses[key], global_step=self.global_step) for key, metric in LOSS_METRICS if key in losses]
        events.append(RawMetricEvent(name="LearningRate", value=self.trainer.get_current_lr(), global_step=self.global_step))
        self._emit_many(events)
        self._last_losses = losses
        for cadence in self.cadences:
            cadence.tick(self.global_step)
        return True

    def _learn(self, quota: int) -> int:
        """Up to ``quota`` learner steps (the source calls this at the point of the round where they overlap best)."""
        done = 0
        while done < quota and self._learner_step():
            done += 1
        return done

    # ------------------------------------------------------------------ round negotiation
    def _steps_wanted(self) -> int:
        if self.trainer is None:
            return 0
        have = len(self.buffer)
        if have < max(self.buffer.min_size_to_train, self.train_config.BATCH_SIZE) or not self.buffer.is_ready():
            return 0
        per_round = int(getattr(self.source, "steps_per_round", 1))
        target = self.train_config.MAX_TRAINING_STEPS
        return per_round if target is None else max(0, min(per_round, target - self.global_step))

    def _negotiate(self) -> RoundQuota:
        steps = self._steps_wanted()
        stop = self.stop_requested.is_set()
        ctx = self.dist
        if ctx is not None and ctx.world_size > 1 and ctx.layout.colocated:
            # every rank takes the same number of optimizer steps, or none; a stop anywhere is a stop everywhere
            agreed = ctx.agree([float(steps), -1.0 if stop else 0.0], "min")
            steps, stop = int(agreed[0]), agreed[1] < 0
        return RoundQuota(steps=steps, stop=stop)

    def _round_metrics(self) -> None:
        self.loop_helpers.log_progress_eta()
        self._emit("Buffer/Size", len(self.buffer))
        self._emit("System/Num_Active_Workers", self.source.active())
        self._emit("System/Num_Pending_Tasks", self.source.pending())
        if self.trieye_actor:
            self.trieye_actor.process_and_log(self.global_step)

    def _sigint_requests_stop(self):  # noqa: ANN202
        """While the loop runs, the first Ctrl-C is a stop request instead of an exception.

        The round in progress finishes and the stop travels through the next negotiation, so the ranks of a multi-GPU
        job leave together (an exception would take this rank out between two collectives and leave the others waiting
        for it until their time-out).  A second Ctrl-C raises KeyboardInterrupt as usual.  Returns the undo function;
        outside the main thread (tests, embedded use) nothing is installed.
        """
        import signal

        if threading.current_thread() is not threading.main_thread():
            return lambda: None
        previous = signal.getsignal(signal.SIGINT)

        def on_sigint(signum, frame):  # noqa: ANN001, ARG001
            if self.stop_requested.is_set():
                raise KeyboardInterrupt
            logger.warning("Interrupt received: stopping after this round (interrupt again to abort).")
            self.request_stop()

        signal.signal(signal.SIGINT, on_sigint)
        return lambda: signal.signal(signal.SIGINT, previous)

    # ------------------------------------------------------------------ main loop
    def run(self) -> None:
        from ..parallel.dist import PeerLostError

        target = self.train_config.MAX_TRAINING_STEPS
        logger.info("Training starts: %s.", f"{target} learner steps" if target else "no step limit")
        self.start_time = time.time()
        restore_sigint = self._sigint_requests_stop()
        try:
            self.source.open()
            while True:
                if target is not None and self.global_step >= target:
                    logger.info("Learner step budget reached (%d).", target)
                    self.training_complete = True
                    self.request_stop()
                quota = self._negotiate()
                if quota.stop:
                    break
                for worker_id, item in self.source.play_round(quota.steps, self._learn):
                    self._account(worker_id, item)
                self.rounds += 1
                self._round_metrics()
        except KeyboardInterrupt:
            logger.warning("Interrupted; stopping after this round.")
            self.request_stop()
        except PeerLostError as exc:
            # the other ranks cannot be reached any more: no further collective is entered, the runner saves and exits 1
            logger.critical("A peer rank was lost (%s); this rank saves its state and leaves.", exc)
            self.peer_lost = True
            self.training_exception = exc
            self.request_stop()
        except Exception as exc:
            logger.critical("The training loop failed: %s", exc, exc_info=True)
            self.training_exception = exc
            self.request_stop()
        finally:
            restore_sigint()
            if self.training_exception is not None or not (target is not None and self.global_step >= target):
                self.training_complete = False
            logger.info("Training loop ended after %d rounds (complete: %s, failed: %s).", self.rounds, self.training_complete,
                        self.training_exception is not None)

    def cleanup_actors(self) -> None:
        self.source.close()
