"""Asynchronous actor / learner roles for the split layout (actor GPUs + learner GPUs).

The reference has one learner process and a pool of Ray CPU workers that exchange pickled
objects; its learner blocks on every weight push (alphatriangle/training/loop.py:298-416,
alphatriangle/training/worker_manager.py:169-209).  On one MI355X node the roles are ranks
of one RCCL job and neither side ever waits for the other's compute:

  actor rank    loop: [stop?] -> [newer weights published? request them, receive, re-pack] ->
                play one move of every game -> announce + send this move's records.
  learner rank  loop: take every announced shipment into the HBM replay -> serve weight
                requests -> a few optimizer steps -> publish (version bump in the control
                store) every ``WORKER_UPDATE_FREQ_STEPS`` steps.

Nothing here is a collective over actors and learners together.  Every transfer is a
point-to-point send / receive on a communicator of its own (one per pair and direction), and
it is only posted after both sides have said so in the control store, so a transfer never
hangs on a peer that is busy, and a peer that died is noticed by its silence: the learner
retires the actor and trains on with the others (alphatriangle/training/worker_manager.py:153-159
removes a dead worker the same way); an actor whose learner is gone leaves with an error.
Several learner ranks keep their optimizer steps in lock-step through a host-side agreement
and all-reduce their flat gradients.
"""

from __future__ import annotations

import logging
import time
from typing import Any, Protocol

import numpy as np
import torch

from ..parallel.dist import PeerLostError
from ..parallel.experience import ExperienceInbox, ExperienceOutbox, RecordLayout, assigned_learner
from ..parallel.weights import WeightWire
from .components import TrainingComponents
from .loop import RoundQuota, TrainingLoop

logger = logging.getLogger(__name__)


class ExperienceSource(Protocol):
    """What an actor rank runs per move."""

    def produce(self) -> tuple[torch.Tensor, dict[str, float]]:
        """Advance self-play; returns ``(records [n, stride] fp32, counters)``."""

    def sync_weights(self, trainer_step: int) -> None: ...


class GpuExperienceSource:
    """GPU-resident actor whose commit kernel writes into a small HBM outbox instead of a replay."""

    def __init__(self, gpu_actor: Any, outbox: Any, layout: RecordLayout):
        self.actor = gpu_actor
        self.outbox = outbox
        self.layout = layout
        self.last_report: Any = None

    def produce(self) -> tuple[torch.Tensor, dict[str, float]]:
        report = self.actor.play_move()
        self.last_report = report
        n = min(int(report.experiences_added), self.outbox.emit_capacity)
        if n > 0:
            idx = self.outbox.emitted_idx[:n].long()
            rec = self.layout.pack(self.outbox.grid[idx], self.outbox.other[idx], self.outbox.policy[idx], self.outbox.ret[idx])
        else:
            rec = torch.empty(0, self.layout.stride, dtype=torch.float32, device=self.outbox.device)
        return rec, {"episodes": report.episodes_finished, "simulations": report.simulations, "moves": report.moves}

    def sync_weights(self, trainer_step: int) -> None:
        self.actor.sync_weights(trainer_step)


class HostExperienceSource:
    """CPU actor: one host-search episode per call (tests, CPU-only rehearsals of the split layout)."""

    def __init__(self, worker: Any, layout: RecordLayout, action_dim: int):
        self.worker = worker
        self.layout = layout
        self.action_dim = action_dim

    def produce(self) -> tuple[torch.Tensor, dict[str, float]]:
        result = self.worker.run_episode()
        exps = result.episode_experiences
        counters = {"episodes": 1, "simulations": result.total_simulations, "moves": result.episode_steps}
        if not exps:
            return torch.empty(0, self.layout.stride), counters
        grid = torch.from_numpy(np.stack([np.asarray(e[0]["grid"], dtype=np.float32).reshape(-1) for e in exps]))
        other = torch.from_numpy(np.stack([np.asarray(e[0]["other_features"], dtype=np.float32) for e in exps]))
        policy = torch.zeros(len(exps), self.action_dim)
        for i, e in enumerate(exps):
            for a, p in e[1].items():
                if 0 <= a < self.action_dim:
                    policy[i, a] = p
        ret = torch.tensor([float(e[2]) for e in exps])
        return self.layout.pack(grid, other, policy, ret), counters

    def sync_weights(self, trainer_step: int) -> None:
        self.worker.set_current_trainer_step(trainer_step)


class _RemoteActors:
    """Scheduler-facing view of the actor ranks that feed this learner."""

    steps_per_round = 1

    def __init__(self, loop: "SplitRoleLoop"):
        self.loop = loop

    def open(self) -> None: ...
    def play_round(self, quota: int, learn: Any) -> list:  # pragma: no cover - the split loop has its own run()
        return []

    def publish(self, step: int) -> None:
        self.loop._announce_weights(step)

    def active(self) -> int:
        return self.loop.inbox.active_actors() if self.loop.inbox is not None else 0

    def pending(self) -> int:
        return 0

    def close(self) -> None: ...


class SplitRoleLoop(TrainingLoop):
    def __init__(self, components: TrainingComponents, source: ExperienceSource | None):
        ctx = components.dist
        if ctx is None or ctx.layout.colocated:
            raise ValueError("SplitRoleLoop needs a DistContext with the split role layout")
        if ctx.control is None:
            raise ValueError("SplitRoleLoop needs the control store of the process group")
        self.ctx = ctx
        self.producer = source
        self.inbox: ExperienceInbox | None = None
        super().__init__(components)
        self.sys = components.system_config
        env, mc = components.env_config, components.model_config
        self.layout = RecordLayout(mc.GRID_INPUT_CHANNELS * env.ROWS * env.COLS, mc.OTHER_NN_INPUT_FEATURES_DIM, env.ACTION_DIM)
        dev = components.nn.device
        self.wire = WeightWire(components.nn._orig_model)
        self.lead = ctx.layout.learner_ranks[0]
        self.moves_played = 0
        self.rows_ingested = 0
        self.busy_s = 0.0            # learner: time spent in optimizer steps (the rest is polling / ingest / idle)
        self._seen_version = 0       # actor: last weight version received; lead learner: last version announced
        self._served: dict[int, int] = {}
        self.outbox: ExperienceOutbox | None = None
        if ctx.is_actor:
            if source is None:
                raise ValueError("actor ranks need an ExperienceSource")
            max_rows = int(self.sys.GAMES_PER_GPU) * (int(self.train_config.N_STEP_RETURNS) + 1)
            self.outbox = ExperienceOutbox(ctx, self.layout, dev, max(max_rows, int(self.sys.SHIP_CAPACITY)))
        else:
            self.inbox = ExperienceInbox(ctx, self.layout, dev, silence_s=float(self.sys.ACTOR_SILENCE_S))

    def _build_source(self) -> Any:
        self.worker_manager = None
        return _RemoteActors(self)

    # ------------------------------------------------------------------ weights: lead learner side
    def _announce_weights(self, step: int) -> None:
        """Snapshot the parameters into the wire buffer and bump the published version (no transfer yet)."""
        if self.ctx.rank != self.lead:
            return
        self.wire.pack()
        self._seen_version += 1
        cp = self.ctx.control
        cp.put("weights/step", step)
        cp.put("weights/version", self._seen_version)

    def _serve_weight_requests(self) -> int:
        """Send the current snapshot to every actor that asked for a version it does not have yet."""
        if self.ctx.rank != self.lead or self._seen_version == 0:
            return 0
        cp = self.ctx.control
        sent = 0
        for a in self.ctx.layout.actor_ranks:
            want = int(cp.peek(f"weights/want/{a}", 0) or 0)
            if want > self._served.get(a, 0):
                learner = assigned_learner(self.ctx, a)
                if learner != self.lead:
                    continue  # pairs exist only between an actor and its own learner; that learner relays (below)
                try:
                    self.ctx.send_to(self.wire.buffer, a, learner=self.lead, actor=a, weights=True)
                except PeerLostError:
                    logger.error("actor rank %d did not take the weights it asked for; skipping it.", a)
                self._served[a] = want
                sent += 1
        return sent

    def _relay_weight_requests(self) -> int:
        """Non-lead learners hold the same parameters (lock-step data parallel) and serve their own actors."""
        if self.ctx.rank == self.lead or not self.ctx.is_learner:
            return 0
        cp = self.ctx.control
        version = int(cp.peek("weights/version", 0) or 0)
        sent = 0
        for feed in (self.inbox.feeds if self.inbox else []):
            want = int(cp.peek(f"weights/want/{feed.rank}", 0) or 0)
            if want > self._served.get(feed.rank, 0) and want <= version:
                self.wire.pack()
                try:
                    self.ctx.send_to(self.wire.buffer, feed.rank, learner=self.ctx.rank, actor=feed.rank, weights=True)
                except PeerLostError:
                    logger.error("actor rank %d did not take the weights it asked for; skipping it.", feed.rank)
                self._served[feed.rank] = want
                sent += 1
        return sent

    # ------------------------------------------------------------------ weights: actor side
    def _take_new_weights(self) -> bool:
        cp = self.ctx.control
        version = int(cp.peek("weights/version", 0) or 0)
        if version <= self._seen_version:
            return False
        me = self.ctx.rank
        learner = assigned_learner(self.ctx, me)
        cp.put(f"weights/want/{me}", version)
        self.ctx.recv_from(self.wire.buffer, learner, learner=learner, actor=me, weights=True)
        self.wire.unpack()
        step = int(cp.peek("weights/step", 0) or 0)
        assert self.producer is not None
        self.producer.sync_weights(step)  # re-folds / re-packs the evaluator's weights in place
        self._seen_version = version
        self.weight_update_count += 1
        return True

    # ------------------------------------------------------------------ learner
    def _ingest(self) -> int:
        assert self.inbox is not None
        rows = 0
        for ship in self.inbox.poll():
            c = ship.counters
            self.episodes_played += c["episodes"]
            self.total_simulations_run += c["simulations"]
            self.moves_played += c["moves"]
            n = int(ship.records.shape[0])
            if n == 0:
                continue
            grid, other, policy, ret = self.layout.unpack(ship.records)
            if hasattr(self.buffer, "add_arrays"):
                self.buffer.add_arrays(grid, other, policy, ret)
            else:
                from ..stats.serializer import experiences_from_arrays

                mc, env = self.components.model_config, self.components.env_config
                g = grid.cpu().numpy().reshape(-1, mc.GRID_INPUT_CHANNELS, env.ROWS, env.COLS)
                self.buffer.add_batch(experiences_from_arrays(g, other.cpu().numpy(), policy.cpu().numpy(), ret.cpu().numpy()))
            rows += n
        if rows:
            self.rows_ingested += rows
            self._announce_ready()
        return rows

    def _steps_wanted(self) -> int:
        want = super()._steps_wanted()
        if want <= 0:
            return 0
        want = min(int(self.sys.TRAIN_STEPS_PER_ROUND), max(1, (self.train_config.MAX_TRAINING_STEPS or 10**12) - self.global_step))
        reuse = getattr(self.sys, "MAX_SAMPLE_REUSE", None)
        if reuse is not None:
            # replay-ratio control: never consume more than `reuse` samples per experience that has arrived
            used = (self.global_step - self.initial_global_step) * self.train_config.BATCH_SIZE
            room = reuse * max(self.rows_ingested, 1) - used
            want = max(0, min(want, int(room // self.train_config.BATCH_SIZE)))
        return want

    def _negotiate(self) -> RoundQuota:
        steps = self._steps_wanted()
        stop = self.stop_requested.is_set()
        if self.ctx.num_learners > 1:
            agreed = self.ctx.agree([float(steps), -1.0 if stop else 0.0], "min", group="learners")
            steps, stop = int(agreed[0]), agreed[1] < 0
        return RoundQuota(steps=steps, stop=stop)

    def _run_learner(self) -> None:
        ctx, cp = self.ctx, self.ctx.control
        target = self.train_config.MAX_TRAINING_STEPS
        assert self.inbox is not None
        if ctx.rank == self.lead:
            self._announce_weights(self.global_step)  # version 1: what the actors start from
        failed = False
        try:
            while True:
                if target is not None and self.global_step >= target:
                    self.training_complete = True
                    self.request_stop()
                got = self._ingest()
                served = self._serve_weight_requests() + self._relay_weight_requests()
                if self.inbox.feeds and all(f.retired for f in self.inbox.feeds):
                    raise PeerLostError("every actor rank that feeds this learner has gone silent")
                quota = self._negotiate()
                if quota.stop:
                    break
                if quota.steps > 0:
                    t0 = time.perf_counter()
                    self._learn(quota.steps)
                    self.busy_s += time.perf_counter() - t0
                elif not got and not served:
                    time.sleep(0.005)  # nothing to learn from yet and nothing arrived: do not spin on the store
                self.rounds += 1
                if ctx.rank == self.lead:
                    self._emit("Progress/Episodes_Played", self.episodes_played)
                    self._emit("Progress/Total_Simulations", self.total_simulations_run)
                self._round_metrics()
        except BaseException:
            failed = True
            raise
        finally:
            if ctx.rank == self.lead:
                cp.put("stop", 2 if failed else 1)
            self._drain(float(self.sys.CONTROL_TIMEOUT_S))

    def _drain(self, patience_s: float) -> None:
        """After the stop flag: keep matching what actors already announced until each has said it is done."""
        assert self.inbox is not None
        deadline = time.monotonic() + patience_s
        try:
            while not self.inbox.drained() and time.monotonic() < deadline:
                if not self._ingest() and not (self._serve_weight_requests() + self._relay_weight_requests()):
                    time.sleep(0.01)
            self._serve_weight_requests()
            self._relay_weight_requests()
        except PeerLostError as exc:
            logger.error("drain ended early: %s", exc)
        if not self.inbox.drained():
            logger.error("some actor ranks never confirmed the stop within %.0f s.", patience_s)

    # ------------------------------------------------------------------ actor
    def _run_actor(self) -> None:
        ctx, cp = self.ctx, self.ctx.control
        assert self.producer is not None and self.outbox is not None
        start_deadline = time.monotonic() + float(self.sys.CONTROL_TIMEOUT_S)
        while int(cp.peek("weights/version", 0) or 0) == 0:  # play with the learner's parameters from the first move on
            if cp.peek("stop", 0) or time.monotonic() > start_deadline:
                break
            time.sleep(0.01)
        try:
            while not self.stop_requested.is_set():
                if cp.peek("stop", 0):
                    break
                self._take_new_weights()
                records, counters = self.producer.produce()
                report = getattr(self.producer, "last_report", None)
                if report is not None:  # per-episode and per-game-step metrics go to this rank's own metric stream
                    self._take_move_report(report)
                else:
                    self.episodes_played += int(counters.get("episodes", 0))
                    self.total_simulations_run += int(counters.get("simulations", 0))
                self.moves_played += int(counters.get("moves", 0))
                self.outbox.ship(records, counters)
                self.rounds += 1
                if self.trieye_actor:
                    self.trieye_actor.process_and_log(self.global_step)
        finally:
            self.outbox.close()
        self.training_complete = int(cp.peek("stop", 0) or 0) == 1

    def _write_summary(self, wall: float) -> None:
        """``role_summary.json`` in this rank's run directory: what the role did and how busy the learner was."""
        import json
        from pathlib import Path

        ctx = self.ctx
        summary = {
            "rank": ctx.rank, "role": "learner" if ctx.is_learner else "actor", "wall_seconds": round(wall, 3), "rounds": self.rounds,
            "learner_steps": self.global_step - self.initial_global_step, "learner_busy_fraction": round(self.busy_s / wall, 4),
            "rows_ingested": self.rows_ingested, "moves_played": self.moves_played, "weight_versions": self.weight_update_count,
            "actors_retired": [f.rank for f in self.inbox.feeds if f.retired] if self.inbox is not None else [],
            "complete": bool(self.training_complete),
        }
        try:
            if self.trieye_actor:
                Path(self.trieye_actor.get_run_base_dir_str(), "role_summary.json").write_text(json.dumps(summary, indent=2))
        except Exception as exc:  # pragma: no cover
            logger.warning("role summary not written: %s", exc)

    # ------------------------------------------------------------------ main loop
    def run(self) -> None:
        ctx = self.ctx
        logger.info("Split layout: rank %d is %s; %d learner(s), %d actor(s).", ctx.rank, "learner" if ctx.is_learner else "actor",
                    len(ctx.layout.learner_ranks), len(ctx.layout.actor_ranks))
        self.start_time = time.time()
        restore_sigint = self._sigint_requests_stop()  # first Ctrl-C: the role loops see stop_requested and leave in order
        try:
            if ctx.is_learner:
                self._run_learner()
            else:
                self._run_actor()
        except KeyboardInterrupt:
            logger.warning("Interrupted on rank %d.", ctx.rank)
            self.request_stop()
            try:
                ctx.control.put("stop", 1)
            except Exception:  # pragma: no cover
                pass
        except PeerLostError as exc:
            logger.critical("rank %d lost its peers: %s", ctx.rank, exc)
            self.peer_lost = True
            self.training_exception = exc
        except Exception as exc:
            logger.critical("Split-layout loop failed on rank %d: %s", ctx.rank, exc, exc_info=True)
            self.training_exception = exc
            try:
                ctx.control.put("stop", 2)
            except Exception:  # pragma: no cover
                pass
        finally:
            restore_sigint()
            if self.training_exception is not None:
                self.training_complete = False
            wall = max(time.time() - self.start_time, 1e-9)
            self._write_summary(wall)
            if ctx.is_learner:
                logger.info("Learner rank %d: %d steps in %d rounds, busy %.0f %% of %.1f s, %d rows ingested.", ctx.rank,
                            self.global_step - self.initial_global_step, self.rounds, 100.0 * self.busy_s / wall, wall, self.rows_ingested)
            else:
                logger.info("Actor rank %d: %d moves in %d rounds, %d weight versions taken.", ctx.rank, self.moves_played, self.rounds,
                            self.weight_update_count)
