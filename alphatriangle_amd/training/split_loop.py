"""Asynchronous actor / learner roles for the split layout (actor GPUs + learner GPUs).

The reference has one learner process and a pool of Ray CPU workers that exchange pickled
objects; its learner blocks on every weight push (alphatriangle/training/loop.py:298-416,
alphatriangle/training/worker_manager.py:169-209).  On one MI355X node the roles are ranks
of one RCCL job and neither side ever waits for the other's compute:

  actor rank    loop: [stop?] -> [weights asked for a move ago: take them, re-pack] -> [newer
                version published? ask for it] -> play one move of every game -> announce +
                send this move's records.
  learner rank  loop: serve every announced operation (shipments into the HBM replay, weight
                requests from the published snapshot) -> a few optimizer steps -> publish
                (version bump in the control store) every ``WORKER_UPDATE_FREQ_STEPS`` steps.

Nothing here is a collective over actors and learners together.  Every transfer is a
point-to-point send / receive on the one communicator of its (learner, actor) pair, initiated by
the actor and announced in the control store before either side posts it (parallel/experience.py),
so a transfer never hangs on a peer that is busy.  An actor asks for a newly published weight
version at the start of a move and takes it at the start of the next one: the delivery happens
while it searches.  A peer that died is noticed by its heartbeat or by a transfer that misses
its deadline: the learner retires the actor - at the cost of that pair's communicator at most -
and trains on with the others (alphatriangle/training/worker_manager.py:153-159 removes a dead
worker the same way); an actor whose learner is gone leaves with an error.
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
        self._packed_version = 0     # learners: the version the wire buffer holds
        self.version_lag_sum = 0     # actor: published version minus the version played with, summed over moves
        self.outbox: ExperienceOutbox | None = None
        if ctx.is_actor:
            if source is None:
                raise ValueError("actor ranks need an ExperienceSource")
            max_rows = int(self.sys.GAMES_PER_GPU) * (int(self.train_config.N_STEP_RETURNS) + 1)
            self.outbox = ExperienceOutbox(ctx, self.layout, dev, max(max_rows, int(self.sys.SHIP_CAPACITY)),
                                           heartbeat_period_s=float(self.sys.HEARTBEAT_PERIOD_S))
        else:
            self.inbox = ExperienceInbox(ctx, self.layout, dev, silence_s=float(self.sys.ACTOR_SILENCE_S),
                                         heartbeat_stale_s=float(self.sys.HEARTBEAT_STALE_S), weights_provider=self._weights_for)

    def _build_source(self) -> Any:
        self.worker_manager = None
        return _RemoteActors(self)

    # ------------------------------------------------------------------ weights: learner side
    def _announce_weights(self, step: int) -> None:
        """Snapshot the parameters into the wire buffer and bump the published version (no transfer yet)."""
        if self.ctx.rank != self.lead:
            return
        self.wire.pack()
        self._seen_version += 1
        self._packed_version = self._seen_version
        cp = self.ctx.control
        cp.put("weights/step", step)
        cp.put("weights/version", self._seen_version)

    def _weights_for(self, wanted: int) -> torch.Tensor | None:
        """The buffer that answers an actor's request for version ``wanted`` (``None``: not published yet).

        The lead learner serves the snapshot it took when it published.  Every other learner holds the same parameters
        (lock-step data parallel) and snapshots its own copy the first time a published version is asked for."""
        if self.ctx.rank == self.lead:
            return self.wire.buffer if 0 < wanted <= self._seen_version else None
        version = int(self.ctx.control.peek("weights/version", 0) or 0)
        if wanted > version or version == 0:
            return None
        if self._packed_version < version:
            self.wire.pack()
            self._packed_version = version
        return self.wire.buffer

    # ------------------------------------------------------------------ weights: actor side
    def _apply_requested_weights(self) -> bool:
        """Take the version asked for earlier (served while the last move was searched) and re-pack the evaluator."""
        assert self.outbox is not None and self.producer is not None
        if not self.outbox.weights_pending():
            return False
        version = self.outbox.take_weights()
        self.wire.unpack()
        step = int(self.ctx.control.peek("weights/step", 0) or 0)
        self.producer.sync_weights(step)  # re-folds / re-packs the evaluator's weights in place
        self._seen_version = version
        self.weight_update_count += 1
        return True

    def _request_new_weights(self) -> bool:
        """Ask for the published version if it is newer than what this actor has or has already asked for."""
        assert self.outbox is not None
        version = int(self.ctx.control.peek("weights/version", 0) or 0)
        if version <= self._seen_version or self.outbox.weights_pending():
            return False
        self.outbox.request_weights(version, self.wire.buffer)
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
                served_before = self.inbox.weights_served
                got = self._ingest()
                served = self.inbox.weights_served - served_before
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
                if not self._ingest():
                    time.sleep(0.01)
            self.inbox.finish_sends()
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
            if self._request_new_weights():  # the first version is waited for: every actor starts from the learner's parameters
                self._apply_requested_weights()
            while not self.stop_requested.is_set():
                if cp.peek("stop", 0):
                    break
                self._apply_requested_weights()  # asked for at the start of the previous move: served meanwhile
                self._request_new_weights()      # arrives under this move's search, used from the next move on
                self.version_lag_sum += max(0, int(cp.peek("weights/version", 0) or 0) - self._seen_version)
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
            "mean_weight_version_lag": round(self.version_lag_sum / max(self.rounds, 1), 3) if ctx.is_actor else None,
            "weights_served": self.inbox.weights_served if self.inbox is not None else None,
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
