"""Lock-step training loop for the split role layout (actor GPUs + learner GPUs).

The reference has one learner process and a pool of Ray CPU workers that exchange pickled
objects (alphatriangle/training/loop.py:298-416, alphatriangle/training/worker_manager.py).
On one MI355X node the same roles become ranks of one RCCL job.  Every round, on every rank:

  1. work      actors advance all their games by one move (experiences land in a local HBM
               outbox); learners run ``TRAIN_STEPS_PER_ROUND`` optimizer steps on their replay
               (gradients all-reduced over the learner group when there are several);
  2. ship      ONE gather per learner of fixed-size record blocks from its actors, inserted
               into the learner's HBM replay at max priority;
  3. control   two tiny all-reduces carry the stop / publish decisions of the first learner and
               the self-play counters of the actors;
  4. publish   when due (every ``WORKER_UPDATE_FREQ_STEPS`` learner steps, and in round 0) ONE
               broadcast of the flat weight wire from the first learner; actors re-fold and
               re-pack their inference engine in place.

Collectives are issued in the same order by every rank, so the loop cannot deadlock on a
role-dependent branch; all role-dependent work sits in phase 1.
"""

from __future__ import annotations

import logging
import time
from typing import Any, Protocol

import numpy as np
import torch
import torch.distributed

from ..parallel.experience import ExperienceShipper, RecordLayout
from ..parallel.weights import WeightWire
from .components import TrainingComponents
from .loop import TrainingLoop

logger = logging.getLogger(__name__)


class ExperienceSource(Protocol):
    """What an actor rank runs in phase 1."""

    def produce(self) -> tuple[torch.Tensor, dict[str, float]]:
        """Advance self-play; returns ``(records [n, stride] fp32, counters)``."""

    def sync_weights(self, trainer_step: int) -> None: ...


class GpuExperienceSource:
    """GPU-resident actor whose commit kernel writes into a small HBM outbox instead of a replay."""

    def __init__(self, gpu_actor: Any, outbox: Any, layout: RecordLayout):
        self.actor = gpu_actor
        self.outbox = outbox
        self.layout = layout

    def produce(self) -> tuple[torch.Tensor, dict[str, float]]:
        report = self.actor.play_move()
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
    """CPU actor: one host-search episode per round (tests, CPU-only rehearsals of the split layout)."""

    def __init__(self, worker: Any, layout: RecordLayout, action_dim: int):
        self.worker = worker
        self.layout = layout
        self.action_dim = action_dim

    def produce(self) -> tuple[torch.Tensor, dict[str, float]]:
        result = self.worker.run_episode()
        exps = result.episode_experiences
        if not exps:
            return torch.empty(0, self.layout.stride), {"episodes": 1, "simulations": result.total_simulations, "moves": result.episode_steps}
        grid = torch.from_numpy(np.stack([np.asarray(e[0]["grid"], dtype=np.float32).reshape(-1) for e in exps]))
        other = torch.from_numpy(np.stack([np.asarray(e[0]["other_features"], dtype=np.float32) for e in exps]))
        policy = torch.zeros(len(exps), self.action_dim)
        for i, e in enumerate(exps):
            for a, p in e[1].items():
                if 0 <= a < self.action_dim:
                    policy[i, a] = p
        ret = torch.tensor([float(e[2]) for e in exps])
        counters = {"episodes": 1, "simulations": result.total_simulations, "moves": result.episode_steps}
        return self.layout.pack(grid, other, policy, ret), counters

    def sync_weights(self, trainer_step: int) -> None:
        self.worker.set_current_trainer_step(trainer_step)


class _NoWorkers:
    """The split loop publishes weights itself; this keeps TrainingLoop's bookkeeping calls harmless."""

    def __init__(self, n_actors: int):
        self.n = n_actors

    def initialize_workers(self) -> None: ...
    def submit_initial_tasks(self) -> None: ...
    def submit_task(self, worker_idx: int) -> None: ...  # noqa: ARG002
    def get_completed_tasks(self, timeout: float = 0.0) -> list:  # noqa: ARG002
        return []

    def update_worker_networks(self, global_step: int) -> None: ...  # noqa: ARG002
    def get_num_active_workers(self) -> int:
        return self.n

    def get_num_pending_tasks(self) -> int:
        return 0

    def cleanup_actors(self) -> None: ...


class SplitRoleLoop(TrainingLoop):
    def __init__(self, components: TrainingComponents, source: ExperienceSource | None):
        ctx = components.dist
        if ctx is None or ctx.layout.colocated:
            raise ValueError("SplitRoleLoop needs a DistContext with the split role layout")
        self.ctx = ctx
        self.source = source
        super().__init__(components)
        self.worker_manager = _NoWorkers(len(ctx.layout.actor_ranks))
        self.sys = components.system_config
        env, mc = components.env_config, components.model_config
        self.layout = RecordLayout(mc.GRID_INPUT_CHANNELS * env.ROWS * env.COLS, mc.OTHER_NN_INPUT_FEATURES_DIM, env.ACTION_DIM)
        dev = components.nn.device
        self.shipper = ExperienceShipper(ctx, self.layout, int(self.sys.SHIP_CAPACITY), dev)
        self.wire = WeightWire(components.nn._orig_model)
        self.ctl_device = torch.device("cpu") if ctx.backend == "gloo" else dev
        self.rounds = 0
        self.moves_played = 0
        self._published_count = -1
        if ctx.is_actor and source is None:
            raise ValueError("actor ranks need an ExperienceSource")

    # ------------------------------------------------------------------ phases
    def _learner_work(self) -> None:
        ready = 1.0 if self.buffer.is_ready() else 0.0
        if self.ctx.num_learners > 1:  # every learner must take the same number of optimizer steps
            t = torch.tensor([ready], dtype=torch.float32, device=self.ctl_device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=self.ctx.learner_group)
            ready = float(t.item())
        if ready < 1.0:
            return
        target = self.train_config.MAX_TRAINING_STEPS
        for _ in range(int(self.sys.TRAIN_STEPS_PER_ROUND)):
            if target is not None and self.global_step >= target:
                break
            if not self._run_training_step():
                break
            ckpt_freq = self.train_config.CHECKPOINT_SAVE_FREQ_STEPS
            if self.ctx.rank == self.ctx.layout.learner_ranks[0] and ckpt_freq > 0 and self.global_step % ckpt_freq == 0:
                self._trigger_save_state(is_best=False, save_buffer=False)

    def _ingest(self, incoming: torch.Tensor) -> None:
        if incoming.shape[0] == 0:
            return
        grid, other, policy, ret = self.layout.unpack(incoming)
        if hasattr(self.buffer, "add_arrays"):
            self.buffer.add_arrays(grid, other, policy, ret)
        else:
            from ..stats.serializer import experiences_from_arrays

            mc, env = self.components.model_config, self.components.env_config
            g = grid.cpu().numpy().reshape(-1, mc.GRID_INPUT_CHANNELS, env.ROWS, env.COLS)
            self.buffer.add_batch(experiences_from_arrays(g, other.cpu().numpy(), policy.cpu().numpy(), ret.cpu().numpy()))
        self._note_buffer_ready()

    def _publish(self, step: int) -> None:
        src = self.ctx.layout.learner_ranks[0]
        if self.ctx.rank == src:
            self.wire.pack()
        self.ctx.broadcast_weights(self.wire.buffer, src)
        if self.ctx.rank != src:
            self.wire.unpack()
        if self.ctx.is_actor:
            assert self.source is not None
            self.components.nn.sync_inference_weights()
            self.source.sync_weights(step)

    # ------------------------------------------------------------------ main loop
    def run(self) -> None:
        ctx = self.ctx
        lead = ctx.layout.learner_ranks[0]
        target = self.train_config.MAX_TRAINING_STEPS
        freq = max(1, int(self.train_config.WORKER_UPDATE_FREQ_STEPS))
        logger.info("SplitRoleLoop: rank %d is %s; %d learner(s), %d actor(s).", ctx.rank, "learner" if ctx.is_learner else "actor",
                    len(ctx.layout.learner_ranks), len(ctx.layout.actor_ranks))
        self.start_time = time.time()
        try:
            while True:
                counters = {"episodes": 0.0, "simulations": 0.0, "moves": 0.0}
                records = None
                try:
                    if ctx.is_actor:
                        assert self.source is not None
                        records, counters = self.source.produce()
                    else:
                        self._learner_work()
                except KeyboardInterrupt:
                    # keep going to the control exchange so every rank learns about the stop in the same round
                    logger.warning("KeyboardInterrupt on rank %d; stopping all ranks at this round's control exchange.", ctx.rank)
                    self.request_stop()
                incoming = self.shipper.exchange(records)
                if ctx.is_learner:
                    self._ingest(incoming)
                # ---- control: MAX carries decisions, SUM carries counters
                done = target is not None and self.global_step >= target
                due = self.global_step // freq
                mx = torch.zeros(3, dtype=torch.float64, device=self.ctl_device)
                mx[0] = 1.0 if (self.stop_requested.is_set() or (ctx.rank == lead and done)) else 0.0
                if ctx.rank == lead:
                    mx[1] = 1.0 if due != self._published_count else 0.0
                    mx[2] = float(self.global_step)
                ctx.all_reduce_max_(mx)
                sm = torch.tensor([counters["episodes"], counters["simulations"], counters["moves"]], dtype=torch.float64, device=self.ctl_device)
                ctx._collective(torch.distributed.all_reduce, sm, op=torch.distributed.ReduceOp.SUM, group=ctx.world_group)
                stop, publish, lead_step = bool(mx[0].item() > 0), bool(mx[1].item() > 0), int(mx[2].item())
                self.episodes_played += int(sm[0].item())
                self.total_simulations_run += int(sm[1].item())
                self.moves_played += int(sm[2].item())
                if publish:
                    self._publish(lead_step)
                    self._published_count = lead_step // freq
                    self.weight_update_count += 1
                    if ctx.rank == lead:
                        self._send_event_async("Progress/Weight_Updates_Total", self.weight_update_count)
                if ctx.rank == lead:
                    self._send_event_async("Progress/Episodes_Played", self.episodes_played)
                    self._send_event_async("Progress/Total_Simulations", self.total_simulations_run)
                    self._send_event_async("Buffer/Size", len(self.buffer))
                    self._send_event_async("System/Num_Active_Workers", len(ctx.layout.actor_ranks))
                    self.loop_helpers.log_progress_eta()
                if self.trieye_actor:
                    self.trieye_actor.process_and_log(self.global_step)
                self.rounds += 1
                if stop:
                    self.training_complete = bool(target is not None and lead_step >= target)
                    self.request_stop()
                    break
        except Exception as exc:
            logger.critical("Unhandled exception in SplitRoleLoop: %s", exc, exc_info=True)
            self.training_exception = exc
            self.request_stop()
        finally:
            logger.info("SplitRoleLoop finished after %d rounds. Complete: %s", self.rounds, self.training_complete)
