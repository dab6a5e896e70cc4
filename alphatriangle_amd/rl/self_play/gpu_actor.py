"""GPU-resident self-play actor: thousands of games, trees and the network on one MI355X.

Replaces a whole pool of the reference's Ray CPU workers
(alphatriangle/rl/self_play/worker.py:35-599 + alphatriangle/training/worker_manager.py):
one ``play_move`` call advances every game by one move - root evaluation, Dirichlet
noise, ``max_simulations`` simulation waves (tree kernel -> network forward over all
leaves), policy target + action sampling, n-step bookkeeping, experience emission into
the HBM replay, re-rooting with subtree reuse.  The steady-state wave
(tree kernel + the network kernels) is captured once into a HIP graph and replayed.

Leaf evaluation is pluggable (:class:`LeafEvaluator`): the hand-written HIP engine in
bf16 or in its fp32-grade split-precision mode, or - for model configurations the
kernels do not tile - the PyTorch module itself in fp32, driving the same tree kernels
(alphatriangle/nn/network.py:243-318 is the reference's leaf-evaluation entry point).

Every evaluation is checked on the device (the reference raises on non-finite outputs:
alphatriangle/nn/network.py:177-191,260-274): the tree kernel counts leaves whose logits
or value were not finite, gives them a uniform prior so the search stays well defined,
and ``finish_move`` raises :class:`NetworkEvaluationError` when the count is not zero.
"""

from __future__ import annotations

import copy
import logging
import time
from dataclasses import dataclass, field
from typing import Any

import numpy as np
import torch

from ...config.env_config import EnvConfig
from ...config.mcts_config import AlphaTriangleMCTSConfig
from ...config.model_config import ModelConfig
from ...config.system_config import SystemConfig
from ...config.train_config import TrainConfig
from ...models.network import NetworkEvaluationError, NeuralNetwork
from ...ops.engine import CENSUS_NON_FINITE, SLOT_DEPTHCAP, SLOT_EXPAND, SLOT_NONE, SLOT_TERMINAL, GpuTreeEngine
from ...rl.core.device_buffer import DeviceReplayBuffer

logger = logging.getLogger(__name__)


@dataclass
class GpuMoveReport:
    """What one ``play_move`` call produced (all games advanced by one move)."""

    moves: int
    simulations: int
    experiences_added: int
    episodes_finished: int
    episode_contexts: list[dict[str, float]] = field(default_factory=list)
    mean_step_reward: float = 0.0
    triangles_cleared: int = 0
    trainer_step: int = 0
    seconds: float = 0.0
    # one entry per game that moved: the per-game-step metrics of the reference's worker
    # (alphatriangle/rl/self_play/worker.py:398-408,443-447)
    per_game: dict[str, np.ndarray] = field(default_factory=dict)
    # leaf slots of this move by kind; ``expand`` slots are the ones that needed a network evaluation
    slot_census: dict[str, int] = field(default_factory=dict)
    # GPU time of the move's phases in milliseconds (HIP events; only with --profile, see gpu_profile.py)
    phase_ms: dict[str, float] = field(default_factory=dict)


class LeafEvaluator:
    """Evaluates the network on the tree's leaf rows; results land in ``logits`` / ``values`` (stable addresses)."""

    logits: torch.Tensor
    values: torch.Tensor
    graph_safe = False
    supports_live = False  # can skip rows beyond a device-side row count
    dtype = "fp32"

    def forward(self, grid: torch.Tensor, other: torch.Tensor, live: torch.Tensor | None = None) -> None:
        """Evaluate rows ``0 .. B-1`` (or ``0 .. live-1`` when a device row count is given and supported)."""
        raise NotImplementedError

    def refresh(self) -> None:
        """Pick up the learner's current parameters."""
        raise NotImplementedError


class EngineEvaluator(LeafEvaluator):
    """The hand-written HIP network (ops/inference.py)."""

    graph_safe = True
    supports_live = True

    def __init__(self, nn: NeuralNetwork, batch: int):
        self.nn = nn
        self.engine = nn.engine
        self.engine.reserve(batch)
        self.logits = self.engine.logits
        self.values = self.engine.values
        self.dtype = getattr(self.engine, "dtype", "bf16")

    def forward(self, grid: torch.Tensor, other: torch.Tensor, live: torch.Tensor | None = None) -> None:
        self.engine.forward(grid, other, live)

    def refresh(self) -> None:
        self.nn.sync_inference_weights()


class ModuleEvaluator(LeafEvaluator):
    """The PyTorch module in eval mode (fp32, or bf16 autocast when that is the requested precision): any ``ModelConfig``
    and any board the reference accepts, same tree kernels.

    It keeps a private eval-mode copy of the module, because the learner switches the shared one to training mode
    (BatchNorm batch statistics, dropout) and may update it on another stream while a search is running.  With batch
    compaction it evaluates the dense rows only: the row count comes back to the host once per wave (one small
    synchronous copy - this evaluator is launched eagerly, so it does not break a graph), rounded up to a quarter of the
    batch, so that the library sees four batch sizes: MIOpen builds its convolution kernels per shape at first use, and
    with a step of 64 rows a bf16 run spent minutes compiling when its first games ended and the batches thinned out.
    """

    supports_live = True

    def __init__(self, nn: NeuralNetwork, batch: int, chunk: int = 2048, dtype: str = "fp32"):
        self.nn = nn
        self.dtype = "bf16" if dtype == "bf16" else "fp32"
        self.rows_evaluated = 0  # rows sent through the module since construction (tests, logging)
        self.row_step = max(64, (int(batch) + 255) // 256 * 64)  # a quarter of the batch, in whole 64-row blocks
        dev = nn.device
        mc, ec = nn.model_config, nn.env_config
        self.module = copy.deepcopy(nn._orig_model).to(dev).eval()
        for p in self.module.parameters():
            p.requires_grad_(False)
        self.A = nn.action_dim
        stride = (self.A + 7) // 8 * 8
        self.logits = torch.zeros(batch, stride, dtype=torch.float32, device=dev)
        self.values = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.support = nn.support.float()
        self.shape = (mc.GRID_INPUT_CHANNELS, ec.ROWS, ec.COLS)
        self.cells = ec.ROWS * ec.COLS
        # channel 0 carries the board; further input channels stay zero (alphatriangle/features/extractor.py:33-46)
        self.planes = torch.zeros(batch, *self.shape, dtype=torch.float32, device=dev) if self.shape[0] > 1 else None
        self.chunk = int(chunk)

    @torch.no_grad()
    def forward(self, grid: torch.Tensor, other: torch.Tensor, live: torch.Tensor | None = None) -> None:
        B = int(grid.shape[0])
        if live is not None:
            n = int(live.item())
            B = min(B, (n + self.row_step - 1) // self.row_step * self.row_step)
            if B == 0:
                return
            grid, other = grid[:B], other[:B]
        self.rows_evaluated += B
        if self.planes is not None:
            self.planes[:B, 0] = grid.view(B, self.shape[1], self.shape[2])
            x = self.planes[:B]
        else:
            x = grid.view(B, *self.shape)
        for s in range(0, B, self.chunk):
            e = min(B, s + self.chunk)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.dtype == "bf16" and x.is_cuda):
                pl, vl = self.module(x[s:e], other[s:e])
            self.logits[s:e, : self.A].copy_(pl.float())
            torch.sum(torch.softmax(vl.float(), dim=1) * self.support, dim=1, out=self.values[s:e])

    @torch.no_grad()
    def refresh(self) -> None:
        src = self.nn._orig_model.state_dict()
        for k, v in self.module.state_dict().items():
            v.copy_(src[k])


def make_evaluator(nn: NeuralNetwork, system_config: SystemConfig, batch: int) -> LeafEvaluator:
    """The fastest evaluator that honours ``SystemConfig.INFERENCE_DTYPE`` for this model configuration."""
    want = str(getattr(system_config, "INFERENCE_DTYPE", "bf16"))
    engine = nn.engine
    if engine is not None and getattr(engine, "dtype", "bf16") == want:
        return EngineEvaluator(nn, batch)
    if engine is not None and want == "bf16":
        return EngineEvaluator(nn, batch)
    if engine is None:
        logger.warning("No HIP inference engine for this model configuration: leaves are evaluated by the PyTorch module (%s).",
                       "bf16 autocast" if want == "bf16" else "fp32")
    else:
        logger.info("INFERENCE_DTYPE=%s: leaves are evaluated by the PyTorch module in fp32.", want)
    return ModuleEvaluator(nn, batch, dtype=want if engine is None else "fp32")


class GpuSelfPlayActor:
    def __init__(
        self,
        env_config: EnvConfig,
        model_config: ModelConfig,
        mcts_config: AlphaTriangleMCTSConfig,
        train_config: TrainConfig,
        system_config: SystemConfig,
        nn: NeuralNetwork,
        replay: DeviceReplayBuffer,
        device: torch.device,
        seed: int = 0,
        evaluator: LeafEvaluator | None = None,
        profile_dir: str | None = None,
        rank: int = 0,
    ):
        self.env_config = env_config
        self.mcts_config = mcts_config
        self.train_config = train_config
        self.system_config = system_config
        self.nn = nn
        self.replay = replay
        self.device = torch.device(device)
        self.G = int(system_config.GAMES_PER_GPU)
        self.L = int(system_config.LEAVES_PER_TREE)
        self.B = self.G * self.L
        self.evaluator = evaluator if evaluator is not None else make_evaluator(nn, system_config, self.B)
        self.compact = bool(getattr(system_config, "COMPACT_LEAF_BATCH", True)) and self.evaluator.supports_live
        self.tree = GpuTreeEngine(
            env_config, mcts_config, self.device, self.G, self.L, model_config.OTHER_NN_INPUT_FEATURES_DIM,
            self.evaluator.logits, self.evaluator.values, seed=seed, compact=self.compact,
        )
        if system_config.TEMPERATURE_SWITCH_GAME_STEP is not None:
            switch = int(system_config.TEMPERATURE_SWITCH_GAME_STEP)
        elif train_config.MAX_TRAINING_STEPS is not None:
            switch = int(train_config.MAX_TRAINING_STEPS * 0.1)  # the reference's rule
        else:
            switch = 2**31 - 1
        if replay.emit_capacity < self.G * (train_config.N_STEP_RETURNS + 1):
            raise ValueError("DeviceReplayBuffer.emit_capacity is too small for this many games")
        self.tree.attach_replay(
            replay, train_config.N_STEP_RETURNS, train_config.GAMMA, base_seed=seed, temp_switch_step=switch,
            tree_reuse=system_config.TREE_REUSE,
        )
        self.waves_per_move = (mcts_config.max_simulations + self.L - 1) // self.L
        seeds = np.random.default_rng(seed).integers(1, 2**63 - 1, size=self.G, dtype=np.int64).astype(np.uint64)
        self.tree.reset_games(seeds)
        self.current_trainer_step = 0
        self.total_moves = 0
        self.total_episodes = 0
        self._ep_seen = 0
        self._graph: torch.cuda.CUDAGraph | None = None
        self._t0 = time.perf_counter()
        self.profile_ranges = bool(train_config.PROFILE_WORKERS)  # --profile: roctx ranges around the move phases
        # --profile also times the phases with HIP events and writes <run>/profile_data/gpu_actor_rank<r>.jsonl
        self.profiler = None
        if self.profile_ranges and self.device.type == "cuda":
            from .gpu_profile import MoveProfiler

            self.profiler = MoveProfiler(self.device, profile_dir, rank)
        self._use_graph = bool(system_config.USE_HIP_GRAPHS) and self.evaluator.graph_safe
        self.per_game_metrics = True
        self._grid_view = self.tree.nn_grid
        self._other_view = self.tree.nn_other
        logger.info(
            "GPU self-play actor: %d games, %d leaves/tree, %d sims/move -> network batch %d, %d waves/move, evaluator %s (%s)",
            self.G, self.L, mcts_config.max_simulations, self.B, self.waves_per_move, type(self.evaluator).__name__,
            self.evaluator.dtype,
        )

    @property
    def dtype(self) -> str:
        """Precision of the leaf evaluations that drive the search."""
        return self.evaluator.dtype

    # ------------------------------------------------------------------ weights
    def sync_weights(self, trainer_step: int) -> None:
        """Make the evaluator use the learner's current parameters (re-fold + re-pack in place for the HIP engine)."""
        self.evaluator.refresh()
        self.current_trainer_step = int(trainer_step)

    # ------------------------------------------------------------------ one simulation wave
    def _forward(self, live: torch.Tensor | None = None) -> None:
        self.evaluator.forward(self._grid_view, self._other_view, live)

    def _steady_wave(self) -> None:
        self.tree.wave(consume=True, select=True)
        self._forward(self.tree.live)  # with compaction: only the rows this wave wrote

    def _capture(self) -> None:
        """Capture the steady-state wave into a HIP graph.

        Every kernel in it has already run eagerly (root evaluation + first wave of this move),
        and the wave allocates nothing, so capture needs no separate warm-up.
        """
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._steady_wave()
        self._graph = g

    @torch.no_grad()
    def begin_move(self) -> None:
        """Enqueue the whole search of one move (root evaluation + every simulation wave) without any host sync.

        Returns after a few milliseconds of launches; the GPU then works for the rest of the move, which leaves
        the host free to enqueue learner steps on another stream before :meth:`finish_move`.
        """
        self._t0 = time.perf_counter()
        tree = self.tree
        tree.move_stats.zero_()
        tree.census.zero_()
        rng = self._range
        prof = self.profiler
        if prof is not None:
            prof.mark("move_start")
        rng("selfplay/root_eval")
        tree.prepare_roots()
        self._forward()
        tree.init_roots()
        rng(None)
        if prof is not None:
            prof.mark("root_done")
        rng("selfplay/waves")
        waves = self.waves_per_move
        tree.wave(consume=False, select=True)
        self._forward(tree.live)
        remaining = waves - 1
        if remaining > 0:
            if self._use_graph:
                if self._graph is None:
                    self._capture()  # capture executes nothing; the replays below do the work
                assert self._graph is not None
                for _ in range(remaining):
                    self._graph.replay()
            else:
                for _ in range(remaining):
                    self._steady_wave()
        tree.wave(consume=True, select=False)
        rng(None)
        if prof is not None:
            prof.mark("waves_done")

    def _range(self, name: str | None) -> None:
        """roctx range markers (visible in rocprofv3 --marker-trace) when profiling is switched on."""
        if not self.profile_ranges:
            return
        if name is None:
            torch.cuda.nvtx.range_pop()
        else:
            torch.cuda.nvtx.range_push(name)

    @torch.no_grad()
    def finish_move(self, wait_event: "torch.cuda.Event | None" = None) -> GpuMoveReport:
        """Commit the move (policy targets, sampling, n-step, emission into the replay), re-root, read the counters.

        ``wait_event``: an event of another stream (a concurrently running learner) that must have completed
        before the replay's sum tree is touched.  Raises :class:`NetworkEvaluationError` when any leaf evaluation
        of this move was not finite; the move is committed first, so every kernel has run to its normal end.
        """
        t0 = self._t0
        tree = self.tree
        if wait_event is not None:
            torch.cuda.current_stream(self.device).wait_event(wait_event)
        self._range("selfplay/commit")
        self.replay.begin_commit()
        tree.commit_moves()
        tree.reroot()
        added = self.replay.end_commit()
        self._range(None)
        if self.profiler is not None:
            self.profiler.mark("move_end")
        stats = tree.move_stats.cpu().numpy()
        census = tree.census_totals().cpu().numpy()
        ep_total = int(tree.ep_count.item())
        new_eps = ep_total - self._ep_seen
        contexts: list[dict[str, float]] = []
        if new_eps > 0:
            take = min(new_eps, tree.ep_ring)
            idx = (torch.arange(ep_total - take, ep_total, device=self.device) % tree.ep_ring).long()
            score = tree.ep_score[idx].cpu().numpy()
            length = tree.ep_len[idx].cpu().numpy()
            sims = tree.ep_sims_out[idx].cpu().numpy()
            cleared = tree.ep_cleared[idx].cpu().numpy()
            for i in range(take):
                contexts.append({"score": float(score[i]), "length": int(length[i]), "simulations": int(sims[i]), "triangles_cleared": int(cleared[i])})
            self._ep_seen = ep_total
        per_game: dict[str, np.ndarray] = {}
        if self.per_game_metrics:
            out = tree.step_out.cpu().numpy()
            moved = np.isfinite(out[0])
            per_game = {"step_reward": out[0][moved], "triangles_cleared_step": out[1][moved], "current_score": out[2][moved]}
        moves = int(stats[2])
        self.total_moves += moves
        self.total_episodes += new_eps
        phase_ms: dict[str, float] = {}
        if self.profiler is not None:
            # every event of this move has completed: the copies above synchronised with the stream they were recorded on
            slots = int(census[SLOT_NONE] + census[SLOT_EXPAND] + census[SLOT_TERMINAL] + census[SLOT_DEPTHCAP])
            phase_ms = self.profiler.finish({
                "games_moved": moves, "simulations": int(stats[4]), "episodes_finished": new_eps,
                "leaf_share": round(float(census[SLOT_EXPAND]) / slots, 4) if slots else None,
                "trainer_step": self.current_trainer_step,
            })
        report = GpuMoveReport(
            moves=moves,
            simulations=int(stats[4]),
            experiences_added=added,
            episodes_finished=new_eps,
            episode_contexts=contexts,
            mean_step_reward=float(stats[0] / max(moves, 1)),
            triangles_cleared=int(stats[1]),
            trainer_step=self.current_trainer_step,
            seconds=time.perf_counter() - t0,
            per_game=per_game,
            slot_census={"none": int(census[SLOT_NONE]), "expand": int(census[SLOT_EXPAND]),
                         "terminal": int(census[SLOT_TERMINAL]), "depth_cap": int(census[SLOT_DEPTHCAP])},
            phase_ms=phase_ms,
        )
        bad = int(census[CENSUS_NON_FINITE])
        if bad:
            raise NetworkEvaluationError(
                f"{bad} leaf evaluations of this move returned non-finite policy logits or values "
                f"(trainer step {self.current_trainer_step}); the learner's parameters have most likely diverged"
            )
        return report

    def play_move(self) -> GpuMoveReport:
        self.begin_move()
        return self.finish_move()

    # ------------------------------------------------------------------ checkpointing of the actor state
    def state_dict(self) -> dict[str, Any]:
        return {"total_moves": self.total_moves, "total_episodes": self.total_episodes}

    def load_state(self, state: dict[str, Any]) -> None:
        self.total_moves = int(state.get("total_moves", 0))
        self.total_episodes = int(state.get("total_episodes", 0))
