"""HBM-resident batched game + tree engine (Python face of csrc/hip/engine_kernels.hip).

``GpuTreeEngine`` owns, for ``G`` concurrent games on one GPU:
  * two node arenas (states 64 B - 160 B on boards above 256 cells -, node meta 32 B, dense 16 B edges) so re-rooting
    can compact the reused subtree from one arena into the other,
  * the per-wave leaf slots (selection paths, network input rows),
  * the per-game self-play state (n-step pending ring, episode counters).

Sized for 288 GB of HBM: with the defaults (4096 games, 802 nodes, 360 actions) the
two edge arenas take 37.8 GB.  One wavefront drives one tree; see the kernel file
for the execution model.  Replaces ``trimcts.run_mcts`` + the per-move worker glue
of alphatriangle/rl/self_play/worker.py:253-458.
"""

from __future__ import annotations

import ctypes as C
import logging
from typing import Any

import numpy as np
import torch

from ..config.env_config import EnvConfig
from ..config.mcts_config import AlphaTriangleMCTSConfig
from ..env.game_state import get_geometry
from . import hipbind as hb

logger = logging.getLogger(__name__)

STATE_BYTES = 64        # boards up to 256 cells (four occupancy words)
WIDE_STATE_BYTES = 160  # boards of 257 - 1024 cells (sixteen words): the wide build of the host and the device engine
META_BYTES = 32
EDGE_BYTES = 16
SLOT_NONE, SLOT_EXPAND, SLOT_TERMINAL, SLOT_DEPTHCAP = 0, 1, 2, 3
CENSUS_WIDTH = 8       # per-tree counters: [0..3] leaf slots by kind, [4] non-finite network evaluations
CENSUS_NON_FINITE = 4


class GpuTreeEngine:
    def __init__(
        self,
        env_config: EnvConfig,
        mcts_config: AlphaTriangleMCTSConfig,
        device: torch.device,
        games: int,
        leaves_per_tree: int,
        other_dim: int,
        logits: torch.Tensor,
        values: torch.Tensor,
        seed: int = 0,
        compact: bool = False,
    ):
        hb.lib()
        self.env_config = env_config
        self.mcts_config = mcts_config
        self.device = torch.device(device)
        self.geom_host = get_geometry(env_config)
        self.geom = hb.DeviceGeometry(self.geom_host, self.device)
        # the kernel file is built twice: entry points atri_* for the 64-byte state, atriw_* for the 160-byte one
        self.wide = int(self.geom_host.nwords) > 4
        self.state_bytes = WIDE_STATE_BYTES if self.wide else STATE_BYTES
        self._api = "atriw_" if self.wide else "atri_"
        self.G = int(games)
        self.L = int(leaves_per_tree)
        self.A = int(self.geom_host.action_dim)
        self.cells = int(self.geom_host.cells)
        self.C = int(mcts_config.capacity())
        self.P = int(mcts_config.max_depth)
        self.other_dim = int(other_dim)
        if logits.shape[0] < self.G * self.L or values.shape[0] < self.G * self.L:
            raise ValueError("network output buffers are smaller than games * leaves_per_tree")
        self.logits = logits
        self.values = values
        dev = self.device
        G, Cn, A, L = self.G, self.C, self.A, self.L
        edge_bytes = 2 * G * Cn * A * EDGE_BYTES
        logger.info("GpuTreeEngine: %d games x %d nodes x %d actions -> %.2f GiB of edges in HBM", G, Cn, A, edge_bytes / 2**30)
        self.states = torch.zeros(2, G, Cn, self.state_bytes, dtype=torch.uint8, device=dev)
        self.meta = torch.zeros(2, G, Cn, META_BYTES, dtype=torch.uint8, device=dev)
        self.edges = torch.empty(2, G, Cn, A, 4, dtype=torch.int32, device=dev)
        self.cur = torch.zeros(G, dtype=torch.int32, device=dev)
        self.n_nodes = torch.zeros(G, dtype=torch.int32, device=dev)
        self.slot_kind = torch.zeros(G * L, dtype=torch.int32, device=dev)
        self.slot_leaf = torch.zeros(G * L, dtype=torch.int32, device=dev)
        self.slot_plen = torch.zeros(G * L, dtype=torch.int32, device=dev)
        self.path_node = torch.zeros(G * L, self.P, dtype=torch.int32, device=dev)
        self.path_action = torch.zeros(G * L, self.P, dtype=torch.int32, device=dev)
        self.nn_grid = torch.zeros(G * L, self.cells, dtype=torch.float32, device=dev)
        self.nn_other = torch.zeros(G * L, self.other_dim, dtype=torch.float32, device=dev)
        self.order = torch.zeros(G, Cn, dtype=torch.int32, device=dev)
        self.noise_rng = torch.from_numpy(
            np.random.default_rng(seed ^ 0x5EED).integers(1, 2**62, size=G, dtype=np.int64)
        ).to(dev)
        self.root_states = torch.zeros(G, self.state_bytes, dtype=torch.uint8, device=dev)
        self.census = torch.zeros(G, CENSUS_WIDTH, dtype=torch.int32, device=dev)
        # batch compaction: a selecting wave writes the rows of the leaves that need an evaluation densely and counts them
        self.compact = bool(compact)
        self.slot_row = torch.zeros(G * L, dtype=torch.int32, device=dev) if self.compact else None
        self.live = torch.zeros(1, dtype=torch.int32, device=dev) if self.compact else None

        tb = hb.TreeBufsC()
        for a in range(2):
            tb.states[a] = self.states[a].data_ptr()
            tb.meta[a] = self.meta[a].data_ptr()
            tb.edges[a] = self.edges[a].data_ptr()
        tb.cur = self.cur.data_ptr()
        tb.n_nodes = self.n_nodes.data_ptr()
        tb.slot_kind = self.slot_kind.data_ptr()
        tb.slot_leaf = self.slot_leaf.data_ptr()
        tb.slot_plen = self.slot_plen.data_ptr()
        tb.path_node = self.path_node.data_ptr()
        tb.path_action = self.path_action.data_ptr()
        tb.nn_grid = self.nn_grid.data_ptr()
        tb.nn_other = self.nn_other.data_ptr()
        tb.nn_logits = self.logits.data_ptr()
        tb.nn_value = self.values.data_ptr()
        tb.order = self.order.data_ptr()
        tb.census = self.census.data_ptr()
        tb.slot_row = self.slot_row.data_ptr() if self.slot_row is not None else 0
        tb.live_cursor = self.live.data_ptr() if self.live is not None else 0
        tb.G, tb.C, tb.A, tb.L, tb.P = G, Cn, A, L, self.P
        tb.other_dim = self.other_dim
        tb.logit_stride = int(self.logits.stride(0))
        self.tb = tb
        self.sp = self._search_params()
        self.gb: hb.GameBufsC | None = None

    def _search_params(self) -> hb.SearchParamsC:
        m = self.mcts_config
        sp = hb.SearchParamsC()
        sp.max_simulations = m.max_simulations
        sp.max_depth = m.max_depth
        sp.leaf_batch = self.L
        sp.fpu_parent = 1 if m.fpu_parent else 0
        sp.cpuct = m.cpuct
        sp.dirichlet_alpha = m.dirichlet_alpha
        sp.dirichlet_epsilon = m.dirichlet_epsilon
        sp.discount = m.discount
        sp.virtual_loss = m.virtual_loss
        return sp

    # ------------------------------------------------------------------ game / tree control
    def reset_games(self, seeds: np.ndarray | torch.Tensor) -> None:
        """Fresh games from explicit seeds and fresh single-node trees."""
        s = torch.as_tensor(np.asarray(seeds, dtype=np.uint64).view(np.int64)) if not isinstance(seeds, torch.Tensor) else seeds
        s = s.to(self.device)
        if s.numel() != self.G:
            raise ValueError("need one seed per game")
        hb.call(self._api + "reset_states", self.geom.ref(), hb.ptr(self.root_states), hb.ptr(s), hb.i32(self.G))
        self.set_roots(self.root_states)

    def set_roots(self, packed_states: torch.Tensor) -> None:
        """Start fresh trees from packed root states ``[G, state_bytes]`` uint8 (``GameState.core.pack()`` of the same board)."""
        st = packed_states.to(self.device).contiguous()
        if st.shape != (self.G, self.state_bytes) or st.dtype != torch.uint8:
            raise ValueError(f"packed_states must be [G, {self.state_bytes}] uint8")
        self.cur.zero_()
        hb.call(self._api + "tree_reset", C.byref(self.tb), hb.ptr(st), hb.ptr(None))

    def prepare_roots(self) -> None:
        hb.call(self._api + "prepare_roots", self.geom.ref(), C.byref(self.tb))

    def init_roots(self) -> None:
        hb.call(self._api + "init_roots", self.geom.ref(), C.byref(self.tb), C.byref(self.sp), hb.ptr(self.noise_rng))

    def wave(self, consume: bool, select: bool) -> None:
        """One simulation wave.  With compaction, a selecting wave is followed by the slot scan and the feature kernel
        (same launch call); ``self.live`` then holds the number of dense feature rows = leaves that need an evaluation."""
        hb.call(self._api + "tree_wave", self.geom.ref(), C.byref(self.tb), C.byref(self.sp), hb.i32(1 if consume else 0), hb.i32(1 if select else 0))

    def census_totals(self) -> torch.Tensor:
        """Per-kind leaf-slot counts and the non-finite evaluation count summed over all trees (``[CENSUS_WIDTH]`` int64)."""
        return self.census.sum(dim=0, dtype=torch.int64)

    def root_visits(self) -> torch.Tensor:
        out = torch.zeros(self.G, self.A, dtype=torch.int32, device=self.device)
        hb.call(self._api + "root_visits", C.byref(self.tb), hb.ptr(out))
        return out

    def current_root_states(self) -> torch.Tensor:
        """Packed root states ``[G, state_bytes]`` of the active arenas."""
        idx = self.cur.long()
        return self.states[idx, torch.arange(self.G, device=self.device), 0]

    # ------------------------------------------------------------------ batched engine ops on packed states
    def _check_states(self, states: torch.Tensor) -> None:
        if states.dim() != 2 or states.shape[1] != self.state_bytes or states.dtype != torch.uint8 or not states.is_contiguous():
            raise ValueError(f"states must be a contiguous [n, {self.state_bytes}] uint8 tensor")

    def step_states(self, states: torch.Tensor, actions: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        self._check_states(states)
        n = states.shape[0]
        rewards = torch.zeros(n, dtype=torch.float32, device=self.device)
        done = torch.zeros(n, dtype=torch.int32, device=self.device)
        hb.call(self._api + "step_states", self.geom.ref(), hb.ptr(states), hb.ptr(actions.to(torch.int32)), hb.ptr(rewards), hb.ptr(done), hb.i32(n))
        return rewards, done

    def valid_mask(self, states: torch.Tensor) -> torch.Tensor:
        self._check_states(states)
        n = states.shape[0]
        mask = torch.zeros(n, self.A, dtype=torch.uint8, device=self.device)
        hb.call(self._api + "valid_mask", self.geom.ref(), hb.ptr(states), hb.ptr(mask), hb.i32(n))
        return mask

    def features(self, states: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        self._check_states(states)
        n = states.shape[0]
        grid = torch.zeros(n, self.cells, dtype=torch.float32, device=self.device)
        other = torch.zeros(n, self.other_dim, dtype=torch.float32, device=self.device)
        hb.call(self._api + "features", self.geom.ref(), hb.ptr(states), hb.ptr(grid), hb.ptr(other), hb.i32(self.other_dim), hb.i32(n))
        return grid, other

    # ------------------------------------------------------------------ self-play state
    def attach_replay(self, replay: Any, n_step: int, gamma: float, base_seed: int, temp_switch_step: int,
                      tree_reuse: bool, ep_ring: int = 1 << 16) -> None:
        """Allocate the per-game self-play buffers and bind the replay tensors the commit kernel writes to."""
        dev, G = self.device, self.G
        R = n_step + 1
        self.R = R
        self.replay = replay
        self.game_rng = torch.from_numpy(np.random.default_rng(base_seed ^ 0xA11CE).integers(1, 2**62, size=G, dtype=np.int64)).to(dev)
        self.episode_idx = torch.zeros(G, dtype=torch.int32, device=dev)
        self.ep_sims = torch.zeros(G, dtype=torch.int32, device=dev)
        self.pend_grid = torch.zeros(G, R, self.cells, dtype=torch.float32, device=dev)
        self.pend_other = torch.zeros(G, R, self.other_dim, dtype=torch.float32, device=dev)
        self.pend_policy = torch.zeros(G, R, self.A, dtype=torch.float32, device=dev)
        self.pend_acc = torch.zeros(G, R, dtype=torch.float32, device=dev)
        self.pend_count = torch.zeros(G, R, dtype=torch.int32, device=dev)
        self.pend_head = torch.zeros(G, dtype=torch.int32, device=dev)
        self.pend_len = torch.zeros(G, dtype=torch.int32, device=dev)
        self.ep_ring = int(ep_ring)
        self.ep_score = torch.zeros(self.ep_ring, dtype=torch.float32, device=dev)
        self.ep_len = torch.zeros(self.ep_ring, dtype=torch.int32, device=dev)
        self.ep_sims_out = torch.zeros(self.ep_ring, dtype=torch.int32, device=dev)
        self.ep_cleared = torch.zeros(self.ep_ring, dtype=torch.int32, device=dev)
        self.ep_count = torch.zeros(1, dtype=torch.int64, device=dev)
        self.move_stats = torch.zeros(8, dtype=torch.float32, device=dev)
        self.next_root = torch.zeros(G, self.state_bytes, dtype=torch.uint8, device=dev)
        self.reroot_child = torch.zeros(G, dtype=torch.int32, device=dev)
        self.chosen_action = torch.zeros(G, dtype=torch.int32, device=dev)
        self.gamma_pow = torch.tensor([gamma**i for i in range(R + 1)], dtype=torch.float32, device=dev)
        self.step_out = torch.zeros(3, G, dtype=torch.float32, device=dev)  # rows: reward (NaN = no move), cleared, score
        if replay.emit_capacity < G * R:
            raise ValueError("replay.emit_capacity must be at least games * (n_step + 1)")
        gb = hb.GameBufsC()
        gb.game_rng = self.game_rng.data_ptr()
        gb.episode_idx = self.episode_idx.data_ptr()
        gb.ep_sims = self.ep_sims.data_ptr()
        gb.pend_grid = self.pend_grid.data_ptr()
        gb.pend_other = self.pend_other.data_ptr()
        gb.pend_policy = self.pend_policy.data_ptr()
        gb.pend_acc = self.pend_acc.data_ptr()
        gb.pend_count = self.pend_count.data_ptr()
        gb.pend_head = self.pend_head.data_ptr()
        gb.pend_len = self.pend_len.data_ptr()
        gb.rp_grid = replay.grid.data_ptr()
        gb.rp_other = replay.other.data_ptr()
        gb.rp_policy = replay.policy.data_ptr()
        gb.rp_ret = replay.ret.data_ptr()
        gb.rp_head = replay.head.data_ptr()
        gb.emitted_idx = replay.emitted_idx.data_ptr()
        gb.emitted_count = replay.emitted_count.data_ptr()
        gb.ep_score = self.ep_score.data_ptr()
        gb.ep_len = self.ep_len.data_ptr()
        gb.ep_sims_out = self.ep_sims_out.data_ptr()
        gb.ep_cleared = self.ep_cleared.data_ptr()
        gb.ep_count = self.ep_count.data_ptr()
        gb.move_stats = self.move_stats.data_ptr()
        gb.next_root = self.next_root.data_ptr()
        gb.reroot_child = self.reroot_child.data_ptr()
        gb.chosen_action = self.chosen_action.data_ptr()
        gb.gamma_pow = self.gamma_pow.data_ptr()
        gb.step_reward = self.step_out[0].data_ptr()
        gb.step_cleared = self.step_out[1].data_ptr()
        gb.step_score = self.step_out[2].data_ptr()
        gb.base_seed = int(base_seed) & (2**64 - 1)
        gb.R = R
        gb.n_step = n_step
        gb.cap = replay.capacity
        gb.emit_cap = replay.emit_capacity
        gb.ep_ring = self.ep_ring
        gb.temp_switch_step = int(min(temp_switch_step, 2**31 - 1))
        gb.tree_reuse = 1 if tree_reuse else 0
        gb.reuse_limit = max(1, self.C - self.mcts_config.max_simulations - 1)
        gb.temp_hi = 1.0
        gb.temp_lo = 0.1
        self.gb = gb

    def commit_moves(self) -> None:
        if self.gb is None:
            raise RuntimeError("attach_replay must be called before commit_moves")
        hb.call(self._api + "commit_moves", self.geom.ref(), C.byref(self.tb), C.byref(self.gb))

    def reroot(self) -> None:
        if self.gb is None:
            raise RuntimeError("attach_replay must be called before reroot")
        hb.call(self._api + "reroot", self.geom.ref(), C.byref(self.tb), C.byref(self.gb))
