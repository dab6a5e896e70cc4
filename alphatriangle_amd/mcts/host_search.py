"""Host (CPU) tree search front-end.

``run_mcts`` keeps the call shape of the external ``trimcts.run_mcts`` used at
alphatriangle/rl/self_play/worker.py:273-280: it returns root visit counts and an
opaque tree handle that enables subtree reuse on the next call.  The search
itself is the native C++ tree (csrc/host/mcts.h); leaves are evaluated in batches
through array-returning network calls.
"""

from __future__ import annotations

import logging
from typing import Any, Protocol

import numpy as np

from .. import native
from ..config.mcts_config import AlphaTriangleMCTSConfig
from ..env.game_state import GameState

logger = logging.getLogger(__name__)


class AlphaZeroNetworkInterface(Protocol):
    """Evaluator protocol (same two methods the reference's network exposes to its search library)."""

    def evaluate_state(self, state: GameState) -> tuple[dict[int, float], float]: ...

    def evaluate_batch(self, states: list[GameState]) -> list[tuple[dict[int, float], float]]: ...


class TreeHandle:
    """Opaque handle returned by :func:`run_mcts`; pass it back to reuse the subtree."""

    def __init__(self, tree: Any, config_key: tuple):
        self.tree = tree
        self.config_key = config_key

    @property
    def n_nodes(self) -> int:
        return int(self.tree.n_nodes)

    @property
    def max_depth_seen(self) -> int:
        return int(self.tree.max_depth_seen)


def _make_evaluator(network_interface: Any, env_config: Any, action_dim: int):
    """Adapt a network object to the native ``evaluator(list[CoreState]) -> (logits, values)`` callback."""
    fast = getattr(network_interface, "evaluate_batch_arrays", None)
    if callable(fast):
        return fast

    def compat(core_states: list[Any]) -> tuple[np.ndarray, np.ndarray]:
        states = [GameState.from_core(env_config, c) for c in core_states]
        results = network_interface.evaluate_batch(states)
        logits = np.full((len(states), action_dim), -1e30, dtype=np.float32)
        values = np.zeros(len(states), dtype=np.float32)
        for i, (policy, value) in enumerate(results):
            for a, p in policy.items():
                if 0 <= a < action_dim and p > 0:
                    logits[i, a] = np.log(p)
            values[i] = value
        return logits, values

    return compat


def run_mcts(
    root_state: GameState,
    network_interface: Any,
    config: AlphaTriangleMCTSConfig,
    previous_tree_handle: TreeHandle | None = None,
    last_action: int = -1,
    seed: int = 0,
) -> tuple[dict[int, int], TreeHandle]:
    """One search from ``root_state``; returns ``({action: visits}, handle)``."""
    geom = root_state.geometry
    key = (id(geom), config.capacity())
    handle = previous_tree_handle
    reuse_action = -1
    if handle is not None and handle.config_key == key and last_action is not None and last_action >= 0:
        reuse_action = int(last_action)
    else:
        handle = TreeHandle(native.host_of(geom).SearchTree(geom, config.capacity(), int(seed) & (2**63 - 1)), key)
    params = config.to_search_params(module=native.host_of(geom))
    evaluator = _make_evaluator(network_interface, root_state.env_config, geom.action_dim)
    visits = handle.tree.search(root_state.core, params, evaluator, reuse_action)
    nz = np.nonzero(visits)[0]
    return {int(a): int(visits[a]) for a in nz}, handle
