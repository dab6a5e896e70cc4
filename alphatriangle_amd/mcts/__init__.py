"""Tree search: native host search (CPU) and the HBM-resident batched tree engine (GPU)."""

from .host_search import AlphaZeroNetworkInterface, TreeHandle, run_mcts

__all__ = ["AlphaZeroNetworkInterface", "TreeHandle", "run_mcts"]
