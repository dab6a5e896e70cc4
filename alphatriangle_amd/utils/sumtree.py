"""Sum tree for prioritized replay, backed by the native C++ tree.

Same observable layout and behaviour as alphatriangle/utils/sumtree.py:12-98:
``tree`` is a float64 array of size ``2*capacity-1`` (a live view of the native
storage), ``data`` an object array, ``data_pointer`` a ring cursor,
``max_priority`` returns 1.0 while empty.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import native


class SumTree:
    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self._core = native.host().SumTreeCore(self.capacity)
        self.tree: np.ndarray = self._core.tree_view()
        self.data = np.zeros(self.capacity, dtype=object)

    # -- ring state lives in the native object; exposed for checkpoint code and tests
    @property
    def data_pointer(self) -> int:
        return int(self._core.data_pointer)

    @data_pointer.setter
    def data_pointer(self, v: int) -> None:
        self._core.data_pointer = int(v)

    @property
    def n_entries(self) -> int:
        return int(self._core.n_entries)

    @n_entries.setter
    def n_entries(self, v: int) -> None:
        self._core.n_entries = int(v)

    @property
    def _max_priority(self) -> float:
        return float(self._core.max_priority_seen)

    @_max_priority.setter
    def _max_priority(self, v: float) -> None:
        self._core.max_priority_seen = float(v)

    def add(self, priority: float, data: Any) -> None:
        slot = self._core.add(float(priority))
        self.data[slot] = data

    def update(self, tree_idx: int, priority: float) -> None:
        self._core.update(int(tree_idx), float(priority))

    def update_batch(self, tree_indices: np.ndarray, priorities: np.ndarray) -> None:
        self._core.update_batch(np.asarray(tree_indices, dtype=np.int64), np.asarray(priorities, dtype=np.float64))

    def get_leaf(self, value: float) -> tuple[int, float, Any]:
        leaf = int(self._core.get_leaf(float(value)))
        return leaf, float(self.tree[leaf]), self.data[leaf - self.capacity + 1]

    def get_leaf_batch(self, values: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        return self._core.get_leaf_batch(np.asarray(values, dtype=np.float64))

    @property
    def total_priority(self) -> float:
        return float(self._core.total())

    @property
    def max_priority(self) -> float:
        return float(self._core.max_priority())

    def __len__(self) -> int:
        return self.n_entries
