"""Pool of CPU self-play workers driven by threads.

Same responsibilities as alphatriangle/training/worker_manager.py:18-238
(create N seeded workers, keep one episode task in flight per worker, collect
finished episodes, push fresh weights + trainer step to every worker, drop
workers that died, clean up).  Ray actors become plain objects on a thread pool:
the native search and the torch forward release the GIL, nothing is pickled, and
a weight update is a direct ``load_state_dict`` on each worker's own network.
"""

from __future__ import annotations

import concurrent.futures as cf
import logging
import threading

from ..rl.self_play.worker import SelfPlayWorker
from ..rl.types import SelfPlayResult
from .components import TrainingComponents

logger = logging.getLogger(__name__)


class WorkerDiedError(RuntimeError):
    """Raised by a worker task to signal that the worker is unusable and must be retired."""


class WorkerManager:
    def __init__(self, components: TrainingComponents):
        self.components = components
        self.train_config = components.train_config
        self.nn = components.nn
        self.stats_actor = components.trieye_actor
        self.actor_name = self.stats_actor.get_actor_name() if self.stats_actor else None
        self.run_base_dir_str = self.stats_actor.get_run_base_dir_str() if self.stats_actor else None
        self.workers: list[SelfPlayWorker | None] = []
        self.worker_tasks: dict[cf.Future, int] = {}
        self.active_worker_indices: set[int] = set()
        self._pool: cf.ThreadPoolExecutor | None = None
        self._weights_lock = threading.Lock()
        self._worker_locks: dict[int, threading.Lock] = {}

    def initialize_workers(self) -> None:
        n = self.train_config.NUM_SELF_PLAY_WORKERS
        logger.info("Initializing %d self-play workers (threads)...", n)
        weights = self.nn.get_weights()
        self.workers = [None] * n
        switch = self.components.system_config.TEMPERATURE_SWITCH_GAME_STEP
        for i in range(n):
            try:
                self.workers[i] = SelfPlayWorker(
                    actor_id=i,
                    env_config=self.components.env_config,
                    mcts_config=self.components.mcts_config,
                    model_config=self.components.model_config,
                    train_config=self.train_config,
                    stats_actor=self.stats_actor,
                    run_base_dir=self.run_base_dir_str,
                    initial_weights=weights,
                    seed=self.train_config.RANDOM_SEED + i,
                    worker_device_str=self.train_config.WORKER_DEVICE,
                    profile_this_worker=self.components.profile_workers and i == 0,
                    temperature_switch_game_step=switch,
                )
                self.active_worker_indices.add(i)
                self._worker_locks[i] = threading.Lock()
            except Exception as exc:
                logger.error("Failed to initialize worker %d: %s", i, exc, exc_info=True)
        self._pool = cf.ThreadPoolExecutor(max_workers=max(1, n), thread_name_prefix="atri-selfplay")
        logger.info("Initialized %d active self-play workers.", len(self.active_worker_indices))

    def _run(self, idx: int) -> SelfPlayResult:
        worker = self.workers[idx]
        if worker is None:
            raise WorkerDiedError(f"worker {idx} is gone")
        with self._worker_locks[idx]:
            return worker.run_episode()

    def submit_initial_tasks(self) -> None:
        for idx in sorted(self.active_worker_indices):
            self.submit_task(idx)

    def submit_task(self, worker_idx: int) -> None:
        if worker_idx not in self.active_worker_indices or self._pool is None:
            return
        if worker_idx in self.worker_tasks.values():
            return
        self.worker_tasks[self._pool.submit(self._run, worker_idx)] = worker_idx

    def get_completed_tasks(self, timeout: float = 0.1) -> list[tuple[int, SelfPlayResult | Exception]]:
        """Finished episodes (all that are done, at least waiting ``timeout`` for the first)."""
        if not self.worker_tasks:
            return []
        done, _ = cf.wait(list(self.worker_tasks.keys()), timeout=timeout, return_when=cf.FIRST_COMPLETED)
        out: list[tuple[int, SelfPlayResult | Exception]] = []
        for fut in done:
            idx = self.worker_tasks.pop(fut)
            try:
                res = fut.result()
                out.append((idx, SelfPlayResult.model_validate(res) if not isinstance(res, SelfPlayResult) else res))
            except WorkerDiedError as exc:
                logger.error("Worker %d died: %s. Retiring it.", idx, exc)
                self.workers[idx] = None
                self.active_worker_indices.discard(idx)
                out.append((idx, exc))
            except Exception as exc:
                logger.error("Error in task of worker %d: %s", idx, exc, exc_info=True)
                out.append((idx, exc))
        return out

    def update_worker_networks(self, global_step: int) -> None:
        """Load the learner's current weights into every active worker and tell it the trainer step."""
        active = [w for i, w in enumerate(self.workers) if i in self.active_worker_indices and w is not None]
        if not active:
            return
        weights = self.nn.get_weights()
        for i in sorted(self.active_worker_indices):
            w = self.workers[i]
            if w is None:
                continue
            # idle worker: load now; busy worker: stage, it swaps at its next move boundary
            lock = self._worker_locks[i]
            if lock.acquire(blocking=False):
                try:
                    w.set_weights(weights)
                    w.set_current_trainer_step(global_step)
                finally:
                    lock.release()
            else:
                w.stage_weights(weights, global_step)
        logger.debug("Worker networks updated for step %d.", global_step)

    def get_num_active_workers(self) -> int:
        return len(self.active_worker_indices)

    def get_num_pending_tasks(self) -> int:
        return len(self.worker_tasks)

    def cleanup_actors(self) -> None:
        for fut in list(self.worker_tasks.keys()):
            fut.cancel()
        self.worker_tasks.clear()
        if self._pool is not None:
            self._pool.shutdown(wait=True, cancel_futures=True)
            self._pool = None
        self.workers = []
        self.active_worker_indices.clear()
        logger.info("Workers cleaned up.")
