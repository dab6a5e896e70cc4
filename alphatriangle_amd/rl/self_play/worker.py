"""CPU self-play worker: plays full episodes with the native host search.

Capability parity with the Ray actor of alphatriangle/rl/self_play/worker.py:35-599
(per-move search with subtree reuse, T=1 policy targets, temperature action
sampling, n-step returns with value bootstrap, metric events, per-episode
cProfile dumps, error containment).  It is a plain object driven by a thread of
the worker manager: the native search and the network forward both release the
GIL, and nothing is pickled between the worker and the learner.
The GPU counterpart (thousands of games per device) is rl/self_play/gpu_actor.py.
"""

from __future__ import annotations

import cProfile
import logging
import random
import time
from pathlib import Path
from typing import Any

import numpy as np

from ...config.env_config import EnvConfig
from ...config.mcts_config import AlphaTriangleMCTSConfig
from ...config.model_config import ModelConfig
from ...config.train_config import TrainConfig
from ...env.game_state import GameState
from ...features.extractor import extract_state_features
from ...mcts.host_search import run_mcts
from ...models.network import NeuralNetwork
from ...stats.events import RawMetricEvent
from ...utils.helpers import get_device
from ...utils.types import Experience
from ..types import SelfPlayResult
from .mcts_helpers import PolicyGenerationError, get_policy_target_from_visits, select_action_from_visits
from .nstep import NStepAccumulator

logger = logging.getLogger(__name__)


class SelfPlayWorker:
    def __init__(
        self,
        actor_id: int,
        env_config: EnvConfig,
        mcts_config: AlphaTriangleMCTSConfig,
        model_config: ModelConfig,
        train_config: TrainConfig,
        stats_actor: Any = None,
        run_base_dir: str | None = None,
        initial_weights: dict | None = None,
        seed: int | None = None,
        worker_device_str: str = "cpu",
        profile_this_worker: bool = False,
        temperature_switch_game_step: int | None = None,
    ):
        self.actor_id = actor_id
        self.env_config = env_config
        self.mcts_config = mcts_config
        self.model_config = model_config
        self.train_config = train_config
        self.stats_actor = stats_actor
        self.run_base_dir = Path(run_base_dir) if run_base_dir else None
        self.seed = seed if seed is not None else random.randint(0, 1_000_000)
        self.profile_this_worker = profile_this_worker
        self.n_step = train_config.N_STEP_RETURNS
        self.gamma = train_config.GAMMA
        self.current_trainer_step = 0
        self._py_rng = random.Random(self.seed)
        self._np_rng = np.random.default_rng(self.seed)
        if temperature_switch_game_step is not None:
            self._explore_steps: float | None = float(temperature_switch_game_step)
        elif train_config.MAX_TRAINING_STEPS is not None:
            # same rule as the reference: game step compared with 10% of the training-step budget
            self._explore_steps = train_config.MAX_TRAINING_STEPS * 0.1
        else:
            self._explore_steps = None
        self.device = get_device(worker_device_str)
        self.nn_evaluator = NeuralNetwork(model_config, env_config, train_config, self.device)
        if initial_weights:
            self.set_weights(initial_weights)
        else:
            self.nn_evaluator.model.eval()
        self.profiler: cProfile.Profile | None = cProfile.Profile() if profile_this_worker else None
        self.episodes_done = 0
        self._staged: tuple[dict, int] | None = None
        logger.debug("Worker %d initialised on %s (seed %d).", actor_id, self.device, self.seed)

    # ------------------------------------------------------------------ remote-callable controls
    def set_weights(self, weights: dict) -> None:
        self.nn_evaluator.set_weights(weights)

    def set_current_trainer_step(self, global_step: int) -> None:
        self.current_trainer_step = int(global_step)

    def stage_weights(self, weights: dict, global_step: int) -> None:
        """Hand over weights while an episode is running; applied at the next move boundary."""
        self._staged = (weights, int(global_step))

    def _apply_staged(self) -> None:
        staged, self._staged = self._staged, None
        if staged is not None:
            self.set_weights(staged[0])
            self.set_current_trainer_step(staged[1])

    def _send_event_async(self, name: str, value: float, context: dict | None = None) -> None:
        if self.stats_actor is None:
            return
        try:
            self.stats_actor.log_event(
                RawMetricEvent(
                    name=name,
                    value=value,
                    global_step=self.current_trainer_step,
                    timestamp=time.time(),
                    context=context or {},
                )
            )
        except Exception as exc:  # metrics must never break self-play
            logger.error("Worker %d: failed to send event %s: %s", self.actor_id, name, exc)

    # ------------------------------------------------------------------ episode
    def _empty_result(self, step_at_start: int, score: float, steps: int, error: bool) -> SelfPlayResult:
        ctx: dict[str, Any] = {"score": score, "length": steps, "simulations": 0, "triangles_cleared": 0, "trainer_step": step_at_start}
        if error:
            ctx["error"] = True
        self._send_event_async("episode_end", 1.0, ctx)
        return SelfPlayResult(
            episode_experiences=[],
            final_score=score,
            episode_steps=steps,
            trainer_step_at_episode_start=step_at_start,
            total_simulations=0,
            avg_root_visits=0.0,
            avg_tree_depth=0.0,
            context=ctx,
        )

    def run_episode(self) -> SelfPlayResult:
        step_at_start = self.current_trainer_step
        episode_seed = self.seed + self._py_rng.randint(0, 1000)
        if self.profiler:
            self.profiler.enable()
        try:
            return self._run_episode_inner(step_at_start, episode_seed)
        except Exception as exc:
            logger.critical("Worker %d: unhandled exception in run_episode: %s", self.actor_id, exc, exc_info=True)
            return self._empty_result(step_at_start, 0.0, 0, error=True)
        finally:
            self.episodes_done += 1
            if self.profiler:
                self.profiler.disable()
                if self.run_base_dir is not None:
                    try:
                        out = self.run_base_dir / "profile_data"
                        out.mkdir(parents=True, exist_ok=True)
                        self.profiler.dump_stats(str(out / f"worker_{self.actor_id}_ep_{episode_seed}.prof"))
                    except Exception as exc:
                        logger.error("Worker %d: failed to write profile: %s", self.actor_id, exc)
                self.profiler = cProfile.Profile()

    def _run_episode_inner(self, step_at_start: int, episode_seed: int) -> SelfPlayResult:
        self.nn_evaluator.model.eval()
        game = GameState(self.env_config, initial_seed=episode_seed)
        if game.is_over() or not game.valid_actions():
            logger.error("Worker %d: game over or no valid actions at reset (seed %d).", self.actor_id, episode_seed)
            return self._empty_result(step_at_start, game.game_score(), 0, error=False)

        nstep = NStepAccumulator(self.n_step, self.gamma)
        experiences: list[Experience] = []
        tree_handle = None
        last_action = -1
        total_sims = 0
        total_cleared = 0
        last_total_visits = 0
        depth_sum = 0
        moves = 0
        action_dim = self.env_config.ACTION_DIM

        while not game.is_over():
            game_step = game.current_step
            if self._staged is not None:
                self._apply_staged()
                tree_handle, last_action = None, -1  # values in the old tree came from the old weights
            try:
                visits, tree_handle = run_mcts(
                    root_state=game,
                    network_interface=self.nn_evaluator,
                    config=self.mcts_config,
                    previous_tree_handle=tree_handle,
                    last_action=last_action,
                    seed=episode_seed + game_step,
                )
            except Exception as exc:
                logger.error("Worker %d: step %d: search failed: %s", self.actor_id, game_step, exc, exc_info=True)
                break
            # bootstrap for experiences whose n-step window closed on this state: V(s) at the root
            root_value = float(tree_handle.tree.root_value())
            for payload, ret in nstep.pop_ready(root_value):
                experiences.append((payload[0], payload[1], ret))
            last_total_visits = sum(visits.values())
            total_sims += last_total_visits
            depth_sum += tree_handle.max_depth_seen
            moves += 1
            self._send_event_async("mcts_step", float(last_total_visits), {"game_step": game_step})
            if not visits:
                logger.error("Worker %d: step %d: search returned no visits.", self.actor_id, game_step)
                break
            sel_temp = 1.0
            if self._explore_steps is not None:
                sel_temp = 1.0 if game_step < self._explore_steps else 0.1
            try:
                policy_target = get_policy_target_from_visits(visits, action_dim, temperature=1.0)
                action = select_action_from_visits(visits, temperature=sel_temp, rng=self._np_rng)
            except PolicyGenerationError as exc:
                logger.error("Worker %d: step %d: policy/action selection failed: %s", self.actor_id, game_step, exc)
                break
            try:
                feats = extract_state_features(game, self.model_config)
            except Exception as exc:
                logger.error("Worker %d: feature extraction error (step %d): %s", self.actor_id, game_step, exc, exc_info=True)
                break
            nstep.push((feats, policy_target))
            try:
                reward, done = game.step(action)
                cleared = game.get_last_cleared_triangles()
                total_cleared += cleared
                last_action = action
            except Exception as exc:
                logger.error("Worker %d: game step error (action %d, step %d): %s", self.actor_id, action, game_step, exc, exc_info=True)
                break
            nstep.add_reward(reward)
            self._send_event_async("step_reward", reward, {"game_step": game_step})
            if cleared > 0:
                self._send_event_async("triangles_cleared_step", cleared, {"game_step": game_step})
            self._send_event_async("current_score", game.game_score(), {"game_step": game_step})
            if done:
                break

        for payload, ret in nstep.flush():
            experiences.append((payload[0], payload[1], ret))
        final_score = game.game_score()
        final_step = game.current_step
        ctx = {
            "score": final_score,
            "length": final_step,
            "simulations": total_sims,
            "triangles_cleared": total_cleared,
            "trainer_step": step_at_start,
        }
        self._send_event_async("episode_end", 1.0, ctx)
        return SelfPlayResult(
            episode_experiences=experiences,
            final_score=final_score,
            episode_steps=final_step,
            trainer_step_at_episode_start=step_at_start,
            total_simulations=total_sims,
            avg_root_visits=float(last_total_visits),
            avg_tree_depth=float(depth_sum / moves) if moves else 0.0,
            context=ctx,
        )
