"""Self-play pieces: visit-count helpers, n-step returns, the CPU worker and the worker manager."""

import numpy as np
import pytest

from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
from alphatriangle_amd.rl import SelfPlayResult, SelfPlayWorker
from alphatriangle_amd.rl.self_play import NStepAccumulator, PolicyGenerationError, get_policy_target_from_visits, select_action_from_visits


def test_select_action_from_visits():
    rng = np.random.default_rng(0)
    assert select_action_from_visits({3: 10, 5: 1}, 0.0, rng) == 3
    assert select_action_from_visits({3: 5, 5: 5}, 0.0, rng) in (3, 5)
    picks = [select_action_from_visits({1: 90, 2: 10}, 1.0, rng) for _ in range(300)]
    assert 0.8 < picks.count(1) / 300 < 0.97
    cold = [select_action_from_visits({1: 60, 2: 40}, 0.1, rng) for _ in range(200)]
    assert cold.count(1) > 190
    assert select_action_from_visits({4: 0, 6: 0}, 1.0, rng) in (4, 6)
    with pytest.raises(PolicyGenerationError):
        select_action_from_visits({}, 1.0, rng)


def test_policy_target_from_visits():
    t = get_policy_target_from_visits({0: 3, 2: 1}, 5, 1.0)
    assert list(t.keys()) == [0, 1, 2, 3, 4] and abs(t[0] - 0.75) < 1e-9 and t[1] == 0.0 and abs(sum(t.values()) - 1) < 1e-9
    t0 = get_policy_target_from_visits({0: 3, 2: 3, 1: 1}, 5, 0.0)
    assert t0[0] == 0.5 and t0[2] == 0.5 and t0[1] == 0.0
    assert sum(get_policy_target_from_visits({}, 4).values()) == 0.0
    assert sum(get_policy_target_from_visits({0: 0}, 4).values()) == 0.0
    sharp = get_policy_target_from_visits({0: 2, 1: 1}, 2, 0.5)
    assert abs(sharp[0] - 0.8) < 1e-9


def test_nstep_accumulator_matches_definition():
    n, gamma = 3, 0.9
    acc = NStepAccumulator(n, gamma)
    rewards = [1.0, 2.0, 3.0, 4.0, 5.0]
    values = [10.0, 20.0, 30.0, 40.0, 50.0]  # V(s_t) known when move t starts
    out = []
    for t in range(5):
        out += acc.pop_ready(values[t])
        acc.push(t)
        acc.add_reward(rewards[t])
    out += acc.flush()
    got = dict(out)
    assert set(got) == {0, 1, 2, 3, 4}  # every experience exactly once
    assert abs(got[0] - (1 + 0.9 * 2 + 0.81 * 3 + 0.729 * values[3])) < 1e-9
    assert abs(got[1] - (2 + 0.9 * 3 + 0.81 * 4 + 0.729 * values[4])) < 1e-9
    assert abs(got[2] - (3 + 0.9 * 4 + 0.81 * 5)) < 1e-9  # episode ended: no bootstrap
    assert abs(got[4] - 5.0) < 1e-9


@pytest.fixture(scope="module")
def tiny():
    env = EnvConfig(ROWS=4, COLS=5, PLAYABLE_RANGE_PER_ROW=[(0, 5)] * 4, NUM_SHAPE_SLOTS=2)
    mc = ModelConfig(CONV_FILTERS=[8], CONV_KERNEL_SIZES=[3], CONV_STRIDES=[1], CONV_PADDING=[1], NUM_RESIDUAL_BLOCKS=0,
                     USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0, FC_DIMS_SHARED=[16], POLICY_HEAD_DIMS=[16], VALUE_HEAD_DIMS=[16],
                     OTHER_NN_INPUT_FEATURES_DIM=22)
    tc = TrainConfig(BATCH_SIZE=8, BUFFER_CAPACITY=500, MIN_BUFFER_SIZE_TO_TRAIN=16, MAX_TRAINING_STEPS=40, N_STEP_RETURNS=3,
                     NUM_SELF_PLAY_WORKERS=2, DEVICE="cpu", WORKER_UPDATE_FREQ_STEPS=5, CHECKPOINT_SAVE_FREQ_STEPS=20)
    mcts = AlphaTriangleMCTSConfig(max_simulations=8, max_depth=4, mcts_batch_size=4)
    return env, mc, tc, mcts


class _Sink:
    def __init__(self):
        self.events = []

    def log_event(self, ev):
        self.events.append(ev)
        return True

    def get_actor_name(self):
        return "sink"

    def get_run_base_dir_str(self):
        return None


def test_worker_runs_an_episode(tiny, tmp_path):
    env, mc, tc, mcts = tiny
    sink = _Sink()
    w = SelfPlayWorker(0, env, mcts, mc, tc, stats_actor=sink, run_base_dir=str(tmp_path), seed=5, profile_this_worker=True)
    w.set_current_trainer_step(7)
    res = w.run_episode()
    assert isinstance(res, SelfPlayResult) and res.episode_steps > 0 and res.trainer_step_at_episode_start == 7
    assert len(res.episode_experiences) == res.episode_steps  # one experience per move, none duplicated
    assert res.total_simulations >= 8 * res.episode_steps
    for state, policy, ret in res.episode_experiences:
        assert state["grid"].shape == (1, 4, 5) and state["other_features"].shape == (22,)
        assert len(policy) == env.ACTION_DIM and abs(sum(policy.values()) - 1.0) < 1e-6 and np.isfinite(ret)
    # last experience: no bootstrap, its return is exactly the reward of the terminal move (placement / clearing reward
    # plus the game-over penalty that replaces the alive bonus)
    last_reward = [e.value for e in sink.events if e.name == "step_reward"][-1]
    assert abs(res.episode_experiences[-1][2] - last_reward) < 1e-6
    assert last_reward < 0.0
    names = {e.name for e in sink.events}
    assert {"mcts_step", "step_reward", "current_score", "episode_end"} <= names
    end = [e for e in sink.events if e.name == "episode_end"][0]
    assert end.global_step == 7 and set(end.context) >= {"score", "length", "simulations", "triangles_cleared", "trainer_step"}
    assert list((tmp_path / "profile_data").glob("worker_0_ep_*.prof"))
    # weights staged during an episode are applied at the next move boundary
    w.stage_weights(w.nn_evaluator.get_weights(), 9)
    w.run_episode()
    assert w.current_trainer_step == 9


def test_worker_survives_search_failure(tiny, monkeypatch):
    env, mc, tc, mcts = tiny
    w = SelfPlayWorker(1, env, mcts, mc, tc, seed=1)

    def boom(**_):
        raise RuntimeError("search exploded")

    monkeypatch.setattr("alphatriangle_amd.rl.self_play.worker.run_mcts", boom)
    res = w.run_episode()
    assert res.episode_experiences == [] and res.episode_steps == 0


def test_self_play_result_drops_malformed(mock_experience):
    bad = ({"grid": np.array([[np.nan]]), "other_features": np.zeros(2)}, {0: 1.0}, 0.5)
    r = SelfPlayResult(episode_experiences=[mock_experience, bad, "junk"], final_score=1.0, episode_steps=2,
                       trainer_step_at_episode_start=0, total_simulations=4, avg_root_visits=2.0, avg_tree_depth=1.0)
    assert len(r.episode_experiences) == 1
