"""The real TrainingLoop with everything around it replaced by light stand-ins
(same strategy as the reference's tests/training/test_loop_integration.py), plus a real
end-to-end run with save + resume on the CPU engine."""

from __future__ import annotations

from unittest.mock import MagicMock

import numpy as np
import pytest

from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
from alphatriangle_amd.presets import build_preset
from alphatriangle_amd.rl import ExperienceBuffer, SelfPlayResult, Trainer
from alphatriangle_amd.stats import Serializer, StatsActor, TrieyeConfig
from alphatriangle_amd.training import TrainingComponents, TrainingLoop, run_training
from alphatriangle_amd.training.worker_manager import WorkerDiedError, WorkerManager
from tests.conftest import make_experience


class FakeManager:
    """One finished episode per poll; records weight pushes."""

    def __init__(self, components):
        self.components = components
        self.weight_updates: list[int] = []
        self.submitted = 0
        rng = np.random.default_rng(0)
        self.exps = [make_experience(components.model_config, components.env_config, rng) for _ in range(3)]

    def initialize_workers(self):
        pass

    def submit_initial_tasks(self):
        self.submitted += 2

    def submit_task(self, idx):
        self.submitted += 1

    def get_completed_tasks(self, timeout=0.0):
        return [(0, SelfPlayResult(episode_experiences=list(self.exps), final_score=1.0, episode_steps=3,
                                   trainer_step_at_episode_start=0, total_simulations=24, avg_root_visits=8.0, avg_tree_depth=1.0))]

    def update_worker_networks(self, step):
        self.weight_updates.append(step)

    def get_num_active_workers(self):
        return 2

    def get_num_pending_tasks(self):
        return 2

    def cleanup_actors(self):
        pass


@pytest.fixture
def components(mock_model_config, mock_env_config, mock_mcts_config, tmp_path):
    tc = TrainConfig(BATCH_SIZE=4, BUFFER_CAPACITY=100, MIN_BUFFER_SIZE_TO_TRAIN=4, USE_PER=True, DEVICE="cpu", MAX_TRAINING_STEPS=20,
                     WORKER_UPDATE_FREQ_STEPS=5, CHECKPOINT_SAVE_FREQ_STEPS=10, NUM_SELF_PLAY_WORKERS=2)
    buffer = MagicMock(spec=ExperienceBuffer)
    buffer.capacity, buffer.min_size_to_train = 100, 4
    buffer.is_ready.return_value = True
    buffer.__len__.return_value = 50
    buffer.sample.return_value = {"batch": [1], "indices": np.zeros(4, dtype=np.int32), "weights": np.ones(4, dtype=np.float32)}
    buffer._calculate_beta.return_value = 0.5
    trainer = MagicMock(spec=Trainer)
    trainer.train_step.return_value = ({"total_loss": 1.0, "policy_loss": 0.6, "value_loss": 0.4, "entropy": 0.1, "mean_td_error": 0.2}, np.ones(4))
    trainer.get_current_lr.return_value = 1e-3
    trainer.optimizer = MagicMock()
    trainer.optimizer.state_dict.return_value = {"state": {}, "param_groups": []}
    actor = MagicMock(spec=StatsActor)
    actor.get_actor_name.return_value = "mock"
    actor.get_run_base_dir_str.return_value = str(tmp_path)
    nn = MagicMock()
    nn.get_weights.return_value = {"w": 1}
    tcfg = TrieyeConfig(app_name="T", run_name="loop")
    tcfg.persistence.SAVE_BUFFER = False
    return TrainingComponents(nn=nn, buffer=buffer, trainer=trainer, trieye_actor=actor, trieye_config=tcfg, serializer=Serializer(),
                              train_config=tc, env_config=mock_env_config, model_config=mock_model_config, mcts_config=mock_mcts_config,
                              profile_workers=False)


def test_loop_cadences(components, monkeypatch):
    monkeypatch.setattr("alphatriangle_amd.training.loop.WorkerManager", FakeManager)
    loop = TrainingLoop(components)
    loop.set_initial_state(0, 0, 0)
    loop.initialize_workers()
    loop.run()
    assert loop.training_complete and loop.global_step == 20 and loop.training_exception is None
    mgr = loop.worker_manager
    assert mgr.weight_updates == [5, 10, 15, 20] and loop.weight_update_count == 4
    events = [c.args[0] for c in components.trieye_actor.log_event.call_args_list]
    wu = [e for e in events if e.name == "Progress/Weight_Updates_Total"]
    assert [e.value for e in wu] == [1, 2, 3, 4] and wu[-1].global_step == 20
    names = {e.name for e in events}
    assert {"step_completed", "PER/Beta", "Buffer/Size", "System/Num_Active_Workers", "System/Num_Pending_Tasks",
            "Progress/Episodes_Played", "Progress/Total_Simulations"} <= names
    batch_names = {e.name for c in components.trieye_actor.log_batch_events.call_args_list for e in c.args[0]}
    assert batch_names == {"Loss/Total", "Loss/Policy", "Loss/Value", "Loss/Entropy", "Loss/Mean_Abs_TD_Error", "LearningRate"}
    saves = components.trieye_actor.save_training_state.call_args_list
    assert [c.kwargs["global_step"] for c in saves] == [10, 20]
    assert all(c.kwargs["is_best"] is False and c.kwargs["save_buffer"] is False for c in saves)
    assert components.buffer.update_priorities.call_count == 20 and components.buffer.add_batch.call_count >= 20
    assert loop.episodes_played >= 20 and loop.total_simulations_run == 24 * loop.episodes_played
    assert components.trieye_actor.process_and_log.call_count >= 20


def test_loop_resume_counters_and_stop(components, monkeypatch):
    monkeypatch.setattr("alphatriangle_amd.training.loop.WorkerManager", FakeManager)
    loop = TrainingLoop(components)
    loop.set_initial_state(12, 7, 300)
    assert loop.weight_update_count == 2 and loop.episodes_played == 7
    loop.request_stop()
    loop.run()
    assert not loop.training_complete and loop.global_step == 12


def test_first_interrupt_is_a_stop_request_and_the_round_finishes(components, monkeypatch):
    """SIGINT while the loop runs: no exception, the current round completes, the loop ends through its normal exit."""
    import os
    import signal

    monkeypatch.setattr("alphatriangle_amd.training.loop.WorkerManager", FakeManager)
    components.train_config.MAX_TRAINING_STEPS = 10_000
    loop = TrainingLoop(components)
    calls = {"n": 0}
    real = components.trainer.train_step.return_value

    def step(*_a, **_k):
        calls["n"] += 1
        if calls["n"] == 7:
            os.kill(os.getpid(), signal.SIGINT)  # delivered to the main thread, i.e. inside this round
        return real

    components.trainer.train_step.side_effect = step
    before = signal.getsignal(signal.SIGINT)
    loop.run()
    assert signal.getsignal(signal.SIGINT) is before, "the loop must restore the previous handler"
    assert loop.training_exception is None and not loop.training_complete
    assert loop.global_step == 7 and components.buffer.update_priorities.call_count == 7  # the interrupted round was completed


def test_loop_survives_training_exception(components, monkeypatch):
    monkeypatch.setattr("alphatriangle_amd.training.loop.WorkerManager", FakeManager)
    components.trainer.train_step.side_effect = RuntimeError("boom")
    loop = TrainingLoop(components)
    loop.run()
    assert isinstance(loop.training_exception, RuntimeError) and not loop.training_complete


def test_worker_manager_retires_dead_workers(components, monkeypatch):
    class W:
        def __init__(self, actor_id, **_):
            self.id, self.weights = actor_id, []

        def run_episode(self):
            if self.id == 1:
                raise WorkerDiedError("gone")
            return SelfPlayResult(episode_experiences=[], final_score=0, episode_steps=0, trainer_step_at_episode_start=0,
                                  total_simulations=0, avg_root_visits=0, avg_tree_depth=0)

        def set_weights(self, w):
            self.weights.append(w)

        def set_current_trainer_step(self, s):
            self.step = s

        def stage_weights(self, w, s):
            self.weights.append(w)

    monkeypatch.setattr("alphatriangle_amd.training.worker_manager.SelfPlayWorker", W)
    mgr = WorkerManager(components)
    mgr.initialize_workers()
    assert mgr.get_num_active_workers() == 2
    mgr.submit_initial_tasks()
    done = []
    for _ in range(20):
        done += mgr.get_completed_tasks(0.5)
        if len(done) >= 2:
            break
    kinds = {i: type(r).__name__ for i, r in done}
    assert kinds == {0: "SelfPlayResult", 1: "WorkerDiedError"} and mgr.get_num_active_workers() == 1
    mgr.submit_task(1)  # retired worker is not resubmitted
    assert mgr.get_num_pending_tasks() == 0
    mgr.update_worker_networks(5)
    assert mgr.workers[0].weights == [{"w": 1}] and mgr.workers[0].step == 5
    mgr.cleanup_actors()


def test_end_to_end_cpu_training_with_resume(tmp_path):
    """BASELINE configuration 1: tiny 5x5 grid, 2-layer CNN, 8 simulations, one CPU worker."""
    env, model, mcts, train, system = build_preset("tiny")
    train = train.model_copy(update={"MAX_TRAINING_STEPS": 30, "CHECKPOINT_SAVE_FREQ_STEPS": 10, "LR_SCHEDULER_T_MAX": 30, "PER_BETA_ANNEAL_STEPS": 30})
    t1 = TrieyeConfig(app_name="E2E", run_name="a")
    t1.persistence.ROOT_DATA_DIR = str(tmp_path)
    assert run_training("WARNING", train, t1, profile=False, model_config_override=model, mcts_config_override=mcts,
                        env_config_override=env, system_config_override=system) == 0
    run = tmp_path / "E2E" / "runs" / "a"
    assert (run / "checkpoints" / "latest.pkl").exists() and (run / "buffers" / "buffer.pkl").exists()
    assert (run / "configs.json").exists() and (run / "metrics.jsonl").stat().st_size > 0
    ck = Serializer().load_checkpoint(run / "checkpoints" / "latest.pkl")
    assert ck.global_step == 30 and ck.episodes_played > 0 and "conv_body.0.0.weight" in ck.model_state_dict
    # second run resumes at step 30 and continues to 45
    train2 = train.model_copy(update={"MAX_TRAINING_STEPS": 45, "LR_SCHEDULER_T_MAX": 45, "PER_BETA_ANNEAL_STEPS": 45})
    t2 = TrieyeConfig(app_name="E2E", run_name="b")
    t2.persistence.ROOT_DATA_DIR = str(tmp_path)
    assert run_training("WARNING", train2, t2, profile=False, model_config_override=model, mcts_config_override=mcts,
                        env_config_override=env, system_config_override=system) == 0
    ck2 = Serializer().load_checkpoint(tmp_path / "E2E" / "runs" / "b" / "checkpoints" / "latest.pkl")
    assert ck2.global_step == 45 and ck2.episodes_played >= ck.episodes_played


def test_resume_with_another_architecture_starts_fresh(tmp_path):
    """Auto-resume finds the latest run under the data root; when that run's weights do not fit the configured
    model the new run starts from scratch instead of failing (and copies nothing from the old weights)."""
    env, model, mcts, train, system = build_preset("tiny")
    train = train.model_copy(update={"MAX_TRAINING_STEPS": 10, "CHECKPOINT_SAVE_FREQ_STEPS": 10, "LR_SCHEDULER_T_MAX": 10, "PER_BETA_ANNEAL_STEPS": 10})
    t1 = TrieyeConfig(app_name="E2E", run_name="a")
    t1.persistence.ROOT_DATA_DIR = str(tmp_path)
    assert run_training("WARNING", train, t1, profile=False, model_config_override=model, mcts_config_override=mcts,
                        env_config_override=env, system_config_override=system) == 0
    wider = model.model_copy(update={"FC_DIMS_SHARED": [model.FC_DIMS_SHARED[0] * 2]})
    t2 = TrieyeConfig(app_name="E2E", run_name="b")
    t2.persistence.ROOT_DATA_DIR = str(tmp_path)
    shorter = train.model_copy(update={"MAX_TRAINING_STEPS": 6, "CHECKPOINT_SAVE_FREQ_STEPS": 6, "LR_SCHEDULER_T_MAX": 6, "PER_BETA_ANNEAL_STEPS": 6})
    assert run_training("WARNING", shorter, t2, profile=False, model_config_override=wider, mcts_config_override=mcts,
                        env_config_override=env, system_config_override=system) == 0
    ck = Serializer().load_checkpoint(tmp_path / "E2E" / "runs" / "b" / "checkpoints" / "latest.pkl")
    assert ck.global_step == 6  # counted from zero: a run resumed at run a's step 10 would have ended there
    first = next(iter(ck.model_state_dict))
    assert ck.model_state_dict["shared_fc.0.weight"].shape[0] == wider.FC_DIMS_SHARED[0], first
