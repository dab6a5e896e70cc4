"""Multi-rank behaviour that a plain command must deliver (CPU, gloo): self-launch from ``--gpus N``,
readiness agreement of data-parallel ranks, a rank that disappears, the asynchronous split layout."""

from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
from alphatriangle_amd.parallel.dist import init_distributed

ROOT = Path(__file__).resolve().parents[1]


def _clean_env() -> dict[str, str]:
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_bench_gpus_2_starts_two_ranks_itself_and_prints_one_line():
    res = subprocess.run(
        [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--engine", "cpu", "--steps", "2", "--warmup", "1", "--games", "2",
         "--sims", "4", "--train-steps", "1"],
        cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0
    assert out["config"]["global_batch"] % 2 == 0 and out["vs_baseline"] is None
    # the whole-job value counts both ranks' games
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] / 1e3)) / out["value"] < 0.02


def test_bench_split_layout_three_ranks_prints_one_line_with_the_learner_figures():
    """`bench.py --gpus 3 --layout split --engine cpu`: two actor ranks + one learner rank over gloo, one JSON line whose
    value is the summed moves/s of the ACTOR ranks and which carries the learner's own figures."""
    res = subprocess.run(
        [sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--layout", "split", "--engine", "cpu", "--steps", "8", "--warmup", "2",
         "--games", "2", "--sims", "4", "--batch", "4"],
        cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["steps"] == 8 and out["warmup"] == 2 and out["backend"] == "gloo" and out["ranks_seen"] == 3
    assert out["config"]["actor_ranks"] == 2 and out["config"]["learner_ranks"] == 1 and out["config"]["parallelism"].startswith("split")
    # two actors x two games per timed move of the slowest actor
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] / 1e3)) / out["value"] < 0.02
    assert out["learner_steps"] > 0 and out["trainer_samples_per_sec"] > 0 and 0 < out["learner_busy_fraction"] <= 1
    assert out["mean_weight_version_lag"] >= 0 and out["actors_retired"] == 0


def test_bench_refuses_a_gpus_flag_that_contradicts_the_launcher():
    env = _clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--engine", "cpu"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 2
    assert "does not match WORLD_SIZE" in res.stdout


def test_launcher_reports_a_failing_rank_and_stops_the_others():
    from alphatriangle_amd.parallel.launch import launch_local_ranks

    code = "import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(60)\n"
    t0 = time.monotonic()
    rc = launch_local_ranks(2, [sys.executable, "-c", code], grace_s=5.0)
    assert rc == 7 and time.monotonic() - t0 < 30


def test_train_command_starts_its_ranks_from_gpus(tmp_path):
    """`train --gpus 2` on the CPU engine: two ranks, data-parallel learner over gloo, both exit 0."""
    res = subprocess.run(
        [sys.executable, "-m", "alphatriangle_amd", "train", "--gpus", "2", "--preset", "tiny", "--engine", "cpu", "--max-steps", "6",
         "--dist-backend", "gloo", "--data-root", str(tmp_path), "--run-name", "two", "--log-level", "WARNING"],
        cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    runs = tmp_path / "AlphaTriangle" / "runs"
    assert (runs / "two" / "checkpoints" / "latest.pkl").exists() and (runs / "two_rank1" / "checkpoints" / "latest.pkl").exists()


# ----------------------------------------------------------------------------------------------------------------------
def _tiny():
    env = EnvConfig(ROWS=3, COLS=3, PLAYABLE_RANGE_PER_ROW=[(0, 3)] * 3, NUM_SHAPE_SLOTS=1)
    mc = ModelConfig(CONV_FILTERS=[4], CONV_KERNEL_SIZES=[3], CONV_STRIDES=[1], CONV_PADDING=[1], NUM_RESIDUAL_BLOCKS=0,
                     USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0, FC_DIMS_SHARED=[8], POLICY_HEAD_DIMS=[8], VALUE_HEAD_DIMS=[8],
                     OTHER_NN_INPUT_FEATURES_DIM=14, USE_BATCH_NORM=False)
    mcts = AlphaTriangleMCTSConfig(max_simulations=4, max_depth=3, mcts_batch_size=2)
    return env, mc, mcts


def _spawn(fn, world, *args):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(fn, args=(world, f"file://{d}/store", *args), nprocs=world, join=True)


def _run_rank(rank, world, init, root, tc_update, sys_update, layout="colocated", timeout_s=60.0):
    from alphatriangle_amd.stats import TrieyeConfig
    from alphatriangle_amd.training.runner import run_training

    torch.set_num_threads(1)
    ctx = init_distributed(rank, world, rank, backend="gloo", layout=layout, num_learners=1, init_method=init,
                           timeout_s=timeout_s, control_timeout_s=timeout_s)
    env, mc, mcts = _tiny()
    tc = TrainConfig(BATCH_SIZE=8, BUFFER_CAPACITY=400, MIN_BUFFER_SIZE_TO_TRAIN=8, DEVICE="cpu", MAX_TRAINING_STEPS=6, USE_PER=False,
                     LEARNING_RATE=1e-2, WORKER_UPDATE_FREQ_STEPS=2, CHECKPOINT_SAVE_FREQ_STEPS=100, NUM_SELF_PLAY_WORKERS=1,
                     N_STEP_RETURNS=2, AUTO_RESUME_LATEST=False, RANDOM_SEED=11 + rank).model_copy(update=tc_update(rank))
    sysc = SystemConfig(ENGINE="cpu", ROLE_LAYOUT=layout, DIST_BACKEND="gloo", CONTROL_TIMEOUT_S=timeout_s,
                        COLLECTIVE_TIMEOUT_S=timeout_s).model_copy(update=sys_update(rank))
    run = "r0" if rank == 0 else f"r0_rank{rank}"
    tcfg = TrieyeConfig(app_name="t", run_name=run)
    tcfg.persistence.ROOT_DATA_DIR = root
    tcfg.persistence.RUN_NAME = run
    code = run_training("WARNING", tc, tcfg, False, model_config_override=mc, mcts_config_override=mcts, env_config_override=env,
                        system_config_override=sysc, dist_context=ctx)
    return code, ctx


def _late_ready(rank, world, init, root):
    # rank 1 needs five times as many experiences before it may train: without the cross-rank agreement rank 0 would
    # enter the gradient all-reduce alone and hang until the collective time-out
    code, ctx = _run_rank(rank, world, init, root, lambda r: {"MIN_BUFFER_SIZE_TO_TRAIN": 8 if r == 0 else 40}, lambda r: {})
    assert code == 0, f"rank {rank} exited with {code}"
    from alphatriangle_amd.stats import Serializer

    ck = Serializer().load_checkpoint(Path(root) / "t" / "runs" / ("r0" if rank == 0 else f"r0_rank{rank}") / "checkpoints" / "latest.pkl")
    assert ck.global_step == 6  # both ranks took exactly the agreed six steps
    ctx.shutdown()


def test_colocated_data_parallel_ranks_agree_on_readiness(tmp_path):
    _spawn(_late_ready, 2, str(tmp_path))


def _one_rank_has_no_batch(rank, world, init, root):
    """Rank 1's replay refuses to hand out its third and fourth batch.  The ranks agreed on their step counts before, so
    rank 0 is already waiting in the gradient all-reduce of those steps: rank 1 has to join them with a zero gradient."""
    if rank == 1:
        from alphatriangle_amd.rl.core import buffer as buffer_mod

        orig = buffer_mod.ExperienceBuffer.sample
        calls = {"n": 0}

        def sample(self, *a, **k):  # noqa: ANN001, ANN002, ANN003
            calls["n"] += 1
            return None if calls["n"] in (3, 4) else orig(self, *a, **k)

        buffer_mod.ExperienceBuffer.sample = sample
    t0 = time.monotonic()
    code, ctx = _run_rank(rank, world, init, root, lambda r: {"MAX_TRAINING_STEPS": 8}, lambda r: {}, timeout_s=30.0)
    assert code == 0, f"rank {rank} exited with {code}"
    assert time.monotonic() - t0 < 60, "a rank without a batch must not strand the others in the all-reduce"
    from alphatriangle_amd.stats import Serializer

    ck = Serializer().load_checkpoint(Path(root) / "t" / "runs" / ("r0" if rank == 0 else f"r0_rank{rank}") / "checkpoints" / "latest.pkl")
    assert ck.global_step == 8, f"rank {rank} stopped at step {ck.global_step}"  # both ranks counted the same eight optimizer steps
    # the replicas went through the same eight updates: identical parameters without any broadcast
    flat = torch.cat([v.reshape(-1).float() for v in ck.model_state_dict.values() if v.dtype.is_floating_point])
    other = flat.clone()
    ctx.broadcast_weights(other, src=0)
    assert torch.allclose(other, flat, atol=1e-7), f"rank {rank}: replicas differ by {(other - flat).abs().max().item():.3e}"
    ctx.shutdown()


def test_a_data_parallel_rank_without_a_batch_still_enters_the_all_reduce(tmp_path):
    _spawn(_one_rank_has_no_batch, 2, str(tmp_path))


def _shutdown_ends_the_group_threads(rank, world, init, root):
    """After a run that went through an optimizer step (torch imports torch._dynamo and, with it,
    torch.distributed.nn.functional on the way), shutdown() has to end the process group's worker threads: a group that
    stays pinned keeps them alive into interpreter shutdown, where a gloo worker dropping a tensor aborts the process."""
    code, ctx = _run_rank(rank, world, init, root, lambda r: {}, lambda r: {})
    assert code == 0
    before = len(os.listdir("/proc/self/task"))
    ctx.shutdown()
    import gc

    gc.collect()
    after = len(os.listdir("/proc/self/task"))
    assert after < before, f"rank {rank}: {before} native threads before shutdown, {after} after: the process group is still alive"
    fn_mod = sys.modules.get("torch.distributed.nn.functional")
    assert fn_mod is not None and fn_mod.all_reduce.__defaults__[-1] is None  # imported before the group existed


def test_shutdown_releases_the_process_group_and_its_threads(tmp_path):
    _spawn(_shutdown_ends_the_group_threads, 2, str(tmp_path))


def _one_rank_leaves(rank, world, init, root):
    if rank == 1:
        # takes part in set-up and three rounds, then disappears without a word
        from alphatriangle_amd.training import loop as loop_mod

        orig = loop_mod.TrainingLoop._negotiate
        calls = {"n": 0}

        def negotiate(self):  # noqa: ANN001
            calls["n"] += 1
            if calls["n"] > 3:
                os._exit(0)
            return orig(self)

        loop_mod.TrainingLoop._negotiate = negotiate
    t0 = time.monotonic()
    code, _ = _run_rank(rank, world, init, root, lambda r: {"MAX_TRAINING_STEPS": 1000}, lambda r: {}, timeout_s=20.0)
    took = time.monotonic() - t0
    assert code == 1, "a run that lost a rank must not report success"
    assert took < 60, f"the surviving rank needed {took:.0f} s to notice"
    assert (Path(root) / "t" / "runs" / "r0" / "checkpoints" / "latest.pkl").exists(), "the survivor saves before leaving"
    os._exit(0)  # the process group cannot be torn down cleanly any more


def test_a_rank_that_disappears_ends_the_others_nonzero_within_a_minute(tmp_path):
    _spawn(_one_rank_leaves, 2, str(tmp_path))


# ----------------------------------------------------------------------------------------------------------------------
def _async_split(rank, world, init, root, kill_rank):
    if rank == kill_rank:
        # this actor takes part in set-up, plays two episodes and then dies without confirming anything
        from alphatriangle_amd.training import split_loop

        orig = split_loop.HostExperienceSource.produce
        calls = {"n": 0}

        def produce(self):  # noqa: ANN001
            calls["n"] += 1
            if calls["n"] > 2:
                os._exit(0)
            return orig(self)

        split_loop.HostExperienceSource.produce = produce
    code, ctx = _run_rank(rank, world, init, root, lambda r: {"MAX_TRAINING_STEPS": 40, "WORKER_UPDATE_FREQ_STEPS": 5},
                          lambda r: {"TRAIN_STEPS_PER_ROUND": 2, "ACTOR_SILENCE_S": 3.0, "NUM_LEARNER_RANKS": 1}, layout="split",
                          timeout_s=30.0)
    assert code == 0, f"rank {rank} exited with {code}"
    if rank == 0:
        seen = json.loads((Path(root) / "t" / "runs" / "r0" / "role_summary.json").read_text())
        assert seen["role"] == "learner" and seen["learner_steps"] == 40 and seen["rows_ingested"] > 0 and seen["complete"]
        # the learner is not tied to the actors' pace: it polls and trains in rounds of its own (two steps each here)
        assert seen["rounds"] >= 20 and 0 < seen["learner_busy_fraction"] <= 1
        assert seen["actors_retired"] == ([kill_rank] if kill_rank is not None else [])
    elif rank != kill_rank:
        seen = json.loads((Path(root) / "t" / "runs" / f"r0_rank{rank}" / "role_summary.json").read_text())
        assert seen["role"] == "actor" and seen["moves_played"] > 0 and seen["weight_versions"] >= 1
    if kill_rank is not None:
        os._exit(0)
    ctx.shutdown()


def test_async_split_layout_three_ranks(tmp_path):
    _spawn(_async_split, 3, str(tmp_path), None)


def test_async_split_layout_survives_a_dead_actor(tmp_path):
    """Degraded mode: actor rank 2 dies after two episodes; the learner retires it and finishes with actor rank 1."""
    _spawn(_async_split, 3, str(tmp_path), 2)
