"""Split role layout on one MI355X: a learner rank and a GPU-actor rank share the card (gloo control plane).

RCCL refuses two ranks on one device, so this rehearsal uses the gloo backend with host staging; what it
covers is the GPU-specific part of the split loop - the HBM outbox, record packing, ingestion into the
learner's device replay, weight-wire unpacking and the engine re-pack on the actor.
"""

from __future__ import annotations

import os
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _rank_main(rank, world, init, root):
    from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
    from alphatriangle_amd.parallel.dist import init_distributed
    from alphatriangle_amd.stats import TrieyeConfig
    from alphatriangle_amd.training.runner import run_training

    torch.set_num_threads(2)
    ctx = init_distributed(rank, world, 0, backend="gloo", layout="split", num_learners=1, init_method=init, timeout_s=180)
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=64, BUFFER_CAPACITY=20_000, MIN_BUFFER_SIZE_TO_TRAIN=256, MAX_TRAINING_STEPS=8,
                     WORKER_UPDATE_FREQ_STEPS=4, CHECKPOINT_SAVE_FREQ_STEPS=1000, N_STEP_RETURNS=2, AUTO_RESUME_LATEST=False)
    sysc = SystemConfig(ENGINE="gpu", ROLE_LAYOUT="split", NUM_LEARNER_RANKS=1, TRAIN_STEPS_PER_ROUND=4, SHIP_CAPACITY=1024,
                        GAMES_PER_GPU=128, DIST_BACKEND="gloo", REPLAY_HBM_BUDGET_GB=4.0)
    mcts = AlphaTriangleMCTSConfig(max_simulations=8, max_depth=4)
    run = "gsplit" if rank == 0 else f"gsplit_rank{rank}"
    tcfg = TrieyeConfig(app_name="t", run_name=run)
    tcfg.persistence.ROOT_DATA_DIR = root
    tcfg.persistence.RUN_NAME = run
    code = run_training("WARNING", tc, tcfg, False, model_config_override=mc, mcts_config_override=mcts, env_config_override=env,
                        system_config_override=sysc, dist_context=ctx)
    assert code == 0, f"rank {rank} exited with {code}"
    ctx.shutdown()


def test_split_roles_gpu_actor_and_gpu_learner_share_one_card():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory() as root, tempfile.TemporaryDirectory() as d:
        mp.spawn(_rank_main, args=(2, f"file://{d}/store", root), nprocs=2, join=True)
