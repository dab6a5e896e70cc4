"""Shared fixtures.  Tiny world mirroring the reference's test fixtures (tests/conftest.py:30-125 there):
3x3 grid, one shape slot, a one-conv network, uniform buffer, 8-simulation search."""

from __future__ import annotations

import random

import numpy as np
import pytest
import torch

from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, TrainConfig

torch.set_num_threads(2)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def mock_env_config() -> EnvConfig:
    return EnvConfig(ROWS=3, COLS=3, PLAYABLE_RANGE_PER_ROW=[(0, 3)] * 3, NUM_SHAPE_SLOTS=1)


@pytest.fixture(scope="session")
def mock_model_config(mock_env_config: EnvConfig) -> ModelConfig:
    return ModelConfig(
        GRID_INPUT_CHANNELS=1,
        CONV_FILTERS=[4],
        CONV_KERNEL_SIZES=[3],
        CONV_STRIDES=[1],
        CONV_PADDING=[1],
        NUM_RESIDUAL_BLOCKS=0,
        RESIDUAL_BLOCK_FILTERS=4,
        USE_TRANSFORMER=False,
        TRANSFORMER_DIM=16,
        TRANSFORMER_HEADS=2,
        TRANSFORMER_LAYERS=0,
        TRANSFORMER_FC_DIM=32,
        FC_DIMS_SHARED=[8],
        POLICY_HEAD_DIMS=[8],
        VALUE_HEAD_DIMS=[8],
        OTHER_NN_INPUT_FEATURES_DIM=10,  # engine yields 14: exercises the truncate path like the reference fixture
        ACTIVATION_FUNCTION="ReLU",
        USE_BATCH_NORM=True,
    )


@pytest.fixture(scope="session")
def mock_train_config() -> TrainConfig:
    return TrainConfig(
        BATCH_SIZE=4,
        BUFFER_CAPACITY=100,
        MIN_BUFFER_SIZE_TO_TRAIN=10,
        USE_PER=False,
        DEVICE="cpu",
        RANDOM_SEED=42,
        NUM_SELF_PLAY_WORKERS=1,
        WORKER_DEVICE="cpu",
        WORKER_UPDATE_FREQ_STEPS=10,
        OPTIMIZER_TYPE="Adam",
        LEARNING_RATE=1e-3,
        WEIGHT_DECAY=0.0,
        LR_SCHEDULER_ETA_MIN=0.0,
        POLICY_LOSS_WEIGHT=1.0,
        VALUE_LOSS_WEIGHT=1.0,
        ENTROPY_BONUS_WEIGHT=0.0,
        CHECKPOINT_SAVE_FREQ_STEPS=50,
        COMPILE_MODEL=False,
        MAX_TRAINING_STEPS=200,
        N_STEP_RETURNS=3,
        GAMMA=0.99,
    )


@pytest.fixture(scope="session")
def mock_mcts_config() -> AlphaTriangleMCTSConfig:
    return AlphaTriangleMCTSConfig(
        max_simulations=8, max_depth=5, cpuct=1.0, dirichlet_alpha=0.3, dirichlet_epsilon=0.25, discount=1.0, mcts_batch_size=4
    )


@pytest.fixture(scope="session")
def action_dim(mock_env_config: EnvConfig) -> int:
    return mock_env_config.ACTION_DIM


def make_state(model_config: ModelConfig, env_config: EnvConfig, rng: np.random.Generator):
    return {
        "grid": rng.random((model_config.GRID_INPUT_CHANNELS, env_config.ROWS, env_config.COLS), dtype=np.float32),
        "other_features": rng.random(model_config.OTHER_NN_INPUT_FEATURES_DIM, dtype=np.float32),
    }


def make_experience(model_config: ModelConfig, env_config: EnvConfig, rng: np.random.Generator):
    a = env_config.ACTION_DIM
    p = rng.random(a)
    p = p / p.sum()
    return (make_state(model_config, env_config, rng), {i: float(p[i]) for i in range(a)}, float(rng.uniform(-1, 1)))


@pytest.fixture
def mock_experience(mock_model_config, mock_env_config):
    return make_experience(mock_model_config, mock_env_config, np.random.default_rng(0))


@pytest.fixture
def seeded():
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
