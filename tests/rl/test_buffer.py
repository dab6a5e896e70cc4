"""Replay buffer, uniform and prioritized (mirrors the reference's tests/rl/test_buffer.py)."""

import numpy as np
import pytest

from alphatriangle_amd.config import TrainConfig
from alphatriangle_amd.rl import ExperienceBuffer
from alphatriangle_amd.utils import SumTree
from tests.conftest import make_experience


def _cfg(per: bool, **kw):
    base = dict(BATCH_SIZE=4, BUFFER_CAPACITY=100, MIN_BUFFER_SIZE_TO_TRAIN=10, USE_PER=per, MAX_TRAINING_STEPS=200, DEVICE="cpu")
    base.update(kw)
    return TrainConfig(**base)


@pytest.fixture
def exps(mock_model_config, mock_env_config):
    rng = np.random.default_rng(0)
    return [make_experience(mock_model_config, mock_env_config, rng) for _ in range(150)]


def test_uniform_buffer(exps):
    b = ExperienceBuffer(_cfg(False))
    assert len(b) == 0 and not b.is_ready() and not b.use_per
    b.add(exps[0])
    b.add_batch(exps[1:5])
    assert len(b) == 5 and b.sample(4) is None  # below the minimum size
    b.add_batch(exps[5:12])
    assert b.is_ready()
    s = b.sample(4)
    assert s is not None and len(s["batch"]) == 4 and s["indices"].shape == (4,) and np.all(s["weights"] == 1.0)
    assert s["indices"].dtype == np.int32 and s["weights"].dtype == np.float32
    b.update_priorities(s["indices"], np.ones(4))  # no-op
    b.add_batch(exps[12:150])
    assert len(b) == 100 and b.buffer[0] is exps[50]  # oldest evicted
    assert len(b.to_list()) == 100


def test_per_buffer_add_and_sample(exps):
    b = ExperienceBuffer(_cfg(True))
    assert isinstance(b.tree, SumTree) and b.tree.max_priority == 1.0
    b.add_batch(exps[:20])
    assert len(b) == 20 and abs(b.tree.total_priority - 20.0) < 1e-9  # inserted at max priority
    with pytest.raises(ValueError):
        b.sample(4)
    s = b.sample(4, current_train_step=10)
    assert s is not None and len(s["batch"]) == 4 and all(isinstance(e, tuple) for e in s["batch"])
    assert np.all((s["weights"] > 0) & (s["weights"] <= 1.0)) and abs(s["weights"].max() - 1.0) < 1e-6
    assert np.all(s["indices"] >= b.capacity - 1) and np.all(s["indices"] < 2 * b.capacity - 1)
    b.add_batch(exps[20:150])
    assert len(b) == 100


def test_per_priority_update_and_beta(exps):
    cfg = _cfg(True, PER_BETA_ANNEAL_STEPS=100)
    b = ExperienceBuffer(cfg)
    b.add_batch(exps[:20])
    s = b.sample(4, current_train_step=1)
    idx = np.array([s["indices"][0], s["indices"][1], s["indices"][1]], dtype=np.int32)
    errs = np.array([0.5, 0.1, 2.0])
    b.update_priorities(idx, errs)
    exp = (np.abs(errs) + cfg.PER_EPSILON) ** cfg.PER_ALPHA
    assert abs(b.tree.tree[idx[0]] - exp[0]) < 1e-9
    assert abs(b.tree.tree[idx[1]] - exp[2]) < 1e-9  # last write wins on duplicates
    assert abs(b.tree.max_priority - max(1.0, exp.max())) < 1e-9
    b.update_priorities(np.array([idx[0]]), np.array([np.nan]))
    assert np.isfinite(b.tree.tree[idx[0]]) and b.tree.tree[idx[0]] >= cfg.PER_EPSILON
    b.update_priorities(np.array([1, 2]), np.array([1.0]))  # length mismatch is ignored
    assert b._calculate_beta(0) == cfg.PER_BETA_INITIAL and b._calculate_beta(100) == cfg.PER_BETA_FINAL
    assert abs(b._calculate_beta(50) - 0.7) < 1e-9 and b._calculate_beta(1000) == cfg.PER_BETA_FINAL


def test_per_sampling_is_proportional(exps):
    b = ExperienceBuffer(_cfg(True))
    b.add_batch(exps[:50])
    heavy = b.capacity - 1 + 7
    b.update_priorities(np.array([heavy]), np.array([50.0]))
    hits = sum(int((b.sample(4, current_train_step=0)["indices"] == heavy).sum()) for _ in range(200))
    assert hits > 40  # ~ 200*4 * p(heavy) with p ~ 0.17


def test_list_roundtrip(exps):
    b = ExperienceBuffer(_cfg(True))
    b.add_batch(exps[:30])
    lst = b.to_list()
    b2 = ExperienceBuffer(_cfg(True))
    b2.load_list(lst)
    assert len(b2) == 30 and abs(b2.tree.total_priority - 30.0) < 1e-9
