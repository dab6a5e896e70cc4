"""Trainer (mirrors the reference's tests/rl/test_trainer.py) plus loss / optimizer / C51 golden tests."""

import numpy as np
import pytest
import torch

from alphatriangle_amd.config import TrainConfig
from alphatriangle_amd.models import NeuralNetwork
from alphatriangle_amd.ops.losses import alphazero_loss_reference, c51_two_hot
from alphatriangle_amd.ops.optim import FlatParameterSpace, FusedFlatOptimizer
from alphatriangle_amd.rl import ExperienceBuffer, Trainer
from tests.conftest import make_experience


@pytest.fixture
def nn_interface(mock_model_config, mock_env_config, mock_train_config):
    torch.manual_seed(0)
    return NeuralNetwork(mock_model_config, mock_env_config, mock_train_config, torch.device("cpu"))


@pytest.fixture
def batch(mock_model_config, mock_env_config):
    rng = np.random.default_rng(1)
    return [make_experience(mock_model_config, mock_env_config, rng) for _ in range(4)]


def test_prepare_batch(nn_interface, mock_train_config, mock_env_config, batch, action_dim):
    tr = Trainer(nn_interface, mock_train_config, mock_env_config)
    g, o, p, r = tr._prepare_batch(batch)
    assert g.shape == (4, 1, 3, 3) and o.shape == (4, 10) and p.shape == (4, action_dim) and r.shape == (4,)
    assert torch.allclose(p.sum(1), torch.ones(4), atol=1e-5) and abs(float(r[2]) - batch[2][2]) < 1e-6


@pytest.mark.parametrize("use_per", [False, True])
def test_train_step_updates_parameters(mock_model_config, mock_env_config, mock_train_config, batch, use_per):
    torch.manual_seed(0)
    cfg = mock_train_config.model_copy(update={"USE_PER": use_per})
    net = NeuralNetwork(mock_model_config, mock_env_config, cfg, torch.device("cpu"))
    tr = Trainer(net, cfg, mock_env_config)
    before = [p.clone() for p in net.model.parameters()]
    sample = {"batch": batch, "indices": np.arange(4, dtype=np.int32), "weights": np.array([1.0, 0.5, 0.8, 1.0], dtype=np.float32)}
    out = tr.train_step(sample)
    assert out is not None
    info, td = out
    assert set(info) == {"total_loss", "policy_loss", "value_loss", "entropy", "mean_td_error"}
    assert info["total_loss"] > 0 and td.shape == (4,) and np.all(np.isfinite(td))
    assert any(not torch.equal(a, b) for a, b in zip(before, net.model.parameters(), strict=True))
    assert tr.train_step({"batch": [], "indices": np.array([]), "weights": np.array([])}) is None
    assert abs(tr.get_current_lr() - cfg.LEARNING_RATE) < 1e-6


def test_c51_projection_golden():
    m = c51_two_hot(torch.tensor([-10.0, 10.0, 0.0, 0.1, -12.0, 15.0, 9.9]), -10.0, 10.0, 51)
    assert torch.allclose(m.sum(1), torch.ones(7), atol=1e-6)
    assert m[0, 0] == 1 and m[1, 50] == 1 and m[2, 25] == 1 and m[4, 0] == 1 and m[5, 50] == 1
    assert abs(float(m[3, 25]) - 0.75) < 1e-5 and abs(float(m[3, 26]) - 0.25) < 1e-5
    assert abs(float(m[6, 49]) - 0.25) < 1e-4 and abs(float(m[6, 50]) - 0.75) < 1e-4
    support = torch.linspace(-10, 10, 51)
    assert torch.allclose((m * support).sum(1)[[2, 3, 6]], torch.tensor([0.0, 0.1, 9.9]), atol=1e-5)


def test_loss_definition():
    torch.manual_seed(0)
    pl, vl = torch.randn(3, 9), torch.randn(3, 51)
    tgt = torch.softmax(torch.randn(3, 9), 1)
    ret, w = torch.tensor([0.3, -2.0, 7.7]), torch.tensor([1.0, 0.5, 0.25])
    out = alphazero_loss_reference(pl, vl, tgt, ret, w, v_min=-10, v_max=10, policy_weight=1.0, value_weight=2.0, entropy_weight=0.1)
    logp = torch.log_softmax(pl, 1)
    pol = (-(tgt * logp).sum(1) * w).mean()
    val_elem = -(c51_two_hot(ret, -10, 10, 51) * torch.log_softmax(vl, 1)).sum(1)
    probs = torch.softmax(pl, 1)
    ent = -(probs * torch.log(probs + 1e-9)).sum(1).mean()
    assert torch.allclose(out.total, pol + 2.0 * (val_elem * w).mean() - 0.1 * ent, atol=1e-6)
    assert torch.allclose(out.td_errors, val_elem, atol=1e-6) and torch.allclose(out.entropy, ent, atol=1e-6)


@pytest.mark.parametrize("kind", ["Adam", "AdamW", "SGD"])
def test_flat_optimizer_reference_path_matches_torch(kind):
    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))

    m1, m2 = make(), make()
    ref = {"Adam": torch.optim.Adam, "AdamW": torch.optim.AdamW}.get(kind)
    ropt = ref(m1.parameters(), lr=1e-2, weight_decay=1e-2) if ref else torch.optim.SGD(m1.parameters(), lr=1e-2, weight_decay=1e-2, momentum=0.9)
    space = FlatParameterSpace(m2)
    opt = FusedFlatOptimizer(space, kind, 1e-2, 1e-2, 0.5)
    x = torch.randn(8, 7)
    for _ in range(4):
        ropt.zero_grad()
        (m1(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), 0.5)
        ropt.step()
        opt.zero_grad()
        (m2(x) ** 2).sum().backward()
        opt.step()
    for a, b in zip(m1.parameters(), m2.parameters(), strict=True):
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
    # state dict is torch.optim shaped and reloadable
    sd = opt.state_dict()
    opt2 = FusedFlatOptimizer(FlatParameterSpace(make()), kind, 1e-2, 1e-2, 0.5)
    opt2.load_state_dict(sd)
    assert opt2._step_count == (4 if kind != "SGD" else 0)
    assert torch.allclose(opt2.flat_m, opt.flat_m)


def test_fused_trainer_on_cpu_and_schedulers(mock_model_config, mock_env_config, batch):
    cfg = TrainConfig(BATCH_SIZE=4, BUFFER_CAPACITY=100, MIN_BUFFER_SIZE_TO_TRAIN=4, DEVICE="cpu", MAX_TRAINING_STEPS=10,
                      LR_SCHEDULER_TYPE="CosineAnnealingLR", USE_PER=False)
    net = NeuralNetwork(mock_model_config, mock_env_config, cfg, torch.device("cpu"))
    tr = Trainer(net, cfg, mock_env_config, fused=True)
    assert tr.flat is not None
    sample = {"batch": batch, "indices": np.zeros(4, dtype=np.int32), "weights": np.ones(4, dtype=np.float32)}
    losses = [tr.train_step(sample)[0]["total_loss"] for _ in range(10)]
    assert losses[-1] < losses[0]
    assert tr.get_current_lr() < cfg.LEARNING_RATE  # cosine decay
    step_cfg = cfg.model_copy(update={"LR_SCHEDULER_TYPE": "StepLR"})
    assert Trainer(net, step_cfg, mock_env_config).scheduler is not None
    none_cfg = cfg.model_copy(update={"LR_SCHEDULER_TYPE": None})
    assert Trainer(net, none_cfg, mock_env_config).scheduler is None
    tr.load_optimizer_state(tr.optimizer.state_dict())


def test_buffer_to_trainer_flow(mock_model_config, mock_env_config, mock_train_config):
    cfg = mock_train_config.model_copy(update={"USE_PER": True})
    net = NeuralNetwork(mock_model_config, mock_env_config, cfg, torch.device("cpu"))
    tr = Trainer(net, cfg, mock_env_config)
    buf = ExperienceBuffer(cfg)
    rng = np.random.default_rng(5)
    buf.add_batch([make_experience(mock_model_config, mock_env_config, rng) for _ in range(20)])
    s = buf.sample(4, current_train_step=0)
    info, td = tr.train_step(s)
    buf.update_priorities(s["indices"], td)
    assert np.isfinite(buf.tree.total_priority)
