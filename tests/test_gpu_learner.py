"""Learner-side HIP kernels vs PyTorch references (needs an MI355X)."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from alphatriangle_amd.config import EnvConfig, ModelConfig, SystemConfig, TrainConfig

pytestmark = pytest.mark.gpu


def test_fused_loss_matches_reference():
    from alphatriangle_amd.ops.losses import alphazero_loss, alphazero_loss_reference

    torch.manual_seed(0)
    dev = torch.device("cuda")
    B, A, atoms = 64, 360, 51
    pl = torch.randn(B, A, device=dev, requires_grad=True)
    vl = torch.randn(B, atoms, device=dev, requires_grad=True)
    tgt = torch.softmax(torch.randn(B, A, device=dev) * 2, dim=1)
    ret = torch.empty(B, device=dev).uniform_(-12, 12)
    ret[0], ret[1], ret[2] = -10.0, 10.0, 0.0  # exact atoms and both clamps
    isw = torch.rand(B, device=dev) + 0.1
    kw = dict(v_min=-10.0, v_max=10.0, policy_weight=1.0, value_weight=0.7, entropy_weight=0.01)
    ref = alphazero_loss_reference(pl, vl, tgt, ret, isw, **kw)
    ref.total.backward()
    gp_ref, gv_ref = pl.grad.clone(), vl.grad.clone()
    pl.grad = None
    vl.grad = None
    out = alphazero_loss(pl, vl, tgt, ret, isw, **kw)
    out.total.backward()
    torch.cuda.synchronize()
    assert torch.allclose(out.total, ref.total, atol=1e-4, rtol=1e-4)
    assert torch.allclose(out.policy_loss, ref.policy_loss, atol=1e-4, rtol=1e-4)
    assert torch.allclose(out.value_loss, ref.value_loss, atol=1e-4, rtol=1e-4)
    assert torch.allclose(out.entropy, ref.entropy, atol=1e-4, rtol=1e-4)
    assert torch.allclose(out.td_errors, ref.td_errors, atol=1e-4, rtol=1e-4)
    assert torch.allclose(pl.grad, gp_ref, atol=1e-6, rtol=1e-3)
    assert torch.allclose(vl.grad, gv_ref, atol=1e-6, rtol=1e-3)


@pytest.mark.parametrize("kind", ["AdamW", "Adam", "SGD"])
def test_flat_optimizer_matches_torch(kind):
    from alphatriangle_amd.ops.optim import FlatParameterSpace, FusedFlatOptimizer

    torch.manual_seed(1)
    dev = torch.device("cuda")

    def make():
        torch.manual_seed(5)
        return torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Tanh(), torch.nn.Linear(53, 11)).to(dev)

    m1, m2 = make(), make()
    lr, wd, clip = 1e-2, 1e-2, 0.5
    if kind == "AdamW":
        ref = torch.optim.AdamW(m1.parameters(), lr=lr, weight_decay=wd)
    elif kind == "Adam":
        ref = torch.optim.Adam(m1.parameters(), lr=lr, weight_decay=wd)
    else:
        ref = torch.optim.SGD(m1.parameters(), lr=lr, weight_decay=wd, momentum=0.9)
    space = FlatParameterSpace(m2)
    opt = FusedFlatOptimizer(space, kind, lr, wd, clip)
    x = torch.randn(16, 37, device=dev)
    for _ in range(5):
        ref.zero_grad()
        (m1(x) ** 2).sum().backward()
        torch.nn.utils.clip_grad_norm_(m1.parameters(), clip)
        ref.step()
        opt.zero_grad()
        (m2(x) ** 2).sum().backward()
        opt.step()
    torch.cuda.synchronize()
    for p1, p2 in zip(m1.parameters(), m2.parameters(), strict=True):
        assert torch.allclose(p1, p2, atol=2e-5, rtol=1e-4), (p1 - p2).abs().max()
    sd = opt.state_dict()
    assert len(sd["state"]) == len(list(m2.parameters()))


def test_device_per_tree_matches_host_semantics():
    from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
    from tests.conftest import make_experience

    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=1000, MIN_BUFFER_SIZE_TO_TRAIN=10, BATCH_SIZE=16, USE_PER=True, MAX_TRAINING_STEPS=100)
    buf = DeviceReplayBuffer(tc, env, mc, dev, SystemConfig())
    rng = np.random.default_rng(0)
    exps = [make_experience(mc, env, rng) for _ in range(300)]
    buf.add_batch(exps)
    assert len(buf) == 300
    assert abs(buf.total_priority() - 300.0) < 1e-9
    idx = torch.tensor([3, 7, 7, 250], dtype=torch.int32, device=dev)
    err = torch.tensor([0.5, 2.0, 2.0, 0.0], device=dev)
    buf.update_priorities(idx, err)
    leaves = buf.tree[buf.P : buf.P + 300].cpu().numpy()
    exp_p = (np.abs(np.array([0.5, 2.0, 0.0])) + tc.PER_EPSILON) ** tc.PER_ALPHA
    assert np.allclose(leaves[[3, 7, 250]], exp_p, rtol=1e-5)
    assert abs(buf.total_priority() - leaves.sum()) < 1e-6  # internal nodes consistent with the leaves
    assert abs(float(buf.max_prio.item()) - max(1.0, exp_p.max())) < 1e-5
    # proportional sampling: the heavy leaf is drawn far more often than a light one
    counts = np.zeros(300)
    for _ in range(200):
        b = buf.sample(16, current_train_step=50)
        for i in b.indices.cpu().numpy():
            counts[i] += 1
        assert float(b.weights.max()) <= 1.0 + 1e-6
    assert counts[7] > counts[250]
    # round trip to the reference list format
    lst = buf.to_list()
    assert len(lst) == 300 and np.allclose(lst[5][0]["grid"], exps[5][0]["grid"])
    assert abs(lst[5][2] - exps[5][2]) < 1e-6


def test_trainer_step_on_device_batch_changes_params():
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
    from alphatriangle_amd.rl.core.trainer import Trainer
    from tests.conftest import make_experience

    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=500, MIN_BUFFER_SIZE_TO_TRAIN=64, BATCH_SIZE=64, MAX_TRAINING_STEPS=100)
    net = NeuralNetwork(mc, env, tc, dev)
    trainer = Trainer(net, tc, env, autocast_bf16=True)
    buf = DeviceReplayBuffer(tc, env, mc, dev, SystemConfig())
    rng = np.random.default_rng(0)
    buf.add_batch([make_experience(mc, env, rng) for _ in range(128)])
    before = trainer.flat.flat_param.clone()
    first = None
    for step in range(8):
        batch = buf.sample(64, current_train_step=step)
        info, td = trainer.train_step(batch)
        buf.update_priorities(batch.indices, td)
        first = first or info["total_loss"]
        assert np.isfinite(info["total_loss"]) and td.shape == (64,)
    assert not torch.equal(before, trainer.flat.flat_param)
    assert info["total_loss"] < first  # fits the tiny fixed buffer
    net.sync_inference_weights()  # engine re-pack after training must still evaluate
    lg, v = net.evaluate_features(batch.grid, batch.other)
    assert torch.isfinite(lg).all() and torch.isfinite(v).all()
