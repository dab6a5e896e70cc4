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


@pytest.mark.parametrize("kind", ["huge-cnn", "18x18-transformer"])
@pytest.mark.parametrize("fp32", [True, False])
def test_learner_trains_on_boards_above_256_cells(kind, fp32):
    """Boards the device engine and the one-sample-per-workgroup kernels do not hold (more than 256 cells / tokens): the
    network has no inference engine (leaves go through the module) and the learner's step still runs and fits a fixed
    batch, on whatever mix of in-tree kernels and library calls its guards pick for that geometry."""
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.presets import build_preset
    from alphatriangle_amd.rl.core.trainer import TensorBatch, Trainer

    dev = torch.device("cuda")
    if kind == "huge-cnn":
        env, mc, _, _, _ = build_preset("huge")
    else:
        env = EnvConfig(ROWS=18, COLS=18, PLAYABLE_RANGE_PER_ROW=[(1, 17)] * 18)
        mc = ModelConfig(TRANSFORMER_LAYERS=1)
    torch.manual_seed(5)
    tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=32, BUFFER_CAPACITY=500, MIN_BUFFER_SIZE_TO_TRAIN=64, MAX_TRAINING_STEPS=100,
                     COMPILE_MODEL=False)
    SystemConfig(LEARNER_FP32_GEMM="bf16x3" if fp32 else "library").apply_learner_gemm_mode()
    try:
        net = NeuralNetwork(mc, env, tc, dev)
        assert net.engine is None  # no HIP inference engine above 256 cells
        trainer = Trainer(net, tc, env, autocast_bf16=not fp32)
        g = torch.Generator(device="cpu").manual_seed(1)
        n = tc.BATCH_SIZE
        grid = (torch.randint(0, 3, (n, mc.GRID_INPUT_CHANNELS, env.ROWS, env.COLS), generator=g).float() - 1.0).to(dev)
        other = torch.rand(n, mc.OTHER_NN_INPUT_FEATURES_DIM, generator=g).to(dev)
        policy = torch.softmax(torch.randn(n, env.ACTION_DIM, generator=g) * 3.0, dim=1).to(dev)
        ret = (torch.rand(n, generator=g) * 4.0 - 2.0).to(dev)
        batch = TensorBatch(grid=grid, other=other, policy_target=policy, returns=ret, weights=torch.ones(n, device=dev),
                            indices=torch.arange(n, device=dev))
        losses = []
        for _ in range(10):
            info, td = trainer.train_step(batch)
            assert np.isfinite(info["total_loss"]) and td.shape == (n,)
            losses.append(info["total_loss"])
        assert losses[-1] < losses[0], losses
        net.sync_inference_weights()
        lg, v = net.evaluate_features(grid, other)
        assert lg.shape == (n, env.ACTION_DIM) and torch.isfinite(lg).all() and torch.isfinite(v).all()
    finally:
        SystemConfig().apply_learner_gemm_mode()


@pytest.mark.parametrize("preset", ["cnn", "default"])
def test_graphed_learner_step_matches_eager_trajectory(preset):
    """COMPILE_MODEL: forward/backward replayed from a HIP graph when the graph passes its guards (no memset nodes,
    replays reproduce eager gradients).  Two trainers from the same weights train on the same batches, one eager and
    one asking for the graph, with eager work and synchronisation between the replays; their parameters have to stay
    together whether the graph was accepted or refused, and a refusal has to come with its reason."""
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.trainer import TensorBatch, Trainer

    dev = torch.device("cuda")
    env = EnvConfig()
    mc = ModelConfig() if preset == "default" else ModelConfig(USE_TRANSFORMER=False)
    trainers = []
    for graph in (False, True):
        torch.manual_seed(3)
        tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=64, MAX_TRAINING_STEPS=100, COMPILE_MODEL=graph)
        net = NeuralNetwork(mc, env, tc, dev)
        for m in net.model.modules():  # dropout off: the two trajectories then differ only by summation order
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, torch.nn.MultiheadAttention):
                m.dropout = 0.0
        trainers.append(Trainer(net, tc, env, autocast_bf16=True))
    eager, graphed = trainers
    assert not eager.use_graph and graphed.use_graph
    assert torch.equal(eager.flat.flat_param, graphed.flat.flat_param)
    start = eager.flat.flat_param.clone()
    gen = torch.Generator(device="cpu").manual_seed(5)
    for step in range(8):
        pol = torch.rand(64, env.ACTION_DIM, generator=gen)
        tb = TensorBatch(
            torch.randint(-1, 2, (64, mc.GRID_INPUT_CHANNELS, env.ROWS, env.COLS), generator=gen).float().to(dev),
            torch.rand(64, mc.OTHER_NN_INPUT_FEATURES_DIM, generator=gen).to(dev),
            (pol / pol.sum(1, keepdim=True)).to(dev),
            torch.empty(64).uniform_(-5, 5, generator=gen).to(dev),
            torch.rand(64, generator=gen).to(dev) + 0.5,
        )
        info_e, td_e = eager.train_step_tensors(tb)
        torch.cuda.synchronize()
        info_g, td_g = graphed.train_step_tensors(tb)
        torch.cuda.synchronize()
        assert abs(float(info_e["total_loss"]) - float(info_g["total_loss"])) < 2e-2 * abs(float(info_e["total_loss"])) + 1e-3
        assert torch.allclose(td_e, td_g, rtol=5e-2, atol=5e-2)
        assert torch.isfinite(graphed.flat.flat_grad).all() and torch.isfinite(graphed.flat.flat_param).all(), step
    report = graphed.graph_report
    print(f"learner graph [{preset}]: {report}")
    assert report["kernel_nodes"] > 50
    if report["replayed"]:
        assert report["memset_nodes"] == 0 and report["max_rel_err"] < 2e-2 and graphed.use_graph
    else:
        assert report["memset_nodes"] > 0 or not report["max_rel_err"] < 2e-2
        assert not graphed.use_graph
    # AdamW normalises every coordinate's step to about lr, so compare against the distance travelled
    apart = float(torch.linalg.vector_norm(eager.flat.flat_param - graphed.flat.flat_param))
    travelled = float(torch.linalg.vector_norm(eager.flat.flat_param - start))
    assert travelled > 0 and apart < 0.2 * travelled, (apart, travelled)
    # BatchNorm running statistics advanced exactly once per step on both sides
    for be, bg in zip(eager.model.buffers(), graphed.model.buffers(), strict=True):
        assert torch.allclose(be.float(), bg.float(), rtol=1e-2, atol=1e-3)
    # a batch of another size runs eagerly through the same trainer
    small = TensorBatch(tb.grid[:32], tb.other[:32], tb.policy_target[:32], tb.returns[:32], tb.weights[:32])
    info_s, td_s = graphed.train_step_tensors(small)
    assert td_s.shape == (32,) and torch.isfinite(info_s["total_loss"])


@pytest.mark.parametrize("n,f,dtype", [(30720, 384, torch.bfloat16), (30720, 128, torch.float32), (1000, 72, torch.bfloat16), (7, 5, torch.float32)])
def test_colsum_kernel_matches_fp32_sum(n, f, dtype):
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(n + f)
    g = torch.randn(n, f, device="cuda").to(dtype)
    ref = g.float().sum(0)
    out = hip_learner.colsum(g)
    assert out.dtype == torch.float32 and out.shape == (f,)
    assert torch.allclose(out, ref, rtol=1e-4, atol=1e-3 * (n ** 0.5))


@pytest.mark.parametrize("rows,d,dtype", [(30720, 128, torch.bfloat16), (1000, 128, torch.float32), (777, 64, torch.bfloat16),
                                            (130, 256, torch.float32), (65, 512, torch.bfloat16)])
def test_layer_norm_kernels_match_torch(rows, d, dtype):
    """Forward and backward of the learner's LayerNorm kernels against torch's fp32 LayerNorm on the same (rounded) input."""
    from alphatriangle_amd.models.model import FastLayerNorm

    torch.manual_seed(rows + d)
    dev = torch.device("cuda")
    ln = FastLayerNorm(d).to(dev)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.3)
        ln.bias.normal_(0.0, 0.3)
    x = (torch.randn(rows, d, device=dev) * 2.0 + 0.5).to(dtype).requires_grad_(True)
    gy = torch.randn(rows, d, device=dev).to(dtype)
    y = ln(x)
    assert y.dtype == dtype and y.shape == x.shape
    gx, gw, gb = torch.autograd.grad(y, [x, ln.weight, ln.bias], gy)
    xr = x.detach().float().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), ln.weight.float(), ln.bias.float(), ln.eps)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, ln.weight, ln.bias], gy.float())
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert (y.float() - yr).abs().max().item() < tol * max(1.0, yr.abs().max().item())
    assert (gx.float() - gxr).abs().max().item() < tol * max(1.0, gxr.abs().max().item())
    assert torch.allclose(gw.float(), gwr, rtol=1e-3, atol=1e-3 * rows ** 0.5)
    assert torch.allclose(gb.float(), gbr, rtol=1e-3, atol=1e-3 * rows ** 0.5)


@pytest.mark.parametrize("rows,d", [(30720, 128), (1000, 256)])
def test_layer_norm_of_fp32_rows_under_bf16_autocast_emits_bf16(rows, d):
    """Under bf16 autocast the LayerNorm of an fp32 residual stream comes out as bf16 - the values an autocast GEMM would
    cast the fp32 result to - and its backward takes the bf16 gradient and returns an fp32 one."""
    from alphatriangle_amd.models.model import FastLayerNorm

    torch.manual_seed(rows)
    dev = torch.device("cuda")
    ln = FastLayerNorm(d).to(dev)
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.3)
        ln.bias.normal_(0.0, 0.3)
    x = (torch.randn(rows, d, device=dev) * 2.0 + 0.5).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = ln(x)
    assert y.dtype == torch.bfloat16
    gy = torch.randn(rows, d, device=dev).bfloat16()
    gx, gw, gb = torch.autograd.grad(y, [x, ln.weight, ln.bias], gy)
    assert gx.dtype == torch.float32
    xr = x.detach().clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (d,), ln.weight, ln.bias, ln.eps)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, ln.weight, ln.bias], gy.float())
    # forward: one bf16 rounding of the fp32 result (the fp32 results of the two implementations differ by ~1e-6, which
    # can flip a rounding tie: allow one bf16 step)
    assert (y.float() - yr.bfloat16().float()).abs().max().item() <= 2 ** -7 * yr.abs().max().item()
    assert (gx - gxr).abs().max().item() < 1e-4 * max(1.0, gxr.abs().max().item())
    assert torch.allclose(gw, gwr, rtol=1e-3, atol=1e-3 * rows ** 0.5)
    assert torch.allclose(gb, gbr, rtol=1e-3, atol=1e-3 * rows ** 0.5)
    assert ln(x).dtype == torch.float32  # outside autocast the rows keep their dtype


@pytest.mark.parametrize("c,b,h,w,k,p", [(128, 256, 8, 15, 3, 1), (1, 256, 8, 15, 3, 1), (32, 64, 8, 15, 3, 0), (16, 8, 12, 21, 5, 2),
                                         (64, 32, 5, 9, 3, 2)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_patch_matrix_kernel_matches_torch_gather(c, b, h, w, k, p, dtype):
    """patch_matrix_kernel (channel-major im2col, one launch) against strided-view gathering in PyTorch: exact copies."""
    from alphatriangle_amd.models.model import _patch_matrix
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(c + b)
    dev = torch.device("cuda")
    x = torch.randn(c, b, h, w, device=dev).to(dtype)
    assert hip_learner.patch_matrix_supported(x, k, k, p, p)
    cols = hip_learner.patch_matrix_cm(x, k, k, p, p)
    ref = _patch_matrix(x.permute(1, 0, 2, 3).contiguous(), k, k, p, p)
    torch.cuda.synchronize()
    assert cols.shape == ref.shape
    assert torch.equal(cols, ref)


@pytest.mark.parametrize("preset", ["default", "cnn"])
def test_channel_major_conv_body_matches_nchw_path(preset):
    """The learner's channel-major convolution body (patch-matrix kernel, no layout copies) against the NCHW form of the
    same module: same outputs, same gradients (bf16 autocast), same BatchNorm running statistics."""
    import copy

    from alphatriangle_amd.models.model import AlphaTriangleNet

    torch.manual_seed(21)
    dev = torch.device("cuda")
    env = EnvConfig()
    mc = ModelConfig() if preset == "default" else ModelConfig.cnn_only()
    net_a = AlphaTriangleNet(mc, env).to(dev).train()
    net_b = copy.deepcopy(net_a)
    net_b.channel_major = False
    for m in list(net_a.modules()) + list(net_b.modules()):
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0  # applied functionally inside the attention
    B = 256
    grid = torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    net_r = copy.deepcopy(net_a)  # fp32 reference, no autocast
    net_r.channel_major = False
    outs, grads = [], []
    for net in (net_a, net_b, net_r):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=net is not net_r):
            assert net._cm_plan_ok(grid) == (net is net_a)
            pl, vl = net(grid, other)
            loss = pl.float().square().mean() + vl.float().square().mean()
        loss.backward()
        outs.append((pl.float().detach(), vl.float().detach()))
        grads.append({n: p.grad.float() for n, p in net.named_parameters()})
    torch.cuda.synchronize()
    # both bf16 paths are compared with the fp32 run: the channel-major path may not be further from it than the NCHW path
    for i in (0, 1):
        err_cm = (outs[0][i] - outs[2][i]).abs().max().item()
        err_nchw = (outs[1][i] - outs[2][i]).abs().max().item()
        assert err_cm < 1.5 * err_nchw + 0.02, (i, err_cm, err_nchw)
    for name, gr in grads[2].items():
        scale = gr.abs().max().item() + 1e-6
        err_cm = (grads[0][name] - gr).abs().max().item() / scale
        err_nchw = (grads[1][name] - gr).abs().max().item() / scale
        assert err_cm < 1.5 * err_nchw + 0.03, (name, err_cm, err_nchw)
    for (name, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        if "running" in name:
            assert torch.allclose(ba.float(), bb.float(), rtol=2e-2, atol=2e-3), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["ReLU", "GELU", "Identity"])
@pytest.mark.parametrize("with_residual", [False, True])
def test_fused_batchnorm_act_kernels_match_torch(dtype, act, with_residual):
    """Channel-major BatchNorm (+ skip) (+ activation), forward and backward, against torch in fp32."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(17)
    dev = torch.device("cuda")
    c, b, h, w = 96, 64, 8, 15
    x = (torch.randn(c, b, h, w, device=dev) * 1.7 + 3.0).to(dtype)  # a large mean: the shifted one-pass variance has to cope
    r = torch.randn(c, b, h, w, device=dev).to(dtype) if with_residual else None
    gy = torch.randn(c, b, h, w, device=dev).to(dtype)
    bn = torch.nn.BatchNorm2d(c).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    ref_bn = torch.nn.BatchNorm2d(c).to(dev).train()
    ref_bn.load_state_dict(bn.state_dict())
    act_id = hip_learner.ACT_IDS[act]
    y, mean, rstd = hip_learner.bn_act_cm_forward(x, r, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, bn.running_mean,
                                                  bn.running_var, bn.num_batches_tracked, act_id)
    dx, dr, dg, db = hip_learner.bn_act_cm_backward(x, r, gy, bn.weight.detach(), bn.bias.detach(), mean, rstd, act_id)
    # reference: NCHW fp32
    xr = x.float().permute(1, 0, 2, 3).contiguous().requires_grad_(True)
    rr = r.float().permute(1, 0, 2, 3).contiguous().requires_grad_(True) if r is not None else None
    z = ref_bn(xr) + (rr if rr is not None else 0)
    yr = getattr(torch.nn, act)()(z)
    grads = torch.autograd.grad(yr, [xr, ref_bn.weight, ref_bn.bias] + ([rr] if rr is not None else []), gy.float().permute(1, 0, 2, 3))
    torch.cuda.synchronize()
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-4

    def close(a, ref, what):
        err = (a.float() - ref).abs().max().item()
        assert err < tol * max(1.0, ref.abs().max().item()), (what, err)

    close(y.permute(1, 0, 2, 3), yr, "y")
    close(dx.permute(1, 0, 2, 3), grads[0], "dx")
    close(dg, grads[1], "dgamma") if dtype == torch.float32 else None
    close(db, grads[2], "dbeta") if dtype == torch.float32 else None
    if rr is not None:
        close(dr.permute(1, 0, 2, 3), grads[3], "dr")
    assert torch.allclose(bn.running_mean, ref_bn.running_mean, rtol=1e-3, atol=2e-3 if dtype == torch.bfloat16 else 1e-5)
    assert torch.allclose(bn.running_var, ref_bn.running_var, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4, atol=1e-4)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("preset,patch_free", [("default", False), ("cnn", False), ("default", True)])
def test_fp32_learner_fast_path_matches_stock_modules(preset, patch_free):
    """fp32 (no autocast): channel-major body + fused BatchNorm / attention kernels + bf16-matrix-core GEMMs against the plain
    NCHW module path on the library's fp32 GEMMs - same outputs, gradients and running statistics to fp32 rounding, both
    measured against an fp64 run of the same step."""
    import copy

    from alphatriangle_amd.models.model import AlphaTriangleNet, EncoderLayer

    torch.manual_seed(23)
    dev = torch.device("cuda")
    env = EnvConfig()
    mc = ModelConfig() if preset == "default" else ModelConfig.cnn_only()
    net_a = AlphaTriangleNet(mc, env).to(dev).train()
    for m in net_a.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    net_b = copy.deepcopy(net_a)
    net_b.channel_major = False
    net_b.fused_tail = False  # stock Linear / BatchNorm1d modules on the comparison side
    for m in net_b.modules():
        if isinstance(m, EncoderLayer):
            m.fused_attention = False  # batched-GEMM attention on the comparison side
    B = 64
    grid = torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    assert net_a._cm_plan_ok(grid) and not net_b._cm_plan_ok(grid)
    import alphatriangle_amd.models.model as model_mod

    net_c = copy.deepcopy(net_b).double()  # fp64 statement of the same step: the yardstick for every other path
    net_d = copy.deepcopy(net_a)           # the fast path under bf16 autocast, for scale
    res = []
    for net, dt, fast_gemm, low in ((net_a, torch.float32, True, False), (net_b, torch.float32, False, False),
                                    (net_c, torch.float64, False, False), (net_d, torch.float32, True, True)):
        model_mod.FP32_GEMM_ON_BF16_CORES = fast_gemm  # the comparison side runs on the library's fp32 GEMMs
        model_mod.PATCH_FREE_CONV = patch_free and net is net_a  # LEARNER_PATCH_FREE_CONV: no patch matrices in memory
        try:
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=low):
                pl, vl = net(grid.to(dt), other.to(dt))
            (pl.float().square().mean() + vl.float().square().mean()).backward() if low else (pl.square().mean() + vl.square().mean()).backward()
        finally:
            model_mod.FP32_GEMM_ON_BF16_CORES = True
            model_mod.PATCH_FREE_CONV = False
        res.append((pl.detach().double(), vl.detach().double(), {n: p.grad.double() for n, p in net.named_parameters()}))
    torch.cuda.synchronize()
    assert (res[0][0] - res[2][0]).abs().max().item() < 2e-4 and (res[0][1] - res[2][1]).abs().max().item() < 2e-4
    top = max(g.abs().max().item() for g in res[2][2].values())
    worst = [0.0, 0.0, 0.0]
    for name, g in res[2][2].items():
        scale = max(g.abs().max().item(), 1e-4 * top)  # gradients that are zero up to rounding are compared on the global scale
        for i, k in enumerate((0, 1, 3)):
            worst[i] = max(worst[i], (res[k][2][name] - g).abs().max().item() / scale)
    print(f"worst gradient error vs fp64: fast fp32 path {worst[0]:.2e}, library fp32 path {worst[1]:.2e}, bf16 autocast {worst[2]:.2e}")
    # Every product of the fast path is good to ~5e-6 of its largest output (tests below; the library's fp32: ~7e-7), but
    # twenty layers of batch-statistics normalisation and ReLU gates amplify forward rounding differences about a
    # thousandfold on the way to the gradients of the most sensitive layers: 1e-3 for the library path, a few 1e-2 for the
    # three-pass path, order one for bf16 autocast.
    assert worst[0] < 0.1 and worst[0] < 0.25 * worst[2], worst
    for (name, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        if "running" in name:
            assert torch.allclose(ba, bb, rtol=1e-3, atol=1e-5), name
        if "num_batches_tracked" in name:
            assert int(ba) == int(bb), name  # the in-kernel counter bump equals the module's own


def _dropout_keep_mask(seed: int, bh: int, seq: int, p: float, device) -> torch.Tensor:
    """Host re-statement of ``drop_hash`` in csrc/hip/learner_attn_kernels.hip: keep[bh, query, key]."""
    lp = 256 if seq > 128 else 128
    m32 = 0xFFFFFFFF
    s0, s1 = seed & m32, (seed >> 32) & m32
    thresh = int(float(np.float32(p)) * 4294967296.0)
    i = torch.arange(bh, device=device, dtype=torch.int64).view(-1, 1, 1)
    q = torch.arange(seq, device=device, dtype=torch.int64).view(1, -1, 1)
    k = torch.arange(seq, device=device, dtype=torch.int64).view(1, 1, -1)
    x = (((i * lp + q) * lp + k) & m32) * 0x9E3779B1 + s0 & m32
    x = ((x ^ (x >> 16)) * 0x85EBCA6B) & m32
    x = ((x ^ (x >> 13)) * 0xC2B2AE35) & m32
    x = ((x ^ (x >> 16)) + s1) & m32
    x = ((x ^ (x >> 15)) * 0x2C1B3C6D) & m32
    x = x ^ (x >> 12)
    return x >= thresh


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seq,batch,heads,p", [(120, 16, 4, 0.0), (120, 16, 4, 0.1), (30, 5, 2, 0.25), (252, 6, 4, 0.1), (129, 3, 1, 0.0)])
def test_fused_attention_kernels_match_torch(dtype, seq, batch, heads, p):
    """One-launch attention forward / backward against an fp64 PyTorch statement of the same arithmetic (same dropout mask)."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(31)
    dev = torch.device("cuda")
    dim = 32 * heads
    qkv = (torch.randn(seq, batch, 3 * dim, device=dev) * 1.5).to(dtype)
    w = torch.randn(seq, batch, dim, device=dev).to(dtype)
    assert hip_learner.attention_supported(qkv, heads)
    ctx, lse, seed = hip_learner.attention_forward(qkv, heads, p)
    dqkv = hip_learner.attention_backward(qkv, w, lse, seed, heads, p)
    torch.cuda.synchronize()

    ref_in = qkv.double().requires_grad_(True)
    q, k, v = (t.reshape(seq, batch, heads, 32).permute(1, 2, 0, 3) for t in ref_in.chunk(3, dim=-1))
    probs = torch.softmax(q @ k.transpose(-2, -1) / 32**0.5, dim=-1)
    if p > 0:
        keep = _dropout_keep_mask(int(seed.item()), batch * heads, seq, p, dev).view(batch, heads, seq, seq)
        rate = 1.0 - keep.double().mean().item()
        assert abs(rate - p) < 0.01, f"drop rate {rate} for p = {p}"
        probs = probs * keep / (1.0 - float(np.float32(p)))
    ref = (probs @ v).permute(2, 0, 1, 3).reshape(seq, batch, dim)
    (ref * w.double()).sum().backward()
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    err_f = (ctx.double() - ref).abs().max().item() / ref.abs().max().item()
    err_b = (dqkv.double() - ref_in.grad).abs().max().item() / ref_in.grad.abs().max().item()
    print(f"[{dtype}, L={seq}, p={p}] forward {err_f:.2e}, backward {err_b:.2e}")
    assert err_f < tol and err_b < tol
    lse_ref = torch.logsumexp(q @ k.transpose(-2, -1) / 32**0.5, dim=-1) / np.log(2.0)
    assert (lse.view(batch, heads, -1)[:, :, :seq].double() - lse_ref).abs().max().item() < (1e-4 if dtype == torch.float32 else 5e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("seq,batch,heads", [(45, 5, 2), (120, 16, 4), (252, 3, 4)])
def test_fused_attention_batch_major_rows_equal_token_major_rows(dtype, seq, batch, heads):
    """The ``batch_major`` row order (``[B, L, ...]``) is the same arithmetic on the same (sample, head) slices: bitwise equal."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(43)
    dev = torch.device("cuda")
    qkv = torch.randn(seq, batch, 96 * heads, device=dev).to(dtype)
    w = torch.randn(seq, batch, 32 * heads, device=dev).to(dtype)
    ctx, lse, seed = hip_learner.attention_forward(qkv, heads, 0.0)
    dqkv = hip_learner.attention_backward(qkv, w, lse, seed, heads, 0.0)
    qkv_b, w_b = qkv.transpose(0, 1).contiguous(), w.transpose(0, 1).contiguous()
    ctx_b, lse_b, seed_b = hip_learner.attention_forward(qkv_b, heads, 0.0, batch_major=True)
    dqkv_b = hip_learner.attention_backward(qkv_b, w_b, lse_b, seed_b, heads, 0.0, batch_major=True)
    torch.cuda.synchronize()
    assert ctx_b.shape == (batch, seq, 32 * heads) and dqkv_b.shape == qkv_b.shape
    assert torch.equal(ctx_b, ctx.transpose(0, 1)) and torch.equal(dqkv_b, dqkv.transpose(0, 1))
    assert torch.equal(lse_b[:, :seq], lse[:, :seq])


def test_fused_attention_draws_a_new_mask_on_every_graph_replay():
    """Captured forward + backward replayed three times: every replay equals an eager run with the seed it drew."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(37)
    dev = torch.device("cuda")
    seq, batch, heads, p = 120, 8, 4, 0.1
    qkv = torch.randn(seq, batch, 96 * heads, device=dev)
    w = torch.randn(seq, batch, 32 * heads, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip_learner.attention_forward(qkv, heads, p)  # warm-up outside the capture
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ctx, lse, seed = hip_learner.attention_forward(qkv, heads, p)
        dqkv = hip_learner.attention_backward(qkv, w, lse, seed, heads, p)
    seeds = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        seeds.append(int(seed.item()))
        eager_ctx = torch.empty_like(ctx)
        eager_lse = torch.empty_like(lse)
        hip_learner.hb.call("atri_attn_fwd", hip_learner.hb.ptr(qkv), hip_learner.hb.i32(0), hip_learner.hb.ptr(eager_ctx),
                            hip_learner.hb.ptr(eager_lse), hip_learner.hb.ptr(seed), hip_learner.hb.i32(seq), hip_learner.hb.i32(batch),
                            hip_learner.hb.i32(heads), hip_learner.hb.f32(p))
        eager_dqkv = hip_learner.attention_backward(qkv, w, eager_lse, seed, heads, p)
        torch.cuda.synchronize()
        assert torch.equal(eager_ctx, ctx) and torch.equal(eager_dqkv, dqkv)
    assert len(set(seeds)) == 3, "every replay has to draw its own dropout seed"


def test_bf16_shadow_follows_the_parameters_through_optimizer_steps_and_outside_writes():
    """Autocast steps read one flat bf16 copy of the parameters (refreshed by one cast per step) instead of casting each
    weight: the copy has to track the optimizer's updates and any write made behind the trainer's back."""
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.trainer import TensorBatch, Trainer

    torch.manual_seed(41)
    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=32, COMPILE_MODEL=False)
    net = NeuralNetwork(mc, env, tc, dev)
    trainer = Trainer(net, tc, env, autocast_bf16=True)
    assert trainer.flat is not None
    B = 32
    pol = torch.rand(B, env.ACTION_DIM, device=dev)
    tb = TensorBatch(grid=torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), device=dev).float(), other=torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev),
                     policy_target=pol / pol.sum(1, keepdim=True), returns=torch.empty(B, device=dev).uniform_(-3, 3), weights=torch.ones(B, device=dev),
                     indices=torch.arange(B, device=dev))
    before = trainer.flat.flat_param.clone()
    trainer.train_step_tensors(tb)
    assert not torch.equal(before, trainer.flat.flat_param)
    conv_w = net._orig_model.conv_body[0][0].weight
    off = trainer.flat.offsets[[id(p) for p in trainer.flat.params].index(id(conv_w))]
    assert conv_w._atri_shadow.dtype == torch.bfloat16
    assert torch.equal(conv_w._atri_shadow, before[off : off + conv_w.numel()].view(conv_w.shape).bfloat16())  # what the step started from
    with torch.no_grad():
        conv_w.mul_(0.5)  # an outside write (checkpoint load, weight broadcast, ...)
    expect = conv_w.detach().clone().bfloat16()
    trainer.train_step_tensors(tb)
    assert torch.equal(conv_w._atri_shadow, expect), "the step has to start from the parameters as they are now"


@pytest.mark.parametrize("m,n,k,a_t,b_t,batch,with_bias", [
    (128, 30720, 1152, False, False, 0, False),   # convolution forward / data gradient: W[Cout, 9 Cin] @ cols[9 Cin, B*H*W]
    (64, 3072, 288, False, False, 0, False),
    (32, 2052, 288, False, False, 0, False),      # partial column tile
    (30720, 384, 128, False, True, 0, True),      # linear forward: x @ W^T + b
    (30720, 128, 256, False, False, 0, False),    # linear data gradient: g @ W
    (128, 1152, 960, False, True, 32, False),     # split-K convolution weight gradient: g chunks @ cols chunks^T
    (384, 128, 960, True, False, 32, False),      # split-K linear weight gradient: g chunks^T @ x chunks
    (100, 260, 100, True, True, 3, True),         # nothing is a multiple of a tile size
    (36, 8, 4, False, False, 0, False),
])
def test_gemm3_matches_fp64(m, n, k, a_t, b_t, batch, with_bias):
    """fp32 GEMM on the bf16 matrix cores (three passes over (hi, lo) splits) against fp64, for every operand layout the
    learner uses; the library's own fp32 result is printed next to it."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(47)
    dev = torch.device("cuda")
    lead = (batch,) if batch else ()
    a = torch.randn(*lead, k, m, device=dev).transpose(-1, -2) if a_t else torch.randn(*lead, m, k, device=dev)
    b = torch.randn(*lead, n, k, device=dev).transpose(-1, -2) if b_t else torch.randn(*lead, k, n, device=dev)
    bias = torch.randn(n, device=dev) if with_bias else None
    out = hip_learner.mm3(a, b, bias)
    assert out is not None and out.shape == (*lead, m, n), "this shape has to take the kernel path"
    torch.cuda.synchronize()
    ref = a.double() @ b.double() + (bias.double() if with_bias else 0.0)
    lib = (torch.bmm(a, b) if batch else a @ b) + (bias if with_bias else 0.0)
    scale = ref.abs().max().item()
    err = (out.double() - ref).abs().max().item() / scale
    lib_err = (lib.double() - ref).abs().max().item() / scale
    print(f"[{m}x{n}x{k} a_t={a_t} b_t={b_t} batch={batch}] kernel {err:.2e}, library fp32 {lib_err:.2e}")
    assert err < 2e-5


def test_gemm3_declines_what_it_cannot_run():
    from alphatriangle_amd.ops import hip_learner

    dev = torch.device("cuda")
    assert hip_learner.mm3(torch.randn(32, 16, device=dev).bfloat16(), torch.randn(16, 32, device=dev).bfloat16()) is None
    assert hip_learner.mm3(torch.randn(64, 32, device=dev)[::2, ::2], torch.randn(16, 32, device=dev)) is None  # no unit stride


@pytest.mark.parametrize("case", ["odd_everything", "value_head", "shared_layer_slices", "column_block_out", "pairs"])
def test_gemm3_with_pitches_and_extents_that_are_not_multiples_of_four(case):
    """Operands whose pitch, extent or base address rules out 16-byte accesses run on the kernel's 8-byte / 4-byte paths:
    the value head's 51 atoms, the shared layer's 15 390-wide rows (column slices of the weight), a column block of a
    larger matrix as the output."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(53)
    dev = torch.device("cuda")
    bias = None
    out = None
    if case == "odd_everything":
        a, b = torch.randn(37, 9, device=dev), torch.randn(9, 31, device=dev)
        bias = torch.randn(31, device=dev)
    elif case == "value_head":  # forward, data gradient and weight gradient of Linear(128 -> 51) at batch 256
        h, w, g = torch.randn(256, 128, device=dev), torch.randn(51, 128, device=dev), torch.randn(256, 51, device=dev)
        for a, b in ((h, w.t()), (g, w), (g.t(), h)):
            got = hip_learner.mm3(a, b)
            ref = a.double() @ b.double()
            assert got is not None and (got.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-5
        return
    elif case == "shared_layer_slices":  # [B, 15360] x column block of W[128, 15390], cut into 60 slices along k
        x, w = torch.randn(64, 15360, device=dev), torch.randn(128, 15390, device=dev)
        a = x.view(64, 60, 256).transpose(0, 1)
        b = w[:, :15360].view(128, 60, 256).permute(1, 2, 0)
        got = hip_learner.mm3(a, b)
        assert got is not None
        ref = x.double() @ w[:, :15360].double().t()
        err = (got.double().sum(0) - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, err
        small = hip_learner.mm3(torch.randn(64, 30, device=dev), w[:, 15360:].t())
        assert small is not None and small.shape == (64, 128)
        return
    elif case == "column_block_out":  # weight gradient written into its column block of the full matrix
        g, x = torch.randn(64, 128, device=dev), torch.randn(64, 30, device=dev)
        full = torch.full((128, 15390), 7.0, device=dev)
        a, b, out = g.t(), x, full[:, 15360:]
    else:  # even but not a multiple of four
        a, b = torch.randn(66, 30, device=dev), torch.randn(30, 54, device=dev)
    got = hip_learner.mm3(a, b, bias, out=out)
    assert got is not None
    torch.cuda.synchronize()
    ref = a.double() @ b.double() + (bias.double() if bias is not None else 0.0)
    assert (got.double() - ref).abs().max().item() / ref.abs().max().item() < 2e-5
    if case == "column_block_out":
        assert got.data_ptr() == out.data_ptr() and bool((full[:, :15360] == 7.0).all()), "only the column block may be written"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("s,shape,with_bias", [(32, (128, 1152), False), (61, (256, 128), True), (3, (7, 51), True), (1, (5, 3), False)])
def test_sum_slices_matches_torch(dtype, s, shape, with_bias):
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(3)
    dev = torch.device("cuda")
    parts = torch.randn(s, *shape, device=dev).to(dtype)
    bias = torch.randn(shape[-1], device=dev) if with_bias else None
    got = hip_learner.sum_slices(parts, bias)
    ref = parts.double().sum(0) + (bias.double() if with_bias else 0.0)
    assert got.dtype == torch.float32 and got.shape == shape
    assert (got.double() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", ["Identity", "ReLU", "GELU", "Tanh"])
@pytest.mark.parametrize("b,c", [(256, 128), (37, 50)])
def test_batchnorm1d_kernels_match_torch(dtype, act, b, c):
    """One-launch BatchNorm1d (+ activation), forward and backward, against torch's module in fp64."""
    from alphatriangle_amd.models.model import _Bn1dAct
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(11)
    dev = torch.device("cuda")
    bn = torch.nn.BatchNorm1d(c).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    import copy

    ref_bn = copy.deepcopy(bn).double()
    x = (torch.randn(b, c, device=dev) * 2 + 0.7).to(dtype).requires_grad_(True)
    xr = x.detach().double().requires_grad_(True)
    y = _Bn1dAct.apply(x, bn.weight, bn.bias, bn, hip_learner.ACT_IDS[act])
    yr = getattr(torch.nn, act)()(ref_bn(xr))
    gy = torch.randn(b, c, device=dev)
    y.backward(gy.to(dtype))
    yr.backward(gy.to(dtype).double())
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-5
    assert (y.double() - yr).abs().max().item() < tol * max(1.0, yr.abs().max().item())
    for got, ref, name in ((x.grad, xr.grad, "dx"), (bn.weight.grad, ref_bn.weight.grad, "dgamma"), (bn.bias.grad, ref_bn.bias.grad, "dbeta")):
        assert (got.double() - ref).abs().max().item() < (10 * tol if dtype == torch.bfloat16 else tol) * max(1.0, ref.abs().max().item()), name
    assert torch.allclose(bn.running_mean.double(), ref_bn.running_mean, rtol=1e-3, atol=1e-5)
    assert torch.allclose(bn.running_var.double(), ref_bn.running_var, rtol=1e-3, atol=1e-5)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cout,cin,k", [(128, 128, 3), (64, 32, 3), (32, 1, 5)])
def test_flipped_filter_kernel_matches_flip_and_permute(dtype, cout, cin, k):
    from alphatriangle_amd.ops import hip_learner

    w = torch.randn(cout, cin, k, k, device="cuda").to(dtype)
    assert torch.equal(hip_learner.conv_wt(w), w.flip(2, 3).permute(1, 0, 2, 3).reshape(cin, cout * k * k))


@pytest.mark.parametrize("variant", ["no_bn_gelu", "deep_shared", "no_hidden_heads", "odd_widths"])
def test_learner_tail_kernels_follow_unusual_head_configurations(variant):
    """The in-tree shared layer / heads path against the stock modules for head stacks the default network does not have:
    no BatchNorm, other activations, several or no hidden layers, widths and feature counts that are not multiples of
    four (5x6 board: 30 cells x 32 channels + 7 other features)."""
    import copy

    from alphatriangle_amd.models.model import AlphaTriangleNet

    torch.manual_seed(83)
    dev = torch.device("cuda")
    env = EnvConfig(ROWS=5, COLS=6, PLAYABLE_RANGE_PER_ROW=[(0, 6)] * 5, NUM_SHAPE_SLOTS=2)
    kw = dict(CONV_FILTERS=[32], CONV_KERNEL_SIZES=[3], CONV_STRIDES=[1], CONV_PADDING=[1], NUM_RESIDUAL_BLOCKS=0,
              RESIDUAL_BLOCK_FILTERS=32, USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0, OTHER_NN_INPUT_FEATURES_DIM=7)
    if variant == "no_bn_gelu":
        mc = ModelConfig(**kw, USE_BATCH_NORM=False, ACTIVATION_FUNCTION="GELU", FC_DIMS_SHARED=[64], POLICY_HEAD_DIMS=[48], VALUE_HEAD_DIMS=[40])
    elif variant == "deep_shared":
        mc = ModelConfig(**kw, FC_DIMS_SHARED=[96, 64, 32], POLICY_HEAD_DIMS=[64, 64], VALUE_HEAD_DIMS=[32], ACTIVATION_FUNCTION="SiLU")
    elif variant == "no_hidden_heads":
        mc = ModelConfig(**kw, FC_DIMS_SHARED=[128], POLICY_HEAD_DIMS=[], VALUE_HEAD_DIMS=[], NUM_VALUE_ATOMS=21)
    else:
        mc = ModelConfig(**kw, FC_DIMS_SHARED=[50], POLICY_HEAD_DIMS=[34], VALUE_HEAD_DIMS=[18], NUM_VALUE_ATOMS=11, ACTIVATION_FUNCTION="Tanh")
    net_a = AlphaTriangleNet(mc, env).to(dev).train()
    net_b = copy.deepcopy(net_a)
    net_b.fused_tail = False
    net_c = copy.deepcopy(net_b).double()
    B = 48
    grid = torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), device=dev).float()
    other = torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, device=dev)
    res = []
    for net, dt in ((net_a, torch.float32), (net_b, torch.float32), (net_c, torch.float64)):
        pl, vl = net(grid.to(dt), other.to(dt))
        (pl.square().mean() + vl.square().mean()).backward()
        res.append((pl.detach().double(), vl.detach().double(), {n: p.grad.double() for n, p in net.named_parameters()}))
    torch.cuda.synchronize()
    assert res[0][0].shape == res[2][0].shape and res[0][1].shape == res[2][1].shape
    for k in (0, 1):
        assert (res[0][k] - res[2][k]).abs().max().item() < 2e-4 * max(1.0, res[2][k].abs().max().item())
    top = max(g.abs().max().item() for g in res[2][2].values())
    for name, g in res[2][2].items():
        scale = max(g.abs().max().item(), 1e-3 * top)
        err_fast = (res[0][2][name] - g).abs().max().item() / scale
        err_lib = (res[1][2][name] - g).abs().max().item() / scale
        assert err_fast < max(5e-3, 50 * err_lib), (name, err_fast, err_lib)
    for (name, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        if "running" in name:
            assert torch.allclose(ba, bb, rtol=1e-3, atol=1e-5), name


def test_dropout_sites_share_one_generator_launch_per_step():
    """Inside begin_step_seeds / end_step_seeds every site gets its own element of one draw; outside, its own draw."""
    from alphatriangle_amd.ops import hip_learner

    dev = torch.device("cuda")
    torch.manual_seed(1)
    hip_learner.begin_step_seeds(dev)
    try:
        a, b = hip_learner._drop_seed(dev), hip_learner._drop_seed(dev)
        assert a.data_ptr() + 8 == b.data_ptr() and int(a) != int(b)
        for _ in range(hip_learner.SEEDS_PER_STEP):  # past the pool: falls back to a draw of its own
            last = hip_learner._drop_seed(dev)
        assert last.numel() == 1
    finally:
        hip_learner.end_step_seeds(dev)
    torch.manual_seed(1)
    hip_learner.begin_step_seeds(dev)
    again = int(hip_learner._drop_seed(dev))
    hip_learner.end_step_seeds(dev)
    assert again == int(a), "torch.manual_seed governs the pool"
    c = hip_learner._drop_seed(dev)
    assert c.numel() == 1 and c.data_ptr() != a.data_ptr()


def test_fp32_training_trajectory_is_the_same_on_both_gemm_paths():
    """Thirty fp32 learner steps from the same initial weights on the same batches: the losses of the bf16-matrix-core GEMM
    path follow those of the library's fp32 GEMMs (LEARNER_FP32_GEMM = 'bf16x3' against 'library')."""
    import alphatriangle_amd.models.model as model_mod
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.trainer import TensorBatch, Trainer

    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=64, COMPILE_MODEL=False, LEARNING_RATE=3e-4)
    B = 64
    gen = torch.Generator(device="cpu").manual_seed(5)
    batches = []
    for _ in range(30):
        pol = torch.rand(B, env.ACTION_DIM, generator=gen)
        batches.append(TensorBatch(
            grid=torch.randint(-1, 2, (B, 1, env.ROWS, env.COLS), generator=gen).float().to(dev), other=torch.rand(B, mc.OTHER_NN_INPUT_FEATURES_DIM, generator=gen).to(dev),
            policy_target=(pol / pol.sum(1, keepdim=True)).to(dev), returns=torch.empty(B).uniform_(-3, 3, generator=gen).to(dev),
            weights=torch.ones(B, device=dev), indices=torch.arange(B, device=dev)))
    curves = []
    try:
        for fast, low in ((True, False), (False, False), (False, True), (False, False)):
            model_mod.FP32_GEMM_ON_BF16_CORES = fast
            torch.manual_seed(9)
            net = NeuralNetwork(mc, env, tc, dev)
            for m in net._orig_model.modules():  # dropout masks would differ between two runs; everything else stays
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
                if isinstance(m, torch.nn.MultiheadAttention):
                    m.dropout = 0.0
            trainer = Trainer(net, tc, env, autocast_bf16=low)
            curves.append([float(trainer.train_step_tensors(tb)[0]["total_loss"]) for tb in batches])
    finally:
        model_mod.FP32_GEMM_ON_BF16_CORES = True
    a, b, c, b2 = (np.asarray(x) for x in curves)
    rel, rel_bf16, rel_rerun = np.abs(a - b) / np.abs(b), np.abs(c - b) / np.abs(b), np.abs(b2 - b) / np.abs(b)
    print(f"loss after 30 steps: bf16x3 {a[-1]:.6f}, library fp32 {b[-1]:.6f}, bf16 autocast {c[-1]:.6f}; largest relative difference to the "
          f"library run along the way: bf16x3 {rel.max():.2e}, bf16 autocast {rel_bf16.max():.2e}, the library run repeated {rel_rerun.max():.2e}")
    assert b[-1] < b[0], "the steps have to train"
    # two fp32 runs that differ in rounding drift apart step by step (the gradients of the first layers amplify forward
    # rounding about a thousandfold, see the fast-path test above); the three-pass path has to stay well inside 1 %
    assert rel.max() < 1e-2


@pytest.mark.parametrize("cin,cout,batch,h,w", [(128, 128, 256, 8, 15), (32, 64, 64, 8, 15), (64, 128, 8, 12, 21), (4, 32, 4, 5, 6), (128, 100, 16, 8, 15)])
def test_patch_free_conv_gemms_match_torch_convolution(cin, cout, batch, h, w):
    """3x3 / padding-1 convolution forward, data gradient and split-K weight gradient through gemm3_kernel's on-the-fly
    patch loaders (no patch matrix in memory) against an fp64 convolution."""
    from alphatriangle_amd.models.model import _split_count
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(53)
    dev = torch.device("cuda")
    x = torch.randn(cin, batch, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    g = torch.randn(cout, batch, h, w, device=dev)
    w2 = wt.reshape(cout, -1)
    assert hip_learner.conv3_supported(x, w2)
    out = hip_learner.conv3_mm(w2, x).view(cout, batch, h, w)
    n = batch * h * w
    s = _split_count(n)
    d_w = hip_learner.conv3_wgrad(g.view(cout, n), x, n // s).sum(0).view(cout, cin, 3, 3)
    d_x = None
    w_t = wt.flip(2, 3).permute(1, 0, 2, 3).reshape(cin, cout * 9)
    if hip_learner.conv3_supported(g, w_t):
        d_x = hip_learner.conv3_mm(w_t, g).view(cin, batch, h, w)
    torch.cuda.synchronize()

    xr = x.double().permute(1, 0, 2, 3).requires_grad_(True)
    wr = wt.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wr, padding=1)
    ref.backward(g.double().permute(1, 0, 2, 3))

    def rel(a, b):
        return (a.double() - b).abs().max().item() / b.abs().max().item()

    e_f, e_w = rel(out, ref.detach().permute(1, 0, 2, 3)), rel(d_w, wr.grad)
    e_x = rel(d_x, xr.grad.permute(1, 0, 2, 3)) if d_x is not None else 0.0
    print(f"[{cin}->{cout}, {batch} x {h}x{w}] forward {e_f:.2e}, weight gradient {e_w:.2e}, data gradient {e_x:.2e}")
    assert e_f < 2e-5 and e_w < 2e-5 and e_x < 2e-5
    assert d_x is not None or cout % 4 != 0


@pytest.mark.parametrize("cin,cout,batch,h,w", [(128, 128, 256, 8, 15), (64, 128, 8, 8, 15), (128, 64, 8, 8, 15), (32, 64, 5, 8, 15),
                                                (32, 128, 3, 8, 15), (128, 128, 4, 12, 21), (64, 64, 3, 5, 6), (128, 128, 2, 16, 16)])
def test_direct_conv_kernel_matches_torch_convolution(cin, cout, batch, h, w):
    """Board-per-workgroup 3x3 convolution (forward with the filter as it is, data gradient with the flipped, transposed
    filter) against torch's convolution in fp64, next to the GEMM-lowered form it replaces."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(61)
    dev = torch.device("cuda")
    x = torch.randn(cin, batch, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    assert hip_learner.conv3_cm_supported(x, cout)
    got = hip_learner.conv3_cm(x, hip_learner.conv3_pack(wt, False), cout)
    ref = torch.nn.functional.conv2d(x.permute(1, 0, 2, 3).double(), wt.double(), padding=1).permute(1, 0, 2, 3)
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    g = torch.randn(cout, batch, h, w, device=dev)
    if hip_learner.conv3_cm_supported(g, cin):
        dx = hip_learner.conv3_cm(g, hip_learner.conv3_pack(wt, True), cin)
        dref = torch.nn.functional.conv_transpose2d(g.permute(1, 0, 2, 3).double(), wt.double(), padding=1).permute(1, 0, 2, 3)
        derr = (dx.double() - dref).abs().max().item() / dref.abs().max().item()
    else:
        derr = 0.0
    print(f"[{cin}->{cout} batch {batch} {h}x{w}] forward {err:.2e}, data gradient {derr:.2e}")
    assert err < 2e-5 and derr < 2e-5


@pytest.mark.parametrize("cin,cout,batch,h,w", [(128, 128, 256, 8, 15), (64, 128, 7, 8, 15), (128, 64, 5, 8, 15), (32, 64, 300, 8, 15),
                                                (32, 128, 3, 5, 6), (64, 64, 2, 7, 9)])
def test_direct_conv_weight_gradient_matches_torch(cin, cout, batch, h, w):
    """Weight gradient from the padded-board kernel (three shifted copies of the input in LDS, slices of the boards added
    by a reduction launch) against autograd in fp64."""
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(71)
    dev = torch.device("cuda")
    x = torch.randn(cin, batch, h, w, device=dev)
    g = torch.randn(cout, batch, h, w, device=dev)
    assert hip_learner.conv3_wgrad_supported(g, x)
    got = hip_learner.conv3_wgrad_cm(g, x)
    wt = torch.zeros(cout, cin, 3, 3, device=dev, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.permute(1, 0, 2, 3).double(), wt, padding=1).backward(g.permute(1, 0, 2, 3).double())
    err = (got.double() - wt.grad).abs().max().item() / wt.grad.abs().max().item()
    print(f"[{cin}->{cout} batch {batch} {h}x{w}] weight gradient {err:.2e}")
    assert err < 2e-5


def test_prepacked_filters_equal_the_per_layer_packing_and_are_released():
    """AlphaTriangleNet.prepack_filters (one launch for all direct convolutions) gives the planes conv3_pack gives layer by
    layer, _ConvGemmCM uses them, and release_packed_filters leaves no stale copy behind."""
    from alphatriangle_amd.models.model import AlphaTriangleNet, GemmConv2d
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(101)
    dev = torch.device("cuda")
    net = AlphaTriangleNet(ModelConfig(), EnvConfig()).to(dev).train()
    net.prepack_filters()
    n = 0
    for m in net.modules():
        if isinstance(m, GemmConv2d) and hasattr(m.weight, "_atri_packed"):
            for dgrad, planes in m.weight._atri_packed.items():
                assert torch.equal(planes, hip_learner.conv3_pack(m.weight, dgrad)), (m, dgrad)
                n += 1
    assert n == 11, n  # 64 -> 128 and four 128 -> 128 layers in both forms, 32 -> 64 forward only
    grid = torch.randint(-1, 2, (8, 1, 8, 15), device=dev).float()
    other = torch.rand(8, 30, device=dev)
    pl_a, vl_a = net(grid, other)
    net.release_packed_filters()
    assert not any(hasattr(p, "_atri_packed") for p in net.parameters())
    for m in net.modules():  # same batch statistics for the second pass
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.momentum = 0.0
    torch.manual_seed(5)
    pl_b, vl_b = net(grid, other)
    torch.manual_seed(5)
    net.prepack_filters()
    pl_c, vl_c = net(grid, other)
    net.release_packed_filters()
    assert torch.equal(pl_b, pl_c) and torch.equal(vl_b, vl_c)


def test_conv_layer_on_the_direct_kernel_trains_like_the_gemm_lowered_layer():
    """_ConvGemmCM with DIRECT_CONV on and off: same output, input gradient and weight gradient."""
    import alphatriangle_amd.models.model as model_mod

    torch.manual_seed(67)
    dev = torch.device("cuda")
    x0 = torch.randn(64, 16, 8, 15, device=dev)
    w0 = torch.randn(128, 64, 3, 3, device=dev) / 24.0
    gy = torch.randn(128, 16, 8, 15, device=dev)
    res = []
    try:
        for direct in (True, False):
            model_mod.DIRECT_CONV = direct
            x, w = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True)
            y = model_mod._ConvGemmCM.apply(x, w, 1, 1)
            y.backward(gy)
            res.append((y.detach(), x.grad, w.grad))
    finally:
        model_mod.DIRECT_CONV = True
    for a, b, name in zip(res[0], res[1], ("y", "dx", "dw")):
        assert (a - b).abs().max().item() < 3e-5 * b.abs().max().item(), name


def _flat_keep_mask(seed: int, n: int, p: float, device) -> torch.Tensor:
    """Host re-statement of ``drop_hash32`` in csrc/hip/learner_fused_kernels.hip over flat element indices."""
    m32 = 0xFFFFFFFF
    s0, s1 = seed & m32, (seed >> 32) & m32
    thresh = int(float(np.float32(p)) * 4294967296.0)
    x = (torch.arange(n, device=device, dtype=torch.int64) * 0x9E3779B1 + s0) & m32
    x = ((x ^ (x >> 16)) * 0x85EBCA6B) & m32
    x = ((x ^ (x >> 13)) * 0xC2B2AE35) & m32
    x = ((x ^ (x >> 16)) + s1) & m32
    x = ((x ^ (x >> 15)) * 0x2C1B3C6D) & m32
    x = x ^ (x >> 12)
    return x >= thresh


@pytest.mark.parametrize("da,db", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
def test_fused_residual_dropout_matches_its_definition(da, db):
    """a + dropout(b) and its two gradients against the same arithmetic in PyTorch with the kernel's own keep mask."""
    from alphatriangle_amd.models.model import _AddDropout
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(3)
    dev = torch.device("cuda")
    p = 0.1
    a = torch.randn(120, 37, 128, device=dev).to(da).requires_grad_(True)
    b = torch.randn(120, 37, 128, device=dev).to(db).requires_grad_(True)
    assert hip_learner.add_dropout_supported(a, b)
    out, seed = hip_learner.add_dropout_forward(a.detach(), b.detach(), p)
    keep = _flat_keep_mask(int(seed.item()), a.numel(), p, dev).view(a.shape)
    assert abs(1.0 - keep.double().mean().item() - p) < 0.01
    scale = 1.0 / (1.0 - float(np.float32(p)))
    ref = a.detach().float() + torch.where(keep, b.detach().float() * scale, torch.zeros((), device=dev))
    assert out.dtype == da and (out.float() - ref.to(da).float()).abs().max().item() <= (0 if da == torch.float32 else 2 ** -6 * ref.abs().max().item())
    g = torch.randn_like(out)
    gb = hip_learner.dropout_backward(g, seed, p, db)
    gb_ref = torch.where(keep, g.float() * scale, torch.zeros((), device=dev)).to(db)
    assert gb.dtype == db and torch.equal(gb, gb_ref)
    # through autograd: the gradient of a is the incoming gradient, that of b is masked and scaled
    torch.manual_seed(11)
    y = _AddDropout.apply(a, b, p)
    ga, gb2 = torch.autograd.grad(y, [a, b], g)
    assert torch.equal(ga, g) and gb2.dtype == db
    assert abs((gb2 == 0).double().mean().item() - p) < 0.01


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act_id,p", [(1, 0.1), (2, 0.1), (1, 0.0)])
def test_fused_activation_dropout_matches_its_definition(dtype, act_id, p):
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(5)
    dev = torch.device("cuda")
    h = (torch.randn(120, 19, 256, device=dev) * 1.5).to(dtype)
    out, seed = hip_learner.act_dropout_forward(h, p, act_id)
    hd = h.double().requires_grad_(True)
    act = torch.relu(hd) if act_id == 1 else torch.nn.functional.gelu(hd)
    if p > 0:
        keep = _flat_keep_mask(int(seed.item()), h.numel(), p, dev).view(h.shape)
        ref = torch.where(keep, act / (1.0 - float(np.float32(p))), torch.zeros((), device=dev, dtype=torch.float64))
    else:
        assert seed is None
        ref = act
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert (out.double() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    g = torch.randn_like(h)
    dh = hip_learner.act_dropout_backward(h, g, seed, p, act_id)
    (ref * g.double()).sum().backward()
    assert (dh.double() - hd.grad).abs().max().item() < tol * max(1.0, hd.grad.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("d,b,h,w,p", [(128, 256, 8, 15, 0.1), (128, 3, 8, 15, 0.0), (64, 5, 5, 6, 0.3), (128, 2, 12, 21, 0.1)])
def test_tokens_kernel_matches_permute_add_dropout(dtype, d, b, h, w, p):
    """Channel-major body output -> tokens + positional table + dropout in one launch (and its backward) against the three ops
    it replaces, with the keep mask re-stated on the host."""
    from alphatriangle_amd.models.model import _TokensPe, sinusoid_table
    from alphatriangle_amd.ops import hip_learner

    torch.manual_seed(91)
    dev = torch.device("cuda")
    x = torch.randn(d, b, h, w, device=dev).to(dtype).requires_grad_(True)
    pe = sinusoid_table(h * w, d).to(dev)
    assert hip_learner.tokens_pe_supported(x, pe)
    hip_learner.begin_step_seeds(dev)
    try:
        tokens = _TokensPe.apply(x, pe, p)
        seed = int(hip_learner._seed_pool[hip_learner._dev_index(dev)][0][0]) if p > 0 else 0
    finally:
        hip_learner.end_step_seeds(dev)
    plain = x.detach().float().flatten(2).permute(2, 1, 0) + pe.unsqueeze(1)  # [HW, B, D]
    keep = _flat_keep_mask(seed, plain.numel(), p, dev).view(plain.shape) if p > 0 else torch.ones_like(plain, dtype=torch.bool)
    ref = torch.where(keep, plain / (1.0 - p), torch.zeros_like(plain))
    assert tokens.dtype == torch.float32 and tokens.shape == ref.shape
    assert torch.allclose(tokens, ref, rtol=1e-6, atol=1e-6)
    if p > 0:
        share = 1.0 - keep.float().mean().item()
        assert abs(share - p) < 0.02, share
    g = torch.randn_like(tokens)
    tokens.backward(g)
    dref = torch.where(keep, g / (1.0 - p), torch.zeros_like(g)).permute(2, 1, 0).reshape(d, b, h, w)
    assert x.grad.dtype == dtype
    assert torch.allclose(x.grad.float(), dref.to(dtype).float(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layer_norm_writing_batch_major_rows_matches_norm_then_permute(dtype):
    from alphatriangle_amd.models.model import _LayerNormFn

    torch.manual_seed(93)
    dev = torch.device("cuda")
    L, B, D = 120, 37, 128
    x = torch.randn(L, B, D, device=dev).to(dtype).requires_grad_(True)
    w = torch.rand(D, device=dev, requires_grad=True)
    bias = torch.randn(D, device=dev, requires_grad=True)
    y = _LayerNormFn.apply(x, w, bias, 1e-5, True)
    xr, wr, br = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), bias.detach().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5).permute(1, 0, 2).flatten(1)
    assert y.shape == (B, L * D)
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-5
    assert (y.double() - yr).abs().max().item() < tol
    g = torch.randn(B, L * D, device=dev)
    y.backward(g.to(y.dtype))
    yr.backward(g.to(y.dtype).double())
    assert (x.grad.double() - xr.grad).abs().max().item() < tol * 3
    assert (w.grad.double() - wr.grad).abs().max().item() < tol * 30 and (bias.grad.double() - br.grad).abs().max().item() < tol * 30


@pytest.mark.parametrize("low", [False, True])
def test_residual_fork_node_gives_the_gradients_of_separate_norm_and_skip(low):
    """EncoderLayer with the (x, LayerNorm(x)) fork as one autograd node (the skip gradient added inside the LayerNorm backward
    kernel) against the same layer with the two consumers of x left to autograd: same output, same gradients."""
    import copy

    from alphatriangle_amd.models.model import EncoderLayer

    torch.manual_seed(97)
    dev = torch.device("cuda")
    layer_a = EncoderLayer(d_model=128, nhead=4, dim_feedforward=256, activation="relu", batch_first=False, norm_first=True).to(dev).train()
    for m in layer_a.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    layer_a.self_attn.dropout = 0.0
    layer_b = copy.deepcopy(layer_a)
    layer_b.fused_norm_fork = False
    x0 = torch.randn(120, 16, 128, device=dev)
    g = torch.randn(120, 16, 128, device=dev)
    res = []
    for layer in (layer_a, layer_b):
        x = x0.clone().requires_grad_(True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=low):
            y = layer(x)
        y.backward(g.to(y.dtype))
        res.append((y.detach().float(), x.grad.float(), {n: p.grad.float() for n, p in layer.named_parameters()}))
    tol = 2e-2 if low else 1e-5
    assert torch.allclose(res[0][0], res[1][0], rtol=tol, atol=tol)
    assert torch.allclose(res[0][1], res[1][1], rtol=tol, atol=tol * max(1.0, res[1][1].abs().max().item()))
    for name, ga in res[0][2].items():
        gb = res[1][2][name]
        assert (ga - gb).abs().max().item() <= tol * max(1.0, gb.abs().max().item()), name


def test_encoder_layer_with_fused_dropout_trains_like_the_stock_modules_in_eval_and_differs_only_by_its_masks():
    """Evaluation mode: identical outputs with and without the fused dropout ops.  Training mode: the layer runs, the
    fused path really drops (outputs differ between two calls) and gradients reach every parameter."""
    from alphatriangle_amd.models.model import EncoderLayer

    torch.manual_seed(9)
    dev = torch.device("cuda")
    layer = EncoderLayer(128, 4, 256, dropout=0.1, activation="relu", norm_first=True).to(dev)
    x = torch.randn(120, 16, 128, device=dev)
    layer.eval()
    with torch.no_grad():
        layer.fused_dropout = True
        y1 = layer(x)
        layer.fused_dropout = False
        y2 = layer(x)
    assert (y1 - y2).abs().max().item() < 1e-5
    layer.train()
    layer.fused_dropout = True
    ya, yb = layer(x), layer(x)
    assert (ya - yb).abs().max().item() > 1e-3
    ya.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in layer.parameters())


def test_per_tree_kernels_at_scale_match_the_host_sum_tree():
    """6 000 touched leaves of a 250 000-leaf tree: the multi-workgroup path update (subtree walk + top rebuild) leaves
    exactly the sums of the native host sum tree (csrc/host/sumtree.h semantics: float64 sums over the same leaves);
    batches above one workgroup sample the same leaves as a host descent of that tree with the same random numbers, and
    the launch costs tens of microseconds, not hundreds."""
    from alphatriangle_amd.ops import hip_learner

    dev = torch.device("cuda")
    cap = 250_000
    P = 1 << (cap - 1).bit_length()
    gen = torch.Generator().manual_seed(3)
    leaves = torch.rand(cap, generator=gen, dtype=torch.float64) + 0.01
    tree = torch.zeros(2 * P, dtype=torch.float64, device=dev)
    tree[P : P + cap] = leaves.to(dev)
    hip_learner.per_rebuild(tree, P)
    # touch 6 000 leaves (with repeats, as PER updates have), some clustered in one subtree to force the list fallback
    idx = torch.cat([torch.randint(0, cap, (3000,), generator=gen), torch.randint(5000, 5900, (3000,), generator=gen)]).to(torch.int32)
    new = torch.rand(6000, generator=gen, dtype=torch.float64) * 3.0 + 0.01
    ref = leaves.clone()
    for i, v in zip(idx.tolist(), new.tolist()):
        ref[i] = v  # last write wins, as on the device (same values for repeated indices below)
    # make repeated indices carry the same value so the result does not depend on write order
    new = ref[idx.long()]
    d_idx = idx.to(dev)
    tree[P + d_idx.long()] = new.to(dev)
    hip_learner.per_update_paths(tree, P, d_idx, None, d_idx.numel())
    torch.cuda.synchronize()
    # host reference: the full float64 tree over the updated leaves
    host = torch.zeros(2 * P, dtype=torch.float64)
    host[P : P + cap] = ref
    level = P >> 1
    while level >= 1:
        host[level : 2 * level] = host[2 * level : 4 * level : 2] + host[2 * level + 1 : 4 * level : 2]
        level >>= 1
    got = tree.cpu()
    assert torch.equal(got[1:], host[1:]), f"largest deviation {(got[1:] - host[1:]).abs().max().item():.3e}"
    # timing of the update (steady state, after a warm-up)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        hip_learner.per_update_paths(tree, P, d_idx, None, d_idx.numel())
    start.record()
    for _ in range(20):
        hip_learner.per_update_paths(tree, P, d_idx, None, d_idx.numel())
    end.record()
    torch.cuda.synchronize()
    us = start.elapsed_time(end) * 1e3 / 20
    print(f"per_update_paths, 6000 leaves of a {cap}-leaf tree: {us:.1f} us")
    assert us < 40.0  # measured 11.8 us; the single-workgroup form took ~200 us
    # sampling a batch of 4096 (above one workgroup): same leaves as a host descent with the same random numbers
    B = 4096
    rnd = torch.rand(B, generator=gen)
    out_idx = torch.empty(B, dtype=torch.int32, device=dev)
    out_w = torch.empty(B, dtype=torch.float32, device=dev)
    hip_learner.per_sample(tree, P, cap, B, rnd.to(dev), 0.6, out_idx, out_w)
    torch.cuda.synchronize()
    total = host[1].item()
    want = []
    for i in range(B):
        value = (i + float(rnd[i])) * (total / B)
        node = 1
        while node < P:
            left = host[2 * node].item()
            if value <= left:
                node = 2 * node
            else:
                value -= left
                node = 2 * node + 1
        want.append(min(node - P, cap - 1))
    assert out_idx.cpu().tolist() == want
    probs = ref[torch.tensor(want)] / total
    w_ref = (cap * probs) ** -0.6
    w_ref = (w_ref / w_ref.max()).float()
    assert torch.allclose(out_w.cpu(), w_ref, rtol=1e-5, atol=1e-6)
