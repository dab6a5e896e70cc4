"""Multi-process runtime on the gloo backend (CPU): collectives, data-parallel learner, role split."""

from __future__ import annotations

import os
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from alphatriangle_amd.config import EnvConfig, ModelConfig, TrainConfig
from alphatriangle_amd.parallel.dist import RoleLayout, init_distributed
from alphatriangle_amd.parallel.experience import ExperienceInbox, ExperienceOutbox, RecordLayout


def _spawn(fn, world, *args):
    with tempfile.TemporaryDirectory() as d:
        init = f"file://{d}/store"
        mp.spawn(fn, args=(world, init, *args), nprocs=world, join=True)


def _collectives(rank, world, init):
    torch.set_num_threads(1)
    ctx = init_distributed(rank, world, rank, backend="gloo", init_method=init, timeout_s=60)
    g = torch.full((10,), float(rank + 1))
    ctx.grad_sync(g)
    assert torch.allclose(g, torch.full((10,), 3.0))
    w = torch.arange(6.0) if rank == 0 else torch.zeros(6)
    ctx.broadcast_weights(w)
    assert torch.equal(w, torch.arange(6.0))
    blocks = ctx.all_gather_records(torch.full((2, 3), float(rank)))
    assert blocks.shape == (2, 2, 3) and blocks[1].eq(1).all() and blocks[0].eq(0).all()
    assert ctx.all_reduce_scalar(rank + 1.0, "max") == 2.0 and ctx.all_reduce_scalar(1.0, "sum") == 2.0
    # control plane: host-side agreement and the key/value store
    assert ctx.agree([float(rank), 5.0 - rank], "min") == [0.0, 4.0]
    assert ctx.control is not None
    ctx.control.add("hits", 1)
    ctx.barrier()
    assert ctx.control.counter("hits") == 2 and ctx.control.peek("absent", -7) == -7
    ctx.control.put(f"v/{rank}", 10 + rank)
    ctx.barrier()
    assert ctx.control.peek(f"v/{1 - rank}") == 11 - rank
    ctx.barrier()
    ctx.shutdown()


def test_collectives_two_ranks():
    _spawn(_collectives, 2)


def _tiny_cfg():
    env = EnvConfig(ROWS=3, COLS=3, PLAYABLE_RANGE_PER_ROW=[(0, 3)] * 3, NUM_SHAPE_SLOTS=1)
    mc = ModelConfig(CONV_FILTERS=[4], CONV_KERNEL_SIZES=[3], CONV_STRIDES=[1], CONV_PADDING=[1], NUM_RESIDUAL_BLOCKS=0,
                     USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0, FC_DIMS_SHARED=[8], POLICY_HEAD_DIMS=[8], VALUE_HEAD_DIMS=[8],
                     OTHER_NN_INPUT_FEATURES_DIM=14, USE_BATCH_NORM=False)
    tc = TrainConfig(BATCH_SIZE=8, BUFFER_CAPACITY=100, MIN_BUFFER_SIZE_TO_TRAIN=8, DEVICE="cpu", MAX_TRAINING_STEPS=10, USE_PER=False,
                     LEARNING_RATE=1e-2)
    return env, mc, tc


def _make_batch(n, seed):
    from alphatriangle_amd.rl.core.trainer import TensorBatch

    g = torch.Generator().manual_seed(seed)
    pol = torch.rand(n, 9, generator=g)
    return TensorBatch(torch.rand(n, 1, 3, 3, generator=g), torch.rand(n, 14, generator=g), pol / pol.sum(1, keepdim=True),
                       torch.empty(n).uniform_(-3, 3, generator=g), torch.ones(n))


def _dp_train(rank, world, init, out_path):
    from alphatriangle_amd.models import NeuralNetwork
    from alphatriangle_amd.rl.core.trainer import TensorBatch, Trainer

    torch.set_num_threads(1)
    ctx = init_distributed(rank, world, rank, backend="gloo", init_method=init, timeout_s=60)
    env, mc, tc = _tiny_cfg()
    torch.manual_seed(0)
    net = NeuralNetwork(mc, env, tc, torch.device("cpu"))
    tr = Trainer(net, tc, env, grad_sync=ctx.grad_sync, world_size=world)
    full = _make_batch(8, 123)
    lo, hi = rank * 4, rank * 4 + 4
    shard = TensorBatch(full.grid[lo:hi], full.other[lo:hi], full.policy_target[lo:hi], full.returns[lo:hi], full.weights[lo:hi])
    for _ in range(3):
        tr.train_step_tensors(shard)
    if rank == 0:
        torch.save(tr.flat.flat_param.clone(), out_path)
    # replicas stay bit-identical without any weight broadcast
    other = tr.flat.flat_param.clone()
    ctx.broadcast_weights(other, src=0)
    assert torch.equal(other, tr.flat.flat_param)
    ctx.shutdown()


def test_data_parallel_learner_equals_single_process(tmp_path):
    from alphatriangle_amd.models import NeuralNetwork
    from alphatriangle_amd.rl.core.trainer import Trainer

    out = str(tmp_path / "dp.pt")
    _spawn(_dp_train, 2, out)
    env, mc, tc = _tiny_cfg()
    torch.manual_seed(0)
    net = NeuralNetwork(mc, env, tc, torch.device("cpu"))
    tr = Trainer(net, tc, env, fused=True)
    full = _make_batch(8, 123)
    for _ in range(3):
        tr.train_step_tensors(full)
    dp = torch.load(out, weights_only=True)
    assert torch.allclose(dp, tr.flat.flat_param, atol=1e-6), (dp - tr.flat.flat_param).abs().max()


def _split_roles(rank, world, init):
    import time

    torch.set_num_threads(1)
    ctx = init_distributed(rank, world, rank, backend="gloo", layout="split", num_learners=1, init_method=init, timeout_s=60)
    assert ctx.is_learner == (rank == 0) and ctx.is_actor == (rank != 0) and ctx.num_learners == 1
    ctx.warm_up()  # includes one tiny transfer per point-to-point communicator
    layout = RecordLayout(grid=9, other=14, policy=9)
    if ctx.is_actor:
        out = ExperienceOutbox(ctx, layout, torch.device("cpu"), max_rows=8)
        n = rank + 1  # actor 1 ships 2 records, actor 2 ships 3; then an empty move
        recs = layout.pack(torch.full((n, 1, 3, 3), float(rank)), torch.zeros(n, 14), torch.full((n, 9), 1 / 9), torch.full((n,), 0.5 * rank))
        assert out.ship(recs, {"episodes": 1, "simulations": 10 * rank, "moves": n}) == n
        assert out.ship(None, {"episodes": 0, "simulations": 0, "moves": 1}) == 0
        out.close()
    else:
        inbox = ExperienceInbox(ctx, layout, torch.device("cpu"))
        got, deadline = [], time.monotonic() + 30
        while not inbox.drained() and time.monotonic() < deadline:
            got += inbox.poll()
            time.sleep(0.01)
        assert inbox.drained() and len(got) == 4
        rows = torch.cat([s.records for s in got])
        assert rows.shape == (5, layout.stride)  # exactly the rows that exist: no fixed-size padding travels
        g, o, p, r = layout.unpack(rows)
        assert sorted(r.tolist()) == [0.5, 0.5, 1.0, 1.0, 1.0] and torch.allclose(p.sum(1), torch.ones(5))
        assert sum(s.counters["simulations"] for s in got) == 30 and sum(s.counters["moves"] for s in got) == 7
    # learner -> actors weight push (collective form)
    w = torch.arange(5.0) if ctx.is_learner else torch.zeros(5)
    ctx.broadcast_weights(w)
    assert torch.equal(w, torch.arange(5.0))
    ctx.grad_sync(torch.ones(3))  # single learner: no-op, must not hang
    ctx.barrier()
    ctx.shutdown()


def test_split_roles_experience_shipping_three_ranks():
    _spawn(_split_roles, 3)


def test_role_layout():
    r = RoleLayout.build(8, "split", 1)
    assert r.learner_ranks == [0] and r.actor_ranks == list(range(1, 8)) and not r.colocated
    c = RoleLayout.build(4, "colocated", 1)
    assert c.learner_ranks == c.actor_ranks == [0, 1, 2, 3] and c.colocated
    assert RoleLayout.build(1, "split", 1).colocated


def _split_training(rank, world, init, root):
    """One learner + two host actors run the real split-role loop end to end (tiny world, gloo)."""
    from alphatriangle_amd.config import AlphaTriangleMCTSConfig, SystemConfig
    from alphatriangle_amd.stats import TrieyeConfig
    from alphatriangle_amd.training.runner import run_training

    torch.set_num_threads(1)
    ctx = init_distributed(rank, world, rank, backend="gloo", layout="split", num_learners=1, init_method=init, timeout_s=120)
    env, mc, tc = _tiny_cfg()
    tc = tc.model_copy(update={"MAX_TRAINING_STEPS": 6, "WORKER_UPDATE_FREQ_STEPS": 2, "CHECKPOINT_SAVE_FREQ_STEPS": 4,
                               "NUM_SELF_PLAY_WORKERS": 1, "N_STEP_RETURNS": 2, "AUTO_RESUME_LATEST": False})
    sysc = SystemConfig(ENGINE="cpu", ROLE_LAYOUT="split", NUM_LEARNER_RANKS=1, TRAIN_STEPS_PER_ROUND=2, SHIP_CAPACITY=64, DIST_BACKEND="gloo")
    mcts = AlphaTriangleMCTSConfig(max_simulations=4, max_depth=3, mcts_batch_size=2)
    run = "split" if rank == 0 else f"split_rank{rank}"
    tcfg = TrieyeConfig(app_name="t", run_name=run)
    tcfg.persistence.ROOT_DATA_DIR = root
    tcfg.persistence.RUN_NAME = run
    code = run_training("WARNING", tc, tcfg, False, model_config_override=mc, mcts_config_override=mcts, env_config_override=env,
                        system_config_override=sysc, dist_context=ctx)
    assert code == 0, f"rank {rank} exited with {code}"
    if rank == 0:
        ckpts = [f for _, _, fs in os.walk(os.path.join(root, "t", "runs", "split", "checkpoints")) for f in fs]
        assert any(f.endswith(".pkl") for f in ckpts), "the lead learner must have written a checkpoint"
    ctx.shutdown()


def test_split_role_training_loop_three_ranks():
    with tempfile.TemporaryDirectory() as root:
        _spawn(_split_training, 3, root)


# ----------------------------------------------------------------------------------------------------------------------
# Peer loss whose failure surfaces later than the call (what RCCL does: a send / receive returns once it is enqueued).
class _DictStore:
    """In-process stand-in of the rendezvous store (add / set / get / check)."""

    def __init__(self):
        self.d: dict[str, bytes] = {}

    def add(self, key, amount):
        v = int(self.d.get(key, b"0")) + int(amount)
        self.d[key] = str(v).encode()
        return v

    def set(self, key, value):
        self.d[key] = str(value).encode()

    def get(self, key):
        return self.d[key]

    def check(self, keys):
        return all(k in self.d for k in keys)


class _StubTransfer:
    def __init__(self, completes: bool, on_finish=None):
        self.completes, self.on_finish, self.finished = completes, on_finish, False

    def done(self):
        return self.completes

    def finish(self, timeout_s=0.0):
        self.finished = True
        if self.on_finish:
            self.on_finish()


class _StubTransport:
    """post_* always succeeds at once; whether the transfer ever completes is decided by ``alive``."""

    def __init__(self):
        self.alive, self.aborts, self.recvs, self.sends = True, 0, 0, 0

    def post_recv(self, tensor, peer):
        self.recvs += 1
        return _StubTransfer(self.alive, on_finish=lambda: tensor.fill_(float(peer)))

    def post_send(self, tensor, peer):
        self.sends += 1
        return _StubTransfer(self.alive)

    def abort(self):
        self.aborts += 1


def _stub_ctx(rank, control):
    from types import SimpleNamespace

    layout = RoleLayout.build(3, "split", 1)
    return SimpleNamespace(rank=rank, control=control, layout=layout, backend="gloo", transfer_timeout_s=0.2,
                           pair_transport=lambda learner, actor: (_ for _ in ()).throw(AssertionError("transport must be injected")))


def _announce_shipment(cp, actor, k, rows):
    slot = k % 64
    cp.put(f"op/{actor}/{slot}/kind", 1)
    cp.put(f"op/{actor}/{slot}/rows", rows)
    for f in ("episodes", "simulations", "moves"):
        cp.put(f"op/{actor}/{slot}/{f}", 1)
    cp.add(f"op/{actor}/n", 1)


def test_learner_survives_a_transfer_whose_failure_surfaces_after_the_call():
    """Actor 2 announces a shipment, its receive is posted without error - and never completes (the actor died after
    announcing).  The learner must give up after the transfer deadline, tear down THAT pair's communicator once, retire
    the actor, and keep serving actor 1."""
    import time

    from alphatriangle_amd.parallel.dist import ControlPlane

    cp = ControlPlane(_DictStore())
    layout = RecordLayout(grid=9, other=14, policy=9)
    t1, t2 = _StubTransport(), _StubTransport()
    inbox = ExperienceInbox(_stub_ctx(0, cp), layout, torch.device("cpu"), silence_s=300.0, heartbeat_stale_s=300.0,
                            transports={1: t1, 2: t2})
    for a in (1, 2):
        cp.add(f"hb/{a}", 1)
    _announce_shipment(cp, 1, 1, 3)
    _announce_shipment(cp, 2, 1, 4)
    t2.alive = False  # from here on nothing actor 2 is part of ever completes
    t0 = time.monotonic()
    got = inbox.poll()
    assert time.monotonic() - t0 < 5.0  # bounded by the 0.2 s transfer deadline, not by a library time-out
    assert [s.actor for s in got] == [1] and got[0].records.shape == (3, layout.stride) and got[0].records.eq(1).all()
    assert t2.recvs == 1 and t2.aborts == 1 and t1.aborts == 0
    retired = {f.rank: f.retired for f in inbox.feeds}
    assert retired == {1: False, 2: True} and inbox.active_actors() == 1
    assert cp.peek("ack/1") == 1 and cp.peek("ack/2", 0) == 0  # the lost shipment is not acknowledged
    # the survivor keeps being served, the retired pair is never touched again
    _announce_shipment(cp, 1, 2, 2)
    _announce_shipment(cp, 2, 2, 2)
    got = inbox.poll()
    assert [s.actor for s in got] == [1] and t2.recvs == 1 and t2.aborts == 1


def test_no_receive_is_posted_for_an_actor_whose_heartbeat_is_stale():
    """A stale heartbeat retires the actor BEFORE anything is posted on its communicator (which therefore stays intact)."""
    import time

    from alphatriangle_amd.parallel.dist import ControlPlane

    cp = ControlPlane(_DictStore())
    layout = RecordLayout(grid=9, other=14, policy=9)
    t1, t2 = _StubTransport(), _StubTransport()
    inbox = ExperienceInbox(_stub_ctx(0, cp), layout, torch.device("cpu"), silence_s=300.0, heartbeat_stale_s=0.15,
                            transports={1: t1, 2: t2})
    for a in (1, 2):
        cp.add(f"hb/{a}", 1)
    assert inbox.poll() == []          # both heartbeats seen
    time.sleep(0.25)
    cp.add("hb/1", 1)                  # actor 1 is alive, actor 2 has stopped beating
    _announce_shipment(cp, 1, 1, 1)
    _announce_shipment(cp, 2, 1, 1)
    got = inbox.poll()
    assert [s.actor for s in got] == [1]
    assert t2.recvs == 0 and t2.aborts == 0 and {f.rank: f.retired for f in inbox.feeds} == {1: False, 2: True}


def test_a_weight_delivery_nobody_takes_costs_only_its_own_pair():
    """Actor 2 asks for weights and dies: the learner's send is posted, never completes, and is given up on later polls;
    actor 1's request is served."""
    import time

    from alphatriangle_amd.parallel.dist import ControlPlane

    cp = ControlPlane(_DictStore())
    layout = RecordLayout(grid=9, other=14, policy=9)
    t1, t2 = _StubTransport(), _StubTransport()
    wire = torch.arange(5.0)
    inbox = ExperienceInbox(_stub_ctx(0, cp), layout, torch.device("cpu"), silence_s=300.0, heartbeat_stale_s=300.0,
                            weights_provider=lambda wanted: wire if wanted <= 1 else None, transports={1: t1, 2: t2})
    for a in (1, 2):
        cp.add(f"hb/{a}", 1)
        slot = 1 % 64
        cp.put(f"op/{a}/{slot}/kind", 2)
        cp.put(f"op/{a}/{slot}/rows", 1)  # version 1
        cp.add(f"op/{a}/n", 1)
    t2.alive = False
    assert inbox.poll() == [] and t1.sends == 1 and t2.sends == 1 and inbox.weights_served == 2
    time.sleep(0.3)
    inbox.poll()
    assert t2.aborts == 1 and t1.aborts == 0 and {f.rank: f.retired for f in inbox.feeds} == {1: False, 2: True}
    # a request for a version that is not published yet stays queued instead of being answered with stale weights
    cp.put(f"op/1/{2 % 64}/kind", 2)
    cp.put(f"op/1/{2 % 64}/rows", 2)
    cp.add("op/1/n", 1)
    inbox.poll()
    assert t1.sends == 1 and inbox.feeds[0].taken == 1
