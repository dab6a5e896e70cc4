"""HIP game / tree engine vs the host C++ engine (bit-exact by construction; needs an MI355X)."""

from __future__ import annotations

import random

import numpy as np
import pytest
import torch

from alphatriangle_amd import native
from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
from alphatriangle_amd.env import GameState

pytestmark = pytest.mark.gpu


def _engine(env, mcts, games, leaves, other_dim=30, compact=False):
    from alphatriangle_amd.ops.engine import GpuTreeEngine

    dev = torch.device("cuda")
    logits = torch.zeros(games * leaves, env.ACTION_DIM, dtype=torch.float32, device=dev)
    values = torch.zeros(games * leaves, dtype=torch.float32, device=dev)
    return GpuTreeEngine(env, mcts, dev, games, leaves, other_dim, logits, values, seed=1, compact=compact)


def _packed(states):
    return torch.from_numpy(np.stack([np.frombuffer(s.core.pack(), dtype=np.uint8) for s in states]).copy())


_WIDE_ENV = EnvConfig(ROWS=18, COLS=18, PLAYABLE_RANGE_PER_ROW=[(1, 17)] * 18)   # 324 cells: six occupancy words
_HUGE_ENV = EnvConfig(ROWS=20, COLS=30, PLAYABLE_RANGE_PER_ROW=[(max(0, 4 - r, r - 15), 30 - max(0, 4 - r, r - 15)) for r in range(20)])


@pytest.mark.parametrize("env", [EnvConfig(), EnvConfig(ROWS=5, COLS=6, PLAYABLE_RANGE_PER_ROW=[(0, 6)] * 5, NUM_SHAPE_SLOTS=2), EnvConfig.large(),
                                 _WIDE_ENV, _HUGE_ENV], ids=["8x15", "5x6", "12x21", "18x18-wide", "20x30-wide"])
def test_step_valid_features_bit_exact(env):
    """Boards above 256 cells run the wide build of both engines (160-byte states, ``atriw_*`` entry points)."""
    rng = random.Random(5)
    other_dim = 8 * env.NUM_SHAPE_SLOTS + 6
    eng = _engine(env, AlphaTriangleMCTSConfig(max_simulations=4), 1, 1, other_dim)
    H = native.host(env.ROWS * env.COLS)
    assert eng.wide == (env.ROWS * env.COLS > 256) and eng.state_bytes == H.STATE_BYTES
    before, actions, after, rewards, dones = [], [], [], [], []
    for seed in range(6):
        g = GameState(env, initial_seed=seed)
        while not g.is_over() and len(before) < 400:
            a = rng.choice(g.valid_actions())
            before.append(g.copy())
            r, d = g.step(a)
            actions.append(a)
            after.append(g.copy())
            rewards.append(r)
            dones.append(d)
    n = len(before)
    st = _packed(before).cuda()
    # valid masks and features of the pre-move states
    mask = eng.valid_mask(st).cpu().numpy().astype(bool)
    grid, other = eng.features(st)
    for i in range(n):
        assert np.array_equal(mask[i], before[i].valid_mask())
    hg, ho = H.extract_features_batch([s.core for s in before], 1, other_dim, env.ROWS, env.COLS)
    assert np.array_equal(grid.cpu().numpy(), hg.reshape(n, -1))
    assert np.array_equal(other.cpu().numpy(), ho)
    # stepping
    rew, done = eng.step_states(st, torch.tensor(actions, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(st.cpu().numpy(), _packed(after).numpy())
    assert np.array_equal(rew.cpu().numpy(), np.asarray(rewards, dtype=np.float32))
    assert np.array_equal(done.cpu().numpy().astype(bool), np.asarray(dones))


class _SyntheticNet:
    """Deterministic stand-in network evaluated on the CPU in float64, shared by both searches."""

    def __init__(self, cells, other_dim, action_dim):
        rng = np.random.default_rng(11)
        self.w1 = rng.normal(size=(cells, 8))
        self.w2 = rng.normal(size=(other_dim, 8))
        self.w3 = rng.normal(size=(8, action_dim))
        self.w4 = rng.normal(size=8)

    def __call__(self, grid, other):
        h = np.tanh(grid.reshape(grid.shape[0], -1).astype(np.float64) @ self.w1 * 0.1 + other.astype(np.float64) @ self.w2)
        logits = np.round(h @ self.w3 * 64.0) / 64.0
        values = np.round(np.tanh(h @ self.w4) * 3.0 * 256.0) / 256.0
        return logits.astype(np.float32), values.astype(np.float32)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("leaves,large", [(1, False), (3, False), (1, True), (1, "wide"), (2, "wide")])
def test_tree_search_equals_host_search(leaves, large, compact):
    # the large board (756 actions) takes the strided edge-scan path; "wide" (18x18, 972 actions) the 160-byte-state build
    env = _WIDE_ENV if large == "wide" else EnvConfig.large() if large else EnvConfig()
    sims = 48
    mcts = AlphaTriangleMCTSConfig(max_simulations=sims, max_depth=6, dirichlet_epsilon=0.0, mcts_batch_size=leaves)
    G = 6
    eng = _engine(env, mcts, G, leaves, compact=compact)  # compact: leaves that need the network get dense rows
    net = _SyntheticNet(env.ROWS * env.COLS, 30, env.ACTION_DIM)
    H = native.host(env.ROWS * env.COLS)
    games = []
    rng = random.Random(3)
    for i in range(G):
        g = GameState(env, initial_seed=100 + i)
        for _ in range(i * 2):  # different depths into the game
            if g.is_over():
                break
            g.step(rng.choice(g.valid_actions()))
        games.append(g)

    def device_eval():
        lg, vl = net(eng.nn_grid.cpu().numpy(), eng.nn_other.cpu().numpy())
        eng.logits.copy_(torch.from_numpy(lg).cuda())
        eng.values.copy_(torch.from_numpy(vl).cuda())

    eng.set_roots(_packed(games).cuda())
    eng.prepare_roots()
    device_eval()
    eng.init_roots()
    waves = (sims + leaves - 1) // leaves
    for w in range(waves):
        eng.wave(consume=w > 0, select=True)
        if compact:  # the row counter equals the number of slots that asked for an evaluation, and rows are a permutation
            kinds = eng.slot_kind.cpu().numpy()
            rows = eng.slot_row.cpu().numpy()[kinds == 1]
            assert int(eng.live.item()) == len(rows) and sorted(rows.tolist()) == list(range(len(rows)))
        device_eval()
    eng.wave(consume=True, select=False)
    dv = eng.root_visits().cpu().numpy()
    torch.cuda.synchronize()

    def host_eval(core_states):
        hg, ho = H.extract_features_batch(core_states, 1, 30, env.ROWS, env.COLS)
        return net(hg, ho)

    params = mcts.to_search_params(leaf_batch=leaves, module=H)
    # the device runs `waves` full waves; give the host the same simulation budget
    params.max_simulations = waves * leaves
    for i, g in enumerate(games):
        tree = H.SearchTree(g.geometry, mcts.capacity(), 0)
        hv = tree.search(g.core, params, host_eval, -1)
        assert np.array_equal(hv, dv[i]), f"game {i}: host {np.nonzero(hv)[0][:8]} vs device {np.nonzero(dv[i])[0][:8]}"
        assert hv.sum() > 0


def _actor(games=48, sims=12, graphs=True, leaves=1, seed=7, dtype="bf16", model=None, hip_engine=None, env=None, compact=True):
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
    from alphatriangle_amd.rl.self_play.gpu_actor import GpuSelfPlayActor

    torch.manual_seed(0)
    dev = torch.device("cuda")
    env, mc = env or EnvConfig(), model or ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=20000, MIN_BUFFER_SIZE_TO_TRAIN=64, BATCH_SIZE=32, N_STEP_RETURNS=3, MAX_TRAINING_STEPS=1000)
    sc = SystemConfig(GAMES_PER_GPU=games, LEAVES_PER_TREE=leaves, USE_HIP_GRAPHS=graphs, INFERENCE_DTYPE=dtype, COMPACT_LEAF_BATCH=compact)
    mcts = AlphaTriangleMCTSConfig(max_simulations=sims, max_depth=5)
    net = NeuralNetwork(mc, env, tc, dev, use_hip_engine=hip_engine, inference_dtype=dtype)
    replay = DeviceReplayBuffer(tc, env, mc, dev, sc)
    actor = GpuSelfPlayActor(env, mc, mcts, tc, sc, net, replay, dev, seed=seed)
    return actor, replay, tc


def test_play_move_invariants_and_replay_flow():
    actor, replay, tc = _actor()
    G = actor.G
    total_added = 0
    for move in range(6):
        rep = actor.play_move()
        assert rep.moves == G
        assert rep.simulations >= G * 12
        total_added += rep.experiences_added
        assert len(replay) == total_added
        assert (actor.tree.n_nodes <= actor.tree.C).all()
    # with 3-step returns the first experiences appear on the 4th move (or earlier when games end)
    assert total_added >= G * 2
    n = len(replay)
    pol = replay.policy[:n]
    assert torch.allclose(pol.sum(1), torch.ones(n, device=pol.device), atol=1e-4)
    assert torch.isfinite(replay.ret[:n]).all()
    assert abs(replay.total_priority() - n * 1.0) < 1e-6  # every new leaf at max priority 1.0
    batch = replay.sample(32, current_train_step=0)
    assert batch is not None and batch.grid.shape == (32, 1, 8, 15) and batch.policy_target.shape == (32, 360)
    assert (batch.indices >= 0).all() and (batch.indices < n).all()
    assert ((batch.weights > 0) & (batch.weights <= 1.0 + 1e-6)).all()


def test_gpu_actor_plays_a_board_above_256_cells():
    """18x18 board: wide device engine (tree, commit, re-root, replay writes), leaves through the PyTorch module because no
    HIP inference engine tiles that board; replay rows and search policies keep their invariants."""
    from alphatriangle_amd.rl.self_play.gpu_actor import ModuleEvaluator

    actor, replay, tc = _actor(games=24, sims=10, dtype="fp32", model=ModelConfig.cnn_only(), env=_WIDE_ENV)
    assert actor.tree.wide and isinstance(actor.evaluator, ModuleEvaluator)
    total = 0
    for _ in range(6):
        rep = actor.play_move()
        assert rep.moves == actor.G and rep.simulations >= actor.G * 10
        total += rep.experiences_added
    n = len(replay)
    assert n == total and n >= actor.G * 2
    pol = replay.policy[:n]
    assert pol.shape[1] == _WIDE_ENV.ACTION_DIM and torch.allclose(pol.sum(1), torch.ones(n, device=pol.device), atol=1e-4)
    assert torch.isfinite(replay.ret[:n]).all()
    grid = replay.grid[:n].view(n, 18, 18)
    assert ((grid == 1) | (grid == 0) | (grid == -1)).all() and (grid[:, :, 0] == -1).all()  # death columns of the board
    batch = replay.sample(16, current_train_step=0)
    assert batch is not None and batch.grid.shape == (16, 1, 18, 18)


def test_per_game_step_metrics_and_slot_census():
    """One metric sample per game step (the CPU worker's per-step events) and the leaf-slot census of the move."""
    actor, _, _ = _actor(games=32, sims=10)
    rep = actor.play_move()
    pg = rep.per_game
    assert set(pg) == {"step_reward", "triangles_cleared_step", "current_score"}
    assert pg["step_reward"].shape == (actor.G,) and np.isfinite(pg["step_reward"]).all()
    assert abs(pg["step_reward"].mean() - rep.mean_step_reward) < 1e-4
    assert int(pg["triangles_cleared_step"].sum()) == rep.triangles_cleared
    c = rep.slot_census
    assert sum(c.values()) == actor.G * actor.waves_per_move and c["expand"] > 0


@pytest.mark.parametrize("dtype,fc_splits", [("fp32", 0), ("fp32", 16), ("bf16", 0), ("bf16", 32)])
def test_compacted_leaf_batch_plays_the_same_games(dtype, fc_splits, monkeypatch):
    """COMPACT_LEAF_BATCH: the engine kernels skip the rows beyond the device-side live count.  Same seeds with and
    without it must give the same moves, rewards and replay contents - also once terminal leaves thin the batch out.
    With the shared FC allowed to split K by the live count (fc_splits > 0) the partial sums are added in another
    order in thin waves: moves, rewards and search policies still have to match, the bootstrapped returns to 1e-4
    (fp32 engine) / 0.05 (bf16 engine, where a last-bit change before the bf16 rounding can move a value by 2^-8)."""
    monkeypatch.setenv("ATRI_FC_MAX_SPLITS", str(fc_splits))
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
    from alphatriangle_amd.rl.self_play.gpu_actor import GpuSelfPlayActor

    dev = torch.device("cuda")
    env = EnvConfig(ROWS=5, COLS=6, PLAYABLE_RANGE_PER_ROW=[(0, 6)] * 5, NUM_SHAPE_SLOTS=2)  # small board: games end within a few moves
    mc = ModelConfig(OTHER_NN_INPUT_FEATURES_DIM=8 * 2 + 6)
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=20000, MIN_BUFFER_SIZE_TO_TRAIN=64, BATCH_SIZE=32, N_STEP_RETURNS=3, MAX_TRAINING_STEPS=1000)
    mcts = AlphaTriangleMCTSConfig(max_simulations=24, max_depth=6)
    runs = []
    for compact in (False, True):
        torch.manual_seed(3)
        sc = SystemConfig(GAMES_PER_GPU=80, INFERENCE_DTYPE=dtype, COMPACT_LEAF_BATCH=compact)
        net = NeuralNetwork(mc, env, tc, dev, inference_dtype=dtype)
        assert net.engine is not None and net.engine.dtype == dtype
        replay = DeviceReplayBuffer(tc, env, mc, dev, sc)
        actor = GpuSelfPlayActor(env, mc, mcts, tc, sc, net, replay, dev, seed=11)
        assert actor.compact == compact and (actor.tree.live is not None) == compact
        reports = [actor.play_move() for _ in range(10)]
        n = len(replay)
        runs.append((reports, replay.policy[:n].clone(), replay.ret[:n].clone(), actor.tree.current_root_states().clone()))
        thin = sum(r.slot_census["terminal"] + r.slot_census["none"] + r.slot_census.get("depth_cap", 0) for r in reports)
        assert thin > 0, "the scenario has to contain waves with fewer live rows than games"
    (ra, pa, va, sa), (rb, pb, vb, sb) = runs
    for x, y in zip(ra, rb):  # per-game values: the move-level means are float atomics, i.e. order dependent in the last bit
        assert np.array_equal(x.per_game["step_reward"], y.per_game["step_reward"])
        assert np.array_equal(x.per_game["current_score"], y.per_game["current_score"])
    assert [r.slot_census for r in ra] == [r.slot_census for r in rb]
    assert torch.equal(sa, sb)

    def canonical(policy, ret):  # replay rows are emitted through an atomic cursor: compare them as multisets
        key = (policy.double() @ torch.linspace(1.0, 2.0, policy.shape[1], dtype=torch.float64, device=policy.device)) + ret.double() * 7.0
        order = torch.argsort(key, stable=True)
        return policy[order], ret[order]

    if fc_splits:
        assert torch.equal(canonical(pa, va * 0)[0], canonical(pb, vb * 0)[0])
        assert torch.allclose(va.sort().values, vb.sort().values, atol=1e-4 if dtype == "fp32" else 0.05, rtol=0)
        return
    (pa, va), (pb, vb) = canonical(pa, va), canonical(pb, vb)
    assert torch.equal(pa, pb) and torch.equal(va, vb)


@pytest.mark.parametrize("evaluator", ["engine-bf16", "engine-fp32", "module"])
def test_non_finite_network_output_raises_and_every_kernel_finishes(evaluator):
    """A diverged learner must not poison the replay silently (alphatriangle/nn/network.py:177-191,260-274)."""
    from alphatriangle_amd.models.network import NetworkEvaluationError

    actor, replay, _ = _actor(games=16, sims=6, dtype="bf16" if evaluator == "engine-bf16" else "fp32",
                              hip_engine=False if evaluator == "module" else None)
    assert type(actor.evaluator).__name__ == ("ModuleEvaluator" if evaluator == "module" else "EngineEvaluator")
    assert actor.dtype == ("bf16" if evaluator == "engine-bf16" else "fp32")
    actor.play_move()  # healthy move first
    with torch.no_grad():
        actor.nn._orig_model.policy_head[-1].bias[3] = float("nan")  # poisoned from the host
    actor.sync_weights(1)
    with pytest.raises(NetworkEvaluationError):
        actor.play_move()
    torch.cuda.synchronize()  # every kernel of the poisoned move ran to its normal end
    assert (actor.tree.n_nodes >= 1).all() and (actor.tree.n_nodes <= actor.tree.C).all()
    # the search itself stayed well defined: uniform priors instead of NaN priors, so visits were still counted
    with torch.no_grad():
        actor.nn._orig_model.policy_head[-1].bias[3] = 0.0
    actor.sync_weights(2)
    rep = actor.play_move()
    assert rep.moves == actor.G and torch.isfinite(replay.ret[: len(replay)]).all()


def test_gpu_tree_with_pytorch_leaf_evaluator_for_untiled_models():
    """Model configurations the HIP kernels do not tile (here: two input planes, a 5x5 first convolution, Tanh) still
    self-play on the GPU tree; the leaves go through the PyTorch module in fp32."""
    mc = ModelConfig(GRID_INPUT_CHANNELS=2, CONV_FILTERS=[16, 24], CONV_KERNEL_SIZES=[5, 3], CONV_STRIDES=[1, 1], CONV_PADDING=[2, 1],
                     NUM_RESIDUAL_BLOCKS=1, RESIDUAL_BLOCK_FILTERS=24, USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0,
                     FC_DIMS_SHARED=[48], POLICY_HEAD_DIMS=[48], VALUE_HEAD_DIMS=[48], ACTIVATION_FUNCTION="Tanh")
    actor, replay, _ = _actor(games=24, sims=8, dtype="fp32", model=mc)
    assert actor.nn.engine is None and type(actor.evaluator).__name__ == "ModuleEvaluator" and actor.dtype == "fp32"
    added = 0
    for _ in range(5):
        rep = actor.play_move()
        assert rep.moves == actor.G and rep.simulations >= actor.G * 8
        added += rep.experiences_added
    assert added > 0 and len(replay) == added
    pol = replay.policy[:added]
    assert torch.allclose(pol.sum(1), torch.ones(added, device=pol.device), atol=1e-4)
    # the evaluator's private copy follows the learner's parameters
    with torch.no_grad():
        for p in actor.nn._orig_model.parameters():
            p.add_(0.01)
    actor.sync_weights(3)
    src, dst = actor.nn._orig_model.state_dict(), actor.evaluator.module.state_dict()
    assert all(torch.equal(src[k], dst[k]) for k in src)


def test_module_evaluator_skips_dead_rows_and_runs_in_bf16():
    """With batch compaction the PyTorch evaluator sends only the dense rows (rounded up to a quarter of the batch) through the module; the
    search it feeds picks (nearly always) the same moves as the one that evaluates every row.  With the bf16 engine
    requested and no HIP engine for the model it runs under bf16 autocast and says so."""
    mc = ModelConfig(GRID_INPUT_CHANNELS=2, CONV_FILTERS=[16, 24], CONV_KERNEL_SIZES=[5, 3], CONV_STRIDES=[1, 1], CONV_PADDING=[2, 1],
                     NUM_RESIDUAL_BLOCKS=1, RESIDUAL_BLOCK_FILTERS=24, USE_TRANSFORMER=False, TRANSFORMER_LAYERS=0,
                     FC_DIMS_SHARED=[48], POLICY_HEAD_DIMS=[48], VALUE_HEAD_DIMS=[48], ACTIVATION_FUNCTION="Tanh")
    dense, _, _ = _actor(games=200, sims=8, dtype="fp32", model=mc)
    assert dense.compact and dense.evaluator.supports_live
    full, _, _ = _actor(games=200, sims=8, dtype="fp32", model=mc, compact=False)
    assert not full.compact
    same = 0
    for _ in range(4):
        dense.play_move()
        full.play_move()
        torch.cuda.synchronize()
        same += int((dense.tree.chosen_action == full.tree.chosen_action).sum())
    assert same >= 0.9 * 4 * 200, same  # library kernels may differ by batch size in the last bits; the games stay together
    # root evaluation (all rows) + 8 waves per move; the dense actor never evaluates more rows than the full one
    assert full.evaluator.rows_evaluated == 4 * 9 * 200
    assert dense.evaluator.row_step == 64 and dense.evaluator.rows_evaluated <= full.evaluator.rows_evaluated + 4 * 9 * 64
    half, _, _ = _actor(games=24, sims=8, dtype="bf16", model=mc)
    assert half.nn.engine is None and half.dtype == "bf16" and half.evaluator.dtype == "bf16"
    rep = half.play_move()
    assert rep.moves == half.G and torch.isfinite(half.evaluator.logits).all()


def test_graph_replay_equals_eager():
    """The captured wave graph is replayed many times per move; its searches must equal eager ones."""
    a1, r1, _ = _actor(graphs=True)
    a2, r2, _ = _actor(graphs=False)
    for _ in range(4):
        a1.play_move()
        a2.play_move()
        torch.cuda.synchronize()
        assert torch.equal(a1.tree.chosen_action, a2.tree.chosen_action)
        assert torch.equal(a1.tree.n_nodes, a2.tree.n_nodes)
    assert len(r1) == len(r2)
    n = len(r1)
    if n:
        # replay slots are handed out by an atomic counter, so the order across games is free:
        # compare the experiences as multisets
        key = torch.arange(1, r1.action_dim + 1, device=r1.device, dtype=torch.float64)

        def signature(r):
            h = (r.policy[:n].double() * key).sum(1) + 1000.0 * r.ret[:n].double() + r.grid[:n].double().sum(1) * 7.0
            return torch.sort(h).values

        assert torch.allclose(signature(r1), signature(r2), atol=0, rtol=0)


def test_large_preset_play_move():
    """12x21 board, 8-layer encoder: generic tile kernel (MT=16), unfused attention (16 key tiles), 756 actions."""
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.presets import build_preset
    from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
    from alphatriangle_amd.rl.self_play.gpu_actor import GpuSelfPlayActor

    torch.manual_seed(0)
    dev = torch.device("cuda")
    env, mc, _, _, _ = build_preset("large")
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=5000, MIN_BUFFER_SIZE_TO_TRAIN=64, BATCH_SIZE=32, N_STEP_RETURNS=2, MAX_TRAINING_STEPS=1000)
    sc = SystemConfig(GAMES_PER_GPU=24, REPLAY_HBM_BUDGET_GB=4.0)
    mcts = AlphaTriangleMCTSConfig(max_simulations=10, max_depth=5)
    net = NeuralNetwork(mc, env, tc, dev)
    assert net.engine is not None and not net.engine.fused_tf
    replay = DeviceReplayBuffer(tc, env, mc, dev, sc)
    actor = GpuSelfPlayActor(env, mc, mcts, tc, sc, net, replay, dev, seed=3)
    added = 0
    for _ in range(4):
        rep = actor.play_move()
        assert rep.moves == actor.G and rep.simulations >= actor.G * 10
        added += rep.experiences_added
    assert added >= actor.G and len(replay) == added
    pol = replay.policy[:added]
    assert pol.shape[1] == env.ACTION_DIM
    assert torch.allclose(pol.sum(1), torch.ones(added, device=dev), atol=1e-4)


@pytest.mark.parametrize("overlap", [True, False])
def test_training_loop_on_gpu_colocated(tmp_path, overlap):
    """The real runner on one card: GPU actor + HBM replay + learner steps (on a second stream when overlapped)."""
    from alphatriangle_amd.stats import TrieyeConfig
    from alphatriangle_amd.training.runner import run_training

    tc = TrainConfig(DEVICE="cuda", BATCH_SIZE=64, BUFFER_CAPACITY=20_000, MIN_BUFFER_SIZE_TO_TRAIN=128, MAX_TRAINING_STEPS=12,
                     WORKER_UPDATE_FREQ_STEPS=4, CHECKPOINT_SAVE_FREQ_STEPS=8, N_STEP_RETURNS=2, AUTO_RESUME_LATEST=False)
    sc = SystemConfig(ENGINE="gpu", GAMES_PER_GPU=96, TRAIN_STEPS_PER_MOVE=3, OVERLAP_LEARNER=overlap, REPLAY_HBM_BUDGET_GB=4.0)
    mcts = AlphaTriangleMCTSConfig(max_simulations=8, max_depth=4)
    run = f"colo_{int(overlap)}"
    tcfg = TrieyeConfig(app_name="t", run_name=run)
    tcfg.persistence.ROOT_DATA_DIR = str(tmp_path)
    tcfg.persistence.RUN_NAME = run
    code = run_training("WARNING", tc, tcfg, False, model_config_override=ModelConfig(), mcts_config_override=mcts,
                        env_config_override=EnvConfig(), system_config_override=sc)
    assert code == 0
    ckpt_dir = tmp_path / "t" / "runs" / run / "checkpoints"
    assert any(p.suffix == ".pkl" for p in ckpt_dir.rglob("*")), "periodic / final checkpoint missing"


def test_profiled_actor_leaves_a_per_move_gpu_timing_file(tmp_path):
    """--profile on the GPU path: three moves leave three records in profile_data/gpu_actor_rank0.jsonl whose phases add
    up to the move, and the move report carries the same numbers for the Perf/* metrics."""
    import json

    from alphatriangle_amd.config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig
    from alphatriangle_amd.models.network import NeuralNetwork
    from alphatriangle_amd.rl.core.device_buffer import DeviceReplayBuffer
    from alphatriangle_amd.rl.self_play.gpu_actor import GpuSelfPlayActor
    from alphatriangle_amd.rl.self_play.gpu_profile import summarise

    dev = torch.device("cuda")
    env, mc = EnvConfig(), ModelConfig()
    tc = TrainConfig(DEVICE="cuda", BUFFER_CAPACITY=20_000, MIN_BUFFER_SIZE_TO_TRAIN=256, PROFILE_WORKERS=True)
    sc = SystemConfig(GAMES_PER_GPU=256, REPLAY_HBM_BUDGET_GB=2.0)
    net = NeuralNetwork(mc, env, tc, dev, inference_dtype=sc.INFERENCE_DTYPE)
    replay = DeviceReplayBuffer(tc, env, mc, dev, sc)
    actor = GpuSelfPlayActor(env, mc, AlphaTriangleMCTSConfig(max_simulations=16, max_depth=5), tc, sc, net, replay, dev, seed=3,
                             profile_dir=str(tmp_path / "profile_data"), rank=0)
    reports = [actor.play_move() for _ in range(3)]
    actor.profiler.close()
    path = tmp_path / "profile_data" / "gpu_actor_rank0.jsonl"
    rows = [json.loads(ln) for ln in path.read_text().splitlines() if ln.strip()]
    assert len(rows) == 3 and [r["move"] for r in rows] == [0, 1, 2]
    for r, rep in zip(rows, reports):
        assert r["games_moved"] == rep.moves == 256 and r["simulations"] == rep.simulations
        assert r["root_eval_ms"] > 0 and r["waves_ms"] > 0 and r["commit_ms"] > 0
        assert abs(r["root_eval_ms"] + r["waves_ms"] + r["commit_ms"] - r["move_ms"]) < 0.05 * r["move_ms"] + 0.05
        assert "learner_ms" not in r  # no overlapped learner in this run
        assert rep.phase_ms["move_ms"] == r["move_ms"]
    assert summarise(path)["phases"]["waves"]["n"] == 3
