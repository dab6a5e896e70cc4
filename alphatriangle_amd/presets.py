"""Named configuration presets, one per benchmark configuration of this project.

* ``tiny``    - 5x5 grid, 2-layer CNN, 8 simulations, one CPU worker (plumbing check, no GPU).
* ``cnn``     - default grid, CNN policy + C51 value network, 100 simulations.
* ``default`` - default grid, CNN + Transformer encoder, 400 simulations on GPU engines
  (the reference's own default of 64 simulations is kept for the cpu engine by ``reference``).
* ``large``   - 12x21 grid, 8-layer Transformer, 800 simulations, HBM-sized replay.
* ``huge``    - 20x30 grid (600 cells: the wide build of the host and the device engine, 160-byte game states): CNN
  network, 64 simulations, 1024 games per GPU with leaves evaluated by the PyTorch module (no HIP inference engine tiles
  that board); host workers on a machine without a GPU.
* ``reference`` - every reference default untouched (64 simulations, 8 CPU workers).
"""

from __future__ import annotations

from .config import AlphaTriangleMCTSConfig, EnvConfig, ModelConfig, SystemConfig, TrainConfig


def build_preset(name: str) -> tuple[EnvConfig, ModelConfig, AlphaTriangleMCTSConfig, TrainConfig, SystemConfig]:
    name = (name or "default").lower()
    if name == "reference":
        return EnvConfig(), ModelConfig(), AlphaTriangleMCTSConfig(), TrainConfig(), SystemConfig()
    if name == "tiny":
        env = EnvConfig(ROWS=5, COLS=5, PLAYABLE_RANGE_PER_ROW=[(0, 5)] * 5, NUM_SHAPE_SLOTS=2)
        model = ModelConfig(
            CONV_FILTERS=[8, 16],
            CONV_KERNEL_SIZES=[3, 3],
            CONV_STRIDES=[1, 1],
            CONV_PADDING=[1, 1],
            NUM_RESIDUAL_BLOCKS=0,
            USE_TRANSFORMER=False,
            TRANSFORMER_LAYERS=0,
            FC_DIMS_SHARED=[32],
            POLICY_HEAD_DIMS=[32],
            VALUE_HEAD_DIMS=[32],
            OTHER_NN_INPUT_FEATURES_DIM=8 * 2 + 6,
        )
        mcts = AlphaTriangleMCTSConfig(max_simulations=8, max_depth=4, mcts_batch_size=4)
        train = TrainConfig(
            BATCH_SIZE=32,
            BUFFER_CAPACITY=5_000,
            MIN_BUFFER_SIZE_TO_TRAIN=64,
            MAX_TRAINING_STEPS=200,
            NUM_SELF_PLAY_WORKERS=1,
            WORKER_UPDATE_FREQ_STEPS=10,
            CHECKPOINT_SAVE_FREQ_STEPS=100,
            N_STEP_RETURNS=3,
            DEVICE="cpu",
        )
        return env, model, mcts, train, SystemConfig(ENGINE="cpu")
    if name == "cnn":
        return (
            EnvConfig(),
            ModelConfig.cnn_only(),
            AlphaTriangleMCTSConfig(max_simulations=100),
            TrainConfig(),
            SystemConfig(),
        )
    if name == "default":
        return EnvConfig(), ModelConfig(), AlphaTriangleMCTSConfig(max_simulations=400), TrainConfig(), SystemConfig()
    if name == "large":
        env = EnvConfig.large()
        return (
            env,
            ModelConfig.large_transformer(other_dim=8 * env.NUM_SHAPE_SLOTS + 6),
            AlphaTriangleMCTSConfig(max_simulations=800),
            TrainConfig(BUFFER_CAPACITY=4_000_000, MIN_BUFFER_SIZE_TO_TRAIN=100_000),
            SystemConfig(GAMES_PER_GPU=2048, REPLAY_HBM_BUDGET_GB=128.0),
        )
    if name == "huge":
        rows, cols = 20, 30
        ranges = [(max(0, 4 - r, r - (rows - 5)), cols - max(0, 4 - r, r - (rows - 5))) for r in range(rows)]
        env = EnvConfig(ROWS=rows, COLS=cols, PLAYABLE_RANGE_PER_ROW=ranges, NUM_SHAPE_SLOTS=3)
        train = TrainConfig(BATCH_SIZE=128, BUFFER_CAPACITY=200_000, MIN_BUFFER_SIZE_TO_TRAIN=2_000, NUM_SELF_PLAY_WORKERS=8)
        return env, ModelConfig.cnn_only(), AlphaTriangleMCTSConfig(max_simulations=64), train, SystemConfig(GAMES_PER_GPU=1024)
    raise ValueError(f"unknown preset '{name}' (choose tiny, cnn, default, large, huge, reference)")
