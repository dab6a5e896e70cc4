import pytest
from pydantic import ValidationError

from alphatriangle_amd.config import (
    APP_NAME,
    AlphaTriangleMCTSConfig,
    EnvConfig,
    ModelConfig,
    SystemConfig,
    TrainConfig,
    print_config_info_and_validate,
)


def test_model_config_defaults_match_reference():
    mc = ModelConfig()
    assert mc.CONV_FILTERS == [32, 64, 128] and mc.NUM_RESIDUAL_BLOCKS == 2 and mc.RESIDUAL_BLOCK_FILTERS == 128
    assert mc.USE_TRANSFORMER and mc.TRANSFORMER_DIM == 128 and mc.TRANSFORMER_HEADS == 4 and mc.TRANSFORMER_LAYERS == 2
    assert mc.TRANSFORMER_FC_DIM == 256 and mc.FC_DIMS_SHARED == [128] and mc.POLICY_HEAD_DIMS == [128]
    assert mc.NUM_VALUE_ATOMS == 51 and mc.VALUE_MIN == -10.0 and mc.VALUE_MAX == 10.0
    assert mc.ACTIVATION_FUNCTION == "ReLU" and mc.USE_BATCH_NORM and mc.OTHER_NN_INPUT_FEATURES_DIM == 30


def test_model_config_validators():
    with pytest.raises(ValidationError):
        ModelConfig(CONV_FILTERS=[32, 64], CONV_KERNEL_SIZES=[3], CONV_STRIDES=[1, 1], CONV_PADDING=[1, 1])
    with pytest.raises(ValidationError):
        ModelConfig(TRANSFORMER_DIM=130, TRANSFORMER_HEADS=4)
    with pytest.raises(ValidationError):
        ModelConfig(VALUE_MIN=1.0, VALUE_MAX=1.0)
    ModelConfig(USE_TRANSFORMER=False, TRANSFORMER_DIM=130)  # not checked when unused


def test_train_config_defaults_and_derived_fields():
    tc = TrainConfig()
    assert tc.BATCH_SIZE == 256 and tc.BUFFER_CAPACITY == 250_000 and tc.MIN_BUFFER_SIZE_TO_TRAIN == 25_000
    assert tc.N_STEP_RETURNS == 5 and tc.GAMMA == 0.99 and tc.OPTIMIZER_TYPE == "AdamW" and tc.LEARNING_RATE == 2e-4
    assert tc.LR_SCHEDULER_T_MAX == 100_000 and tc.PER_BETA_ANNEAL_STEPS == 100_000
    assert tc.WORKER_UPDATE_FREQ_STEPS == 10 and tc.CHECKPOINT_SAVE_FREQ_STEPS == 2500 and tc.USE_PER
    assert tc.RUN_NAME.startswith("train_")
    t2 = TrainConfig(MAX_TRAINING_STEPS=500)
    assert t2.LR_SCHEDULER_T_MAX == 500 and t2.PER_BETA_ANNEAL_STEPS == 500
    t3 = TrainConfig(MAX_TRAINING_STEPS=None)
    assert t3.LR_SCHEDULER_T_MAX == 100_000 and t3.PER_BETA_ANNEAL_STEPS == 100_000


def test_train_config_validators():
    with pytest.raises(ValidationError):
        TrainConfig(MIN_BUFFER_SIZE_TO_TRAIN=10, BUFFER_CAPACITY=5, BATCH_SIZE=2)
    with pytest.raises(ValidationError):
        TrainConfig(BATCH_SIZE=500, BUFFER_CAPACITY=100, MIN_BUFFER_SIZE_TO_TRAIN=10)
    with pytest.raises(ValidationError):
        TrainConfig(GRADIENT_CLIP_VALUE=0.0)
    with pytest.raises(ValidationError):
        TrainConfig(PER_BETA_INITIAL=0.8, PER_BETA_FINAL=0.5)
    with pytest.raises(ValidationError):
        TrainConfig(LR_SCHEDULER_T_MAX=0)


def test_mcts_config():
    m = AlphaTriangleMCTSConfig()
    assert (m.max_simulations, m.max_depth, m.cpuct, m.mcts_batch_size) == (64, 8, 1.5, 32)
    assert m.dirichlet_alpha == 0.3 and m.dirichlet_epsilon == 0.25 and m.discount == 1.0
    with pytest.raises(ValidationError):
        m.max_simulations = 0  # validate_assignment
    sp = m.to_trimcts_config()
    assert sp.max_simulations == 64 and sp.leaf_batch == 32 and abs(sp.cpuct - 1.5) < 1e-6
    assert m.capacity() == 130


def test_env_config():
    e = EnvConfig()
    assert (e.ROWS, e.COLS, e.NUM_SHAPE_SLOTS, e.ACTION_DIM) == (8, 15, 3, 360)
    with pytest.raises(ValidationError):
        EnvConfig(ROWS=2, COLS=3, PLAYABLE_RANGE_PER_ROW=[(0, 3)])
    assert EnvConfig(ROWS=20, COLS=20, PLAYABLE_RANGE_PER_ROW=[(0, 20)] * 20).ACTION_DIM == 1200  # wide host engine
    with pytest.raises(ValidationError):
        EnvConfig(ROWS=40, COLS=30, PLAYABLE_RANGE_PER_ROW=[(0, 30)] * 40)  # above 1024 cells
    big = EnvConfig.large()
    assert big.ROWS * big.COLS <= 256  # the largest preset the device engine plays


def test_print_config_and_system(capsys):
    out = print_config_info_and_validate()
    assert set(out) == {"env", "model", "train", "mcts", "system"}
    text = capsys.readouterr().out
    assert "MAX_TRAINING_STEPS" in text and "max_simulations" in text and "GAMES_PER_GPU" in text
    with pytest.raises(ValueError):
        print_config_info_and_validate("not a config")  # type: ignore[arg-type]
    assert APP_NAME == "AlphaTriangle"
    with pytest.raises(ValidationError):
        SystemConfig(REPLAY_HBM_BUDGET_GB=400)


def test_learner_gemm_mode_is_applied_to_the_model_module():
    import alphatriangle_amd.models.model as model_mod

    try:
        SystemConfig(LEARNER_FP32_GEMM="library", LEARNER_PATCH_FREE_CONV=True).apply_learner_gemm_mode()
        assert model_mod.FP32_GEMM_ON_BF16_CORES is False and model_mod.PATCH_FREE_CONV is True
        SystemConfig().apply_learner_gemm_mode()
        assert model_mod.FP32_GEMM_ON_BF16_CORES is True and model_mod.PATCH_FREE_CONV is False
        with pytest.raises(ValidationError):
            SystemConfig(LEARNER_FP32_GEMM="fp16")
    finally:
        model_mod.FP32_GEMM_ON_BF16_CORES, model_mod.PATCH_FREE_CONV = True, False
    sc = SystemConfig()
    assert sc.COMPACT_LEAF_BATCH is True and sc.INFERENCE_DTYPE == "fp32" and sc.learner_autocast() is False
    assert SystemConfig(INFERENCE_DTYPE="bf16").learner_autocast() is True
