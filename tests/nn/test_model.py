"""Network module (mirrors the reference's tests/nn/test_model.py: shapes, dtypes, transformer on/off)."""

import pytest
import torch

from alphatriangle_amd.config import EnvConfig, ModelConfig
from alphatriangle_amd.models import AlphaTriangleNet
from alphatriangle_amd.models.model import GemmConv2d, PositionalEncoding, sinusoid_table


def test_model_initialization(mock_model_config, mock_env_config, action_dim):
    m = AlphaTriangleNet(mock_model_config, mock_env_config)
    assert m.action_dim == action_dim == 9
    assert m.model_config is mock_model_config and m.env_config is mock_env_config


@pytest.mark.parametrize("batch_size", [1, 4])
def test_forward_shapes(mock_model_config, mock_env_config, action_dim, batch_size):
    m = AlphaTriangleNet(mock_model_config, mock_env_config).eval()
    grid = torch.randn(batch_size, 1, 3, 3)
    other = torch.randn(batch_size, mock_model_config.OTHER_NN_INPUT_FEATURES_DIM)
    with torch.no_grad():
        p, v = m(grid, other)
    assert p.shape == (batch_size, action_dim) and v.shape == (batch_size, mock_model_config.NUM_VALUE_ATOMS)
    assert p.dtype == torch.float32 and v.dtype == torch.float32


@pytest.mark.parametrize("use_transformer", [False, True])
def test_transformer_toggle(mock_env_config, use_transformer):
    mc = ModelConfig(
        CONV_FILTERS=[8], CONV_KERNEL_SIZES=[3], CONV_STRIDES=[1], CONV_PADDING=[1], NUM_RESIDUAL_BLOCKS=1, RESIDUAL_BLOCK_FILTERS=8,
        USE_TRANSFORMER=use_transformer, TRANSFORMER_DIM=16, TRANSFORMER_HEADS=2, TRANSFORMER_LAYERS=1 if use_transformer else 0,
        TRANSFORMER_FC_DIM=32, FC_DIMS_SHARED=[8], POLICY_HEAD_DIMS=[8], VALUE_HEAD_DIMS=[8], OTHER_NN_INPUT_FEATURES_DIM=14,
    )
    m = AlphaTriangleNet(mc, mock_env_config).eval()
    assert (m.transformer_body is not None) == use_transformer
    if use_transformer:
        assert isinstance(m.input_proj, torch.nn.Conv2d)  # 8 -> 16 projection
        assert m.transformer_output_size == 9 * 16
    with torch.no_grad():
        p, v = m(torch.randn(2, 1, 3, 3), torch.randn(2, 14))
    assert p.shape == (2, 9) and v.shape == (2, 51)


def test_default_net_size_and_state_dict_keys():
    m = AlphaTriangleNet(ModelConfig(), EnvConfig())
    assert sum(p.numel() for p in m.parameters()) == 3_005_819
    keys = set(m.state_dict().keys())
    for k in (
        "conv_body.0.0.weight", "conv_body.2.1.running_var", "res_body.0.conv1.0.weight", "res_body.1.conv2.weight",
        "res_body.1.bn2.bias", "pos_encoder.pe", "transformer_body.layers.0.self_attn.in_proj_weight",
        "transformer_body.layers.1.linear2.bias", "transformer_body.layers.0.norm1.weight", "transformer_body.norm.weight",
        "shared_fc.0.weight", "shared_fc.1.running_mean", "policy_head.3.weight", "value_head.3.bias",
    ):
        assert k in keys, k
    assert m.state_dict()["pos_encoder.pe"].shape == (5000, 1, 128)
    assert m.state_dict()["shared_fc.0.weight"].shape == (128, 15390)
    assert isinstance(m.conv_body[0][0], GemmConv2d)


def test_positional_encoding_validation():
    pe = PositionalEncoding(8, dropout=0.0, max_len=10)
    x = torch.zeros(4, 2, 8)
    assert torch.allclose(pe(x)[:, 0], sinusoid_table(10, 8)[:4])
    with pytest.raises(ValueError):
        pe(torch.zeros(11, 1, 8))
    with pytest.raises(ValueError):
        pe(torch.zeros(4, 1, 6))
    with pytest.raises(ValueError):
        PositionalEncoding(0)


def test_gemm_conv_lowering_equals_conv2d():
    conv = GemmConv2d(3, 5, 3, 1, 1)
    x = torch.randn(2, 3, 8, 15)
    cols = torch.nn.functional.unfold(x, (3, 3), padding=(1, 1))
    lowered = (conv.weight.reshape(5, -1) @ cols + conv.bias.view(1, -1, 1)).view(2, 5, 8, 15)
    assert torch.allclose(lowered, conv(x), atol=1e-5)


@pytest.mark.parametrize("cin,cout,k,p,bias", [(1, 8, 3, 1, False), (6, 10, 3, 1, True), (5, 7, 1, 0, True), (4, 6, 3, 0, False), (3, 5, 3, 2, True)])
def test_gemm_conv_matches_conv2d_forward_and_gradients(cin, cout, k, p, bias):
    """The learner's patch-matrix convolution (split-K weight gradient included) against ``F.conv2d`` in float64."""
    import torch.nn.functional as F

    from alphatriangle_amd.models.model import GemmConv2d, _split_count

    assert _split_count(30720) == 32 and _split_count(7) == 1 and 1920 % _split_count(1920) == 0
    torch.manual_seed(0)
    conv = GemmConv2d(cin, cout, k, 1, p, bias=bias).double()
    x = torch.randn(16, cin, 8, 15, dtype=torch.double, requires_grad=True)
    params = [x, conv.weight] + ([conv.bias] if bias else [])
    y_gemm = conv.gemm_forward(x)
    y_ref = F.conv2d(x, conv.weight, conv.bias, 1, p)
    assert y_gemm.shape == y_ref.shape and y_gemm.is_contiguous()
    assert torch.allclose(y_gemm, y_ref, atol=1e-12)
    g = torch.randn_like(y_ref)
    for a, b in zip(torch.autograd.grad(y_gemm, params, g), torch.autograd.grad(y_ref, params, g), strict=True):
        assert torch.allclose(a, b, atol=1e-10)


@pytest.mark.parametrize("act", ["relu", "gelu"])
def test_encoder_layer_fast_path_matches_stock_layer(act):
    """The written-out pre-LN layer (split-K linear layers, 2048 rows -> 4 chunks) against ``nn.TransformerEncoderLayer``."""
    from alphatriangle_amd.models.model import EncoderLayer, _split_count

    assert _split_count(64 * 32) == 4
    torch.manual_seed(0)
    layer = EncoderLayer(d_model=32, nhead=4, dim_feedforward=64, activation=act, batch_first=False, norm_first=True).double()
    layer.eval()  # dropout off: the two paths draw their masks differently
    x = torch.randn(64, 32, 32, dtype=torch.double, requires_grad=True)
    y_fast = layer.fast_forward(x)
    y_ref = torch.nn.TransformerEncoderLayer.forward(layer, x)
    assert torch.allclose(y_fast, y_ref, atol=1e-10)
    g = torch.randn_like(y_ref)
    params = [x] + list(layer.parameters())
    for a, b in zip(torch.autograd.grad(y_fast, params, g), torch.autograd.grad(y_ref, params, g), strict=True):
        assert torch.allclose(a, b, atol=1e-9)
    # state_dict keys are those of the stock layer (checkpoints stay interchangeable)
    stock = torch.nn.TransformerEncoderLayer(d_model=32, nhead=4, dim_feedforward=64, activation=act, norm_first=True)
    assert list(stock.state_dict().keys()) == list(layer.state_dict().keys())
    # CPU tensors and masked calls use the stock implementation
    assert not layer._fast_ok(x, None, None, False)


@pytest.mark.parametrize("use_transformer", [False, True])
@pytest.mark.parametrize("residual_filters", [16, 24])
def test_channel_major_body_equals_nchw_body_on_cpu(use_transformer: bool, residual_filters: int):
    """The learner's channel-major convolution body (activations [C, B, H, W], BatchNorm on a one-image view) must give
    the NCHW body's features, gradients and BatchNorm running statistics (fp32 on the host: equal up to summation order).
    24 residual filters after a 16-filter stem also exercise the 1x1 projection."""
    import copy

    from alphatriangle_amd.config import EnvConfig, ModelConfig
    from alphatriangle_amd.models.model import AlphaTriangleNet

    torch.manual_seed(11)
    env = EnvConfig(ROWS=4, COLS=6, PLAYABLE_RANGE_PER_ROW=[(0, 6)] * 4, NUM_SHAPE_SLOTS=1)
    mc = ModelConfig(
        CONV_FILTERS=[8, 16], CONV_KERNEL_SIZES=[3, 3], CONV_STRIDES=[1, 1], CONV_PADDING=[1, 1], NUM_RESIDUAL_BLOCKS=1,
        RESIDUAL_BLOCK_FILTERS=residual_filters, USE_TRANSFORMER=use_transformer, TRANSFORMER_DIM=16, TRANSFORMER_HEADS=2,
        TRANSFORMER_LAYERS=1 if use_transformer else 0, TRANSFORMER_FC_DIM=32, FC_DIMS_SHARED=[16], POLICY_HEAD_DIMS=[16],
        VALUE_HEAD_DIMS=[16], OTHER_NN_INPUT_FEATURES_DIM=14,
    )
    net_a = AlphaTriangleNet(mc, env).double().train()
    for m in net_a.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0  # applied functionally inside the attention
    net_b = copy.deepcopy(net_a)
    grid = torch.randint(-1, 2, (5, 1, env.ROWS, env.COLS)).double()
    fa = net_a.body_channel_major(grid)
    fb = net_b.body(grid)
    assert fa.shape == fb.shape
    assert torch.allclose(fa, fb, atol=1e-9)
    fa.square().sum().backward()
    fb.square().sum().backward()
    for (name, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        if pb.grad is None:
            assert pa.grad is None, name
            continue
        assert torch.allclose(pa.grad, pb.grad, atol=1e-8), name
    for (name, ba), (_, bb) in zip(net_a.named_buffers(), net_b.named_buffers()):
        assert torch.allclose(ba.double(), bb.double(), atol=1e-10), name
