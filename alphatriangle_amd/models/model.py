"""PyTorch definition of the policy / distributional-value network.

This is the learner-side (autograd) definition and the numerical reference for
the hand-written HIP inference engine (ops/inference.py).  The module tree and
therefore every ``state_dict`` key equals the reference network
(alphatriangle/nn/model.py:15-297: ``conv_body.{i}.{0,1}``, ``res_body.{i}.conv1.{0,1}``,
``res_body.{i}.conv2``, ``res_body.{i}.bn2``, ``input_proj``, ``pos_encoder.pe``,
``transformer_body.layers.{i}.*``, ``transformer_body.norm``, ``shared_fc``,
``policy_head``, ``value_head``) so checkpoints are interchangeable.
"""

from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..config.env_config import EnvConfig
from ..config.model_config import ModelConfig


class GemmConv2d(nn.Conv2d):
    """``nn.Conv2d`` (same parameters, same ``state_dict`` keys) whose GPU path is im2col + GEMM.

    On gfx950 MIOpen has no tuned bf16 / fp32 convolution for these tiny 8x15 images and falls
    back to naive kernels (11 ms for one weight-gradient call at batch 256).  Lowering the
    convolution to a patch matrix + one batched matmul puts forward, data-gradient and
    weight-gradient on hipBLASLt GEMMs, which is what the learner needs.  CPU tensors keep the
    stock implementation.
    """

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (
            not x.is_cuda
            or self.stride != (1, 1)
            or self.dilation != (1, 1)
            or self.groups != 1
            or isinstance(self.padding, str)
            or self.padding_mode != "zeros"
        ):
            return super().forward(x)
        b, _, h, w = x.shape
        kh, kw = self.kernel_size
        ph, pw = self.padding
        oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
        if kh == 1 and kw == 1 and ph == 0 and pw == 0:
            cols = x.reshape(b, x.shape[1], h * w)
        else:
            # F.unfold launches one im2col kernel PER SAMPLE (1536 launches per step at batch 256); strided views of the
            # padded input plus one gathering copy build the same [B, Cin*kh*kw, oh*ow] matrix in two launches
            xp = torch.nn.functional.pad(x, (pw, pw, ph, ph))
            patches = xp.unfold(2, kh, 1).unfold(3, kw, 1)  # [B, Cin, oh, ow, kh, kw] (view)
            cols = patches.permute(0, 1, 4, 5, 2, 3).reshape(b, x.shape[1] * kh * kw, oh * ow)
        out = torch.matmul(self.weight.reshape(self.out_channels, -1), cols)  # [B, Cout, oh*ow]
        if self.bias is not None:
            out = out + self.bias.view(1, -1, 1)
        return out.view(b, self.out_channels, oh, ow)


def conv_block(
    in_channels: int,
    out_channels: int,
    kernel_size: int | tuple[int, int],
    stride: int | tuple[int, int],
    padding: int | tuple[int, int] | str,
    use_batch_norm: bool,
    activation: type[nn.Module],
) -> nn.Sequential:
    """Conv2d (bias only without BN) -> optional BatchNorm2d -> activation."""
    mods: list[nn.Module] = [
        GemmConv2d(in_channels, out_channels, kernel_size, stride, padding, bias=not use_batch_norm)
    ]
    if use_batch_norm:
        mods.append(nn.BatchNorm2d(out_channels))
    mods.append(activation())
    return nn.Sequential(*mods)


class ResidualBlock(nn.Module):
    """conv-bn-act, conv-bn, skip add, act."""

    def __init__(self, channels: int, use_batch_norm: bool, activation: type[nn.Module]):
        super().__init__()
        self.conv1 = conv_block(channels, channels, 3, 1, 1, use_batch_norm, activation)
        self.conv2 = GemmConv2d(channels, channels, 3, 1, 1, bias=not use_batch_norm)
        self.bn2 = nn.BatchNorm2d(channels) if use_batch_norm else nn.Identity()
        self.activation = activation()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.bn2(self.conv2(self.conv1(x)))
        return self.activation(y + x)


class PositionalEncoding(nn.Module):
    """Sinusoidal positional encoding for ``[seq, batch, dim]`` inputs (persistent buffer ``pe``)."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000):
        super().__init__()
        if d_model <= 0:
            raise ValueError("d_model must be positive for PositionalEncoding")
        self.dropout = nn.Dropout(p=dropout)
        self.register_buffer("pe", sinusoid_table(max_len, d_model).unsqueeze(1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        pe: torch.Tensor = self.pe  # type: ignore[assignment]
        if x.shape[0] > pe.shape[0]:
            raise ValueError(f"Input sequence length {x.shape[0]} exceeds max_len {pe.shape[0]} of PositionalEncoding")
        if x.shape[2] != pe.shape[2]:
            raise ValueError(
                f"Input embedding dimension {x.shape[2]} does not match PositionalEncoding dimension {pe.shape[2]}"
            )
        return self.dropout(x + pe[: x.size(0)])


def sinusoid_table(length: int, d_model: int) -> torch.Tensor:
    """``[length, d_model]`` table: sin on even columns, cos on odd columns."""
    pos = torch.arange(length, dtype=torch.float32).unsqueeze(1)
    freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    table = torch.zeros(length, d_model)
    table[:, 0::2] = torch.sin(pos * freq)
    if d_model > 1:
        table[:, 1::2] = torch.cos(pos * freq)[:, : d_model // 2]
    return table


def _mlp(in_features: int, hidden: list[int], use_bn: bool, activation: type[nn.Module]) -> tuple[list[nn.Module], int]:
    mods: list[nn.Module] = []
    for width in hidden:
        mods.append(nn.Linear(in_features, width))
        if use_bn:
            mods.append(nn.BatchNorm1d(width))
        mods.append(activation())
        in_features = width
    return mods, in_features


class AlphaTriangleNet(nn.Module):
    """CNN body -> residual body -> optional Transformer encoder -> shared FC -> policy / C51 value heads."""

    def __init__(self, model_config: ModelConfig, env_config: EnvConfig):
        super().__init__()
        self.model_config = model_config
        self.env_config = env_config
        self.action_dim = int(env_config.NUM_SHAPE_SLOTS * env_config.ROWS * env_config.COLS)
        mc = model_config
        act: type[nn.Module] = getattr(nn, mc.ACTIVATION_FUNCTION)

        channels = mc.GRID_INPUT_CHANNELS
        stem: list[nn.Module] = []
        for i, width in enumerate(mc.CONV_FILTERS):
            stem.append(
                conv_block(channels, width, mc.CONV_KERNEL_SIZES[i], mc.CONV_STRIDES[i], mc.CONV_PADDING[i], mc.USE_BATCH_NORM, act)
            )
            channels = width
        self.conv_body = nn.Sequential(*stem)

        res: list[nn.Module] = []
        if mc.NUM_RESIDUAL_BLOCKS > 0:
            if channels != mc.RESIDUAL_BLOCK_FILTERS:
                res.append(conv_block(channels, mc.RESIDUAL_BLOCK_FILTERS, 1, 1, 0, mc.USE_BATCH_NORM, act))
                channels = mc.RESIDUAL_BLOCK_FILTERS
            res.extend(ResidualBlock(channels, mc.USE_BATCH_NORM, act) for _ in range(mc.NUM_RESIDUAL_BLOCKS))
        self.res_body = nn.Sequential(*res)
        self.cnn_output_channels = channels

        self.transformer_body: nn.TransformerEncoder | None = None
        self.pos_encoder: PositionalEncoding | None = None
        self.input_proj: nn.Module = nn.Identity()
        self.transformer_output_size = 0
        self.flattened_cnn_size = 0
        self.uses_transformer = bool(mc.USE_TRANSFORMER and mc.TRANSFORMER_LAYERS > 0)

        with torch.no_grad():
            probe = torch.zeros(1, mc.GRID_INPUT_CHANNELS, env_config.ROWS, env_config.COLS)
            body_out = self.res_body(self.conv_body(probe))
        self.body_hw = (int(body_out.shape[2]), int(body_out.shape[3]))

        if self.uses_transformer:
            d = mc.TRANSFORMER_DIM
            if channels != d:
                self.input_proj = GemmConv2d(channels, d, kernel_size=1)
            self.pos_encoder = PositionalEncoding(d, dropout=0.1)
            layer = nn.TransformerEncoderLayer(
                d_model=d,
                nhead=mc.TRANSFORMER_HEADS,
                dim_feedforward=mc.TRANSFORMER_FC_DIM,
                activation=mc.ACTIVATION_FUNCTION.lower(),
                batch_first=False,
                norm_first=True,
            )
            self.transformer_body = nn.TransformerEncoder(
                layer, num_layers=mc.TRANSFORMER_LAYERS, norm=nn.LayerNorm(d), enable_nested_tensor=False
            )
            self.transformer_output_size = self.body_hw[0] * self.body_hw[1] * d
            feat = self.transformer_output_size
        else:
            self.flattened_cnn_size = int(body_out.numel())
            feat = self.flattened_cnn_size

        shared, width = _mlp(feat + mc.OTHER_NN_INPUT_FEATURES_DIM, list(mc.FC_DIMS_SHARED), mc.USE_BATCH_NORM, act)
        self.shared_fc = nn.Sequential(*shared)

        pol, pw = _mlp(width, list(mc.POLICY_HEAD_DIMS), mc.USE_BATCH_NORM, act)
        pol.append(nn.Linear(pw, self.action_dim))
        self.policy_head = nn.Sequential(*pol)

        val, vw = _mlp(width, list(mc.VALUE_HEAD_DIMS), mc.USE_BATCH_NORM, act)
        val.append(nn.Linear(vw, mc.NUM_VALUE_ATOMS))
        self.value_head = nn.Sequential(*val)

    def body(self, grid_state: torch.Tensor) -> torch.Tensor:
        """Feature extractor up to (and excluding) the shared FC: returns ``[B, feat]``."""
        x = self.res_body(self.conv_body(grid_state))
        if self.uses_transformer and self.transformer_body is not None and self.pos_encoder is not None:
            x = self.input_proj(x)
            tokens = x.flatten(2).permute(2, 0, 1)  # [H*W, B, D]
            tokens = self.transformer_body(self.pos_encoder(tokens))
            return tokens.permute(1, 0, 2).flatten(1)
        return x.reshape(x.size(0), -1)

    def forward(self, grid_state: torch.Tensor, other_features: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Returns ``(policy_logits [B, A], value_logits [B, atoms])``."""
        feat = torch.cat([self.body(grid_state), other_features], dim=1)
        shared = self.shared_fc(feat)
        return self.policy_head(shared), self.value_head(shared)
