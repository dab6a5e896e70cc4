"""Hand-written HIP inference engine for :class:`AlphaTriangleNet` on gfx950.

``InferenceEngine.load_from_module`` folds every BatchNorm into the preceding
conv / linear layer, converts weights to bf16 and packs them in MFMA fragment
order; ``forward`` then evaluates the network with the kernels of
``csrc/hip/nn_kernels.hip`` (about 25 launches, no PyTorch ops).  The numerical
reference is the eval-mode PyTorch module (models/model.py); the network itself
is the reference's (alphatriangle/nn/model.py:109-297).

Layer -> kernel map (default config, B leaves, HW = 120 tokens):
  conv stem 1 -> 32 -> 64 -> 128         rowconv_kernel<32, FIRST, PAIR, 64>: one launch, intermediates in LDS halo tiles
                                         (boards above 8x15: flatconv_kernel<32, TPW, PAIR, FIRST, 64>, flat tiles)
  conv stem layer 0 (Cin = 1), others    evaluated inside the staging of layer 1's rowconv_kernel (boards up to 8x15),
                                         else conv_first_kernel (VALU, memory bound)
  residual blocks (128 channels)         rowconv_kernel<PAIR>: both convolutions in one launch, intermediate in LDS
  conv stem 1.., residual convs          rowconv_kernel (boards up to 8x15: halo tile in LDS, one sample per
                                         workgroup, coalesced write-back), flatconv_kernel (larger boards with
                                         H * (W + 1) <= 320: flat pixel run in LDS) else tile_gemm_kernel taps=9
  1x1 projections, QKV, out-proj, FFN    tile_gemm_kernel taps=1  (+LayerNorm prologue,
                                         residual / activation / positional epilogue)
  4-head attention                       attention_kernel         (MFMA, softmax in registers)
  encoder layer, d_model 128, 129..256 tok qkv_attn_kernel (LN1 + QKV + attention per sample and head pair, Q / K / V stay
                                         on the chip) + proj_ffn_rows_kernel (out-projection and FFN half in one
                                         launch); also as qkv_rows + attn_core + proj_rows + ffn_block kernels
  encoder layer, d_model 128 / <=128 tok encoder_layer_kernel (tf_kernels.hip: one launch per layer, the residual stream
                                         stays in registers between the attention and FFN halves); the two halves
                                         also exist as attn_block_kernel + ffn_block_kernel
  final LayerNorm + flatten + shared FC  fc_rows_splitk_kernel (128-channel bodies: LN applied while staging the
  (K = 15390)                            rows, no flattened copy) + fc_finish_kernel; else rows_to_fc_kernel +
                                         fc_splitk_kernel
  FC reduction, policy / value heads,    heads_kernel (head_kernels.hip: 128-wide hidden layers, 16 samples per
  C51 expectation                        workgroup, one launch) else fc_finish_kernel + tile_gemm_kernel taps=1
                                         (rows = samples) + c51_expect_kernel
"""

from __future__ import annotations

import logging
import math
import os
import threading
from dataclasses import dataclass
from typing import Any

import torch
import torch.nn as nn

from ..config.env_config import EnvConfig
from ..config.model_config import ModelConfig
from . import hipbind as hb

logger = logging.getLogger(__name__)

# the live row count is host state of the kernel library between atri_set_live_count and the launches that pick it up
_LIVE_LOCK = threading.Lock()

ACT_IDS = {"none": 0, "ReLU": 1, "GELU": 2, "SiLU": 3, "Tanh": 4, "Sigmoid": 5}


def _pow2(x: int) -> bool:
    return x > 0 and (x & (x - 1)) == 0


def _split_flat_ok(rows: int, cols: int) -> bool:
    """Whether the fp32-grade flat-tile conv kernels hold a board of this size (host-side geometry check of the library;
    without the library - CPU-only installs - the documented bound is applied: tile rows + halo <= 320, <= 256 cells)."""
    try:
        return bool(hb.lib().atri_split_flatconv_supported(int(rows), int(cols)))
    except Exception:  # noqa: BLE001 - no kernel library on this host
        tiles = (rows * (cols + 1) + 15) // 16
        tpw = max(3, (tiles + 3) // 4)
        return tpw <= 5 and rows * cols <= 256 and 16 * max(tiles, 4 * (tpw - 1)) + 2 * cols + 4 <= 320


def engine_supports(mc: ModelConfig, ec: EnvConfig, dtype: str = "bf16") -> tuple[bool, str]:
    """Whether the HIP engine implements this architecture at this precision (else callers use the PyTorch module)."""
    if dtype not in ("bf16", "fp32"):
        return False, f"no {dtype} kernels"
    if dtype == "fp32":
        # The fp32-grade kernels (csrc/hip/split_*.hip) are written for the reference's default architecture family:
        # 3x3 stem 1 -> 32 -> 64 -> 128, 128-channel residual blocks, d_model 128 / 4 heads, 128-wide dense tail.
        # boards up to 8x15: halo-tile convolutions + one-launch encoder layer; larger boards (up to 256 cells): flat-tile
        # convolutions (csrc/hip/split_flatconv_kernels.hip) + the row-tiled encoder layer (split_tf_rows_kernels.hip)
        if (ec.ROWS > 8 or ec.COLS > 15) and not _split_flat_ok(ec.ROWS, ec.COLS):
            return False, "fp32 kernels need a board of at most 8 x 15 cells or one the flat-tile kernels can hold"
        if list(mc.CONV_FILTERS) != [32, 64, 128]:
            return False, "fp32 kernels need the conv stem 32 / 64 / 128"
        if mc.NUM_RESIDUAL_BLOCKS > 0 and mc.RESIDUAL_BLOCK_FILTERS != 128:
            return False, "fp32 kernels need 128-channel residual blocks"
        if mc.USE_TRANSFORMER and mc.TRANSFORMER_LAYERS > 0:
            if mc.TRANSFORMER_DIM != 128 or mc.TRANSFORMER_HEADS != 4 or mc.TRANSFORMER_FC_DIM % 128 != 0 or mc.TRANSFORMER_FC_DIM > 512:
                return False, "fp32 kernels need d_model 128, 4 heads and a feed-forward width of 128 / 256 / 384 / 512"
            if mc.ACTIVATION_FUNCTION not in ("ReLU", "GELU"):
                return False, "fp32 encoder kernel implements ReLU and GELU"
        if list(mc.FC_DIMS_SHARED) != [128] or list(mc.POLICY_HEAD_DIMS) != [128] or list(mc.VALUE_HEAD_DIMS) != [128]:
            return False, "fp32 kernels need the 128-wide dense tail"
        if mc.NUM_VALUE_ATOMS > 128:
            return False, "fp32 tail kernel holds at most 128 value atoms"
    if mc.GRID_INPUT_CHANNELS != 1:
        return False, "GRID_INPUT_CHANNELS != 1"
    if not mc.CONV_FILTERS:
        return False, "no conv stem"
    for k, s, p in zip(mc.CONV_KERNEL_SIZES, mc.CONV_STRIDES, mc.CONV_PADDING, strict=True):
        if k not in (3, (3, 3)) or s not in (1, (1, 1)) or p not in (1, (1, 1)):
            return False, "conv stem must be 3x3 / stride 1 / padding 1"
    chans = list(mc.CONV_FILTERS)
    if mc.NUM_RESIDUAL_BLOCKS > 0:
        chans.append(mc.RESIDUAL_BLOCK_FILTERS)
    if chans[0] % 8 != 0:
        return False, "first conv width must be a multiple of 8"
    for c in chans:
        if not (_pow2(c) and 32 <= c <= 512):
            return False, f"channel width {c} must be a power of two in [32, 512]"
    hw = ec.ROWS * ec.COLS
    if hw > 256:
        return False, "more than 256 cells"
    uses_tf = mc.USE_TRANSFORMER and mc.TRANSFORMER_LAYERS > 0
    if uses_tf:
        d = mc.TRANSFORMER_DIM
        if not (_pow2(d) and 32 <= d <= 512) or d // mc.TRANSFORMER_HEADS != 32:
            return False, "transformer needs power-of-two dim with head_dim 32"
        if not (_pow2(mc.TRANSFORMER_FC_DIM) and 32 <= mc.TRANSFORMER_FC_DIM <= 512):
            return False, "TRANSFORMER_FC_DIM must be a power of two in [32, 512]"
        if mc.ACTIVATION_FUNCTION not in ("ReLU", "GELU"):
            return False, "transformer activation must be ReLU or GELU"
    if not mc.FC_DIMS_SHARED:
        return False, "FC_DIMS_SHARED must not be empty"
    if mc.FC_DIMS_SHARED[0] > 128 or mc.FC_DIMS_SHARED[0] % 32 != 0:
        return False, "first shared FC width must be a multiple of 32 and <= 128"
    for w in [*mc.FC_DIMS_SHARED, *mc.POLICY_HEAD_DIMS, *mc.VALUE_HEAD_DIMS]:
        if not (_pow2(w) and 32 <= w <= 512):
            return False, f"dense width {w} must be a power of two in [32, 512]"
    return True, ""


_pack_index_cache: dict[tuple[int, int, int, str], tuple[torch.Tensor, int]] = {}


def _pack_index(taps: int, cin: int, cout: int, device: torch.device) -> tuple[torch.Tensor, int]:
    """Flat gather indices into the zero-padded ``[taps, cin, NB*128]`` weight for the fragment layout."""
    key = (taps, cin, cout, str(device))
    hit = _pack_index_cache.get(key)
    if hit is not None:
        return hit
    assert cin % 32 == 0
    kc_n = cin // 32
    nb_n = (cout + 127) // 128
    lane = torch.arange(64)
    j = torch.arange(8)
    kidx = torch.arange(kc_n)[:, None, None] * 32 + 8 * (lane[None, :, None] >> 4) + j[None, None, :]  # [KC,64,8]
    r = lane & 15
    cidx = (
        torch.arange(nb_n)[:, None, None, None] * 128
        + torch.arange(4)[None, :, None, None] * 32
        + ((r >> 2) * 8)[None, None, None, :]
        + torch.arange(2)[None, None, :, None] * 4
        + (r & 3)[None, None, None, :]
    )  # [NB,4,2,64]
    shape = (nb_n, taps, kc_n, 4, 2, 64, 8)
    k_full = kidx[None, None, :, None, None, :, :].expand(shape)
    c_full = cidx[:, None, None, :, :, :, None].expand(shape)
    t_full = torch.arange(taps)[None, :, None, None, None, None, None].expand(shape)
    flat = ((t_full * cin + k_full) * (nb_n * 128) + c_full).reshape(-1).to(device)
    _pack_index_cache[key] = (flat, nb_n)
    return flat, nb_n


def split_operand_dtype() -> torch.dtype:
    """16-bit format of the (hi, lo) operand pairs the loaded kernel library was built for (csrc/hip/common.hpp)."""
    try:
        return torch.float16 if int(hb.lib().atri_split_operand_format()) == 1 else torch.bfloat16
    except Exception:  # noqa: BLE001 - no kernel library on this host
        return torch.bfloat16


def pack_gemm_weights(w: torch.Tensor, device: torch.device, dtype: torch.dtype = torch.bfloat16) -> tuple[torch.Tensor, int]:
    """``w[taps, Cin, Cout]`` (f32) -> 16-bit fragments ``[NB][taps*KC][wave 4][n 2][lane 64][8]`` (+1 spare step).

    Fragment element ``j`` of lane ``l`` is ``W[k = kc*32 + 8*(l>>4) + j][cout]`` with
    ``cout = nb*128 + wave*32 + (r>>2)*8 + n*4 + (r&3)``, ``r = l & 15``: after the MFMA
    a lane then holds eight consecutive output channels of one pixel.  Runs on ``device``.
    """
    taps, cin, cout = w.shape
    flat, nb_n = _pack_index(taps, cin, cout, device)
    wpad = torch.zeros(taps, cin, nb_n * 128, dtype=torch.float32, device=device)
    wpad[:, :, :cout] = w.detach().to(device=device, dtype=torch.float32)
    packed = wpad.reshape(-1)[flat]
    out = torch.cat([packed, torch.zeros(4 * 2 * 64 * 8, dtype=torch.float32, device=device)]).to(dtype)
    return out, nb_n


def pack_gemm_weights_split(w: torch.Tensor, device: torch.device) -> tuple[torch.Tensor, int, int]:
    """fp32 weights as two packed 16-bit fragment streams, ``hi = r16(w)`` and ``lo = r16(w - hi)``, concatenated
    (``r16`` = rounding to the library's split operand format, bf16 or fp16).

    Returns ``(packed, n_blocks, lo_offset_in_elements)``; both streams use the layout of :func:`pack_gemm_weights`.
    """
    dt = split_operand_dtype()
    w32 = w.detach().to(device=device, dtype=torch.float32)
    hi = w32.to(dt)
    lo = w32 - hi.float()
    p_hi, nb = pack_gemm_weights(hi.float(), device, dt)
    p_lo, _ = pack_gemm_weights(lo, device, dt)
    return torch.cat([p_hi, p_lo]), nb, int(p_hi.numel())


def _fold_bn(weight: torch.Tensor, bias: torch.Tensor | None, bn: nn.Module | None) -> tuple[torch.Tensor, torch.Tensor]:
    """Fold an eval-mode BatchNorm into ``weight`` (out-channel first) and ``bias``."""
    w = weight.detach().float()
    b = bias.detach().float() if bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None and not isinstance(bn, nn.Identity):
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        w = w * scale.view(-1, *([1] * (w.dim() - 1)))
        b = b * scale + shift
    return w, b


@dataclass
class _GemmLayer:
    w: torch.Tensor
    bias: torch.Tensor
    cin: int
    cout: int       # padded to a multiple of 8
    taps: int
    act: int
    n_blocks: int
    ln_g: torch.Tensor | None = None
    ln_b: torch.Tensor | None = None
    ln_eps: float = 1e-5
    post_add: torch.Tensor | None = None
    lo_off: int = 0   # fp32 mode: element offset of the lo fragment stream inside ``w``


def _split_seq(seq: nn.Sequential) -> list[tuple[nn.Module, nn.Module | None, bool]]:
    """Group a Sequential of (Linear|Conv, [BN], [act]) into ``(layer, bn, has_act)`` triples."""
    out: list[tuple[nn.Module, nn.Module | None, bool]] = []
    mods = list(seq)
    i = 0
    while i < len(mods):
        layer = mods[i]
        bn = None
        has_act = False
        i += 1
        if i < len(mods) and isinstance(mods[i], (nn.BatchNorm1d, nn.BatchNorm2d)):
            bn = mods[i]
            i += 1
        if i < len(mods) and not isinstance(mods[i], (nn.Linear, nn.Conv2d)):
            has_act = True
            i += 1
        out.append((layer, bn, has_act))
    return out


class InferenceEngine:
    def __init__(self, model_config: ModelConfig, env_config: EnvConfig, device: torch.device, dtype: str = "bf16"):
        ok, why = engine_supports(model_config, env_config, dtype)
        if not ok:
            raise ValueError(f"HIP inference engine does not support this config: {why}")
        hb.lib()  # fail loudly when the kernel library is missing
        self.dtype = dtype
        self.split = dtype == "fp32"  # fp32 activations, every GEMM as three bf16 MFMA passes over (hi, lo) splits
        self.mc = model_config
        self.ec = env_config
        self.device = torch.device(device)
        self.H, self.W = env_config.ROWS, env_config.COLS
        self.HW = self.H * self.W
        self.A = env_config.NUM_SHAPE_SLOTS * self.HW
        self.act = ACT_IDS[model_config.ACTIVATION_FUNCTION]
        self.other_dim = model_config.OTHER_NN_INPUT_FEATURES_DIM
        self.other_pad = (self.other_dim + 31) // 32 * 32
        self.rows_per_wg = max(1, 256 // self.HW) * self.HW if self.HW <= 256 else self.HW
        if self.rows_per_wg > 240 and self.HW <= 240:
            self.rows_per_wg = (240 // self.HW) * self.HW
        self.uses_tf = bool(model_config.USE_TRANSFORMER and model_config.TRANSFORMER_LAYERS > 0)
        self.logit_stride = (self.A + 7) // 8 * 8  # rows padded so the 8-channel output groups never straddle rows
        self.value_stride = (model_config.NUM_VALUE_ATOMS + 7) // 8 * 8
        self.cap = 0
        self.loaded = False
        self.fc_sk = 8
        self.fc_fused = False
        # one-workgroup-per-sample fused encoder-layer kernels (csrc/hip/tf_kernels.hip)
        self.fused_tf = bool(
            self.uses_tf and model_config.TRANSFORMER_DIM == 128 and model_config.TRANSFORMER_HEADS == 4
            and self.HW <= 128
        )
        # the FFN half of an encoder layer is token-wise: its fused kernel is tiled over flat rows and works for any board
        self.fused_ffn = bool(
            self.uses_tf and model_config.TRANSFORMER_DIM == 128 and model_config.TRANSFORMER_FC_DIM % 128 == 0
            and model_config.TRANSFORMER_FC_DIM <= 512
        )
        # attention half for boards the one-workgroup-per-sample kernel cannot hold (129..256 tokens): row-tiled QKV / out-proj
        # kernels around a register-resident attention core
        self.rows_attn = bool(
            self.uses_tf and not self.fused_tf and model_config.TRANSFORMER_DIM == 128 and model_config.TRANSFORMER_HEADS == 4
            and self.HW <= 256 and self.HW % 4 == 0
        )
        self.use_rowconv = True  # tests flip this to cover the generic tile kernel on small boards
        self.fuse_res_blocks = True  # tests flip this to compare the fused residual-block kernel with two launches
        self.fuse_stem = True  # tests flip this to compare the one-launch conv stem with per-layer launches
        self.tf_chunk = 0  # samples per pass of the 129..256-token layer stack (0 = whole batch; chunking measured no gain)
        self.fuse_layer = True  # tests flip this to compare the one-launch encoder layer with attn_block + ffn_block
        self.use_fused_heads = True  # tests flip this to compare the fused tail kernel with the per-layer launches
        self.heads_fused = False
        self._store: list[torch.Tensor] = []
        self._store_cursor = 0
        # tests set this to a list: the fp32-grade forward then appends (stage name, copy of the activation rows) after
        # every body / encoder launch, so a numerics failure names its kernel
        self.trace: list[tuple[str, torch.Tensor]] | None = None

    # ------------------------------------------------------------------ weights
    def _put(self, value: torch.Tensor) -> torch.Tensor:
        """Device tensor with a stable address across reloads (HIP graphs keep raw pointers)."""
        i = self._store_cursor
        self._store_cursor += 1
        value = value.to(self.device)
        if i < len(self._store) and self._store[i].shape == value.shape and self._store[i].dtype == value.dtype:
            self._store[i].copy_(value)
            return self._store[i]
        value = value.contiguous().clone()
        if i < len(self._store):
            self._store[i] = value
        else:
            self._store.append(value)
        return value

    def _gemm(self, w_tkc: torch.Tensor, bias: torch.Tensor, act: int, **kw: Any) -> _GemmLayer:
        taps, cin, cout = w_tkc.shape
        cout8 = (cout + 7) // 8 * 8
        lo_off = 0
        if self.split:
            packed, nb, lo_off = pack_gemm_weights_split(w_tkc, self.device)
        else:
            packed, nb = pack_gemm_weights(w_tkc, self.device)
        b = torch.zeros(nb * 128, dtype=torch.float32, device=self.device)
        b[:cout] = bias.to(self.device)
        for k in ("ln_g", "ln_b", "post_add"):
            if kw.get(k) is not None:
                kw[k] = self._put(kw[k])
        return _GemmLayer(self._put(packed), self._put(b), cin, cout8, taps, act, nb, lo_off=lo_off, **kw)

    def _conv_layer(self, conv: nn.Conv2d, bn: nn.Module | None, act: int, **kw: Any) -> _GemmLayer:
        w, b = _fold_bn(conv.weight, conv.bias, bn)  # [Cout, Cin, kh, kw]
        cout, cin, kh, kw_ = w.shape
        w_tkc = w.permute(2, 3, 1, 0).reshape(kh * kw_, cin, cout)
        return self._gemm(w_tkc, b, act, **kw)

    def _linear_layer(self, lin: nn.Linear, bn: nn.Module | None, act: int, **kw: Any) -> _GemmLayer:
        w, b = _fold_bn(lin.weight, lin.bias, bn)  # [N, K]
        return self._gemm(w.t().unsqueeze(0).contiguous(), b, act, **kw)

    @torch.no_grad()
    def load_from_module(self, model: nn.Module) -> None:
        mc, dev = self.mc, self.device
        D = mc.TRANSFORMER_DIM
        self._store_cursor = 0
        # ---- conv stem
        stem = [(_split_seq(blk)[0]) for blk in model.conv_body]
        conv0, bn0, _ = stem[0]
        w0, b0 = _fold_bn(conv0.weight, conv0.bias, bn0)  # [C0, 1, 3, 3]
        self.c0 = w0.shape[0]
        self.w_first = self._put(w0.reshape(self.c0, 9).t().contiguous())  # [9][C0]
        self.b_first = self._put(b0)
        self.body_layers: list[tuple[str, _GemmLayer]] = []
        for conv, bn, _ in stem[1:]:
            self.body_layers.append(("plain", self._conv_layer(conv, bn, self.act)))
        # ---- residual body
        for blk in model.res_body:
            if isinstance(blk, nn.Sequential):
                conv, bn, _ = _split_seq(blk)[0]
                self.body_layers.append(("plain", self._conv_layer(conv, bn, self.act)))
            else:
                c1, bn1, _ = _split_seq(blk.conv1)[0]
                self.body_layers.append(("res_a", self._conv_layer(c1, bn1, self.act)))
                self.body_layers.append(("res_b", self._conv_layer(blk.conv2, blk.bn2, self.act)))
        self.body_channels = self.body_layers[-1][1].cout if self.body_layers else self.c0
        # ---- transformer
        self.tf_layers: list[dict[str, Any]] = []
        self.proj: _GemmLayer | None = None
        self.final_ln: tuple[torch.Tensor | None, torch.Tensor | None, float] | None = None
        if self.uses_tf:
            pe = model.pos_encoder.pe[: self.HW, 0, :].detach().float().contiguous().to(dev)
            if isinstance(model.input_proj, nn.Conv2d):
                self.proj = self._conv_layer(model.input_proj, None, 0, post_add=pe)
            elif self.body_layers:
                self.body_layers[-1][1].post_add = self._put(pe)
            else:
                raise ValueError("transformer directly on the first conv layer is not supported")
            hd = D // mc.TRANSFORMER_HEADS
            # The fused / row-tiled encoder kernels only normalise; LayerNorm's affine part is folded into the
            # following projection: (xhat * g + b) W = xhat (g (.) W) + b W.  The tile-kernel fallback keeps g / b.
            fold_attn = self.fused_tf or self.rows_attn or self.split
            fold_ffn = self.fused_ffn or self.split
            for layer in model.transformer_body.layers:
                w_in = layer.self_attn.in_proj_weight.detach().float().clone()  # [3D, D]
                b_in = layer.self_attn.in_proj_bias.detach().float().clone()
                g1, be1 = layer.norm1.weight.detach().float(), layer.norm1.bias.detach().float()
                if fold_attn:
                    b_in = b_in + w_in @ be1
                    w_in = w_in * g1[None, :]
                qscale = math.log2(math.e) / math.sqrt(hd)  # the kernels' softmax is exp2-based
                w_in[:D] *= qscale
                b_in[:D] *= qscale
                qkv = self._gemm(
                    w_in.t().unsqueeze(0).contiguous(), b_in, 0,
                    ln_g=None if fold_attn else g1.to(dev), ln_b=None if fold_attn else be1.to(dev),
                    ln_eps=float(layer.norm1.eps),
                )
                out = self._linear_layer(layer.self_attn.out_proj, None, 0)
                w1 = layer.linear1.weight.detach().float().clone()  # [F, D]
                b1 = layer.linear1.bias.detach().float().clone()
                g2, be2 = layer.norm2.weight.detach().float(), layer.norm2.bias.detach().float()
                if fold_ffn:
                    b1 = b1 + w1 @ be2
                    w1 = w1 * g2[None, :]
                ff1 = self._gemm(
                    w1.t().unsqueeze(0).contiguous(), b1, self.act,
                    ln_g=None if fold_ffn else g2.to(dev), ln_b=None if fold_ffn else be2.to(dev),
                    ln_eps=float(layer.norm2.eps),
                )
                ff2 = self._linear_layer(layer.linear2, None, 0)
                self.tf_layers.append({"qkv": qkv, "out": out, "ff1": ff1, "ff2": ff2})
            fn = model.transformer_body.norm
            self.final_ln = (self._put(fn.weight.detach().float()), self._put(fn.bias.detach().float()), float(fn.eps))
            self.feat_c = D
        else:
            self.feat_c = self.body_channels
        # ---- shared FC (first layer split-K, the rest dense)
        shared = _split_seq(model.shared_fc)
        lin0, bnf, _ = shared[0]
        w, b = _fold_bn(lin0.weight, lin0.bias, bnf)  # [N, feat + other]
        feat = self.HW * self.feat_c
        wf = w[:, :feat]
        if self.split and self.uses_tf:
            # fp32 engine: the final LayerNorm's affine part goes into the FC (the kernel only normalises the rows)
            fn = model.transformer_body.norm
            gf, bf = fn.weight.detach().float().to(w.device), fn.bias.detach().float().to(w.device)
            b = b + wf @ bf.repeat(self.HW)
            wf = wf * gf.repeat(self.HW)[None, :]
            self.final_ln = (None, None, float(fn.eps))
        if not self.uses_tf:  # PyTorch flattens NCHW (c*HW + p); activations here are NHWC (p*C + c)
            wf = wf.reshape(-1, self.feat_c, self.HW).permute(0, 2, 1).reshape(-1, feat)
        wo = torch.zeros(w.shape[0], self.other_pad, device=w.device)
        wo[:, : self.other_dim] = w[:, feat : feat + self.other_dim]
        wk = torch.cat([wf, wo], dim=1).t().contiguous()  # [Kp, N]
        self.fc_kp = wk.shape[0]
        self.fc_n = wk.shape[1]
        if self.split:
            fc_packed, _, self.fc_lo = pack_gemm_weights_split(wk.unsqueeze(0), dev)
            self.fc_w = self._put(fc_packed)
        else:
            self.fc_lo = 0
            self.fc_w = self._put(pack_gemm_weights(wk.unsqueeze(0), dev)[0])
        self.fc_b = self._put(b)
        self.shared_rest = [self._linear_layer(l, bn, self.act if a else 0) for l, bn, a in shared[1:]]
        self.policy_layers = [self._linear_layer(l, bn, self.act if a else 0) for l, bn, a in _split_seq(model.policy_head)]
        self.value_layers = [self._linear_layer(l, bn, self.act if a else 0) for l, bn, a in _split_seq(model.value_head)]
        total_steps = self.fc_kp // 32
        # K splits of the shared FC.  The FC kernels run one 8-wave workgroup per CU on 64 (fp32) or 128 (bf16) samples
        # x one split: 4 / 8 splits of a 4096-leaf wave are one workgroup per CU of a 256-CU device.
        self.fc_sk = max(1, min(4 if self.split else 8, total_steps // 16))
        # 128-channel bodies feed the FC straight from their NHWC rows (flatten + final LayerNorm fused into its staging)
        self.fc_fused = self.feat_c == 128 and self.fc_n <= 128
        if self.fc_fused:
            self.fc_steps_per_split = ((total_steps + self.fc_sk - 1) // self.fc_sk + 3) // 4 * 4
            self.fc_sk = (total_steps + self.fc_steps_per_split - 1) // self.fc_steps_per_split
        # with a device-side live count below the batch, the fused FC's workgroups re-deal themselves over the row blocks
        # that hold leaves and up to this many (shorter) K splits; the fused tail kernel derives the same plan.  On by
        # default in the fp32 engine only: the bf16 FC is half as long to begin with and measured no gain from it.
        self.fc_max_splits = int(os.environ.get("ATRI_FC_MAX_SPLITS", "16" if self.split else "0")) if self.fc_fused else 0
        self.fc_total_steps = total_steps
        # default-shaped tail (128-wide shared / policy / value hidden layers): one fused kernel
        pl, vl = self.policy_layers, self.value_layers
        self.heads_fused = (
            self.fc_n == 128 and not self.shared_rest and len(pl) == 2 and len(vl) == 2
            and all(l.cin == 128 and l.taps == 1 and l.ln_g is None and l.post_add is None for l in pl + vl)
            and pl[0].cout == 128 and vl[0].cout == 128 and pl[0].act == self.act and vl[0].act == self.act
            and pl[1].act == 0 and vl[1].act == 0 and vl[1].cout <= 128 and pl[1].cout <= self.logit_stride
            and vl[1].cout <= self.value_stride
        )
        self.loaded = True

    # ------------------------------------------------------------------ buffers
    def reserve(self, batch: int) -> None:
        """Allocate activation / output buffers for up to ``batch`` samples (pointers stay stable until it grows)."""
        if batch <= self.cap:
            return
        dev = self.device
        rows = batch * self.HW
        widths = [self.c0, self.body_channels, self.feat_c]
        for _, l in self.body_layers:
            widths.append(l.cout)
        if self.uses_tf:
            widths += [3 * self.mc.TRANSFORMER_DIM, self.mc.TRANSFORMER_FC_DIM]
        maxc = max(widths)
        if self.split:
            self.bufs = [torch.empty(rows * 128, dtype=torch.float32, device=dev) for _ in range(3)]
            self.fc_in = self.vt = None
            self.slabs = torch.empty(max(self.fc_sk, self.fc_max_splits), batch, 128, dtype=torch.float32, device=dev)
            self.dense = []
            self.logits = torch.zeros(batch, self.logit_stride, dtype=torch.float32, device=dev)
            self.vlogits = torch.zeros(batch, self.value_stride, dtype=torch.float32, device=dev)
            self.values = torch.zeros(batch, dtype=torch.float32, device=dev)
            self.cap = batch
            return
        self.bufs = [torch.empty(rows * maxc, dtype=torch.bfloat16, device=dev) for _ in range(5)]
        self.fc_in = None if self.fc_fused else torch.zeros(batch, self.fc_kp, dtype=torch.bfloat16, device=dev)
        self.vt = torch.zeros(batch * 4 * 32 * 256, dtype=torch.bfloat16, device=dev) if self.rows_attn else None
        self.slabs = torch.empty(max(self.fc_sk, self.fc_max_splits), batch, 128, dtype=torch.float32, device=dev)
        dense_w = max([self.fc_n] + [l.cout for l in self.shared_rest + self.policy_layers + self.value_layers])
        self.dense = [torch.empty(batch * dense_w, dtype=torch.bfloat16, device=dev) for _ in range(3)]
        self.logits = torch.zeros(batch, self.logit_stride, dtype=torch.float32, device=dev)
        self.vlogits = torch.zeros(batch, self.value_stride, dtype=torch.float32, device=dev)
        self.values = torch.zeros(batch, dtype=torch.float32, device=dev)
        self.cap = batch

    # ------------------------------------------------------------------ launches
    def _run_gemm(self, layer: _GemmLayer, src: torch.Tensor, dst: torch.Tensor, rows: int, *, conv: bool,
                  residual: torch.Tensor | None = None, out_f32: bool = False, out_stride: int | None = None) -> None:
        if (conv and layer.taps == 9 and self.use_rowconv and not out_f32 and out_stride is None and layer.ln_g is None
                and self.H <= 8 and self.W <= 15 and 32 <= layer.cin <= 128 and _pow2(layer.cout)):
            # halo-tiled one-sample-per-workgroup conv (csrc/hip/conv_kernels.hip)
            hb.call("atri_rowconv", hb.ptr(src), hb.ptr(layer.w), hb.ptr(layer.bias), hb.ptr(residual), hb.ptr(layer.post_add),
                    hb.ptr(dst), hb.i32(rows // self.HW), hb.i32(self.H), hb.i32(self.W), hb.i32(layer.cin), hb.i32(layer.cout),
                    hb.i32(layer.act), hb.i32(layer.n_blocks))
            return
        if (conv and layer.taps == 9 and self.use_rowconv and not out_f32 and out_stride is None and layer.ln_g is None
                and layer.n_blocks == 1 and self._flat_ok(layer.cin, layer.cout)):
            # flat-tiled one-sample-per-workgroup conv for boards above 8x15 (csrc/hip/flatconv_kernels.hip)
            hb.call("atri_flatconv", hb.ptr(src), hb.ptr(layer.w), hb.ptr(layer.bias), hb.ptr(residual), hb.ptr(layer.post_add),
                    hb.ptr(dst), hb.i32(rows // self.HW), hb.i32(self.H), hb.i32(self.W), hb.i32(layer.cin), hb.i32(layer.cout),
                    hb.i32(layer.act))
            return
        if conv:
            hw, h, w, rpw = self.HW, self.H, self.W, self.rows_per_wg
        else:
            # dense layers (rows = samples): small row blocks so a few thousand rows still cover every CU
            rpw = min(240, max(16, 16 * ((rows + 16 * 256 - 1) // (16 * 256))))
            hw, h, w = 1, 1, 1
        hb.call(
            "atri_tile_gemm",
            hb.ptr(src), hb.ptr(layer.w), hb.ptr(layer.bias), hb.ptr(residual), hb.ptr(layer.post_add),
            hb.ptr(layer.ln_g), hb.ptr(layer.ln_b), hb.ptr(dst),
            hb.i32(rows), hb.i32(rpw), hb.i32(hw), hb.i32(h), hb.i32(w), hb.i32(layer.cin), hb.i32(layer.cout),
            hb.i32(layer.taps), hb.i32(layer.act), hb.i32(1 if out_f32 else 0),
            hb.i32(out_stride if out_stride is not None else layer.cout), hb.i32(layer.n_blocks), hb.f32(layer.ln_eps),
        )

    def _flat_ok(self, cin: int, cout: int) -> bool:
        """The flat-tile conv kernel serves this layer: a board the 8x15 halo-tile kernel refuses, H * (W + 1) <= 320."""
        if self.H <= 8 and self.W <= 15:
            return False
        return bool(hb.lib().atri_flatconv_supported(int(self.H), int(self.W), int(cin), int(cout)))

    def _pair_ok(self, l1: _GemmLayer, l2: _GemmLayer) -> bool:
        """Both convolutions of a residual block fit a fused block kernel (128 channels, same activation; the halo-tile
        kernel up to 8x15 boards, the flat-tile kernel above)."""
        return (
            self.use_rowconv and self.fuse_res_blocks and ((self.H <= 8 and self.W <= 15) or self._flat_ok(128, 128))
            and l1.taps == 9 and l2.taps == 9 and l1.cin == l1.cout == l2.cin == l2.cout == 128
            and l1.act == l2.act and l1.post_add is None and l1.ln_g is None and l2.ln_g is None
        )

    def _forward_split(self, grid: torch.Tensor, other: torch.Tensor, B: int) -> tuple[torch.Tensor, torch.Tensor]:
        """fp32-grade evaluation (csrc/hip/split_*.hip): fp32 activations between launches, one launch per stem /
        residual block / encoder layer, split-K FC with the final LayerNorm fused, one tail launch."""
        bufs = self.bufs
        body = self.body_layers
        l1, l2 = body[0][1], body[1][1]
        small = self.H <= 8 and self.W <= 15
        hb.call("atri_split_conv_stem" if small else "atri_split_flatconv_stem", hb.ptr(grid), hb.ptr(self.w_first), hb.ptr(self.b_first), hb.ptr(l1.w), hb.i64(l1.lo_off),
                hb.ptr(l1.bias), hb.ptr(l2.w), hb.i64(l2.lo_off), hb.ptr(l2.bias), hb.ptr(l2.post_add), hb.ptr(bufs[0]),
                hb.i32(B), hb.i32(self.H), hb.i32(self.W), hb.i32(self.c0), hb.i32(l1.cout), hb.i32(l2.cout), hb.i32(self.act))
        cur = 0
        rest = body[2:]

        def note(name: str) -> None:
            if self.trace is not None:
                self.trace.append((name, bufs[cur][: B * self.HW * 128].view(B, self.HW, 128).clone()))

        note("stem")
        for li in range(0, len(rest), 2):
            (ka, la), (kb, lb) = rest[li], rest[li + 1]
            assert ka == "res_a" and kb == "res_b"
            dst = (cur + 1) % 3
            hb.call("atri_split_conv_pair" if small else "atri_split_flatconv_pair", hb.ptr(bufs[cur]), hb.ptr(la.w), hb.i64(la.lo_off), hb.ptr(la.bias), hb.ptr(lb.w),
                    hb.i64(lb.lo_off), hb.ptr(lb.bias), hb.ptr(lb.post_add), hb.ptr(bufs[dst]), hb.i32(B), hb.i32(self.H),
                    hb.i32(self.W), hb.i32(la.cin), hb.i32(la.act))
            cur = dst
            note(f"block{li // 2}")
        ln = None
        if self.uses_tf:
            for tl in self.tf_layers:
                qkv, out, ff1, ff2 = tl["qkv"], tl["out"], tl["ff1"], tl["ff2"]
                dst = (cur + 1) % 3
                if self.HW > 128:
                    # 129..256 tokens: (LN1 + QKV + attention) -> (out-projection + FFN over 128-row tiles)
                    att = (cur + 2) % 3
                    hb.call("atri_split_qkv_attn", hb.ptr(bufs[cur]), hb.ptr(bufs[att]), hb.ptr(qkv.w), hb.i64(qkv.lo_off),
                            hb.ptr(qkv.bias), hb.f32(qkv.ln_eps), hb.i32(B), hb.i32(self.HW))
                    hb.call("atri_split_proj_ffn_rows", hb.ptr(bufs[att]), hb.ptr(bufs[cur]), hb.ptr(bufs[dst]), hb.ptr(out.w),
                            hb.i64(out.lo_off), hb.ptr(out.bias), hb.ptr(ff1.w), hb.i64(ff1.lo_off), hb.ptr(ff1.bias), hb.ptr(ff2.w),
                            hb.i64(ff2.lo_off), hb.ptr(ff2.bias), hb.f32(ff1.ln_eps), hb.i32(B), hb.i32(self.HW), hb.i32(ff1.cout),
                            hb.i32(self.act))
                    if self.trace is not None:
                        self.trace.append((f"attn{len(self.trace)}", bufs[att][: B * self.HW * 128].view(B, self.HW, 128).clone()))
                    cur = dst
                    note("layer")
                    continue
                hb.call("atri_split_encoder_layer", hb.ptr(bufs[cur]), hb.ptr(bufs[dst]), hb.ptr(qkv.w), hb.i64(qkv.lo_off),
                        hb.ptr(qkv.bias), hb.ptr(out.w), hb.i64(out.lo_off), hb.ptr(out.bias), hb.f32(qkv.ln_eps), hb.ptr(ff1.w),
                        hb.i64(ff1.lo_off), hb.ptr(ff1.bias), hb.ptr(ff2.w), hb.i64(ff2.lo_off), hb.ptr(ff2.bias), hb.f32(ff1.ln_eps),
                        hb.i32(B), hb.i32(self.HW), hb.i32(ff1.cout), hb.i32(self.act))
                cur = dst
                note("layer")
            ln = self.final_ln
        g, bt, eps = ln if ln is not None else (None, None, 0.0)
        hb.call("atri_split_fc_rows_dyn", hb.ptr(bufs[cur]), hb.ptr(other), hb.i32(self.other_dim), hb.ptr(g), hb.ptr(bt), hb.f32(eps),
                hb.ptr(self.fc_w), hb.i64(self.fc_lo), hb.ptr(self.slabs), hb.i32(B), hb.i32(self.HW), hb.i32(self.feat_c),
                hb.i32(self.other_pad), hb.i32(self.fc_n), hb.i32(self.fc_sk), hb.i32(self.fc_steps_per_split),
                hb.i32(self.fc_max_splits))
        p1, p2 = self.policy_layers
        v1, v2 = self.value_layers
        dz = (self.mc.VALUE_MAX - self.mc.VALUE_MIN) / (self.mc.NUM_VALUE_ATOMS - 1)
        hb.call("atri_split_heads_dyn", hb.ptr(self.slabs), hb.i32(self.fc_sk), hb.i32(self.fc_steps_per_split),
                hb.i32(self.fc_total_steps), hb.i32(self.fc_max_splits), hb.ptr(self.fc_b), hb.ptr(p1.w), hb.i64(p1.lo_off),
                hb.ptr(p1.bias), hb.ptr(p2.w), hb.i64(p2.lo_off), hb.ptr(p2.bias), hb.i32(p2.n_blocks), hb.i32(p2.cout),
                hb.ptr(v1.w), hb.i64(v1.lo_off), hb.ptr(v1.bias), hb.ptr(v2.w), hb.i64(v2.lo_off), hb.ptr(v2.bias),
                hb.i32(self.mc.NUM_VALUE_ATOMS), hb.i32(v2.cout), hb.ptr(self.logits), hb.i32(self.logit_stride),
                hb.ptr(self.vlogits), hb.i32(self.value_stride), hb.ptr(self.values), hb.f32(self.mc.VALUE_MIN), hb.f32(dz),
                hb.i32(B), hb.i32(self.act))
        return self.logits[:B, : self.A], self.values[:B]

    @torch.no_grad()
    def forward(self, grid: torch.Tensor, other: torch.Tensor, live: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """``grid`` f32 [B, 1, H, W] (or [B, HW]), ``other`` f32 [B, Do] -> ``(logits [B, A], value [B])``.

        The returned tensors are views of the engine's persistent output buffers.  ``live`` (int32 [1] on the device):
        only rows ``0 .. live-1`` hold leaves; the batch kernels are launched for all ``B`` rows and every workgroup
        beyond the live count ends at once, so a search wave that needs fewer evaluations costs less without any host
        synchronisation (the count is read on the device, also when the launches are replayed from a HIP graph).
        Kernels of the less common paths that do not read the count simply evaluate all rows.
        """
        if live is not None and (live.dtype != torch.int32 or live.numel() != 1 or not live.is_cuda):
            raise ValueError("live must be one int32 on the engine's device")
        set_live = getattr(hb.lib(), "atri_set_live_count")
        with _LIVE_LOCK:  # also without a count: another thread's count must not leak into these launches
            set_live(hb.ptr(live))
            try:
                return self._forward_all(grid, other)
            finally:
                set_live(hb.VP(0))

    def _forward_all(self, grid: torch.Tensor, other: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        if not self.loaded:
            raise RuntimeError("InferenceEngine.load_from_module has not been called")
        B = int(grid.shape[0])
        if B == 0:
            return torch.empty(0, self.A, device=self.device), torch.empty(0, device=self.device)
        self.reserve(B)
        if grid.dtype != torch.float32 or not grid.is_contiguous():
            grid = grid.float().contiguous()
        if other.dtype != torch.float32 or not other.is_contiguous():
            other = other.float().contiguous()
        if grid.numel() != B * self.HW or other.numel() != B * self.other_dim:
            raise ValueError("feature tensor shapes do not match the engine configuration")
        if self.split:
            return self._forward_split(grid, other, B)
        rows = B * self.HW
        bufs = self.bufs
        cur = 0  # index of the buffer holding the current activation
        body = self.body_layers
        fuse_first = (
            self.use_rowconv and body and body[0][0] == "plain" and body[0][1].post_add is None and body[0][1].taps == 9
            and self.H <= 8 and self.W <= 15 and self.c0 in (32, 64, 128) and _pow2(body[0][1].cout)
        )
        fuse_stem = (
            fuse_first and self.fuse_stem and len(body) >= 2 and body[1][0] == "plain" and self.c0 == 32
            and body[0][1].cout == 64 and body[1][1].cin == 64 and body[1][1].cout == 128 and body[1][1].taps == 9
            and body[0][1].act == self.act and body[1][1].act == self.act and body[1][1].ln_g is None and self.HW <= 128
        )
        flat_stem = (
            not fuse_stem and self.use_rowconv and self.fuse_stem and len(body) >= 2 and body[0][0] == "plain"
            and body[1][0] == "plain" and self.c0 == 32 and body[0][1].taps == 9 and body[1][1].taps == 9
            and body[0][1].cin == 32 and body[0][1].cout == 64 and body[1][1].cin == 64 and body[1][1].cout == 128
            and body[0][1].post_add is None and body[0][1].ln_g is None and body[1][1].ln_g is None
            and body[0][1].act == self.act and body[1][1].act == self.act and self._flat_ok(32, 64) and self._flat_ok(64, 128)
        )
        if flat_stem:
            # the same one-launch conv stem on a board above 8x15 (csrc/hip/flatconv_kernels.hip)
            l1, l2 = body[0][1], body[1][1]
            hb.call("atri_flatconv_stem", hb.ptr(grid), hb.ptr(self.w_first), hb.ptr(self.b_first), hb.ptr(l1.w), hb.ptr(l1.bias),
                    hb.ptr(l2.w), hb.ptr(l2.bias), hb.ptr(l2.post_add), hb.ptr(bufs[1]), hb.i32(B), hb.i32(self.H), hb.i32(self.W),
                    hb.i32(self.c0), hb.i32(l1.cout), hb.i32(l2.cout), hb.i32(self.act))
            cur = 1
            body = body[2:]
        elif fuse_stem:
            # whole conv stem (1 -> 32 -> 64 -> 128 channels) in one launch: only the last layer's output goes to memory
            l1, l2 = body[0][1], body[1][1]
            hb.call("atri_rowconv_stem", hb.ptr(grid), hb.ptr(self.w_first), hb.ptr(self.b_first), hb.ptr(l1.w), hb.ptr(l1.bias),
                    hb.ptr(l2.w), hb.ptr(l2.bias), hb.ptr(l2.post_add), hb.ptr(bufs[1]), hb.i32(B), hb.i32(self.H), hb.i32(self.W),
                    hb.i32(self.c0), hb.i32(l1.cout), hb.i32(l2.cout), hb.i32(self.act))
            cur = 1
            body = body[2:]
        elif fuse_first:
            # first layer (1 input channel) evaluated inside the staging of the second one: no separate kernel
            l1 = body[0][1]
            hb.call("atri_rowconv_first", hb.ptr(grid), hb.ptr(self.w_first), hb.ptr(self.b_first), hb.i32(self.act), hb.ptr(l1.w),
                    hb.ptr(l1.bias), hb.ptr(bufs[1]), hb.i32(B), hb.i32(self.H), hb.i32(self.W), hb.i32(l1.cin), hb.i32(l1.cout),
                    hb.i32(l1.act), hb.i32(l1.n_blocks))
            cur = 1
            body = body[1:]
        else:
            hb.call("atri_conv_first", hb.ptr(grid), hb.ptr(self.w_first), hb.ptr(self.b_first), hb.ptr(bufs[0]),
                    hb.i32(rows), hb.i32(self.HW), hb.i32(self.H), hb.i32(self.W), hb.i32(self.c0), hb.i32(self.act))
        skip_next = False
        for li, (kind, layer) in enumerate(body):
            if skip_next:
                skip_next = False
                continue
            if kind == "res_a" and self._pair_ok(layer, body[li + 1][1]):
                # whole residual block in one launch (the intermediate activation never leaves LDS)
                l2 = body[li + 1][1]
                dst = (cur + 1) % 3
                pair_fn = "atri_rowconv_pair" if (self.H <= 8 and self.W <= 15) else "atri_flatconv_pair"
                hb.call(pair_fn, hb.ptr(bufs[cur]), hb.ptr(layer.w), hb.ptr(layer.bias), hb.ptr(l2.w), hb.ptr(l2.bias),
                        hb.ptr(l2.post_add), hb.ptr(bufs[dst]), hb.i32(B), hb.i32(self.H), hb.i32(self.W), hb.i32(layer.cin),
                        hb.i32(layer.act))
                cur = dst
                skip_next = True
                continue
            if kind in ("plain", "res_a"):
                dst = (cur + 1) % 3
                self._run_gemm(layer, bufs[cur], bufs[dst], rows, conv=True)
                if kind == "res_a":
                    res_in, cur = cur, dst
                else:
                    cur = dst
            else:  # res_b: second conv of a residual block, skip connection from the block input
                dst = 3 - cur - res_in  # the third of buffers {0,1,2}
                self._run_gemm(layer, bufs[cur], bufs[dst], rows, conv=True, residual=bufs[res_in])
                cur = dst
        if self.uses_tf:
            if self.proj is not None:
                dst = (cur + 1) % 3
                self._run_gemm(self.proj, bufs[cur], bufs[dst], rows, conv=True)
                cur = dst
            D, heads = self.mc.TRANSFORMER_DIM, self.mc.TRANSFORMER_HEADS
            x = cur
            rows_path = (
                not self.fused_tf and self.rows_attn and self.fused_ffn and self.fuse_layer and self.act in (1, 2)
            )
            if rows_path:
                # 129..256 tokens: (LN1 + QKV + attention) -> (out-proj + FFN) rows per layer.  tf_chunk > 0 sends the batch
                # through the whole layer stack in chunks that share one chunk's worth of scratch.
                free = [i for i in range(5) if i != x]
                src, dst, q, t, h = x, free[0], free[1], free[2], free[3]
                chunk = self.tf_chunk if 0 < self.tf_chunk < B else B
                for s0 in range(0, B, chunk):
                    n = min(chunk, B - s0)
                    off = s0 * self.HW * D
                    a, b = src, dst
                    for tl in self.tf_layers:
                        qkv, out, ff1, ff2 = tl["qkv"], tl["out"], tl["ff1"], tl["ff2"]
                        hb.call("atri_layer_rows", hb.ptr(bufs[a][off:]), hb.ptr(bufs[b][off:]), hb.ptr(qkv.w), hb.ptr(qkv.bias),
                                hb.ptr(out.w), hb.ptr(out.bias), hb.ptr(qkv.ln_g), hb.ptr(qkv.ln_b), hb.f32(qkv.ln_eps),
                                hb.ptr(ff1.w), hb.ptr(ff1.bias), hb.ptr(ff2.w), hb.ptr(ff2.bias), hb.ptr(ff1.ln_g), hb.ptr(ff1.ln_b),
                                hb.f32(ff1.ln_eps), hb.ptr(bufs[q]), hb.ptr(bufs[t]), hb.ptr(self.vt), hb.ptr(bufs[h]), hb.i32(n),
                                hb.i32(self.HW), hb.i32(D), hb.i32(heads), hb.i32(ff1.cout), hb.i32(self.act))
                        a, b = b, a
                x = src if len(self.tf_layers) % 2 == 0 else dst
            for tl in (() if rows_path else self.tf_layers):
                free = [i for i in range(5) if i != x]
                q, t, y, h = free[0], free[1], free[2], free[3]
                qkv, out, ff1, ff2 = tl["qkv"], tl["out"], tl["ff1"], tl["ff2"]
                if self.fused_tf and self.fused_ffn and self.fuse_layer and self.act in (1, 2):
                    # whole layer in one launch: bufs[x] -> bufs[q] (the residual stream stays in registers)
                    hb.call("atri_encoder_layer", hb.ptr(bufs[x]), hb.ptr(bufs[q]), hb.ptr(qkv.w), hb.ptr(qkv.bias),
                            hb.ptr(out.w), hb.ptr(out.bias), hb.ptr(qkv.ln_g), hb.ptr(qkv.ln_b), hb.f32(qkv.ln_eps),
                            hb.ptr(ff1.w), hb.ptr(ff1.bias), hb.ptr(ff2.w), hb.ptr(ff2.bias), hb.ptr(ff1.ln_g), hb.ptr(ff1.ln_b),
                            hb.f32(ff1.ln_eps), hb.i32(B), hb.i32(self.HW), hb.i32(D), hb.i32(heads), hb.i32(ff1.cout),
                            hb.i32(self.act))
                    x = q
                    continue
                # ---- attention half: bufs[x] -> bufs[y]
                if self.fused_tf:
                    hb.call("atri_attn_block", hb.ptr(bufs[x]), hb.ptr(bufs[y]), hb.ptr(qkv.w), hb.ptr(qkv.bias),
                            hb.ptr(out.w), hb.ptr(out.bias), hb.ptr(qkv.ln_g), hb.ptr(qkv.ln_b), hb.i32(B), hb.i32(self.HW),
                            hb.i32(D), hb.i32(heads), hb.f32(qkv.ln_eps))
                elif self.rows_attn:
                    hb.call("atri_attn_rows", hb.ptr(bufs[x]), hb.ptr(bufs[y]), hb.ptr(qkv.w), hb.ptr(qkv.bias), hb.ptr(out.w),
                            hb.ptr(out.bias), hb.ptr(qkv.ln_g), hb.ptr(qkv.ln_b), hb.ptr(bufs[q]), hb.ptr(bufs[t]), hb.ptr(self.vt),
                            hb.ptr(bufs[h]), hb.i32(B), hb.i32(self.HW), hb.i32(D), hb.i32(heads), hb.f32(qkv.ln_eps))
                else:
                    self._run_gemm(qkv, bufs[x], bufs[q], rows, conv=True)
                    hb.call("atri_attention", hb.ptr(bufs[q]), hb.ptr(bufs[t]), hb.i32(B), hb.i32(self.HW), hb.i32(D), hb.i32(heads))
                    self._run_gemm(out, bufs[t], bufs[y], rows, conv=True, residual=bufs[x])
                # ---- FFN half: bufs[y] -> bufs[q]
                if self.fused_ffn:
                    hb.call("atri_ffn_block", hb.ptr(bufs[y]), hb.ptr(bufs[q]), hb.ptr(ff1.w), hb.ptr(ff1.bias),
                            hb.ptr(ff2.w), hb.ptr(ff2.bias), hb.ptr(ff1.ln_g), hb.ptr(ff1.ln_b), hb.i32(B), hb.i32(self.HW),
                            hb.i32(D), hb.i32(ff1.cout), hb.i32(self.act), hb.f32(ff1.ln_eps))
                else:
                    self._run_gemm(ff1, bufs[y], bufs[h], rows, conv=True)
                    self._run_gemm(ff2, bufs[h], bufs[q], rows, conv=True, residual=bufs[y])
                x = q
            cur = x
            ln = self.final_ln
        else:
            ln = None
        fused_tail = self.heads_fused and self.use_fused_heads
        fc_splits = self.fc_max_splits if self.fc_fused and fused_tail else 0  # only the fused tail knows the plan
        if self.fc_fused:
            g, bt, eps = ln if ln is not None else (None, None, 0.0)
            hb.call("atri_fc_rows_splitk_dyn", hb.ptr(bufs[cur]), hb.ptr(other), hb.i32(self.other_dim), hb.ptr(g), hb.ptr(bt), hb.f32(eps),
                    hb.ptr(self.fc_w), hb.ptr(self.slabs), hb.i32(B), hb.i32(self.HW), hb.i32(self.feat_c), hb.i32(self.other_pad),
                    hb.i32(self.fc_n), hb.i32(self.fc_sk), hb.i32(self.fc_steps_per_split), hb.i32(fc_splits))
        else:
            g, bt, eps = ln if ln is not None else (None, None, 0.0)
            hb.call("atri_rows_to_fc", hb.ptr(bufs[cur]), hb.ptr(g), hb.ptr(bt), hb.ptr(self.fc_in), hb.i32(rows),
                    hb.i32(self.HW), hb.i32(self.feat_c), hb.i32(self.fc_kp), hb.f32(eps))
            hb.call("atri_other_to_fc", hb.ptr(other), hb.ptr(self.fc_in), hb.i32(B), hb.i32(self.other_dim),
                    hb.i32(self.other_pad), hb.i32(self.fc_kp), hb.i32(self.HW * self.feat_c))
            hb.call("atri_fc_splitk", hb.ptr(self.fc_in), hb.ptr(self.fc_w), hb.ptr(self.slabs), hb.i32(B), hb.i32(self.fc_kp),
                    hb.i32(self.fc_n), hb.i32(self.fc_sk))
        if fused_tail:
            # split-K reduction, both heads and the C51 expectation in one launch (csrc/hip/head_kernels.hip)
            p1, p2 = self.policy_layers
            v1, v2 = self.value_layers
            dz = (self.mc.VALUE_MAX - self.mc.VALUE_MIN) / (self.mc.NUM_VALUE_ATOMS - 1)
            block_rows = getattr(hb.lib(), "atri_fc_rows_splitk_block_rows")(hb.i32(B), hb.i32(self.fc_sk)) if fc_splits else 0
            hb.call("atri_heads_dyn", hb.ptr(self.slabs), hb.i32(self.fc_sk), hb.i32(block_rows),
                    hb.i32(self.fc_steps_per_split if fc_splits else 0), hb.i32(self.fc_total_steps), hb.i32(fc_splits),
                    hb.ptr(self.fc_b), hb.ptr(p1.w), hb.ptr(p1.bias),
                    hb.ptr(p2.w), hb.ptr(p2.bias), hb.i32(p2.n_blocks), hb.i32(p2.cout), hb.ptr(v1.w), hb.ptr(v1.bias),
                    hb.ptr(v2.w), hb.ptr(v2.bias), hb.i32(self.mc.NUM_VALUE_ATOMS), hb.i32(v2.cout), hb.ptr(self.logits),
                    hb.i32(self.logit_stride), hb.ptr(self.vlogits), hb.i32(self.value_stride), hb.ptr(self.values),
                    hb.f32(self.mc.VALUE_MIN), hb.f32(dz), hb.i32(B), hb.i32(self.act))
            return self.logits[:B, : self.A], self.values[:B]
        # slabs are laid out [SK][cap][128]; the kernel indexes them with the live batch size
        hb.call("atri_fc_finish", hb.ptr(self.slabs), hb.ptr(self.fc_b), hb.ptr(self.dense[0]), hb.i32(B), hb.i32(self.fc_n),
                hb.i32(self.fc_sk), hb.i32(self.act))
        d = 0
        for layer in self.shared_rest:
            self._run_gemm(layer, self.dense[d], self.dense[1 - d], B, conv=False)
            d = 1 - d
        shared = d
        src = shared
        for i, layer in enumerate(self.policy_layers):
            last = i == len(self.policy_layers) - 1
            if last:
                self._run_gemm(layer, self.dense[src], self.logits, B, conv=False, out_f32=True, out_stride=self.logit_stride)
            else:
                dst = 2 if src == shared else (1 - shared if src == 2 else 2)
                self._run_gemm(layer, self.dense[src], self.dense[dst], B, conv=False)
                src = dst
        src = shared
        for i, layer in enumerate(self.value_layers):
            last = i == len(self.value_layers) - 1
            if last:
                self._run_gemm(layer, self.dense[src], self.vlogits, B, conv=False, out_f32=True, out_stride=self.value_stride)
            else:
                dst = 2 if src == shared else (1 - shared if src == 2 else 2)
                self._run_gemm(layer, self.dense[src], self.dense[dst], B, conv=False)
                src = dst
        dz = (self.mc.VALUE_MAX - self.mc.VALUE_MIN) / (self.mc.NUM_VALUE_ATOMS - 1)
        hb.call("atri_c51_expect", hb.ptr(self.vlogits), hb.i32(self.value_stride), hb.i32(self.mc.NUM_VALUE_ATOMS),
                hb.f32(self.mc.VALUE_MIN), hb.f32(dz), hb.ptr(self.values), hb.i32(B))
        return self.logits[:B, : self.A], self.values[:B]
