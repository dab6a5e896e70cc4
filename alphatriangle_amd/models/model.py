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


def _grad_dtype(partial: torch.Tensor) -> torch.dtype:
    """Weight gradients of half-precision GEMMs are summed straight into fp32 (the parameters' dtype under autocast),
    which saves the cast kernel autograd would otherwise append."""
    return torch.float32 if partial.dtype in (torch.bfloat16, torch.float16) else partial.dtype


def _autocast_dtype(t: torch.Tensor) -> torch.dtype | None:
    """The dtype autocast would run a GEMM on ``t``'s device in, or ``None`` outside autocast."""
    if t.is_cuda and torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return None


def _as_dtype(weight: torch.Tensor, dtype: torch.dtype | None) -> torch.Tensor:
    """``weight`` in ``dtype``: the optimizer-side low-precision view when one is attached
    (``ops.optim.FlatParameterSpace.refresh_shadow``), else a cast."""
    if dtype is None or weight.dtype == dtype:
        return weight
    shadow = getattr(weight, "_atri_shadow", None)
    if shadow is not None and shadow.dtype == dtype and shadow.shape == weight.shape:
        return shadow
    return weight.to(dtype)


# fp32 CUDA products of the GEMM-lowered layers run on the bf16 matrix cores (three passes over (hi, lo) operand splits,
# ops.hip_learner.mm3) instead of the library's native-fp32 GEMMs; False restores the library path (A/B runs, tests).
FP32_GEMM_ON_BF16_CORES = True
# 3x3 convolutions of the fp32 path without a patch matrix in memory (ops.hip_learner.conv3_mm / conv3_wgrad: the GEMM's
# operand loader generates the patches): nine times less activation memory per layer kept for the backward pass, about
# 75 % slower products (the loader's per-value address and bounds arithmetic costs more than the patch traffic it saves;
# the plain products already run at the speed their patch matrix can be read at).  SystemConfig.LEARNER_PATCH_FREE_CONV.
PATCH_FREE_CONV = False
# 3x3 / padding-1 convolutions of the fp32 path (forward and data gradient) on the board-per-workgroup kernel
# (ops.hip_learner.conv3_cm): the input is staged once into LDS and the nine taps are shifted reads of it, where the
# GEMM-lowered form writes and reads a nine times larger patch matrix; the weight gradient comes from its own kernel
# (ops.hip_learner.conv3_wgrad_cm) over the same unpadded operands.  False restores the GEMM-lowered products (A/B runs,
# tests).
DIRECT_CONV = True


def _mm(a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """``a @ b (+ bias)`` for 2-D or batched 3-D operands; fp32 CUDA operands go through :func:`ops.hip_learner.mm3`."""
    if FP32_GEMM_ON_BF16_CORES and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32:
        from ..ops import hip_learner

        out = hip_learner.mm3(a, b, bias)
        if out is not None:
            return out
    if bias is not None:
        return torch.addmm(bias, a, b)
    return torch.bmm(a, b) if a.dim() == 3 else a @ b


def _sum_parts(parts: torch.Tensor) -> torch.Tensor:
    """Sum of the split-K slices ``parts [S, ...]`` in the gradient dtype (one in-tree launch on the GPU)."""
    if parts.shape[0] == 1 and parts.dtype == _grad_dtype(parts):
        return parts[0]
    if parts.is_cuda and parts.is_contiguous() and parts.dtype in (torch.float32, torch.bfloat16):
        from ..ops import hip_learner

        if hip_learner.available():
            return hip_learner.sum_slices(parts)
    return parts.sum(0, dtype=_grad_dtype(parts))


def _flipped_filter(weight: torch.Tensor) -> torch.Tensor:
    """``[Cin, Cout * kh * kw]`` operand of a convolution's data gradient (flipped taps, channels swapped)."""
    cout, cin, kh, kw = weight.shape
    if weight.is_cuda and weight.dtype in (torch.float32, torch.bfloat16):
        from ..ops import hip_learner

        if hip_learner.available():
            return hip_learner.conv_wt(weight)
    return weight.flip(2, 3).permute(1, 0, 2, 3).reshape(cin, cout * kh * kw)


def _split_count(n: int, limit: int = 32, min_chunk: int = 512, step: int = 1) -> int:
    """Largest divisor of ``n`` that is at most ``limit`` and leaves chunks of at least ``min_chunk`` (whole multiples of
    ``step`` when such a divisor exists)."""
    fallback = 1
    for s in range(min(limit, max(1, n // min_chunk)), 1, -1):
        if n % s == 0:
            if (n // s) % step == 0:
                return s
            fallback = max(fallback, s)
    return fallback


def _linear_wgrad_slices(n: int, fp32_kernel: bool) -> int:
    """Slices of the ``n`` rows a linear layer's weight gradient reduces over.  The in-tree fp32 GEMM gets one workgroup
    per (slice, 128 x 128 output tile) and these gradients have one to three tiles, so it takes about 120 slices of whole
    32-row steps to fill 256 compute units (32 slices: 56 us per product at 30 720 rows)."""
    return _split_count(n, 128, 192, 32) if fp32_kernel else _split_count(n)


def _patch_matrix(x: torch.Tensor, kh: int, kw: int, ph: int, pw: int) -> torch.Tensor:
    """im2col for stride 1: ``[B, C, H, W]`` -> ``[C*kh*kw, B*oh*ow]`` (row order ``(c, kh, kw)``, column order ``(b, oh, ow)``).

    ``F.unfold`` launches one im2col kernel PER SAMPLE (1536 launches per learner step at batch 256); strided views of
    the padded input plus one gathering copy build the matrix in two launches.
    """
    b, c, h, w = x.shape
    if kh == 1 and kw == 1 and ph == 0 and pw == 0:
        return x.permute(1, 0, 2, 3).reshape(c, b * h * w)
    xp = torch.nn.functional.pad(x, (pw, pw, ph, ph))
    patches = xp.unfold(2, kh, 1).unfold(3, kw, 1)  # [B, C, oh, ow, kh, kw] (view)
    oh, ow = patches.shape[2], patches.shape[3]
    return patches.permute(1, 4, 5, 0, 2, 3).reshape(c * kh * kw, b * oh * ow)


class _ConvGemm(torch.autograd.Function):
    """Stride-1 convolution as ``W[Cout, K] @ patches[K, N]`` with all three GEMMs shaped for hipBLASLt.

    ``N = batch * pixels`` (30 720 at batch 256), ``K = Cin*kh*kw`` (1152 for the residual convolutions).

    * forward and data gradient are both "small matrix times patch matrix" products: the data gradient is the
      convolution of the output gradient with the flipped, transposed filter, so it gathers patches of the output
      gradient and reuses the forward GEMM shape.  (The transposed product ``W^T @ grad`` followed by a col2im
      scatter measured 164 us + 2 x 42 us per residual convolution against 16 us + 25 us this way.)
    * the weight gradient reduces over ``N`` into only ``Cout x K`` (9 tiles of 128 x 128): one plain GEMM keeps 9 of
      256 compute units busy, so ``N`` is cut into up to 32 chunks that run as ONE strided batched GEMM over views
      (no copies) and the partial products are added (164 us -> 28 us).
    """

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, ph: int, pw: int) -> torch.Tensor:  # noqa: ANN001
        low = _autocast_dtype(x)
        with torch.autocast(device_type=x.device.type, enabled=False):
            x, weight = _as_dtype(x, low), _as_dtype(weight, low)
            b, _, h, w = x.shape
            cout, _, kh, kw = weight.shape
            cols = _patch_matrix(x, kh, kw, ph, pw)
            ctx.save_for_backward(cols, weight)
            ctx.geom = (b, h, w, ph, pw)
            oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
            out = weight.reshape(cout, -1) @ cols  # [Cout, B*oh*ow]
            return out.view(cout, b, oh, ow).permute(1, 0, 2, 3).contiguous()

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        cols, weight = ctx.saved_tensors
        b, h, w, ph, pw = ctx.geom
        cout, cin, kh, kw = weight.shape
        d_x = d_w = None
        if ctx.needs_input_grad[1]:
            n = cols.shape[1]
            k = cols.shape[0]
            g2 = grad_out.permute(1, 0, 2, 3).reshape(cout, n)
            s = _split_count(n)
            parts = torch.bmm(g2.view(cout, s, n // s).transpose(0, 1), cols.view(k, s, n // s).permute(1, 2, 0))
            d_w = _sum_parts(parts).view(cout, cin, kh, kw)
        if ctx.needs_input_grad[0]:
            g_cols = _patch_matrix(grad_out, kh, kw, kh - 1 - ph, kw - 1 - pw)  # [Cout*kh*kw, B*H*W]
            w_t = weight.flip(2, 3).permute(1, 0, 2, 3).reshape(cin, cout * kh * kw)
            d_x = (w_t @ g_cols).view(cin, b, h, w).permute(1, 0, 2, 3).contiguous()
        return d_x, d_w, None, None


class _ConvGemmCM(torch.autograd.Function):
    """:class:`_ConvGemm` on channel-major activations: ``x [Cin, B, H, W]`` -> ``[Cout, B, oh, ow]``.

    With the batch folded into the column dimension the GEMM output ``[Cout, B*oh*ow]`` IS the next layer's input, and
    the incoming gradient is the GEMM operand as it stands: no NCHW <-> channel-major copies anywhere (the NCHW form
    spends about ten strided-copy launches per convolution and step on padding, gathering and permuting).  The patch
    matrices come from one HIP launch each (``ops.hip_learner.patch_matrix_cm``).
    """

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, ph: int, pw: int) -> torch.Tensor:  # noqa: ANN001
        from ..ops import hip_learner

        low = _autocast_dtype(x)
        with torch.autocast(device_type=x.device.type, enabled=False):
            x, weight = _as_dtype(x, low), _as_dtype(weight, low)
            cin, b, h, w = x.shape
            cout, _, kh, kw = weight.shape
            oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
            ctx.geom = (b, h, w, ph, pw)
            x = x.contiguous()
            w2 = weight.reshape(cout, -1)
            # fp32 3x3 / padding-1 layers: the patch matrix is generated inside the GEMM's loader and never stored
            ctx.implicit = bool(FP32_GEMM_ON_BF16_CORES and PATCH_FREE_CONV and (kh, kw, ph, pw) == (3, 3, 1, 1) and x.is_cuda
                                and hip_learner.conv3_supported(x, w2))
            if ctx.implicit:
                ctx.save_for_backward(x, weight)
                return hip_learner.conv3_mm(w2, x).view(cout, b, oh, ow)
            ctx.direct = bool(FP32_GEMM_ON_BF16_CORES and DIRECT_CONV and (kh, kw, ph, pw) == (3, 3, 1, 1) and weight.dtype == torch.float32
                              and hip_learner.conv3_cm_supported(x, cout))
            if ctx.direct:
                ctx.save_for_backward(x, weight)
                packed = getattr(weight, "_atri_packed", None)  # AlphaTriangleNet.prepack_filters: all layers in one launch
                ctx.packed_dgrad = packed.get(True) if packed else None
                planes = packed.get(False) if packed else None
                return hip_learner.conv3_cm(x, planes if planes is not None else hip_learner.conv3_pack(weight, False), cout)
            if kh == 1 and kw == 1:
                cols = x.reshape(cin, b * h * w)
            elif x.is_cuda:
                cols = hip_learner.patch_matrix_cm(x, kh, kw, ph, pw)
            else:  # host tensors (tests of the channel-major plumbing): strided-view gather
                cols = _patch_matrix(x.permute(1, 0, 2, 3), kh, kw, ph, pw)
            ctx.save_for_backward(cols, weight)
            return _mm(w2, cols).view(cout, b, oh, ow)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        cols, weight = ctx.saved_tensors  # with ctx.implicit / ctx.direct, `cols` is the layer input itself
        b, h, w, ph, pw = ctx.geom
        cout, cin, kh, kw = weight.shape
        g = grad_out.contiguous()
        d_x = d_w = None
        direct = getattr(ctx, "direct", False)
        wgrad_done = False
        if direct and ctx.needs_input_grad[1]:
            if hip_learner.conv3_wgrad_supported(g, cols):
                d_w = hip_learner.conv3_wgrad_cm(g, cols)
                wgrad_done = True
            else:
                cols = hip_learner.patch_matrix_cm(cols, kh, kw, ph, pw)  # the split-K GEMM's operand, built here instead of kept
        if ctx.needs_input_grad[1] and not wgrad_done:
            n = b * h * w if ctx.implicit else cols.shape[1]
            g2 = g.view(cout, n)
            s = _split_count(n)
            if ctx.implicit:
                parts = hip_learner.conv3_wgrad(g2, cols, n // s)
            else:
                k = cols.shape[0]
                parts = _mm(g2.view(cout, s, n // s).transpose(0, 1), cols.view(k, s, n // s).permute(1, 2, 0))
            d_w = _sum_parts(parts).view(cout, cin, kh, kw)
        if ctx.needs_input_grad[0]:
            if kh == 1 and kw == 1:
                d_x = _mm(weight.reshape(cout, cin).t(), g.view(cout, -1)).view(cin, b, h, w)
            elif direct and hip_learner.conv3_cm_supported(g, cin):
                planes = getattr(ctx, "packed_dgrad", None)
                d_x = hip_learner.conv3_cm(g, planes if planes is not None else hip_learner.conv3_pack(weight, True), cin)
            else:
                w_t = _flipped_filter(weight)
                if ctx.implicit and hip_learner.conv3_supported(g, w_t):
                    d_x = hip_learner.conv3_mm(w_t, g).view(cin, b, h, w)
                else:
                    if g.is_cuda:
                        g_cols = hip_learner.patch_matrix_cm(g, kh, kw, kh - 1 - ph, kw - 1 - pw)  # [Cout*kh*kw, B*H*W]
                    else:
                        g_cols = _patch_matrix(g.permute(1, 0, 2, 3), kh, kw, kh - 1 - ph, kw - 1 - pw)
                    d_x = _mm(w_t, g_cols).view(cin, b, h, w)
        return d_x, d_w, None, None


class _BnActCM(torch.autograd.Function):
    """Training-mode BatchNorm2d (+ skip connection) (+ activation) on a channel-major activation ``[C, B, H, W]``.

    One fused forward (two launches: statistics, apply) and one fused backward (two launches) through
    ``ops.hip_learner.bn_act_cm_*`` replace the library BatchNorm call, the skip add and the activation kernel of
    alphatriangle/nn/model.py:15-60 and their three backward kernels.  The activation's pre-image is recomputed in
    the backward pass from the saved input, so only per-channel statistics are stored besides what autograd keeps
    alive anyway.
    """

    @staticmethod
    def forward(ctx, x, residual, weight, bias, bn, act_id):  # noqa: ANN001
        from ..ops import hip_learner

        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        track = bn.track_running_stats and bn.running_mean is not None
        y, mean, rstd = hip_learner.bn_act_cm_forward(
            x, residual, w32, b32, float(bn.eps), momentum, bn.running_mean if track else None,
            bn.running_var if track else None, bn.num_batches_tracked if track else None, act_id,
        )
        ctx.save_for_backward(x, residual if residual is not None else x.new_empty(0), w32, b32, mean, rstd)
        ctx.has_res = residual is not None
        ctx.act_id = act_id
        return y

    @staticmethod
    def backward(ctx, grad_out):  # noqa: ANN001
        from ..ops import hip_learner

        x, res, w32, b32, mean, rstd = ctx.saved_tensors
        dx, dr, dg, db = hip_learner.bn_act_cm_backward(
            x.contiguous(), res.contiguous() if ctx.has_res else None, grad_out.to(x.dtype), w32, b32, mean, rstd, ctx.act_id
        )
        return dx, dr, dg, db, None, None


def _act_id(mod: nn.Module | None) -> int | None:
    """Activation code of the fused BatchNorm kernels (``None``: not a fusable activation)."""
    if mod is None or isinstance(mod, nn.Identity):
        return 0
    if getattr(mod, "approximate", "none") != "none":  # tanh-approximated GELU is not what the kernels compute
        return None
    return {nn.ReLU: 1, nn.GELU: 2, nn.SiLU: 3, nn.Tanh: 4, nn.Sigmoid: 5}.get(type(mod))


class GemmConv2d(nn.Conv2d):
    """``nn.Conv2d`` (same parameters, same ``state_dict`` keys) whose GPU path is im2col + GEMM.

    On gfx950 MIOpen has no tuned bf16 / fp32 convolution for these tiny 8x15 images and falls
    back to naive kernels (11 ms for one weight-gradient call at batch 256).  Lowering the
    convolution to a patch matrix ``[Cin*kh*kw, batch*pixels]`` + one large GEMM (:class:`_ConvGemm`)
    puts forward, data gradient and weight gradient on well-shaped hipBLASLt GEMMs, which is what
    the learner needs.  CPU tensors keep the stock implementation.
    """

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (
            not x.is_cuda
            or self.stride != (1, 1)
            or self.dilation != (1, 1)
            or self.groups != 1
            or isinstance(self.padding, str)
            or self.padding_mode != "zeros"
            or self.kernel_size[0] - 1 < self.padding[0]
            or self.kernel_size[1] - 1 < self.padding[1]
        ):
            return super().forward(x)
        return self.gemm_forward(x)

    def gemm_forward(self, x: torch.Tensor) -> torch.Tensor:
        out = _ConvGemm.apply(x, self.weight, int(self.padding[0]), int(self.padding[1]))
        if self.bias is not None:
            out = out + self.bias.to(out.dtype).view(1, -1, 1, 1)
        return out

    def cm_ok(self, batch: int, h: int, w: int) -> bool:
        """Whether :meth:`cm_forward` handles this layer for ``batch`` boards of ``h x w`` (bf16 autocast on a GPU)."""
        if self.stride != (1, 1) or self.dilation != (1, 1) or self.groups != 1 or isinstance(self.padding, str):
            return False
        kh, kw = self.kernel_size
        ph, pw = self.padding
        if self.padding_mode != "zeros" or kh - 1 < ph or kw - 1 < pw:
            return False
        if (kh, kw) == (1, 1):
            return (ph, pw) == (0, 0)
        oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
        # forward gathers the input, the data gradient gathers the output gradient: both column counts in eights
        return (kh, kw) in ((3, 3), (5, 5)) and oh > 0 and ow > 0 and (batch * oh * ow) % 8 == 0 and (batch * h * w) % 8 == 0

    def cm_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Channel-major form: ``[Cin, B, H, W]`` -> ``[Cout, B, oh, ow]`` (see :class:`_ConvGemmCM`)."""
        out = _ConvGemmCM.apply(x, self.weight, int(self.padding[0]), int(self.padding[1]))
        if self.bias is not None:
            out = out + self.bias.to(out.dtype).view(-1, 1, 1, 1)
        return out


class _SplitKLinear(torch.autograd.Function):
    """``y[N, out] = x[N, in] @ W[out, in]^T + b`` whose weight gradient is a split-K batched GEMM.

    The encoder's linear layers see ``N = tokens * batch`` = 30 720 rows at batch 256 while their weight gradients
    are at most 384 x 128: hipBLASLt runs that reduction on two to six workgroups (127 - 164 us per layer).  Cutting
    the rows into up to 32 contiguous chunks turns it into one batched GEMM with enough tiles to fill the chip; the
    bias gradient is a two-stage column sum (``ops.hip_learner.colsum``) without semaphores, so that a captured
    learner step holds no memset nodes.
    """

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x2: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:  # noqa: ANN001
        low = _autocast_dtype(x2)
        with torch.autocast(device_type=x2.device.type, enabled=False):
            x2, weight = _as_dtype(x2, low), _as_dtype(weight, low)
            ctx.save_for_backward(x2, weight)
            ctx.has_bias = bias is not None
            return _mm(x2, weight.t(), _as_dtype(bias, low) if bias is not None else None)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        x2, weight = ctx.saved_tensors
        g = grad_out.contiguous()
        n, out_f = g.shape
        s = _linear_wgrad_slices(n, bool(FP32_GEMM_ON_BF16_CORES and g.is_cuda and g.dtype == torch.float32))
        gs = g.view(s, n // s, out_f)
        d_x = _mm(g, weight) if ctx.needs_input_grad[0] else None
        d_w = None
        if ctx.needs_input_grad[1]:
            parts = _mm(gs.transpose(1, 2), x2.reshape(s, n // s, x2.shape[1]))
            d_w = _sum_parts(parts)
        d_b = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if g.is_cuda and g.dtype in (torch.bfloat16, torch.float32):
                from ..ops import hip_learner  # torch's multi-block column sum records a memset node under capture

                d_b = hip_learner.colsum(g)  # fp32, like the parameter
            else:
                d_b = gs.sum(1).sum(0)
        return d_x, d_w, d_b


def _linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """``F.linear`` over the flattened leading dimensions through :class:`_SplitKLinear`."""
    lead = x.shape[:-1]
    y = _SplitKLinear.apply(x.reshape(-1, x.shape[-1]), weight, bias)
    return y.view(*lead, weight.shape[0])


def _k_slices(k: int, limit: int = 128, min_chunk: int = 128) -> int:
    """Number of slices a ``k``-long dot product is cut into: the largest divisor of ``k`` up to ``limit`` whose chunks are
    multiples of four of at least ``min_chunk`` values."""
    for s in range(min(limit, max(1, k // min_chunk)), 1, -1):
        if k % s == 0 and (k // s) % 4 == 0:
            return s
    return 1


class _SharedFc(torch.autograd.Function):
    """First layer after the body: ``y = [body | other] @ W^T + b`` without the concatenation (fp32, GPU).

    ``body [B, Fb]`` (15 360 features at the default size) and ``other [B, Fo]`` (30) multiply their own column blocks of
    ``W [out, Fb + Fo]``.  With 256 rows and 128 outputs the product has two output tiles, so the long dot product is
    cut into slices that run as one batched launch of ``ops.hip_learner.mm3`` and are added (with the bias) by
    ``sum_slices``; the small block is one more slice.  The backward pass writes both column blocks of the weight
    gradient in place and produces no gradient for ``other`` (a network input).
    Reference layer: alphatriangle/nn/model.py:180-196.
    """

    @staticmethod
    def forward(ctx, body: torch.Tensor, other: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:  # noqa: ANN001
        from ..ops import hip_learner

        body, other = body.contiguous(), other.contiguous()
        n, fb = body.shape
        out_f = weight.shape[0]
        s = _k_slices(fb)
        chunk = fb // s
        parts = torch.empty(s + 1, n, out_f, dtype=torch.float32, device=body.device)
        w_body, w_other = weight[:, :fb], weight[:, fb:]
        ok = hip_learner.mm3(body.view(n, s, chunk).transpose(0, 1), w_body.view(out_f, s, chunk).permute(1, 2, 0), out=parts[:s])
        ok2 = hip_learner.mm3(other, w_other.t(), out=parts[s])
        if ok is None or ok2 is None:
            raise RuntimeError("_SharedFc: operands rejected by the fp32-grade GEMM")
        ctx.save_for_backward(body, other, weight)
        ctx.has_bias = bias is not None
        return hip_learner.sum_slices(parts, bias)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        body, other, weight = ctx.saved_tensors
        g = grad_out.contiguous()
        fb = body.shape[1]
        d_body = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            d_body = hip_learner.mm3(g, weight[:, :fb])
        if ctx.needs_input_grad[2]:
            d_w = torch.empty_like(weight)
            hip_learner.mm3(g.t(), body, out=d_w[:, :fb])
            hip_learner.mm3(g.t(), other, out=d_w[:, fb:])
        if ctx.has_bias and ctx.needs_input_grad[3]:
            d_b = hip_learner.colsum(g)
        return d_body, None, d_w, d_b


class _Bn1dAct(torch.autograd.Function):
    """Training-mode BatchNorm1d (+ activation) on ``[B, C]`` rows: one launch forward, one backward
    (``ops.hip_learner.bn1d_*``) in place of the library's statistics / apply / activation kernels and their backward
    twins (alphatriangle/nn/model.py:183-189, 201-207, 214-220 build these layers)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, act_id):  # noqa: ANN001
        from ..ops import hip_learner

        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        momentum = 0.1 if bn.momentum is None else float(bn.momentum)
        track = bn.track_running_stats and bn.running_mean is not None
        x = x.contiguous()
        y, mean, rstd = hip_learner.bn1d_forward(
            x, w32, b32, float(bn.eps), momentum, bn.running_mean if track else None, bn.running_var if track else None,
            bn.num_batches_tracked if track else None, act_id,
        )
        ctx.save_for_backward(x, w32, b32, mean, rstd)
        ctx.act_id = act_id
        return y

    @staticmethod
    def backward(ctx, grad_out):  # noqa: ANN001
        from ..ops import hip_learner

        x, w32, b32, mean, rstd = ctx.saved_tensors
        dx, dg, db = hip_learner.bn1d_backward(x, grad_out, w32, b32, mean, rstd, ctx.act_id)
        return dx, dg, db, None, None


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dimension through the learner's HIP kernels (``ops.hip_learner.layer_norm_*``).

    bf16 rows are normalised with fp32 statistics and come out as bf16, where torch's autocast policy would run the op
    in fp32 between two cast kernels.  fp32 rows under bf16 autocast (the residual stream of the encoder) come out as
    bf16 as well: every consumer of a LayerNorm output in this network is a GEMM that autocast would feed with exactly
    that cast, so the values are the same and the cast kernels (and their backward twins) are gone.
    """

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, batch_major_out: bool = False) -> torch.Tensor:  # noqa: ANN001
        """``batch_major_out`` (``x`` is ``[tokens, batch, D]``): the result comes out as ``[batch, tokens * D]`` - the
        permute + flatten in front of the shared layer done by the same pass (and undone by the backward kernel)."""
        from ..ops import hip_learner

        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        out_dtype = x2.dtype
        if x2.dtype == torch.float32 and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            out_dtype = torch.bfloat16
        perm = int(x.shape[1]) if batch_major_out else 0
        y, mean, rstd = hip_learner.layer_norm_forward(x2, w32, b32, eps, out_dtype, perm)
        ctx.save_for_backward(x2, w32, mean, rstd)
        ctx.shape = x.shape
        ctx.perm = perm
        return y.view(x.shape[1], -1) if batch_major_out else y.view(x.shape)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        x2, w32, mean, rstd = ctx.saved_tensors
        dy2 = grad_out.reshape(x2.shape)  # with ctx.perm: batch-major rows, which the kernel reads through the same permutation
        if dy2.dtype != torch.bfloat16 or x2.dtype != torch.float32:  # bf16 gradients of fp32 rows are read as they are
            dy2 = dy2.to(x2.dtype)
        dy2 = dy2.contiguous()
        dx, dg, db = hip_learner.layer_norm_backward(x2, dy2, w32, mean, rstd, ctx.perm)
        return dx.view(ctx.shape), dg, db, None, None


class _NormFork(torch.autograd.Function):
    """``(x, LayerNorm(x))``: the fork of a pre-norm residual branch as ONE autograd node.

    The residual connection and the normalised branch both start at ``x``; as two consumers of one tensor their
    gradients would be added by a kernel of autograd's own (four per step for two encoder layers).  Here the first output
    aliases ``x``, so ``x`` has a single consumer, and the backward kernel adds the gradient that arrives through the
    alias to the LayerNorm gradient while it writes it (``ln_bwd_kernel``'s ``skip`` operand)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float):  # noqa: ANN001, ANN205
        from ..ops import hip_learner

        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w32, b32 = weight.float().contiguous(), bias.float().contiguous()
        out_dtype = x2.dtype
        if x2.dtype == torch.float32 and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            out_dtype = torch.bfloat16
        y, mean, rstd = hip_learner.layer_norm_forward(x2, w32, b32, eps, out_dtype)
        ctx.save_for_backward(x2, w32, mean, rstd)
        ctx.shape = x.shape
        return x.view_as(x), y.view(x.shape)

    @staticmethod
    def backward(ctx, g_skip, g_y):  # noqa: ANN001
        from ..ops import hip_learner

        x2, w32, mean, rstd = ctx.saved_tensors
        if g_y is None:
            return g_skip, None, None, None
        dy2 = g_y.reshape(x2.shape)
        if dy2.dtype != torch.bfloat16 or x2.dtype != torch.float32:
            dy2 = dy2.to(x2.dtype)
        skip = g_skip.reshape(x2.shape).to(x2.dtype).contiguous() if g_skip is not None else None
        dx, dg, db = hip_learner.layer_norm_backward(x2, dy2.contiguous(), w32, mean, rstd, 0, skip)
        return dx.view(ctx.shape), dg, db, None


class _TokensPe(torch.autograd.Function):
    """Channel-major body output ``[D, B, H, W]`` -> encoder tokens ``[H * W, B, D]`` with the positional table added and
    its dropout applied, one launch each way (``ops.hip_learner.tokens_pe_*``) instead of a permuting copy, an add and
    ``nn.Dropout`` (alphatriangle/nn/model.py:268-283).  The table is a buffer and takes no gradient."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, pe: torch.Tensor, p_drop: float) -> torch.Tensor:  # noqa: ANN001
        from ..ops import hip_learner

        tokens, seed = hip_learner.tokens_pe_forward(x, pe, p_drop)
        ctx.save_for_backward(seed if seed is not None else tokens.new_empty(0))
        ctx.p_drop, ctx.x_shape, ctx.x_dtype = p_drop, tuple(x.shape), x.dtype
        return tokens

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        (seed,) = ctx.saved_tensors
        return hip_learner.tokens_pe_backward(grad_out, seed if ctx.p_drop > 0 else None, ctx.p_drop, ctx.x_shape, ctx.x_dtype), None, None


class FastLayerNorm(nn.LayerNorm):
    """``nn.LayerNorm`` (same parameters, same ``state_dict`` keys) whose GPU path is :class:`_LayerNormFn`."""

    def fast_ok(self, x: torch.Tensor) -> bool:
        """Whether the kernels take ``x`` (else the stock implementation does)."""
        return bool(
            x.is_cuda and self.elementwise_affine and self.bias is not None and len(self.normalized_shape) == 1
            and self.normalized_shape[0] in (64, 128, 256, 512) and x.dtype in (torch.bfloat16, torch.float32)
            and x.shape[-1] == self.normalized_shape[0]
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.fast_ok(x):
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)

    @classmethod
    def adopt(cls, norm: nn.LayerNorm) -> "FastLayerNorm":
        """Re-class an existing ``nn.LayerNorm`` (keeps its parameters)."""
        norm.__class__ = cls
        return norm  # type: ignore[return-value]


class _FusedAttention(torch.autograd.Function):
    """Multi-head self-attention over packed ``qkv [L, B, 3D]`` rows through ``ops.hip_learner.attention_*``.

    One launch forward and one backward per layer; scores, probabilities and the dropout mask never reach memory
    (the backward recomputes them from ``qkv``, one log-sum-exp per query and the dropout seed).
    """

    @staticmethod
    def forward(ctx, qkv: torch.Tensor, heads: int, p_drop: float) -> torch.Tensor:  # noqa: ANN001
        from ..ops import hip_learner

        qkv = qkv.contiguous()
        out, lse, seed = hip_learner.attention_forward(qkv, heads, p_drop)
        ctx.save_for_backward(qkv, lse, seed if seed is not None else lse.new_empty(0))
        ctx.heads, ctx.p_drop = heads, p_drop
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        qkv, lse, seed = ctx.saved_tensors
        dqkv = hip_learner.attention_backward(qkv, grad_out, lse, seed if ctx.p_drop > 0 else None, ctx.heads, ctx.p_drop)
        return dqkv, None, None


class _AddDropout(torch.autograd.Function):
    """``a + dropout(b, p)`` in one launch (``ops.hip_learner.add_dropout_*``): the residual adds of the encoder layer.
    The keep mask is a hash of a device-side seed and the element index, recomputed in the backward pass."""

    @staticmethod
    def forward(ctx, a: torch.Tensor, b: torch.Tensor, p_drop: float) -> torch.Tensor:  # noqa: ANN001
        from ..ops import hip_learner

        out, seed = hip_learner.add_dropout_forward(a, b, p_drop)
        ctx.save_for_backward(seed)
        ctx.p_drop, ctx.b_dtype = p_drop, b.dtype
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        (seed,) = ctx.saved_tensors
        g = grad_out.contiguous()
        return g, hip_learner.dropout_backward(g, seed, ctx.p_drop, ctx.b_dtype), None


class _ActDropout(torch.autograd.Function):
    """``dropout(act(h), p)`` in one launch forward and one backward (``ops.hip_learner.act_dropout_*``)."""

    @staticmethod
    def forward(ctx, h: torch.Tensor, p_drop: float, act_id: int) -> torch.Tensor:  # noqa: ANN001
        from ..ops import hip_learner

        h = h.contiguous()
        out, seed = hip_learner.act_dropout_forward(h, p_drop, act_id)
        ctx.save_for_backward(h, seed if seed is not None else h.new_empty(0))
        ctx.p_drop, ctx.act_id = p_drop, act_id
        return out

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # noqa: ANN001
        from ..ops import hip_learner

        h, seed = ctx.saved_tensors
        return hip_learner.act_dropout_backward(h, grad_out, seed if ctx.p_drop > 0 else None, ctx.p_drop, ctx.act_id), None, None


class EncoderLayer(nn.TransformerEncoderLayer):
    """``nn.TransformerEncoderLayer`` (same sub-modules, same ``state_dict`` keys) with a learner fast path.

    For the configuration the network uses (pre-LayerNorm, ``[tokens, batch, dim]`` input, no masks) the GPU forward
    is written out here so that every linear layer goes through :class:`_SplitKLinear` and attention is two batched
    GEMMs around a softmax; the arithmetic is that of the stock layer
    (``x + drop(attn(norm1(x)))``, then ``x + drop(ffn(norm2(x)))``).  Everything else falls back to the stock
    implementation.
    """

    def __init__(self, *args, **kwargs):  # noqa: ANN002, ANN003
        super().__init__(*args, **kwargs)
        FastLayerNorm.adopt(self.norm1)
        FastLayerNorm.adopt(self.norm2)

    def _fast_ok(self, src: torch.Tensor, src_mask, src_key_padding_mask, is_causal: bool) -> bool:  # noqa: ANN001
        attn = self.self_attn
        return (
            src.is_cuda and src.dim() == 3 and self.norm_first and src_mask is None and src_key_padding_mask is None
            and not is_causal and not attn.batch_first and attn._qkv_same_embed_dim and attn.in_proj_bias is not None
            and attn.bias_k is None and not attn.add_zero_attn
        )

    def forward(self, src, src_mask=None, src_key_padding_mask=None, is_causal=False):  # noqa: ANN001
        if not self._fast_ok(src, src_mask, src_key_padding_mask, is_causal):
            return super().forward(src, src_mask=src_mask, src_key_padding_mask=src_key_padding_mask, is_causal=is_causal)
        return self.fast_forward(src)

    def fast_forward(self, src: torch.Tensor) -> torch.Tensor:
        attn = self.self_attn
        seq, batch, dim = src.shape
        heads = attn.num_heads
        src, y = self._norm_fork(self.norm1, src)
        qkv = _linear_rows(y, attn.in_proj_weight, attn.in_proj_bias)  # [L, B, 3D]
        ctx = self._attention(qkv, seq, batch, dim, heads)
        x = self._add_dropout(src, _linear_rows(ctx, attn.out_proj.weight, attn.out_proj.bias), self.dropout1)
        x, y2 = self._norm_fork(self.norm2, x)
        hidden = self._act_dropout(_linear_rows(y2, self.linear1.weight, self.linear1.bias))
        return self._add_dropout(x, _linear_rows(hidden, self.linear2.weight, self.linear2.bias), self.dropout2)

    fused_norm_fork = True  # residual fork + LayerNorm as one autograd node (tests flip this)

    def _norm_fork(self, norm: nn.Module, t: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """``(t, norm(t))`` - as one node whose backward kernel adds the two gradients of ``t`` when the kernels take it."""
        if self.fused_norm_fork and isinstance(norm, FastLayerNorm) and norm.fast_ok(t) and torch.is_grad_enabled() and t.requires_grad:
            return _NormFork.apply(t, norm.weight, norm.bias, norm.eps)
        return t, norm(t)

    fused_dropout = True  # dropout fused with the residual adds / the FFN activation on the GPU (False: stock modules)

    def _add_dropout(self, a: torch.Tensor, b: torch.Tensor, drop: nn.Dropout) -> torch.Tensor:
        from ..ops import hip_learner

        if self.fused_dropout and self.training and 0.0 < drop.p < 1.0 and hip_learner.add_dropout_supported(a, b):
            return _AddDropout.apply(a, b, float(drop.p))
        return a + drop(b)

    def _act_dropout(self, h: torch.Tensor) -> torch.Tensor:
        from ..ops import hip_learner

        act_id = {torch.nn.functional.relu: 1, torch.nn.functional.gelu: 2}.get(self.activation)
        p = float(self.dropout.p) if self.training else 0.0
        if self.fused_dropout and act_id is not None and 0.0 <= p < 1.0 and hip_learner.act_dropout_supported(h):
            return _ActDropout.apply(h, p, act_id)
        return self.dropout(self.activation(h))


    fused_attention = True  # one-launch HIP attention where it applies; False keeps the batched-GEMM formulation

    def _attention(self, qkv: torch.Tensor, seq: int, batch: int, dim: int, heads: int) -> torch.Tensor:
        from ..ops import hip_learner

        attn = self.self_attn
        if self.fused_attention and dim == 32 * heads and hip_learner.attention_supported(qkv, heads):
            return _FusedAttention.apply(qkv, heads, float(attn.dropout) if self.training else 0.0)
        q, k, v = (t.reshape(seq, batch, heads, dim // heads).permute(1, 2, 0, 3) for t in qkv.chunk(3, dim=-1))
        # Written-out attention (two batched GEMMs around a softmax) instead of the flash kernel: the flash backward
        # zero-fills its accumulators through memset nodes, which keeps the learner step out of a replayable HIP graph.
        scores = (q * (dim // heads) ** -0.5) @ k.transpose(-2, -1)  # [B, H, L, L]
        with torch.autocast(device_type=qkv.device.type, enabled=False):
            probs = torch.softmax(scores, dim=-1)  # bf16 in / bf16 out under autocast, fp32 accumulation inside
        probs = torch.nn.functional.dropout(probs, attn.dropout, self.training)
        return (probs @ v).permute(2, 0, 1, 3).reshape(seq, batch, dim)


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
            layer = EncoderLayer(
                d_model=d,
                nhead=mc.TRANSFORMER_HEADS,
                dim_feedforward=mc.TRANSFORMER_FC_DIM,
                activation=mc.ACTIVATION_FUNCTION.lower(),
                batch_first=False,
                norm_first=True,
            )
            self.transformer_body = nn.TransformerEncoder(
                layer, num_layers=mc.TRANSFORMER_LAYERS, norm=FastLayerNorm(d), enable_nested_tensor=False
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

    channel_major = True  # learner fast path: convolution body on [C, B, H, W] activations (tests flip this)

    def _cm_plan_ok(self, grid_state: torch.Tensor) -> bool:
        """The channel-major convolution body needs a GPU (fp32, or bf16 autocast), the patch-matrix kernel and eligible layers."""
        if not (self.channel_major and grid_state.is_cuda and grid_state.dim() == 4):
            return False
        if torch.is_autocast_enabled("cuda"):
            if torch.get_autocast_dtype("cuda") != torch.bfloat16:
                return False
        elif grid_state.dtype != torch.float32:
            return False
        from ..ops import hip_learner

        if not hip_learner.available():
            return False
        b, _, h, w = grid_state.shape
        for m in list(self.conv_body.modules()) + list(self.res_body.modules()) + [self.input_proj]:
            if isinstance(m, nn.Conv2d):
                if not isinstance(m, GemmConv2d) or not m.cm_ok(b, h, w):
                    return False
                h = h + 2 * m.padding[0] - m.kernel_size[0] + 1
                w = w + 2 * m.padding[1] - m.kernel_size[1] + 1
        return True

    fused_batch_norm = True  # learner fast path: BatchNorm (+ skip) (+ activation) as one fused op (tests flip this)

    def _bn_fused_ok(self, bn: nn.Module, x: torch.Tensor, act: nn.Module | None) -> bool:
        from ..ops import hip_learner

        return bool(
            self.fused_batch_norm and isinstance(bn, nn.BatchNorm2d) and bn.training and bn.affine and _act_id(act) is not None
            and torch.is_grad_enabled() and hip_learner.bn_act_cm_supported(x)
        )

    def _cm_apply(self, mod: nn.Module, x: torch.Tensor) -> torch.Tensor:
        """``mod`` on a channel-major activation ``[C, B, H, W]`` (same parameters, buffers and statistics)."""
        if isinstance(mod, nn.Sequential):
            mods = list(mod)
            i = 0
            while i < len(mods):
                m = mods[i]
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                fuse_act = nxt is not None and not isinstance(nxt, (nn.Conv2d, nn.BatchNorm2d)) and _act_id(nxt) not in (None, 0)
                if isinstance(m, nn.BatchNorm2d) and self._bn_fused_ok(m, x, nxt if fuse_act else None):
                    # BatchNorm followed by an activation: one fused op
                    x = _BnActCM.apply(x, None, m.weight, m.bias, m, _act_id(nxt) if fuse_act else 0)
                    i += 2 if fuse_act else 1
                    continue
                x = self._cm_apply(m, x)
                i += 1
            return x
        if isinstance(mod, GemmConv2d):
            return mod.cm_forward(x)
        if isinstance(mod, nn.BatchNorm2d):
            if self._bn_fused_ok(mod, x, None):
                return _BnActCM.apply(x, None, mod.weight, mod.bias, mod, 0)
            # per-channel statistics over (B, H, W): the channel-major buffer viewed as ONE image of B*H rows
            c, b, h, w = x.shape
            return mod(x.reshape(1, c, b * h, w)).view(c, b, h, w)
        if isinstance(mod, ResidualBlock):
            y = self._cm_apply(mod.conv2, self._cm_apply(mod.conv1, x))
            if self._bn_fused_ok(mod.bn2, y, mod.activation):
                return _BnActCM.apply(y, x, mod.bn2.weight, mod.bn2.bias, mod.bn2, _act_id(mod.activation))  # bn2 + skip + activation
            return mod.activation(self._cm_apply(mod.bn2, y) + x)
        return mod(x)  # activations and Identity are layout-agnostic

    def body_channel_major(self, grid_state: torch.Tensor) -> torch.Tensor:
        """:meth:`body` with the convolution body on channel-major activations (same result, no layout copies)."""
        x = grid_state.permute(1, 0, 2, 3)  # [C, B, H, W]: a view for the usual single input channel
        x = self._cm_apply(self.res_body, self._cm_apply(self.conv_body, x))
        if self.uses_transformer and self.transformer_body is not None and self.pos_encoder is not None:
            x = self._cm_apply(self.input_proj, x)
            fused = self._encoder_io_fused(x)
            if fused:
                pe: torch.Tensor = self.pos_encoder.pe  # type: ignore[assignment]
                p = float(self.pos_encoder.dropout.p) if self.training else 0.0
                tokens = _TokensPe.apply(x.contiguous(), pe[: x.shape[2] * x.shape[3], 0, :], p)
            else:
                tokens = self.pos_encoder(x.flatten(2).permute(2, 1, 0).contiguous())  # [H*W, B, D]
            body = self.transformer_body
            norm = body.norm
            if fused and isinstance(norm, FastLayerNorm) and norm.fast_ok(tokens):
                for layer in body.layers:  # the stack written out, so that the final LayerNorm can write [B, H*W*D] itself
                    tokens = layer(tokens)
                return _LayerNormFn.apply(tokens, norm.weight, norm.bias, norm.eps, True)
            return body(tokens).permute(1, 0, 2).flatten(1)
        return x.permute(1, 0, 2, 3).reshape(x.size(1), -1)

    fused_encoder_io = True  # learner fast path: tokens + positional table + dropout in one launch, final LayerNorm writes the flattened rows

    def _encoder_io_fused(self, x: torch.Tensor) -> bool:
        from ..ops import hip_learner

        if not (self.fused_encoder_io and self.pos_encoder is not None and x.is_cuda and torch.is_grad_enabled()):
            return False
        p = float(self.pos_encoder.dropout.p) if self.training else 0.0
        pe: torch.Tensor = self.pos_encoder.pe  # type: ignore[assignment]
        hw = int(x.shape[2] * x.shape[3])
        if not (0.0 <= p < 1.0) or hw > pe.shape[0] or x.shape[0] != pe.shape[2]:
            return False
        return hip_learner.tokens_pe_supported(x.contiguous(), pe[:hw, 0, :])

    def body(self, grid_state: torch.Tensor) -> torch.Tensor:
        """Feature extractor up to (and excluding) the shared FC: returns ``[B, feat]``."""
        if self._cm_plan_ok(grid_state):
            return self.body_channel_major(grid_state)
        x = self.res_body(self.conv_body(grid_state))
        if self.uses_transformer and self.transformer_body is not None and self.pos_encoder is not None:
            x = self.input_proj(x)
            tokens = x.flatten(2).permute(2, 0, 1)  # [H*W, B, D]
            tokens = self.transformer_body(self.pos_encoder(tokens))
            return tokens.permute(1, 0, 2).flatten(1)
        return x.reshape(x.size(0), -1)

    def prepack_filters(self) -> None:
        """Pack the filters of every 3x3 convolution the board-per-workgroup kernel serves (forward form, and the flipped
        form where the data gradient runs on it too) in ONE launch and attach them to their weights; :class:`_ConvGemmCM`
        picks them up instead of packing layer by layer (eleven launches per step for the default network).  The packed
        copies are only valid until the weights change: the caller releases them (:meth:`release_packed_filters`) before
        the optimizer step."""
        from ..ops import hip_learner

        if not (FP32_GEMM_ON_BF16_CORES and DIRECT_CONV and hip_learner.available()):
            return
        h, w = int(self.env_config.ROWS), int(self.env_config.COLS)
        jobs: list[tuple[torch.Tensor, bool]] = []
        slots: list[tuple[torch.Tensor, bool]] = []
        for m in self.modules():
            if not (isinstance(m, GemmConv2d) and m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride == (1, 1) and m.groups == 1
                    and m.weight.is_cuda and m.weight.dtype == torch.float32):
                continue
            cout, cin = int(m.out_channels), int(m.in_channels)
            if not hip_learner.conv3_cm_sizes_ok(cin, cout, h, w):
                continue
            jobs.append((m.weight, False))
            slots.append((m.weight, False))
            if hip_learner.conv3_cm_sizes_ok(cout, cin, h, w):  # the data gradient is a cout -> cin convolution
                jobs.append((m.weight, True))
                slots.append((m.weight, True))
        if not jobs:
            return
        for (weight, dgrad), planes in zip(slots, hip_learner.conv3_pack_many(jobs), strict=True):
            packed = getattr(weight, "_atri_packed", None)
            if packed is None:
                packed = weight._atri_packed = {}
            packed[dgrad] = planes

    def release_packed_filters(self) -> None:
        for p in self.parameters():
            if hasattr(p, "_atri_packed"):
                del p._atri_packed

    fused_tail = True  # learner fast path: shared layer and heads on the in-tree GEMM / BatchNorm1d kernels (tests flip this)

    def _tail_ok(self, body: torch.Tensor, other: torch.Tensor) -> bool:
        if not (self.fused_tail and self.training and torch.is_grad_enabled() and body.is_cuda and other.is_cuda and body.dim() == 2):
            return False
        if torch.is_autocast_enabled("cuda"):
            if torch.get_autocast_dtype("cuda") != torch.bfloat16:
                return False
        elif body.dtype != torch.float32 or other.dtype != torch.float32:
            return False
        from ..ops import hip_learner

        return bool(hip_learner.available())

    def _mlp_apply(self, seq: nn.Sequential, x: torch.Tensor, other: torch.Tensor | None = None) -> torch.Tensor:
        """``seq`` (Linear [-> BatchNorm1d] -> activation, ..., [Linear]) on ``[B, F]`` rows through the learner's kernels.
        With ``other`` the first Linear multiplies ``[x | other]`` (the shared layer's input) without concatenating."""
        from ..ops import hip_learner

        mods = list(seq)
        i = 0
        if other is not None and other.shape[1] == 0:
            other = None
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, nn.Linear):
                if other is not None:
                    if x.dtype == torch.float32 and not torch.is_autocast_enabled("cuda") and FP32_GEMM_ON_BF16_CORES:
                        x = _SharedFc.apply(x, other, m.weight, m.bias)
                    else:
                        x = _linear_rows(torch.cat([x, other.to(x.dtype)], dim=1), m.weight, m.bias)
                    other = None
                else:
                    x = _linear_rows(x, m.weight, m.bias)
                i += 1
                continue
            if isinstance(m, nn.BatchNorm1d) and m.training and m.affine and hip_learner.bn1d_supported(x):
                fuse = nxt is not None and not isinstance(nxt, (nn.Linear, nn.BatchNorm1d)) and _act_id(nxt) not in (None, 0)
                x = _Bn1dAct.apply(x, m.weight, m.bias, m, _act_id(nxt) if fuse else 0)
                i += 2 if fuse else 1
                continue
            if _act_id(m) not in (None, 0) and hip_learner.act_dropout_supported(x):
                x = _ActDropout.apply(x, 0.0, _act_id(m))
                i += 1
                continue
            x = m(x)
            i += 1
        if other is not None:  # no Linear in the shared stack: the heads see the concatenation itself
            x = torch.cat([x, other.to(x.dtype)], dim=1)
        return x

    def forward(self, grid_state: torch.Tensor, other_features: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Returns ``(policy_logits [B, A], value_logits [B, atoms])``."""
        body = self.body(grid_state)
        if self._tail_ok(body, other_features):
            shared = self._mlp_apply(self.shared_fc, body, other_features)
            return self._mlp_apply(self.policy_head, shared), self._mlp_apply(self.value_head, shared)
        feat = torch.cat([body, other_features], dim=1)
        shared = self.shared_fc(feat)
        return self.policy_head(shared), self.value_head(shared)
