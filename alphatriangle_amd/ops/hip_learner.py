"""Python face of the learner-side HIP kernels (csrc/hip/learner_kernels.hip)."""

from __future__ import annotations

import ctypes as C

import torch

from .. import native
from . import hipbind as hb
from .losses import LossOutputs

_KIND = {"Adam": 0, "AdamW": 1, "SGD": 2}


def available() -> bool:
    return native.hip_available()


def require() -> None:
    hb.lib()


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plog, vlog, ptgt, ret, isw, v_min, v_max, wp, wv, we):  # noqa: ANN001
        B, A = plog.shape
        atoms = vlog.shape[1]
        pl = plog.detach().float().contiguous()
        vl = vlog.detach().float().contiguous()
        pt = ptgt.detach().float().contiguous()
        rt = ret.detach().float().contiguous()
        iw = isw.detach().float().contiguous()
        dev = pl.device
        pce = torch.empty(B, dtype=torch.float32, device=dev)
        vce = torch.empty(B, dtype=torch.float32, device=dev)
        ent = torch.empty(B, dtype=torch.float32, device=dev)
        dpl = torch.empty_like(pl)
        dvl = torch.empty_like(vl)
        hb.call(
            "atri_az_loss", hb.ptr(pl), hb.ptr(vl), hb.ptr(pt), hb.ptr(rt), hb.ptr(iw), hb.ptr(pce), hb.ptr(vce),
            hb.ptr(ent), hb.ptr(dpl), hb.ptr(dvl), hb.i32(B), hb.i32(A), hb.i32(atoms), hb.f32(v_min), hb.f32(v_max),
            hb.f32(wp), hb.f32(wv), hb.f32(we),
        )
        ctx.save_for_backward(dpl, dvl)
        ctx.in_dtypes = (plog.dtype, vlog.dtype)
        scalars = torch.empty(4, dtype=torch.float32, device=dev)  # total, policy, value, entropy: one launch for all four
        hb.call("atri_loss_reduce", hb.ptr(pce), hb.ptr(vce), hb.ptr(ent), hb.ptr(iw), hb.i32(B), hb.f32(wp), hb.f32(wv), hb.f32(we),
                hb.ptr(scalars))
        total, policy_loss, value_loss, entropy = scalars[0], scalars[1], scalars[2], scalars[3]
        ctx.mark_non_differentiable(policy_loss, value_loss, entropy, vce)
        ctx.set_materialize_grads(False)  # no zero tensors for the outputs nobody differentiates
        return total, policy_loss, value_loss, entropy, vce

    @staticmethod
    def backward(ctx, g_total, *_unused):  # noqa: ANN001
        dpl, dvl = ctx.saved_tensors
        pd, vd = ctx.in_dtypes
        if g_total is None:
            return (None,) * 10
        if pd == vd and pd in (torch.float32, torch.bfloat16) and g_total.numel() == 1:
            gp, gv = torch.empty_like(dpl, dtype=pd), torch.empty_like(dvl, dtype=vd)
            hb.call("atri_scale_pair", hb.ptr(dpl), hb.i64(dpl.numel()), hb.ptr(dvl), hb.i64(dvl.numel()),
                    hb.ptr(g_total.detach().float().contiguous()), hb.ptr(gp), hb.ptr(gv), hb.i32(1 if pd == torch.bfloat16 else 0))
            return gp, gv, None, None, None, None, None, None, None, None
        return (dpl * g_total).to(pd), (dvl * g_total).to(vd), None, None, None, None, None, None, None, None


def fused_loss(policy_logits, value_logits, policy_target, returns, is_weights, *, v_min, v_max, policy_weight,
               value_weight, entropy_weight) -> LossOutputs:  # noqa: ANN001
    total, pl, vl, ent, td = _FusedLoss.apply(
        policy_logits, value_logits, policy_target, returns, is_weights, float(v_min), float(v_max),
        float(policy_weight), float(value_weight), float(entropy_weight),
    )
    return LossOutputs(total, pl, vl, ent, td)


_partial_cache: dict[torch.device, torch.Tensor] = {}


def flat_optimizer_step(*, kind: str, param: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor | None,
                        norm_buf: torch.Tensor, step: int, lr: float, beta1: float, beta2: float, eps: float,
                        weight_decay: float, momentum: float, max_grad_norm: float | None, grad_scale: float) -> torch.Tensor:
    """Gradient-norm clip + optimizer update of the whole flat buffer; returns the (device) gradient norm."""
    dev = param.device
    partial = _partial_cache.get(dev)
    if partial is None:
        partial = torch.empty(1024, dtype=torch.float32, device=dev)
        _partial_cache[dev] = partial
    bc1 = 1.0 - beta1**step
    bc2 = 1.0 - beta2**step
    hb.call(
        "atri_flat_optimizer", hb.i32(_KIND[kind]), hb.ptr(param), hb.ptr(grad), hb.ptr(m), hb.ptr(v if v is not None else m),
        hb.i64(param.numel()), hb.ptr(norm_buf), hb.ptr(partial), hb.i32(1024), hb.f32(lr), hb.f32(beta1), hb.f32(beta2),
        hb.f32(eps), hb.f32(weight_decay), hb.f32(momentum), hb.f32(bc1), hb.f32(bc2**0.5),
        hb.f32(max_grad_norm if max_grad_norm is not None else 0.0), hb.f32(grad_scale), hb.i32(1 if step == 1 else 0),
    )
    return norm_buf[0]


def colsum(g: torch.Tensor) -> torch.Tensor:
    """fp32 column sums of a contiguous ``[N, F]`` bf16 / fp32 matrix (two kernels, no atomics, graph-capturable)."""
    if g.dim() != 2 or not g.is_contiguous() or g.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError(f"colsum needs a contiguous 2-D bf16/fp32 matrix, got {tuple(g.shape)} {g.dtype}")
    n, f = g.shape
    rows_per_block = 128 if n >= 8192 else 64  # 240 row blocks at 30 720 rows: enough workgroups to hide the load latency
    blocks = (n + rows_per_block - 1) // rows_per_block
    partial = torch.empty(blocks * f, dtype=torch.float32, device=g.device)
    out = torch.empty(f, dtype=torch.float32, device=g.device)
    hb.call("atri_colsum", hb.ptr(g), hb.i32(1 if g.dtype == torch.bfloat16 else 0), hb.ptr(partial), hb.ptr(out),
            hb.i32(n), hb.i32(f), hb.i32(rows_per_block))
    return out


LN_WIDTHS = (64, 128, 256, 512)


def layer_norm_forward(x2: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
                       out_dtype: torch.dtype | None = None, perm_batch: int = 0) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``y, mean, rstd`` of LayerNorm over the last dimension of a contiguous ``[rows, D]`` bf16 / fp32 matrix.

    ``out_dtype=torch.bfloat16`` with fp32 rows writes the normalised rows as bf16 (what an autocast GEMM would cast them
    to anyway).  ``perm_batch = B > 0``: the rows are token-major ``[tokens][B]`` and ``y`` comes out batch-major
    ``[B][tokens]`` (the flattening in front of the shared layer, done by the same pass).
    """
    rows, d = x2.shape
    out_dtype = x2.dtype if out_dtype is None else out_dtype
    if x2.dtype == torch.bfloat16 and out_dtype != torch.bfloat16:
        raise ValueError("bf16 rows are normalised to bf16")
    y = torch.empty(rows, d, dtype=out_dtype, device=x2.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2.device)
    hb.call("atri_ln_fwd_perm", hb.ptr(x2), hb.i32(1 if x2.dtype == torch.bfloat16 else 0), hb.ptr(weight), hb.ptr(bias), hb.ptr(y),
            hb.i32(1 if out_dtype == torch.bfloat16 else 0), hb.ptr(mean), hb.ptr(rstd), hb.i32(rows), hb.i32(d), hb.f32(eps),
            hb.i32(perm_batch))
    return y, mean, rstd


def layer_norm_backward(x2: torch.Tensor, dy2: torch.Tensor, weight: torch.Tensor, mean: torch.Tensor,
                        rstd: torch.Tensor, perm_batch: int = 0, skip: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``dx [rows, D]`` (dtype of x), ``dgamma [D]``, ``dbeta [D]`` (fp32); ``dy2`` in the dtype of x or, for fp32 rows, bf16."""
    rows, d = x2.shape
    rows_per_block = max(32, (rows + 511) // 512 + 31 & ~31)  # about 500 workgroups, whole 32-row passes
    blocks = (rows + rows_per_block - 1) // rows_per_block
    dx = torch.empty_like(x2)
    partial = torch.empty(blocks * 2 * d, dtype=torch.float32, device=x2.device)
    dgb = torch.empty(2 * d, dtype=torch.float32, device=x2.device)
    hb.call("atri_ln_bwd_perm", hb.ptr(x2), hb.ptr(dy2), hb.i32(1 if x2.dtype == torch.bfloat16 else 0),
            hb.i32(1 if dy2.dtype == torch.bfloat16 else 0), hb.ptr(weight), hb.ptr(mean), hb.ptr(rstd), hb.ptr(dx), hb.ptr(partial),
            hb.ptr(dgb), hb.ptr(dgb[d:]), hb.i32(rows), hb.i32(d), hb.i32(rows_per_block), hb.i32(perm_batch), hb.ptr(skip))
    return dx, dgb[:d], dgb[d:]


# ---------------------------------------------------------------- dropout fused with its neighbours
_seed_pool: dict[int, list] = {}  # device index -> [int64 seeds of the running step, next unused]
SEEDS_PER_STEP = 64


def _dev_index(device: torch.device | str | int) -> int:
    d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
    return d.index if d.index is not None else torch.cuda.current_device()


def begin_step_seeds(device: torch.device | str | int) -> None:
    """Draw the dropout seeds of one forward pass with ONE generator call; every dropout site then takes its own element
    (:func:`_drop_seed`) instead of launching a generator kernel of its own (eight per step for the default network)."""
    _seed_pool[_dev_index(device)] = [torch.randint(0, 2**62, (SEEDS_PER_STEP,), dtype=torch.int64, device=device), 0]


def end_step_seeds(device: torch.device | str | int) -> None:
    _seed_pool.pop(_dev_index(device), None)


def _drop_seed(device: torch.device) -> torch.Tensor:
    """One int64 on the device from torch's generator: ``torch.manual_seed`` governs it, and a captured step draws a
    new one on every replay.  Inside :func:`begin_step_seeds` / :func:`end_step_seeds` it is a view of that step's draw."""
    ent = _seed_pool.get(_dev_index(device))
    if ent is not None and ent[1] < ent[0].numel():
        i = ent[1]
        ent[1] = i + 1
        return ent[0][i : i + 1]
    return torch.randint(0, 2**62, (1,), dtype=torch.int64, device=device)


def add_dropout_supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    ok = (torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)
    return bool(a.is_cuda and b.is_cuda and a.shape == b.shape and (a.dtype, b.dtype) in ok and a.numel() % 8 == 0 and a.numel() > 0)


def add_dropout_forward(a: torch.Tensor, b: torch.Tensor, p_drop: float) -> tuple[torch.Tensor, torch.Tensor]:
    """``a + dropout(b, p)`` in the dtype of ``a`` and the seed of its keep mask (``0 < p < 1``)."""
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    seed = _drop_seed(a.device)
    hb.call("atri_add_dropout_fwd", hb.ptr(a), hb.i32(1 if a.dtype == torch.bfloat16 else 0), hb.ptr(b),
            hb.i32(1 if b.dtype == torch.bfloat16 else 0), hb.ptr(out), hb.ptr(seed), hb.i64(a.numel()), hb.f32(p_drop))
    return out, seed


def dropout_backward(g: torch.Tensor, seed: torch.Tensor, p_drop: float, dtype: torch.dtype) -> torch.Tensor:
    """Gradient of the dropped branch of :func:`add_dropout_forward` (``dtype``: that branch's dtype)."""
    g = g.contiguous()
    db = torch.empty(g.shape, dtype=dtype, device=g.device)
    hb.call("atri_dropout_bwd", hb.ptr(g), hb.i32(1 if g.dtype == torch.bfloat16 else 0), hb.ptr(db),
            hb.i32(1 if dtype == torch.bfloat16 else 0), hb.ptr(seed), hb.i64(g.numel()), hb.f32(p_drop))
    return db


def act_dropout_supported(h: torch.Tensor) -> bool:
    return bool(h.is_cuda and h.dtype in (torch.float32, torch.bfloat16) and h.numel() % 8 == 0 and h.numel() > 0)


def act_dropout_forward(h: torch.Tensor, p_drop: float, act_id: int) -> tuple[torch.Tensor, torch.Tensor | None]:
    """``dropout(act(h), p)`` (``p`` may be 0) and the seed of the keep mask."""
    h = h.contiguous()
    out = torch.empty_like(h)
    seed = _drop_seed(h.device) if p_drop > 0 else None
    hb.call("atri_act_dropout", hb.ptr(h), hb.ptr(None), hb.ptr(out), hb.i32(1 if h.dtype == torch.bfloat16 else 0), hb.ptr(seed),
            hb.i64(h.numel()), hb.f32(p_drop), hb.i32(act_id))
    return out, seed


def act_dropout_backward(h: torch.Tensor, g: torch.Tensor, seed: torch.Tensor | None, p_drop: float, act_id: int) -> torch.Tensor:
    g = g.to(h.dtype).contiguous()
    dh = torch.empty_like(h)
    hb.call("atri_act_dropout", hb.ptr(h), hb.ptr(g), hb.ptr(dh), hb.i32(1 if h.dtype == torch.bfloat16 else 0), hb.ptr(seed),
            hb.i64(h.numel()), hb.f32(p_drop), hb.i32(act_id))
    return dh


# ---------------------------------------------------------------- prioritized replay tree
def per_set_leaves(tree: torch.Tensor, P: int, idx: torch.Tensor, count: torch.Tensor | None, count_max: int, prio: torch.Tensor) -> None:
    hb.call("atri_per_set_leaves", hb.ptr(tree), hb.i64(P), hb.ptr(idx), hb.ptr(count), hb.i32(count_max), hb.ptr(prio))


def per_update_paths(tree: torch.Tensor, P: int, idx: torch.Tensor, count: torch.Tensor | None, count_max: int) -> None:
    hb.call("atri_per_update_paths", hb.ptr(tree), hb.i64(P), hb.ptr(idx), hb.ptr(count), hb.i32(count_max))


def per_rebuild(tree: torch.Tensor, P: int) -> None:
    hb.call("atri_per_rebuild", hb.ptr(tree), hb.i64(P))


def per_sample(tree: torch.Tensor, P: int, size: int, batch: int, rnd: torch.Tensor, beta: float, out_idx: torch.Tensor, out_w: torch.Tensor) -> None:
    if batch <= 1024:  # one workgroup, one launch
        hb.call("atri_per_sample", hb.ptr(tree), hb.i64(P), hb.i32(size), hb.i32(batch), hb.ptr(rnd), hb.f32(beta), hb.ptr(out_idx), hb.ptr(out_w))
        return
    scratch = torch.empty((batch + 255) // 256, dtype=torch.float32, device=tree.device)
    hb.call("atri_per_sample_large", hb.ptr(tree), hb.i64(P), hb.i32(size), hb.i32(batch), hb.ptr(rnd), hb.f32(beta), hb.ptr(out_idx),
            hb.ptr(out_w), hb.ptr(scratch))


def per_priorities(tree: torch.Tensor, P: int, idx: torch.Tensor, err: torch.Tensor, alpha: float, eps: float, max_prio: torch.Tensor) -> None:
    hb.call("atri_per_priorities", hb.ptr(tree), hb.i64(P), hb.ptr(idx), hb.ptr(err), hb.i32(idx.numel()), hb.f32(alpha), hb.f32(eps), hb.ptr(max_prio))


# ---------------------------------------------------------------- fused BatchNorm (+ skip) (+ activation), channel-major
ACT_IDS = {"Identity": 0, "ReLU": 1, "GELU": 2, "SiLU": 3, "Tanh": 4, "Sigmoid": 5}


def bn_act_cm_supported(x: torch.Tensor) -> bool:
    return bool(x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() >= 2 and (x[0].numel() % 8) == 0
                and x.shape[0] <= 65535 and available())


def _bn_partial(c: int, n: int, device: torch.device) -> torch.Tensor:
    fn = getattr(hb.lib(), "atri_bn_cm_partial_floats")
    fn.restype = C.c_int
    return torch.empty(int(fn(C.c_int(c), C.c_int(n))), dtype=torch.float32, device=device)


def bn_act_cm_forward(x: torch.Tensor, residual: torch.Tensor | None, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                      momentum: float, running_mean: torch.Tensor | None, running_var: torch.Tensor | None,
                      num_batches: torch.Tensor | None, act: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Training-mode BatchNorm over channel-major ``x [C, ...]`` (+ ``residual``) (+ activation) in two launches.

    Returns ``(y, batch_mean, batch_rstd)``; the running statistics and the batch counter are updated in place.
    """
    x = x.contiguous()
    c = int(x.shape[0])
    n = x.numel() // c
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    r = residual.contiguous() if residual is not None else None
    hb.call("atri_bn_cm_fwd", hb.ptr(x), hb.ptr(r), hb.i32(1 if x.dtype == torch.bfloat16 else 0), hb.ptr(gamma), hb.ptr(beta),
            hb.ptr(_bn_partial(c, n, x.device)), hb.i32(c), hb.i32(n), hb.f32(eps), hb.f32(momentum), hb.ptr(running_mean),
            hb.ptr(running_var), hb.ptr(num_batches), hb.ptr(y), hb.ptr(mean), hb.ptr(rstd), hb.i32(act))
    return y, mean, rstd


def bn_act_cm_backward(x: torch.Tensor, residual: torch.Tensor | None, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                       mean: torch.Tensor, rstd: torch.Tensor, act: int) -> tuple[torch.Tensor, torch.Tensor | None, torch.Tensor, torch.Tensor]:
    """Backward of :func:`bn_act_cm_forward`: ``(dx, d_residual, dgamma, dbeta)`` in two launches."""
    c = int(x.shape[0])
    n = x.numel() // c
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    dr = torch.empty_like(x) if residual is not None else None
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    hb.call("atri_bn_cm_bwd", hb.ptr(x), hb.ptr(residual), hb.ptr(dy), hb.i32(1 if x.dtype == torch.bfloat16 else 0), hb.ptr(gamma),
            hb.ptr(beta), hb.ptr(mean), hb.ptr(rstd), hb.ptr(_bn_partial(c, n, x.device)), hb.i32(c), hb.i32(n), hb.i32(act),
            hb.ptr(dx), hb.ptr(dr), hb.ptr(dgamma), hb.ptr(dbeta))
    return dx, dr, dgamma, dbeta


# ---------------------------------------------------------------- fp32 GEMM on the bf16 matrix cores (three passes)
def _gemm3_layout(t: torch.Tensor) -> tuple[int, int] | None:
    """(row stride, column stride) of the last two dimensions when one of them is the contiguous one."""
    sr, sc = int(t.stride(-2)), int(t.stride(-1))
    if t.shape[-1] == 1:
        sc = 1 if sr != 1 else sc
    if t.shape[-2] == 1:
        sr = 1 if sc != 1 else sr
    if sc == 1 and sr >= t.shape[-1]:
        return sr, 1
    if sr == 1 and sc >= t.shape[-2]:
        return 1, sc
    return None


def mm3(a: torch.Tensor, b: torch.Tensor, bias: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor | None:
    """``a @ b (+ bias)`` for fp32 CUDA operands through ``gemm3_kernel`` (fp32-grade: three bf16 MFMA passes over (hi, lo)
    splits), or ``None`` when the operands do not fit it (the caller then uses the library GEMM).

    ``a`` is ``[M, K]`` or ``[S, M, K]``, ``b`` is ``[K, N]`` or ``[S, K, N]`` (batched for split-K products); either
    dimension of each matrix may be the contiguous one, so transposed views and column slices need no copy, and pitches
    or extents that are not multiples of four floats only narrow the kernel's vector accesses.  ``out`` (fp32, last
    dimension contiguous, e.g. a column block of a larger matrix) receives the product instead of a new tensor.
    """
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == b.dim() and a.dim() in (2, 3) and available()):
        return None
    batch = int(a.shape[0]) if a.dim() == 3 else 1
    if a.dim() == 3 and (b.shape[0] != batch or batch > 65535):
        return None
    m, k = int(a.shape[-2]), int(a.shape[-1])
    n = int(b.shape[-1])
    if int(b.shape[-2]) != k or min(m, n, k) < 1:
        return None
    la, lb = _gemm3_layout(a), _gemm3_layout(b)
    if la is None or lb is None:
        return None
    a_bs = int(a.stride(0)) if a.dim() == 3 else 0
    b_bs = int(b.stride(0)) if b.dim() == 3 else 0
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() != n or not bias.is_contiguous()):
        return None
    if out is None:
        out = torch.empty((batch, m, n) if a.dim() == 3 else (m, n), dtype=torch.float32, device=a.device)
    elif (out.dtype != torch.float32 or out.device != a.device or tuple(out.shape) != ((batch, m, n) if a.dim() == 3 else (m, n))
          or out.stride(-1) != 1 or out.stride(-2) < n):
        raise ValueError("mm3: `out` must be an fp32 tensor of the product's shape with a contiguous last dimension")
    ldc = int(out.stride(-2))
    c_bs = int(out.stride(0)) if out.dim() == 3 else m * n
    hb.call("atri_gemm3", hb.ptr(a), hb.ptr(b), hb.ptr(out), hb.ptr(bias), hb.i32(m), hb.i32(n), hb.i32(k), hb.i64(la[0]), hb.i64(la[1]),
            hb.i64(lb[0]), hb.i64(lb[1]), hb.i64(ldc), hb.i32(batch), hb.i64(a_bs), hb.i64(b_bs), hb.i64(c_bs))
    return out


def sum_slices(parts: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 sum over the first dimension of contiguous ``parts [S, ...]`` (fp32 or bf16) in one launch, plus ``bias`` along the
    last dimension: the reduction half of a split-K product (``sum_slices_kernel``)."""
    if not parts.is_contiguous() or parts.dtype not in (torch.float32, torch.bfloat16) or parts.dim() < 2:
        raise ValueError("sum_slices needs a contiguous fp32 / bf16 tensor [S, ...]")
    s = int(parts.shape[0])
    out = torch.empty(parts.shape[1:], dtype=torch.float32, device=parts.device)
    period = 0
    if bias is not None:
        if bias.dtype != torch.float32 or not bias.is_contiguous() or bias.numel() != parts.shape[-1]:
            raise ValueError("sum_slices: bias must be a contiguous fp32 vector over the last dimension")
        period = int(bias.numel())
    hb.call("atri_sum_slices", hb.ptr(parts), hb.i32(1 if parts.dtype == torch.bfloat16 else 0), hb.ptr(out), hb.i32(s), hb.i64(out.numel()),
            hb.ptr(bias), hb.i32(period))
    return out


def bn1d_supported(x: torch.Tensor) -> bool:
    return bool(x.is_cuda and x.dim() == 2 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[0] >= 1 and available())


def bn1d_forward(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, momentum: float,
                 running_mean: torch.Tensor | None, running_var: torch.Tensor | None, num_batches: torch.Tensor | None,
                 act: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Training-mode BatchNorm1d (+ activation) over the rows of ``x [B, C]`` in one launch: ``(y, batch_mean, batch_rstd)``;
    the running statistics and the batch counter are updated in place (unbiased variance, like ``torch.nn.BatchNorm1d``)."""
    x = x.contiguous()
    b, c = (int(v) for v in x.shape)
    y = torch.empty_like(x)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    rstd = torch.empty(c, dtype=torch.float32, device=x.device)
    hb.call("atri_bn1d_fwd", hb.ptr(x), hb.i32(1 if x.dtype == torch.bfloat16 else 0), hb.ptr(gamma), hb.ptr(beta), hb.i32(b), hb.i32(c),
            hb.f32(eps), hb.f32(momentum), hb.ptr(running_mean), hb.ptr(running_var), hb.ptr(num_batches), hb.ptr(y), hb.ptr(mean),
            hb.ptr(rstd), hb.i32(act))
    return y, mean, rstd


def bn1d_backward(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                  act: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``(dx, dgamma, dbeta)`` of :func:`bn1d_forward` in one launch."""
    b, c = (int(v) for v in x.shape)
    dy = dy.to(x.dtype).contiguous()
    dx = torch.empty_like(x)
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x.device)
    hb.call("atri_bn1d_bwd", hb.ptr(x), hb.ptr(dy), hb.i32(1 if x.dtype == torch.bfloat16 else 0), hb.ptr(gamma), hb.ptr(beta), hb.ptr(mean),
            hb.ptr(rstd), hb.i32(b), hb.i32(c), hb.i32(act), hb.ptr(dx), hb.ptr(dgamma), hb.ptr(dbeta))
    return dx, dgamma, dbeta


def conv3_cm_supported(x: torch.Tensor, cout: int) -> bool:
    """True when :func:`conv3_cm` convolves channel-major ``x [Cin, B, H, W]`` (fp32, contiguous) into ``cout`` channels."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and available()):
        return False
    cin, nb, h, w = (int(v) for v in x.shape)
    fn = getattr(hb.lib(), "atri_conv3_cm_supported")
    fn.restype = C.c_int
    return bool(nb >= 1 and cin * nb * h * w < 2**31 and fn(C.c_int(cin), C.c_int(int(cout)), C.c_int(h), C.c_int(w)))


def conv3_cm_sizes_ok(cin: int, cout: int, h: int, w: int) -> bool:
    """Whether :func:`conv3_cm` serves a ``cin -> cout`` convolution on ``h x w`` boards."""
    if not available():
        return False
    fn = getattr(hb.lib(), "atri_conv3_cm_supported")
    fn.restype = C.c_int
    return bool(fn(C.c_int(int(cin)), C.c_int(int(cout)), C.c_int(int(h)), C.c_int(int(w))))


def conv3_pack(weight: torch.Tensor, dgrad: bool) -> torch.Tensor:
    """The filter ``[Cout, Cin, 3, 3]`` (fp32) as the (hi, lo) bf16 planes :func:`conv3_cm` reads, ``[2, 9 * Cin * Cout]``:
    as it is for the forward product, flipped with the channels swapped for the data gradient.  One launch."""
    cout, cin, kh, kw = (int(v) for v in weight.shape)
    if (kh, kw) != (3, 3) or weight.dtype != torch.float32:
        raise ValueError("conv3_pack: needs an fp32 [Cout, Cin, 3, 3] filter")
    w = weight.contiguous()
    planes = torch.empty(2, 9 * cin * cout, dtype=torch.bfloat16, device=w.device)
    hb.call("atri_conv3_pack", hb.ptr(w), hb.ptr(planes[0]), hb.ptr(planes[1]), hb.i32(cout), hb.i32(cin), hb.i32(1 if dgrad else 0))
    return planes


def conv3_pack_many(jobs: list[tuple[torch.Tensor, bool]]) -> list[torch.Tensor]:
    """:func:`conv3_pack` for several ``(filter, dgrad)`` jobs in ONE launch (a learner step packs eleven: every direct
    convolution's forward form and, where the data gradient runs on the kernel too, its flipped form)."""
    if not jobs:
        return []
    if len(jobs) > 24:
        return conv3_pack_many(jobs[:24]) + conv3_pack_many(jobs[24:])
    dev = jobs[0][0].device
    total = sum(2 * 9 * int(w.shape[0]) * int(w.shape[1]) for w, _ in jobs)
    pool = torch.empty(total, dtype=torch.bfloat16, device=dev)
    outs, off = [], 0
    n = len(jobs)
    wp, hp, lp = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    co, ci, dg = (C.c_int * n)(), (C.c_int * n)(), (C.c_int * n)()
    keep = []
    for e, (w, dgrad) in enumerate(jobs):
        cout, cin, kh, kw = (int(v) for v in w.shape)
        if (kh, kw) != (3, 3) or w.dtype != torch.float32 or w.device != dev:
            raise ValueError("conv3_pack_many: needs fp32 [Cout, Cin, 3, 3] filters on one device")
        wc = w.contiguous()
        keep.append(wc)
        cnt = 9 * cin * cout
        planes = pool[off : off + 2 * cnt].view(2, cnt)
        off += 2 * cnt
        outs.append(planes)
        wp[e], hp[e], lp[e] = wc.data_ptr(), planes[0].data_ptr(), planes[1].data_ptr()
        co[e], ci[e], dg[e] = cout, cin, 1 if dgrad else 0
    hb.call("atri_conv3_pack_many", hb.i32(n), wp, hp, lp, co, ci, dg)
    return outs


def conv3_cm(x: torch.Tensor, planes: torch.Tensor, cout: int) -> torch.Tensor:
    """3x3, padding-1 convolution of channel-major ``x [Cin, B, H, W]`` with a filter packed by :func:`conv3_pack`:
    ``[cout, B, H, W]``, one workgroup per board, no patch matrix (``conv3_cm_kernel``)."""
    cin, nb, h, w = (int(v) for v in x.shape)
    if planes.numel() != 2 * 9 * cin * cout or not conv3_cm_supported(x, cout):
        raise ValueError("conv3_cm: unsupported input or filter size")
    out = torch.empty(cout, nb, h, w, dtype=torch.float32, device=x.device)
    hb.call("atri_conv3_cm", hb.ptr(x), hb.ptr(planes[0]), hb.ptr(planes[1]), hb.ptr(out), hb.i32(cin), hb.i32(cout), hb.i32(nb), hb.i32(h),
            hb.i32(w))
    return out


def conv3_wgrad_supported(g: torch.Tensor, x: torch.Tensor) -> bool:
    """True when :func:`conv3_wgrad_cm` handles channel-major ``g [Cout, B, H, W]`` and ``x [Cin, B, H, W]`` (fp32, contiguous)."""
    if not (g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.dim() == 4 and x.dim() == 4 and g.shape[1:] == x.shape[1:]
            and g.is_contiguous() and x.is_contiguous() and available()):
        return False
    fn = getattr(hb.lib(), "atri_conv3_wgrad_supported")
    fn.restype = C.c_int
    cin, nb, h, w = (int(v) for v in x.shape)
    return bool(nb >= 1 and max(g.numel(), x.numel()) < 2**31 and fn(C.c_int(cin), C.c_int(int(g.shape[0])), C.c_int(h), C.c_int(w)))


def conv3_wgrad_cm(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Weight gradient ``[Cout, Cin, 3, 3]`` of a 3x3, padding-1 convolution from the output gradient ``g [Cout, B, H, W]`` and the
    input ``x [Cin, B, H, W]`` (channel-major fp32) without a patch matrix: ``conv3_wgrad_kernel`` over slices of the boards
    plus one reduction launch."""
    if not conv3_wgrad_supported(g, x):
        raise ValueError("conv3_wgrad_cm: unsupported operands")
    cin, nb, h, w = (int(v) for v in x.shape)
    cout = int(g.shape[0])
    fn = getattr(hb.lib(), "atri_conv3_wgrad_slices")
    fn.restype = C.c_int
    slices = int(fn(C.c_int(cin), C.c_int(nb)))
    part = torch.empty(slices * 9 * cout * cin, dtype=torch.float32, device=x.device)
    dw = torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=x.device)
    hb.call("atri_conv3_wgrad", hb.ptr(g), hb.ptr(x), hb.ptr(part), hb.ptr(dw), hb.i32(cin), hb.i32(cout), hb.i32(nb), hb.i32(h), hb.i32(w))
    return dw


def tokens_pe_supported(x: torch.Tensor, pe: torch.Tensor) -> bool:
    """True when :func:`tokens_pe_forward` handles channel-major ``x [D, B, H, W]`` and the table ``pe [H * W, D]``."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and available()):
        return False
    d, b, h, w = (int(v) for v in x.shape)
    return bool(d % 32 == 0 and pe.dtype == torch.float32 and pe.is_contiguous() and tuple(pe.shape) == (h * w, d) and pe.device == x.device
                and b * h * w * d < 2**32 and pe.data_ptr() % 16 == 0)


def tokens_pe_forward(x: torch.Tensor, pe: torch.Tensor, p_drop: float) -> tuple[torch.Tensor, torch.Tensor | None]:
    """Encoder tokens ``[H * W, B, D]`` (fp32) = dropout(``x`` permuted + ``pe``) from the channel-major body output in one launch;
    returns them with the seed of the keep mask (``None`` without dropout)."""
    d, b, h, w = (int(v) for v in x.shape)
    tokens = torch.empty(h * w, b, d, dtype=torch.float32, device=x.device)
    seed = _drop_seed(x.device) if p_drop > 0 else None
    hb.call("atri_tokens_pe", hb.ptr(x), hb.i32(1 if x.dtype == torch.bfloat16 else 0), hb.ptr(pe), hb.ptr(None), hb.ptr(tokens), hb.ptr(None),
            hb.ptr(seed), hb.i32(d), hb.i32(b), hb.i32(h * w), hb.f32(p_drop))
    return tokens, seed


def tokens_pe_backward(g: torch.Tensor, seed: torch.Tensor | None, p_drop: float, shape: tuple[int, ...], dtype: torch.dtype) -> torch.Tensor:
    """Gradient of the channel-major input ``[D, B, H, W]`` (in ``dtype``) from the token gradient ``g [H * W, B, D]`` (fp32)."""
    d, b, h, w = (int(v) for v in shape)
    g = g.float().contiguous()
    dx = torch.empty(d, b, h, w, dtype=dtype, device=g.device)
    hb.call("atri_tokens_pe", hb.ptr(None), hb.i32(1 if dtype == torch.bfloat16 else 0), hb.ptr(None), hb.ptr(g), hb.ptr(None), hb.ptr(dx),
            hb.ptr(seed), hb.i32(d), hb.i32(b), hb.i32(h * w), hb.f32(p_drop))
    return dx


def conv_wt(weight: torch.Tensor) -> torch.Tensor:
    """``[Cin, Cout * kh * kw]``: the spatially flipped filter with input and output channels swapped - the A operand of a
    convolution's data-gradient GEMM - in one launch (``weight.flip(2, 3).permute(1, 0, 2, 3).reshape(...)`` takes two)."""
    cout, cin, kh, kw = (int(v) for v in weight.shape)
    w = weight.contiguous()
    out = torch.empty(cin, cout * kh * kw, dtype=w.dtype, device=w.device)
    hb.call("atri_conv_wt", hb.ptr(w), hb.ptr(out), hb.i32(1 if w.dtype == torch.bfloat16 else 0), hb.i32(cout), hb.i32(cin), hb.i32(kh * kw))
    return out


def conv3_supported(x: torch.Tensor, a: torch.Tensor) -> bool:
    """True when the 3x3 / padding-1 products of ``a [M, K]`` with the patch matrix of channel-major ``x [Cin, B, H, W]`` can
    run without materialising the patch matrix (:func:`conv3_mm`, :func:`conv3_wgrad`)."""
    if not (x.is_cuda and x.dtype == torch.float32 and a.dtype == torch.float32 and x.dim() == 4 and a.dim() == 2 and available()):
        return False
    cin, nb, h, w = (int(v) for v in x.shape)
    pos = nb * h * w
    return bool(x.is_contiguous() and cin % 4 == 0 and pos % 4 == 0 and pos < 2**24 and a.stride(1) == 1 and a.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and a.data_ptr() % 16 == 0)


def conv3_mm(a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``a [M, Cin * 9] @ patches(x)`` -> ``[M, B * H * W]`` for a 3x3, padding-1 convolution over channel-major ``x [Cin, B, H, W]``;
    the patch matrix is generated inside the GEMM's operand loader (``gemm3_kernel`` B_MODE 2)."""
    cin, nb, h, w = (int(v) for v in x.shape)
    m = int(a.shape[0])
    if a.shape[1] != cin * 9:
        raise ValueError("conv3_mm: a must be [M, Cin * 9]")
    out = torch.empty(m, nb * h * w, dtype=torch.float32, device=x.device)
    hb.call("atri_conv3_gemm", hb.ptr(a), hb.ptr(x), hb.ptr(out), hb.i32(m), hb.i32(cin), hb.i32(nb), hb.i32(h), hb.i32(w),
            hb.i64(a.stride(0)), hb.i32(0), hb.i32(0))
    return out


def conv3_wgrad(g2: torch.Tensor, x: torch.Tensor, chunk: int) -> torch.Tensor:
    """Split-K weight-gradient partials ``[positions / chunk, M, Cin * 9]`` of a 3x3, padding-1 convolution: slice ``z`` is
    ``g2[:, z*chunk:(z+1)*chunk] @ patches(x)[:, same]^T`` (``gemm3_kernel`` B_MODE 3); the caller adds the slices."""
    cin, nb, h, w = (int(v) for v in x.shape)
    m, pos = int(g2.shape[0]), nb * h * w
    if g2.shape[1] != pos or pos % chunk or chunk % 4:
        raise ValueError("conv3_wgrad: g2 must be [M, B * H * W] and chunk a multiple of 4 that divides the positions")
    out = torch.empty(pos // chunk, m, cin * 9, dtype=torch.float32, device=x.device)
    hb.call("atri_conv3_gemm", hb.ptr(g2), hb.ptr(x), hb.ptr(out), hb.i32(m), hb.i32(cin), hb.i32(nb), hb.i32(h), hb.i32(w),
            hb.i64(g2.stride(0)), hb.i32(1), hb.i32(chunk))
    return out


# ---------------------------------------------------------------- fused multi-head self-attention (32-wide heads)
def attention_supported(qkv: torch.Tensor, heads: int) -> bool:
    """True when :func:`attention_forward` handles packed ``qkv [L, B, 3 * 32 * heads]`` (bf16 or f32, up to 256 tokens)."""
    if not (qkv.is_cuda and qkv.dim() == 3 and qkv.dtype in (torch.bfloat16, torch.float32) and available()):
        return False
    seq, batch, width = qkv.shape
    pitch = 256 if seq > 128 else 128
    return bool(width == 96 * heads and 1 <= seq <= 256 and batch * heads * pitch * pitch < 2**32)


def _lse_pitch(seq: int) -> int:
    return 256 if seq > 128 else 128


def attention_forward(qkv: torch.Tensor, heads: int, p_drop: float, batch_major: bool = False) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None]:
    """``dropout(softmax(Q K^T / sqrt(32))) V`` per (sample, head) in one launch (``attn_fwd_kernel``).

    ``qkv`` is ``[L, B, 3 * 32 * heads]`` or, with ``batch_major``, ``[B, L, 3 * 32 * heads]`` (a sample's rows contiguous).
    Returns ``(ctx, lse, seed)`` with ``ctx`` in the layout of ``qkv``: ``lse`` (log2-domain log-sum-exp per query) and
    ``seed`` (the dropout seed, drawn from torch's generator so that ``torch.manual_seed`` governs it and a captured step
    draws a new one on every replay) are what :func:`attention_backward` needs next to ``qkv``.
    """
    qkv = qkv.contiguous()
    if batch_major:
        batch, seq, width = qkv.shape
    else:
        seq, batch, width = qkv.shape
    ctx = torch.empty(*qkv.shape[:2], width // 3, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(batch * heads, _lse_pitch(seq), dtype=torch.float32, device=qkv.device)
    seed = _drop_seed(qkv.device) if p_drop > 0 else None
    hb.call("atri_attn_fwd_layout", hb.ptr(qkv), hb.i32(1 if qkv.dtype == torch.bfloat16 else 0), hb.ptr(ctx), hb.ptr(lse), hb.ptr(seed),
            hb.i32(seq), hb.i32(batch), hb.i32(heads), hb.f32(p_drop), hb.i32(1 if batch_major else 0))
    return ctx, lse, seed


def attention_backward(qkv: torch.Tensor, dctx: torch.Tensor, lse: torch.Tensor, seed: torch.Tensor | None, heads: int,
                       p_drop: float, batch_major: bool = False) -> torch.Tensor:
    """Gradient of the packed ``qkv`` rows for :func:`attention_forward` in one launch (``attn_bwd_kernel``)."""
    if batch_major:
        batch, seq, _ = qkv.shape
    else:
        seq, batch, _ = qkv.shape
    dctx = dctx.to(qkv.dtype).contiguous()
    dqkv = torch.empty_like(qkv)
    hb.call("atri_attn_bwd_layout", hb.ptr(qkv), hb.ptr(dctx), hb.i32(1 if qkv.dtype == torch.bfloat16 else 0), hb.ptr(dqkv), hb.ptr(lse),
            hb.ptr(seed), hb.i32(seq), hb.i32(batch), hb.i32(heads), hb.f32(p_drop), hb.i32(1 if batch_major else 0))
    return dqkv


def patch_matrix_supported(x: torch.Tensor, kh: int, kw: int, ph: int, pw: int) -> bool:
    """True when :func:`patch_matrix_cm` can build the patch matrix of ``x`` ``[C, B, H, W]``."""
    if not (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.dim() == 4 and available()):
        return False
    if (kh, kw) not in ((3, 3), (5, 5)) or not (0 <= ph < kh and 0 <= pw < kw):
        return False
    c, b, h, w = x.shape
    oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    if oh <= 0 or ow <= 0 or (b * oh * ow) % 8 != 0 or c > 65535 or c * b * h * w >= 2**31 or b * oh * ow >= 2**31:
        return False
    return (2048 // (oh * ow) + 2) * h * w * x.element_size() <= 64 * 1024  # boards one workgroup stages in LDS


def patch_matrix_cm(x: torch.Tensor, kh: int, kw: int, ph: int, pw: int) -> torch.Tensor:
    """Stride-1 im2col of a channel-major activation ``[C, B, H, W]`` (bf16 or f32) -> ``[C*kh*kw, B*oh*ow]`` in one launch.

    Row order ``(c, ky, kx)``, column order ``(b, oy, ox)``: the B operand of the learner's convolution GEMMs
    (``patch_matrix_kernel`` in csrc/hip/learner_kernels.hip).
    """
    if not patch_matrix_supported(x, kh, kw, ph, pw):
        raise ValueError("patch_matrix_cm: unsupported input (needs a CUDA bf16 / f32 [C, B, H, W] tensor, 3x3 or 5x5 taps)")
    x = x.contiguous()
    c, b, h, w = x.shape
    oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    cols = torch.empty(c * kh * kw, b * oh * ow, dtype=x.dtype, device=x.device)
    hb.call("atri_patch_matrix", hb.ptr(x), hb.ptr(cols), hb.i32(1 if x.dtype == torch.bfloat16 else 0), hb.i32(c), hb.i32(b), hb.i32(h), hb.i32(w), hb.i32(kh), hb.i32(kw),
            hb.i32(ph), hb.i32(pw))
    return cols
