"""Flat parameter space and the fused flat-buffer optimizer.

``FlatParameterSpace`` re-points every parameter (and its gradient) of a module
into one contiguous fp32 buffer: one RCCL all-reduce moves all gradients, one
broadcast moves all weights, and one HIP kernel pair (csrc/hip/learner_kernels.hip:
``sumsq`` + ``flat_optimizer_step``) does gradient clipping and the Adam / AdamW /
SGD-momentum update for the whole model.

``FusedFlatOptimizer`` is a ``torch.optim.Optimizer`` so LR schedulers and
``state_dict()`` keep working; its per-parameter state tensors are views into flat
state buffers, which makes the saved optimizer state interchangeable with
``torch.optim`` (alphatriangle/rl/core/trainer.py:48-64 creates Adam / AdamW /
SGD(momentum=0.9), all with weight decay; clipping is
``clip_grad_norm_(max_norm=GRADIENT_CLIP_VALUE)`` at :276-282).
"""

from __future__ import annotations

import logging
from typing import Any

import torch

logger = logging.getLogger(__name__)


class FlatParameterSpace:
    """All parameters of ``module`` as views of one flat fp32 buffer (same for gradients)."""

    def __init__(self, module: torch.nn.Module):
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        device = self.params[0].device
        # 64-element alignment keeps every tensor 256-byte aligned inside the buffer.
        self.offsets: list[int] = []
        total = 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 63) // 64 * 64
        self.total = total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets, strict=True):
                n = p.numel()
                self.flat_param[off : off + n].copy_(p.detach().reshape(-1).float())
                p.data = self.flat_param[off : off + n].view(p.shape)
                p.grad = self.flat_grad[off : off + n].view(p.shape)

    def refresh_shadow(self, dtype: torch.dtype = torch.bfloat16) -> None:
        """Low-precision copy of every parameter in ONE cast kernel over the flat buffer.

        Each parameter gets its view of the copy as ``p._atri_shadow``; the learner's GEMM-lowered layers
        (``models/model.py``) take that view under autocast instead of casting their weight themselves, which replaces
        one small cast kernel per weight and step (46 of them for the default network) by this single launch.  It is
        called at the start of every autocast step, so whoever wrote the parameters in between (optimizer kernel,
        checkpoint load, weight broadcast) is picked up.
        """
        shadow = getattr(self, "flat_shadow", None)
        if shadow is None or shadow.dtype != dtype:
            shadow = torch.empty(self.total, dtype=dtype, device=self.flat_param.device)
            self.flat_shadow = shadow
            for p, off in zip(self.params, self.offsets, strict=True):
                p._atri_shadow = shadow[off : off + p.numel()].view(p.shape)
        with torch.no_grad():
            shadow.copy_(self.flat_param)

    def drop_shadow(self) -> None:
        self.flat_shadow = None
        for p in self.params:
            if hasattr(p, "_atri_shadow"):
                del p._atri_shadow

    def rebind(self) -> None:
        """Re-attach views after something replaced ``p.data`` / ``p.grad`` (e.g. ``zero_grad(set_to_none=True)``)."""
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets, strict=True):
                n = p.numel()
                view = self.flat_param[off : off + n].view(p.shape)
                if p.data.data_ptr() != view.data_ptr():
                    view.copy_(p.data)
                    p.data = view
                gview = self.flat_grad[off : off + n].view(p.shape)
                if p.grad is None or p.grad.data_ptr() != gview.data_ptr():
                    if p.grad is not None:
                        gview.copy_(p.grad)
                    p.grad = gview

    def zero_grad(self) -> None:
        self.flat_grad.zero_()

    def release_grads(self) -> None:
        """Detach ``p.grad`` from the flat buffer so that autograd hands each parameter its gradient tensor as it is.

        Accumulating into the views costs one small add kernel per parameter and step (70 of them for the default
        network); with ``p.grad = None`` autograd just stores the tensor it computed and :meth:`gather_grads` moves
        all of them into the flat buffer with one multi-tensor copy.
        """
        for p in self.params:
            p.grad = None

    def gather_grads(self) -> None:
        """Copy the gradients autograd left in ``p.grad`` into the flat buffer and re-attach the views."""
        views = self.view_like_params(self.flat_grad)
        src, dst = [], []
        with torch.no_grad():
            for p, v in zip(self.params, views, strict=True):
                if p.grad is None:
                    v.zero_()  # parameter took no part in this step
                elif p.grad.data_ptr() != v.data_ptr():
                    src.append(p.grad)
                    dst.append(v)
            if src:
                torch._foreach_copy_(dst, src)
            for p, v in zip(self.params, views, strict=True):
                p.grad = v

    def view_like_params(self, flat: torch.Tensor) -> list[torch.Tensor]:
        return [flat[off : off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets, strict=True)]


class FusedFlatOptimizer(torch.optim.Optimizer):
    """Adam / AdamW / SGD-momentum over a :class:`FlatParameterSpace` with fused gradient clipping."""

    def __init__(
        self,
        space: FlatParameterSpace,
        kind: str,
        lr: float,
        weight_decay: float,
        max_grad_norm: float | None,
        betas: tuple[float, float] = (0.9, 0.999),
        eps: float = 1e-8,
        momentum: float = 0.9,
    ):
        if kind not in ("Adam", "AdamW", "SGD"):
            raise ValueError(f"Unsupported optimizer type: {kind}")
        self.space = space
        self.kind = kind
        self.max_grad_norm = max_grad_norm
        self.grad_scale = 1.0  # set to 1/world_size for summed data-parallel gradients
        defaults = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps, momentum=momentum)
        super().__init__(space.params, defaults)
        dev = space.flat_param.device
        self.flat_m = torch.zeros(space.total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(space.total, dtype=torch.float32, device=dev) if kind != "SGD" else None
        self._step_count = 0
        self._norm_buf = torch.zeros(2, dtype=torch.float32, device=dev)
        m_views = space.view_like_params(self.flat_m)
        v_views = space.view_like_params(self.flat_v) if self.flat_v is not None else None
        for i, p in enumerate(space.params):
            st: dict[str, Any] = self.state[p]
            if kind == "SGD":
                st["momentum_buffer"] = m_views[i]
            else:
                st["step"] = torch.zeros((), dtype=torch.float32)
                st["exp_avg"] = m_views[i]
                assert v_views is not None
                st["exp_avg_sq"] = v_views[i]
        self.last_grad_norm: torch.Tensor | None = None

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: ARG002 - grads must stay views
        self.space.zero_grad()

    def load_state_dict(self, state_dict: dict[str, Any]) -> None:
        """Accepts ``torch.optim`` format; copies into the flat state buffers (views stay attached)."""
        groups = state_dict.get("param_groups", [])
        for g_saved, g in zip(groups, self.param_groups, strict=False):
            for k in ("lr", "weight_decay", "betas", "eps", "momentum", "initial_lr"):
                if k in g_saved:
                    g[k] = g_saved[k]
        saved = state_dict.get("state", {})
        max_step = 0
        for i, p in enumerate(self.space.params):
            st_saved = saved.get(i, saved.get(str(i)))
            if not st_saved:
                continue
            st = self.state[p]
            for k in ("exp_avg", "exp_avg_sq", "momentum_buffer"):
                if k in st_saved and st_saved[k] is not None and k in st:
                    st[k].copy_(st_saved[k].to(st[k].device))
            if "step" in st_saved and "step" in st:
                step_val = float(st_saved["step"])
                st["step"] = torch.tensor(step_val, dtype=torch.float32)
                max_step = max(max_step, int(step_val))
        self._step_count = max_step

    @torch.no_grad()
    def step(self, closure=None):  # noqa: ANN001
        if closure is not None:
            raise RuntimeError("FusedFlatOptimizer does not support closures")
        g = self.param_groups[0]
        self._step_count += 1
        sp = self.space
        if sp.flat_param.is_cuda:
            from . import hip_learner

            hip_learner.require()
            self.last_grad_norm = hip_learner.flat_optimizer_step(
                kind=self.kind,
                param=sp.flat_param,
                grad=sp.flat_grad,
                m=self.flat_m,
                v=self.flat_v,
                norm_buf=self._norm_buf,
                step=self._step_count,
                lr=float(g["lr"]),
                beta1=float(g["betas"][0]),
                beta2=float(g["betas"][1]),
                eps=float(g["eps"]),
                weight_decay=float(g["weight_decay"]),
                momentum=float(g["momentum"]),
                max_grad_norm=self.max_grad_norm,
                grad_scale=self.grad_scale,
            )
        else:
            self.last_grad_norm = _flat_step_reference(self, g)
        if self.kind != "SGD":
            for p in sp.params:
                self.state[p]["step"] = torch.tensor(float(self._step_count), dtype=torch.float32)
        return None


def _flat_step_reference(opt: FusedFlatOptimizer, g: dict[str, Any]) -> torch.Tensor:
    """Pure PyTorch twin of the HIP kernels (used on CPU and as the kernel-test reference)."""
    sp = opt.space
    grad = sp.flat_grad
    if opt.grad_scale != 1.0:
        grad.mul_(opt.grad_scale)
    norm = torch.linalg.vector_norm(grad)
    if opt.max_grad_norm is not None:
        coef = torch.clamp(opt.max_grad_norm / (norm + 1e-6), max=1.0)
        grad.mul_(coef)
    lr, wd = float(g["lr"]), float(g["weight_decay"])
    p = sp.flat_param
    if opt.kind == "SGD":
        d = grad + wd * p if wd != 0 else grad.clone()
        mom = float(g["momentum"])
        if opt._step_count == 1:
            opt.flat_m.copy_(d)
        else:
            opt.flat_m.mul_(mom).add_(d)
        p.add_(opt.flat_m, alpha=-lr)
        return norm
    b1, b2 = g["betas"]
    eps = float(g["eps"])
    t = opt._step_count
    if opt.kind == "AdamW":
        p.mul_(1.0 - lr * wd)
        d = grad
    else:
        d = grad + wd * p if wd != 0 else grad
    opt.flat_m.mul_(b1).add_(d, alpha=1 - b1)
    assert opt.flat_v is not None
    opt.flat_v.mul_(b2).addcmul_(d, d, value=1 - b2)
    bc1 = 1 - b1**t
    bc2 = 1 - b2**t
    denom = (opt.flat_v.sqrt() / (bc2**0.5)).add_(eps)
    p.addcdiv_(opt.flat_m, denom, value=-lr / bc1)
    return norm


def build_optimizer(model: torch.nn.Module, train_config: Any, fused: bool) -> tuple[torch.optim.Optimizer, FlatParameterSpace | None]:
    """Reference-equivalent optimizer; ``fused`` selects the flat-buffer implementation."""
    lr, wd, kind = train_config.LEARNING_RATE, train_config.WEIGHT_DECAY, train_config.OPTIMIZER_TYPE
    if fused:
        space = FlatParameterSpace(model)
        return FusedFlatOptimizer(space, kind, lr, wd, train_config.GRADIENT_CLIP_VALUE), space
    params = model.parameters()
    if kind == "Adam":
        return torch.optim.Adam(params, lr=lr, weight_decay=wd), None
    if kind == "AdamW":
        return torch.optim.AdamW(params, lr=lr, weight_decay=wd), None
    if kind == "SGD":
        return torch.optim.SGD(params, lr=lr, weight_decay=wd, momentum=0.9), None
    raise ValueError(f"Unsupported optimizer type: {kind}")
