"""ctypes binding of ``libatri_hip.so`` (C ABI of the gfx950 kernel library).

Kernels are launched on PyTorch's current HIP stream, so they order with tensor
ops and can be captured into HIP graphs by ``torch.cuda.graph``.  Every entry
point returns a HIP error code (0 = success, negative = argument rejected on the
host before any launch); ``check`` turns that into an exception so a GPU code
path can never fail silently.
"""

from __future__ import annotations

import ctypes as C
from typing import Any

import numpy as np
import torch

from .. import native

VP = C.c_void_p


class GeomViewC(C.Structure):
    _fields_ = [
        ("rows", C.c_int32),
        ("cols", C.c_int32),
        ("cells", C.c_int32),
        ("nwords", C.c_int32),
        ("slots", C.c_int32),
        ("n_templates", C.c_int32),
        ("n_lines", C.c_int32),
        ("action_dim", C.c_int32),
        ("playable_count", C.c_int32),
        ("native_other_dim", C.c_int32),
        ("r_place", C.c_float),
        ("r_clear", C.c_float),
        ("r_alive", C.c_float),
        ("r_over", C.c_float),
        ("playable", VP),
        ("place_mask", VP),
        ("place_ok", VP),
        ("line_mask", VP),
        ("shape_cdf", VP),
        ("shape_feat", VP),
        ("tmpl_ntris", VP),
    ]


class SearchParamsC(C.Structure):
    _fields_ = [
        ("max_simulations", C.c_int32),
        ("max_depth", C.c_int32),
        ("leaf_batch", C.c_int32),
        ("fpu_parent", C.c_int32),
        ("cpuct", C.c_float),
        ("dirichlet_alpha", C.c_float),
        ("dirichlet_epsilon", C.c_float),
        ("discount", C.c_float),
        ("virtual_loss", C.c_float),
    ]


class TreeBufsC(C.Structure):
    _fields_ = [
        ("states", VP * 2),
        ("meta", VP * 2),
        ("edges", VP * 2),
        ("cur", VP),
        ("n_nodes", VP),
        ("slot_kind", VP),
        ("slot_leaf", VP),
        ("slot_plen", VP),
        ("path_node", VP),
        ("path_action", VP),
        ("nn_grid", VP),
        ("nn_other", VP),
        ("nn_logits", VP),
        ("nn_value", VP),
        ("order", VP),
        ("census", VP),
        ("slot_row", VP),
        ("live_cursor", VP),
        ("G", C.c_int32),
        ("C", C.c_int32),
        ("A", C.c_int32),
        ("L", C.c_int32),
        ("P", C.c_int32),
        ("other_dim", C.c_int32),
        ("logit_stride", C.c_int32),
        ("pad", C.c_int32),
    ]


class GameBufsC(C.Structure):
    _fields_ = [
        ("game_rng", VP),
        ("episode_idx", VP),
        ("ep_sims", VP),
        ("pend_grid", VP),
        ("pend_other", VP),
        ("pend_policy", VP),
        ("pend_acc", VP),
        ("pend_count", VP),
        ("pend_head", VP),
        ("pend_len", VP),
        ("rp_grid", VP),
        ("rp_other", VP),
        ("rp_policy", VP),
        ("rp_ret", VP),
        ("rp_head", VP),
        ("emitted_idx", VP),
        ("emitted_count", VP),
        ("ep_score", VP),
        ("ep_len", VP),
        ("ep_sims_out", VP),
        ("ep_cleared", VP),
        ("ep_count", VP),
        ("move_stats", VP),
        ("next_root", VP),
        ("reroot_child", VP),
        ("chosen_action", VP),
        ("gamma_pow", VP),
        ("step_reward", VP),
        ("step_cleared", VP),
        ("step_score", VP),
        ("base_seed", C.c_uint64),
        ("R", C.c_int32),
        ("n_step", C.c_int32),
        ("cap", C.c_int32),
        ("emit_cap", C.c_int32),
        ("ep_ring", C.c_int32),
        ("temp_switch_step", C.c_int32),
        ("tree_reuse", C.c_int32),
        ("reuse_limit", C.c_int32),
        ("temp_hi", C.c_float),
        ("temp_lo", C.c_float),
    ]


class HipError(RuntimeError):
    pass


_checked = False


def lib() -> C.CDLL:
    """The loaded kernel library (struct sizes verified against the C side once)."""
    global _checked
    handle = native.hip()
    if not _checked:
        for fn, cls in (
            ("atri_sizeof_geomview", GeomViewC),
            ("atri_sizeof_searchparams", SearchParamsC),
            ("atri_sizeof_treebufs", TreeBufsC),
            ("atri_sizeof_gamebufs", GameBufsC),
            ("atriw_sizeof_geomview", GeomViewC),
            ("atriw_sizeof_searchparams", SearchParamsC),
            ("atriw_sizeof_treebufs", TreeBufsC),
            ("atriw_sizeof_gamebufs", GameBufsC),
        ):
            f = getattr(handle, fn)
            f.restype = C.c_int
            if f() != C.sizeof(cls):
                raise HipError(f"ABI mismatch: {fn}() = {f()} but ctypes {cls.__name__} is {C.sizeof(cls)} bytes")
        _checked = True
    return handle


def stream_ptr() -> VP:
    return VP(torch.cuda.current_stream().cuda_stream)


def ptr(t: torch.Tensor | None) -> VP:
    return VP(0) if t is None else VP(t.data_ptr())


def check(code: int, what: str) -> None:
    if code != 0:
        raise HipError(f"{what} failed with code {code}")


def call(name: str, *args: Any) -> None:
    """Call ``name(*args, stream)`` and raise on a non-zero return code."""
    fn = getattr(lib(), name)
    fn.restype = C.c_int
    check(fn(*args, stream_ptr()), name)


def f32(x: float) -> C.c_float:
    return C.c_float(float(x))


def i32(x: int) -> C.c_int:
    return C.c_int(int(x))


def i64(x: int) -> C.c_longlong:
    return C.c_longlong(int(x))


class DeviceGeometry:
    """Geometry tables mirrored into HBM plus the by-value ``GeomView`` kernel argument."""

    def __init__(self, geom: Any, device: torch.device):
        self.geom = geom
        self.device = device

        def up(arr: np.ndarray, dtype: torch.dtype | None = None) -> torch.Tensor:
            a = np.ascontiguousarray(arr)
            if a.dtype == np.uint64:
                a = a.view(np.int64)
            elif a.dtype == np.uint32:
                a = a.view(np.int32)
            t = torch.from_numpy(a.copy()).to(device)
            return t if dtype is None else t.to(dtype)

        self.playable = up(geom.playable_mask())
        self.place_mask = up(geom.place_mask())
        self.place_ok = up(geom.place_ok())
        lm = geom.line_mask()
        self.line_mask = up(lm) if lm.size else torch.zeros(1, dtype=torch.int64, device=device)
        self.shape_cdf = up(geom.shape_cdf())
        self.shape_feat = up(geom.shape_feat())
        self.tmpl_ntris = up(geom.tmpl_ntris())
        r_place, r_clear, r_alive, r_over = geom.rewards()
        v = GeomViewC()
        v.rows, v.cols, v.cells, v.nwords = geom.rows, geom.cols, geom.cells, geom.nwords
        v.slots, v.n_templates, v.n_lines, v.action_dim = geom.slots, geom.n_templates, geom.n_lines, geom.action_dim
        v.playable_count = geom.playable_count
        v.native_other_dim = 8 * geom.slots + 6
        v.r_place, v.r_clear, v.r_alive, v.r_over = r_place, r_clear, r_alive, r_over
        v.playable = self.playable.data_ptr()
        v.place_mask = self.place_mask.data_ptr()
        v.place_ok = self.place_ok.data_ptr()
        v.line_mask = self.line_mask.data_ptr()
        v.shape_cdf = self.shape_cdf.data_ptr()
        v.shape_feat = self.shape_feat.data_ptr()
        v.tmpl_ntris = self.tmpl_ntris.data_ptr()
        self.view = v

    def ref(self) -> Any:
        return C.byref(self.view)
