"""Loaders for the in-tree native libraries.

``host()`` returns the pybind11 module (built on demand with g++ when missing:
it needs no GPU); ``host(cells)`` returns the module whose game state holds a
board of that many cells (the wide build above 256 cells), ``host_of(obj)`` the
module a bound object came from.  ``hip()`` returns the ctypes handle of the gfx950 kernel
library; it is never built implicitly on a GPU box and raises loudly when it is
missing, so that GPU code paths cannot silently fall back to eager PyTorch.
"""

from __future__ import annotations

import ctypes
import os
import importlib
import sys
import threading
from pathlib import Path
from types import ModuleType

_PKG = Path(__file__).resolve().parent
_lock = threading.Lock()
_host_mods: dict[str, ModuleType] = {}
_hip_lib: ctypes.CDLL | None = None

NARROW_CELLS = 256   # four 64-bit occupancy words: the 64-byte state shared with the device engine
WIDE_CELLS = 1024    # sixteen words: the wide host module (and the atriw_* entry points of the kernel library)


class NativeLibraryError(RuntimeError):
    """Raised when a required native library is missing or fails to load."""


def _load_host(name: str) -> ModuleType:
    mod = _host_mods.get(name)
    if mod is not None:
        return mod
    with _lock:
        mod = _host_mods.get(name)
        if mod is None:
            try:
                mod = importlib.import_module(f"alphatriangle_amd.{name}")
            except ImportError:
                from .build_native import build_host

                build_host(verbose=False)
                importlib.invalidate_caches()
                mod = importlib.import_module(f"alphatriangle_amd.{name}")
            _host_mods[name] = mod
    return mod


def host(cells: int | None = None) -> ModuleType:
    """The host runtime.  ``cells`` picks the build whose game state fits a board of that size; everything that does not
    touch a game state (sum tree, metric ring) uses the default module."""
    if cells is not None and cells > WIDE_CELLS:
        raise NativeLibraryError(f"a board of {cells} cells exceeds the widest host engine build ({WIDE_CELLS} cells)")
    return _load_host("_atri_host_wide" if cells is not None and cells > NARROW_CELLS else "_atri_host")


def host_of(obj: object) -> ModuleType:
    """The host module that created a bound object (geometry, state, tree): the two builds do not accept each other's."""
    return sys.modules[type(obj).__module__]


def hip_library_path() -> Path:
    """The in-tree kernel library; ``ATRI_HIP_LIB`` names another build of it (same-box A/B runs of kernel variants)."""
    override = os.environ.get("ATRI_HIP_LIB")
    return Path(override) if override else _PKG / "libatri_hip.so"


def hip_available() -> bool:
    """True when the kernel library exists and a GPU is visible to PyTorch."""
    import torch

    return hip_library_path().exists() and torch.cuda.is_available()


def hip() -> ctypes.CDLL:
    """ctypes handle of ``libatri_hip.so`` (import torch first so both share one HIP runtime)."""
    global _hip_lib
    if _hip_lib is not None:
        return _hip_lib
    with _lock:
        if _hip_lib is None:
            import torch  # noqa: F401  (loads libamdhip64 before our library binds to it)

            path = hip_library_path()
            if not path.exists():
                raise NativeLibraryError(
                    f"{path} is missing. Build it with `python -m alphatriangle_amd.build_native --hip` "
                    "(hipcc cross-compiles gfx950 without a GPU)."
                )
            try:
                _hip_lib = ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)
            except OSError as exc:
                raise NativeLibraryError(f"failed to load {path}: {exc}") from exc
    return _hip_lib
