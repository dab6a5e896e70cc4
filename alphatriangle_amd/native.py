"""Loaders for the two in-tree native libraries.

``host()`` returns the pybind11 module (built on demand with g++ when missing:
it needs no GPU).  ``hip()`` returns the ctypes handle of the gfx950 kernel
library; it is never built implicitly on a GPU box and raises loudly when it is
missing, so that GPU code paths cannot silently fall back to eager PyTorch.
"""

from __future__ import annotations

import ctypes
import os
import importlib
import threading
from pathlib import Path
from types import ModuleType

_PKG = Path(__file__).resolve().parent
_lock = threading.Lock()
_host_mod: ModuleType | None = None
_hip_lib: ctypes.CDLL | None = None


class NativeLibraryError(RuntimeError):
    """Raised when a required native library is missing or fails to load."""


def host() -> ModuleType:
    global _host_mod
    if _host_mod is not None:
        return _host_mod
    with _lock:
        if _host_mod is None:
            try:
                _host_mod = importlib.import_module("alphatriangle_amd._atri_host")
            except ImportError:
                from .build_native import build_host

                build_host(verbose=False)
                importlib.invalidate_caches()
                _host_mod = importlib.import_module("alphatriangle_amd._atri_host")
    return _host_mod


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
