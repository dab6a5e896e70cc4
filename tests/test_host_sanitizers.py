"""Host C++ runtime under AddressSanitizer + UBSan and under ThreadSanitizer (CPU only).

The reference has no sanitizer coverage at all; here the engine, the host search, the sum tree and the
lock-free metric ring (four producer threads against one consumer) run in a stand-alone sanitized binary.
"""

from __future__ import annotations

import shutil
import subprocess

import pytest

from alphatriangle_amd import build_native


@pytest.mark.parametrize("wide", [False, True], ids=["256-cell-state", "1024-cell-state"])
@pytest.mark.parametrize("kind", ["address", "thread"])
def test_host_runtime_is_clean_under_sanitizer(kind, wide):
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    try:
        exe = build_native.build_selftest(kind, verbose=False, wide=wide)
    except subprocess.CalledProcessError as exc:  # toolchain without the sanitizer runtime
        pytest.skip(f"sanitizer build not available: {exc}")
    env = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "TSAN_OPTIONS": "halt_on_error=1", "PATH": "/usr/bin:/bin"}
    proc = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert proc.returncode == 0, f"{kind} sanitizer run failed:\n{proc.stdout}\n{proc.stderr[-4000:]}"
    assert "host selftest: ok" in proc.stdout
