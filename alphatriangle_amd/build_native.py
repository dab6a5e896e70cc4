"""In-tree builder of the two native libraries.

* ``_atri_host``  - host C++ runtime (pybind11, g++): game engine, features,
  host tree search, sum tree, metric ring.
* ``_atri_host_wide`` - the same sources compiled with 16 occupancy words per
  game state: boards of 257 - 1024 cells (the kernel library holds the
  device engine for them as a second object of ``engine_kernels.hip``).
* ``libatri_hip`` - HIP/CDNA4 kernels for gfx950 (hipcc): batched game + tree
  engine, inference network, replay and learner kernels.  Exposed with a C ABI
  and loaded through ctypes after ``import torch`` so that it binds to the same
  HIP runtime as PyTorch.

Both are written next to this file so that they travel with the source tree.
Run ``python -m alphatriangle_amd.build_native`` (``--host`` / ``--hip`` to pick one).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
_EXT = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
HOST_SO = PKG_DIR / ("_atri_host" + _EXT)
HOST_WIDE_SO = PKG_DIR / ("_atri_host_wide" + _EXT)
WIDE_WORDS = 16  # 64 cells per word: boards of up to 1024 cells
HIP_SO = PKG_DIR / "libatri_hip.so"

HIP_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-Xclang",
    "-target-feature",
    "-Xclang",
    "-packed-fp32-ops",
    "-Wno-unused-command-line-argument",
    "-Wno-unknown-warning-option",
]


def _newer(target: Path, sources: list[Path]) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(s.stat().st_mtime <= t for s in sources)


def host_sources() -> list[Path]:
    return sorted((CSRC / "host").glob("*.cpp"))


def hip_sources() -> list[Path]:
    return sorted((CSRC / "hip").glob("*.hip"))


def _all_headers() -> list[Path]:
    return sorted(CSRC.rglob("*.h")) + sorted(CSRC.rglob("*.hpp"))


def build_host(force: bool = False, verbose: bool = True) -> Path:
    """Build both host modules (the 256-cell one and the wide one) when they are older than their sources."""
    import pybind11

    srcs = host_sources()
    deps = srcs + _all_headers() + [Path(__file__)]
    for target, defines in ((HOST_SO, []), (HOST_WIDE_SO, [f"-DATRI_MAX_WORDS={WIDE_WORDS}", "-DATRI_HOST_MODULE=_atri_host_wide"])):
        if not force and _newer(target, deps):
            continue
        cxx = os.environ.get("CXX", "g++")
        cmd = [
            cxx,
            "-O2",
            "-std=c++17",
            "-fPIC",
            "-shared",
            "-ffp-contract=off",
            "-fvisibility=hidden",
            "-Wall",
            "-Wno-unused-function",
            *defines,
            f"-I{pybind11.get_include()}",
            f"-I{sysconfig.get_paths()['include']}",
            *[str(s) for s in srcs],
            "-o",
            str(target),
        ]
        if verbose:
            print("[build_native] host:", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return HOST_SO


def hip_objects() -> list[tuple[Path, str, list[str]]]:
    """(source, object stem, extra defines) of every object of the kernel library: one per ``.hip`` file, plus the device
    engine a second time with the wide host engine's 16-word game state (boards of 257 - 1024 cells, ``atriw_*`` entries)."""
    objs: list[tuple[Path, str, list[str]]] = [(s, s.stem, []) for s in hip_sources()]
    objs.append((CSRC / "hip" / "engine_kernels.hip", "engine_kernels_wide", [f"-DATRI_MAX_WORDS={WIDE_WORDS}"]))
    return objs


def build_hip(force: bool = False, verbose: bool = True) -> Path:
    srcs = hip_sources()
    if not srcs:
        raise RuntimeError("no HIP sources found")
    deps = srcs + _all_headers() + [Path(__file__)]
    if not force and _newer(HIP_SO, deps):
        return HIP_SO
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    obj_dir = PKG_DIR / "build" / "hip"
    obj_dir.mkdir(parents=True, exist_ok=True)
    procs = []
    for s, stem, defines in hip_objects():
        o = obj_dir / (stem + ".o")
        objs.append(o)
        if not force and _newer(o, [s] + _all_headers() + [Path(__file__)]):
            continue
        cmd = [hipcc, *[f for f in HIP_FLAGS if f != "-shared"], *defines, "-c", str(s), "-o", str(o)]
        if verbose:
            print("[build_native] hip:", " ".join(cmd), flush=True)
        procs.append((stem, subprocess.Popen(cmd)))
    for stem, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {stem}")
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(HIP_SO)]
    if verbose:
        print("[build_native] link:", " ".join(link), flush=True)
    subprocess.check_call(link)
    return HIP_SO


SANITIZERS = {"address": "address,undefined", "thread": "thread"}


def build_selftest(kind: str = "address", verbose: bool = True, wide: bool = False) -> Path:
    """Build the stand-alone host-runtime exerciser with ``-fsanitize=<kind>`` (address+undefined or thread);
    ``wide`` compiles the 16-word game state of ``_atri_host_wide`` and adds a 600-cell board to the run.

    The sanitized process contains only the C++ runtime (engine, search, sum tree, metric ring), so no
    Python allocator noise and no preloading is needed; ``tests/test_host_sanitizers.py`` runs the binaries.
    """
    if kind not in SANITIZERS:
        raise ValueError(f"unknown sanitizer '{kind}' (choose from {sorted(SANITIZERS)})")
    out_dir = PKG_DIR / "build" / "selftest"
    out_dir.mkdir(parents=True, exist_ok=True)
    out = out_dir / (f"host_selftest_{kind}" + ("_wide" if wide else ""))
    src = CSRC / "selftest" / "host_selftest.cpp"
    cmd = [
        os.environ.get("CXX", "g++"), "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-omit-frame-pointer",
        f"-fsanitize={SANITIZERS[kind]}", *([f"-DATRI_MAX_WORDS={WIDE_WORDS}"] if wide else []), str(src), "-o", str(out), "-pthread",
    ]
    if verbose:
        print("[build_native] selftest:", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_all(force: bool = False, verbose: bool = True) -> None:
    build_host(force, verbose)
    build_hip(force, verbose)


if __name__ == "__main__":
    args = set(sys.argv[1:])
    force = "--force" in args
    if "--selftest" in args:
        kinds = [a for a in args if a in SANITIZERS] or list(SANITIZERS)
        for k in kinds:
            print(build_selftest(k))
    elif "--host" in args:
        build_host(force)
    elif "--hip" in args:
        build_hip(force)
    else:
        build_all(force)
