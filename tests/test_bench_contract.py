"""bench.py's output contract: exactly one JSON line on stdout with the fields the benchmark driver reads."""

from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _run(args: list[str], timeout: int = 300) -> subprocess.CompletedProcess:
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only behaviour")
def test_bench_without_gpu_reports_one_json_error_line():
    res = _run(["--steps", "1", "--warmup", "0"])
    assert res.returncode == 1
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    assert "error" in json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_bench_prints_exactly_one_json_line_with_the_contract_fields(dtype):
    res = _run(["--steps", "2", "--warmup", "1", "--games", "512", "--sims", "16", "--train-steps", "2", "--dtype", dtype])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert REQUIRED <= set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["dtype"] == dtype
    assert out["vs_baseline"] is None  # 16 simulations: no reference-style figure exists for this configuration
    assert out["trainer_samples_per_sec_unthrottled"] > out["trainer_samples_per_sec"]
    assert 0 < out["leaf_slots_needing_evaluation"] <= 1
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # value = games * steps / elapsed: consistent with the per-step time it reports
    assert abs(out["value"] - 512 / (out["ms_per_step"] / 1e3)) / out["value"] < 0.02
    assert {"model", "global_batch", "seq_len", "parallelism"} <= set(out["config"])
    assert out["trainer_samples_per_sec"] > 0


def test_bench_metric_string_is_the_one_baseline_json_names():
    metric = json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    assert f'"metric": "{metric}"' in (ROOT / "bench.py").read_text()


def test_all_configs_appends_the_other_presets_and_leaves_the_headline_alone():
    """--all-configs (cpu engine stand-in): the extra key lists the other configurations, the headline fields keep their
    meaning (value = games / time per step of the timed run) and the line says which backend and precision it ran on."""
    res = _run(["--engine", "cpu", "--steps", "2", "--warmup", "1", "--games", "2", "--sims", "4", "--all-configs"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert REQUIRED | {"dtype_detail", "backend", "ranks_seen", "configs"} <= set(out)
    assert out["backend"] == "none" and out["ranks_seen"] == 1 and "fp32" in out["dtype_detail"]
    assert abs(out["value"] - 2 / (out["ms_per_step"] / 1e3)) / out["value"] < 0.02
    assert len(out["configs"]) >= 2
    for c in out["configs"]:
        assert {"preset", "sims", "dtype", "moves_per_s"} <= set(c) and c["moves_per_s"] > 0
    assert "all-reduce" not in out["config"]["parallelism"]  # one rank: no collective is claimed
    plain = json.loads([ln for ln in _run(["--engine", "cpu", "--steps", "2", "--warmup", "1", "--games", "2", "--sims", "4"]).stdout.splitlines() if ln.strip()][0])
    assert "configs" not in plain and set(plain) == set(out) - {"configs"}


@pytest.mark.gpu
def test_bench_line_names_its_precision_scheme():
    res = _run(["--steps", "1", "--warmup", "1", "--games", "256", "--sims", "8", "--train-steps", "1"])
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert out["dtype"] == "fp32" and out["dtype_detail"].startswith("fp16x3") and "22-bit operands" in out["dtype_detail"]
    assert out["backend"] == "none" and out["ranks_seen"] == 1
