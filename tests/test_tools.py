"""Command-line helpers under tools/ that work on files copied back from a GPU box."""
from __future__ import annotations

import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_learning_curve_averages_events_per_window_and_skips_a_cut_off_line(tmp_path):
    src = tmp_path / "metrics.jsonl"
    with open(src, "w", encoding="utf-8") as fh:
        for step in range(0, 250, 10):
            fh.write(json.dumps({"name": "RL/Score", "value": float(step), "step": step, "time": 0.0}) + "\n")
            fh.write(json.dumps({"name": "Loss/Total", "value": 5.0 - step / 100, "step": step, "time": 0.0}) + "\n")
        fh.write(json.dumps({"name": "Custom/Ignored", "value": 1.0, "step": 900, "time": 0.0}) + "\n")
        fh.write('{"name": "RL/Sc')  # the run ended in the middle of a line
    out = tmp_path / "curve.csv"
    subprocess.run([sys.executable, str(ROOT / "tools" / "learning_curve.py"), str(src), str(out), "100"], check=True)
    lines = out.read_text().strip().split("\n")
    header = lines[0].split(",")
    assert header[0] == "learner_step" and "RL/Score" in header and "Loss/Total" in header
    rows = [dict(zip(header, line.split(","))) for line in lines[1:]]
    assert [r["learner_step"] for r in rows] == ["0", "100", "200"]  # the window with only an unlisted metric is skipped
    assert float(rows[0]["RL/Score"]) == 45.0 and float(rows[1]["RL/Score"]) == 145.0 and float(rows[2]["RL/Score"]) == 220.0
    assert abs(float(rows[1]["Loss/Total"]) - 3.55) < 1e-9
    assert rows[0]["Buffer/Size"] == ""


def test_kernel_stats_top_prints_the_requested_number_of_rows(tmp_path):
    src = tmp_path / "stats.csv"
    src.write_text(
        '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
        '"kernel_a(int)",10,5000000,500000.0,62.5,1,2,0.1\n'
        '"kernel_b(int)",20,3000000,150000.0,37.5,1,2,0.1\n'
    )
    res = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_stats_top.py"), str(src), "1"], check=True,
                         capture_output=True, text=True)
    lines = res.stdout.strip().split("\n")
    assert lines[0].startswith("total 8.00 ms in 30 kernel launches")
    assert len(lines) == 2 and "kernel_a" in lines[1] and "500.0 us" in lines[1]
