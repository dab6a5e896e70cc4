"""One CSV row per `every` learner steps from a run's ``metrics.jsonl`` (the JSONL sink of the stats service).

usage: python tools/learning_curve.py <run_dir>/metrics.jsonl [out.csv] [every=100]

Every metric is averaged over the events whose step falls into the row's window (``step // every``); columns are the
metric names in the order given below, rows without any event are skipped.
"""
from __future__ import annotations

import json
import sys
from collections import defaultdict

COLUMNS = ["RL/Score", "RL/Episode_Length", "RL/Triangles_Cleared_Episode", "Loss/Total", "Loss/Policy", "Loss/Value",
           "Loss/Entropy", "Progress/Episodes_Played", "Rate/Simulations_Per_Sec", "Rate/Steps_Per_Sec", "Buffer/Size"]


def main() -> None:
    src = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    sums: dict[int, dict[str, list[float]]] = defaultdict(lambda: defaultdict(lambda: [0.0, 0.0]))
    with open(src, encoding="utf-8") as fh:
        for line in fh:
            try:
                ev = json.loads(line)
            except json.JSONDecodeError:
                continue  # a line cut off by the end of the run
            cell = sums[int(ev["step"]) // every][ev["name"]]
            cell[0] += float(ev["value"])
            cell[1] += 1.0
    print("learner_step," + ",".join(COLUMNS), file=out)
    for window in sorted(sums):
        row = sums[window]
        if not any(c in row for c in COLUMNS):
            continue
        cells = [f"{row[c][0] / row[c][1]:.4g}" if c in row else "" for c in COLUMNS]
        print(f"{window * every}," + ",".join(cells), file=out)


if __name__ == "__main__":
    main()
