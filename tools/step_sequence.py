"""One learner step as the ordered list of its kernels, from a rocprofv3 ``*_kernel_trace.csv``.

The flat optimizer kernel ends every step; the kernels between its last two launches are one steady-state step.

usage: python tools/step_sequence.py gpurun_out/prof/<name>_kernel_trace.csv [marker-substring]
"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main() -> None:
    rows = list(csv.DictReader(open(sys.argv[1])))
    marker = sys.argv[2] if len(sys.argv) > 2 else "flat_optimizer"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(ends) < 3:
        raise SystemExit(f"fewer than three '{marker}' launches in the trace")
    a, b = ends[-3], ends[-2]
    step = rows[a + 1 : b + 1]
    t0 = int(step[0]["Start_Timestamp"])
    busy = 0
    for i, r in enumerate(step):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        print(f"{i:4d} {(s - t0) / 1e3:9.1f} us {(e - s) / 1e3:7.1f} us  {short(r['Kernel_Name'])}")
    span = int(step[-1]["End_Timestamp"]) - t0
    print(f"{len(step)} launches, {span / 1e3:.1f} us from first start to last end, {busy / 1e3:.1f} us inside kernels")


if __name__ == "__main__":
    main()
