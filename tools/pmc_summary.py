"""Per-kernel averages of a rocprofv3 --pmc counter_collection CSV.

usage: python tools/pmc_summary.py gpurun_out/pmc/p_counter_collection.csv
"""
import csv
import sys
from collections import defaultdict


def main() -> None:
    acc: dict[str, dict[str, float]] = defaultdict(lambda: defaultdict(float))
    disp: dict[str, set] = defaultdict(set)
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    names = sorted({c for v in acc.values() for c in v})
    print(f"{'kernel':40s} {'calls':>6s} " + " ".join(f"{n[-16:]:>16s}" for n in names))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get(names[0], 0))):
        n = len(disp[k])
        print(f"{k:40s} {n:6d} " + " ".join(f"{v.get(c, 0) / n:16.0f}" for c in names))


if __name__ == "__main__":
    main()
