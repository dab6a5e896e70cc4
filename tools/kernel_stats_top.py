"""Top kernels of a rocprofv3 ``*_kernel_stats.csv`` (already copied back from the GPU box).

usage: python tools/kernel_stats_top.py gpurun_out/prof/bench_kernel_stats.csv [n]
"""
import csv
import sys


def main() -> None:
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    print(f"total {total / 1e6:.2f} ms in {calls} kernel launches")
    for r in rows[:n]:
        print(f"{float(r['TotalDurationNs']) / 1e6:8.2f} ms {r['Calls']:>6} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")


if __name__ == "__main__":
    main()
