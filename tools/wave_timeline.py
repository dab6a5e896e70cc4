"""Print the kernel timeline of one MCTS wave from a rocprofv3 kernel trace CSV.

usage: python tools/wave_timeline.py gpurun_out/prof/bench_kernel_trace.csv [n_kernels] [position]

`position` in 0 .. 1 picks the wave by its place in the trace (default 0.5: the middle one; 0.95 of a run with a long
warm-up shows a wave of the late moves).
"""
import csv
import sys


def main() -> None:
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 26
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("tree_wave")]
    pos = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    i = idx[min(int(len(idx) * pos), len(idx) - 1)]
    t0 = int(rows[i]["Start_Timestamp"])
    for r in rows[i : i + n]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f} us  grid={r['Grid_Size_X']:>8} wg={r['Workgroup_Size_X']:>4} "
              f"vgpr={r['VGPR_Count']:>4}+{r['Accum_VGPR_Count']:<4} {r['Kernel_Name'][:48]}")


if __name__ == "__main__":
    main()
