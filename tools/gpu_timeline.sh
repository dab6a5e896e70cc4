#!/bin/bash
# Kernel timeline of single MCTS waves of a bench.py run (on the GPU box): tools/gpu_timeline.sh <name> "<positions>" [bench.py args...]
# positions: places in the trace (0 .. 1) whose wave is printed, e.g. "0.3 0.95".
set -eo pipefail
name="$1"; positions="$2"; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$name"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$out" -o run -- python3 "$root/bench.py" --no-all-configs "$@" > "$out/bench.log" 2>&1
cd "$root"
for p in $positions; do
  echo "--- wave at $p of the trace" >> "$out/timeline.txt"
  python3 tools/wave_timeline.py "$out/run_kernel_trace.csv" 14 "$p" >> "$out/timeline.txt"
done
cat "$out/timeline.txt"
tail -1 "$out/bench.log" | cut -c1-300
rm -f "$out/run_kernel_trace.csv"
