#!/bin/bash
# Per-kernel timing of a bench.py run under rocprofv3 (run on the GPU box): tools/gpu_profile.sh <name> [bench.py args...]
# Writes gpurun_out/<name>/ (kernel stats, trace) and prints the top kernels.
set -eo pipefail
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$name"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$root/bench.py" --no-all-configs "$@" > "$out/bench.log" 2>&1
cd "$root"
python3 tools/kernel_stats_top.py "$out/run_kernel_stats.csv" 13
tail -1 "$out/bench.log" | cut -c1-400
