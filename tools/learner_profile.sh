#!/bin/bash
# Learner-only step under rocprofv3 (run on the GPU box): tools/learner_profile.sh <name> [learner_bench.py args...]
# Writes gpurun_out/<name>/ (kernel stats + ordered trace), prints the top kernels and the launch count of one step.
set -eo pipefail
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$name"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$root/bench/learner_bench.py" "$@" > "$out/bench.log" 2>&1
cd "$root"
python3 tools/kernel_stats_top.py "$out/run_kernel_stats.csv" 40 > "$out/top.txt"
python3 tools/step_sequence.py "$out/run_kernel_trace.csv" > "$out/step_sequence.txt"
tail -1 "$out/step_sequence.txt"
tail -1 "$out/bench.log" | cut -c1-300
