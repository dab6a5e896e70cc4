#!/bin/bash
# Hardware counters of a short bench.py run (run on the GPU box): tools/gpu_pmc.sh <name> "<counters>" [bench.py args...]
# One rocprofv3 --pmc pass (at most 8 SQ counters); per-kernel per-dispatch averages go to gpurun_out/<name>/pmc.txt.
set -eo pipefail
name="$1"; counters="$2"; shift 2
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$name"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $counters --output-format csv -d "$out" -o p -- python3 "$root/bench.py" --no-all-configs "$@" > "$out/bench.log" 2>&1
cd "$root"
python3 tools/pmc_summary.py "$out/p_counter_collection.csv" > "$out/pmc.txt"
head -8 "$out/pmc.txt" | cut -c1-220
