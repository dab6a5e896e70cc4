#!/bin/bash
# Same-box series of bench.py runs (run on the GPU box): tools/bench_matrix.sh <name> "<args of run 1>" "<args of run 2>" ...
# One line per run goes to gpurun_out/<name>/matrix.txt: the arguments, moves/s, ms per step, share of leaf slots evaluated.
set -eo pipefail
name="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/$name"
mkdir -p "$out"
cd "$root"
for args in "$@"; do
  timeout -k 10 500 python3 -u bench.py --no-all-configs $args 2> "$out/last.err" | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$args |', d['value'], 'moves/s |', d['ms_per_step'], 'ms |', d.get('leaf_slots_needing_evaluation'), '|', d.get('trainer_samples_per_sec_unthrottled'))" | tee -a "$out/matrix.txt"
done
