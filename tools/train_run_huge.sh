#!/bin/bash
# Short training run of the 20x30 preset (600 cells: wide build of the device engine for self-play, leaves through the PyTorch
# module, learner on the same GPU) on the GPU box: tools/train_run_huge.sh [max steps] [simulations] [extra train options].
# Log excerpt and learning curve go to gpurun_out/train_huge/.
set -eo pipefail
steps="${1:-300}"; sims="${2:-64}"; shift 2 || true
root="$(cd "$(dirname "$0")/.." && pwd)"
out="$root/gpurun_out/train_huge"
mkdir -p "$out"
cd "$root"
( while true; do sleep 60; echo "[tick] $(date +%T)" >> "$out/ticks.txt"; done ) &
TICK=$!
trap "kill $TICK 2>/dev/null || true" EXIT
timeout -k 10 500 python3 -u -m alphatriangle_amd train --preset huge --max-steps "$steps" --simulations "$sims" "$@" \
  --data-root "$out/data" --run-name huge --log-level INFO > "$out/train.log" 2>&1
grep -E "host engine|Determined|Initialized|GpuTreeEngine|GpuSelfPlayActor|leaf|Training finished|completed|steps/sec|Final|Error|error" "$out/train.log" | tail -25 > "$out/train_log_excerpt.txt" || true
python3 tools/learning_curve.py "$out/data/AlphaTriangle/runs/huge/metrics.jsonl" "$out/learning_curve_huge.csv" 25 || true
rm -rf "$out/data"
tail -5 "$out/train_log_excerpt.txt"
