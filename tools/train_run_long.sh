#!/bin/bash
# 16 000 learner steps from the command line on the GPU box (1024 games, 64 simulations per move, fp32): learning curve under
# gpurun_out/r3long/.
set -eo pipefail
mkdir -p gpurun_out/r3long
( while true; do sleep 50; echo "[tick] $(date +%T)"; done ) &
TICK=$!
t0=$(date +%s)
timeout -k 10 1100 python -u -m alphatriangle_amd train --games-per-gpu 1024 --simulations 64 --max-steps 16000 --run-name r3long > gpurun_out/r3long/train.log 2>&1 || { kill $TICK; tail -20 gpurun_out/r3long/train.log; exit 1; }
t1=$(date +%s)
kill $TICK
f=$(find . -name metrics.jsonl -path "*r3long*" | head -1)
python tools/learning_curve.py "$f" gpurun_out/r3long/learning_curve_16k.csv 1000
echo "wall seconds: $((t1 - t0))" | tee gpurun_out/r3long/wall.txt
cut -c1-150 gpurun_out/r3long/learning_curve_16k.csv
