#!/bin/bash
# The headline configuration as a training run from the command line on the GPU box (4096 games, 400 simulations per move,
# fp32, one learner step per move): sustained moves/s over games of all ages.  Leaves the learning curve under gpurun_out/r3headline/.
set -eo pipefail
mkdir -p gpurun_out/r3headline
( while true; do sleep 50; echo "[tick] $(date +%T)"; done ) &
TICK=$!
t0=$(date +%s)
timeout -k 10 1000 python -u -m alphatriangle_amd train --max-steps "${1:-900}" --run-name r3headline > gpurun_out/r3headline/train.log 2>&1 || { kill $TICK; tail -20 gpurun_out/r3headline/train.log; exit 1; }
t1=$(date +%s)
kill $TICK
f=$(find . -name metrics.jsonl -path "*r3headline*" | head -1)
python tools/learning_curve.py "$f" gpurun_out/r3headline/learning_curve_headline_config.csv 100
echo "wall seconds: $((t1 - t0))" | tee gpurun_out/r3headline/wall.txt
cut -c1-160 gpurun_out/r3headline/learning_curve_headline_config.csv
grep "Progress: Step" gpurun_out/r3headline/train.log | tail -2 | sed 's/\x1b\[[0-9;]*m//g' | cut -c1-300
