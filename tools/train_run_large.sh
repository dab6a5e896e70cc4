#!/bin/bash
# The 12x21 preset (8 encoder layers) trained in fp32 from the command line on the GPU box: fp32-grade flat-tile / row-tiled engine
# kernels for the search, direct convolution kernels (16-tile boards) for the learner.  Learning curve under gpurun_out/r3large/.
set -eo pipefail
mkdir -p gpurun_out/r3large
( while true; do sleep 50; echo "[tick] $(date +%T)"; done ) &
TICK=$!
t0=$(date +%s)
timeout -k 10 1000 python -u -m alphatriangle_amd train --preset large --games-per-gpu 512 --simulations 64 --max-steps "${1:-1500}" --run-name r3large > gpurun_out/r3large/train.log 2>&1 || { kill $TICK; tail -20 gpurun_out/r3large/train.log; exit 1; }
t1=$(date +%s)
kill $TICK
f=$(find . -name metrics.jsonl -path "*r3large*" | head -1)
python tools/learning_curve.py "$f" gpurun_out/r3large/learning_curve_large_fp32.csv 250
echo "wall seconds: $((t1 - t0))" | tee gpurun_out/r3large/wall.txt
cut -c1-150 gpurun_out/r3large/learning_curve_large_fp32.csv
grep -i "evaluator\|engine" gpurun_out/r3large/train.log | head -5 | sed 's/\x1b\[[0-9;]*m//g' | cut -c1-200
