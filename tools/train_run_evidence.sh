#!/bin/bash
# A short training run from the command line on the GPU box with --profile; leaves the learning curve and the GPU-actor
# phase summary under gpurun_out/r3train/ (tools/learning_curve.py, analyze_profiles.py).
set -eo pipefail
mkdir -p gpurun_out/r3train

( while true; do sleep 50; echo "[tick] $(date +%T)"; done ) &
TICK=$!
timeout -k 10 900 python -u -m alphatriangle_amd train --games-per-gpu 1024 --simulations 64 --max-steps 4000 --run-name r3run --profile > gpurun_out/r3train/train.log 2>&1 || { kill $TICK; tail -20 gpurun_out/r3train/train.log; exit 1; }
kill $TICK
f=$(find . -name metrics.jsonl -path "*r3run*" | head -1)
echo "metrics: $f"
python tools/learning_curve.py "$f" gpurun_out/r3train/learning_curve.csv 500
cat gpurun_out/r3train/learning_curve.csv | cut -c1-200
g=$(find . -name "gpu_actor_rank0.jsonl" -path "*r3run*" | head -1)
[ -n "$g" ] && python analyze_profiles.py "$g" > gpurun_out/r3train/gpu_actor_profile.txt 2>&1 && tail -8 gpurun_out/r3train/gpu_actor_profile.txt
grep -i "steps/sec\|Training finished\|Final\|error" gpurun_out/r3train/train.log | tail -5
