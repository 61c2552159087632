#!/bin/bash
# Samples the GPU's shader clock and socket power while the VGG-D step runs (is the contraction clock- / power-limited?).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
(rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power|fclk|mclk") > gpurun_out/clock_idle.txt
python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-extra-configs > gpurun_out/clock_bench.log 2>&1 &
BPID=$!
: > gpurun_out/clock_samples.txt
for i in $(seq 1 60); do
  sleep 0.5
  kill -0 $BPID 2>/dev/null || break
  (rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Socket Power|Average Graphics" | tr '\n' ' '; echo) >> gpurun_out/clock_samples.txt
done
wait $BPID
echo "idle:"; cat gpurun_out/clock_idle.txt
echo "under load (last 25 samples):"; tail -25 gpurun_out/clock_samples.txt | cut -c1-200
