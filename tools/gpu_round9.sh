#!/bin/bash
# Round 9: grid cap of the grid-stride kernels (none vs 8 workgroups per CU) on whole steps; element-wise rates.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ew_bw_bench.py > gpurun_out/ew_bw_bench4.txt 2>&1; tail -4 gpurun_out/ew_bw_bench4.txt
for cap in 8 0; do
  NNC_MI355X_GRID_WG_PER_CU=$cap timeout 600 python bench.py --steps 6 --warmup 2 --no-via-host --no-cpu-baseline > gpurun_out/bench_cap$cap.json 2> gpurun_out/bench_cap$cap.err
  echo "cap $cap: $(cut -c1-200 gpurun_out/bench_cap$cap.json)"
  for c in cifar10-dawn-f16-bs512 resnet50-nchw-bs256; do
    NNC_MI355X_GRID_WG_PER_CU=$cap timeout 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/bench_${c}_cap$cap.json 2> gpurun_out/bench_${c}_cap$cap.err
    echo "cap $cap: $(cut -c1-200 gpurun_out/bench_${c}_cap$cap.json)"
  done
done
