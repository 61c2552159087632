#!/bin/bash
# round 4, GPU call 11: PMC traffic (fetch / write passes) of configs 4-f16 and 5 -> profiles/pmc_traffic_<config>.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512; do
  rm -rf gpurun_out/pmc
  PMC_BATCH=$cfg PMC_GROUPS="fetch write" PMC_BENCH_ARGS="--config $cfg --steps 1 --warmup 1 --no-cpu-baseline" timeout 900 tools/pmc_pass.sh > gpurun_out/pmc_$cfg.log 2>&1
  cp gpurun_out/pmc_traffic_bs$cfg.json gpurun_out/pmc_traffic_$cfg.json 2>/dev/null
  cp gpurun_out/pmc_traffic.txt gpurun_out/pmc_traffic_$cfg.txt 2>/dev/null
  grep -i "bn_" gpurun_out/pmc_traffic_$cfg.txt | head -4
done
rm -rf gpurun_out/pmc
