#!/bin/bash
# round 4, last GPU call: what the driver runs -- the GPU tier, smoke(), the default bench line as the driver calls it -- plus configs 4 / 4-f16 / 5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err; echo "bench exit $?"; cut -c1-250 gpurun_out/bench_driver.json
for cfg in resnet50-nchw-bs256 resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512; do
  timeout 900 python bench.py --config $cfg --steps 8 --warmup 2 > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err; echo "exit $?" >> gpurun_out/bench_$cfg.err
  python -c "
import json
d=json.load(open('gpurun_out/bench_$cfg.json')); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['frac'], (d.get('cpu_baseline') or {}).get('value'), d['config'].get('oracle_gate'))"
done
