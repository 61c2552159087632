#!/bin/bash
# round 4, GPU call 7: the whole GPU tier on the current tree + configs 4 / 4-f16 / 5 + the default line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log
for cfg in resnet50-nchw-bs256 resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512; do
  timeout 900 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err; echo "exit $?" >> gpurun_out/bench_$cfg.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$cfg.json"))
    print("$cfg", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
except Exception as e:
    print("$cfg failed", e)
PY
done
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --records gpurun_out/records.txt > gpurun_out/bench.json 2> gpurun_out/bench.err; cut -c1-300 gpurun_out/bench.json
