#!/bin/bash
# round 4, GPU call: thin transposes, cluster batch norm on 8 / 4-byte chunks, the process-per-GPU form at world 1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bn_cluster.py tests/test_parity_ops.py tests/test_half.py tests/test_bench_stdout.py tests/test_via_host.py -m gpu -q -x -p no:cacheprovider -k "cluster or norm or transform or transpose or process_per_gpu or nchw or reference_model_api" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sel.log; tail -5 gpurun_out/pytest_sel.log
timeout 600 python tools/bn_bench.py > gpurun_out/bn_bench.txt 2>&1; echo "exit $?" >> gpurun_out/bn_bench.txt; tail -18 gpurun_out/bn_bench.txt
for cfg in resnet50-nchw-bs256 resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512; do
  timeout 900 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$cfg.json 2> gpurun_out/bench_$cfg.err; echo "exit $?" >> gpurun_out/bench_$cfg.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$cfg.json"))
    print("$cfg", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d["config"].get("host_enqueue", {}).get("ms_per_step_median"))
except Exception as e:
    print("$cfg failed", e)
PY
  tail -2 gpurun_out/bench_$cfg.err
done
