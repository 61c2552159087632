#!/bin/bash
# round 4, GPU call 6: the vector epilogue -- parity subset, the 1x1 sweep again, configs 4 / 4-f16 / 5 and the default line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_half.py tests/test_conv_random.py tests/test_resnet_block.py tests/test_peephole.py tests/test_via_host.py tests/test_parity_fullsize.py -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sel.log; tail -4 gpurun_out/pytest_sel.log
timeout 400 python tools/conv1x1_bench.py 256 f32 > gpurun_out/conv1x1_f32.txt 2>&1; timeout 400 python tools/conv1x1_bench.py 256 f16 > gpurun_out/conv1x1_f16.txt 2>&1
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
