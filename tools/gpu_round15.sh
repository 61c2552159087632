#!/bin/bash
# Round 15: first-layer forward with the next group's loads issued before the stores (counted wait) -- parity and rate.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_fullsize.py tests/test_parity_ops.py tests/test_vgg_step.py tests/test_conv_random.py -m gpu -q -p no:cacheprovider -x > gpurun_out/round15_tests.log 2>&1; tail -2 gpurun_out/round15_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 6 --warmup 2 --no-via-host --no-cpu-baseline > gpurun_out/bench_r15.json 2> gpurun_out/bench_r15.err
echo "$(cut -c56-120 gpurun_out/bench_r15.json) $(grep -o '"nnc::conv3x3_c3_fwd_kernel": {[^}]*}' gpurun_out/bench_r15.json) $(grep -o '"nnc::conv3x3_c3_wgrad_kernel": {[^}]*}' gpurun_out/bench_r15.json)"
