#!/bin/bash
# Round 19: the look-ahead on the GPU -- parity suites, then the step through the reference host with and without it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_peephole.py tests/test_parity_ops.py tests/test_vgg_step.py tests/test_via_host.py -m gpu -q -p no:cacheprovider -x > gpurun_out/round19_tests.log 2>&1; tail -2 gpurun_out/round19_tests.log
NNC_MI355X_PEEPHOLE_STATS=1 timeout 300 oracle/_ref/host_vgg_bench.gpu 256 225 6 2 2>&1 | tail -2 | cut -c1-400
NNC_MI355X_PEEPHOLE=0 timeout 300 oracle/_ref/host_vgg_bench.gpu 256 225 6 2 2>&1 | tail -1 | cut -c1-300
