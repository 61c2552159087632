#!/bin/bash
# round 4, GPU call 4 (after the container was re-created): the whole GPU tier, the dataframe stage's throughput, fresh kernel stats of configs 4 / 4-f16 / 5,
# the default bench line, PMC traffic of config 4's batch-norm / transpose kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -8 gpurun_out/pytest_gpu.log
(cd oracle/_ref && for a in "4096 256 imagenet 32" "4096 256 imagenet 16" "16384 512 cifar 16"; do timeout 300 ./host_dataframe_test.gpu $a bench; done) > gpurun_out/dataframe_bench.txt 2>&1; cat gpurun_out/dataframe_bench.txt
PROF_TIMEOUT=600 tools/gpu_round.sh prof:resnet50-nchw-bs256 prof:resnet50-nchw-bs256-f16 prof:cifar10-dawn-f16-bs512 > gpurun_out/prof_round.log 2>&1; tail -5 gpurun_out/prof_round.log
STEPS=10 BENCH_ARGS="--no-cpu-baseline" tools/gpu_round.sh bench > gpurun_out/bench_round.log 2>&1; cut -c1-600 gpurun_out/bench.json
PMC_BATCH=resnet50 PMC_GROUPS="fetch write" PMC_BENCH_ARGS="--config resnet50-nchw-bs256 --steps 1 --warmup 1 --no-cpu-baseline" timeout 900 tools/pmc_pass.sh > gpurun_out/pmc_resnet.log 2>&1; grep -i "bn_cluster\|transpose" gpurun_out/pmc_traffic.txt | head
