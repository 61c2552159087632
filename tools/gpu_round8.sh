#!/bin/bash
# Round 8: native half rows + fold kernels -- configs 4 and 5 in both precisions, the new GPU-tier half tests, kernel trace of dawn f16.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
timeout 600 python -m pytest tests/test_half.py tests/test_parity_ops.py tests/test_resnet_block.py -m gpu -q -p no:cacheprovider -x > gpurun_out/half_gpu_tests.log 2>&1
tail -3 gpurun_out/half_gpu_tests.log
for c in cifar10-dawn-f16-bs512 cifar10-dawn-f32-bs512 resnet50-nchw-bs256 resnet50-nchw-bs256-f16; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "exit $?" >> gpurun_out/bench_$c.err
  cut -c1-260 gpurun_out/bench_$c.json; tail -n 2 gpurun_out/bench_$c.err
done
rm -rf gpurun_out/prof_dawn16
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_dawn16" -o dawn -- "$R/oracle/_ref/host_resnet_bench.gpu" 512 32 4 2 16 dawn > "$R/gpurun_out/prof_dawn16.log" 2>&1)
find gpurun_out/prof_dawn16 -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_dawn16.md; done
rm -rf gpurun_out/prof_dawn16
head -24 gpurun_out/kernel_stats_dawn16.md | cut -c1-160
