#!/bin/bash
# Round 13: stride-2 data gradient by parity classes -- parity on the GPU, ResNet-50 in both precisions, the trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_conv_random.py tests/test_resnet_block.py tests/test_jitter.py tests/test_via_host.py -m gpu -q -p no:cacheprovider > gpurun_out/round13_tests.log 2>&1; tail -2 gpurun_out/round13_tests.log
for c in resnet50-nchw-bs256 resnet50-nchw-bs256-f16; do
  timeout 600 python bench.py --config $c --steps 5 --warmup 1 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "exit $?" >> gpurun_out/bench_$c.err
  cut -c1-200 gpurun_out/bench_$c.json; tail -n 1 gpurun_out/bench_$c.err
done
rm -rf gpurun_out/prof_rn32
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_rn32" -o rn -- "$R/oracle/_ref/host_resnet_bench.gpu" 256 224 3 1 32 > "$R/gpurun_out/prof_rn32.log" 2>&1)
find gpurun_out/prof_rn32 -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_resnet32.md; done
rm -rf gpurun_out/prof_rn32
head -14 gpurun_out/kernel_stats_resnet32.md | cut -c1-170; grep -n "parity" gpurun_out/kernel_stats_resnet32.md | cut -c1-120
