#!/bin/bash
# Round 10: after the pooling-gradient change -- pooling / step parity on the GPU, the bench line, the kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_vgg_step.py tests/test_parity_fullsize.py tests/test_half.py -m gpu -q -p no:cacheprovider -k "pool or step or native" > gpurun_out/pool_gpu_tests.log 2>&1; tail -2 gpurun_out/pool_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py --steps 6 --warmup 2 --records gpurun_out/records.txt > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o vgg -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-via-host > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1)
find gpurun_out/prof -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats.md; done
rm -rf gpurun_out/prof
cut -c1-330 gpurun_out/bench.log; tail -2 gpurun_out/bench.err; head -16 gpurun_out/kernel_stats.md | cut -c1-150
