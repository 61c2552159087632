#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench line, rocprofv3 kernel trace. Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-4}
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; free -g | head -2) > gpurun_out/box.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps $STEPS --warmup 1 --records gpurun_out/records.txt > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
if [ -z "$NO_PROF" ]; then
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o vgg -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-via-host --no-alt-leg > "$OLDPWD/gpurun_out/prof_bench.log" 2>&1; echo "prof exit $?" >> "$OLDPWD/gpurun_out/prof_bench.log")
  find gpurun_out/prof -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats.md; done
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
fi
tail -5 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/smoke.log; cat gpurun_out/bench.log; tail -3 gpurun_out/bench.err
