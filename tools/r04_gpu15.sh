#!/bin/bash
# round 4, closing GPU call: the LSTM rows through the reference's own model API on the MI355X, the ABI / smoke checks after the registry grew to 130 rows,
# a kernel trace of the LSTM timing script, and the driver's own bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_via_host.py tests/test_abi.py tests/test_lstm.py -m gpu -q -p no:cacheprovider -k "lstm or abi" > gpurun_out/pytest_lstm2.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_lstm2.log; tail -4 gpurun_out/pytest_lstm2.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
rm -rf gpurun_out/prof_lstm
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_lstm" -o t -- python "$OLDPWD/tools/lstm_bench.py" > "$OLDPWD/gpurun_out/prof_lstm.log" 2>&1; echo "prof exit $?" >> "$OLDPWD/gpurun_out/prof_lstm.log")
find gpurun_out/prof_lstm -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_lstm.md; done
find gpurun_out/prof_lstm -name "*kernel_trace*" -size +20M -delete; find gpurun_out/prof_lstm -name "*.db" -size +20M -delete
head -14 gpurun_out/kernel_stats_lstm.md | cut -c1-200
timeout 200 python bench.py --steps 6 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err; cut -c1-400 gpurun_out/bench.json; tail -2 gpurun_out/bench.err
