#!/bin/bash
# round 4, last GPU call: the LSTM rows on the MI355X (parity tests, the reference's own int cases, a timing line)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_lstm.py tests/test_ref_int.py tests/test_via_host.py -m gpu -q -p no:cacheprovider -k "lstm or LSTM" > gpurun_out/pytest_lstm.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_lstm.log; tail -5 gpurun_out/pytest_lstm.log
timeout 100 python tools/lstm_bench.py > gpurun_out/lstm_bench.txt 2>&1; echo "exit $?" >> gpurun_out/lstm_bench.txt; cat gpurun_out/lstm_bench.txt
