#!/bin/bash
# round 5, re-entry session: ONE short call (10 GPU-minutes were left).  In order of what matters most: the new palettize rows on the real GPU (parity tests +
# the reference's own int cases through its unmodified host), their bandwidth, the LSTM rows kernel's tests / timing / trace, smoke.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
(rocminfo | grep -E "Marketing Name|gfx" | head -2; nproc) > gpurun_out/box.txt 2>&1
timeout 120 python -m pytest tests/test_palettize.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_palettize.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_palettize.log; tail -3 gpurun_out/pytest_gpu_palettize.log
(timeout 150 python tools/ref_int_tests.py run gpu palettize --timeout 60 --out gpurun_out/ref_int_palettize.txt | tail -2) &
(timeout 150 python tools/ref_int_tests.py run gpu cublas cudnn --timeout 120 --match palettize --out gpurun_out/ref_int_palettize_rows.txt | tail -4) &
timeout 120 python tools/palette_bench.py > gpurun_out/palette_bench.txt 2>&1; echo "exit $?" >> gpurun_out/palette_bench.txt; tail -26 gpurun_out/palette_bench.txt
wait
echo "=== $(( $(date +%s) - S )) s: lstm"
timeout 150 python -m pytest tests/test_lstm.py tests/test_pool_alloc.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_lstm.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu_lstm.log; tail -3 gpurun_out/pytest_gpu_lstm.log
timeout 120 python tools/lstm_bench.py > gpurun_out/lstm_bench.txt 2>&1; echo "exit $?" >> gpurun_out/lstm_bench.txt; cat gpurun_out/lstm_bench.txt
echo "=== $(( $(date +%s) - S )) s: imdb line + trace"
STEPS=5 tools/gpu_round.sh config:imdb-lstm-bs64 | cut -c1-900
PROF_TIMEOUT=150 tools/gpu_round.sh prof:imdb-lstm-bs64 | head -14
echo "=== $(( $(date +%s) - S )) s: smoke"
tools/gpu_round.sh smoke
echo "=== $(( $(date +%s) - S )) s: depalettize trace"
rm -rf gpurun_out/prof_pal; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_pal" -o t -- python "$OLDPWD/tools/palette_bench.py" > "$OLDPWD/gpurun_out/prof_pal.log" 2>&1)
find gpurun_out/prof_pal -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_palette.md; done
find gpurun_out/prof_pal -size +20M -delete
head -16 gpurun_out/kernel_stats_palette.md
echo "final3 total $(( $(date +%s) - S )) s"
