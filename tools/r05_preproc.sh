#!/bin/bash
# round 5: the 8-bit area resample, old kernel (NNC_MI355X_RESAMPLE_ROWS=0) vs the row-per-workgroup kernel; kernel trace + HBM counters of the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_preproc.py tests/test_jitter.py tests/test_dataframe_binding.py tests/test_pool_alloc.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_preproc.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/pytest_preproc.log
for v in 0 1; do echo "== NNC_MI355X_RESAMPLE_ROWS=$v"; NNC_MI355X_RESAMPLE_ROWS=$v timeout 300 python tools/preproc_bench.py; done > gpurun_out/preproc_bench.txt 2>&1
cat gpurun_out/preproc_bench.txt
rm -rf gpurun_out/prof_preproc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_preproc" -o t -- python "$OLDPWD/tools/preproc_bench.py" > "$OLDPWD/gpurun_out/prof_preproc.log" 2>&1)
find gpurun_out/prof_preproc -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/preproc_kernel_stats.md; done
head -8 gpurun_out/preproc_kernel_stats.md
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_preproc_$c
  (cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OLDPWD/gpurun_out/pmc_preproc_$c" -o pmc -- python "$OLDPWD/tools/preproc_bench.py" > "$OLDPWD/gpurun_out/pmc_preproc_$c.log" 2>&1)
done
python - <<'PY'
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for f in glob.glob("gpurun_out/pmc_preproc_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            if "area_8u_rows" in row["Kernel_Name"] and row["Counter_Name"] == c:
                tot += float(row["Counter_Value"]); n += 1
    print(c, "per launch of area_8u_rows_kernel (raw counter units, KB):", tot / n if n else None, "launches", n)
PY
find gpurun_out -name "*kernel_trace*" -size +8M -delete; find gpurun_out -name "*.db" -size +8M -delete
