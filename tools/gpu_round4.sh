#!/bin/bash
# Full visit: gpu test tier, smoke, default bench line + records, rocprofv3 kernel stats, marker-trace check, PMC traffic passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
NO_PROF= STEPS=8 bash tools/gpu_round.sh > gpurun_out/round.log 2>&1
(cd /tmp && NNC_MI355X_MARKERS=1 timeout 300 rocprofv3 --marker-trace --kernel-trace -d "$OLDPWD/gpurun_out/prof_markers" -o m -- python -c "import sys; sys.path.insert(0, '$OLDPWD'); import __graft_entry__ as g; g.smoke()" > "$OLDPWD/gpurun_out/markers.log" 2>&1; echo "exit $?" >> "$OLDPWD/gpurun_out/markers.log")
find gpurun_out/prof_markers -name "*_results.db" | head -1 | while read f; do python - "$f" > gpurun_out/markers_summary.txt 2>&1 <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, count(*) from regions group by name order by 2 desc"))
print("marker regions (roctx ranges) by name:")
for n, k in rows[:40]: print("%6d  %s" % (k, n))
PY
done
PMC_GROUPS="fetch write" PMC_BATCH=256 timeout 900 bash tools/pmc_pass.sh > gpurun_out/pmc_pass.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; cut -c1-1500 gpurun_out/bench.log; tail -2 gpurun_out/bench.err; head -30 gpurun_out/markers_summary.txt; tail -30 gpurun_out/pmc_traffic.txt
