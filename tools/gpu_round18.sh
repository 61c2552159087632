#!/bin/bash
# Round 18: kernel-time breakdown of the step with ReLU folded into its neighbours.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof18
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof18 -o r18 --output-format csv -- python bench.py --steps 4 --warmup 2 --no-via-host --no-cpu-baseline --no-alt-leg > gpurun_out/bench_r18.json 2> gpurun_out/bench_r18.err
f=$(find gpurun_out/prof18 -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print("%-90s %5d %9.3f ms %5.1f%%" % (r["Name"][:90], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("total", tot / 1e6)
PY
find gpurun_out/prof18 -name '*kernel_trace.csv' -delete
