#!/bin/bash
# round 5: counter calibration (tools/fetch_calib.cpp); STEP=1 adds the size-resolved read-request counters over the batch-256 VGG-D step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/calib
rm -rf $OUT; mkdir -p $OUT
tools/bin/fetch_calib > $OUT/run.txt 2>&1; cat $OUT/run.txt
(cd /tmp && timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/rd -o pmc -- $OLDPWD/tools/bin/fetch_calib > $OUT/rd.log 2>&1)
(cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- $OLDPWD/tools/bin/fetch_calib > $OUT/fetch.log 2>&1)
(cd /tmp && timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $OUT/hit -o pmc -- $OLDPWD/tools/bin/fetch_calib > $OUT/hit.log 2>&1)
[ -n "$STEP" ] && (cd /tmp && timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $OUT/step -o pmc -- python $OLDPWD/bench.py --steps 1 --warmup 0 --batch 256 --no-cpu-baseline --no-via-host --no-alt-leg --no-extra-configs > $OUT/step.log 2>&1)
python - <<'PY' | tee gpurun_out/calib_summary.txt
import csv, glob
from collections import defaultdict
def table(d):
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in glob.glob("gpurun_out/calib/%s/**/*counter_collection.csv" % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-70:]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    return acc, n
for d in ("rd", "fetch", "hit", "step"):
    acc, n = table(d)
    print("==", d)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:14]:
        c = acc[k]; L = len(n[k])
        if "FETCH_SIZE" in c:
            print("%-72s launches %3d  FETCH_SIZE %.1f MB per launch (KiB units)" % (k, L, c["FETCH_SIZE"] * 1024 / L / 1e6))
        elif "TCC_HIT_sum" in c:
            print("%-72s launches %3d  per launch: req %.4g read %.4g hit %.4g miss %.4g" % (k, L, c["TCC_REQ_sum"] / L, c["TCC_READ_sum"] / L, c["TCC_HIT_sum"] / L, c["TCC_MISS_sum"] / L))
        else:
            r, r32, r64, r128 = c["TCC_EA0_RDREQ_sum"], c["TCC_EA0_RDREQ_32B_sum"], c["TCC_EA0_RDREQ_64B_sum"], c["TCC_EA0_RDREQ_128B_sum"]
            print("%-72s launches %3d  requests %.3g (32B %.3g, 64B %.3g, 128B %.3g, other %.3g)  => %.1f MB per launch" % (k, L, r, r32, r64, r128, r - r32 - r64 - r128, (32 * r32 + 64 * r64 + 128 * r128 + 64 * (r - r32 - r64 - r128)) / L / 1e6))
PY
find gpurun_out/calib -name "*kernel_trace*" -size +4M -delete
