#!/bin/bash
# round 5: the fused Winograd kernel's store cache policies (tools/wf5_probe.cpp, WF5_POLICY=1) -- time without the profiler, then L2 -> fabric read / write requests per launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp WF5_POLICY=1
OUT=$PWD/gpurun_out/policy
rm -rf $OUT; mkdir -p $OUT
IFS=";" read -ra SH <<< "${SHAPES:-256 224 64 64;256 112 64 128;256 56 128 256;256 56 256 256}"
for shape in "${SH[@]}"; do
	tag=$(echo $shape | tr ' ' _)
	tools/bin/wf5_probe $shape | tee $OUT/time_$tag.txt
	(cd /tmp && timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/rd_$tag -o pmc -- $OLDPWD/tools/bin/wf5_probe $shape > $OUT/rd_$tag.log 2>&1)
done
python - <<'PY' | tee gpurun_out/policy_summary.txt
import csv, glob, os
from collections import defaultdict
for d in sorted(glob.glob("gpurun_out/policy/rd_*")):
    if not os.path.isdir(d): continue
    acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "wino_fused_kernel" not in row["Kernel_Name"]: continue
            k = row["Kernel_Name"].split("(")[0].replace("void nnc::", "")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    print("==", d)
    for k in sorted(acc):
        c = acc[k]; L = len(n[k])
        print("%-50s launches %d  reads %.1f MB  writes %.1f MB per launch (write requests: %.3g, of them 64-byte %.3g)" % (k, L, 128 * c["TCC_EA0_RDREQ_128B_sum"] / L / 1e6,
            (64 * c["TCC_EA0_WRREQ_64B_sum"] + 32 * (c["TCC_EA0_WRREQ_sum"] - c["TCC_EA0_WRREQ_64B_sum"])) / L / 1e6, c["TCC_EA0_WRREQ_sum"] / L, c["TCC_EA0_WRREQ_64B_sum"] / L))
PY
find gpurun_out/policy -name "*kernel_trace*" -size +4M -delete
