#!/bin/bash
# round 5: SQ counters of ONE tool's kernels.  usage: tools/r05_pmc_kernel.sh <kernel substring> <command...>   -> gpurun_out/pmc_<substring>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
K=$1; shift
OUT=$PWD/gpurun_out/pmck
rm -rf $OUT; mkdir -p $OUT
pass() { n=$1; shift; (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -o pmc -- "${CMD[@]}" > $OUT/$n.log 2>&1); }
CMD=("$@")
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
pass d TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
K="$K" python - <<'PY' > gpurun_out/pmc_kernel.txt
import csv, glob, os
from collections import defaultdict
k = os.environ["K"]
acc, n = defaultdict(float), defaultdict(int)
for f in glob.glob("gpurun_out/pmck/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if k in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for c in sorted(acc):
    print("%-32s %16.0f per launch (%d launches)" % (c, acc[c] / n[c], n[c]))
PY
cat gpurun_out/pmc_kernel.txt
find gpurun_out/pmck -name "*kernel_trace*" -size +4M -delete
