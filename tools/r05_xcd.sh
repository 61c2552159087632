#!/bin/bash
# round 5: the XCD-contiguous block mapping of the halo-overlap kernels (nnc_xcd_block: wino_input_kernel, pool_forw_v4 / pool_back_v4): per-kernel time of the
# batch-256 step + the L2 -> fabric read requests per launch; compare with profiles/r05_v7_rocprofv3_kernel_stats.md / r05_v7_pmc_traffic.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
tools/gpu_round.sh prof > gpurun_out/xcd_prof.log 2>&1
grep -E "wino_input_kernel|pool_forw_v4|pool_back_v4|wino_fused_kernel<4, 4, 0, false|mfma_gemm_f32_kernel<nnc::BufMatLoader<true>, nnc::BufMatLoader<true>, nnc::EpiStore" gpurun_out/kernel_stats_vgg.md
PMC_BATCH=256 PMC_GROUPS="rdsize" tools/pmc_pass.sh > gpurun_out/pmc_pass.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(set)
for f in glob.glob("gpurun_out/pmc/rdsize/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-60:]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k in sorted(acc):
    if any(s in k for s in ("wino_input", "pool_", "wino_fused_kernel", "wino_outgrad", "wino_output")):
        print("%-62s launches %3d  reads %.1f MB per launch" % (k, len(n[k]), 128 * acc[k]["TCC_EA0_RDREQ_128B_sum"] / len(n[k]) / 1e6))
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-extra-configs --no-cpu-baseline --no-alt-leg --no-via-host 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('value', d['value'], d['ms_per_step'])"
