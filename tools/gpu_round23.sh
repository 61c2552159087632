#!/bin/bash
# Round 23: kernel trace of the step driven by the UNMODIFIED reference host (look-ahead on): which ReLU kernels are left.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof23
(cd /tmp && NNC_MI355X_PEEPHOLE_STATS=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof23" -o host -- "$OLDPWD/oracle/_ref/host_vgg_bench.gpu" 256 225 4 1 > "$OLDPWD/gpurun_out/prof23_host.log" 2>&1; echo "exit $?")
find gpurun_out/prof23 -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/via_host_kernel_stats.md; done
find gpurun_out/prof23 -name "*kernel_trace*" -size +20M -delete
grep -i "look-ahead" gpurun_out/prof23_host.log; grep -E "Relu|relu|pool_back|wino_fused_kernel<4, 4" gpurun_out/via_host_kernel_stats.md | cut -c1-140
