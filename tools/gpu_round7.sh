#!/bin/bash
# Round 7: where the DawnNet step goes (fp16 vs fp32) -- kernel traces of both.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
for p in 16 32; do
  rm -rf gpurun_out/prof_dawn$p
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_dawn$p" -o dawn -- "$R/oracle/_ref/host_resnet_bench.gpu" 512 32 4 2 $p dawn > "$R/gpurun_out/prof_dawn$p.log" 2>&1; echo "prof exit $?" >> "$R/gpurun_out/prof_dawn$p.log")
  find gpurun_out/prof_dawn$p -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_dawn$p.md; done
  rm -rf gpurun_out/prof_dawn$p
  head -32 gpurun_out/kernel_stats_dawn$p.md | cut -c1-200
done
