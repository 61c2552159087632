#!/bin/bash
# Round 11: kernel traces of the ResNet-50 step (fp32 and f16) after the round's changes; attention / detection rows on the GPU.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$PWD"
timeout 600 python -m pytest tests/test_attention.py -m gpu -q -p no:cacheprovider > gpurun_out/attention_gpu.log 2>&1; tail -2 gpurun_out/attention_gpu.log
timeout 900 python tools/ref_int_tests.py run gpu nms roi_align compression cublas --match "" --timeout 300 > gpurun_out/ref_int_new_rows.txt 2>&1; grep -c PASS gpurun_out/ref_int_new_rows.txt; grep -v PASS gpurun_out/ref_int_new_rows.txt | head -10
for p in 32 16; do
  rm -rf gpurun_out/prof_rn$p
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_rn$p" -o rn -- "$R/oracle/_ref/host_resnet_bench.gpu" 256 224 3 1 $p > "$R/gpurun_out/prof_rn$p.log" 2>&1)
  find gpurun_out/prof_rn$p -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_resnet$p.md; done
  rm -rf gpurun_out/prof_rn$p
  head -26 gpurun_out/kernel_stats_resnet$p.md | cut -c1-170
done
