#!/bin/bash
# ResNet-50 (config 4) bench lines + rocprofv3 kernel table of the fp32 step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-a}
timeout 600 python bench.py --config resnet50-nchw-bs256 --steps 3 --warmup 1 > gpurun_out/bench_resnet_f32_$TAG.json 2> gpurun_out/bench_resnet_f32_$TAG.err; echo "exit $?" >> gpurun_out/bench_resnet_f32_$TAG.err
timeout 600 python bench.py --config resnet50-nchw-bs256-f16 --steps 3 --warmup 1 > gpurun_out/bench_resnet_f16_$TAG.json 2> gpurun_out/bench_resnet_f16_$TAG.err; echo "exit $?" >> gpurun_out/bench_resnet_f16_$TAG.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_resnet_$TAG" -o resnet -- "$OLDPWD/oracle/_ref/host_resnet_bench.gpu" 256 224 2 1 32 > "$OLDPWD/gpurun_out/prof_resnet_$TAG.log" 2>&1; echo "prof exit $?" >> "$OLDPWD/gpurun_out/prof_resnet_$TAG.log")
find gpurun_out/prof_resnet_$TAG -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/resnet_kernel_stats_$TAG.md; done
find gpurun_out/prof_resnet_$TAG -name "*.db" -size +40M -delete
cut -c1-900 gpurun_out/bench_resnet_f32_$TAG.json; cut -c1-400 gpurun_out/bench_resnet_f16_$TAG.json; tail -n 2 gpurun_out/bench_resnet_f32_$TAG.err gpurun_out/bench_resnet_f16_$TAG.err; head -22 gpurun_out/resnet_kernel_stats_$TAG.md
