#!/bin/bash
# Second GPU visit shape: configs 2 / 4 bench lines, half-precision contraction rates, rocprofv3 of the ResNet-50 step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/half_bench.py > gpurun_out/half_bench.txt 2>&1; echo "half_bench exit $?" >> gpurun_out/half_bench.txt
timeout 600 python bench.py --config resnet50-nchw-bs256 --steps 3 --warmup 1 > gpurun_out/bench_resnet_f32.json 2> gpurun_out/bench_resnet_f32.err; echo "exit $?" >> gpurun_out/bench_resnet_f32.err
timeout 600 python bench.py --config resnet50-nchw-bs256-f16 --steps 3 --warmup 1 > gpurun_out/bench_resnet_f16.json 2> gpurun_out/bench_resnet_f16.err; echo "exit $?" >> gpurun_out/bench_resnet_f16.err
timeout 600 python bench.py --config vggd-fwd-bs64 --steps 8 --warmup 2 > gpurun_out/bench_vggd_fwd64.json 2> gpurun_out/bench_vggd_fwd64.err; echo "exit $?" >> gpurun_out/bench_vggd_fwd64.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_resnet" -o resnet -- "$OLDPWD/oracle/_ref/host_resnet_bench.gpu" 256 224 2 1 32 > "$OLDPWD/gpurun_out/prof_resnet.log" 2>&1; echo "prof exit $?" >> "$OLDPWD/gpurun_out/prof_resnet.log")
find gpurun_out/prof_resnet -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/resnet_kernel_stats.md; done
find gpurun_out/prof_resnet -name "*.db" -size +40M -delete
(cd oracle/_ref/int && for c in "cudnn forward convolution in half precision" "cudnn backward convolution in half precision"; do timeout 300 ./cudnn.gpu "$c" 2>&1 | tail -2; done) > gpurun_out/conv_half_cases.txt 2>&1
timeout 300 python -m pytest tests/test_via_host.py -m gpu -q -p no:cacheprovider > gpurun_out/via_host_gpu.log 2>&1
tail -12 gpurun_out/half_bench.txt; cat gpurun_out/bench_resnet_f32.json gpurun_out/bench_resnet_f16.json | cut -c1-1200; tail -2 gpurun_out/bench_resnet_f32.err gpurun_out/bench_resnet_f16.err gpurun_out/bench_vggd_fwd64.err; cut -c1-700 gpurun_out/bench_vggd_fwd64.json; cat gpurun_out/conv_half_cases.txt; tail -2 gpurun_out/via_host_gpu.log; head -14 gpurun_out/resnet_kernel_stats.md
