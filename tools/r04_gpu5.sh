#!/bin/bash
# round 4, GPU call 5: per-launch records of one step of configs 4 / 4-f16 / 5 (which layer runs at which rate)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
cd oracle/_ref
HOST_BENCH_RECORDS=../../gpurun_out/records_resnet50_f32.txt timeout 300 ./host_resnet_bench.gpu 256 224 2 1 32 full > ../../gpurun_out/rec_f32.json 2> ../../gpurun_out/rec_f32.err
HOST_BENCH_RECORDS=../../gpurun_out/records_resnet50_f16.txt timeout 300 ./host_resnet_bench.gpu 256 224 2 1 16 full > ../../gpurun_out/rec_f16.json 2> ../../gpurun_out/rec_f16.err
HOST_BENCH_RECORDS=../../gpurun_out/records_dawn_f16.txt timeout 300 ./host_resnet_bench.gpu 512 32 2 1 16 dawn > ../../gpurun_out/rec_dawn.json 2> ../../gpurun_out/rec_dawn.err
wc -l ../../gpurun_out/records_*.txt
