#!/bin/bash
# Pipeline throughput, config 5, config 4 re-run, the N > 1 code path on one GPU (communicator of one), new GPU-tier tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/jitter_bench.py > gpurun_out/jitter_bench.txt 2>&1; echo "exit $?" >> gpurun_out/jitter_bench.txt
timeout 600 python bench.py --config cifar10-dawn-f16-bs512 --steps 5 --warmup 1 > gpurun_out/bench_cifar_f16.json 2> gpurun_out/bench_cifar_f16.err; echo "exit $?" >> gpurun_out/bench_cifar_f16.err
timeout 600 python bench.py --config cifar10-dawn-f32-bs512 --steps 5 --warmup 1 > gpurun_out/bench_cifar_f32.json 2> gpurun_out/bench_cifar_f32.err; echo "exit $?" >> gpurun_out/bench_cifar_f32.err
timeout 600 python bench.py --config resnet50-nchw-bs256 --steps 3 --warmup 1 > gpurun_out/bench_resnet_f32_c.json 2> gpurun_out/bench_resnet_f32_c.err; echo "exit $?" >> gpurun_out/bench_resnet_f32_c.err
NNC_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-via-host --no-cpu-baseline > gpurun_out/bench_force_comm.json 2> gpurun_out/bench_force_comm.err; echo "exit $?" >> gpurun_out/bench_force_comm.err
timeout 600 python -m pytest tests/test_jitter.py tests/test_via_host.py tests/test_preproc.py tests/test_parity_fullsize.py -m gpu -q -p no:cacheprovider -k "jitter or dawnnet or filter or winograd_unit or cubic" > gpurun_out/new_gpu_tests.log 2>&1
cat gpurun_out/jitter_bench.txt; cut -c1-600 gpurun_out/bench_cifar_f16.json; cut -c1-300 gpurun_out/bench_cifar_f32.json; cut -c1-300 gpurun_out/bench_resnet_f32_c.json; cut -c1-900 gpurun_out/bench_force_comm.json; tail -n 3 gpurun_out/bench_cifar_f16.err gpurun_out/bench_force_comm.err; tail -3 gpurun_out/new_gpu_tests.log
