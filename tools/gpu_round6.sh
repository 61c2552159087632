#!/bin/bash
# Round 6: the fp32 DawnNet fault (scratch growth under a staging scope) and the N > 1 path's exit (no torch in the ranks).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
NNC_MI355X_SYNC_TRACE=1 timeout 300 oracle/_ref/host_resnet_bench.gpu 8 32 1 1 32 dawn > gpurun_out/dawn_f32_trace.json 2> gpurun_out/dawn_f32_trace.err; echo "exit $?" >> gpurun_out/dawn_f32_trace.err
timeout 600 python bench.py --config cifar10-dawn-f32-bs512 --steps 5 --warmup 1 > gpurun_out/bench_cifar_f32.json 2> gpurun_out/bench_cifar_f32.err; echo "exit $?" >> gpurun_out/bench_cifar_f32.err
NNC_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-via-host --no-cpu-baseline > gpurun_out/bench_force_comm.json 2> gpurun_out/bench_force_comm.err; echo "exit $?" >> gpurun_out/bench_force_comm.err
timeout 600 python -m pytest tests/test_via_host.py tests/test_rccl_single.py -m gpu -q -p no:cacheprovider > gpurun_out/new_gpu_tests.log 2>&1
tail -n 6 gpurun_out/dawn_f32_trace.err; cut -c1-300 gpurun_out/bench_cifar_f32.json; tail -n 2 gpurun_out/bench_cifar_f32.err; cut -c1-400 gpurun_out/bench_force_comm.json; tail -n 3 gpurun_out/bench_force_comm.err; tail -3 gpurun_out/new_gpu_tests.log
