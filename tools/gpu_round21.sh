#!/bin/bash
# Round 21: new look-ahead tests on the GPU; bench.py's N > 1 code path (RCCL communicator of one, and under torch.distributed.run with one rank).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_peephole.py tests/test_rccl_single.py -m gpu -q -p no:cacheprovider > gpurun_out/round21_tests.log 2>&1; tail -2 gpurun_out/round21_tests.log
NNC_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-via-host > gpurun_out/bench_r21_forced_comm.json 2> gpurun_out/bench_r21_forced_comm.err; echo "exit $?"; cut -c1-330 gpurun_out/bench_r21_forced_comm.json; tail -2 gpurun_out/bench_r21_forced_comm.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-via-host > gpurun_out/bench_r21_torchrun1.json 2> gpurun_out/bench_r21_torchrun1.err; echo "exit $?"; cut -c1-200 gpurun_out/bench_r21_torchrun1.json; tail -2 gpurun_out/bench_r21_torchrun1.err
