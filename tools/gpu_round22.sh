#!/bin/bash
# Round 22: stdout carries exactly one line (the JSON) also when RCCL prints its banner.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
NNC_BENCH_FORCE_COMM=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-via-host --no-alt-leg > gpurun_out/bench_r22_forced_comm.json 2> gpurun_out/bench_r22_forced_comm.err; echo "exit $? lines $(wc -l < gpurun_out/bench_r22_forced_comm.json)"; grep -c "RCCL version" gpurun_out/bench_r22_forced_comm.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 4 --warmup 1 > gpurun_out/bench_r22_torchrun1.json 2> gpurun_out/bench_r22_torchrun1.err; echo "exit $? lines $(wc -l < gpurun_out/bench_r22_torchrun1.json)"
timeout 300 python bench.py --config cifar10-dawn-f16-bs512 --no-cpu-baseline > gpurun_out/bench_r22_dawn.json 2> gpurun_out/bench_r22_dawn.err; echo "exit $? lines $(wc -l < gpurun_out/bench_r22_dawn.json)"
python - <<'PY'
import json
for f in ("bench_r22_forced_comm", "bench_r22_torchrun1", "bench_r22_dawn"):
    d = json.load(open("gpurun_out/%s.json" % f)); print(f, d["value"], d["unit"])
PY
