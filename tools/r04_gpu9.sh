#!/bin/bash
# round 4, GPU call 9: pooling on NCHW rows (A / B inside one box), parity of the pool / conv changes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_half.py tests/test_peephole.py tests/test_via_host.py -m gpu -q -x -p no:cacheprovider -k "pool or conv or via or backward_pairs or vector" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sel.log; tail -3 gpurun_out/pytest_sel.log
for rep in 1 2; do
for cfg in resnet50-nchw-bs256 resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512; do
  for v in 0 1; do
    NNC_MI355X_POOL_ROWS=$v timeout 600 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/pool_ab_${cfg}_$v.json 2> gpurun_out/pool_ab_${cfg}_$v.err
    python -c "
import json
d=json.load(open('gpurun_out/pool_ab_${cfg}_$v.json')); print('$cfg pool_rows=$v rep=$rep', d['value'], d['ms_per_step'])"
  done
done
done
