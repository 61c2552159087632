#!/bin/bash
# round 4, GPU call 12: the half kernel's 16-byte chunks (A / B inside one box) + parity subset
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_half.py tests/test_parity_ops.py -m gpu -q -x -p no:cacheprovider -k "half or conv or gemm or vector" > gpurun_out/pytest_sel.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_sel.log; tail -3 gpurun_out/pytest_sel.log
for rep in 1 2; do
for cfg in resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512; do
  for v in 0 1; do
    NNC_MI355X_GEMM_HALF_CHUNK8=$v timeout 600 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/c8_ab_${cfg}_$v.json 2> gpurun_out/c8_ab_${cfg}_$v.err
    python -c "
import json
d=json.load(open('gpurun_out/c8_ab_${cfg}_$v.json')); print('$cfg chunk8=$v rep=$rep', d['value'], d['ms_per_step'], d['roofline_f16_contractions']['all_f16_contractions'])"
  done
done
done
timeout 600 python tools/conv_half_bench.py > gpurun_out/conv_half_bench_c8.txt 2>&1; cut -c1-100 gpurun_out/conv_half_bench_c8.txt
