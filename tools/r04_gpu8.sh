#!/bin/bash
# round 4, GPU call 8: A / B inside one box -- epilogue forms (0 scalar, 2 row vectors, 1 + planar) on configs 4-f16 / 5 / 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512 resnet50-nchw-bs256; do
  for v in 0 2 1; do
    NNC_MI355X_GEMM_VEC_EPILOGUE=$v timeout 600 python bench.py --config $cfg --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_${cfg}_$v.json 2> gpurun_out/ab_${cfg}_$v.err
    python -c "
import json
d=json.load(open('gpurun_out/ab_${cfg}_$v.json')); print('$cfg vec=$v rep=$rep', d['value'], d['ms_per_step'])"
  done
done
done
