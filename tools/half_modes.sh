#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_half.py -q -m gpu -k "buffer_kernel_shapes" 2>&1 | tail -2
for m in 1 2 4; do echo "== GEMM_BUFFER_LOADS=$m"; NNC_MI355X_GEMM_BUFFER_LOADS=$m timeout 300 python tools/half_bench.py 2>&1 | grep -E "gemm f16"; done
STEPS=4 tools/gpu_round.sh config:resnet50-nchw-bs256-f16 2>&1 | cut -c1-200 | tail -2
