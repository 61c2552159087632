#!/bin/bash
# round 5, last call: the evidence set of the final tree (tools/r05_final.sh), the kernel traces of the other configurations, the GPU test tier, smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
S=$(date +%s)
tools/r05_final.sh
STEPS=3 tools/gpu_round.sh prof:vggd-fwd-bs64 prof:resnet50-nchw-bs256 prof:resnet50-nchw-bs256-f16 prof:cifar10-dawn-f16-bs512 prof:imdb-lstm-bs64 > gpurun_out/prof_configs.log 2>&1; grep -c "^|" gpurun_out/kernel_stats_*.md
tools/gpu_round.sh smoke tests | tail -6
echo "final2 total $(( $(date +%s) - S )) s"
