#!/bin/bash
# Round 17: ReLU in the convolution's epilogue (NNC_MI355X_CONV_ALGO_FUSE_RELU) -- parity, then both settings timed by bench.py.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_ops.py tests/test_vgg_step.py tests/test_parity_fullsize.py -m gpu -q -p no:cacheprovider -x > gpurun_out/round17_tests.log 2>&1; tail -2 gpurun_out/round17_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 8 --warmup 2 --no-via-host --no-cpu-baseline > gpurun_out/bench_r17.json 2> gpurun_out/bench_r17.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r17.json"))
print(d["value"], d["ms_per_step"], d.get("relu_as_separate_commands"), d["roofline"]["frac"], d["config"]["final_loss"])
PY
