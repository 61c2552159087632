#!/bin/bash
# round 4, GPU call 1: the cluster batch-norm kernels (tests, microbenchmark, config 4 with and without them) + the host-enqueue numbers of configs 4 / 5
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
(rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc) > gpurun_out/box.txt 2>&1
timeout 600 python -m pytest tests/test_bn_cluster.py tests/test_parity_ops.py tests/test_half.py tests/test_peephole.py tests/test_staging_ring.py tests/test_comm_multidev.py tests/test_rccl_single.py -m gpu -q -p no:cacheprovider -k "cluster or norm or staging or comm or rccl or relu_pair" > gpurun_out/pytest_bn.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_bn.log; tail -4 gpurun_out/pytest_bn.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 600 python tools/bn_bench.py > gpurun_out/bn_bench.txt 2>&1; echo "exit $?" >> gpurun_out/bn_bench.txt; cat gpurun_out/bn_bench.txt
for mode in 1 0; do
  NNC_MI355X_BN_CLUSTER=$mode timeout 900 python bench.py --config resnet50-nchw-bs256 --steps 4 --warmup 2 $( [ $mode = 0 ] && echo --no-cpu-baseline ) > gpurun_out/bench_resnet50_f32_cluster$mode.json 2> gpurun_out/bench_resnet50_f32_cluster$mode.err; echo "exit $?" >> gpurun_out/bench_resnet50_f32_cluster$mode.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_resnet50_f32_cluster$mode.json"))
print("BN_CLUSTER=$mode", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"), d["config"].get("host_enqueue"), d.get("cpu_baseline"), d["config"].get("oracle_gate"))
PY
  tail -2 gpurun_out/bench_resnet50_f32_cluster$mode.err
done
timeout 900 python bench.py --config resnet50-nchw-bs256-f16 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_resnet50_f16.json 2> gpurun_out/bench_resnet50_f16.err; echo "exit $?" >> gpurun_out/bench_resnet50_f16.err
timeout 900 python bench.py --config cifar10-dawn-f16-bs512 --steps 20 --warmup 3 > gpurun_out/bench_dawn_f16.json 2> gpurun_out/bench_dawn_f16.err; echo "exit $?" >> gpurun_out/bench_dawn_f16.err
python - <<PY
import json
for f in ("bench_resnet50_f16", "bench_dawn_f16"):
    try:
        d = json.load(open("gpurun_out/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["config"].get("host_enqueue"), d.get("cpu_baseline"), d["config"].get("oracle_gate"))
    except Exception as e:
        print(f, "failed", e)
PY
