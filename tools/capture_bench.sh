#!/bin/bash
# HIP-graph capture on the MI355X box: the captured-step tests, then configs 5 / 4-f16 / 4 through the reference host with the step
# issued command by command and replayed from the captured graph: step time, host time per step (one thread), devices one thread can feed.
#   gpurun --timeout 1500 -- tools/capture_bench.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out/capture
timeout 900 python -m pytest tests/test_capture.py tests/test_via_host.py -m gpu -q -p no:cacheprovider -k "capture" > ${O}_pytest.log 2>&1; echo "pytest exit $?" >> ${O}_pytest.log; tail -8 ${O}_pytest.log
H=oracle/_ref/host_resnet_bench.gpu
run() { # tag, args...
  tag=$1; shift
  for c in 0 1; do
    HOST_BENCH_CAPTURE=$c timeout 600 $H "$@" > ${O}_${tag}_c$c.json 2> ${O}_${tag}_c$c.err; echo "exit $?" >> ${O}_${tag}_c$c.err
    python - "$tag" "$c" ${O}_${tag}_c$c.json <<'PY'
import json, sys
tag, c, path = sys.argv[1:4]
try:
    h = json.loads(open(path).read().strip().splitlines()[-1])
    he = h["host_enqueue"]
    print("%-22s capture=%s  %8.3f ms/step  %9.1f img/s  host %7.4f ms/step (median of 5, drained)  -> one thread feeds %5.1f devices   graph nodes %d  commands/step %d  probe %r" % (
        tag, c, h["ms_per_step"], h["images_per_s"], he["ms_per_step_median"], h["ms_per_step"] / he["ms_per_step_median"] if he["ms_per_step_median"] > 0 else float("nan"), h["capture"]["graph_nodes"], he["commands_per_step"], h["replica_probe_sumsq"]))
except Exception as e:
    print(tag, c, "FAILED", e, open(path.replace(".json", ".err")).read()[-800:])
PY
  done
}
run dawn-f16-bs512 512 32 30 3 16 dawn
NNC_MI355X_CAPTURE_STREAMS=1 run dawn-f16-bs512-kept-streams 512 32 30 3 16 dawn
run resnet50-f16-bs256 256 224 8 2 16 full
run resnet50-f32-bs256 256 224 6 2 32 full
