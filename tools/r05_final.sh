#!/bin/bash
# round 5: the evidence set of the final tree, one GPU call.  Everything lands in gpurun_out/; the summaries are copied into profiles/r05_* afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
S=$(date +%s)
timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 --records gpurun_out/records.txt > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench exit $? after $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_default.json"))
print("value", d["value"], "ms", d["ms_per_step"], "via_host", d.get("via_host_images_per_s"), "roofline", d["roofline"]["frac"], d["roofline"]["avg_ms"], "traffic", d["roofline"]["traffic"])
print("look-ahead", d["config"]["via_host"].get("relu_look_ahead"))
for k, v in d.get("extra_configs", {}).items():
    print(k, {a: v.get(a) for a in ("value", "ms_per_step", "wall_s", "status", "via_host_images_per_s")}, (v.get("roofline") or {}).get("frac"))
PY
tools/gpu_round.sh prof profhost | tail -3
PMC_BATCH=256 PMC_GROUPS="mfma wait inst lds fetch write rdsize wrsize" tools/pmc_pass.sh > gpurun_out/pmc_pass.log 2>&1; tail -12 gpurun_out/pmc_traffic.txt
tools/r05_preproc.sh | tail -8
echo "total $(( $(date +%s) - S )) s"
