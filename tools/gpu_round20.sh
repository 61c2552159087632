#!/bin/bash
# Round 20: the other configurations with the look-ahead on (and off, for the host-driven ones), and the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in resnet50-nchw-bs256 resnet50-nchw-bs256-f16 cifar10-dawn-f16-bs512 cifar10-dawn-f32-bs512; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline > gpurun_out/bench_r20_$cfg.json 2> gpurun_out/bench_r20_$cfg.err
  NNC_MI355X_PEEPHOLE=0 timeout 600 python bench.py --config $cfg --no-cpu-baseline > gpurun_out/bench_r20_${cfg}_off.json 2>> gpurun_out/bench_r20_$cfg.err
  python - $cfg <<'PY'
import json, sys
c = sys.argv[1]
a = json.load(open("gpurun_out/bench_r20_%s.json" % c)); b = json.load(open("gpurun_out/bench_r20_%s_off.json" % c))
print(c, "on %.1f off %.1f img/s" % (a["value"], b["value"]))
PY
done
timeout 600 python bench.py --config vggd-fwd-bs64 --no-cpu-baseline > gpurun_out/bench_r20_fwd64.json 2> gpurun_out/bench_r20_fwd64.err; cut -c1-200 gpurun_out/bench_r20_fwd64.json
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_r20.json 2> gpurun_out/bench_r20.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r20.json"))
print(d["value"], d["ms_per_step"], d.get("relu_as_separate_passes"), d["config"]["via_host"].get("images_per_s"), d["roofline"]["frac"])
PY
