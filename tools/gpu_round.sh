#!/bin/bash
# ONE parameterised GPU-box recipe (replaces the numbered one-off scripts of rounds 1-2).  Everything lands in gpurun_out/.
#   gpurun --timeout 2400 -- 'tools/gpu_round.sh <step> [<step> ...]'
# steps (run in the order given):
#   box            what the box is (GPU, cores, memory)
#   tests          pytest -m gpu (PYTEST_ARGS narrows it: PYTEST_ARGS="-k via_host")
#   smoke          __graft_entry__.smoke()
#   bench          bench.py default (config 3): the JSON line + per-launch contraction records        [STEPS, BENCH_ARGS]
#   config:<name>  bench.py --config <name>  (vggd-fwd-bs64, resnet50-nchw-bs256[-f16], cifar10-dawn-f16-bs512, ...)
#   prof           rocprofv3 --kernel-trace --stats of a short default bench -> gpurun_out/kernel_stats.md   [PROF_ARGS]
#   prof:<name>    the same for bench.py --config <name> -> gpurun_out/kernel_stats_<name>.md
#   profhost       kernel trace of the VGG-D step driven by the unmodified reference host -> gpurun_out/via_host_kernel_stats.md
#   pmc            tools/pmc_pass.sh (separate --pmc passes, HBM traffic per launch)                 [PMC_BATCH, PMC_GROUPS]
#   int:<suite>    the reference's own int cases of one suite, case by case (tools/ref_int_tests.py) [INT_MATCH]
#   run:<script>   python <script> > gpurun_out/<basename>.log   (tools/half_bench.py, tools/conv_algo_sweep.py, ...)
# Every step is wrapped in its own timeout; a failing step does not stop the following ones.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STEPS=${STEPS:-6}
prof_cmd() { # $1 = tag, rest = command
  tag=$1; shift
  rm -rf gpurun_out/prof_$tag
  (cd /tmp && timeout ${PROF_TIMEOUT:-900} rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$tag" -o t -- "$@" > "$OLDPWD/gpurun_out/prof_$tag.log" 2>&1; echo "prof exit $?" >> "$OLDPWD/gpurun_out/prof_$tag.log")
  find gpurun_out/prof_$tag -name "*_results.db" | head -1 | while read f; do python tools/prof_summary.py "$f" > gpurun_out/kernel_stats_$tag.md; done
  find gpurun_out/prof_$tag -name "*kernel_trace*" -size +20M -delete
  find gpurun_out/prof_$tag -name "*.db" -size +20M -delete
}
for step in "$@"; do
  echo "=== $step"
  case "$step" in
    box) (rocminfo | grep -E "Marketing Name|gfx" | head -4; nproc; free -g | head -2) > gpurun_out/box.txt 2>&1; cat gpurun_out/box.txt;;
    tests) timeout ${TESTS_TIMEOUT:-1800} python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log; tail -6 gpurun_out/pytest_gpu.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log;;
    bench) timeout 1800 python bench.py --steps $STEPS --warmup 2 --records gpurun_out/records.txt ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err;;
    config:*) n=${step#config:}; timeout 1500 python bench.py --config $n --steps $STEPS --warmup 2 ${BENCH_ARGS} > gpurun_out/bench_$n.json 2> gpurun_out/bench_$n.err; echo "exit $?" >> gpurun_out/bench_$n.err; cut -c1-1500 gpurun_out/bench_$n.json; tail -3 gpurun_out/bench_$n.err;;
    prof) prof_cmd vgg python "$PWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-via-host --no-alt-leg --no-extra-configs ${PROF_ARGS}; cp gpurun_out/kernel_stats_vgg.md gpurun_out/kernel_stats.md 2>/dev/null; head -40 gpurun_out/kernel_stats.md;;
    prof:*) n=${step#prof:}; prof_cmd $n python "$PWD/bench.py" --config $n --steps 2 --warmup 1 --no-cpu-baseline ${PROF_ARGS}; head -40 gpurun_out/kernel_stats_$n.md;;
    profhost) NNC_MI355X_PEEPHOLE_STATS=1 prof_cmd host "$PWD/oracle/_ref/host_vgg_bench.gpu" 256 225 4 1; cp gpurun_out/kernel_stats_host.md gpurun_out/via_host_kernel_stats.md 2>/dev/null; grep -i "look-ahead" gpurun_out/prof_host.log;;
    pmc) timeout 2400 tools/pmc_pass.sh;;
    int:*) s=${step#int:}; timeout 1800 python tools/ref_int_tests.py run gpu $s --timeout 180 ${INT_MATCH:+--match "$INT_MATCH"} --out "gpurun_out/ref_int_${s}${INT_MATCH:+_$(echo "$INT_MATCH" | tr -c "a-zA-Z0-9" _)}.txt" | tail -25;;
    run:*) f=${step#run:}; b=$(basename $f .py); timeout ${RUN_TIMEOUT:-900} python $f ${RUN_ARGS} > gpurun_out/$b.log 2>&1; echo "exit $?" >> gpurun_out/$b.log; tail -40 gpurun_out/$b.log;;
    *) echo "unknown step $step";;
  esac
done
