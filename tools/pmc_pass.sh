#!/bin/bash
# PMC counter passes (rocprofv3 --pmc, one small counter group per pass; never combined with sys/runtime tracing) over a
# reduced VGG-D step (PMC_BENCH_ARGS: another bench.py command line, e.g. "--config resnet50-nchw-bs256 --steps 1 --warmup 1 --no-cpu-baseline").  Output: gpurun_out/pmc/<group>/... csv + gpurun_out/pmc_summary.md (per-kernel sums).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
mkdir -p $OUT
BATCH=${PMC_BATCH:-64}
(cd /tmp && rocprofv3 -L > $OUT/counters_list.txt 2>&1; true)
GROUPS_WANTED=${PMC_GROUPS:-mfma wait inst lds fetch write l2 rdsize wrsize}
run_pass() {
  name=$1; shift
  case " $GROUPS_WANTED " in *" $name "*) ;; *) return;; esac
  (cd /tmp && timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- python $OLDPWD/bench.py ${PMC_BENCH_ARGS:---steps 1 --warmup 0 --batch $BATCH --no-cpu-baseline --no-via-host --no-alt-leg --no-extra-configs} > $OUT/$name.log 2>&1; echo "exit $?" >> $OUT/$name.log)
}
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run_pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
run_pass inst SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass l2 TCC_HIT_sum TCC_MISS_sum
# (round 5) the L2 -> fabric requests by size: an exact byte count (128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B; writes 64 x WRREQ_64B + 32 x the rest),
# calibrated on known volumes by tools/fetch_calib.cpp (profiles/r05_v7_counter_calibration.txt) -- the cross-check of FETCH_SIZE x 2 and the calibration WRITE_SIZE lacks
run_pass rdsize TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run_pass wrsize TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
python tools/pmc_summary.py $OUT > gpurun_out/pmc_summary.md 2>&1
python tools/pmc_traffic.py $OUT gpurun_out/pmc_traffic_bs$BATCH.json $BATCH > gpurun_out/pmc_traffic.txt 2>&1
find $OUT -name "*kernel_trace*" -size +8M -delete
find $OUT -name "*counter_collection*" -size +24M -delete
tail -30 gpurun_out/pmc_traffic.txt
