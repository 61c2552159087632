#!/bin/bash
# PMC passes over tools/kprobe.cpp (every DBG variant is its own kernel symbol, so the per-kernel rows separate them).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/kpmc
rm -rf $OUT; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I ccv_amd/csrc tools/kprobe.cpp -o /tmp/kprobe 2>/dev/null
run_pass() {
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o pmc -- /tmp/kprobe ${KPROBE_ARGS:-64 55 256 256} > $OUT/$name.log 2>&1; echo "exit $?" >> $OUT/$name.log)
}
run_pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
run_pass vmem SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run_pass tcp TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES
run_pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES
run_pass ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS
run_pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES
python tools/pmc_summary.py $OUT > gpurun_out/kprobe_pmc.md 2>&1
find $OUT -name "*.csv" -size +4M -delete
cat gpurun_out/kprobe_pmc.md | cut -c1-220
