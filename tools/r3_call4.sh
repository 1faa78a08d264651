#!/bin/bash
# round 3, GPU call 4: queue-fed persistent copy; L1-side latency / stall counters for the two ends of the lifetime effect
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 $R/tools/kstream 5 dyn > $R/gpurun_out/r3_kstream_dyn.log 2>&1 || exit 1
pass() {  # name, counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_$n -- $R/tools/kstream 1 pmc > $R/gpurun_out/r3_pmc_$n.log 2>&1 || exit 1
}
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
pass tcpstall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES
pass ta TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_SPI_STALL_sum
echo done
