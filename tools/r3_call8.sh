#!/bin/bash
# round 3, GPU call 8: L1-side latency counters of the product kernels (T=1 against the LDS kernel at T=4 and T=2); small-grid launch times
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {  # name T counters...
  n=$1; T=$2; shift; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmcK_${n}_T$T -- python3 $R/tools/quick_bench.py 16384 16384 4 periodic $T > $R/gpurun_out/r3_pmcK_${n}_T$T.log 2>&1 || exit 1
}
for T in 1 4 2; do
pass lat $T TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
pass ea $T TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum
pass sq $T SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
done
cd $R
: > gpurun_out/r3_small.log
for cfg in "400 800 200 reference 1 0 1" "400 800 100 reference 6 0 6" "400 800 100 reference 6 4 6" "400 800 100 reference 6 8 6" "400 800 100 reference 6 16 6" "400 800 100 reference 6 32 6" "400 800 100 reference 3 0 3" "400 800 100 reference 3 8 3" "400 800 100 reference 2 0 2" "1355 3782 50 reference 6 0 6"; do
  timeout -k 10 60 python tools/quick_bench.py $cfg 2>/dev/null | grep GLUPS | tail -1 | sed "s/^/[$cfg] /" >> gpurun_out/r3_small.log || exit 1
done
cat gpurun_out/r3_small.log
