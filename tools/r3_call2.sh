#!/bin/bash
# round 3, GPU call 2: in-place copies (fixed), kt1 on a narrow grid, counters for the wave-lifetime effect
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 $R/tools/kt1 16384 16384 5 > $R/gpurun_out/r3_kt1_b.log 2>&1 || exit 1
timeout -k 10 200 $R/tools/kt1 91136 2944 5 > $R/gpurun_out/r3_kt1_b_narrow.log 2>&1 || exit 1
pass() {  # name, counters...
  n=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/r3_pmc_$n -- $R/tools/kstream 1 pmc > $R/gpurun_out/r3_pmc_$n.log 2>&1 || exit 1
}
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass tlbstall TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass credit TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum
echo done
