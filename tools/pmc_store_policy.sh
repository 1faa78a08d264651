#!/bin/bash
# rocprofv3 counter passes of the T = 1 kernel with `nt` stores (lib_nt.so, a throw-away build with
# -DLBM_ST_POLICY='"nt"') against the shipped `nt sc1`.  Development aid; output: gpurun_out/r3_pmc_st/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in lib_nt.so latticeboltzmann_amd/liblbm_hip.so; do
  tag=$(basename $lib .so)
  i=0
  for ctrs in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
              "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
              "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
              "TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
              "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1))
    LBM_HIP_LIB=$R/$lib timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/r3_pmc_st/$tag.$i -- python3 $R/tools/quick_bench.py 16384 16384 5 periodic 1 0 6 f32 1e-14 > $R/gpurun_out/r3_pmc_st/$tag.$i.log 2>&1 || { tail -5 $R/gpurun_out/r3_pmc_st/$tag.$i.log; exit 1; }
  done
done
