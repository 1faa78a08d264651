#!/bin/bash
# L2 write counters of the LDS kernel at T = 4: plain stores (lib_plain.so, a throw-away build with
# -DLBM_TB_STREAM_STORE=0) against the shipped line-aligned write-through stores.  Development aid;
# output: gpurun_out/r3_pmc_tb/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in lib_plain.so latticeboltzmann_amd/liblbm_hip.so; do
  tag=$(basename $lib .so)
  i=0
  for ctrs in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" \
              "TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
              "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
    i=$((i+1))
    LBM_HIP_LIB=$R/$lib timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/r3_pmc_tb/$tag.$i -- python3 $R/tools/quick_bench.py 16384 16384 5 periodic 4 0 6 f32 1e-14 > $R/gpurun_out/r3_pmc_tb/$tag.$i.log 2>&1 || { tail -5 $R/gpurun_out/r3_pmc_tb/$tag.$i.log; exit 1; }
  done
done
