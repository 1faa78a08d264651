#!/bin/bash
# Residency cap of the one-update kernels (unused dynamic LDS per single-wave workgroup) re-swept after
# the stores became `nt sc1`.  Development aid; output: gpurun_out/r3_cap_sweep.log
for rep in 1 2; do
for cap in 0 6144 7168 8192 9216 10240 11264 12288 14336; do
  for dt in f32 f16; do
    echo -n "cap=$cap $dt: "
    LBM_T1_LDS_CAP=$cap timeout -k 10 120 python tools/quick_bench.py 16384 16384 20 periodic 1 0 6 $dt 1e-14 2>&1 | grep GLUPS | tail -1 | awk '{print $9}' || exit 1
  done
done
done
for cap in 12288 14336 16384 18432 20480 24576 32768; do
  echo -n "q21 cap=$cap: "
  LBM_Q21_LDS_CAP=$cap timeout -k 10 120 python tools/q21_bench.py 2>&1 | grep -i glups | tail -1 || exit 1
done
