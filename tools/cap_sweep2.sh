#!/bin/bash
# finer residency-cap sweep, fp32 (see cap_sweep.sh); output: gpurun_out/r3_cap_sweep2.log
for rep in 1 2 3; do
for cap in 9216 10240 10496 10752 10880 11008 11264 11520 11648 11776; do
    echo -n "cap=$cap f32: "
    LBM_T1_LDS_CAP=$cap timeout -k 10 120 python tools/quick_bench.py 16384 16384 20 periodic 1 0 6 f32 1e-14 2>&1 | grep GLUPS | tail -1 | awk '{print $9}' || exit 1
done
done
for cap in 9216 11264; do for b in reference; do
    echo -n "cap=$cap f32 $b: "
    LBM_T1_LDS_CAP=$cap timeout -k 10 120 python tools/quick_bench.py 16384 16384 20 $b 1 0 6 f32 1e-14 2>&1 | grep GLUPS | tail -1 | awk '{print $9}' || exit 1
    echo -n "cap=$cap f32 $b omega 0.05 yeet: "
    LBM_T1_LDS_CAP=$cap timeout -k 10 120 python tools/quick_bench.py 16384 16384 20 $b 1 0 6 f32 0.05 yeet 2>&1 | grep GLUPS | tail -1 | awk '{print $9}' || exit 1
done; done
