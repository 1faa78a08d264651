#!/bin/bash
# round 3, GPU call 9: what obstacles cost the LDS kernel (same box): none / yeet tiled / stripes / dots; D2Q21 cap sweep
mkdir -p gpurun_out; : > gpurun_out/r3_obst.log
for T in 6 4 1; do for m in none yeet stripes128 stripes1024 dots128 dots16; do
  r=$(timeout -k 10 120 python tools/quick_bench.py 16384 16384 $((T==1?30:12)) periodic $T 0 6 f32 1e-14 $m 2>/dev/null | grep GLUPS | tail -1 | awk '{print $9}') || exit 1
  echo "T=$T mask=$m $r" >> gpurun_out/r3_obst.log
done; done
cat gpurun_out/r3_obst.log
for cap in 14336 16384 18432 20480 22528; do echo cap=$cap; LBM_Q21_LDS_CAP=$cap timeout -k 10 120 python tools/q21_bench.py 16384 16384 20 | grep GLUPS | tail -1; done
