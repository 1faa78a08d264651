#!/bin/bash
# round 3, GPU call 7: row-stride padding sweep with the product library (T = 1, 4, 6)
mkdir -p gpurun_out
: > gpurun_out/r3_pad_sweep.log
for T in 1 4 2; do
for pad in 0 1 2 3 4 5 6 7 8 12 16 0; do
  r=$(LBM_ROW_PAD=$pad timeout -k 10 120 python tools/quick_bench.py 16384 16384 $((T==1?30:12)) periodic $T 2>/dev/null | grep GLUPS | tail -1 | awk '{print $9}') || exit 1
  echo "T=$T pad=$pad $r" >> gpurun_out/r3_pad_sweep.log
done; done
cat gpurun_out/r3_pad_sweep.log
