#!/bin/bash
# Does the LDS kernel run faster when a strip's rows are contiguous in memory?  One-strip-wide grids
# (M = the strip width of each depth, N so that the cell count stays 2^28) against the square grid.
# Development aid; output: gpurun_out/r3_panel_probe.log
run() { timeout -k 10 150 python tools/quick_bench.py $1 $2 10 periodic $3 0 6 f32 1e-14 2>&1 | grep GLUPS | tail -1; }
for cfg in "2 736" "3 608" "4 368" "6 240" "4 736" "4 1472" "2 1472" "1 368"; do
  set -- $cfg; T=$1; M=$2; N=$(( 268435456 / M ))
  run $N $M $T || exit 1
done
for T in 2 3 4 6; do run 16384 16384 $T || exit 1; done
