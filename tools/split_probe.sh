#!/bin/bash
# The balanced split of the LDS kernel (tb_rows = 0) against the plain grid with the chunk height the
# old chooser picked, same library, same box.  Development aid; output: gpurun_out/r3_split_probe.log
run() { echo -n "T=$1 tb_rows=$2 rounds=${LBM_TB_ROUNDS:-1} bands=${LBM_TB_BANDS:-1}: "; timeout -k 10 150 python tools/quick_bench.py 16384 16384 20 periodic $1 $2 6 f32 1e-14 2>&1 | grep GLUPS | tail -1 | awk '{print $9}'; }
for rep in 1 2; do
for cfg in "2 164" "3 352" "4 484" "5 444" "6 444"; do
  set -- $cfg
  run $1 $2 || exit 1
  run $1 0 || exit 1
  LBM_TB_ROUNDS=2 run $1 0 || exit 1
  LBM_TB_BANDS=4 run $1 0 || exit 1
  LBM_TB_BANDS=8 run $1 0 || exit 1
  LBM_TB_ROUNDS=4 LBM_TB_BANDS=8 run $1 0 || exit 1
done; done
