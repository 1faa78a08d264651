#!/bin/bash
# round 3, GPU call 1: is the narrow grid still faster with the final round-2 kernel; panel layout and in-place copies
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 ./tools/kt1 16384 16384 5 > gpurun_out/r3_kt1_a.log 2>&1 || exit 1
for g in "16384 16384" "182272 1472" "91136 2944"; do
  timeout -k 10 150 python tools/quick_bench.py $g 30 periodic 1 >> gpurun_out/r3_qb_shapes.log 2>&1 || exit 1
done
rocprofv3 -L > gpurun_out/r3_counters_avail.txt 2>&1 || true
echo done
