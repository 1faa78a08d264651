#!/bin/bash
# round 3, GPU call 3: panel layout and padded strides with the product kernel's addressing
mkdir -p gpurun_out
timeout -k 10 300 ./tools/kt1 16384 16384 5 > gpurun_out/r3_kt1_c.log 2>&1 || exit 1
timeout -k 10 150 python tools/quick_bench.py 16384 16384 30 periodic 1 > gpurun_out/r3_qb_c.log 2>&1 || exit 1
echo done
