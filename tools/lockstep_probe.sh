#!/bin/bash
# One launch = exactly one round of 256 workgroups that start together (strips x chunks = 256, 512-row
# chunks), against two and four rounds of the same shape: does the LDS kernel run faster while its
# workgroups are still in step?  Development aid; output: gpurun_out/r3_lockstep_probe.log
run() { timeout -k 10 150 python tools/quick_bench.py $1 $2 40 periodic $3 512 6 f32 1e-14 2>&1 | grep GLUPS | tail -1; }
for k in 1 2 4 8; do
  run $((8192 * k)) 11776 2 || exit 1
  run $((8192 * k)) 9728 3 || exit 1
  run $((4096 * k)) 11776 4 || exit 1
done
