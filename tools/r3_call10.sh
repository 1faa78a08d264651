#!/bin/bash
# round 3, GPU call 10: parity of the T=3 window change; chunk-height sweep beyond 512 rows
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_temporal_blocking_gpu.py tests/test_fullsize_gpu.py tests/test_parity_gpu.py tests/test_slabs_gpu.py tests/test_fp16_render_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_tests_nb3.log 2>&1 || { tail -30 gpurun_out/r3_tests_nb3.log; exit 1; }
tail -2 gpurun_out/r3_tests_nb3.log
timeout -k 10 200 python -u tools/stress_parity.py 3000 99 > gpurun_out/r3_stress_nb3.log 2>&1 || { tail -5 gpurun_out/r3_stress_nb3.log; exit 1; }
tail -1 gpurun_out/r3_stress_nb3.log
: > gpurun_out/r3_rows.log
for T in 2 3 4 5 6; do for rows in 0 256 512 768 1024 2048 4096; do
  r=$(timeout -k 10 120 python tools/quick_bench.py 16384 16384 12 periodic $T $rows 6 2>/dev/null | grep GLUPS | tail -1 | awk '{print $9}') || exit 1
  echo "T=$T rows=$rows $r" >> gpurun_out/r3_rows.log
done; done
cat gpurun_out/r3_rows.log
