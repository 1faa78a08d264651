#!/bin/bash
# round 3, GPU call 5: slab tests with host-side bounded waits, one slab soak at default priorities, de-phased persistent copy
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_slabs_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread > gpurun_out/r3_tests_slabs.log 2>&1 || { tail -30 gpurun_out/r3_tests_slabs.log; exit 1; }
tail -3 gpurun_out/r3_tests_slabs.log
timeout -k 10 300 python -u tools/stress_slabs.py 150 31 > gpurun_out/r3_stress_slabs.log 2>&1 || { tail -30 gpurun_out/r3_stress_slabs.log; exit 1; }
tail -2 gpurun_out/r3_stress_slabs.log
timeout -k 10 120 python -u tools/slab_self_bench.py > gpurun_out/r3_slab_self.log 2>&1 || { tail -20 gpurun_out/r3_slab_self.log; exit 1; }
tail -4 gpurun_out/r3_slab_self.log
timeout -k 10 200 ./tools/kstream 5 dyn > gpurun_out/r3_kstream_dyn2.log 2>&1 || exit 1
cat gpurun_out/r3_kstream_dyn2.log
