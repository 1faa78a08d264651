#!/bin/bash
# round 3, GPU call 11: soaks on the final library; the N = 8 bench rehearsal log
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/stress_q21.py 3000 11 > gpurun_out/r3_stress_q21b.log 2>&1 || { tail -5 gpurun_out/r3_stress_q21b.log; exit 1; }
tail -1 gpurun_out/r3_stress_q21b.log
timeout -k 10 500 python -u tools/stress_slabs.py 400 77 > gpurun_out/r3_stress_slabsB.log 2>&1 || { tail -30 gpurun_out/r3_stress_slabsB.log; exit 1; }
tail -1 gpurun_out/r3_stress_slabsB.log
LBM_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 8 --rows 2048 --cols 4096 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3_bench_n8_gloo_rehearsal.log 2>&1 || { tail -30 gpurun_out/r3_bench_n8_gloo_rehearsal.log; exit 1; }
grep "^{" gpurun_out/r3_bench_n8_gloo_rehearsal.log | cut -c1-400
