#!/bin/bash
# round 3: the committed profile summaries (profiles/r03_*): kernel trace + stats of bench.py, FETCH_SIZE / WRITE_SIZE passes
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3prof/stats -- python3 $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r3prof/stats.log 2>&1 || { tail -20 $R/gpurun_out/r3prof/stats.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r3prof/fetch -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3prof/fetch.log 2>&1 || { tail -20 $R/gpurun_out/r3prof/fetch.log; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r3prof/write -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r3prof/write.log 2>&1 || { tail -20 $R/gpurun_out/r3prof/write.log; exit 1; }
cd $R
timeout -k 10 400 python bench.py > gpurun_out/r3_bench_b.json 2> gpurun_out/r3_bench_b.err || { tail gpurun_out/r3_bench_b.err; exit 1; }
tail -c 600 gpurun_out/r3_bench_b.json
ls gpurun_out/r3prof/stats/*/
