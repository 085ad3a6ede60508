#!/bin/bash
# rocprofv3 kernel trace of the timed decode steps of bench.py -> gpurun_out/<tag>_bench_timed_steps_kernel_breakdown.txt
TAG=${1:-trace}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prefill-roofline --small-batches > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample --last-steps 20 > $R/gpurun_out/${TAG}_bench_timed_steps_kernel_breakdown.txt 2>&1
head -26 $R/gpurun_out/${TAG}_bench_timed_steps_kernel_breakdown.txt | cut -c1-150
find $R/gpurun_out/${TAG}_kt -name "*.db" -delete
tail -1 $R/gpurun_out/${TAG}_kt.log | cut -c1-200
