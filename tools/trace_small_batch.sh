#!/bin/bash
# rocprofv3 kernel trace of decode steps at a small batch (default 1) -> gpurun_out/<tag>_b<bs>_kernel_breakdown.txt
TAG=${1:-trace}; BS=${2:-1}
R=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout ${TRACE_TIMEOUT:-400} rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_ktb -- python $R/bench.py --batch $BS --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-prefill-roofline --small-batches > $R/gpurun_out/${TAG}_ktb.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_ktb -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample --last-steps 20 > $R/gpurun_out/${TAG}_b${BS}_kernel_breakdown.txt 2>&1
head -24 $R/gpurun_out/${TAG}_b${BS}_kernel_breakdown.txt | cut -c1-150
find $R/gpurun_out/${TAG}_ktb -name "*.db" -delete
tail -2 $R/gpurun_out/${TAG}_ktb.log | cut -c1-300
