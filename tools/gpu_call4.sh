#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
( time timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c4_prof -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-prefill ) > $R/gpurun_out/c4_prof.log 2>&1
cd $R
tail -2 gpurun_out/c4_prof.log | cut -c1-400
find gpurun_out/c4_prof -type f | head -20
DB=$(find gpurun_out/c4_prof -name "*results.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py $DB --top 40 --csv gpurun_out/c4_kernel_stats.csv > gpurun_out/c4_kernel_stats.txt 2>&1; cat gpurun_out/c4_kernel_stats.txt | cut -c1-200; fi
find gpurun_out/c4_prof -name "*kernel_stats*.csv" | head -3
# keep the merge small: drop the raw trace db
find gpurun_out/c4_prof -name "*.db" -size +30M -delete
