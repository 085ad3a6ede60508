#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/c26_pmc -- python $R/tools/profile_gemm.py > $R/gpurun_out/c26_pmc.log 2>&1
DB=$(find $R/gpurun_out/c26_pmc -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 30 > $R/gpurun_out/c26_pmc_fetch.txt 2>&1
grep -A12 "kernel,counter" $R/gpurun_out/c26_pmc_fetch.txt | cut -c1-170
grep algorithmic $R/gpurun_out/c26_pmc.log
find $R/gpurun_out/c26_pmc -name "*.db" -delete
