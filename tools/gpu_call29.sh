#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $R/gpurun_out/c29_pmc -- python $R/tools/microbench.py --only prefill --out $R/gpurun_out/c29_microbench_prefill.json > $R/gpurun_out/c29_pmc.log 2>&1
DB=$(find $R/gpurun_out/c29_pmc -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 30 > $R/gpurun_out/c29_pmc_mfma.txt 2>&1
grep -A10 "kernel,counter" $R/gpurun_out/c29_pmc_mfma.txt | cut -c1-170
grep "^prefill" $R/gpurun_out/c29_pmc.log | cut -c1-200
find $R/gpurun_out/c29_pmc -name "*.db" -delete
