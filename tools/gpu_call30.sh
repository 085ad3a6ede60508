#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/c30_pmc_attn -- python $R/tools/profile_attn.py > $R/gpurun_out/c30_pmc_attn.log 2>&1
DB=$(find $R/gpurun_out/c30_pmc_attn -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 40 > $R/gpurun_out/c30_pmc_attn_sq.txt 2>&1
grep -A12 "kernel,counter" $R/gpurun_out/c30_pmc_attn_sq.txt | grep "attn_decode" | cut -c1-170
find $R/gpurun_out/c30_pmc_attn -name "*.db" -delete
timeout 90 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -d $R/gpurun_out/c30_pmc_lds -- python $R/tools/profile_gemm.py > $R/gpurun_out/c30_pmc_lds.log 2>&1
DB=$(find $R/gpurun_out/c30_pmc_lds -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 40 > $R/gpurun_out/c30_pmc_lds.txt 2>&1
grep -A14 "kernel,counter" $R/gpurun_out/c30_pmc_lds.txt | grep "gemm\|counter" | cut -c1-170
tail -3 $R/gpurun_out/c30_pmc_lds.log | cut -c1-200
find $R/gpurun_out/c30_pmc_lds -name "*.db" -delete
