R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/r02g_pmc_mfma -- python $R/tools/microbench.py --only prefill --out $R/gpurun_out/r02g_microbench_prefill_under_pmc.json > $R/gpurun_out/r02g_pmc_mfma.log 2>&1
echo rc=$?
DB=$(find $R/gpurun_out/r02g_pmc_mfma -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/r02g_pmc_mfma_prefill_attention.txt 2>&1
grep -A8 "kernel,counter" $R/gpurun_out/r02g_pmc_mfma_prefill_attention.txt | cut -c1-160
find $R/gpurun_out/r02g_pmc_mfma -name "*.db" -delete
