#!/bin/bash
# decode attention (reduce-scatter + buffer loads): parity, microbench, PMC traffic passes, e2e offline bench.
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_attn_decode.py tests/test_gpu_model.py -m gpu -q -x ) > gpurun_out/c9_pytest.log 2>&1
tail -4 gpurun_out/c9_pytest.log | cut -c1-200
( timeout 300 python tools/microbench.py --only decode --out gpurun_out/c9_microbench_decode.json ) > gpurun_out/c9_microbench_decode.log 2>&1
grep "^decode" gpurun_out/c9_microbench_decode.log | sed -E "s/'bytes.*'chunk': [0-9]+, //" | cut -c1-200
for M in qwen3-0.6b qwen3-14b; do
  ( time timeout 600 python tools/offline_bench.py --model $M --out gpurun_out/c9_offline_$M.json ) > gpurun_out/c9_offline_$M.log 2>&1
  grep '^{' gpurun_out/c9_offline_$M.log | cut -c1-700; tail -4 gpurun_out/c9_offline_$M.log | grep -v "^{" | cut -c1-200
done
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/c9_pmc_$C -- python $R/tools/profile_attn.py > $R/gpurun_out/c9_pmc_$C.log 2>&1
  DB=$(find $R/gpurun_out/c9_pmc_$C -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/c9_pmc_$C.txt 2>&1
  grep -A8 "kernel,counter" $R/gpurun_out/c9_pmc_$C.txt | cut -c1-160
  find $R/gpurun_out/c9_pmc_$C -name "*.db" -delete
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c9_kt -- python $R/tools/profile_attn.py > $R/gpurun_out/c9_kt.log 2>&1
DB=$(find $R/gpurun_out/c9_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 8 > $R/gpurun_out/c9_kt.txt 2>&1; cut -c1-160 $R/gpurun_out/c9_kt.txt
find $R/gpurun_out/c9_kt -name "*.db" -delete
grep algorithmic $R/gpurun_out/c9_kt.log
