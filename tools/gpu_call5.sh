#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/c5_pytest.log 2>&1
tail -4 gpurun_out/c5_pytest.log
( time timeout 900 python bench.py ) > gpurun_out/c5_bench.log 2>&1
grep '^{"metric' gpurun_out/c5_bench.log > gpurun_out/c5_bench.json
cut -c1-700 gpurun_out/c5_bench.json; tail -c 900 gpurun_out/c5_bench.json
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/c5_pmc_$C -- python $R/tools/profile_attn.py > $R/gpurun_out/c5_pmc_$C.log 2>&1
  DB=$(find $R/gpurun_out/c5_pmc_$C -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/c5_pmc_$C.txt 2>&1
  grep -A14 "kernel,counter" $R/gpurun_out/c5_pmc_$C.txt | cut -c1-160
  find $R/gpurun_out/c5_pmc_$C -name "*.db" -delete
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5_kt -- python $R/tools/profile_attn.py > $R/gpurun_out/c5_kt.log 2>&1
DB=$(find $R/gpurun_out/c5_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 8 > $R/gpurun_out/c5_kt.txt 2>&1; cut -c1-160 $R/gpurun_out/c5_kt.txt
find $R/gpurun_out/c5_kt -name "*.db" -delete
grep algorithmic $R/gpurun_out/c5_kt.log
