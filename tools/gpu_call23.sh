#!/bin/bash
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c23_kt -- python $R/bench.py --no-cpu-baseline --no-prefill-roofline --small-batches > $R/gpurun_out/c23_kt.log 2>&1
DB=$(find $R/gpurun_out/c23_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 40 > $R/gpurun_out/c23_kt_stats.txt 2>&1
python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample_logits_kernel --last-steps 40 > $R/gpurun_out/c23_kt_timed_steps.txt 2>&1
head -24 $R/gpurun_out/c23_kt_timed_steps.txt | cut -c1-170
grep '^{"metric' $R/gpurun_out/c23_kt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])"
find $R/gpurun_out/c23_kt -name "*.db" -delete
