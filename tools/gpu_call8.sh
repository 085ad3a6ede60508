#!/bin/bash
# A/B of the decode-attention compute (reduce-scatter vs all-reduce), row kernels, decode-only step trace.
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 600 python -m pytest tests/test_gpu_attn_decode.py tests/test_gpu_model.py tests/test_gpu_rowops.py -m gpu -q -x ) > gpurun_out/c8_pytest.log 2>&1
tail -6 gpurun_out/c8_pytest.log | cut -c1-200
for RS in 1 0; do
  ( MSGL_DECODE_RS=$RS timeout 300 python tools/microbench.py --only decode --out gpurun_out/c8_microbench_decode_rs$RS.json ) > gpurun_out/c8_microbench_decode_rs$RS.log 2>&1
  echo "== RS=$RS"; grep "^decode" gpurun_out/c8_microbench_decode_rs$RS.log | sed -E "s/'bytes.*'chunk': [0-9]+, //" | cut -c1-260
done
( timeout 300 python tools/microbench.py --only rows --out gpurun_out/c8_microbench_rows.json ) > gpurun_out/c8_microbench_rows.log 2>&1
grep "_T256\|_T8192" gpurun_out/c8_microbench_rows.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/c8_kt -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-prefill --no-prefill-roofline > $R/gpurun_out/c8_kt.log 2>&1
DB=$(find $R/gpurun_out/c8_kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample_kernel --last-steps 10 > $R/gpurun_out/c8_kt_decode_step.txt 2>&1; cut -c1-200 $R/gpurun_out/c8_kt_decode_step.txt | head -40
grep '^{"metric' $R/gpurun_out/c8_kt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
find $R/gpurun_out/c8_kt -name "*.db" -delete
