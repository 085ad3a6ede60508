#!/bin/bash
# One parameterised entry for every GPU-box job of a round (replaces the per-call rNN_runK.sh scripts):
#     gpurun --timeout 900 -- 'bash tools/gpu_job.sh <tag> <job> [<job> ...]'
# Every job writes under gpurun_out/<tag>_* and prints a short digest; jobs are independent and each is bounded by its
# own `timeout`.  Jobs:
#   kv_probe          tools/kv_stream_probe.hip (request shape x policy x landing place for the paged K/V stream)
#   decode_ab:<impls> tools/decode_ab.py on the six shapes with the given select codes (comma-separated)
#   tests:<expr>      pytest -m gpu -k <expr>          tests_all   the whole GPU suite
#   bench             the driver's bench command       bench_fast  the same without e2e / cpu baseline / prefill roofline
#   trace_step        rocprofv3 kernel trace of the timed steps -> <tag>_bench_timed_steps_kernel_breakdown.txt
#   rank_shard        bench.py --rank-shard on the five TP configurations
#   trace_replay:<model>[:<requests>]  tools/trace_replay.py at the reference's six scales, --cache naive
#   tp2_share[:<model>]  bench.py --gpus 2 with both ranks on one GPU (code-path check, incl. the collectives preflight)
#   pmc_attn          rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes + kernel trace of the decode attention launch -> <tag>_pmc_attn_decode.json
#   pmc_lm_head       FETCH_SIZE / WRITE_SIZE / L2-request counters + kernel trace of the LM head (library solution and row-owner kernel)
#   smoke             __graft_entry__.smoke()
#   py:<script args>  python tools/<script args>
TAG=$1; shift
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for JOB in "$@"; do
  NAME=${JOB%%:*}; ARG=${JOB#*:}; [ "$ARG" == "$JOB" ] && ARG=""
  echo "=== $TAG $JOB"
  case $NAME in
    kv_probe)
      mkdir -p tools/build
      [ -x tools/build/kv_stream_probe ] || hipcc -O3 --offload-arch=gfx950 tools/kv_stream_probe.hip -o tools/build/kv_stream_probe
      timeout 300 tools/build/kv_stream_probe ${ARG:-3} > gpurun_out/${TAG}_kv_stream_probe.txt 2>&1  # kv_probe:<rounds>[ <skew %>]
      cat gpurun_out/${TAG}_kv_stream_probe.txt | cut -c1-200 ;;
    decode_ab)  # decode_ab:<impls>[:<page>[:<alloc>]]
      IFS=: read -r IMPLS PAGE ALLOC <<< "$ARG"
      SUF=""; [ -n "$PAGE" ] && SUF="_page${PAGE}_${ALLOC:-shuffled}"
      ( time timeout 600 python tools/decode_ab.py --impls ${IMPLS:-1,0,72} --page ${PAGE:-256} --alloc ${ALLOC:-shuffled} --shape 14b,14b_tp4,32b_tp4,70b_tp8,0.6b,14b_b32 --out gpurun_out/${TAG}_decode_ab$SUF.json ) > gpurun_out/${TAG}_decode_ab$SUF.txt 2>&1
      grep -v "^$" gpurun_out/${TAG}_decode_ab$SUF.txt | tail -40 | cut -c1-200 ;;
    tests)
      ( time timeout 1500 python -m pytest tests -m gpu -q -x -k "$ARG" ) > gpurun_out/${TAG}_pytest_k.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_k.log | cut -c1-300 ;;
    tests_all)
      ( time timeout 1700 python -m pytest tests -m gpu -q -x ) > gpurun_out/${TAG}_pytest.log 2>&1; tail -12 gpurun_out/${TAG}_pytest.log | cut -c1-300 ;;
    bench|bench_fast)
      EXTRA=""; [ $NAME == bench_fast ] && EXTRA="--no-e2e --no-cpu-baseline --no-prefill-roofline"
      ( time timeout 900 python bench.py --steps 20 --warmup 5 $EXTRA $ARG ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
      tail -3 gpurun_out/${TAG}_bench.err | cut -c1-300
      python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("summary", {k: d[k] for k in ("value", "ms_per_step")})))
    print(json.dumps(d["roofline"])[:900])
    for r in d.get("gemm_tune", {}).get("refined_in_graph", []):
        print(r["name"], r.get("chosen"), json.dumps(r.get("tried", r.get("ms", {})))[:400])
except Exception as e:
    print("bench line unreadable:", e)
PY
      ;;
    trace_step) bash tools/trace_step.sh $TAG 2>&1 | tail -30 ;;
    rank_shard)
      MS=${ARG:-17.5}
      for cfg in "qwen3-14b 4" "qwen3-32b 4" "qwen3-14b 2" "qwen3-14b 8" "llama-3.1-70b 8"; do set -- $cfg
        timeout 240 python bench.py --model $1 --rank-shard $2 --tp1-ms $MS --steps 20 --warmup 5 2>gpurun_out/${TAG}_rank_shard_$1_tp$2.err | tail -1 > gpurun_out/${TAG}_rank_shard_$1_tp$2.json
        python -c "import json; d=json.loads(open('gpurun_out/${TAG}_rank_shard_$1_tp$2.json').read()); print('$1 tp$2', round(d['ms_per_step'],3), 'ms')" 2>&1 | tail -1
      done ;;
    trace_replay)
      IFS=: read -r MODEL NREQ <<< "$ARG"
      ( time timeout 1700 python tools/trace_replay.py --model ${MODEL:-qwen3-14b} --requests ${NREQ:-300} --out gpurun_out/${TAG}_trace_replay_${MODEL:-qwen3-14b}_tp1.json ) 2>&1 | tail -12 | cut -c1-400 ;;
    tp2_share)  # bench.py --gpus 2 self-launched with both ranks on ONE GPU (code-path check of the N > 1 path incl. the preflight)
      ( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --model ${ARG:-qwen3-0.6b} ) > gpurun_out/${TAG}_bench_tp2_share.json 2> gpurun_out/${TAG}_bench_tp2_share.err
      grep "bench preflight" gpurun_out/${TAG}_bench_tp2_share.err | cut -c1-1500; tail -3 gpurun_out/${TAG}_bench_tp2_share.err | cut -c1-300
      python -c "
import json
d=json.loads(open('gpurun_out/${TAG}_bench_tp2_share.json').read().strip().splitlines()[-1])
print('N=2 on one GPU:', d.get('launch'), '| ms/step', d.get('ms_per_step'), '|', json.dumps(d.get('collectives'))[:1200])" 2>&1 | tail -3 ;;
    pmc_attn)  # HBM traffic of the decode attention launch: separate --pmc passes (+ the probe's nt whole-line kernel as a second calibration)
      cd /tmp && export TMPDIR=/tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/${TAG}_pmc_$C -- python $R/tools/profile_attn.py --advance 26 > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
        DB=$(find $R/gpurun_out/${TAG}_pmc_$C -name "*results.db" | head -1)
        timeout 120 python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/${TAG}_pmc_$C.txt 2>&1
        grep -A6 "kernel,counter" $R/gpurun_out/${TAG}_pmc_$C.txt | cut -c1-160
        find $R/gpurun_out/${TAG}_pmc_$C -name "*.db" -delete
      done
      mkdir -p $R/tools/build; [ -x $R/tools/build/kv_stream_probe ] || hipcc -O3 --offload-arch=gfx950 $R/tools/kv_stream_probe.hip -o $R/tools/build/kv_stream_probe
      timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_probe -- $R/tools/build/kv_stream_probe 1 > $R/gpurun_out/${TAG}_pmc_probe.log 2>&1
      DB=$(find $R/gpurun_out/${TAG}_pmc_probe -name "*results.db" | head -1)
      timeout 120 python $R/tools/rocpd_summary.py $DB --top 24 > $R/gpurun_out/${TAG}_pmc_probe_FETCH_SIZE.txt 2>&1
      grep -A12 "kernel,counter" $R/gpurun_out/${TAG}_pmc_probe_FETCH_SIZE.txt | cut -c1-160
      find $R/gpurun_out/${TAG}_pmc_probe -name "*.db" -delete
      timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_attn_kt -- python $R/tools/profile_attn.py --advance 26 > $R/gpurun_out/${TAG}_attn_kt.log 2>&1
      DB=$(find $R/gpurun_out/${TAG}_attn_kt -name "*results.db" | head -1)
      timeout 120 python $R/tools/rocpd_summary.py $DB --top 8 > $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt 2>&1; cut -c1-160 $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt | head -8
      find $R/gpurun_out/${TAG}_attn_kt -name "*.db" -delete
      ALGO=$(grep algorithmic_bytes_per_launch $R/gpurun_out/${TAG}_attn_kt.log | awk '{print $2}')
      python $R/tools/pmc_json.py $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}_pmc_WRITE_SIZE.txt \
        $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt $ALGO $R/gpurun_out/${TAG}_pmc_attn_decode.json "tools/gpu_job.sh $TAG pmc_attn" | cut -c1-600
      cd $R ;;
    pmc_lm_head)
      cd /tmp && export TMPDIR=/tmp
      for C in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
        NAME=$(echo $C | tr ' ' '_')
        timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/${TAG}_pmclm_$NAME -- python $R/tools/pmc_lm_head.py > $R/gpurun_out/${TAG}_pmclm_$NAME.log 2>&1
        DB=$(find $R/gpurun_out/${TAG}_pmclm_$NAME -name "*results.db" | head -1)
        timeout 120 python $R/tools/rocpd_summary.py $DB --top 8 > $R/gpurun_out/${TAG}_pmc_lm_head_$NAME.txt 2>&1
        grep -A8 "kernel,counter" $R/gpurun_out/${TAG}_pmc_lm_head_$NAME.txt | cut -c1-200
        find $R/gpurun_out/${TAG}_pmclm_$NAME -name "*.db" -delete
      done
      timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_pmclm_kt -- python $R/tools/pmc_lm_head.py > $R/gpurun_out/${TAG}_pmclm_kt.log 2>&1
      DB=$(find $R/gpurun_out/${TAG}_pmclm_kt -name "*results.db" | head -1)
      timeout 120 python $R/tools/rocpd_summary.py $DB --top 6 > $R/gpurun_out/${TAG}_pmc_lm_head_kernel_trace.txt 2>&1; cut -c1-200 $R/gpurun_out/${TAG}_pmc_lm_head_kernel_trace.txt | head -8; grep algorithmic $R/gpurun_out/${TAG}_pmclm_kt.log
      find $R/gpurun_out/${TAG}_pmclm_kt -name "*.db" -delete
      cd $R ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    py) ( time timeout 900 python tools/$ARG ) 2>&1 | tail -40 | cut -c1-300 ;;
    *) echo "unknown job $JOB" ;;
  esac
done
