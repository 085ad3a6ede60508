#!/bin/bash
# End-of-round validation on the GPU box (run through gpurun from the repo root): GPU parity suite, the driver's bench
# command, rocprofv3 kernel trace of the same command, PMC traffic passes of the dominant kernel.  Every step is
# bounded by `timeout`; summaries land in gpurun_out/<tag>_* (copy what is to be judged into profiles/).
TAG=${1:-r04}
R=$(pwd)
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > gpurun_out/${TAG}_pytest.log 2>&1
tail -4 gpurun_out/${TAG}_pytest.log | cut -c1-200
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("bench:", round(d["value"]), "tok/s", round(d["ms_per_step"], 2), "ms/step; attn", round(d["roofline"]["us_per_launch"], 1), "us frac",
          round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"], "; step frac", round(d["step_roofline"]["frac"], 3),
          "; e2e", {k: round(v.get("throughput_tok_s", 0)) for k, v in d.get("e2e_offline", {}).items()}, "; refdriven", d.get("reference_driven", {}).get("decode_ms_per_step"))
except Exception as e:
    print("bench line unreadable:", e)
PY
tail -3 gpurun_out/${TAG}_bench.err | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-prefill-roofline --small-batches > $R/gpurun_out/${TAG}_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_kt -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 > $R/gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1
timeout 120 python $R/tools/rocpd_summary.py $DB --top 30 --steps-by sample --last-steps 20 > $R/gpurun_out/${TAG}_bench_timed_steps_kernel_breakdown.txt 2>&1
head -24 $R/gpurun_out/${TAG}_bench_timed_steps_kernel_breakdown.txt | cut -c1-170
find $R/gpurun_out/${TAG}_kt -name "*.db" -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C -d $R/gpurun_out/${TAG}_pmc_$C -- python $R/tools/profile_attn.py --advance 26 > $R/gpurun_out/${TAG}_pmc_$C.log 2>&1
  DB=$(find $R/gpurun_out/${TAG}_pmc_$C -name "*results.db" | head -1)
  timeout 120 python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/${TAG}_pmc_$C.txt 2>&1
  grep -A8 "kernel,counter" $R/gpurun_out/${TAG}_pmc_$C.txt | cut -c1-160
  find $R/gpurun_out/${TAG}_pmc_$C -name "*.db" -delete
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_attn_kt -- python $R/tools/profile_attn.py --advance 26 > $R/gpurun_out/${TAG}_attn_kt.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_attn_kt -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 8 > $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt 2>&1; cut -c1-160 $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt | head -12
find $R/gpurun_out/${TAG}_attn_kt -name "*.db" -delete
grep algorithmic $R/gpurun_out/${TAG}_attn_kt.log
ALGO=$(grep algorithmic_bytes_per_launch $R/gpurun_out/${TAG}_attn_kt.log | awk '{print $2}')
python $R/tools/pmc_json.py $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}_pmc_WRITE_SIZE.txt \
  $R/gpurun_out/${TAG}_attn_decode_kernel_trace.txt $ALGO $R/gpurun_out/${TAG}_pmc_attn_decode.json "validate_round.sh $TAG" | cut -c1-400
# PMC passes over the M = 256 projection path (per GEMM kernel, slab consumers), rank-shard TP bounds
cd $R
bash tools/pmc_gemm.sh ${TAG} > gpurun_out/${TAG}_pmc_gemm.log 2>&1; grep -c "," gpurun_out/${TAG}_pmc_gemm_FETCH_SIZE.txt
MS=$(python - <<PY
import json
try:
    print(round(json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])["ms_per_step"], 3))
except Exception:
    print(17.8)
PY
)
for cfg in "qwen3-14b 2" "qwen3-14b 4" "qwen3-14b 8" "qwen3-32b 4" "llama-3.1-70b 8"; do set -- $cfg; timeout 200 python bench.py --model $1 --rank-shard $2 --tp1-ms $MS --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${TAG}_rank_shard_$1_tp$2.json; cut -c1-200 gpurun_out/${TAG}_rank_shard_$1_tp$2.json; done
bash tools/trace_small_batch.sh ${TAG} 1 > gpurun_out/${TAG}_b1_trace.log 2>&1; head -3 gpurun_out/${TAG}_b1_kernel_breakdown.txt
# batches <= 8: the row-streaming projection against the 16-row kernel back to back, and inside the captured step with / without the folds
hipcc -O3 --offload-arch=gfx950 tools/persist_probe.hip -o tools/build/persist_probe 2>/dev/null; timeout 100 tools/build/persist_probe > gpurun_out/${TAG}_persist_probe.txt 2>&1; head -8 gpurun_out/${TAG}_persist_probe.txt
timeout 300 python tools/rowstream_bench.py --iters 8 --out gpurun_out/${TAG}_rowstream_bench.json 2>/dev/null | tail -1 | cut -c1-400
timeout 400 python tools/small_batch_ab.py --out gpurun_out/${TAG}_small_batch_ab.json 2>/dev/null | grep -v gemm_tune | head -5
# same-box A/B of the decode attention kernels, per-wave clock stamps of the default one, MFMA counters of prefill attention
cd $R
timeout 300 python tools/decode_ab.py --shape 14b,14b_tp4,32b_tp4,70b_tp8,0.6b,14b_b32 --impls 1,94,0,22,23,32,72,92 --out gpurun_out/${TAG}_decode_ab.json 2>&1 | grep impl > gpurun_out/${TAG}_decode_ab.txt
head -8 gpurun_out/${TAG}_decode_ab.txt
timeout 200 python tools/decode_trace.py --out gpurun_out/${TAG}_decode_trace.json > /dev/null 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_mfma -- python $R/tools/microbench.py --only prefill --out $R/gpurun_out/${TAG}_microbench_prefill_under_pmc.json > $R/gpurun_out/${TAG}_pmc_mfma.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_pmc_mfma -name "*results.db" | head -1)
timeout 120 python $R/tools/rocpd_summary.py $DB --top 12 > $R/gpurun_out/${TAG}_pmc_mfma_prefill_attention.txt 2>&1
grep -A6 "kernel,counter" $R/gpurun_out/${TAG}_pmc_mfma_prefill_attention.txt | cut -c1-160
find $R/gpurun_out/${TAG}_pmc_mfma -name "*.db" -delete
# prefill attention: kernel generations timed interleaved + the default kernel's ablations, wave-level counters, segment stamps of
# the counter-phase kernel; fused gate_up + SiLU.mul of the weight-streaming kernel against projection + activation
cd $R
timeout 200 python tools/prefill_ablate.py --rounds 3 --only "tr: full" "dma: full" --out gpurun_out/${TAG}_prefill_variants.json > gpurun_out/${TAG}_prefill_variants.log 2>&1
python - <<PY
import json
try:
    r = json.load(open("gpurun_out/${TAG}_prefill_variants.json"))
    for n, c in r.items():
        print(n, {k: v["frac_if_full"] for k, v in c["variants"].items() if isinstance(v, dict) and k in ("tr: full", "dma: full")})
except Exception as e:
    print("prefill variants unreadable:", e)
PY
timeout 300 bash tools/pmc_prefill.sh ${TAG} 2 4 > gpurun_out/${TAG}_pmc_prefill.log 2>&1; grep -c "SQ_" gpurun_out/${TAG}_pmc_prefill.txt
timeout 100 python tools/skinny_silu_bench.py --out gpurun_out/${TAG}_skinny_silu_bench.json 2>&1 | tail -3 | cut -c1-200
# round 4: the N > 1 path of bench.py end to end through its SELF-LAUNCH (python3 bench.py --gpus N, no torchrun around it), all
# ranks on this box's one GPU over the peer-to-peer communicator (MSGL_BENCH_SHARE_GPU=1: a code-path check, not a
# measurement; --gpus 4 also runs the Qwen3-32B TP4 entry); what one CU's memory pipe delivers by source (tools/cu_pipe_probe.hip);
# kernel choices A/B'd inside the captured step; BASELINE config 4 as a scheduler-level trace replay (Qwen3-32B, TP1 here)
cd $R
# (four 14B ranks time-slicing ONE GPU did not finish in 15 minutes -- three spinning peers starve the fourth -- so the N = 4
# code path (second workload in the same processes: communicators torn down and rebuilt) runs at N = 2 on a small model)
( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off timeout 500 python bench.py --gpus 2 --steps 3 --warmup 1 ) > gpurun_out/${TAG}_bench_tp2_ranks_on_one_gpu_code_path_check.json 2> gpurun_out/${TAG}_bench_tp2_share.err
( time MSGL_BENCH_SHARE_GPU=1 MSGL_GEMM_TUNE=off MSGL_BENCH_SECOND_MODEL=qwen3-0.6b timeout 400 python bench.py --gpus 2 --model qwen3-0.6b --steps 3 --warmup 1 ) > gpurun_out/${TAG}_bench_tp2_two_workloads_code_path_check.json 2> gpurun_out/${TAG}_bench_tp2_two_share.err
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_tp2_ranks_on_one_gpu_code_path_check.json", "gpurun_out/${TAG}_bench_tp2_two_workloads_code_path_check.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("self-launched --gpus 2 on one GPU:", d["launch"], "| ms/step", round(d["ms_per_step"], 1), "|", d["collectives"]["paths"],
              "| second workload:", {k: ({kk: (round(vv, 1) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("ms_per_step", "error")}) for k, v in d.items() if k.startswith("second_config") or k == "qwen3_32b_tp4"})
    except Exception as e:
        print("self-launch check unreadable:", f, e)
PY
[ -x tools/build/cu_pipe_probe ] || { mkdir -p tools/build; hipcc -O3 --offload-arch=gfx950 tools/cu_pipe_probe.hip -o tools/build/cu_pipe_probe; }
timeout 120 tools/build/cu_pipe_probe > gpurun_out/${TAG}_cu_pipe_probe.txt 2>&1; tail -13 gpurun_out/${TAG}_cu_pipe_probe.txt | cut -c1-200
timeout 400 python tools/step_ab.py --rounds 3 --knobs decode72 decode71 lib_o lib_gate_up no_slab_norm --out gpurun_out/${TAG}_step_ab.json 2>&1 | tail -2 | cut -c1-600
timeout 900 python tools/trace_replay.py --model qwen3-32b --requests 300 --rate 6.0 --out gpurun_out/${TAG}_trace_replay_qwen3-32b_tp1.json 2>&1 | tail -2 | cut -c1-700
