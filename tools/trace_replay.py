"""BASELINE config 4 ("Qwen3-32B ..., Qwen-trace 1000-req replay", README online benchmark) as a SCHEDULER-LEVEL replay on
one MI355X: the REFERENCE's own Scheduler / PrefillManager / radix cache / Engine / GraphRunner (through
minisgl_plugin.install(), as in tests/test_gpu_reference_driven.py) fed requests at the trace's arrival times; TTFT, TPOT and
E2E percentiles by the rule of the reference's benchmark client (P/benchmark/client.py:324-384).

    python tools/trace_replay.py [--model qwen3-14b] [--requests 300] [--rate 6.0] [--scales 0.4,0.5,0.6,0.7,0.8,1.6]
                                 [--cache naive] [--out gpurun_out/trace_replay.json]

Round 6: sweeps the reference's own scale list (benchmark/online/bench_qwen.py:41) with `--cache naive` (README :169), one
worker process per scale, and writes ONE summary carrying the fingerprint of the product sources; bench.py surfaces the newest
committed summary of its model as `online_trace_replay` (kind "committed", stale-marked when the sources have changed since).

What it is NOT: the HTTP server / tokenizer / ZMQ path (out of scope), the real Qwen trace (no network:
tests/refdrive.synth_qwen_trace documents the synthetic stand-in), or TP = 4 (one GPU here: Qwen3-32B runs at TP = 1 in
64 GB of the 288 GB).  Random weights (`use_dummy_weight`): only time is measured.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-32b")
    ap.add_argument("--requests", type=int, default=300)
    ap.add_argument("--rate", type=float, default=6.0, help="mean arrivals per second of the synthetic trace")
    ap.add_argument("--scales", default="0.4,0.5,0.6,0.7,0.8,1.6", help="timestamps x scale, one replay each (bench_qwen.py:41)")
    ap.add_argument("--cache", default="naive", choices=["naive", "radix"], help="prefix cache of the reference's scheduler")
    ap.add_argument("--gemm-tune", default="heuristic")
    ap.add_argument("--out", default="gpurun_out/trace_replay.json")
    args = ap.parse_args()
    import refdrive
    from bench import product_code_fingerprint

    trace = refdrive.synth_qwen_trace(args.requests, args.rate)
    kw = dict(page_size=256, max_running_req=256, cuda_graph_bs=[1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256], max_seq_len_override=8192,
              max_extend_tokens=8192, cache_type=args.cache, memory_ratio=0.9)
    out = dict(what="scheduler-level replay of a synthetic Qwen-like trace through the reference's Scheduler on the HIP backend "
                    "(percentiles by the reference client's rule: sorted[int(n p)], P/benchmark/client.py:324-348)",
               model=args.model, tp=1, cache=args.cache, code_fingerprint=product_code_fingerprint(),
               trace=dict(kind="synthetic (tests/refdrive.synth_qwen_trace)", requests=args.requests, rate_per_s=args.rate), scales={})
    for sc in [float(x) for x in args.scales.split(",")]:
        rec = refdrive.run_worker(dict(model=args.model, weights="dummy", llm_kwargs=kw, gemm_tune=args.gemm_tune, vectorized_glue=True,
                                       native_radix=True, trace=trace, trace_scale=sc, max_position=40960), timeout=1500)
        r = rec["trace_replay"]
        out.update(device=rec["device"], backend=rec["backend"], prefix_cache=rec["prefix_cache"], graph_bs=rec["graph_bs"])
        out["trace"].update(input_tokens=r["input_tokens"], output_tokens=r["tokens"])
        out["scales"][str(sc)] = dict(integrity=rec["integrity"], init_and_run_s=rec["init_and_run_s"],
                                      **{k: r[k] for k in ("complete", "duration_s", "throughput_tok_s", "req_per_s", "ttft_ms", "tpot_ms", "e2e_s")})
        print(json.dumps(dict(model=args.model, scale=sc, complete=r["complete"], duration_s=round(r["duration_s"], 1),
                              tok_s=round(r["throughput_tok_s"]), ttft_ms={k: round(v, 1) for k, v in r["ttft_ms"].items()},
                              tpot_ms={k: round(v, 2) for k, v in r["tpot_ms"].items()})), flush=True)
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
