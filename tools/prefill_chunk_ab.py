"""What the prefill-chunk GEMM solution search (gemm_plan.tune_prefill_gemms, EngineConfig.prefill_tokens) is worth at the
reference's default chunk size: the bench's 256 prompts prefilled in `max_extend_tokens`-token chunks by two engines of the
same model, one with the library's heuristic picks at M = chunk, one with the searched solutions.

    python tools/prefill_chunk_ab.py [--model qwen3-14b] [--chunk 8192] [--out gpurun_out/prefill_chunk_ab.json]
"""
from __future__ import annotations

import argparse
import gc
import json
import random
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run(model, chunk, tuned, dev):
    from bench import bench_contexts
    from mini_sglang_amd import ops
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.offline import OfflineRunner

    B = 256
    ecfg = EngineConfig(model=PRESETS[model], dtype=torch.bfloat16, max_running_req=B, cuda_graph_bs=[], page_size=256,
                        max_seq_len_override=4096, num_page_override=2048,  # a fixed pool that holds the batch: the engines of this
                        # process follow each other, and the caching allocator keeps the previous pool's memory
                        gemm_tune="heuristic", refine_in_graph=False,
                        prefill_tokens=chunk if tuned else None)
    eng = Engine(ecfg, dev)
    runner = OfflineRunner(eng, max_extend_tokens=chunk, seed=0)
    rnd = random.Random(1234)
    prompts = [[rnd.randint(0, 10000) for _ in range(n)] for n in bench_contexts(B)]
    sp = SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=4)
    runner.warmup_prefill()
    times = []
    for rep in range(2):
        states = [runner.add_request(p, sp) for p in prompts]
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        chunks = sum(1 for _ in runner.prefill(states))
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1))
        for st in states:
            runner._free(st.req)
    rep_ = [dict(name=r["name"], M=r["M"], default_us=round(r["default_us"], 1), best_us=round(r["best_us"], 1), kernel=r["kernel"][:60])
            for r in eng.gemm_report if r.get("prefill")]
    tokens = sum(len(p) for p in prompts)
    eng.shutdown()
    del eng, runner
    gc.collect()
    torch.cuda.empty_cache()
    ops.reset_gemm_plans()
    return dict(tuned=tuned, prefill_ms=min(times), chunks=chunks, tokens=tokens, tok_per_s=tokens / min(times) * 1e3, search=rep_)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--chunk", type=int, default=8192)
    ap.add_argument("--out", default="gpurun_out/prefill_chunk_ab.json")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = dict(model=a.model, chunk=a.chunk, runs=[run(a.model, a.chunk, t, dev) for t in (False, True, False, True)])
    h = min(r["prefill_ms"] for r in res["runs"] if not r["tuned"])
    t = min(r["prefill_ms"] for r in res["runs"] if r["tuned"])
    res["summary"] = dict(heuristic_ms=h, searched_ms=t, speedup=h / t)
    print(json.dumps(res["summary"]), json.dumps(res["runs"][1]["search"]))
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
