"""End-to-end offline benchmark, the shape of the reference's README run (benchmark/offline/bench.py:11-38):
random.seed(0); 256 prompts of randint(100, 1024) ids in [0, 10000); max_tokens randint(100, 1024);
ignore_eos; max_seq_len_override 4096, max_extend_tokens 16384, cuda_graph_max_bs 256, page_size 256;
one untimed warm-up; throughput = sum(max_tokens) / wall of generate() (prefill included).

    python tools/offline_bench.py --model qwen3-0.6b [--temperature 0.6] [--out gpurun_out/offline.json]

Drives the restated scheduler glue of mini_sglang_amd/offline.py (all 256 requests fit one running set, so the
reference's prefill-first policy degenerates to: chunked prefill of everything, then decode until done).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402


def run(model: str = "qwen3-0.6b", num_seqs: int = 256, temperature: float = 0.6, page_size: int = 256,
        gemm_tune: str = "heuristic", device=None) -> dict:
    """One timed generate() of the README workload; returns the result dict (throughput = sum(max_tokens) / wall)."""
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.offline import OfflineRunner

    dev = device or torch.device("cuda:0")
    torch.cuda.set_device(dev)
    rnd = random.Random(0)
    n = num_seqs
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, 1024))] for _ in range(n)]
    params = [SamplingParams(temperature=temperature, ignore_eos=True, max_tokens=rnd.randint(100, 1024))
              for _ in range(n)]
    mcfg = PRESETS[model]
    ecfg = EngineConfig(model=mcfg, dtype=torch.bfloat16, max_running_req=n, cuda_graph_max_bs=n,
                        page_size=page_size, max_seq_len_override=4096, memory_ratio=0.9, gemm_tune=gemm_tune)
    eng = Engine(ecfg, dev)
    runner = OfflineRunner(eng, max_extend_tokens=16384, seed=0)
    runner.warmup_prefill()
    # warm-up generate (bench.py:32: llm.generate(["Benchmark: "], SamplingParams()))
    runner.generate([[1, 2, 3, 4]], [SamplingParams(temperature=0.0, ignore_eos=True, max_tokens=8)])
    r = runner.generate(prompts, params)
    out_tokens = sum(p.max_tokens for p in params)
    in_tokens = sum(len(p) for p in prompts)
    ttft = sorted(r["ttft_ms"])
    res = {
        "model": mcfg.name, "num_seqs": n, "input_tokens": in_tokens, "output_tokens": out_tokens,
        "wall_s": r["wall_s"], "throughput_tok_s": out_tokens / r["wall_s"],
        "prefill_ms": r["prefill_ms"], "decode_ms": r["decode_ms"], "decode_steps": r["decode_steps"],
        "decode_tok_s": r["decode_tokens"] / (r["decode_ms"] * 1e-3),
        "prefill_tok_s": in_tokens / (r["prefill_ms"] * 1e-3),
        "ms_per_decode_step": r["decode_ms"] / max(r["decode_steps"], 1),
        "ttft_p50_ms": ttft[int(len(ttft) * 0.5)], "ttft_p99_ms": ttft[int(len(ttft) * 0.99)],
        "temperature": temperature, "page_size": page_size, "graph_bs": len(eng.graph_runner.graph_bs_list),
        "gemm_tune": gemm_tune,
    }
    eng.shutdown()
    del runner, eng
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--num-seqs", type=int, default=256)
    ap.add_argument("--temperature", type=float, default=0.6)  # the reference bench's value
    ap.add_argument("--page-size", type=int, default=256)
    ap.add_argument("--gemm-tune", default="heuristic")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    res = run(args.model, args.num_seqs, args.temperature, args.page_size, args.gemm_tune)
    print(json.dumps(res), flush=True)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
