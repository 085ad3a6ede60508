"""Solution search over the library GEMMs of one decode step.

    python tools/gemm_sweep.py --model qwen3-14b --batch 256 --mode full --out gpurun_out/gemm_sweep.json

For each projection shape prints the library's heuristic pick vs the fastest solution found
(timed on rotating weight buffers, i.e. from HBM), as us, TFLOP/s and TB/s of weights, and the
per-step GEMM total.  Also checks the tuned result against torch's own matmul in fp32.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402
from mini_sglang_amd.model import PRESETS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batch", type=int, nargs="+", default=[256])
    ap.add_argument("--tp", type=int, default=1)
    ap.add_argument("--mode", default="full", choices=["heuristic", "full"])
    ap.add_argument("--out", default="gpurun_out/gemm_sweep.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = PRESETS[args.model]
    D, tp = m.head_dim, args.tp
    hq, hkv, inter = m.num_qo_heads // tp, max(m.num_kv_heads // tp, 1), m.intermediate_size // tp
    vocab = (m.vocab_size + tp - 1) // tp
    shapes = [("qkv", (hq + 2 * hkv) * D, m.hidden_size), ("o", m.hidden_size, hq * D),
              ("gate_up", 2 * inter, m.hidden_size), ("down", m.hidden_size, inter),
              ("lm_head", vocab, m.hidden_size)]
    per_step = {"qkv": m.num_layers, "o": m.num_layers, "gate_up": m.num_layers, "down": m.num_layers, "lm_head": 1}
    cand = 0 if args.mode == "full" else -32
    rows = []
    for bs in args.batch:
        tot_def = tot_best = 0.0
        for name, N, K in shapes:
            nbuf = max(1, min(8, (600 << 20) // (N * K * 2) + 1))  # > 2x the 256 MiB Infinity Cache in rotation
            ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
            x = torch.randn((bs, K), device=dev, dtype=torch.float32).to(torch.bfloat16)
            t0 = time.perf_counter()
            r = ops.gemm_tune(x, ws, max_candidates=cand, iters=10)
            r["tune_s"] = time.perf_counter() - t0
            if bs <= ops.SKINNY_MAX_M:
                sk = ops.skinny_tune(x, ws, r["best_us"], iters=10)
                r.update(skinny_us=sk["skinny_us"], skinny_slices=sk["slices"], skinny_row_tiles=sk["row_tiles"],
                         skinny_used=sk["used"])
            if 32 < bs <= ops.WSTREAM_MAX_M:
                wsr = ops.wstream_tune(x, ws, min(r["best_us"], r.get("skinny_us") or 1e30), iters=10)
                r.update(wstream_us=wsr["wstream_us"], wstream_row_tiles=wsr["row_tiles"],
                         wstream_k_splits=wsr["k_splits"], wstream_used=wsr["used"], wstream_all=wsr.get("all"))
            got = ops.linear(x, ws[0]).float()
            ref = x.float() @ ws[0].float().t()
            err = (got - ref).abs().max().item()
            scale = ref.abs().max().item()
            flops, wbytes = 2.0 * bs * N * K, 2.0 * N * K
            r.update(name=name, batch=bs, n_rot=nbuf, max_abs_err=err, ref_absmax=scale,
                     default_tflops=flops / r["default_us"] / 1e6, best_tflops=flops / r["best_us"] / 1e6,
                     default_tbps=wbytes / r["default_us"] / 1e6, best_tbps=wbytes / r["best_us"] / 1e6)
            rows.append(r)
            tot_def += r["default_us"] * per_step[name]
            eff = min(r["best_us"], r.get("skinny_us") or 1e30, r.get("wstream_us") or 1e30)
            if r.get("wstream_us"):
                print(f"      wstream: {r['wstream_us']:8.1f} us ({wbytes / r['wstream_us'] / 1e6:.2f} TB/s) row tiles "
                      f"{r['wstream_row_tiles']} k splits {r['wstream_k_splits']}  {'<< used' if r['wstream_used'] else ''}"
                      f"  {r['wstream_all']}", flush=True)
            tot_best += eff * per_step[name]
            if r.get("skinny_us"):
                print(f"      skinny: {r['skinny_us']:8.1f} us ({wbytes / r['skinny_us'] / 1e6:.2f} TB/s) slices "
                      f"{r['skinny_slices']} row tiles {r['skinny_row_tiles']}  {'<< used' if r['skinny_used'] else ''}", flush=True)
            print(f"bs={bs:4d} {name:8s} N={N:6d} K={K:6d}: heuristic {r['default_us']:8.1f} us "
                  f"({r['default_tflops']:6.0f} TF, {r['default_tbps']:.2f} TB/s) -> best {r['best_us']:8.1f} us "
                  f"({r['best_tflops']:6.0f} TF, {r['best_tbps']:.2f} TB/s) of {r['tried']} in {r['tune_s']:.1f}s "
                  f"err {err:.3g}/{scale:.3g}\n      {r['kernel'][:170]}", flush=True)
            assert err <= 2e-2 * max(scale, 1.0), "tuned solution disagrees with the fp32 reference"
            del ws
        print(f"bs={bs}: GEMM time per decode step  heuristic {tot_def / 1e3:.2f} ms -> tuned {tot_best / 1e3:.2f} ms",
              flush=True)
        rows.append(dict(batch=bs, name="step_total", default_ms=tot_def / 1e3, best_ms=tot_best / 1e3))
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
