"""Per-shape timing of the full-batch projection kernel (csrc/gemm_m256.hip) against the library's best solution.

    python tools/m256_bench.py [--model qwen3-14b] [--batch 256] [--library-candidates -16] [--out gpurun_out/m256.json]

Weights of several layers are rotated so every launch streams from HBM.  Prints us, TB/s of weights and TFLOP/s
for every plan (grid, whole tiles, k-slices) and the library's pick.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402
from mini_sglang_amd._lib import lib  # noqa: E402
from mini_sglang_amd.model import PRESETS  # noqa: E402


def time_us(fn, weights, iters=20, rounds=3, warm=40):
    for i in range(warm):  # clocks ramp over milliseconds: time at the sustained state
        fn(weights[i % len(weights)])
    best = 1e30
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(weights[(i + 1) % len(weights)])
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--library-candidates", type=int, default=-16)
    ap.add_argument("--shapes", nargs="*", default=["qkv", "o", "gate_up", "down", "lm_head"])
    ap.add_argument("--pad-x", type=int, default=0, help="extra elements in x's row stride (channel-conflict probe)")
    ap.add_argument("--pad-w", type=int, default=0, help="extra elements in w's row stride")
    ap.add_argument("--out", default="gpurun_out/m256_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m, D = PRESETS[args.model], PRESETS[args.model].head_dim
    shapes = {"qkv": ((m.num_qo_heads + 2 * m.num_kv_heads) * D, m.hidden_size), "o": (m.hidden_size, m.num_qo_heads * D),
              "gate_up": (2 * m.intermediate_size, m.hidden_size), "down": (m.hidden_size, m.intermediate_size),
              "lm_head": (m.vocab_size, m.hidden_size)}
    cus = int(lib().msgl_device_cu_count())
    M = args.batch
    rows = []
    for name in args.shapes:
        N, K = shapes[name]
        nbuf = max(1, min(8, (600 << 20) // (N * K * 2) + 1))
        ws = [(torch.randn((N, K + args.pad_w), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16)[:, :K]
              for _ in range(nbuf)]
        x = torch.randn((M, K + args.pad_x), device=dev, dtype=torch.float32).to(torch.bfloat16)[:, :K]
        ref = x.float() @ ws[0].float().t()
        lib_rep = ops.gemm_tune(x, ws, max_candidates=args.library_candidates, iters=10)
        row = dict(name=name, M=M, N=N, K=K, library_us=lib_rep["best_us"], library_default_us=lib_rep["default_us"],
                   library_kernel=lib_rep["kernel"][:90], plans={})
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for plan in ops.m256_candidates(M, N, K, cus):
            us = time_us(lambda w: ops.m256_linear(x, w, *plan, out=out), ws)
            ops.m256_linear(x, ws[0], *plan, out=out)
            err = (out.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
            row["plans"]["/".join(map(str, plan))] = dict(us=round(us, 1), weight_TBps=round(2.0 * N * K / us / 1e6, 2),
                                                          tflops=round(2.0 * M * N * K / us / 1e6, 1), rel_err=err)
        best = min(row["plans"].items(), key=lambda kv: kv[1]["us"])
        row["best_plan"], row["best_us"] = best[0], best[1]["us"]
        rows.append(row)
        print(f"{name:8s} N={N:6d} K={K:5d} library {row['library_us']:7.1f} us (default {row['library_default_us']:.1f}) | "
              f"m256 best {best[0]} {best[1]['us']:7.1f} us = {best[1]['weight_TBps']} TB/s, {best[1]['tflops']} TF | "
              + "  ".join(f"{k}:{v['us']}" for k, v in row["plans"].items()), flush=True)
        del ws
        torch.cuda.empty_cache()
    L = m.num_layers
    per = {r["name"]: r for r in rows}
    if all(k in per for k in ("qkv", "o", "gate_up", "down")):
        lib_ms = sum(per[k]["library_us"] for k in ("qkv", "o", "gate_up", "down")) * L / 1e3
        new_ms = sum(min(per[k]["library_us"], per[k]["best_us"]) for k in ("qkv", "o", "gate_up", "down")) * L / 1e3
        print(f"per step ({L} layers, without lm_head): library {lib_ms:.2f} ms -> best-of {new_ms:.2f} ms")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
