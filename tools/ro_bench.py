"""Row-owner projection kernel (csrc/gemm_ro.hip): per-shape timing of its plans against the incumbents -- the library's best
solution, gemm_g3.hip (M > 128), gemm_wstream.hip (M <= 256) -- back to back on rotating weights (every launch streams from
HBM), at several batch sizes; gate_up also fused with SiLU.mul against projection + activation.

    python tools/ro_bench.py [--model qwen3-14b] [--batches 256 128 64] [--shapes qkv o gate_up down] [--out gpurun_out/ro_bench.json]
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


def time_us(fn, weights, iters=40, rounds=3, warm=40):
    for i in range(warm):
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


def quick_parity(dev) -> bool:
    ok = True
    torch.manual_seed(1)
    for (M, N, K, plans) in [(256, 1024, 512, [(8, 1), (16, 2), (64, 1)]), (128, 2048, 1024, [(8, 1), (32, 4)]),
                             (40, 7168, 256, [(256, 1), (64, 2)])]:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
        ref = x.float() @ w.float().t()
        for p in plans:
            y = ops.ro_linear(x, w, *p)
            err = (y.float() - ref).abs().max().item() / ref.abs().max().item()
            good = err < 2 ** -7
            ok &= good
            print(f"parity M={M} N={N} K={K} plan={p}: rel err {err:.2e} {'ok' if good else 'FAIL'}", flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batches", type=int, nargs="*", default=[256, 128, 64])
    ap.add_argument("--shapes", nargs="*", default=["qkv", "o", "gate_up", "down"])
    ap.add_argument("--tp", type=int, default=1)
    ap.add_argument("--no-library", action="store_true")
    ap.add_argument("--out", default="gpurun_out/ro_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"parity_ok": quick_parity(dev), "rows": []}
    m, D = PRESETS[args.model], PRESETS[args.model].head_dim
    tp = args.tp
    shapes = {"qkv": ((m.num_qo_heads + 2 * m.num_kv_heads) * D // tp, m.hidden_size), "o": (m.hidden_size, m.num_qo_heads * D // tp),
              "gate_up": (2 * m.intermediate_size // tp, m.hidden_size), "down": (m.hidden_size, m.intermediate_size // tp),
              "lm_head": (m.vocab_size // tp // 16 * 16, m.hidden_size)}
    cus = int(lib().msgl_device_cu_count())
    for name in args.shapes:
        N, K = shapes[name]
        nbuf = max(2, min(8, (600 << 20) // (N * K * 2) + 1))
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
        for M in args.batches:
            x = torch.randn((M, K), device=dev, dtype=torch.float32).to(torch.bfloat16)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            row = dict(name=name, M=M, N=N, K=K, weight_MB=round(N * K * 2 / 1e6, 1), ro={})
            if not args.no_library:
                rep = ops.gemm_tune(x, ws, max_candidates=-16, iters=8)
                row.update(library_us=round(rep["best_us"], 1), library_default_us=round(rep["default_us"], 1))
                ops.reset_gemm_plans()
            for p in ops.ro_candidates(M, N, K, cus):
                row["ro"]["/".join(map(str, p))] = round(time_us(lambda w: ops.ro_linear(x, w, p[0], p[1], out), ws), 1)
            if M > 128 and ops.m256_supported(M, N, K):
                row["g3"] = {"/".join(map(str, p)): round(time_us(lambda w: ops.g3_linear(x, w, *p, out=out), ws), 1)
                             for p in ops.m256_candidates(M, N, K, cus)}
            if ops.wstream_supported(M, N, K) and name != "lm_head":
                row["wstream"] = {"/".join(map(str, p)): round(time_us(lambda w: ops.wstream_linear(x, w, p[0], p[1], out), ws), 1)
                                  for p in ops.wstream_candidates(M, N, K)}
            if name == "gate_up" and N % 64 == 0:
                half = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
                row["ro_silu"] = {"/".join(map(str, p)): round(time_us(lambda w: ops.ro_linear(x, w, p[0], 1, half, silu=True), ws), 1)
                                  for p in ops.ro_candidates(M, N, K, cus, silu=True)}
                row["silu_and_mul_us"] = round(time_us(lambda w: ops.silu_and_mul_interleaved(out, half), ws), 1)
            best = min(row["ro"].items(), key=lambda kv: kv[1])
            row["best_ro"] = best
            row["best_ro_TBps"] = round(N * K * 2 / best[1] / 1e6, 2)
            others = {k: min(row[k].values()) for k in ("g3", "wstream") if row.get(k)}
            if "library_us" in row:
                others["library"] = row["library_us"]
            row["best_other"] = min(others.items(), key=lambda kv: kv[1]) if others else None
            res["rows"].append(row)
            print(f"{name:8s} M={M:3d} N={N:6d} K={K:5d} ro {best[0]:>7s} {best[1]:6.1f} us ({row['best_ro_TBps']:.2f} TB/s)"
                  f" | others {others}" + (f" | ro+silu {min(row['ro_silu'].values()):.1f} vs +act {row['silu_and_mul_us']:.1f}" if "ro_silu" in row else ""),
                  flush=True)
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.out).write_text(json.dumps(res, indent=1))
        del ws
        torch.cuda.empty_cache()
    print(json.dumps({"parity_ok": res["parity_ok"], "rows": len(res["rows"])}))


if __name__ == "__main__":
    main()
