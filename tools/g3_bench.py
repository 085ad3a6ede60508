"""Generation-3 full-batch projection kernel (csrc/gemm_g3.hip): parity + per-shape timing against the register-staged
kernel (csrc/gemm_m256.hip) and the library's best solution, and the ablation table of its loader / matrix waves.

    python tools/g3_bench.py [--model qwen3-14b] [--batch 256] [--out gpurun_out/g3_bench.json] [--no-ablate]

Weights of several layers are rotated so every launch streams from HBM; `warm` launches first (sustained clocks).
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


def time_us(fn, weights, iters=60, rounds=3, warm=120):
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


def parity(dev):
    """g3 against fp32 x @ w.T, against the register-staged kernel (same k order: expected bit-equal), and the fused
    activation against unfused + msgl_silu_and_mul_interleaved (must be bit-equal)."""
    torch.manual_seed(1)
    ok = True
    for (M, N, K) in [(256, 512, 256), (200, 1024, 1024), (256, 7168, 5120), (129, 5120, 5120)]:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        w = (torch.randn((N, K), device=dev) * 0.05).to(torch.bfloat16)
        ref = x.float() @ w.float().t()
        scale = ref.abs().max().item()
        for plan in sorted(set(ops.m256_candidates(M, N, K, 256) + [(8, N // 128, 1), (8, 1, 3), (256, 0, 2)])):
            if plan[1] > N // 128 or plan[2] > K // 64:
                continue
            y = ops.g3_linear(x, w, *plan)
            y2 = ops.g3_linear(x, w, *plan)
            y0 = ops.m256_linear(x, w, *plan)
            err = (y.float() - ref).abs().max().item() / scale
            rep = torch.equal(y, y2)
            same = torch.equal(y, y0)
            good = err < 2 ** -7 and rep
            ok &= good
            print(f"parity M={M} N={N} K={K} plan={plan}: rel err {err:.2e} repeatable {rep} == m256 {same} {'ok' if good else 'FAIL'}",
                  flush=True)
    for (M, I, K, plan) in [(256, 1024, 512, (256, 16, 1)), (256, 1024, 512, (16, 3, 4)), (256, 17408, 5120, (256, 256, 16)),
                            (200, 3072, 1024, (256, 0, 4))]:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        wg = (torch.randn((2 * I, K), device=dev) * 0.05).to(torch.bfloat16)
        wi = ops.interleave_gate_up(wg)
        fused = ops.g3_linear(x, wi, *plan, silu=True)
        unf = ops.silu_and_mul_interleaved(ops.g3_linear(x, wi, *plan))
        plain = ops.silu_and_mul(ops.g3_linear(x, wg, *plan))
        e1, e2 = torch.equal(fused, unf), torch.equal(fused, plain)
        ok &= e1 and e2
        print(f"silu  M={M} I={I} K={K} plan={plan}: fused == unfused-interleaved {e1}, == reference-layout path {e2}", flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--shapes", nargs="*", default=["qkv", "o", "gate_up", "down"])
    ap.add_argument("--no-ablate", action="store_true")
    ap.add_argument("--no-library", action="store_true")
    ap.add_argument("--out", default="gpurun_out/g3_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {"parity_ok": parity(dev)}
    m, D = PRESETS[args.model], PRESETS[args.model].head_dim
    shapes = {"qkv": ((m.num_qo_heads + 2 * m.num_kv_heads) * D, m.hidden_size), "o": (m.hidden_size, m.num_qo_heads * D),
              "gate_up": (2 * m.intermediate_size, m.hidden_size), "down": (m.hidden_size, m.intermediate_size),
              "lm_head": (m.vocab_size // 128 * 128, m.hidden_size)}
    cus = int(lib().msgl_device_cu_count())
    M = args.batch
    rows = []
    for name in args.shapes:
        N, K = shapes[name]
        nbuf = max(2, min(8, (600 << 20) // (N * K * 2) + 1))
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
        x = torch.randn((M, K), device=dev, dtype=torch.float32).to(torch.bfloat16)
        row = dict(name=name, M=M, N=N, K=K, m256={}, g3={}, g3_default_policy={})
        if not args.no_library:
            lib_rep = ops.gemm_tune(x, ws, max_candidates=-16, iters=10)
            row.update(library_us=lib_rep["best_us"], library_default_us=lib_rep["default_us"])
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for plan in ops.m256_candidates(M, N, K, cus):
            key = "/".join(map(str, plan))
            row["m256"][key] = round(time_us(lambda w: ops.m256_linear(x, w, *plan, out=out), ws), 1)
            row["g3"][key] = round(time_us(lambda w: ops.g3_linear(x, w, *plan, out=out), ws), 1)
            row["g3_default_policy"][key] = round(time_us(lambda w: ops.g3_linear(x, w, *plan, out=out, variant=1), ws), 1)
        if name == "gate_up":
            half = torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
            row["g3_silu"] = {}
            for plan in ops.m256_candidates(M, N, K, cus):
                row["g3_silu"]["/".join(map(str, plan))] = round(
                    time_us(lambda w: ops.g3_linear(x, w, *plan, out=half, silu=True), ws), 1)
            row["silu_and_mul_us"] = round(time_us(lambda w: ops.silu_and_mul(out, half), ws), 1)
        b3 = min(row["g3"].items(), key=lambda kv: kv[1])
        b2 = min(row["m256"].items(), key=lambda kv: kv[1])
        row["best_g3"], row["best_m256"] = b3, b2
        rows.append(row)
        print(f"{name:8s} N={N:6d} K={K:5d} library {row.get('library_us', 0):7.1f} | m256 {b2[0]} {b2[1]:6.1f} | g3 {b3[0]} {b3[1]:6.1f} "
              f"({2.0 * N * K / b3[1] / 1e6:.2f} TB/s) | g3 {row['g3']} | default policy {row['g3_default_policy']}"
              + (f" | silu-fused {row['g3_silu']} silu kernel {row['silu_and_mul_us']}" if name == "gate_up" else ""), flush=True)
        del ws
        torch.cuda.empty_cache()
    res["shapes"] = rows
    if not args.no_ablate:
        # N = 32768: one whole tile per workgroup; K = 5120
        N, K = 32768, 5120
        ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
        x = torch.randn((256, K), device=dev).to(torch.bfloat16)
        out = torch.empty((256, N), dtype=torch.bfloat16, device=dev)
        names = {0: "full (nt weights)", 1: "full (default policy)", 2: "no x loads", 4: "no compute (x + w loads)",
                 8: "no w loads", 6: "w stream only", 12: "x re-reads only", 10: "compute only"}
        abl = {}
        for v, label in names.items():
            abl[label] = round(time_us(lambda w: ops.g3_linear(x, w, 256, 256, 1, out=out, variant=v), ws, iters=100, warm=300), 1)
        abl["m256 full"] = round(time_us(lambda w: ops.m256_linear(x, w, 256, 256, 1, out=out), ws, iters=100, warm=300), 1)
        res["ablation_N32768_K5120"] = abl
        print("ablation N=32768 K=5120 (us):", abl, flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
