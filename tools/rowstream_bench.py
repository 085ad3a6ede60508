"""Row-streaming projection (csrc/gemm_rowstream.hip) against the matrix-core kernel of the same batch sizes
(csrc/gemm_skinny.hip, best of its settings) on a model's five projections, back to back on rotating weights, and the
folded launches (fused_add_rmsnorm / SiLU.mul in the staging pass) against the pairs they replace.

    python tools/rowstream_bench.py [--model qwen3-14b] [--batches 1 4] [--out gpurun_out/rowstream_bench.json]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batches", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--out", default="gpurun_out/rowstream_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from mini_sglang_amd import ops
    from mini_sglang_amd.model import PRESETS

    m = PRESETS[args.model]
    H, D, inter = m.hidden_size, m.head_dim, m.intermediate_size
    shapes = [("qkv", (m.num_qo_heads + 2 * m.num_kv_heads) * D, H), ("o", H, m.num_qo_heads * D), ("gate_up", 2 * inter, H),
              ("down", H, inter), ("lm_head", m.vocab_size, H)]
    t = lambda fn, ws: ops._time_launches_us(fn, ws, args.iters, 3)  # noqa: E731
    res = {"model": args.model, "rows": []}
    for name, N, K in shapes:
        copies = max(3, min(8, (600 << 20) // (N * K * 2)))  # > 600 MB in rotation: nothing comes back from a cache
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.03).to(torch.bfloat16) for _ in range(copies)]
        for M in args.batches:
            x = (torch.randn((M, K), device=dev) * 0.5).to(torch.bfloat16)
            out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            row = dict(name=name, M=M, N=N, K=K, MB=round(N * K * 2 / 1e6, 1))
            sk = [(sl, nt) for sl, nt in ops.skinny_candidates(M, N, K) if sl > 0]
            ranked = sorted((ops._time_launches_us(lambda w: ops.skinny_linear(x, w, sl, out, nt), ws, 6, 1), sl, nt) for sl, nt in sk)[:3]
            best = min((t(lambda w: ops.skinny_linear(x, w, sl, out, nt), ws), sl, nt) for _, sl, nt in ranked)
            row.update(skinny_us=round(best[0], 2), skinny_plan=best[1:], skinny_tbs=round(N * K * 2 / best[0] / 1e6, 2))
            best_rs = None
            for v, tag in ((0, "vec"), (1, "mfma")):
                if not ops.rowstream_supported(M, N, K, 0, v):
                    continue
                for d in (8, 16):
                    us = t(lambda w: ops.rowstream_linear(x, w, d, out, variant=v), ws)
                    row[f"{tag}{d}_us"], row[f"{tag}{d}_tbs"] = round(us, 2), round(N * K * 2 / us / 1e6, 2)
                    if best_rs is None or us < best_rs[0]:
                        best_rs = (us, d, v)
            if best_rs is not None:
                _, d, v = best_rs
                row["rowstream_best"] = dict(us=round(best_rs[0], 2), depth=d, variant=v)
                if name in ("qkv", "gate_up", "lm_head") and ops.rowstream_supported(M, N, K, ops.ROWSTREAM_ADD_NORM, v):
                    res_in, res_out = (torch.randn((M, K), device=dev).to(torch.bfloat16) for _ in range(2))
                    gamma = torch.ones((K,), dtype=torch.bfloat16, device=dev)
                    xs = x.clone()

                    def pair(w):
                        ops.fused_add_rmsnorm(xs, res_in, gamma, 1e-6)
                        ops.rowstream_linear(xs, w, d, out, variant=v)

                    row["norm_then_rowstream_us"] = round(t(pair, ws), 2)
                    row["rowstream_with_norm_us"] = round(t(lambda w: ops.rowstream_linear(x, w, d, out, mode=ops.ROWSTREAM_ADD_NORM, res_in=res_in,
                                                                                           res_out=res_out, gamma=gamma, eps=1e-6, variant=v), ws), 2)
                if name == "down" and ops.rowstream_supported(M, N, K, ops.ROWSTREAM_SILU_INTERLEAVED, v):
                    gu = (torch.randn((M, 2 * K), device=dev)).to(torch.bfloat16)
                    act = torch.empty((M, K), dtype=torch.bfloat16, device=dev)

                    def pair(w):
                        ops.silu_and_mul_interleaved(gu, act)
                        ops.rowstream_linear(act, w, d, out, variant=v)

                    row["act_then_rowstream_us"] = round(t(pair, ws), 2)
                    row["rowstream_with_act_us"] = round(t(lambda w: ops.rowstream_linear(gu, w, d, out, mode=ops.ROWSTREAM_SILU_INTERLEAVED,
                                                                                          variant=v), ws), 2)
            print(json.dumps(row), flush=True)
            res["rows"].append(row)
        del ws
        torch.cuda.empty_cache()
    for M in args.batches:
        rows = [r for r in res["rows"] if r["M"] == M and r["name"] != "lm_head"]
        res[f"layer_us_M{M}"] = dict(skinny=round(sum(r["skinny_us"] for r in rows), 1),
                                     rowstream=round(sum(r.get("rowstream_best", {}).get("us", r["skinny_us"]) for r in rows), 1),
                                     best=round(sum(min(r["skinny_us"], r.get("rowstream_best", {}).get("us", 1e9)) for r in rows), 1))
    print(json.dumps({k: v for k, v in res.items() if k.startswith("layer_us")}))
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
