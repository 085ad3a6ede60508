"""Per-kernel micro-benchmarks on one MI355X: GB/s against algorithmic bytes.

    python tools/microbench.py [--out gpurun_out/microbench.json]
"""
from __future__ import annotations

import argparse
import json
import math
import random
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402


def time_us(fn, iters=20, warmup=3, stats=None):
    """Median of per-launch event timings (a one-off hiccup must not masquerade as kernel time);
    `stats`, if given, receives min / max / back-to-back average as well."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for e0, e1 in evs:
        e0.record()
        fn()
        e1.record()
    b.record()
    b.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
    if stats is not None:
        stats.update(min_us=ts[0], max_us=ts[-1], avg_us=a.elapsed_time(b) * 1e3 / iters)
    return ts[len(ts) // 2]


def bench_lens(n=256, seed=0):
    """Context lengths of BASELINE config 2/3 mid-run: input 100..1024 plus a uniform share of the output."""
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        i, o = rnd.randint(100, 1024), rnd.randint(100, 1024)
        out.append(i + rnd.randint(0, o - 1))
    return out


def decode_case(B, hq, hkv, lens, page_size, dev, dtype=torch.bfloat16, D=128, shuffle=True):
    max_seq = (max(lens) + 31) // 32 * 32
    pages_per_req = (max_seq + page_size - 1) // page_size
    n_pages = B * pages_per_req + 1
    slots = n_pages * page_size
    k = torch.randn((slots, hkv, D), device=dev, dtype=dtype)
    v = torch.randn((slots, hkv, D), device=dev, dtype=dtype)
    perm = torch.randperm(n_pages - 1) if shuffle else torch.arange(n_pages - 1)  # pages in shuffled / allocation order
    table = torch.zeros((B, max_seq), dtype=torch.int32)
    p = 0
    for b in range(B):
        need = (lens[b] + page_size - 1) // page_size
        pg = perm[p:p + need].to(torch.int32) * page_size
        p += need
        tok = (pg.unsqueeze(1) + torch.arange(page_size, dtype=torch.int32)).flatten()
        table[b, : min(need * page_size, max_seq)] = tok[:max_seq]
    q = torch.randn((B, hq, D), device=dev, dtype=dtype)
    return k, v, table.to(dev), q


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16


def prefill_case(q_lens, k_lens, hq, hkv, page_size, dev, dtype=torch.bfloat16, D=128, q_tile=128):
    """A prefill batch in a shuffled paged pool: returns the ops.attn_prefill argument tuple + causal flops."""
    import numpy as np

    from mini_sglang_amd.attention import prefill_tile_order

    B = len(q_lens)
    max_seq = (max(k_lens) + 31) // 32 * 32
    pages = [-(-n // page_size) for n in k_lens]
    n_pages = sum(pages) + 1
    k = torch.randn((n_pages * page_size, hkv, D), device=dev, dtype=dtype)
    v = torch.randn((n_pages * page_size, hkv, D), device=dev, dtype=dtype)
    perm = torch.randperm(n_pages - 1)
    table = torch.zeros((B, max_seq), dtype=torch.int32)
    pp = 0
    for b in range(B):
        pg = perm[pp: pp + pages[b]].to(torch.int32) * page_size
        pp += pages[b]
        tok = (pg.unsqueeze(1) + torch.arange(page_size, dtype=torch.int32)).flatten()
        table[b, : k_lens[b]] = tok[: k_lens[b]]
    T = sum(q_lens)
    q = torch.randn((T, hq, D), device=dev, dtype=dtype)
    out = torch.empty_like(q)
    cu_q = torch.tensor([0] + list(q_lens), dtype=torch.int32).cumsum(0).to(torch.int32).to(dev)
    tiles = np.array([(n + q_tile - 1) // q_tile for n in q_lens], dtype=np.int64)
    tile_cu = torch.tensor([0] + tiles.tolist(), dtype=torch.int32).cumsum(0).to(torch.int32).to(dev)
    order = torch.from_numpy(prefill_tile_order(np.array(q_lens, dtype=np.int64), np.array(k_lens, dtype=np.int64),
                                                tiles, q_tile)).to(dev)
    seq = torch.tensor(k_lens, dtype=torch.int32, device=dev)
    # flops = 4 Hq D sum_i [q_i (k_i - q_i) + q_i (q_i + 1) / 2]
    flops = 4 * hq * D * sum(qi * (ki - qi) + qi * (qi + 1) // 2 for qi, ki in zip(q_lens, k_lens))
    return dict(out=out, q=q, k=k, v=v, table=table.to(dev), seq=seq, cu_q=cu_q, tile_cu=tile_cu, order=order,
                B=B, total_tiles=int(tiles.sum()), flops=flops, T=T)


def prefill_chunk_lens(budget, lens):
    """Requests of `lens` packed into one chunk of `budget` new tokens (last one cut), no cache hit."""
    q, left = [], budget
    for n in lens:
        if left <= 0:
            break
        q.append(min(n, left))
        left -= q[-1]
    return q


def run_prefill(res, dev):
    import bench as _bench

    ctx = _bench.bench_contexts(256)
    chunk = prefill_chunk_lens(16384, ctx)
    cases = [
        ("prefill_14b_chunk16384_bench", chunk, chunk, 40, 8, 256),
        ("prefill_14b_8x2048", [2048] * 8, [2048] * 8, 40, 8, 256),
        ("prefill_14b_chunked_hit", [512] * 32, [1024] * 32, 40, 8, 256),
        ("prefill_14b_tp4_chunk16384", chunk, chunk, 10, 2, 256),
        ("prefill_0.6b_chunk16384", chunk, chunk, 16, 8, 256),
    ]
    for name, ql, kl, hq, hkv, page in cases:
        c = prefill_case(ql, kl, hq, hkv, page, dev)
        r = dict(flops=c["flops"], tokens=c["T"], requests=c["B"], q_tiles=c["total_tiles"])
        for label, impl, order in (("dma_heavy", 4, c["order"]), ("dma_natural", 4, None), ("tr_heavy", 2, c["order"])):
            f = lambda: ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"],  # noqa: E731
                                         c["tile_cu"], c["B"], c["total_tiles"], 128 ** -0.5, tile_order=order,
                                         impl=impl)
            us = time_us(f, iters=10, warmup=10)
            r[f"us_{label}"] = us
            r[f"TFLOPs_{label}"] = c["flops"] / us / 1e6
            r[f"mfma_frac_{label}"] = c["flops"] / us / 1e6 / MFMA_PEAK_TFLOPS
        res[name] = r
        print(name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
        del c


def run(args):
    dev = torch.device("cuda:0")
    res = {"device": torch.cuda.get_device_name(0), "cus": ops.lib().msgl_device_cu_count()}
    it = 2  # bytes per element
    if args.only == "rows":
        res = {}
        run_rows(res, dev)
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(res, indent=1))
        return
    if args.only in ("all", "prefill"):
        run_prefill(res, dev)
        if args.only == "prefill":
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.out).write_text(json.dumps(res, indent=1))
            return
    # ---- decode attention, Qwen3-14B TP1 shape, B=256
    for name, B, hq, hkv, page in [("decode_14b_b256_p256", 256, 40, 8, 256), ("decode_14b_b256_p1", 256, 40, 8, 1),
                                   ("decode_14b_tp2_b256", 256, 20, 4, 256), ("decode_14b_tp4_b256", 256, 10, 2, 256),
                                   ("decode_14b_tp8_b256", 256, 5, 1, 256),
                                   ("decode_0.6b_b256", 256, 16, 8, 256), ("decode_32b_tp4_b256", 256, 16, 2, 256),
                                   ("decode_70b_tp8_b256", 256, 8, 1, 256), ("decode_14b_b1", 1, 40, 8, 256),
                                   ("decode_14b_b16", 16, 40, 8, 256)]:
        lens = bench_lens(B)
        k, v, table, q = decode_case(B, hq, hkv, lens, page, dev)
        D = 128
        cap = max(4096, 2 * B)
        plan = torch.zeros(ops.attn_decode_plan_words(B, cap), dtype=torch.int32, device=dev)
        ws = torch.empty(ops.attn_decode_workspace_bytes(cap, hq, D), dtype=torch.uint8, device=dev)
        seq = torch.tensor(lens, dtype=torch.int32, device=dev)
        out = torch.empty_like(q)
        ops.attn_decode_plan(plan, seq, B, B, cap, hq, hkv)
        S = sum(lens)
        bytes_ = S * 2 * hkv * D * it + 2 * B * hq * D * it + S * 4 + 2 * B * 4
        pl = plan[:2].tolist()
        res[name] = dict(bytes=bytes_, sum_len=S, n_items=pl[0], chunk=pl[1])
        for run in ([1] if page < 16 else [page, 1]):  # slot_run: scalar table walk vs per-token walk
            f = lambda: ops.attn_decode(out, q, k, v, table, None, seq, plan, ws, B, B, cap, D ** -0.5, slot_run=run)
            st = {}
            us = time_us(f, stats=st)
            res[name][f"us_run{run}"] = us
            res[name][f"GBps_run{run}"] = bytes_ / us / 1e3
            res[name][f"minmaxavg_run{run}"] = [round(st["min_us"], 1), round(st["max_us"], 1), round(st["avg_us"], 1)]
        print(name, res[name], flush=True)
        del k, v, table, q, ws
    if args.only == "decode":
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(res, indent=1))
        return
    run_rows(res, dev)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


def run_rows(res, dev):
    it = 2
    # ---- row ops at decode (T=256) and prefill (T=8192) sizes, hidden 5120
    for T in (256, 8192):
        H = 5120
        x = torch.randn((T, H), device=dev, dtype=torch.bfloat16)
        r = torch.randn((T, H), device=dev, dtype=torch.bfloat16)
        w = torch.ones(H, device=dev, dtype=torch.bfloat16)
        us = time_us(lambda: ops.fused_add_rmsnorm(x, r, w, 1e-6))
        res[f"fused_add_rmsnorm_T{T}"] = dict(us=us, GBps=(4 * T * H * it + H * it) / us / 1e3)
        us = time_us(lambda: ops.rmsnorm(x, w, 1e-6, out=r))
        res[f"rmsnorm_T{T}"] = dict(us=us, GBps=(2 * T * H * it + H * it) / us / 1e3)
        I = 17408
        gu = torch.randn((T, 2 * I), device=dev, dtype=torch.bfloat16)
        o = torch.empty((T, I), device=dev, dtype=torch.bfloat16)
        us = time_us(lambda: ops.silu_and_mul(gu, out=o))
        res[f"silu_and_mul_T{T}"] = dict(us=us, GBps=(3 * T * I * it) / us / 1e3)
        hq, hk, D = 40, 8, 128
        qkv = torch.randn((T, (hq + 2 * hk) * D), device=dev, dtype=torch.bfloat16)
        q_, k_, v_ = qkv.split([hq * D, hk * D, hk * D], dim=-1)
        cache = torch.randn((40960, D), device=dev, dtype=torch.float32)
        pos = torch.randint(0, 4096, (T,), device=dev, dtype=torch.int32)
        us = time_us(lambda: ops.rope_neox_inplace(pos, q_, k_, D, cache))
        res[f"rope_T{T}"] = dict(us=us, GBps=(2 * T * (hq + hk) * D * it + T * D * 4) / us / 1e3)
        kc = torch.zeros((300000, hk * D), device=dev, dtype=torch.bfloat16)
        vc = torch.zeros_like(kc)
        loc = torch.randperm(300000, device=dev)[:T].to(torch.int32)
        us = time_us(lambda: ops.store_kv(kc, vc, loc, k_, v_))
        res[f"store_kv_T{T}"] = dict(us=us, GBps=(4 * T * hk * D * it + T * 4) / us / 1e3)
        qw = torch.ones(D, device=dev, dtype=torch.bfloat16)
        us = time_us(lambda: ops.qk_norm_rope_store(q_, k_, v_, qw, qw, 1e-6, pos, cache, kc, vc, loc, D))
        res[f"qk_norm_rope_store_T{T}"] = dict(
            us=us, GBps=(2 * T * (hq + hk) * D * it + 3 * T * hk * D * it + T * D * 4) / us / 1e3)
        emb = torch.randn((151936, H), device=dev, dtype=torch.bfloat16)
        ids = torch.randint(0, 151936, (T,), device=dev, dtype=torch.int32)
        us = time_us(lambda: ops.embedding_gather(emb, ids))
        res[f"embedding_gather_T{T}"] = dict(us=us, GBps=(2 * T * H * it) / us / 1e3)
        for kname in list(res):
            if kname.endswith(f"_T{T}"):
                print(kname, res[kname], flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/microbench.json")
    ap.add_argument("--only", default="all", choices=["all", "decode", "prefill", "rows"])
    run(ap.parse_args())
