"""Same-box A/B of the decode attention kernels on the bench shape (boxes differ by a few percent, so variants are
only comparable inside one process): interleaved rounds, long warm-up (clocks), median of per-launch event timings.

    python tools/decode_ab.py [--impls 1,24,23,22,32,33,42] [--rounds 3] [--shape 14b]
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
from tools.microbench import bench_lens, decode_case  # noqa: E402

SHAPES = {"14b": (256, 40, 8), "14b_tp4": (256, 10, 2), "32b_tp4": (256, 16, 2), "0.6b": (256, 16, 8), "70b_tp8": (256, 8, 1),
          "14b_b32": (32, 40, 8)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--impls", default="1,24,23,22,32,33,42")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--shape", default="14b")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--scale", type=float, default=1.0, help="multiply every context length (marginal bandwidth)")
    ap.add_argument("--page", type=int, default=256, help="page size of the pool (slot_run = page if >= 16 else 1: token-granular table)")
    ap.add_argument("--alloc", default="shuffled", choices=["shuffled", "sequential"], help="pages handed out in shuffled / allocation order")
    ap.add_argument("--out", default="gpurun_out/decode_ab.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    impls = [int(x) for x in args.impls.split(",")]
    res = {}
    for shape in args.shape.split(","):
        B, hq, hkv = SHAPES[shape]
        lens = [max(1, int(n * args.scale)) for n in bench_lens(B)]
        k, v, table, q = decode_case(B, hq, hkv, lens, args.page, dev, shuffle=args.alloc == "shuffled")
        slot_run = args.page if args.page >= 16 else 1
        D, cap = 128, max(4096, 2 * B)
        ws = torch.empty(ops.attn_decode_workspace_bytes(cap, hq, D), dtype=torch.uint8, device=dev)
        seq = torch.tensor(lens, dtype=torch.int32, device=dev)
        out = torch.empty_like(q)
        S = sum(lens)
        bytes_ = S * 2 * hkv * D * 2 + 2 * B * hq * D * 2 + S * 4 + 2 * B * 4
        plans, outs = {}, {}
        for impl in impls:
            ops.attn_decode_select(impl)
            plans[impl] = torch.zeros(ops.attn_decode_plan_words(B, cap), dtype=torch.int32, device=dev)
            ops.attn_decode_plan(plans[impl], seq, B, B, cap, hq, hkv)
        times = {impl: [] for impl in impls}
        for rnd in range(args.rounds):
            for impl in impls:
                ops.attn_decode_select(impl)
                f = lambda: ops.attn_decode(out, q, k, v, table, None, seq, plans[impl], ws, B, B, cap, D ** -0.5,  # noqa: E731
                                            slot_run=slot_run)
                for _ in range(60 if rnd == 0 else 15):
                    f()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
                for e0, e1 in evs:
                    e0.record()
                    f()
                    e1.record()
                torch.cuda.synchronize()
                ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)
                times[impl].append(ts[len(ts) // 2])
                outs[impl] = out.clone()
        ops.attn_decode_select(0)
        ref = outs[impls[0]].float()
        res[shape] = {}
        for impl in impls:
            med = sorted(times[impl])[len(times[impl]) // 2]
            res[shape][impl] = dict(us=[round(t, 1) for t in times[impl]], median_us=med, GBps=bytes_ / med / 1e3,
                                    slots=int(plans[impl][3]), max_diff_vs_first=(outs[impl].float() - ref).abs().max().item())
            print(f"{shape} page {args.page} {args.alloc} impl {impl:>2}: {[round(t, 1) for t in times[impl]]} us  median {med:.1f}  {bytes_ / med / 1e3:.0f} GB/s  "
                  f"slots {int(plans[impl][3])}  maxdiff {res[shape][impl]['max_diff_vs_first']:.1e}", flush=True)
        del k, v, table, q, ws
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
