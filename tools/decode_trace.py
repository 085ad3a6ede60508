"""Where a wave of the matrix-core decode attention kernel spends its time on the bench shape: shader-clock stamps
per wave (msgl_attn_decode_trace + variant 93), reduced to phase statistics.

    python tools/decode_trace.py [--out gpurun_out/decode_trace.json]
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402
from mini_sglang_amd._lib import check, lib  # noqa: E402
from tools.microbench import bench_lens, decode_case  # noqa: E402


def stat(x):
    x = np.asarray(x, dtype=np.float64)
    return dict(mean=float(x.mean()), p50=float(np.median(x)), p95=float(np.percentile(x, 95)), max=float(x.max()))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/decode_trace.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, hq, hkv, D = 256, 40, 8, 128
    lens = bench_lens(B)
    k, v, table, q = decode_case(B, hq, hkv, lens, 256, dev)
    cap = max(4096, 2 * B)
    ws = torch.empty(ops.attn_decode_workspace_bytes(cap, hq, D), dtype=torch.uint8, device=dev)
    seq = torch.tensor(lens, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    ops.attn_decode_select(93)
    plan = torch.zeros(ops.attn_decode_plan_words(B, cap), dtype=torch.int32, device=dev)
    ops.attn_decode_plan(plan, seq, B, B, cap, hq, hkv)
    stamps = torch.zeros((4096, 16), dtype=torch.int64, device=dev)
    check(lib().msgl_attn_decode_trace(stamps.data_ptr()), "trace")
    f = lambda: ops.attn_decode(out, q, k, v, table, None, seq, plan, ws, B, B, cap, D ** -0.5, slot_run=256)  # noqa: E731
    for _ in range(50):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stamps.zero_()
    torch.cuda.synchronize()
    e0.record()
    f()
    e1.record()
    torch.cuda.synchronize()
    check(lib().msgl_attn_decode_trace(None), "trace off")
    ops.attn_decode_select(0)
    st = stamps.cpu().numpy().astype(np.int64)
    live = st[:, 15] != 0
    wave_in_block = np.nonzero(live)[0] % 8
    st = st[live]
    n = st[:, 14]
    pieces = (n - 2) // 4
    life = st[:, 15] - st[:, 0]
    event_us = e0.elapsed_time(e1) * 1e3
    res = dict(waves=int(live.sum()), event_us_kernel_plus_merge=event_us, pieces_per_wave=stat(pieces),
               wave_lifetime_clocks=stat(life),
               lifetime_by_wave_of_the_workgroup=[float(life[wave_in_block == w].mean()) for w in range(8)],
               slot_known=stat(st[:, 1] - st[:, 0]))
    for i in range(3):
        has = pieces > i
        if not has.any():
            break
        s = st[has]
        base = 2 + 4 * i
        prev = s[:, 1] if i == 0 else s[:, base - 1]
        res[f"piece{i}"] = dict(waves=int(has.sum()), metadata=stat(s[:, base] - prev), first_tile=stat(s[:, base + 1] - s[:, base]),
                                stream=stat(s[:, base + 2] - s[:, base + 1]), store=stat(s[:, base + 3] - s[:, base + 2]))
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
