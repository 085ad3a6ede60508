"""Launch the paged decode attention kernel at the bench shape a few times (for rocprofv3
--kernel-trace / --pmc passes, and a float4-copy calibration kernel with a known byte count).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python tools/profile_attn.py
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402
from tools.microbench import bench_lens, decode_case  # noqa: E402

sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    import argparse

    ap = argparse.ArgumentParser()
    # bench.py measures its roofline launch after the default 5 + 40 steps: context = prompt + 1 + 45
    ap.add_argument("--advance", type=int, default=46)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, hq, hkv, D = 256, 40, 8, 128
    lens = [n + args.advance for n in bench.bench_contexts(B)]
    k, v, table, q = decode_case(B, hq, hkv, lens, 256, dev)
    cap = 4096
    plan = torch.zeros(ops.attn_decode_plan_words(B, cap), dtype=torch.int32, device=dev)
    ws = torch.empty(ops.attn_decode_workspace_bytes(cap, hq, D), dtype=torch.uint8, device=dev)
    seq = torch.tensor(lens, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    ops.attn_decode_plan(plan, seq, B, B, cap, hq, hkv)
    S = sum(lens)
    print("algorithmic_bytes_per_launch", S * 2 * hkv * D * 2 + 2 * B * hq * D * 2 + S * 4 + 2 * B * 4, flush=True)
    # calibration: a 1 GiB device-to-device copy (known: 1 GiB read + 1 GiB written)
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    for _ in range(10):
        ops.attn_decode(out, q, k, v, table, None, seq, plan, ws, B, B, cap, D ** -0.5, slot_run=256)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
