"""Row-owner kernel (csrc/gemm_ro.hip): what its memory engine and its matrix side cost on their own -- the ablation switches
of msgl_ro_gemm_nt (flags bits 8-15) timed back to back on rotating weights: everything; no MFMAs (both streams); weight stream
only; x re-reads only; MFMAs + LDS only (no loads).

    python tools/ro_ablate.py [--out gpurun_out/ro_ablate.json]
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
from tools.ro_bench import time_us  # noqa: E402

CASES = [("gate_up", 256, 34816, 5120, (256, 1)), ("gate_up", 128, 34816, 5120, (256, 1)), ("gate_up", 64, 34816, 5120, (256, 1)),
         ("qkv", 256, 7168, 5120, (85, 3)), ("down", 256, 5120, 17408, (42, 6)), ("o", 256, 5120, 5120, (51, 5)),
         ("down", 128, 5120, 17408, (42, 6)), ("qkv", 128, 7168, 5120, (64, 4))]
VARIANTS = [("everything", 0), ("no MFMAs (both streams)", 2), ("weight stream only", 3), ("x re-reads only", 6), ("MFMAs + LDS reads only", 5)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/ro_ablate.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    res = []
    for name, M, N, K, plan in CASES:
        nbuf = max(2, min(8, (600 << 20) // (N * K * 2) + 1))
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
        x = torch.randn((M, K), device=dev, dtype=torch.float32).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        row = dict(name=name, M=M, N=N, K=K, plan=plan, weight_MB=round(N * K * 2 / 1e6, 1), us={})
        for label, abl in VARIANTS:
            row["us"][label] = round(time_us(lambda w: ops.ro_linear(x, w, plan[0], plan[1], out, ablate=abl), ws), 1)
        res.append(row)
        print(f"{name:8s} M={M:3d} plan {plan}: " + "; ".join(f"{k} {v}" for k, v in row["us"].items()), flush=True)
        del ws
        torch.cuda.empty_cache()
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
