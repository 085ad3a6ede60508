"""gate_up + SiLU.mul at decode-sized batches: the weight-streaming kernel with the activation in its epilogue
(msgl_skinny_gemm_silu_nt, every setting) against the plain kernel's best setting followed by the activation kernel.

    python tools/skinny_silu_bench.py [--model qwen3-14b] [--batches 1 8 32]
Weights of several layers rotated (every launch streams from HBM); microseconds per launch (pair), back to back.
"""
import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402
from mini_sglang_amd.model import PRESETS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batches", type=int, nargs="*", default=[1, 8, 32])
    ap.add_argument("--out", default="gpurun_out/skinny_silu_bench.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = PRESETS[args.model]
    N, K = 2 * m.intermediate_size, m.hidden_size
    ws = [ops.interleave_gate_up((torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16)) for _ in range(4)]
    res = {}
    for M in args.batches:
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        full, half = torch.empty((M, N), dtype=torch.bfloat16, device=dev), torch.empty((M, N // 2), dtype=torch.bfloat16, device=dev)
        t = lambda f: ops._time_launches_us(f, ws, 12, 3)  # noqa: E731
        plain = {f"{sl}/{nt}": t(lambda w: ops.skinny_linear(x, w, sl, full, nt)) for sl, nt in ops.skinny_candidates(M, N, K)}
        bp = min(plain, key=plain.get)
        sl0, nt0 = map(int, bp.split("/"))
        pair = t(lambda w: ops.silu_and_mul_interleaved(ops.skinny_linear(x, w, sl0, full, nt0), half))
        fused = {f"{sl}/{nt}": t(lambda w: ops.skinny_linear_silu(x, w, sl, half, nt)) for sl, nt in ops.skinny_silu_candidates(M, N, K)}
        bf = min(fused, key=fused.get)
        res[M] = dict(plain_best=bp, plain_us=round(plain[bp], 1), plain_plus_activation_us=round(pair, 1), fused_best=bf,
                      fused_us=round(fused[bf], 1), fused_all={k: round(v, 1) for k, v in fused.items()},
                      plain_all={k: round(v, 1) for k, v in plain.items()})
        print(M, json.dumps(res[M]), flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
