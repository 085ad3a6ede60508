"""Round-4 experiments on the k-sliced full-batch projection kernel (csrc/gemm_g3.hip), M = 256, rotating weights:

  * slab-store cache policy: plain (variant 0) / sc1 write-through (32) / nt (33) -- GEMM alone and GEMM + the slab-consuming
    fused-add RMSNorm that follows it in the decode step (the pair is what the step pays: the kernel boundary behind a
    kernel that leaves tens of MB dirty in L2 is part of the cost);
  * loader roles: every loader wave issues a quarter of the x AND of the w pieces (0) / two loaders x only, two w only (64);
  * both (96).
Every variant must produce the same slabs bit for bit (checked through the reduced output).

    python tools/g3_exp.py [--out gpurun_out/g3_exp.json]
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
from mini_sglang_amd._lib import check, lib  # noqa: E402
from tools.g3_bench import time_us  # noqa: E402

VARIANTS = {0: "plain stores, mixed loaders", 32: "sc1 slab stores", 33: "nt slab stores", 64: "split loaders",
            96: "split loaders + sc1 slab stores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/g3_exp.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    M = 256
    res = {}
    wsp = ops.gemm_workspace(dev)
    for name, N, K, split in (("qkv", 7168, 5120, 4), ("o", 5120, 5120, 6), ("down", 5120, 17408, 6),
                              ("gate_up tail only (16 tiles x 16 slices)", 2048, 5120, 16)):
        ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(8 if K < 10000 else 4)]
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        resid = torch.randn((M, N), device=dev).to(torch.bfloat16)
        nw = torch.ones(N, dtype=torch.bfloat16, device=dev)
        y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)

        def slabs_only(w, variant):
            check(lib().msgl_g3_gemm_nt(None, x.data_ptr(), w.data_ptr(), M, N, K, K, K, N, ops._dt(x), 256, 0, split,
                                        ops.G3_SLABS_ONLY | (variant << 8), wsp.data_ptr(), wsp.numel(), ops._stream()), "g3")

        def pair(w, variant):
            slabs_only(w, variant)
            check(lib().msgl_fused_add_rmsnorm_slabs(y.data_ptr(), resid.data_ptr(), nw.data_ptr(), 1e-6, M, N, N, N,
                                                     wsp.data_ptr(), split, M * N, N, ops._dt(x), ops._stream()), "norm")

        row = dict(N=N, K=K, k_slices=split, slab_MB=round(split * M * N * 4 / 1e6, 1), weights_MB=round(2 * N * K / 1e6, 1))
        ref = ops.g3_linear(x, ws[0], 256, 0, split, out=out).clone()
        for v, label in VARIANTS.items():
            got = ops.g3_linear(x, ws[0], 256, 0, split, variant=v)
            same = bool(torch.equal(got, ref))
            t_gemm = time_us(lambda w: slabs_only(w, v), ws)
            t_pair = time_us(lambda w: pair(w, v), ws) if N <= 8192 else None
            row[label] = dict(bit_identical=same, gemm_slabs_only_us=round(t_gemm, 1), gemm_plus_norm_us=None if t_pair is None else round(t_pair, 1))
        t_norm_plain = time_us(lambda w: ops.fused_add_rmsnorm(y, resid, nw, 1e-6), ws)
        row["fused_add_rmsnorm without slabs us"] = round(t_norm_plain, 1)
        res[name] = row
        print(name, json.dumps(row), flush=True)
        del ws
        torch.cuda.empty_cache()
    # whole-tile regime (gate_up: 256 whole tiles + k-sliced tail; N = 32768: exactly one whole tile per workgroup)
    for name, N, K, plan in (("gate_up 256 whole + 16x16 tail", 34816, 5120, (256, 256, 16)), ("N=32768 one whole tile per CU", 32768, 5120, (256, 256, 1))):
        ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ref = ops.g3_linear(x, ws[0], *plan, out=out).clone()
        row = {}
        for v in (0, 64):
            got = ops.g3_linear(x, ws[0], *plan, variant=v)
            row[VARIANTS[v]] = dict(bit_identical=bool(torch.equal(got, ref)),
                                    us=round(time_us(lambda w: ops.g3_linear(x, w, *plan, out=out, variant=v), ws, iters=40, warm=80), 1))
        res[name] = row
        print(name, json.dumps(row), flush=True)
        del ws
        torch.cuda.empty_cache()
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
