"""Launch the two hand-written decode GEMM kernels at Qwen3-14B shapes (for rocprofv3 --kernel-trace / --pmc
FETCH_SIZE passes): skinny at M = 8, wstream at M = 64, rotating weights so every launch streams from HBM, plus
three 1-GiB device copies as the FETCH_SIZE calibration (MI355X_MICROARCH.md: gfx950 reports half of a wide read).

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_gemm -- python tools/profile_gemm.py
"""
from __future__ import annotations

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    shapes = [("qkv", 7168, 5120), ("o", 5120, 5120), ("gate_up", 34816, 5120), ("down", 5120, 17408)]
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    for name, N, K in shapes:
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(torch.bfloat16) for _ in range(4)]
        for M, kind in ((8, "skinny"), (64, "wstream")):
            x = torch.randn((M, K), device=dev, dtype=torch.float32).to(torch.bfloat16)
            cands = ops.skinny_candidates(M, N, K) if kind == "skinny" else ops.wstream_candidates(M, N, K)
            # a mid-table setting per kernel: the point is bytes per launch, not the tuned time
            sl_or_nt, second = (4, 2) if kind == "skinny" else (1, 4)
            if (sl_or_nt, second) not in cands:
                sl_or_nt, second = cands[0]
            algo = 2 * N * K + 2 * M * K + 2 * M * N
            print(f"{kind} {name} M={M} N={N} K={K} setting=({sl_or_nt},{second}) algorithmic_bytes={algo}", flush=True)
            for i in range(8):
                if kind == "skinny":
                    ops.skinny_linear(x, ws[i % 4], sl_or_nt, row_tiles=second)
                else:
                    ops.wstream_linear(x, ws[i % 4], sl_or_nt, second)
        del ws
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
