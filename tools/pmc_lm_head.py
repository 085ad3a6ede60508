"""The LM head at the full decode batch ([256, 5120] x [151936, 5120]^T, 1.556 GB of weights) under rocprofv3 counters: the
library's heuristic solution (what the search also keeps, MT192x256x64) and the row-owner kernel on its best plan, launched a
few times each on a rotating pair of weights, with three 1-GiB copies first for the FETCH_SIZE calibration (VERDICT r5 item 3c:
"own it, or the counters that say why not").

    rocprofv3 --pmc FETCH_SIZE -d out -- python tools/pmc_lm_head.py
    rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum -d out -- python tools/pmc_lm_head.py
    rocprofv3 --kernel-trace --stats -d out -- python tools/pmc_lm_head.py
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
    M, N, K = 256, 151936, 5120
    bf = torch.bfloat16
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(bf) for _ in range(2)]
    x = torch.randn((M, K), device=dev).to(bf)
    a = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    out = torch.empty((M, N), dtype=bf, device=dev)
    for i in range(6):  # the library's heuristic solution
        ops.linear(x, ws[i % 2], out)
    torch.cuda.synchronize()
    cus = int(ops.lib().msgl_device_cu_count())
    plans = [p for p in ops.ro_candidates(M, N, K, cus) if p[1] == 1][:1] or ops.ro_candidates(M, N, K, cus)[:1]
    print("algorithmic_bytes", 2 * N * K + 2 * M * K + 2 * M * N, "ro plan", plans[0], flush=True)
    for i in range(6):
        ops.ro_linear(x, ws[i % 2], plans[0][0], plans[0][1], out)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
