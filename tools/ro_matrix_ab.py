"""Row-owner kernel (csrc/gemm_ro.hip): same-process A/B of its matrix side at M > 128 -- the software-pipelined body (default)
against read-all-then-multiply (flag bit 16 << 8), with and without the loads (ablation bits) -- bit equality of the two bodies
first.  python tools/ro_matrix_ab.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
from mini_sglang_amd import ops
from tools.ro_bench import time_us
dev = torch.device("cuda:0")
for name, M, N, K, plan in [("gate_up", 256, 34816, 5120, (256, 1)), ("gate_up", 128, 34816, 5120, (256, 1)), ("down", 256, 5120, 17408, (42, 6)), ("qkv", 256, 7168, 5120, (85, 3))]:
    nbuf = max(2, min(8, (600 << 20) // (N * K * 2) + 1))
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(nbuf)]
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    r = {}
    same = torch.equal(ops.ro_linear(x, ws[0], plan[0], plan[1]), ops.ro_linear(x, ws[0], plan[0], plan[1], ablate=16))
    r["both bodies give the same bits"] = same
    for rep in range(2):
        for label, abl in [("pipelined (default)", 0), ("read-all-then-multiply", 16), ("no loads, pipelined", 5), ("no loads, read-all-then-multiply", 21)]:
            r.setdefault(label, []).append(round(time_us(lambda w: ops.ro_linear(x, w, plan[0], plan[1], out, ablate=abl), ws), 1))
    print(name, M, r, flush=True)
