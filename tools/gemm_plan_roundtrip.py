"""A searched library solution carried to another plan table as data (ops.export_gemm_plans / import_gemm_plans over
msgl_gemm_get_plan / msgl_gemm_set_plan): same bits before and after, plain and split-K solutions.  python tools/gemm_plan_roundtrip.py"""
import sys, torch, ctypes as C
sys.path.insert(0, "/root/repo")
from mini_sglang_amd import ops, _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(256, 5120, 17408), (1, 5120, 5120), (256, 32768, 5120)]:
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(2)]
    r = ops.gemm_tune(x, ws, max_candidates=-16, iters=4)
    y1 = ops.linear(x, ws[0]).clone()
    code = ops._dt(x)
    plans = ops.export_gemm_plans([(M, N, K, K, K, N, code)])
    print(M, N, K, r["kernel"][:60], plans["library"], flush=True)
    ops.import_gemm_plans(plans)
    y2 = ops.linear(x, ws[0])
    print("  same bits after import:", torch.equal(y1, y2), flush=True)
    ops.reset_gemm_plans()
