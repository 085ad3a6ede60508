"""Per-step cost of the full-batch projection kernel under its ablation switches (profiles/r02b_m256_gemm_ablation.txt).

    MSGL_M256_ABLATE=<bits> python tools/m256_ablate.py     bits: 1 no x loads, 2 no compute, 4 no w loads, 8 nt w loads

N = 32768 (one whole tile per workgroup), K = 5120 and 10240: the difference gives the cost of 80 steps.
"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mini_sglang_amd import ops
dev = torch.device("cuda:0")
def t_us(fn, iters=300):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
N = 32768
res = {}
for K in (5120, 10240):
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
    x = torch.randn((256, K), device=dev).to(torch.bfloat16)
    out = torch.empty((256, N), dtype=torch.bfloat16, device=dev)
    i = [0]
    def fn():
        i[0] += 1
        ops.m256_linear(x, ws[i[0] % 3], 256, 256, 1, out=out)
    res[K] = t_us(fn)
    del ws
print(f"ABL {os.environ.get('MSGL_M256_ABLATE','0')}: K=5120 {res[5120]:.1f} us, K=10240 {res[10240]:.1f} us -> per step {(res[10240]-res[5120])/80*1000:.0f} ns, fixed {2*res[5120]-res[10240]:.1f} us")
