"""What the k-sliced plans of the small projections pay for their fp32 slabs: g3 with / without epilogue stores, with /
without the reduce launch, at the plans the search picks (qkv 4, o 6, down 6 k-slices), M = 256, rotating weights."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops
from mini_sglang_amd._lib import lib
from tools.g3_bench import time_us
dev = torch.device("cuda:0")
M = 256
for name, N, K, split in (("qkv", 7168, 5120, 4), ("o", 5120, 5120, 6), ("down", 5120, 17408, 6), ("o_3", 5120, 5120, 3), ("o_12", 5120, 5120, 12)):
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(8 if K < 10000 else 4)]
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    wsp = ops.gemm_workspace(dev)
    def slabs_only(w, variant=0):
        ops.check(lib().msgl_g3_gemm_nt(None, x.data_ptr(), w.data_ptr(), M, N, K, K, K, N, ops._dt(x), 256, 0, split,
                                        ops.G3_SLABS_ONLY | (variant << 8), wsp.data_ptr(), wsp.numel(), ops._stream()), "g3")
    t_full = time_us(lambda w: ops.g3_linear(x, w, 256, 0, split, out=out), ws)
    t_slabs = time_us(lambda w: slabs_only(w), ws)
    t_nostore = time_us(lambda w: ops.g3_linear(x, w, 256, 0, split, out=out, variant=16), ws)
    t_nocomp = time_us(lambda w: ops.g3_linear(x, w, 256, 0, split, out=out, variant=4), ws)
    print(f"{name:5s} N={N} K={K} {split} k-slices: gemm + reduce launch {t_full:6.1f} us | slabs only {t_slabs:6.1f} | no epilogue stores (+reduce) {t_nostore:6.1f} | "
          f"loads only (+stores, reduce) {t_nocomp:6.1f} | slab bytes {split * M * N * 4 / 1e6:.1f} MB, weights {2 * N * K / 1e6:.1f} MB", flush=True)
