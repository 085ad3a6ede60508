"""A few launches of the prefill attention kernels on one steady-state shape, for `rocprofv3 --pmc ...` passes
(tools/pmc_prefill.sh).  impl 2 = tr-read kernel, 4 = DMA-staged kernel; ablations as in tools/prefill_ablate.py."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from mini_sglang_amd import ops  # noqa: E402
from microbench import prefill_case  # noqa: E402

impls = [int(a) for a in sys.argv[1:]] or [2, 4]
dev = torch.device("cuda:0")
cases = {}
for impl in impls:
    qt = ops.prefill_q_tile(impl)
    if qt not in cases:
        cases[qt] = prefill_case([8192] * 2, [8192] * 2, 40, 8, 256, dev, q_tile=qt)
    c = cases[qt]
    for _ in range(3):
        ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"], c["tile_cu"], c["B"],
                         c["total_tiles"], 128 ** -0.5, tile_order=c["order"], impl=impl)
torch.cuda.synchronize()
print("flops per launch", c["flops"])
