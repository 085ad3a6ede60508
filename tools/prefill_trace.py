"""Per-segment clock stamps of the counter-phase prefill kernel (impl 5 + 128): where a period's cycles go.

    python tools/prefill_trace.py [--out gpurun_out/prefill_trace.json]

For the first 32 workgroups (the heaviest tiles) and periods 8..23 every wave records s_memtime at: period start,
after its first K fragment reads, after P.V, end of its matrix segment, after its DMA wait, after the barrier, end of
its softmax segment (the second barrier follows).  Reported: mean cycles of each piece for group A (waves 0-3) and group B (waves 4-7).
"""
import argparse
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from mini_sglang_amd import ops  # noqa: E402
from mini_sglang_amd._lib import check, lib  # noqa: E402
from microbench import prefill_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prefill_trace.json")
    ap.add_argument("--impl", type=int, default=128 + 128)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    c = prefill_case([8192] * 2, [8192] * 2, 40, 8, 256, dev, q_tile=256)
    stamps = torch.zeros(32 * 8 * 256, dtype=torch.int64, device=dev)
    run = lambda: ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"], c["tile_cu"],  # noqa: E731
                                   c["B"], c["total_tiles"], 128 ** -0.5, tile_order=c["order"], impl=args.impl)
    for _ in range(3):
        run()
    check(lib().msgl_attn_prefill_trace(stamps.data_ptr()), "trace")
    run()
    torch.cuda.synchronize()
    check(lib().msgl_attn_prefill_trace(None), "trace off")
    t = stamps.cpu().numpy().reshape(32, 8, 16, 16).astype(np.int64)
    res = {}
    for g, name in ((0, "group A (waves 0-3)"), (1, "group B (waves 4-7)")):
        x = t[:, 4 * g: 4 * g + 4]  # [wg, wave, period, 5]
        ok = (x[..., 0] > 0).all(axis=-1)
        x = x[ok]
        d = dict(
            matrix_segment=float((x[..., 1] - x[..., 0]).mean()), first_k_reads=float((x[..., 5] - x[..., 0]).mean()),
            pv_part=float((x[..., 6] - x[..., 5]).mean()), qk_part=float((x[..., 1] - x[..., 6]).mean()), dma_wait=float((x[..., 7] - x[..., 1]).mean()), barrier_1_wait=float((x[..., 2] - x[..., 7]).mean()),
            softmax_segment=float((x[..., 3] - x[..., 2]).mean()),
            sm_mask_max=float((x[..., 8] - x[..., 2]).mean()), sm_piece_k0=float((x[..., 9] - x[..., 8]).mean()),
            sm_exp_block0=float((x[..., 10] - x[..., 9]).mean()), sm_piece_k1=float((x[..., 11] - x[..., 10]).mean()),
            sm_exp_block1=float((x[..., 12] - x[..., 11]).mean()), sm_piece_v0=float((x[..., 13] - x[..., 12]).mean()),
            sm_sum_rescale=float((x[..., 14] - x[..., 13]).mean()), sm_piece_v1=float((x[..., 15] - x[..., 14]).mean()),
            sm_slots_pack_preread=float((x[..., 3] - x[..., 15]).mean()),
            period=float((x[:, 1:, 0] - x[:, :-1, 0]).mean()), barrier_2_wait=float((x[:, 1:, 0] - x[:, :-1, 3]).mean()),
            waves=int(ok.sum()))
        res[name] = {k: round(v, 1) if isinstance(v, float) else v for k, v in d.items()}
        print(name, res[name], flush=True)
    # one workgroup's timeline (wave 0 and wave 4), relative to wave 0's first stamp
    base = t[0, 0, 0, 0]
    res["wg0_wave0_periods_8_11"] = (t[0, 0, :4] - base).tolist()
    res["wg0_wave4_periods_8_11"] = (t[0, 4, :4] - base).tolist()
    print("wg 0 wave 0:", res["wg0_wave0_periods_8_11"])
    print("wg 0 wave 4:", res["wg0_wave4_periods_8_11"])
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
