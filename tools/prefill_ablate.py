"""Where the prefill attention kernels' time goes: the register-staged kernel (csrc/attn_prefill.hip, impl 2) and the
DMA-staged default (impl 4) with parts of their loops removed (impl 16 + bits / 64 + bits; results are wrong by
construction, timing only).  The ablation variants exist only in a diagnostic build of the library:

    MSGL_PREFILL_DIAG=1 python mini-sglang_amd/build.py --force && python tools/prefill_ablate.py [--out gpurun_out/prefill_ablate.json]
    python tools/prefill_ablate.py --only "tr: full" "dma: full"          (production build: the two kernels, no ablations)

bits: 1 = K/V tile loaded once (no global loads / LDS writes in the loop), 2 = no QK^T MFMAs (and no K fragment reads),
4 = no softmax arithmetic, 8 = no PV MFMAs (and no V^T fragment reads), 16 = no barrier in the loop.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from mini_sglang_amd import ops  # noqa: E402
from microbench import MFMA_PEAK_TFLOPS, prefill_case, prefill_chunk_lens, time_us  # noqa: E402

VARIANTS = {
    0: "full", 32: "full, one workgroup per CU", 1: "no K/V stream", 16: "no barrier", 17: "no stream, no barrier", 4: "no softmax", 2: "no QK^T", 8: "no PV",
    6: "no QK^T, no softmax (PV + stream)", 12: "no softmax, no PV (QK^T + stream)", 10: "no MFMA (softmax + stream)",
    14: "stream + barrier only", 15: "loop skeleton", 31: "skeleton, no barrier",
}


DMA_VARIANTS = {0: "full", 16: "full, s_setprio 1 on the MFMA blocks", 32: "full, s_setprio 1 on the softmax", 1: "no K/V stream", 4: "no softmax", 2: "no QK^T", 8: "no PV", 6: "PV + stream", 12: "QK^T + stream",
                10: "softmax + stream", 14: "stream + barrier only", 15: "loop skeleton"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prefill_ablate.json")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", nargs="*", help="label prefixes to run, e.g. 'dma: full' 'tr: full'")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import bench as _bench

    chunk = prefill_chunk_lens(16384, _bench.bench_contexts(256))
    cases = [("14b_8x2048", [2048] * 8, [2048] * 8, 40, 8), ("14b_chunk16384_bench", chunk, chunk, 40, 8),
             ("14b_2x8192", [8192] * 2, [8192] * 2, 40, 8)]
    res = {}
    for name, ql, kl, hq, hkv in cases:
        c = prefill_case(ql, kl, hq, hkv, 256, dev)

        def launch(cc, impl):
            ops.attn_prefill(cc["out"], cc["q"], cc["k"], cc["v"], cc["table"], None, cc["seq"], cc["cu_q"], cc["tile_cu"],
                             cc["B"], cc["total_tiles"], 128 ** -0.5, tile_order=cc["order"], impl=impl)

        row, outs = {}, {}
        for impl, cc in ((2, c), (4, c)):
            cc["out"].zero_()
            launch(cc, impl)
            outs[impl] = cc["out"].clone()
        row["dma == tr (bitwise)"] = bool(torch.equal(outs[2], outs[4]))
        runs = [("tr: " + label, c, (16 + bits) if bits else 2) for bits, label in VARIANTS.items()]
        runs += [("dma: " + label, c, (64 + bits) if bits else 4) for bits, label in DMA_VARIANTS.items()]
        if args.only:  # exact labels ("dma: full") or prefixes ending in a space ("dma: ")
            runs = [r for r in runs if any(r[0] == o or (o.endswith(" ") and r[0].startswith(o)) for o in args.only)]
        best = {}
        for _ in range(args.rounds):  # variants interleaved, best of the rounds: the first launches after a change of kernel run slower
            for label, cc, impl in runs:
                us = time_us(lambda: launch(cc, impl), iters=8, warmup=4)
                best[label] = min(best.get(label, 1e30), us)
        for label, us in best.items():
            row[label] = dict(us=round(us, 1), frac_if_full=round(c["flops"] / us / 1e6 / MFMA_PEAK_TFLOPS, 3))
        res[name] = dict(flops=c["flops"], q_tiles=c["total_tiles"], variants=row)
        print(name, json.dumps(row), flush=True)
        del c
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
