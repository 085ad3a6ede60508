"""Where the prefill attention kernel's time goes: the tr-read kernel (csrc/attn_prefill.hip, impl 2) with parts of its
loop removed (impl 16 + bits; results are wrong by construction, timing only).

    python tools/prefill_ablate.py [--out gpurun_out/prefill_ablate.json]

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


DMA_VARIANTS = {0: "full", 16: "full, no s_setprio", 32: "full, s_setprio 1 on the softmax instead of the MFMA blocks", 1: "no K/V stream", 4: "no softmax", 2: "no QK^T", 8: "no PV", 6: "PV + stream", 12: "QK^T + stream",
                10: "softmax + stream", 14: "stream + barrier only", 15: "loop skeleton"}


PP_VARIANTS = {11: "softmax only (no MFMA, no LDS reads, no stream)", 5: "MFMA + LDS reads only (no softmax, no stream)", 0: "full", 16: "full, softmax segment at priority 1", 32: "full, matrix segment at priority 1",
               64: "full, group B at priority 1", 1: "no K/V stream", 4: "no softmax", 2: "no QK^T", 8: "no PV", 10: "softmax + stream",
               14: "stream + barriers only", 15: "loop skeleton"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/prefill_ablate.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import bench as _bench

    chunk = prefill_chunk_lens(16384, _bench.bench_contexts(256))
    cases = [("14b_8x2048", [2048] * 8, [2048] * 8, 40, 8), ("14b_chunk16384_bench", chunk, chunk, 40, 8),
             ("14b_2x8192", [8192] * 2, [8192] * 2, 40, 8)]
    res = {}
    for name, ql, kl, hq, hkv in cases:
        c = prefill_case(ql, kl, hq, hkv, 256, dev)
        row = {}
        for bits, label in VARIANTS.items():
            f = lambda: ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"],  # noqa: E731
                                         c["tile_cu"], c["B"], c["total_tiles"], 128 ** -0.5, tile_order=c["order"],
                                         impl=(16 + bits) if bits else 2)
            us = time_us(f, iters=10, warmup=10)
            row[label] = dict(us=round(us, 1), frac_if_full=round(c["flops"] / us / 1e6 / MFMA_PEAK_TFLOPS, 3))
        outs = {}
        for impl in (2, 4):
            c["out"].zero_()
            ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"], c["tile_cu"], c["B"],
                             c["total_tiles"], 128 ** -0.5, tile_order=c["order"], impl=impl)
            outs[impl] = c["out"].clone()
        row["dma == tr (bitwise)"] = bool(torch.equal(outs[2], outs[4]))
        row["dma vs tr max abs diff"] = float((outs[2].float() - outs[4].float()).abs().max())
        for bits, label in DMA_VARIANTS.items():
            f = lambda: ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"],  # noqa: E731
                                         c["tile_cu"], c["B"], c["total_tiles"], 128 ** -0.5, tile_order=c["order"],
                                         impl=(64 + bits) if bits else 4)
            us = time_us(f, iters=10, warmup=10)
            row["dma: " + label] = dict(us=round(us, 1), frac_if_full=round(c["flops"] / us / 1e6 / MFMA_PEAK_TFLOPS, 3))
        c2 = prefill_case(ql, kl, hq, hkv, 256, dev, q_tile=256)
        for k_ in ("q", "k", "v", "table"):
            c2[k_] = c[k_]
        c2["out"].zero_()
        ops.attn_prefill(c2["out"], c2["q"], c2["k"], c2["v"], c2["table"], None, c2["seq"], c2["cu_q"], c2["tile_cu"], c2["B"],
                         c2["total_tiles"], 128 ** -0.5, tile_order=c2["order"], impl=5)
        row["pp == tr (bitwise)"] = bool(torch.equal(outs[2], c2["out"]))
        row["pp vs tr max abs diff"] = float((outs[2].float() - c2["out"].float()).abs().max())
        for bits, label in PP_VARIANTS.items():
            f = lambda: ops.attn_prefill(c2["out"], c2["q"], c2["k"], c2["v"], c2["table"], None, c2["seq"], c2["cu_q"],  # noqa: E731
                                         c2["tile_cu"], c2["B"], c2["total_tiles"], 128 ** -0.5, tile_order=c2["order"],
                                         impl=(128 + bits) if bits else 5)
            us = time_us(f, iters=10, warmup=10)
            row["pp: " + label] = dict(us=round(us, 1), frac_if_full=round(c["flops"] / us / 1e6 / MFMA_PEAK_TFLOPS, 3))
        res[name] = dict(flops=c["flops"], q_tiles=c["total_tiles"], variants=row)
        print(name, json.dumps(row), flush=True)
        del c
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
