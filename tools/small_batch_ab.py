"""Decode step at the smallest batch sizes, inside the captured graph, with and without the row-streaming projection
(csrc/gemm_rowstream.hip) and with and without the row kernels folded into its staging pass.

    python tools/small_batch_ab.py [--model qwen3-14b] [--batches 1 2 4 8] [--out gpurun_out/small_batch_ab.json]

configurations (same engine, plans re-searched / graphs re-captured in between; median of 20 timed replays each):
    skinny           the plan search without the row-streaming kernel (MSGL_DISABLE_ROWSTREAM=1): round-3 state
    rowstream        the search with it (what an engine start does now), row kernels as their own launches
    rowstream+fold   same plans, fused_add_rmsnorm / SiLU.mul folded where the consuming projection is the row-streaming kernel
    forced+fold      every projection the kernel supports on it (depth 8; vector units at B = 1, matrix cores above), folded:
                     what the folding is worth when nothing is left outside
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batches", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--out", default="gpurun_out/small_batch_ab.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ["MSGL_DISABLE_REFINE"] = "1"

    from mini_sglang_amd import _lib
    from mini_sglang_amd import model as model_mod
    from mini_sglang_amd import ops
    from mini_sglang_amd.core import Batch, Req
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.plan_refine import StepBench

    bss = sorted(args.batches)
    ecfg = EngineConfig(model=PRESETS[args.model], dtype=torch.bfloat16, max_running_req=max(bss), cuda_graph_bs=bss, page_size=256,
                        max_seq_len_override=4096, num_page_override=64 * max(bss), gemm_tune="off")
    eng = Engine(ecfg, dev)
    gr = eng.graph_runner

    def measure_all():
        out = {}
        for bs in bss:
            sb = StepBench(bs=bs, page_table=eng.page_table, page_size=ecfg.page_size, num_pages=eng.num_pages,
                           row_len=eng.aligned_max_seq_len, device=dev, Req=Req, Batch=Batch,
                           prepare_metadata=eng.attn_backend.prepare_metadata, capture=lambda bs=bs: gr.capture(bs), replay=gr.replay,
                           forward_ctx=eng.ctx.forward_batch)
            assert sb.ok
            try:
                out[bs] = round(sb.measure(args.replays), 4)
            finally:
                sb.close()
        return out

    def search(disable_rowstream: bool):
        os.environ["MSGL_DISABLE_ROWSTREAM"] = "1" if disable_rowstream else "0"
        ops.reset_gemm_plans()
        rep = eng.model.tune_gemms(bss, "heuristic")
        return {f"{r['name']}@{r['M']}": f"{r['kernel'][:60]} {r['best_us']:.1f} us" for r in rep}

    res = {"model": args.model, "batches": bss, "ms_per_step": {}, "plans": {}}
    try:
        res["plans"]["skinny"] = search(True)
        model_mod._ROWSTREAM_FUSE = False
        res["ms_per_step"]["skinny"] = measure_all()
        print("skinny", res["ms_per_step"]["skinny"], flush=True)
        model_mod._ROWSTREAM_FUSE = True  # the search credits the folds it will get
        res["plans"]["rowstream"] = search(False)
        model_mod._ROWSTREAM_FUSE = False
        res["ms_per_step"]["rowstream"] = measure_all()
        print("rowstream", res["ms_per_step"]["rowstream"], flush=True)
        model_mod._ROWSTREAM_FUSE = True
        res["ms_per_step"]["rowstream+fold"] = measure_all()
        print("rowstream+fold", res["ms_per_step"]["rowstream+fold"], flush=True)
        ws = [eng.model.lm_head] + [w for lw in eng.model.layers[:1] for w in (lw.qkv, lw.o, lw.gate_up, lw.down)]
        for M in bss:
            for w in ws:
                N, K = w.shape
                key = (w.device.index or 0, M, N, K, K, w.stride(0), _lib.BF16)
                v = 0 if M == 1 else 1  # vector units at one row, matrix cores above
                if ops.rowstream_supported(M, N, K, 0, v):
                    ops._SKINNY_PLAN[key] = (-v, 8)
                    ops._WSTREAM_PLAN.pop(key, None)
        res["ms_per_step"]["forced+fold"] = measure_all()
        print("forced+fold", res["ms_per_step"]["forced+fold"], flush=True)
        model_mod._ROWSTREAM_FUSE = False
        res["ms_per_step"]["forced"] = measure_all()
        print("forced", res["ms_per_step"]["forced"], flush=True)
        for k, v in res["plans"]["rowstream"].items():
            print(k, "|", res["plans"]["skinny"].get(k), "->", v)
    finally:
        eng.shutdown()
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
