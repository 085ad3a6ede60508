"""A/B of kernel choices INSIDE the captured decode step (Qwen3-14B, 256 sequences, synthetic contexts of the bench's mean):
what a variant is worth where it runs -- cold L2 after the attention stream, its consumer in the same measurement -- not
back to back.  Same machinery as the in-graph re-ranking of the projection plans (plan_refine.StepBench): every setting is
re-captured and replayed, rounds interleaved, median of individually timed replays.

    python tools/step_ab.py [--model qwen3-14b] [--rounds 3] [--replays 20] [--out gpurun_out/step_ab.json] [--knobs decode72 ...]

knobs:
    decode72      paged decode attention with the split-KV merge folded into the last-arriving piece (msgl_attn_decode_select 72)
                  instead of the separate merge kernel
    decode71      the default variant by its number (72's A/B partner)
    decode1       the streaming (VALU) decode attention kernel
    no_slab_norm  o_proj / down_proj slabs reduced by their own launch instead of by the following norm
    lib_o / lib_qkv / lib_down / lib_gate_up     the library's best solution for that projection instead of the planned kernel
    plan:<o|qkv|down|gate_up>:<grid>/<whole tiles>/<k-slices>/<0 m256 | 1 g3>    that full-batch plan instead of the planned kernel
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
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--knobs", nargs="*", default=["decode72", "lib_o", "lib_gate_up"])
    ap.add_argument("--out", default="gpurun_out/step_ab.json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MSGL_DISABLE_REFINE", "1")  # the plans of the back-to-back search stay; this tool does the A/Bs

    from mini_sglang_amd import model as model_mod
    from mini_sglang_amd import ops
    from mini_sglang_amd.core import Batch, Req
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.plan_refine import StepBench

    B = args.batch
    ecfg = EngineConfig(model=PRESETS[args.model], dtype=torch.bfloat16, max_running_req=B, cuda_graph_bs=[B], page_size=256,
                        max_seq_len_override=4096, memory_ratio=0.9, gemm_tune=os.environ.get("MSGL_GEMM_TUNE", "full"))
    eng = Engine(ecfg, dev)
    gr = eng.graph_runner
    sb = StepBench(bs=B, page_table=eng.page_table, page_size=ecfg.page_size, num_pages=eng.num_pages,
                   row_len=eng.aligned_max_seq_len, device=dev, Req=Req, Batch=Batch,
                   prepare_metadata=eng.attn_backend.prepare_metadata, capture=lambda: gr.capture(B), replay=gr.replay,
                   forward_ctx=eng.ctx.forward_batch)
    assert sb.ok
    keys = {ops._CANDIDATES[k]["name"]: k for k in ops._CANDIDATES if k[1] == B}

    def lib_knob(name):
        key = keys[name]
        snap = ops.snapshot_plan(key)
        return (lambda: ops.apply_candidate(key, ("lib", 0))), (lambda: ops.restore_search_pick(key, snap))

    knobs = {
        "decode72": (lambda: ops.attn_decode_select(72), lambda: ops.attn_decode_select(0)),
        "decode71": (lambda: ops.attn_decode_select(71), lambda: ops.attn_decode_select(0)),
        "decode1": (lambda: ops.attn_decode_select(1), lambda: ops.attn_decode_select(0)),
        "no_slab_norm": (lambda: setattr(model_mod, "_SLAB_NORM", False), lambda: setattr(model_mod, "_SLAB_NORM", True)),
    }
    for name in ("o", "qkv", "down", "gate_up"):
        if name in keys:
            knobs["lib_" + name] = lib_knob(name)

    def plan_knob(spec):  # "plan:<projection>:<grid>/<whole tiles>/<k-slices>/<impl 0 m256 | 1 g3>", e.g. plan:o:256/0/3/1
        _, name, plan = spec.split(":")
        key, p4 = keys[name], tuple(int(v) for v in plan.split("/"))
        snap = ops.snapshot_plan(key)
        return (lambda: ops.apply_candidate(key, ("hand", p4))), (lambda: ops.restore_search_pick(key, snap))

    for kn in args.knobs:
        if kn.startswith("plan:"):
            knobs[kn] = plan_knob(kn)
        elif kn.startswith("decode:"):  # any msgl_attn_decode_select code, e.g. decode:73
            code = int(kn.split(":")[1])
            knobs[kn] = ((lambda c=code: ops.attn_decode_select(c)), (lambda: ops.attn_decode_select(0)))
    res = {"model": args.model, "batch": B, "plans": {n: ops.current_candidate(k) for n, k in keys.items()}, "base_ms": [], "knobs": {}}
    try:
        for rnd in range(args.rounds):
            res["base_ms"].append(round(sb.measure(args.replays), 4))
            for kn in args.knobs:
                on, off = knobs[kn]
                on()
                try:
                    ms = sb.measure(args.replays)
                finally:
                    off()
                res["knobs"].setdefault(kn, []).append(round(ms, 4))
            print(f"round {rnd}: base {res['base_ms'][-1]} " + " ".join(f"{k} {v[-1]}" for k, v in res["knobs"].items()), flush=True)
        res["base_ms"].append(round(sb.measure(args.replays), 4))
        base = sorted(res["base_ms"])[len(res["base_ms"]) // 2]
        res["summary"] = {"base_median_ms": base, **{k: dict(median_ms=sorted(v)[len(v) // 2], delta_ms=round(sorted(v)[len(v) // 2] - base, 4))
                                                    for k, v in res["knobs"].items()}}
        print(json.dumps(res["summary"]), flush=True)
    finally:
        sb.close()
        eng.shutdown()
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
