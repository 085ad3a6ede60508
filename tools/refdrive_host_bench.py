"""TEST INFRASTRUCTURE (imports oracle/_ref through tests/refdrive.py): the reference's own LLM / Scheduler running the
README offline benchmark on Qwen3-0.6B dims (256 requests, in/out 100-1024, page 256, dummy weights, temperature 0.6)
through minisgl_plugin.install(), for the four combinations of

    f4 off / on   vectorised scheduler glue + native radix tree + deterministic decode order
    overlap scheduling on / off   MINISGL_DISABLE_OVERLAP_SCHEDULING (P/env.py:69, P/scheduler/scheduler.py:121-131)

Reports, per combination: README throughput (sum of max_tokens / wall, prefill included), ms per decode step between
consecutive forwards (GPU events), and the HOST microseconds per scheduler iteration spent in _schedule_next_batch
(incl. prepare_metadata), _forward (launches) and _process_last_data.  VERDICT r2 "next" 6.

    python tools/refdrive_host_bench.py [--out gpurun_out/r03_refdrive_0p6b_host.json] [--max-out 1024]
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main():
    import refdrive

    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-0.6b")
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--max-out", type=int, default=1024)
    ap.add_argument("--out", default="gpurun_out/r03_refdrive_0p6b_host.json")
    a = ap.parse_args()
    assert refdrive.reference_root() is not None, "needs oracle/_ref (oracle/build_ref.sh)"
    prompts, outs = refdrive.offline_bench_requests(a.n, max_out=a.max_out)
    sampling = [dict(temperature=0.6, max_tokens=o, ignore_eos=True) for o in outs]
    kw = dict(page_size=256, max_running_req=256, cuda_graph_max_bs=256, max_seq_len_override=4096, max_extend_tokens=16384,
              cache_type="radix")
    results = {}
    for f4 in (False, True):
        for overlap in (True, False):
            name = f"f4_{'on' if f4 else 'off'}__overlap_{'on' if overlap else 'off'}"
            if overlap:
                os.environ.pop("MINISGL_DISABLE_OVERLAP_SCHEDULING", None)
            else:
                os.environ["MINISGL_DISABLE_OVERLAP_SCHEDULING"] = "1"
            spec = dict(model=a.model, weights="dummy", llm_kwargs=kw, gemm_tune="heuristic", record="timing", host_timing=True,
                        vectorized_glue=f4, native_radix=f4, deterministic_decode_order=f4,
                        rounds=[dict(prompts=prompts[:8], sampling=sampling[:8]),      # warm-up as the reference bench does
                                dict(prompts=prompts, sampling=sampling)])
            rec = refdrive.run_worker(spec, timeout=900.0)
            fw = [f for f in rec["forwards"] if f["round"] == 1]
            dec = [f["ms_to_next"] for f in fw if f["phase"] == "decode" and f.get("ms_to_next")]
            full = [f["ms_to_next"] for f in fw if f["phase"] == "decode" and f.get("ms_to_next") and f["size"] >= 0.9 * a.n]
            h = rec["host"]
            it = max(h["iterations"], 1)
            results[name] = dict(
                throughput_tok_s=sum(outs) / rec["walls"][1], wall_s=rec["walls"][1], decode_steps=len(dec),
                ms_per_decode_step_mean=statistics.mean(dec) if dec else None,
                ms_per_decode_step_median=statistics.median(dec) if dec else None,
                ms_per_full_batch_decode_step_median=statistics.median(full) if full else None,
                host_us_per_iteration=dict(schedule=h["schedule_s"] / it * 1e6, forward=h["forward_s"] / it * 1e6,
                                           process=h["process_s"] / it * 1e6,
                                           total=(h["schedule_s"] + h["forward_s"] + h["process_s"]) / it * 1e6),
                iterations=h["iterations"], prefix_cache=rec["prefix_cache"], integrity=rec["integrity"])
            r = results[name]
            print(f"{name}: {r['throughput_tok_s']:.0f} tok/s, {r['ms_per_decode_step_mean']:.3f} ms/decode step (median "
                  f"{r['ms_per_decode_step_median']:.3f}), host {r['host_us_per_iteration']['total']:.0f} us/iteration "
                  f"(schedule {r['host_us_per_iteration']['schedule']:.0f}, forward {r['host_us_per_iteration']['forward']:.0f}, "
                  f"process {r['host_us_per_iteration']['process']:.0f}), cache {r['prefix_cache']}", flush=True)
    os.environ.pop("MINISGL_DISABLE_OVERLAP_SCHEDULING", None)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(dict(model=a.model, requests=a.n, sum_out=sum(outs), results=results), indent=1))


if __name__ == "__main__":
    main()
