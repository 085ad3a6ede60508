"""Re-ranking of the projection kernels' finalists INSIDE the captured decode step.

Back-to-back timing (gemm_plan.tune_projection_gemms) separates the best few kernels of a projection by a percent or
so.  Inside the captured decode step -- cold L2 / Infinity Cache after the attention stream, another clock and power
state, and above all the CONSUMER of the projection in the same measurement -- the same candidates differ by up to
9 %: the library's gate_up solutions 105 vs 115 us; o_proj through the library (37.9 us back to back, bf16 output) beats
the k-sliced full-batch kernel (30.9 us back to back, but 31 MB of fp32 slabs to write and for the next norm to read)
by 0.14 ms per step.  Which candidate the back-to-back search ranks first also varied from run to run.  So the finalists
are re-ranked WHERE THEY RUN: for each projection of the largest graph batch in turn (biggest first) each candidate is
made the plan, the graph is re-captured and a synthetic full batch replayed; the fastest stays.

Used by this repository's engine (engine.Engine.refine_plans_in_graph) and, through `minisgl_plugin.install()`, by the
reference's GraphRunner (P/engine/graph.py:105-150) -- the same function, handed the host's own Req / Batch types and
capture / replay callables.
"""
from __future__ import annotations

from typing import Any, Callable, List

import torch

from . import ops


def refine_plans_in_graph(*, bs: int, page_table: torch.Tensor, page_size: int, num_pages: int, row_len: int,
                          device: torch.device, Req: Any, Batch: Any, prepare_metadata: Callable[[Any], None],
                          capture: Callable[[], None], replay: Callable[[Any], Any], forward_ctx: Callable[[Any], Any],
                          mean_context: int = 900, replays: int = 6) -> List[dict]:
    """Returns one dict per refined projection: step times (ms) per candidate, the one kept, whether it changed.
    Nothing happens (empty list) when no shape of batch `bs` has more than one candidate or the KV pool cannot hold the
    synthetic batch (bs requests of up to 2 * mean_context tokens in contiguous slots; K/V contents are left as they
    are: timing does not depend on them).  The page-table rows it borrows are restored."""
    keys = [k for k in ops._CANDIDATES if k[1] == bs and len(ops._CANDIDATES[k]["cands"]) > 1]
    L = -(-(2 * mean_context) // page_size) * page_size  # slots per request row (page multiple)
    L = min(L, row_len // page_size * page_size, page_table.shape[1] // page_size * page_size)
    if not keys or L < 64 or bs * L > num_pages * page_size or bs > page_table.shape[0]:
        return []
    saved = page_table[:bs, :L].clone()
    page_table[:bs, :L] = (torch.arange(bs, device=device, dtype=torch.int32)[:, None] * L
                           + torch.arange(L, device=device, dtype=torch.int32)[None, :])
    lens = [max(16, min(L - 1, int(mean_context * (0.25 + 1.5 * ((i * 37) % 101) / 100.0)))) for i in range(bs)]
    reqs = [Req(input_ids=torch.zeros(n + 1, dtype=torch.int32), table_idx=i, cached_len=n, output_len=1 << 20, uid=-2 - i,
                **_req_extras(Req)) for i, n in enumerate(lens)]
    batch = Batch(reqs=reqs, phase="decode")
    batch.padded_reqs = reqs
    batch.input_ids = torch.zeros(bs, dtype=torch.int32, device=device)
    batch.positions = torch.tensor(lens, dtype=torch.int32, device=device)
    batch.out_loc = page_table[torch.arange(bs, device=device), batch.positions.long()].contiguous()
    prepare_metadata(batch)

    def measure() -> float:
        capture()
        with forward_ctx(batch):
            replay(batch)
            replay(batch)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(replays):
                replay(batch)
            e1.record()
            e1.synchronize()
        return e0.elapsed_time(e1) / replays

    report = []
    try:
        base = measure()
        for key in sorted(keys, key=lambda k: -k[2] * k[3]):  # biggest projection first
            info = ops._CANDIDATES[key]
            snap, start = ops.snapshot_plan(key), ops.current_candidate(key)
            best_ms, best_spec, tried = base, None, {"(search's pick) " + start: round(base, 4)}
            for label, spec in info["cands"]:
                try:
                    ops.apply_candidate(key, spec)
                    ms = measure()
                except Exception as e:  # a candidate that cannot be captured is skipped, never fatal
                    tried[label] = type(e).__name__
                    continue
                tried[label] = round(ms, 4)
                if ms < best_ms * 0.998:
                    best_ms, best_spec = ms, spec
            ops.restore_search_pick(key, snap)
            if best_spec is not None:
                ops.apply_candidate(key, best_spec)
            base = best_ms
            report.append(dict(name=info["name"], M=key[1], N=key[2], K=key[3], step_ms=tried,
                               chosen=ops.current_candidate(key), changed=best_spec is not None))
        capture()  # the graph that stays = the plans that stay
    finally:
        page_table[:bs, :L] = saved
    return report


def _req_extras(Req: Any) -> dict:
    """Required constructor arguments beyond the five set above: the reference's Req is a dataclass that also wants
    `sampling_params` and `cache_handle` (P/core.py:29-36); this repository's has defaults for them."""
    import inspect

    extras = {}
    for name, p in inspect.signature(Req).parameters.items():
        if name in ("self", "input_ids", "table_idx", "cached_len", "output_len", "uid") or p.default is not inspect.Parameter.empty:
            continue
        if name == "sampling_params":
            try:
                from minisgl.core import SamplingParams
            except ImportError:
                from .core import SamplingParams
            extras[name] = SamplingParams()
        else:
            extras[name] = None
    return extras
