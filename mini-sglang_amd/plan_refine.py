"""Re-ranking of the projection kernels' finalists INSIDE the captured decode step.

Back-to-back timing (gemm_plan.tune_projection_gemms) separates the best few kernels of a projection by a percent or
so.  Inside the captured decode step -- cold L2 / Infinity Cache after the attention stream, another clock and power
state, and above all the CONSUMER of the projection in the same measurement -- the same candidates differ by up to
9 %: the library's gate_up solutions 105 vs 115 us; o_proj through the library (37.9 us back to back, bf16 output) beats
the k-sliced full-batch kernel (30.9 us back to back, but 31 MB of fp32 slabs to write and for the next norm to read)
by 0.14 ms per step.  Which candidate the back-to-back search ranks first also varied from run to run.  So the finalists
are re-ranked WHERE THEY RUN: for each projection of the largest graph batch in turn (biggest first) each candidate is
made the plan, the graph is re-captured and a synthetic full batch replayed (median of 20 individually timed replays).
A challenger replaces the search's pick only if it is faster by `margin` (0.5 %) AND wins a confirmation round against a
fresh measurement of the incumbent by the same margin (round 3 ranked on the mean of 6 replays with a 0.2 % threshold:
inside run-to-run noise, plans flipped between runs).

Used by this repository's engine (engine.Engine.refine_plans_in_graph) and, through `minisgl_plugin.install()`, by the
reference's GraphRunner (P/engine/graph.py:105-150) -- the same function, handed the host's own Req / Batch types and
capture / replay callables.
"""
from __future__ import annotations

from typing import Any, Callable, List

import torch

from . import ops


class StepBench:
    """A synthetic full decode batch (bs requests of up to 2 * mean_context tokens in contiguous pool slots; K/V contents are
    left as they are: timing does not depend on them) replayed through the host's captured decode graph.  `ok` is False --
    and nothing was touched -- when the KV pool or the page table cannot hold it.  close() restores the page-table rows it
    borrowed."""

    def __init__(self, *, bs: int, page_table: torch.Tensor, page_size: int, num_pages: int, row_len: int,
                 device: torch.device, Req: Any, Batch: Any, prepare_metadata: Callable[[Any], None],
                 capture: Callable[[], None], replay: Callable[[Any], Any], forward_ctx: Callable[[Any], Any],
                 mean_context: int = 900) -> None:
        self.capture, self.replay, self.forward_ctx = capture, replay, forward_ctx
        self.page_table, self.bs = page_table, bs
        L = -(-(2 * mean_context) // page_size) * page_size  # slots per request row (page multiple)
        L = min(L, row_len // page_size * page_size, page_table.shape[1] // page_size * page_size)
        self.L = L
        self.ok = not (L < 64 or bs * L > num_pages * page_size or bs > page_table.shape[0])
        self.saved = None
        if not self.ok:
            return
        self.saved = page_table[:bs, :L].clone()
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
        self.batch = batch

    def measure(self, replays: int = 20) -> float:
        """Re-capture with the plans / kernel choices in force, then the MEDIAN (ms) of `replays` individually timed
        replays (one event pair each: a clock ramp or a stray host stall moves single samples, not the median)."""
        self.capture()
        batch = self.batch
        with self.forward_ctx(batch):
            self.replay(batch)
            self.replay(batch)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(replays + 1)]
            evs[0].record()
            for i in range(replays):
                self.replay(batch)
                evs[i + 1].record()
            evs[-1].synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        return ts[len(ts) // 2]

    def close(self) -> None:
        if self.saved is not None:
            self.page_table[: self.bs, : self.L] = self.saved
            self.saved = None


def refine_plans_in_graph(*, bs: int, page_table: torch.Tensor, page_size: int, num_pages: int, row_len: int,
                          device: torch.device, Req: Any, Batch: Any, prepare_metadata: Callable[[Any], None],
                          capture: Callable[[], None], replay: Callable[[Any], Any], forward_ctx: Callable[[Any], Any],
                          mean_context: int = 900, replays: int = 20, margin: float = 0.005) -> List[dict]:
    """Returns one dict per refined projection: step times (ms) per candidate, the one kept, whether it changed.
    Nothing happens (empty list) when no shape of batch `bs` has more than one candidate or the KV pool cannot hold the
    synthetic batch (StepBench).  The page-table rows it borrows are restored."""
    keys = [k for k in ops._CANDIDATES if k[1] == bs and len(ops._CANDIDATES[k]["cands"]) > 1]
    if not keys:
        return []
    sb = StepBench(bs=bs, page_table=page_table, page_size=page_size, num_pages=num_pages, row_len=row_len, device=device,
                   Req=Req, Batch=Batch, prepare_metadata=prepare_metadata, capture=capture, replay=replay,
                   forward_ctx=forward_ctx, mean_context=mean_context)
    if not sb.ok:
        return []

    def measure() -> float:
        return sb.measure(replays)

    def forget_pending() -> None:
        # a candidate that failed mid-forward may have left a deferred reduce / all-reduce registered: the next capture
        # must not trip over it
        dev = device.index or 0
        ops._PENDING_SLABS[dev] = None
        ops._PENDING_ALLREDUCE[dev] = None

    report = []
    try:
        base = measure()
        for key in sorted(keys, key=lambda k: -k[2] * k[3]):  # biggest projection first
            info = ops._CANDIDATES[key]
            snap, start = ops.snapshot_plan(key), ops.current_candidate(key)
            best_ms, best_spec, best_label, tried = base, None, None, {"(search's pick) " + start: round(base, 4)}
            for label, spec in info["cands"]:
                try:
                    ops.apply_candidate(key, spec)
                    ms = measure()
                except Exception as e:  # a candidate that cannot be captured is skipped, never fatal
                    forget_pending()
                    tried[label] = f"{type(e).__name__}: {e}"[:160]
                    continue
                tried[label] = round(ms, 4)
                if ms < best_ms:
                    best_ms, best_spec, best_label = ms, spec, label
            ops.restore_search_pick(key, snap)
            accepted = False
            if best_spec is not None and best_ms < base * (1.0 - margin):
                # confirmation round, incumbent first: box clocks drift over the seconds a sweep takes, so the challenger
                # must win again against a fresh measurement of the incumbent, by the same margin
                try:
                    base2 = measure()
                    ops.apply_candidate(key, best_spec)
                    ms2 = measure()
                    tried["confirm: incumbent / " + best_label] = [round(base2, 4), round(ms2, 4)]
                    accepted = ms2 < base2 * (1.0 - margin)
                    if accepted:
                        base = ms2
                    else:
                        ops.restore_search_pick(key, snap)
                        base = base2
                except Exception as e:
                    forget_pending()
                    ops.restore_search_pick(key, snap)
                    tried["confirm"] = f"{type(e).__name__}: {e}"[:160]
            report.append(dict(name=info["name"], M=key[1], N=key[2], K=key[3], step_ms=tried, replays=replays, margin=margin,
                               chosen=ops.current_candidate(key), changed=accepted))
        sb.capture()  # the graph that stays = the plans that stay
    finally:
        sb.close()
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
