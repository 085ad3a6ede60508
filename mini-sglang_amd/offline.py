"""Offline driver: feeds a fixed request set through the engine (chunked prefill, then
continuous decode with graph padding), producing the (Batch, positions, out_loc, page table)
inputs the hot path consumes.

It plays the part of the reference's scheduler for benchmarks on the GPU box, where the
reference itself is absent.  Only the tensor glue of P/scheduler/scheduler.py:204-267,
P/scheduler/cache.py:42-53,127-146 and P/scheduler/table.py:4-12 is restated (page-aligned slot
allocation, `out_loc = page_table[(table_idx, position)]`, device-side token feedback through
`token_pool`); admission policy, radix reuse and I/O are out of scope.
Host work per step is vectorised (numpy), and nothing synchronises with the device: step N+1 is
prepared while step N runs, like the reference's overlap loop.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .core import Batch, Req, SamplingParams
from .engine import Engine


@dataclass
class RequestState:
    req: Req
    prompt_len: int
    max_tokens: int
    t_submit: float = 0.0
    t_first: Optional[float] = None
    output_events: List = field(default_factory=list)


class PageAllocator:
    """Host side of the reference's page allocation (P/scheduler/cache.py:15-20, 42-53, 106-146): `free_slots` holds
    page-START token slots, a batch takes the first `needed` of them in request order, every page expands to
    page_size consecutive token slots, and those land at positions [first_page, last_page) x page_size of each
    request's table row.  Pure numpy, no device: the caller scatters (rows, positions, slots) into the page table.
    Pinned against the reference's own CacheManager trace in tests/test_cpu_host_logic.py."""

    def __init__(self, num_pages: int, page_size: int, order: Optional[np.ndarray] = None) -> None:
        self.page_size = page_size
        pages = np.arange(num_pages, dtype=np.int64) if order is None else np.asarray(order, dtype=np.int64)
        self.free_slots = (pages * page_size).astype(np.int32)
        self.row_pages: Dict[int, List[np.ndarray]] = {}  # which pages each table row received (to give back)

    def allocate(self, reqs: Sequence) -> Optional[tuple]:
        """Pages for positions [cached_len, device_len) of every request -> (rows, positions, token slots) or None."""
        ps = self.page_size
        first = np.array([-(-r.cached_len // ps) for r in reqs], dtype=np.int64)
        last = np.array([-(-r.device_len // ps) for r in reqs], dtype=np.int64)
        need = np.maximum(last - first, 0)
        total = int(need.sum())
        if total == 0:
            return None
        if total > len(self.free_slots):
            raise RuntimeError("KV pool exhausted")
        pages, self.free_slots = self.free_slots[:total], self.free_slots[total:]
        off = 0
        for r, n in zip(reqs, need.tolist()):
            if n:
                self.row_pages.setdefault(r.table_idx, []).append(pages[off: off + n])
                off += n
        rows = np.repeat(np.array([r.table_idx for r in reqs], dtype=np.int64), need * ps)
        starts = np.repeat(first * ps, need * ps)
        seg = np.repeat(np.cumsum(need * ps) - need * ps, need * ps)
        pos = starts + (np.arange(total * ps, dtype=np.int64) - seg)
        tok = (pages[:, None] + np.arange(ps, dtype=np.int32)[None, :]).reshape(-1)
        return rows, pos, tok

    def free_row(self, table_idx: int) -> None:
        """Give a finished request's pages back, appended in the order they were taken (cache.py:115-119)."""
        owned = self.row_pages.pop(table_idx, [])
        if owned:
            self.free_slots = np.concatenate([self.free_slots] + owned)


class OfflineRunner:
    def __init__(self, engine: Engine, max_extend_tokens: int = 8192, seed: Optional[int] = 0) -> None:
        self.engine = engine
        self.device = engine.device
        self.page_size = engine.cfg.page_size
        self.max_extend_tokens = max_extend_tokens
        self.page_table = engine.page_table
        self.token_pool = torch.zeros_like(self.page_table)  # P/scheduler/table.py:9-10
        # free pages shuffled (seed) unless seed is None (the reference's initial order, cache.py:19): a long-running
        # server's free list is not sorted, and the kernels must not depend on it being so
        order = None if seed is None else np.random.default_rng(seed).permutation(engine.num_pages)
        self.pages = PageAllocator(engine.num_pages, self.page_size, order)
        self.free_rows = list(range(engine.cfg.max_running_req))  # pop() hands out the highest row first (table.py:6,17)

    # ------------------------------------------------------------------ allocation
    def _allocate_paged(self, reqs: Sequence[Req]) -> None:
        """Pages for positions [cached_len, device_len) of every request (cache.py:42-53) -> page table."""
        got = self.pages.allocate(reqs)
        if got is None:
            return
        rows, pos, tok = got
        rows_t = torch.from_numpy(rows).pin_memory().to(self.device, non_blocking=True)
        pos_t = torch.from_numpy(pos).pin_memory().to(self.device, non_blocking=True)
        tok_t = torch.from_numpy(tok).pin_memory().to(self.device, non_blocking=True)
        self.page_table[rows_t, pos_t] = tok_t

    def _free(self, req: Req) -> None:
        self.pages.free_row(req.table_idx)
        self.free_rows.append(req.table_idx)

    # ------------------------------------------------------------------ batches
    def _make_batch(self, reqs: List[Req], phase: str) -> tuple:
        batch = Batch(reqs=reqs, phase=phase)  # type: ignore[arg-type]
        self.engine.graph_runner.pad_batch(batch)
        self._allocate_paged(reqs)
        padded = batch.padded_reqs
        ext = np.array([r.extend_len for r in padded], dtype=np.int64)
        cached = np.array([r.cached_len for r in padded], dtype=np.int64)
        rows = np.array([r.table_idx for r in padded], dtype=np.int64)
        total = int(ext.sum())
        seg = np.repeat(np.cumsum(ext) - ext, ext)
        positions = (np.repeat(cached, ext) + (np.arange(total, dtype=np.int64) - seg))
        row_rep = np.repeat(rows, ext)
        pos_t = torch.from_numpy(positions.astype(np.int32)).pin_memory().to(self.device, non_blocking=True)
        row_t = torch.from_numpy(row_rep).pin_memory().to(self.device, non_blocking=True)
        batch.positions = pos_t
        pos64 = pos_t.to(torch.int64)
        batch.out_loc = self.page_table[row_t, pos64]  # scheduler.py:210
        batch.input_ids = self.token_pool[row_t, pos64]  # scheduler.py:229
        # where the sampled token goes (scheduler.py:262-267): column device_len, or -1 (junk)
        n = len(reqs)
        wrow = torch.from_numpy(rows[:n]).pin_memory().to(self.device, non_blocking=True)
        wcol = torch.from_numpy(np.array([(r.device_len if r.can_decode else -1) for r in reqs],
                                         dtype=np.int64)).pin_memory().to(self.device, non_blocking=True)
        self.engine.attn_backend.prepare_metadata(batch)
        return batch, (wrow, wcol)

    def _forward(self, batch: Batch, write, sample_args):
        out = self.engine.forward_batch(batch, sample_args)
        self.token_pool[write] = out.next_tokens_gpu
        return out

    # ------------------------------------------------------------------ public API
    def add_request(self, prompt_ids: Sequence[int], params: SamplingParams) -> RequestState:
        row = self.free_rows.pop()
        ids = torch.tensor(list(prompt_ids), dtype=torch.int32)
        out_len = min(params.max_tokens, self.engine.max_seq_len - len(ids))
        req = Req(input_ids=ids, table_idx=row, cached_len=0, output_len=out_len, uid=row, sampling_params=params)
        self.token_pool[row, : len(ids)] = ids.pin_memory().to(self.device, non_blocking=True)
        return RequestState(req=req, prompt_len=len(ids), max_tokens=out_len, t_submit=time.perf_counter())

    def warmup_prefill(self, tokens: Optional[int] = None) -> None:
        """One untimed prefill forward of `tokens` (default: the chunk budget) new tokens of dummy requests.
        They all sit on the engine's dummy table row, whose every entry is the dummy page (P/engine/engine.py:
        89-98), so K/V land there and no request state is touched.  Purpose: first-use costs of the chunk
        shape (library GEMM kernel load + heuristic, allocator growth) stay out of TTFT, like the one-prompt
        warm-up generate() of the reference's benchmark (benchmark/offline/bench.py:32)."""
        eng = self.engine
        tokens = tokens or self.max_extend_tokens
        per = max(1, min(eng.max_seq_len - 1, tokens))
        lens = [per] * (tokens // per) + ([tokens % per] if tokens % per else [])
        row = eng.dummy_req.table_idx
        reqs = [Req(input_ids=torch.zeros(n, dtype=torch.int32), table_idx=row, cached_len=0, output_len=1, uid=-1,
                    sampling_params=SamplingParams()) for n in lens]
        batch = Batch(reqs=reqs, phase="prefill")  # type: ignore[arg-type]
        batch.padded_reqs = reqs
        pos = np.concatenate([np.arange(n, dtype=np.int32) for n in lens])
        batch.positions = torch.from_numpy(pos).pin_memory().to(self.device, non_blocking=True)
        batch.out_loc = torch.full((len(pos),), eng.num_pages * self.page_size, dtype=torch.int32, device=self.device)
        batch.input_ids = torch.zeros(len(pos), dtype=torch.int32, device=self.device)
        eng.attn_backend.prepare_metadata(batch)
        with eng.ctx.forward_batch(batch):
            logits = eng.model.forward(eng.ctx, batch)
        eng.sampler.sample(logits, eng.sampler.prepare(batch))
        del logits

    def prefill(self, states: List[RequestState]):
        """Chunked prefill of all requests under the token budget (P/scheduler/prefill.py:65-90:
        a request cut by the budget continues in the next forward with cached_len advanced).
        A generator: yields (forward output, requests that got their first token) right after each
        chunk is enqueued, so a caller can drop a device event between chunks (per-request TTFT)."""
        pending = list(states)
        while pending:
            budget = self.max_extend_tokens
            reqs: List[Req] = []
            finals: List[RequestState] = []
            while pending and budget > 0:
                st = pending[0]
                r = st.req
                remain = st.prompt_len - r.cached_len
                take = min(remain, budget)
                budget -= take
                r.device_len = r.cached_len + take  # this forward covers [cached_len, device_len)
                reqs.append(r)
                if take == remain:
                    finals.append(st)
                    pending.pop(0)
                else:
                    break
            batch, write = self._make_batch(reqs, "prefill")
            # chunked requests' sampled token is junk: route it to column -1
            if len(finals) != len(reqs):
                write[1][-1] = -1
            args = self.engine.sampler.prepare(batch)
            out = self._forward(batch, write, args)  # complete_one: cached_len = device_len, device_len += 1
            if len(finals) != len(reqs):  # undo the +1 of the chunked request: it has not produced a token
                reqs[-1].device_len -= 1
            yield out, finals

    def decode_step(self, running: List[RequestState]):
        reqs = [s.req for s in running]
        batch, write = self._make_batch(reqs, "decode")
        args = self.engine.sampler.prepare(batch)
        return self._forward(batch, write, args)

    def generate(self, prompts: Sequence[Sequence[int]], params: Sequence[SamplingParams]) -> Dict:
        """Run every request to max_tokens (ignore_eos semantics of benchmark/offline/bench.py).
        Returns wall time, per-request TTFT (all requests are submitted at t = 0; a request's first
        token exists when the prefill forward holding its last chunk finishes) and the decode-phase
        time, all measured with device events -- the host never waits inside the loop."""
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        states = [self.add_request(p, sp) for p, sp in zip(prompts, params)]
        first_token_events = []
        for out, finals in self.prefill(states):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            first_token_events.append((ev, len(finals)))
        ev_prefill_done = torch.cuda.Event(enable_timing=True)
        ev_prefill_done.record()
        running = [s for s in states if s.req.can_decode]
        for s in states:
            if not s.req.can_decode:
                self._free(s.req)
        n_decode_steps, decode_tokens = 0, 0
        while running:
            self.decode_step(running)
            n_decode_steps += 1
            decode_tokens += len(running)
            still = []
            for s in running:
                if s.req.can_decode:
                    still.append(s)
                else:
                    self._free(s.req)
            running = still
        ev_end = torch.cuda.Event(enable_timing=True)
        ev_end.record()
        torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        ttft_ms: List[float] = []
        for ev, n in first_token_events:
            ttft_ms += [ev0.elapsed_time(ev)] * n
        self.last_states = states
        return dict(wall_s=t1 - t0, decode_steps=n_decode_steps, decode_tokens=decode_tokens,
                    prefill_ms=ev0.elapsed_time(ev_prefill_done), decode_ms=ev_prefill_done.elapsed_time(ev_end),
                    ttft_ms=ttft_ms, total_output_tokens=decode_tokens + len(states))

    def output_ids(self, st: RequestState) -> List[int]:
        """Generated token ids of a (finished or running) request, read back from the token pool."""
        n_out = st.req.device_len - st.prompt_len
        return self.token_pool[st.req.table_idx, st.prompt_len: st.prompt_len + n_out].cpu().tolist()
