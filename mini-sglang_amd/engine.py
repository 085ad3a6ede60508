"""Per-rank engine: model + KV pool + page table + attention backend + sampler + hipGraph runner.

Host-side mirror of the reference's L4 (P/engine/engine.py:29-211, graph.py:21-171,
sample.py:24-75) restricted to what drives the hot path.  The reference's own Engine runs
unchanged on the plugin when `minisgl` is installed; this one exists for the
GPU box, where only this repository is present.
"""
from __future__ import annotations

import os

import gc
from dataclasses import dataclass, field
from typing import Any, Dict, List, NamedTuple, Optional

import numpy as np
import torch

from . import flashinfer_compat as fi
from . import ops
from .attention import HipAttnBackend
from .core import Batch, Context, Req, SamplingParams, set_global_ctx
from .kvcache import create_kvcache_pool, heads_per_rank
from .model import Communicator, DenseDecoder, ModelConfig


# ------------------------------------------------------------------------------ sampler
_T_FLOOR = 1e-6  # the reference sampler's lower clamp on temperature and on top_p (P/engine/sample.py:58)


class SamplingPlan(NamedTuple):
    """What one batch's token draw needs on the device.  `temperatures is None` = every row is an argmax.  The field
    names are the ones P/engine/sample.py:13-17 gives its argument record (tests and the plugin read them)."""
    temperatures: Optional[torch.Tensor]
    top_k: Optional[torch.Tensor] = None
    top_p: Optional[torch.Tensor] = None


def plan_sampling_host(temperature: "np.ndarray", top_k: "np.ndarray", top_p: "np.ndarray", vocab_size: int):
    """The clamps of P/engine/sample.py:53-68 over whole columns (numpy; no per-request Python arithmetic).

    Input: one entry per request of SamplingParams.{temperature, top_k, top_p}.  Output: None when the whole batch is
    greedy, else an (f32 temperatures, i32 top_k | None, f32 top_p | None) triple.  A greedy row inside a sampled batch gets
    the floor temperature (its softmax is then a one-hot up to ties); top_k < 1 means 'no limit' = the vocabulary size;
    top_p is clamped into [floor, 1].  A filter column that filters nothing for any row is dropped, so the cheaper
    draw kernel is chosen."""
    greedy = ((temperature <= 0.0) | (top_k == 1)) & (top_p == 1.0)  # SamplingParams.is_greedy, P/core.py:23-25
    if bool(greedy.all()):
        return None
    t = np.maximum(np.where(greedy, 0.0, temperature), _T_FLOOR).astype(np.float32)
    k = np.where(top_k >= 1, top_k, vocab_size).astype(np.int32)
    p = np.clip(top_p, _T_FLOOR, 1.0).astype(np.float32)
    return t, (k if bool((k != vocab_size).any()) else None), (p if bool((top_p < 1.0).any()) else None)


class Sampler:
    """Batch sampler behind the interface of P/engine/sample.py:49-75 (`prepare` on the host, `sample` enqueues).

    prepare: the requests' three sampling scalars are gathered into columns once, clamped by `plan_sampling_host`, packed
    into ONE pinned staging row and sent with ONE asynchronous copy; the device tensors are views of that upload
    (top_k reinterpreted as int32)."""

    def __init__(self, device: torch.device, vocab_size: int) -> None:
        self.device, self.vocab_size = device, vocab_size
        self._pin = device.type == "cuda"

    def prepare(self, batch: Batch) -> SamplingPlan:
        n = len(batch.reqs)
        cols = np.fromiter((v for r in batch.reqs for v in
                            (r.sampling_params.temperature, r.sampling_params.top_k, r.sampling_params.top_p)),
                           dtype=np.float64, count=3 * n).reshape(n, 3)
        plan = plan_sampling_host(cols[:, 0], cols[:, 1].astype(np.int64), cols[:, 2], self.vocab_size)
        if plan is None:
            return SamplingPlan(None)
        live = [c for c in plan if c is not None]
        host = torch.empty((len(live), n), dtype=torch.float32, pin_memory=self._pin)
        for row, c in zip(host.numpy(), live):
            row.view(c.dtype)[:] = c  # int32 top_k travels as raw bits in the f32 staging row
        dev = iter(host.to(self.device, non_blocking=True).unbind(0))
        t, k, p = plan
        return SamplingPlan(next(dev), None if k is None else next(dev).view(torch.int32),
                            None if p is None else next(dev))

    def sample(self, logits: torch.Tensor, args: SamplingPlan) -> torch.Tensor:
        if args.temperatures is None:  # greedy: first index of the row max
            return ops.argmax_rows(logits)
        # same op sequence as sample_impl (sample.py:24-45); `softmax` is deferred, so a temperature-only batch
        # is one fused draw from the logits and only the top-k / top-p variants materialise probabilities
        probs = fi.sampling.softmax(logits, args.temperatures)
        draw = {
            (False, False): lambda: fi.sampling.sampling_from_probs(probs),
            (True, False): lambda: fi.sampling.top_k_sampling_from_probs(probs, args.top_k),
            (False, True): lambda: fi.sampling.top_p_sampling_from_probs(probs, args.top_p),
            (True, True): lambda: fi.sampling.top_k_top_p_sampling_from_probs(probs, args.top_k, args.top_p),
        }
        return draw[(args.top_k is not None, args.top_p is not None)]()


BatchSamplingArgs = SamplingPlan  # the reference's name for the record


# ------------------------------------------------------------------------------ config
@dataclass
class EngineConfig:
    """Fields of P/engine/config.py:16-55 that reach the hot path."""
    model: ModelConfig
    dtype: torch.dtype = torch.bfloat16
    tp_rank: int = 0
    tp_size: int = 1
    max_running_req: int = 256
    cuda_graph_bs: Optional[List[int]] = None
    cuda_graph_max_bs: Optional[int] = None
    page_size: int = 1
    memory_ratio: float = 0.9
    max_seq_len_override: Optional[int] = None
    num_page_override: Optional[int] = None
    fused_qkv_path: bool = True
    comm: Any = None  # communicator with all_reduce / all_gather (tp_size > 1): kernel.init_pynccl(...)
    comm_side: Any = None  # second communicator for side-stream collectives (see model.DenseDecoder.row_parallel)
    comm_split_tokens: int = 0  # token-split + side-stream all-reduce for forwards of at least this many tokens
    comm_overlap: bool = True
    tp_cpu_group: Any = None  # torch.distributed CPU group of the TP ranks (pool sizing agreement)
    gemm_tune: str = "heuristic"  # "off" | "heuristic" | "full": library solution search per graph batch size
    refine_in_graph: bool = True  # re-rank the search's finalists at the largest graph batch inside the captured step
    prefill_tokens: Optional[int] = None  # the scheduler's max_extend_tokens: library solution search at that chunk size too
    seed: int = 42

    @property
    def max_seq_len(self) -> int:
        return self.max_seq_len_override or self.model.max_position


class ForwardOutput(NamedTuple):
    next_tokens_gpu: torch.Tensor
    next_tokens_cpu: torch.Tensor
    copy_done_event: torch.cuda.Event


def _align_up_32(n: int) -> int:
    return (n + 31) // 32 * 32


def determine_num_pages(free_before: int, free_after: int, cfg: EngineConfig) -> int:
    """P/engine/engine.py:148-168: (memory_ratio * free_before - model_bytes) // bytes_per_page."""
    m = cfg.model
    cache_per_page = (2 * m.head_dim * heads_per_rank(m.num_kv_heads, cfg.tp_size, replicate=True) * cfg.page_size
                      * torch.empty((), dtype=cfg.dtype).element_size() * m.num_layers)
    if cfg.num_page_override is not None:
        return cfg.num_page_override
    model_memory = free_before - free_after
    available = int(cfg.memory_ratio * free_before) - model_memory
    num_pages = available // cache_per_page
    assert num_pages > 1, "Not enough memory for KV cache"
    return num_pages


def determine_graph_bs(cuda_graph_bs: Optional[List[int]], cuda_graph_max_bs: Optional[int],
                       free_memory: int) -> List[int]:
    """P/engine/graph.py:49-67."""
    if cuda_graph_bs is not None:
        return cuda_graph_bs
    if cuda_graph_max_bs is None:
        cuda_graph_max_bs = 256 if free_memory / (1 << 30) > 80 else 160
    if cuda_graph_max_bs < 1:
        return []
    return [b for b in [1, 2, 4] if b <= cuda_graph_max_bs] + list(range(8, cuda_graph_max_bs + 1, 8))


# ------------------------------------------------------------------------------ graph runner
class GraphRunner:
    """hipGraph capture/replay of the decode forward, P/engine/graph.py:78-171."""

    def __init__(self, engine: "Engine", bs_list: List[int]) -> None:
        self.engine = engine
        self.graph_bs_list = sorted(bs_list)
        self.max_graph_bs = max(bs_list) if bs_list else 0
        self.graph_map: Dict[int, torch.cuda.CUDAGraph] = {}
        if not bs_list:
            return
        dev, V = engine.device, engine.cfg.model.vocab_size
        self.input_ids = torch.zeros(self.max_graph_bs, dtype=torch.int32, device=dev)
        self.out_loc = torch.zeros(self.max_graph_bs, dtype=torch.int32, device=dev)
        self.positions = torch.zeros(self.max_graph_bs, dtype=torch.int32, device=dev)
        # The reference copies the model's logits into a static fp32 buffer inside the graph (P/engine/graph.py:33,
        # 139-141).  Here the LM-head projection writes straight into the static buffer, which stays in the model dtype
        # (the sampling kernels read it directly): no 78-MB conversion pass per step, half the bytes for the sampler;
        # the VALUES are the same.
        self.logits = torch.empty((self.max_graph_bs, V), dtype=engine.dtype, device=dev)
        backend = engine.attn_backend
        backend.init_capture_graph(max_seq_len=engine.aligned_max_seq_len, bs_list=self.graph_bs_list)
        torch.cuda.synchronize(dev)
        self._pool = None
        for bs in sorted(self.graph_bs_list, reverse=True):
            self.capture(bs)

    def capture(self, bs: int) -> None:
        """(Re)capture the decode forward at batch size `bs` with the kernel plans in force now."""
        engine, backend = self.engine, self.engine.attn_backend
        graph = torch.cuda.CUDAGraph()
        batch = Batch(reqs=[engine.dummy_req] * bs, phase="decode")
        batch.padded_reqs = batch.reqs
        backend.prepare_for_capture(batch)
        batch.input_ids, batch.out_loc, batch.positions = self.input_ids[:bs], self.out_loc[:bs], self.positions[:bs]
        static = self.logits[:bs]

        def forward_into_static() -> None:
            # the model writes the LM head straight into the static buffer where it can (tp = 1); whatever else it
            # returns is copied there INSIDE the graph, as the reference does (P/engine/graph.py:139-141) -- a replay
            # must never hand back a buffer the captured kernels did not write
            out = engine.model.forward(engine.ctx, batch, logits_out=static)
            if out.data_ptr() != static.data_ptr() or out.shape != static.shape or out.stride() != static.stride():
                static.copy_(out)

        with engine.ctx.forward_batch(batch):
            forward_into_static()
            # with a communicator inside the graph, RCCL's proxy thread may touch the HIP API while this
            # thread captures: only this thread's calls are held to the capture rules then
            mode = "thread_local" if engine.cfg.tp_size > 1 else "global"
            with torch.cuda.graph(graph, pool=self._pool, stream=engine.stream, capture_error_mode=mode):
                forward_into_static()
        if self._pool is None:
            self._pool = graph.pool()
        self.graph_map[bs] = graph

    def can_use_cuda_graph(self, batch: Batch) -> bool:
        return batch.is_decode and batch.size <= self.max_graph_bs

    def pad_batch(self, batch: Batch) -> None:
        padded = (next(bs for bs in self.graph_bs_list if bs >= batch.size)
                  if self.can_use_cuda_graph(batch) else batch.size)
        batch.padded_reqs = batch.reqs + [self.engine.dummy_req] * (padded - batch.size)

    def replay(self, batch: Batch) -> torch.Tensor:
        n = batch.padded_size
        self.input_ids[:n] = batch.input_ids
        self.out_loc[:n] = batch.out_loc
        self.positions[:n] = batch.positions
        self.engine.attn_backend.prepare_for_replay(batch)
        self.graph_map[n].replay()
        return self.logits[: batch.size]

    def destroy(self) -> None:
        self.graph_map = {}
        gc.collect()


# ------------------------------------------------------------------------------ engine
class Engine:
    def __init__(self, cfg: EngineConfig, device: Optional[torch.device] = None) -> None:
        self.cfg = cfg
        self.device = device or torch.device(f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        torch.manual_seed(cfg.seed)  # every TP rank seeds identically (P/engine/engine.py:37)
        self.stream = torch.cuda.Stream()
        torch.cuda.set_stream(self.stream)
        self.dtype = cfg.dtype
        self.ctx = Context(cfg.page_size)
        set_global_ctx(self.ctx, force=True)

        torch.cuda.synchronize(self.device)
        free_before = torch.cuda.mem_get_info(self.device)[0]
        free_before = self._agreed_free_memory(free_before)
        comm = Communicator(cfg.comm, cfg.tp_size, side=cfg.comm_side)
        # two collectives of ONE communicator must never be in flight at once (shared flags / sequence counters; RCCL is
        # not safe across two streams either): without a second communicator the token halves run serially
        overlap = cfg.comm_overlap and not (cfg.tp_size > 1 and cfg.comm_split_tokens > 0 and cfg.comm_side is None)
        self.model = DenseDecoder(cfg.model, dtype=cfg.dtype, device=self.device, tp_rank=cfg.tp_rank,
                                  tp_size=cfg.tp_size, seed=cfg.seed, comm=comm, fused=cfg.fused_qkv_path,
                                  comm_split_tokens=cfg.comm_split_tokens, comm_overlap=overlap)
        torch.cuda.synchronize(self.device)
        free_after = self._agreed_free_memory(torch.cuda.mem_get_info(self.device)[0])

        self.num_pages = determine_num_pages(free_before, free_after, cfg)
        num_tokens = self.num_pages * cfg.page_size
        self.ctx.kv_cache = self.kv_cache = create_kvcache_pool(
            cfg.model, self.num_pages + 1, cfg.page_size, cfg.dtype, self.device, tp_size=cfg.tp_size)  # +1 dummy
        self.max_seq_len = min(cfg.max_seq_len, num_tokens)
        self.aligned_max_seq_len = _align_up_32(self.max_seq_len)
        self.ctx.page_table = self.page_table = torch.zeros(
            (cfg.max_running_req + 1, self.aligned_max_seq_len), dtype=torch.int32, device=self.device)
        self.ctx.attn_backend = self.attn_backend = HipAttnBackend(cfg.model, self.ctx, tp_size=cfg.tp_size)
        self.sampler = Sampler(self.device, cfg.model.vocab_size)
        self.dummy_req = Req(input_ids=torch.tensor([0], dtype=torch.int32), table_idx=cfg.max_running_req,
                             cached_len=0, output_len=1, uid=-1)
        self.page_table[self.dummy_req.table_idx].fill_(num_tokens)  # the dummy page
        bs_list = determine_graph_bs(cfg.cuda_graph_bs, cfg.cuda_graph_max_bs, free_before)
        bs_list = [b for b in bs_list if b <= cfg.max_running_req]
        # solution search happens before capture (it synchronises); full search only where it pays
        self.gemm_report = self.model.tune_gemms(bs_list, cfg.gemm_tune,
                                                 prefill_tokens=[cfg.prefill_tokens] if cfg.prefill_tokens else ())
        self.graph_runner = GraphRunner(self, bs_list)
        self.refine_report: List[dict] = []
        if cfg.refine_in_graph and cfg.gemm_tune != "off" and bs_list and cfg.tp_size == 1 and \
                os.environ.get("MSGL_DISABLE_REFINE") != "1":
            self.refine_report = self.refine_plans_in_graph(max(bs_list))
        if cfg.tp_size > 1 and cfg.tp_cpu_group is not None:
            # kernel search and capture take a rank-dependent time; the device-side barriers of the peer-to-peer
            # collectives spin for a bounded time only, so the ranks meet on the CPU before the first forward
            import torch.distributed as dist

            torch.cuda.synchronize(self.device)
            dist.barrier(group=cfg.tp_cpu_group)

    def refine_plans_in_graph(self, bs: int) -> List[dict]:
        """Re-rank the search's finalists inside the captured decode step (plan_refine.refine_plans_in_graph); tp = 1
        only: every replay runs the collectives, so the ranks would have to agree on the number of replays."""
        from .plan_refine import refine_plans_in_graph

        gr = self.graph_runner
        if bs not in gr.graph_map:
            return []
        return refine_plans_in_graph(
            bs=bs, page_table=self.page_table, page_size=self.cfg.page_size, num_pages=self.num_pages,
            row_len=self.aligned_max_seq_len, device=self.device, Req=Req, Batch=Batch,
            prepare_metadata=self.attn_backend.prepare_metadata, capture=lambda: gr.capture(bs), replay=gr.replay,
            forward_ctx=self.ctx.forward_batch)

    def _agreed_free_memory(self, free: int) -> int:
        """Every TP rank must derive the same num_pages / max_seq_len / page-table width (the schedulers are
        replicated): take the MIN of the ranks' free memory over the CPU group and refuse an imbalance above 2 GiB,
        as _sync_get_memory does (P/engine/engine.py:170-189)."""
        cfg = self.cfg
        if cfg.tp_size == 1 or cfg.tp_cpu_group is None:
            return free
        import torch.distributed as dist

        t = torch.tensor([free, -free], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=cfg.tp_cpu_group)
        lo, hi = int(t[0]), -int(t[1])
        if hi - lo > 2 * 1024 ** 3:
            raise RuntimeError(f"Memory across TP ranks are imbalanced: min {lo / 2**30:.2f} GiB, max {hi / 2**30:.2f} GiB")
        return lo

    def forward_batch(self, batch: Batch, args: BatchSamplingArgs) -> ForwardOutput:
        """P/engine/engine.py:191-206."""
        with self.ctx.forward_batch(batch):
            if self.graph_runner.can_use_cuda_graph(batch):
                logits = self.graph_runner.replay(batch)
            else:
                logits = self.model.forward(self.ctx, batch)
        for req in batch.reqs:
            req.complete_one()
        next_tokens_gpu = self.sampler.sample(logits[: batch.size], args).to(torch.int32)
        next_tokens_cpu = next_tokens_gpu.to("cpu", non_blocking=True)
        if self.cfg.tp_size > 1:  # a peer-to-peer barrier that gave up poisons its output: make that an exception
            self._forwards = getattr(self, "_forwards", 0) + 1
            if self._forwards % 8 == 0:
                self._poll_comm_errors(sync=False)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ForwardOutput(next_tokens_gpu, next_tokens_cpu, ev)

    def _poll_comm_errors(self, sync: bool) -> None:
        for c in (self.cfg.comm, self.cfg.comm_side):
            if c is not None and hasattr(c, "poll_error"):
                c.poll_error(sync)

    def shutdown(self) -> None:
        self.graph_runner.destroy()
        if self.cfg.tp_size > 1:
            self._poll_comm_errors(sync=True)
