"""HipAttnBackend: the MI355X attention backend behind the reference's plugin API.

Implements the five abstract methods of BaseAttnBackend (P/attention/base.py:18-34) and the
metadata contract (`get_last_indices`, P/attention/base.py:12-15, fa.py:32-33) with the gfx950
kernels.  Differences from the reference backends that matter for speed, none for results:

* the page table is read IN PLACE: the kernels walk `ctx.page_table[req.table_idx, :len]`
  (token slots, page-size agnostic) directly, so `prepare_metadata` builds no per-step
  `[B, max_k / page]` table (cf. fa.py:92-97: B slice views + a stack + a div per step) and
  graph replay copies two `[B]` int vectors instead of a `[B, max_seq/page]` table;
* all per-batch integers travel in ONE pinned host buffer / ONE async H2D copy;
* decode work is split by a device-side plan (no host sync, fixed grids => graph-safe).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional

import numpy as np
import torch

from . import _lib, ops
from . import core as _core


@dataclass
class HipAttnMetadata:
    cu_seqlens_q: torch.Tensor  # [B+1] int32, device
    seq_lens: torch.Tensor      # [B]   int32, device (= cache_seqlens / device_len)
    req_rows: torch.Tensor      # [B]   int32, device (= req.table_idx)
    batch: int
    max_seqlen_q: int
    max_seqlen_k: int
    tile_cu: Optional[torch.Tensor] = None  # [B+1] int32 (prefill only)
    total_tiles: int = 0
    tile_order: Optional[torch.Tensor] = None  # [total_tiles] int32: q tiles, most keys first (prefill only)
    plan: Optional[torch.Tensor] = None     # decode work list (device)

    def get_last_indices(self, bs: int) -> torch.Tensor:
        return self.cu_seqlens_q[1: 1 + bs] - 1


def prefill_tile_order(seqlens_q: np.ndarray, seqlens_k: np.ndarray, tiles: np.ndarray,
                       q_tile: Optional[int] = None) -> np.ndarray:
    """Launch order of the q tiles (q_tile rows each; default: the default prefill kernel's) of a prefill batch: by
    decreasing number of keys the tile attends (tile t of request b sees keys [0, min(k_b, k_b - q_b + min((t + 1) *
    q_tile, q_b)))), ties in natural order.  A scheduling hint only (longest-processing-time first shortens the launch
    tail); results do not depend on it."""
    if q_tile is None:
        q_tile = ops.prefill_q_tile()
    total = int(tiles.sum())
    req = np.repeat(np.arange(len(tiles)), tiles)
    first = np.cumsum(tiles) - tiles
    t_in = np.arange(total) - first[req]
    q, k = seqlens_q[req], seqlens_k[req]
    kend = np.minimum(k, k - q + np.minimum((t_in + 1) * q_tile, q))
    return np.argsort(-kend, kind="stable").astype(np.int32)


def fill_metadata_host(h: np.ndarray, seqlens_q: np.ndarray, seqlens_k: np.ndarray, rows: np.ndarray,
                       decode: bool, q_tile: int = _lib.PREFILL_QTILE) -> None:
    """Per-batch integers of `prepare_metadata` (fa.py:67-105) in one int32 buffer of 4 B + 2 (+ tiles) words:
    [seq_lens = device_len (cache_seqlens) | rows = table_idx | cu_seqlens_q [B+1] | tile_cu [B+1] | tile_order].
    cu_seqlens_q is the reference's in all three regimes (arange for decode, = cu_seqlens_k without a cache
    hit, own cumsum with one): it is always the exclusive prefix of extend_len."""
    bs = len(seqlens_q)
    h[:bs] = seqlens_k
    h[bs: 2 * bs] = rows
    h[2 * bs] = 0
    h[2 * bs + 1: 3 * bs + 1] = np.cumsum(seqlens_q)
    if not decode:
        tiles = (seqlens_q + q_tile - 1) // q_tile
        h[3 * bs + 1] = 0
        h[3 * bs + 2: 4 * bs + 2] = np.cumsum(tiles)
        h[4 * bs + 2:] = prefill_tile_order(seqlens_q, seqlens_k, tiles, q_tile)


class HipAttnBackend:
    """Paged attention (prefill + decode) over the reference's KV pool and global page table."""

    def __init__(self, config: Any, ctx: Any = None, *, tp_size: int = 1, decode_capacity: int = 0) -> None:
        ctx = ctx if ctx is not None else _core.get_global_ctx()
        self.ctx = ctx
        self.config = config
        self.kvcache = ctx.kv_cache
        self.page_size = ctx.page_size
        # aligned runs of this many positions never cross a page => consecutive slots (cache.py:42-53)
        self.slot_run = self.page_size & -self.page_size
        self.device = self.kvcache.device
        self.head_dim = config.head_dim
        self.scale = config.head_dim ** -0.5
        self.tp_size = tp_size
        self._q_tile = ops.prefill_q_tile()  # query rows per tile of the prefill kernel in use (unit of tile_cu / tile_order)
        self.qo_heads = config.num_qo_heads // tp_size
        self.kv_heads = max(config.num_kv_heads // tp_size, 1)
        self.max_bs = int(ctx.page_table.shape[0])
        self.capacity = decode_capacity or max(4096, 4 * self.max_bs)
        self._workspace = torch.empty(ops.attn_decode_workspace_bytes(self.capacity, self.qo_heads, self.head_dim),
                                      dtype=torch.uint8, device=self.device)
        self._plan_words = ops.attn_decode_plan_words(self.max_bs, self.capacity)
        # graph state
        self.capture_bs: List[int] = []
        self.max_graph_bs = 0
        self._cap_seq: Optional[torch.Tensor] = None
        self._cap_rows: Optional[torch.Tensor] = None
        self._cap_plan: Optional[torch.Tensor] = None
        self._cap_cu_q: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ forward
    def _kv_tokens(self, layer_id: int):
        k = self.kvcache.k_cache(layer_id)
        v = self.kvcache.v_cache(layer_id)
        return k.view(-1, k.shape[-2], k.shape[-1]), v.view(-1, v.shape[-2], v.shape[-1])

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, layer_id: int, batch: Any) -> torch.Tensor:
        """Reference contract (fa.py:48-65): persist k, v at batch.out_loc, then attend."""
        self.kvcache.store_kv(k, v, batch.out_loc, layer_id)
        return self.attend(q, layer_id, batch)

    def attend(self, q: torch.Tensor, layer_id: int, batch: Any) -> torch.Tensor:
        """Attention only (K/V for this batch already in the pool, e.g. via the fused
        qk-norm + RoPE + store kernel)."""
        md: HipAttnMetadata = batch.attn_metadata
        q = q.view(-1, self.qo_heads, self.head_dim)
        out = torch.empty((q.shape[0], self.qo_heads, self.head_dim), dtype=q.dtype, device=q.device)
        k_tok, v_tok = self._kv_tokens(layer_id)
        table = self.ctx.page_table
        if md.max_seqlen_q == 1:
            if md.plan is None:  # prepared for replay but run eagerly by the host: plan on first use
                md.plan = torch.empty(self._plan_words, dtype=torch.int32, device=self.device)
                ops.attn_decode_plan(md.plan, md.seq_lens, md.batch, self.max_bs, self.capacity, self.qo_heads,
                                     self.kv_heads)
            ops.attn_decode(out, q, k_tok, v_tok, table, md.req_rows, md.seq_lens, md.plan, self._workspace,
                            md.batch, self.max_bs, self.capacity, self.scale, slot_run=self.slot_run)
        else:
            ops.attn_prefill(out, q, k_tok, v_tok, table, md.req_rows, md.seq_lens, md.cu_seqlens_q, md.tile_cu,
                             md.batch, md.total_tiles, self.scale, tile_order=md.tile_order)
        return out

    # ------------------------------------------------------------------ metadata
    def prepare_metadata(self, batch: Any) -> None:
        if self.tp_size > 1:
            # driven by the reference's scheduler nobody else sees the communicator once decode runs from captured
            # graphs: every 16th step look at the peer-to-peer error word (one 4-byte async copy; raises on a timeout)
            self._md_calls = getattr(self, "_md_calls", 0) + 1
            if self._md_calls % 16 == 0:
                from .kernel import poll_communicator_errors

                poll_communicator_errors(sync=False)
        reqs = batch.padded_reqs
        bs = len(reqs)
        seqlens_q = np.fromiter((r.extend_len for r in reqs), dtype=np.int64, count=bs)
        seqlens_k = np.fromiter((r.device_len for r in reqs), dtype=np.int64, count=bs)
        rows = np.fromiter((r.table_idx for r in reqs), dtype=np.int64, count=bs)
        max_q, max_k = int(seqlens_q.max()), int(seqlens_k.max())
        decode = max_q == 1
        q_tile = self._q_tile
        total_tiles = 0 if decode else int(((seqlens_q + q_tile - 1) // q_tile).sum())
        # one pinned buffer, one async H2D copy: [seq_lens | rows | cu_q | tile_cu | tile_order]
        host = torch.empty(4 * bs + 2 + total_tiles, dtype=torch.int32, pin_memory=True)
        fill_metadata_host(host.numpy(), seqlens_q, seqlens_k, rows, decode, q_tile)
        dev = host.to(self.device, non_blocking=True)
        md = HipAttnMetadata(
            cu_seqlens_q=dev[2 * bs: 3 * bs + 1], seq_lens=dev[:bs], req_rows=dev[bs: 2 * bs], batch=bs,
            max_seqlen_q=max_q, max_seqlen_k=max_k, tile_cu=None if decode else dev[3 * bs + 1: 4 * bs + 2],
            total_tiles=total_tiles, tile_order=None if decode else dev[4 * bs + 2:],
        )
        if decode and not self._replayed_from_graph(batch, bs):
            # runs eagerly: plan now (device side, no sync).  Only batches the engine will REPLAY are planned
            # later, in prepare_for_replay; replay is keyed on the phase (P/engine/graph.py:149-150
            # can_use_cuda_graph = batch.is_decode and size <= max_graph_bs), NOT on max_q == 1: a prefill
            # batch whose every request extends by one token (radix full hit on a repeated prompt at
            # page_size 1, P/scheduler/cache.py:27-30; a one-token chunk remainder) takes this kernel eagerly
            md.plan = torch.empty(self._plan_words, dtype=torch.int32, device=self.device)
            ops.attn_decode_plan(md.plan, md.seq_lens, bs, self.max_bs, self.capacity, self.qo_heads, self.kv_heads)
        batch.attn_metadata = md

    def _replayed_from_graph(self, batch: Any, padded_bs: int) -> bool:
        phase = getattr(batch, "phase", None)
        is_decode = (phase == "decode") if phase is not None else True
        return bool(is_decode and self._cap_plan is not None and padded_bs in self.capture_bs
                    and len(batch.reqs) <= self.max_graph_bs)

    # ------------------------------------------------------------------ graph hooks
    def init_capture_graph(self, max_seq_len: int, bs_list: List[int]) -> None:
        assert self._cap_plan is None, "Capture already initialized."
        max_bs = max(bs_list)
        assert max_bs <= self.max_bs
        self.max_graph_bs = max_bs
        self.capture_bs = sorted(bs_list)
        self._cap_seq = torch.ones(max_bs, dtype=torch.int32, device=self.device)
        self._cap_rows = torch.zeros(max_bs, dtype=torch.int32, device=self.device)
        self._cap_cu_q = torch.arange(0, max_bs + 1, dtype=torch.int32, device=self.device)
        self._cap_plan = torch.zeros(self._plan_words, dtype=torch.int32, device=self.device)

    def prepare_for_capture(self, batch: Any) -> None:
        bs = batch.size
        assert bs in self.capture_bs and self._cap_plan is not None
        # dummy requests: length 1, all rows = the dummy request's table row
        self._cap_rows[:bs].fill_(batch.reqs[0].table_idx)
        self._cap_seq[:bs].fill_(1)
        ops.attn_decode_plan(self._cap_plan, self._cap_seq, bs, self.max_bs, self.capacity, self.qo_heads, self.kv_heads)
        batch.attn_metadata = HipAttnMetadata(
            cu_seqlens_q=self._cap_cu_q[: bs + 1], seq_lens=self._cap_seq[:bs], req_rows=self._cap_rows[:bs],
            batch=bs, max_seqlen_q=1, max_seqlen_k=int(self.ctx.page_table.shape[1]), plan=self._cap_plan,
        )

    def prepare_for_replay(self, batch: Any) -> None:
        md, bs = batch.attn_metadata, batch.padded_size
        assert isinstance(md, HipAttnMetadata) and bs in self.capture_bs and self._cap_plan is not None
        self._cap_seq[:bs].copy_(md.seq_lens)
        self._cap_rows[:bs].copy_(md.req_rows)
        ops.attn_decode_plan(self._cap_plan, self._cap_seq, bs, self.max_bs, self.capacity, self.qo_heads, self.kv_heads)
