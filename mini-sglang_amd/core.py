"""Attribute holders for the standalone driver (engine.py / offline.py / bench.py on the GPU box).

When the reference is installed, its own `minisgl.core` objects (P/core.py:16-136) are what reach the
attention backend and the KV pool; this module is NOT a replacement for them.  The backend only reads
attributes -- `Batch.{reqs, padded_reqs, phase, positions, out_loc, input_ids, attn_metadata}` and
`Req.{table_idx, cached_len, device_len, extend_len}` -- so the standalone driver hands
it these minimal holders carrying the same attribute names and nothing else of the reference's class bodies.
"""
from __future__ import annotations

from contextlib import contextmanager
from typing import Any, Iterator, List, Optional

import torch


class SamplingParams:
    __slots__ = ("temperature", "top_k", "top_p", "ignore_eos", "max_tokens")

    def __init__(self, temperature: float = 0.0, top_k: int = -1, top_p: float = 1.0, ignore_eos: bool = False,
                 max_tokens: int = 1024) -> None:
        self.temperature, self.top_k, self.top_p = temperature, top_k, top_p
        self.ignore_eos, self.max_tokens = ignore_eos, max_tokens

    @property
    def is_greedy(self) -> bool:
        """Greedy rule of the reference sampler (P/core.py:23-25)."""
        no_randomness = self.temperature <= 0.0 or self.top_k == 1
        return no_randomness and self.top_p == 1.0


class Req:
    """One sequence's progress: `cached_len` tokens have K/V in the pool, `device_len` tokens exist on the
    device; a forward extends the pool by `extend_len = device_len - cached_len` tokens."""

    __slots__ = ("input_ids", "table_idx", "cached_len", "device_len", "max_device_len", "uid", "sampling_params",
                 "cache_handle")

    def __init__(self, input_ids: torch.Tensor, table_idx: int, cached_len: int, output_len: int, uid: int,
                 sampling_params: Optional[SamplingParams] = None, cache_handle: Any = None) -> None:
        n = int(input_ids.shape[0])
        if not (input_ids.device.type == "cpu" and 0 <= cached_len < n):
            raise ValueError(f"bad request: {n} host tokens, cached_len {cached_len}")
        self.input_ids, self.table_idx, self.uid = input_ids, table_idx, uid
        self.cached_len, self.device_len, self.max_device_len = cached_len, n, n + output_len
        self.sampling_params, self.cache_handle = sampling_params, cache_handle

    extend_len = property(lambda self: self.device_len - self.cached_len)
    remain_len = property(lambda self: self.max_device_len - self.device_len)
    can_decode = property(lambda self: self.max_device_len > self.device_len)

    def complete_one(self) -> None:
        """Host bookkeeping right after a forward is enqueued (P/core.py:52-54): everything on the device is now
        cached and one sampled token is about to exist."""
        self.cached_len, self.device_len = self.device_len, self.device_len + 1


class Batch:
    __slots__ = ("reqs", "phase", "input_ids", "positions", "out_loc", "padded_reqs", "attn_metadata")

    def __init__(self, reqs: List[Req], phase: str) -> None:
        if phase not in ("prefill", "decode"):
            raise ValueError(phase)
        self.reqs, self.phase = reqs, phase
        self.padded_reqs: List[Req] = reqs

    is_prefill = property(lambda self: self.phase == "prefill")
    is_decode = property(lambda self: self.phase == "decode")
    size = property(lambda self: len(self.reqs))
    padded_size = property(lambda self: len(self.padded_reqs))


class Context:
    """Per-process state the backend is constructed from (P/core.py:101-126): the token-slot page table
    (page-size agnostic), the KV pool, the backend itself, and the batch being forwarded."""

    def __init__(self, page_size: int) -> None:
        self.page_size = page_size
        self.page_table: torch.Tensor = None  # type: ignore[assignment]
        self.kv_cache: Any = None
        self.attn_backend: Any = None
        self._batch: Optional[Batch] = None

    @property
    def batch(self) -> Batch:
        if self._batch is None:
            raise RuntimeError("no batch is being forwarded")
        return self._batch

    @contextmanager
    def forward_batch(self, batch: Batch) -> Iterator[None]:
        if self._batch is not None:
            raise RuntimeError("forward_batch does not nest")
        self._batch = batch
        try:
            yield
        finally:
            self._batch = None


_CTX: Optional[Context] = None


def set_global_ctx(ctx: Optional[Context], *, force: bool = False) -> None:
    global _CTX
    if _CTX is not None and ctx is not None and not force:
        raise RuntimeError("global context already set")
    _CTX = ctx


def get_global_ctx() -> Context:
    if _CTX is None:
        raise RuntimeError("global context not set")
    return _CTX
