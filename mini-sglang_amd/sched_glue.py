"""Vectorised versions of the per-step tensor glue of the reference's scheduler:
`_make_positions`, `_make_input_tuple`, `_make_write_tuple` (python/minisgl/scheduler/scheduler.py:236-267) build
their index tensors with one Python-level tensor operation PER REQUEST (256 `torch.arange(out=...)` / `fill_` calls for
a full decode batch); here each is a handful of numpy operations over the whole batch.  Same signatures, same
tensors (dtype, pinned staging + non-blocking copy, values) -- `tests/test_cpu_reference_native.py` compares them with
the reference's own functions.  Installed by `minisgl_plugin.install(vectorized_glue=True)`.
"""
from __future__ import annotations

from typing import Any, Tuple

import numpy as np
import torch


def _pinned(n: int, dtype: torch.dtype) -> torch.Tensor:
    return torch.empty(n, dtype=dtype, pin_memory=torch.cuda.is_available())


def _lens(reqs) -> Tuple[np.ndarray, np.ndarray]:
    n = len(reqs)
    cached = np.fromiter((r.cached_len for r in reqs), dtype=np.int64, count=n)
    device = np.fromiter((r.device_len for r in reqs), dtype=np.int64, count=n)
    return cached, device


def make_positions(batch: Any, device: torch.device) -> torch.Tensor:
    """positions[j] for every new token of every padded request: cached_len .. device_len - 1, request after request."""
    cached, dev_len = _lens(batch.padded_reqs)
    lens = dev_len - cached
    total = int(lens.sum())
    host = _pinned(total, torch.int32)
    if total:
        if total == len(lens):  # decode: one token per request
            host.numpy()[:] = cached
        else:
            starts = np.cumsum(lens) - lens
            host.numpy()[:] = np.arange(total, dtype=np.int64) - np.repeat(starts - cached, lens)
    return host.to(device, non_blocking=True)


def make_input_tuple(batch: Any, device: torch.device):
    """(table row of every new token, its position) -- the gather index into token_pool / page_table."""
    cached, dev_len = _lens(batch.padded_reqs)
    rows = np.fromiter((r.table_idx for r in batch.padded_reqs), dtype=np.int64, count=len(cached))
    host = _pinned(len(batch.positions), torch.int64)
    if len(host):
        host.numpy()[:] = np.repeat(rows, dev_len - cached)
    return host.to(device, non_blocking=True), batch.positions.to(torch.int64)


def make_write_tuple(batch: Any, device: torch.device):
    """(table row, column) where each request's sampled token goes: device_len, or the junk column -1 when it cannot
    decode further."""
    n = len(batch.reqs)
    rows = _pinned(n, torch.int64)
    cols = _pinned(n, torch.int64)
    if n:
        rows.numpy()[:] = np.fromiter((r.table_idx for r in batch.reqs), dtype=np.int64, count=n)
        cols.numpy()[:] = np.fromiter((r.device_len if r.can_decode else -1 for r in batch.reqs), dtype=np.int64, count=n)
    return rows.to(device, non_blocking=True), cols.to(device, non_blocking=True)
