"""Mirror of `minisgl.kernel` (P/kernel/__init__.py): same names, argument meaning and
error behaviour, implemented on the gfx950 C-ABI instead of tvm-ffi JIT modules."""
from __future__ import annotations

import ctypes
from typing import Literal, Optional, Tuple

import torch

from . import _lib, ops


def store_cache(k_cache: torch.Tensor, v_cache: torch.Tensor, indices: torch.Tensor, k: torch.Tensor,
                v: torch.Tensor) -> None:
    """P/kernel/store.py:30-42: caches are viewed as [num_tokens, -1] rows."""
    num_tokens = k_cache.shape[0]
    ops.store_kv(k_cache.view(num_tokens, -1), v_cache.view(num_tokens, -1), indices,
                 k.view(k.shape[0], -1) if k.dim() > 2 else k, v.view(v.shape[0], -1) if v.dim() > 2 else v)


def indexing(weights: torch.Tensor, indices: torch.Tensor, *, output: Optional[torch.Tensor] = None,
             vocab_range: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """P/kernel/index.py:30-50."""
    return ops.embedding_gather(weights, indices, out=output, vocab_range=vocab_range)


def fast_compare_key(x: torch.Tensor, y: torch.Tensor) -> int:
    """P/kernel/radix.py:18-20."""
    return ops.fast_compare_key(x, y)


class RcclCommunicator:
    """Same surface as PyNCCLCommunicator (P/kernel/pynccl.py:16-26)."""

    def __init__(self, rank: int, world_size: int, max_bytes: int, unique_id: bytes) -> None:
        self._lib = _lib.comm_lib()
        self._handle = ctypes.c_void_p()
        self.rank, self.world_size = rank, world_size
        _lib.check_comm(
            self._lib.msgl_comm_create(ctypes.byref(self._handle), rank, world_size, unique_id, max_bytes),
            "comm_create",
        )

    def all_reduce(self, input: torch.Tensor, op: Literal["sum"] = "sum") -> None:
        if op != "sum":
            raise ValueError(f"unsupported reduce op {op!r}")
        if not (input.is_cuda and input.is_contiguous()):
            raise RuntimeError("Tensor must be a contiguous device tensor")
        _lib.check_comm(
            self._lib.msgl_comm_all_reduce_sum(self._handle, input.data_ptr(), input.numel(), ops._dt(input),
                                               torch.cuda.current_stream().cuda_stream),
            "all_reduce",
        )

    def all_gather(self, output: torch.Tensor, input: torch.Tensor) -> None:
        if not (input.is_cuda and input.is_contiguous() and output.is_cuda and output.is_contiguous()):
            raise RuntimeError("Tensor must be a contiguous device tensor")
        if output.shape[0] != input.shape[0] * self.world_size:
            raise RuntimeError("Destination tensor has incorrect size")
        _lib.check_comm(
            self._lib.msgl_comm_all_gather(self._handle, output.data_ptr(), input.data_ptr(), input.numel(),
                                           ops._dt(input), torch.cuda.current_stream().cuda_stream),
            "all_gather",
        )

    def get_buffer(self) -> int:
        return int(self._lib.msgl_comm_get_buffer(self._handle) or 0)

    def destroy(self) -> None:
        if self._handle:
            self._lib.msgl_comm_destroy(self._handle)
            self._handle = ctypes.c_void_p()


PyNCCLCommunicator = RcclCommunicator


def create_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check_comm(_lib.comm_lib().msgl_comm_unique_id(buf), "comm_unique_id")
    return buf.raw


def init_pynccl(*, tp_rank: int, tp_size: int, tp_cpu_group, max_size_bytes: int = 0) -> RcclCommunicator:
    """P/kernel/pynccl.py:47-78: rank 0 creates the unique id, broadcast over the CPU (gloo) group."""
    import torch.distributed as dist

    id_list = [create_unique_id() if tp_rank == 0 else None]
    dist.broadcast_object_list(id_list, src=0, group=tp_cpu_group)
    uid = id_list[0]
    assert uid is not None, f"Failed to get RCCL unique ID on {tp_rank = }"
    return RcclCommunicator(tp_rank, tp_size, max_size_bytes, uid)


__all__ = ["indexing", "fast_compare_key", "store_cache", "init_pynccl", "PyNCCLCommunicator", "RcclCommunicator"]
