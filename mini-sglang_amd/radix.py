"""Radix prefix cache with the tree walk in native code (csrc/radix.cpp, msgl_radix_* in include/msgl_hip.h).

Mirrors `RadixPrefixCache` / `RadixCacheHandle` (python/minisgl/kvcache/radix_cache.py:80-203): same methods, same
arguments, same results for the same calls -- the structure (keys, reference counts, timestamps, LRU order) lives in
the native tree, the per-node value tensors (KV pool slots, device memory) stay here and follow every split and
eviction the tree reports.  SURVEY.md section 8(f) rank 4.

`NativeRadixTree` has no dependency on the reference; `make_prefix_cache_class()` binds the cache to the reference's
abstract base classes when it is plugged in (minisgl_plugin registers it as cache type "hip_radix").
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib


class NativeRadixTree:
    """ctypes wrapper of one native tree.  Node ids are ints; the root is 0."""

    def __init__(self, page_size: int, clock: Optional[Callable[[], int]] = None):
        self._lib = _lib.lib()
        self.page_size = int(page_size)
        self.clock = clock or (lambda: time.monotonic_ns())  # the reference's clock (radix_cache.py:27,209)
        h = C.c_void_p()
        _lib.check(self._lib.msgl_radix_create(C.byref(h), self.page_size, self.clock()), "radix_create")
        self._h = h
        self._out = np.zeros(8, dtype=np.int64)
        self._ids = np.zeros(1024, dtype=np.int64)

    def close(self) -> None:
        if self._h:
            self._lib.msgl_radix_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _as_i32(ids: torch.Tensor) -> np.ndarray:
        if ids.is_cuda:
            raise RuntimeError("radix keys are host tensors (the reference keeps Req.input_ids on the CPU)")
        a = ids.detach().numpy()
        if a.dtype != np.int32 or not a.flags.c_contiguous:
            a = np.ascontiguousarray(a, dtype=np.int32)
        return a

    def walk(self, ids: torch.Tensor) -> Tuple[int, int, Optional[Tuple[int, int, int]]]:
        """(node, matched length, (split head, split tail, position) or None) -- `_tree_walk`."""
        a = self._as_i32(ids)
        _lib.check(self._lib.msgl_radix_walk(self._h, a.ctypes.data, a.shape[0], self.clock(), self._out.ctypes.data),
                   "radix_walk")
        o = self._out
        split = (int(o[2]), int(o[3]), int(o[4])) if o[2] >= 0 else None
        return int(o[0]), int(o[1]), split

    def add_child(self, parent: int, key: torch.Tensor) -> int:
        a = self._as_i32(key)
        r = self._lib.msgl_radix_add_child(self._h, parent, a.ctypes.data, a.shape[0], self.clock())
        _lib.check(r, "radix_add_child")
        return int(r)

    def lock(self, node: int, unlock: bool) -> None:
        _lib.check(self._lib.msgl_radix_lock(self._h, node, 1 if unlock else 0), "radix_lock")

    def evict(self, size: int) -> List[int]:
        info = self.info()
        if self._ids.shape[0] < info[2]:
            self._ids = np.zeros(int(info[2]) * 2, dtype=np.int64)
        r = self._lib.msgl_radix_evict(self._h, int(size), self._ids.ctypes.data, self._ids.shape[0])
        if r < 0:
            msg = self._lib.msgl_last_error().decode(errors="replace")
            if msg.startswith("Cannot evict"):
                raise AssertionError(msg)  # the reference's own assertion text (radix_cache.py:150-152, 160-162)
            _lib.check(r, "radix_evict")
        return [int(x) for x in self._ids[:r]]

    def path(self, node: int) -> List[int]:
        info = self.info()
        if self._ids.shape[0] < info[2]:
            self._ids = np.zeros(int(info[2]) * 2, dtype=np.int64)
        r = self._lib.msgl_radix_path(self._h, node, self._ids.ctypes.data, self._ids.shape[0])
        _lib.check(r, "radix_path")
        return [int(x) for x in self._ids[:r]]

    def info(self, node: int = -1) -> Tuple[int, int, int, int]:
        _lib.check(self._lib.msgl_radix_info(self._h, node, self._out.ctypes.data), "radix_info")
        o = self._out
        return int(o[0]), int(o[1]), int(o[2]), int(o[3])

    def check(self) -> None:
        _lib.check(self._lib.msgl_radix_check(self._h), "radix_check")


def make_prefix_cache_class(base_cache: type, base_handle: type, match_result: type, insert_result: type,
                            size_info: type, page_size_of: Callable[[], int]) -> Tuple[type, type]:
    """(cache class, handle class) deriving from the given abstract bases (the reference's
    `BasePrefixCache` / `BaseCacheHandle`, python/minisgl/kvcache/base.py:38-135) and returning its result tuples."""
    from dataclasses import dataclass

    @dataclass(frozen=True)
    class NativeRadixHandle(base_handle):  # type: ignore[misc, valid-type]
        node: int
        cache: Any

        def get_matched_indices(self) -> torch.Tensor:  # radix_cache.py:87-94
            return self.cache._matched_indices(self.node)

    class NativeRadixPrefixCache(base_cache):  # type: ignore[misc, valid-type]
        def __init__(self, device: torch.device, clock: Optional[Callable[[], int]] = None):
            super().__init__()
            self.device = device
            self.page_size = int(page_size_of())
            self.tree = NativeRadixTree(self.page_size, clock)
            self.empty_tensor = torch.empty(0, dtype=torch.int32, device=device)
            self.values: Dict[int, torch.Tensor] = {}

        # ---- value tensors follow the tree
        def _mirror_split(self, split: Optional[Tuple[int, int, int]]) -> None:
            if split is not None:
                head, tail, pos = split
                v = self.values[tail]
                self.values[head], self.values[tail] = v[:pos], v[pos:]

        def _matched_indices(self, node: int) -> torch.Tensor:
            return torch.cat([self.values[n] for n in self.tree.path(node)])

        # ---- BasePrefixCache
        def lock_handle(self, handle, unlock: bool = False) -> None:
            assert isinstance(handle, NativeRadixHandle)
            self.tree.lock(handle.node, unlock)

        def match_prefix(self, input_ids: torch.Tensor):
            node, prefix_len, split = self.tree.walk(input_ids)
            self._mirror_split(split)
            return match_result(NativeRadixHandle(prefix_len, node, self))

        def insert_prefix(self, input_ids: torch.Tensor, indices: torch.Tensor):
            insert_len = len(input_ids) // self.page_size * self.page_size
            input_ids, indices = input_ids[:insert_len], indices[:insert_len]
            node, prefix_len, split = self.tree.walk(input_ids)
            self._mirror_split(split)
            if prefix_len != insert_len:
                node = self.tree.add_child(node, input_ids[prefix_len:])
                self.values[node] = indices[prefix_len:].clone()
            return insert_result(prefix_len, NativeRadixHandle(insert_len, node, self))

        def evict(self, size: int) -> torch.Tensor:
            if size == 0:
                return self.empty_tensor
            return torch.cat([self.values.pop(n) for n in self.tree.evict(size)])

        def reset(self) -> None:
            raise NotImplementedError("RadixManager.reset is not implemented")

        @property
        def size_info(self):
            ev, pr, _, _ = self.tree.info()
            return size_info(evictable_size=ev, protected_size=pr)

        @property
        def evictable_size(self) -> int:
            return self.tree.info()[0]

        @property
        def protected_size(self) -> int:
            return self.tree.info()[1]

        def check_integrity(self) -> None:
            self.tree.check()
            ev, pr, live, _ = self.tree.info()
            if live != len(self.values) + 1 or sum(len(v) for v in self.values.values()) != ev + pr:
                raise RuntimeError("radix values out of step with the tree")

    return NativeRadixPrefixCache, NativeRadixHandle
