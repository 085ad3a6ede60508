"""KV pool mirror of MHAKVCache (P/kvcache/mha_pool.py:10-68, interface P/kvcache/base.py:10-37).

Layout in HBM (one allocation for the process lifetime, sized for 288 GB parts):
    [2 (k|v), num_layers, num_pages, page_size, local_kv_heads, head_dim]
so one layer's K (or V) is a contiguous [num_pages * page_size, local_kv_heads * head_dim] slab
of token rows: a token's row is local_kv_heads x 256 B contiguous (2048 B at TP1 ... 256 B at
TP8), slots are token-granular indices into that slab, and the extra last page is the dummy
target of padded graph rows (P/engine/engine.py:57-63, 89-98).
"""
from __future__ import annotations

import torch

from . import ops


def div_even(a: int, b: int, allow_replicate: bool = False) -> int:
    """P/utils/misc.py:20-26."""
    if allow_replicate and b > a:
        assert b % a == 0, f"{b = } must be divisible by {a = } for KV head replication"
        return 1
    assert a % b == 0, f"{a = } must be divisible by {b = }"
    return a // b


class MHAKVCache:
    def __init__(self, num_kv_heads: int, num_layers: int, head_dim: int, num_pages: int, page_size: int,
                 dtype: torch.dtype, device: torch.device, tp_size: int = 1) -> None:
        local_kv_heads = div_even(num_kv_heads, tp_size, allow_replicate=True)
        self._kv_buffer = torch.empty((2, num_layers, num_pages, page_size, local_kv_heads, head_dim),
                                      device=device, dtype=dtype)
        self._num_layers = num_layers
        self._k_buffer = self._kv_buffer[0]
        self._v_buffer = self._kv_buffer[1]
        self._device = device
        self._storage_shape = (num_pages * page_size, local_kv_heads, head_dim)
        self._row_shape = (num_pages * page_size, local_kv_heads * head_dim)

    def k_cache(self, index: int) -> torch.Tensor:
        return self._k_buffer[index]

    def v_cache(self, index: int) -> torch.Tensor:
        return self._v_buffer[index]

    # token-row views used by the kernels
    def k_rows(self, index: int) -> torch.Tensor:
        return self._k_buffer[index].view(self._row_shape)

    def v_rows(self, index: int) -> torch.Tensor:
        return self._v_buffer[index].view(self._row_shape)

    def k_tokens(self, index: int) -> torch.Tensor:
        return self._k_buffer[index].view(self._storage_shape)

    def v_tokens(self, index: int) -> torch.Tensor:
        return self._v_buffer[index].view(self._storage_shape)

    def store_kv(self, k: torch.Tensor, v: torch.Tensor, out_loc: torch.Tensor, layer_id: int) -> None:
        ops.store_kv(self.k_rows(layer_id), self.v_rows(layer_id), out_loc,
                     k.view(k.shape[0], -1), v.view(v.shape[0], -1))

    @property
    def device(self) -> torch.device:
        return self._device

    @property
    def dtype(self) -> torch.dtype:
        return self._kv_buffer.dtype

    @property
    def num_layers(self) -> int:
        return self._num_layers


def create_kvcache_pool(model_config, num_pages: int, page_size: int, dtype: torch.dtype, device: torch.device,
                        tp_size: int = 1) -> MHAKVCache:
    """P/kvcache/__init__.py:27-44."""
    return MHAKVCache(num_kv_heads=model_config.num_kv_heads, num_pages=num_pages, page_size=page_size,
                      num_layers=model_config.num_layers, head_dim=model_config.head_dim, device=device,
                      dtype=dtype, tp_size=tp_size)
