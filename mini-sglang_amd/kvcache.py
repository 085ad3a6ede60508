"""KV pool behind the reference's KV-cache interface (P/kvcache/base.py:10-37; the pool it replaces: P/kvcache/mha_pool.py:10-68).

One HBM allocation for the process lifetime, sized for 288 GB parts:
    pool[2 (k|v), num_layers, num_pages, page_size, local_kv_heads, head_dim]
One layer's K (or V) is therefore a contiguous slab of token rows, a token's row is local_kv_heads x 256 B
(2048 B at TP1 ... 256 B at TP8), slots are token-granular indices into that slab, and the extra last page is
the dummy target of padded graph rows (P/engine/engine.py:57-63, 89-98).  Only the stand-alone driver of this
repository (engine.py) builds this class; behind the plugin the reference's own MHAKVCache object is used and
its `store_kv` is what gets rebound.
"""
from __future__ import annotations

import torch

from . import ops


def heads_per_rank(total: int, ranks: int, replicate: bool = False) -> int:
    """How many of `total` heads (or MLP columns) one of `ranks` tensor-parallel ranks owns.  With replicate=True a
    group larger than the head count keeps one head per rank, shared by ranks // total neighbours (KV heads at
    tp > Hkv).  Same arithmetic as the reference's sharding helper (P/utils/misc.py:20-26)."""
    if replicate and ranks > total:
        if ranks % total:
            raise ValueError(f"cannot replicate {total} heads over {ranks} ranks (not a multiple)")
        return 1
    if total % ranks:
        raise ValueError(f"cannot split {total} evenly over {ranks} ranks")
    return total // ranks


class MHAKVCache:
    def __init__(self, num_kv_heads: int, num_layers: int, head_dim: int, num_pages: int, page_size: int,
                 dtype: torch.dtype, device: torch.device, tp_size: int = 1) -> None:
        heads = heads_per_rank(num_kv_heads, tp_size, replicate=True)
        slots = num_pages * page_size
        self.pool = torch.empty((2, num_layers, num_pages, page_size, heads, head_dim), device=device, dtype=dtype)
        self._layers = num_layers
        self._tok_shape = (slots, heads, head_dim)   # [slot, head, dim] view of one layer
        self._row_shape = (slots, heads * head_dim)  # [slot, row] view of one layer

    # the reference's accessors ([num_pages, page_size, heads, dim] per layer)
    def k_cache(self, index: int) -> torch.Tensor:
        return self.pool[0, index]

    def v_cache(self, index: int) -> torch.Tensor:
        return self.pool[1, index]

    # token-row views used by the kernels
    def k_rows(self, index: int) -> torch.Tensor:
        return self.pool[0, index].view(self._row_shape)

    def v_rows(self, index: int) -> torch.Tensor:
        return self.pool[1, index].view(self._row_shape)

    def k_tokens(self, index: int) -> torch.Tensor:
        return self.pool[0, index].view(self._tok_shape)

    def v_tokens(self, index: int) -> torch.Tensor:
        return self.pool[1, index].view(self._tok_shape)

    def store_kv(self, k: torch.Tensor, v: torch.Tensor, out_loc: torch.Tensor, layer_id: int) -> None:
        ops.store_kv(self.k_rows(layer_id), self.v_rows(layer_id), out_loc,
                     k.view(k.shape[0], -1), v.view(v.shape[0], -1))

    @property
    def device(self) -> torch.device:
        return self.pool.device

    @property
    def dtype(self) -> torch.dtype:
        return self.pool.dtype

    @property
    def num_layers(self) -> int:
        return self._layers


def create_kvcache_pool(model_config, num_pages: int, page_size: int, dtype: torch.dtype, device: torch.device,
                        tp_size: int = 1) -> MHAKVCache:
    """P/kvcache/__init__.py:27-44."""
    return MHAKVCache(num_kv_heads=model_config.num_kv_heads, num_pages=num_pages, page_size=page_size,
                      num_layers=model_config.num_layers, head_dim=model_config.head_dim, device=device,
                      dtype=dtype, tp_size=tp_size)
