"""ctypes view of oracle_bytes.c (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import torch

_DIR = Path(__file__).resolve().parent
_SO = _DIR / "_build" / "liboracle_bytes.so"


def build() -> Path:
    subprocess.run(["make", "-s", "-C", str(_DIR)], check=True)
    return _SO


_dll = None


def _lib() -> C.CDLL:
    global _dll
    if _dll is None:
        if not _SO.exists():
            build()
        _dll = C.CDLL(str(_SO))
        _dll.orc_compare_key.restype = C.c_int64
    return _dll


def _p(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _i64(x: int) -> C.c_int64:
    return C.c_int64(int(x))


def store_kv(k_cache, v_cache, indices, k, v) -> None:
    """Row-strided CPU tensors [slots,row] / [T,row]; in place on the caches."""
    es = k.element_size()
    assert k.stride(0) == v.stride(0) and k_cache.stride(0) == v_cache.stride(0)
    _lib().orc_store_kv(_p(k_cache), _p(v_cache), _p(indices), int(indices.dtype == torch.int64), _p(k), _p(v),
                        _i64(k.shape[0]), _i64(k.shape[1] * es), _i64(k_cache.stride(0) * es),
                        _i64(k.stride(0) * es))


def index(weights, indices, vocab_range=None) -> torch.Tensor:
    out = torch.empty((indices.shape[0], weights.shape[1]), dtype=weights.dtype)
    rb = weights.shape[1] * weights.element_size()
    if vocab_range is None:
        _lib().orc_index(_p(out), _p(weights), _p(indices), int(indices.dtype == torch.int64),
                         _i64(indices.shape[0]), _i64(rb))
    else:
        _lib().orc_index_masked(_p(out), _p(weights), _p(indices), int(indices.dtype == torch.int64),
                                _i64(indices.shape[0]), _i64(rb), C.c_uint64(vocab_range[0]),
                                C.c_uint64(vocab_range[1]))
    return out


def compare_key(x, y) -> int:
    return int(_lib().orc_compare_key(_p(x), _i64(x.numel()), _p(y), _i64(y.numel()), x.element_size()))


def page_to_token(pages: torch.Tensor, page_size: int) -> torch.Tensor:
    out = torch.empty(pages.numel() * page_size, dtype=torch.int32)
    _lib().orc_page_to_token(_p(out), _p(pages), _i64(pages.numel()), C.c_int32(page_size))
    return out
