"""Registers the MI355X core into an installed `minisgl` (the reference), without touching its
source: after `install()` the reference's scheduler, radix cache, engine, models and server run
unchanged with `--attn hip` (see INTEGRATION.md for the two-line bootstrap).

Seams filled:
  attention backend   SUPPORTED_ATTENTION_BACKENDS.register("hip")        P/attention/__init__.py:19-40
  minisgl.kernel      store_cache / indexing / fast_compare_key / init_pynccl   P/kernel/__init__.py
  flashinfer names    rmsnorm, fused_add_rmsnorm, apply_rope_with_cos_sin_cache_inplace,
                      silu_and_mul, sampling.*                            P/layers/norm.py:10-30 ...
"""
from __future__ import annotations

import importlib
import sys
import types


def _stub_zmq() -> None:
    """Offline `LLM` mode never opens a socket (P/scheduler/io.py:30-33) but `minisgl.utils`
    imports zmq eagerly (P/utils/mp.py:6-7)."""
    try:
        importlib.import_module("zmq")
        return
    except ImportError:
        pass
    zmq = types.ModuleType("zmq")
    for name in ("PUSH", "PULL", "PUB", "SUB", "SUBSCRIBE"):
        setattr(zmq, name, 0)
    zmq.Context = type("Context", (), {})
    zmq_asyncio = types.ModuleType("zmq.asyncio")
    zmq_asyncio.Context = type("Context", (), {})
    zmq.asyncio = zmq_asyncio
    sys.modules["zmq"], sys.modules["zmq.asyncio"] = zmq, zmq_asyncio


def _install_flashinfer_shim() -> None:
    try:
        importlib.import_module("flashinfer")
        return  # a real flashinfer is present: leave it alone
    except ImportError:
        pass
    from . import flashinfer_compat as fc

    mod = types.ModuleType("flashinfer")
    for name in ("rmsnorm", "fused_add_rmsnorm", "apply_rope_with_cos_sin_cache_inplace", "silu_and_mul",
                 "gelu_and_mul"):
        setattr(mod, name, getattr(fc, name))
    samp = types.ModuleType("flashinfer.sampling")
    for name in ("softmax", "sampling_from_probs", "top_k_sampling_from_probs", "top_p_sampling_from_probs",
                 "top_k_top_p_sampling_from_probs"):
        setattr(samp, name, getattr(fc.sampling, name))
    mod.sampling = samp
    sys.modules["flashinfer"], sys.modules["flashinfer.sampling"] = mod, samp


def install(stub_zmq: bool = True) -> None:
    if stub_zmq:
        _stub_zmq()
    _install_flashinfer_shim()
    from . import kernel as k

    import minisgl.kernel as mk  # noqa: E402  (tvm_ffi is only imported inside its functions)

    for name in ("store_cache", "indexing", "fast_compare_key", "init_pynccl", "PyNCCLCommunicator"):
        setattr(mk, name, getattr(k, name))
    for sub, name in (("radix", "fast_compare_key"), ("store", "store_cache"), ("index", "indexing"),
                      ("pynccl", "init_pynccl")):
        setattr(importlib.import_module(f"minisgl.kernel.{sub}"), name, getattr(k, name))

    from minisgl.attention import SUPPORTED_ATTENTION_BACKENDS
    from minisgl.attention.base import BaseAttnBackend

    from .attention import HipAttnBackend

    BaseAttnBackend.register(HipAttnBackend)
    if "hip" not in getattr(SUPPORTED_ATTENTION_BACKENDS, "_registry", {}):
        @SUPPORTED_ATTENTION_BACKENDS.register("hip")
        def create_hip_backend(config):
            from minisgl.core import get_global_ctx
            from minisgl.distributed import get_tp_info

            return HipAttnBackend(config, ctx=get_global_ctx(), tp_size=get_tp_info().size)
