"""ctypes binding of the C-ABI declared in include/msgl_hip.h.

The shared objects are built in-tree by build.py (hipcc --offload-arch=gfx950).  There is
no CPU fallback: if a library is missing, `lib()` raises; every wrapper in ops.py goes
through it, so the product path fails loudly instead of silently computing elsewhere.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

# Load order matters: PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64.  Importing torch
# first makes our shared objects bind to that already-loaded runtime (same SONAME) instead of
# pulling a second copy from /opt/rocm, which would see no device.
import torch  # noqa: F401

_PKG = Path(__file__).resolve().parent
LIB_DIR = _PKG / "lib"
HIP_SO = LIB_DIR / "libmsgl_hip.so"
COMM_SO = LIB_DIR / "libmsgl_comm.so"
GEMM_SO = LIB_DIR / "libmsgl_gemm.so"

BF16, FP16, F32 = 0, 1, 2
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
PREFILL_QTILE = 128
ABI_VERSION = 5

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_u64, _sz = C.c_uint64, C.c_size_t

# name -> (restype, argtypes); mirrors include/msgl_hip.h one to one
HIP_SIGNATURES = {
    "msgl_last_error": (C.c_char_p, []),
    "msgl_abi_version": (_i, []),
    "msgl_device_cu_count": (_i, []),
    "msgl_store_kv": (_i, [_p, _p, _p, _i, _p, _p, _l, _l, _l, _l, _l, _p]),
    "msgl_embedding_gather": (_i, [_p, _p, _p, _i, _l, _l, _i, _l, _l, _p]),
    "msgl_fast_compare_key": (_l, [_p, _l, _p, _l, _i]),
    "msgl_rmsnorm": (_i, [_p, _p, _p, _f, _l, _l, _l, _l, _l, _l, _l, _i, _p]),
    "msgl_fused_add_rmsnorm": (_i, [_p, _p, _p, _f, _l, _l, _l, _l, _i, _p]),
    "msgl_fused_add_rmsnorm_slabs": (_i, [_p, _p, _p, _f, _l, _l, _l, _l, _p, _i, _l, _l, _i, _p]),
    "msgl_rope_neox_inplace": (_i, [_p, _p, _p, _i, _p, _l, _i, _i, _i, _l, _l, _i, _p]),
    "msgl_qk_norm_rope_store": (
        _i,
        [_p, _p, _p, _p, _p, _f, _p, _i, _p, _p, _p, _p, _i, _l, _i, _i, _i, _l, _l, _l, _l, _i, _p],
    ),
    "msgl_qk_norm_rope_store_slabs": (
        _i,
        [_p, _l, _p, _i, _l, _l, _p, _p, _f, _p, _i, _p, _p, _p, _p, _i, _l, _i, _i, _i, _l, _i, _p],
    ),
    "msgl_silu_and_mul": (_i, [_p, _p, _l, _l, _l, _l, _i, _p]),
    "msgl_gelu_and_mul": (_i, [_p, _p, _l, _l, _l, _l, _i, _p]),
    "msgl_silu_and_mul_interleaved": (_i, [_p, _p, _l, _l, _l, _l, _i, _p]),
    "msgl_attn_decode_select": (_i, [_i]),
    "msgl_attn_decode_trace": (_i, [_p]),
    "msgl_attn_decode_plan_words": (_l, [_i, _i]),
    "msgl_attn_decode_workspace_bytes": (_l, [_i, _i, _i]),
    "msgl_attn_decode_plan": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "msgl_attn_decode": (
        _i,
        [_p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _l, _l, _l, _f, _i, _i, _p],
    ),
    "msgl_attn_prefill": (
        _i,
        [_p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _i, _i, _i, _i, _i, _l, _l, _l, _l, _f, _i, _p, _i, _p],
    ),
    "msgl_attn_prefill_q_tile": (_i, [_i]),
    "msgl_argmax_rows": (_i, [_p, _p, _l, _l, _l, _i, _p]),
    "msgl_softmax_temperature": (_i, [_p, _p, _p, _l, _l, _l, _l, _i, _p]),
    "msgl_sample_top_k_top_p": (_i, [_p, _p, _p, _p, _l, _l, _l, _u64, _u64, _p]),
    "msgl_sample_from_logits": (_i, [_p, _p, _p, _l, _l, _l, _i, _u64, _u64, _p]),
    "msgl_skinny_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _p]),
    "msgl_skinny_gemm_silu_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _p]),
    "msgl_rowstream_gemm_supported": (_i, [_i, _i, _i, _i, _i]),
    "msgl_rowstream_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _p, _p, _p, _f, _l, _l, _p]),
    "msgl_wstream_gemm_workspace_bytes": (_l, [_i, _i, _i]),
    "msgl_wstream_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _p, _l, _p]),
    "msgl_wstream_gemm_slabs_nt": (_i, [_p, _p, _i, _i, _i, _l, _l, _i, _i, _i, _p, _l, _p]),
    "msgl_p2p_create": (_i, [C.POINTER(_p), _i, _i, _sz]),
    "msgl_p2p_ipc_handle": (_i, [_p, C.c_char_p]),
    "msgl_p2p_open": (_i, [_p, C.c_char_p]),
    "msgl_p2p_configure": (_i, [_p, _sz, _i]),
    "msgl_p2p_all_reduce_sum": (_i, [_p, _p, _sz, _i, _p]),
    "msgl_p2p_all_gather": (_i, [_p, _p, _p, _sz, _i, _p]),
    "msgl_p2p_all_reduce_add_rmsnorm": (_i, [_p, _p, _p, _p, _f, _l, _l, _l, _l, _i, _p]),
    "msgl_p2p_error": (_i, [_p]),
    "msgl_p2p_error_async": (_i, [_p, _p, _p]),
    "msgl_p2p_set_spin_limit": (_i, [_p, C.c_uint32]),
    "msgl_p2p_get_buffer": (_p, [_p]),
    "msgl_p2p_destroy": (_i, [_p]),
    "msgl_p2p_release_all": (_i, []),
    "msgl_radix_create": (_i, [C.POINTER(_p), _i, _l]),
    "msgl_radix_destroy": (_i, [_p]),
    "msgl_radix_walk": (_i, [_p, _p, _l, _l, _p]),
    "msgl_radix_add_child": (_l, [_p, _l, _p, _l, _l]),
    "msgl_radix_lock": (_i, [_p, _l, _i]),
    "msgl_radix_evict": (_l, [_p, _l, _p, _l]),
    "msgl_radix_path": (_l, [_p, _l, _p, _l]),
    "msgl_radix_info": (_i, [_p, _l, _p]),
    "msgl_radix_check": (_i, [_p]),
    "msgl_m256_gemm_workspace_bytes": (_l, [_i, _i, _i, _i]),
    "msgl_m256_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _p, _l, _p]),
    "msgl_m256_gemm_slabs_nt": (_i, [_p, _p, _i, _i, _i, _l, _l, _i, _i, _i, _p, _l, _p]),
    "msgl_g3_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _i, _p, _l, _p]),
    "msgl_ro_gemm_max_units": (_i, [_i]),
    "msgl_ro_gemm_workspace_bytes": (_l, [_i, _i, _i]),
    "msgl_ro_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _p, _l, _p]),
}

COMM_SIGNATURES = {
    "msgl_comm_unique_id": (_i, [C.c_char_p]),
    "msgl_comm_create": (_i, [C.POINTER(_p), _i, _i, C.c_char_p, _sz]),
    "msgl_comm_all_reduce_sum": (_i, [_p, _p, _sz, _i, _p]),
    "msgl_comm_all_gather": (_i, [_p, _p, _p, _sz, _i, _p]),
    "msgl_comm_get_buffer": (_p, [_p]),
    "msgl_comm_info": (_i, [_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "msgl_comm_destroy": (_i, [_p]),
    "msgl_comm_last_error": (C.c_char_p, []),
}


GEMM_SIGNATURES = {
    "msgl_gemm_nt": (_i, [_p, _p, _p, _i, _i, _i, _l, _l, _l, _i, _p, _l, _p]),
    "msgl_gemm_tune": (
        _i,
        [_p, _p, C.POINTER(_p), _i, _i, _i, _i, _l, _l, _l, _i, _p, _l, _i, _i, _i, C.POINTER(_f), C.POINTER(_f),
         C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _p],
    ),
    "msgl_gemm_solution_name": (_i, [_i, _i, _i, _l, _l, _l, _i, C.c_char_p, _i]),
    "msgl_gemm_reset_plans": (_i, []),
    "msgl_gemm_finalists": (_i, [_i, _i, _i, _l, _l, _l, _i, C.POINTER(_f), _i]),
    "msgl_gemm_select_finalist": (_i, [_i, _i, _i, _l, _l, _l, _i, _i]),
    "msgl_gemm_get_plan": (_i, [_i, _i, _i, _l, _l, _l, _i, C.POINTER(_i), C.POINTER(_i)]),
    "msgl_gemm_set_plan": (_i, [_i, _i, _i, _l, _l, _l, _i, _i, _i, _p, _l]),
    "msgl_gemm_last_error": (C.c_char_p, []),
}


class MsglError(RuntimeError):
    """Raised when a C-ABI call returns a negative code (reference: host::PanicError)."""


def _bind(path: Path, signatures: dict) -> C.CDLL:
    if not path.exists():
        raise RuntimeError(
            f"{path} is missing: build it with `python mini-sglang_amd/build.py` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path."
        )
    dll = C.CDLL(str(path))
    for name, (res, args) in signatures.items():
        fn = getattr(dll, name, None)
        if fn is None:  # declared in the header but not exported: calling it raises
            MISSING_SYMBOLS.append(name)
            setattr(dll, name, _missing(name, path))
            continue
        fn.restype = res
        fn.argtypes = args
    return dll


MISSING_SYMBOLS: list[str] = []


def _missing(name: str, path: Path):
    def raiser(*_a, **_k):
        raise RuntimeError(f"{path.name} does not export {name}; rebuild the library")

    return raiser


_hip: C.CDLL | None = None
_comm: C.CDLL | None = None
_gemm: C.CDLL | None = None


def lib() -> C.CDLL:
    global _hip
    if _hip is None:
        _hip = _bind(HIP_SO, HIP_SIGNATURES)
        if _hip.msgl_abi_version() != ABI_VERSION:
            raise RuntimeError("libmsgl_hip.so ABI version mismatch; rebuild")
    return _hip


def comm_lib() -> C.CDLL:
    global _comm
    if _comm is None:
        _comm = _bind(COMM_SO, COMM_SIGNATURES)
    return _comm


def gemm_lib() -> C.CDLL:
    global _gemm
    if _gemm is None:
        _gemm = _bind(GEMM_SO, GEMM_SIGNATURES)
    return _gemm


def check_gemm(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = gemm_lib().msgl_gemm_last_error().decode(errors="replace")
        raise MsglError(f"{what or 'msgl gemm call'} failed ({rc}): {msg}")


def check(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = lib().msgl_last_error().decode(errors="replace")
        raise MsglError(f"{what or 'msgl call'} failed ({rc}): {msg}")


def check_comm(rc: int, what: str = "") -> None:
    if rc < 0:
        msg = comm_lib().msgl_comm_last_error().decode(errors="replace")
        raise MsglError(f"{what or 'msgl comm call'} failed ({rc}): {msg}")
