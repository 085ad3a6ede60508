"""ctypes view of oracle/_ref/libref_radix.so -- the REFERENCE's own C/src/radix.cpp compiled unmodified by
oracle/build_ref.sh (TEST INFRASTRUCTURE ONLY).  Arguments travel as DLPack `DLTensor` structs, which is what
the reference's tvm-ffi `TensorView` wraps (C/src/radix.cpp:12-24)."""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import torch

SO = Path(__file__).resolve().parent / "_ref" / "libref_radix.so"


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int), ("device_id", C.c_int)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int), ("dtype", DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


_CODES = {torch.int32: (0, 32), torch.int64: (0, 64), torch.float32: (2, 32), torch.uint8: (1, 8)}
_dll: Optional[C.CDLL] = None


def available() -> bool:
    return SO.exists()


def _lib() -> C.CDLL:
    global _dll
    if _dll is None:
        _dll = C.CDLL(str(SO))
        _dll.ref_fast_compare_key.restype = C.c_int64
        _dll.ref_fast_compare_key.argtypes = [C.POINTER(DLTensor), C.POINTER(DLTensor), C.c_char_p, C.c_size_t]
    return _dll


def _dl(t: torch.Tensor):
    code, bits = _CODES[t.dtype]
    shape = (C.c_int64 * t.dim())(*t.shape)
    strides = (C.c_int64 * t.dim())(*t.stride())
    dl = DLTensor(C.c_void_p(t.data_ptr()), DLDevice(1 if t.device.type == "cpu" else 2, 0), t.dim(),
                  DLDataType(code, bits, 1), shape, strides, 0)
    return dl, (shape, strides, t)  # keep the arrays alive


def fast_compare_key(x: torch.Tensor, y: torch.Tensor) -> int:
    """The reference's `fast_compare_key` (C/src/radix.cpp:19-40); raises RuntimeError with the reference's own
    PanicError text where the reference would."""
    a, ka = _dl(x)
    b, kb = _dl(y)
    err = C.create_string_buffer(512)
    r = _lib().ref_fast_compare_key(C.byref(a), C.byref(b), err, 512)
    if r < 0:
        raise RuntimeError(err.value.decode(errors="replace"))
    return int(r)
