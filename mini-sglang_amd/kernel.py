"""Mirror of `minisgl.kernel` (P/kernel/__init__.py): same names, argument meaning and
error behaviour, implemented on the gfx950 C-ABI instead of tvm-ffi JIT modules."""
from __future__ import annotations

import ctypes
import os
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

    def info(self) -> dict:
        """What RCCL itself reports for this communicator: ranks in it, this rank, the device it is bound to."""
        n, r, d = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _lib.check_comm(self._lib.msgl_comm_info(self._handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)),
                        "comm_info")
        return dict(nranks=n.value, rank=r.value, device=d.value)

    def destroy(self) -> None:
        if self._handle:
            self._lib.msgl_comm_destroy(self._handle)
            self._handle = ctypes.c_void_p()


def describe_p2p_error(e: int, rank: int, world_size: int) -> str:
    """The sticky error word of a peer-to-peer communicator in words (layout: csrc/comm_p2p.hip, `Error word`)."""
    phase, kind, block, peer, told = (e & 15) - 1, (e >> 4) & 15, (e >> 8) & 255, (e >> 16) & 15, bool(e & (1 << 20))
    what = {1: "one-shot all-reduce", 2: "two-shot all-reduce", 3: "fused all-reduce + add + RMSNorm", 4: "all-gather"}.get(kind, "collective")
    if told:
        return (f"peer-to-peer collective: rank {peer} gave up waiting for a peer in a {what} (barrier phase {phase}, block "
                f"{block}) and told rank {rank} of {world_size}; outputs of every later collective are NaN-poisoned")
    return (f"peer-to-peer collective: rank {rank} of {world_size} gave up waiting for a peer (rank {peer}) in a {what} "
            f"at barrier phase {phase}, block {block} (spin limit reached); outputs of that and every later collective are "
            "NaN-poisoned")


class P2PCommunicator:
    """Peer-to-peer collectives over mapped buffers (csrc/comm_p2p.hip): the symmetric-buffer path of the reference's
    NCCLWrapper (C/src/pynccl.cu:81-90, 105-123) for messages <= max_bytes.  Same surface as PyNCCLCommunicator.
    Needs only a CPU group for the one-time exchange of IPC handles, so -- unlike RCCL -- two ranks may also share
    one device (how the N > 1 path is exercised on a 1-GPU box)."""

    def __init__(self, rank: int, world_size: int, cpu_group, max_bytes: int, *, one_shot_max_bytes: int = 256 << 10,
                 blocks: int = 32) -> None:
        import torch.distributed as dist

        self._lib = _lib.lib()
        self._handle = ctypes.c_void_p()
        self.rank, self.world_size, self.max_bytes = rank, world_size, int(max_bytes)
        # every step that can fail on ONE rank (allocation, handle export, mapping a peer's buffer) is followed by an
        # exchange over the CPU group, so that all ranks learn of it and raise together instead of one raising while
        # the others wait in a collective
        mine, err = None, None
        try:
            _lib.check(self._lib.msgl_p2p_create(ctypes.byref(self._handle), rank, world_size, self.max_bytes), "p2p_create")
            buf = ctypes.create_string_buffer(_lib.IPC_HANDLE_BYTES)
            _lib.check(self._lib.msgl_p2p_ipc_handle(self._handle, buf), "p2p_ipc_handle")
            mine = buf.raw
        except Exception as e:  # noqa: BLE001 - reported to every rank below
            err = f"rank {rank}: {e}"
        handles = [None] * world_size
        dist.all_gather_object(handles, (mine, err), group=cpu_group)
        self._raise_if_any([h[1] for h in handles])
        try:
            _lib.check(self._lib.msgl_p2p_open(self._handle, b"".join(h[0] for h in handles)), "p2p_open")
            _lib.check(self._lib.msgl_p2p_configure(self._handle, one_shot_max_bytes, blocks), "p2p_configure")
            if os.environ.get("MSGL_P2P_SPIN_LIMIT"):  # diagnostics: fail fast instead of after tens of seconds
                _lib.check(self._lib.msgl_p2p_set_spin_limit(self._handle, int(os.environ["MSGL_P2P_SPIN_LIMIT"])), "p2p_set_spin_limit")
        except Exception as e:  # noqa: BLE001
            err = f"rank {rank}: {e}"
        errs = [None] * world_size
        dist.all_gather_object(errs, err, group=cpu_group)  # also: every rank has mapped every buffer before the first collective
        self._raise_if_any(errs)

    def _raise_if_any(self, errs) -> None:
        bad = [e for e in errs if e]
        if bad:
            self.destroy()
            raise RuntimeError("peer-to-peer communicator setup failed: " + "; ".join(bad))

    def self_test(self, cpu_group) -> Optional[str]:
        """Known answers (rank-valued data -> n(n+1)/2, tests/kernel/test_comm.py:106-114 of the reference) through the
        one-shot and the two-shot kernel and the all-gather, on this communicator's real links.  Returns None if every
        rank saw the right values and no barrier timed out, else a description -- the same on all ranks."""
        import torch.distributed as dist

        dev = torch.device("cuda", torch.cuda.current_device())
        problems = []
        try:
            for name, numel in (("one-shot", 4096), ("two-shot", max(8192, (self.max_bytes // 2) // 16 * 8))):
                if numel * 2 > self.max_bytes:
                    continue
                x = torch.full((numel,), float(self.rank + 1), dtype=torch.bfloat16, device=dev)
                self.all_reduce(x)
                torch.cuda.synchronize()
                want = float(self.world_size * (self.world_size + 1) // 2)
                if not bool((x == want).all()):
                    problems.append(f"{name} all-reduce: got {x[:2].tolist()} .. {x[-2:].tolist()}, want {want}")
            n = min(4096, self.max_bytes // 2 // self.world_size // 8 * 8)
            if n > 0:
                src = torch.full((n,), float(self.rank), dtype=torch.bfloat16, device=dev)
                dst = torch.empty((n * self.world_size,), dtype=torch.bfloat16, device=dev)
                self.all_gather(dst, src)
                torch.cuda.synchronize()
                want = torch.arange(self.world_size, device=dev, dtype=torch.bfloat16).repeat_interleave(n)
                if not torch.equal(dst, want):
                    problems.append("all-gather: wrong contents")
            if self.error():
                problems.append(self.describe_error(self.error()))
        except Exception as e:  # noqa: BLE001
            problems.append(f"{type(e).__name__}: {e}")
        mine = f"rank {self.rank}: " + "; ".join(problems) if problems else None
        everyone = [None] * self.world_size
        dist.all_gather_object(everyone, mine, group=cpu_group)
        bad = [e for e in everyone if e]
        return "; ".join(bad) if bad else None

    def fits(self, t: torch.Tensor) -> bool:
        n = t.numel() * t.element_size()
        return n <= self.max_bytes and n % 16 == 0 and t.data_ptr() % 16 == 0

    def all_reduce(self, input: torch.Tensor, op: Literal["sum"] = "sum") -> None:
        if op != "sum":
            raise ValueError(f"unsupported reduce op {op!r}")
        if not (input.is_cuda and input.is_contiguous()):
            raise RuntimeError("Tensor must be a contiguous device tensor")
        _lib.check(self._lib.msgl_p2p_all_reduce_sum(self._handle, input.data_ptr(), input.numel(), ops._dt(input),
                                                      torch.cuda.current_stream().cuda_stream), "p2p_all_reduce")

    def all_gather(self, output: torch.Tensor, input: torch.Tensor) -> None:
        if not (input.is_cuda and input.is_contiguous() and output.is_cuda and output.is_contiguous()):
            raise RuntimeError("Tensor must be a contiguous device tensor")
        if output.shape[0] != input.shape[0] * self.world_size:
            raise RuntimeError("Destination tensor has incorrect size")
        _lib.check(self._lib.msgl_p2p_all_gather(self._handle, output.data_ptr(), input.data_ptr(), input.numel(),
                                                  ops._dt(input), torch.cuda.current_stream().cuda_stream),
                   "p2p_all_gather")

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> bool:
        """x <- rmsnorm(all_reduce(x) + residual) * weight, residual <- the rounded sum, in ONE launch (the row-parallel
        projection's all-reduce fused with the RMSNormFused that follows it, P/layers/linear.py:102-106 + P/layers/norm.py:
        33-38).  Returns False -- nothing launched -- for shapes the kernel does not cover (the caller then issues
        all_reduce + fused_add_rmsnorm); bit-identical to that pair."""
        rows, dim = x.shape
        if (x.dim() != 2 or x.stride(1) != 1 or residual.stride(1) != 1 or residual.shape != x.shape or dim % 8 or dim > 8192
                or rows < 1 or rows > 512 or rows * dim * x.element_size() > self.max_bytes or x.data_ptr() % 16
                or residual.data_ptr() % 16):
            return False
        _lib.check(self._lib.msgl_p2p_all_reduce_add_rmsnorm(self._handle, x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                                             float(eps), rows, dim, x.stride(0), residual.stride(0), ops._dt(x),
                                                             torch.cuda.current_stream().cuda_stream),
                   "p2p_all_reduce_add_rmsnorm")
        return True

    def error(self) -> int:
        """0, or the sticky error word of a barrier that timed out (describe_error decodes it; synchronises the device)."""
        return int(self._lib.msgl_p2p_error(self._handle))

    def set_spin_limit(self, spins: int) -> None:
        """Flag polls per barrier before it gives up (default 40 M ~ tens of seconds)."""
        _lib.check(self._lib.msgl_p2p_set_spin_limit(self._handle, int(spins)), "p2p_set_spin_limit")

    def poll_error(self, sync: bool = False) -> None:
        """Raise if a barrier of this communicator ever timed out (its kernels then returned NaN-poisoned outputs, never
        a partial sum).  sync=False costs one 4-byte async copy: it enqueues a read of the device's error word on the
        current stream and examines the value the PREVIOUS poll fetched (complete by now in any loop that synchronises
        once per step, as the token copy of a decode step does); sync=True reads the word now (device synchronise)."""
        if not self._handle:
            return
        if sync:
            e = self.error()
        else:
            if getattr(self, "_err_host", None) is None:
                self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._err_event = None
            e = 0
            if self._err_event is not None and self._err_event.query():
                e = int(self._err_host[0])
                self._err_event = None
            if e == 0 and self._err_event is None:
                _lib.check(self._lib.msgl_p2p_error_async(self._handle, self._err_host.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), "p2p_error_async")
                self._err_event = torch.cuda.Event()
                self._err_event.record()
        if e:
            raise _lib.MsglError(self.describe_error(e))

    def describe_error(self, e: int) -> str:
        return describe_p2p_error(e, self.rank, self.world_size)

    def get_buffer(self) -> int:
        return int(self._lib.msgl_p2p_get_buffer(self._handle) or 0)

    def destroy(self) -> None:
        if self._handle:
            self._lib.msgl_p2p_destroy(self._handle)
            self._handle = ctypes.c_void_p()


_LIVE_COMMUNICATORS: set = set()  # communicators init_pynccl handed out and nobody destroyed yet


def poll_communicator_errors(sync: bool = False) -> None:
    """poll_error() of every live communicator: the per-step hook of hosts that own no communicator object themselves
    (the attention backend calls it every few prepare_metadata calls when driven by the reference's scheduler, whose
    captured decode graphs never pass through Python's all_reduce)."""
    for c in list(_LIVE_COMMUNICATORS):
        c.poll_error(sync)


class HybridCommunicator:
    """What `init_pynccl` returns: messages that fit the mapped buffers go peer to peer (the reference's symmetric
    window path), the rest through RCCL (its direct path, C/src/pynccl.cu:125-132)."""

    def __init__(self, p2p: Optional[P2PCommunicator], rccl: Optional[RcclCommunicator]) -> None:
        assert p2p is not None or rccl is not None
        self.p2p, self.rccl = p2p, rccl
        self.rank = (p2p or rccl).rank
        self.world_size = (p2p or rccl).world_size
        # a second, independent communicator for collectives issued on a side stream while this one is busy on the
        # compute stream (minisgl_plugin's row-parallel overlap); None unless init_pynccl(side=True) made one
        self.side: Optional["HybridCommunicator"] = None
        # collectives ISSUED from Python per path (a launch captured into a hipGraph is counted once, at capture, not
        # per replay): bench.py reports the split next to `rccl_ranks_seen`
        self.issued = dict(p2p_calls=0, p2p_bytes=0, rccl_calls=0, rccl_bytes=0)
        _LIVE_COMMUNICATORS.add(self)

    def _count(self, path: str, t: torch.Tensor) -> None:
        self.issued[path + "_calls"] += 1
        self.issued[path + "_bytes"] += t.numel() * t.element_size()

    def all_reduce(self, input: torch.Tensor, op: Literal["sum"] = "sum") -> None:
        if self.p2p is not None and (self.rccl is None or self.p2p.fits(input)):
            self._count("p2p", input)
            return self.p2p.all_reduce(input, op)
        self._count("rccl", input)
        return self.rccl.all_reduce(input, op)

    def all_gather(self, output: torch.Tensor, input: torch.Tensor) -> None:
        if self.p2p is not None and (self.rccl is None or self.p2p.fits(input)):
            self._count("p2p", input)
            return self.p2p.all_gather(output, input)
        self._count("rccl", input)
        return self.rccl.all_gather(output, input)

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
        """all_reduce(x) followed by fused_add_rmsnorm(x, residual, weight, eps): one peer-to-peer launch where that
        kernel applies (decode-size row blocks), else the two operations."""
        if self.p2p is not None and self.p2p.all_reduce_add_rmsnorm(x, residual, weight, eps):
            self._count("p2p", x)
            return
        self.all_reduce(x, "sum")
        ops.fused_add_rmsnorm(x, residual, weight, eps)

    def get_buffer(self) -> int:
        return (self.p2p or self.rccl).get_buffer()

    def poll_error(self, sync: bool = False) -> None:
        """Raise if a peer-to-peer barrier timed out (see P2PCommunicator.poll_error); RCCL reports through its calls."""
        if self.p2p is not None:
            self.p2p.poll_error(sync)
        if self.side is not None:
            self.side.poll_error(sync)

    def destroy(self) -> None:
        try:
            self.poll_error(sync=True)  # a timed-out barrier must not pass unnoticed because nobody polled
        finally:
            for c in (self.p2p, self.rccl):
                if c is not None:
                    c.destroy()
            _LIVE_COMMUNICATORS.discard(self)
            if self.side is not None:
                side, self.side = self.side, None
                side.destroy()


PyNCCLCommunicator = HybridCommunicator


def create_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check_comm(_lib.comm_lib().msgl_comm_unique_id(buf), "comm_unique_id")
    return buf.raw


def init_pynccl(*, tp_rank: int, tp_size: int, tp_cpu_group, max_size_bytes: int = 0,
                backend: Optional[str] = None, side: Optional[bool] = None) -> HybridCommunicator:
    """P/kernel/pynccl.py:47-78.  backend: "hybrid" (peer-to-peer buffers of max_size_bytes + RCCL for larger
    messages), "rccl" (library only), "p2p" (mapped buffers only: every message must fit; the only choice when two
    ranks share a device).  RCCL bootstrap as in the reference: rank 0 creates the unique id, broadcast over the CPU
    (gloo) group."""
    import os

    import torch.distributed as dist

    # the reference calls init_pynccl(tp_rank, tp_size, tp_cpu_group, max_size_bytes) (P/distributed/impl.py:81-88): the
    # backend of a drop-in run is chosen by the environment
    backend = backend or os.environ.get("MSGL_COMM_BACKEND", "hybrid")
    if backend not in ("hybrid", "rccl", "p2p"):
        raise ValueError(backend)
    rccl = p2p = None
    if backend != "p2p":
        id_list = [create_unique_id() if tp_rank == 0 else None]
        dist.broadcast_object_list(id_list, src=0, group=tp_cpu_group)
        uid = id_list[0]
        assert uid is not None, f"Failed to get RCCL unique ID on {tp_rank = }"
        rccl = RcclCommunicator(tp_rank, tp_size, 0, uid)
    if backend != "rccl" and max_size_bytes > 0 and tp_size <= 8:
        # the mapped-buffer path is checked on the links it will run on before anything depends on it; with RCCL at
        # hand a failure (setup or known answers) is reported and the library path carries all messages
        problem = None
        try:
            p2p = P2PCommunicator(tp_rank, tp_size, tp_cpu_group, max_size_bytes)
            problem = p2p.self_test(tp_cpu_group)
        except RuntimeError as e:
            problem = str(e)
        if problem is not None:
            if p2p is not None:
                p2p.destroy()
                p2p = None
            if rccl is None:
                raise RuntimeError(f"peer-to-peer collectives unusable and no RCCL communicator to fall back to: {problem}")
            if tp_rank == 0:
                import sys

                print(f"[msgl] peer-to-peer collectives disabled, RCCL carries every message: {problem}", file=sys.stderr)
    comm = HybridCommunicator(p2p, rccl)
    # side=True (or MSGL_COMM_OVERLAP=1 when the reference calls us): a second communicator of the same kind for the
    # side stream -- the same sequence of CPU-group exchanges on every rank, so all ranks must ask for it alike
    if side is None:
        side = os.environ.get("MSGL_COMM_OVERLAP", "0") == "1"
    if side and tp_size > 1:
        comm.side = init_pynccl(tp_rank=tp_rank, tp_size=tp_size, tp_cpu_group=tp_cpu_group, max_size_bytes=max_size_bytes,
                                backend=backend, side=False)
    return comm


__all__ = ["indexing", "fast_compare_key", "store_cache", "init_pynccl", "PyNCCLCommunicator", "RcclCommunicator",
           "P2PCommunicator", "HybridCommunicator", "poll_communicator_errors"]
