"""Registers the MI355X core into an installed `minisgl` (the reference), without touching its
source: after `install()` the reference's scheduler, radix cache, engine, models and server run
unchanged with `--attn hip` (see INTEGRATION.md for the two-line bootstrap).

Seams filled:
  attention backend   SUPPORTED_ATTENTION_BACKENDS.register("hip")        P/attention/__init__.py:19-40
  minisgl.kernel      store_cache / indexing / fast_compare_key / init_pynccl   P/kernel/__init__.py
  flashinfer names    rmsnorm, fused_add_rmsnorm, apply_rope_with_cos_sin_cache_inplace,
                      silu_and_mul, gelu_and_mul, sampling.*              P/layers/norm.py:10-30 ...

and, so that the path the reference drives is the path bench.py measures:
  F.linear            the name `F` of P/layers/linear.py:32,103,124 and P/layers/embedding.py:98 resolves to a
                      proxy of torch.nn.functional whose `linear` is ops.linear (bias / CPU / odd dtypes fall
                      back to torch)
  AttentionLayer      P/layers/attention.py:47-57 (split, q-norm, k-norm, RoPE, store_kv, attend: 5 launches)
                      becomes one qk_norm_rope_store launch + attend, bit-identical to the unfused sequence
  GEMM plans          before GraphRunner captures (P/engine/graph.py:105-147) every projection shape of every
                      captured batch size is timed once (gemm_plan.tune_projection_gemms)
All three are class / module attribute assignments made at install time; no reference file is edited.
"""
from __future__ import annotations

import weakref

import importlib
import os
import sys
import types
from typing import Any, Dict, List, Optional

_STATE: Dict[str, Any] = {"gemm_tune": "heuristic", "gemm_report": [], "fast_linear": False, "fused_attention": False,
                          "deferred_reduce_weights": set()}


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


# ------------------------------------------------------------------------------ F.linear
class _FunctionalProxy:
    """Stands in for the module-level name `F` (= torch.nn.functional) of the reference's layer files: every
    attribute is torch's except `linear`."""

    def __init__(self) -> None:
        import torch.nn.functional as F

        self._F = F

    def __getattr__(self, name: str) -> Any:
        return getattr(self._F, name)

    def linear(self, x, weight, bias=None):
        import torch

        from . import ops

        if (bias is None and x.is_cuda and x.dtype == weight.dtype and x.dtype in (torch.bfloat16, torch.float16)
                and weight.dim() == 2 and x.dim() >= 1 and x.shape[-1] == weight.shape[1] and weight.stride(1) == 1):
            x2 = x if x.dim() == 2 else x.reshape(-1, x.shape[-1])
            if x2.stride(1) == 1:
                if x.dim() == 2 and weight.data_ptr() in _STATE["deferred_reduce_weights"]:
                    # o_proj / down_proj (tp = 1) or qkv_proj of a dense decoder layer: the tensor returned here is
                    # the very object the reference hands to fused_add_rmsnorm (RMSNormFused.forward) resp. to
                    # AttentionLayer.forward, which then does the split-K reduce
                    y, slabs = ops.linear_slabs(x2, weight)
                    if slabs is not None:
                        y._msgl_slabs = slabs
                    return y
                y = ops.linear(x2, weight)
                return y if x.dim() == 2 else y.view(*x.shape[:-1], weight.shape[0])
        return self._F.linear(x, weight, bias)


def _install_fast_linear() -> None:
    proxy = _FunctionalProxy()
    for mod in ("minisgl.layers.linear", "minisgl.layers.embedding"):
        importlib.import_module(mod).F = proxy
    _STATE["fast_linear"] = True


# ------------------------------------------------------------------------------ fused attention layer
def _install_fused_attention() -> None:
    from minisgl.core import get_global_ctx
    from minisgl.layers.attention import AttentionLayer

    from . import ops
    from .attention import HipAttnBackend

    if getattr(AttentionLayer.forward, "_msgl_fused", False):
        return
    reference_forward = AttentionLayer.forward

    def forward(self, qkv):
        ctx = get_global_ctx()
        backend = ctx.attn_backend
        if not isinstance(backend, HipAttnBackend) or not qkv.is_cuda:
            if getattr(qkv, "_msgl_slabs", None) is not None:
                raise RuntimeError("qkv_proj deferred its reduce to the fused attention pass, but the backend is not 'hip'")
            return reference_forward(self, qkv)
        batch = ctx.batch
        D = self.head_dim
        q, k, v = qkv.split([self.qo_attn_dim, self.kv_attn_dim, self.kv_attn_dim], dim=-1)
        kv = ctx.kv_cache
        kc, vc = kv.k_cache(self.layer_id), kv.v_cache(self.layer_id)
        qn, kn = self.q_norm, self.k_norm
        slabs = getattr(qkv, "_msgl_slabs", None)  # qkv_proj's F.linear left its split-K reduce to this pass
        if slabs is not None:
            del qkv._msgl_slabs
            ops.qk_norm_rope_store_slabs(qkv, slabs, self.num_qo_heads, self.num_kv_heads,
                                         qn.weight if qn is not None else None, kn.weight if kn is not None else None,
                                         qn.eps if qn is not None else 0.0, batch.positions, self.rotary._cos_sin_cache,
                                         kc.view(-1, self.kv_attn_dim), vc.view(-1, self.kv_attn_dim), batch.out_loc, D)
        else:
            ops.qk_norm_rope_store(q, k, v, qn.weight if qn is not None else None, kn.weight if kn is not None else None,
                                   qn.eps if qn is not None else 0.0, batch.positions, self.rotary._cos_sin_cache,
                                   kc.view(-1, self.kv_attn_dim), vc.view(-1, self.kv_attn_dim), batch.out_loc, D)
        o = backend.attend(q.view(-1, self.num_qo_heads, D), self.layer_id, batch)
        return o.view(-1, self.qo_attn_dim)

    forward._msgl_fused = True  # type: ignore[attr-defined]
    forward._msgl_reference = reference_forward  # type: ignore[attr-defined]
    AttentionLayer.forward = forward
    _STATE["fused_attention"] = True


# ------------------------------------------------------------------------------ row-parallel projections: side-stream overlap
def _install_row_parallel_overlap() -> None:
    """`LinearOProj.forward` / `LinearRowParallel.forward` (P/layers/linear.py:102-106, 123-127) are
    `y = F.linear(x, w, b); y = self._comm.all_reduce(y)` on one stream.  north_star: "RCCL all-reduce over xGMI
    overlapped on a side HIP stream".  With at least MSGL_COMM_SPLIT_TOKENS tokens (default 2048: prefill chunks; a decode
    batch would stream the weights twice for nothing) and a second communicator at hand (init_pynccl(side=True) /
    MSGL_COMM_OVERLAP=1 -- two collectives of ONE communicator must never be in flight at once), the projection runs per
    token half: the first half's all-reduce goes to the side stream through the second communicator while the compute
    stream runs the second half's GEMM; the second half's all-reduce follows on the compute stream, which then waits for
    the side stream.  The same split rule and kernels as model.DenseDecoder.row_parallel (bit-identical to it and to the
    serial issue of the same kernels).  Anything else (bias, tp = 1, small batches, capture) takes the reference's
    forward unchanged."""
    import torch

    from minisgl.layers.linear import LinearOProj, LinearRowParallel

    split_tokens = int(os.environ.get("MSGL_COMM_SPLIT_TOKENS", "2048"))

    def side_comm():
        import minisgl.distributed.impl as dimpl

        comm = getattr(dimpl.DistributedCommunicator.plugins[-1], "comm", None)
        return comm, getattr(comm, "side", None)

    def wrap(cls):
        if getattr(cls.forward, "_msgl_overlap", False):
            return
        reference_forward = cls.forward

        def forward(self, x):
            T = x.shape[0] if x.dim() == 2 else 0
            if (self._tp_size > 1 and self.bias is None and 0 < T < max(split_tokens, 1) and x.is_cuda
                    and self.weight.data_ptr() in _STATE.get("deferred_allreduce_weights", ())):
                # decode-size batch of a dense decoder layer: the tensor returned here goes, untouched, into the next
                # RMSNormFused (fused_add_rmsnorm shim), which runs all-reduce + residual add + norm as ONE launch
                comm, _side = side_comm()
                if comm is not None and hasattr(comm, "all_reduce_add_rmsnorm") and x.dtype == self.weight.dtype:
                    from . import ops

                    y = ops.linear(x, self.weight)
                    y._msgl_allreduce = comm
                    ops._PENDING_ALLREDUCE[x.device.index or 0] = y
                    return y
            if (self._tp_size == 1 or self.bias is not None or split_tokens <= 0 or T < split_tokens or not x.is_cuda
                    or torch.cuda.is_current_stream_capturing()):
                return reference_forward(self, x)
            comm, side = side_comm()
            if comm is None or side is None:
                return reference_forward(self, x)
            from . import ops

            h = (T // 2 + 7) // 8 * 8
            y = torch.empty((T, self.weight.shape[0]), dtype=x.dtype, device=x.device)
            ops.linear(x[:h], self.weight, out=y[:h])
            main = torch.cuda.current_stream()
            stream = _STATE.get("side_stream")
            if stream is None or stream.device != x.device:
                stream = _STATE["side_stream"] = torch.cuda.Stream(device=x.device)
            stream.wait_event(main.record_event())
            with torch.cuda.stream(stream):
                side.all_reduce(y[:h], "sum")
                done = stream.record_event()
            ops.linear(x[h:], self.weight, out=y[h:])
            comm.all_reduce(y[h:], "sum")
            main.wait_event(done)
            _STATE["overlapped_projections"] = _STATE.get("overlapped_projections", 0) + 1
            return y

        forward._msgl_overlap = True  # type: ignore[attr-defined]
        forward._msgl_reference = reference_forward  # type: ignore[attr-defined]
        cls.forward = forward

    wrap(LinearOProj)
    wrap(LinearRowParallel)
    _STATE["row_parallel_overlap"] = True


# ------------------------------------------------------------------------------ fused gate_up projection + SiLU.mul
def _install_fused_gated_mlp() -> None:
    """`GatedMLP.forward` (P/models/utils.py:45-51) = gate_up_proj (F.linear, P/layers/linear.py:32) -> act_fn
    (flashinfer.silu_and_mul, P/layers/activation.py:9-12) -> down_proj.  For layers whose gate_up weight
    `_interleave_gated_mlps` has put into the block-32 interleaved row order, the first two run as
    ops.linear_silu: one launch of csrc/gemm_g3.hip with the activation in its epilogue where the pre-capture search
    planned it, else the projection followed by the interleaved activation kernel.  Every other layer: unchanged."""
    from minisgl.models.utils import GatedMLP

    from . import ops

    if getattr(GatedMLP.forward, "_msgl_fused", False):
        return
    reference_forward = GatedMLP.forward

    def forward(self, x):
        if not getattr(self, "_msgl_gate_up_ilv", False):
            return reference_forward(self, x)
        # the weight rows ARE permuted: the reference forward (contiguous [gate | up] halves) would be silently wrong on
        # them, so every input either takes the interleaved path or is refused
        w = self.gate_up_proj.weight
        if not _holds_permuted_rows(self, w):
            # load_state_dict (P/layers/base.py:31-49) REPLACES the tensor: the new one holds the reference's [gate; up]
            # rows, the permuted storage is gone -- this layer is a plain reference layer again until the next capture
            # converts it (an IN-PLACE rewrite cannot be seen from here: restore_gate_up_layout() first, see its docstring)
            self._msgl_gate_up_ilv, self._msgl_gate_up_ptr, self._msgl_gate_up_ref = False, None, None
            return reference_forward(self, x)
        if not x.is_cuda:
            raise RuntimeError("GatedMLP with interleaved gate_up rows runs on the HIP device only (no CPU fallback)")
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y = self.down_proj.forward(ops.linear_silu(x2, w))
        return y if x.dim() == 2 else y.view(*lead, y.shape[-1])

    forward._msgl_fused = True  # type: ignore[attr-defined]
    forward._msgl_reference = reference_forward  # type: ignore[attr-defined]
    GatedMLP.forward = forward
    _STATE["fused_gated_mlp"] = True


def _holds_permuted_rows(op: Any, w: Any) -> bool:
    """True when `w` (the layer's current gate_up weight) is the very tensor `_interleave_gated_mlps` permuted.  The
    address alone is not enough: after load_state_dict replaces the tensor the caching allocator may hand the NEW tensor the
    freed block at the same address, so the mark is (address, a weak reference to the tensor object): a replaced tensor is a
    different object (or the weak reference is dead) whatever its address."""
    ref = getattr(op, "_msgl_gate_up_ref", None)
    return ref is not None and ref() is w and w.data_ptr() == getattr(op, "_msgl_gate_up_ptr", None)


def _interleave_gated_mlps(model: Any) -> int:
    """Once per model, after its weights are loaded and before its first forward (GraphRunner.__init__ ->
    _capture_graphs, P/engine/graph.py:77-100, runs exactly there): permute the rows of every SiLU GatedMLP's
    gate_up_proj.weight ([gate shard; up shard], P/layers/linear.py:50-62) in place into ops.interleave_gate_up order
    and mark the layer.  Returns the number of layers converted."""
    import torch

    from minisgl.layers.base import BaseOP
    from minisgl.layers.linear import LinearColParallelMerged
    from minisgl.models.utils import GatedMLP

    from . import ops

    done = 0

    def walk(op: Any) -> None:
        nonlocal done
        if type(op) is GatedMLP:
            gu = getattr(op, "gate_up_proj", None)
            act = getattr(getattr(op, "act_fn", None), "__name__", "")
            if (type(gu) is LinearColParallelMerged and gu.bias is None and act == "silu_and_mul" and gu.weight.is_cuda
                    and gu.weight.dim() == 2 and (gu.weight.shape[0] // 2) % 64 == 0 and gu.weight.is_contiguous()
                    and gu.weight.dtype in (torch.bfloat16, torch.float16)
                    and not getattr(op, "_msgl_gate_up_ilv", False)):
                w = gu.weight.data if hasattr(gu.weight, "data") else gu.weight
                w.copy_(ops.interleave_gate_up(w))
                op._msgl_gate_up_ilv = True
                op._msgl_gate_up_ptr = gu.weight.data_ptr()  # the storage that holds permuted rows (checked per forward)
                op._msgl_gate_up_ref = weakref.ref(gu.weight)  # ... and the tensor object itself (_holds_permuted_rows)
                done += 1
            return
        if isinstance(op, BaseOP):
            for sub in vars(op).values():
                for s_ in (sub if isinstance(sub, (list, tuple)) else (sub,)):
                    if isinstance(s_, BaseOP):
                        walk(s_)

    walk(model)
    return done


def gate_up_reference(op: Any) -> Any:
    """gate_up_proj.weight of a GatedMLP in the REFERENCE's row order [gate; up] (P/layers/linear.py:50-62), whatever
    order the storage is in: what a state_dict export or a comparison against a checkpoint must read."""
    import torch

    from . import ops

    w = op.gate_up_proj.weight
    if not getattr(op, "_msgl_gate_up_ilv", False):
        return w
    idx = ops.gate_up_interleave_index(w.shape[0] // 2, w.device)
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(idx.numel(), device=w.device)
    return w.index_select(0, inv)


def restore_gate_up_layout(model: Any) -> int:
    """Undo `_interleave_gated_mlps` in place (rows back to [gate; up], flag cleared) on every converted GatedMLP:
    call before anything rewrites or exports the weights (load_state_dict, a reload, state_dict); the next graph
    capture converts them again.  Returns the number of layers restored."""
    from minisgl.layers.base import BaseOP

    done = 0

    def walk(op: Any) -> None:
        nonlocal done
        if getattr(op, "_msgl_gate_up_ilv", False):
            w = op.gate_up_proj.weight
            if _holds_permuted_rows(op, w):  # else: already replaced by un-permuted rows
                (w.data if hasattr(w, "data") else w).copy_(gate_up_reference(op))
            op._msgl_gate_up_ilv = False
            op._msgl_gate_up_ptr = op._msgl_gate_up_ref = None
            done += 1
            return
        if isinstance(op, BaseOP):
            for sub in vars(op).values():
                for s_ in (sub if isinstance(sub, (list, tuple)) else (sub,)):
                    if isinstance(s_, BaseOP):
                        walk(s_)

    walk(model)
    return done


def _gate_up_is_interleaved(model: Any) -> set:
    """data_ptr of the gate_up weights that are in interleaved order (their tuning group times the fused launch)."""
    from minisgl.layers.base import BaseOP

    ptrs: set = set()

    def walk(op: Any) -> None:
        if getattr(op, "_msgl_gate_up_ilv", False):
            ptrs.add(op.gate_up_proj.weight.data_ptr())
            return
        if isinstance(op, BaseOP):
            for sub in vars(op).values():
                for s_ in (sub if isinstance(sub, (list, tuple)) else (sub,)):
                    if isinstance(s_, BaseOP):
                        walk(s_)

    walk(model)
    return ptrs


# ------------------------------------------------------------------------------ GEMM plans before capture
def _projection_groups(model: Any, require_device: bool = True) -> List[tuple]:
    """(name, same-shaped weights of up to 8 layers, K) for every distinct linear shape in the reference's op tree
    (P/layers/base.py:15-53: ops hold sub-ops as attributes, OPList in `op_list`) + the LM head."""
    from minisgl.layers.base import BaseOP
    from minisgl.layers.embedding import ParallelLMHead
    from minisgl.layers.linear import _LinearTPImpl

    found: Dict[tuple, List] = {}
    names: Dict[tuple, str] = {}

    def walk(op: Any, name: str) -> None:
        if isinstance(op, _LinearTPImpl):
            if op.bias is None and (op.weight.is_cuda or not require_device):
                key = tuple(op.weight.shape)
                found.setdefault(key, []).append(op.weight)
                names.setdefault(key, name)
            return
        if isinstance(op, ParallelLMHead):
            w = (op.tied_embedding or op).weight
            if op.bias is None and (w.is_cuda or not require_device):
                found.setdefault(tuple(w.shape), []).append(w)
                names.setdefault(tuple(w.shape), "lm_head")
            return
        if isinstance(op, BaseOP):
            for attr, sub in vars(op).items():
                if isinstance(sub, BaseOP):
                    walk(sub, attr)
                elif isinstance(sub, (list, tuple)):
                    for s in sub:
                        if isinstance(s, BaseOP):
                            walk(s, attr)

    walk(model, "model")
    groups = []
    for shape, ws in found.items():
        step = max(1, len(ws) // 8)
        groups.append((names[shape].replace("_proj", ""), ws[::step][:8], shape[1]))
    return groups


def _deferred_reduce_weights(model: Any, tp_gt1: bool = False) -> set:
    """data_ptr of every projection weight whose output goes, untouched, into a kernel of ours that can add the
    projection's k-slice sums itself: o_proj and down_proj of the dense decoder layers into the next RMSNormFused's
    fused residual add (P/models/qwen3.py:37-41, llama.py:39-43, qwen2.py; the last layer's down_proj feeds the final
    norm, qwen3.py:63) when no all-reduce sits in between (tp = 1), and qkv_proj into AttentionLayer.forward
    (P/models/utils.py:118-123).  Recognised structurally; anything that does not look exactly like that layer is left
    alone."""
    from minisgl.layers.base import BaseOP
    from minisgl.core import get_global_ctx
    from minisgl.layers.linear import LinearOProj, LinearQKVMerged, LinearRowParallel
    from minisgl.layers.norm import RMSNormFused

    from .attention import HipAttnBackend

    try:
        hip_backend = isinstance(get_global_ctx().attn_backend, HipAttnBackend)
    except Exception:
        hip_backend = False
    from minisgl.models.utils import GatedMLP, RopeAttn

    ptrs: set = set()

    def walk(op: Any) -> None:
        attn, mlp = getattr(op, "self_attn", None), getattr(op, "mlp", None)
        if (type(attn) is RopeAttn and type(mlp) is GatedMLP
                and type(getattr(op, "post_attention_layernorm", None)) is RMSNormFused
                and type(getattr(op, "input_layernorm", None)) is RMSNormFused):
            o, down = getattr(attn, "o_proj", None), getattr(mlp, "down_proj", None)
            if (type(o) is LinearOProj and type(down) is LinearRowParallel and o.bias is None and down.bias is None
                    and (o._tp_size > 1 and down._tp_size > 1 if tp_gt1 else o._tp_size == 1 and down._tp_size == 1)
                    and o.weight.is_cuda):
                ptrs.update((o.weight.data_ptr(), down.weight.data_ptr()))
            if tp_gt1:  # only the row-parallel pair: qkv_proj has no all-reduce
                return
            # qkv_proj -> AttentionLayer.forward (P/models/utils.py:118-123), column-parallel: any tp size; only when
            # that forward is the fused one installed above and the backend is ours
            qkv = getattr(attn, "qkv_proj", None)
            if (type(qkv) is LinearQKVMerged and qkv.bias is None and qkv.weight.is_cuda and _STATE["fused_attention"]
                    and hip_backend):
                ptrs.add(qkv.weight.data_ptr())
            return
        if isinstance(op, BaseOP):
            for sub in vars(op).values():
                for s in (sub if isinstance(sub, (list, tuple)) else (sub,)):
                    if isinstance(s, BaseOP):
                        walk(s)

    walk(model)

    # the LAST layer's down_proj feeds the model's final norm (P/models/qwen3.py:62-63: `self.norm.forward(x, residual)`):
    # deferred only if that norm is the fused-add kind whose shim adds the slabs; anything else reads `x` as a tensor
    def final_norms(op: Any):
        layers, norm = getattr(op, "layers", None), getattr(op, "norm", None)
        if layers is not None and norm is not None and getattr(layers, "op_list", None):
            yield layers.op_list[-1], norm
        if isinstance(op, BaseOP):
            for sub in vars(op).values():
                for s_ in (sub if isinstance(sub, (list, tuple)) else (sub,)):
                    if isinstance(s_, BaseOP):
                        yield from final_norms(s_)

    for last, norm in final_norms(model):
        down = getattr(getattr(last, "mlp", None), "down_proj", None)
        if down is not None and type(norm) is not RMSNormFused:
            ptrs.discard(down.weight.data_ptr())
    return ptrs


def _refine_reference_graphs(gr: Any, model: Any) -> List[dict]:
    """plan_refine.refine_plans_in_graph on the reference's GraphRunner: the finalists of the pre-capture search are
    re-ranked inside the reference's own captured decode graph of the largest batch size.  `capture` below restates
    the body of the reference's capture loop for ONE size (P/engine/graph.py:130-145) on the reference's objects."""
    import torch

    from minisgl.core import Batch, Req, get_global_ctx

    from .plan_refine import refine_plans_in_graph

    ctx, bs = get_global_ctx(), gr.max_graph_bs
    if bs not in gr.graph_map:
        return []

    def capture() -> None:
        pool = gr.graph_map[bs].pool()
        graph = torch.cuda.CUDAGraph()
        batch = Batch(reqs=[gr.dummy_req] * bs, phase="decode")
        batch.padded_reqs = batch.reqs
        gr.attn_backend.prepare_for_capture(batch)
        gr.buffer.set_batch(batch)
        with ctx.forward_batch(batch):
            gr.buffer.logits[:bs] = model.forward()
            with torch.cuda.graph(graph, pool=pool, stream=gr.stream):
                gr.buffer.logits[:bs] = model.forward()
        gr.graph_map[bs] = graph

    k0 = ctx.kv_cache.k_cache(0)
    return refine_plans_in_graph(
        bs=bs, page_table=ctx.page_table, page_size=ctx.page_size, num_pages=int(k0.shape[0]) - 1,
        row_len=int(ctx.page_table.shape[1]), device=gr.device, Req=Req, Batch=Batch,
        prepare_metadata=gr.attn_backend.prepare_metadata, capture=capture, replay=gr.replay,
        forward_ctx=ctx.forward_batch)


def _install_tune_before_capture() -> None:
    from minisgl.engine.graph import GraphRunner

    if getattr(GraphRunner._capture_graphs, "_msgl_tuned", False):
        return
    reference_capture = GraphRunner._capture_graphs

    def _capture_graphs(self, max_seq_len, vocab_size, model):
        mode = _STATE["gemm_tune"]
        if _STATE.get("fused_gated_mlp") and os.environ.get("MSGL_DISABLE_FUSED_SILU") != "1":
            _STATE["interleaved_mlps"] = _interleave_gated_mlps(model)
        if mode != "off" and self.max_graph_bs > 0 and _STATE["fast_linear"]:
            import torch

            from .gemm_plan import tune_projection_gemms

            groups = _projection_groups(model)
            ilv = _gate_up_is_interleaved(model)
            groups = [g + ({"silu_interleaved": True},) if all(w.data_ptr() in ilv for w in g[1]) else g for g in groups]
            if groups:
                dtype = groups[0][1][0].dtype
                log = (lambda m: print(m, file=sys.stderr)) if os.environ.get("MSGL_PLUGIN_VERBOSE") else None
                _STATE["gemm_report"] = tune_projection_gemms(groups, list(self.graph_bs_list), mode, dtype,
                                                              self.device, log=log)
                # the chunk size a loaded server prefills at: the reference's default max_extend_tokens
                # (P/scheduler/config.py:16) unless $MSGL_PREFILL_TOKENS says otherwise ("0": skip)
                chunk = int(os.environ.get("MSGL_PREFILL_TOKENS", "8192"))
                if chunk > 0:
                    from .gemm_plan import tune_prefill_gemms

                    _STATE["gemm_report"] += tune_prefill_gemms(groups, [chunk], dtype, self.device, log=log)
                torch.cuda.synchronize(self.device)
        if _STATE["fast_linear"] and os.environ.get("MSGL_DISABLE_SLAB_NORM") != "1":
            _STATE["deferred_reduce_weights"] = _deferred_reduce_weights(model)
        if _STATE.get("row_parallel_overlap") and os.environ.get("MSGL_FUSED_ALLREDUCE_NORM") == "1":  # opt-in, see model.py
            _STATE["deferred_allreduce_weights"] = _deferred_reduce_weights(model, tp_gt1=True)
        out = reference_capture(self, max_seq_len, vocab_size, model)
        world = 1
        try:
            import torch
            import torch.distributed as dist

            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # kernel search and capture take a rank-dependent time, and the device-side barriers of the peer-to-peer
                # collectives give up after a bounded spin: the ranks meet on the CPU before the first forward (the
                # reference's default group is its gloo TP group when pynccl is on, P/engine/engine.py:113-126)
                torch.cuda.synchronize(self.device)
                dist.barrier()
        except ImportError:
            pass
        if (mode != "off" and self.max_graph_bs > 0 and _STATE["fast_linear"] and world == 1
                and os.environ.get("MSGL_DISABLE_REFINE") != "1"):
            _STATE["refine_report"] = _refine_reference_graphs(self, model)
        return out

    _capture_graphs._msgl_tuned = True  # type: ignore[attr-defined]
    GraphRunner._capture_graphs = _capture_graphs


# ------------------------------------------------------------------------------ deterministic decode order (opt-in)
def _install_deterministic_decode_order(only_under_tp: bool = False) -> None:
    """`DecodeManager.running_reqs` is a `set` of eq=False dataclasses (P/scheduler/decode.py:12,35, P/core.py:28): the
    decode batch order -- and with it the order `allocate_paged` hands out free pages (P/scheduler/cache.py:42-53) -- is
    the set's iteration order, i.e. object addresses.  Opt-in: schedule decode batches in uid order, so that two runs
    of the same request stream produce the same batches and KV block indices."""
    from minisgl.core import Batch
    from minisgl.scheduler.decode import DecodeManager

    if getattr(DecodeManager.schedule_next_batch, "_msgl_sorted", False):
        if not only_under_tp:
            DecodeManager.schedule_next_batch._msgl_always = True  # type: ignore[attr-defined]
        return
    reference_schedule = DecodeManager.schedule_next_batch

    def schedule_next_batch(self):
        if not schedule_next_batch._msgl_always:
            try:
                from minisgl.distributed import get_tp_info

                if get_tp_info().size == 1:
                    return reference_schedule(self)
            except Exception:
                return reference_schedule(self)
        if not self.runnable:
            return None
        return Batch(reqs=sorted(self.running_reqs, key=lambda r: r.uid), phase="decode")

    schedule_next_batch._msgl_always = not only_under_tp  # type: ignore[attr-defined]

    schedule_next_batch._msgl_sorted = True  # type: ignore[attr-defined]
    DecodeManager.schedule_next_batch = schedule_next_batch


# ------------------------------------------------------------------------------ native radix prefix cache
def _install_native_radix(replace_radix: bool) -> None:
    """Cache type "hip_radix" in the reference's registry (P/kvcache/__init__.py:23,59-64): RadixPrefixCache with the
    tree walk in native code (radix.py, csrc/radix.cpp).  replace_radix=True also points the name "radix" (the
    scheduler's default, BASELINE configs 2-3) at it."""
    import minisgl.kvcache as kvc
    from minisgl.core import get_global_ctx
    from minisgl.kvcache.base import BaseCacheHandle, BasePrefixCache, InsertResult, MatchResult, SizeInfo

    from .radix import make_prefix_cache_class

    if "hip_radix" not in kvc.SUPPORTED_CACHE_MANAGER.supported_names():
        cache_cls, _ = make_prefix_cache_class(BasePrefixCache, BaseCacheHandle, MatchResult, InsertResult, SizeInfo,
                                               lambda: get_global_ctx().page_size)
        _STATE["native_radix_class"] = cache_cls

        def create_native_radix(device):
            return cache_cls(device=device)

        kvc.SUPPORTED_CACHE_MANAGER.register("hip_radix")(create_native_radix)
    if replace_radix:
        reg = kvc.SUPPORTED_CACHE_MANAGER._registry
        _STATE.setdefault("reference_radix_factory", reg["radix"])
        reg["radix"] = reg["hip_radix"]


# ------------------------------------------------------------------------------ vectorised scheduler glue (opt-in)
def _install_vectorized_glue() -> None:
    """`_make_positions`, `_make_input_tuple`, `_make_write_tuple` of P/scheduler/scheduler.py:236-267 (module-level
    names looked up by `Scheduler._prepare_batch` at call time) -> the numpy versions of sched_glue.py."""
    import minisgl.scheduler.scheduler as sched

    from . import sched_glue

    if getattr(sched._make_positions, "_msgl_vectorized", False):
        return
    _STATE["reference_glue"] = (sched._make_positions, sched._make_input_tuple, sched._make_write_tuple)
    for name, fn in (("_make_positions", sched_glue.make_positions), ("_make_input_tuple", sched_glue.make_input_tuple),
                     ("_make_write_tuple", sched_glue.make_write_tuple)):
        fn._msgl_vectorized = True  # type: ignore[attr-defined]
        setattr(sched, name, fn)


def gemm_report() -> List[dict]:
    """What the last pre-capture search chose (one dict per (batch size, projection))."""
    return list(_STATE["gemm_report"])


def install(stub_zmq: bool = True, *, fast_linear: bool = True, fused_attention: bool = True, fused_mlp: bool = True,
            comm_overlap: bool = True,
            gemm_tune: Optional[str] = None, deterministic_decode_order: Optional[bool] = None, native_radix: bool = True,
            vectorized_glue: bool = True) -> None:
    """gemm_tune: "off" | "heuristic" | "full" (default: $MSGL_GEMM_TUNE or "heuristic"); unless "off", the library's solutions
        are also searched at the prefill chunk size $MSGL_PREFILL_TOKENS (default 8192 = the reference's max_extend_tokens,
        P/scheduler/config.py:16; "0" skips it): o_proj / down_proj gain 1.4x over the heuristic pick there.
    fused_mlp: gate_up_proj + silu_and_mul of the dense GatedMLP as ops.linear_silu (weights interleaved once, in place).
    comm_overlap: under TP, row-parallel projections of >= $MSGL_COMM_SPLIT_TOKENS (2048) tokens run as two token halves with the
        first half's all-reduce on a side stream (second communicator); see _install_row_parallel_overlap.
    deterministic_decode_order: decode batches in uid order instead of set-iteration order (reproducible KV indices);
        None (default) = only under tensor parallelism, where the replicated schedulers of the ranks MUST build the same
        batch (the reference's order is the iteration order of a set of id-hashed objects, i.e. heap addresses).
    native_radix (default on): cache_type="radix" uses the native tree walk too (cache_type="hip_radix" always does);
        same matches, inserts, evictions and free lists as the reference's tree (tests/test_cpu_native_radix.py).
    vectorized_glue (default on): the scheduler's per-step index tensors (positions, input / write tuples) by numpy over
        the whole batch; the same tensors as the reference's functions (tests/test_cpu_reference_native.py).
    Measured through the reference's scheduler on the README offline benchmark at Qwen3-0.6B dims
    (profiles/r03_refdrive_0p6b_host.json): host time in _schedule_next_batch 350 -> 202 us per iteration; with overlap
    scheduling on the step is GPU-bound either way (4.65 ms, +0.4 % throughput), with it off +4 %."""
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

    _STATE["gemm_tune"] = gemm_tune or os.environ.get("MSGL_GEMM_TUNE", "heuristic")
    if fast_linear:
        _install_fast_linear()
        _install_tune_before_capture()
    if fused_attention:
        _install_fused_attention()
    if fused_mlp and fast_linear:
        _install_fused_gated_mlp()
    if fast_linear and comm_overlap:
        os.environ.setdefault("MSGL_COMM_OVERLAP", "1")  # init_pynccl (called by the reference) then builds the side communicator
        _install_row_parallel_overlap()
    if deterministic_decode_order is None or deterministic_decode_order:
        _install_deterministic_decode_order(only_under_tp=deterministic_decode_order is None)
    _install_native_radix(replace_radix=native_radix)
    if vectorized_glue:
        _install_vectorized_glue()
