"""TEST INFRASTRUCTURE ONLY: an INDEPENDENT bf16 forward of the dense decoder on the GPU, in plain torch ops on ROCm.

Why it exists (VERDICT r3 "next" 2): north_star asks for "bf16 logits within 1e-3" of the reference, the reference's
floating-point kernels (flashinfer, sgl_kernel) are absent, and the repository's kernels differ from the fp32-accumulating
CPU oracle (oracle/ref_model.py) by up to ~0.08 logit std at Qwen3-14B width.  Is that band the floor of ANY bf16
pipeline, or these kernels?  This file is the second opinion: the same op order and the same rounding points as the
oracle (P/models/qwen3.py:18-81, P/layers/attention.py:47-57) -- bf16 activations between ops, fp32 math
inside norm / RoPE / activation / softmax -- but every GEMM is torch's own bf16 `F.linear` (hipBLASLt / rocBLAS, their
accumulation order) and attention is written out with torch matmuls, P rounded to bf16 before P.V (the tensor-core
flash-attention convention).  Nothing here comes from mini-sglang_amd: no kernel, no plan, no layout helper.

tests/ compare   |ours - fp32 oracle|   with   |this - fp32 oracle|   on the same teacher-forced batches.

Round 6: `linear="fp32acc"` replaces every F.linear by an fp32 matmul of the bf16 values rounded once to
bf16 -- with CPU tensors that is a bf16 pipeline whose GEMMs owe nothing to hipBLASLt / rocBLAS (the floor is then independent
of the device and of its libraries).
"""
from __future__ import annotations

from typing import Any, List, Optional, Sequence

import torch
import torch.nn.functional as F


def weights_to(w: Any, device: torch.device) -> Any:
    """A ref_model.CpuWeights moved to `device` (same bits)."""
    from .ref_model import CpuWeights

    mv = lambda t: None if t is None else t.to(device)  # noqa: E731
    return CpuWeights(mv(w.embed), [{k: mv(v) for k, v in lw.items()} for lw in w.layers], mv(w.final_norm), mv(w.lm_head),
                      mv(w.cos_sin))


def _rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return (y * weight.float()).to(x.dtype)


def _add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """flashinfer.fused_add_rmsnorm semantics: residual <- round(x + residual); x <- norm of the UNROUNDED sum."""
    s = x.float() + residual.float()
    y = s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + eps)
    return (y * weight.float()).to(x.dtype), s.to(x.dtype)


def _rope_neox(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor, head_dim: int, cos_sin: torch.Tensor):
    """NeoX rotation from the cat(cos, sin) fp32 cache (P/layers/rotary.py:45-51), fp32 math, one rounding."""
    cs = cos_sin[positions.long()]
    half = head_dim // 2
    cos, sin = cs[:, None, :half], cs[:, None, half:]

    def rot(t):
        T = t.shape[0]
        tf = t.float().view(T, -1, head_dim)
        a, b = tf[..., :half], tf[..., half:]
        return torch.cat([a * cos - b * sin, b * cos + a * sin], -1).to(t.dtype).view(T, -1)

    return rot(q), rot(k)


def _attention(q: torch.Tensor, k_pool: torch.Tensor, v_pool: torch.Tensor, table: torch.Tensor, rows: Sequence[int],
               k_lens: Sequence[int], q_lens: Sequence[int], scale: float) -> torch.Tensor:
    """q [T, Hq, D]; pools [slots, Hkv, D]; bottom-right aligned causal mask; S and softmax in fp32, P rounded to the
    model dtype, O accumulated in fp32 and rounded once."""
    T, hq, d = q.shape
    hkv = k_pool.shape[1]
    g = hq // hkv
    out = torch.empty_like(q)
    off = 0
    for row, kl, ql in zip(rows, k_lens, q_lens):
        slots = table[row, :kl].long()
        K = k_pool[slots].float().permute(1, 0, 2)            # [Hkv, k, D]
        V = v_pool[slots].float().permute(1, 0, 2)
        Q = q[off: off + ql].float().view(ql, hkv, g, d).permute(1, 2, 0, 3).reshape(hkv, g * ql, d)
        s = torch.bmm(Q, K.transpose(1, 2)) * scale           # [Hkv, g*q, k]
        if ql > 1:
            qpos = torch.arange(ql, device=q.device).repeat(g) + (kl - ql)
            s = s.masked_fill(torch.arange(kl, device=q.device)[None, None, :] > qpos[None, :, None], float("-inf"))
        p = torch.softmax(s, dim=-1).to(q.dtype).float()
        o = torch.bmm(p, V)                                   # [Hkv, g*q, D]
        out[off: off + ql] = o.view(hkv, g, ql, d).permute(2, 0, 1, 3).reshape(ql, hq, d).to(q.dtype)
        off += ql
    return out


def _linear_fp32acc(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    return (x.float() @ w.float().t()).to(x.dtype)


def forward(cfg: Any, w: Any, input_ids: torch.Tensor, positions: torch.Tensor, out_loc: torch.Tensor,
            k_pool: List[torch.Tensor], v_pool: List[torch.Tensor], page_table: torch.Tensor, req_rows: Sequence[int],
            k_lens: Sequence[int], q_lens: Sequence[int], is_prefill: bool,
            hq: Optional[int] = None, hkv: Optional[int] = None, linear: str = "library") -> torch.Tensor:
    """Same contract as ref_model.forward, every tensor on the GPU; k_pool[l] / v_pool[l] [slots, Hkv, D] are updated in
    place at out_loc.  Returns logits [B, vocab] in the model dtype."""
    D = cfg.head_dim
    hq = hq or cfg.num_qo_heads
    hkv = hkv or cfg.num_kv_heads
    eps = cfg.rms_norm_eps
    lin = F.linear if linear == "library" else _linear_fp32acc
    x = F.embedding(input_ids.long(), w.embed)
    residual = None
    for li, lw in enumerate(w.layers):
        if residual is None:
            residual = x
            x = _rmsnorm(x, lw["input_norm"], eps)
        else:
            x, residual = _add_rmsnorm(x, residual, lw["input_norm"], eps)
        qkv = lin(x, lw["qkv"])
        q, k, v = qkv.split([hq * D, hkv * D, hkv * D], dim=-1)
        T = q.shape[0]
        if lw["q_norm"] is not None:
            q = _rmsnorm(q.reshape(T, hq, D), lw["q_norm"], eps).reshape(T, hq * D)
            k = _rmsnorm(k.reshape(T, hkv, D), lw["k_norm"], eps).reshape(T, hkv * D)
        q, k = _rope_neox(positions, q, k, D, w.cos_sin)
        k_pool[li][out_loc.long()] = k.view(T, hkv, D)
        v_pool[li][out_loc.long()] = v.reshape(T, hkv, D)
        o = _attention(q.view(T, hq, D), k_pool[li], v_pool[li], page_table, req_rows, k_lens, q_lens, D ** -0.5)
        x = lin(o.view(T, hq * D), lw["o"])
        x, residual = _add_rmsnorm(x, residual, lw["post_norm"], eps)
        gu = lin(x, lw["gate_up"])
        half = gu.shape[-1] // 2
        y = (F.silu(gu[:, :half].float()) * gu[:, half:].float()).to(gu.dtype)
        x = lin(y, lw["down"])
    x, _ = _add_rmsnorm(x, residual, w.final_norm, eps)
    if is_prefill:
        x = x[torch.tensor(q_lens, device=x.device).cumsum(0) - 1]
    return lin(x, w.lm_head)
