"""Mirror of the flashinfer functions the reference binds on the dense path.

The reference reaches flashinfer through attributes bound in constructors
(P/layers/norm.py:10-30, P/layers/rotary.py:35-37) and lazy imports
(P/layers/activation.py:10,16, P/engine/sample.py:30); a module exposing the same names
with the same argument meaning is a drop-in.  Everything forwards to
the hand-written gfx950 kernels via ops.py.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import torch

from . import ops


# ---- norm (P/layers/norm.py:17,20,37) -----------------------------------------------------
def rmsnorm(input: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
            out: Optional[torch.Tensor] = None, enable_pdl: Optional[bool] = None) -> torch.Tensor:
    return ops.rmsnorm(input, weight, eps, out=out)


def fused_add_rmsnorm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
                      enable_pdl: Optional[bool] = None) -> None:
    slabs = getattr(input, "_msgl_slabs", None)
    comm = getattr(input, "_msgl_allreduce", None)
    if slabs is not None:  # output of a split-K projection whose reduce was left to this kernel (ops.linear_slabs)
        del input._msgl_slabs
        ops.fused_add_rmsnorm_slabs(input, residual, weight, eps, slabs)
    elif comm is not None:  # output of a row-parallel projection whose all-reduce was left to this norm (TP decode)
        del input._msgl_allreduce
        ops._PENDING_ALLREDUCE[input.device.index or 0] = None
        comm.all_reduce_add_rmsnorm(input, residual, weight, eps)
    else:
        ops.fused_add_rmsnorm(input, residual, weight, eps)


# ---- rope (P/layers/rotary.py:45-51) ------------------------------------------------------
def apply_rope_with_cos_sin_cache_inplace(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor,
                                          head_size: int, cos_sin_cache: torch.Tensor,
                                          is_neox: bool = True) -> None:
    if not is_neox:
        raise NotImplementedError("only NeoX (rotate-half) RoPE is on the reference's path")
    ops.rope_neox_inplace(positions, query, key, head_size, cos_sin_cache)


def build_cos_sin_cache(rotary_dim: int, max_position: int, base: float,
                        rope_scaling: Optional[Dict[str, Any]] = None,
                        device: torch.device | str = "cpu") -> torch.Tensor:
    """fp32 [max_position, rotary_dim] = cat(cos, sin); same construction as RotaryEmbedding.__init__ /
    _get_rope (P/layers/rotary.py:24-32, 55-114): default, llama3 and yarn inverse-frequency post-processing.
    Like the reference (`with torch.device(_ROPE_DEVICE)`, rotary.py:133-141) every tensor op runs ON `device`,
    so the table has the reference's bits on the CPU (golden fixtures) and on the GPU (reference-driven runs)."""
    with torch.device(device):
        return _cos_sin_cache(rotary_dim, max_position, base, rope_scaling)


def _cos_sin_cache(rotary_dim: int, max_position: int, base: float,
                   rope_scaling: Optional[Dict[str, Any]]) -> torch.Tensor:
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    kind = None if rope_scaling is None else rope_scaling.get("rope_type", "default")
    if kind == "llama3":
        factor = rope_scaling["factor"]
        low, high = rope_scaling["low_freq_factor"], rope_scaling["high_freq_factor"]
        orig = rope_scaling["original_max_position_embeddings"]
        wave_len = 2 * math.pi / inv_freq
        if low == high:
            inv_freq = torch.where(wave_len < orig / high, inv_freq, inv_freq / factor)
        else:
            smooth = torch.clamp((orig / wave_len - low) / (high - low), 0, 1)
            inv_freq = ((1 - smooth) / factor + smooth) * inv_freq
    elif kind == "yarn":
        factor = rope_scaling["factor"]
        beta_fast = rope_scaling.get("beta_fast", 32.0)
        beta_slow = rope_scaling.get("beta_slow", 1.0)
        orig = rope_scaling["original_max_position_embeddings"]

        def corr_dim(rotations: float) -> float:
            return rotary_dim * math.log(orig / (rotations * 2 * math.pi)) / (2 * math.log(base))

        lo = max(math.floor(corr_dim(beta_fast)), 0)
        hi = min(math.ceil(corr_dim(beta_slow)), rotary_dim // 2 - 1)
        ramp = torch.clamp((torch.arange(rotary_dim // 2, dtype=torch.float32) - lo) / max(hi - lo, 1), 0, 1)
        inv_freq = (inv_freq / factor) * ramp + inv_freq * (1 - ramp)
    elif kind not in (None, "default"):
        raise ValueError(f"Unsupported rope_scaling = {rope_scaling}")
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


# ---- activation (P/layers/activation.py:9-18) ---------------------------------------------
def silu_and_mul(input: torch.Tensor, out: Optional[torch.Tensor] = None,
                 enable_pdl: Optional[bool] = None) -> torch.Tensor:
    return ops.silu_and_mul(input, out=out)


def gelu_and_mul(input: torch.Tensor, out: Optional[torch.Tensor] = None,
                 enable_pdl: Optional[bool] = None) -> torch.Tensor:
    return ops.gelu_and_mul(input, out=out)


# ---- sampling (P/engine/sample.py:30-45) --------------------------------------------------
class _DeferredProbs:
    """What `softmax(logits, T)` hands back: the reference's `sample_impl` (P/engine/sample.py:24-45) passes it
    straight to one of the sampling functions and never looks inside.  Temperature-only batches are then drawn
    by ONE fused kernel from the logits (no [B, V] fp32 probabilities tensor); the top-k / top-p functions
    materialise the probabilities and run the radix-select kernel.  Anything else that treats the object as a
    tensor gets the materialised tensor through `__torch_function__`-free explicit conversion (`.probs()`)."""

    __slots__ = ("logits", "temperature", "_probs")

    def __init__(self, logits: torch.Tensor, temperature: torch.Tensor) -> None:
        self.logits, self.temperature, self._probs = logits, temperature, None

    @property
    def shape(self):
        return self.logits.shape

    @property
    def device(self):
        return self.logits.device

    def fusable(self) -> bool:
        lg = self.logits
        return (lg.stride(0) * lg.element_size()) % 16 == 0 and lg.data_ptr() % 16 == 0

    def probs(self) -> torch.Tensor:
        if self._probs is None:
            self._probs = ops.softmax_temperature(self.logits, self.temperature)
        return self._probs


def _as_probs(p) -> torch.Tensor:
    return p.probs() if isinstance(p, _DeferredProbs) else p


class _SamplingNamespace:
    """`import flashinfer.sampling as sampling` surface."""

    def __init__(self) -> None:
        self._offset = 0

    def _next_offset(self, rows: int) -> int:
        off = self._offset
        self._offset += rows
        return off

    @staticmethod
    def softmax(logits: torch.Tensor, temperature: Optional[torch.Tensor] = None,
                enable_pdl: Optional[bool] = None, *, deferred: bool = True):
        if temperature is None:
            temperature = torch.ones(logits.shape[0], dtype=torch.float32, device=logits.device)
        d = _DeferredProbs(logits, temperature)
        return d if deferred else d.probs()

    def _sample(self, probs, top_k, top_p):
        probs = _as_probs(probs)
        dev = probs.device
        if isinstance(top_k, int):
            top_k = torch.full((probs.shape[0],), top_k, dtype=torch.int32, device=dev)
        if isinstance(top_p, float):
            top_p = torch.full((probs.shape[0],), top_p, dtype=torch.float32, device=dev)
        # every TP rank seeds identically (P/engine/engine.py:37) => identical draws on all ranks
        return ops.sample_top_k_top_p(probs, top_k, top_p, torch.initial_seed(), self._next_offset(probs.shape[0]))

    def sampling_from_probs(self, probs, **_kw):
        if isinstance(probs, _DeferredProbs) and probs.fusable():
            return self.sampling_from_logits(probs.logits, probs.temperature)
        return self._sample(probs, None, None)

    def sampling_from_logits(self, logits, temperature, **_kw):
        """Fused softmax(logits / T) + draw (same Philox stream as the probs functions)."""
        return ops.sample_from_logits(logits, temperature, torch.initial_seed(), self._next_offset(logits.shape[0]))

    def top_k_sampling_from_probs(self, probs, top_k, **_kw):
        return self._sample(probs, top_k, None)

    def top_p_sampling_from_probs(self, probs, top_p, **_kw):
        return self._sample(probs, None, top_p)

    def top_k_top_p_sampling_from_probs(self, probs, top_k, top_p, **_kw):
        return self._sample(probs, top_k, top_p)


sampling = _SamplingNamespace()
