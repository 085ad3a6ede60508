"""CPU oracle of the dense decoder forward (TEST INFRASTRUCTURE ONLY; also the `cpu_baseline`
leg of bench.py).  torch-eager restatement of P/models/qwen3.py:18-81, P/models/utils.py:25-123,
P/layers/attention.py:47-57, P/layers/embedding.py:33-110 and P/engine/sample.py:71-75 on top of
oracle/ref_ops.py: F.embedding, fp32-math RMSNorm, F.linear, NeoX RoPE from the cat(cos, sin)
cache, attention over K/V gathered through the page table, silu*mul, argmax.  Activations are
rounded to the model dtype at the same op boundaries as the device pipeline."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ref_ops


@dataclass
class CpuWeights:
    embed: torch.Tensor
    layers: List[Dict[str, Optional[torch.Tensor]]]
    final_norm: torch.Tensor
    lm_head: torch.Tensor
    cos_sin: torch.Tensor


def weights_from_device_model(model: Any) -> CpuWeights:
    """Copy a DenseDecoder's tensors to the host (the oracle must see the same bits)."""
    layers = []
    for li, lw in enumerate(model.layers):
        layers.append({k: (None if getattr(lw, k) is None else getattr(lw, k).detach().cpu())
                       for k in ("input_norm", "qkv", "q_norm", "k_norm", "o", "post_norm", "gate_up", "down")})
        if getattr(model, "gate_up_ilv", False):  # the device model keeps gate_up rows interleaved for its fused epilogue
            layers[-1]["gate_up"] = model.gate_up_reference(li).detach().cpu()
    return CpuWeights(model.embed.cpu(), layers, model.final_norm.cpu(), model.lm_head.cpu(), model.cos_sin.cpu())


def random_weights(cfg: Any, dtype: torch.dtype, seed: int = 42, init_std: float = 0.02,
                   num_layers: Optional[int] = None) -> CpuWeights:
    """Seeded N(0, std^2) weights with norm weights = 1."""
    g = torch.Generator().manual_seed(seed)
    D, H = cfg.head_dim, cfg.hidden_size
    qd, kd = cfg.num_qo_heads * D, cfg.num_kv_heads * D

    def w(*shape):
        return (torch.randn(shape, generator=g) * init_std).to(dtype)

    ones = lambda n: torch.ones(n, dtype=dtype)  # noqa: E731
    layers = []
    for _ in range(num_layers if num_layers is not None else cfg.num_layers):
        layers.append(dict(input_norm=ones(H), qkv=w(qd + 2 * kd, H), q_norm=ones(D) if cfg.qk_norm else None,
                           k_norm=ones(D) if cfg.qk_norm else None, o=w(H, qd), post_norm=ones(H),
                           gate_up=w(2 * cfg.intermediate_size, H), down=w(H, cfg.intermediate_size)))
    embed = w(cfg.vocab_size, H)
    lm = embed if cfg.tie_word_embeddings else w(cfg.vocab_size, H)
    cos_sin = ref_ops.rope_cos_sin_cache(D, cfg.max_position, cfg.rope_base, cfg.rope_scaling)
    return CpuWeights(embed, layers, ones(H), lm, cos_sin)


def forward(cfg: Any, w: CpuWeights, input_ids: torch.Tensor, positions: torch.Tensor, out_loc: torch.Tensor,
            k_pool: List[torch.Tensor], v_pool: List[torch.Tensor], page_table: torch.Tensor,
            req_rows: Sequence[int], k_lens: Sequence[int], q_lens: Sequence[int], is_prefill: bool,
            layers: Optional[range] = None) -> torch.Tensor:
    """One forward over a batch.  k_pool[l] / v_pool[l]: [slots, Hkv, D] CPU tensors (updated in
    place at out_loc).  Returns logits [B, vocab] in the model dtype."""
    D = cfg.head_dim
    hq, hkv = cfg.num_qo_heads, cfg.num_kv_heads
    dt = w.embed.dtype
    x = F.embedding(input_ids.long(), w.embed)
    residual = None
    for li in (layers if layers is not None else range(len(w.layers))):
        lw = w.layers[li]
        if residual is None:
            residual = x
            x = ref_ops.rmsnorm_ref(x, lw["input_norm"], cfg.rms_norm_eps)
        else:
            x, residual = ref_ops.fused_add_rmsnorm_ref(x, residual, lw["input_norm"], cfg.rms_norm_eps)
        qkv = F.linear(x.float(), lw["qkv"].float()).to(dt)
        q, k, v = qkv.split([hq * D, hkv * D, hkv * D], dim=-1)
        T = q.shape[0]
        if lw["q_norm"] is not None:
            q = ref_ops.rmsnorm_ref(q.reshape(T, hq, D), lw["q_norm"], cfg.rms_norm_eps).reshape(T, hq * D)
            k = ref_ops.rmsnorm_ref(k.reshape(T, hkv, D), lw["k_norm"], cfg.rms_norm_eps).reshape(T, hkv * D)
        q, k = ref_ops.rope_neox_ref(positions, q, k, D, w.cos_sin)
        ref_ops.store_kv_ref(k_pool[li].view(-1, hkv * D), v_pool[li].view(-1, hkv * D), out_loc, k, v)
        o = ref_ops.paged_attention_ref(q.reshape(T, hq, D), k_pool[li], v_pool[li], page_table, req_rows, k_lens,
                                        q_lens, D ** -0.5)
        x = F.linear(o.reshape(T, hq * D).float(), lw["o"].float()).to(dt)
        x, residual = ref_ops.fused_add_rmsnorm_ref(x, residual, lw["post_norm"], cfg.rms_norm_eps)
        gu = F.linear(x.float(), lw["gate_up"].float()).to(dt)
        y = ref_ops.silu_and_mul_ref(gu)
        x = F.linear(y.float(), lw["down"].float()).to(dt)
    x, _ = ref_ops.fused_add_rmsnorm_ref(x, residual, w.final_norm, cfg.rms_norm_eps)
    if is_prefill:
        last = torch.tensor(q_lens).cumsum(0) - 1
        x = x[last]
    return F.linear(x.float(), w.lm_head.float()).to(dt)


# --------------------------------------------------------------------------- tensor parallel
def shard_weights(cfg: Any, full: CpuWeights, tp_size: int, tp_rank: int) -> CpuWeights:
    """Megatron-style shards of full weights, as the reference's loader cuts them
    (P/models/weight.py:34-52, P/layers/linear.py:56-127, P/layers/embedding.py:25-31):
    q/k/v by heads (kv heads replicated when tp > Hkv), gate/up by the intermediate dim,
    o/down on the input dim, embedding / LM head by vocab rows (zero-padded last shard)."""
    D = cfg.head_dim
    hq, hkv, inter = cfg.num_qo_heads, cfg.num_kv_heads, cfg.intermediate_size
    hq_l = hq // tp_size
    hkv_l = max(hkv // tp_size, 1)
    kv_rank = tp_rank if tp_size <= hkv else tp_rank // (tp_size // hkv)
    i_l = inter // tp_size
    layers = []
    for lw in full.layers:
        qkv = lw["qkv"]
        q, k, v = qkv[: hq * D], qkv[hq * D: (hq + hkv) * D], qkv[(hq + hkv) * D:]
        q_s = q[tp_rank * hq_l * D: (tp_rank + 1) * hq_l * D]
        k_s = k[kv_rank * hkv_l * D: (kv_rank + 1) * hkv_l * D]
        v_s = v[kv_rank * hkv_l * D: (kv_rank + 1) * hkv_l * D]
        gate, up = lw["gate_up"][:inter], lw["gate_up"][inter:]
        layers.append(dict(
            input_norm=lw["input_norm"], post_norm=lw["post_norm"], q_norm=lw["q_norm"], k_norm=lw["k_norm"],
            qkv=torch.cat([q_s, k_s, v_s], 0),
            o=lw["o"][:, tp_rank * hq_l * D: (tp_rank + 1) * hq_l * D],
            gate_up=torch.cat([gate[tp_rank * i_l: (tp_rank + 1) * i_l], up[tp_rank * i_l: (tp_rank + 1) * i_l]], 0),
            down=lw["down"][:, tp_rank * i_l: (tp_rank + 1) * i_l]))
    per = (cfg.vocab_size + tp_size - 1) // tp_size

    def rows(t):
        out = torch.zeros((per, t.shape[1]), dtype=t.dtype)
        part = t[per * tp_rank: per * (tp_rank + 1)]
        out[: part.shape[0]] = part
        return out

    return CpuWeights(rows(full.embed), layers, full.final_norm, rows(full.lm_head), full.cos_sin)


def forward_tp(cfg: Any, w: CpuWeights, tp_size: int, tp_rank: int, all_reduce, all_gather, input_ids, positions,
               out_loc, k_pool, v_pool, page_table, req_rows, k_lens, q_lens, is_prefill: bool) -> torch.Tensor:
    """One rank of the tensor-parallel forward: same op order as forward(), with the collectives
    of P/layers/linear.py:105,126 and P/layers/embedding.py:42,102 (`all_reduce(t) -> t` in place
    SUM, `all_gather(t) -> [tp * rows, ...]`)."""
    from mini_sglang_amd.model import lm_head_unshard, vocab_shard  # host-side layout helpers of the product

    D = cfg.head_dim
    hq, hkv = cfg.num_qo_heads // tp_size, max(cfg.num_kv_heads // tp_size, 1)
    dt = w.embed.dtype
    _, rng = vocab_shard(cfg.vocab_size, tp_size, tp_rank)
    x = all_reduce(ref_ops.indexing_ref(w.embed, input_ids, rng))
    residual = None
    for li, lw in enumerate(w.layers):
        if residual is None:
            residual = x
            x = ref_ops.rmsnorm_ref(x, lw["input_norm"], cfg.rms_norm_eps)
        else:
            x, residual = ref_ops.fused_add_rmsnorm_ref(x, residual, lw["input_norm"], cfg.rms_norm_eps)
        qkv = F.linear(x.float(), lw["qkv"].float()).to(dt)
        q, k, v = qkv.split([hq * D, hkv * D, hkv * D], dim=-1)
        T = q.shape[0]
        if lw["q_norm"] is not None:
            q = ref_ops.rmsnorm_ref(q.reshape(T, hq, D), lw["q_norm"], cfg.rms_norm_eps).reshape(T, hq * D)
            k = ref_ops.rmsnorm_ref(k.reshape(T, hkv, D), lw["k_norm"], cfg.rms_norm_eps).reshape(T, hkv * D)
        q, k = ref_ops.rope_neox_ref(positions, q, k, D, w.cos_sin)
        ref_ops.store_kv_ref(k_pool[li].view(-1, hkv * D), v_pool[li].view(-1, hkv * D), out_loc, k, v)
        o = ref_ops.paged_attention_ref(q.reshape(T, hq, D), k_pool[li], v_pool[li], page_table, req_rows, k_lens,
                                        q_lens, D ** -0.5)
        x = all_reduce(F.linear(o.reshape(T, hq * D).float(), lw["o"].float()).to(dt))
        x, residual = ref_ops.fused_add_rmsnorm_ref(x, residual, lw["post_norm"], cfg.rms_norm_eps)
        gu = F.linear(x.float(), lw["gate_up"].float()).to(dt)
        y = ref_ops.silu_and_mul_ref(gu)
        x = all_reduce(F.linear(y.float(), lw["down"].float()).to(dt))
    x, _ = ref_ops.fused_add_rmsnorm_ref(x, residual, w.final_norm, cfg.rms_norm_eps)
    if is_prefill:
        x = x[torch.tensor(q_lens).cumsum(0) - 1]
    logits = F.linear(x.float(), w.lm_head.float()).to(dt)
    return lm_head_unshard(all_gather(logits), tp_size, logits.shape[0], cfg.vocab_size)
