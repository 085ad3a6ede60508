"""Dense decoder stack (Qwen3 / Qwen2 / Llama family) built on the hot-path seams.

This is the caller of the path, restated compactly from the reference's composition
(P/models/qwen3.py:18-81, P/models/llama.py, P/models/utils.py:25-123, P/layers/*): the
reference's own model classes run unchanged on top of the plugin; this copy
exists because /root/reference is not present on the GPU box and bench.py / smoke need a
driver.  GEMMs are the library's (hipBLASLt) as in the reference (P/layers/linear.py:32), reached through
msgl_gemm_nt so that the per-shape solution search of csrc/gemm.cpp applies (`tune_gemms`); everything
else goes through the hand-written gfx950 kernels.

Two execution modes, bit-identical by construction (tests/test_gpu_model.py):
  fused=False  the reference's op order through its seams: q_norm, k_norm, rope (flashinfer
               names), attn_backend.forward (store_kv + attention)   [P/layers/attention.py:47-57]
  fused=True   one qk_norm_rope_store kernel + attn_backend.attend
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import os

import torch

from . import flashinfer_compat as fi
from . import ops
from .kvcache import heads_per_rank

# MSGL_DISABLE_SLAB_NORM=1: keep the split-K reduce of o_proj / down_proj as its own launch (A/B switch)
_SLAB_NORM = os.environ.get("MSGL_DISABLE_SLAB_NORM") != "1"
_FUSE_SILU = os.environ.get("MSGL_DISABLE_FUSED_SILU") != "1"
# MSGL_DISABLE_ROWSTREAM_FUSE=1: at batches <= 8 keep fused_add_rmsnorm / SiLU.mul as their own launches even where the
# projection that consumes them is the row-streaming kernel (A/B switch; the results are bit-identical either way)
_ROWSTREAM_FUSE = os.environ.get("MSGL_DISABLE_ROWSTREAM_FUSE") != "1"
# all-reduce + residual add + RMSNorm as one peer-to-peer launch (csrc/comm_p2p.hip): OPT-IN.  The only measurement
# available without a multi-GPU box -- one rank's shard with looped-back collectives, bench.py --rank-shard 4 -- has it
# SLOWER than the two launches (10.24 vs 9.07 ms per step): the kernel keeps the few dozen blocks its flag barriers want,
# and those cannot move the 10 MB of local x / residual traffic of a 256-row batch as fast as the 256-block norm kernel.
_FUSE_AR_NORM = os.environ.get("MSGL_FUSED_ALLREDUCE_NORM") == "1"


@dataclass(frozen=True)
class ModelConfig:
    """Subset of P/models/config.py:19-87 used by dense models."""
    num_layers: int
    num_qo_heads: int
    num_kv_heads: int
    head_dim: int
    hidden_size: int
    vocab_size: int
    intermediate_size: int
    rms_norm_eps: float = 1e-6
    rope_base: float = 1000000.0
    rope_scaling: Optional[Dict[str, Any]] = None
    max_position: int = 40960
    tie_word_embeddings: bool = False
    qk_norm: bool = True  # Qwen3: per-head RMSNorm on q and k (P/models/qwen3.py:21)
    name: str = "custom"


PRESETS: Dict[str, ModelConfig] = {
    # public HF config.json values
    "qwen3-0.6b": ModelConfig(28, 16, 8, 128, 1024, 151936, 3072, tie_word_embeddings=True, name="Qwen3-0.6B"),
    "qwen3-14b": ModelConfig(40, 40, 8, 128, 5120, 151936, 17408, name="Qwen3-14B"),
    "qwen3-32b": ModelConfig(64, 64, 8, 128, 5120, 151936, 25600, name="Qwen3-32B"),
    "llama-3.1-70b": ModelConfig(
        80, 64, 8, 128, 8192, 128256, 28672, rms_norm_eps=1e-5, rope_base=500000.0, max_position=131072,
        rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                          original_max_position_embeddings=8192),
        qk_norm=False, name="Llama-3.1-70B-Instruct"),
    # small shapes for tests (same structure, GQA group 5 like Qwen3-14B)
    "tiny": ModelConfig(2, 10, 2, 128, 256, 1024, 512, max_position=4096, name="tiny"),
    "tiny-llama": ModelConfig(2, 8, 2, 128, 256, 1024, 512, rms_norm_eps=1e-5, qk_norm=False, max_position=4096,
                              name="tiny-llama"),
    # Qwen3-14B's layer at full width (every projection has the real decode shapes, so the searched plans are the real ones),
    # two layers and a 32 k vocabulary: 2 GB of weights -- small enough to write as a checkpoint in a test
    "qwen3-14b-width-2l": ModelConfig(2, 40, 8, 128, 5120, 32768, 17408, max_position=4096, name="Qwen3-14B width, 2 layers"),
}


@dataclass
class LayerWeights:
    input_norm: torch.Tensor
    qkv: torch.Tensor       # [(Hq_l + 2 Hkv_l) * D, hidden]   (P/layers/linear.py:74-88)
    q_norm: Optional[torch.Tensor]
    k_norm: Optional[torch.Tensor]
    o: torch.Tensor         # [hidden, Hq_l * D]               row-parallel
    post_norm: torch.Tensor
    gate_up: torch.Tensor   # [2 * I_l, hidden]                (P/layers/linear.py:56-71); rows in
                            # ops.interleave_gate_up order when DenseDecoder.gate_up_ilv
    down: torch.Tensor      # [hidden, I_l]                    row-parallel


def vocab_shard(vocab_size: int, tp_size: int, tp_rank: int):
    """(shard rows incl. padding, (start, length)) of the vocab-parallel tables, P/layers/embedding.py:25-31."""
    per = (vocab_size + tp_size - 1) // tp_size
    start = per * tp_rank
    return per, (start, min(start + per, vocab_size) - start)


def lm_head_unshard(gathered: torch.Tensor, tp_size: int, rows: int, vocab_size: int) -> torch.Tensor:
    """all-gathered [tp * rows, V/tp] -> [rows, vocab] (P/layers/embedding.py:102-110)."""
    return gathered.view(tp_size, rows, -1).permute(1, 0, 2).reshape(rows, -1)[:, :vocab_size]


class Communicator:
    """all_reduce / all_gather seam (P/distributed/impl.py:63-70); identity at tp = 1.

    `side` is a second, independent communicator (own buffers / own RCCL communicator) for collectives issued on the
    side stream: two collectives of one communicator must never be in flight at once."""

    def __init__(self, impl: Any = None, tp_size: int = 1, side: Any = None) -> None:
        self.impl, self.tp_size, self.side = impl, tp_size, side

    def all_reduce(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size > 1:
            self.impl.all_reduce(x, "sum")
        return x

    def all_reduce_side(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size > 1:
            (self.side or self.impl).all_reduce(x, "sum")
        return x

    def all_reduce_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
        """all_reduce(x), then fused_add_rmsnorm(x, residual, weight, eps): ONE launch where the communicator has the
        fused peer-to-peer kernel for the shape (decode batches), otherwise the two operations."""
        if self.tp_size > 1 and hasattr(self.impl, "all_reduce_add_rmsnorm") and _FUSE_AR_NORM:
            self.impl.all_reduce_add_rmsnorm(x, residual, weight, eps)
            return x
        self.all_reduce(x)
        fi.fused_add_rmsnorm(x, residual, weight, eps)
        return x

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size == 1:
            return x
        out = x.new_empty((x.shape[0] * self.tp_size,) + tuple(x.shape[1:]))
        self.impl.all_gather(out, x)
        return out


class DenseDecoder:
    def __init__(self, cfg: ModelConfig, *, dtype: torch.dtype, device: torch.device, tp_rank: int = 0,
                 tp_size: int = 1, seed: int = 42, comm: Optional[Communicator] = None, fused: bool = True,
                 init_std: float = 0.02, comm_split_tokens: int = 0, comm_overlap: bool = True) -> None:
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.tp_rank, self.tp_size = tp_rank, tp_size
        self.fused = fused
        self.comm = comm or Communicator(None, tp_size)
        # row-parallel projections of >= comm_split_tokens tokens run as two token halves so that the all-reduce of
        # the first half (side stream, second communicator) overlaps the GEMM of the second (0 = never).  Meant for
        # prefill chunks: a decode batch would stream the weights twice for nothing (its GEMMs are weight-bound).
        self.comm_split_tokens = comm_split_tokens if tp_size > 1 else 0
        if self.comm_split_tokens and comm_overlap and self.comm.side is None:
            raise ValueError("comm_overlap with comm_split_tokens > 0 needs a second communicator (Communicator.side): two "
                             "collectives of one communicator must never be in flight at once")
        self.comm_overlap = comm_overlap
        self.side_stream = torch.cuda.Stream(device=device) if (self.comm_split_tokens and comm_overlap) else None
        D = cfg.head_dim
        self.hq = heads_per_rank(cfg.num_qo_heads, tp_size)
        self.hkv = heads_per_rank(cfg.num_kv_heads, tp_size, replicate=True)
        self.inter = heads_per_rank(cfg.intermediate_size, tp_size)
        self.vocab_tp, self.vocab_range = vocab_shard(cfg.vocab_size, tp_size, tp_rank)
        self.q_dim, self.kv_dim = self.hq * D, self.hkv * D

        g = torch.Generator(device=device)
        g.manual_seed(seed + 1000 * tp_rank)

        def w(*shape):  # seeded N(0, std^2): no checkpoints exist offline (use_dummy_weight analogue)
            return (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * init_std).to(dtype)

        def ones(n):
            return torch.ones(n, device=device, dtype=dtype)

        H = cfg.hidden_size
        self.embed = w(self.vocab_tp, H)
        self.layers: List[LayerWeights] = []
        for _ in range(cfg.num_layers):
            self.layers.append(LayerWeights(
                input_norm=ones(H), qkv=w(self.q_dim + 2 * self.kv_dim, H),
                q_norm=ones(D) if cfg.qk_norm else None, k_norm=ones(D) if cfg.qk_norm else None,
                o=w(H, self.q_dim), post_norm=ones(H), gate_up=w(2 * self.inter, H), down=w(H, self.inter)))
        # gate_up rows in the block-32 interleaved order the fused projection + SiLU.mul epilogue needs (csrc/gemm_g3.hip);
        # every consumer goes through ops.linear_silu, the oracle gets the reference layout back (gate_up_reference)
        self.gate_up_ilv = bool(fused and _FUSE_SILU and device.type == "cuda" and self.inter % 64 == 0)
        if self.gate_up_ilv:
            idx = ops.gate_up_interleave_index(self.inter, device)
            for lw in self.layers:
                lw.gate_up = lw.gate_up.index_select(0, idx)
        self.final_norm = ones(H)
        self.lm_head = self.embed if cfg.tie_word_embeddings else w(self.vocab_tp, H)
        self.cos_sin = fi.build_cos_sin_cache(D, cfg.max_position, cfg.rope_base, cfg.rope_scaling, device=device)

    def weight_bytes(self) -> int:
        n = self.embed.numel() + self.final_norm.numel()
        if not self.cfg.tie_word_embeddings:
            n += self.lm_head.numel()
        for lw in self.layers:
            n += sum(t.numel() for t in (lw.input_norm, lw.qkv, lw.o, lw.post_norm, lw.gate_up, lw.down))
            n += 0 if lw.q_norm is None else lw.q_norm.numel() + lw.k_norm.numel()
        return n * self.embed.element_size()

    def streamed_bytes_per_step(self) -> int:
        """Weight bytes a decode step must read: everything but the embedding table (gathered rows only)."""
        b = self.weight_bytes()
        if not self.cfg.tie_word_embeddings:
            b -= self.embed.numel() * self.embed.element_size()
        return b

    # ------------------------------------------------------------------ weights from a checkpoint
    def load_hf_state(self, state: Dict[str, torch.Tensor]) -> None:
        """Overwrite the seeded weights with HF-named tensors (full, unsharded), sharding and merging as
        P/models/weight.py:34-124 does: q/k/v -> qkv, gate/up -> gate_up, column shards on dim 0, row shards
        (o, down) on dim 1, vocab shards of embed_tokens / lm_head, KV heads replicated when tp > Hkv."""
        cfg, r, n, D = self.cfg, self.tp_rank, self.tp_size, self.cfg.head_dim

        def col(t):
            return t.chunk(n, dim=0)[r]

        def kv(t):
            if cfg.num_kv_heads < n:
                h = r * cfg.num_kv_heads // n
                return t[h * D:(h + 1) * D]
            return col(t)

        def put(dst: torch.Tensor, src: torch.Tensor) -> None:
            assert dst.shape == src.shape, (dst.shape, src.shape)
            dst.copy_(src.to(device=self.device, dtype=self.dtype))

        start, length = self.vocab_range
        put(self.embed[:length], state["model.embed_tokens.weight"][start:start + length])
        if not cfg.tie_word_embeddings:
            put(self.lm_head[:length], state["lm_head.weight"][start:start + length])
        put(self.final_norm, state["model.norm.weight"])
        for i, lw in enumerate(self.layers):
            p = f"model.layers.{i}."
            put(lw.input_norm, state[p + "input_layernorm.weight"])
            put(lw.post_norm, state[p + "post_attention_layernorm.weight"])
            put(lw.qkv, torch.cat([col(state[p + "self_attn.q_proj.weight"]), kv(state[p + "self_attn.k_proj.weight"]),
                                   kv(state[p + "self_attn.v_proj.weight"])], dim=0))
            if lw.q_norm is not None:
                put(lw.q_norm, state[p + "self_attn.q_norm.weight"])
                put(lw.k_norm, state[p + "self_attn.k_norm.weight"])
            put(lw.o, state[p + "self_attn.o_proj.weight"].chunk(n, dim=1)[r])
            gu = torch.cat([col(state[p + "mlp.gate_proj.weight"]), col(state[p + "mlp.up_proj.weight"])], dim=0)
            put(lw.gate_up, gu.index_select(0, ops.gate_up_interleave_index(self.inter, gu.device)) if self.gate_up_ilv
                else gu)
            put(lw.down, state[p + "mlp.down_proj.weight"].chunk(n, dim=1)[r])

    def gate_up_reference(self, layer: int) -> torch.Tensor:
        """Layer `layer`'s gate_up weight in the reference's row order [gate; up] (P/layers/linear.py:56-71)."""
        w = self.layers[layer].gate_up
        if not self.gate_up_ilv:
            return w
        idx = ops.gate_up_interleave_index(self.inter, w.device)
        inv = torch.empty_like(idx)
        inv[idx] = torch.arange(idx.numel(), device=w.device)
        return w.index_select(0, inv)

    # ------------------------------------------------------------------ GEMM solution search
    def projection_groups(self):
        """(name, same-shaped weights of up to 8 layers, K) of the five projection shapes."""
        step = max(1, len(self.layers) // 8)
        pick = self.layers[::step][:8]
        # "fold": the row kernel forward() folds into the projection where the row-streaming kernel is planned (batches <= 8)
        fold = self.fused and self.tp_size == 1 and _ROWSTREAM_FUSE
        norm = {"fold": "norm"} if fold else {}
        act = {"fold": "act_interleaved" if self.gate_up_ilv else "act"} if fold else {}
        return [("qkv", [l.qkv for l in pick], self.cfg.hidden_size, norm), ("o", [l.o for l in pick], self.q_dim),
                ("gate_up", [l.gate_up for l in pick], self.cfg.hidden_size, {"silu_interleaved": self.gate_up_ilv, **norm}),
                ("down", [l.down for l in pick], self.inter, act), ("lm_head", [self.lm_head], self.cfg.hidden_size, norm)]

    def tune_gemms(self, batch_sizes: List[int], mode: str = "heuristic", log=None, prefill_tokens=()) -> List[dict]:
        from .gemm_plan import tune_prefill_gemms, tune_projection_gemms

        report = tune_projection_gemms(self.projection_groups(), batch_sizes, mode, self.dtype, self.device, log=log)
        if mode != "off" and prefill_tokens:  # after the decode search: "off" resets every plan
            report += tune_prefill_gemms(self.projection_groups(), prefill_tokens, self.dtype, self.device, log=log)
        return report

    # ------------------------------------------------------------------ row-parallel projection + all-reduce
    def row_parallel(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """all_reduce(x @ w^T) (P/layers/linear.py:102-106, 123-127).  north_star: "RCCL all-reduce over xGMI overlapped
        on a side HIP stream": with >= comm_split_tokens tokens the projection runs per token half; the first half's
        collective goes to the side stream (its own communicator) while the compute stream runs the second half's
        GEMM, the second half's collective follows on the compute stream, which then waits for the side stream.
        comm_overlap=False issues the very same kernels on one stream (the serial path the overlap must equal)."""
        T = x.shape[0]
        if self.tp_size == 1:
            # the caller's next operation is fused_add_rmsnorm: a k-sliced full-batch plan leaves its reduce to it
            if not _SLAB_NORM:
                return ops.linear(x, w)
            y, slabs = ops.linear_slabs(x, w)
            if slabs is not None:
                y._msgl_slabs = slabs
            return y
        if not self.comm_split_tokens or T < self.comm_split_tokens:
            return self.comm.all_reduce(ops.linear(x, w))
        h = (T // 2 + 7) // 8 * 8
        y = torch.empty((T, w.shape[0]), dtype=x.dtype, device=x.device)
        ops.linear(x[:h], w, out=y[:h])
        if self.comm_overlap:
            main, side = torch.cuda.current_stream(), self.side_stream
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                self.comm.all_reduce_side(y[:h])
                done = side.record_event()
            ops.linear(x[h:], w, out=y[h:])
            self.comm.all_reduce(y[h:])
            main.wait_event(done)
        else:
            self.comm.all_reduce_side(y[:h])
            ops.linear(x[h:], w, out=y[h:])
            self.comm.all_reduce(y[h:])
        return y

    def row_parallel_norm(self, x: torch.Tensor, w: torch.Tensor, residual: torch.Tensor, norm_w: torch.Tensor) -> torch.Tensor:
        """fused_add_rmsnorm(all_reduce(x @ w^T), residual, norm_w) (P/models/qwen3.py:36-41): at tp = 1 the projection's
        split-K reduce joins the norm (slab hand-off); at tp > 1 a decode-size batch takes the fused all-reduce + add +
        norm launch of the peer-to-peer communicator, larger ones the (overlapped) all-reduce and then the norm."""
        eps = self.cfg.rms_norm_eps
        if self.tp_size > 1 and not (self.comm_split_tokens and x.shape[0] >= self.comm_split_tokens):
            return self.comm.all_reduce_add_rmsnorm(ops.linear(x, w), residual, norm_w, eps)
        y = self.row_parallel(x, w)
        fi.fused_add_rmsnorm(y, residual, norm_w, eps)
        return y

    # ------------------------------------------------------------------ forward
    def forward(self, ctx: Any, batch: Any, logits_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """P/models/qwen3.py:77-81 -> logits [B, vocab] (model dtype); written into `logits_out` (rows x vocab, model
        dtype) when given -- the graph runner's static buffer."""
        cfg, D = self.cfg, self.cfg.head_dim
        backend, kv = ctx.attn_backend, ctx.kv_cache
        x = ops.embedding_gather(self.embed, batch.input_ids,
                                 vocab_range=self.vocab_range if self.tp_size > 1 else None)
        x = self.comm.all_reduce(x)
        residual: Optional[torch.Tensor] = None
        normed_ahead = False  # the previous layer's down_proj already ran this layer's input norm (row_parallel_norm)
        # Batches <= 8 (tp = 1): where a projection's plan is the row-streaming kernel (csrc/gemm_rowstream.hip) the row
        # kernel in front of it rides in that kernel's staging pass -- fused_add_rmsnorm in front of qkv / gate_up / lm_head
        # (`pending` = the previous projection's output still waiting for its add + norm; the new residual goes to the
        # other one of two buffers, every workgroup is still reading the old one), SiLU.mul in front of down_proj.  Same bits.
        # Measured inside the captured step (tools/small_batch_ab.py, profiles/r04_small_batch_ab.json): folding the norms is
        # worth 0.13 / 0.07-0.11 ms per step at one / two rows and nothing at four; SiLU.mul pays at one row only.
        T = x.shape[0]
        rs_on = _ROWSTREAM_FUSE and self.fused and self.tp_size == 1 and T <= ops.ROWSTREAM_FOLD_NORM_MAX_M
        pending: Optional[torch.Tensor] = None
        spare: Optional[torch.Tensor] = None

        def norm_into(w: torch.Tensor, y: torch.Tensor, norm_w: torch.Tensor, plan, out=None) -> torch.Tensor:
            nonlocal residual, spare
            if spare is None:
                spare = torch.empty_like(residual)
            r = ops.rowstream_linear(y, w, plan[0], out=out, mode=ops.ROWSTREAM_ADD_NORM, res_in=residual, res_out=spare,
                                     gamma=norm_w, eps=cfg.rms_norm_eps, variant=plan[1])
            residual, spare = spare, residual
            return r

        for li, lw in enumerate(self.layers):
            if residual is None:  # P/layers/norm.py:35-36
                residual = x
                x = fi.rmsnorm(x, lw.input_norm, cfg.rms_norm_eps)
            elif not normed_ahead:
                fi.fused_add_rmsnorm(x, residual, lw.input_norm, cfg.rms_norm_eps)
            if pending is not None:  # the previous layer's down_proj output: add + input norm in the qkv launch
                qkv, slabs = norm_into(lw.qkv, pending, lw.input_norm, ops.rowstream_planned(T, lw.qkv, ops.ROWSTREAM_ADD_NORM)), None
                pending = None
            else:
                # a k-sliced full-batch plan leaves the qkv projection's reduce to the fused norm / RoPE / store pass
                qkv, slabs = ops.linear_slabs(x, lw.qkv) if self.fused and _SLAB_NORM else (ops.linear(x, lw.qkv), None)
            q, k, v = qkv.split([self.q_dim, self.kv_dim, self.kv_dim], dim=-1)
            if self.fused:
                kc, vc = kv.k_cache(li), kv.v_cache(li)
                if slabs is not None:
                    ops.qk_norm_rope_store_slabs(qkv, slabs, self.hq, self.hkv, lw.q_norm, lw.k_norm, cfg.rms_norm_eps,
                                                 batch.positions, self.cos_sin, kc.view(-1, self.kv_dim),
                                                 vc.view(-1, self.kv_dim), batch.out_loc, D)
                else:
                    ops.qk_norm_rope_store(q, k, v, lw.q_norm, lw.k_norm, cfg.rms_norm_eps, batch.positions,
                                           self.cos_sin, kc.view(-1, self.kv_dim), vc.view(-1, self.kv_dim),
                                           batch.out_loc, D)
                o = backend.attend(q.view(-1, self.hq, D), li, batch)
            else:  # P/layers/attention.py:47-57
                if lw.q_norm is not None:
                    fi.rmsnorm(q.view(-1, self.hq, D), lw.q_norm, cfg.rms_norm_eps, out=q.view(-1, self.hq, D))
                    fi.rmsnorm(k.view(-1, self.hkv, D), lw.k_norm, cfg.rms_norm_eps, out=k.view(-1, self.hkv, D))
                fi.apply_rope_with_cos_sin_cache_inplace(positions=batch.positions, query=q, key=k, head_size=D,
                                                         cos_sin_cache=self.cos_sin)
                o = backend.forward(q.view(-1, self.hq, D), k, v, li, batch)
            # down_proj feeds the NEXT layer's input norm (or the final norm): its all-reduce / slab reduce joins that norm
            last = li + 1 == len(self.layers)
            nxt = self.final_norm if last else self.layers[li + 1].input_norm
            d_gu = ops.rowstream_planned(T, lw.gate_up, ops.ROWSTREAM_ADD_NORM) if rs_on else None
            act_mode = ops.ROWSTREAM_SILU_INTERLEAVED if self.gate_up_ilv else ops.ROWSTREAM_SILU
            # SiLU.mul in the staging pass pays at one row only (M = 4: 43 vs 39 us for the pair, profiles/r04_rowstream_bench.json)
            d_dn = ops.rowstream_planned(T, lw.down, act_mode) if rs_on and T <= ops.ROWSTREAM_FOLD_ACT_MAX_M else None
            if d_gu:  # o_proj, then post-attention add + norm inside the gate_up launch
                gu = norm_into(lw.gate_up, ops.linear(o.view(-1, self.q_dim), lw.o), lw.post_norm, d_gu)
            else:
                x = self.row_parallel_norm(o.view(-1, self.q_dim), lw.o, residual, lw.post_norm)
                gu = None
            # the next consumer of the residual stream (next layer's qkv, or the LM head of a decode batch) takes down_proj's
            # add + norm into its own launch if it can
            fold_next = (ops.rowstream_planned(T, self.lm_head if last else self.layers[li + 1].qkv, ops.ROWSTREAM_ADD_NORM)
                         if rs_on and not (last and batch.is_prefill) else None)
            if d_dn:  # SiLU.mul while down_proj stages its input
                if gu is None:
                    gu = ops.linear(x, lw.gate_up)
                x = ops.rowstream_linear(gu, lw.down, d_dn[0], mode=act_mode, variant=d_dn[1])
            else:
                if gu is not None:
                    y = ops.silu_and_mul_interleaved(gu) if self.gate_up_ilv else fi.silu_and_mul(gu)
                elif self.gate_up_ilv:  # projection + SiLU.mul: one launch where planned (P/models/utils.py:45-51)
                    y = ops.linear_silu(x, lw.gate_up)
                else:
                    y = fi.silu_and_mul(ops.linear(x, lw.gate_up))
                x = ops.linear(y, lw.down) if fold_next else self.row_parallel_norm(y, lw.down, residual, nxt)
            if fold_next:
                pending = x
            elif d_dn:
                fi.fused_add_rmsnorm(x, residual, nxt, cfg.rms_norm_eps)
            normed_ahead = True
        # LM head (P/layers/embedding.py:88-110)
        bs = batch.size
        if pending is not None:  # decode batch: the final add + norm inside the LM head's launch
            out = logits_out if logits_out is not None and logits_out.shape == (T, self.lm_head.shape[0]) else None
            return norm_into(self.lm_head, pending, self.final_norm, ops.rowstream_planned(T, self.lm_head, ops.ROWSTREAM_ADD_NORM), out)
        if batch.is_prefill:
            x = x[batch.attn_metadata.get_last_indices(bs)].contiguous()
        if self.tp_size == 1:
            if logits_out is not None and logits_out.shape == (x.shape[0], self.lm_head.shape[0]):
                return ops.linear(x, self.lm_head, out=logits_out)
            return ops.linear(x, self.lm_head)
        logits = ops.linear(x, self.lm_head)
        full = lm_head_unshard(self.comm.all_gather(logits), self.tp_size, logits.shape[0], cfg.vocab_size)
        if logits_out is not None:
            logits_out.copy_(full)
            return logits_out
        return full
