"""The smallest decode batches (1 .. 4 sequences) of a Qwen3-14B-dimension decoder with every projection on the
row-streaming kernel (csrc/gemm_rowstream.hip) and the layer's row kernels folded into its staging pass
(model.DenseDecoder.forward: at one or two rows fused_add_rmsnorm in front of qkv / gate_up / lm_head, at one row SiLU.mul in
front of down_proj):

  * the folded forward equals the unfolded one BIT FOR BIT (logits and KV pool), eager and under graph replay;
  * both are the reference's composition (P/models/qwen3.py:18-81) within the bf16 band of tests/test_gpu_model_14b.py
    against oracle/ref_model.py, teacher-forced on the recorded batches;
  * the folding really happens: no fused_add_rmsnorm launch is left in a decode forward, and no activation launch at one row.
"""
import random

import pytest
import torch

import parity_stats
from oracle import ref_model

pytestmark = pytest.mark.gpu


def _cfg(layers):
    from mini_sglang_amd.model import PRESETS, ModelConfig

    m = PRESETS["qwen3-14b"]
    return ModelConfig(layers, m.num_qo_heads, m.num_kv_heads, m.head_dim, m.hidden_size, m.vocab_size,
                       m.intermediate_size, name="Qwen3-14B dims, reduced depth")


def _plan_rowstream(ops, model, batch_sizes, depth, variant):
    """What the search does when the row-streaming kernel wins a shape: record (-variant, depth) for it."""
    from mini_sglang_amd import _lib

    ws = [model.lm_head] + [w for lw in model.layers for w in (lw.qkv, lw.o, lw.gate_up, lw.down)]
    n = 0
    for M in batch_sizes:
        for w in ws:
            N, K = w.shape
            if ops.rowstream_supported(M, N, K, 0, variant):
                ops._SKINNY_PLAN[(w.device.index or 0, M, N, K, K, w.stride(0), _lib.BF16)] = (-variant, depth)
                n += 1
    return n


@pytest.mark.parametrize("variant", [0, 1], ids=["vector-units", "matrix-cores"])
def test_small_decode_batches_rowstream_folded_equals_unfolded_and_matches_oracle(dev, monkeypatch, variant):
    from mini_sglang_amd import flashinfer_compat as fi
    from mini_sglang_amd import model as model_mod
    from mini_sglang_amd import ops
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.offline import OfflineRunner
    from replay_util import record_offline_runner, replay_forward

    layers = 2
    m = _cfg(layers)
    cfg = EngineConfig(model=m, dtype=torch.bfloat16, max_running_req=4, page_size=16, cuda_graph_bs=[1, 2, 4],
                       max_seq_len_override=512, num_page_override=256, seed=7, gemm_tune="off")
    eng = Engine(cfg, dev)
    try:
        assert not ops._SKINNY_PLAN
        rnd = random.Random(3)
        eng.kv_cache.pool.zero_()
        rec = []
        runner = OfflineRunner(eng, max_extend_tokens=64, seed=1)
        record_offline_runner(runner, eng, rec)
        for n_req in (3, 1, 2):  # decode batches of 3 (padded to the 4-graph), 1, 2; the prefills carry 3 .. 40 tokens
            prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(1, 14))] for _ in range(n_req)]
            runner.generate(prompts, [SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True) for _ in prompts])
        decodes = [f for f in rec if f["phase"] == "decode"]
        assert {f["size"] for f in decodes} == {1, 2, 3} and all(f["graph"] for f in decodes)

        def run_all(fold: bool):
            """Every recorded forward again (KV pool from zero), decode graphs re-captured with the plans in force."""
            monkeypatch.setattr(model_mod, "_ROWSTREAM_FUSE", fold)
            eng.kv_cache.pool.zero_()
            for bs in (4, 2, 1):
                eng.graph_runner.capture(bs)
            out = [replay_forward(eng, f).float().cpu() for f in rec]
            torch.cuda.synchronize()
            return out, eng.kv_cache.pool.clone()

        base, _ = run_all(False)  # library GEMMs (no plans), separate row kernels: the recorded run itself
        for f, lg in zip(rec, base):
            assert torch.equal(lg, f["logits"])
        planned = _plan_rowstream(ops, eng.model, (1, 2, 3, 4), 16, variant)
        assert planned == 4 * (1 + 4 * layers)
        unfolded, unfolded_pool = run_all(False)
        folded, folded_pool = run_all(True)
        # count the row-kernel launches of one eager decode forward, folded and not
        calls = {"norm": 0, "act": 0}
        real_norm, real_act = fi.fused_add_rmsnorm, ops.silu_and_mul_interleaved
        monkeypatch.setattr(fi, "fused_add_rmsnorm", lambda *a, **k: (calls.__setitem__("norm", calls["norm"] + 1), real_norm(*a, **k))[1])
        monkeypatch.setattr(ops, "silu_and_mul_interleaved", lambda *a, **k: (calls.__setitem__("act", calls["act"] + 1), real_act(*a, **k))[1])
        monkeypatch.setattr(eng.graph_runner, "can_use_cuda_graph", lambda batch: False)
        # the LAST recorded forward: its pages are still what they were (earlier requests' pages have been recycled since)
        assert rec[-1] is decodes[-1]
        one = dict(decodes[-1], graph=False)
        monkeypatch.setattr(model_mod, "_ROWSTREAM_FUSE", True)
        eager_folded = replay_forward(eng, one).float().cpu()
        # a batch of two: the norms are folded, SiLU.mul (folded at one row only) still is a launch per layer
        n_act = layers if eng.model.gate_up_ilv else 0  # (fi.silu_and_mul is not counted)
        assert one["input_ids"].numel() == 2 and calls == {"norm": 0, "act": n_act}, calls
        monkeypatch.setattr(model_mod, "_ROWSTREAM_FUSE", False)
        eager_unfolded = replay_forward(eng, one).float().cpu()
        assert calls == {"norm": 2 * layers, "act": 2 * n_act}, calls
        single = next(f for f in reversed(rec) if f["phase"] == "decode" and f["input_ids"].numel() == 1)
        monkeypatch.setattr(model_mod, "_ROWSTREAM_FUSE", True)
        replay_forward(eng, dict(single, graph=False))  # one row: nothing is left outside the projections
        assert calls == {"norm": 2 * layers, "act": 2 * n_act}, calls
        assert torch.equal(eager_folded, eager_unfolded)
        for i, (f, a, b) in enumerate(zip(rec, unfolded, folded)):
            assert torch.equal(a, b), (i, f["phase"], f["size"])
        assert torch.equal(unfolded_pool, folded_pool)
        assert torch.equal(eager_folded, folded[-1])  # eager == graph replay

        # teacher-forced against the oracle (same band as the full-batch test at these dims)
        w = ref_model.weights_from_device_model(eng.model)
        table = eng.page_table.cpu()
        slots = eng.kv_cache.pool.shape[2] * eng.kv_cache.pool.shape[3]
        kp = [torch.zeros((slots, m.num_kv_heads, m.head_dim), dtype=torch.bfloat16) for _ in range(layers)]
        vp = [torch.zeros_like(k) for k in kp]
        stats, stats_base = {}, {}
        for i, f in enumerate(rec):
            k_lens, q_lens = f["device_lens"], [d - c for d, c in zip(f["device_lens"], f["cached_lens"])]
            tb = torch.zeros_like(table)  # pages are recycled between the three generate() calls: this forward's rows
            tb[torch.tensor(f["rows"]), : f["table"].shape[1]] = f["table"]
            want = ref_model.forward(m, w, f["input_ids"], f["positions"], f["out_loc"], kp, vp, tb, f["rows"], k_lens, q_lens,
                                     f["phase"] == "prefill").float()[: f["size"]]
            st = parity_stats.logit_error_stats(folded[i], want)
            stats = parity_stats.merge_stats(stats, st)
            stats_base = parity_stats.merge_stats(stats_base, parity_stats.logit_error_stats(base[i], want))
            assert st["max_abs"] <= 1.5e-1, (i, parity_stats.fmt(st))
        print(f"\n[14B dims, {layers} layers, batches of 1-3] row-streaming + folded: {parity_stats.fmt(stats)}")
        print(f"[14B dims, {layers} layers, batches of 1-3] library GEMMs, separate row kernels: {parity_stats.fmt(stats_base)}")
        assert stats["mean_abs"] <= max(2e-2, 1.25 * stats_base["mean_abs"])
    finally:
        eng.shutdown()
        ops.reset_gemm_plans()


@pytest.mark.parametrize("kind", ["qkv", "gate_up", "down", "lm_head", "qkv+down", "gate_up+lm_head"])
def test_rowstream_folds_with_mixed_plans_equal_unfolded(dev, monkeypatch, kind):
    """The search plans the row-streaming kernel per SHAPE, so a layer usually has it for some projections only: every
    mixed state of model.DenseDecoder.forward's fold logic (only qkv, only gate_up, only down_proj -- SiLU.mul folded but
    the add + norm left to its own launch -- only the LM head, and two pairs) must equal the unfolded forward bit for bit,
    logits and KV pool, eager decode at one and two rows."""
    from mini_sglang_amd import _lib
    from mini_sglang_amd import model as model_mod
    from mini_sglang_amd import ops
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.offline import OfflineRunner
    from replay_util import record_offline_runner, replay_forward

    layers = 2
    m = _cfg(layers)
    cfg = EngineConfig(model=m, dtype=torch.bfloat16, max_running_req=2, page_size=16, cuda_graph_bs=[],
                       max_seq_len_override=256, num_page_override=64, seed=11, gemm_tune="off")
    eng = Engine(cfg, dev)
    try:
        rnd = random.Random(5)
        rec = []
        runner = OfflineRunner(eng, max_extend_tokens=64, seed=1)
        record_offline_runner(runner, eng, rec)
        for n_req in (1, 2):
            prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(2, 9))] for _ in range(n_req)]
            runner.generate(prompts, [SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True) for _ in prompts])
        assert {f["size"] for f in rec if f["phase"] == "decode"} == {1, 2}
        chosen = kind.split("+")
        mdl = eng.model
        group = {"qkv": [lw.qkv for lw in mdl.layers], "gate_up": [lw.gate_up for lw in mdl.layers],
                 "down": [lw.down for lw in mdl.layers], "lm_head": [mdl.lm_head]}
        for M in (1, 2):
            for k in chosen:
                for w in group[k]:
                    N, K = w.shape
                    assert ops.rowstream_supported(M, N, K, 0, 1)
                    ops._SKINNY_PLAN[(w.device.index or 0, M, N, K, K, w.stride(0), _lib.BF16)] = (-1, 16)

        def run_all(fold: bool):
            monkeypatch.setattr(model_mod, "_ROWSTREAM_FUSE", fold)
            eng.kv_cache.pool.zero_()
            out = [replay_forward(eng, dict(f, graph=False)).float().cpu() for f in rec]
            torch.cuda.synchronize()
            return out, eng.kv_cache.pool.clone()

        unfolded, pool_u = run_all(False)
        folded, pool_f = run_all(True)
        for i, (f, a, b) in enumerate(zip(rec, unfolded, folded)):
            assert torch.isfinite(a).all() and torch.equal(a, b), (kind, i, f["phase"], f["size"])
        assert torch.equal(pool_u, pool_f)
    finally:
        eng.shutdown()
        ops.reset_gemm_plans()
