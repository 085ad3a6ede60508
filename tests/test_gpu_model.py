"""End-to-end on one MI355X: engine + offline driver on a tiny Qwen3-shaped model.

* fused (qk_norm_rope_store + attend) vs the reference op order through the seams: token ids,
  logits and the whole KV pool must be bit-identical;
* hipGraph replay vs eager: identical greedy ids;
* teacher-forced parity vs the CPU oracle (oracle/ref_model.py): logits within tolerance at
  every step, greedy ids equal wherever the oracle's top-1/top-2 margin exceeds the tolerance
  (GEMM accumulation order alone can flip near-ties).
"""
import pytest
import torch

import parity_stats
from oracle import ref_model, ref_ops, torch_bf16

pytestmark = pytest.mark.gpu

# north_star: "bf16 logits within 1e-3".  One bf16 ulp is 2^-8 |x| ... 2^-7 |x|: 1e-3 is below one ulp for every
# |logit| > 0.25, and the GPU pipeline rounds to bf16 at the same 9 points per layer as the oracle but sums its GEMMs
# in a different order, so single-ulp flips of intermediate activations are unavoidable and propagate.  What is
# asserted instead (error distribution printed by every test below): per OP on the oracle's own input <= 2 ulp
# (test_op_chain...), and end to end a bound set from the measured distribution: tiny models (2 layers, logit std
# 0.33) max 7.8e-3 / p99 4.9e-3 / mean 1.1e-3 = 1.9 ulp => LOGIT_TOL 1.6e-2; Qwen3-0.6B dims (28 layers, logit std
# 0.64) see the config-0 test.  1e-3 itself holds for 60 % of the tiny-model logits (frac > 1e-3 = 0.40).
LOGIT_TOL = 1.6e-2


def make_engine(dev, name="tiny", fused=True, graphs=True, page_size=4, seed=42):
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS

    cfg = EngineConfig(model=PRESETS[name], dtype=torch.bfloat16, max_running_req=16, page_size=page_size,
                       cuda_graph_bs=[1, 2, 4, 8] if graphs else [], max_seq_len_override=512,
                       num_page_override=4096 // page_size, fused_qkv_path=fused, seed=seed)
    eng = Engine(cfg, dev)
    eng.kv_cache.pool.zero_()  # torch.empty pool: make untouched slots comparable across engines
    return eng


def prompts(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    lens = [5, 17, 64, 33, 1, 100, 129, 48][:n]
    return [torch.randint(0, 1000, (l,), generator=g).tolist() for l in lens]


def run(engine, ps, max_tokens, record=None, max_extend=8192):
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.offline import OfflineRunner

    runner = OfflineRunner(engine, max_extend_tokens=max_extend, seed=1)
    if record is not None:
        orig_fwd, orig_sample = runner._forward, engine.sampler.sample

        def fwd(batch, write, args):
            md = batch.attn_metadata
            record.append(dict(phase=batch.phase, input_ids=batch.input_ids.cpu(), positions=batch.positions.cpu(),
                               out_loc=batch.out_loc.cpu(), rows=[r.table_idx for r in batch.padded_reqs],
                               k_lens=[r.device_len for r in batch.padded_reqs],
                               q_lens=[r.extend_len for r in batch.padded_reqs], size=batch.size))
            return orig_fwd(batch, write, args)

        def sample(logits, args):
            record[-1]["logits"] = logits.float().cpu()
            return orig_sample(logits, args)

        runner._forward, engine.sampler.sample = fwd, sample
    sp = [SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True) for _ in ps]
    stats = runner.generate(ps, sp)
    ids = [runner.output_ids(s) for s in runner.last_states]
    return ids, stats, runner


@pytest.mark.parametrize("name", ["tiny", "tiny-llama"])
def test_fused_path_is_bit_identical_to_reference_op_order(dev, name):
    ps = prompts(6)
    e1 = make_engine(dev, name, fused=True, graphs=False)
    ids1, _, _ = run(e1, ps, 6)
    kv1 = e1.kv_cache.pool.clone()
    e1.shutdown()
    e2 = make_engine(dev, name, fused=False, graphs=False)
    ids2, _, _ = run(e2, ps, 6)
    assert ids1 == ids2
    assert torch.equal(kv1[:, :, :-1].cpu(), e2.kv_cache.pool[:, :, :-1].cpu())  # all but the dummy page
    e2.shutdown()


def test_graph_replay_matches_eager(dev):
    ps = prompts(7)
    e1 = make_engine(dev, graphs=True)
    ids1, st, _ = run(e1, ps, 12)
    assert st["decode_steps"] == 11 and all(len(i) == 12 for i in ids1)
    e1.shutdown()
    e2 = make_engine(dev, graphs=False)
    ids2, _, _ = run(e2, ps, 12)
    e2.shutdown()
    assert ids1 == ids2


def test_chunked_prefill_matches_unchunked(dev):
    ps = prompts(8)
    e1 = make_engine(dev, graphs=False)
    ids1, _, _ = run(e1, ps, 4)
    e1.shutdown()
    e2 = make_engine(dev, graphs=False)
    ids2, _, _ = run(e2, ps, 4, max_extend=50)  # forces q_len < k_len continuation chunks
    e2.shutdown()
    assert ids1 == ids2


def test_prefill_batch_of_one_token_extensions_runs_eagerly_with_graphs_captured(dev):
    """Round-1 crash (ADVICE high, attention.py:149): a PREFILL-phase batch whose every request extends by one
    token (chunk remainder of 1 here; a radix full hit in the reference) at a captured batch size took the decode
    kernel with no plan.  prompt_len = budget + 1 => the second chunk is exactly that batch."""
    g = torch.Generator().manual_seed(5)
    ps = [torch.randint(0, 1000, (65,), generator=g).tolist()]
    e1 = make_engine(dev, graphs=True)
    rec = []
    ids1, _, _ = run(e1, ps, 5, record=rec, max_extend=64)
    e1.shutdown()
    assert [r["phase"] for r in rec[:2]] == ["prefill", "prefill"] and rec[1]["q_lens"] == [1] and rec[1]["k_lens"] == [65]
    e2 = make_engine(dev, graphs=True)
    rec2 = []
    ids2, _, _ = run(e2, ps, 5, record=rec2)  # unchunked
    e2.shutdown()
    # The one-token chunk goes through the DECODE kernel, the unchunked prompt's last token through the PREFILL kernel: two
    # summation orders of the same attention, so the runs agree within the logit tolerance, not bit for bit, and the greedy
    # continuation is the same until a near-tie (round 6: the matrix-core decode kernel now also serves this model's
    # 4-token pages and flips one such tie that the streaming kernel happened not to).  Compared forward by forward up to the
    # first divergence, which must BE a near-tie in both runs.
    for a, b in zip(rec[1:], rec2):
        la, lb = a["logits"][0], b["logits"][0]
        assert (la - lb).abs().max().item() <= LOGIT_TOL
        ia, ib = int(la.argmax()), int(lb.argmax())
        if ia != ib:
            assert abs(float(lb[ib] - lb[ia])) <= 2 * LOGIT_TOL and abs(float(la[ia] - la[ib])) <= 2 * LOGIT_TOL
            break
    else:
        assert ids1 == ids2


@pytest.mark.parametrize("name,page_size", [("tiny", 4), ("tiny-llama", 1)])
def test_teacher_forced_parity_vs_cpu_oracle(dev, name, page_size):
    ps = prompts(5)
    rec = []
    eng = make_engine(dev, name, fused=True, graphs=True, page_size=page_size)
    ids, _, runner = run(eng, ps, 5, record=rec)
    cfg = eng.cfg.model
    w = ref_model.weights_from_device_model(eng.model)
    table = eng.page_table.cpu()
    slots = eng.kv_cache.pool.shape[2] * eng.kv_cache.pool.shape[3]
    kp = [torch.zeros((slots, cfg.num_kv_heads, cfg.head_dim), dtype=torch.bfloat16) for _ in range(cfg.num_layers)]
    vp = [torch.zeros_like(k) for k in kp]
    agree = total = floor_agree = cfloor_agree = 0
    stats, floor, cfloor = {}, {}, {}
    # the independent bf16 forward (plain torch ops on the GPU, oracle/torch_bf16.py): where a bf16 pipeline sits; and the same
    # forward on the CPU with fp32-accumulating matmuls of the bf16 values (no device library in it)
    wd, table_d = torch_bf16.weights_to(w, dev), table.to(dev)
    kpd = [torch.zeros_like(k, device=dev) for k in kp]
    vpd = [torch.zeros_like(k) for k in kpd]
    kpc, vpc = [torch.zeros_like(k) for k in kp], [torch.zeros_like(k) for k in kp]
    for r in rec:
        logits = ref_model.forward(cfg, w, r["input_ids"], r["positions"], r["out_loc"], kp, vp, table, r["rows"],
                                   r["k_lens"], r["q_lens"], r["phase"] == "prefill").float()[: r["size"]]
        tb = torch_bf16.forward(cfg, wd, r["input_ids"].to(dev), r["positions"].to(dev), r["out_loc"].to(dev), kpd, vpd, table_d,
                                r["rows"], r["k_lens"], r["q_lens"], r["phase"] == "prefill").float().cpu()[: r["size"]]
        floor = parity_stats.merge_stats(floor, parity_stats.logit_error_stats(tb, logits))
        floor_agree += int((tb.argmax(-1) == logits.argmax(-1)).sum())
        tc = torch_bf16.forward(cfg, w, r["input_ids"], r["positions"], r["out_loc"], kpc, vpc, table, r["rows"], r["k_lens"],
                                r["q_lens"], r["phase"] == "prefill", linear="fp32acc").float()[: r["size"]]
        cfloor = parity_stats.merge_stats(cfloor, parity_stats.logit_error_stats(tc, logits))
        cfloor_agree += int((tc.argmax(-1) == logits.argmax(-1)).sum())
        st = parity_stats.logit_error_stats(r["logits"], logits)
        stats = parity_stats.merge_stats(stats, st)
        assert st["max_abs"] <= LOGIT_TOL, parity_stats.fmt(st)
        top2 = logits.topk(2, dim=-1).values
        sure = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL
        same = r["logits"].argmax(-1) == logits.argmax(-1)
        assert bool(same[sure].all())
        agree += int(same.sum())
        total += same.numel()
    print(f"\n[teacher-forced {name} page {page_size}] {parity_stats.fmt(stats)}; argmax agreement {agree}/{total}")
    assert agree >= 0.9 * total and stats["p99_abs"] <= LOGIT_TOL / 2
    parity_stats.assert_not_above_bf16_floor(f"{name}, {cfg.num_layers} layers", stats, floor, agree, floor_agree, total,
                                             floor_cpu=cfloor, floor_cpu_agree=cfloor_agree)
    # KV pool contents: every slot the run wrote agrees with the oracle's pool
    dev_k = eng.kv_cache.pool[0].cpu().view(cfg.num_layers, slots, cfg.num_kv_heads, cfg.head_dim)
    for li in range(cfg.num_layers):
        torch.testing.assert_close(dev_k[li][:-page_size].float(), kp[li][:-page_size].float(), atol=3e-2, rtol=3e-2)
    eng.shutdown()


def test_sampling_path_runs_and_is_seed_deterministic(dev):
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.offline import OfflineRunner

    outs = []
    for _ in range(2):
        eng = make_engine(dev, graphs=True)
        from mini_sglang_amd import flashinfer_compat as fi

        fi.sampling._offset = 0
        runner = OfflineRunner(eng, seed=1)
        ps = prompts(4)
        sp = [SamplingParams(temperature=0.8, top_k=20, top_p=0.9, max_tokens=6, ignore_eos=True),
              SamplingParams(temperature=0.0, max_tokens=6, ignore_eos=True),
              SamplingParams(temperature=1.0, max_tokens=6, ignore_eos=True),
              SamplingParams(temperature=0.5, top_p=0.5, max_tokens=6, ignore_eos=True)]
        runner.generate(ps, sp)
        outs.append([runner.output_ids(s) for s in runner.last_states])
        eng.shutdown()
    assert outs[0] == outs[1]
    assert all(0 <= t < 1024 for row in outs[0] for t in row)


@pytest.mark.parametrize("page_size,new_tokens", [(1, 32), (16, 12)])
def test_baseline_config0_qwen3_0p6b_single_prompt_greedy(dev, page_size, new_tokens):
    """BASELINE.json configs[0] at the real Qwen3-0.6B dimensions: seeded N(0, 0.02^2)
    weights, one prompt of 32 ids randint(0, 10000) seed 0, temperature 0, 32 new tokens at page_size 1 (12 at
    page_size 16: the oracle re-materialises 2.4 GB of fp32 weights per forward).
    The GPU run (chunk-free prefill + graph-replayed decode steps) is teacher-forced through the CPU oracle:
    logits within 8e-2 at every step (28 bf16 layers, logit std 0.64: the tiny models' 3e-2 was exceeded by
    16 of 151 936 logits, max 4.3e-2), greedy id equal wherever the oracle's margin exceeds twice that,
    out_loc trace = the page table's slots for the request's positions, K pool equal to the oracle's."""
    import random

    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS

    cfg = EngineConfig(model=PRESETS["qwen3-0.6b"], dtype=torch.bfloat16, max_running_req=4, page_size=page_size,
                       cuda_graph_bs=[1], max_seq_len_override=256, num_page_override=1024 // page_size, seed=42)
    eng = Engine(cfg, dev)
    eng.kv_cache.pool.zero_()
    rnd = random.Random(0)
    prompt = [rnd.randint(0, 10000) for _ in range(32)]
    rec = []
    tol = 6e-2  # measured: max 4.3e-2 over 151 936 x 32 logits (std 0.64); distribution printed below
    ids, stats, runner = run(eng, [prompt], new_tokens, record=rec)
    dist = {}
    assert len(ids[0]) == new_tokens and stats["decode_steps"] == new_tokens - 1
    m = eng.cfg.model
    w = ref_model.weights_from_device_model(eng.model)
    table = eng.page_table.cpu()
    slots = eng.kv_cache.pool.shape[2] * eng.kv_cache.pool.shape[3]
    kp = [torch.zeros((slots, m.num_kv_heads, m.head_dim), dtype=torch.bfloat16) for _ in range(m.num_layers)]
    vp = [torch.zeros_like(k) for k in kp]
    row = rec[0]["rows"][0]
    sure_n = same_n = floor_same = cfloor_same = 0
    floor, cfloor = {}, {}
    wd, table_d = torch_bf16.weights_to(w, dev), table.to(dev)
    kpd = [torch.zeros_like(k, device=dev) for k in kp]
    vpd = [torch.zeros_like(k) for k in kpd]
    kpc, vpc = [torch.zeros_like(k) for k in kp], [torch.zeros_like(k) for k in kp]
    for step, r in enumerate(rec):
        # KV block indices: the slots this forward writes are the table's entries for its positions
        assert torch.equal(r["out_loc"].long(), table[row, r["positions"].long()].long())
        logits = ref_model.forward(m, w, r["input_ids"], r["positions"], r["out_loc"], kp, vp, table, r["rows"],
                                   r["k_lens"], r["q_lens"], r["phase"] == "prefill").float()[: r["size"]]
        tb = torch_bf16.forward(m, wd, r["input_ids"].to(dev), r["positions"].to(dev), r["out_loc"].to(dev), kpd, vpd, table_d,
                                r["rows"], r["k_lens"], r["q_lens"], r["phase"] == "prefill").float().cpu()[: r["size"]]
        floor = parity_stats.merge_stats(floor, parity_stats.logit_error_stats(tb, logits))
        floor_same += int(tb.argmax(-1)[0]) == int(logits.argmax(-1)[0])
        tc = torch_bf16.forward(m, w, r["input_ids"], r["positions"], r["out_loc"], kpc, vpc, table, r["rows"], r["k_lens"],
                                r["q_lens"], r["phase"] == "prefill", linear="fp32acc").float()[: r["size"]]
        cfloor = parity_stats.merge_stats(cfloor, parity_stats.logit_error_stats(tc, logits))
        cfloor_same += int(tc.argmax(-1)[0]) == int(logits.argmax(-1)[0])
        st = parity_stats.logit_error_stats(r["logits"], logits)
        dist = parity_stats.merge_stats(dist, st)
        assert st["max_abs"] <= tol, (step, parity_stats.fmt(st))
        top2 = logits.topk(2, dim=-1).values
        sure = bool((top2[0, 0] - top2[0, 1]) > 2 * tol)
        same = int(r["logits"].argmax(-1)[0]) == int(logits.argmax(-1)[0])
        assert same or not sure, f"step {step}: greedy id differs at a margin of {float(top2[0, 0] - top2[0, 1])}"
        sure_n += sure
        same_n += same
        assert ids[0][step] == int(r["logits"].argmax(-1)[0])  # the id fed back is the argmax of these logits
    assert same_n >= sure_n
    print(f"\n[config 0, Qwen3-0.6B dims, page {page_size}] {parity_stats.fmt(dist)}; greedy ids equal at {same_n}/{len(rec)} "
          f"steps ({sure_n} with a sure oracle margin)")
    assert dist["p99_abs"] <= tol / 2 and dist["mean_ulp"] <= 8.0
    parity_stats.assert_not_above_bf16_floor(f"Qwen3-0.6B dims, 28 layers, page {page_size}", dist, floor, same_n, floor_same, len(rec),
                                             floor_cpu=cfloor, floor_cpu_agree=cfloor_same)
    dev_k = eng.kv_cache.pool[0].cpu().view(m.num_layers, slots, m.num_kv_heads, m.head_dim)
    for li in (0, m.num_layers // 2, m.num_layers - 1):  # one bf16 ulp of a K element of magnitude 4 is 3e-2
        torch.testing.assert_close(dev_k[li][:-page_size].float(), kp[li][:-page_size].float(), atol=tol, rtol=tol)
    eng.shutdown()


def test_op_chain_one_layer_each_op_on_the_oracles_input(dev):
    """Where end-to-end drift comes from: every op of one Qwen3-0.6B-sized layer runs on the GPU on the ORACLE's
    input for that op and is compared with the oracle's output for it, in bf16 ulp of the oracle value.  Each op
    stays within 2 ulp (norms, activation: 1; RoPE, GEMMs: 2, the fp32 accumulation order; attention: p99 3 ulp and
    8e-3 absolute, the bf16 rounding of P before P.V) -- the end-to-end bounds above are these single-ulp differences compounding through
    28 layers, not a loose kernel."""
    import torch.nn.functional as F

    from mini_sglang_amd import flashinfer_compat as fi
    from mini_sglang_amd import ops
    from mini_sglang_amd.model import PRESETS

    cfg, D = PRESETS["qwen3-0.6b"], 128
    H, hq, hkv, inter, eps = cfg.hidden_size, cfg.num_qo_heads, cfg.num_kv_heads, cfg.intermediate_size, cfg.rms_norm_eps
    T, page = 96, 16
    g = torch.Generator().manual_seed(12)

    def w(*shape, std=0.02):
        return (torch.randn(shape, generator=g) * std).to(torch.bfloat16)

    W = dict(input_norm=(1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16), qkv=w((hq + 2 * hkv) * D, H),
             q_norm=(1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16),
             k_norm=(1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16), o=w(H, hq * D),
             post_norm=(1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16), gate_up=w(2 * inter, H), down=w(H, inter))
    x0 = (torch.randn((T, H), generator=g) * 0.7).to(torch.bfloat16)
    res0 = (torch.randn((T, H), generator=g) * 0.7).to(torch.bfloat16)
    positions = torch.arange(T, dtype=torch.int32)
    slots = torch.randperm(8 * page, generator=g)[: T // page].to(torch.int32) * page
    table = (slots[:, None] + torch.arange(page, dtype=torch.int32)[None, :]).reshape(1, -1).contiguous()
    out_loc = table[0, :T].clone()
    n_slots = 9 * page * 16
    cos_sin = ref_ops.rope_cos_sin_cache(D, 4096, cfg.rope_base, None)

    # ---- oracle chain (P/layers/attention.py:47-57, P/models/qwen3.py:33-44), keeping every intermediate
    r = {}
    r["norm"], r["res1"] = ref_ops.fused_add_rmsnorm_ref(x0, res0, W["input_norm"], eps)
    r["qkv"] = F.linear(r["norm"].float(), W["qkv"].float()).to(torch.bfloat16)
    q, k, v = r["qkv"].split([hq * D, hkv * D, hkv * D], dim=-1)
    qn = ref_ops.rmsnorm_ref(q.reshape(T, hq, D), W["q_norm"], eps).reshape(T, hq * D)
    kn = ref_ops.rmsnorm_ref(k.reshape(T, hkv, D), W["k_norm"], eps).reshape(T, hkv * D)
    r["q_rope"], r["k_rope"] = ref_ops.rope_neox_ref(positions, qn, kn, D, cos_sin)
    kp = torch.zeros((n_slots, hkv, D), dtype=torch.bfloat16)
    vp = torch.zeros_like(kp)
    ref_ops.store_kv_ref(kp.view(-1, hkv * D), vp.view(-1, hkv * D), out_loc, r["k_rope"], v)
    r["attn"] = ref_ops.paged_attention_ref(r["q_rope"].reshape(T, hq, D), kp, vp, table, [0], [T], [T], D ** -0.5)
    r["o"] = F.linear(r["attn"].reshape(T, hq * D).float(), W["o"].float()).to(torch.bfloat16)
    r["norm2"], r["res2"] = ref_ops.fused_add_rmsnorm_ref(r["o"], r["res1"], W["post_norm"], eps)
    r["gate_up"] = F.linear(r["norm2"].float(), W["gate_up"].float()).to(torch.bfloat16)
    r["act"] = ref_ops.silu_and_mul_ref(r["gate_up"])
    r["down"] = F.linear(r["act"].float(), W["down"].float()).to(torch.bfloat16)

    # ---- each GPU op on the oracle's input
    d = lambda t: t.to(dev)  # noqa: E731
    Wd = {k_: d(v_) for k_, v_ in W.items()}
    got = {}
    xg, rg = d(x0).clone(), d(res0).clone()
    fi.fused_add_rmsnorm(xg, rg, Wd["input_norm"], eps)
    got["norm"], got["res1"] = xg, rg
    got["qkv"] = ops.linear(d(r["norm"]), Wd["qkv"])
    qkv_g = d(r["qkv"]).clone()
    qg, kg, vg = qkv_g.split([hq * D, hkv * D, hkv * D], dim=-1)
    kpd, vpd = d(kp).zero_(), d(vp).zero_()
    ops.qk_norm_rope_store(qg, kg, vg, Wd["q_norm"], Wd["k_norm"], eps, d(positions), d(cos_sin), kpd.view(-1, hkv * D),
                           vpd.view(-1, hkv * D), d(out_loc), D)
    got["q_rope"], got["k_rope"] = qg, kg
    assert torch.equal(kpd.cpu()[out_loc.long()].reshape(T, -1), kg.cpu()) and torch.equal(vpd.cpu()[out_loc.long()].reshape(T, -1), v)
    PREFILL_QTILE = ops.prefill_q_tile()
    tiles = (T + PREFILL_QTILE - 1) // PREFILL_QTILE
    attn = torch.empty((T, hq, D), dtype=torch.bfloat16, device=dev)
    ops.attn_prefill(attn, d(r["q_rope"]).view(T, hq, D), d(kp), d(vp), d(table), torch.zeros(1, dtype=torch.int32, device=dev),
                     torch.tensor([T], dtype=torch.int32, device=dev), torch.tensor([0, T], dtype=torch.int32, device=dev),
                     torch.tensor([0, tiles], dtype=torch.int32, device=dev), 1, tiles, D ** -0.5)
    got["attn"] = attn
    got["o"] = ops.linear(d(r["attn"]).view(T, hq * D), Wd["o"])
    x2, r2 = d(r["o"]).clone(), d(r["res1"]).clone()
    fi.fused_add_rmsnorm(x2, r2, Wd["post_norm"], eps)
    got["norm2"], got["res2"] = x2, r2
    got["gate_up"] = ops.linear(d(r["norm2"]), Wd["gate_up"])
    got["act"] = fi.silu_and_mul(d(r["gate_up"]))
    got["down"] = ops.linear(d(r["act"]), Wd["down"])
    torch.cuda.synchronize()
    bound = dict(norm=1, res1=0, qkv=2, q_rope=2, k_rope=2, attn=3, o=2, norm2=1, res2=0, gate_up=2, act=1, down=2)
    lines = []
    for name, ulps in bound.items():
        st = parity_stats.logit_error_stats(got[name].float().cpu().reshape(-1), r[name].float().reshape(-1))
        lines.append(f"{name:8s} max {st['max_ulp']:.2f} ulp  p99 {st['p99_ulp']:.2f}  mean {st['mean_ulp']:.3f}  |err| max {st['max_abs']:.2e}")
        if name == "attn":  # P is rounded to bf16 before P.V (FA-style): 2^-6 relative + cancellation near zero outputs
            assert st["max_abs"] <= 8e-3 and st["p99_ulp"] <= ulps + 1e-6, (name, parity_stats.fmt(st))
        else:
            assert st["max_ulp"] <= ulps + 1e-6, (name, parity_stats.fmt(st))
    print("\n[op chain, Qwen3-0.6B layer dims, each op on the oracle's input]\n" + "\n".join(lines))


class _OneGpuTpComm:
    """Stands in for rank 0's RCCL communicator of a tp-rank group on a box with ONE GPU: collectives go through
    a one-rank RCCL communicator (world-1 shortcut disabled, so they are real ncclAllReduce / ncclAllGather
    enqueues), the all-gather fills shard 0 of the destination and zeroes the others.  Numerically that is
    'every other rank contributed zeros' -- enough to run the engine's TP code path end to end."""

    def __init__(self, tp_size):
        from mini_sglang_amd import kernel

        self.tp_size = tp_size
        self.inner = kernel.RcclCommunicator(0, 1, 0, kernel.create_unique_id())

    def all_reduce(self, x, op="sum"):
        self.inner.all_reduce(x, op)

    def all_gather(self, out, x):
        rows = x.shape[0]
        self.inner.all_gather(out[:rows], x)
        out[rows:].zero_()

    def destroy(self):
        self.inner.destroy()


@pytest.mark.parametrize("tp", [2])
def test_tp_code_path_runs_on_one_gpu_with_collectives_in_the_graph(dev, tp, monkeypatch):
    """Engine(tp_size=2, tp_rank=0): sharded weights and KV pool (5 q heads / 1 kv head per rank), masked
    vocab-parallel gather, row-parallel all-reduces and the LM-head all-gather captured inside the decode
    hipGraph (thread-local capture mode), GEMM search on the shard shapes.  Graph replay must equal eager."""
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS

    monkeypatch.setenv("MSGL_COMM_NO_SHORTCUT", "1")
    outs = []
    for graphs in (True, False):
        comm = _OneGpuTpComm(tp)
        cfg = EngineConfig(model=PRESETS["tiny"], dtype=torch.bfloat16, tp_rank=0, tp_size=tp, max_running_req=8,
                           page_size=16, cuda_graph_bs=[1, 2, 4] if graphs else [], max_seq_len_override=256,
                           num_page_override=64, comm=comm, seed=7)
        eng = Engine(cfg, dev)
        assert eng.model.hq == 5 and eng.model.hkv == 1 and eng.attn_backend.kv_heads == 1
        assert eng.model.lm_head.shape[0] == (PRESETS["tiny"].vocab_size + tp - 1) // tp
        ids, stats, _ = run(eng, prompts(3), 6)
        assert stats["decode_steps"] == 5 and all(len(i) == 6 for i in ids)
        assert all(0 <= t < PRESETS["tiny"].vocab_size for seq in ids for t in seq)
        outs.append(ids)
        eng.shutdown()
        comm.destroy()
    assert outs[0] == outs[1]
