"""CPU suite (runs without a GPU): pins the oracle against the fixtures generated from the
reference's own Python (tests/golden/make_golden.py), and checks the host-side product logic
that needs no device."""
import re
from pathlib import Path

import pytest
import torch

from oracle import bytes_c, ref_ops

ROOT = Path(__file__).resolve().parent.parent


# ------------------------------------------------------------------ RoPE cache (rotary.py:12-114)
def test_rope_cache_oracle_and_product_match_reference_rows(golden_dir):
    from mini_sglang_amd.flashinfer_compat import build_cos_sin_cache

    gold = torch.load(golden_dir / "rope_cache.pt")
    assert set(gold) == {"qwen3_default", "llama3", "yarn", "hd64_default"}
    for name, c in gold.items():
        kw = c["kwargs"]
        for build in (ref_ops.rope_cos_sin_cache, build_cos_sin_cache):
            cache = build(kw["rotary_dim"], kw["max_position"], kw["base"], kw["rope_scaling"])
            assert tuple(cache.shape) == c["shape"], name
            assert torch.equal(cache[c["positions"]], c["rows"]), (name, build.__module__)


# ------------------------------------------------------------------ sampler clamps (sample.py:53-68)
def test_sampler_prepare_matches_reference(golden_dir):
    gold = torch.load(golden_dir / "sampler_prepare.pt")
    for name, c in gold["sets"].items():
        params = [ref_ops.SamplingParamsRef(*p) for p in c["params"]]
        assert [p.is_greedy for p in params] == c["is_greedy"], name
        ts, tk, tp = ref_ops.sampler_prepare_ref(params, gold["vocab"])
        for mine, ref, dt in ((ts, c["temperatures"], torch.float32), (tk, c["top_k"], torch.int32),
                              (tp, c["top_p"], torch.float32)):
            if ref is None:
                assert mine is None, name
            else:
                assert torch.equal(torch.tensor(mine, dtype=dt), ref), name


def test_product_sampler_prepare_matches_reference(golden_dir):
    """The product's Sampler.prepare (column-wise numpy clamps, one packed upload) against the outputs of the reference's
    Sampler.prepare stored by tests/golden/make_golden.py (P/engine/sample.py:53-68): same tensors, same dtypes, a filter
    column present exactly when the reference builds one."""
    from types import SimpleNamespace

    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Sampler

    gold = torch.load(golden_dir / "sampler_prepare.pt")
    sampler = Sampler(torch.device("cpu"), gold["vocab"])
    for name, c in gold["sets"].items():
        reqs = [SimpleNamespace(sampling_params=SamplingParams(temperature=t, top_k=k, top_p=p)) for t, k, p in c["params"]]
        plan = sampler.prepare(SimpleNamespace(reqs=reqs))
        for mine, ref, dt in ((plan.temperatures, c["temperatures"], torch.float32), (plan.top_k, c["top_k"], torch.int32),
                              (plan.top_p, c["top_p"], torch.float32)):
            if ref is None:
                assert mine is None, name
            else:
                assert mine is not None and mine.dtype == dt and torch.equal(mine, ref), name


def test_product_sampling_params_is_greedy_matches_reference(golden_dir):
    from mini_sglang_amd.core import SamplingParams

    gold = torch.load(golden_dir / "sampler_prepare.pt")
    for c in gold["sets"].values():
        for (t, k, p), g in zip(c["params"], c["is_greedy"]):
            assert SamplingParams(temperature=t, top_k=k, top_p=p).is_greedy == g


# ------------------------------------------------------------------ attention metadata (fa.py:67-105)
def test_fa_metadata_oracle_matches_reference(golden_dir):
    gold = torch.load(golden_dir / "fa_metadata.pt")
    for name, c in gold.items():
        reqs = [ref_ops.ReqRef(*s) for s in c["specs"]]
        md = ref_ops.fa_metadata_ref(reqs, c["table"], c["page_size"])
        for key in ("cu_seqlens_k", "cu_seqlens_q", "cache_seqlens", "page_table", "last_indices"):
            assert torch.equal(md[key].to(torch.int64), c[key].to(torch.int64)), (name, key)
        assert md["max_seqlen_k"] == c["max_seqlen_k"] and md["max_seqlen_q"] == c["max_seqlen_q"]


def test_product_metadata_host_buffer_matches_reference(golden_dir):
    """The product's host-side metadata assembly (attention.fill_metadata_host, no device needed) against the
    reference's FlashAttentionBackend.prepare_metadata outputs: cache_seqlens, cu_seqlens_q, last indices."""
    import numpy as np

    from mini_sglang_amd import _lib
    from mini_sglang_amd.attention import fill_metadata_host, prefill_tile_order

    gold = torch.load(golden_dir / "fa_metadata.pt")
    for name, c in gold.items():
        specs = c["specs"]  # (table_idx, cached_len, device_len)
        rows = np.array([s[0] for s in specs], dtype=np.int64)
        q = np.array([s[2] - s[1] for s in specs], dtype=np.int64)
        k = np.array([s[2] for s in specs], dtype=np.int64)
        bs = len(specs)
        decode = int(q.max()) == 1
        for qt in (_lib.PREFILL_QTILE, 256):  # rows per q tile: the 4-wave kernels' and the counter-phase kernel's
            tiles = (q + qt - 1) // qt
            total = 0 if decode else int(tiles.sum())
            h = np.full(4 * bs + 2 + total, -7, dtype=np.int32)
            fill_metadata_host(h, q, k, rows, decode, qt)
            assert np.array_equal(h[:bs], c["cache_seqlens"].numpy()), name
            assert np.array_equal(h[bs: 2 * bs], rows), name
            assert np.array_equal(h[2 * bs: 3 * bs + 1], c["cu_seqlens_q"].numpy()), name
            assert np.array_equal(h[2 * bs + 1: 3 * bs + 1] - 1, c["last_indices"].numpy()), name
            if not decode:
                assert h[3 * bs + 1] == 0 and np.array_equal(np.diff(h[3 * bs + 1: 4 * bs + 2]), tiles), name
                order = h[4 * bs + 2:]
                assert sorted(order.tolist()) == list(range(total)), name  # a permutation of the q tiles


def test_prefill_tile_order_is_heaviest_first():
    import numpy as np

    from mini_sglang_amd.attention import prefill_tile_order

    q = np.array([300, 1, 128, 129, 700], dtype=np.int64)
    k = np.array([300, 900, 128, 1000, 700], dtype=np.int64)
    tiles = (q + 127) // 128
    order = prefill_tile_order(q, k, tiles)
    req = np.repeat(np.arange(5), tiles)
    t_in = np.arange(tiles.sum()) - (np.cumsum(tiles) - tiles)[req]
    kend = np.minimum(k[req], k[req] - q[req] + np.minimum((t_in + 1) * 128, q[req]))
    assert sorted(order.tolist()) == list(range(int(tiles.sum())))
    assert (np.diff(kend[order]) <= 0).all()
    assert kend[order[0]] == 1000 and kend.min() == kend[order[-1]]


def test_in_place_page_table_walk_equals_reference_page_table(golden_dir):
    """The kernels read ctx.page_table[table_idx, t] (token slots) directly.  That must address
    the same KV rows as the reference's per-step table: slot(t) == page_table_new[b, t // page] * page
    + t % page for every valid t (fa.py:92-97 + flash-attn paged addressing)."""
    gold = torch.load(golden_dir / "fa_metadata.pt")
    for name, c in gold.items():
        ps = c["page_size"]
        for b, (ti, _, dl) in enumerate(c["specs"]):
            t = torch.arange(dl)
            ours = c["table"][ti, :dl].long()
            ref = c["page_table"][b, t // ps].long() * ps + t % ps
            assert torch.equal(ours, ref), name


# ------------------------------------------------------------------ radix key compare + page slots
def test_fast_compare_key_matches_reference_radix_calls(golden_dir):
    from mini_sglang_amd import ops

    gold = torch.load(golden_dir / "cache_allocate.pt")
    calls = gold["compare_calls"]
    assert len(calls) >= 3
    for x, y, r in calls:
        assert ref_ops.fast_compare_key_ref(x, y) == r
        assert bytes_c.compare_key(x.contiguous(), y.contiguous()) == r
        assert ops.fast_compare_key(x.contiguous(), y.contiguous()) == r  # product (host function of the C-ABI)
    # edge cases of C/src/radix.cpp:19-40
    a = torch.arange(10, dtype=torch.int64)
    assert ops.fast_compare_key(a, a[:4].clone()) == 4
    assert ops.fast_compare_key(a[:0].clone(), a) == 0
    b = a.clone(); b[7] = -1
    assert ops.fast_compare_key(a, b) == 7
    with pytest.raises(RuntimeError):
        ops.fast_compare_key(a, a.to(torch.int32))
    with pytest.raises(RuntimeError):
        ops.fast_compare_key(a.view(2, 5), a)


def test_page_aligned_allocation_matches_reference_trace(golden_dir):
    """First request of the reference CacheManager trace (no prefix hit): allocate_paged hands out the
    first pages of free_slots, expanded to token slots (cache.py:42-53,121-126)."""
    gold = torch.load(golden_dir / "cache_allocate.pt")
    for key, ps in (("page1", 1), ("page4", 4)):
        tr = gold[key]["trace"][0]
        n = len(tr["input_ids"])
        pages = -(-n // ps)
        free = torch.arange(gold[key]["num_pages"], dtype=torch.int32) * ps
        tok = bytes_c.page_to_token(free[:pages].contiguous(), ps)
        assert torch.equal(tok[:n], tr["prefill_row"])
        # every later row stays page-consistent: slot(t) - slot(page start) == t % ps
        for t in gold[key]["trace"]:
            row = t["final_row"].long()
            idx = torch.arange(len(row))
            assert torch.equal(row - row[idx - idx % ps], idx % ps)


# ------------------------------------------------------------------ byte movers
def test_indexing_oracles_match_reference(golden_dir):
    gold = torch.load(golden_dir / "indexing.pt")
    w, idx, rng = gold["weights"], gold["indices"], gold["mask_range"]
    for impl in (ref_ops.indexing_ref, bytes_c.index):
        assert torch.equal(impl(w, idx), gold["plain"])
        assert torch.equal(impl(w, idx, rng), gold["masked"])
        assert torch.equal(impl(gold["masked_local_weights"].contiguous(), idx, rng), gold["masked_local"])
    assert torch.equal(bytes_c.index(w, idx.long()), gold["plain"])  # int64 indices


def test_store_oracles_match_reference(golden_dir):
    gold = torch.load(golden_dir / "store.pt")
    H = gold["before"].shape[2]
    for impl in (ref_ops.store_kv_ref, bytes_c.store_kv):
        kv = gold["before"].clone()
        qkv = gold["qkv"]
        impl(kv[:, 0, :], kv[:, 1, :], gold["indices"], qkv[:, :H], qkv[:, H: 2 * H])
        assert torch.equal(kv, gold["after"])


# ------------------------------------------------------------------ known answers of tests/kernel/test_comm.py
@pytest.mark.parametrize("tp", [2, 4, 8])
def test_collective_oracle_known_answers(tp):
    x = [torch.ones(64, dtype=torch.float16) for _ in range(tp)]
    for _ in range(4):
        s = ref_ops.all_reduce_sum_ref(x)
        x = [s.clone() for _ in range(tp)]
    assert torch.equal(x[0], torch.full((64,), float(tp ** 4), dtype=torch.float16))
    r = ref_ops.all_reduce_sum_ref([torch.full((64,), float(i), dtype=torch.float16) for i in range(tp)])
    assert torch.equal(r, torch.full((64,), tp * (tp - 1) / 2, dtype=torch.float16))
    g = ref_ops.all_gather_ref([torch.full((8,), float(i), dtype=torch.float16) for i in range(tp)])
    assert torch.equal(g, torch.arange(tp, dtype=torch.float16).repeat_interleave(8))


# ------------------------------------------------------------------ filtered sampling distribution
def test_top_k_top_p_filter_semantics():
    p = torch.tensor([[0.5, 0.2, 0.15, 0.1, 0.05]])
    f = ref_ops.top_k_top_p_filter_ref(p, [3], None)[0]
    assert torch.allclose(f, torch.tensor([0.5, 0.2, 0.15, 0, 0], dtype=torch.float64) / 0.85)
    f = ref_ops.top_k_top_p_filter_ref(p, None, [0.6])[0]
    assert torch.allclose(f, torch.tensor([0.5, 0.2, 0, 0, 0], dtype=torch.float64) / 0.7)
    f = ref_ops.top_k_top_p_filter_ref(p, [4], [1e-6])[0]
    assert torch.allclose(f, torch.tensor([1.0, 0, 0, 0, 0], dtype=torch.float64))


# ------------------------------------------------------------------ C-ABI surface
def _declared(header: Path):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(msgl_[a-z0-9_]+)\s*\(", text)))


def test_c_abi_exports_every_declared_symbol():
    import ctypes

    from mini_sglang_amd import _lib

    names = _declared(ROOT / "include" / "msgl_hip.h")
    assert len(names) >= 25
    hip, comm = ctypes.CDLL(str(_lib.HIP_SO)), ctypes.CDLL(str(_lib.COMM_SO))
    gemm = ctypes.CDLL(str(_lib.GEMM_SO))
    for n in names:
        lib = comm if n.startswith("msgl_comm_") else gemm if n.startswith("msgl_gemm_") else hip
        assert hasattr(lib, n), f"{n} declared in include/msgl_hip.h but not exported"
    _lib.lib(); _lib.comm_lib(); _lib.gemm_lib()
    assert _lib.MISSING_SYMBOLS == []
    assert set(_lib.HIP_SIGNATURES) | set(_lib.COMM_SIGNATURES) | set(_lib.GEMM_SIGNATURES) == set(names)


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    """Argument checks run before any launch (reference: TensorMatcher / RuntimeCheck panics)."""
    import ctypes as C

    from mini_sglang_amd import _lib

    L = _lib.lib()
    buf = (C.c_char * 4096)()
    p = C.addressof(buf)
    p16 = (p + 15) // 16 * 16
    assert L.msgl_store_kv(p16, p16, p16, 0, p16, p16, 4, 8, 256, 256, 256, None) == -1  # 8-byte rows
    assert b"multiple of 16" in L.msgl_last_error()
    assert L.msgl_store_kv(p16 + 2, p16, p16, 0, p16, p16, 4, 256, 256, 256, 256, None) == -1  # misaligned
    assert L.msgl_store_kv(p16, p16, p16, 0, p16, p16, 0, 256, 256, 256, 256, None) == 0  # empty = no-op
    assert L.msgl_rmsnorm(p16, p16, p16, 1e-6, 4, 1, 100, 100, 0, 100, 0, 0, None) == -1  # dim % 8
    assert L.msgl_rmsnorm(p16, p16, p16, 1e-6, 4, 1, 128, 128, 0, 128, 0, 7, None) == -1  # dtype code
    assert L.msgl_rope_neox_inplace(p16, p16, p16, 0, p16, 4, 2, 2, 96, 512, 512, 0, None) == -1  # head_dim
    assert L.msgl_attn_decode(p16, p16, p16, p16, p16, 32, None, p16, p16, p16, 2, 4, 16, 8, 3, 128, 1024, 384, 128,
                              1024, 0.1, 1, 0, None) == -1  # 8 q heads / 3 kv heads
    assert L.msgl_attn_decode_plan_words(8, 4) == -1
    # hdr 4 | item_start 8 | n_chunks 8 | tile_start 8 | slot_first 65 -> 93, int4-aligned 96 | items 4 * 64 | items2 4 * 64
    assert L.msgl_attn_decode_plan_words(8, 64) == 96 + 256 + 256 + 8 * 64  # + arrival counters: 8 requests x 64 kv heads
    assert L.msgl_attn_decode_workspace_bytes(64, 40, 128) == 64 * 40 * 130 * 4
    assert L.msgl_fast_compare_key(None, 3, None, 3, 4) == -1
    # round-3 entry points: tile rows per prefill kernel, fused gate_up + SiLU.mul, slab-only weight-streaming GEMM
    assert L.msgl_attn_prefill_q_tile(0) == 128 and L.msgl_attn_prefill_q_tile(4) == 128 and L.msgl_attn_prefill_q_tile(2) == 128
    for code in (7, 1, 3, 5, 16, 17, 64, 65, 128, -1):  # unknown / removed generations / ablation codes of a diagnostic build
        assert L.msgl_attn_prefill(p16, p16, p16, p16, p16, 32, None, p16, p16, p16, 1, 1, 8, 2, 128, 1024, 256, 128, 1024, 0.1,
                                   0, None, code, None) == -1, code
    assert L.msgl_attn_prefill(p16, p16, p16, p16, p16, 32, None, p16, p16, p16, 1, 1, 8, 2, 128, 1024, 1 << 33, 128, 1024,
                               0.1, 0, None, 0, None) == -1  # token stride beyond the 32-bit slot-offset multiply
    assert L.msgl_skinny_gemm_silu_nt(p16, p16, p16, 4, 256, 128, 128, 128, 128, 0, 2, 3, None) == -1  # row tiles 1, 2, 4
    assert L.msgl_skinny_gemm_silu_nt(p16, p16, p16, 4, 96, 128, 128, 128, 128, 0, 2, 2, None) == -1   # gate / up blocks of 32 rows
    assert L.msgl_skinny_gemm_silu_nt(p16, p16, p16, 4, 256, 128, 128, 128, 64, 0, 2, 2, None) == -1   # ldo < N / 2
    assert L.msgl_wstream_gemm_slabs_nt(p16, p16, 64, 256, 128, 128, 128, 0, 1, 1, p16, 1 << 20, None) == -1  # needs >= 2 k splits
    assert L.msgl_wstream_gemm_slabs_nt(p16, p16, 64, 256, 128, 128, 128, 0, 1, 2, p16, 1024, None) == -1     # workspace too small
    with pytest.raises(RuntimeError):
        from mini_sglang_amd import ops

        ops.rmsnorm(torch.zeros(2, 128, dtype=torch.bfloat16), torch.ones(128, dtype=torch.bfloat16), 1e-6)  # CPU tensor


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under mini-sglang_amd/ may reference it."""
    for f in (ROOT / "mini-sglang_amd").glob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_tp_layout_helpers():
    from mini_sglang_amd.model import lm_head_unshard, vocab_shard

    assert vocab_shard(151936, 8, 7) == (18992, (132944, 18992))
    assert vocab_shard(10, 4, 3) == (3, (9, 1))
    tp, rows, per, V = 4, 3, 3, 10
    full = torch.arange(rows * tp * per).view(rows, tp * per)
    gathered = torch.cat([full[:, r * per:(r + 1) * per] for r in range(tp)], 0)
    assert torch.equal(lm_head_unshard(gathered, tp, rows, V), full[:, :V])
