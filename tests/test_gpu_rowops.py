"""GPU parity: byte movers (bit-exact) and row ops (bf16 tolerance) vs the CPU oracle.

Shapes follow the reference's own kernel tests (tests/kernel/test_store.py:10-34,
tests/kernel/test_index.py:33-91) and the call sites in P/layers/.
"""
import pytest
import torch

from oracle import bytes_c, ref_ops

pytestmark = pytest.mark.gpu

# bf16 has 8 significand bits: 1 ulp = 2^-8 relative; allow 2 ulp for differing op order
BF16_TOL = dict(atol=1e-2, rtol=2 ** -7)
FP16_TOL = dict(atol=2e-3, rtol=2 ** -9)


def _tol(dtype):
    return BF16_TOL if dtype == torch.bfloat16 else FP16_TOL


@pytest.fixture(scope="module")
def ops(dev):
    from mini_sglang_amd import ops as _ops

    return _ops


# ------------------------------------------------------------------ store_kv (bit exact)
@pytest.mark.parametrize("bs", [1, 2, 3, 17, 256, 4096, 32768])
@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64])
def test_store_cache_interleaved_and_fused_views(ops, dev, bs, idx_dtype):
    """tests/kernel/test_store.py:15-34: K/V caches are strided views of one interleaved
    tensor, k/v are column slices of a fused qkv; equality must be exact."""
    HEAD, NUM = 128, 65536
    g = torch.Generator().manual_seed(bs)
    kv_cache = torch.randn((NUM, 2, HEAD), generator=g).to(torch.float16)
    kv_dev = kv_cache.to(dev)
    k_cache, v_cache = kv_dev[:, 0, :], kv_dev[:, 1, :]
    indices = torch.randperm(NUM, generator=g)[:bs].to(idx_dtype)
    qkv = torch.randn((bs, HEAD * 4), generator=g).to(torch.float16)
    qkv_dev = qkv.to(dev)
    ops.store_kv(k_cache, v_cache, indices.to(dev), qkv_dev[:, :HEAD], qkv_dev[:, HEAD: 2 * HEAD])
    # oracle (plain C, same strided views)
    bytes_c.store_kv(kv_cache[:, 0, :], kv_cache[:, 1, :], indices, qkv[:, :HEAD], qkv[:, HEAD: 2 * HEAD])
    assert torch.equal(kv_dev.cpu(), kv_cache)


@pytest.mark.parametrize("row_elems", [128, 256, 512, 1024])  # 256 B (TP8) ... 2048 B (TP1) rows
def test_store_cache_row_sizes(ops, dev, row_elems):
    g = torch.Generator().manual_seed(row_elems)
    cache_k = torch.zeros((1000, row_elems), dtype=torch.bfloat16)
    cache_v = torch.zeros((1000, row_elems), dtype=torch.bfloat16)
    k = torch.randn((77, row_elems), generator=g).to(torch.bfloat16)
    v = torch.randn((77, row_elems), generator=g).to(torch.bfloat16)
    idx = torch.randperm(1000, generator=g)[:77].to(torch.int32)
    ck, cv = cache_k.to(dev), cache_v.to(dev)
    ops.store_kv(ck, cv, idx.to(dev), k.to(dev), v.to(dev))
    ref_ops.store_kv_ref(cache_k, cache_v, idx, k, v)
    assert torch.equal(ck.cpu(), cache_k) and torch.equal(cv.cpu(), cache_v)


def test_store_cache_golden(ops, dev, golden_dir):
    gold = torch.load(golden_dir / "store.pt")
    kv = gold["before"].clone().to(dev)
    qkv = gold["qkv"].to(dev)
    H = kv.shape[2]
    ops.store_kv(kv[:, 0, :], kv[:, 1, :], gold["indices"].to(dev), qkv[:, :H], qkv[:, H: 2 * H])
    assert torch.equal(kv.cpu(), gold["after"])


def test_store_empty_and_bad_args(ops, dev):
    from mini_sglang_amd._lib import MsglError

    c = torch.zeros((8, 128), dtype=torch.bfloat16, device=dev)
    e = torch.zeros((0, 128), dtype=torch.bfloat16, device=dev)
    ops.store_kv(c, c.clone(), torch.zeros(0, dtype=torch.int32, device=dev), e, e)  # no-op
    bad = torch.zeros((2, 4), dtype=torch.bfloat16, device=dev)  # 8-byte rows
    with pytest.raises(MsglError):
        ops.store_kv(torch.zeros((8, 4), dtype=torch.bfloat16, device=dev),
                     torch.zeros((8, 4), dtype=torch.bfloat16, device=dev),
                     torch.zeros(2, dtype=torch.int32, device=dev), bad, bad)


# ------------------------------------------------------------------ gather (bit exact)
@pytest.mark.parametrize("bs", [1, 2, 5, 64, 1000, 32768])
@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64])
def test_indexing(ops, dev, bs, idx_dtype):
    """tests/kernel/test_index.py:33-52 (scaled-down table, same row size 4096 x fp16)."""
    EMBED, NUM = 4096, 8192
    g = torch.Generator().manual_seed(bs)
    w = torch.randn((NUM, EMBED), generator=g).to(torch.float16)
    idx = torch.randint(0, NUM, (bs,), generator=g).to(idx_dtype)
    out = ops.embedding_gather(w.to(dev), idx.to(dev))
    assert torch.equal(out.cpu(), bytes_c.index(w, idx))


@pytest.mark.parametrize("bs", [1, 7, 512, 4096])
def test_indexing_with_mask(ops, dev, bs):
    """tests/kernel/test_index.py:65-91: vocab range (V/4, V/4)."""
    EMBED, NUM, TP = 1024, 4096, 4
    g = torch.Generator().manual_seed(bs)
    w = torch.randn((NUM, EMBED), generator=g).to(torch.bfloat16)
    idx = torch.randint(0, NUM, (bs,), generator=g).to(torch.int32)
    rng = (NUM // TP, NUM // TP)
    out = ops.embedding_gather(w.to(dev), idx.to(dev), vocab_range=rng)
    assert torch.equal(out.cpu(), bytes_c.index(w, idx, rng))
    assert torch.equal(out.cpu(), ref_ops.indexing_ref(w, idx, rng))


def test_indexing_golden(ops, dev, golden_dir):
    gold = torch.load(golden_dir / "indexing.pt")
    w, idx = gold["weights"].to(dev), gold["indices"].to(dev)
    assert torch.equal(ops.embedding_gather(w, idx).cpu(), gold["plain"])
    assert torch.equal(ops.embedding_gather(w, idx, vocab_range=gold["mask_range"]).cpu(), gold["masked"])
    wl = gold["masked_local_weights"].to(dev)
    assert torch.equal(ops.embedding_gather(wl, idx, vocab_range=gold["mask_range"]).cpu(), gold["masked_local"])


# ------------------------------------------------------------------ RMSNorm
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,dim", [(1, 1024), (7, 1024), (256, 5120), (33, 8192), (5, 64), (300, 128), (3, 2048),
                                      (2, 16384)])
def test_rmsnorm_2d(ops, dev, dtype, rows, dim):
    g = torch.Generator().manual_seed(dim + rows)
    x = (torch.randn((rows, dim), generator=g) * 3).to(dtype)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype)
    eps = 1e-6
    out = ops.rmsnorm(x.to(dev), w.to(dev), eps)
    torch.testing.assert_close(out.cpu().float(), ref_ops.rmsnorm_ref(x, w, eps).float(), **_tol(dtype))


def test_rmsnorm_strided_3d_inplace(ops, dev):
    """qk-norm call of P/layers/attention.py:50-53: in place on a [T, heads, 128] view of qkv."""
    T, HQ, HK, D = 37, 5, 2, 128
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn((T, (HQ + 2 * HK) * D), generator=g).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16)
    qkv_dev = qkv.to(dev)
    q_view = qkv_dev[:, : HQ * D].view(-1, HQ, D)
    ops.rmsnorm(q_view, w.to(dev), 1e-6, out=q_view)
    ref = qkv.clone()
    ref[:, : HQ * D] = ref_ops.rmsnorm_ref(qkv[:, : HQ * D].reshape(T, HQ, D), w, 1e-6).reshape(T, HQ * D)
    got = qkv_dev.cpu()
    torch.testing.assert_close(got.float(), ref.float(), **BF16_TOL)
    assert torch.equal(got[:, HQ * D:], qkv[:, HQ * D:])  # k, v untouched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,dim", [(1, 1024), (256, 5120), (19, 8192), (4, 128)])
def test_fused_add_rmsnorm(ops, dev, dtype, rows, dim):
    g = torch.Generator().manual_seed(rows * dim)
    x = torch.randn((rows, dim), generator=g).to(dtype)
    r = (torch.randn((rows, dim), generator=g) * 2).to(dtype)
    w = (1 + 0.1 * torch.randn(dim, generator=g)).to(dtype)
    xd, rd = x.to(dev), r.to(dev)
    ops.fused_add_rmsnorm(xd, rd, w.to(dev), 1e-5)
    x_ref, r_ref = ref_ops.fused_add_rmsnorm_ref(x, r, w, 1e-5)
    assert torch.equal(rd.cpu(), r_ref)  # the rounded fp32 sum is exact
    torch.testing.assert_close(xd.cpu().float(), x_ref.float(), **_tol(dtype))


# ------------------------------------------------------------------ RoPE
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,HQ,HK,D", [(1, 16, 8, 128), (256, 40, 8, 128), (33, 5, 1, 128), (9, 4, 2, 64)])
@pytest.mark.parametrize("pos_dtype", [torch.int32, torch.int64])
def test_rope_neox_inplace_strided(ops, dev, dtype, T, HQ, HK, D, pos_dtype):
    """q, k are row-strided column slices of the fused qkv (P/layers/attention.py:49,54)."""
    g = torch.Generator().manual_seed(T + HQ)
    qkv = torch.randn((T, (HQ + 2 * HK) * D), generator=g).to(dtype)
    cache = ref_ops.rope_cos_sin_cache(D, 4096, 1000000.0)
    pos = torch.randint(0, 4096, (T,), generator=g).to(pos_dtype)
    qkv_dev = qkv.to(dev)
    q, k, v = qkv_dev.split([HQ * D, HK * D, HK * D], dim=-1)
    ops.rope_neox_inplace(pos.to(dev), q, k, D, cache.to(dev))
    qr, kr = ref_ops.rope_neox_ref(pos, qkv[:, : HQ * D], qkv[:, HQ * D: (HQ + HK) * D], D, cache)
    got = qkv_dev.cpu()
    torch.testing.assert_close(got[:, : HQ * D].float(), qr.float(), **_tol(dtype))
    torch.testing.assert_close(got[:, HQ * D: (HQ + HK) * D].float(), kr.float(), **_tol(dtype))
    assert torch.equal(got[:, (HQ + HK) * D:], qkv[:, (HQ + HK) * D:])


def test_rope_cache_golden(golden_dir, dev):
    """The device rope cache is built by the host mirror; it must reproduce the reference's rows."""
    from mini_sglang_amd.flashinfer_compat import build_cos_sin_cache

    gold = torch.load(golden_dir / "rope_cache.pt")
    for name, c in gold.items():
        kw = c["kwargs"]
        cache = build_cos_sin_cache(kw["rotary_dim"], kw["max_position"], kw["base"], kw["rope_scaling"])
        assert tuple(cache.shape) == c["shape"], name
        assert torch.equal(cache[c["positions"]], c["rows"]), name


# ------------------------------------------------------------------ silu_and_mul
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T,d", [(1, 3072), (256, 17408), (7, 8)])
def test_silu_and_mul(ops, dev, dtype, T, d):
    g = torch.Generator().manual_seed(T + d)
    x = (torch.randn((T, 2 * d), generator=g) * 2).to(dtype)
    out = ops.silu_and_mul(x.to(dev))
    torch.testing.assert_close(out.cpu().float(), ref_ops.silu_and_mul_ref(x).float(), **_tol(dtype))


# ------------------------------------------------------------------ fused qk-norm + rope + store
@pytest.mark.parametrize("has_norm", [True, False])
@pytest.mark.parametrize("T,HQ,HK,D", [(1, 16, 8, 128), (256, 40, 8, 128), (100, 5, 1, 128), (13, 8, 2, 64)])
def test_qk_norm_rope_store_matches_unfused_bitexact(ops, dev, has_norm, T, HQ, HK, D):
    """The fused pass must equal the reference op order (P/layers/attention.py:47-56 then
    mha_pool.store_kv) bit for bit: compare with our own unfused kernels, which are
    themselves oracle-checked above."""
    g = torch.Generator().manual_seed(T * 7 + HQ)
    dtype = torch.bfloat16
    qkv = torch.randn((T, (HQ + 2 * HK) * D), generator=g).to(dtype)
    qw = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype).to(dev)
    kw = (1 + 0.1 * torch.randn(D, generator=g)).to(dtype).to(dev)
    cache = ref_ops.rope_cos_sin_cache(D, 2048, 1000000.0).to(dev)
    pos = torch.randint(0, 2048, (T,), generator=g).to(torch.int32).to(dev)
    SLOTS = 4096
    loc = torch.randperm(SLOTS, generator=g)[:T].to(torch.int32).to(dev)

    a = qkv.to(dev)
    kc_a = torch.zeros((SLOTS, HK * D), dtype=dtype, device=dev)
    vc_a = torch.zeros_like(kc_a)
    q, k, v = a.split([HQ * D, HK * D, HK * D], dim=-1)
    if has_norm:
        ops.rmsnorm(q.view(-1, HQ, D), qw, 1e-6, out=q.view(-1, HQ, D))
        ops.rmsnorm(k.view(-1, HK, D), kw, 1e-6, out=k.view(-1, HK, D))
    ops.rope_neox_inplace(pos, q, k, D, cache)
    ops.store_kv(kc_a, vc_a, loc, k, v)

    b = qkv.to(dev)
    kc_b = torch.zeros_like(kc_a)
    vc_b = torch.zeros_like(kc_a)
    q2, k2, v2 = b.split([HQ * D, HK * D, HK * D], dim=-1)
    ops.qk_norm_rope_store(q2, k2, v2, qw if has_norm else None, kw if has_norm else None, 1e-6, pos, cache, kc_b,
                           vc_b, loc, D)
    assert torch.equal(a.cpu(), b.cpu())
    assert torch.equal(kc_a.cpu(), kc_b.cpu()) and torch.equal(vc_a.cpu(), vc_b.cpu())
