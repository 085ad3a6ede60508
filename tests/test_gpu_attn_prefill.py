"""GPU parity: paged varlen causal prefill attention (MFMA) vs the CPU oracle.

Covers the three metadata regimes of P/attention/fa.py:84-90: no cache hit (q_len == k_len),
partial radix hit / chunked prefill (q_len < k_len, bottom-right aligned mask), and mixed
batches with ragged lengths around the 128-row / 64-key tile edges.
"""
import math

import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu

TOL = dict(atol=8e-3, rtol=2 ** -6)  # P is rounded to bf16 before P.V (as FA-style kernels do)


@pytest.fixture(scope="module")
def ops(dev):
    from mini_sglang_amd import ops as _ops

    return _ops


def build(g, specs, hq, hkv, page_size, dtype=torch.bfloat16):
    """specs: list of (cached_len, device_len) per request."""
    D = 128
    B = len(specs)
    max_seq = (max(dl for _, dl in specs) + 31) // 32 * 32
    pages_per = (max_seq + page_size - 1) // page_size
    n_pages = (B + 2) * pages_per + 2
    slots = n_pages * page_size
    k = torch.randn((slots, hkv, D), generator=g).to(dtype)
    v = torch.randn((slots, hkv, D), generator=g).to(dtype)
    poison = (n_pages - 1) * page_size
    k[poison:] = float("nan")
    v[poison:] = float("nan")
    perm = torch.randperm(n_pages - 2, generator=g)
    table = torch.full((B + 2, max_seq), poison, dtype=torch.int32)
    rows = torch.randperm(B + 2, generator=g)[:B].tolist()
    pp = 0
    for (cl, dl), row in zip(specs, rows):
        need = (dl + page_size - 1) // page_size
        pg = perm[pp: pp + need].to(torch.int32) * page_size
        pp += need
        tok = (pg.unsqueeze(1) + torch.arange(page_size, dtype=torch.int32)).flatten()
        table[row, :dl] = tok[:dl]
    q_lens = [dl - cl for cl, dl in specs]
    T = sum(q_lens)
    qkv = torch.randn((T, (hq + 2 * hkv) * D), generator=g).to(dtype)
    return dict(k=k, v=v, table=table, rows=rows, q_lens=q_lens, k_lens=[dl for _, dl in specs], qkv=qkv, hq=hq,
                hkv=hkv)


@pytest.fixture(params=[0, 2], ids=["default_dma", "tr_read"])
def impl(request):
    """Both kernels (include/msgl_hip.h: 0 = the default = impl 4, DMA-staged; 2 = its register-staged predecessor, the A/B
    partner)."""
    return request.param


def run(ops, dev, c, impl=0, order="heavy"):
    import numpy as np

    from mini_sglang_amd.attention import prefill_tile_order

    D, hq = 128, c["hq"]
    qkv = c["qkv"].to(dev)
    T = qkv.shape[0]
    q = qkv[:, : hq * D].view(T, hq, D)
    out = torch.zeros((T, hq, D), dtype=qkv.dtype, device=dev)
    cu_q = torch.tensor([0] + c["q_lens"], dtype=torch.int32).cumsum(0).to(torch.int32)
    qt = ops.prefill_q_tile(impl)  # rows per q tile of the kernel (128)
    tiles = [(n + qt - 1) // qt for n in c["q_lens"]]
    tile_cu = torch.tensor([0] + tiles, dtype=torch.int32).cumsum(0).to(torch.int32)
    tile_order = None
    if order == "heavy":
        tile_order = torch.from_numpy(prefill_tile_order(np.array(c["q_lens"], dtype=np.int64),
                                                         np.array(c["k_lens"], dtype=np.int64),
                                                         np.array(tiles, dtype=np.int64), qt)).to(dev)
    elif order == "reversed":
        tile_order = torch.arange(int(tile_cu[-1]) - 1, -1, -1, dtype=torch.int32, device=dev)
    ops.attn_prefill(out, q, c["k"].to(dev), c["v"].to(dev), c["table"].to(dev),
                     torch.tensor(c["rows"], dtype=torch.int32, device=dev),
                     torch.tensor(c["k_lens"], dtype=torch.int32, device=dev), cu_q.to(dev), tile_cu.to(dev),
                     len(c["q_lens"]), int(tile_cu[-1]), D ** -0.5, tile_order=tile_order, impl=impl)
    torch.cuda.synchronize()
    return out.cpu()


def oracle(c):
    D, hq = 128, c["hq"]
    T = c["qkv"].shape[0]
    q = c["qkv"][:, : hq * D].reshape(T, hq, D)
    return ref_ops.paged_attention_ref(q, c["k"], c["v"], c["table"], c["rows"], c["k_lens"], c["q_lens"],
                                       D ** -0.5, double=True)


@pytest.mark.parametrize("hq,hkv", [(16, 8), (40, 8), (8, 1), (4, 4)])
@pytest.mark.parametrize("page_size", [1, 16])
def test_prefill_no_cache_hit(ops, dev, impl, hq, hkv, page_size):
    g = torch.Generator().manual_seed(hq + page_size)
    specs = [(0, n) for n in (1, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 300, 517)]
    c = build(g, specs, hq, hkv, page_size)
    out = run(ops, dev, c, impl)
    assert torch.isfinite(out.float()).all()
    torch.testing.assert_close(out.double(), oracle(c), **TOL)


def test_prefill_partial_hit_and_chunked(ops, dev, impl):
    """q_len < k_len: radix prefix hits (page-aligned cached_len) and chunked-prefill continuation."""
    g = torch.Generator().manual_seed(1)
    specs = [(16, 40), (64, 65), (128, 400), (1000, 1001), (512, 1024), (0, 7), (256, 257 + 128), (48, 49)]
    c = build(g, specs, 40, 8, 16)
    out = run(ops, dev, c, impl)
    torch.testing.assert_close(out.double(), oracle(c), **TOL)


def test_prefill_tile_order_does_not_change_results(ops, dev):
    """tile_order is a scheduling hint: natural, heaviest-first and reversed orders are bit-identical."""
    g = torch.Generator().manual_seed(11)
    specs = [(0, 300), (128, 400), (0, 1), (512, 1024), (0, 129)]
    c = build(g, specs, 10, 2, 16)
    for impl in (0, 2):
        a = run(ops, dev, c, impl, order="heavy")
        assert torch.equal(a, run(ops, dev, c, impl, order=None))
        assert torch.equal(a, run(ops, dev, c, impl, order="reversed"))
        torch.testing.assert_close(a.double(), oracle(c), **TOL)


def test_prefill_generations_agree(ops, dev):
    """Same fragment ownership and accumulation order in both kernels => identical bits."""
    g = torch.Generator().manual_seed(12)
    c = build(g, [(0, 517), (64, 200), (1000, 1100)], 16, 8, 1)
    # the DMA-staged kernel (4, the default) keeps the register-staged kernel's (2) math, fragment ownership and accumulation
    # order per query row: identical bits, also on a batch with cache hits, a ragged tail, a request shorter than one tile
    # and one spanning several tiles
    for cc in (c, build(torch.Generator().manual_seed(13), [(0, 1), (0, 255), (0, 257), (300, 1100), (0, 700), (4096, 4200)], 10, 2, 16)):
        a = run(ops, dev, cc, 2)
        assert torch.equal(a, run(ops, dev, cc, 4))
        assert torch.equal(a, run(ops, dev, cc, 0))


def test_prefill_rejects_removed_and_diagnostic_impl_codes(ops, dev):
    """a stray MSGL_PREFILL_IMPL / impl argument must not select a timing-only ablation kernel (wrong results)
    or a removed generation: production builds accept 0, 2, 4 only."""
    from mini_sglang_amd._lib import MsglError

    c = build(torch.Generator().manual_seed(14), [(0, 40)], 4, 2, 16)
    for bad in (1, 3, 5, 16, 17, 64, 65, 128, 511, -1):
        with pytest.raises(MsglError):
            run(ops, dev, c, bad)


def test_prefill_long(ops, dev, impl):
    g = torch.Generator().manual_seed(2)
    specs = [(0, 2048), (1024, 3000)]
    c = build(g, specs, 16, 8, 1)
    out = run(ops, dev, c, impl)
    torch.testing.assert_close(out.double(), oracle(c), **TOL)


def test_prefill_fp16(ops, dev, impl):
    g = torch.Generator().manual_seed(3)
    c = build(g, [(0, 200), (32, 90)], 16, 8, 1, dtype=torch.float16)
    out = run(ops, dev, c, impl)
    torch.testing.assert_close(out.double(), oracle(c), atol=2e-3, rtol=2 ** -8)


def test_prefill_transpose_detecting(ops, dev, impl):
    """Asymmetric structured inputs (guide rule 16): V[key, d] = key + d/1000 with uniform
    attention => O[q, d] = mean_key(key) + d/1000; a swapped row/col layout cannot pass."""
    g = torch.Generator().manual_seed(4)
    c = build(g, [(0, 96)], 8, 8, 1)
    D = 128
    slots = c["table"][c["rows"][0], :96].long()
    c["k"][slots] = 0
    keys = torch.arange(96, dtype=torch.float32).view(96, 1, 1) / 16.0
    dd = torch.arange(D, dtype=torch.float32).view(1, 1, D) / 64.0
    hh = torch.arange(8, dtype=torch.float32).view(1, 8, 1)
    c["v"][slots] = (keys + dd + hh).to(torch.bfloat16)
    out = run(ops, dev, c, impl)
    torch.testing.assert_close(out.double(), oracle(c), **TOL)


def test_prefill_matches_decode_on_last_token(ops, dev):
    """The last query row of a prefill equals a decode step over the same KV."""
    import test_gpu_attn_decode as td

    g = torch.Generator().manual_seed(5)
    c = build(g, [(0, 333), (100, 777)], 40, 8, 1)
    out = run(ops, dev, c)
    D, hq = 128, 40
    last = [c["q_lens"][0] - 1, sum(c["q_lens"]) - 1]
    case = dict(k=c["k"], v=c["v"], table=c["table"], rows=c["rows"], lens=c["k_lens"],
                qkv=c["qkv"][last].contiguous(), hq=hq, hkv=8, D=D)
    dec, _ = td.run_decode(ops, dev, case)
    torch.testing.assert_close(out[last].float(), dec.float(), atol=8e-3, rtol=2 ** -6)
