"""Row-owner generation of the decode projection (csrc/gemm_ro.hip, msgl_ro_gemm_nt) against the fp32 statement of the
reference's `F.linear` (P/layers/linear.py:32,103,124; P/layers/embedding.py:98):

  * every plan (tiles, k-slices) of the search + adversarial ones (one tile, ragged widths, more items than CUs, every M
    class: 3 .. 256, partial token tiles) within atol = 2^-7 * max|ref| of x.float() @ w.float().T (bf16 output rounding
    2^-9 relative plus the accumulation-order term; the same bound tests/test_gpu_gemm.py writes for every other GEMM path);
  * the result of a row does not depend on the batch it is in, on the tile cut, or on the launch (bit equality);
  * rows past M are never written; the k-sliced plans' slabs added in slice order by the kernel's own reduce launch equal
    the slab consumers of the decoder layer (fused_add_rmsnorm_slabs, qk_norm_rope_store_slabs) bit for bit;
  * MSGL_RO_SILU == the plain launch followed by the activation kernel, bit for bit (P/layers/activation.py:9-12).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from mini_sglang_amd import ops as _ops

    return _ops


def _ref(x, w):
    return x.float() @ w.float().t()


def _check(out, ref):
    tol = 2 ** -7 * max(ref.abs().max().item(), 1e-3)
    assert torch.isfinite(out.float()).all()
    assert (out.float() - ref).abs().max().item() <= tol, ((out.float() - ref).abs().max().item(), tol)


def _plans(ops, M, N, K):
    units, nsteps, umax = N // 16, K // 64, ops.ro_max_units(M)
    lo = -(-units // umax)
    plans = set(ops.ro_candidates(M, N, K, 256))
    plans |= {(lo, 1), (units, 1), (lo, min(3, nsteps)), (min(units, lo + 1), min(2, nsteps)), (min(units, 2 * lo + 1), min(7, nsteps)),
              (min(units, 300), 1)}
    return sorted(p for p in plans if lo <= p[0] <= units and (p[1] == 1 or p[1] * M * N * 4 <= ops.GEMM_WORKSPACE_BYTES))


@pytest.mark.parametrize("M,N,K", [(256, 5120, 5120), (256, 7168, 1024), (256, 34816, 512), (200, 2064, 640), (129, 144, 64),
                                   (128, 5120, 2048), (128, 34816, 256), (100, 7168, 512), (64, 1024, 17408), (48, 4352, 320),
                                   (17, 288, 128), (9, 16, 64), (16, 151936, 128), (136, 1008, 192), (3, 5120, 5120), (8, 34816, 512),
                                   (4, 7168, 1024)])
def test_ro_gemm_matches_fp32_reference(ops, dev, M, N, K):
    g = torch.Generator(device=dev).manual_seed(M * 31 + N + K)
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    ref = _ref(x, w)
    first = {}
    for tiles, slices in _plans(ops, M, N, K):
        out = torch.full((M + 3, N), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.ro_linear(x, w, tiles, slices, out=out[:M])
        _check(out[:M], ref)
        assert bool(out[M:].isnan().all()), (tiles, slices)
        assert torch.equal(out[:M], ops.ro_linear(x, w, tiles, slices)), (tiles, slices)   # repeatable
        # the tile cut is not part of the arithmetic: plans with the same k-slicing agree bit for bit
        if slices in first:
            assert torch.equal(out[:M], first[slices]), (tiles, slices)
        first.setdefault(slices, out[:M].clone())


def test_ro_row_result_does_not_depend_on_the_batch(ops, dev):
    """Row m of x @ w^T is the same bits at M = 256, 130, 128, 40 (both accumulator shapes, partial token tiles)."""
    g = torch.Generator(device=dev).manual_seed(5)
    x = (torch.randn((256, 1024), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((2304, 1024), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    for slices in (1, 4):
        full = ops.ro_linear(x, w, 64, slices)
        for M in (130, 128, 40, 9):
            assert torch.equal(ops.ro_linear(x[:M], w, 64 if M > 128 else 16, slices), full[:M]), (M, slices)


def test_ro_identity_fp16_strided_operands_and_argument_checks(ops, dev):
    M, N, K = 256, 256, 256
    x = torch.eye(K, dtype=torch.bfloat16, device=dev)[:M]
    w = (torch.arange(N, device=dev)[:, None] * 0.25 + torch.arange(K, device=dev)[None, :] * 3.0).to(torch.bfloat16)
    for plan in ((2, 1), (16, 1), (4, 2), (3, 4)):
        assert torch.equal(ops.ro_linear(x, w, *plan).float(), w.float().t()[:M].contiguous()), plan
    g = torch.Generator(device=dev).manual_seed(11)
    big = (torch.randn((200, 3 * 512), generator=g, device=dev) * 0.5).to(torch.float16)
    xs = big[:, 512:1024]
    ws = (torch.randn((768, 1024), generator=g, device=dev) * 0.05).to(torch.float16)[:, :512]
    fused = torch.zeros((200, 2048), dtype=torch.float16, device=dev)
    for plan in ((6, 1), (48, 1), (8, 4), (48, 2)):
        _check(ops.ro_linear(xs, ws, *plan, out=fused[:, 256:1024]), _ref(xs, ws))
    assert fused[:, :256].abs().max().item() == 0 and fused[:, 1024:].abs().max().item() == 0
    with pytest.raises(RuntimeError):
        ops.ro_linear(x, torch.zeros((200, 256), dtype=torch.bfloat16, device=dev), 4, 1)   # N % 16
    with pytest.raises(RuntimeError):
        ops.ro_linear(x, w, 1, 1)        # 16 units in one tile at M = 256: over the accumulator budget
    with pytest.raises(RuntimeError):
        ops.ro_linear(x, w, 17, 1)       # more tiles than units
    with pytest.raises(RuntimeError):
        ops.ro_linear(x, w, 4, 5)        # more k-slices than 64-k steps
    with pytest.raises(RuntimeError):
        ops.ro_linear(x, w, 4, 2, silu=True)   # the fused activation needs whole-K items
    with pytest.raises(RuntimeError):
        ops.ro_linear(torch.zeros((257, 256), dtype=torch.bfloat16, device=dev), w, 4, 1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,inter,K,tiles", [(256, 17408, 5120, 256), (256, 1024, 512, 16), (200, 3072, 1024, 100), (129, 512, 640, 8),
                                             (128, 17408, 1024, 256), (64, 4352, 512, 31), (24, 64, 64, 1), (9, 2048, 128, 256), (4, 17408, 512, 256)])
def test_ro_fused_silu_equals_projection_then_activation(ops, dev, dtype, M, inter, K, tiles):
    """gate_up_proj + silu_and_mul (P/models/utils.py:45-51) as ONE launch on the interleaved weight: bit-identical to the
    projection rounded to 16 bits followed by the activation kernel; against the fp32 oracle within the activation's bound."""
    from oracle import ref_ops

    g = torch.Generator(device=dev).manual_seed(M + inter + K)
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(dtype)
    w = (torch.randn((2 * inter, K), generator=g, device=dev) * 0.05).to(dtype)
    wi = ops.interleave_gate_up(w)
    idx = ops.gate_up_interleave_index(inter, dev)
    fused = torch.full((M + 1, inter), float("nan"), dtype=dtype, device=dev)
    ops.ro_linear(x, wi, tiles, 1, out=fused[:M], silu=True)
    assert bool(fused[M:].isnan().all())
    gu = ops.ro_linear(x, wi, tiles, 1)
    assert torch.equal(gu, ops.ro_linear(x, w, tiles, 1).index_select(1, idx))   # a row's dot product does not know its neighbours
    assert torch.equal(fused[:M], ops.silu_and_mul_interleaved(gu))
    want = ref_ops.silu_and_mul_ref(_ref(x, w).to(dtype).cpu()).float().to(dev)
    err = (fused[:M].float() - want).abs()
    assert err.max().item() <= 2 ** -6 * max(want.abs().max().item(), 1e-3), err.max().item()
    key = (dev.index or 0, M, 2 * inter, K, x.stride(0), wi.stride(0), ops._dt(x))
    try:
        ops._RO_SILU_PLAN[key] = (tiles, 1)
        assert torch.equal(ops.linear_silu(x, wi), fused[:M])
    finally:
        ops._RO_SILU_PLAN.clear()


@pytest.mark.parametrize("M,N,K,plan", [(256, 5120, 5120, (64, 4)), (130, 5120, 1024, (37, 3)), (128, 5120, 17408, (32, 8)), (40, 1024, 512, (4, 2))])
def test_ro_slabs_folded_into_fused_add_rmsnorm(ops, dev, M, N, K, plan):
    """A k-sliced plan's reduce left to the norm that follows o_proj / down_proj (P/models/qwen3.py:36-41): same bits as the
    kernel's own reduce launch + fused_add_rmsnorm, through ops.linear_slabs as the decoder layer calls it."""
    g = torch.Generator(device=dev).manual_seed(M + N)
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    res = torch.randn((M, N), generator=g, device=dev).to(torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(N, generator=g, device=dev)).to(torch.bfloat16)
    key = (dev.index or 0, M, N, K, x.stride(0), w.stride(0), ops._dt(x))
    try:
        ops._RO_PLAN[key] = plan
        y = ops.linear(x, w)
        assert torch.equal(y, ops.ro_linear(x, w, *plan))
        r1 = res.clone()
        ops.fused_add_rmsnorm(y, r1, gamma, 1e-6)
        out, slabs = ops.linear_slabs(x, w)
        assert slabs is not None and slabs.count == plan[1]
        r2 = res.clone()
        ops.fused_add_rmsnorm_slabs(out, r2, gamma, 1e-6, slabs)
        assert torch.equal(out, y) and torch.equal(r1, r2)
    finally:
        ops.reset_gemm_plans()


def test_ro_slabs_folded_into_qk_norm_rope_store(ops, dev):
    """qkv_proj's k-sliced reduce left to the fused qk-norm / RoPE / store pass (P/layers/attention.py:47-57): same q, k, v and
    pool rows as reduce-then-pass."""
    M, hq, hk, D, K = 200, 8, 2, 128, 1024
    N = (hq + 2 * hk) * D
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    qn = (1 + 0.1 * torch.randn(D, generator=g, device=dev)).to(torch.bfloat16)
    kn = (1 + 0.1 * torch.randn(D, generator=g, device=dev)).to(torch.bfloat16)
    pos = torch.randint(0, 4096, (M,), generator=g, device=dev, dtype=torch.int32)
    loc = torch.randperm(512, generator=g, device=dev)[:M].to(torch.int32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, device=dev, dtype=torch.float32) / D))
    ang = torch.arange(4096, device=dev, dtype=torch.float32)[:, None] * inv[None, :]
    cos_sin = torch.cat([ang.cos(), ang.sin()], dim=-1).contiguous()
    key = (dev.index or 0, M, N, K, x.stride(0), w.stride(0), ops._dt(x))
    try:
        ops._RO_PLAN[key] = (24, 4)
        pools = []
        outs = []
        for use_slabs in (False, True):
            kc = torch.zeros((512, hk * D), dtype=torch.bfloat16, device=dev)
            vc = torch.zeros_like(kc)
            if use_slabs:
                qkv, slabs = ops.linear_slabs(x, w)
                assert slabs is not None
                ops.qk_norm_rope_store_slabs(qkv, slabs, hq, hk, qn, kn, 1e-6, pos, cos_sin, kc, vc, loc, D)
            else:
                qkv = ops.linear(x, w)
                q, k, v = qkv.split([hq * D, hk * D, hk * D], dim=-1)
                ops.qk_norm_rope_store(q, k, v, qn, kn, 1e-6, pos, cos_sin, kc, vc, loc, D)
            outs.append(qkv.clone())
            pools.append((kc, vc))
        assert torch.equal(outs[0], outs[1])
        assert torch.equal(pools[0][0], pools[1][0]) and torch.equal(pools[0][1], pools[1][1])
    finally:
        ops.reset_gemm_plans()


def test_ro_under_graph_capture_and_tune_dispatch(ops, dev):
    """The launch is capturable (no allocation, no synchronisation) and ro_tune's plan is what ops.linear then runs."""
    g = torch.Generator(device=dev).manual_seed(1)
    M, N, K = 128, 4096, 1024
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    ws = [(torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16) for _ in range(3)]
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.ro_linear(x, ws[0], 128, 2, out)
    want = out.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            ops.ro_linear(x, ws[0], 128, 2, out)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
    assert torch.equal(out, want)
    try:
        r = ops.ro_tune(x, ws, incumbent_us=1e9)
        assert r["used"] and r["plan"] in ops.ro_candidates(M, N, K, int(ops.lib().msgl_device_cu_count()))
        _check(ops.linear(x, ws[1]), _ref(x, ws[1]))
        assert torch.equal(ops.linear(x, ws[1]), ops.ro_linear(x, ws[1], *r["plan"]))
        r2 = ops.ro_tune(x, ws, incumbent_us=1e-3)   # nothing beats a nanosecond: no plan is left behind
        assert not r2["used"] and not ops._RO_PLAN
    finally:
        ops.reset_gemm_plans()
