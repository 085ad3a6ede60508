"""GPU parity: the row-streaming projection of the smallest decode batches (csrc/gemm_rowstream.hip, msgl_rowstream_gemm_nt)
vs an fp32 torch reference of `F.linear` (P/layers/linear.py:32), and its fused staging modes vs the kernels they replace.

Tolerance of the product: inputs 16-bit, accumulation fp32, output rounded once => atol = 2^-7 * max|ref| (as tests/test_gpu_gemm.py).
The staging modes are asserted BIT-IDENTICAL to the unfused sequence: activation kernel (P/layers/activation.py:9-12) or
fused_add_rmsnorm (P/layers/norm.py:33-38), then the same projection.
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
    assert (out.float() - ref).abs().max().item() <= tol


def _rand(shape, g, dev, scale, dtype):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(dtype)


# N not a multiple of anything (1000, 300), fewer rows than CUs (100), one row (1), the 14B / 32B / 70B projections, the LM head
SHAPES = [(7168, 5120), (5120, 17408), (34816, 5120), (151936, 5120), (1000, 1024), (300, 512), (100, 2048), (1, 512),
          (10240, 8192), (5120, 25600)]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8])
@pytest.mark.parametrize("N,K", SHAPES + [(5120, 4352), (5120, 1280), (8, 128)])
def test_rowstream_matches_fp32_reference(ops, dev, M, N, K, variant):
    """variant 0: products on the vector units (units of one row x 512 k); 1: on the matrix cores (four rows x 128 k)."""
    if not ops.rowstream_supported(M, N, K, 0, variant):
        why = ("multiple of 512" if K % 512 else "do not fit") if variant == 0 else ("variant 1 needs" if K % 128 or N % 4 else "do not fit")
        if why == "do not fit":  # x (padded to 1 / 2 / 4 / 8 rows) + the per-row wave sums must fit the CU's LDS
            assert M * K * 2 + (N + 255) // 256 * 32 * M > 100 * 1024
        with pytest.raises(RuntimeError, match=why):
            ops.rowstream_linear(torch.zeros((M, K), dtype=torch.bfloat16, device=dev),
                                 torch.zeros((N, K), dtype=torch.bfloat16, device=dev), variant=variant)
        return
    g = torch.Generator(device=dev).manual_seed(M * 131 + N + K)
    x = _rand((M, K), g, dev, 0.5, torch.bfloat16)
    w = _rand((N, K), g, dev, 0.05, torch.bfloat16)
    out16 = ops.rowstream_linear(x, w, 16, variant=variant)
    _check(out16, _ref(x, w))
    out8 = ops.rowstream_linear(x, w, 8, variant=variant)
    assert torch.equal(out8, out16)  # the ring's depth changes what is in flight, not the order of the sums


def test_rowstream_fp16_strides_and_untouched_neighbours(ops, dev):
    g = torch.Generator(device=dev).manual_seed(5)
    M, N, K = 3, 777, 1536
    bigx = _rand((M, 3 * K), g, dev, 0.5, torch.float16)
    bigw = _rand((N, K + 64), g, dev, 0.05, torch.float16)
    x, w = bigx[:, K:2 * K], bigw[:, :K]
    for variant, n in ((0, N), (1, N - 1)):  # 777 rows on the vector units, 776 (a multiple of 4) on the matrix cores
        fused = torch.zeros((M, n + 200), dtype=torch.float16, device=dev)
        out = ops.rowstream_linear(x, w[:n], 8, out=fused[:, 100:100 + n], variant=variant)
        _check(out, _ref(x, w[:n]))
        assert fused[:, :100].abs().max().item() == 0 and fused[:, 100 + n:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,inter,N", [(1, 17408, 5120), (2, 3072, 1024), (4, 17408, 5120), (7, 1024, 333), (8, 6144, 2048)])
@pytest.mark.parametrize("interleaved", [False, True])
def test_rowstream_silu_staging_equals_activation_then_projection(ops, dev, dtype, M, inter, N, interleaved):
    """down_proj(act_fn(gate_up)) (P/models/utils.py:45-51) in one launch: the activation is applied while x is staged."""
    g = torch.Generator(device=dev).manual_seed(M + inter + N)
    gu = _rand((M, 2 * inter), g, dev, 1.5, dtype)
    w = _rand((N, inter), g, dev, 0.03, dtype)
    act = ops.silu_and_mul_interleaved(gu) if interleaved else ops.silu_and_mul(gu)
    mode = ops.ROWSTREAM_SILU_INTERLEAVED if interleaved else ops.ROWSTREAM_SILU
    before = gu.clone()
    for variant in (0, 1):
        if not ops.rowstream_supported(M, N, inter, mode, variant):
            assert variant == 1 and N % 4
            continue
        want = ops.rowstream_linear(act, w, 16, variant=variant)
        for depth in (8, 16):
            got = ops.rowstream_linear(gu, w, depth, mode=mode, variant=variant)
            assert torch.equal(got, want)
        _check(want, _ref(act, w))
    assert torch.equal(gu, before)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N", [(1, 5120, 7168), (2, 5120, 34816), (3, 8192, 10240), (4, 5120, 151936), (4, 1536, 100), (1, 4096, 4096),
                                   (2, 1152, 512)])
def test_rowstream_add_norm_staging_equals_fused_add_rmsnorm_then_projection(ops, dev, dtype, M, K, N):
    """The decoder layer's `x, residual = norm(x, residual); y = proj(x)` (P/models/qwen3.py:36-41) in one launch."""
    g = torch.Generator(device=dev).manual_seed(M + K + N)
    eps = 1e-6
    x = _rand((M, K), g, dev, 0.7, dtype)
    res = _rand((M, K), g, dev, 2.0, dtype)
    gamma = (1.0 + 0.2 * torch.randn((K,), generator=g, device=dev)).to(dtype)
    w = _rand((N, K), g, dev, 0.03, dtype)
    x_ref, res_ref = x.clone(), res.clone()
    ops.fused_add_rmsnorm(x_ref, res_ref, gamma, eps)  # in place: x_ref = normed, res_ref = new residual
    x_before, res_before = x.clone(), res.clone()
    s = x.float() + res.float()
    y = (s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + eps) * gamma.float()).to(dtype)
    for variant in (0, 1):
        if not ops.rowstream_supported(M, N, K, ops.ROWSTREAM_ADD_NORM, variant):
            assert variant == 0 and K % 512
            continue
        want = ops.rowstream_linear(x_ref, w, 16, variant=variant)
        for depth in (8, 16):
            res_out = torch.full_like(res, float("nan"))
            got = ops.rowstream_linear(x, w, depth, mode=ops.ROWSTREAM_ADD_NORM, res_in=res, res_out=res_out, gamma=gamma, eps=eps,
                                       variant=variant)
            assert torch.equal(got, want)
            assert torch.equal(res_out, res_ref)
        _check(want, _ref(y, w))  # the pair against the fp32 statement of the op
    assert torch.equal(x, x_before) and torch.equal(res, res_before)


def test_rowstream_add_norm_strided_rows(ops, dev):
    g = torch.Generator(device=dev).manual_seed(9)
    M, K, N = 2, 2048, 640
    bx, br, bo = (_rand((M, 2 * K), g, dev, 1.0, torch.bfloat16) for _ in range(3))
    x, res, res_out = bx[:, :K], br[:, K:], bo[:, :K]
    gamma = (1.0 + 0.1 * torch.randn((K,), generator=g, device=dev)).to(torch.bfloat16)
    w = _rand((N, K), g, dev, 0.03, torch.bfloat16)
    keep = bo[:, K:].clone()
    x_ref, res_ref = x.contiguous(), res.contiguous()
    ops.fused_add_rmsnorm(x_ref, res_ref, gamma, 1e-5)
    got = ops.rowstream_linear(x, w, 16, mode=ops.ROWSTREAM_ADD_NORM, res_in=res, res_out=res_out, gamma=gamma, eps=1e-5, variant=1)
    assert torch.equal(got, ops.rowstream_linear(x_ref, w, 16, variant=1)) and torch.equal(res_out, res_ref)
    assert torch.equal(bo[:, K:], keep)


def test_rowstream_under_graph_capture(ops, dev):
    g = torch.Generator(device=dev).manual_seed(21)
    M, K, N = 2, 5120, 7168
    x = _rand((M, K), g, dev, 0.5, torch.bfloat16)
    res = _rand((M, K), g, dev, 1.0, torch.bfloat16)
    gamma = torch.ones((K,), dtype=torch.bfloat16, device=dev)
    w = _rand((N, K), g, dev, 0.05, torch.bfloat16)
    res_out, out = torch.empty_like(res), torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    kw = dict(mode=ops.ROWSTREAM_ADD_NORM, res_in=res, res_out=res_out, gamma=gamma, eps=1e-6)
    want = ops.rowstream_linear(x, w, 16, **kw).clone()
    want_res = res_out.clone()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ops.rowstream_linear(x, w, 16, out=out, **kw)
    torch.cuda.current_stream().wait_stream(s)
    out.zero_()
    res_out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(res_out, want_res)


def test_rowstream_rejects_what_it_cannot_do(ops, dev):
    z = lambda *s: torch.zeros(s, dtype=torch.bfloat16, device=dev)  # noqa: E731
    with pytest.raises(RuntimeError, match="M outside"):
        ops.rowstream_linear(z(9, 512), z(64, 512))
    with pytest.raises(RuntimeError, match="multiple of 512"):
        ops.rowstream_linear(z(1, 576), z(64, 576))
    with pytest.raises(RuntimeError, match="variant 1 needs"):
        ops.rowstream_linear(z(1, 512), z(66, 512), variant=1)
    with pytest.raises(RuntimeError, match="variant outside"):
        ops.rowstream_linear(z(1, 512), z(64, 512), variant=2)
    with pytest.raises(RuntimeError, match="depth"):
        ops.rowstream_linear(z(1, 512), z(64, 512), 4)
    with pytest.raises(RuntimeError, match="mode 3"):
        ops.rowstream_linear(z(1, 1024), z(64, 1024), 16, mode=3, res_in=z(1, 1024), res_out=z(1, 1024), gamma=z(1024))
    with pytest.raises(RuntimeError, match="mode 3"):
        ops.rowstream_linear(z(5, 2048), z(64, 2048), 16, mode=3, res_in=z(5, 2048), res_out=z(5, 2048), gamma=z(2048))
    r = z(1, 2048)
    with pytest.raises(RuntimeError, match="alias"):
        ops.rowstream_linear(z(1, 2048), z(64, 2048), 16, mode=3, res_in=r, res_out=r, gamma=z(2048))
    with pytest.raises(RuntimeError, match="CPU tensor"):
        ops.rowstream_linear(torch.zeros((1, 512), dtype=torch.bfloat16), z(64, 512))


def test_skinny_tune_times_the_rowstream_kernel_and_linear_dispatches(ops, dev):
    """The plan search of the small-batch projections (ops.skinny_tune) includes the row-streaming kernel as the settings
    (0, depth) -- vector units -- and (-1, depth) -- matrix cores; whatever wins, ops.linear must then give that kernel's bits."""
    g = torch.Generator(device=dev).manual_seed(33)
    M, N, K = 1, 5120, 17408
    x = _rand((M, K), g, dev, 0.5, torch.bfloat16)
    ws = [_rand((N, K), g, dev, 0.05, torch.bfloat16) for _ in range(3)]
    assert {(0, 8), (0, 16), (-1, 8), (-1, 16)} <= set(ops.skinny_candidates(M, N, K))
    assert not [c for c in ops.skinny_candidates(16, N, K) if c[0] <= 0]
    rep = ops.skinny_tune(x, ws, library_us=1e9)
    assert rep["used"]
    want = ops.skinny_linear(x, ws[0], rep["slices"], None, rep["row_tiles"])
    assert torch.equal(ops.linear(x, ws[0]), want)
    _check(want, _ref(x, ws[0]))
    print(f"down_proj M=1: best plan {(rep['slices'], rep['row_tiles'])} {rep['skinny_us']:.1f} us "
          f"({N * K * 2 / rep['skinny_us'] / 1e6:.2f} TB/s)")
    ops._SKINNY_PLAN.clear()
