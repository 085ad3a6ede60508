"""GPU parity: projection GEMMs through msgl_gemm_nt (library solutions) vs an
fp32 torch reference of the same op (`F.linear`, P/layers/linear.py:32).

Tolerance: inputs are bf16, accumulation fp32, output rounded to bf16 once => |err| <= 2^-8 |ref| + a
split-K reordering term; asserted as atol = 2^-7 * max|ref| (written here, floating-point GEMM).
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


@pytest.mark.parametrize("M,N,K", [(1, 512, 256), (3, 1024, 512), (16, 7168, 5120), (256, 5120, 5120),
                                   (256, 1024, 17408), (37, 640, 1280), (1024, 2048, 1024)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_matches_fp32_reference(ops, dev, M, N, K, dtype):
    g = torch.Generator(device=dev).manual_seed(M * 7 + N + K)
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(dtype)
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(dtype)
    out = ops.linear(x, w)
    assert out.shape == (M, N) and out.dtype == dtype
    _check(out, _ref(x, w))


def test_linear_strided_operands(ops, dev):
    """x a column slice of a wider tensor (row stride > K), out a column slice of a fused buffer."""
    g = torch.Generator(device=dev).manual_seed(3)
    big = (torch.randn((64, 3 * 512), generator=g, device=dev)).to(torch.bfloat16)
    x = big[:, 512:1024]
    w = (torch.randn((768, 512), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    fused = torch.zeros((64, 2048), dtype=torch.bfloat16, device=dev)
    out = ops.linear(x, w, out=fused[:, 256:1024])
    _check(out, _ref(x, w))
    assert fused[:, :256].abs().max().item() == 0 and fused[:, 1024:].abs().max().item() == 0


@pytest.mark.parametrize("mode", [-8, 0])
def test_tuned_solution_is_correct_and_not_slower(ops, dev, mode):
    g = torch.Generator(device=dev).manual_seed(11)
    M, N, K = 256, 2048, 1024
    x = (torch.randn((M, K), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    ws = [(torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16) for _ in range(3)]
    rep = ops.gemm_tune(x, ws, max_candidates=mode, iters=5)
    assert rep["tried"] >= 1 and rep["best_us"] > 0
    assert rep["best_us"] <= rep["default_us"] * 1.05 + 1.0
    assert rep["kernel"].startswith("[tuned]")
    for w in ws:
        _check(ops.linear(x, w), _ref(x, w))


def test_linear_under_graph_capture(ops, dev):
    """No allocation / sync inside msgl_gemm_nt: legal under stream capture, replays bit-identically."""
    g = torch.Generator(device=dev).manual_seed(5)
    x = (torch.randn((32, 1024), generator=g, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((2048, 1024), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty((32, 2048), dtype=torch.bfloat16, device=dev)
    eager = ops.linear(x, w).clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.linear(x, w, out=out)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ops.linear(x, w, out=out)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)


def test_linear_rejects_cpu_tensors(ops):
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros((2, 8), dtype=torch.bfloat16), torch.zeros((4, 8), dtype=torch.bfloat16))
