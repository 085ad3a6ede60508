"""GPU parity: greedy argmax (bit exact), temperature softmax, top-k / top-p sampling.

The reference's RNG stream (flashinfer) cannot be reproduced, so non-greedy sampling is checked
distributionally (chi-square against the exact filtered distribution of the oracle) plus exactness
on degenerate rows (top_k = 1, top_p -> 0  =>  argmax), as SURVEY.md Appendix B prescribes.
"""
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from mini_sglang_amd import ops as _ops

    return _ops


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("rows,vocab", [(1, 151936), (256, 151936), (7, 128256), (3, 1000), (5, 8), (2, 33333)])
def test_argmax_first_max_index(ops, dev, dtype, rows, vocab):
    g = torch.Generator().manual_seed(rows + vocab)
    logits = torch.randn((rows, vocab), generator=g).to(dtype)
    # force ties: duplicate the max at a later index -> the first one must win
    for r in range(rows):
        i = int(torch.argmax(logits[r].float()))
        if i + 1 < vocab:
            logits[r, vocab - 1] = logits[r, i]
    out = ops.argmax_rows(logits.to(dev))
    assert torch.equal(out.cpu(), ref_ops.argmax_ref(logits))


def test_argmax_strided_rows(ops, dev):
    """The graph path samples from logits[:B] of a wider fp32 buffer (P/engine/graph.py:33,165)."""
    g = torch.Generator().manual_seed(0)
    buf = torch.randn((8, 1000 + 24), generator=g)
    view = buf[:, :1000]
    out = ops.argmax_rows(view.to(dev)[:, :1000])
    assert torch.equal(out.cpu(), ref_ops.argmax_ref(view))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_temperature(ops, dev, dtype):
    g = torch.Generator().manual_seed(1)
    logits = (torch.randn((16, 151936), generator=g) * 3).to(dtype)
    t = torch.tensor([1e-6, 0.1, 0.6, 1.0, 2.0, 0.3, 0.9, 1.5] * 2, dtype=torch.float32)
    probs = ops.softmax_temperature(logits.to(dev), t.to(dev)).cpu()
    ref = ref_ops.softmax_temperature_ref(logits, t)
    torch.testing.assert_close(probs.double(), ref, atol=1e-6, rtol=2e-4)
    torch.testing.assert_close(probs.sum(-1), torch.ones(16), atol=1e-4, rtol=0)


def test_sampling_degenerate_rows_are_argmax(ops, dev):
    g = torch.Generator().manual_seed(2)
    V = 151936
    probs = torch.softmax(torch.randn((6, V), generator=g) * 2, -1)
    pd = probs.to(dev)
    am = ref_ops.argmax_ref(probs)
    k1 = torch.ones(6, dtype=torch.int32, device=dev)
    assert torch.equal(ops.sample_top_k_top_p(pd, k1, None, 42, 0).cpu(), am)
    p0 = torch.full((6,), 1e-6, dtype=torch.float32, device=dev)
    assert torch.equal(ops.sample_top_k_top_p(pd, None, p0, 42, 0).cpu(), am)
    assert torch.equal(ops.sample_top_k_top_p(pd, k1, p0, 7, 123).cpu(), am)


def test_sampling_is_deterministic_in_seed_and_offset(ops, dev):
    g = torch.Generator().manual_seed(3)
    probs = torch.softmax(torch.randn((64, 32000), generator=g), -1).to(dev)
    a = ops.sample_top_k_top_p(probs, None, None, 42, 100)
    b = ops.sample_top_k_top_p(probs, None, None, 42, 100)
    c = ops.sample_top_k_top_p(probs, None, None, 42, 164)
    assert torch.equal(a, b) and not torch.equal(a, c)


def _chi2_ok(counts, expected_p, n):
    keep = expected_p > 0
    assert counts[~keep].sum() == 0, "sampled a token outside the filtered support"
    # merge tiny cells
    e = expected_p[keep] * n
    o = counts[keep].double()
    order = torch.argsort(e, descending=True)
    e, o = e[order], o[order]
    big = e >= 5
    e2 = torch.cat([e[big], e[~big].sum().view(1)])
    o2 = torch.cat([o[big], o[~big].sum().view(1)])
    ok = e2 > 0
    chi2 = (((o2 - e2) ** 2)[ok] / e2[ok]).sum().item()
    dof = int(ok.sum()) - 1
    # mean dof, variance 2 dof; 6 sigma bound keeps the test stable yet sharp
    assert chi2 < dof + 6 * (2 * max(dof, 1)) ** 0.5 + 10, (chi2, dof)


@pytest.mark.parametrize("top_k,top_p", [(None, None), (50, None), (None, 0.9), (20, 0.7), (5, 0.99)])
def test_sampling_distribution(ops, dev, top_k, top_p):
    """Draw N samples of ONE row (replicated), compare frequencies with the oracle's filtered
    distribution."""
    g = torch.Generator().manual_seed(5)
    V, N = 4096, 40000
    row = torch.softmax(torch.randn(V, generator=g) * 2.5, -1)
    probs = row.unsqueeze(0).repeat(N, 1).to(dev)
    tk = None if top_k is None else torch.full((N,), top_k, dtype=torch.int32, device=dev)
    tp = None if top_p is None else torch.full((N,), top_p, dtype=torch.float32, device=dev)
    out = ops.sample_top_k_top_p(probs, tk, tp, 1234, 0).cpu().long()
    assert out.min() >= 0 and out.max() < V
    counts = torch.bincount(out, minlength=V)
    expect = ref_ops.top_k_top_p_filter_ref(row.unsqueeze(0), None if top_k is None else [top_k],
                                            None if top_p is None else [top_p])[0]
    _chi2_ok(counts, expect, N)


def test_sampling_per_row_parameters(ops, dev):
    """Different (k, p) per row, as Sampler.prepare builds them (P/engine/sample.py:53-68)."""
    g = torch.Generator().manual_seed(6)
    V = 151936
    probs = torch.softmax(torch.randn((8, V), generator=g) * 3, -1)
    ks = [1, 5, V, 50, 1000, V, 2, 10]
    ps = [1.0, 0.5, 0.3, 1.0, 0.9, 1e-6, 0.99, 0.1]
    support = ref_ops.top_k_top_p_filter_ref(probs, ks, ps) > 0
    pd = probs.to(dev)
    tk = torch.tensor(ks, dtype=torch.int32, device=dev)
    tp = torch.tensor(ps, dtype=torch.float32, device=dev)
    for off in range(0, 80, 8):
        out = ops.sample_top_k_top_p(pd, tk, tp, 99, off).cpu().long()
        for r in range(8):
            assert support[r, out[r]], (r, int(out[r]))


# ---------------------------------------------------------------- fused softmax + draw (temperature only)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("temperature", [0.6, 1.0, 1.7])
def test_sample_from_logits_distribution(ops, dev, dtype, temperature):
    """One row replicated N times: frequencies of the fused kernel against softmax(logits / T) of the oracle."""
    g = torch.Generator().manual_seed(11)
    V, N = 4096, 40000
    logits = (torch.randn(V, generator=g) * 2.5).to(dtype)
    t = torch.full((N,), temperature, dtype=torch.float32)
    out = ops.sample_from_logits(logits.unsqueeze(0).repeat(N, 1).to(dev), t.to(dev), 4321, 0).cpu().long()
    assert out.min() >= 0 and out.max() < V
    expect = ref_ops.softmax_temperature_ref(logits.unsqueeze(0), t[:1])[0]
    _chi2_ok(torch.bincount(out, minlength=V), expect, N)


@pytest.mark.parametrize("vocab", [151936, 128256, 33336, 1000, 8])
def test_sample_from_logits_edges(ops, dev, vocab):
    """T -> 0 rows return the (unique) argmax; results are a function of (seed, offset) only; the graph path's
    fp32 buffer view (row stride > vocab) and a ragged last wave segment are covered by the vocab list."""
    g = torch.Generator().manual_seed(vocab)
    rows = 6
    pad = 24 if vocab % 8 == 0 else 0
    buf = torch.randn((rows, vocab + pad), generator=g) * 3
    logits = buf[:, :vocab]
    dl = buf.to(dev)[:, :vocab]
    cold = torch.full((rows,), 1e-6, dtype=torch.float32, device=dev)
    assert torch.equal(ops.sample_from_logits(dl, cold, 42, 0).cpu(), ref_ops.argmax_ref(logits))
    warm = torch.full((rows,), 0.8, dtype=torch.float32, device=dev)
    a = ops.sample_from_logits(dl, warm, 42, 100)
    b = ops.sample_from_logits(dl, warm, 42, 100)
    assert torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) < vocab
    draws = torch.stack([ops.sample_from_logits(dl, warm, 42, 1000 + 8 * i) for i in range(12)])
    if vocab > 8:
        assert len(set(draws[:, 0].tolist())) > 1  # the offset moves the Philox stream


def test_sample_from_logits_matches_probs_path_statistically(ops, dev):
    """Same logits through softmax -> sampling_from_probs and through the fused kernel: the two empirical
    distributions agree (two-sample chi-square on the pooled top cells)."""
    g = torch.Generator().manual_seed(17)
    V, N = 2048, 30000
    logits = torch.randn(V, generator=g) * 2
    t = torch.full((N,), 0.6, dtype=torch.float32, device=dev)
    dl = logits.unsqueeze(0).repeat(N, 1).to(dev)
    a = torch.bincount(ops.sample_from_logits(dl, t, 5, 0).cpu().long(), minlength=V).double()
    b = torch.bincount(ops.sample_top_k_top_p(ops.softmax_temperature(dl, t), None, None, 6, 0).cpu().long(),
                       minlength=V).double()
    big = (a + b) >= 20
    chi2 = (((a - b) ** 2)[big] / (a + b)[big]).sum().item()
    dof = int(big.sum()) - 1
    assert chi2 < dof + 6 * (2 * dof) ** 0.5 + 10, (chi2, dof)


def test_top_k_larger_than_the_number_of_positive_probabilities(ops, dev):
    """ADVICE r1 (sampling.hip:249): a low-temperature softmax underflows the tail to exactly 0; with top_k above
    the count of positive entries the count search finds no bin and must behave as 'keep every positive entry'
    (what flashinfer returns), also when top_p follows.  The sample can only ever be a positive-probability index."""
    V = 151936
    g = torch.Generator().manual_seed(5)
    probs = torch.zeros((6, V), dtype=torch.float32)
    hot = [int(torch.randint(0, V, (1,), generator=g)) for _ in range(6)]
    for r, h in enumerate(hot):
        probs[r, h] = 1.0                      # rows 0-2: one positive entry (T = 1e-6 on a non-greedy row)
    for r in (3, 4, 5):                        # rows 3-5: three positive entries, k = 50
        probs[r] = 0
        probs[r, hot[r]] = 0.7
        probs[r, (hot[r] + 17) % V] = 0.2
        probs[r, (hot[r] + 40000) % V] = 0.1
    pd = probs.to(dev)
    k = torch.full((6,), 50, dtype=torch.int32, device=dev)
    p = torch.tensor([0.9, 1e-6, 1.0, 0.5, 0.95, 1.0], dtype=torch.float32, device=dev)
    allowed = [set(torch.nonzero(probs[r]).flatten().tolist()) for r in range(6)]
    seen = [set() for _ in range(6)]
    for it in range(40):
        for kk, pp in ((k, p), (k, None)):
            out = ops.sample_top_k_top_p(pd, kk, pp, seed=11, offset=it * 6).cpu().tolist()
            for r, o in enumerate(out):
                assert o in allowed[r], (r, o)
                if pp is None:
                    seen[r].add(o)
    for r in range(3):
        assert seen[r] == {hot[r]}
    # row 3 with top_p = 0.5 keeps only the 0.7 entry; without top_p all three positives are reachable
    outs = [ops.sample_top_k_top_p(pd, k, p, seed=3, offset=i * 6).cpu().tolist()[3] for i in range(30)]
    assert set(outs) == {hot[3]}
    assert len(seen[5]) >= 2
