"""GPU parity: paged decode attention vs the CPU oracle (Appendix B "paged attention").

Synthetic page tables as SURVEY.md section 7 step 3 asks: randomly permuted token slots,
ragged lengths, dummy-page padding rows, table tails holding stale (even invalid) slots
that must never be dereferenced, q as a strided view of a fused qkv tensor.
"""
import math

import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu

TOL = dict(atol=6e-3, rtol=2 ** -7)  # 2 bf16 ulp on O(1) outputs


def make_case(g, B, hq, hkv, lens, page_size, dtype=torch.bfloat16, max_seq=None, poison_tail=True):
    D = 128
    max_seq = max_seq or ((max(lens) + 31) // 32 * 32)
    n_rows = B + 3
    pages_per_req = (max_seq + page_size - 1) // page_size
    n_pages = n_rows * pages_per_req + 2  # + dummy page (valid data) + poison page (NaN)
    slots = n_pages * page_size
    k_cache = torch.randn((slots, hkv, D), generator=g).to(dtype)
    v_cache = torch.randn((slots, hkv, D), generator=g).to(dtype)
    poison = (n_pages - 1) * page_size
    k_cache[poison:] = float("nan")
    v_cache[poison:] = float("nan")
    perm = torch.randperm(n_pages - 2, generator=g)
    table = torch.zeros((n_rows, max_seq), dtype=torch.int32)
    rows = torch.randperm(n_rows, generator=g)[:B].tolist()
    p = 0
    for b, row in enumerate(rows):
        need = (lens[b] + page_size - 1) // page_size
        pg = perm[p: p + need].to(torch.int32) * page_size
        p += need
        tok = (pg.unsqueeze(1) + torch.arange(page_size, dtype=torch.int32)).flatten()
        n = min(need * page_size, max_seq)
        table[row, :n] = tok[:n]
        if poison_tail and lens[b] < max_seq:
            # stale entries beyond the sequence must never be dereferenced: they point at NaN rows
            table[row, lens[b]:] = poison
    qkv = torch.randn((B, (hq + 2 * hkv) * D), generator=g).to(dtype)
    return dict(k=k_cache, v=v_cache, table=table, rows=rows, lens=lens, qkv=qkv, hq=hq, hkv=hkv, D=D)


def run_decode(ops, dev, case, max_bs=None, capacity=None, min_chunk=64, slot_run=1):
    B, hq, hkv, D = len(case["lens"]), case["hq"], case["hkv"], case["D"]
    max_bs = max_bs or B
    capacity = capacity or max(4 * max_bs, 1024)
    qkv = case["qkv"].to(dev)
    q = qkv[:, : hq * D].view(B, hq, D)
    out = torch.empty((B, hq, D), dtype=qkv.dtype, device=dev)
    plan = torch.zeros(ops.attn_decode_plan_words(max_bs, capacity), dtype=torch.int32, device=dev)
    ws = torch.empty(ops.attn_decode_workspace_bytes(capacity, hq, D), dtype=torch.uint8, device=dev)
    seq = torch.tensor(case["lens"], dtype=torch.int32, device=dev)
    rows = torch.tensor(case["rows"], dtype=torch.int32, device=dev)
    ops.attn_decode_plan(plan, seq, B, max_bs, capacity, hq, hkv, min_chunk)
    ops.attn_decode(out, q, case["k"].to(dev), case["v"].to(dev), case["table"].to(dev), rows, seq, plan, ws, B,
                    max_bs, capacity, 1.0 / math.sqrt(D), slot_run=slot_run)
    torch.cuda.synchronize()
    return out.cpu(), plan.cpu()


def oracle(case):
    B, hq, D = len(case["lens"]), case["hq"], case["D"]
    q = case["qkv"][:, : hq * D].reshape(B, hq, D)
    table = case["table"].clone()
    return ref_ops.paged_attention_ref(q, case["k"], case["v"], table, case["rows"], case["lens"], [1] * B,
                                       1.0 / math.sqrt(D), double=True)


@pytest.fixture(scope="module")
def ops(dev):
    from mini_sglang_amd import ops as _ops

    return _ops


@pytest.mark.parametrize("hq,hkv", [(16, 8), (40, 8), (64, 8), (8, 8), (32, 8), (5, 1), (8, 1), (10, 2), (24, 8),
                                    (28, 4), (48, 8), (16, 1)])
@pytest.mark.parametrize("page_size", [1, 16])
def test_decode_matches_oracle_groups(ops, dev, hq, hkv, page_size):
    g = torch.Generator().manual_seed(hq * 100 + hkv + page_size)
    lens = [1, 2, 15, 16, 17, 63, 64, 65, 200, 777, 1024, 33]
    case = make_case(g, len(lens), hq, hkv, lens, page_size)
    ref = oracle(case)
    # slot_run = 1: per-token table walk; slot_run = page_size: one scalar table read per 16-token tile
    for slot_run in ([1] if page_size < 16 else [1, page_size]):
        out, plan = run_decode(ops, dev, case, slot_run=slot_run)
        assert torch.isfinite(out.float()).all()
        torch.testing.assert_close(out.double(), ref, **TOL)
        if slot_run >= 16:  # both request shapes of the matrix-core kernel, whatever the default picks for this row width
            try:
                outs = []
                for code in (60, 61):
                    ops.attn_decode_select(code)
                    outs.append(run_decode(ops, dev, case, slot_run=slot_run)[0])
            finally:
                ops.attn_decode_select(0)
            assert torch.equal(outs[0], out) and torch.equal(outs[1], out)


@pytest.mark.parametrize("page_size,min_chunk", [(16, 16), (64, 64), (256, 64), (256, 256), (48, 32)])
def test_decode_slot_run_pages(ops, dev, page_size, min_chunk):
    """Page-aligned allocation (P/scheduler/cache.py:42-53): with slot_run = the largest power of two
    dividing page_size the kernel reads one table entry per tile.  Streaming kernel: results equal the per-token
    walk bit for bit (same arithmetic, same order).  Default choice (matrix-core kernel for runs >= 16): the same
    tolerance against the oracle."""
    g = torch.Generator().manual_seed(page_size * 7 + min_chunk)
    lens = [1, 16, 17, 31, 32, 33, 47, 48, 49, 255, 256, 257, 700, 1023, 1024, 1025, 2047, 3000, 5, 64]
    case = make_case(g, len(lens), 40, 8, lens, page_size)
    run = page_size & -page_size
    ref = oracle(case)
    try:
        ops.attn_decode_select(1)
        out_run, _ = run_decode(ops, dev, case, min_chunk=min_chunk, slot_run=run)
        out_tok, _ = run_decode(ops, dev, case, min_chunk=min_chunk, slot_run=1)
        assert torch.isfinite(out_run.float()).all()
        assert torch.equal(out_run, out_tok)
        torch.testing.assert_close(out_run.double(), ref, **TOL)
    finally:
        ops.attn_decode_select(0)
    out_mc, _ = run_decode(ops, dev, case, min_chunk=min_chunk, slot_run=run)
    assert torch.isfinite(out_mc.float()).all()
    torch.testing.assert_close(out_mc.double(), ref, **TOL)


@pytest.mark.parametrize("hq,hkv,dtype", [(40, 8, torch.bfloat16), (64, 8, torch.bfloat16), (16, 8, torch.float16),
                                          (8, 8, torch.bfloat16)])
def test_decode_long_pieces_through_the_request_ring(ops, dev, hq, hkv, dtype):
    """Enough tokens per slot (~50 tiles) that every wave runs the steady state of the matrix-core kernel's register
    ring (kStages - 1 tiles in flight), with pieces starting and ending inside requests, ragged last tiles, and
    requests shorter than the ring.  Both kernels against the oracle; repeatable bit for bit."""
    g = torch.Generator().manual_seed(hq + hkv)
    lens = [int(x) for x in torch.randint(2500, 4200, (60,), generator=g)] + [1, 15, 16, 17, 47, 48, 49, 64, 3, 33]
    case = make_case(g, len(lens), hq, hkv, lens, 256, dtype=dtype)
    ref = oracle(case)
    out, _ = run_decode(ops, dev, case, slot_run=256)
    torch.testing.assert_close(out.double(), ref, **TOL)
    again, _ = run_decode(ops, dev, case, slot_run=256)
    assert torch.equal(out, again)
    try:
        ops.attn_decode_select(1)
        out_s, _ = run_decode(ops, dev, case, slot_run=256)
        torch.testing.assert_close(out_s.double(), ref, **TOL)
        # round 6: whole-line nt requests with K transposed through LDS (default) vs round 5's 16-row x 64-B requests
        # (select 60): the same products on the same operands in the same order => the same bits
        ops.attn_decode_select(60)
        out_r5, _ = run_decode(ops, dev, case, slot_run=256)
        assert torch.equal(out, out_r5)
    finally:
        ops.attn_decode_select(0)
    err = (out.double() - ref).abs()
    print(f"\n[decode long pieces hq={hq} hkv={hkv} {dtype}] max abs err {err.max():.2e}, mean {err.mean():.2e}; "
          f"streaming kernel max {(out_s.double() - ref).abs().max():.2e}")


def read_plan(plan, max_bs, capacity, batch):
    """Decode the device plan (layout: csrc/attn_decode.hip plan_off_*)."""
    plan = plan.tolist()
    n_items, slot_tokens, _, n_slots = plan[:4]
    o_start, o_chunks, o_tiles, o_first = 4, 4 + max_bs, 4 + 2 * max_bs, 4 + 3 * max_bs
    o_items = (4 + 3 * max_bs + capacity + 1 + 3) // 4 * 4
    items = [tuple(plan[o_items + 4 * i: o_items + 4 * i + 4]) for i in range(n_items)]
    o_items2 = o_items + 4 * capacity  # per piece: (seq_len, pieces of the request, its first piece, 0)
    items2 = [tuple(plan[o_items2 + 4 * i: o_items2 + 4 * i + 4]) for i in range(n_items)]
    return dict(n_items=n_items, slot_tokens=slot_tokens, n_slots=n_slots, items=items, items2=items2,
                item_start=plan[o_start: o_start + batch], n_chunks=plan[o_chunks: o_chunks + batch],
                tile_start=plan[o_tiles: o_tiles + batch], slot_first=plan[o_first: o_first + n_slots + 1])


def check_plan(pl, lens):
    """Every tile of every request is covered exactly once, request-major; a slot holds consecutive
    pieces worth at most slot_tokens / 16 tiles; all slots but the last are full."""
    q = pl["slot_tokens"] // 16
    nts = [(n + 15) // 16 for n in lens]
    assert pl["n_slots"] == (sum(nts) + q - 1) // q
    cover = [0] * len(lens)
    load = [0] * pl["n_slots"]
    for i, (b, t0, t1, k) in enumerate(pl["items"]):
        assert 0 <= t0 < t1 <= nts[b]
        assert t0 == cover[b]  # pieces of a request are consecutive and ordered
        cover[b] = t1
        load[k] += t1 - t0
        assert pl["slot_first"][k] <= i < pl["slot_first"][k + 1]
    assert cover == nts
    for (b, _t0, _t1, _k), (seq, nch, first, _z) in zip(pl["items"], pl["items2"]):  # the per-piece copy of the request's scalars
        assert (seq, nch, first) == (lens[b], pl["n_chunks"][b], pl["item_start"][b])
    assert all(x == q for x in load[:-1]) and 0 < load[-1] <= q
    assert pl["slot_first"][-1] == pl["n_items"]
    for b, n in enumerate(nts):
        mine = [i for i, it in enumerate(pl["items"]) if it[0] == b]
        assert len(mine) == pl["n_chunks"][b]
        assert not mine or (mine[0] == pl["item_start"][b] and mine == list(range(mine[0], mine[0] + len(mine))))


@pytest.mark.parametrize("min_chunk", [16, 64, 256, 1024])
def test_decode_split_kv_chunks(ops, dev, min_chunk):
    """Long and short requests mixed: multi-piece merge and single-piece direct write; the balanced
    plan covers every tile exactly once."""
    g = torch.Generator().manual_seed(min_chunk)
    lens = [3000, 5, 1, 2047, 2048, 2049, 300, 17]
    case = make_case(g, len(lens), 40, 8, lens, 1)
    out, plan = run_decode(ops, dev, case, min_chunk=min_chunk)
    pl = read_plan(plan, len(lens), max(4 * len(lens), 1024), len(lens))
    assert pl["slot_tokens"] >= min_chunk
    check_plan(pl, lens)
    torch.testing.assert_close(out.double(), oracle(case), **TOL)


def test_decode_plan_balances_the_bench_batch(ops, dev):
    """256 ragged requests (the offline-bench context distribution): one slot per resident wave of a kv
    head, all equal (the uniform-chunk plan of round 1 left the slowest wave with 1.4x the mean)."""
    import random

    rnd = random.Random(0)
    lens = [rnd.randint(100, 2048) for _ in range(256)]
    hq, hkv, cap = 40, 8, 4096
    plan = torch.zeros(ops.attn_decode_plan_words(257, cap), dtype=torch.int32, device=dev)
    seq = torch.tensor(lens, dtype=torch.int32, device=dev)
    ops.attn_decode_plan(plan, seq, 256, 257, cap, hq, hkv)
    torch.cuda.synchronize()
    pl = read_plan(plan.cpu(), 257, cap, 256)
    check_plan(pl, lens)
    assert pl["n_items"] <= pl["n_slots"] + 256
    assert 64 <= pl["n_slots"] <= 2048


def test_decode_plan_respects_capacity(ops, dev):
    g = torch.Generator().manual_seed(5)
    lens = [4000] * 6 + [1, 9]
    case = make_case(g, len(lens), 16, 8, lens, 1)
    out, plan = run_decode(ops, dev, case, max_bs=16, capacity=16, min_chunk=16)
    assert int(plan[0]) <= 16
    check_plan(read_plan(plan, 16, 16, len(lens)), lens)
    torch.testing.assert_close(out.double(), oracle(case), **TOL)


def test_decode_padded_dummy_rows(ops, dev):
    """Graph padding (P/engine/graph.py:160-166): rows of the dummy request have length 1
    and all point at the dummy slot; they must not disturb real rows."""
    g = torch.Generator().manual_seed(9)
    lens = [120, 1, 1, 1, 64, 1, 1, 1]
    case = make_case(g, len(lens), 40, 8, lens, 1)
    dummy_row, dummy_slot = case["rows"][1], case["k"].shape[0] - 2  # last slot of the dummy page
    for i in (1, 2, 3, 5, 6, 7):
        case["rows"][i] = dummy_row
    case["table"][dummy_row].fill_(dummy_slot)
    out, _ = run_decode(ops, dev, case, max_bs=8)
    torch.testing.assert_close(out.double(), oracle(case), **TOL)


def test_decode_bench_shape_statistics(ops, dev):
    """BASELINE config 2/3 shape family at reduced batch: 32 requests, lengths 100..2048, GQA 5."""
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(100, 2049, (32,), generator=g).tolist()
    case = make_case(g, len(lens), 40, 8, lens, 16)
    out, _ = run_decode(ops, dev, case)
    torch.testing.assert_close(out.double(), oracle(case), **TOL)


def test_decode_fp16(ops, dev):
    g = torch.Generator().manual_seed(21)
    lens = [77, 300, 1, 1500]
    case = make_case(g, len(lens), 16, 8, lens, 1, dtype=torch.float16)
    out, _ = run_decode(ops, dev, case)
    torch.testing.assert_close(out.double(), oracle(case), atol=1e-3, rtol=2 ** -9)


def test_decode_softmax_extremes(ops, dev):
    """One key dominates (forces the running-max rescale path) and tokens identical (uniform P)."""
    g = torch.Generator().manual_seed(33)
    lens = [700, 700]
    case = make_case(g, 2, 40, 8, lens, 1)
    D = 128
    q = case["qkv"][:, : 40 * D].reshape(2, 40, D)
    # request 0: spike key at position 650 aligned with head 0's query
    slot = int(case["table"][case["rows"][0], 650])
    case["k"][slot, 0] = (q[0, 0].float() * 4).to(torch.bfloat16)
    # request 1: all keys identical => uniform attention => output = mean(V)
    slots1 = case["table"][case["rows"][1], :700].long()
    case["k"][slots1] = case["k"][slots1[0]].clone()
    out, _ = run_decode(ops, dev, case)
    ref = oracle(case)
    torch.testing.assert_close(out.double(), ref, **TOL)
    mean_v = case["v"][slots1].double().mean(0).repeat_interleave(5, dim=0)
    torch.testing.assert_close(out[1].double(), mean_v, atol=1e-2, rtol=1e-2)


def test_decode_graph_replay(ops, dev):
    """Capture once with the static-buffer contract (P/attention/fa.py:107-136), replay with
    different lengths / rows copied into the static buffers."""
    g = torch.Generator().manual_seed(77)
    D, hq, hkv, max_bs, capacity = 128, 40, 8, 8, 1024
    lens_a = [1] * 8
    case = make_case(g, 8, hq, hkv, [900, 40, 1, 333, 2000, 64, 65, 512], 1)
    kd, vd, td = case["k"].to(dev), case["v"].to(dev), case["table"].to(dev)
    seq = torch.tensor(lens_a, dtype=torch.int32, device=dev)
    rows = torch.tensor(case["rows"], dtype=torch.int32, device=dev)
    qkv = torch.zeros_like(case["qkv"]).to(dev)
    q = qkv[:, : hq * D].view(8, hq, D)
    out = torch.empty((8, hq, D), dtype=torch.bfloat16, device=dev)
    plan = torch.zeros(ops.attn_decode_plan_words(max_bs, capacity), dtype=torch.int32, device=dev)
    ws = torch.empty(ops.attn_decode_workspace_bytes(capacity, hq, D), dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ops.attn_decode_plan(plan, seq, 8, max_bs, capacity, hq, hkv)
        ops.attn_decode(out, q, kd, vd, td, rows, seq, plan, ws, 8, max_bs, capacity, D ** -0.5)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            ops.attn_decode_plan(plan, seq, 8, max_bs, capacity, hq, hkv)
            ops.attn_decode(out, q, kd, vd, td, rows, seq, plan, ws, 8, max_bs, capacity, D ** -0.5)
        # replay with the real batch
        seq.copy_(torch.tensor(case["lens"], dtype=torch.int32))
        qkv.copy_(case["qkv"])
        graph.replay()
        torch.cuda.synchronize()
    torch.testing.assert_close(out.cpu().double(), oracle(case), **TOL)


@pytest.mark.parametrize("hq,hkv,lens_kind", [(40, 8, "bench"), (16, 2, "ragged"), (64, 8, "ragged"), (8, 1, "many_pieces")])
def test_decode_in_kernel_combine_equals_the_merge_kernel(ops, dev, hq, hkv, lens_kind):
    """Matrix-core kernel, select code 72: the piece of a request that arrives LAST combines the partial sums inside the
    kernel (write-through stores, arrival counter, sc1 loads).  The result must be the merge kernel's bit for bit -- pieces
    are combined in piece order whoever arrives last -- on every launch of a sequence (the counters return to zero), and
    under hipGraph replay."""
    g = torch.Generator().manual_seed(hq * 7 + hkv)
    if lens_kind == "bench":
        lens = [int(x) for x in torch.randint(300, 2000, (96,), generator=g)]
    elif lens_kind == "ragged":
        lens = [1, 17, 300, 4000, 33, 2500, 64, 999, 16, 1500, 7, 3100]
    else:
        lens = [9000, 40, 7000, 3]
    case = make_case(g, len(lens), hq, hkv, lens, 256)
    ref = oracle(case)
    ops.attn_decode_select(22)          # the same kernel with the separate merge kernel
    want, plan = run_decode(ops, dev, case, slot_run=256, min_chunk=64)
    n_items = int(plan[0])
    assert n_items > len(lens), "the scenario must split requests"
    torch.testing.assert_close(want.double(), ref, **TOL)
    try:
        ops.attn_decode_select(72)
        _combine_checks(ops, dev, case, lens, hq, hkv, want)
    finally:
        ops.attn_decode_select(0)


def _combine_checks(ops, dev, case, lens, hq, hkv, want):
    for _ in range(3):
        got, _ = run_decode(ops, dev, case, slot_run=256, min_chunk=64)
        assert torch.equal(got, want)
    # many launches on one plan (as the 40 layers of a step), then under graph replay
    B, D = len(lens), 128
    cap = max(4 * B, 1024)
    qkv = case["qkv"].to(dev)
    q = qkv[:, : hq * D].view(B, hq, D)
    k, v, table = case["k"].to(dev), case["v"].to(dev), case["table"].to(dev)
    rows = torch.tensor(case["rows"], dtype=torch.int32, device=dev)
    seq = torch.tensor(lens, dtype=torch.int32, device=dev)
    plan = torch.zeros(ops.attn_decode_plan_words(B, cap), dtype=torch.int32, device=dev)
    ws = torch.empty(ops.attn_decode_workspace_bytes(cap, hq, D), dtype=torch.uint8, device=dev)
    ops.attn_decode_plan(plan, seq, B, B, cap, hq, hkv, 64)
    outs = [torch.empty((B, hq, D), dtype=qkv.dtype, device=dev) for _ in range(12)]
    for o in outs:
        ops.attn_decode(o, q, k, v, table, rows, seq, plan, ws, B, B, cap, D ** -0.5, slot_run=256)
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o.cpu(), want)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            for o in outs[:4]:
                ops.attn_decode(o, q, k, v, table, rows, seq, plan, ws, B, B, cap, D ** -0.5, slot_run=256)
        for _ in range(3):
            for o in outs[:4]:
                o.zero_()
            graph.replay()
            torch.cuda.synchronize()
            for o in outs[:4]:
                assert torch.equal(o.cpu(), want)
