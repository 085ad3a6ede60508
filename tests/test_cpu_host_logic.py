"""CPU checks of host-side arithmetic the GPU path depends on: KV-pool sizing at 288 GB (BASELINE.json
configs[4]), graph batch-size list, the benchmark workload generators (SURVEY.md 8d figures), the
search space of the decode-batch GEMM kernel."""
import random

import pytest
import torch

GIB = 1 << 30


def test_kv_pool_sizing_at_288gb_matches_survey_table():
    """determine_num_pages = (memory_ratio * free - model_bytes) // bytes_per_page (P/engine/engine.py:148-168).
    SURVEY.md section 8: KV bytes/token/rank 160 / 40 / 64 / 40 KiB; Llama-3.1-70B TP8 ~ 5.9 M tokens per rank."""
    from mini_sglang_amd.engine import EngineConfig, determine_num_pages
    from mini_sglang_amd.model import PRESETS

    free = 288 * 10 ** 9  # what the part advertises; the formula is linear in it
    cases = [  # model, tp, weight bytes per rank (SURVEY table, GiB), KV bytes per token per rank
        ("qwen3-14b", 1, 27.5, 160 * 1024), ("qwen3-14b", 4, 6.9, 40 * 1024),
        ("qwen3-32b", 4, 15.3, 64 * 1024), ("llama-3.1-70b", 8, 16.4, 40 * 1024),
    ]
    for name, tp, w_gib, kv_tok in cases:
        for page in (1, 16, 256):
            cfg = EngineConfig(model=PRESETS[name], tp_size=tp, page_size=page, memory_ratio=0.9)
            w_bytes = int(w_gib * GIB)
            pages = determine_num_pages(free, free - w_bytes, cfg)
            assert pages == (int(0.9 * free) - w_bytes) // (kv_tok * page)
    cfg = EngineConfig(model=PRESETS["llama-3.1-70b"], tp_size=8, page_size=1)
    tokens = determine_num_pages(free, free - int(16.4 * GIB), cfg)
    assert 5.8e6 < tokens < 6.0e6
    cfg = EngineConfig(model=PRESETS["qwen3-14b"], tp_size=1, page_size=1, num_page_override=1234)
    assert determine_num_pages(free, free, cfg) == 1234


def test_graph_batch_sizes_match_reference_rule():
    """P/engine/graph.py:49-67: [1, 2, 4] + range(8, max + 1, 8); max 256 above 80 GiB free, else 160."""
    from mini_sglang_amd.engine import determine_graph_bs

    assert determine_graph_bs(None, None, 200 * GIB) == [1, 2, 4] + list(range(8, 257, 8))
    assert determine_graph_bs(None, None, 64 * GIB)[-1] == 160
    assert determine_graph_bs(None, 20, 200 * GIB) == [1, 2, 4, 8, 16]
    assert determine_graph_bs(None, 2, 200 * GIB) == [1, 2]
    assert determine_graph_bs([3, 5], 256, 200 * GIB) == [3, 5]
    assert determine_graph_bs(None, 0, 200 * GIB) == []


def test_offline_benchmark_workload_matches_survey_figures():
    """benchmark/offline/bench.py:11-31 replicated: random.seed(0); 256 x (ids, len 100..1024), then
    256 x max_tokens 100..1024.  SURVEY.md 8d: sum in = 142 827, sum out = 133 966."""
    random.seed(0)
    prompts = [[random.randint(0, 10000) for _ in range(random.randint(100, 1024))] for _ in range(256)]
    outs = [random.randint(100, 1024) for _ in range(256)]
    assert sum(len(p) for p in prompts) == 142827 and sum(outs) == 133966
    assert min(len(p) for p in prompts) == 107 and max(len(p) for p in prompts) == 1024


def test_bench_contexts_are_the_token_weighted_distribution():
    import bench

    ctx = bench.bench_contexts(256)
    assert ctx == bench.bench_contexts(256) and len(ctx) == 256
    assert 100 <= min(ctx) and max(ctx) < 2048
    assert 850 < sum(ctx) / 256 < 1000  # mean context per decoded token 902.8
    assert len(bench.bench_contexts(8)) == 8


def test_skinny_gemm_search_space_respects_kernel_limits():
    """(k-slices, row tiles) offered by ops.skinny_candidates must satisfy csrc/gemm_skinny.hip's checks."""
    from mini_sglang_amd import ops

    for M in (1, 16, 17, 32, 33, 64):
        mt = 1 if M <= 16 else 2 if M <= 32 else 4
        for N, K in ((5120, 5120), (48, 64), (34816, 5120), (5120, 17408), (16, 128)):
            cands = ops.skinny_candidates(M, N, K)
            assert cands and len(set(cands)) == len(cands)
            for sl, nt in cands:
                if sl <= 0:  # the row-streaming kernel (csrc/gemm_rowstream.hip): (-variant, loads in flight per lane)
                    assert M <= 8 and K % (128 if sl else 512) == 0 and nt in (8, 16) and ops.rowstream_supported(M, N, K, 0, -sl)
                    continue
                assert N % (16 * nt) == 0 and 1 <= sl <= K // 64
                assert sl <= (16 if mt * nt <= 2 else 8 if mt * nt <= 8 else 4)
            assert ((0, 16) in cands) == (M <= 8 and K % 512 == 0) and ((-1, 8) in cands) == (M <= 8 and K % 128 == 0 and N % 4 == 0)
    assert not ops.skinny_supported(65, 5120, 5120) and not ops.skinny_supported(8, 40, 128)
    assert not ops.skinny_supported(8, 48, 100) and ops.skinny_supported(64, 48, 64)
    # the host-side shape check of the row-streaming kernel: staged x (padded to 1 / 2 / 4 / 8 rows) + [rows per CU][8 waves][rows of x]
    # fp32 partial sums within 159 KB of LDS; the fused add + RMSNorm staging only for the hidden sizes rmsnorm_wide_row_kernel runs
    assert ops.rowstream_supported(4, 151936, 5120) and not ops.rowstream_supported(8, 151936, 5120)
    assert ops.rowstream_supported(4, 5120, 17408, ops.ROWSTREAM_SILU_INTERLEAVED) and not ops.rowstream_supported(5, 5120, 17408)
    assert ops.rowstream_supported(4, 7168, 5120, ops.ROWSTREAM_ADD_NORM) and not ops.rowstream_supported(5, 7168, 5120, ops.ROWSTREAM_ADD_NORM)
    assert not ops.rowstream_supported(1, 7168, 1024, ops.ROWSTREAM_ADD_NORM) and not ops.rowstream_supported(1, 64, 576)
    assert ops.rowstream_supported(8, 5120, 4352, 0, ops.ROWSTREAM_MATRIX) and not ops.rowstream_supported(8, 5122, 4352, 0, ops.ROWSTREAM_MATRIX)
    assert ops.rowstream_planned(1, torch.zeros((64, 512), dtype=torch.bfloat16)) is None  # no plan, no folding
    assert ops.linear.__doc__ and not ops._SKINNY_PLAN  # nothing is planned until skinny_tune ran on a GPU


def test_page_allocator_reproduces_the_reference_cache_manager_trace(golden_dir):
    """a12: the product driver's page allocation (offline.PageAllocator, used by OfflineRunner._allocate_paged) against
    the trace recorded from the REFERENCE's CacheManager + RadixPrefixCache (tests/golden/make_golden.py,
    gen_cache_allocate: P/scheduler/cache.py:42-53,106-146).  The driver has no prefix cache, so per trace step it is
    given what the reference had at that point -- the free list left by the previous step and the matched prefix
    length -- and must hand out exactly the same token slots for the prefill and for each of the three decode steps
    (pages taken from the head of the free list in order, expanded page-aligned, one new page when a decode step
    crosses a page boundary), leaving exactly the reference's free list minus what the radix cache gave back."""
    import types

    import numpy as np

    from mini_sglang_amd.offline import PageAllocator

    gold = torch.load(golden_dir / "cache_allocate.pt")
    checked = 0
    for key, ps in (("page1", 1), ("page4", 4)):
        free_before = torch.arange(gold[key]["num_pages"], dtype=torch.int32) * ps
        for tr in gold[key]["trace"]:
            n, cached, row = len(tr["input_ids"]), tr["matched"], tr["table_idx"]
            alloc = PageAllocator(0, ps)
            alloc.free_slots = free_before.numpy().copy()
            table = np.zeros((8, 64), dtype=np.int32)
            table[row, :cached] = tr["prefill_row"][:cached].numpy()  # the matched prefix comes from the radix tree
            req = types.SimpleNamespace(table_idx=row, cached_len=cached, device_len=n)

            def step():
                got = alloc.allocate([req])
                if got is not None:
                    rows, pos, tok = got
                    table[rows, pos] = tok

            try:
                step()
                assert np.array_equal(table[row, :n], tr["prefill_row"].numpy()), (key, tr["step"])
                for _ in range(3):  # complete_one, then the next decode step allocates [cached_len, device_len)
                    req.cached_len, req.device_len = req.device_len, req.device_len + 1
                    step()
            except RuntimeError:  # the reference evicted from its radix cache here; the driver has none
                free_before = tr["free_slots"]
                continue
            assert np.array_equal(table[row, : req.device_len], tr["final_row"].numpy()), (key, tr["step"])
            # what the driver still holds free is a prefix-preserving part of the reference's next free list: the
            # reference appends what cache_req frees (cache.py:71-79) behind the untouched remainder
            rest = alloc.free_slots
            assert np.array_equal(tr["free_slots"].numpy()[: len(rest)], rest), (key, tr["step"])
            free_before = tr["free_slots"]
            checked += 1
    assert checked >= 8, checked

    # freeing appends in take order (cache.py:115-119), rows are recycled
    a = PageAllocator(8, 2)
    r0 = types.SimpleNamespace(table_idx=3, cached_len=0, device_len=5)
    r1 = types.SimpleNamespace(table_idx=1, cached_len=0, device_len=2)
    rows, pos, tok = a.allocate([r0, r1])
    assert tok.tolist() == [0, 1, 2, 3, 4, 5, 6, 7] and rows.tolist() == [3] * 6 + [1] * 2 and pos.tolist() == [0, 1, 2, 3, 4, 5, 0, 1]
    a.free_row(3)
    assert a.free_slots.tolist() == [8, 10, 12, 14, 0, 2, 4]
    with pytest.raises(RuntimeError):
        a.allocate([types.SimpleNamespace(table_idx=0, cached_len=0, device_len=100)])


def test_bench_self_launch_command_is_the_drivers_contract_line():
    """`python3 bench.py --gpus N` outside a launcher re-executes itself through torch.distributed.run: one process per GPU,
    loopback rendezvous on a free port, this file's own arguments, the dmabuf IPC mode exported."""
    from pathlib import Path

    import bench

    cmd, env = bench.self_launch_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and cmd[i + 2] == "--master-port" and 1024 <= int(cmd[i + 3]) < 65536
    j = cmd.index(str(Path(bench.__file__).resolve()))
    assert cmd[j + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
    assert env["MSGL_BENCH_SELF_LAUNCHED"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_p2p_error_word_decodes_to_a_message_on_both_sides():
    """csrc/comm_p2p.hip `Error word`: bits 0-3 = 1 + phase, 4-7 = collective kind, 8-15 = block, 16-19 = peer, bit 20 = the
    copy a failing rank wrote into a peer's header."""
    from mini_sglang_amd.kernel import describe_p2p_error

    own = (1 + 0) | (2 << 4) | (5 << 8) | (3 << 16)
    m = describe_p2p_error(own, 0, 4)
    assert "rank 0 of 4 gave up waiting for a peer (rank 3)" in m and "two-shot all-reduce" in m and "phase 0, block 5" in m
    told = (1 + 2) | (3 << 4) | (9 << 8) | (1 << 16) | (1 << 20)
    m = describe_p2p_error(told, 2, 4)
    assert "rank 1 gave up waiting" in m and "told rank 2 of 4" in m and "fused all-reduce + add + RMSNorm" in m and "phase 2" in m


def test_synthetic_qwen_trace_is_deterministic_and_in_range():
    import refdrive

    a, b = refdrive.synth_qwen_trace(50, 5.0), refdrive.synth_qwen_trace(50, 5.0)
    assert a == b and len(a) == 50
    assert all(x["t"] <= y["t"] for x, y in zip(a, a[1:]))
    assert all(16 <= r["input_length"] <= 6000 and 8 <= r["output_length"] <= 1000 for r in a)
    assert 5.0 < a[-1]["t"] < 20.0  # ~50 arrivals at 5 / s


def test_ro_plan_arithmetic_matches_the_kernel_and_covers_every_unit():
    """Host side of the row-owner projection (csrc/gemm_ro.hip): the Python accumulator budget equals the library's, every
    candidate plan respects it, and the balanced tile cut the kernel computes (tile t = units [t U / T, (t + 1) U / T)) covers
    every 16-row unit exactly once with widths that differ by at most one."""
    from mini_sglang_amd import _lib, ops

    for M in (3, 16, 64, 128, 129, 200, 256):
        assert _lib.lib().msgl_ro_gemm_max_units(M) == ops.ro_max_units(M)
    for (M, N, K) in [(256, 34816, 5120), (256, 5120, 17408), (128, 7168, 5120), (64, 151936, 5120), (9, 16, 64), (200, 2064, 640),
                      (256, 8704, 5120), (256, 1792, 5120)]:
        units, umax = N // 16, ops.ro_max_units(M)
        cands = ops.ro_candidates(M, N, K, 256)
        assert cands, (M, N, K)
        for tiles, slices in cands:
            cut = [t * units // tiles for t in range(tiles + 1)]
            widths = [b - a for a, b in zip(cut[:-1], cut[1:])]
            assert cut[0] == 0 and cut[-1] == units and min(widths) >= 1 and max(widths) <= umax, (M, N, K, tiles)
            assert max(widths) - min(widths) <= 1 and 1 <= slices <= K // 64
        assert all(s == 1 for _, s in ops.ro_candidates(M, N, K, 256, silu=True))
    assert _lib.lib().msgl_ro_gemm_workspace_bytes(256, 5120, 6) == 6 * 256 * 5120 * 4
