"""The native radix prefix cache (mini-sglang_amd/radix.py + csrc/radix.cpp) against the REFERENCE's RadixPrefixCache
(python/minisgl/kvcache/radix_cache.py), both driven by the same random stream of calls under the same clock:
every observable must be equal -- matched lengths, matched indices, insert results, evicted slots and their order,
size_info after each call.  A coarse clock (many equal timestamps) makes the LRU order depend on heapq's tie
behaviour and on the children's dict order, which the native tree reproduces.  Also: the reference's CacheManager
(allocate / free-and-cache / evict, P/scheduler/cache.py) running on top of either cache hands out the same pages."""
from __future__ import annotations

import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

import refdrive

ROOT = Path(__file__).resolve().parent.parent

needs_reference = pytest.mark.skipif(refdrive.reference_root() is None, reason="no importable reference")


def run(code: str, timeout: int = 600) -> str:
    pre = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(refdrive.reference_root())!r}); sys.path.insert(0, {str(ROOT)!r})
        import random, torch
        import mini_sglang_amd.minisgl_plugin as plugin
        plugin.install(gemm_tune="off", native_radix=False, vectorized_glue=False)  # the tests below compare the reference's own pieces with ours
        import minisgl.core as core
        from minisgl.core import Context
        import minisgl.kvcache.radix_cache as rc
        from minisgl.kvcache import create_prefix_cache

        class Clock:
            def __init__(self, div): self.n, self.div = 0, div
            def __call__(self):
                self.n += 1
                return 1000 + self.n // self.div

        def pair(page_size, div):
            core._GLOBAL_CTX = None
            core.set_global_ctx(Context(page_size))
            ref_clock, my_clock = Clock(div), Clock(div)
            rc.time.monotonic_ns = ref_clock          # the reference module's clock (radix_cache.py:27,209)
            ref = create_prefix_cache(torch.device("cpu"), "radix")
            mine = plugin._STATE["native_radix_class"](torch.device("cpu"), clock=my_clock)
            return ref, mine, ref_clock, my_clock
    """)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=str(ROOT / "tests"))
    r = subprocess.run([sys.executable, "-c", pre + textwrap.dedent(code)], env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    return r.stdout


@needs_reference
def test_native_radix_equals_reference_on_random_call_streams():
    out = run("""
        def stream(page_size, div, seed, steps):
            rnd = random.Random(seed)
            ref, mine, c1, c2 = pair(page_size, div)
            vocab = 6
            docs = [[rnd.randrange(vocab) for _ in range(rnd.randrange(1, 12) * page_size + rnd.randrange(page_size))] for _ in range(6)]
            locked = []      # (reference handle, native handle)
            next_slot = 0
            stats = dict(match=0, insert=0, evict=0, unlock=0, splits=0, evicted=0)
            for step in range(steps):
                op = rnd.random()
                if op < 0.40:      # a request arrives: shares a prefix with an earlier document, then diverges
                    base = rnd.choice(docs)
                    cut = rnd.randrange(0, len(base) + 1)
                    ids = base[:cut] + [rnd.randrange(vocab) for _ in range(rnd.randrange(0, 5 * page_size))]
                    if not ids:
                        ids = [rnd.randrange(vocab)]
                    docs[rnd.randrange(len(docs))] = ids
                    t = torch.tensor(ids, dtype=torch.int32)
                    nodes_before = mine.tree.info()[2]
                    a, b = ref.match_prefix(t).cuda_handle, mine.match_prefix(t).cuda_handle
                    stats["splits"] += mine.tree.info()[2] - nodes_before
                    assert a.cached_len == b.cached_len, (step, a.cached_len, b.cached_len)
                    if a.cached_len:
                        assert torch.equal(a.get_matched_indices(), b.get_matched_indices()), step
                    else:
                        assert b.node == 0
                    ref.lock_handle(a); mine.lock_handle(b)
                    locked.append((a, b, t))
                    stats["match"] += 1
                elif op < 0.70 and locked:   # a request finishes: its tokens are cached, the old handle is released
                    a, b, t = locked.pop(rnd.randrange(len(locked)))
                    grown = torch.cat([t, torch.tensor([rnd.randrange(vocab) for _ in range(rnd.randrange(0, 3 * page_size))], dtype=torch.int32)])
                    idx = torch.arange(next_slot, next_slot + len(grown), dtype=torch.int32)
                    next_slot += len(grown)
                    ra, rb = ref.insert_prefix(grown, idx), mine.insert_prefix(grown, idx)
                    assert ra.cached_len == rb.cached_len and ra.handle.cached_len == rb.handle.cached_len, step
                    if ra.handle.cached_len:
                        assert torch.equal(ra.handle.get_matched_indices(), rb.handle.get_matched_indices()), step
                    ref.lock_handle(a, unlock=True); mine.lock_handle(b, unlock=True)
                    stats["insert"] += 1
                elif op < 0.80 and locked:
                    a, b, _ = locked.pop(rnd.randrange(len(locked)))
                    ref.lock_handle(a, unlock=True); mine.lock_handle(b, unlock=True)
                    stats["unlock"] += 1
                else:
                    ev = ref.size_info.evictable_size
                    if ev:
                        size = rnd.randrange(1, ev + 1)
                        ea, eb = ref.evict(size), mine.evict(size)
                        assert torch.equal(ea, eb), (step, ea.tolist(), eb.tolist())
                        stats["evict"] += 1; stats["evicted"] += len(ea)
                assert tuple(ref.size_info) == tuple(mine.size_info), (step, ref.size_info, mine.size_info)
                assert c1.n == c2.n, "the two caches read the clock a different number of times"
                if step % 50 == 0:
                    mine.check_integrity()
            mine.check_integrity()
            return stats

        total = dict()
        for page_size in (1, 4, 16):
            for div in (1, 7, 10 ** 9):      # distinct stamps / frequent ties / every stamp equal
                for seed in range(3):
                    st = stream(page_size, div, seed * 31 + page_size, 500)
                    for k, v in st.items():
                        total[k] = total.get(k, 0) + v
        assert total["evict"] > 300 and total["splits"] > 200 and total["insert"] > 2000 and total["evicted"] > 5000, total
        # the reference's failure modes
        ref, mine, _, _ = pair(4, 1)
        for c in (ref, mine):
            try:
                c.evict(8)
            except AssertionError as e:
                assert "Cannot evict 8, only 0 is evictable" in str(e)
            else:
                raise SystemExit("evict beyond the evictable size must fail")
            assert c.evict(0).numel() == 0
            try:
                c.reset()
            except NotImplementedError:
                pass
        h = mine.insert_prefix(torch.arange(8, dtype=torch.int32), torch.arange(8, dtype=torch.int32)).handle
        mine.evict(8)
        try:
            h.get_matched_indices()
        except Exception as e:
            assert "evicted" in str(e)
        else:
            raise SystemExit("a stale handle must not resolve")
        print("radix parity ok", total)
    """)
    assert "radix parity ok" in out


@needs_reference
def test_cache_manager_hands_out_the_same_pages_over_either_cache():
    out = run("""
        import time, types
        if not torch.cuda.is_available():   # pinned host memory needs a GPU runtime; the values do not depend on pinning
            def _no_pin(fn):
                def wrapped(*a, **k):
                    k.pop("pin_memory", None)
                    return fn(*a, **k)
                return wrapped
            torch.empty, torch.tensor = _no_pin(torch.empty), _no_pin(torch.tensor)
        from minisgl.scheduler.cache import CacheManager
        from minisgl.core import Req, SamplingParams

        def drive(kind, page_size, seed):
            core._GLOBAL_CTX = None
            core.set_global_ctx(Context(page_size))
            time.monotonic_ns = Clock(5)       # both caches read time.monotonic_ns at call time
            num_pages, max_len = 40, 96
            table = torch.zeros((8, max_len), dtype=torch.int32)
            cm = CacheManager(num_pages, page_size, table, type=kind)
            assert type(cm.prefix_cache).__name__ == ("RadixPrefixCache" if kind == "radix" else "NativeRadixPrefixCache")
            rnd = random.Random(seed)
            trace, live, free_rows = [], [], list(range(8))
            for step in range(220):
                if free_rows and (rnd.random() < 0.55 or not live):
                    # P/scheduler/prefill.py flow: match, lock, copy the hit into the table row, allocate the rest
                    base = rnd.choice(live).input_ids.tolist() if live and rnd.random() < 0.7 else []
                    ids = (base[: rnd.randrange(0, len(base) + 1)] + [rnd.randrange(4) for _ in range(rnd.randrange(2, 30))])[: max_len - 8]
                    t = torch.tensor(ids, dtype=torch.int32)
                    handle = cm.match_req(types.SimpleNamespace(input_ids=t, input_len=len(ids))).cuda_handle
                    need = (len(ids) - handle.cached_len + page_size - 1) // page_size * page_size
                    cm.lock(handle)
                    if need > cm.available_size:
                        cm.unlock(handle)
                        trace.append(("skip", handle.cached_len, cm.available_size))
                        continue
                    req = Req(input_ids=t, table_idx=free_rows.pop(), cached_len=handle.cached_len, output_len=4, uid=step,
                              sampling_params=SamplingParams(), cache_handle=handle)
                    if handle.cached_len:
                        table[req.table_idx, : handle.cached_len] = handle.get_matched_indices()
                    cm.allocate_paged([req])
                    req.complete_one()               # the forward was launched: [0, cached_len) is what the request holds
                    trace.append(("alloc", handle.cached_len, table[req.table_idx, : len(ids)].tolist(), len(cm.free_slots)))
                    live.append(req)
                else:
                    req = live.pop(rnd.randrange(len(live)))
                    cm.cache_req(req, finished=True)   # P/scheduler/scheduler.py:_free_req_resources
                    free_rows.append(req.table_idx)
                    trace.append(("free", len(cm.free_slots), tuple(cm.prefix_cache.size_info), cm.free_slots.tolist()))
                if not live:                       # the reference's page count only closes with no request in flight
                    cm.check_integrity()
                cm.prefix_cache.check_integrity()
            return trace

        for page_size in (1, 4):
            a, b = drive("radix", page_size, 11), drive("hip_radix", page_size, 11)
            assert len(a) == len(b)
            for i, (x, y) in enumerate(zip(a, b)):
                assert x == y, (page_size, i, x, y)
            hits = sum(t[0] == "alloc" and t[1] > 0 for t in a)
            assert hits > 10 and any(t[0] == "skip" for t in a) or hits > 10, "no prefix hits in the scenario"
            print(page_size, "allocs", sum(t[0] == "alloc" for t in a), "hits", hits, "skips", sum(t[0] == "skip" for t in a))
        # install(native_radix=True) points the scheduler's default cache type at the native tree
        plugin.install(gemm_tune="off", native_radix=True)
        core._GLOBAL_CTX = None
        core.set_global_ctx(Context(4))
        cm = CacheManager(8, 4, torch.zeros((2, 32), dtype=torch.int32), type="radix")
        assert type(cm.prefix_cache).__name__ == "NativeRadixPrefixCache"
        print("cache manager parity ok")
    """)
    assert "cache manager parity ok" in out


def test_native_radix_tree_c_abi_without_the_reference():
    """The C-ABI of the tree on its own (runs wherever libmsgl_hip.so loads, reference or not): a hand-checked scenario --
    insert, partial match with a page-aligned split, lock / unlock sizes, LRU eviction order, stale ids, ragged keys."""
    import torch

    from mini_sglang_amd._lib import MsglError
    from mini_sglang_amd.radix import NativeRadixTree

    tick = iter(range(100, 10 ** 6))
    t = NativeRadixTree(4, clock=lambda: next(tick))
    ids = lambda *x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    assert t.walk(ids(1, 2, 3, 4, 5, 6, 7, 8)) == (0, 0, None)               # empty tree: root, nothing matched
    a = t.add_child(0, ids(1, 2, 3, 4, 5, 6, 7, 8))                           # node a: two pages
    assert t.info(a) == (8, 0, 2, 8)
    # 6 equal tokens -> one whole page matches -> a is split at 4: head (new id) + tail (a keeps its id)
    node, matched, split = t.walk(ids(1, 2, 3, 4, 5, 6, 9, 9))
    assert matched == 4 and split == (node, a, 4) and t.info(node)[3] == 4 and t.info(a)[3] == 4
    head = node
    b = t.add_child(head, ids(5, 6, 9, 9))                                    # sibling of a's tail under the head
    assert t.path(b) == [head, b] and t.path(a) == [head, a]
    assert t.walk(ids(1, 2, 3))[:2] == (0, 0)                                 # shorter than a page: no lookup
    t.lock(b, unlock=False)
    assert t.info()[:2] == (4, 8)                                             # head + b protected, a's tail evictable
    with pytest.raises(AssertionError, match="Cannot evict 8, only 4 is evictable"):
        t.evict(8)
    assert t.evict(1) == [a]                                                  # the only unreferenced leaf
    with pytest.raises(MsglError, match="unknown node"):
        t.path(a)
    t.lock(b, unlock=True)
    assert t.info()[:2] == (8, 0)
    c = t.add_child(0, ids(7, 7, 7, 7))
    t.walk(ids(1, 2, 3, 4, 5, 6, 9, 9))                                       # touches head and b: c is now the oldest
    assert t.evict(4) == [c]
    assert t.evict(8) == [b, head]                                            # leaf first, then its parent becomes a leaf
    assert t.info()[:3] == (0, 0, 1)
    with pytest.raises(MsglError, match="whole number of pages"):
        t.add_child(0, ids(1, 2, 3))
    with pytest.raises(MsglError, match="not locked"):
        t.lock(t.add_child(0, ids(1, 1, 1, 1)), unlock=True)
    t.check()
    t.close()
