"""Generate golden fixtures by IMPORTING the reference's own Python (runs only where
/root/reference exists, i.e. the build container; the fixtures it writes are committed and
are what the tests read -- nothing under tests/ touches /root/reference at run time).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is pinned (everything on this path that the reference can compute without a GPU):
  rope_cache.pt       RotaryEmbedding._cos_sin_cache rows   (P/layers/rotary.py:12-114)
  sampler_prepare.pt  Sampler.prepare outputs               (P/engine/sample.py:53-68)
  fa_metadata.pt      FlashAttentionBackend.prepare_metadata (P/attention/fa.py:67-105)
  cache_allocate.pt   CacheManager.allocate_paged page tables + radix traces
                      (P/scheduler/cache.py:42-146, P/kvcache/radix_cache.py) incl. every
                      (x, y) -> fast_compare_key call it made
  indexing.pt         ref_indexing of tests/kernel/test_index.py:13-30 (masked gather)
  store.pt            the index-assign baseline of tests/kernel/test_store.py:41-44
"""
from __future__ import annotations

import ast
import os
import sys
import types
from pathlib import Path

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import torch  # noqa: E402

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent

# ---- stubs for modules the reference imports but this image lacks (import-only) -------------
zmq = types.ModuleType("zmq")
for name in ("PUSH", "PULL", "PUB", "SUB", "SUBSCRIBE"):
    setattr(zmq, name, 0)
zmq.Context = type("Context", (), {})
zmq_asyncio = types.ModuleType("zmq.asyncio")
zmq_asyncio.Context = type("Context", (), {})
zmq.asyncio = zmq_asyncio
sys.modules["zmq"] = zmq
sys.modules["zmq.asyncio"] = zmq_asyncio

fi = types.ModuleType("flashinfer")  # constructors bind these names; never called here
for name in ("rmsnorm", "fused_add_rmsnorm", "apply_rope_with_cos_sin_cache_inplace", "silu_and_mul",
             "gelu_and_mul"):
    setattr(fi, name, lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub")))
sys.modules["flashinfer"] = fi

sys.path.insert(0, str(REF / "python"))

# pinned host memory needs a GPU runtime; the values do not depend on pinning
_orig_tensor, _orig_empty = torch.tensor, torch.empty


def _no_pin(fn):
    def wrapped(*a, **k):
        k.pop("pin_memory", None)
        return fn(*a, **k)

    return wrapped


torch.tensor = _no_pin(_orig_tensor)
torch.empty = _no_pin(_orig_empty)
torch.arange = _no_pin(torch.arange)
torch.ones = _no_pin(torch.ones)


def gen_rope():
    from minisgl.layers.rotary import _get_rope

    cases = {
        "qwen3_default": dict(head_dim=128, rotary_dim=128, max_position=40960, base=1000000.0, rope_scaling=None),
        "llama3": dict(head_dim=128, rotary_dim=128, max_position=131072, base=500000.0,
                       rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0,
                                         high_freq_factor=4.0, original_max_position_embeddings=8192)),
        "yarn": dict(head_dim=128, rotary_dim=128, max_position=131072, base=1000000.0,
                     rope_scaling=dict(rope_type="yarn", factor=4.0, original_max_position_embeddings=32768)),
        "hd64_default": dict(head_dim=64, rotary_dim=64, max_position=8192, base=10000.0, rope_scaling=None),
    }
    out = {}
    for name, kw in cases.items():
        rope = _get_rope(**kw)
        cache = rope._cos_sin_cache
        pos = torch.tensor([0, 1, 2, 3, 7, 63, 100, 1000, 4095, kw["max_position"] // 2, kw["max_position"] - 1])
        out[name] = dict(kwargs=kw, positions=pos, rows=cache[pos].clone(), shape=tuple(cache.shape))
    torch.save(out, OUT / "rope_cache.pt")


def gen_sampler():
    from minisgl.core import SamplingParams
    from minisgl.engine.sample import Sampler

    V = 151936
    sets = {
        "all_greedy": [SamplingParams(), SamplingParams(temperature=0.7, top_k=1)],
        "temp_only": [SamplingParams(temperature=0.6), SamplingParams(temperature=1.3)],
        "mixed": [SamplingParams(), SamplingParams(temperature=0.6, top_k=50),
                  SamplingParams(temperature=0.9, top_p=0.8), SamplingParams(temperature=0.0, top_p=0.5),
                  SamplingParams(temperature=2.0, top_k=0, top_p=0.0), SamplingParams(temperature=1e-9, top_k=V)],
        "topp_only": [SamplingParams(temperature=1.0, top_p=0.95), SamplingParams(temperature=1.0)],
    }
    out = {}
    for name, ps in sets.items():
        batch = types.SimpleNamespace(reqs=[types.SimpleNamespace(sampling_params=p) for p in ps])
        args = Sampler(torch.device("cpu"), V).prepare(batch)
        out[name] = dict(
            params=[(p.temperature, p.top_k, p.top_p) for p in ps],
            is_greedy=[p.is_greedy for p in ps],
            temperatures=args.temperatures, top_k=args.top_k, top_p=args.top_p,
        )
    torch.save(dict(vocab=V, sets=out), OUT / "sampler_prepare.pt")


def gen_fa_metadata():
    import minisgl.core as core
    from minisgl.attention.fa import FlashAttentionBackend
    from minisgl.core import Batch, Context, Req

    out = {}
    for name, page_size, specs, phase in [
        # (table_idx, cached_len, device_len)
        ("decode_p1", 1, [(0, 9, 10), (3, 99, 100), (1, 0, 1), (7, 511, 512)], "decode"),
        ("prefill_nohit_p1", 1, [(2, 0, 17), (0, 0, 5), (5, 0, 130)], "prefill"),
        ("extend_hit_p1", 1, [(2, 16, 40), (0, 0, 5), (5, 64, 65)], "prefill"),
        ("decode_p16", 16, [(0, 31, 32), (4, 32, 33), (2, 200, 201)], "decode"),
        ("extend_hit_p16", 16, [(1, 32, 100), (6, 0, 47)], "prefill"),
    ]:
        core._GLOBAL_CTX = None
        ctx = Context(page_size)
        g = torch.Generator().manual_seed(len(name))
        rows, width = 8, 544
        # page-aligned token slots like CacheManager would write them
        n_pages = width // page_size
        table = torch.stack([
            (torch.randperm(4096, generator=g)[:n_pages].to(torch.int32) * page_size).repeat_interleave(page_size)
            + torch.arange(page_size, dtype=torch.int32).repeat(n_pages) for _ in range(rows)])
        ctx.page_table = table
        ctx.kv_cache = types.SimpleNamespace(device=torch.device("cpu"))
        core.set_global_ctx(ctx)
        reqs = []
        for (ti, cl, dl) in specs:
            r = Req(input_ids=torch.zeros(dl, dtype=torch.int32), table_idx=ti, cached_len=cl, output_len=4,
                    uid=ti, sampling_params=None, cache_handle=None)
            reqs.append(r)
        batch = Batch(reqs=reqs, phase=phase)
        batch.padded_reqs = reqs
        backend = FlashAttentionBackend(types.SimpleNamespace(head_dim=128))
        backend.prepare_metadata(batch)
        md = batch.attn_metadata
        out[name] = dict(page_size=page_size, specs=specs, table=table.clone(), cu_seqlens_k=md.cu_seqlens_k,
                         cu_seqlens_q=md.cu_seqlens_q, cache_seqlens=md.cache_seqlens,
                         max_seqlen_k=md.max_seqlen_k, max_seqlen_q=md.max_seqlen_q,
                         page_table=md.page_table.clone(), last_indices=md.get_last_indices(len(reqs)))
    core._GLOBAL_CTX = None
    torch.save(out, OUT / "fa_metadata.pt")


def gen_cache_allocate(use_product_compare: bool = False):
    """Drive the reference CacheManager + RadixPrefixCache on CPU (the scenario family of
    tests/core/test_cache_allocate.py) and record page tables, free lists and every key
    compare the radix tree performed."""
    import minisgl.core as core
    import minisgl.kernel.radix as kradix
    from minisgl.core import Context, Req
    from minisgl.scheduler.cache import CacheManager

    calls = []

    def py_compare(x, y):  # C/src/radix.cpp:19-40 restated; results recorded as golden
        n = min(len(x), len(y))
        neq = (x[:n] != y[:n]).nonzero()
        r = int(neq[0]) if neq.numel() else n
        if len(calls) < 64:
            calls.append((x.clone(), y.clone(), r))
        return r

    import minisgl.kernel as mk

    if use_product_compare:  # the drop-in: mini_sglang_amd's C-ABI function behind minisgl.kernel
        sys.path.insert(0, str(OUT.parent.parent))
        from mini_sglang_amd import minisgl_plugin

        minisgl_plugin.install()
        assert mk.fast_compare_key.__module__.startswith("mini_sglang_amd")
    else:
        kradix.fast_compare_key = py_compare
        mk.fast_compare_key = py_compare

    out = {}
    for page_size in (1, 4):
        core._GLOBAL_CTX = None
        core.set_global_ctx(Context(page_size))
        num_pages, max_len = 64, 64
        table = torch.zeros((8, max_len), dtype=torch.int32)
        cm = CacheManager(num_pages, page_size, table, type="radix")
        trace = []
        g = torch.Generator().manual_seed(7 + page_size)
        prompts = [torch.randint(0, 50, (n,), generator=g, dtype=torch.int32) for n in (13, 22, 9, 30)]
        prompts.append(torch.cat([prompts[1][:16], torch.randint(50, 99, (7,), generator=g, dtype=torch.int32)]))
        prompts.append(prompts[0].clone())
        with torch.no_grad():
            for step, ids in enumerate(prompts):
                ti = step % 8
                pending = types.SimpleNamespace(input_ids=ids, input_len=len(ids))
                handle = cm.match_req(pending).cuda_handle
                cm.lock(handle)
                cached = handle.cached_len
                if cached:
                    table[ti, :cached] = handle.get_matched_indices()
                req = Req(input_ids=ids, table_idx=ti, cached_len=cached, output_len=8, uid=step,
                          sampling_params=None, cache_handle=handle)
                cm.allocate_paged([req])  # _prepare_batch of the prefill forward
                prefill_row = table[ti, : req.device_len].clone()
                req.complete_one()  # engine.forward_batch (P/engine/engine.py:199-200)
                req.append_host(torch.tensor([1], dtype=torch.int32))
                for _ in range(3):  # decode forwards: allocate -> forward -> complete_one
                    cm.allocate_paged([req])
                    req.complete_one()
                    req.append_host(torch.tensor([1], dtype=torch.int32))
                row = table[ti, : req.cached_len].clone()
                cm.cache_req(req, finished=True)
                cm.check_integrity()
                trace.append(dict(step=step, input_ids=ids.clone(), table_idx=ti, matched=cached,
                                  prefill_row=prefill_row, final_row=row,
                                  free_slots=cm.free_slots.clone(),
                                  evictable=cm.prefix_cache.size_info.evictable_size))
        out[f"page{page_size}"] = dict(num_pages=num_pages, trace=trace)
    out["compare_calls"] = calls
    core._GLOBAL_CTX = None
    if use_product_compare:
        return out
    torch.save(out, OUT / "cache_allocate.pt")


def gen_indexing_and_store():
    src = (REF / "tests/kernel/test_index.py").read_text()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "ref_indexing")
    ns = {"torch": torch, "F": torch.nn.functional, "Tuple": tuple}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "ref_indexing", "exec"), ns)
    ref_indexing = ns["ref_indexing"]
    g = torch.Generator().manual_seed(11)
    V, E, TP = 512, 64, 4
    w = torch.randn((V, E), generator=g).to(torch.float16)
    idx = torch.randint(0, V, (97,), generator=g, dtype=torch.int32)
    mask = (V // TP, V // TP)  # (start, length) exactly as tests/kernel/test_index.py:74-75
    out = dict(weights=w, indices=idx, plain=ref_indexing(w, idx.clone()),
               mask_range=mask, masked=ref_indexing(w, idx.clone(), vocab_range=mask))
    # shard-local weights as VocabParallelEmbedding holds them (P/layers/embedding.py:25-31)
    w_local = w[: V // TP]
    out["masked_local_weights"] = w_local
    out["masked_local"] = ref_indexing(w_local, idx.clone(), vocab_range=mask)
    torch.save(out, OUT / "indexing.pt")

    # tests/kernel/test_store.py:15-34: strided views of an interleaved cache and of a fused qkv
    H = 128
    kv_cache = torch.randn((96, 2, H), generator=g).to(torch.float16)
    before = kv_cache.clone()
    k_cache, v_cache = kv_cache[:, 0, :], kv_cache[:, 1, :]
    indices = torch.randperm(96, generator=g)[:33].to(torch.int32)
    qkv = torch.randn((33, H * 4), generator=g).to(torch.float16)
    k, v = qkv[:, :H], qkv[:, H: H * 2]
    k_cache[indices.long()] = k  # the reference test's own baseline (index assignment)
    v_cache[indices.long()] = v
    torch.save(dict(before=before, indices=indices, qkv=qkv, after=kv_cache.clone()), OUT / "store.pt")


def check_plugin() -> None:
    """The reference's radix cache + CacheManager driven through the plugin's fast_compare_key must
    reproduce the committed golden trace bit for bit; the registry must know the `hip` backend."""
    got = gen_cache_allocate(use_product_compare=True)
    gold = torch.load(OUT / "cache_allocate.pt")
    for key in ("page1", "page4"):
        for a, b in zip(got[key]["trace"], gold[key]["trace"]):
            for f in ("matched", "table_idx"):
                assert a[f] == b[f], (key, a["step"], f)
            for f in ("prefill_row", "final_row", "free_slots"):
                assert torch.equal(a[f], b[f]), (key, a["step"], f)
            assert a["evictable"] == b["evictable"]
    from minisgl.attention import SUPPORTED_ATTENTION_BACKENDS, validate_attn_backend

    assert "hip" in SUPPORTED_ATTENTION_BACKENDS.supported_names()
    validate_attn_backend("hip,hip")
    print("plugin check ok")


if __name__ == "__main__":
    assert REF.exists(), "run this where /root/reference is mounted"
    if "--check-plugin" in sys.argv:
        check_plugin()
        sys.exit(0)
    gen_rope()
    gen_sampler()
    gen_fa_metadata()
    gen_cache_allocate()
    gen_indexing_and_store()
    for f in sorted(OUT.glob("*.pt")):
        print(f"{f.name}: {f.stat().st_size} bytes")
