"""CPU checks against the reference's own code:

* `oracle/_ref/libref_radix.so` is /root/reference/python/minisgl/kernel/csrc/src/radix.cpp compiled unmodified
  (oracle/build_ref.sh); the product's `msgl_fast_compare_key` and the plain-C oracle must agree with it on
  every input, and fail where it fails.
* `minisgl_plugin.install()` against the reference's Python package: every seam the INTEGRATION table lists is
  actually rebound (F.linear proxy, fused AttentionLayer.forward, pre-capture GEMM search, kernels, registry).
"""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest
import torch

import refdrive
from oracle import bytes_c, ref_native

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not ref_native.available(), reason="oracle/_ref/libref_radix.so not built")
def test_fast_compare_key_matches_the_compiled_reference():
    from mini_sglang_amd import ops

    g = torch.Generator().manual_seed(0)
    cases = []
    for dt in (torch.int32, torch.int64):
        for n, m in [(0, 0), (0, 5), (1, 1), (7, 7), (64, 33), (1000, 1000), (4097, 40000)]:
            x = torch.randint(0, 50000, (n,), generator=g).to(dt)
            y = torch.randint(0, 50000, (m,), generator=g).to(dt)
            k = min(n, m)
            for cut in {0, min(1, k), k // 2, max(k - 1, 0), k}:
                y2 = y.clone()
                y2[:cut] = x[:cut]
                cases.append((x, y2))
            cases.append((x, x.clone()))
    assert len(cases) > 60
    for x, y in cases:
        want = ref_native.fast_compare_key(x, y)
        assert ops.fast_compare_key(x, y) == want
        assert bytes_c.compare_key(x, y) == want
    # error behaviour (C/src/radix.cpp:22-23): mixed dtypes, non-contiguous, wrong rank
    a, b = torch.arange(8, dtype=torch.int32), torch.arange(8, dtype=torch.int64)
    for bad in [(a, b), (a[::2], a), (a.view(2, 4), a)]:
        with pytest.raises(RuntimeError):
            ref_native.fast_compare_key(*bad)
        with pytest.raises(RuntimeError):
            ops.fast_compare_key(*bad)


@pytest.mark.skipif(refdrive.reference_root() is None, reason="no importable reference")
def test_install_rebinds_every_seam_of_the_reference():
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(refdrive.reference_root())!r}); sys.path.insert(0, {str(ROOT)!r})
        import torch
        import mini_sglang_amd.minisgl_plugin as plugin
        plugin.install(gemm_tune="off")
        import minisgl.kernel as mk, minisgl.layers.linear as ll, minisgl.layers.embedding as le
        from minisgl.layers.attention import AttentionLayer
        from minisgl.engine.graph import GraphRunner
        from minisgl.attention import SUPPORTED_ATTENTION_BACKENDS, validate_attn_backend
        from mini_sglang_amd import kernel as k
        assert mk.store_cache is k.store_cache and mk.indexing is k.indexing and mk.init_pynccl is k.init_pynccl
        assert validate_attn_backend("hip") == "hip" and validate_attn_backend("hip,hip")
        assert type(ll.F).__name__ == "_FunctionalProxy" and le.F is ll.F
        x, w, b = torch.randn(3, 8), torch.randn(5, 8), torch.randn(5)
        assert torch.equal(ll.F.linear(x, w, b), torch.nn.functional.linear(x, w, b))   # CPU / bias: torch
        assert ll.F.silu is torch.nn.functional.silu                                     # everything else: torch
        assert AttentionLayer.forward._msgl_fused and GraphRunner._capture_graphs._msgl_tuned
        import flashinfer, flashinfer.sampling
        for n in ("rmsnorm", "fused_add_rmsnorm", "apply_rope_with_cos_sin_cache_inplace", "silu_and_mul", "gelu_and_mul"):
            assert callable(getattr(flashinfer, n))
        # the op tree walk finds the five projection shapes of a dense model
        from minisgl.distributed import set_tp_info
        from minisgl.layers import set_rope_device
        from minisgl.models import ModelConfig, create_model
        from minisgl.utils import cached_load_hf_config
        import refdrive, tempfile, pathlib
        d = refdrive.write_model_dir(pathlib.Path(tempfile.mkdtemp()) / "tiny", "tiny", weights=False, max_position=512)
        set_tp_info(rank=0, size=1); set_rope_device(torch.device("cpu"))
        model = create_model(ModelConfig.from_hf(cached_load_hf_config(str(d))))
        groups = plugin._projection_groups(model, require_device=False)
        got = sorted((n, tuple(ws[0].shape), len(ws), kk) for n, ws, kk in groups)
        # plans are keyed by shape: at the tiny dims the LM head (1024, 256) shares gate_up's shape and group
        want = sorted([("qkv", (1792, 256), 2, 256), ("o", (256, 1280), 2, 1280), ("gate_up", (1024, 256), 3, 256),
                       ("down", (256, 512), 2, 512)])
        assert got == want, got
        plugin.install(gemm_tune="off")   # idempotent
        # opt-in: decode batches in uid order (the reference iterates a set of eq=False objects)
        from minisgl.core import Req, SamplingParams
        from minisgl.scheduler.decode import DecodeManager
        dm = DecodeManager(page_size=1)
        reqs = [Req(input_ids=torch.zeros(3, dtype=torch.int32), table_idx=i, cached_len=0, output_len=4, uid=u,
                    sampling_params=SamplingParams(), cache_handle=None) for i, u in enumerate([5, 2, 9, 0, 7])]
        dm.filter_reqs(reqs)
        plugin.install(gemm_tune="off", deterministic_decode_order=True)
        assert [r.uid for r in dm.schedule_next_batch().reqs] == [0, 2, 5, 7, 9]
        print("seams ok")
    """)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=str(ROOT / "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "seams ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(refdrive.reference_root() is None, reason="no importable reference")
def test_vectorized_scheduler_glue_equals_the_reference_functions():
    """sched_glue.make_positions / make_input_tuple / make_write_tuple against `_make_positions`, `_make_input_tuple`,
    `_make_write_tuple` of the reference (P/scheduler/scheduler.py:236-267) on random prefill, chunked and decode
    batches incl. padded dummy requests: equal dtype, shape and values; plus the host time of both at a 256-request
    decode batch."""
    code = textwrap.dedent(f"""
        import sys, time, random, types
        sys.path.insert(0, {str(refdrive.reference_root())!r}); sys.path.insert(0, {str(ROOT)!r})
        import torch
        if not torch.cuda.is_available():   # pinned host memory needs a GPU runtime; the values do not depend on pinning
            def _no_pin(fn):
                def wrapped(*a, **k):
                    k.pop("pin_memory", None)
                    return fn(*a, **k)
                return wrapped
            torch.empty, torch.tensor = _no_pin(torch.empty), _no_pin(torch.tensor)
        import mini_sglang_amd.minisgl_plugin as plugin
        plugin.install(gemm_tune="off", vectorized_glue=False)  # `ref` below must be the reference's own functions
        import minisgl.scheduler.scheduler as sched
        from minisgl.core import Batch, Req, SamplingParams
        from mini_sglang_amd import sched_glue
        ref = (sched._make_positions, sched._make_input_tuple, sched._make_write_tuple)
        rnd = random.Random(5)
        dev = torch.device("cpu")

        def make_batch(n, phase, pad):
            reqs = []
            for i in range(n):
                dl = rnd.randint(2, 300)
                cl = dl - 1 if phase == "decode" else rnd.randint(0, dl - 1)
                reqs.append(Req(input_ids=torch.zeros(dl, dtype=torch.int32), table_idx=rnd.randrange(64), cached_len=cl,
                                output_len=rnd.choice([0, 0, 3]), uid=i, sampling_params=SamplingParams(), cache_handle=None))
            b = Batch(reqs=reqs, phase=phase)
            dummy = Req(input_ids=torch.zeros(1, dtype=torch.int32), table_idx=64, cached_len=0, output_len=1, uid=-1,
                        sampling_params=SamplingParams(), cache_handle=None)
            b.padded_reqs = reqs + [dummy] * pad
            return b

        def same(a, b):
            assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), (a, b)

        for trial in range(200):
            b = make_batch(rnd.randint(1, 40), rnd.choice(["prefill", "decode"]), rnd.choice([0, 0, 3]))
            b.positions = ref[0](b, dev)
            same(b.positions, sched_glue.make_positions(b, dev))
            for x, y in zip(ref[1](b, dev), sched_glue.make_input_tuple(b, dev)):
                same(x, y)
            for x, y in zip(ref[2](b, dev), sched_glue.make_write_tuple(b, dev)):
                same(x, y)
        # host time at a full decode batch
        b = make_batch(256, "decode", 0)
        def timed(fns):
            t0 = time.perf_counter()
            for _ in range(50):
                b.positions = fns[0](b, dev); fns[1](b, dev); fns[2](b, dev)
            return (time.perf_counter() - t0) / 50 * 1e6
        t_ref, t_mine = timed(ref), timed((sched_glue.make_positions, sched_glue.make_input_tuple, sched_glue.make_write_tuple))
        print(f"glue us per step: reference {{t_ref:.0f}} vectorised {{t_mine:.0f}}")
        assert t_mine < t_ref
        plugin.install(gemm_tune="off", vectorized_glue=True)
        assert sched._make_positions is sched_glue.make_positions and sched._make_write_tuple is sched_glue.make_write_tuple
        print("glue ok")
    """)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=str(ROOT / "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "glue ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    print(r.stdout.strip().splitlines()[-2])
