"""N > 1 path on CPU: 2 processes over gloo run the tensor-parallel restatement (sharded
weights, vocab-parallel embedding with masked gather, row-parallel all-reduce, LM-head
all-gather + unshard) and must reproduce the single-rank forward; plus the all-reduce /
all-gather known answers of the reference's tests/kernel/test_comm.py:96-149."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # the container hostname may not resolve
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mini_sglang_amd.model import PRESETS
        from oracle import ref_model

        torch.set_num_threads(2)
        # ---- known answers (test_comm.py:96-149) over the same collective seam
        def all_reduce(t):
            f = t.float()  # gloo has no bf16 sum on every build; sum in fp32 like a ring in fp32 would
            dist.all_reduce(f)
            return f.to(t.dtype)

        def all_gather(t):
            out = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(out, t.contiguous())
            return torch.cat(out, 0)

        x = torch.ones(4096, dtype=torch.bfloat16)
        for _ in range(4):
            x = all_reduce(x)
        assert torch.equal(x, torch.full((4096,), float(world ** 4), dtype=torch.bfloat16))
        r = all_reduce(torch.full((4096,), float(rank), dtype=torch.bfloat16))
        assert torch.equal(r, torch.full((4096,), world * (world - 1) / 2, dtype=torch.bfloat16))
        g = all_gather(torch.full((8,), float(rank), dtype=torch.bfloat16))
        assert torch.equal(g, torch.arange(world, dtype=torch.bfloat16).repeat_interleave(8))

        # ---- sharded forward == unsharded forward (prefill with a cache hit, then a decode step)
        cfg = PRESETS["tiny"]
        dt = torch.float32  # exact-ish comparison: fp32 everywhere, only the summation order differs
        full = ref_model.random_weights(cfg, dt, seed=7)
        mine = ref_model.shard_weights(cfg, full, world, rank)
        D, hkv, hkv_l = cfg.head_dim, cfg.num_kv_heads, max(cfg.num_kv_heads // world, 1)
        slots = 256
        gen = torch.Generator().manual_seed(1)
        table = torch.randperm(slots, generator=gen).to(torch.int32).view(4, 64)
        rows, k_lens, q_lens = [2, 0], [20, 33], [20, 13]
        ids = torch.randint(0, cfg.vocab_size, (33,), generator=gen)
        pos = torch.cat([torch.arange(0, 20), torch.arange(20, 33)]).to(torch.int32)
        loc = torch.cat([table[2, :20], table[0, 20:33]])
        kf = [torch.randn((slots, hkv, D), generator=gen) for _ in range(cfg.num_layers)]
        vf = [torch.randn((slots, hkv, D), generator=gen) for _ in range(cfg.num_layers)]
        ks = [k[:, rank * hkv_l:(rank + 1) * hkv_l].clone() for k in kf]
        vs = [v[:, rank * hkv_l:(rank + 1) * hkv_l].clone() for v in vf]
        ref = ref_model.forward(cfg, full, ids, pos, loc, kf, vf, table, rows, k_lens, q_lens, True)
        got = ref_model.forward_tp(cfg, mine, world, rank, all_reduce, all_gather, ids, pos, loc, ks, vs, table, rows,
                                   k_lens, q_lens, True)
        torch.testing.assert_close(got, ref, atol=2e-4, rtol=2e-4)
        for li in range(cfg.num_layers):  # each rank stored exactly its kv-head slice
            torch.testing.assert_close(ks[li], kf[li][:, rank * hkv_l:(rank + 1) * hkv_l], atol=1e-5, rtol=1e-5)
        # decode step on top
        ids2 = ref.argmax(-1)
        pos2 = torch.tensor([20, 33], dtype=torch.int32)
        loc2 = torch.stack([table[2, 20], table[0, 33]])
        ref2 = ref_model.forward(cfg, full, ids2, pos2, loc2, kf, vf, table, rows, [21, 34], [1, 1], False)
        got2 = ref_model.forward_tp(cfg, mine, world, rank, all_reduce, all_gather, ids2, pos2, loc2, ks, vs, table,
                                    rows, [21, 34], [1, 1], False)
        torch.testing.assert_close(got2, ref2, atol=2e-4, rtol=2e-4)
        assert torch.equal(got2.argmax(-1), ref2.argmax(-1))
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_tensor_parallel_forward_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
