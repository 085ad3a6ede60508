"""N > 1 host logic on CPU (2 gloo ranks).  The product's tensor-parallel FORWARD needs the GPU and is checked there
with two real ranks (tests/test_gpu_tp.py: DenseDecoder(tp_size=2) == tp = 1, peer-to-peer collectives).  Here:
* the product's sharding rules -- DenseDecoder.load_hf_state (q/k/v and gate/up merging, column / row shards, vocab
  shards), model.vocab_shard, model.lm_head_unshard -- are run on each rank and must equal the oracle's shard of the
  same full weights, and the LM-head unshard of a gloo all-gather must rebuild the unsharded logits;
* the oracle's tensor-parallel restatement reproduces its single-rank forward over the same collectives (keeps the
  oracle honest), with the all-reduce / all-gather known answers of the reference's tests/kernel/test_comm.py:96-149."""
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

        # ---- bench.py's multi-GPU preflight runs the SAME known answers (in place, its collective seam) before a --gpus N run:
        #      exercised here over gloo so that the code a first 8-GPU run depends on has run at world 2 somewhere
        import sys
        from pathlib import Path

        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        import bench

        def ar_inplace(t):
            t.copy_(all_reduce(t))

        def ag_into(dst, src):
            dst.copy_(all_gather(src))

        assert bench._known_answers(ar_inplace, ag_into, rank, world, torch.device("cpu"), 4096) == []
        wrong = bench._known_answers(lambda t: None, ag_into, rank, world, torch.device("cpu"), 4096)  # a collective that does nothing
        assert len(wrong) >= 2 and "ones x4" in wrong[0]

        # ---- the PRODUCT's sharding (DenseDecoder.load_hf_state on this rank) == the oracle's shard of the same weights

        sys.path.insert(0, str(Path(__file__).resolve().parent))
        import refdrive
        from mini_sglang_amd.model import DenseDecoder, lm_head_unshard, vocab_shard

        cfg = PRESETS["tiny"]
        state = refdrive.seeded_hf_state("tiny", seed=11)
        dec = DenseDecoder(cfg, dtype=torch.bfloat16, device=torch.device("cpu"), tp_rank=rank, tp_size=world)
        dec.load_hf_state(state)
        layers = []
        for i in range(cfg.num_layers):
            p = f"model.layers.{i}."
            layers.append(dict(
                input_norm=state[p + "input_layernorm.weight"],
                qkv=torch.cat([state[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0),
                q_norm=state[p + "self_attn.q_norm.weight"], k_norm=state[p + "self_attn.k_norm.weight"],
                o=state[p + "self_attn.o_proj.weight"], post_norm=state[p + "post_attention_layernorm.weight"],
                gate_up=torch.cat([state[p + "mlp.gate_proj.weight"], state[p + "mlp.up_proj.weight"]], 0),
                down=state[p + "mlp.down_proj.weight"]))
        full_w = ref_model.CpuWeights(state["model.embed_tokens.weight"], layers, state["model.norm.weight"],
                                      state["lm_head.weight"], dec.cos_sin)
        want = ref_model.shard_weights(cfg, full_w, world, rank)
        for li, lw in enumerate(dec.layers):
            for name in ("qkv", "o", "gate_up", "down", "input_norm", "post_norm", "q_norm", "k_norm"):
                assert torch.equal(getattr(lw, name), want.layers[li][name]), (li, name)
        per, (start, length) = vocab_shard(cfg.vocab_size, world, rank)
        assert torch.equal(dec.embed[:length], want.embed[:length]) and torch.equal(dec.lm_head[:length], want.lm_head[:length])
        # LM head: shard logits -> all-gather -> product's unshard == unsharded logits
        h = torch.randn((5, cfg.hidden_size), generator=torch.Generator().manual_seed(5))
        mine_logits = h @ dec.lm_head.float().t()
        rebuilt = lm_head_unshard(all_gather(mine_logits), world, 5, cfg.vocab_size)
        torch.testing.assert_close(rebuilt, h @ state["lm_head.weight"].float().t(), atol=1e-5, rtol=1e-5)

        # ---- the oracle's sharded forward == its unsharded forward (prefill with a cache hit, then a decode step)
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
