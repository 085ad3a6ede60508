"""TEST INFRASTRUCTURE: one tensor-parallel rank (a process) of the product code.

    RANK=r WORLD=n PORT=p python tests/tp_worker.py <mode> <out.pt> [device index]

All ranks may share ONE device (the peer-to-peer communicator needs no RCCL, which refuses two ranks per device):
that is how N > 1 execution of the product is covered on a 1-GPU box.  Modes:
  collectives   the reference's known answers (tests/kernel/test_comm.py:96-149: ones -> n^rounds, rank-valued ->
                n(n-1)/2 with a lagging rank 0, half zeros / half ones, all-gather of rank-valued chunks) at one-shot and
                two-shot sizes, ragged counts, bit-identity across ranks, hipGraph capture + replay
  absent_rank   rank 0 enters collectives nobody else joins: NaN-poisoned outputs, sticky error word, polls raise
  tp_model      DenseDecoder(tp_size = n) through Engine + OfflineRunner on the tiny model: logits of every forward
                (rank 0 also runs the tp = 1 engine on the same full weights for the parent to compare), KV shards,
                token-split side-stream overlap on vs off
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def collectives(rank: int, world: int, dev: torch.device) -> dict:
    from mini_sglang_amd.kernel import P2PCommunicator

    res = {"world": world}
    comm = P2PCommunicator(rank, world, dist.group.WORLD, max_bytes=8 << 20, one_shot_max_bytes=256 << 10, blocks=16)
    for name, numel in (("one_shot", 4096), ("two_shot", 256 * 5120), ("ragged_two_shot", 8 * 12345 + 8)):
        # (1) ones -> n^4 after 4 rounds (exact in bf16 up to 8^4 = 4096)
        x = torch.ones(numel, dtype=torch.bfloat16, device=dev)
        for _ in range(4):
            comm.all_reduce(x)
        torch.cuda.synchronize()
        assert bool((x == float(world ** 4)).all()), (name, x[:4])
        # (2) rank-valued, rank 0 arrives late
        x = torch.full((numel,), float(rank), dtype=torch.bfloat16, device=dev)
        if rank == 0:
            time.sleep(0.3)
        comm.all_reduce(x)
        torch.cuda.synchronize()
        assert bool((x == float(world * (world - 1) // 2)).all()), (name, x[:4])
        # (3) first half zeros, second half ones
        x = torch.zeros(numel, dtype=torch.bfloat16, device=dev)
        x[numel // 2:] = 1
        comm.all_reduce(x)
        torch.cuda.synchronize()
        assert bool((x[: numel // 2] == 0).all()) and bool((x[numel // 2:] == world).all()), name
        # (4) random data: every rank must hold the SAME bits (rank-ordered sum), close to the fp32 sum
        g = torch.Generator(device=dev).manual_seed(1234)
        parts = [torch.randn(numel, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16) for _ in range(world)]
        x = parts[rank].clone()
        comm.all_reduce(x)
        torch.cuda.synchronize()
        want = torch.stack([p.float() for p in parts]).sum(0)
        assert (x.float() - want).abs().max().item() <= 2 ** -7 * want.abs().max().item(), name
        res[f"sum_{name}"] = x.cpu()
        # (5) fp16
        xh = torch.full((numel,), 0.5 + rank, dtype=torch.float16, device=dev)
        comm.all_reduce(xh)
        torch.cuda.synchronize()
        assert bool((xh == 0.5 * world + world * (world - 1) / 2).all()), name
    # fused all-reduce + residual add + RMSNorm == all-reduce, then fused_add_rmsnorm: same bits (x, residual), every rank
    from mini_sglang_amd import ops

    fused_checked = 0
    for rows, dim, dt in ((256, 5120, torch.bfloat16), (130, 5120, torch.bfloat16), (8, 1024, torch.float16), (1, 8192, torch.bfloat16),
                          (64, 128, torch.bfloat16)):
        g = torch.Generator(device=dev).manual_seed(99 + rows)
        parts = [torch.randn((rows, dim), generator=g, device=dev).to(dt) for _ in range(world)]
        res0 = torch.randn((rows, dim), generator=g, device=dev).to(dt)
        wn = (1 + 0.1 * torch.randn(dim, generator=g, device=dev)).to(dt)
        xa, ra = parts[rank].clone(), res0.clone()
        comm.all_reduce(xa)
        ops.fused_add_rmsnorm(xa, ra, wn, 1e-6)
        big = torch.zeros((rows, dim + 64), dtype=dt, device=dev)   # row-strided x as well
        xb, rb = big[:, :dim], res0.clone()
        xb.copy_(parts[rank])
        used = comm.all_reduce_add_rmsnorm(xb, rb, wn, 1e-6)
        torch.cuda.synchronize()
        if used:
            assert torch.equal(xa, xb) and torch.equal(ra, rb), (rows, dim, dt)
            assert big[:, dim:].abs().max().item() == 0
            fused_checked += 1
    res["fused_allreduce_norm_shapes"] = fused_checked
    # all-gather of rank-valued chunks
    src = torch.full((64, 1184), float(rank), dtype=torch.bfloat16, device=dev)
    dst = torch.empty((64 * world, 1184), dtype=torch.bfloat16, device=dev)
    comm.all_gather(dst, src)
    torch.cuda.synchronize()
    for r in range(world):
        assert bool((dst[64 * r: 64 * (r + 1)] == float(r)).all())
    # hipGraph: a captured all-reduce replays with fresh data (flags are device-side sequence numbers)
    buf = torch.zeros(256 * 1024, dtype=torch.bfloat16, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        comm.all_reduce(buf)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
            comm.all_reduce(buf)
        for it in range(3):
            buf.fill_(float(rank + it))
            graph.replay()
            torch.cuda.synchronize()
            assert bool((buf == float(world * it + world * (world - 1) // 2)).all()), it
    # timing (same device for all ranks here: a protocol check, not an xGMI number)
    x = torch.ones(256 * 5120, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        comm.all_reduce(x)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        comm.all_reduce(x)
    torch.cuda.synchronize()
    res["two_shot_2p6MB_us"] = (time.perf_counter() - t0) / 20 * 1e6
    res["error"] = comm.error()
    dist.barrier()
    comm.destroy()
    return res


def absent_rank(rank: int, world: int, dev: torch.device) -> dict:
    """A peer that never arrives: rank 0 runs a collective nobody else joins.  Its barrier
    must give up, the output must be NaN-poisoned (never a partial sum), the error word sticky, every poll form must
    raise, and later collectives must fail fast instead of spinning again."""
    from mini_sglang_amd._lib import MsglError
    from mini_sglang_amd.kernel import HybridCommunicator, P2PCommunicator

    res = {"world": world}
    comm = P2PCommunicator(rank, world, dist.group.WORLD, max_bytes=4 << 20, one_shot_max_bytes=256 << 10, blocks=8)
    comm.set_spin_limit(100_000)
    for numel in (4096, 256 * 1024):  # a healthy round first (one-shot, two-shot)
        x = torch.full((numel,), float(rank + 1), dtype=torch.bfloat16, device=dev)
        comm.all_reduce(x)
        torch.cuda.synchronize()
        assert bool((x == float(world * (world + 1) // 2)).all())
    comm.poll_error()            # enqueue a first async read: nothing wrong yet
    torch.cuda.synchronize()
    comm.poll_error()
    dist.barrier()
    if rank == 0:
        for name, numel in (("two_shot", 256 * 1024), ("one_shot", 4096)):
            x = torch.full((numel,), 1.0, dtype=torch.bfloat16, device=dev)
            t0 = time.perf_counter()
            comm.all_reduce(x)
            torch.cuda.synchronize()
            res[f"{name}_seconds"] = time.perf_counter() - t0
            res[f"{name}_all_nan"] = bool(x.isnan().all())
        dst = torch.zeros((world * 64, 128), dtype=torch.bfloat16, device=dev)
        comm.all_gather(dst, torch.ones((64, 128), dtype=torch.bfloat16, device=dev))
        torch.cuda.synchronize()
        res["gather_all_nan"] = bool(dst.isnan().all())
        res["error_word"] = comm.error()
        raised = {}
        for form in ("sync", "async"):
            try:
                if form == "sync":
                    comm.poll_error(sync=True)
                else:
                    comm._err_event = None
                    comm.poll_error()        # enqueues the copy
                    torch.cuda.synchronize()
                    comm.poll_error()        # sees it
                raised[form] = None
            except MsglError as e:
                raised[form] = str(e)
        res["raised"] = raised
        try:
            HybridCommunicator(comm, None).destroy()   # destroying a communicator with a pending error raises too
            res["destroy_raised"] = False
        except MsglError:
            res["destroy_raised"] = True
    dist.barrier()
    if rank != 0:
        # the rank that stayed away: rank 0's verdict arrived in its header, so a collective it enters NOW (alone: rank 0 is
        # done) poisons at once instead of spinning, and its host's poll raises
        res["error_word"] = comm.error()
        x = torch.full((256 * 1024,), 1.0, dtype=torch.bfloat16, device=dev)
        t0 = time.perf_counter()
        comm.all_reduce(x)
        torch.cuda.synchronize()
        res["late_seconds"] = time.perf_counter() - t0
        res["late_all_nan"] = bool(x.isnan().all())
        try:
            comm.poll_error(sync=True)
            res["late_raised"] = None
        except MsglError as e:
            res["late_raised"] = str(e)
        comm.destroy()
    return res


def tp_model(rank: int, world: int, dev: torch.device) -> dict:
    import refdrive
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.kernel import init_pynccl
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.offline import OfflineRunner

    state = refdrive.seeded_hf_state("tiny", seed=7)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, 1000, (n,), generator=g).tolist() for n in (5, 40, 130, 17, 64, 9)]
    sp = [SamplingParams(temperature=0.0, max_tokens=5, ignore_eos=True) for _ in prompts]
    max_bytes = 256 * 256 * 2 * 4

    def engine(tp_size, tp_rank, comm, comm_side, split, overlap):
        cfg = EngineConfig(model=PRESETS["tiny"], dtype=torch.bfloat16, tp_rank=tp_rank, tp_size=tp_size, max_running_req=8,
                           page_size=4, cuda_graph_bs=[1, 2, 4, 8], max_seq_len_override=512, num_page_override=512,
                           comm=comm, comm_side=comm_side, comm_split_tokens=split, comm_overlap=overlap,
                           tp_cpu_group=dist.group.WORLD if tp_size > 1 else None, gemm_tune="off")
        eng = Engine(cfg, dev)
        eng.model.load_hf_state(state)
        eng.kv_cache.pool.zero_()
        return eng

    def run(tp_size, tp_rank, comm, comm_side, split, overlap):
        from replay_util import record_offline_runner

        eng = engine(tp_size, tp_rank, comm, comm_side, split, overlap)
        runner = OfflineRunner(eng, max_extend_tokens=96, seed=0)
        forwards = []
        record_offline_runner(runner, eng, forwards)
        runner.generate(prompts, sp)
        ids = [runner.output_ids(s) for s in runner.last_states]
        kv = eng.kv_cache.pool.cpu()
        eng.shutdown()
        return dict(ids=ids, forwards=forwards, logits=[f["logits"] for f in forwards], kv=kv)

    def replay_tp1(forwards):
        """The tp = 1 engine, teacher-forced with the batches the tp = 2 run executed."""
        from replay_util import replay_forward

        eng = engine(1, 0, None, None, 0, True)
        logits = [replay_forward(eng, f).float().cpu() for f in forwards]
        kv = eng.kv_cache.pool.cpu()
        eng.shutdown()
        return dict(logits=logits, kv=kv)

    out = {}
    comm = init_pynccl(tp_rank=rank, tp_size=world, tp_cpu_group=dist.group.WORLD, max_size_bytes=max_bytes, backend="p2p")
    side = init_pynccl(tp_rank=rank, tp_size=world, tp_cpu_group=dist.group.WORLD, max_size_bytes=max_bytes, backend="p2p")
    out["tp"] = run(world, rank, comm, side, 0, True)                 # collectives on the compute stream
    out["tp_split_serial"] = run(world, rank, comm, side, 32, False)  # token halves, one stream
    out["tp_split_overlap"] = run(world, rank, comm, side, 32, True)  # first half's all-reduce on the side stream
    out["comm_error"] = (comm.p2p.error(), side.p2p.error())
    dist.barrier()
    comm.destroy()
    side.destroy()
    if rank == 0:
        out["tp1"] = replay_tp1(out["tp"]["forwards"])
    return out


def root_cause(rank: int, world: int, dev: torch.device) -> dict:
    """World 3: rank 2 stays away; ranks 0 and 1 enter the same all-reduce, rank 0 with a short spin limit, rank 1
    with a long one.  Rank 0 gives up first (waiting for rank 2) and tells its peers; rank 1 leaves its spin BECAUSE of that
    word and must NOT tell anybody: rank 0's header keeps the root cause (no bit 20, the peer it waited for), rank 1's and
    rank 2's headers hold rank 0's told copy (bit 20, rank field 0)."""
    from mini_sglang_amd.kernel import P2PCommunicator

    assert world == 3
    comm = P2PCommunicator(rank, world, dist.group.WORLD, max_bytes=4 << 20, one_shot_max_bytes=256 << 10, blocks=8)
    comm.set_spin_limit(100_000 if rank == 0 else 400_000_000)
    x = torch.full((256 * 1024,), float(rank + 1), dtype=torch.bfloat16, device=dev)
    comm.all_reduce(x)  # healthy round
    torch.cuda.synchronize()
    assert bool((x == 6.0).all())
    dist.barrier()
    res = {"world": world}
    if rank != 2:
        x = torch.ones(256 * 1024, dtype=torch.bfloat16, device=dev)
        t0 = time.perf_counter()
        comm.all_reduce(x)
        torch.cuda.synchronize()
        res["seconds"] = time.perf_counter() - t0
        res["all_nan"] = bool(x.isnan().all())
    dist.barrier()
    time.sleep(0.2)
    res["error_word"] = comm.error()
    dist.barrier()
    return res


def main() -> None:
    mode, out_path = sys.argv[1], sys.argv[2]
    dev = torch.device(f"cuda:{sys.argv[3] if len(sys.argv) > 3 else 0}")
    rank, world, port = int(os.environ["RANK"]), int(os.environ["WORLD"]), int(os.environ["PORT"])
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    res = {"collectives": collectives, "tp_model": tp_model, "absent_rank": absent_rank, "root_cause": root_cause}[mode](rank, world, dev)
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
