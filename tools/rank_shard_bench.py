"""One rank's shard of a tensor-parallel decode step on ONE GPU, collectives looped back (VERDICT r2 "next" 3c).

    python tools/rank_shard_bench.py --model qwen3-14b --tp 4 [--batch 256] [--steps 20] [--tp1-ms 18.2]
    python bench.py --rank-shard 4 ...                      (same thing)

What a TP = N run does on each GPU, minus the links: rank 0's weight shard (Hq/N query heads, max(Hkv/N, 1) KV heads,
inter/N MLP columns, vocab/N LM-head rows; P/models/weight.py:34-52), its KV pool shard, the same kernels, plans and
launch count as the real rank -- GEMM search on the shard shapes, captured decode graph -- with every collective run as
the SAME peer-to-peer kernel on a one-rank communicator (copy-in, flag barrier, rank-ordered sum, copy-out; the
all-gather's N - 1 remote shards are written locally).  What it bounds: TP = N cannot be faster than this step time
(the links only add), so `tp1_ms / ms_per_step` is an UPPER bound on the strong-scaling speed-up at N GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class LoopbackCommunicator:
    """all_reduce / all_gather of a tp_size-rank group executed by rank 0 alone through csrc/comm_p2p.hip."""

    def __init__(self, tp_size: int, max_bytes: int) -> None:
        import torch.distributed as dist

        from mini_sglang_amd.kernel import P2PCommunicator

        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(backend="gloo", rank=0, world_size=1)
        self.tp_size, self.rank, self.world_size = tp_size, 0, tp_size
        self.p2p = P2PCommunicator(0, 1, dist.group.WORLD, max_bytes)
        self.calls = 0

    def all_reduce(self, x: torch.Tensor, op: str = "sum") -> None:
        self.calls += 1
        self.p2p.all_reduce(x, op)

    def all_gather(self, output: torch.Tensor, input: torch.Tensor) -> None:
        self.calls += 1
        rows = input.shape[0]
        self.p2p.all_gather(output[:rows], input)
        output.view(self.tp_size, rows, *input.shape[1:])[1:] = input  # the peers' shards arriving

    def all_reduce_add_rmsnorm(self, x, residual, weight, eps) -> None:
        from mini_sglang_amd import ops

        self.calls += 1
        if not self.p2p.all_reduce_add_rmsnorm(x, residual, weight, eps):
            self.p2p.all_reduce(x, "sum")
            ops.fused_add_rmsnorm(x, residual, weight, eps)

    def poll_error(self, sync: bool = False) -> None:
        self.p2p.poll_error(sync)

    def destroy(self) -> None:
        self.p2p.destroy()


def run(model: str, tp: int, batch: int, steps: int, warmup: int, page_size: int, tp1_ms: float | None,
        device: torch.device) -> dict:
    from mini_sglang_amd import _lib
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.offline import OfflineRunner
    from bench import bench_contexts

    mcfg = PRESETS[model]
    B = batch
    max_seq = 4096
    p2p_bytes = max(B * mcfg.hidden_size * 2, B * (-(-mcfg.vocab_size // tp)) * 2)
    comm = LoopbackCommunicator(tp, p2p_bytes)
    ecfg = EngineConfig(model=mcfg, dtype=torch.bfloat16, tp_rank=0, tp_size=tp, max_running_req=B, cuda_graph_bs=[B],
                        page_size=page_size, max_seq_len_override=max_seq, comm=comm, comm_side=None,
                        num_page_override=B * (max_seq // 2) // page_size, comm_split_tokens=0,
                        gemm_tune=os.environ.get("MSGL_GEMM_TUNE", "full"))
    engine = Engine(ecfg, device)
    runner = OfflineRunner(engine, max_extend_tokens=16384, seed=0)
    contexts = bench_contexts(B)
    rnd = random.Random(1234)
    prompts = [[rnd.randint(0, 10000) for _ in range(n)] for n in contexts]
    sp = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=steps + warmup + 8) for _ in range(B)]
    states = [runner.add_request(p, s) for p, s in zip(prompts, sp)]
    # contexts are filled with random K/V (no prefill: the shard's prefill is not what this bounds)
    engine.kv_cache.pool.normal_(0.0, 1.0)
    for st in states:
        st.req.cached_len, st.req.device_len = st.prompt_len - 1, st.prompt_len
    runner._allocate_paged([type("R", (), dict(table_idx=s.req.table_idx, cached_len=0, device_len=s.prompt_len))()
                            for s in states])
    running = list(states)
    for _ in range(warmup):
        runner.decode_step(running)
    torch.cuda.synchronize(device)
    calls0 = comm.calls
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.decode_step(running)
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) * 1e3 / steps
    comm.poll_error(sync=True)
    S = sum(s.req.device_len for s in running)
    be = engine.attn_backend
    kv_tok = 2 * mcfg.num_layers * be.kv_heads * mcfg.head_dim * 2
    step_bytes = engine.model.streamed_bytes_per_step() + (S + B) * kv_tok + B * (-(-mcfg.vocab_size // tp)) * 2
    res = {
        "metric": "rank-shard decode step (one rank of a TP group on one GPU, collectives looped back)",
        "model": mcfg.name, "tp": tp, "batch": B, "ms_per_step": ms, "mean_context": S / B,
        "collectives_per_step": 2 * mcfg.num_layers + 2, "collective_launches_outside_graph_per_step": (comm.calls - calls0) / steps,
        "rank_bytes_per_step": step_bytes, "rank_hbm_frac": step_bytes / (ms * 1e-3) / 8e12,
        "tokens_per_s_bound_at_tp": B / (ms * 1e-3),
        "gemm_plans_at_full_batch": {r["name"]: dict(us=round(r["best_us"], 1), kernel=r["kernel"][:70])
                                     for r in engine.gemm_report if r["M"] == B},
    }
    if tp1_ms:
        res["tp1_ms_per_step"] = tp1_ms
        res["speedup_upper_bound_vs_tp1"] = tp1_ms / ms
    engine.shutdown()
    comm.destroy()
    return res


def main(argv=None) -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--tp", type=int, default=4)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--page-size", type=int, default=256)
    ap.add_argument("--tp1-ms", type=float, default=None, help="measured TP1 ms/step of the same workload (for the bound)")
    ap.add_argument("--out")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = run(args.model, args.tp, args.batch, args.steps, args.warmup, args.page_size, args.tp1_ms, dev)
    print(json.dumps(res), flush=True)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
