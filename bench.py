#!/usr/bin/env python3
"""Headline benchmark: decode tokens/s of the MI355X paged-attention serving core.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: spawns its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W        (N > 1: TP = N, one rank per GPU)

Workload (BASELINE.json: metric quoted on Qwen3-14B TP=1; fits one GPU => configs[2]'s model on
configs[1]'s request shape): Qwen3-14B bf16, 256 sequences, context lengths drawn from the
token-weighted context distribution of the reference's offline benchmark
(benchmark/offline/bench.py:11-31: in/out 100..1024, mean context per decoded token 902.8),
page_size 256, temperature 0.6, seeded N(0, 0.02^2) weights (no checkpoints offline), synthetic
token ids.  The prompts are first prefetched with the real chunked-prefill path
(max_extend_tokens 16384) -- that gives p50 TTFT -- then ONE STEP = one decode forward of the
whole batch: metadata prep, hipGraph replay of the 40-layer model (hand-written gfx950 kernels +
hipBLASLt GEMMs), sampling, token feedback.  Nothing is skipped inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (paged decode attention:
~58% of the step's algorithmic bytes), timed live with HIP events on the launch stream;
`cpu_baseline` is the oracle's torch-eager forward on this node's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
METRIC = "decode tokens/s/GPU + p50 TTFT, Qwen3-14B TP=1 and Qwen3-32B TP=4"


def product_code_fingerprint() -> str:
    """sha256 over the product's sources (kernels, host C++, package Python, the C header): a number recorded under profiles/
    by another process (the reference-driven step time) carries the fingerprint of the code that produced it, and this file
    marks it stale when HEAD's sources differ."""
    import hashlib

    h = hashlib.sha256()
    pkg = ROOT / "mini-sglang_amd"
    files = sorted(list((pkg / "csrc").glob("*")) + list(pkg.glob("*.py")) + list((ROOT / "include").glob("*.h")))
    for f in files:
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def bench_contexts(num_seqs: int, seed: int = 0):
    """(request, decode step) pairs drawn uniformly from all decoded tokens of the reference's
    offline benchmark => context lengths with its token-weighted mean (902.8 at 256 seqs)."""
    rnd = random.Random(seed)
    ins = [rnd.randint(100, 1024) for _ in range(256)]
    outs = [rnd.randint(100, 1024) for _ in range(256)]
    pick = random.Random(seed + 1)
    reqs = pick.choices(range(256), weights=outs, k=num_seqs)
    return [ins[r] + pick.randint(0, outs[r] - 1) for r in reqs]


def cpu_baseline(cfg, contexts, rounds: int = 3):
    """Oracle (oracle/ref_model.py, torch eager fp32) timed on the host cores ON THE GPU WORKLOAD'S SHAPE: ONE decoder
    layer of the same dims at the full batch (256 requests, the bench's own contexts, K/V gathered through a page
    table) + embedding, final norm and LM head, x num_layers.  A reported baseline, not a target."""
    from oracle import ref_model

    avail = len(os.sched_getaffinity(0))
    forced = os.environ.get("MSGL_CPU_BASELINE_THREADS")
    B, L = len(contexts), 1
    lens = [int(n) for n in contexts]
    D = cfg.head_dim
    g = torch.Generator().manual_seed(0)
    small_vocab = 4096  # the embedding table is only gathered from: a small one keeps the host allocation bounded
    w = ref_model.random_weights(
        type("C", (), dict(cfg.__dict__, vocab_size=small_vocab, tie_word_embeddings=True))(), torch.float32, num_layers=L)
    w.lm_head = torch.randn((cfg.vocab_size, cfg.hidden_size), generator=g) * 0.02
    # bf16 weights, fp32 accumulation: every weight rounded to bf16, held and multiplied as fp32
    # (torch's CPU bf16 matmul is several times slower than its fp32 one on these hosts and would flatter the GPU)
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    w.embed, w.final_norm, w.lm_head = bf(w.embed), bf(w.final_norm), bf(w.lm_head)
    for lw in w.layers:
        for k_, v_ in list(lw.items()):
            if torch.is_tensor(v_) and v_.is_floating_point():
                lw[k_] = bf(v_)
    row = max(lens) + rounds + 2
    table = torch.arange(B * row, dtype=torch.int32).view(B, row)
    kp = [torch.randn((B * row, cfg.num_kv_heads, D), generator=g) for _ in range(L)]
    vp = [torch.randn((B * row, cfg.num_kv_heads, D), generator=g) for _ in range(L)]
    ids = torch.randint(0, small_vocab, (B,), generator=g)
    no_head = ref_model.CpuWeights(w.embed, w.layers, w.final_norm, w.lm_head[:8], w.cos_sin)

    def run(weights, cur):
        pos = torch.tensor([n - 1 for n in cur], dtype=torch.int32)
        loc = table[torch.arange(B), pos.long()]
        t0 = time.perf_counter()
        ref_model.forward(cfg, weights, ids, pos, loc, kp, vp, table, list(range(B)), cur, [1] * B, False)
        return time.perf_counter() - t0

    # thread count: torch-eager on a many-core host is not monotone in threads (the per-request attention loop and the
    # M = 256 GEMMs want different counts; 64 threads measured 3.5x SLOWER per layer than 8 on one box): one layer at each
    # of a few counts, the fastest is the baseline's (and is what `cores` reports)
    # (all 256 cores of a GPU box took 287 s for ONE layer in round 6 -- oversubscribed per-request attention loop: never probed)
    counts = [int(forced)] if forced else sorted({c for c in (8, 16, 32, 64) if c <= avail} or {avail})
    torch.set_num_threads(counts[0])
    run(no_head, lens)  # warm-up (thread pool, allocator)
    probe = {}
    for c in counts:
        torch.set_num_threads(c)
        probe[c] = run(no_head, lens)
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    t_layer, t_full = [], []
    for i in range(rounds):
        cur = [n + i + 1 for n in lens]
        t_layer.append(run(no_head, cur))  # embedding + L layers + final norm + an 8-row LM head
        t_full.append(run(w, cur))         # the same with the full LM head
    t_layer.sort(); t_full.sort()
    lay = t_layer[len(t_layer) // 2]
    head = t_full[len(t_full) // 2] - lay
    if head <= 0.05 * lay:
        # the difference of two noisy ~1-s runs can come out at (or below) zero: time the LM head's matmul on its own
        xs = torch.randn((B, cfg.hidden_size), generator=g)
        ts = []
        for _ in range(rounds):
            t0 = time.perf_counter()
            (xs @ w.lm_head.t()).sum().item()
            ts.append(time.perf_counter() - t0)
        head = sorted(ts)[len(ts) // 2]
    est_step = lay * cfg.num_layers / L + head
    return dict(value=B / est_step, unit="tokens/s", cores=cores, threads=cores, host_cores=avail,
                kind="port",
                sample=f"torch-eager oracle (bf16-rounded weights, fp32 math), {cfg.name} dims, the GPU step's own batch: {B} requests, contexts mean "
                       f"{sum(lens) / B:.0f} through a page table; {L} of {cfg.num_layers} decoder layers timed "
                       f"({lay * 1e3:.0f} ms, median of {rounds}) x {cfg.num_layers // L} + embedding / final norm / LM head "
                       f"({head * 1e3:.0f} ms): {est_step * 1e3:.0f} ms/step; host has {avail} cores (sched_getaffinity), `cores` = the thread count used = the fastest of those tried (s per layer): "
                       + ", ".join(f"{c}: {t:.2f}" for c, t in probe.items()))


def pmc_traffic(attn_bytes: int, shape: str):
    """HBM bytes per launch of the decode attention kernel (+merge) from the committed rocprofv3 PMC passes
    (FETCH_SIZE x the gfx950 correction of the same pass + WRITE_SIZE, separate passes; profiles/*_pmc_attn_decode.json).
    Counters cannot be read from inside this process: the newest profile of the SAME launch shape (batch, heads, page
    size, context distribution) gives the measured traffic / algorithmic ratio, applied to this run's algorithmic
    bytes (the contexts grow by one token per step, so the byte count differs by a fraction of a percent between
    --steps settings).  null if no profile of this shape is committed."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_pmc_attn_decode.json")):
        try:
            d = json.loads(f.read_text())
        except Exception:
            continue
        alg = float(d.get("algorithmic_bytes_per_launch", 0))
        same_shape = d.get("launch_shape") == shape or (
            "launch_shape" not in d and "Qwen3-14B TP1 shape, B=256" in d.get("workload", "") and shape == DEFAULT_SHAPE)
        if alg > 0 and same_shape and abs(alg - attn_bytes) / attn_bytes < 0.05:
            ratio = float(d["hbm_bytes_per_launch"]) / alg
            best = (ratio * attn_bytes, f"profiles/{f.name}", ratio)
    return best if best else (None, None, None)


DEFAULT_SHAPE = "qwen3-14b tp1 B256 page256 bench_contexts"


def gemm_traffic_over_algorithmic():
    """HBM bytes the projection path of one layer moves (FETCH_SIZE x 2 + WRITE_SIZE of the GEMM kernels, the slab-consuming
    norm / qk pass and SiLU.mul; committed rocprofv3 PMC passes of tools/pmc_gemm.py, k-sliced plans) over its algorithmic
    bytes (weights + activations in and out).  (ratio, source) or (None, None)."""
    try:
        fp = sorted((ROOT / "profiles").glob("r*_pmc_gemm_FETCH_SIZE.json"))[-1]  # the newest round's passes
        wp = fp.with_name(fp.name.replace("FETCH_SIZE", "WRITE_SIZE"))
        f, w = json.loads(fp.read_text()), json.loads(wp.read_text())
    except Exception:
        return None, None
    man = f["manifest"]
    if "path_ops" in man:  # round 5 on: the manifest names the launches of the step's path and their algorithmic bytes
        ops_, algo = man["path_ops"], man["path_algorithmic_bytes"]
    else:
        ops_ = [k for k in f["per_launch"] if not k.startswith("gate_up g3") and not k.startswith("tail reduce")]
        algo = sum(man["algorithmic_bytes"].values()) + 2 * man["norm_algorithmic_bytes"] + man["silu_algorithmic_bytes"]
    moved = sum(2.0 * f["per_launch"][k]["FETCH_SIZE"] + w["per_launch"][k]["WRITE_SIZE"] for k in ops_) * 1024.0
    return moved / algo, f"profiles/{fp.name} + {wp.name}"


def measure_prefill_attention(device, hq: int, hkv: int, budget: int = 16384):
    """MFMA roofline of the prefill attention kernel on ONE chunk of the bench's own prompts (the first
    `budget` prompt tokens, no cache hit), one layer: causal flops / HIP-event time."""
    from tools.microbench import MFMA_PEAK_TFLOPS, prefill_case, prefill_chunk_lens, time_us
    from mini_sglang_amd import ops

    lens = prefill_chunk_lens(budget, bench_contexts(256))
    c = prefill_case(lens, lens, hq, hkv, 256, device, q_tile=ops.prefill_q_tile())
    us = time_us(lambda: ops.attn_prefill(c["out"], c["q"], c["k"], c["v"], c["table"], None, c["seq"], c["cu_q"],
                                          c["tile_cu"], c["B"], c["total_tiles"], 128 ** -0.5,
                                          tile_order=c["order"]), iters=20, warmup=10)
    tf = c["flops"] / us / 1e6
    return {"bound": "mfma", "kernel": "attn_prefill_dma_kernel (the default prefill kernel), one layer, one chunk", "achieved": tf,
            "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "us_per_launch": us,
            "flops": c["flops"], "chunk_tokens": c["T"], "requests": c["B"]}


def self_launch(gpus: int) -> int:
    """`python3 bench.py --gpus N` with no launcher around it: start the N ranks through torch.distributed.run (one
    process per GPU, loopback rendezvous on a free port) and hand back its exit code; rank 0's JSON line goes to this
    process's stdout."""
    import subprocess

    cmd, env = self_launch_command(gpus, sys.argv[1:])
    return subprocess.call(cmd, env=env)


def self_launch_command(gpus: int, argv):
    """(command, environment) of the self-launch: the driver's own contract line for N > 1 with this file's arguments."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MSGL_BENCH_SELF_LAUNCHED="1")
    env.setdefault("OMP_NUM_THREADS", "8")
    return cmd, env


LAST_PREFLIGHT: dict = {}  # what check_collectives last measured (for the line a failed multi-GPU run still prints)


def _known_answers(f_all_reduce, f_all_gather, rank: int, world: int, device, numel: int) -> list:
    """The reference's own known answers (tests/kernel/test_comm.py:96-149) through one path at one message size: ones
    reduced four times -> world^4; rank-valued with rank 0 arriving 0.2 s late -> n(n-1)/2; half zeros / half ones ->
    [0, world]; all-gather of rank-valued chunks -> 0,0,..,1,1,..  Returns the list of failed checks (empty = pass)."""
    bad = []
    sync = (lambda: torch.cuda.synchronize(device)) if torch.device(device).type == "cuda" else (lambda: None)
    x = torch.ones(numel, dtype=torch.bfloat16, device=device)
    for _ in range(4):
        f_all_reduce(x)
    sync()
    if not bool((x == float(world ** 4)).all()):
        bad.append(f"ones x4 -> {x[0].item()} .. {x[-1].item()}, want {world ** 4}")
    x = torch.full((numel,), float(rank), dtype=torch.bfloat16, device=device)
    if rank == 0:  # a lagging rank: the others must wait for it, not sum stale data
        sync()
        time.sleep(0.2)
    f_all_reduce(x)
    sync()
    if not bool((x == float(world * (world - 1) // 2)).all()):
        bad.append(f"rank-valued with a lagging rank -> {x[0].item()}, want {world * (world - 1) // 2}")
    x = torch.cat([torch.zeros(numel // 2, dtype=torch.bfloat16, device=device), torch.ones(numel - numel // 2, dtype=torch.bfloat16, device=device)])
    f_all_reduce(x)
    sync()
    if not (bool((x[: numel // 2] == 0).all()) and bool((x[numel // 2:] == float(world)).all())):
        bad.append("half zeros / half ones: wrong halves")
    n = max(8, (numel // world) // 8 * 8)
    src = torch.full((n,), float(rank), dtype=torch.bfloat16, device=device)
    dst = torch.empty((n * world,), dtype=torch.bfloat16, device=device)
    f_all_gather(dst, src)
    sync()
    if not torch.equal(dst, torch.arange(world, dtype=torch.bfloat16, device=device).repeat_interleave(n)):
        bad.append("all-gather: wrong contents")
    return bad


def check_collectives(comm, rank: int, world: int, device) -> dict:
    """PREFLIGHT on the links the run is about to use, printed to stderr by rank 0 BEFORE the workload
    starts and carried in the final line, so that a multi-GPU run that dies later still says how far the links got:
      * `ncclCommCount` / the device every RCCL rank bound (`rccl_ranks_seen`, `rccl_devices`);
      * the reference's known answers (tests/kernel/test_comm.py:96-149) through RCCL and through the peer-to-peer kernels,
        each at the three message sizes of the path -- 256 KB (one-shot range), 2.6 MB ([256, hidden] decode all-reduce),
        84 MB ([8192, hidden] prefill chunk; RCCL only unless the mapped buffers were sized for it);
      * all-reduce time and bus bandwidth per path and size: busbw = 2 (n - 1) / n x bytes / t, to be read
        against one xGMI link (153 GB/s, ring) and all seven (peer-to-peer kernels)."""
    import torch.distributed as dist

    sizes = {"256KB": 256 * 1024, "2.6MB": 256 * 5120 * 2, "84MB": 8192 * 5120 * 2}
    paths = {}
    if comm.rccl is not None:
        paths["rccl"] = (comm.rccl.all_reduce, comm.rccl.all_gather, lambda nbytes: True)
    if comm.p2p is not None:
        paths["p2p"] = (comm.p2p.all_reduce, comm.p2p.all_gather, lambda nbytes: nbytes <= comm.p2p.max_bytes)
    out = {"p2p": "ok (one-shot / two-shot / all-gather self-test at init)" if comm.p2p is not None else "absent: RCCL carries every message",
           "known_answers": {}, "all_reduce": {}}
    for pname, (ar, ag, fits) in paths.items():
        for sname, nbytes in sizes.items():
            key = f"{pname} {sname}"
            if not fits(nbytes):
                out["known_answers"][key] = "skipped: larger than the mapped peer buffers (this size goes through RCCL)"
                continue
            try:
                bad = _known_answers(ar, ag, rank, world, device, nbytes // 2)
                x = torch.ones(nbytes // 2, dtype=torch.bfloat16, device=device)
                for _ in range(3):
                    ar(x)
                    x.fill_(1.0)
                torch.cuda.synchronize(device)
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 10
                e0.record()
                for _ in range(reps):
                    ar(x)
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
            except Exception as e:  # noqa: BLE001 -- the report must survive a broken path
                bad, us = [f"{type(e).__name__}: {e}"], None
            everyone = [None] * world
            dist.all_gather_object(everyone, (bad, us))
            fails = [f"rank {r}: {'; '.join(b)}" for r, (b, _) in enumerate(everyone) if b]
            out["known_answers"][key] = "pass" if not fails else "FAIL " + " | ".join(fails)
            times = [u for _, u in everyone if u is not None]
            if times:
                t = max(times)
                out["all_reduce"][key] = {"us": t, "busbw_GBps": 2.0 * (world - 1) / world * nbytes / t / 1e3}
    if comm.rccl is not None:
        info = comm.rccl.info()
        devs = [None] * world
        dist.all_gather_object(devs, (info["nranks"], info["device"]))
        out.update(rccl_ranks_seen=min(d[0] for d in devs), rccl_devices=[d[1] for d in devs])
    else:
        out.update(rccl_ranks_seen=0, rccl_devices=[])
    out["rccl_known_answer_ok"] = (all(v == "pass" for k, v in out["known_answers"].items() if k.startswith("rccl"))
                                   if comm.rccl is not None else None)
    out["p2p_known_answer_ok"] = (all(v == "pass" or v.startswith("skipped") for k, v in out["known_answers"].items() if k.startswith("p2p"))
                                  if comm.p2p is not None else None)
    LAST_PREFLIGHT.clear()
    LAST_PREFLIGHT.update(out)
    if rank == 0:
        print("[bench preflight] " + json.dumps(out), file=sys.stderr, flush=True)
    return out


def run_workload(args, model_name: str, rank: int, local_rank: int, world: int, device, share_gpu: bool,
                 primary: bool) -> dict:
    """One decode workload (chunked prefill, W + K decode steps, roofline of the attention launch) of `model_name` at
    TP = world.  primary: the headline entry (with the latency regime, the prefill-attention roofline, the GEMM search
    report); a secondary entry (Qwen3-32B at --gpus 4) carries the step time, TTFT and the collectives report only."""
    from mini_sglang_amd import ops
    from mini_sglang_amd.core import SamplingParams
    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.model import PRESETS
    from mini_sglang_amd.offline import OfflineRunner

    import torch.distributed as dist

    mcfg = PRESETS[model_name]
    comm = comm_side = None
    collectives = None
    if world > 1:
        from mini_sglang_amd.kernel import init_pynccl

        # decode-size messages (all-reduce [B, hidden], logits all-gather [B, V / tp]) go peer to peer over mapped
        # buffers, prefill-size ones through RCCL; a second communicator serves the side stream
        p2p_bytes = max(args.batch * mcfg.hidden_size * 2, args.batch * (-(-mcfg.vocab_size // world)) * 2)
        if share_gpu:
            # ranks time-slicing ONE GPU: a rank's barrier may spin through whole scheduling quanta of its peer (one 14B N = 2
            # run in five gave up at its first collective with the default limit); a code-path check wants patience, a real
            # node (one GPU per rank) keeps the default
            os.environ.setdefault("MSGL_P2P_SPIN_LIMIT", "2000000000")
        if share_gpu:  # every message must fit the mapped buffers: size them for a prefill chunk as well
            p2p_bytes = max(p2p_bytes, 16384 * mcfg.hidden_size * 2)
        backend = "p2p" if share_gpu else os.environ.get("MSGL_COMM_BACKEND", "hybrid")
        comm = init_pynccl(tp_rank=rank, tp_size=world, tp_cpu_group=dist.group.WORLD, max_size_bytes=p2p_bytes, backend=backend)
        comm_side = init_pynccl(tp_rank=rank, tp_size=world, tp_cpu_group=dist.group.WORLD, max_size_bytes=p2p_bytes,
                                backend=backend)
        collectives = check_collectives(comm, rank, world, device)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()

    B = args.batch
    # the latency-regime section is a single-GPU extra: under TP every additional graph is another capture
    # with collectives inside, which the headline measurement does not need
    small_batches = list(args.small_batches) if (world == 1 and primary) else []
    contexts = bench_contexts(B)
    use_graph = os.environ.get("MSGL_BENCH_NO_GRAPH", "0") != "1"
    max_seq = 4096  # max_seq_len_override of the reference bench
    ecfg = EngineConfig(model=mcfg, dtype=torch.bfloat16, tp_rank=rank, tp_size=world, max_running_req=B,
                        cuda_graph_bs=sorted(set([b for b in small_batches if b < B] + [B])) if use_graph else [],
                        page_size=args.page_size,
                        max_seq_len_override=max_seq, comm=comm, comm_side=comm_side, memory_ratio=0.9,
                        # ranks sharing one GPU cannot each take 90 % of its memory: a fixed pool that holds the batch
                        num_page_override=(B * (max_seq // 2) // args.page_size) if share_gpu else None,
                        comm_split_tokens=int(os.environ.get("MSGL_BENCH_COMM_SPLIT", "2048")) if world > 1 else 0, tp_cpu_group=dist.group.WORLD if world > 1 else None,
                        gemm_tune=os.environ.get("MSGL_GEMM_TUNE", "full"))
    engine, err = None, None
    try:
        engine = Engine(ecfg, device)
    except Exception as e:  # graph capture with a collective inside may be refused: run eager
        if not use_graph:
            raise
        err = f"{type(e).__name__}: {e}"
    if world > 1:  # every rank must take the same path: one rank falling back to eager alone would desynchronise the collectives
        flags = [None] * world
        dist.all_gather_object(flags, err)
        err = next((f for f in flags if f), None)
        if err is not None and engine is not None:
            engine.shutdown()
            engine = None
    if engine is None:
        # outside the except block: the failed engine (weights + 0.9 of HBM as KV pool) is only
        # referenced by the dead traceback; drop it before sizing a second pool
        import gc

        print(f"[bench] graph capture failed ({err}); falling back to eager", file=sys.stderr)
        gc.collect()
        torch.cuda.empty_cache()
        ecfg.cuda_graph_bs = []
        use_graph = False
        engine = Engine(ecfg, device)
    runner = OfflineRunner(engine, max_extend_tokens=16384, seed=0)

    rnd = random.Random(1234)
    prompts = [[rnd.randint(0, 10000) for _ in range(n)] for n in contexts]
    total_steps = args.steps + args.warmup
    sp = [SamplingParams(temperature=0.6, ignore_eos=True, max_tokens=total_steps + 8 + 13 * len(small_batches)) for _ in range(B)]
    states = [runner.add_request(p, s) for p, s in zip(prompts, sp)]

    # ---------------- prefill (untimed for `value`; yields TTFT) ----------------
    ttft_p50 = None
    if not args.no_prefill:
        # one untimed chunk of dummy requests first (library GEMM kernels for the chunk shape, allocator
        # pools), as the reference bench warms up with one generate() before timing (bench.py:32)
        runner.warmup_prefill()
    barrier()
    if args.no_prefill:
        engine.kv_cache.pool.normal_(0.0, 1.0)
        for st in states:
            st.req.cached_len, st.req.device_len = st.prompt_len - 1, st.prompt_len
        runner._allocate_paged([type("R", (), dict(table_idx=s.req.table_idx, cached_len=0,
                                                   device_len=s.prompt_len))() for s in states])
    else:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        evs = []
        for out, finals in runner.prefill(states):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            evs.append((ev, len(finals)))
        torch.cuda.synchronize(device)
        ttft = sorted(t for ev, n in evs for t in [ev0.elapsed_time(ev)] * n)
        ttft_p50 = ttft[int(len(ttft) * 0.5)]  # percentile rule of P/benchmark/client.py:324-348
    running = list(states)

    # ---------------- decode: W warm-up + K timed steps ----------------
    for _ in range(args.warmup):
        runner.decode_step(running)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.decode_step(running)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    ms_per_step = elapsed * 1e3 / args.steps
    tokens_per_s = B * args.steps / elapsed

    # ---------------- roofline of the dominant kernel (paged decode attention) ----------------
    lens_now = [s.req.device_len for s in running]  # contexts of the next step
    be = engine.attn_backend
    it = 2
    hq, hkv, D = be.qo_heads, be.kv_heads, mcfg.head_dim
    seq = torch.tensor(lens_now, dtype=torch.int32, device=device)
    rows = torch.tensor([s.req.table_idx for s in running], dtype=torch.int32, device=device)
    plan = torch.empty(be._plan_words, dtype=torch.int32, device=device)
    ops.attn_decode_plan(plan, seq, B, be.max_bs, be.capacity, hq, hkv)
    q = torch.randn((B, hq, D), device=device, dtype=torch.bfloat16)
    o = torch.empty_like(q)
    k_tok, v_tok = be._kv_tokens(0)
    launch = lambda: ops.attn_decode(o, q, k_tok, v_tok, engine.page_table, rows, seq, plan, be._workspace, B,  # noqa: E731
                                     be.max_bs, be.capacity, be.scale, slot_run=be.slot_run)
    for _ in range(3):
        launch()
    # One layer's launch = attention kernel + merge kernel.  Three HIP-event readings on the launch stream:
    #  * `us_per_launch` (what `achieved` / `frac` use): ONE event pair around 40 launches queued back to back, divided by
    #    40 -- covers both kernels and the gaps between them, the stream never idles (as inside the captured step); this is
    #    the figure rocprofv3 --kernel-trace reproduces as (attention + merge) average duration (+ ~0.3 us of gaps);
    #  * `us_per_launch_event_gap_median` (round 5's headline, kept for comparison): an event after every launch, median
    #    distance of consecutive events -- reads ~3 % LOW (an event's stamp is taken before the merge kernel has drained);
    #  * `us_per_launch_idle_stream` (rounds 1-4): one synchronised event pair per launch -- adds the host-to-idle-GPU
    #    launch latency to every sample (5-8 % high, box dependent).
    reps, tot = 30, 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    attn_us_idle = tot * 1e3 / reps
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 4)]
    for _ in range(3):
        launch()
    evs[0].record()
    for i in range(reps + 3):
        launch()
        evs[i + 1].record()
    evs[-1].synchronize()
    gaps = sorted(x.elapsed_time(y) for x, y in zip(evs[3:-1], evs[4:]))  # the first three gaps = ramp of the queue
    attn_us_gap = gaps[len(gaps) // 2] * 1e3
    pair_reps, pair = 40, []
    for _ in range(3):
        for _ in range(5):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(pair_reps):
            launch()
        e1.record()
        e1.synchronize()
        pair.append(e0.elapsed_time(e1) * 1e3 / pair_reps)
    attn_us = sorted(pair)[1]  # partial + merge kernels of one layer, median of three runs of 40
    S = sum(lens_now)
    attn_bytes = S * 2 * hkv * D * it + 2 * B * hq * D * it + S * 4 + 2 * B * 4
    achieved = attn_bytes / attn_us / 1e3
    # whole-step algorithmic bytes per rank: streamed weights + KV read/write + logits
    kv_tok = 2 * mcfg.num_layers * hkv * D * it
    step_bytes = engine.model.streamed_bytes_per_step() + (S + B) * kv_tok + B * mcfg.vocab_size * it
    step_gbps = step_bytes / (ms_per_step * 1e-3) / 1e9

    # ---------------- latency regime: the same model at small decode batches (not part of `value`) --------
    small = {}
    if use_graph:
        for sb in [b for b in small_batches if b < B]:
            sub = running[:sb]
            for _ in range(3):
                runner.decode_step(sub)
            barrier()
            t1 = time.perf_counter()
            for _ in range(10):
                runner.decode_step(sub)
            barrier()
            small[str(sb)] = (time.perf_counter() - t1) * 1e3 / 10

    # the same algorithmic-byte model per small batch: weights + KV of the first sb requests + their logits
    small_frac = {}
    for k_, ms_ in small.items():
        sb = int(k_)
        sb_bytes = engine.model.streamed_bytes_per_step() + (sum(lens_now[:sb]) + sb) * kv_tok + sb * mcfg.vocab_size * it
        small_frac[k_] = round(sb_bytes / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    shape = f"{model_name} tp{world} B{B} page{args.page_size} bench_contexts"
    traffic, traffic_src, traffic_ratio = pmc_traffic(attn_bytes, shape)
    prefill_roofline = None
    if rank == 0 and primary and not args.no_prefill_roofline:
        try:
            prefill_roofline = measure_prefill_attention(device, hq, hkv)
        except Exception as e:  # never lose the headline numbers to the side measurement
            prefill_roofline = {"error": f"{type(e).__name__}: {e}"}

    if collectives is not None:
        collectives["issued_from_python"] = dict(comm.issued, note="collectives issued per path; a launch captured into the decode "
                                                 "graph counts once (at capture), prefill chunks run eager")
        collectives["paths"] = ("p2p only, all ranks on ONE gpu (code-path check)" if share_gpu else
                                "p2p (decode-size) + rccl (prefill-size)" if comm.p2p is not None else "rccl")
    result = {
        "metric": METRIC, "value": tokens_per_s, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {
            "workload": f"{mcfg.name} bf16 decode step, {B} seqs, contexts from the offline-bench token-weighted "
                        f"distribution (mean {S / B:.0f}), page_size {args.page_size}, temperature 0.6, hipGraph "
                        f"{'on' if use_graph else 'off'}",
            "batch": B, "parallelism": f"tp{world}", "mean_context": S / B,
            "collectives": None if world == 1 else collectives["paths"],
        },
        "ttft_p50_ms": ttft_p50,
        "small_batch_ms_per_step": small,
        "small_batch_step_roofline_frac": small_frac,
        "roofline": {
            "bound": "hbm", "kernel": "attn_decode_mfma_kernel + attn_decode_merge_kernel, one layer (both kernels timed)", "achieved": achieved,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
            "traffic_source": traffic_src, "traffic_over_algorithmic": traffic_ratio, "us_per_launch": attn_us,
            "us_per_launch_event_gap_median": attn_us_gap, "us_per_launch_idle_stream": attn_us_idle,
            "timing": "us_per_launch = one HIP-event pair around 40 launches (attention + merge each) queued back to back / 40, median of 3: covers both kernels; event_gap_median = round 5's method (reads ~3 % low); idle_stream = one synchronised pair per launch (rounds 1-4, reads 5-8 % high)",
            "algorithmic_bytes": attn_bytes, "launch_shape": shape,
        },
        "prefill_roofline": prefill_roofline,
        "step_roofline": {
            "bound": "hbm", "bytes_per_step_per_rank": step_bytes, "achieved": step_gbps, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": step_gbps / HBM_PEAK_GBPS,
            "roofline_tokens_per_s": B * HBM_PEAK_GBPS * 1e9 / step_bytes,
        },
    }
    result["_lens_now"] = list(lens_now)  # for the cpu_baseline leg (the GPU step's own contexts); popped before printing
    if collectives is not None:
        result["collectives"] = collectives
    # projection GEMMs at the full batch: weight bytes over the search's back-to-back times (what the kernels sustain on
    # their own; inside the step the consumers of their slabs are part of the cost: profiles/r0*_kernel_breakdown.txt)
    full = [r for r in engine.gemm_report if r["M"] == B]
    if full:
        L = mcfg.num_layers
        wbytes = sum(2.0 * r["N"] * r["K"] * (1 if r["name"] == "lm_head" else L) for r in full)
        us = sum(r["best_us"] * (1 if r["name"] == "lm_head" else L) for r in full)
        result["step_roofline"]["gemm_weight_TBps"] = wbytes / us / 1e6
        result["step_roofline"]["gemm_ms_per_step_back_to_back"] = us / 1e3
    if primary and world == 1:
        ratio, src = gemm_traffic_over_algorithmic()
        # "committed": from the rocprofv3 PMC passes kept under profiles/ (counters cannot be read from inside this process)
        result["step_roofline"]["gemm_traffic_over_algorithmic"] = {"kind": "committed", "value": ratio, "source": src}
    # library GEMM solution search done at engine start (csrc/gemm.cpp): heuristic pick vs chosen, per launch
    if primary:
        result["gemm_tune"] = {
            "mode": ecfg.gemm_tune,
            "shapes": [dict(name=r["name"], M=r["M"], N=r["N"], K=r["K"], heuristic_us=round(r["default_us"], 1),
                            tuned_us=round(r["best_us"], 1), candidates=r["tried"],
                            tflops=round(2.0 * r["M"] * r["N"] * r["K"] / r["best_us"] / 1e6, 1),
                            weight_TBps=round(2.0 * r["N"] * r["K"] / r["best_us"] / 1e6, 2),
                            **({"hand_written": r["kernel"], "library_best_us": round(r["library_best_us"], 1)}
                               if r.get("skinny_used") else {}),
                            **({"fold_credit_us": r["fold_credit_us"]} if r.get("fold_credit_us") else {}),
                            **({"projection_then_silu_us": round(r["silu_unfused_us"], 1),
                                "fused_silu_epilogue_us": round(r["silu_fused_us"], 1), "fused_silu_used": r["silu_fused_used"]}
                               if r.get("silu_fused_us") else {})) for r in engine.gemm_report],
        }
        result["gemm_tune"]["refined_in_graph"] = getattr(engine, "refine_report", [])
    else:
        result["gemm_plans_at_full_batch"] = {r["name"]: dict(us=round(r["best_us"], 1), kernel=r["kernel"][:70]) for r in full}
    engine.shutdown()  # raises if a peer-to-peer barrier ever timed out (NaN-poisoned collectives)
    del runner, states, running, be, k_tok, v_tok, launch, engine, q, o
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    for c in (comm, comm_side):
        if c is not None:
            c.destroy()
    return result


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="qwen3-14b")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--page-size", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end README offline benchmark (extra keys)")
    ap.add_argument("--no-prefill", action="store_true", help="fill the KV pool with random data instead")
    ap.add_argument("--no-prefill-roofline", action="store_true", help="skip the prefill-attention MFMA measurement")
    ap.add_argument("--no-second-config", action="store_true",
                    help="--gpus 4: skip the Qwen3-32B TP4 entry (the metric's second configuration)")
    ap.add_argument("--small-batches", type=int, nargs="*", default=[1, 8, 32, 64, 128],
                    help="also report ms per decode step at these batch sizes (latency regime; [] to skip)")
    ap.add_argument("--rank-shard", type=int, default=0,
                    help="N > 1: time ONE rank's shard of a TP = N decode step on this GPU with the collectives looped back "
                         "(tools/rank_shard_bench.py): an upper bound on the TP = N speed-up before any link is paid")
    ap.add_argument("--tp1-ms", type=float, default=None, help="with --rank-shard: the TP1 ms/step to compare against")
    args = ap.parse_args()
    if args.rank_shard > 1:
        from tools.rank_shard_bench import main as rank_shard_main

        argv = ["--model", args.model, "--tp", str(args.rank_shard), "--batch", str(args.batch), "--steps", str(args.steps),
                "--warmup", str(args.warmup), "--page-size", str(args.page_size)]
        rank_shard_main(argv + (["--tp1-ms", str(args.tp1_ms)] if args.tp1_ms else []))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us (the driver's `python3 bench.py --gpus N`): start the ranks ourselves
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU fallback)"
    # MSGL_BENCH_SHARE_GPU=1: every rank on cuda:0 with the peer-to-peer communicator only (RCCL refuses two ranks per
    # device) -- exercises the whole N > 1 code path of this file on a 1-GPU box; not a measurement
    share_gpu = os.environ.get("MSGL_BENCH_SHARE_GPU") == "1"
    if not share_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py --gpus {world}: only {torch.cuda.device_count()} GPU(s) visible "
                         f"(MSGL_BENCH_SHARE_GPU=1 runs all ranks on one device as a code-path check)")
    device = torch.device("cuda:0" if share_gpu else f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    import torch.distributed as dist

    from mini_sglang_amd.model import PRESETS

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node: rendezvous and the RCCL bootstrap over loopback (the box's hostname may not resolve);
        # the data path is xGMI peer-to-peer either way
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    try:
        result = run_workload(args, args.model, rank, local_rank, world, device, share_gpu, primary=True)
    except BaseException as e:  # noqa: BLE001
        if world > 1 and rank == 0:
            # a multi-GPU run that dies after its preflight still leaves ONE line saying how far the links got
            print(json.dumps({"metric": METRIC, "value": None, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16",
                              "data": "synthetic", "config": {"workload": f"{args.model} bf16 decode step (FAILED before a measurement)",
                                                               "parallelism": f"tp{world}"},
                              "error": f"{type(e).__name__}: {e}", "collectives": dict(LAST_PREFLIGHT) or None}), flush=True)
        raise
    result["launch"] = ("self-launched: bench.py started its ranks through torch.distributed.run"
                        if os.environ.get("MSGL_BENCH_SELF_LAUNCHED") == "1" else
                        "external launcher (torchrun)" if world > 1 else "single process")
    mcfg = PRESETS[args.model]
    contexts = bench_contexts(args.batch)
    ms_per_step = result["ms_per_step"]
    # the metric's second configuration (BASELINE.json: "Qwen3-32B TP=4"): at --gpus 4 the same request set on Qwen3-32B.
    # $MSGL_BENCH_SECOND_MODEL runs a second workload at any N (how the two-workloads-in-one-process path -- communicators
    # torn down and rebuilt, a second engine in the freed memory -- is exercised on a 1-GPU box with a small model)
    second = os.environ.get("MSGL_BENCH_SECOND_MODEL") or ("qwen3-32b" if world == 4 and args.model == "qwen3-14b" else None)
    if second and not args.no_second_config:
        key = "qwen3_32b_tp4" if (second == "qwen3-32b" and world == 4) else f"second_config_{second}_tp{world}"
        try:
            r2 = run_workload(args, second, rank, local_rank, world, device, share_gpu, primary=False)
            result[key] = {k: r2[k] for k in ("value", "unit", "ms_per_step", "ttft_p50_ms", "config", "roofline",
                                              "step_roofline", "collectives", "gemm_plans_at_full_batch") if k in r2}
        except Exception as e:
            result[key] = {"error": f"{type(e).__name__}: {e}"}
    # what the REFERENCE's own LLM / Scheduler / GraphRunner measure on this path through the plugin: recorded by
    # tests/test_gpu_reference_driven.py (it may import oracle/_ref, this file may not) and committed under profiles/
    ref_runs = sorted((ROOT / "profiles").glob("r*_refdrive_14b.json"))
    if ref_runs and args.model == "qwen3-14b" and world == 1:
        try:
            d = json.loads(ref_runs[-1].read_text())
            # "committed": recorded by the GPU test suite on a box of the same kind and kept under profiles/; this file may
            # not import oracle/_ref (the reference), so it cannot re-measure it
            # stale = the sources have changed since that run was recorded (fingerprint stored by the test that wrote the file;
            # files from before round 5 carry none): the number then describes an EARLIER state of the kernels
            fp = product_code_fingerprint()
            result["reference_driven"] = {"kind": "committed", "decode_ms_per_step": d["decode_ms_per_step_median"], "tokens_per_s": d["tokens_per_s"],
                                          "vs_this_run_ms_per_step": d["decode_ms_per_step_median"] / ms_per_step,
                                          "driver": d["driver"], "source": f"profiles/{ref_runs[-1].name}",
                                          "recorded_on_code": d.get("code_fingerprint"), "this_code": fp,
                                          "stale": d.get("code_fingerprint") != fp}
        except Exception as e:
            result["reference_driven"] = {"error": f"{type(e).__name__}: {e}"}
    # the ONLINE half of the metric (BASELINE config 4's protocol): the reference's Scheduler fed a synthetic Qwen-like trace at
    # the reference's scale list through the plugin -- recorded by tools/trace_replay.py (it imports oracle/_ref, this file may
    # not) and committed under profiles/; stale when the product sources have changed since
    if world == 1:
        fp = product_code_fingerprint()
        replays = {}
        for f in sorted((ROOT / "profiles").glob("r*_trace_replay_*.json")):
            try:
                d = json.loads(f.read_text())
            except Exception:
                continue
            if "scales" in d and d.get("model"):
                replays[d["model"]] = {"kind": "committed", "source": f"profiles/{f.name}", "cache": d.get("cache"), "tp": d.get("tp"),
                                       "trace": d.get("trace"), "recorded_on_code": d.get("code_fingerprint"), "this_code": fp,
                                       "stale": d.get("code_fingerprint") != fp,
                                       "by_scale": {k: {"complete": v["complete"], "throughput_tok_s": v["throughput_tok_s"],
                                                        "ttft_ms": {q: v["ttft_ms"][q] for q in ("p50", "p90", "p99")},
                                                        "tpot_ms": {q: v["tpot_ms"][q] for q in ("p50", "p90", "p99")}}
                                                    for k, v in d["scales"].items()}}
        if replays:
            result["online_trace_replay"] = replays
        # what the GPU parity suite compared on its last committed run (tests/parity_stats.py -> profiles/r*_parity_summary.json)
        par = sorted((ROOT / "profiles").glob("r*_parity_summary.json"))
        if par:
            try:
                result["parity"] = dict(json.loads(par[-1].read_text()), kind="committed", source=f"profiles/{par[-1].name}")
            except Exception as e:
                result["parity"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_e2e:
        # the reference's own throughput definition (benchmark/offline/bench.py:32-38): sum(max_tokens) / wall of one
        # generate() over 256 requests, in/out 100..1024, prefill included, one untimed warm-up -- BASELINE configs 1, 2
        from tools.offline_bench import run as offline_run

        result["e2e_offline"] = {}
        for name in (["qwen3-0.6b"] + ([args.model] if args.model != "qwen3-0.6b" else [])):
            try:
                r = offline_run(name, 256, 0.6, args.page_size, os.environ.get("MSGL_GEMM_TUNE", "full"), device)
                result["e2e_offline"][name] = {k: r[k] for k in ("throughput_tok_s", "wall_s", "output_tokens", "input_tokens",
                                                                 "decode_steps", "ms_per_decode_step", "prefill_tok_s",
                                                                 "ttft_p50_ms")}
            except Exception as e:
                result["e2e_offline"][name] = {"error": f"{type(e).__name__}: {e}"}
        # the reference's DEFAULT page size (P/engine/config.py:25: page_size = 1; its FlashInfer backend always sees token-granular
        # tables, P/attention/fi.py:179-187): the same workload on a token-granular pool (round 6: the matrix-core decode kernel
        # gathers token rows by per-lane addresses)
        try:
            r = offline_run(args.model, 256, 0.6, 1, os.environ.get("MSGL_GEMM_TUNE", "full"), device)
            result["e2e_offline_page_size_1"] = {args.model: {k: r[k] for k in ("throughput_tok_s", "wall_s", "decode_steps", "ms_per_decode_step",
                                                                               "prefill_tok_s", "ttft_p50_ms", "page_size")}}
        except Exception as e:
            result["e2e_offline_page_size_1"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(mcfg, result.pop("_lens_now", contexts))
        except Exception as e:  # never lose the GPU numbers to a host-side problem
            result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    result.pop("_lens_now", None)
    if rank == 0:
        # the LAST object of the line (a reader that keeps only the tail of stdout still sees the headline numbers)
        e2e = result.get("e2e_offline", {})
        result["summary"] = {
            "ms_per_step": round(ms_per_step, 3), "tokens_per_s": round(result["value"]),
            "step_frac_of_8TBps": round(result["step_roofline"]["frac"], 4),
            "attn_decode_us": round(result["roofline"]["us_per_launch"], 1), "attn_decode_frac": round(result["roofline"]["frac"], 4),
            "small_batch_ms_per_step": result.get("small_batch_ms_per_step"),
            "ttft_p50_ms": result.get("ttft_p50_ms"),
            "e2e_offline_tok_s": {k: round(v["throughput_tok_s"]) for k, v in e2e.items() if "throughput_tok_s" in v},
            "e2e_offline_page_size_1_tok_s": {k: round(v["throughput_tok_s"]) for k, v in result.get("e2e_offline_page_size_1", {}).items()
                                              if isinstance(v, dict) and "throughput_tok_s" in v},
            "small_batch_step_roofline_frac": result.get("small_batch_step_roofline_frac"),
            "bit_identical_forwards_vs_reference_driven": (result.get("parity") or {}).get("bit_identical_forwards"),
            "prefill_attn_frac": (result.get("prefill_roofline") or {}).get("frac"),
            "reference_driven_ms": (result.get("reference_driven") or {}).get("decode_ms_per_step"),
            "reference_driven_stale": (result.get("reference_driven") or {}).get("stale"),
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
