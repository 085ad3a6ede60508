"""TEST INFRASTRUCTURE: one reference-driven scenario in a fresh process (see tests/refdrive.py).

    python tests/refdrive_worker.py spec.json out.pt <dir holding the reference's `minisgl` package>

Runs the REFERENCE's `LLM` (P/llm/llm.py:28-101: Scheduler + Engine + GraphRunner + CacheManager/radix cache,
overlap loop) with `attention_backend="hip"` after `minisgl_plugin.install()`, and records, per forward the
reference's engine executes, exactly what the hot path was handed (the `Batch`: phase, request rows, lengths,
input ids, positions, out_loc, the page-table rows) and what it produced (logits summary, sampled ids), plus the
generated token ids.  The recording wraps methods of the reference's *instances* from the outside; no reference
source is modified.
"""
from __future__ import annotations

import json
import os
import socket
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def logits_summary(logits):
    """Bit-level fingerprint of a logits matrix: argmax, top-2 values, and an order-independent checksum of the
    raw bit patterns per row (two matrices with equal checksums, argmax and top-2 are taken as bit-identical)."""
    import torch

    lf = logits.float()
    top2 = lf.topk(2, dim=-1)
    bits = lf.contiguous().view(torch.int32).to(torch.int64)
    weights = torch.arange(1, lf.shape[1] + 1, device=lf.device, dtype=torch.int64)
    return dict(argmax=top2.indices[:, 0].to(torch.int32).cpu(), top2=top2.values.cpu(),
                checksum=(bits.sum(-1) + (bits * weights).sum(-1)).cpu(), dtype=str(logits.dtype))


def replay_through_repo_engine(spec, forwards, ref_engine, tp_rank, tp_size, rec):
    """Inside the rank processes of a tp > 1 scenario: the recorded batches through THIS repository's Engine at the
    same tp (same ranks, a second peer-to-peer communicator), bit-compared with what the reference's engine produced;
    rank 0 also replays them through a tp = 1 engine (all weights) for the tolerance check."""
    import torch
    import torch.distributed as dist
    from safetensors.torch import load_file

    from mini_sglang_amd.engine import Engine, EngineConfig
    from mini_sglang_amd.kernel import init_pynccl
    from mini_sglang_amd.model import PRESETS
    from replay_util import replay_forward

    kw, dev = spec["llm_kwargs"], ref_engine.device
    state = load_file(str(Path(spec["model_dir"]) / "model.safetensors"))
    group = dist.group.WORLD
    max_bytes = kw.get("max_extend_tokens", 8192) * PRESETS[spec["model"]].hidden_size * 2

    split = int(os.environ.get("MSGL_COMM_SPLIT_TOKENS", "2048"))  # the plugin's rule: the repo engine must split alike

    def build(size, rank, comm):
        cfg = EngineConfig(model=PRESETS[spec["model"]], dtype=torch.bfloat16, tp_rank=rank, tp_size=size,
                           max_running_req=kw["max_running_req"], page_size=kw["page_size"],
                           cuda_graph_bs=list(rec["graph_bs"]), max_seq_len_override=kw["max_seq_len_override"],
                           num_page_override=rec["num_pages"], fused_qkv_path=True, gemm_tune="off", comm=comm,
                           comm_side=getattr(comm, "side", None), comm_split_tokens=split if size > 1 else 0,
                           tp_cpu_group=group if size > 1 else None)
        eng = Engine(cfg, dev)
        eng.model.load_hf_state(state)
        return eng

    out = {}
    comm = init_pynccl(tp_rank=tp_rank, tp_size=tp_size, tp_cpu_group=group, max_size_bytes=max_bytes, backend="p2p",
                       side=True)
    eng = build(tp_size, tp_rank, comm)
    same, tp_logits = [], []
    for f in forwards:
        lg = replay_forward(eng, {k: v for k, v in f.items()})
        mine = logits_summary(lg)
        r = f["summary"]
        same.append(bool(torch.equal(r["argmax"], mine["argmax"])
                         and torch.equal(r["top2"], mine["top2"]) and torch.equal(r["checksum"], mine["checksum"])))
        tp_logits.append(lg.float().cpu())
    out["bit_identical"] = same
    out["comm_error"] = comm.p2p.error()
    eng.shutdown()
    dist.barrier()
    comm.destroy()
    if tp_rank == 0:
        eng1 = build(1, 0, None)
        out["max_abs_vs_tp1"] = [float((replay_forward(eng1, f).float().cpu() - t).abs().max()) for f, t in zip(forwards, tp_logits)]
        eng1.shutdown()
    dist.barrier()
    return out


def trace_replay(LLM, model_dir, kw, spec, tp_args):
    """BASELINE config 4 as a scheduler-level replay (benchmark/online/bench_qwen.py:37-51, P/benchmark/client.py:284-384):
    requests ARRIVE at the trace's timestamps, prompts are prefixes of one token sequence (the benchmark's `dummy=True`
    mode, client.py:428-431: the radix cache sees shared prefixes), every request generates exactly `output_length` tokens
    (ignore_eos), and per-token arrival times give TTFT / TPOT / E2E with the percentile rule of process_benchmark_results.
    The HTTP front end, tokenizer workers and ZMQ hops of the online server are NOT in the loop (out of scope, SURVEY.md
    section 8): the clock runs from a request's scheduled arrival to the scheduler handing its tokens to offline_send_result."""
    import time

    import torch
    from minisgl.core import SamplingParams
    from minisgl.llm.llm import RequestAllFinished
    from minisgl.message import UserMsg

    trace = sorted(spec["trace"], key=lambda r: r["t"])
    scale = float(spec.get("trace_scale", 1.0))
    g = torch.Generator().manual_seed(7)
    base = torch.randint(0, 10000, (max(r["input_length"] for r in trace) + 1,), generator=g, dtype=torch.int32)

    class TraceLLM(LLM):
        tracing = False

        def begin(self):
            self.nxt, self.t0, self.tics, self.open_, self.tracing = 0, time.perf_counter(), {}, 0, True

        def offline_receive_msg(self, blocking=False):
            if not self.tracing:  # the warm-up generate()
                return super().offline_receive_msg(blocking)
            while True:
                now = time.perf_counter() - self.t0
                out = []
                while self.nxt < len(trace) and trace[self.nxt]["t"] * scale <= now and len(out) < 64:
                    r = trace[self.nxt]
                    uid = self.nxt
                    self.tics[uid] = [r["t"] * scale]
                    out.append(UserMsg(uid=uid, input_ids=base[: r["input_length"]].clone(),
                                       sampling_params=SamplingParams(temperature=0.0, max_tokens=int(r["output_length"]), ignore_eos=True)))
                    self.nxt += 1
                    self.open_ += 1
                if out or not blocking:
                    return out
                if self.nxt >= len(trace):
                    raise RequestAllFinished()
                time.sleep(max(0.0, min(0.005, trace[self.nxt]["t"] * scale - now)))

        def offline_send_result(self, reply):
            if not self.tracing:
                return super().offline_send_result(reply)
            t = time.perf_counter() - self.t0
            for msg in reply:
                self.tics[msg.uid].append(t)
                if msg.finished:
                    self.open_ -= 1

    llm = TraceLLM(model_dir, *tp_args, **kw)
    # one untimed warm-up request (library kernels for the shapes, allocator pools), as the offline benchmark does
    llm.generate([[1, 2, 3, 4, 5, 6, 7, 8]], SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True))
    llm.begin()
    try:
        llm.run_forever()
    except RequestAllFinished:
        pass
    torch.cuda.synchronize()
    tics = [llm.tics[i] for i in range(len(trace))]
    first = sorted(t[1] - t[0] for t in tics if len(t) > 1)
    accum = sorted(d for t in tics for d in (b - a for a, b in zip(t[1:-1], t[2:])))
    e2e = sorted(t[-1] - t[0] for t in tics)

    def stats(v, k=1.0):
        return dict(avg=k * sum(v) / len(v), p50=k * v[int(len(v) * 0.5)], p90=k * v[int(len(v) * 0.9)], p99=k * v[int(len(v) * 0.99)],
                    max=k * v[-1]) if v else None

    dur = max(t[-1] for t in tics) - min(t[0] for t in tics)
    ntok = sum(len(t) - 1 for t in tics)
    complete = all(len(t) - 1 == r["output_length"] for t, r in zip(tics, trace))
    return llm, dict(requests=len(trace), tokens=ntok, complete=complete, duration_s=dur, throughput_tok_s=ntok / dur,
                     req_per_s=len(trace) / dur, ttft_ms=stats(first, 1e3), tpot_ms=stats(accum, 1e3), e2e_s=stats(e2e),
                     input_tokens=sum(r["input_length"] for r in trace), trace_scale=scale)


def gated_mlp_checks(plugin, engine, model_dir):
    """a GatedMLP whose gate_up rows the plugin interleaved must never take the reference forward on
    them.  Inputs the fused path used to hand to the reference forward (3-D, non-contiguous), the un-permute helpers, and
    a weight REPLACED by load_state_dict-style setattr (P/layers/base.py:31-49)."""
    import torch
    from safetensors.torch import load_file

    layers = engine.model.model.layers.op_list
    m0, m1 = layers[0].mlp, layers[1].mlp
    out = dict(interleaved=bool(getattr(m0, "_msgl_gate_up_ilv", False) and getattr(m1, "_msgl_gate_up_ilv", False)))
    if not out["interleaved"]:
        return out
    state = load_file(str(Path(model_dir) / "model.safetensors"))
    hf = [torch.cat([state[f"model.layers.{i}.mlp.gate_proj.weight"], state[f"model.layers.{i}.mlp.up_proj.weight"]], 0).to(engine.device)
          for i in (0, 1)]
    H = hf[0].shape[1]
    g = torch.Generator(device=engine.device).manual_seed(5)
    x = torch.randn((5, H), generator=g, device=engine.device).to(torch.bfloat16)
    wide = torch.randn((5, 2 * H), generator=g, device=engine.device).to(torch.bfloat16)
    y0 = m0.forward(x)
    out["three_d_equal"] = bool(torch.equal(m0.forward(x.view(1, 5, H)), y0.view(1, 5, -1)))
    xs = wide[:, ::2]  # stride(1) == 2
    out["strided_equal"] = bool(torch.equal(m0.forward(xs), m0.forward(xs.contiguous())))
    out["reference_layout_equal"] = bool(torch.equal(plugin.gate_up_reference(m0), hf[0]))
    y1 = m1.forward(x)
    # a reload replaces the tensor object (reference layout): the layer must turn back into a plain reference layer
    m1.gate_up_proj.weight = hf[1].clone()
    y1b = m1.forward(x)
    out["replaced_flag_cleared"] = not getattr(m1, "_msgl_gate_up_ilv", True)
    out["replaced_max_abs"] = float((y1b.float() - y1.float()).abs().max())
    out["out_absmax"] = float(y1.float().abs().max())
    # restore everything in place: rows back to [gate; up], flags cleared, outputs unchanged
    out["restored_layers"] = plugin.restore_gate_up_layout(engine.model)
    out["restored_equal_hf"] = bool(torch.equal(m0.gate_up_proj.weight, hf[0]))
    out["restored_max_abs"] = float((m0.forward(x).float() - y0.float()).abs().max())
    out["reinterleaved"] = plugin._interleave_gated_mlps(engine.model)
    out["reinterleaved_equal"] = bool(torch.equal(m0.forward(x), y0))
    torch.cuda.synchronize()
    return out


def main() -> None:
    spec_path, out_path, ref_root = sys.argv[1:4]
    spec = json.loads(Path(spec_path).read_text())
    sys.path.insert(0, ref_root)
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    tp_size = int(spec.get("tp_size", 1))
    tp_rank = int(os.environ.get("MSGL_REFDRIVE_RANK", "0"))
    if tp_size > 1:
        out_path = f"{out_path}.{tp_rank}"
        os.environ.setdefault("MSGL_COMM_BACKEND", spec.get("comm_backend", "p2p"))
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")

    import mini_sglang_amd.minisgl_plugin as plugin

    plugin.install(fast_linear=spec.get("fast_linear", True), fused_attention=spec.get("fused_attention", True),
                   gemm_tune=spec.get("gemm_tune", "off"),
                   deterministic_decode_order=spec.get("deterministic_decode_order", False),
                   vectorized_glue=spec.get("vectorized_glue", False), native_radix=spec.get("native_radix", False))
    import torch

    from refdrive import write_model_dir

    # forced full-batch GEMM plans [[M, N, K, grid, full, tail_split], ...] (contiguous operands, bf16): lets a
    # bit-identity scenario exercise the hand-written kernel and the reduce-in-norm hand-off without a timing search
    if spec.get("m256_plans"):
        from mini_sglang_amd import ops

        for M, N, K, grid, full, split in spec["m256_plans"]:
            ops._M256_PLAN[(0, M, N, K, K, K, ops._dt(torch.empty(0, dtype=torch.bfloat16)))] = (grid, full, split)

    model_dir = spec.get("model_dir")
    if model_dir is None:
        model_dir = str(write_model_dir(Path(tempfile.mkdtemp(prefix="msgl_model_")) / spec["model"], spec["model"],
                                        weights=spec.get("weights", "seeded") == "seeded",
                                        max_position=spec.get("max_position", 40960)))

    # the reference hard-wires tcp://127.0.0.1:2333 for its gloo group (P/engine/config.py:54-55); successive
    # worker processes on one box would race for it
    from minisgl.engine.config import EngineConfig

    port = spec.get("port")
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    EngineConfig.distributed_addr = property(lambda self: f"tcp://127.0.0.1:{port}")  # type: ignore[assignment]

    from minisgl.core import SamplingParams
    from minisgl.llm import LLM

    kw = dict(spec.get("llm_kwargs", {}))
    kw.setdefault("attention_backend", "hip")
    kw["use_dummy_weight"] = spec.get("weights", "seeded") == "dummy"
    t0 = time.perf_counter()
    if spec.get("trace"):
        llm, result = trace_replay(LLM, model_dir, kw, spec, ())
        cm = llm.cache_manager
        rec = dict(spec={k: v for k, v in spec.items() if k != "trace"}, trace_replay=result, init_and_run_s=time.perf_counter() - t0,
                   gemm_report=plugin.gemm_report(), backend=type(llm.engine.attn_backend).__name__,
                   prefix_cache=type(cm.prefix_cache).__name__, graph_bs=list(llm.engine.graph_runner.graph_bs_list),
                   num_pages=llm.engine.num_pages, device=torch.cuda.get_device_name(0))
        try:
            cm.check_integrity()
            rec["integrity"] = "ok"
        except Exception as e:
            rec["integrity"] = f"{type(e).__name__}: {e}"
        torch.save(rec, out_path)
        try:
            llm.shutdown()
        except Exception as e:
            print(f"[worker] shutdown: {type(e).__name__}: {e}", file=sys.stderr)
        return
    if tp_size == 1:
        llm = LLM(model_dir, **kw)
    else:
        # The reference's offline `LLM` hard-wires tp_info = (0, 1) (P/llm/llm.py:31); its TP ranks are scheduler
        # processes fed over ZMQ.  Here every rank is an offline LLM with its real tp_info (offline mode takes the same
        # early exit on every rank, P/scheduler/io.py:30-33): the replicated schedulers see the same request list.
        # All ranks share ONE GPU on a 1-GPU box: the reference picks its device as f"cuda:{tp_info.rank}"
        # (P/engine/engine.py:35) -- the rank below formats as "0" there and is the true rank everywhere else.
        from minisgl.distributed import DistributedInfo
        from minisgl.scheduler import Scheduler, SchedulerConfig

        class _RankOnDevice0(int):
            def __format__(self, spec_):
                return "0"

        class TPLLM(LLM):
            def __init__(self, model_path, tp_info, dtype=torch.bfloat16, **kwargs):
                config = SchedulerConfig(model_path=model_path, tp_info=tp_info, dtype=dtype, offline_mode=True, **kwargs)
                Scheduler.__init__(self, config)
                self.pending_requests, self.status_map, self.counter = [], {}, 0

        share = torch.cuda.device_count() < tp_size
        llm = TPLLM(model_dir, DistributedInfo(_RankOnDevice0(tp_rank) if share else tp_rank, tp_size), **kw)
    init_s = time.perf_counter() - t0
    engine = llm.engine

    record_level = spec.get("record", "batches")  # "none" | "timing" | "batches"
    full_logits = int(spec.get("full_logits_forwards", 0))
    forwards = []
    state = dict(round=0)
    orig_forward_batch, orig_sample = engine.forward_batch, engine.sampler.sample

    def forward_batch(batch, args):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        e = dict(round=state["round"], phase=batch.phase, size=batch.size, padded_size=batch.padded_size, event=ev,
                 total_tokens=int(sum(r.extend_len for r in batch.padded_reqs)))
        if record_level == "batches":
            padded = batch.padded_reqs
            rows = [r.table_idx for r in padded]
            max_len = max(r.device_len for r in padded)
            e.update(uids=[r.uid for r in padded], rows=rows, cached_lens=[r.cached_len for r in padded],
                     device_lens=[r.device_len for r in padded], chunked=[type(r).__name__ == "ChunkedReq" for r in padded],
                     input_ids=batch.input_ids.clone(), positions=batch.positions.clone(), out_loc=batch.out_loc.clone(),
                     table=engine.page_table[torch.tensor(rows, device=engine.device)][:, :max_len].clone(),
                     metadata_type=type(batch.attn_metadata).__name__,
                     graph=bool(engine.graph_runner.can_use_cuda_graph(batch)))
        forwards.append(e)
        out = orig_forward_batch(batch, args)
        if record_level == "batches":
            e["next_tokens"] = out.next_tokens_gpu.clone()
        return out

    def sample(logits, args):
        e = forwards[-1]
        if record_level == "batches":
            e["summary"] = logits_summary(logits)
            if len([f for f in forwards if "logits" in f]) < full_logits:
                e["logits"] = logits.float().cpu()
        return orig_sample(logits, args)

    if record_level != "none":
        engine.forward_batch = forward_batch
        engine.sampler.sample = sample

    # host time of the scheduler's three phases per loop iteration (P/scheduler/scheduler.py:83-119): what the backend's
    # host side (prepare_metadata, the glue tensors, the radix cache) costs next to the GPU step
    host = dict(schedule_s=0.0, forward_s=0.0, process_s=0.0, iterations=0)
    if spec.get("host_timing"):
        o_sched, o_fwd, o_proc = llm._schedule_next_batch, llm._forward, llm._process_last_data

        def t_sched():
            t = time.perf_counter()
            r = o_sched()
            host["schedule_s"] += time.perf_counter() - t
            host["iterations"] += r is not None
            return r

        def t_fwd(fi_):
            t = time.perf_counter()
            r = o_fwd(fi_)
            host["forward_s"] += time.perf_counter() - t
            return r

        def t_proc(ld):
            t = time.perf_counter()
            r = o_proc(ld)
            host["process_s"] += time.perf_counter() - t
            return r

        llm._schedule_next_batch, llm._forward, llm._process_last_data = t_sched, t_fwd, t_proc

    outputs, walls = [], []
    for ri, rnd in enumerate(spec["rounds"]):
        state["round"] = ri
        sps = [SamplingParams(**sp) for sp in rnd["sampling"]]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = llm.generate(rnd["prompts"], sps)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        outputs.append([r["token_ids"] for r in res])

    # timing: ms between the starts of consecutive forwards (steady-state step time incl. host work)
    for a, b in zip(forwards[:-1], forwards[1:]):
        a["ms_to_next"] = a["event"].elapsed_time(b["event"]) if a["round"] == b["round"] else None
    for f in forwards:
        f.pop("event", None)
        for k, v in list(f.items()):
            if isinstance(v, torch.Tensor):
                f[k] = v.cpu()

    cm = llm.cache_manager
    rec = dict(spec=spec, outputs=outputs, walls=walls, forwards=forwards, init_s=init_s, host=host,
               num_pages=engine.num_pages, max_seq_len=engine.max_seq_len, page_table_shape=tuple(engine.page_table.shape),
               graph_bs=list(engine.graph_runner.graph_bs_list), backend=type(engine.attn_backend).__name__,
               attention_forward_fused=bool(getattr(type(engine.model.model.layers.op_list[0].self_attn.attn).forward,
                                                    "_msgl_fused", False)),
               gemm_report=plugin.gemm_report(), refine_report=plugin._STATE.get("refine_report", []), prefix_cache=type(cm.prefix_cache).__name__, deferred_reduce_weights=len(plugin._STATE["deferred_reduce_weights"]), free_pages_end=int(len(cm.free_slots)),
               evictable_end=int(cm.prefix_cache.size_info.evictable_size), device=torch.cuda.get_device_name(0))
    if spec.get("export_gemm_plans"):
        # the search's result as data (hand-written kernels' plan tables + the library's picks per searched shape): the test
        # process installs exactly these plans in its own engine before it replays the recorded forwards
        from mini_sglang_amd import ops as _ops

        code = _ops._dt(torch.empty(0, dtype=torch.bfloat16))
        shapes = [(r["M"], r["N"], r["K"], r["K"], r["K"], r["N"], code) for r in plugin.gemm_report()]
        shapes += [(r["M"], r["N"], r["K"], r["K"], r["K"], r["N"] // 2, code) for r in plugin.gemm_report()]
        rec["gemm_plans"] = _ops.export_gemm_plans(shapes)
        rec["plan_labels"] = {f"{r['name']}@{r['M']}": r["kernel"][:90] for r in plugin.gemm_report()}
    rec["tp_rank"], rec["tp_size"] = tp_rank, tp_size
    if tp_size > 1:
        import minisgl.distributed.impl as dimpl

        plug = dimpl.DistributedCommunicator.plugins[-1]
        comm = getattr(plug, "comm", None)
        rec["comm_class"] = type(comm).__name__
        rec["comm_p2p_error"] = comm.p2p.error() if getattr(comm, "p2p", None) is not None else None
        rec["comm_has_rccl"] = getattr(comm, "rccl", None) is not None
        rec["interleaved_mlps"] = plugin._STATE.get("interleaved_mlps")
        rec["overlapped_projections"] = plugin._STATE.get("overlapped_projections", 0)
        rec["comm_has_side"] = getattr(comm, "side", None) is not None
    if spec.get("replay_repo_engine"):
        rec["repo_replay"] = replay_through_repo_engine(spec, forwards, engine, tp_rank, tp_size, rec)
    if spec.get("mlp_checks"):
        rec["mlp_checks"] = gated_mlp_checks(plugin, engine, model_dir)
    try:
        cm.check_integrity()
        rec["integrity"] = "ok"
    except Exception as e:  # recorded, asserted by the test
        rec["integrity"] = f"{type(e).__name__}: {e}"
    torch.save(rec, out_path)
    try:
        llm.shutdown()
    except Exception as e:
        print(f"[worker] shutdown: {type(e).__name__}: {e}", file=sys.stderr)


if __name__ == "__main__":
    main()
