"""TEST INFRASTRUCTURE: drive the REFERENCE's own host code (LLM -> Scheduler -> Engine -> GraphRunner ->
radix cache, /root/reference/python/minisgl/llm/llm.py:28,77, scheduler/scheduler.py:83-233,
engine/graph.py:105-166) on the MI355X through `mini_sglang_amd.minisgl_plugin.install()`.

The reference is pure Python above the seams; `oracle/build_ref.sh` places a git-ignored copy of the package
under oracle/_ref/ (it ships to the GPU box with the snapshot, /root/reference does not).  Nothing here is
imported by the product; only tests/ use it.

Pieces:
  reference_root()        where an importable `minisgl` lies (oracle/_ref, else /root/reference/python), or None
  hf_config(...)          the config.json the reference reads through AutoConfig (P/utils/hf.py:31-37)
  seeded_hf_state(...)    HF-named weight tensors N(0, 0.02^2), norms = 1
  write_model_dir(...)    config.json + tokenizer stub (+ model.safetensors) -- what SURVEY.md Appendix C lists
  run_worker(spec)        run tests/refdrive_worker.py in a fresh process (the reference's Engine insists on an
                          uninitialised CUDA context, P/engine/engine.py:31) and load what it recorded
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path
from typing import Any, Dict, List, Optional

ROOT = Path(__file__).resolve().parent.parent


def reference_root() -> Optional[Path]:
    for cand in (ROOT / "oracle" / "_ref", Path("/root/reference/python")):
        if (cand / "minisgl" / "llm" / "llm.py").exists():
            return cand
    return None


# ------------------------------------------------------------------------------ model directory
DIMS = {
    # name: (layers, hidden, q heads, kv heads, head_dim, intermediate, vocab, tied)
    "tiny": (2, 256, 10, 2, 128, 512, 1024, False),
    "qwen3-0.6b": (28, 1024, 16, 8, 128, 3072, 151936, True),
    "qwen3-14b": (40, 5120, 40, 8, 128, 17408, 151936, False),
    "qwen3-32b": (64, 5120, 64, 8, 128, 25600, 151936, False),
    "qwen3-14b-width-2l": (2, 5120, 40, 8, 128, 17408, 32768, False),   # model.PRESETS: the 14B layer at full width, 2 GB
}


def hf_config(model: str, max_position: int = 40960) -> Dict[str, Any]:
    L, H, hq, hkv, D, inter, V, tied = DIMS[model]
    return {
        "architectures": ["Qwen3ForCausalLM"], "model_type": "qwen3", "hidden_size": H, "intermediate_size": inter,
        "num_hidden_layers": L, "num_attention_heads": hq, "num_key_value_heads": hkv, "head_dim": D,
        "vocab_size": V, "hidden_act": "silu", "rms_norm_eps": 1e-6, "max_position_embeddings": max_position,
        "tie_word_embeddings": tied, "attention_bias": False, "torch_dtype": "bfloat16",
        "rope_parameters": {"rope_theta": 1000000.0, "rope_type": "default"}, "rope_theta": 1000000.0,
    }


def seeded_hf_state(model: str, seed: int = 42, device: str = "cpu", std: float = 0.02):
    """Full (unsharded) HF-named tensors, bf16: projections N(0, std^2) from one seeded generator in a fixed
    order, norm weights 1."""
    import torch

    L, H, hq, hkv, D, inter, V, tied = DIMS[model]
    g = torch.Generator(device=device).manual_seed(seed)

    def w(*shape):
        return (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std).to(torch.bfloat16)

    def ones(n):
        return torch.ones(n, device=device, dtype=torch.bfloat16)

    st = {"model.embed_tokens.weight": w(V, H)}
    for i in range(L):
        p = f"model.layers.{i}."
        st[p + "input_layernorm.weight"] = ones(H)
        st[p + "self_attn.q_proj.weight"] = w(hq * D, H)
        st[p + "self_attn.k_proj.weight"] = w(hkv * D, H)
        st[p + "self_attn.v_proj.weight"] = w(hkv * D, H)
        st[p + "self_attn.q_norm.weight"] = ones(D)
        st[p + "self_attn.k_norm.weight"] = ones(D)
        st[p + "self_attn.o_proj.weight"] = w(H, hq * D)
        st[p + "post_attention_layernorm.weight"] = ones(H)
        st[p + "mlp.gate_proj.weight"] = w(inter, H)
        st[p + "mlp.up_proj.weight"] = w(inter, H)
        st[p + "mlp.down_proj.weight"] = w(H, inter)
    st["model.norm.weight"] = ones(H)
    if not tied:
        st["lm_head.weight"] = w(V, H)
    return st


def write_model_dir(path: Path, model: str, *, weights: bool, seed: int = 42, max_position: int = 40960,
                    device: str = "cpu") -> Path:
    """A local 'checkpoint' the reference accepts as `model_path` (P/utils/hf.py:40-43 takes a directory)."""
    from tokenizers import Tokenizer, models, pre_tokenizers

    path.mkdir(parents=True, exist_ok=True)
    cfg = hf_config(model, max_position)
    (path / "config.json").write_text(json.dumps(cfg, indent=1))
    V = cfg["vocab_size"]
    vocab = {f"t{i}": i for i in range(V - 1)}
    vocab["<eos>"] = V - 1
    tok = Tokenizer(models.WordLevel(vocab, unk_token="t0"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(str(path / "tokenizer.json"))
    (path / "tokenizer_config.json").write_text(json.dumps(
        {"tokenizer_class": "PreTrainedTokenizerFast", "eos_token": "<eos>", "unk_token": "t0"}))
    if weights:
        from safetensors.torch import save_file

        st = {k: v.contiguous().cpu() for k, v in seeded_hf_state(model, seed, device).items()}
        save_file(st, str(path / "model.safetensors"))
    return path


def synth_qwen_trace(n: int, rate_per_s: float, seed: int = 42):
    """A stand-in for the Qwen usage trace the reference's online benchmark downloads (benchmark/online/bench_qwen.py:21,
    fields timestamp / input_length / output_length, client.py:413-421): no network here, so Poisson arrivals at
    `rate_per_s` with log-normal lengths (input median ~900 tokens, sigma 1.0, clipped to [16, 6000]; output median ~200,
    sigma 0.8, clipped to [8, 1000]).  NOT the real trace: a workload of the same kind, stated as such wherever reported."""
    import math
    import random

    rnd = random.Random(seed)
    t, out = 0.0, []
    for _ in range(n):
        t += rnd.expovariate(rate_per_s)
        inp = int(min(6000, max(16, math.exp(rnd.gauss(math.log(900), 1.0)))))
        outl = int(min(1000, max(8, math.exp(rnd.gauss(math.log(200), 0.8)))))
        out.append(dict(t=round(t, 4), input_length=inp, output_length=outl))
    return out


# ------------------------------------------------------------------------------ worker process
def run_worker(spec: Dict[str, Any], timeout: float = 900.0) -> Dict[str, Any]:
    """Run one scenario in a fresh interpreter; returns the recorded dict (torch.load of the worker's file)."""
    import torch

    ref = reference_root()
    assert ref is not None, "no importable reference (run oracle/build_ref.sh in the build container)"
    with tempfile.TemporaryDirectory(prefix="msgl_refdrive_") as td:
        spec_path, out_path = Path(td) / "spec.json", Path(td) / "out.pt"
        spec_path.write_text(json.dumps(spec))
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("MSGL_GEMM_TUNE", None)
        proc = subprocess.run([sys.executable, str(ROOT / "tests" / "refdrive_worker.py"), str(spec_path),
                               str(out_path), str(ref)], env=env, capture_output=True, text=True, timeout=timeout)
        if proc.returncode != 0 or not out_path.exists():
            tail = (proc.stdout[-3000:] + "\n--- stderr ---\n" + proc.stderr[-6000:])
            raise AssertionError(f"reference-driven worker failed (rc {proc.returncode}):\n{tail}")
        rec = torch.load(out_path, weights_only=False)
        rec["stderr_tail"] = proc.stderr[-2000:]
        return rec


def run_tp_workers(spec: Dict[str, Any], tp_size: int, timeout: float = 900.0) -> list:
    """The same scenario as `tp_size` rank processes (one reference `LLM` each, tp_info = (rank, tp_size)); on a box
    with fewer GPUs than ranks they share device 0 over the peer-to-peer communicator.  Returns the ranks' records."""
    import socket

    import torch

    ref = reference_root()
    assert ref is not None, "no importable reference (run oracle/build_ref.sh in the build container)"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    spec = dict(spec, tp_size=tp_size, port=port)
    with tempfile.TemporaryDirectory(prefix="msgl_refdrive_tp_") as td:
        spec_path, out_path = Path(td) / "spec.json", Path(td) / "out.pt"
        spec_path.write_text(json.dumps(spec))
        procs = []
        for r in range(tp_size):
            env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MSGL_REFDRIVE_RANK=str(r),
                       GLOO_SOCKET_IFNAME="lo", **{k: str(v) for k, v in spec.get("env", {}).items()})
            env.pop("MSGL_GEMM_TUNE", None)
            procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "refdrive_worker.py"), str(spec_path),
                                           str(out_path), str(ref)], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        logs, failed = [], False
        for p in procs:
            try:
                o, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
                failed = True
            logs.append(o)
            failed |= p.returncode != 0
        if failed or not all(Path(f"{out_path}.{r}").exists() for r in range(tp_size)):
            raise AssertionError("reference-driven tp workers failed:\n" + "\n=====\n".join(l[-4000:] for l in logs))
        return [torch.load(f"{out_path}.{r}", weights_only=False) for r in range(tp_size)]


def offline_bench_requests(n: int, seed: int = 0, max_out: Optional[int] = None):
    """First `n` requests of the reference's offline benchmark (benchmark/offline/bench.py:11-31: seed(0), 256
    prompts of randint(100,1024) ids in [0,10000], max_tokens randint(100,1024)), generated in the same call
    order so that request i is the benchmark's request i."""
    import random

    rnd = random.Random(seed)
    prompts = [[rnd.randint(0, 10000) for _ in range(rnd.randint(100, 1024))] for _ in range(256)]
    outs = [rnd.randint(100, 1024) for _ in range(256)]
    outs = [min(o, max_out) if max_out else o for o in outs]
    return prompts[:n], outs[:n]
