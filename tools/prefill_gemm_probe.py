"""Library solution search at PREFILL chunk sizes (M = 16384 / 8192): how far the heuristic pick is from the best solution
for each projection of the model.  python tools/prefill_gemm_probe.py [--model qwen3-14b] [--ms 16384 8192]"""
import argparse, json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mini_sglang_amd import ops
from mini_sglang_amd.model import PRESETS

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="qwen3-14b")
ap.add_argument("--ms", type=int, nargs="*", default=[16384, 8192])
ap.add_argument("--candidates", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/prefill_gemm_probe.json")
a = ap.parse_args()
dev = torch.device("cuda:0")
m, D = PRESETS[a.model], PRESETS[a.model].head_dim
shapes = {"qkv": ((m.num_qo_heads + 2 * m.num_kv_heads) * D, m.hidden_size), "o": (m.hidden_size, m.num_qo_heads * D),
          "gate_up": (2 * m.intermediate_size, m.hidden_size), "down": (m.hidden_size, m.intermediate_size)}
rows = []
for M in a.ms:
    tot_d = tot_b = 0.0
    for name, (N, K) in shapes.items():
        ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(2)]
        x = torch.randn((M, K), device=dev).to(torch.bfloat16)
        r = ops.gemm_tune(x, ws, max_candidates=a.candidates, iters=3, split_k=False)
        fl = 2.0 * M * N * K
        rows.append(dict(M=M, name=name, N=N, K=K, default_us=r["default_us"], best_us=r["best_us"], tried=r["tried"],
                         default_tflops=fl / r["default_us"] / 1e6, best_tflops=fl / r["best_us"] / 1e6, kernel=r["kernel"][:80]))
        tot_d += r["default_us"]; tot_b += r["best_us"]
        print(f"M={M} {name:8s} default {r['default_us']:8.1f} us ({fl / r['default_us'] / 1e6:6.0f} TF)  best {r['best_us']:8.1f} us ({fl / r['best_us'] / 1e6:6.0f} TF) of {r['tried']}  {r['kernel'][:70]}", flush=True)
        del ws, x
        torch.cuda.empty_cache()
    print(f"M={M}: per layer default {tot_d:.0f} us -> best {tot_b:.0f} us ({tot_d / tot_b:.3f}x)")
ops.reset_gemm_plans()
Path(a.out).parent.mkdir(parents=True, exist_ok=True)
Path(a.out).write_text(json.dumps(rows, indent=1))
