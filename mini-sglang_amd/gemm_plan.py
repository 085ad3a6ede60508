"""Per-shape kernel choice for the projection GEMMs of a decode step.

`F.linear` of the reference (P/layers/linear.py:32,103,124, P/layers/embedding.py:98) is one library call
with the library's heuristic; here each (M, N, K) a captured decode graph will launch is timed once, before
capture, over: the library's solutions (incl. split-K), and the hand-written weight-streaming kernels where
they apply.  `ops.linear` then dispatches on the recorded plan.  Used by this repository's engine
(`DenseDecoder.tune_gemms`) and by the plugin for the reference's engine (`minisgl_plugin.install`), so the
benchmarked path and the drop-in path choose kernels the same way.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import ops

Group = Tuple[str, Sequence[torch.Tensor], int]  # (name, same-shaped weights rotated while timing, K)


def tune_projection_gemms(groups: Sequence[Group], batch_sizes: Sequence[int], mode: str, dtype: torch.dtype,
                          device: torch.device, log: Optional[Callable[[str], None]] = None) -> List[dict]:
    """mode: "off", "heuristic" (library's top 16 + hand-written kernels), "full" (every library solution at the
    largest batch size, top 16 elsewhere).  Weights of different layers are rotated so candidates are timed from
    HBM, not from the Infinity Cache.  Synchronises: call before graph capture."""
    if mode == "off":  # plans are process state: "off" means the library's heuristic, not an earlier engine's search
        ops.reset_gemm_plans()
        return []
    if not batch_sizes:
        return []
    biggest = max(batch_sizes)
    cands = {bs: ({"heuristic": -16, "full": 0}[mode] if bs == biggest else -16) for bs in batch_sizes}
    report: List[dict] = []
    for bs in batch_sizes:
        for name, ws, k in groups:
            ws = list(ws)
            x = torch.randn((bs, k), device=device, dtype=torch.float32).to(dtype)
            r = ops.gemm_tune(x, ws, max_candidates=cands[bs], iters=8)
            r["name"] = name
            if bs <= ops.SKINNY_MAX_M:  # hand-written weight-streaming kernel vs the library's best
                sk = ops.skinny_tune(x, ws, r["best_us"])
                r.update(skinny_us=sk["skinny_us"], skinny_slices=sk["slices"], skinny_row_tiles=sk["row_tiles"],
                         skinny_used=sk["used"])
                if sk["used"]:
                    r["library_best_us"], r["best_us"] = r["best_us"], sk["skinny_us"]
                    r["kernel"] = f"msgl::skinny_gemm_kernel[slices {sk['slices']}, row tiles {sk['row_tiles']}]"
            if 32 < bs <= ops.WSTREAM_MAX_M and name != "lm_head":  # LDS-shared weight-streaming kernel
                wsr = ops.wstream_tune(x, ws, r["best_us"])
                r.update(wstream_us=wsr["wstream_us"], wstream_row_tiles=wsr["row_tiles"],
                         wstream_k_splits=wsr["k_splits"], wstream_used=wsr["used"])
                if wsr["used"]:
                    r.setdefault("library_best_us", r["best_us"])
                    r["best_us"] = wsr["wstream_us"]
                    r["skinny_used"] = True  # reported as hand-written by bench.py
                    r["kernel"] = (f"msgl::wstream_gemm_kernel[row tiles {wsr['row_tiles']}, "
                                   f"k splits {wsr['k_splits']}]")
            if ops.m256_supported(bs, r["N"], r["K"]) and os.environ.get("MSGL_DISABLE_M256") != "1":  # one workgroup per CU, LDS-DMA ring (gemm_m256.hip)
                mr = ops.m256_tune(x, ws, r["best_us"])
                r.update(m256_us=mr["m256_us"], m256_plan=mr["plan"], m256_used=mr["used"], m256_all=mr.get("all"))
                if mr["used"]:
                    r.setdefault("library_best_us", r["best_us"])
                    r["best_us"] = mr["m256_us"]
                    r["skinny_used"] = True
                    r["kernel"] = "msgl::m256_gemm_kernel[grid %d, whole tiles %d, k-slices %d]" % tuple(mr["plan"])
            report.append(r)
            if log is not None:
                log(f"[gemm_tune] bs={bs} {name}: {r['default_us']:.1f} -> {r['best_us']:.1f} us "
                    f"({r['tried']} candidates) {r['kernel'][:100]}")
    return report
