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

# (name, same-shaped weights rotated while timing, K[, {"silu_interleaved": True}]): the flag marks a gate_up group whose
# rows are in ops.interleave_gate_up order, i.e. whose consumer is ops.linear_silu (fused epilogue candidates are timed)
Group = Tuple


def tune_prefill_gemms(groups: Sequence[Group], prefill_tokens: Sequence[int], dtype: torch.dtype, device: torch.device,
                       log: Optional[Callable[[str], None]] = None) -> List[dict]:
    """Library solution search at PREFILL chunk sizes.  A chunked-prefill forward under load carries exactly
    `max_extend_tokens` tokens (P/scheduler/prefill.py:65-90 fills the budget), the reference's default being 8192
    (P/scheduler/config.py:16): at that M the library's heuristic pick is 1.4x off its best solution for o_proj and down_proj
    (tools/prefill_gemm_probe.py; at 16384 the heuristic already is the best of 230).  Library only -- the hand-written
    kernels stop at M = 256 -- the 16 solutions the library ranks first, no split-K; the plan is keyed by the exact M, every
    other chunk size keeps the heuristic.  The LM head is skipped (a prefill forward projects one row per finished request)."""
    report: List[dict] = []
    for M in sorted(set(int(m) for m in prefill_tokens if int(m) > ops.M256_MAX_M)):
        for grp in groups:
            name, ws, k = grp[0], list(grp[1]), grp[2]
            if name == "lm_head":
                continue
            x = torch.randn((M, k), device=device, dtype=torch.float32).to(dtype)
            r = ops.gemm_tune(x, ws, max_candidates=-16, iters=3, split_k=False)
            r.update(name=name, prefill=True)
            report.append(r)
            if log is not None:
                log(f"[gemm_tune] prefill M={M} {name}: {r['default_us']:.1f} -> {r['best_us']:.1f} us ({r['tried']} candidates)")
            del x
    return report


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
        for grp in groups:
            name, ws, k = grp[0], grp[1], grp[2]
            flags = grp[3] if len(grp) > 3 else {}
            ws = list(ws)
            x = torch.randn((bs, k), device=device, dtype=torch.float32).to(dtype)
            r = ops.gemm_tune(x, ws, max_candidates=cands[bs], iters=8)
            r["name"] = name
            if bs <= ops.SKINNY_MAX_M:  # hand-written weight-streaming kernel vs the library's best
                # flags["fold"]: the row kernel the caller folds into this projection where the row-streaming kernel is planned
                kind = flags.get("fold", "")
                fold = {"norm": ops.ROWSTREAM_ADD_NORM, "act": ops.ROWSTREAM_SILU, "act_interleaved": ops.ROWSTREAM_SILU_INTERLEAVED}.get(kind)
                if bs > (ops.ROWSTREAM_FOLD_ACT_MAX_M if kind.startswith("act") else ops.ROWSTREAM_FOLD_NORM_MAX_M):
                    fold = None
                sk = ops.skinny_tune(x, ws, r["best_us"], fold_mode=fold)
                r.update(skinny_us=sk["skinny_us"], skinny_slices=sk["slices"], skinny_row_tiles=sk["row_tiles"],
                         skinny_used=sk["used"], fold_credit_us=sk.get("fold_credit_us", 0.0))
                if sk["used"]:
                    r["library_best_us"], r["best_us"] = r["best_us"], sk["skinny_us"]
                    r["kernel"] = (f"msgl::rowstream{'4' if sk['slices'] else ''}_gemm_kernel[{sk['row_tiles']} loads in flight per lane]" if sk["slices"] <= 0 else
                                   f"msgl::skinny_gemm_kernel[slices {sk['slices']}, row tiles {sk['row_tiles']}]")
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
                    r["kernel"] = ("msgl::%s[grid %d, whole tiles %d, k-slices %d]"
                                   % ((("m256_gemm_kernel", "g3_gemm_kernel")[mr["plan"][3]],) + tuple(mr["plan"][:3])))
            if ops.ro_supported(bs, r["N"], r["K"]) and os.environ.get("MSGL_DISABLE_RO") != "1":
                # row-owner generation (gemm_ro.hip): balanced 16-row-unit tiles x k-slices, against the best so far
                rr = ops.ro_tune(x, ws, r["best_us"])
                r.update(ro_us=rr["ro_us"], ro_plan=rr["plan"], ro_used=rr["used"], ro_all=rr.get("all"))
                if rr["used"]:
                    r.setdefault("library_best_us", r["best_us"])
                    r["best_us"] = rr["ro_us"]
                    r["skinny_used"] = True
                    r["kernel"] = "msgl::ro_gemm_kernel[tiles %d, k-slices %d]" % tuple(rr["plan"])
            if flags.get("silu_interleaved") and (ops.m256_supported(bs, r["N"], r["K"]) or bs <= ops.SKINNY_MAX_M):
                fr = ops.fused_silu_tune(x, ws)  # projection + activation as one launch vs the two just planned
                r.update(silu_unfused_us=fr["unfused_us"], silu_fused_us=fr["fused_us"], silu_fused_plan=fr["plan"],
                         silu_fused_used=fr["used"], silu_fused_all=fr.get("all"))
                if fr["used"]:
                    r.setdefault("library_best_us", r["best_us"])
                    r["best_us"] = fr["fused_us"]  # projection AND activation
                    r["skinny_used"] = True
                    if fr.get("kind") == "skinny":
                        r["kernel"] = "msgl::skinny_gemm_kernel<silu>[slices %d, row tiles %d]" % tuple(fr["plan"])
                    else:
                        r["kernel"] = "msgl::g3_gemm_kernel<silu>[grid %d, whole tiles %d, k-slices %d]" % tuple(fr["plan"])
            if flags.get("silu_interleaved") and ops.ro_supported(bs, r["N"], r["K"]) and os.environ.get("MSGL_DISABLE_RO") != "1":
                # the row-owner launch with SiLU.mul in its epilogue against the best (projection, activation) pair or fused
                # launch found so far
                pair = r["best_us"] if (r.get("silu_fused_used")) else (r.get("silu_unfused_us") or ops.silu_pair_us(x, ws))
                rs = ops.ro_silu_tune(x, ws, pair)
                r.update(ro_silu_us=rs["fused_us"], ro_silu_plan=rs["plan"], ro_silu_used=rs["used"], ro_silu_all=rs.get("all"),
                         ro_silu_against_us=pair)
                if rs["used"]:
                    r.setdefault("library_best_us", r["best_us"])
                    r["best_us"] = rs["fused_us"]  # projection AND activation
                    r["skinny_used"] = True
                    r["kernel"] = "msgl::ro_gemm_kernel<silu>[tiles %d]" % rs["plan"][0]
            ops.register_candidates(name, x, ws[0], r)
            report.append(r)
            if log is not None:
                log(f"[gemm_tune] bs={bs} {name}: {r['default_us']:.1f} -> {r['best_us']:.1f} us "
                    f"({r['tried']} candidates) {r['kernel'][:100]}")
    return report
