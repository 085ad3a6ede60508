"""Error-distribution report for logit parity (north_star: "bf16 logits within 1e-3").

One bf16 ulp at magnitude |x| is 2^(floor(log2|x|) - 7): 1e-3 is less than one ulp for every |logit| > 0.25, so the
flat figure cannot hold for a bf16 pipeline compared with an fp32-accumulating CPU restatement whose GEMMs sum in a
different order.  Tests therefore report the whole distribution (absolute and in ulp of the oracle value), assert a
bound set from the measured distribution plus a margin, and state where 1e-3 is and is not met.
"""
from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Any, Dict, List

import torch

# What the session compared, for the pytest terminal summary and gpurun_out/parity_summary.json (tests/conftest.py writes both;
# bench.py surfaces the committed copy under profiles/ as `parity`, kind "committed").  Two kinds of record:
#   bit_identical: label -> number of forwards whose logits matched BIT FOR BIT (reference-driven replays, rank pairs)
#   logits:        label -> error distribution of ours, of the torch-bf16 floor on the GPU's library GEMMs, and of the
#                  CPU floor (fp32-accumulating matmul on bf16-rounded inputs: independent of the device's libraries)
RECORDS: List[Dict[str, Any]] = []


def record(kind: str, label: str, **fields: Any) -> None:
    RECORDS.append(dict(kind=kind, label=label, **fields))


def _brief(st: Dict[str, float]) -> Dict[str, float]:
    return {k: st[k] for k in ("n", "ref_std", "max_abs", "p99_abs", "mean_abs", "frac_gt_1e3")} if st else None


def summary_lines() -> List[str]:
    out = []
    bit = [r for r in RECORDS if r["kind"] == "bit_identical"]
    if bit:
        out.append(f"bit-identical logits: {sum(r['forwards'] for r in bit)} forwards in {len(bit)} scenarios "
                   f"(reference-driven through the plugin vs the repository's engine, rank vs rank)")
        out += [f"    {r['forwards']:4d}  {r['label']}" for r in bit]
    for r in (r for r in RECORDS if r["kind"] == "logits"):
        f = lambda st: "n/a" if not st else (f"max {st['max_abs']:.2e} p99 {st['p99_abs']:.2e} mean {st['mean_abs']:.2e} "  # noqa: E731
                                             f"frac>1e-3 {st['frac_gt_1e3']:.3f}")
        out.append(f"logits vs the fp32 oracle [{r['label']}] (ref std {r['ours']['ref_std']:.2f}, n {r['ours']['n']})")
        out.append(f"    ours                       {f(r['ours'])}  argmax {r.get('ours_agree')}/{r.get('total')}")
        out.append(f"    torch-bf16 (GPU library)   {f(r.get('floor_gpu'))}  argmax {r.get('floor_gpu_agree')}/{r.get('total')}")
        out.append(f"    bf16-rounded fp32-acc (CPU) {f(r.get('floor_cpu'))}  argmax {r.get('floor_cpu_agree')}/{r.get('total')}")
    return out


def write_summary(path: Path) -> None:
    if not RECORDS:
        return
    bit = [r for r in RECORDS if r["kind"] == "bit_identical"]
    path.parent.mkdir(parents=True, exist_ok=True)
    path.write_text(json.dumps(dict(
        bit_identical_forwards=sum(r["forwards"] for r in bit), bit_identical_scenarios=[dict(label=r["label"], forwards=r["forwards"]) for r in bit],
        logits=[dict(label=r["label"], ours=_brief(r["ours"]), floor_gpu_library=_brief(r.get("floor_gpu")),
                     floor_cpu_fp32acc=_brief(r.get("floor_cpu")), argmax=dict(ours=r.get("ours_agree"), floor_gpu=r.get("floor_gpu_agree"),
                                                                               floor_cpu=r.get("floor_cpu_agree"), total=r.get("total")))
                for r in RECORDS if r["kind"] == "logits"],
        note="north_star's 1e-3 is below one bf16 ulp for |logit| > 0.25; frac_gt_1e3 shows where each bf16 pipeline sits"), indent=1))


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """Spacing of bf16 numbers at |x| (8 significand bits); floored at the spacing of 2^-6 so that near-zero
    references do not blow the ratio up."""
    mag = x.abs().float().clamp_min(2.0 ** -6)
    return torch.exp2(torch.floor(torch.log2(mag)) - 7)


def logit_error_stats(got: torch.Tensor, ref: torch.Tensor) -> Dict[str, float]:
    got, ref = got.float().flatten(), ref.float().flatten()
    err = (got - ref).abs()
    ulp = err / bf16_ulp(ref)
    k99 = max(int(0.99 * err.numel()), 1)
    return dict(
        n=err.numel(), ref_std=float(ref.std()), ref_absmax=float(ref.abs().max()),
        max_abs=float(err.max()), p99_abs=float(err.kthvalue(k99).values), mean_abs=float(err.mean()),
        frac_gt_1e3=float((err > 1e-3).float().mean()),
        max_ulp=float(ulp.max()), p99_ulp=float(ulp.kthvalue(k99).values), mean_ulp=float(ulp.mean()),
    )


def merge_stats(a: Dict[str, float], b: Dict[str, float]) -> Dict[str, float]:
    """Worst case over steps for the max / p99 figures, element-weighted means elsewhere."""
    if not a:
        return dict(b)
    n = a["n"] + b["n"]
    out = dict(n=n)
    for k in ("max_abs", "p99_abs", "max_ulp", "p99_ulp", "ref_absmax", "ref_std"):
        out[k] = max(a[k], b[k])
    for k in ("mean_abs", "frac_gt_1e3", "mean_ulp"):
        out[k] = (a[k] * a["n"] + b[k] * b["n"]) / n
    return out


def fmt(stats: Dict[str, float]) -> str:
    return (f"n={stats['n']} ref_std={stats['ref_std']:.3f} |err| max={stats['max_abs']:.2e} p99={stats['p99_abs']:.2e} "
            f"mean={stats['mean_abs']:.2e} frac>1e-3={stats['frac_gt_1e3']:.3f} | ulp(bf16 of ref) max="
            f"{stats['max_ulp']:.2f} p99={stats['p99_ulp']:.2f} mean={stats['mean_ulp']:.3f}")


def floor_report(label: str, ours: Dict[str, float], floor: Dict[str, float], ours_agree: int, floor_agree: int, total: int) -> str:
    return (f"[{label}] vs the fp32 oracle, same teacher-forced batches:\n"
            f"    ours        {fmt(ours)}; argmax agreement {ours_agree}/{total}\n"
            f"    torch-bf16  {fmt(floor)}; argmax agreement {floor_agree}/{total}\n"
            f"    ratio ours / torch-bf16: max {ours['max_abs'] / max(floor['max_abs'], 1e-12):.2f}  p99 "
            f"{ours['p99_abs'] / max(floor['p99_abs'], 1e-12):.2f}  mean {ours['mean_abs'] / max(floor['mean_abs'], 1e-12):.2f}")


def assert_not_above_bf16_floor(label: str, ours: Dict[str, float], floor: Dict[str, float], ours_agree: int, floor_agree: int,
                                total: int, floor_cpu: Dict[str, float] = None, floor_cpu_agree: int = None) -> None:
    """|ours - oracle| must be statistically no larger than |independent torch-bf16 forward - oracle| (oracle/torch_bf16.py):
    mean and p99 within 25 % of the floor's, the maximum (one sample of the tail) within 50 %, and the argmax of ours
    agrees with the oracle's at least as often as the floor's does (minus 1 % of the rows for ties broken the other way)."""
    print(floor_report(label, ours, floor, ours_agree, floor_agree, total))
    record("logits", label, ours=ours, floor_gpu=floor, floor_cpu=floor_cpu, ours_agree=ours_agree, floor_gpu_agree=floor_agree,
           floor_cpu_agree=floor_cpu_agree, total=total)
    if floor_cpu:
        # the floor that owes nothing to the device's GEMM libraries: exact-input fp32 accumulation, one rounding per op
        print(f"    cpu-floor   {fmt(floor_cpu)}; argmax agreement {floor_cpu_agree}/{total}")
        assert ours["mean_abs"] <= 1.35 * floor_cpu["mean_abs"] + 1e-7, label
        assert ours["p99_abs"] <= 1.35 * floor_cpu["p99_abs"] + 1e-7, label
    assert ours["mean_abs"] <= 1.25 * floor["mean_abs"] + 1e-7, label
    assert ours["p99_abs"] <= 1.25 * floor["p99_abs"] + 1e-7, label
    assert ours["max_abs"] <= 1.5 * floor["max_abs"] + 1e-7, label
    assert ours_agree >= floor_agree - max(2, total // 100), label
