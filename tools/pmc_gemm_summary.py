"""Attribute the dispatches of a `rocprofv3 --pmc ... -- python tools/pmc_gemm.py` run to the operations the script lists
in its manifest (by launch order: the measured region is the tail of the run) and print counter sums per operation and
launch.

    python tools/pmc_gemm_summary.py <results.db> <manifest.json> [--json out.json]
"""
import argparse
import json
import re
import sqlite3
from collections import defaultdict


def kind(n):
    if "reduce_kernel" in n:
        return "reduce"
    if "g3_gemm_kernel" in n or "m256_gemm_kernel" in n or "wstream_gemm" in n or "ro_gemm_kernel" in n or "Cijk_" in n:
        return "gemm"
    if "qk_norm_rope_store" in n:
        return "qk"
    if "rmsnorm" in n:
        return "norm"
    if "silu_mul" in n:
        return "silu"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("manifest")
    ap.add_argument("--json")
    a = ap.parse_args()
    man = json.load(open(a.manifest))
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute("select dispatch_id, name, counter_name, counter_value from pmc_events order by dispatch_id").fetchall()
    disp = {}
    for did, name, cname, val in rows:
        d = disp.setdefault(did, dict(name=name, counters={}))
        d["counters"][cname] = d["counters"].get(cname, 0) + (val or 0)
    order = [disp[k] for k in sorted(disp)]
    counters = sorted({c for d in order for c in d["counters"]})
    out = {"counters": counters, "calibration": {}}
    for c in counters:
        big = sorted(d["counters"].get(c, 0) for d in order if "copyBuffer" in d["name"])[-3:]
        out["calibration"][c] = big
        print(f"# calibration {c}: the three 1-GiB copyBuffer dispatches {big}")
    rel = [d for d in order if kind(d["name"])]
    exp = man["expected"]
    tail = rel[-len(exp):]
    bad = [(i, e, d["name"][:50]) for i, (e, d) in enumerate(zip(exp, tail)) if kind(d["name"]) != e[1]]
    if len(tail) != len(exp) or bad:
        print(f"# WARNING: launch sequence does not match the manifest ({len(tail)} vs {len(exp)}; first mismatches {bad[:3]})")
    agg, names, cnt = defaultdict(lambda: defaultdict(float)), {}, defaultdict(int)
    for (label, _k), d in zip(exp, tail):
        for c, v in d["counters"].items():
            agg[label][c] += v
        names[label] = d["name"][:110]
        cnt[label] += 1
    print(f"# {len(order)} dispatches in the run, {len(exp)} in the measured region; counters per LAUNCH")
    print("operation,kernel," + ",".join(counters))
    out["per_launch"] = {}
    seen = []
    for label, _k in exp:
        if label in seen:
            continue
        seen.append(label)
        vals = {c: agg[label][c] / cnt[label] for c in counters}
        out["per_launch"][label] = dict(kernel=names[label], launches=cnt[label], **vals)
        print(f"{label},\"{names[label][:60]}\"," + ",".join(f"{vals[c]:.0f}" for c in counters))
    out["manifest"] = {k: v for k, v in man.items() if k != "expected"}
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
