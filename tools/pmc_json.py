"""Assemble profiles/rNN_pmc_attn_decode.json from the rocprofv3 passes over tools/profile_attn.py:
two --pmc passes (FETCH_SIZE, WRITE_SIZE; separate, as MI355X_MICROARCH.md prescribes) summarised by
tools/rocpd_summary.py, plus the --kernel-trace pass.  The FETCH_SIZE correction is CALIBRATED in the same
pass on the 1-GiB device copies profile_attn.py issues (gfx950 reports half of a wide streaming read).

    python tools/pmc_json.py FETCH.txt WRITE.txt KTRACE.txt ALGO_BYTES OUT.json [note]
"""
from __future__ import annotations

import json
import sys

GIB = 1 << 30


def per_dispatch(path, counter):
    out = {}
    for line in open(path):
        if f",{counter}," not in line:
            continue
        name, rest = line.rsplit(f",{counter},", 1)
        n, total, per = rest.strip().split(",")
        out[name.strip('"')] = (int(n), float(total), float(per))
    return out


def avg_us(path):
    out = {}
    for line in open(path):
        if line.startswith('"') and line.count(",") >= 4 and "FETCH_SIZE" not in line and "WRITE_SIZE" not in line:
            name, calls, total, avg, pct = line.rsplit(",", 4)
            out[name.strip('"')] = float(avg)
    return out


def pick(d, key, default=None):
    for k, v in d.items():
        if key in k:
            return v
    if default is not None:
        return default
    raise KeyError(key)


def pick_attn(d):
    """(kernel name, value) of the partial-attention kernel that ran: matrix-core or streaming."""
    for key in ("attn_decode_mfma_kernel", "attn_decode_kernel"):
        for k, v in d.items():
            if key in k:
                return k, v
    raise KeyError("attn_decode*_kernel")


def main():
    fetch_txt, write_txt, kt_txt, algo, out = sys.argv[1:6]
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    algo = int(algo)
    f, w, t = per_dispatch(fetch_txt, "FETCH_SIZE"), per_dispatch(write_txt, "WRITE_SIZE"), avg_us(kt_txt)
    # calibration: each 1-GiB copy reads 1 GiB and writes 1 GiB; counters are in KB
    cf, cw = pick(f, "copyBuffer"), pick(w, "copyBuffer")
    copies_f = cf[1] / cf[0] if cf[0] else 0.0
    # rocclr may split one copy into several dispatches: use totals over the 3 copies of profile_attn.py
    fetch_ratio = cf[1] * 1024 / (3 * GIB)
    write_ratio = cw[1] * 1024 / (3 * GIB)
    corr = 1.0 / fetch_ratio
    kname, fa = pick_attn(f)
    # round 4: at full batches the split-KV merge runs INSIDE the attention kernel (no merge dispatch: zeros)
    fa, fm = fa[2], pick(f, "attn_decode_merge", (0, 0, 0.0))[2]
    wa, wm = pick_attn(w)[1][2], pick(w, "attn_decode_merge", (0, 0, 0.0))[2]
    hbm = ((fa + fm) * corr + (wa + wm) / write_ratio) * 1024
    ta, tm = pick_attn(t)[1], pick(t, "attn_decode_merge", 0.0)
    res = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) and --kernel-trace --stats "
                  "-- python tools/profile_attn.py, MI355X" + (", " + note if note else ""),
        "workload": f"{kname.split('(')[0][:60]} + merge (in-kernel when no merge dispatch is listed), Qwen3-14B TP1 shape, B=256, bench contexts + 46 "
                    "(= bench.py's roofline launch at default steps), page_size 256",
        "launch_shape": "qwen3-14b tp1 B256 page256 bench_contexts",
        "algorithmic_bytes_per_launch": algo,
        "fetch_size_kb": {"attn": fa, "merge": fm},
        "write_size_kb": {"attn": wa, "merge": wm},
        "fetch_correction": corr,
        "calibration": f"same pass: three 1-GiB device copies read FETCH_SIZE = {cf[1]:.0f} KB "
                       f"({fetch_ratio:.3f} x true) and WRITE_SIZE = {cw[1]:.0f} KB ({write_ratio:.3f} x true)",
        "hbm_bytes_per_launch": hbm,
        "ratio_to_algorithmic": hbm / algo,
        "kernel_trace_avg_us": {"attn": ta, "merge": tm},
        "achieved_GBps_kernel_trace": algo / (ta + tm) / 1e3,
    }
    open(out, "w").write(json.dumps(res, indent=1))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
