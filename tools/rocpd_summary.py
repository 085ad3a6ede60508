"""Summarise a rocprofv3 (rocpd sqlite) result database: per-kernel count / total / average
duration (the `--stats` view) and, for --pmc runs, per-kernel counter sums per dispatch.

    python tools/rocpd_summary.py gpurun_out/prof_bench/*/*_results.db [--top 25] [--csv out.csv]
"""
from __future__ import annotations

import argparse
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\[clone .*\]", "", name)
    name = re.sub(r"\(.*", "", name)
    return name.strip()[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--csv")
    ap.add_argument("--steps-by", help="kernel-name substring that occurs once per step (e.g. sample_kernel): "
                    "summarise only the window spanned by its last --last-steps occurrences, per step")
    ap.add_argument("--last-steps", type=int, default=10)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if "name" in cols else []
    per = 1
    if args.steps_by and rows:
        rows.sort(key=lambda r: r[1])
        marks = [r[2] for r in rows if args.steps_by in r[0]]
        if len(marks) > args.last_steps:
            lo, hi = marks[-args.last_steps - 1], marks[-1]
            rows = [r for r in rows if lo < r[1] <= hi]
            per = args.last_steps
            busy = sum(e - s for _, s, e in rows)
            print(f"# window: last {per} steps by '{args.steps_by}': wall {(hi - lo) / 1e3 / per:.1f} us/step, "
                  f"kernel busy {busy / 1e3 / per:.1f} us/step, {len(rows) / per:.1f} launches/step")
    stats = {}
    for name, s, e in rows:
        d = stats.setdefault(short(name), [0, 0])
        d[0] += 1
        d[1] += e - s
    total = sum(v[1] for v in stats.values()) or 1
    lines = ["kernel,calls,total_us,avg_us,percent"]
    for k, (n, t) in sorted(stats.items(), key=lambda kv: -kv[1][1])[: args.top]:
        lines.append(f"\"{k}\",{n / per:g},{t / 1e3 / per:.1f},{t / 1e3 / n:.2f},{100 * t / total:.2f}")
    print("\n".join(lines))
    # counters
    try:
        pm = cur.execute("select * from pmc_events limit 1").fetchall()
        pcols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
    except sqlite3.Error:
        pm, pcols = [], []
    if pm:
        print("\n# pmc_events columns:", pcols)
        try:
            q = ("select name, counter_name, count(*), sum(counter_value) from pmc_events "
                 "group by name, counter_name")
            agg = cur.execute(q).fetchall()
        except sqlite3.Error as e:
            print("pmc join failed:", e)
            agg = []
        print("kernel,counter,dispatches,sum,per_dispatch")
        for name, cname, n, v in sorted(agg, key=lambda r: -(r[3] or 0))[: args.top]:
            print(f"\"{short(name)}\",{cname},{n},{v:.0f},{(v or 0) / n:.1f}")
    if args.csv:
        open(args.csv, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
