"""Per-kernel averages of the counters of a `rocprofv3 --pmc ... -- python tools/pmc_prefill.py` run."""
import sqlite3
import sys
from collections import defaultdict

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select dispatch_id, name, counter_name, counter_value from pmc_events order by dispatch_id").fetchall()
disp = {}
for did, name, cname, val in rows:
    d = disp.setdefault(did, dict(name=name, counters=defaultdict(float)))
    d["counters"][cname] += val or 0
agg, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for d in disp.values():
    if "attn_prefill" not in d["name"]:
        continue
    key = d["name"][:90]
    cnt[key] += 1
    for c, v in d["counters"].items():
        agg[key][c] += v
for k in agg:
    print(k, f"({cnt[k]} launches)")
    for c in sorted(agg[k]):
        print(f"    {c:34s} {agg[k][c] / cnt[k]:16.0f}")
