"""Per-kernel sums of PMC counters from a rocprofv3 --pmc results .db (rocpd sqlite)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"pgr::(\w+)(<[^>]*>)?", name)
    return ("pgr::" + m.group(1) + (m.group(2) or "")) if m else name[:60]


db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ix else [c for c in cols if "kernel" in c and "name" in c][0]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for r in rows:
    k = short(r[ix[name_col]])
    acc[k][r[ix["counter_name"]]] += float(r[ix["value"]])
    cnt[k].add(r[ix["dispatch_id"]])
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{k}  dispatches={len(cnt[k])}")
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:18.0f}   per-dispatch {v / max(1, len(cnt[k])):16.0f}")
