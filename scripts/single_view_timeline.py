"""Kernel timeline of single-view render() calls from a rocprofv3 --kernel-trace rocpd .db: per kernel the average duration
and the average gap before it (idle time on the GPU between the previous kernel's end and this one's start), over the calls
after the warm-up.  Usage: python3 scripts/single_view_timeline.py <db> [n_warm_calls]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = ([x for x in tables if x == "kernels"] or [x for x in tables if "kernel_dispatch" in x])[0]
cols = [d[1] for d in db.execute(f"pragma table_info({t})")]
start = [c for c in cols if c in ("start", "start_timestamp")][0]
end = [c for c in cols if c in ("end", "end_timestamp")][0]
namec = [c for c in cols if c in ("name", "kernel_name")][0]
rows = list(db.execute(f"select {start},{end},{namec} from {t} order by {start}"))


def short(name):
    m = re.search(r"pgr::(\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:200]


# a call = the kernels from one preprocess_batch_kernel to the next, cut at the kernel that packs its camera
cutter = "batch_header" if any("batch_header_kernel" in n for _, _, n in rows) else "pack_camera"
calls, cur = [], None
for s, e, n in rows:
    k = short(n)
    if k.startswith(cutter):
        if cur:
            calls.append(cur)
        cur = []
    if cur is not None:
        cur.append((s, e, k))
if cur:
    calls.append(cur)
calls = calls[skip:-1]
print(f"{len(calls)} calls analysed")
dur, gap, cnt, order = defaultdict(float), defaultdict(float), defaultdict(int), []
span = busy = 0.0
for c in calls:
    prev_end = None
    seen = defaultdict(int)
    for s, e, k in c:
        seen[k] += 1
        key = f"{k}#{seen[k]}" if seen[k] > 1 else k
        if key not in dur:
            order.append(key)
        dur[key] += e - s
        cnt[key] += 1
        if prev_end is not None:
            gap[key] += max(0, s - prev_end)
        prev_end = max(prev_end or e, e)
    span += c[-1][1] - c[0][0]
    busy += sum(e - s for s, e, _ in c)
n = len(calls)
print(f"{'kernel':50s} {'avg us':>9s} {'gap before us':>14s}")
for k in order:
    print(f"{k[:50]:50s} {dur[k] / cnt[k] / 1e3:9.2f} {gap[k] / cnt[k] / 1e3:14.2f}")
for k in order:
    if len(k) > 50:
        print("#", k)
print(f"per call: first kernel start -> last kernel end {span / n / 1e3:.1f} us, kernels busy {busy / n / 1e3:.1f} us, "
      f"idle between kernels {(span - busy) / n / 1e3:.1f} us")
gaps_between = [(b[0][0] - a[-1][1]) / 1e3 for a, b in zip(calls, calls[1:])]
if gaps_between:
    print(f"between calls (last kernel end -> next call's first kernel): {sum(gaps_between) / len(gaps_between):.1f} us")
