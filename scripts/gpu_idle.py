"""GPU idle share of the pipelined bench from a rocprofv3 --kernel-trace rocpd .db: union of the kernel intervals
(all streams) over the span of the last `tail` fraction of the trace (the timed steps), and the largest gaps."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tables if "kernel_dispatch" in t]
cand = [t for t in tables if t == "kernels"] or kd
t = cand[0]
cols = [d[1] for d in db.execute(f"pragma table_info({t})")]
start = [c for c in cols if c in ("start", "start_timestamp")][0]
end = [c for c in cols if c in ("end", "end_timestamp")][0]
namec = [c for c in cols if c in ("name", "kernel_name")]
q = f"select {start},{end}" + (f",{namec[0]}" if namec else ",''") + f" from {t} order by {start}"
rows = list(db.execute(q))
print("table", t, "rows", len(rows))
# the timed steps: the longest run of fused frame compositor launches (composite_quarter_kernel<false, true>) whose
# successive starts are less than 12 ms apart
comp = [r for r in rows if "composite_quarter_kernel" in r[2] and "true>" in r[2].split("composite_quarter_kernel")[1][:14]]
best, run = [], []
for r in comp:
    if run and r[0] - run[-1][0] > 12e6:
        if len(run) > len(best):
            best = run
        run = []
    run.append(r)
if len(run) > len(best):
    best = run
print("steady run of", len(best), "compositor launches")
t0, t1 = best[1][0], best[-2][1]
rows = [r for r in rows if r[0] >= t0 and r[1] <= t1]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e, gaps = 0, rows[0][0], rows[0][1], []
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(1 - busy / span):.2%}  gaps {len(gaps)}")
gaps.sort(reverse=True)
for g, n in gaps[:12]:
    print(f"   gap {g / 1e3:8.1f} us before {n[:70]}")

if len(sys.argv) > 2 and sys.argv[2] == "detail":
    print("columns:", cols)
    extra = [c for c in cols if c in ("stream_id", "queue_id", "stream", "queue", "tid")]
    q = f"select {start},{end},{namec[0]}," + ",".join(extra) + f" from {t} where {start} >= {t0} and {end} <= {t1} order by {start}"
    rr = list(db.execute(q))
    # the events around the three largest gaps
    ends = sorted(rr, key=lambda r: r[1])
    import bisect
    cur_e, shown = rr[0][1], 0
    for i, r in enumerate(rr[1:], 1):
        if r[0] > cur_e + 200e3 and shown < 3:
            shown += 1
            print(f"--- gap {(r[0] - cur_e) / 1e3:.0f} us; last 6 kernels to END before it, then the next 6 to START ({extra})")
            before = [x for x in ends if x[1] <= cur_e][-6:]
            for x in before:
                print(f"   end {(x[1] - cur_e) / 1e3:9.1f} us  dur {(x[1] - x[0]) / 1e3:8.1f}  {x[2][:60]:60s} {x[3:]}")
            for x in rr[i:i + 6]:
                print(f"   start +{(x[0] - cur_e) / 1e3:8.1f} us  dur {(x[1] - x[0]) / 1e3:8.1f}  {x[2][:60]:60s} {x[3:]}")
        cur_e = max(cur_e, r[1])

if len(sys.argv) > 2 and sys.argv[2] == "window":
    q = f"select {start},{end},{namec[0]} from {t} where {start} >= {t0} and {end} <= {t1} order by {start}"
    rr = list(db.execute(q))
    cur_e, shown = rr[0][1], 0
    for i, r in enumerate(rr[1:], 1):
        if r[0] > cur_e + 200e3 and shown < 2 and i > 200:
            shown += 1
            g0 = cur_e
            print(f"--- gap {(r[0] - cur_e) / 1e3:.0f} us: every kernel overlapping [-7 ms, +7 ms] around its start")
            for x in rr:
                if x[1] > g0 - 7e6 and x[0] < g0 + 7e6 and (x[1] - x[0]) > 50e3:
                    print(f"   {(x[0] - g0) / 1e3:9.1f} .. {(x[1] - g0) / 1e3:9.1f}  {x[2][:75]}")
        cur_e = max(cur_e, r[1])
