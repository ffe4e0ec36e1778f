"""Per-dispatch durations of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace rocpd .db:
   python scripts/kernel_calls.py <db> <pattern>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
t = ([x for x in tables if x == "kernels"] or [x for x in tables if "kernel_dispatch" in x])[0]
cols = [r[1] for r in db.execute(f"pragma table_info({t})")]
start = [c for c in cols if c in ("start", "start_timestamp")][0]
end = [c for c in cols if c in ("end", "end_timestamp")][0]
namec = [c for c in cols if c in ("name", "kernel_name")][0]
rows = [(s, e, n) for s, e, n in db.execute(f"select {start},{end},{namec} from {t} order by {start}") if sys.argv[2] in n]
print(len(rows), "dispatches; durations in us:")
print(" ".join(f"{(e - s) / 1e3:.0f}" for s, e, _ in rows))
