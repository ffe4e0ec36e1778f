"""Turns a rocprofv3 results .db (rocpd sqlite, what `rocprofv3 --kernel-trace --stats` writes on this
image) into a small text/CSV kernel summary to commit under profiles/."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    m = re.search(r"pgr::(\w+)(<[^>]*>)?", name)
    if m:
        return "pgr::" + m.group(1) + (m.group(2) or "")
    m = re.search(r"radix_sort_onesweep_(\w+?)<", name)
    if m:
        return "rocprim::radix_sort_onesweep_" + m.group(1) + ("#2" if "#2}" in name[-300:] else "")
    m = re.search(r"(at::native::\w+)", name)
    if m:
        return m.group(1)
    return name[:80]


def main(db_path, out_path=None, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats summary ({note})", "# durations in microseconds",
             f"{'kernel':58s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}"]
    for name, calls, total, avg, pct in rows:
        lines.append(f"{short(name):58s} {calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
