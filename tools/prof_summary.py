"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db): per-kernel time, calls, share.
Usage: python tools/prof_summary.py <results.db> [--skip-first-frac F] > profiles/xxx.txt"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("cl::", "")
    name = re.sub(r"at::native::", "at::", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    if not rows:
        print("no kernels", cols); return
    t0, t1 = rows[0][1], rows[-1][2]
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1; a[1] += (e - s)
    tot = sum(a[1] for a in agg.values())
    print(f"kernels: {len(rows)} dispatches, {tot/1e6:.2f} ms GPU busy, {(t1-t0)/1e6:.2f} ms wall span")
    print(f"{'name':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{n:110s} {c:7d} {t/1e6:10.3f} {t/c/1e3:9.2f} {100*t/tot:6.2f}")


if __name__ == "__main__":
    main()
