"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db): per-kernel time, calls, share.
Usage: python tools/prof_summary.py <results.db> [--steady MARKER N] > profiles/xxx.txt

--steady MARKER N: restrict to the window spanned by the last N occurrences of the kernel whose name contains
MARKER (e.g. `adamw_dev_kernel` ends every training step, `ddim_step_dev_kernel` every denoise step), i.e. exactly N
steady-state steps without model build / packing / warm-up; totals are then also printed per step."""
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
    nsteps = None
    if "--steady" in sys.argv:
        i = sys.argv.index("--steady")
        marker, nsteps = sys.argv[i + 1], int(sys.argv[i + 2])
        ends = [e for n, s, e in rows if marker in n]
        if len(ends) <= nsteps:
            print(f"only {len(ends)} occurrences of {marker}"); return
        lo, hi = ends[-nsteps - 1], ends[-1]
        rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    t0, t1 = rows[0][1], rows[-1][2]
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0.0])
        a[0] += 1; a[1] += (e - s)
    tot = sum(a[1] for a in agg.values())
    print(f"kernels: {len(rows)} dispatches, {tot/1e6:.2f} ms GPU busy, {(t1-t0)/1e6:.2f} ms wall span")
    if nsteps:
        print(f"steady state: {nsteps} steps -> {len(rows)/nsteps:.1f} dispatches, {tot/1e6/nsteps:.3f} ms GPU busy, "
              f"{(t1-t0)/1e6/nsteps:.3f} ms wall per step")
    print(f"{'name':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{n:110s} {c:7d} {t/1e6:10.3f} {t/c/1e3:9.2f} {100*t/tot:6.2f}")


if __name__ == "__main__":
    main()
