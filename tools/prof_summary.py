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
    if "--gaps" in sys.argv:
        # timeline coverage: time with >= 1 kernel running (union of intervals), time with >= 2, and the idle gaps between kernels
        ev = sorted([(s, 1) for _, s, e in rows] + [(e, -1) for _, s, e in rows])
        depth, last, cover, over, gaps = 0, ev[0][0], 0, 0, []
        for t, d in ev:
            if depth >= 1: cover += t - last
            if depth >= 2: over += t - last
            if depth == 0 and t > last: gaps.append(t - last)
            depth += d; last = t
        k = max(1, nsteps or 1)
        wall = t1 - t0
        # the largest gaps with the kernels on either side (copies / memsets / event waits of the graph are not kernels: they show here)
        srt = sorted(rows, key=lambda r: r[1])
        big, end_so_far, last_name = [], srt[0][2], srt[0][0]
        for n, s_, e_ in srt[1:]:
            if s_ > end_so_far: big.append((s_ - end_so_far, short(last_name)[:60], short(n)[:60]))
            if e_ > end_so_far: end_so_far, last_name = e_, n
        big.sort(reverse=True)
        for g_, a_, b_ in big[:16 * k if False else 24]:
            print(f"  gap {g_/1e3:8.1f} us   after {a_:60s} before {b_}")
        gaps.sort()
        k = max(1, nsteps or 1)
        print(f"timeline: covered {cover/1e6/k:.3f} ms, two or more kernels at once {over/1e6/k:.3f} ms, idle {(wall-cover)/1e6/k:.3f} ms per step "
              f"({100*(wall-cover)/wall:.1f} % of the wall) in {len(gaps)/k:.0f} gaps per step; gap median {gaps[len(gaps)//2]/1e3:.2f} us, "
              f"p90 {gaps[int(len(gaps)*0.9)]/1e3:.2f} us, max {gaps[-1]/1e3:.1f} us" if gaps else "timeline: no gaps")
    print(f"{'name':110s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{n:110s} {c:7d} {t/1e6:10.3f} {t/c/1e3:9.2f} {100*t/tot:6.2f}")


if __name__ == "__main__":
    main()
