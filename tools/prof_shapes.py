"""Per-SHAPE table of the contraction kernels inside a profiled run: joins a rocprofv3 --kernel-trace result (rocpd sqlite .db
or the *_kernel_trace.csv) with the launch-tag table `bench.py --tag-gemm tags.json` wrote during the SAME run.

While tagging is on (csrc/debug_hooks.h: cl_debug_gemm_tag) every product signature launches `tag` extra, empty workgroups, so a
dispatch's workgroup count names its signature: tile kernels  workgroups = real + tag ; x-stationary kernel (gemm_xs)
workgroups = (row blocks + tag) * column runs.  Per signature: calls per step, average duration INSIDE the replayed step, TF/s,
algorithmic bytes, fraction of the shape's own roofline max(flops / 2.5 PF, bytes / 8 TB/s).

usage: python tools/prof_shapes.py <results.db | kernel_trace.csv> tags.json [--steady MARKER N] [--top 30]"""
import csv
import json
import re
import sqlite3
import sys

PEAK_TF, PEAK_TBS = 2500.0, 8.0


def load_rows(path):
    if path.endswith(".csv"):
        rows = []
        for r in csv.DictReader(open(path)):
            wg = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0)
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size"]), wg))
        return sorted(rows, key=lambda t: t[1])
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    if gx is None:
        raise SystemExit(f"no grid columns in the kernels view: {cols}")
    wx = "workgroup_x" if "workgroup_x" in cols else "workgroup_size_x"
    q = (f"select name, start, end, {gx} * {gx.replace('_x', '_y')} * {gx.replace('_x', '_z')}, "
         f"{wx} * {wx.replace('_x', '_y')} * {wx.replace('_x', '_z')} from kernels order by start")
    return list(cur.execute(q))


def main():
    rows = load_rows(sys.argv[1])
    tags = json.load(open(sys.argv[2]))
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 30
    nsteps = 1
    if "--steady" in sys.argv:
        i = sys.argv.index("--steady")
        marker, nsteps = sys.argv[i + 1], int(sys.argv[i + 2])
        ends = [e for n, s, e, g, w in rows if marker in n]
        lo, hi = ends[-nsteps - 1], ends[-1]
        rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    agg, unmatched, un_t = {}, 0, 0.0
    for name, s, e, grid, wg in rows:
        if "gemm_fl" not in name and "gemm_kernel" not in name and "gemm_xs" not in name:
            continue
        wgs = grid // max(wg, 1)
        xs = "gemm_xs" in name
        cand = []
        for t in tags:
            if not t["workgroups"] or t["wg_size"] != wg:
                continue
            extra = wgs - t["workgroups"]
            if xs:
                if extra > 0 and extra % t["tag"] == 0 and t["workgroups"] % (extra // t["tag"]) == 0:
                    cand.append(t)
            elif extra == t["tag"]:
                cand.append(t)
        if len(cand) != 1:
            unmatched += 1; un_t += e - s
            continue
        t = cand[0]
        key = (t["mode"], t["M"], t["N"], t["K1"], t["K2"], t["act"], t["residual"], re.sub(r"\(.*$", "", name).replace("void cl::", "")[:48])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += e - s
    out = []
    for (mode, M, N, K1, K2, act, res, kern), (n, tt) in agg.items():
        taps = 1 if mode == 0 else 9
        fl = 2.0 * M * N * (taps * K1 + K2)
        no = N // 2 if act in (2, 3) else N
        by = 2.0 * (M * (K1 + K2) + N * (taps * K1 + K2) + M * no * (2 if res else 1))
        us = tt / n / 1e3
        roof_us = max(fl / PEAK_TF * 1e-6, by / PEAK_TBS * 1e-6)
        out.append((tt / nsteps / 1e3, n / nsteps, us, fl / us * 1e-6, by / us * 1e-6, roof_us / us, mode, M, N, K1, K2, act, res, kern))
    out.sort(reverse=True)
    tot = sum(o[0] for o in out)
    print(f"contraction kernels by product signature INSIDE the profiled step(s): {len(out)} signatures x kernels, {tot * 1e-3:.2f} ms per step "
          f"({unmatched} dispatches = {un_t / nsteps * 1e-6:.3f} ms per step could not be attributed); peaks {PEAK_TF:.0f} TF/s, {PEAK_TBS} TB/s")
    print(f"{'us/step':>8} {'n/step':>6} {'avg us':>8} {'TF/s':>7} {'TB/s':>6} {'of roof':>7}  mode M N K1 K2 act res  kernel")
    for o in out[:top]:
        print(f"{o[0]:8.0f} {o[1]:6.1f} {o[2]:8.1f} {o[3]:7.1f} {o[4]:6.2f} {o[5]:7.3f}  {o[6]} {o[7]} {o[8]} {o[9]} {o[10]} {o[11]} {o[12]}  {o[13]}")


if __name__ == "__main__":
    main()
