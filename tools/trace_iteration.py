#!/usr/bin/env python
"""The dispatches of ONE trust-region iteration from a rocpd kernel trace, in order, with their durations and the idle time in front of each:
trace_iteration.py file.db [marker_kernel [where]]   (an iteration = from one launch of the marker kernel to the next; `where` in 0..1 picks the
iteration by its position in the run: 0.5 = the middle, the default; a trace of two workloads one after the other has the first one early)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_tprep"
idx = [i for i, r in enumerate(rows) if marker in r[0]]
if len(idx) < 6:
    sys.exit("not enough iterations in the trace")
where = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
at = min(max(int(len(idx) * where), 0), len(idx) - 2)
lo, hi = idx[at], idx[at + 1]
busy = 0.0
print(f"| # | kernel | workgroups x threads | us | idle before, us |\n|---|---|---|---|---|")
for k, (name, st, en, grid, wg) in enumerate(rows[lo:hi]):
    gap = (st - rows[lo + k - 1][2]) / 1e3
    busy += (en - st) / 1e3
    print(f"| {k + 1} | {name.split('(')[0][:48]} | {grid // max(wg, 1)} x {wg} | {(en - st) / 1e3:.2f} | {gap:.2f} |")
span = (rows[hi][1] - rows[lo][1]) / 1e3
print(f"\n{hi - lo} dispatches, {busy:.1f} us in kernels, {span:.1f} us from the first to the next iteration's first ({span - busy:.1f} us idle)")
