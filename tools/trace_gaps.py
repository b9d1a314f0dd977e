#!/usr/bin/env python
"""Idle time of the GPU inside the timed region of a kernel trace: trace_gaps.py file.db [marker_kernel]
Groups dispatches into LM iterations by the Schur kernel (k_schur_reg / k_schur_tile) and prints, per iteration, the sum of
kernel durations, the wall span and the gaps larger than 3 us with the kernel that follows them."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
marker = sys.argv[2] if len(sys.argv) > 2 else "k_schur_reg"
idx = [i for i, r in enumerate(rows) if marker in r[0]]
if len(idx) < 12:
    sys.exit("not enough iterations in the trace")
lo, hi = idx[len(idx) // 2], idx[len(idx) // 2 + 8]  # eight iterations from the middle of the run
seg = rows[lo:hi]
busy = sum(e - s for _, s, e in seg) / 1e3
span = (seg[-1][2] - seg[0][1]) / 1e3
gaps = collections.defaultdict(lambda: [0, 0.0])
for (n0, s0, e0), (n1, s1, e1) in zip(seg[:-1], seg[1:]):
    g = (s1 - e0) / 1e3
    if g > 3.0:
        key = f"{n0.split('(')[0][:28]} -> {n1.split('(')[0][:28]}"
        gaps[key][0] += 1; gaps[key][1] += g
print(f"8 iterations: span {span:.1f} us, kernels {busy:.1f} us, idle {span - busy:.1f} us ({100 * (span - busy) / span:.1f} %), {len(seg)} dispatches")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / 8:7.1f} us/iter  x{n / 8:4.1f}  {k}")
