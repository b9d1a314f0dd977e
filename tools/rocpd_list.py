#!/usr/bin/env python
"""List the dispatches of kernels matching a substring (duration us, grid) from a rocpd db: rocpd_list.py file.db substr [max]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {disp} d join {sym} s on d.kernel_id = s.id order by d.start").fetchall()
n = 0; prev_end = None
for name, st, en, grid in rows:
    if sys.argv[2] in name:
        gap = (st - prev_end) / 1e3 if prev_end else 0.0
        print(f"{name[:40]:40s} grid {grid:8d} dur {(en-st)/1e3:8.2f} us  gap_before {gap:7.2f} us")
        n += 1
        if len(sys.argv) > 3 and n >= int(sys.argv[3]): break
    prev_end = en
