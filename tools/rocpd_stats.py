#!/usr/bin/env python
"""Per-kernel statistics from a rocprofv3 rocpd (.db) kernel trace:  python tools/rocpd_stats.py file.db [out.md]"""
import re
import sqlite3
import sys


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), max(s.arch_vgpr_count) "
        f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    out = ["| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | LDS B | VGPR |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, n, tot, mn, mx, grid, wg, lds, vgpr in rows:
        short = re.sub(r"\(.*", "", name or "?")
        out.append(f"| {short[:60]} | {n} | {tot/1e6:.3f} | {tot/n/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.1f} | {grid} | {wg} | {lds} | {vgpr} |")
    return "\n".join(out)


if __name__ == "__main__":
    text = stats(sys.argv[1])
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)
