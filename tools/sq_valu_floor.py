#!/usr/bin/env python
"""profiles/rNN_<workload>_sq_counters.md (tools/sq_counter_summary.py) -> profiles/sq_<workload>.json: SQ_INSTS_VALU (and the other counters) per
launch of the dominant kernels, the table bench.py's `roofline.iteration.valu_floor` is computed from.

    python tools/sq_valu_floor.py profiles/r05_cfg4_sq_counters.md profiles/sq_cfg4.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
lines = [l for l in open(src).read().splitlines() if l.startswith("|")]
head = [c.strip() for c in lines[0].strip("|").split("|")]
out = {}
for l in lines[2:]:
    cells = [c.strip() for c in l.strip("|").split("|")]
    if len(cells) != len(head) or cells[0] in ("kernel", "---"):
        break  # (the second table of the file: shares)
    try:
        out[cells[0]] = {h: float(v) for h, v in zip(head[1:], cells[1:])}
    except ValueError:
        break
json.dump(out, open(dst, "w"), indent=1)
print(dst, {k: v.get("SQ_INSTS_VALU") for k, v in out.items()})
