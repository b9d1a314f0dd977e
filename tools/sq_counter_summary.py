#!/usr/bin/env python
"""Per-kernel means of the SQ counters of one or more rocprofv3 --pmc passes (csv): sq_counter_summary.py out.md dir [dir ...]
Counts are per dispatch, summed over the device's shader engines as rocprofv3 reports them.  SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count
quad-cycles per wave (MI355X_MICROARCH.md, PMC section); ratios between them are what is read off."""
import glob
import sys

import pandas as pd

out, dirs = sys.argv[1], sys.argv[2:]
frames = []
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        frames.append(pd.read_csv(f))
df = pd.concat(frames)
df["kernel"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("cba::", "")
tab = df.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
cnt = df[df["Counter_Name"] == df["Counter_Name"].iloc[0]].groupby("kernel").size()
keep = [k for k in tab.index if k.startswith("k_schur_reg3") or k.startswith("k_build_cs") or k.startswith("k_tprep") or k.startswith("k_backsub") or k.startswith("k_jv") or k.startswith("k_chol_step")]
cols = list(tab.columns)
lines = ["| kernel | launches | " + " | ".join(cols) + " |", "|---|---|" + "---|" * len(cols)]
for k in keep:
    lines.append(f"| {k} | {int(cnt.get(k, 0))} | " + " | ".join(f"{tab.loc[k, c]:.4g}" for c in cols) + " |")
w = "SQ_WAVE_CYCLES"
if w in cols:
    lines += ["", "Shares of SQ_WAVE_CYCLES:", "", "| kernel | " + " | ".join(c for c in cols if c != w and (c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE"))) + " |"]
    sub = [c for c in cols if c != w and (c.startswith("SQ_WAIT") or c.startswith("SQ_ACTIVE"))]
    lines.append("|---|" + "---|" * len(sub))
    for k in keep:
        lines.append(f"| {k} | " + " | ".join(f"{tab.loc[k, c] / tab.loc[k, w]:.3f}" for c in sub) + " |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
