#!/usr/bin/env python
"""Per-kernel means of every counter in a rocprofv3 --pmc csv run:  python tools/pmc_table.py <dir>[,<dir>...]"""
import glob, sys
import pandas as pd

frames = []
for d in sys.argv[1].split(","):
    for f in glob.glob(f"{d}/*counter_collection.csv"):
        frames.append(pd.read_csv(f))
df = pd.concat(frames)
df["kernel"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("cba::", "").str.slice(0, 34)
t = df.pivot_table(index="kernel", columns="Counter_Name", values="Counter_Value", aggfunc="mean")
pd.set_option("display.width", 250); pd.set_option("display.max_columns", 30)
print(t.sort_values(t.columns[0], ascending=False).head(int(sys.argv[2]) if len(sys.argv) > 2 else 14).to_string(float_format=lambda v: f"{v:.3e}"))
