#!/usr/bin/env python
"""Mean of every collected PMC counter for the kernels matching a substring:  python tools/pmc_kernel.py <dir> <substr>"""
import glob, sys
import pandas as pd

for d in sys.argv[1].split(","):
    f = glob.glob(f"{d}/*counter_collection.csv")
    if not f:
        continue
    df = pd.read_csv(f[0])
    df = df[df["Kernel_Name"].str.contains(sys.argv[2])]
    print(df.groupby("Counter_Name")["Counter_Value"].mean().to_string())
