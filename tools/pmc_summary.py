#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (csv) into per-kernel HBM traffic.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.md> <out.json>

Counters are reported in KiB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section), on gfx950
FETCH_SIZE reads exactly half of the bytes of a wide coalesced streaming read, so the read side is doubled;
WRITE_SIZE is taken as is.  Calibration on this workload: k_cost streams a known 24 N + 24 P bytes.
"""
import json
import sys

import pandas as pd


def load(d, counter):
    import glob

    f = glob.glob(f"{d}/*counter_collection.csv")[0]
    df = pd.read_csv(f)
    df = df[df["Counter_Name"] == counter]
    df["kernel"] = df["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("void ", "").str.replace("cba::", "")
    return df.groupby("kernel")["Counter_Value"].agg(["mean", "count"])


def main(fetch_dir, write_dir, out_md, out_json):
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fe.index) | set(wr.index)):
        f = float(fe["mean"].get(k, 0.0)) * 1024.0
        w = float(wr["mean"].get(k, 0.0)) * 1024.0
        rows[k] = {"launches": int(fe["count"].get(k, wr["count"].get(k, 0))), "fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f,
                   "write_bytes": w, "hbm_bytes": 2 * f + w}
    lines = ["| kernel | launches | FETCH_SIZE raw MB | read MB (x2, gfx950) | WRITE_SIZE MB | HBM MB per launch |", "|---|---|---|---|---|---|"]
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes"]):
        if v["hbm_bytes"] < 1e5:
            continue
        lines.append(f"| {k} | {v['launches']} | {v['fetch_bytes_raw']/1e6:.1f} | {v['fetch_bytes_x2']/1e6:.1f} | {v['write_bytes']/1e6:.1f} | {v['hbm_bytes']/1e6:.1f} |")
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(rows, open(out_json, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:5])
