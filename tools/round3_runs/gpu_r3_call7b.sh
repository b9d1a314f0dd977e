#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c7; mkdir -p $O
timeout 300 python tools/create_timing.py > $O/create_full.log 2>&1
python - <<'PY'
import re
txt=open("gpurun_out/c7/create_full.log").read()
# print the block before the second cfg4-half create line
blocks=txt.split("== ")
for b in blocks[5:8]: print("== "+b[-1500:] if len(b)>1500 else "== "+b)
PY
timeout 120 python - <<'PY'
import time, sys, os
sys.path.insert(0, ".")
import numpy as np, bench
from caliscope_amd import engine_cache
from caliscope_amd.engine import BAProblem
sc, par, x0, prob, cfg = bench.build_problem("cfg4", n_points=100_000, n_obs=1_000_000)
for rep in range(3):
    t=time.perf_counter(); p2 = BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices); t1=time.perf_counter()
    fp = engine_cache.fingerprint(p2, 0, False); t2=time.perf_counter()
    print(f"BAProblem {1e3*(t1-t):.2f} ms, fingerprint {1e3*(t2-t1):.2f} ms")
PY
