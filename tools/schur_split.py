"""Time the Schur pair kernel with parts of it switched off (CBA_DEBUG_SCHUR_SKIP bits: 1 pair loop, 2 record gather,
4 LDS stores, 8 index / code loads): what each part costs when the others are gone.  Results are garbage, timings are not."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
masks = [int(m) for m in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 6, 7, 8, 15]
sc, par, x0, prob, cfg = bench.build_problem(name)
from caliscope_amd.hip_engine import HipEngine
for skip in masks:
    os.environ['CBA_DEBUG_SCHUR_SKIP'] = str(skip)
    eng = HipEngine(prob)
    eng.begin(x0); eng.linearize()
    try:
        eng.newton_step(1e-6)
    except Exception as exc:
        print('skip', skip, 'warm-up step raised', exc, flush=True)
    eng.enable_timers(True); eng.reset_timers()
    for _ in range(6):
        try:
            eng.newton_step(1e-6)
        except Exception:
            pass
    t = eng.timers()
    print('skip', skip, {k: round(v[0]/max(v[1],1)*1e3,1) for k, v in t.items() if v[1] and k in ('schur', 'schur_pairs', 'schur_reduce_finalize', 'cholesky_solve')}, flush=True)
    eng.close()
