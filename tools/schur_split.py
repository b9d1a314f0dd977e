import os, sys, json
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
for skip in ('0', '1', '2', '3'):
    os.environ['CBA_DEBUG_SCHUR_SKIP'] = skip
    from caliscope_amd.hip_engine import HipEngine
    sc, par, x0, prob, cfg = bench.build_problem('cfg4')
    eng = HipEngine(prob)
    eng.begin(x0); eng.linearize()
    eng.enable_timers(True); eng.reset_timers()
    for _ in range(5): eng.newton_step(1e-6)
    t = eng.timers()
    print('skip', skip, {k: round(v[0]/max(v[1],1)*1e3,1) for k, v in t.items() if v[1]}, flush=True)
    eng.close()
