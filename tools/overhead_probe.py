"""Wall-clock per trust-region iteration with and without the HIP-event timers, and the host-side share."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from caliscope_amd.hip_engine import HipEngine
for name in ('cfg2', 'cfg3', 'cfg4'):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    eng = HipEngine(prob)
    eng.begin(x0)
    bench.run_iterations(eng, 8, {})
    for timers in (False, True, False):
        eng.enable_timers(timers); eng.reset_timers()
        t0 = time.perf_counter(); bench.run_iterations(eng, 40, {}); dt = time.perf_counter() - t0
        ksum = sum(v[0] for v in eng.timers().values()) if timers else float('nan')
        print(f"{name} timers={timers}: {dt/40*1e3:.4f} ms/step, kernel-family sum {ksum/40:.4f} ms/step", flush=True)
    # pure primitive sequence without the Python driver logic
    eng.enable_timers(False)
    eng.begin(x0)
    t0 = time.perf_counter()
    for _ in range(20):
        eng.linearize(); eng.newton_step(1e-6); eng.trial(0.0, 1e-3)
    print(f"{name} raw primitive triple: {(time.perf_counter()-t0)/20*1e3:.4f} ms", flush=True)
    eng.close()
