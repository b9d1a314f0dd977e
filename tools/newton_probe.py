"""Run a few linearize / newton_step / trial calls on one workload (for rocprofv3 kernel traces)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from caliscope_amd.hip_engine import HipEngine

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc, par, x0, prob, cfg = bench.build_problem(name)
import time
t0 = time.time()
eng = HipEngine(prob)
print('engine created in %.3f s' % (time.time() - t0), file=sys.stderr)
eng.begin(x0)
eng.linearize()
for _ in range(reps):
    eng.newton_step(1e-6)
    eng.trial(1.0, 0.0)
eng.close()
