import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from caliscope_amd.hip_engine import HipEngine
sc, par, x0, prob, cfg = bench.build_problem('cfg4')
eng = HipEngine(prob)
eng.begin(x0); eng.linearize()
for _ in range(3): eng.newton_step(1e-6)
eng.close()
