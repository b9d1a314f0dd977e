"""Timing experiments on the pair kernel: damped steps on one workload with the device timers on, for libraries built with one of the
CBA_EXP_* macros of csrc/cba_kernels.h (their sums are wrong on purpose; only the timers are read).
    CALISCOPE_BA_LIB=tools/exp/libcba_PAIR_NOMATH.so python tools/pair_phase_probe.py cfg4 10"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench
from caliscope_amd.hip_engine import HipEngine

name = sys.argv[1] if len(sys.argv) > 1 else 'cfg4'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sc, par, x0, prob, cfg = bench.build_problem(name)
eng = HipEngine(prob)
eng.begin(x0)
eng.linearize()
for _ in range(3):
    eng.newton_step(1e-3)
eng.enable_timers(True)
eng.reset_timers()
for _ in range(reps):
    eng.newton_step(1e-3)
t = eng.timers()
print(os.environ.get('CALISCOPE_BA_LIB', 'product library').split('/')[-1], {k: round(v[0] * 1e3 / max(v[1], 1), 1) for k, v in t.items() if k in ('schur', 'schur_pairs', 'schur_reduce_finalize', 'cholesky_solve')})
eng.close()
