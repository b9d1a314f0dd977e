"""The protocol of the persistent dense solve (caliscope_amd/csrc/chol_persist.h) on the CPU: task order, waits, signals and block arithmetic replayed by
tests/chol_persist_model.py under random interleavings and any number of task workgroups — no deadlock, no task reads a block another one is writing,
every block carries the panels the kernel's comments say, and x = T y solves the system."""
import numpy as np
import pytest

from chol_persist_model import FEED, PANEL, TSTEP, UPD, Model, task_list


def _system(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n + 40, n))
    S = A.T @ A + 1e-3 * np.eye(n)
    return S, rng.standard_normal(n)


@pytest.mark.parametrize("n,workers,seed", [(128, 1, 0), (128, 5, 1), (160, 2, 2), (100, 3, 3), (390, 7, 4), (384, 300, 5), (230, 1, 6)])
def test_protocol_solves_the_system_under_any_interleaving(n, workers, seed):
    S, rhs = _system(n, seed)
    m = Model(S, rhs, n_workers=workers, seed=seed)
    x = m.run()
    ref = np.linalg.solve(S, rhs)
    assert np.max(np.abs(x - ref)) <= 1e-8 * np.max(np.abs(ref))
    assert np.max(np.abs(m.factor() - np.linalg.cholesky(S))) <= 1e-9 * np.max(np.abs(S)) ** 0.5


def test_task_list_is_in_dependency_order():
    """Every task's inputs are produced by tasks with SMALLER tickets (or by the chain workgroup): what makes the launch free of deadlocks with any number
    of resident workgroups."""
    for nbk in (4, 5, 12, 36):
        tasks = task_list(nbk)
        pos = {t: q for q, t in enumerate(tasks)}
        assert len(pos) == len(tasks)  # no duplicates
        for q, (kind, k, b, j) in enumerate(tasks):
            need = []
            if kind == PANEL:
                need += [(UPD, kk, b, k) for kk in range(k)]
            elif kind == UPD:
                need += [(PANEL, k, b, 0), (PANEL, k, j, 0)] + ([(UPD, k - 1, b, j)] if k >= 1 else [])
                if b < nbk and k == b - 3 and j >= b - 2:
                    need.append((FEED, max(b - 3, 0), b, 0))
            elif kind == TSTEP:
                i, m = b, k - 1
                need.append((PANEL, m, i, 0))
                if m != j:
                    need += [(TSTEP, m, m, j), (TSTEP, k - 1, i, j)]
            else:
                k3 = b - 3
                if k3 >= 1:
                    need += [(UPD, k3 - 1, b, k3), (UPD, k3 - 1, b - 2, k3), (UPD, k3 - 1, b - 1, k3), (UPD, k3 - 1, b, b - 2), (UPD, k3 - 1, b, b - 1), (UPD, k3 - 1, b, b)]
            for t in need:
                assert t in pos and pos[t] < q, (nbk, tasks[q], t)
        # rows 2 .. nbk-1 get their mail exactly once; the diagonal block takes panels 0 .. b-4 only
        assert sorted(b for kind, k, b, j in tasks if kind == FEED) == list(range(2, nbk))
        assert all(k <= b - 4 for kind, k, b, j in tasks if kind == UPD and b == j)
