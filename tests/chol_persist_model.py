"""Host model of the persistent dense solve (caliscope_amd/csrc/chol_persist.h): the same task list, waits, signals and block arithmetic with numpy,
the chain workgroup and the task workgroups run as coroutines that yield where the kernel spins.  It checks what cannot be seen from a passing GPU
test alone: that every wait is satisfied by a task with a smaller ticket (or by the chain workgroup) for ANY number of task workgroups, that no two
tasks write a block another one is reading, and that the blocks each task reads carry exactly the panels the kernel's comments say they carry.

Used by tests/test_chol_persist_model.py (CPU suite); the task list itself comes from the library's own generator through the C test harness when that
is built, else from the port below (the two are compared there)."""

from __future__ import annotations

import numpy as np

NB = 32
PANEL, UPD, TSTEP, FEED = 0, 1, 2, 3


def task_list(nbk: int) -> list[tuple[int, int, int, int]]:
    """chol_persist_tasks (chol_persist.h), line for line."""
    out = []
    for k in range(nbk):
        if k == 0 and 2 <= nbk - 1:
            out.append((FEED, 0, 2, 0))
        if k + 3 <= nbk - 1:
            out.append((FEED, k, k + 3, 0))
        for b in range(k + 1, nbk + 1):
            out.append((PANEL, k, b, 0))
        d = 1
        while k + d <= nbk - 1:
            j = k + d
            for b in range(j, nbk + 1):
                if b == j and k > b - 4:
                    continue
                out.append((UPD, k, b, j))
            d += 1
        if k >= 1:
            for i in range(k, nbk):
                for j in range(k):
                    out.append((TSTEP, k, i, j))
    return out


class Model:
    def __init__(self, S: np.ndarray, rhs: np.ndarray, n_workers: int = 3, seed: int = 0):
        n = S.shape[0]
        self.n, self.nbk = n, (n + NB - 1) // NB
        self.W = np.zeros((n + 1, n))
        self.W[:n] = S
        self.W[n] = rhs
        self.Lm = np.zeros((n + 1, n))
        self.T = np.zeros((n, n))
        self.X = {}
        self.mail = {}
        self.F = 0
        self.Pn, self.Sn, self.Un, self.Tn = {}, {}, {}, {}
        self.tasks = task_list(self.nbk)
        self.n_workers = n_workers
        self.rng = np.random.default_rng(seed)
        self.applied = {}       # block (b, j) -> set of panels applied to the U block in W
        self.reading = {}       # block -> number of tasks currently between their load and their end (a writer must not overlap)
        self.log = []

    # --- geometry
    def r0(self, b):
        return b * NB if b < self.nbk else self.n

    def rc(self, b):
        return min(NB, self.n - b * NB) if b < self.nbk else 1

    def blk(self, M, b, j):
        return M[self.r0(b): self.r0(b) + self.rc(b), j * NB: j * NB + self.rc(j)]

    def get(self, table, key):
        return table.get(key, 0)

    # --- the chain workgroup
    def chain(self):
        n, nbk = self.n, self.nbk
        D = self.blk(self.W, 0, 0).copy()
        Bn = self.blk(self.W, 1, 0).copy() if nbk > 1 else None
        Dn = self.blk(self.W, 1, 1).copy() if nbk > 1 else None
        Lcur = None
        for k in range(nbk):
            Lkk = np.linalg.cholesky(D)
            Xk = np.linalg.inv(Lkk)
            yield ("work", "factor", k)
            self.blk(self.Lm, k, k)[:] = Lkk
            self.X[k] = Xk
            self.blk(self.T, k, k)[:] = Xk.T
            self.F = k + 1
            if k + 1 >= nbk:
                break
            Lcur_new = Bn @ Xk.T
            D = Dn - Lcur_new @ Lcur_new.T
            # the job of the NEXT factorisation (row k + 2), from the mail, X_k and L_k+1,k
            if k + 2 < nbk:
                b = k + 2
                while self.get(self.Sn, b) < 1:
                    yield ("wait", "Sn", b)
                m0, m1, m2 = self.mail[b]
                La = m0 @ Xk.T
                Bn = m1 - La @ Lcur_new.T
                Dn = m2 - La @ La.T
            Lcur = Lcur_new
        yield ("done",)

    # --- one task workgroup
    def worker(self, take):
        while True:
            t = take()
            if t is None:
                yield ("done",)
                return
            kind, k, b, j = self.tasks[t]
            if kind == PANEL:
                while not (self.F >= k + 1 and self.get(self.Un, (b, k)) >= k):
                    yield ("wait", "PANEL", t)
                assert self.applied.get((b, k), set()) == set(range(k)), ("PANEL reads a block with the wrong panels", b, k, self.applied.get((b, k)))
                self.blk(self.Lm, b, k)[:] = self.blk(self.W, b, k) @ self.X[k].T
                yield ("work", "PANEL", t)
                self.Pn[b] = k + 1
            elif kind == UPD:
                after_feed = b < self.nbk and k == b - 3 and j >= b - 2
                while not (self.get(self.Pn, b) >= k + 1 and self.get(self.Pn, j) >= k + 1 and self.get(self.Un, (b, j)) >= k and
                           (not after_feed or self.get(self.Sn, b) >= 1)):
                    yield ("wait", "UPD", t)
                assert self.reading.get((b, j), 0) == 0, ("UPD writes a block somebody is reading", b, j, k)
                assert self.applied.get((b, j), set()) == set(range(k)), ("UPD out of order", b, j, k)
                self.blk(self.W, b, j)[:] -= self.blk(self.Lm, b, k) @ self.blk(self.Lm, j, k).T
                self.applied.setdefault((b, j), set()).add(k)
                yield ("work", "UPD", t)
                self.Un[(b, j)] = k + 1
            elif kind == TSTEP:
                i, m = b, k - 1
                while not ((self.F >= j + 1 if m == j else self.get(self.Tn, (j, m)) >= m - j + 1) and self.get(self.Pn, i) >= m + 1 and
                           self.get(self.Tn, (j, i)) >= m - j):
                    yield ("wait", "TSTEP", t)
                Tb = self.blk(self.T, j, i)
                acc = (Tb if m != j else 0.0) + self.blk(self.T, j, m) @ self.blk(self.Lm, i, m).T
                if i != k:
                    Tb[:] = acc
                else:
                    while not self.F >= k + 1:
                        yield ("wait", "TSTEP-final", t)
                    Tb[:] = -acc @ self.X[k].T
                yield ("work", "TSTEP", t)
                self.Tn[(j, i)] = m - j + 1 + (1 if i == k else 0)
            else:  # FEED
                k3 = b - 3
                if k3 < 0:
                    cols = [b - 2, b - 1, b]
                    for c in cols:
                        assert self.applied.get((b, c), set()) == set(), ("FEED(2) wants the original blocks", b, c)
                    self.mail[b] = tuple(self.blk(self.W, b, c).copy() for c in cols)
                    yield ("work", "FEED", t)
                    self.Sn[b] = 1
                    continue
                while not (self.F >= k3 + 1 and all(self.get(self.Un, key) >= k3 for key in
                                                    ((b, k3), (b - 2, k3), (b - 1, k3), (b, b - 2), (b, b - 1), (b, b)))):
                    yield ("wait", "FEED", t)
                keys = [(b, b - 2), (b, b - 1), (b, b)]
                for key in keys:
                    assert self.applied.get(key, set()) == set(range(k3)), ("FEED reads a block with the wrong panels", key, self.applied.get(key))
                    self.reading[key] = self.reading.get(key, 0) + 1
                X = self.X[k3]
                L0 = self.blk(self.W, b, k3) @ X.T
                L1 = self.blk(self.W, b - 2, k3) @ X.T
                L2 = self.blk(self.W, b - 1, k3) @ X.T
                yield ("work", "FEED-load", t)  # (other tasks run between this task's loads and its stores)
                self.mail[b] = (self.blk(self.W, b, b - 2) - L0 @ L1.T, self.blk(self.W, b, b - 1) - L0 @ L2.T, self.blk(self.W, b, b) - L0 @ L0.T)
                for key in keys:
                    self.reading[key] -= 1
                yield ("work", "FEED", t)
                self.Sn[b] = 1

    def run(self, max_rounds: int = 2_000_000):
        next_ticket = [0]
        taken_by = {}

        def make_take(w):
            def take():
                t = next_ticket[0]
                next_ticket[0] += 1
                if t >= len(self.tasks):
                    return None
                taken_by[t] = w
                return t
            return take

        procs = [self.chain()] + [self.worker(make_take(w)) for w in range(self.n_workers)]
        alive = [True] * len(procs)
        idle_rounds = 0
        for _ in range(max_rounds):
            if not any(alive):
                break
            progressed = False
            order = self.rng.permutation(len(procs))  # a random schedule: any interleaving must work
            for q in order:
                if not alive[q]:
                    continue
                ev = next(procs[q])
                if ev[0] == "done":
                    alive[q] = False
                    progressed = True
                elif ev[0] == "work":
                    progressed = True
            idle_rounds = 0 if progressed else idle_rounds + 1
            assert idle_rounds < 50, "deadlock: every workgroup is waiting"
        assert not any(alive), "did not finish"
        y = self.Lm[self.n, : self.n]
        return self.T @ y

    def factor(self):
        return np.tril(self.Lm[: self.n])
