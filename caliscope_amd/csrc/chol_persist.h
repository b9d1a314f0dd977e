// Dense solve of the reduced camera system in ONE persistent launch (round 5).  Included by cba_kernels.h behind the launch-per-panel kernels, whose
// building blocks it reuses (chol_factor_block, chol_factor_store, the FP64 MFMA tile product).
//
// Why.  k_chol_step needs one launch per 32-column panel, and the critical workgroup of every launch pays a global round trip for its operands, three
// MFMA stages with barriers, the 32-pivot chain (5.7 us) and the launch boundary: 10.5 us per panel, 142 us for cfg4's 12 panels, 475 us for cfg5's 36 —
// a fifth of the iteration, repeated on every rank of a sharded solve.  Only the pivot chain is inherently serial.  Here ONE workgroup (the chain
// workgroup, block 0) does nothing but that chain and the two small products that connect one diagonal block to the next, with everything it needs
// already in its LDS; all other work is done around it by the other workgroups, which take TASKS from a ticket counter:
//
//   chain workgroup, for k = 0 .. nbk - 1:   [D_k is in LDS]  wave 0 factors it (L_kk, X_k = L_kk^-1)          -- 5.7 us
//        store L_kk, X_k;  publish F = k + 1;  L_k+1,k = B X_k^T;  D_k+1 = D - L_k+1,k L_k+1,k^T                 -- ~1 us, two MFMA stages
//     where B = block (k+1, k) and D = block (k+1, k+1) with the panels 0 .. k-1 applied.  They were prepared DURING the previous factorisation by two
//     otherwise idle waves of the same workgroup from the "mail" of row k+1 — its blocks (k+1, k-1), (k+1, k), (k+1, k+1) with panels 0 .. k-2 applied,
//     which a FEED task left in global memory a whole step earlier — and from X_k-1, L_k,k-1, which the workgroup has in LDS.
//
//   tasks (8 waves each, operands staged in LDS, 32 x 32 x 32 products on the FP64 matrix cores):
//     PANEL(b, k)     L_bk = U_bk X_k^T                                        needs F > k and the block's earlier updates
//     UPD(b, j, k)    block (b, j) -= L_bk L_jk^T          (k < j <= b)         needs both panels; one writer per block at a time (fixed order)
//     FEED(b)         the mail of row b (see above), panel b-3 applied with L blocks it recomputes itself, so that it hangs on ONE hand-over only
//     TSTEP(i, j, k)  T = L^-T, built beside the factorisation as in k_chol_step's inverse role; x = T y stays k_chol_apply
//   U blocks live in the work matrix W, L blocks go to a second matrix Lm (a FEED task reads U blocks a PANEL task would otherwise overwrite).
//
// Synchronisation: monotone counters in global memory (one writer at a time each), written with a release / read with an acquire at agent scope — the
// XCDs' L2s are not coherent with each other, the fences write back / invalidate (MI355X_MICROARCH.md; measured hand-over of an 8 KB block between two
// workgroups: 2.2-2.6 us, tools note in profiles/r05_chol_persist.txt).  Counters carry an epoch base, so nothing is cleared between solves.
// Deadlock freedom: the task list is in dependency order and a workgroup takes tickets in order, so every task depends only on tasks with smaller
// tickets (taken by workgroups that are running) and on the chain workgroup, which is block 0 and therefore resident whenever any block is — the
// launch makes progress with ANY number of resident workgroups (ranks sharing a device, a busy GPU).  Every spin has a time-out (50 ms): it raises the
// abort word, all workgroups leave, the host reports CBA_ERR_HIP and the handle falls back to the launch-per-panel path.
#pragma once

namespace cba {

struct CholTask { unsigned char kind, pad; unsigned short k, b, j; };  // 8 bytes
static_assert(sizeof(CholTask) == 8, "task record");
enum { CT_PANEL = 0, CT_UPD = 1, CT_TSTEP = 2, CT_FEED = 3 };
constexpr int CP_THREADS = 512;
constexpr int CP_MIN_BLOCKS = 4;                    // fewer row blocks: launch-per-panel path (k_small_solve covers ncp <= 96 on one rank)
constexpr int CP_MAX_BLOCKS = 48;                   // task list ~nbk^3 / 3 entries: beyond 1536 camera parameters the launch-per-panel path
constexpr long long CP_TIMEOUT_TICKS = 5000000LL;   // 50 ms of the 100 MHz wall clock

struct CholP {
  double *W, *Lm, *Xinv, *Tinv, *mail;  // mail: [nbk][3][NB * NB]
  const CholTask* tasks;
  int n_tasks;
  unsigned long long* ticket;
  unsigned long long ticket_base;
  int* sync;   // [0] F, [1] abort, then Pn[nbk + 1], Sn[nbk + 1], Un[(nbk + 1)^2], Tn[nbk^2]
  int n, ldw, nbk, base;
  int* flags;
  long long* trace;  // profiling build (-DCBA_PROFILING, CBA_CHOL_TRACE=persist): [nbk + 1][8] stamps of the chain workgroup, 100 MHz; else nullptr
};
#ifdef CBA_PROFILING
#define CP_STAMP(c, k, ph) do { if ((c).trace) (c).trace[(long)(k) * 8 + (ph)] = wall_clock64(); } while (0)
#else
#define CP_STAMP(c, k, ph) do { } while (0)
#endif

__device__ __forceinline__ int* cp_F(const CholP& c) { return c.sync; }
__device__ __forceinline__ int* cp_abort(const CholP& c) { return c.sync + 1; }
__device__ __forceinline__ int* cp_Pn(const CholP& c, int b) { return c.sync + 2 + b; }
__device__ __forceinline__ int* cp_Sn(const CholP& c, int b) { return c.sync + 2 + (c.nbk + 1) + b; }
__device__ __forceinline__ int* cp_Un(const CholP& c, int b, int j) { return c.sync + 2 + 2 * (c.nbk + 1) + b * (c.nbk + 1) + j; }
__device__ __forceinline__ int* cp_Tn(const CholP& c, int j, int i) { return c.sync + 2 + 2 * (c.nbk + 1) + (c.nbk + 1) * (c.nbk + 1) + j * c.nbk + i; }
__host__ __device__ inline size_t cp_sync_ints(int nbk) { return 2 + 2 * (size_t)(nbk + 1) + (size_t)(nbk + 1) * (nbk + 1) + (size_t)nbk * nbk; }

// Polling.  A counter is written by a workgroup on another XCD; the reader polls it with RELAXED device-scope loads (they are served from the memory
// side, not from this XCD's L2) and executes ONE acquire fence when the value is there.  Measured (profiles/r05_chol_persist.txt): with an acquire per
// poll — an L2 invalidate each — 250 polling workgroups slowed every memory access of the device down so much that a step of the chain workgroup took
// 50-90 us instead of 7; with relaxed polls every 0.1 us it took 17.  So polls are RARE: the first re-check after 0.25 us, then every 0.5 .. 2 us, and a
// wait whose counter is still several steps of the chain away sleeps through them.
__device__ __forceinline__ void cp_nap(unsigned spins, int far_steps) {
  if (far_steps > 1) {  // the awaited value is >= 2 steps of the chain workgroup (~7 us each) away: ~3.4 us per s_sleep 127
    for (int q = 0; q < min(far_steps - 1, 4); ++q) __builtin_amdgcn_s_sleep(127);
    return;
  }
  if (spins < 2u) __builtin_amdgcn_s_sleep(9);        // 0.25 us
  else if (spins < 6u) __builtin_amdgcn_s_sleep(18);  // 0.5 us
  else if (spins < 12u) __builtin_amdgcn_s_sleep(36); // 1 us
  else __builtin_amdgcn_s_sleep(72);                  // 2 us
}
// one thread: wait until *p >= base + need.  false: aborted (by a peer, or by this wait's own time-out).  The caller issues the acquire fence.
__device__ __forceinline__ bool cp_wait(const CholP& c, const int* p, int need) {
  if (need <= 0) return true;  // (nothing to wait for: counters of earlier solves lie below this solve's base)
  const int want = c.base + need;
  const long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= want) return true;
    if ((spins & 7u) == 7u) {
      if (__hip_atomic_load(cp_abort(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      if (wall_clock64() - t0 > CP_TIMEOUT_TICKS) {
        __hip_atomic_store(cp_abort(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    cp_nap(spins, 0);
  }
}
// wave 0 of a task workgroup: up to 8 conditions *p[q] >= base + need[q], one per lane, polled TOGETHER (a FEED task has seven: one after the other
// they cost seven memory round trips before the task even starts).  `far`: index of the condition that counts steps of the chain workgroup (F), or -1.
// Returns false on abort / time-out.  All 64 lanes call it.
struct CpWaits { const int* p[8]; int need[8]; int n; int far; };
__device__ __forceinline__ bool cp_wait_many(const CholP& c, const CpWaits& w, int lane) {
  const int* mine = nullptr;
  int want = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (q == lane && q < w.n && w.need[q] > 0) { mine = w.p[q]; want = c.base + w.need[q]; }
  const long long t0 = wall_clock64();
  for (unsigned spins = 0;; ++spins) {
    const int v = mine ? __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const bool ok = mine == nullptr || v >= want;
    if (__all(ok)) return true;
    if ((spins & 7u) == 7u) {
      const bool stop = __hip_atomic_load(cp_abort(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (__any(stop)) return false;
      if (wall_clock64() - t0 > CP_TIMEOUT_TICKS) {
        if (lane == 0) __hip_atomic_store(cp_abort(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    int far_steps = 0;
    if (w.far >= 0) far_steps = __shfl((lane == w.far && mine) ? want - v : 0, w.far < 0 ? 0 : w.far, WAVE);
    cp_nap(spins, far_steps);
  }
}
__device__ __forceinline__ void cp_signal(const CholP& c, int* p, int value) {  // one thread, behind a barrier that ordered the workgroup's stores
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __hip_atomic_store(p, c.base + value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef double CpBlk[NB][NB + 1];

// all threads of the 512-thread workgroup: dst = the rows x cols corner of the 32 x 32 block at src (row stride ld), zero beyond
__device__ __forceinline__ void cp_load(CpBlk dst, const double* __restrict__ src, int ld, int rows, int cols, int tid) {
#pragma unroll
  for (int h = 0; h < NB * NB / CP_THREADS; ++h) {
    const int i = (tid >> 5) + h * (CP_THREADS / NB), j = tid & 31;
    dst[i][j] = (i < rows && j < cols) ? src[(long)i * ld + j] : 0.0;
  }
}
// one wave: the same
__device__ __forceinline__ void cp_load_wave(CpBlk dst, const double* __restrict__ src, int ld, int rows, int cols, int lane) {
#pragma unroll
  for (int h = 0; h < NB * NB / WAVE; ++h) {
    const int e = h * WAVE + lane, i = e >> 5, j = e & 31;
    dst[i][j] = (i < rows && j < cols) ? src[(long)i * ld + j] : 0.0;
  }
}
// 16 x 16 tile (ti, tj) of A B^T, depth NB, operands row-major in LDS: four values per lane, value r belongs to row ti*16 + (lane >> 4) + 4 r,
// column tj*16 + (lane & 15)
__device__ __forceinline__ v4f64 cp_tile(const CpBlk A, const CpBlk B, int ti, int tj, int lane) {
  v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < NB / 4; ++t) {
    const int q = 4 * t + (lane >> 4);
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ti * 16 + (lane & 15)][q], B[tj * 16 + (lane & 15)][q], c, 0, 0, 0);
  }
  return c;
}
#define CP_TILE_ROW(ti, r, lane) ((ti) * 16 + ((lane) >> 4) + 4 * (r))
#define CP_TILE_COL(tj, lane) ((tj) * 16 + ((lane) & 15))

__device__ __forceinline__ void cp_wave_sync() {  // a wave's own LDS writes before its reads (DS instructions execute in order; the fences pin the compiler)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the chain workgroup ------------------------------------------------------------------------------------------------------------------------
struct CpChainLds {
  double D[2 * NB][NB + 1];   // chol_factor_block's block: L in rows 0 .. NB-1, column c of X = L^-1 in row NB + c
  CpBlk X;                    // X_k, row-major (operand of the products; the job of the next factorisation reads it)
  CpBlk Lcur;                 // L_k+1,k
  CpBlk Bn, Dn;               // blocks (k+1, k) and (k+1, k+1) with the panels 0 .. k-1 applied
  CpBlk scrA[2], scrL[2];     // scratch of the two job waves
  int abort_seen;
};

// waves 5 and 6 during the factorisation of D_k+1: row b = k + 2 from its mail (panels .. k-1 applied), X_k and L_k+1,k:
//   La = m0 X_k^T (= L_b,k);   wave 5: Bn = m1 - La L_k+1,k^T;   wave 6: Dn = m2 - La La^T
__device__ __forceinline__ void cp_chain_job(const CholP& c, CpChainLds& s, int b, int k, int wv, int lane) {
  const int which = wv - 5;  // 0: Bn, 1: Dn
  const int n = c.n, rcb = min(NB, n - b * NB), nbk_k = min(NB, n - k * NB), rc1 = min(NB, n - (k + 1) * NB);
  bool ok = true;
  if (lane == 0) ok = cp_wait(c, cp_Sn(c, b), 1);
  ok = __shfl(ok ? 1 : 0, 0, WAVE) != 0;
  if (!ok) { if (lane == 0) s.abort_seen = 1; return; }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (lane == 0 && which == 0) CP_STAMP(c, k, 4);  // the mail is there
  const double* mail = c.mail + (long)b * 3 * NB * NB;
  cp_load_wave(s.scrA[which], mail, NB, rcb, nbk_k, lane);  // m0: block (b, k)
  cp_wave_sync();
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const int ti = t4 >> 1, tj = t4 & 1;
    const v4f64 v = cp_tile(s.scrA[which], s.X, ti, tj, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = CP_TILE_ROW(ti, r, lane), j = CP_TILE_COL(tj, lane);
      s.scrL[which][i][j] = (i < rcb && j < nbk_k) ? v[r] : 0.0;
    }
  }
  cp_wave_sync();
  const double* msrc = mail + (long)(1 + which) * NB * NB;     // m1: block (b, k+1) / m2: block (b, b)
  const int cols = which == 0 ? rc1 : rcb;
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4) {
    const int ti = t4 >> 1, tj = t4 & 1;
    const v4f64 v = which == 0 ? cp_tile(s.scrL[0], s.Lcur, ti, tj, lane) : cp_tile(s.scrL[1], s.scrL[1], ti, tj, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = CP_TILE_ROW(ti, r, lane), j = CP_TILE_COL(tj, lane);
      const double old = (i < rcb && j < cols) ? msrc[i * NB + j] : 0.0;
      if (which == 0) s.Bn[i][j] = (i < rcb && j < cols) ? old - v[r] : 0.0;
      else s.Dn[i][j] = (i < rcb && j < cols) ? old - v[r] : 0.0;
    }
  }
  if (lane == 0) CP_STAMP(c, k, 5 + which);  // this wave's share of the job is done
}

__device__ __forceinline__ void cp_chain(const CholP& c, CpChainLds& s) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int n = c.n, ldw = c.ldw, nbk = c.nbk;
  if (tid == 0) s.abort_seen = 0;
  {  // prologue: D_0, and row 1 as it is (no panel to apply yet)
    const int rc0 = min(NB, n), rc1 = min(NB, n - NB);
#pragma unroll
    for (int h = 0; h < NB * NB / CP_THREADS; ++h) {
      const int i = (tid >> 5) + h * (CP_THREADS / NB), j = tid & 31;
      s.D[i][j] = (i < rc0 && j < rc0) ? c.W[(long)i * ldw + j] : (i == j ? 1.0 : 0.0);
    }
    cp_load(s.Bn, c.W + (long)NB * ldw, ldw, rc1, rc0, tid);
    cp_load(s.Dn, c.W + (long)NB * ldw + NB, ldw, rc1, rc1, tid);
    __syncthreads();
    if (wv == 0) chol_factor_block(s.D, rc0, c.flags);
    __syncthreads();
  }
  for (int k = 0; k < nbk; ++k) {
    const int r0 = k * NB, rck = min(NB, n - r0);
    if (tid == 0) CP_STAMP(c, k, 0);  // D_k is factored
    // X_k row-major for this workgroup's products (waves 0-6); L_kk, X_k and the diagonal block of T go to global memory for everybody else from WAVE 7
    // ALONE, which then publishes F = k + 1 by itself: no barrier of the workgroup waits for global stores (0.5-1 us each step on the chain before)
    if (wv < 7) {
      for (int e = tid; e < NB * NB; e += 7 * WAVE) { const int i = e >> 5, j = e & 31; s.X[i][j] = s.D[NB + j][i]; }
    } else {
      chol_factor_store(s.D, rck, c.Lm + (long)r0 * ldw + r0, ldw, c.Xinv + (long)k * NB * NB, lane, WAVE, c.Tinv + (long)r0 * ldw + r0);
    }
    __syncthreads();  // (wave 7 has read D; its stores are in flight)
    if (wv == 7) {
      __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have left
      if (lane == 0) { cp_signal(c, cp_F(c), k + 1); CP_STAMP(c, k, 1); }
    }
    if (k + 1 >= nbk) break;
    const int rc1 = min(NB, n - (k + 1) * NB);
    if (wv < 4) {  // L_k+1,k = Bn X_k^T
      const int ti = wv >> 1, tj = wv & 1;
      const v4f64 v = cp_tile(s.Bn, s.X, ti, tj, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = CP_TILE_ROW(ti, r, lane), j = CP_TILE_COL(tj, lane);
        s.Lcur[i][j] = (i < rc1 && j < rck) ? v[r] : 0.0;
      }
    }
    __syncthreads();
    if (wv < 4) {  // D_k+1 = Dn - L_k+1,k L_k+1,k^T, identity beyond the live rows
      const int ti = wv >> 1, tj = wv & 1;
      const v4f64 v = cp_tile(s.Lcur, s.Lcur, ti, tj, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = CP_TILE_ROW(ti, r, lane), j = CP_TILE_COL(tj, lane);
        s.D[i][j] = (i < rc1 && j < rc1) ? s.Dn[i][j] - v[r] : (i == j ? 1.0 : 0.0);
      }
    }
    __syncthreads();
    if (tid == 0) CP_STAMP(c, k, 2);  // the next diagonal block is in place
    if (wv == 0) { chol_factor_block(s.D, rc1, c.flags); if (lane == 0) CP_STAMP(c, k, 3); }
    else if ((wv == 5 || wv == 6) && k + 2 < nbk) cp_chain_job(c, s, k + 2, k, wv, lane);
    __syncthreads();
    if (s.abort_seen) return;
  }
}

// ---- tasks ----------------------------------------------------------------------------------------------------------------------------------------
struct CpTaskLds {
  CpBlk A, B, L0, L1, L2;
  int ok;
  unsigned long long ticket;
};

// wave 0 waits for the conditions of `w` (cp_wait_many), everybody learns the outcome; an acquire at agent scope follows
#define CP_WAIT_ALL(w)                                                          \
  do {                                                                          \
    if (wv == 0) { const bool ok_ = cp_wait_many(c, (w), lane); if (lane == 0) s.ok = ok_ ? 1 : 0; } \
    __syncthreads();                                                            \
    if (!s.ok) return false;                                                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                          \
  } while (0)

__device__ __forceinline__ bool cp_run_task(const CholP& c, CpTaskLds& s, const CholTask t) {
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int n = c.n, ldw = c.ldw, nbk = c.nbk;
  auto r0 = [&](int b) { return b < nbk ? b * NB : n; };                  // first row of row block b (nbk: the rhs row)
  auto rc = [&](int b) { return b < nbk ? min(NB, n - b * NB) : 1; };     // its live rows
  const int k = t.k, b = t.b, j = t.j;
  if (t.kind == CT_PANEL) {  // L_bk = U_bk X_k^T
    const CpWaits w{{cp_F(c), cp_Un(c, b, k)}, {k + 1, k}, 2, 0};
    CP_WAIT_ALL(w);
    cp_load(s.A, c.W + (long)r0(b) * ldw + k * NB, ldw, rc(b), rc(k), tid);
    cp_load(s.B, c.Xinv + (long)k * NB * NB, NB, NB, NB, tid);
    __syncthreads();
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
      const v4f64 v = cp_tile(s.A, s.B, ti, tj, lane);
      double* out = c.Lm + (long)r0(b) * ldw + k * NB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = CP_TILE_ROW(ti, r, lane), jj = CP_TILE_COL(tj, lane);
        if (i < rc(b) && jj < rc(k)) out[(long)i * ldw + jj] = v[r];
      }
    }
    __syncthreads();
    if (tid == 0) cp_signal(c, cp_Pn(c, b), k + 1);
    return true;
  }
  if (t.kind == CT_UPD) {  // block (b, j) -= L_bk L_jk^T
    // (the three blocks a FEED task reads with panel b-4 applied wait for it before panel b-3 goes in)
    const bool after_feed = b < nbk && k == b - 3 && j >= b - 2;
    const CpWaits w{{cp_Pn(c, b), cp_Pn(c, j), cp_Un(c, b, j), cp_Sn(c, b)}, {k + 1, k + 1, k, after_feed ? 1 : 0}, 4, 0};
    CP_WAIT_ALL(w);
    cp_load(s.A, c.Lm + (long)r0(b) * ldw + k * NB, ldw, rc(b), rc(k), tid);
    cp_load(s.B, c.Lm + (long)r0(j) * ldw + k * NB, ldw, rc(j), rc(k), tid);
    __syncthreads();
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
      const v4f64 v = cp_tile(s.A, s.B, ti, tj, lane);
      double* out = c.W + (long)r0(b) * ldw + j * NB;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = CP_TILE_ROW(ti, r, lane), jj = CP_TILE_COL(tj, lane);
        if (i < rc(b) && jj < rc(j)) out[(long)i * ldw + jj] -= v[r];
      }
    }
    __syncthreads();
    if (tid == 0) cp_signal(c, cp_Un(c, b, j), k + 1);
    return true;
  }
  if (t.kind == CT_TSTEP) {
    // T = L^-T (upper block triangle, block (j, i) = M_ij^T, M = L^-1):  acc_ji += T_jm L_im^T for m = j .. i-1, then T_ji = -acc_ji X_i^T.
    // This task: block (j, i = b), term m = k - 1; i == k: the last term, finalised with X_k (published by the chain workgroup at the end of step k).
    const int i = b, m = k - 1;
    const CpWaits w{{m == j ? cp_F(c) : cp_Tn(c, j, m), cp_Pn(c, i), cp_Tn(c, j, i)}, {m == j ? j + 1 : m - j + 1, m + 1, m - j}, 3, 1};
    CP_WAIT_ALL(w);
    const int rj = j * NB, ri = i * NB, rcj = rc(j), rci = rc(i), rcm = rc(m);
    cp_load(s.A, c.Tinv + (long)rj * ldw + m * NB, ldw, rcj, rcm, tid);
    cp_load(s.B, c.Lm + (long)ri * ldw + m * NB, ldw, rci, rcm, tid);
    __syncthreads();
    double* Tb = c.Tinv + (long)rj * ldw + ri;
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
      const v4f64 v = cp_tile(s.A, s.B, ti, tj, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = CP_TILE_ROW(ti, r, lane), jj = CP_TILE_COL(tj, lane);
        const bool live = ii < rcj && jj < rci;
        const double acc = (live ? ((m != j) ? Tb[(long)ii * ldw + jj] : 0.0) + v[r] : 0.0);
        if (i != k) { if (live) Tb[(long)ii * ldw + jj] = acc; }
        else s.L0[ii][jj] = acc;
      }
    }
    if (i == k) {
      const CpWaits wf{{cp_F(c)}, {k + 1}, 1, 0};
      CP_WAIT_ALL(wf);  // (the barrier inside also orders L0)
      cp_load(s.B, c.Xinv + (long)k * NB * NB, NB, NB, NB, tid);
      __syncthreads();
      if (wv < 4) {
        const int ti = wv >> 1, tj = wv & 1;
        const v4f64 v = cp_tile(s.L0, s.B, ti, tj, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = CP_TILE_ROW(ti, r, lane), jj = CP_TILE_COL(tj, lane);
          if (ii < rcj && jj < rci) Tb[(long)ii * ldw + jj] = -v[r];
        }
      }
    }
    __syncthreads();
    if (tid == 0) cp_signal(c, cp_Tn(c, j, i), m - j + 1 + (i == k ? 1 : 0));
    return true;
  }
  // CT_FEED: the mail of row b for the chain workgroup: blocks (b, b-2), (b, b-1), (b, b) with the panels 0 .. b-3 applied
  {
    const int k3 = b - 3;
    double* mail = c.mail + (long)b * 3 * NB * NB;
    const int rcb = rc(b);
    if (k3 < 0) {  // b == 2: nothing to apply
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int col = b - 2 + q;
#pragma unroll
        for (int h = 0; h < NB * NB / CP_THREADS; ++h) {
          const int i = (tid >> 5) + h * (CP_THREADS / NB), jj = tid & 31;
          mail[(long)q * NB * NB + i * NB + jj] = (i < rcb && jj < rc(col)) ? c.W[(long)(r0(b) + i) * ldw + col * NB + jj] : 0.0;
        }
      }
      __syncthreads();
      if (tid == 0) cp_signal(c, cp_Sn(c, b), 1);
      return true;
    }
    const CpWaits w{{cp_F(c), cp_Un(c, b, k3), cp_Un(c, b - 2, k3), cp_Un(c, b - 1, k3), cp_Un(c, b, b - 2), cp_Un(c, b, b - 1), cp_Un(c, b, b)},
                    {k3 + 1, k3, k3, k3, k3, k3, k3}, 7, 0};
    CP_WAIT_ALL(w);
    cp_load(s.B, c.Xinv + (long)k3 * NB * NB, NB, NB, NB, tid);
    // the three L blocks of panel k3 this row's update needs, recomputed here: rows b, b-2, b-1
    CpBlk* Ls[3] = {&s.L0, &s.L1, &s.L2};
    const int rows3[3] = {b, b - 2, b - 1};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int rb = rows3[q];
      __syncthreads();  // (the previous product has read A)
      cp_load(s.A, c.W + (long)r0(rb) * ldw + k3 * NB, ldw, rc(rb), NB, tid);
      __syncthreads();
      if (wv < 4) {
        const int ti = wv >> 1, tj = wv & 1;
        const v4f64 v = cp_tile(s.A, s.B, ti, tj, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = CP_TILE_ROW(ti, r, lane), jj = CP_TILE_COL(tj, lane);
          (*Ls[q])[ii][jj] = (ii < rc(rb)) ? v[r] : 0.0;
        }
      }
    }
    __syncthreads();
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
#pragma unroll
      for (int q = 0; q < 3; ++q) {  // m_q = block (b, b-2+q) - L_b,k3 L_col,k3^T
        const int col = b - 2 + q;
        const v4f64 v = q == 0 ? cp_tile(s.L0, s.L1, ti, tj, lane) : (q == 1 ? cp_tile(s.L0, s.L2, ti, tj, lane) : cp_tile(s.L0, s.L0, ti, tj, lane));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ii = CP_TILE_ROW(ti, r, lane), jj = CP_TILE_COL(tj, lane);
          const bool live = ii < rcb && jj < rc(col);
          mail[(long)q * NB * NB + ii * NB + jj] = live ? c.W[(long)(r0(b) + ii) * ldw + col * NB + jj] - v[r] : 0.0;
        }
      }
    }
    __syncthreads();
    if (tid == 0) cp_signal(c, cp_Sn(c, b), 1);
    return true;
  }
}

union CpLds {
  CpChainLds chain;
  CpTaskLds task;
};

__global__ void __launch_bounds__(CP_THREADS)
k_chol_persist(CholP c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cp_lds_raw[];
  CpLds& lds = *reinterpret_cast<CpLds*>(cp_lds_raw);
  if (blockIdx.x == 0) { cp_chain(c, lds.chain); return; }
  CpTaskLds& s = lds.task;
  const int tid = threadIdx.x;
  for (;;) {
    if (tid == 0) s.ticket = __hip_atomic_fetch_add(c.ticket, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - c.ticket_base;
    __syncthreads();
    const unsigned long long t = s.ticket;
    __syncthreads();
    if (t >= (unsigned long long)c.n_tasks) return;
    if (!cp_run_task(c, s, c.tasks[t])) return;  // aborted
    __syncthreads();
  }
}

// x = T y behind the persistent launch (k_chol_apply's body); an aborted launch (time-out) is reported as a failed factorisation: flags[2]
__global__ void __launch_bounds__(APPLY_THREADS)
k_chol_apply_checked(const double* __restrict__ Tinv, const double* __restrict__ Lm, int n, int ldw, double* __restrict__ out, const int* __restrict__ abort_word,
                     int* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) double y[];  // n
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0 && *abort_word != 0) flags[2] = 1;
  for (int i = tid; i < n; i += APPLY_THREADS) y[i] = Lm[(long)n * ldw + i];
  __syncthreads();
  const int r = tid >> 4, cl = tid & 15, row = blockIdx.x * NB + r;
  double a0 = 0.0, a1 = 0.0;
  if (row < n) {
    const double* Tr = Tinv + (long)row * ldw;
    int c = blockIdx.x * NB + cl;
    for (; c + 16 < n; c += 32) { a0 = fma(Tr[c], y[c], a0); a1 = fma(Tr[c + 16], y[c + 16], a1); }
    if (c < n) a0 = fma(Tr[c], y[c], a0);
  }
  double s = a0 + a1;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
  if (cl == 0 && row < n) out[row] = s;
}

// ---- host: the task list in dependency order -----------------------------------------------------------------------------------------------------
// Step k holds: the FEED tasks whose last input is X_k (row k + 3; rows 2 and 3 at step 0), the panel solves of column k (rows k+1 .. nbk: nbk is the
// rhs row), the updates with panel k — blocks near the diagonal first —, and the T-steps of launch k of the old scheme (term m = k - 1).
// The diagonal block (b, b) takes panels 0 .. b-4 only: FEED(b) applies panel b-3 for the chain workgroup, which applies the last two itself.
inline std::vector<CholTask> chol_persist_tasks(int nbk) {
  std::vector<CholTask> out;
  auto push = [&](int kind, int k, int b, int j) { out.push_back(CholTask{(unsigned char)kind, 0, (unsigned short)k, (unsigned short)b, (unsigned short)j}); };
  for (int k = 0; k < nbk; ++k) {
    if (k == 0 && 2 <= nbk - 1) push(CT_FEED, 0, 2, 0);
    if (k + 3 <= nbk - 1) push(CT_FEED, k, k + 3, 0);
    for (int b = k + 1; b <= nbk; ++b) push(CT_PANEL, k, b, 0);
    for (int d = 1; k + d <= nbk - 1; ++d) {  // column j = k + d of the update, rows j .. nbk
      const int j = k + d;
      for (int b = j; b <= nbk; ++b) {
        if (b == j && k > b - 4) continue;  // (the diagonal block's last three panels are not applied here)
        push(CT_UPD, k, b, j);
      }
    }
    if (k >= 1)
      for (int i = k; i <= nbk - 1; ++i)
        for (int j = 0; j < k; ++j) push(CT_TSTEP, k, i, j);
  }
  return out;
}

}  // namespace cba
