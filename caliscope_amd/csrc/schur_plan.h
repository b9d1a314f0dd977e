// Static plan of the register-accumulating Schur kernel (k_schur_reg2), host side only: plain C++, no HIP, so that the
// CPU suite can build it with g++ and replay it against a direct sum (tests/native/plan_harness.cpp).
//
// The kernel (cba_kernels.h) binds a 256-thread workgroup to one tile (camera group a x camera group b) of the reduced
// camera system.  Every thread owns one camera-pair block of the tile and keeps its nc x nc accumulators in registers.  A
// tile's work is a stream of T records (one per observation whose camera is in group a or b), cut into CHUNKS that fit
// LDS; per chunk every thread multiplies the record pairs (T_i, T_j) that belong to its block.  A wave runs at the pace
// of its busiest lane, so what the plan decides is the time the pass takes: the chunks must hold the same number of
// pairs for every block.
//
// Round 1 filled one open chunk greedily from a window of 32 points (48 % of the lane-iterations did work on cfg4).
// Here a REGION of ~128 chunks is open at once and the points of the region are DEALT: heaviest first, each to the
// first chunk (first fit) that has room for its records and where none of its blocks has reached the cap `t` (pairs per
// thread per chunk); what fits nowhere is dealt again with t + 1.  With 384 records per chunk a thread sees 1.8 pairs
// per chunk on cfg4 and the cap is 2: 82 % of the lane-iterations do work (tools/plan_sim.py).
//
// The pair list leaves the plan TRANSPOSED: for chunk c and wave w, nit[c].w iterations of 64 codes each, code =
// (i_addr | j_addr << 16) of the pair lane `l` multiplies in that iteration (LDS addresses of the two records in 16-byte
// units), or ZERO (both halves point at an all-zero
// record kept behind the chunk in LDS) when the lane has nothing left.  The kernel's pair loop is therefore branch-free
// with a wave-uniform trip count, and a lane fetches its code with one coalesced load.
//
// LDS bank conflicts.  A lane reads its two records with ds_read_b128; the LDS serves a wave's b128 read in four fixed
// groups of 16 lanes, one cycle per group when the 16 addresses fall into 16 different 16-byte bank groups
// ((address / 16) mod 16, /opt/skills/guides/MI355X_MICROARCH.md, LDS).  A record starts at slot * stride with an odd
// stride in 16-byte units, so the bank group of piece m is (m + c * slot) mod 16 with c odd: what matters is the slot's
// residue mod 16.  With the records of a chunk in arrival order the residues of the 16 records a lane group reads are
// random (3.1 cycles per group instead of 1: the pair loop was LDS-bound on exactly that).  The order of the records
// inside a chunk is free, so the plan COLOURS them: every (iteration, lane group, operand) is a clique of up to 16
// records that want 16 different residues; a greedy pass plus two refinement sweeps assign residues, then slots.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <type_traits>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace cba {

// An array without value-initialisation: the worker threads that fill it are the first to touch its pages (a std::vector of the plan's 40+ MB would
// be zero-filled, page by page, by the one thread that resizes it).  Large arrays (>= 4 MB) come from a process-wide pool of 2 MB-aligned blocks
// advised for transparent huge pages, and go back to it: measured on the MI355X boxes (CBA_PLAN_TIMING, round 4), half of the cheap plan's wall time
// was the CONCATENATION of the jobs' arrays into freshly mapped memory — 5.4 of 12 ms for cfg4's 160 MB, 104 of 190 ms for cfg5's 946 MB: page
// faults of 64 threads, not copying.  The pool keeps at most kKeepBlocks blocks / kKeepBytes (what a process that made a cfg5-sized plan holds on
// to until it exits); a request takes the smallest kept block that fits and is at most twice as large.
namespace rawvec_detail {
constexpr size_t kPooledFrom = (size_t)4 << 20, kHuge = (size_t)2 << 20, kKeepBytes = (size_t)3 << 30;
constexpr int kKeepBlocks = 32;
struct Block { void* ptr; size_t bytes; };
inline std::mutex& pool_mu() { static std::mutex m; return m; }
inline std::vector<Block>& pool() { static std::vector<Block> v; return v; }
inline void* acquire(size_t bytes, size_t* capacity) {
  if (bytes < kPooledFrom) { *capacity = bytes; return std::malloc(std::max<size_t>(bytes, 1)); }
  {
    std::lock_guard<std::mutex> lock(pool_mu());
    std::vector<Block>& v = pool();
    int best = -1;
    for (int i = 0; i < (int)v.size(); ++i)
      if (v[i].bytes >= bytes && v[i].bytes <= 2 * bytes && (best < 0 || v[i].bytes < v[best].bytes)) best = i;
    if (best >= 0) {
      const Block b = v[best];
      v.erase(v.begin() + best);
      *capacity = b.bytes;
      return b.ptr;
    }
  }
  const size_t cap = (bytes + kHuge - 1) / kHuge * kHuge;
  void* ptr = nullptr;
  if (posix_memalign(&ptr, kHuge, cap) != 0) return nullptr;
  (void)madvise(ptr, cap, MADV_HUGEPAGE);
  *capacity = cap;
  return ptr;
}
inline void release(void* ptr, size_t capacity) {
  if (!ptr) return;
  if (capacity >= kPooledFrom) {
    std::lock_guard<std::mutex> lock(pool_mu());
    std::vector<Block>& v = pool();
    size_t kept = 0;
    for (const Block& b : v) kept += b.bytes;
    if ((int)v.size() < kKeepBlocks && kept + capacity <= kKeepBytes) { v.push_back(Block{ptr, capacity}); return; }
  }
  std::free(ptr);
}
// everything the pool keeps goes back to the allocator (cba_trim); returns the bytes released
inline size_t trim() {
  std::vector<Block> v;
  { std::lock_guard<std::mutex> lock(pool_mu()); v.swap(pool()); }
  size_t bytes = 0;
  for (const Block& b : v) { bytes += b.bytes; std::free(b.ptr); }
  return bytes;
}
// for containers that hand back (pointer, requested bytes) only — the std::vector allocator of cba_create's observation-sized host arrays: the
// capacity of a pooled block is remembered beside the pool
inline std::map<void*, size_t>& tracked() { static std::map<void*, size_t> m; return m; }
inline void* acquire_tracked(size_t bytes) {
  size_t cap = 0;
  void* ptr = acquire(bytes, &cap);
  if (ptr && cap >= kPooledFrom) { std::lock_guard<std::mutex> lock(pool_mu()); tracked()[ptr] = cap; }
  return ptr;
}
inline void release_tracked(void* ptr, size_t bytes) {
  size_t cap = bytes;
  if (ptr && bytes >= kPooledFrom) {
    std::lock_guard<std::mutex> lock(pool_mu());
    auto it = tracked().find(ptr);
    if (it != tracked().end()) { cap = it->second; tracked().erase(it); }
  }
  release(ptr, cap);
}
}  // namespace rawvec_detail

template <typename T>
struct RawVec {
  static_assert(std::is_trivially_destructible<T>::value && std::is_trivially_default_constructible<T>::value, "plain data only");
  T* p = nullptr;
  size_t n = 0, capacity_bytes = 0;
  RawVec() = default;
  RawVec(const RawVec&) = delete;
  RawVec& operator=(const RawVec&) = delete;
  RawVec(RawVec&& o) noexcept : p(o.p), n(o.n), capacity_bytes(o.capacity_bytes) { o.p = nullptr; o.n = 0; o.capacity_bytes = 0; }
  RawVec& operator=(RawVec&& o) noexcept {
    if (this != &o) { rawvec_detail::release(p, capacity_bytes); p = o.p; n = o.n; capacity_bytes = o.capacity_bytes; o.p = nullptr; o.n = 0; o.capacity_bytes = 0; }
    return *this;
  }
  ~RawVec() { rawvec_detail::release(p, capacity_bytes); }
  void resize_uninit(size_t m) {
    rawvec_detail::release(p, capacity_bytes);
    p = static_cast<T*>(rawvec_detail::acquire(m * sizeof(T), &capacity_bytes));
    if (!p && m) throw std::bad_alloc();
    n = m;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
  T* begin() { return p; }
  T* end() { return p + n; }
};

// CPUs this process may actually burn: the hardware count capped by the cgroup's CPU quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us).  A container
// that shows 256 cores and is throttled to 16 CPUs of quota (the MI355X boxes of rounds 1-4) freezes EVERY thread of the process — the one driving the
// GPU included — for the rest of the 100 ms period once 64 plan threads have used the quota up: measured, a cfg4 solve of 3 ms took 60 ms while the
// dealt plan was being made in the background.
inline unsigned usable_cpus() {
  static const unsigned n = [] {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    auto read = [](const char* path, long long* a, long long* b) {
      FILE* f = std::fopen(path, "r");
      if (!f) return 0;
      char x[64] = {0}, y[64] = {0};
      const int got = std::fscanf(f, "%63s %63s", x, y);
      std::fclose(f);
      if (got >= 1) *a = std::strcmp(x, "max") == 0 ? -1 : std::atoll(x);
      if (got >= 2) *b = std::atoll(y);
      return got;
    };
    long long quota = -1, period = 100000;
    if (read("/sys/fs/cgroup/cpu.max", &quota, &period) < 1) {
      long long unused = 0;
      if (read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &quota, &unused) >= 1) (void)read("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &period, &unused);
    }
    if (quota > 0 && period > 0) hw = (unsigned)std::max<long long>(1, std::min<long long>(hw, quota / period));
    return hw;
  }();
  return n;
}

struct Reg2Params {
  int C = 0, P = 0;       // cameras, world points
  int G = 1, g = 1;       // camera groups, cameras per group (max)
  int rep = 1;            // threads per camera-pair block (small groups: 256 / g^2)
  int chunk_cap = 384;    // records (slots) per chunk
  // LDS layout of a staged chunk, in 16-byte pieces: slot s lives at (s / slots_per_wave) * wave_pieces + (s % slots_per_wave) *
  // rec_pieces (every wave of the kernel stages its own run of slots, padded to whole load instructions); the codes carry
  // these piece addresses, so the pair loop does no index arithmetic.  slots_per_wave must be a multiple of 16 and
  // rec_pieces odd (then the bank group of a record is decided by slot mod 16, see below).
  int slots_per_wave = 96, wave_pieces = 896, rec_pieces = 9;
  int zero_piece = 0;     // piece address of the all-zero record
  int region_chunks = 32;   // chunks dealt together: larger regions pack better (lane utilisation 0.77 at 128, 0.71 at 32 on cfg4) but the tiles that
                            // gather one record then do so further apart in time: at 32 half of those gathers hit the L2 (PMC FETCH_SIZE 0.59 vs 0.90 GB)
  int heavy_obs = 0;      // > 0: points with more observations are left out (k_heavy_schur forms their share)
  int threads = 0;        // host threads (0: hardware concurrency)
  int pair_cap = 0;       // > 0: pairs of one block per chunk the kernel takes without an extra load; the dealing opens new chunks rather than going beyond
                          // (a single point with more pairs of one block than this still gets its chunk)
  int n_waves = 4;        // waves of a workgroup that multiply pairs (4: 256 threads, 16: the 1024-thread kernel of 32 x 32 tiles)
  bool cheap = false;     // the plan a handle starts with while the dealt one is being made: one open chunk filled in point order (no dealing), the
                          // records in arrival order (no colouring) — a quarter of the host time; the pair kernel then runs at ~0.5 lane utilisation
                          // and ~2.4 LDS cycles per read group instead of 0.7 / 1.4
  bool cheap_lean = true; // (tests: false runs the cheap plan through the general path, which must produce the same arrays)
  const std::atomic<bool>* cancel = nullptr;  // set by the owner to make the workers stop between jobs (build_reg2_plan then returns -2)
  int colour_sweeps = 0;  // refinement sweeps of the slot colouring behind the greedy pass.  Two sweeps (round 2) buy 1.35 instead of 1.36 LDS cycles
                          // per 16-lane read group on cfg4 and cost a quarter of the plan's time: off
};

struct Reg2Plan {
  RawVec<int> obs;                    // stream entry -> observation (index into the T records); padded with 2 * chunk_cap zeros
  std::vector<int> chunk_start;       // [n_chunks + 1] offsets into obs
  std::vector<int> code_start;        // [n_chunks + 1] offsets into codes
  std::vector<unsigned> nit;          // [n_chunks][n_waves / 4] iterations of the waves, one byte each
  RawVec<unsigned> codes;             // transposed pair codes; padded with 4 * 64 n_waves ZERO codes
  std::vector<int> tile_chunk_begin;  // [n_tiles + 1]
  long n_pairs = 0;                   // pair codes that do work
  long lane_iters = 0;                // 64 x wave-iterations (n_pairs / lane_iters = lane utilisation)
  long lds_groups = 0;                // (iteration, lane group, operand) read groups
  long lds_cycles = 0;                // LDS cycles they take with the plan's slots (1 per group when conflict-free)
  long lds_cycles_arrival = 0;        // ... and with the records in arrival order, for comparison
  int n_regions = 1;
  double seconds_runs = 0.0, seconds_jobs = 0.0, seconds_concat = 0.0;  // wall time of the three phases of build_reg2_plan (CBA_PLAN_TIMING prints them)
};

namespace reg2_detail {

struct Cand { int q; int n_rec; int key_begin, key_end; };

struct Job {
  int tile = 0, q_begin = 0, q_end = 0;
  std::vector<int> obs, chunk_start, code_start;
  std::vector<unsigned> nit, codes;
  long n_pairs = 0, lane_iters = 0, lds_groups = 0, lds_cycles = 0, lds_cycles_arrival = 0;
  int rc = 0;
};

// lane groups of ds_read_b128 (wave64): group of lane l
inline int b128_group(int lane) {
  const int l = lane & 31, hi = (lane >> 5) * 2;
  const bool g0 = (l < 4) || (l >= 12 && l < 16) || (l >= 20 && l < 28);
  return hi + (g0 ? 0 : 1);
}

}  // namespace reg2_detail

// hcam / hps: cameras of the observations sorted by (point, camera) and the first observation of every point.
// Returns 0, or -1 when a single point does not fit a chunk.
inline int build_reg2_plan(const Reg2Params& prm, const int* hcam, const int* hps, Reg2Plan& out) {
  using namespace reg2_detail;
  const auto t_begin = std::chrono::steady_clock::now();
  const int G = prm.G, g = prm.g, C = prm.C, P = prm.P, rep = std::max(1, prm.rep);
  const int nT = G * (G + 1) / 2, nblk = g * g, R = prm.chunk_cap;
  const int NW = std::max(4, prm.n_waves / 4 * 4), NWORD = NW / 4;
  const unsigned ZERO = (unsigned)prm.zero_piece | ((unsigned)prm.zero_piece << 16);
  auto piece_of = [&](int slot) { return (unsigned)((slot / prm.slots_per_wave) * prm.wave_pieces + (slot % prm.slots_per_wave) * prm.rec_pieces); };
  std::vector<int> gcam(G + 1);
  for (int a = 0; a <= G; ++a) gcam[a] = std::min(a * g, C);
  std::vector<int> ta(nT), tb(nT);
  {
    int t = 0;
    for (int a = 0; a < G; ++a)
      for (int b = a; b < G; ++b, ++t) { ta[t] = a; tb[t] = b; }
  }
  // (more than ~64 workers do not pay: a job is a few milliseconds, and starting a thread costs the main thread ~20 us)
  // (foreground work may burst above a cgroup quota: 64 threads finish the plan of a 1M-observation problem in 15 ms, inside one accounting period)
  // Default cap (round 5): 16 workers below two million observations — the reference runs inside a desktop GUI process on a workstation, and 64
  // threads bought a 1M-observation plan 3 ms of a 11 ms set-up; from 2M observations on (cfg4: 9-12 ms of wall on 64 threads) the burst stays.
  const long n_obs_plan = (long)hps[P];
  const unsigned n_threads = prm.threads > 0 ? (unsigned)prm.threads
                                             : std::min(n_obs_plan >= 2000000 ? 64u : 16u, std::max(1u, std::thread::hardware_concurrency()));
  // observations of a point are sorted by camera => by group; pgb[q*(G+1) + a] .. [a+1] is group a's run
  RawVec<int> pgb;
  pgb.resize_uninit((size_t)P * (G + 1));
  long stream_total = 0;
  {
    auto runs = [&](int q0, int q1, long* total) {
      long tot = 0;
      for (int q = q0; q < q1; ++q) {
        int cur = hps[q];
        const int s1 = hps[q + 1];
        for (int a = 0; a < G; ++a) {
          pgb[(size_t)q * (G + 1) + a] = cur;
          while (cur < s1 && hcam[cur] < gcam[a + 1]) ++cur;
        }
        pgb[(size_t)q * (G + 1) + G] = s1;
        tot += (long)(s1 - hps[q]) * G;  // an observation takes part in the G tiles of its group (upper bound)
      }
      *total = tot;
    };
    const int nth = (int)std::max(1u, std::min(n_threads, (unsigned)(P / 8192)));  // (a thread per 8k points at least: below that starting it costs more)
    std::vector<long> part((size_t)nth, 0);
    std::vector<std::thread> pool;
    for (int t = 1; t < nth; ++t) pool.emplace_back(runs, (int)((long)P * t / nth), (int)((long)P * (t + 1) / nth), &part[t]);
    runs(0, (int)((long)P / nth), &part[0]);
    for (auto& th : pool) th.join();
    for (long v : part) stream_total += v;
  }
  const auto t_phase0 = std::chrono::steady_clock::now();
  // regions: the same point ranges for every tile, about region_chunks chunks of an average tile each
  const long per_tile = std::max<long>(1, stream_total / std::max(nT, 1));
  int n_regions = (int)std::max<long>(1, (per_tile + (long)R * prm.region_chunks / 2) / ((long)R * std::max(prm.region_chunks, 1)));
  n_regions = std::min(n_regions, std::max(1, P));
  out.n_regions = n_regions;

  std::vector<Job> jobs((size_t)nT * n_regions);
  for (int t = 0; t < nT; ++t)
    for (int r = 0; r < n_regions; ++r) {
      Job& j = jobs[(size_t)t * n_regions + r];
      j.tile = t;
      j.q_begin = (int)((long)P * r / n_regions);
      j.q_end = (int)((long)P * (r + 1) / n_regions);
    }

  // The cheap plan (one open chunk filled in point order, records in arrival order) needs none of the dealing's bookkeeping: one pass over the job's
  // points, the pair codes written in their final form (LDS piece addresses from a table) into fixed-size per-thread lists, a transposing copy per
  // chunk.  Same plan as the general path below produces with prm.cheap, bit for bit (tests/test_schur_plan.py compares digests), for 0.55 of its
  // CPU time — the cheap plan is what a two-stage handle waits for in cba_create.  A chunk is also closed when a thread's list is half full
  // (never on real data: ~1 pair per block and chunk); a list that would overflow hands the job to the general path.
  constexpr int LCAP = 255;                  // (a byte per wave in `nit`)
  std::vector<unsigned> piece_tab((size_t)R);
  for (int s = 0; s < R; ++s) piece_tab[(size_t)s] = piece_of(s);
  auto run_job_cheap = [&](Job& job) -> bool {
    const int a = ta[job.tile], b = tb[job.tile];
    const bool diag = (a == b);
    const int na_t = gcam[a + 1] - gcam[a];
    std::vector<std::vector<int>> helpers;
    if (diag) {
      helpers.assign(g, {});
      int k = 0;
      for (int li = 0; li < g; ++li)
        for (int lj = 0; lj <= li; ++lj, ++k) helpers[k % std::max(na_t, 1)].push_back(li * g + lj);
    }
    long tot_rec = 0, tot_pairs = 0;
    for (int q = job.q_begin; q < job.q_end; ++q) {  // sizes for the two reservations (upper bounds)
      const int* gb = &pgb[(size_t)q * (G + 1)];
      const int na = gb[a + 1] - gb[a], nb = diag ? 0 : gb[b + 1] - gb[b];
      if (prm.heavy_obs > 0 && hps[q + 1] - hps[q] > prm.heavy_obs) continue;
      if (na <= 0 || (!diag && nb <= 0)) continue;
      if (na + nb > R) { job.rc = -1; return true; }
      tot_rec += na + nb;
      tot_pairs += diag ? (long)na * (na + 1) / 2 : (long)na * nb;
    }
    job.chunk_start.push_back(0);
    job.code_start.push_back(0);
    if (!tot_rec) return true;
    job.obs.reserve((size_t)(tot_rec + tot_rec / 8) + 64);
    job.codes.reserve((size_t)((double)tot_pairs * 3.6) + 256 * (size_t)NW);
    const int NL = NW * 64;
    std::vector<unsigned> lst((size_t)NL * LCAP);
    std::vector<unsigned short> cnt((size_t)NL, 0);
    std::vector<unsigned> rr((size_t)nblk, 0), rrc((size_t)g, 0), packed((size_t)NWORD, 0u);
    int fill = 0, maxcnt = 0;
    bool overflow = false;
    auto flush = [&]() {
      if (!fill) return;
      std::fill(packed.begin(), packed.end(), 0u);
      for (int w = 0; w < NW; ++w) {
        int mx = 0;
        for (int l = 0; l < 64; ++l) mx = std::max<int>(mx, cnt[(size_t)w * 64 + l]);
        packed[w / 4] |= (unsigned)mx << (8 * (w % 4));
        const size_t at = job.codes.size();
        job.codes.resize(at + (size_t)mx * 64);
        unsigned* dst = job.codes.data() + at;
        for (int l = 0; l < 64; ++l) {
          const unsigned* src = &lst[((size_t)w * 64 + l) * LCAP];
          const int n = cnt[(size_t)w * 64 + l];
          for (int it = 0; it < n; ++it) dst[(size_t)it * 64 + l] = src[it];
          for (int it = n; it < mx; ++it) dst[(size_t)it * 64 + l] = ZERO;
          job.n_pairs += n;
        }
        job.lane_iters += (long)mx * 64;
      }
      job.nit.insert(job.nit.end(), packed.begin(), packed.end());
      job.chunk_start.push_back((int)job.obs.size());
      job.code_start.push_back((int)job.codes.size());
      std::fill(cnt.begin(), cnt.end(), (unsigned short)0);
      std::fill(rr.begin(), rr.end(), 0u);
      std::fill(rrc.begin(), rrc.end(), 0u);
      fill = 0; maxcnt = 0;
    };
    auto emit = [&](int blk, unsigned code) {
      const int slot = (int)(rr[blk]++ % (unsigned)rep);
      const size_t li = (size_t)slot * nblk + blk;
      if (cnt[li] >= LCAP) { overflow = true; return; }
      lst[li * LCAP + cnt[li]] = code;
      maxcnt = std::max<int>(maxcnt, ++cnt[li]);
    };
    for (int q = job.q_begin; q < job.q_end && !overflow; ++q) {
      const int* gb = &pgb[(size_t)q * (G + 1)];
      const int na = gb[a + 1] - gb[a], nb = diag ? 0 : gb[b + 1] - gb[b];
      if (prm.heavy_obs > 0 && hps[q + 1] - hps[q] > prm.heavy_obs) continue;
      if (na <= 0 || (!diag && nb <= 0)) continue;
      if (fill + na + nb > R || maxcnt > LCAP / 2) flush();
      const int base = fill;
      for (int i = gb[a]; i < gb[a] + na; ++i) job.obs.push_back(i);
      for (int i = gb[b]; !diag && i < gb[b] + nb; ++i) job.obs.push_back(i);
      fill += na + nb;
      for (int i = 0; i < na; ++i) {
        const int li = hcam[gb[a] + i] - gcam[a];
        const unsigned pi = piece_tab[(size_t)(base + i)];
        const int j0 = diag ? i : na, j1 = diag ? na : na + nb;
        for (int j = j0; j < j1; ++j) {
          const int lj = diag ? hcam[gb[a] + j] - gcam[a] : hcam[gb[b] + (j - na)] - gcam[b];
          const unsigned pj = piece_tab[(size_t)(base + j)];
          if (diag && lj == li) {
            const std::vector<int>& h = helpers[li];
            emit(h[rrc[li]++ % h.size()], pi | (pj << 16));
            if (j != i) emit(h[rrc[li]++ % h.size()], pj | (pi << 16));
          } else {
            emit(li * g + lj, pi | (pj << 16));
          }
        }
      }
    }
    if (overflow) return false;
    flush();
    return true;
  };

  auto run_job = [&](Job& job) {
    if (prm.cheap && prm.cheap_lean) {
      if (run_job_cheap(job)) return;
      job = Job{job.tile, job.q_begin, job.q_end};  // (a thread's list overflowed: the general path has no such limit below 255 pairs per wave-iteration byte)
    }
    const int a = ta[job.tile], b = tb[job.tile];
    const bool diag = (a == b);
    const int na_t = gcam[a + 1] - gcam[a];
    const int min_rec = diag ? 1 : 2;  // records of the smallest point of this tile
    // key space of the load counters: [0, nblk) real camera-pair blocks, [nblk, nblk + g) the (i, i) items of a camera
    // (diagonal tiles), which its helper threads share
    std::vector<std::vector<int>> helpers;
    if (diag) {
      helpers.assign(g, {});
      int k = 0;
      for (int li = 0; li < g; ++li)
        for (int lj = 0; lj <= li; ++lj, ++k) helpers[k % std::max(na_t, 1)].push_back(li * g + lj);
    }
    std::vector<Cand> cand;
    std::vector<unsigned short> keys;
    long tot_rec = 0, tot_pairs = 0;
    std::vector<long> key_tot((size_t)nblk + g, 0);
    for (int q = job.q_begin; q < job.q_end; ++q) {
      const int* gb = &pgb[(size_t)q * (G + 1)];
      const int na = gb[a + 1] - gb[a], nb = diag ? 0 : gb[b + 1] - gb[b];
      if (prm.heavy_obs > 0 && hps[q + 1] - hps[q] > prm.heavy_obs) continue;
      if (na <= 0 || (!diag && nb <= 0)) continue;
      if (na + nb > R) { job.rc = -1; return; }
      Cand c{q, na + nb, (int)keys.size(), 0};
      for (int i = gb[a]; i < gb[a] + na; ++i) {
        const int li = hcam[i] - gcam[a];
        const int j0 = diag ? i : gb[b], j1 = diag ? gb[a] + na : gb[b] + nb;
        for (int j = j0; j < j1; ++j) {
          const int lj = hcam[j] - gcam[b];
          if (diag && lj == li) {  // (i, i) item, or two rows of one camera: T_i T_j^T + T_j T_i^T, two codes
            keys.push_back((unsigned short)(nblk + li));
            if (j != i) keys.push_back((unsigned short)(nblk + li));
          } else {
            keys.push_back((unsigned short)(li * g + lj));
          }
        }
      }
      c.key_end = (int)keys.size();
      for (int k = c.key_begin; k < c.key_end; ++k) key_tot[keys[k]]++;
      tot_rec += c.n_rec;
      tot_pairs += c.key_end - c.key_begin;
      cand.push_back(c);
    }
    job.chunk_start.push_back(0);
    job.code_start.push_back(0);
    if (cand.empty()) return;
    // one allocation each for what the job emits (growing them by doubling re-maps tens of megabytes under all the worker threads at once)
    job.obs.reserve((size_t)(tot_rec + tot_rec / 8) + 64);
    job.codes.reserve((size_t)((double)tot_pairs * (prm.cheap ? 3.6 : 1.7)) + 256 * (size_t)NW);  // (1 / lane utilisation: ~0.7 dealt, ~0.35 in point order)
    // capacity of a key per unit of the cap t: rep thread slots per block, all helper slots of a camera
    std::vector<int> unit((size_t)nblk + g, rep);
    int active_slots = 0;
    for (int k = 0; k < nblk + g; ++k) {
      if (k >= nblk) unit[k] = diag ? rep * (int)helpers[k - nblk].size() : 0;
      if (key_tot[k] > 0) active_slots += unit[k];
    }
    active_slots = std::max(active_slots, 1);
    // heaviest first (counting sort by pair count, stable in point order)
    std::vector<int> order(cand.size());
    {
      int maxp = 0;
      for (const Cand& c : cand) maxp = std::max(maxp, c.key_end - c.key_begin);
      std::vector<int> start((size_t)maxp + 2, 0);
      for (const Cand& c : cand) start[(size_t)maxp - (c.key_end - c.key_begin) + 1]++;
      for (int i = 0; i <= maxp; ++i) start[i + 1] += start[i];
      for (int i = 0; i < (int)cand.size(); ++i) order[start[(size_t)maxp - (cand[i].key_end - cand[i].key_begin)]++] = i;
    }

    struct Deal {
      std::vector<std::vector<int>> members;  // candidates of every chunk, in placement order
      long cost = 0;
    };
    const double mean = (double)tot_pairs / active_slots / std::max(1.0, (double)tot_rec / R);  // pairs per thread slot per full chunk
    // per candidate the set of its keys as a bit mask, per chunk the set of keys that have reached the cap: a chunk where the two intersect cannot take
    // the candidate — one AND per 64 keys instead of a walk over the candidate's keys with increments to take back (most first-fit probes fail)
    const int W = nblk + g, MW = (W + 63) / 64;
    std::vector<unsigned long long> cand_mask(prm.cheap ? 0 : cand.size() * (size_t)MW, 0ull);
    for (size_t ci = 0; ci < cand.size() && !prm.cheap; ++ci)
      for (int k = cand[ci].key_begin; k < cand[ci].key_end; ++k) cand_mask[ci * MW + keys[k] / 64] |= 1ull << (keys[k] % 64);
    auto deal = [&](int t0, int r_eff, Deal& d) {
      const int want = (int)((tot_rec + r_eff - 1) / r_eff);
      const int max_chunks = want + std::max(2, want / 12);
      std::vector<unsigned short> cnt;  // [chunk][nblk + g]
      std::vector<unsigned long long> atcap;  // [chunk][MW]
      std::vector<int> fill;
      auto open_chunk = [&]() { cnt.resize(cnt.size() + W, 0); atcap.resize(atcap.size() + MW, 0ull); fill.push_back(0); d.members.emplace_back(); };
      std::vector<int> todo(order), left;
      for (int t = t0; !todo.empty(); ++t) {
        if (t > t0)  // a higher cap: what was full may have room again
          for (size_t ch = 0; ch < fill.size(); ++ch) {
            unsigned long long* am = &atcap[ch * MW];
            const unsigned short* cc = &cnt[ch * W];
            for (int w = 0; w < MW; ++w) am[w] = 0ull;
            for (int k = 0; k < W; ++k)
              if (cc[k] >= t * unit[k]) am[k / 64] |= 1ull << (k % 64);
          }
        left.clear();
        int first = 0;
        for (int ci : todo) {
          const Cand& c = cand[ci];
          const unsigned long long* cm = &cand_mask[(size_t)ci * MW];
          while (first < (int)fill.size() && fill[first] + min_rec > r_eff) ++first;
          bool placed = false;
          for (int ch = first; ch <= (int)fill.size() && !placed; ++ch) {
            if (ch == (int)fill.size()) {
              if (ch >= max_chunks && t < t0 + 6 && (prm.pair_cap <= 0 || t < prm.pair_cap)) break;  // the region is full at this cap: try again with t + 1
              open_chunk();
              if (t > 0) {  // (keys without capacity — unit 0 — are at the cap from the start)
                unsigned long long* am = &atcap[(size_t)ch * MW];
                for (int k = 0; k < W; ++k)
                  if (t * unit[k] <= 0) am[k / 64] |= 1ull << (k % 64);
              }
            }
            if (fill[ch] + c.n_rec > r_eff) continue;
            const bool fresh = fill[ch] == 0 && ch + 1 == (int)fill.size();
            {
              const unsigned long long* am = &atcap[(size_t)ch * MW];
              unsigned long long hit = 0ull;
              for (int w = 0; w < MW; ++w) hit |= am[w] & cm[w];
              if (hit) {
                if (fresh) break;  // does not even fit an empty chunk at this cap
                continue;
              }
            }
            unsigned short* cc = &cnt[(size_t)ch * W];
            int k = c.key_begin;
            for (; k < c.key_end; ++k)
              if (++cc[keys[k]] > t * unit[keys[k]]) break;
            if (k < c.key_end) {  // over the cap (a key the candidate holds more than once): take the increments back
              for (int k2 = c.key_begin; k2 <= k; ++k2) --cc[keys[k2]];
              if (fresh) break;  // does not even fit an empty chunk at this cap
              continue;
            }
            unsigned long long* am = &atcap[(size_t)ch * MW];
            for (k = c.key_begin; k < c.key_end; ++k)
              if (cc[keys[k]] >= t * unit[keys[k]]) am[keys[k] / 64] |= 1ull << (keys[k] % 64);
            fill[ch] += c.n_rec;
            d.members[ch].push_back(ci);
            placed = true;
          }
          if (!placed) left.push_back(ci);
        }
        todo.swap(left);
      }
      // cost model: wave-iterations (four waves, each at the pace of its busiest slot) + a fixed share per chunk
      d.cost = 0;
      for (size_t ch = 0; ch < fill.size(); ++ch) {
        if (!fill[ch]) continue;
        const unsigned short* cc = &cnt[ch * W];
        int mx = 0;
        for (int k = 0; k < W; ++k)
          if (unit[k] > 0) mx = std::max(mx, (cc[k] + unit[k] - 1) / unit[k]);
        d.cost += NW * mx + 6;
      }
    };
    Deal best;
    if (prm.cheap) {
      int fill = 0;
      best.members.emplace_back();
      for (int ci = 0; ci < (int)cand.size(); ++ci) {
        if (fill + cand[ci].n_rec > R) { best.members.emplace_back(); fill = 0; }
        best.members.back().push_back(ci);
        fill += cand[ci].n_rec;
      }
    } else {
      int t_hi = std::max(1, (int)std::ceil(mean - 1e-9));
      if (prm.pair_cap > 0) t_hi = std::min(t_hi, prm.pair_cap);
      int r_hi = R;
      if (mean / t_hi > 0.93) r_hi = std::max(R / 2, (int)(R * 0.93 * t_hi / mean));
      deal(t_hi, r_hi, best);
      if (t_hi > 1 && mean / (t_hi - 1) < 1.35) {
        Deal alt;
        deal(t_hi - 1, std::max(R / 2, (int)(R * 0.93 * (t_hi - 1) / mean)), alt);
        if (alt.cost < best.cost) best = std::move(alt);
      }
    }
    // emit: stream entries (in slot order), transposed codes
    std::vector<std::vector<unsigned>> lists((size_t)NW * 64);
    std::vector<unsigned> rr((size_t)nblk, 0), rrc((size_t)g, 0);
    std::vector<int> rec_obs;                 // chunk record id (arrival order) -> observation
    std::vector<int> slot_of, residue;        // record id -> slot / residue
    // cliques: the distinct record ids a lane group reads in one instruction (at most 16: one per lane of the group), flat arrays;
    // rec_cl: record id -> the cliques it belongs to, as CSR.  (Round 2 kept both as vectors of vectors: half of the plan's time went into
    // their allocation and into the strided walk of best_residue.)
    std::vector<int> clq_item;                // [clique][16]
    std::vector<unsigned char> clq_n;         // [clique]
    std::vector<int> rcl_start, rcl_item, rcl_fill;
    std::vector<int> clq_cnt;                 // [clique][16] members per residue
    std::vector<int> mark, class_size, order_r, deg_start;
    std::vector<size_t> nit_w((size_t)NW);
    std::vector<unsigned> packed((size_t)NWORD, 0u);
    for (const std::vector<int>& mem : best.members) {
      if (mem.empty()) continue;
      for (auto& l : lists) l.clear();
      std::fill(rr.begin(), rr.end(), 0u);
      std::fill(rrc.begin(), rrc.end(), 0u);
      rec_obs.clear();
      for (int ci : mem) {
        const int q = cand[ci].q;
        const int* gb = &pgb[(size_t)q * (G + 1)];
        const int na = gb[a + 1] - gb[a], nb = diag ? 0 : gb[b + 1] - gb[b];
        const int base = (int)rec_obs.size();
        for (int i = gb[a]; i < gb[a] + na; ++i) rec_obs.push_back(i);
        for (int i = gb[b]; !diag && i < gb[b] + nb; ++i) rec_obs.push_back(i);
        auto emit = [&](int blk, unsigned code) {
          const int slot = (int)(rr[blk]++ % (unsigned)rep);
          lists[(size_t)slot * nblk + blk].push_back(code);
        };
        for (int i = 0; i < na; ++i) {
          const int li = hcam[gb[a] + i] - gcam[a];
          const int j0 = diag ? i : na, j1 = diag ? na : na + nb;
          for (int j = j0; j < j1; ++j) {
            const int lj = diag ? hcam[gb[a] + j] - gcam[a] : hcam[gb[b] + (j - na)] - gcam[b];
            const unsigned code = (unsigned)(base + i) | ((unsigned)(base + j) << 16);
            if (diag && lj == li) {
              const std::vector<int>& h = helpers[li];
              emit(h[rrc[li]++ % h.size()], code);
              if (j != i) emit(h[rrc[li]++ % h.size()], (unsigned)(base + j) | ((unsigned)(base + i) << 16));
            } else {
              emit(li * g + lj, code);
            }
          }
        }
      }
      const int n_rec = (int)rec_obs.size();
      for (int w = 0; w < NW; ++w) {
        size_t mx = 0;
        for (int l = 0; l < 64; ++l) mx = std::max(mx, lists[(size_t)w * 64 + l].size());
        nit_w[w] = std::min<size_t>(mx, 255);  // a byte per wave; 255 pairs of one block in one chunk cannot happen (chunk_cap records)
      }
      slot_of.assign((size_t)n_rec, 0);
      if (prm.cheap) {
        for (int r = 0; r < n_rec; ++r) slot_of[r] = r;  // arrival order
      } else {
        // cliques of the LDS reads: per wave, iteration, lane group and operand the distinct records read together
        size_t n_clq = 0;
        for (int w = 0; w < NW; ++w) n_clq += nit_w[w] * 8;
        clq_item.resize(n_clq * 16);
        clq_n.assign(n_clq, 0);
        mark.assign((size_t)n_rec, -1);
        {
          size_t c0 = 0;
          for (int w = 0; w < NW; ++w)
            for (size_t it = 0; it < nit_w[w]; ++it)
              for (int side = 0; side < 2; ++side, c0 += 4)
                for (int l = 0; l < 64; ++l) {
                  const std::vector<unsigned>& li = lists[(size_t)w * 64 + l];
                  if (it >= li.size()) continue;
                  const int r = side ? (int)(li[it] >> 16) : (int)(li[it] & 0xffffu);
                  const int cq = (int)c0 + b128_group(l);
                  if (mark[r] == cq) continue;  // the same record twice in a group: one address, a broadcast
                  mark[r] = cq;
                  clq_item[(size_t)cq * 16 + clq_n[cq]++] = r;
                }
        }
        auto clique_cost = [&](const std::vector<int>& res) {
          long cyc = 0;
          int cnt[16];
          for (size_t cq = 0; cq < n_clq; ++cq) {
            if (!clq_n[cq]) continue;
            std::fill(cnt, cnt + 16, 0);
            int mx = 0;
            for (int k = 0; k < clq_n[cq]; ++k) mx = std::max(mx, ++cnt[res[clq_item[cq * 16 + k]] & 15]);
            cyc += mx;
          }
          return cyc;
        };
        residue.resize((size_t)n_rec);
        for (int r = 0; r < n_rec; ++r) residue[r] = r & 15;
        long groups = 0;
        for (size_t cq = 0; cq < n_clq; ++cq) groups += clq_n[cq] ? 1 : 0;
        job.lds_groups += groups;
        job.lds_cycles_arrival += clique_cost(residue);
        // colouring
        rcl_start.assign((size_t)n_rec + 1, 0);
        for (size_t cq = 0; cq < n_clq; ++cq)
          for (int k = 0; k < clq_n[cq]; ++k) rcl_start[(size_t)clq_item[cq * 16 + k] + 1]++;
        for (int r = 0; r < n_rec; ++r) rcl_start[(size_t)r + 1] += rcl_start[r];
        rcl_item.resize((size_t)rcl_start[n_rec]);
        rcl_fill.assign(rcl_start.begin(), rcl_start.end() - 1);
        for (size_t cq = 0; cq < n_clq; ++cq)  // ascending clique order per record, as the vectors of round 2 had it
          for (int k = 0; k < clq_n[cq]; ++k) rcl_item[(size_t)rcl_fill[clq_item[cq * 16 + k]]++] = (int)cq;
        clq_cnt.assign(n_clq * 16, 0);
        class_size.assign(16, 0);
        order_r.resize((size_t)n_rec);
        {  // records by the number of cliques they are in, most first, stable (counting sort)
          int maxd = 0;
          for (int r = 0; r < n_rec; ++r) maxd = std::max(maxd, rcl_start[r + 1] - rcl_start[r]);
          deg_start.assign((size_t)maxd + 2, 0);
          for (int r = 0; r < n_rec; ++r) deg_start[(size_t)maxd - (rcl_start[r + 1] - rcl_start[r]) + 1]++;
          for (int d = 0; d <= maxd; ++d) deg_start[(size_t)d + 1] += deg_start[d];
          for (int r = 0; r < n_rec; ++r) order_r[(size_t)deg_start[(size_t)maxd - (rcl_start[r + 1] - rcl_start[r])]++] = r;
        }
        std::fill(residue.begin(), residue.end(), -1);
        auto best_residue = [&](int r) {
          long cost[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int e = rcl_start[r]; e < rcl_start[r + 1]; ++e) {
            const int* row = &clq_cnt[(size_t)rcl_item[e] * 16];  // one contiguous row per clique: the 16 sums vectorise
            for (int rho = 0; rho < 16; ++rho) cost[rho] += row[rho];
          }
          int best_rho = -1;
          long best_cost = 0;
          for (int rho = 0; rho < 16; ++rho) {
            if (class_size[rho] >= (R - rho + 15) / 16) continue;  // slots rho, rho + 16, ... below the chunk's R slots
            const long c = cost[rho] * 64 + class_size[rho];
            if (best_rho < 0 || c < best_cost) { best_rho = rho; best_cost = c; }
          }
          return best_rho;
        };
        auto put = [&](int r, int rho, int d) {
          residue[r] = d > 0 ? rho : -1;
          class_size[rho] += d;
          for (int e = rcl_start[r]; e < rcl_start[r + 1]; ++e) clq_cnt[(size_t)rcl_item[e] * 16 + rho] += d;
        };
        for (int r : order_r) put(r, best_residue(r), +1);
        for (int sweep = 0; sweep < prm.colour_sweeps; ++sweep)
          for (int r : order_r) {
            const int old = residue[r];
            put(r, old, -1);
            put(r, best_residue(r), +1);
          }
        job.lds_cycles += clique_cost(residue);
        // slots: residue rho takes rho, rho + 16, rho + 32, ...
        int next_k[16] = {0};
        for (int r = 0; r < n_rec; ++r) slot_of[r] = residue[r] + 16 * next_k[residue[r]]++;
      }
      int n_slots = 0;
      for (int r = 0; r < n_rec; ++r) n_slots = std::max(n_slots, slot_of[r] + 1);
      const size_t open = job.obs.size();
      job.obs.resize(open + n_slots, 0);  // holes point at observation 0: loaded, never referenced
      for (int r = 0; r < n_rec; ++r) job.obs[open + slot_of[r]] = rec_obs[r];
      std::fill(packed.begin(), packed.end(), 0u);
      for (int w = 0; w < NW; ++w) {
        const size_t mx = nit_w[w];
        packed[w / 4] |= (unsigned)mx << (8 * (w % 4));
        for (size_t it = 0; it < mx; ++it)
          for (int l = 0; l < 64; ++l) {
            const std::vector<unsigned>& li = lists[(size_t)w * 64 + l];
            unsigned code = ZERO;
            if (it < li.size()) code = piece_of(slot_of[li[it] & 0xffffu]) | (piece_of(slot_of[li[it] >> 16]) << 16);
            job.codes.push_back(code);
          }
        job.lane_iters += (long)mx * 64;
      }
      for (auto& l : lists) job.n_pairs += (long)l.size();
      job.nit.insert(job.nit.end(), packed.begin(), packed.end());
      job.chunk_start.push_back((int)job.obs.size());
      job.code_start.push_back((int)job.codes.size());
    }
  };

  {
    unsigned hw = n_threads;
    hw = std::min<unsigned>(hw, (unsigned)jobs.size());
    std::atomic<size_t> next{0};
    auto worker = [&]() {
      for (size_t j = next++; j < jobs.size(); j = next++) {
        if (prm.cancel && prm.cancel->load(std::memory_order_relaxed)) return;
        run_job(jobs[j]);
      }
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < hw; ++i) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
  }
  if (prm.cancel && prm.cancel->load(std::memory_order_relaxed)) return -2;
  for (const Job& j : jobs)
    if (j.rc) return j.rc;
  const auto t_phase1 = std::chrono::steady_clock::now();

  // concatenate in (tile, region) order: offsets first, then the copies by the worker threads
  size_t n_obs = 0, n_codes = 0, n_chunks = 0;
  std::vector<size_t> job_o(jobs.size()), job_c(jobs.size()), job_ch(jobs.size());
  for (size_t j = 0; j < jobs.size(); ++j) {
    job_o[j] = n_obs; job_c[j] = n_codes; job_ch[j] = n_chunks;
    n_obs += jobs[j].obs.size(); n_codes += jobs[j].codes.size(); n_chunks += jobs[j].nit.size() / NWORD;
  }
  out.obs.resize_uninit(n_obs + 2 * (size_t)R);
  out.codes.resize_uninit(n_codes + (size_t)4 * 64 * NW);
  std::fill(out.obs.begin() + (long)n_obs, out.obs.end(), 0);
  std::fill(out.codes.begin() + (long)n_codes, out.codes.end(), ZERO);
  out.chunk_start.assign(n_chunks + 1, 0);
  out.code_start.assign(n_chunks + 1, 0);
  out.nit.assign(n_chunks * NWORD, 0);
  out.tile_chunk_begin.assign(nT + 1, 0);
  out.n_pairs = 0; out.lane_iters = 0; out.lds_groups = 0; out.lds_cycles = 0; out.lds_cycles_arrival = 0;
  for (int t = 0; t < nT; ++t) out.tile_chunk_begin[t] = (int)job_ch[(size_t)t * n_regions];
  {
    std::atomic<size_t> next{0};
    auto copier = [&]() {
      for (size_t ji = next++; ji < jobs.size(); ji = next++) {
        const Job& j = jobs[ji];
        const size_t o = job_o[ji], cpos = job_c[ji], ch = job_ch[ji];
        std::copy(j.obs.begin(), j.obs.end(), out.obs.begin() + (long)o);
        std::copy(j.codes.begin(), j.codes.end(), out.codes.begin() + (long)cpos);
        const size_t jch = j.nit.size() / NWORD;
        std::copy(j.nit.begin(), j.nit.end(), out.nit.begin() + (long)(ch * NWORD));
        for (size_t c = 0; c < jch; ++c) {
          out.chunk_start[ch + c + 1] = (int)(o + j.chunk_start[c + 1]);
          out.code_start[ch + c + 1] = (int)(cpos + j.code_start[c + 1]);
        }
      }
    };
    const unsigned nth = std::min<unsigned>(n_threads, (unsigned)jobs.size());
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < nth; ++i) pool.emplace_back(copier);
    copier();
    for (auto& th : pool) th.join();
  }
  for (const Job& j : jobs) {
    out.n_pairs += j.n_pairs; out.lane_iters += j.lane_iters;
    out.lds_groups += j.lds_groups; out.lds_cycles += j.lds_cycles; out.lds_cycles_arrival += j.lds_cycles_arrival;
  }
  const size_t ch = n_chunks;
  out.tile_chunk_begin[nT] = (int)ch;
  out.seconds_runs = std::chrono::duration<double>(t_phase0 - t_begin).count();
  out.seconds_jobs = std::chrono::duration<double>(t_phase1 - t_phase0).count();
  out.seconds_concat = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_phase1).count();
  return 0;
}

inline int build_reg2_plan(const Reg2Params& prm, const std::vector<int>& hcam, const std::vector<int>& hps, Reg2Plan& out) {
  return build_reg2_plan(prm, hcam.data(), hps.data(), out);
}

}  // namespace cba
