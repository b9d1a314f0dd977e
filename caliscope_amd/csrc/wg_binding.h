// Workgroup -> tile binding of the tiled Schur kernels, host side only: plain C++, so that the CPU suite can build it with g++ and check
// that every chunk is bound exactly once (tests/native/plan_harness.cpp, tests/test_schur_plan.py).
//
// A kernel's workgroups are persistent: workgroup b belongs to tile wt[b] and walks the chunks wfirst[b], wfirst[b] + wstride[b], ... <
// wend[b].  Workgroups are handed out to the tiles in proportion to the tiles' estimated cost (at least one each): a diagonal tile's
// chunks hold 3 pair iterations per wave where an off-diagonal tile's hold 2 — with equal chunk COUNTS per workgroup the diagonal
// tiles' workgroups ran 30 % longer than the rest, and the kernel lasts as long as its slowest workgroup.
#pragma once
#include <algorithm>
#include <vector>

#if defined(__HIPCC__)
#define CBA_WG_HD __host__ __device__ __forceinline__
#else
#define CBA_WG_HD inline
#endif

namespace cba {

// The id a workgroup of k_schur_reg3 works under (the index into wt / wfirst / ... above and of its partial row): groups of eight consecutive
// hardware ids (one per XCD) alternate between the two halves of the grid.  With two workgroups per CU the ones dispatched later run slower
// (phase clocks: the time per trip grows by a third from the first to the last quarter of the grid); spread like this every tile gets the
// same mix, and the cost model of the binding holds for all of them.  A permutation of [0, grid) that keeps id mod 8 (the XCD); the
// identity when the grid is not a multiple of 16.
CBA_WG_HD int logical_workgroup(int block, int grid) {
  if ((grid & 15) != 0) return block;
  const int gh = block >> 3, halfg = grid >> 4;
  return ((gh < halfg ? 2 * gh : 2 * (gh - halfg) + 1) << 3) | (block & 7);
}

struct WgBinding {
  std::vector<int> wgb;                         // [nT + 1] first workgroup of every tile
  std::vector<int> wt, wfirst, wend, wstride;   // [grid] tile, first chunk, end of the chunk range, stride
  int grid = 0;
  bool xcd_mode = false;
};

// TCB: first chunk of every tile ([nT + 1]); tile_cost: optional, default the chunk counts; reg: the register kernels (XCD-aware binding);
// chunk_cost: optional, per chunk (what tile_cost sums): the XCD slices of a tile are then cut in proportion to COST, not to chunk counts.
inline WgBinding bind_workgroups(const std::vector<int>& TCB, int nT, int max_blocks, bool reg, const std::vector<double>* tile_cost = nullptr,
                                 const std::vector<float>* chunk_cost = nullptr, bool fine = false) {
  WgBinding out;
  const int n_tile_chunks = TCB[nT] - TCB[0];
  std::vector<long> nch(nT);
  for (int t = 0; t < nT; ++t) nch[t] = TCB[t + 1] - TCB[t];
  std::vector<double> wt_cost(nT);
  double cost_total = 0.0;
  for (int t = 0; t < nT; ++t) { wt_cost[t] = tile_cost ? (*tile_cost)[t] : (double)nch[t]; cost_total += wt_cost[t]; }
  auto allocate = [&](int budget) {
    std::vector<int> nwg(nT);
    long used = 0;
    for (int t = 0; t < nT; ++t) {
      const long w = cost_total > 0.0 ? (long)(wt_cost[t] * budget / cost_total) : 1;
      nwg[t] = (int)std::max<long>(1, std::min<long>(w, std::max<long>(nch[t], 1)));
      used += nwg[t];
    }
    // hand out what is left (or take back the excess) where the cost per workgroup is most uneven
    while (used != budget) {
      int best = -1;
      double score = 0.0;
      for (int t = 0; t < nT; ++t) {
        if (used < budget) {
          if (nwg[t] >= nch[t]) continue;
          const double sc = wt_cost[t] / nwg[t];
          if (best < 0 || sc > score) { best = t; score = sc; }
        } else {
          if (nwg[t] <= 1) continue;
          const double sc = -wt_cost[t] / (nwg[t] - 1);
          if (best < 0 || sc > score) { best = t; score = sc; }
        }
      }
      if (best < 0) break;
      if (used < budget) { nwg[best]++; used++; } else { nwg[best]--; used--; }
    }
    return nwg;
  };
  const int budget = std::max(nT, std::min(max_blocks, std::max(1, n_tile_chunks)));
  // XCD-aware binding (register kernel): workgroup b runs on XCD b mod 8.  Every XCD walks its own slice of every tile's chunk range (the point
  // range), so the G tiles that gather a given T record do so through the same L2 at about the same time; HBM then serves each record once
  // instead of G times.
  // Round 6, `fine` (an option, not the default — see prepare_install in cba_lib.hip for what it measured): a tile's workgroup count is not a multiple of eight.  With 512 workgroups over 10 tiles that rounding left the tiles 5-8 units of
  // eight each — up to +-8 % off their share — and the planned cost per workgroup spread 193 / 226 / 247 (min / mean / max), which is what the device
  // stamps show (lifetimes 80 / 107 / 126 us: the kernel lasts as long as its slowest workgroup).  Now the counts follow the cost to one workgroup, a
  // tile's workgroups keep contiguous ids (k_reg_reduce relies on it), workgroup id b sits on XCD b mod 8 — so a tile holds w_x = 6 or 7 workgroups
  // on XCD x — and the tile's chunk range is cut into eight slices in proportion to w_x (by cost when the caller gives chunk costs): planned spread
  // 218 / 226 / 233.
  constexpr int XCDS = 8;
  const bool xcd_mode = reg && budget % XCDS == 0 && nT <= budget / XCDS && n_tile_chunks >= 4 * budget;
  std::vector<int> nwg = allocate((xcd_mode && !fine) ? budget / XCDS : budget), wgb(nT + 1, 0);
  if (xcd_mode && !fine) for (int& w : nwg) w *= XCDS;  // (rounds 2-5: multiples of eight, equal eighths of the chunk range)
  for (int t = 0; t < nT; ++t) wgb[t + 1] = wgb[t] + nwg[t];
  const int grid = wgb[nT];
  std::vector<int> wt(grid), wfirst(grid), wend(grid), wstride(grid);
  for (int t = 0; t < nT; ++t) {
    if (!xcd_mode) {
      for (int r = 0; r < nwg[t]; ++r) { const int b = wgb[t] + r; wt[b] = t; wfirst[b] = TCB[t] + r; wend[b] = TCB[t + 1]; wstride[b] = nwg[t]; }
      continue;
    }
    if (!fine) {
      for (int r = 0; r < nwg[t]; ++r) {
        const int b = wgb[t] + r, x = r % XCDS, sidx = r / XCDS;
        const long lo = nch[t] * x / XCDS, hi = nch[t] * (x + 1) / XCDS;
        wt[b] = t; wfirst[b] = TCB[t] + (int)lo + sidx; wend[b] = TCB[t] + (int)hi; wstride[b] = nwg[t] / XCDS;
      }
      continue;
    }
    int w[XCDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < nwg[t]; ++r) w[(wgb[t] + r) % XCDS]++;
    int last_x = 0;
    for (int x = 0; x < XCDS; ++x) if (w[x] > 0) last_x = x;
    // slice boundaries: cumulative cost (or count) nearest to the XCD's cumulative share of the tile's workgroups
    const long n = nch[t];
    std::vector<double> pre((size_t)n + 1, 0.0);
    for (long c = 0; c < n; ++c) pre[(size_t)c + 1] = pre[(size_t)c] + (chunk_cost ? (double)(*chunk_cost)[(size_t)TCB[t] + c] : 1.0);
    long lo_x[XCDS], hi_x[XCDS];
    long lo = 0;
    int wcum = 0;
    for (int x = 0; x < XCDS; ++x) {
      wcum += w[x];
      long hi = lo;
      if (w[x] > 0) {
        const double goal = pre[(size_t)n] * wcum / nwg[t];
        while (hi < n && (pre[(size_t)hi + 1] <= goal || goal - pre[(size_t)hi] > pre[(size_t)hi + 1] - goal)) ++hi;
        if (x == last_x) hi = n;
      }
      lo_x[x] = lo; hi_x[x] = hi;
      lo = hi;
    }
    for (int r = 0; r < nwg[t]; ++r) {
      const int b = wgb[t] + r, x = b % XCDS, sidx = r / XCDS;  // (the s-th workgroup of this tile on XCD x: ids on one XCD are eight apart)
      wt[b] = t;
      wfirst[b] = TCB[t] + (int)lo_x[x] + sidx; wend[b] = TCB[t] + (int)hi_x[x]; wstride[b] = w[x];
    }
  }
  out.wgb = std::move(wgb); out.wt = std::move(wt); out.wfirst = std::move(wfirst); out.wend = std::move(wend); out.wstride = std::move(wstride);
  out.grid = grid; out.xcd_mode = xcd_mode;
  return out;
}

// Estimated cost of every tile for the binding above, from the plan's iteration counts.  Per chunk: `cost_a` for gather + barrier (in units of
// one pair iteration; phase clocks on cfg4: ~2200 against ~830 clocks) + the pair iterations of its slowest PHYSICAL wave (a physical wave
// runs `code_waves / phys_waves` of the plan's waves one after the other: their iterations add up).
inline std::vector<double> tile_costs(const std::vector<unsigned>& nit, const std::vector<int>& tile_chunk_begin, int nT, int code_waves, int phys_waves,
                                      double cost_a, std::vector<float>* chunk_cost_out = nullptr) {
  std::vector<double> cost(nT, 0.0);
  if (chunk_cost_out) chunk_cost_out->assign((size_t)tile_chunk_begin[nT], 0.0f);
  const int nword = code_waves / 4, vb = std::max(1, code_waves / std::max(phys_waves, 1));
  const int pw = code_waves / vb;
  for (int t = 0; t < nT; ++t)
    for (int ch = tile_chunk_begin[t]; ch < tile_chunk_begin[t + 1]; ++ch) {
      int mx = 0;
      for (int w = 0; w < pw; ++w) {
        int its = 0;
        for (int v = 0; v < vb; ++v) { const int vw = v * pw + w; its += (int)((nit[(size_t)ch * nword + vw / 4] >> (8 * (vw % 4))) & 0xffu); }
        mx = std::max(mx, its);
      }
      cost[t] += cost_a + mx;
      if (chunk_cost_out) (*chunk_cost_out)[(size_t)ch] = (float)(cost_a + mx);
    }
  return cost;
}

}  // namespace cba
