// CPU replay of the dealt Schur plan (caliscope_amd/csrc/schur_plan.h): walks the transposed pair codes exactly as
// k_schur_reg3 does (chunk by chunk, wave by wave, iteration by iteration, lane by lane) and accumulates T_i T_j^T per
// owner thread, so that tests/test_schur_plan.py can compare the per-block sums with a direct sum over the points.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../caliscope_amd/csrc/schur_plan.h"
#include "../../caliscope_amd/csrc/wg_binding.h"

extern "C" {

// acc_out: [n_tiles][64 n_waves][nc*nc] (thread = slot * g*g + block); stats_out[10]: n_chunks, n_pairs, lane_iters, n_regions, stream entries, LDS read groups / cycles / cycles in arrival order, iterations over the cap, digest of the plan arrays
int plan_replay(int C, int P, int G, int g, int rep, int nc, int rec, int chunk_cap, int slots_per_wave, int lds_stride, int wave_pieces, int region_chunks,
                int heavy_obs, int threads, int n_waves, int pair_cap, int cheap, const int* hcam, const int* hps, const double* T, double* acc_out, long* stats_out) {
  cba::Reg2Params prm;
  prm.C = C; prm.P = P; prm.G = G; prm.g = g; prm.rep = rep; prm.chunk_cap = chunk_cap;
  // the kernel's staging layout (cba_kernels.h, Reg2Cfg): a wave stages slots_per_wave slots, lds_stride pieces apart, in a run of wave_pieces pieces (k_schur_reg3 pads it to whole loads)
  prm.rec_pieces = lds_stride; prm.slots_per_wave = slots_per_wave; prm.wave_pieces = wave_pieces;
  const int zero_piece = (chunk_cap + slots_per_wave - 1) / slots_per_wave * prm.wave_pieces;
  prm.zero_piece = zero_piece;
  prm.region_chunks = region_chunks; prm.heavy_obs = heavy_obs; prm.threads = threads; prm.n_waves = n_waves; prm.pair_cap = pair_cap;
  prm.cheap = cheap != 0;  // the plan a handle starts with: one open chunk in point order, records in arrival order
  prm.cheap_lean = cheap != 2;  // 2: the cheap plan through the general path (the lean path must produce the same arrays: stats_out[8] is their digest)
  const long N = hps[P];
  std::vector<int> vcam(hcam, hcam + N), vps(hps, hps + P + 1);
  cba::Reg2Plan plan;
  const int rc = cba::build_reg2_plan(prm, vcam, vps, plan);
  if (rc) return rc;
  const int nT = G * (G + 1) / 2, bsz = nc * nc;
  const int n_chunks = plan.tile_chunk_begin[nT];
  for (int t = 0; t < nT; ++t)
    for (int ch = plan.tile_chunk_begin[t]; ch < plan.tile_chunk_begin[t + 1]; ++ch) {
      const int c0 = plan.chunk_start[ch], len = plan.chunk_start[ch + 1] - c0;
      if (len <= 0 || len > chunk_cap) return -10;
      long code = plan.code_start[ch];
      for (int w = 0; w < n_waves; ++w) {
        const int n = (plan.nit[(size_t)ch * (n_waves / 4) + w / 4] >> (8 * (w % 4))) & 0xff;
        for (int it = 0; it < n; ++it)
          for (int lane = 0; lane < 64; ++lane, ++code) {
            const unsigned cd = plan.codes[code];
            const int ia = cd & 0xffff, ja = cd >> 16;
            if (ia == zero_piece && ja == zero_piece) continue;
            // piece address -> slot (the inverse of the kernel's staging layout)
            auto slot_at = [&](int addr) {
              const int w = addr / prm.wave_pieces, off = addr % prm.wave_pieces;
              return (off % prm.rec_pieces) ? -1 : w * prm.slots_per_wave + off / prm.rec_pieces;
            };
            const int il = slot_at(ia), jl = slot_at(ja);
            if (il < 0 || jl < 0 || il >= len || jl >= len) return -11;
            const double* Ti = T + (long)plan.obs[c0 + il] * rec;
            const double* Tj = T + (long)plan.obs[c0 + jl] * rec;
            double* a = acc_out + ((long)t * 64 * n_waves + w * 64 + lane) * bsz;
            for (int r = 0; r < nc; ++r)
              for (int c = 0; c < nc; ++c) a[r * nc + c] += Ti[3 * r] * Tj[3 * c] + Ti[3 * r + 1] * Tj[3 * c + 1] + Ti[3 * r + 2] * Tj[3 * c + 2];
          }
      }
      if (code != plan.code_start[ch + 1]) return -12;
    }
  if (std::getenv("PLAN_TILE_STATS"))
    for (int t = 0; t < nT; ++t) {
      long its = 0, slots = 0, pairs = 0;
      for (int ch = plan.tile_chunk_begin[t]; ch < plan.tile_chunk_begin[t + 1]; ++ch) {
        slots += plan.chunk_start[ch + 1] - plan.chunk_start[ch];
        for (int w = 0; w < n_waves; ++w) its += (plan.nit[(size_t)ch * (n_waves / 4) + w / 4] >> (8 * (w % 4))) & 0xff;
        for (long c = plan.code_start[ch]; c < plan.code_start[ch + 1]; ++c) pairs += plan.codes[c] != ((unsigned)zero_piece | ((unsigned)zero_piece << 16));
      }
      const int nch = plan.tile_chunk_begin[t + 1] - plan.tile_chunk_begin[t];
      std::fprintf(stderr, "tile %d: %d chunks, %.1f slots/chunk, %.2f wave-iterations per chunk and wave, utilisation %.3f\n", t, nch, (double)slots / std::max(nch, 1),
                   (double)its / std::max(nch, 1) / n_waves, (double)pairs / std::max(1L, its * 64));
    }
  stats_out[0] = n_chunks; stats_out[1] = plan.n_pairs; stats_out[2] = plan.lane_iters; stats_out[3] = plan.n_regions;
  stats_out[4] = (long)plan.obs.size() - 2L * chunk_cap;
  {  // wave-iterations beyond the cap (codes the kernel loads inside its pair loop)
    long over = 0;
    for (int ch = 0; ch < n_chunks; ++ch)
      for (int w = 0; w < n_waves; ++w) {
        const int n = (plan.nit[(size_t)ch * (n_waves / 4) + w / 4] >> (8 * (w % 4))) & 0xff;
        if (pair_cap > 0 && n > pair_cap) over += n - pair_cap;
      }
    stats_out[8] = over;
  }
  stats_out[5] = plan.lds_groups; stats_out[6] = plan.lds_cycles; stats_out[7] = plan.lds_cycles_arrival;
  {  // FNV-1a over every array of the plan (63 bits: the caller's array is signed)
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void* ptr, size_t bytes) { const unsigned char* c = static_cast<const unsigned char*>(ptr); for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
    mix(plan.obs.data(), plan.obs.size() * sizeof(int)); mix(plan.chunk_start.data(), plan.chunk_start.size() * sizeof(int));
    mix(plan.code_start.data(), plan.code_start.size() * sizeof(int)); mix(plan.nit.data(), plan.nit.size() * sizeof(unsigned));
    mix(plan.codes.data(), plan.codes.size() * sizeof(unsigned)); mix(plan.tile_chunk_begin.data(), plan.tile_chunk_begin.size() * sizeof(int));
    stats_out[9] = (long)(h >> 1);
  }
  return 0;
}

// Workgroup binding (csrc/wg_binding.h) of a plan built like plan_replay's: per workgroup (tile, first, end, stride), per tile its cost.
// Returns the grid (or a negative plan error); out arrays sized by the caller for `max_blocks + n_tiles` workgroups.
int bind_replay(int C, int P, int G, int g, int rep, int chunk_cap, int slots_per_wave, int lds_stride, int wave_pieces, int region_chunks, int n_waves,
                int pair_cap, int phys_waves, double cost_a, int max_blocks, int reg, const int* hcam, const int* hps, int* wt, int* wfirst, int* wend,
                int* wstride, int* tcb_out, double* cost_out, int* xcd_out) {
  cba::Reg2Params prm;
  prm.C = C; prm.P = P; prm.G = G; prm.g = g; prm.rep = rep; prm.chunk_cap = chunk_cap;
  prm.rec_pieces = lds_stride; prm.slots_per_wave = slots_per_wave; prm.wave_pieces = wave_pieces;
  prm.zero_piece = (chunk_cap + slots_per_wave - 1) / slots_per_wave * wave_pieces;
  prm.region_chunks = region_chunks; prm.threads = 4; prm.n_waves = n_waves; prm.pair_cap = pair_cap;
  const long N = hps[P];
  std::vector<int> vcam(hcam, hcam + N), vps(hps, hps + P + 1);
  cba::Reg2Plan plan;
  const int rc = cba::build_reg2_plan(prm, vcam, vps, plan);
  if (rc) return rc;
  const int nT = G * (G + 1) / 2;
  std::vector<float> chunk_cost;
  const std::vector<double> cost = cba::tile_costs(plan.nit, plan.tile_chunk_begin, nT, n_waves, phys_waves, cost_a, &chunk_cost);
  const cba::WgBinding b = cba::bind_workgroups(plan.tile_chunk_begin, nT, max_blocks, reg != 0, cost_a >= 0.0 ? &cost : nullptr, cost_a >= 0.0 ? &chunk_cost : nullptr, std::getenv("PLAN_BIND_FINE") != nullptr);
  for (int i = 0; i < b.grid; ++i) { wt[i] = b.wt[i]; wfirst[i] = b.wfirst[i]; wend[i] = b.wend[i]; wstride[i] = b.wstride[i]; }
  for (int t = 0; t <= nT; ++t) tcb_out[t] = plan.tile_chunk_begin[t];
  for (int t = 0; t < nT; ++t) cost_out[t] = cost[t];
  *xcd_out = b.xcd_mode ? 1 : 0;
  return b.grid;
}

int logical_workgroup_of(int block, int grid) { return cba::logical_workgroup(block, grid); }
}
