// Host-side plan of the observation order (include/caliscope_ba.h: cba_host_plan): stable sort by (world point, camera),
// first observation of every point, chunk table.  Plain C++: shared by the device library (cba_lib.hip) and the CPU test
// build of the C ABI (tests/native/cpu_library.cpp).  `fail(code, fmt, ...)` is the including file's error reporter.
#pragma once
#include <cstdint>
#include <vector>

template <typename Fail>
static int64_t host_plan_impl(Fail fail, int32_t n_points, int64_t n_obs, const int32_t* obs_pt, const int32_t* obs_cam, int32_t n_cams,
                      int32_t chunk_cap, int64_t* order_out, int64_t* pt_start_out, int64_t* chunk_start_out) {
  if (n_points < 0 || n_obs < 0 || chunk_cap <= 0 || (n_obs > 0 && !obs_pt)) return fail(CBA_ERR_INVALID, "cba_host_plan: bad arguments");
  // Already in (point, camera) order?  One sequential pass decides (the arrays CaptureVolume hands over after a first optimize() are, and so is
  // anything produced point by point); the two scattering passes below cost ~9 ns per observation.
  bool sorted = true;
  for (int64_t i = 0; i < n_obs; ++i) {
    const int32_t p = obs_pt[i];
    if (p < 0 || p >= n_points) return fail(CBA_ERR_INVALID, "observation %lld: world-point index %d out of range", (long long)i, p);
    if (obs_cam && n_cams > 0 && (obs_cam[i] < 0 || obs_cam[i] >= n_cams)) return fail(CBA_ERR_INVALID, "observation %lld: camera index %d out of range", (long long)i, obs_cam[i]);
    if (i > 0 && (p < obs_pt[i - 1] || (p == obs_pt[i - 1] && obs_cam && n_cams > 0 && obs_cam[i] < obs_cam[i - 1]))) sorted = false;
  }
  // optional first key: camera (stable counting sort), so that the final order is (point, camera, input order)
  std::vector<int64_t> by_cam;
  if (!sorted && obs_cam && n_cams > 0) {
    std::vector<int64_t> cc((size_t)n_cams + 1, 0);
    for (int64_t i = 0; i < n_obs; ++i) {
      if (obs_cam[i] < 0 || obs_cam[i] >= n_cams) return fail(CBA_ERR_INVALID, "observation %lld: camera index %d out of range", (long long)i, obs_cam[i]);
      cc[obs_cam[i] + 1]++;
    }
    for (int32_t c = 0; c < n_cams; ++c) cc[c + 1] += cc[c];
    by_cam.resize((size_t)n_obs);
    for (int64_t i = 0; i < n_obs; ++i) by_cam[cc[obs_cam[i]]++] = i;
  }
  std::vector<int64_t> count((size_t)n_points + 1, 0);
  for (int64_t i = 0; i < n_obs; ++i) {
    const int32_t p = obs_pt[i];
    if (p < 0 || p >= n_points) return fail(CBA_ERR_INVALID, "observation %lld: world-point index %d out of range", (long long)i, p);
    count[p + 1]++;
  }
  for (int32_t p = 0; p < n_points; ++p) count[p + 1] += count[p];
  for (int32_t p = 0; p <= n_points; ++p) pt_start_out[p] = count[p];
  if (sorted) {
    for (int64_t q = 0; q < n_obs; ++q) order_out[q] = q;
  } else {
    std::vector<int64_t> cursor(count.begin(), count.end() - 1);
    for (int64_t q = 0; q < n_obs; ++q) {  // stable counting sort by point
      const int64_t i = by_cam.empty() ? q : by_cam[q];
      order_out[cursor[obs_pt[i]]++] = i;
    }
  }
  int64_t n_chunks = 0;
  int64_t start = 0;
  chunk_start_out[0] = 0;
  for (int32_t p = 0; p < n_points; ++p) {
    const int64_t begin = pt_start_out[p], end = pt_start_out[p + 1];
    if (end - begin > chunk_cap) {
      // a point that does not fit one chunk gets chunks of its own ("fragments" of at most chunk_cap observations)
      if (begin > start) chunk_start_out[++n_chunks] = begin;
      for (int64_t o = begin + chunk_cap; o < end; o += chunk_cap) chunk_start_out[++n_chunks] = o;
      chunk_start_out[++n_chunks] = end;
      start = end;
    } else if (end - start > chunk_cap) {  // close the chunk before this point
      chunk_start_out[++n_chunks] = begin;
      start = begin;
    }
  }
  if (n_obs > start) chunk_start_out[++n_chunks] = n_obs;
  return n_chunks;
}
