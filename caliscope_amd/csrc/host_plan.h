// Host-side plan of the observation order (include/caliscope_ba.h: cba_host_plan): stable sort by (world point, camera),
// first observation of every point, chunk table.  Plain C++: shared by the device library (cba_lib.hip) and the CPU test
// build of the C ABI (tests/native/cpu_library.cpp).  `fail(code, fmt, ...)` is the including file's error reporter.
#pragma once
#include <algorithm>
#include <cstdint>
#include <thread>
#include <vector>

template <typename Fail>
static int64_t host_plan_impl(Fail fail, int32_t n_points, int64_t n_obs, const int32_t* obs_pt, const int32_t* obs_cam, int32_t n_cams,
                      int32_t chunk_cap, int64_t* order_out, int64_t* pt_start_out, int64_t* chunk_start_out) {
  if (n_points < 0 || n_obs < 0 || chunk_cap <= 0 || (n_obs > 0 && !obs_pt)) return fail(CBA_ERR_INVALID, "cba_host_plan: bad arguments");
  // Already in (point, camera) order?  One pass decides (the arrays CaptureVolume hands over after a first optimize() are, and so is anything
  // produced point by point); the two scattering passes below cost ~9 ns per observation.  The pass and, for sorted input, the point table and
  // the identity order are split over a few threads (1M observations: 4 ms on one).
  const bool with_cam = obs_cam && n_cams > 0;
  const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, std::thread::hardware_concurrency()), n_obs / 131072));
  auto slices = [&](auto&& body) {  // body(thread, lo, hi)
    std::vector<std::thread> pool;
    for (int t = 1; t < nth; ++t) pool.emplace_back(body, t, n_obs * t / nth, n_obs * (t + 1) / nth);
    body(0, (int64_t)0, n_obs / nth);
    for (auto& th : pool) th.join();
  };
  std::vector<char> slice_ok((size_t)nth, 1), slice_sorted((size_t)nth, 1);
  slices([&](int t, int64_t lo, int64_t hi) {
    bool ok = true, srt = true;
    for (int64_t i = lo; i < hi; ++i) {
      const int32_t p = obs_pt[i];
      if (p < 0 || p >= n_points || (with_cam && (obs_cam[i] < 0 || obs_cam[i] >= n_cams))) { ok = false; break; }
      if (i > 0 && (p < obs_pt[i - 1] || (p == obs_pt[i - 1] && with_cam && obs_cam[i] < obs_cam[i - 1]))) srt = false;
    }
    slice_ok[(size_t)t] = ok; slice_sorted[(size_t)t] = srt;
  });
  bool sorted = true;
  for (int t = 0; t < nth; ++t) {
    if (!slice_ok[(size_t)t])  // report the first offender, in input order
      for (int64_t i = 0; i < n_obs; ++i) {
        const int32_t p = obs_pt[i];
        if (p < 0 || p >= n_points) return fail(CBA_ERR_INVALID, "observation %lld: world-point index %d out of range", (long long)i, p);
        if (with_cam && (obs_cam[i] < 0 || obs_cam[i] >= n_cams)) return fail(CBA_ERR_INVALID, "observation %lld: camera index %d out of range", (long long)i, obs_cam[i]);
      }
    sorted = sorted && slice_sorted[(size_t)t];
  }
  if (sorted) {
    // first observation of every point straight from the runs of equal indices: thread t fills the entries of the points that BEGIN in its slice
    // (and of the unobserved points in front of them); the identity order on the way
    slices([&](int, int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        order_out[i] = i;
        const int32_t p = obs_pt[i], prev = i > 0 ? obs_pt[i - 1] : -1;
        for (int32_t q = prev + 1; q <= p; ++q) pt_start_out[q] = i;
      }
    });
    for (int32_t q = (n_obs > 0 ? obs_pt[n_obs - 1] : -1) + 1; q <= n_points; ++q) pt_start_out[q] = n_obs;
  } else {
    // optional first key: camera (stable counting sort), so that the final order is (point, camera, input order)
    std::vector<int64_t> by_cam;
    if (with_cam) {
      std::vector<int64_t> cc((size_t)n_cams + 1, 0);
      for (int64_t i = 0; i < n_obs; ++i) cc[obs_cam[i] + 1]++;
      for (int32_t c = 0; c < n_cams; ++c) cc[c + 1] += cc[c];
      by_cam.resize((size_t)n_obs);
      for (int64_t i = 0; i < n_obs; ++i) by_cam[cc[obs_cam[i]]++] = i;
    }
    std::vector<int64_t> count((size_t)n_points + 1, 0);
    for (int64_t i = 0; i < n_obs; ++i) count[obs_pt[i] + 1]++;
    for (int32_t p = 0; p < n_points; ++p) count[p + 1] += count[p];
    for (int32_t p = 0; p <= n_points; ++p) pt_start_out[p] = count[p];
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
