// libcaliscope_ba.so — host side of the MI355X bundle-adjustment engine (C ABI of include/caliscope_ba.h).
//
// One cba_problem owns: the observations re-sorted by world point (SoA, HBM-resident for the whole solve),
// the chunk table, every vector of the trust-region iteration and the scratch of the Schur solve.  Every
// entry point enqueues its kernels on the handle's own HIP stream, waits once, and returns scalars.
// There is no CPU code path for the arithmetic: without a device, cba_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <memory>
#include <string>
#include <type_traits>
#include <thread>
#include <utility>
#include <vector>

#include <rccl/rccl.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include "../../include/caliscope_ba.h"
#include "cba_kernels.h"
#include "schur_plan.h"
#include "host_plan.h"
#include "wg_binding.h"

using namespace cba;

// sha256 over the sources this library was built from (caliscope_amd/build.py: source_digest), as a string inside the binary:
// __graft_entry__.smoke() and tests/test_library_abi.py look for it, so a stale .so next to newer sources is the driver's finding.
#ifndef CBA_SOURCE_DIGEST
#define CBA_SOURCE_DIGEST "unknown"
#endif
extern "C" __attribute__((used, visibility("hidden"))) const char cba_source_digest_marker[] = "CBA_SOURCE_DIGEST=" CBA_SOURCE_DIGEST;

// ---------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess)                                                                              \
      return fail(CBA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

enum TimerId {
  T_CAM_PREP = 0, T_COST, T_BUILD, T_BUILD_REDUCE, T_SCALE_SCALARS, T_JV, T_SCHUR, T_SCHUR_REDUCE, T_CHOLESKY,
  T_BACKSUB, T_VECTOR, T_SCHUR_PAIRS, T_EXCHANGE, T_COUNT
};
static const char* kTimerNames[T_COUNT] = {
    "cam_prep", "cost", "build", "build_reduce", "scale_scalars", "jv", "schur", "schur_reduce_finalize",
    "cholesky_solve", "backsub", "vector_ops", "schur_pairs", "exchange"};

struct EventPair { hipEvent_t a, b; };
struct cba_group;

struct cba_problem {
  int device = 0;
  hipStream_t stream = nullptr;
  int C = 0, P = 0, ncp = 0, nct = 6;
  long N = 0;
  VecLayout lay{};
  int n_chunks = 0, grid = 0, max_obs_per_point = 0;
  int G = 1, gsz = 1, n_tiles = 1, n_tile_chunks = 0, tile_grid = 0;
  int cus = 256;           // compute units of the device
  double plan_lane_util = 0.0;  // share of the lane-iterations of the pair loops that multiply a real pair
  long tile_stream_len = 0, n_pairs = 0;
  TilePlan tp{};
  int* tile_wg_begin = nullptr;
  int loss = 0;
  double f_scale = 1.0;
  long device_bytes = 0;
  // observations (sorted by point) + plan
  double *obs_u = nullptr, *obs_v = nullptr;
  int *obs_cam = nullptr, *obs_pt = nullptr, *pt_start = nullptr, *chunk_start = nullptr, *order = nullptr;
  int* chunk_pts = nullptr;  // [n_chunks][2] first point and number of points (observed or not) in a chunk's range
  // cameras
  double* cam_const = nullptr;
  int *cam_model = nullptr, *cam_np = nullptr, *cam_off = nullptr, *param_cam = nullptr, *param_loc = nullptr;
  double *tab = nullptr, *tab_new = nullptr;
  // vectors
  double *x0 = nullptr, *x = nullptr, *x_new = nullptr, *g = nullptr, *s = nullptr, *sinv = nullptr, *v1 = nullptr, *v2 = nullptr;
  double *V = nullptr, *Upacked = nullptr;
  double *partial = nullptr, *partial4 = nullptr, *partial1 = nullptr;
  long partial_width = 0;
  bool want_chol_trace = false;
  long long* chol_trace = nullptr;  // CBA_CHOL_TRACE=1: phase stamps of k_chol_step (tools/chol_trace.py)
  int ldw = 0;             // row stride of the Cholesky work matrix Lbuf (multiple of 4 doubles)
  double *Sacc = nullptr, *S = nullptr, *Lbuf = nullptr, *rhs = nullptr, *red = nullptr, *Trec = nullptr, *partial_b = nullptr;
  CsPlan cs{};              // camera-sorted super-chunks of the build pass (k_build_cs); cs.n_sc == 0: k_build (CBA_BUILD_CS=0, deterministic sums, fragments)
  bool tab_global = false;  // the per-observation kernels read the camera table from global memory (CAMG variants): its LDS copy would not fit
  int det_m = 0;  // cba_options.deterministic: tasks per thread of the fixed-order camera sums (3, 5, 8 or 16; 0: atomics)
  DetPlan det{nullptr, nullptr};
  double* tri = nullptr;   // packed upper triangle of Sacc + b for the exchange of a sharded solve
  double* Xinv = nullptr;  // inverses of the diagonal blocks of the Cholesky factor, [blocks][NB][NB] (k_chol_step)
  double* Tinv = nullptr;  // T = L^-T, built block by block next to the factorisation (inverse role of k_chol_step)
  bool fuse_reg_finalize = true;  // k_reg_finalize in the place of k_reg_reduce + k_schur_finalize where the route allows (CBA_REG_FINALIZE=0: off)
  double* scal = nullptr;  // device scalars
  double* xbuf = nullptr;  // staging of the one all-reduce per primitive (sharded solves)
  double *sinv_state_c = nullptr, *cam_diag = nullptr, *cam_over1 = nullptr, *cam_over2 = nullptr;  // [ncp_pad] each (cba_set_camera_scaling)
  bool cam_scaled = false, cam_state_saved = false;
  // fused iteration (cba_step): device scalars [lam, radius, alpha, beta], second set of build outputs for the trial point
  double *fz = nullptr, *V2 = nullptr, *g2 = nullptr, *U2 = nullptr, *partial4b = nullptr;
  bool eval_only = false;  // cba_options.evaluation_only: no Schur plan, no solver buffers (residual hook, begin / trial costs)
  bool peer_needs_primitives = false;  // sharded solves: some rank cannot run cba_step, so none does
  bool have_build = false;   // V, g, Upacked are valid at the current x (a trial built by cba_step was accepted)
  bool trial_built = false;  // the pending trial point carries its own build in V2, g2, U2
  bool cost_pending = false; // the build of this linearisation ran here: its rho sum (scal[8]) is the cost at x
  double cost_x = 0.0;       // cost at the current x
  double trial_cost = 0.0;   // cost at the pending trial point
  int grid_backsub = 1;  // k_backsub keeps 34 KB of LDS (cfg4): more resident workgroups than the 57-65 KB kernels sharing p->grid
  int jv_grid = 0;       // workgroups of k_jv: what the device holds at once (its 150 registers per thread admit three 256-thread workgroups per CU)
  bool has_fragments = false;  // some chunk is a fragment of a point with more than CHUNK observations (k_backsub adds its sums by atomics)
  bool backsub_rec = false;    // the back-substitution streams the T records (k_backsub_rec) instead of linearising every observation again
  int n_heavy = 0; int* heavy_pts = nullptr; int* heavy_frag = nullptr; double* heavy_W = nullptr;  // heavy points (k_heavy_schur)
  std::vector<int> h_heavy_pts;
  ConPlan con{};           // rigid-distance constraint rows (cba_set_constraints); con.n_con == 0: none
  int con_grid = 0;        // workgroups of the per-constraint kernels
  int* flags = nullptr;
  double* h_scal = nullptr;  // pinned, mapped: k_publish writes it (d_hscal is the same memory seen from the device)
  int* h_flags = nullptr;
  double* d_hscal = nullptr;
  double *h_cam = nullptr, *d_hcam = nullptr;  // pinned, mapped: camera blocks of up to three vectors + a sequence word
  double *h_bcam = nullptr, *d_hbcam = nullptr;  // pinned, mapped: the four camera blocks a bounded fused iteration sends with its packet
  // bounded fused iteration (cba_set_bounds): bounds of the camera block on the device, second copies of the Jacobi state and of the Coleman-Li
  // diagonal for the speculative linearisation of the trial point
  double *lb_dev = nullptr, *ub_dev = nullptr, *sinv_state_c2 = nullptr, *cam_diag2 = nullptr;
  bool bounds_on = false;
  unsigned long long publish_seq = 0;
  bool spin_wait = true;  // CBA_SPIN=0: sleep in hipStreamSynchronize instead
  int* d_hflags = nullptr;
  bool first_scale = true;
  // Speculative linearisation (single-rank fused iterations): right behind the k_publish of a cba_step — while the host reads the packet, decides
  // and enqueues the next iteration — the Jacobi scale, the vector sums and ||J_h g_h||^2 of the TRIAL point are formed as if it were accepted (it
  // usually is), into shadow outputs: sinv2 instead of sinv, the reduction partials.  cba_accept makes them current; the next cba_step then starts
  // with k_lin_finish.  A rejected trial just leaves ~50 us of device work unused.
  double* sinv2 = nullptr;
  bool spec_enqueued = false, spec_valid = false;
  int spec_rows_jv = 0;
  bool lf_pending = false;  // the reductions of the linearisation and the damping are left to the next k_tprep (LinFin lf)
  LinFin lf{};
  bool have_x0 = false;
  std::vector<int> h_tile_wg_begin;  // host copy (profiling print)
  struct PlanTask* plan_task = nullptr;  // two-stage plan: the thread still dealing the Schur plan while the handle works with the cheap one
  int reg_reduce_y = 4;                  // y extent of k_reg_reduce's workgroups (4 or 16: by the partial rows per tile)
  int plan_max_blocks = 0;               // workgroup budget of the pair kernel (the dealt plan is bound with the same one when it is swapped in)
  size_t partial_capacity = 0;           // doubles behind `partial`
  bool plan_is_cheap = false;
  int plan_error = 0;  // CBA_ERR_* of a failed background plan build (the handle keeps the quick plan)
  bool schur_clock = false;  // profiling build only (-DCBA_PROFILING, CBA_SCHUR_CLOCK=1): phase clocks of k_schur_reg3
  long long* stamps = nullptr;  // profiling build only (CBA_STAMPS=1): device-side entry / exit stamps of the kernels of the last fused iteration
  bool begun = false, linearized = false, stepped = false, have_trial = false;
  double gh_sq = 0.0;
  std::vector<int> h_cam_off, h_cam_np;
  double* h_vec = nullptr;     // pinned staging for layout conversion (vectors in and out); pageable, a hipMemcpyAsync of cfg4's 4.8 MB x0 waited
                               // 25-40 ms inside the runtime whenever the plan thread's workers were faulting their arrays in beside it
  size_t h_vec_doubles = 0;
  hipEvent_t h_vec_sent = nullptr;  // recorded behind an asynchronous copy OUT of h_vec: the next writer of h_vec waits for it (stage_wait)
  bool h_vec_in_flight = false;
  // timers
  bool timers_on = false;
  std::vector<EventPair> pending[T_COUNT];
  std::vector<EventPair> free_events;
  double t_ms[T_COUNT] = {0};
  long t_calls[T_COUNT] = {0};
  std::vector<void*> allocs;   // arena chunks (dev_alloc)
  std::vector<size_t> alloc_bytes;
  size_t mail_doubles = 0;     // capacity of the mapped host mailbox (h_scal)
  char* arena_cur = nullptr; size_t arena_left = 0, arena_next = (size_t)4 << 20;
  // small uploads of cba_create / cba_set_constraints go through a pooled pinned buffer and the handle's stream (one wait at the end) instead of ~40
  // synchronous hipMemcpy calls of 10-20 us each: most of the millisecond a handle for the reference's own 4-camera session took to build
  char* up_stage = nullptr; size_t up_cap = 0, up_used = 0;
  // sharded solve (points partitioned over ranks, cameras replicated): RCCL over xGMI
  std::atomic<ncclComm_t> comm{nullptr};
  std::atomic<bool> comm_aborted{false};  // cba_comm_abort was called (from another thread): every collective of this handle fails from now on
  cba_group* group = nullptr;  // in-process device group (cba_group_join): direct peer-to-peer exchange instead of RCCL
  int rank = 0, world = 1;
  unsigned long long group_generation = 0;
  bool sharded() const { return comm.load() != nullptr || group != nullptr || comm_aborted.load(); }
};

// What a small handle is made of is recycled: per device the library keeps up to four first arena chunks (4 MB), streams and mapped host
// mailboxes of destroyed handles and hands them to the next cba_create.  hipMalloc / hipFree / hipStreamCreate / hipHostMalloc are
// 0.3 - 1 ms each; on the reference's own 4-camera session creating and destroying a handle took 3.5 + 2.4 ms next to a 0.7 ms solve.
constexpr size_t kPoolChunk = (size_t)4 << 20;
constexpr size_t kPoolKeep = 4;
constexpr size_t kPoolStagingMax = (size_t)8 << 20;  // doubles: staging buffers above 64 MB are not kept
struct DevicePool {
  std::vector<void*> chunks;
  std::vector<hipStream_t> streams;
  std::vector<std::pair<double*, size_t>> mail;  // (mapped host pointer, doubles)
  std::vector<std::pair<double*, size_t>> staging;  // (pinned host pointer, doubles): h_vec
  std::vector<char*> upstage;                      // pinned buffers of kUpStage bytes for small uploads
};
constexpr size_t kUpStage = (size_t)256 << 10, kUpStageMax = (size_t)64 << 10;
static std::mutex g_pool_mu;
static std::map<int, DevicePool> g_pool;

// Device memory of a handle comes from an arena: a handle has ~60 buffers, and on the reference's own 4-camera session
// creating and freeing them one hipMalloc / hipFree at a time was more than half of an optimize() call (the solve itself takes
// 1 ms).  Buffers are carved out of chunks (the first 4 MB, then doubling; a large buffer gets a chunk of its own), 256-byte
// aligned; cba_destroy frees the chunks.
template <typename T>
static int dev_alloc(cba_problem* p, T** out, size_t count) {
  const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
  if (bytes > p->arena_left) {
    const size_t chunk = std::max(bytes, p->arena_next);
    void* ptr = nullptr;
    if (chunk == kPoolChunk) {
      std::lock_guard<std::mutex> lock(g_pool_mu);
      DevicePool& pool = g_pool[p->device];
      if (!pool.chunks.empty()) { ptr = pool.chunks.back(); pool.chunks.pop_back(); }
    }
    if (!ptr) HIPCHK(hipMalloc(&ptr, chunk));
    p->allocs.push_back(ptr);
    p->alloc_bytes.push_back(chunk);
    if (bytes >= p->arena_next) {  // a buffer of its own: the open chunk stays open
      p->device_bytes += (long)bytes;
      *out = static_cast<T*>(ptr);
      return CBA_OK;
    }
    p->arena_cur = static_cast<char*>(ptr);
    p->arena_left = chunk;
    p->arena_next = std::min<size_t>(p->arena_next * 2, (size_t)256 << 20);
  }
  *out = reinterpret_cast<T*>(p->arena_cur);
  p->arena_cur += bytes;
  p->arena_left -= bytes;
  p->device_bytes += (long)bytes;
  return CBA_OK;
}
template <typename T, typename V>
static int dev_upload(cba_problem* p, T** out, const V& h) {  // V: any contiguous host array of T (std::vector, HostVec, RawVec)
  static_assert(std::is_same<typename std::remove_cv<typename std::remove_pointer<decltype(h.data())>::type>::type, T>::value, "element type");
  int rc = dev_alloc(p, out, h.size());
  if (rc) return rc;
  const size_t bytes = h.size() * sizeof(T);
  if (bytes && p->up_stage && p->stream && bytes <= kUpStageMax && p->up_used + bytes <= p->up_cap) {
    std::memcpy(p->up_stage + p->up_used, h.data(), bytes);
    HIPCHK(hipMemcpyAsync(*out, p->up_stage + p->up_used, bytes, hipMemcpyHostToDevice, p->stream));
    p->up_used += (bytes + 63) & ~(size_t)63;
  } else if (bytes) {
    HIPCHK(hipMemcpy(*out, h.data(), bytes, hipMemcpyHostToDevice));
  }
  return CBA_OK;
}
// a pinned buffer for the small uploads of one set-up call (cba_create, cba_set_constraints); given back — after the stream has been waited for — by
// upload_stage_release
static void upload_stage_acquire(cba_problem* p) {
  if (p->up_stage) return;
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    DevicePool& pool = g_pool[p->device];
    if (!pool.upstage.empty()) { p->up_stage = pool.upstage.back(); pool.upstage.pop_back(); }
  }
  if (!p->up_stage && hipHostMalloc((void**)&p->up_stage, kUpStage, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p->up_stage = nullptr; }
  p->up_cap = p->up_stage ? kUpStage : 0;
  p->up_used = 0;
}
static void upload_stage_release(cba_problem* p) {  // the caller has synchronised the stream (or the device)
  if (!p->up_stage) return;
  std::lock_guard<std::mutex> lock(g_pool_mu);
  DevicePool& pool = g_pool[p->device];
  if (pool.upstage.size() < kPoolKeep) pool.upstage.push_back(p->up_stage); else (void)hipHostFree(p->up_stage);
  p->up_stage = nullptr; p->up_cap = p->up_used = 0;
}

// h_vec is pinned: a hipMemcpyAsync out of it returns while the copy engine still reads it
static int stage_wait(cba_problem* p) {
  if (p->h_vec_in_flight) { HIPCHK(hipEventSynchronize(p->h_vec_sent)); p->h_vec_in_flight = false; }
  return CBA_OK;
}
static int stage_sent(cba_problem* p) {
  HIPCHK(hipEventRecord(p->h_vec_sent, p->stream));
  p->h_vec_in_flight = true;
  return CBA_OK;
}

// std::vector without value-initialisation of its elements: the threads that fill the observation-sized host arrays of cba_create are the first to
// touch their pages (zero-filling ~70 bytes per observation on the calling thread was a third of "reorder on host")
template <typename T>
struct NoInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = NoInitAlloc<U>; };
  NoInitAlloc() = default;
  template <typename U> NoInitAlloc(const NoInitAlloc<U>&) {}
  template <typename U, typename... A> void construct(U* ptr, A&&... args) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)ptr) U; else ::new ((void*)ptr) U(std::forward<A>(args)...);
  }
  // large arrays from the pool of huge-page blocks the plan's arrays use (schur_plan.h: a handle's set-up was paying for the page faults of ~100 MB of
  // freshly mapped host memory per call)
  T* allocate(size_t n) {
    void* ptr = rawvec_detail::acquire_tracked(n * sizeof(T));
    if (!ptr) throw std::bad_alloc();
    return static_cast<T*>(ptr);
  }
  void deallocate(T* ptr, size_t n) noexcept { rawvec_detail::release_tracked(ptr, n * sizeof(T)); }
};
template <typename T> using HostVec = std::vector<T, NoInitAlloc<T>>;

// roctx range around the host-side enqueue of a phase: `rocprofv3 --marker-trace --kernel-trace` shows the kernels of an
// iteration under build / linearize / schur / cholesky / backsub / trial (free when no profiler is attached)
struct RoctxRange {
  explicit RoctxRange(const char* name) { roctxRangePushA(name); }
  ~RoctxRange() { roctxRangePop(); }
};

struct ScopedTimer {
  cba_problem* p;
  int id;
  EventPair ev{};
  bool on;
  ScopedTimer(cba_problem* p_, int id_) : p(p_), id(id_), on(p_->timers_on) {
    if (!on) return;
    if (!p->free_events.empty()) { ev = p->free_events.back(); p->free_events.pop_back(); }
    else { (void)hipEventCreate(&ev.a); (void)hipEventCreate(&ev.b); }
    (void)hipEventRecord(ev.a, p->stream);
  }
  ~ScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(ev.b, p->stream);
    p->pending[id].push_back(ev);
  }
};

static void drain_timers(cba_problem* p) {
  for (int t = 0; t < T_COUNT; ++t) {
    for (auto& ev : p->pending[t]) {
      float ms = 0.f;
      if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
        p->t_ms[t] += ms;
        p->t_calls[t] += 1;
      }
      p->free_events.push_back(ev);
    }
    p->pending[t].clear();
  }
}

// part_a / part_b: per-workgroup partial columns whose sums belong to scal[slot_a] / scal[slot_b] (compact fused step)
static unsigned long long publish_enqueue(cba_problem* p, int n_scal, const double* part_a = nullptr, int rows_a = 0, int slot_a = 0,
                                          const double* part_b = nullptr, int rows_b = 0, int slot_b = 0) {
  const unsigned long long seq = ++p->publish_seq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(BLOCK), 0, p->stream, p->scal, n_scal, p->flags, p->d_hscal, p->d_hflags, seq, part_a, rows_a,
                     slot_a, part_b, rows_b, slot_b);
  return seq;
}
// wait until the k_publish with sequence number `seq` has written its packet.  `drain`: nothing was enqueued behind it, a stream synchronize
// is equivalent (the CBA_SPIN=0 route); with speculative work behind the publish only the sequence word says when the packet is there.
static int publish_wait(cba_problem* p, unsigned long long seq, bool drain = true) {
  if (p->spin_wait || !drain) {
    // the solver owns this host thread anyway: poll the sequence number k_publish writes last (a few hundred ns per poll of
    // pinned memory) rather than sleep in hipStreamSynchronize and pay its wake-up latency once per iteration
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(p->h_scal) + 63;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *flag != seq; ++spins) {
      if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHK(hipStreamSynchronize(p->stream));  // something is wrong or very slow: let the runtime report it
        if (*flag != seq) return fail(CBA_ERR_HIP, "k_publish did not complete");
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return CBA_OK;
  }
  HIPCHK(hipStreamSynchronize(p->stream));
  return CBA_OK;
}
static int sync_scalars(cba_problem* p, int n_scal, const double* part_a = nullptr, int rows_a = 0, int slot_a = 0,
                        const double* part_b = nullptr, int rows_b = 0, int slot_b = 0) {
  // every primitive ends here: scalars and flags to the host, flags cleared for the next primitive
  return publish_wait(p, publish_enqueue(p, n_scal, part_a, rows_a, slot_a, part_b, rows_b, slot_b));
}

#define NCCLCHK(expr)                                                                                  \
  do {                                                                                                 \
    ncclResult_t _r = (expr);                                                                          \
    if (_r != ncclSuccess) return fail(CBA_ERR_COMM, "%s failed: %s", #expr, ncclGetErrorString(_r));  \
  } while (0)

// In-process device group: the ranks are handles of ONE process, one host thread each (cba_group_join).  The exchange is
// direct: every rank parks its contribution in a staging buffer of its own device, then sums the staging buffers of all
// ranks in rank order — its own and, through peer access over xGMI, everybody else's — so all ranks end with the same
// bits and no ring is involved ("direct ... over the 7 xGMI links, never a ring for these sizes").
// Ordering across the streams: events.  Collective g uses staging buffer g & 1:
//   rank r, stream r:  wait done_q[g & 1] of all q (their reads of generation g - 2)  ->  copy  ->  record ready_r[g & 1]
//   host:              barrier (every rank has recorded its ready event)
//   rank r, stream r:  wait ready_q[g & 1] of all q  ->  sum kernel  ->  record done_r[g & 1]
// The one host barrier per collective also orders generation g + 2's waits behind generation g's records.
struct cba_group {
  int world = 0;
  std::atomic<int> arrived{0};
  std::atomic<unsigned> phase{0};
  std::atomic<int> joined{0}, failed{0}, aborted{0};
  std::vector<cba_problem*> member;
  std::vector<double*> stage[2];
  std::vector<hipEvent_t> ready[2], done[2];
  size_t capacity = 0;  // doubles per staging buffer
  int timeout_s = 120;  // CBA_GROUP_TIMEOUT_S
  unsigned long long generation = 0;  // advanced by rank 0 behind the barrier; every rank keeps its own copy in step
  const double** d_src[2] = {nullptr, nullptr};  // per rank (device memory of that rank): the `world` staging pointers of a parity
  std::vector<const double**> src_tab[2];
  // sense-reversing spin barrier: the member threads are dedicated to their devices and meet every few hundred microseconds
  // returns false when the group was aborted (cba_group_abort: a member failed and will never arrive)
  bool barrier() {
    const unsigned ph = phase.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == world) {
      arrived.store(0, std::memory_order_relaxed);
      phase.store(ph + 1, std::memory_order_release);
    } else {
      long spins = 0;
      std::chrono::steady_clock::time_point t0;
      while (phase.load(std::memory_order_acquire) == ph) {
        if (aborted.load(std::memory_order_acquire)) return false;
        if (++spins > 2000) {
          std::this_thread::yield();
          // watchdog: a rank that never arrives (it failed, or the ranks disagree about the next collective) must not
          // leave the others spinning for ever
          if (spins == 2001) t0 = std::chrono::steady_clock::now();
          else if ((spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
            aborted.store(1, std::memory_order_release);
            return false;
          }
        }
      }
    }
    return !aborted.load(std::memory_order_acquire);
  }
};
constexpr int GROUP_MAX = 16;

struct GroupSrc { const double* p[GROUP_MAX]; };
__global__ void __launch_bounds__(256)
k_group_sum(GroupSrc src, int world, size_t count, double* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    double s = src.p[0][i];
    for (int q = 1; q < world; ++q) s += src.p[q][i];  // rank order: identical bits on every rank
    out[i] = s;
  }
}

static int group_allreduce(cba_problem* p, double* buf, size_t count) {
  cba_group* g = p->group;
  if (count > g->capacity) return fail(CBA_ERR_INVALID, "group exchange of %zu doubles exceeds the staging capacity %zu", count, g->capacity);
  const int r = p->rank, par = (int)(p->group_generation & 1ull);
  for (int q = 0; q < g->world; ++q)
    if (q != r) HIPCHK(hipStreamWaitEvent(p->stream, g->done[par][q], 0));
  HIPCHK(hipMemcpyAsync(g->stage[par][r], buf, count * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  HIPCHK(hipEventRecord(g->ready[par][r], p->stream));
  if (!g->barrier())
    return fail(CBA_ERR_INVALID, "device group aborted at collective %llu of rank %d (%zu doubles): another rank failed or never arrived",
                p->group_generation, r, count);
  GroupSrc src;
  for (int q = 0; q < g->world; ++q) {
    src.p[q] = g->stage[par][q];
    if (q != r) HIPCHK(hipStreamWaitEvent(p->stream, g->ready[par][q], 0));
  }
  const int grid = (int)std::max<size_t>(1, std::min<size_t>((count + 255) / 256, 256));
  hipLaunchKernelGGL(k_group_sum, dim3(grid), dim3(256), 0, p->stream, src, g->world, count, buf);
  HIPCHK(hipEventRecord(g->done[par][r], p->stream));
  p->group_generation++;
  return CBA_OK;
}

// in-place sum over the ranks of a sharded solve, enqueued on the engine's stream (no-op for world 1)
static int allreduce_sum(cba_problem* p, double* buf, size_t count) {
  if (p->comm_aborted.load(std::memory_order_acquire)) return fail(CBA_ERR_COMM, "the communicator of rank %d was aborted (another rank failed)", p->rank);
  ncclComm_t comm = p->comm.load();
  if (!p->group && !comm) return CBA_OK;
  ScopedTimer t(p, T_EXCHANGE);  // nested inside the family that needs the sum: comm time per step, reported next to the families
  if (p->group) return group_allreduce(p, buf, count);
  NCCLCHK(ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, p->stream));
  return CBA_OK;
}
// one all-reduce for the host-visible scalars of a primitive: scal slots in `mask`, the flags, optionally max |g|
// `buf` / `prefix`: the packed scalars ride at the tail of a payload that needs the same sum (buf[0 .. prefix)), so
// that payload and scalars cost one collective; buf must have room for prefix + 64 + world doubles.
static int exchange_at(cba_problem* p, unsigned long long mask, bool with_max, double* buf, size_t prefix) {
  if (!p->sharded()) return CBA_OK;
  hipLaunchKernelGGL(k_xpack, dim3(1), dim3(64), 0, p->stream, p->scal, p->flags, mask, with_max ? 1 : 0, p->rank, p->world, buf + prefix);
  const size_t n = (size_t)__builtin_popcountll(mask) + 4 + (with_max ? p->world : 0);
  const int rc = allreduce_sum(p, buf, prefix + n);
  if (rc) return rc;
  hipLaunchKernelGGL(k_xunpack, dim3(1), dim3(64), 0, p->stream, buf + prefix, mask, with_max ? 1 : 0, p->world, p->scal, p->flags);
  return CBA_OK;
}
static int exchange(cba_problem* p, unsigned long long mask, bool with_max) { return exchange_at(p, mask, with_max, p->xbuf, 0); }
#define SLOT(i) (1ull << (i))

static inline int vec_grid(long total) { return (int)std::min<long>((total + BLOCK - 1) / BLOCK, 1024); }

// ---------------------------------------------------------------------------------------------------
extern "C" {

int cba_triangulate(const cba_triangulate_desc* d, int32_t device, double* xyz_out, double* undistorted_out) {
  if (!d || !xyz_out) return fail(CBA_ERR_INVALID, "cba_triangulate: null argument");
  if (d->n_cams <= 0 || d->n_points < 0 || !d->cam_P || (d->n_points > 0 && (!d->pt_start || !d->obs_cam || !d->obs_xy)))
    return fail(CBA_ERR_INVALID, "cba_triangulate: bad descriptor");
  if (d->cam_intr && !d->cam_model) return fail(CBA_ERR_INVALID, "cba_triangulate: cam_intr given without cam_model");
  if (d->n_points == 0) return CBA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(CBA_ERR_NO_DEVICE, "cba_triangulate: no HIP device");
  if (device < 0 || device >= ndev) return fail(CBA_ERR_INVALID, "cba_triangulate: device %d of %d", device, ndev);
  HIPCHK(hipSetDevice(device));
  const int64_t n_obs = d->pt_start[d->n_points];
  for (int64_t i = 0; i < n_obs; ++i)
    if (d->obs_cam[i] < 0 || d->obs_cam[i] >= d->n_cams) return fail(CBA_ERR_INVALID, "cba_triangulate: obs_cam[%lld] out of range", (long long)i);
  std::vector<void*> bufs;
  auto cleanup = [&]() { for (void* b : bufs) (void)hipFree(b); };
  auto up = [&](const void* src, size_t bytes, void** dst) -> int {
    void* ptr = nullptr;
    if (hipMalloc(&ptr, std::max<size_t>(bytes, 8)) != hipSuccess) return CBA_ERR_HIP;
    bufs.push_back(ptr);
    if (src && bytes && hipMemcpy(ptr, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return CBA_ERR_HIP;
    *dst = ptr;
    return CBA_OK;
  };
  void *dps = nullptr, *dcam = nullptr, *dxy = nullptr, *dmodel = nullptr, *dintr = nullptr, *dP = nullptr, *dxyz = nullptr, *dund = nullptr;
  int rc = up(d->pt_start, (size_t)(d->n_points + 1) * sizeof(int64_t), &dps);
  if (!rc) rc = up(d->obs_cam, (size_t)n_obs * sizeof(int32_t), &dcam);
  if (!rc) rc = up(d->obs_xy, (size_t)n_obs * 2 * sizeof(double), &dxy);
  if (!rc) rc = up(d->cam_P, (size_t)d->n_cams * 12 * sizeof(double), &dP);
  if (!rc && d->cam_intr) rc = up(d->cam_intr, (size_t)d->n_cams * 9 * sizeof(double), &dintr);
  if (!rc && d->cam_intr) rc = up(d->cam_model, (size_t)d->n_cams * sizeof(int32_t), &dmodel);
  if (!rc) rc = up(nullptr, (size_t)d->n_points * 3 * sizeof(double), &dxyz);
  if (!rc && undistorted_out) rc = up(nullptr, (size_t)n_obs * 2 * sizeof(double), &dund);
  if (rc) { cleanup(); return fail(CBA_ERR_HIP, "cba_triangulate: device allocation / upload failed"); }
  static_assert(sizeof(long) == sizeof(int64_t), "pt_start is passed as long");
  const int grid = (int)((d->n_points + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL(k_triangulate, dim3(grid), dim3(BLOCK), 0, 0, (long)d->n_points, (const long*)dps, (const int*)dcam, (const double*)dxy,
                     (const int*)dmodel, (const double*)dintr, (const double*)dP, d->float32_io ? 1 : 0, (double*)dxyz, (double*)dund);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(xyz_out, dxyz, (size_t)d->n_points * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && undistorted_out) e = hipMemcpy(undistorted_out, dund, (size_t)n_obs * 2 * sizeof(double), hipMemcpyDeviceToHost);
  cleanup();
  if (e != hipSuccess) return fail(CBA_ERR_HIP, "cba_triangulate: %s", hipGetErrorString(e));
  return CBA_OK;
}

int cba_set_error(int32_t code, const char* message) { return fail(code, "%s", message ? message : ""); }

const char* cba_last_error(void) { return g_last_error.c_str(); }
int cba_version(void) { return CBA_VERSION; }
int cba_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int64_t cba_trim(void) {
  int64_t bytes = (int64_t)rawvec_detail::trim();
  std::map<int, DevicePool> pools;
  { std::lock_guard<std::mutex> lock(g_pool_mu); pools.swap(g_pool); }
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (auto& kv : pools) {
    if (hipSetDevice(kv.first) != hipSuccess) continue;
    DevicePool& pool = kv.second;
    for (void* c : pool.chunks) { (void)hipFree(c); bytes += (int64_t)kPoolChunk; }
    for (hipStream_t st : pool.streams) (void)hipStreamDestroy(st);
    for (auto& m : pool.mail) { (void)hipHostFree(m.first); bytes += (int64_t)(m.second * sizeof(double)); }
    for (auto& m : pool.staging) { (void)hipHostFree(m.first); bytes += (int64_t)(m.second * sizeof(double)); }
    for (char* u : pool.upstage) { (void)hipHostFree(u); bytes += (int64_t)kUpStage; }
  }
  if (have_cur) (void)hipSetDevice(cur);
  return bytes;
}

int cba_timer_count(void) { return T_COUNT; }
const char* cba_timer_name(int32_t i) { return (i >= 0 && i < T_COUNT) ? kTimerNames[i] : ""; }

int64_t cba_host_plan(int32_t n_points, int64_t n_obs, const int32_t* obs_pt, const int32_t* obs_cam, int32_t n_cams,
                      int32_t chunk_cap, int64_t* order_out, int64_t* pt_start_out, int64_t* chunk_start_out) {
  return host_plan_impl([](int code, const char* fmt, auto... args) { return fail(code, fmt, args...); }, n_points, n_obs, obs_pt, obs_cam, n_cams, chunk_cap,
                        order_out, pt_start_out, chunk_start_out);
}

static void drop_plan_task(cba_problem* p);  // (PlanTask is defined with the plans further down)

#ifdef CBA_PROFILING
// CBA_STAMPS=1 (profiling build): the kernels of the LAST fused iteration in the order they started, with the device's own clock — duration from the
// first workgroup's entry to the last wave's exit, and the idle time in front of each (previous kernel's last exit -> this kernel's first entry).
static void dump_stamps(cba_problem* p) {
  static const char* names[] = {"k_tprep", "k_schur_reg3", "k_reg_reduce (or k_reg_finalize)", "k_schur_finalize", "k_chol_apply", "k_backsub", "k_step_cam", "k_build_cs",
                                "k_reduce_rows_pub", "k_scale_lin", "k_jv", "k_small_solve"};
  const size_t n = (size_t)STAMP_SLOTS * STAMP_BLOCKS * STAMP_ROW;
  std::vector<long long> h(n);
  long long* null_ptr = nullptr;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cba_stamps), &null_ptr, sizeof(null_ptr));
  if (hipMemcpy(h.data(), p->stamps, n * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return;
  struct Row { int slot; long long t0, t1, first_exit, last_entry; int blocks; double life_sum; long long life_min, life_max; };
  std::vector<Row> rows;
  for (int s = 0; s < STAMP_SLOTS; ++s) {
    Row r{s, 0, 0, 0, 0, 0, 0.0, 0, 0};
    for (int b = 0; b < STAMP_BLOCKS; ++b) {
      const long long* q = &h[((size_t)s * STAMP_BLOCKS + b) * STAMP_ROW];
      if (!q[0]) continue;
      r.blocks++;
      r.t0 = r.t0 ? std::min(r.t0, q[0]) : q[0];
      long long e = 0;
      for (int w = 0; w < STAMP_WAVES; ++w) e = std::max(e, q[1 + w]);
      r.t1 = std::max(r.t1, e);
      r.first_exit = r.first_exit ? std::min(r.first_exit, e) : e;
      r.last_entry = std::max(r.last_entry, q[0]);
      const long long life = e - q[0];
      r.life_sum += (double)life; r.life_min = r.life_min ? std::min(r.life_min, life) : life; r.life_max = std::max(r.life_max, life);
    }
    if (r.blocks) rows.push_back(r);
  }
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.t0 < b.t0; });
  // the last iteration: from the last k_tprep on
  size_t first = 0;
  for (size_t i = 0; i < rows.size(); ++i) if (rows[i].slot == ST_TPREP) first = i;
  fprintf(stderr, "device stamps of the last fused iteration (100 MHz wall clock; blocks = workgroups seen, at most %d):\n", STAMP_BLOCKS);
  fprintf(stderr, "| # | kernel | workgroups | first entry -> last exit, us | last entry after the first, us | workgroup lifetime min / mean / max, us | idle before, us |\n|---|---|---|---|---|---|---|\n");
  double busy = 0.0, idle = 0.0;
  for (size_t i = first; i < rows.size(); ++i) {
    const Row& r = rows[i];
    char nm[48];
    if (r.slot >= ST_CHOL_STEP) snprintf(nm, sizeof nm, "k_chol_step k=%d", r.slot - ST_CHOL_STEP - 1); else snprintf(nm, sizeof nm, "%s", names[r.slot]);
    const double gap = i > first ? (r.t0 - rows[i - 1].t1) * 0.01 : 0.0;
    busy += (r.t1 - r.t0) * 0.01; if (i > first) idle += gap;
    fprintf(stderr, "| %zu | %s | %d | %.2f | %.2f | %.2f / %.2f / %.2f | %.2f |\n", i - first + 1, nm, r.blocks, (r.t1 - r.t0) * 0.01, (r.last_entry - r.t0) * 0.01,
            r.life_min * 0.01, r.life_sum / r.blocks * 0.01, r.life_max * 0.01, gap);
  }
  if (first < rows.size())
    fprintf(stderr, "%zu kernels, %.1f us inside kernels, %.1f us between them, %.1f us from the first entry to the last exit\n", rows.size() - first, busy, idle,
            (rows.back().t1 - rows[first].t0) * 0.01);
  if (!p->h_tile_wg_begin.empty() && p->tile_grid <= STAMP_BLOCKS) {  // the pair kernel's workgroups by tile and by XCD
    const std::vector<int>& wgb = p->h_tile_wg_begin;
    const int nT = (int)wgb.size() - 1;
    std::vector<double> life(p->tile_grid, 0.0);
    for (int b = 0; b < p->tile_grid; ++b) {
      const long long* q = &h[((size_t)ST_PAIRS * STAMP_BLOCKS + b) * STAMP_ROW];
      long long e = 0;
      for (int w = 0; w < STAMP_WAVES; ++w) e = std::max(e, q[1 + w]);
      life[logical_workgroup(b, p->tile_grid)] = q[0] ? (e - q[0]) * 0.01 : 0.0;
    }
    fprintf(stderr, "k_schur_reg3 workgroup lifetimes by tile (us): tile: workgroups min / mean / max\n");
    for (int t = 0; t < nT; ++t) {
      double mn = 1e300, mx = 0.0, sm = 0.0;
      for (int b = wgb[t]; b < wgb[t + 1]; ++b) { mn = std::min(mn, life[b]); mx = std::max(mx, life[b]); sm += life[b]; }
      fprintf(stderr, "  %d: %d  %.1f / %.1f / %.1f\n", t, wgb[t + 1] - wgb[t], mn, sm / std::max(1, wgb[t + 1] - wgb[t]), mx);
    }
    fprintf(stderr, "... by XCD (logical id mod 8): min / mean / max\n");
    for (int x = 0; x < 8; ++x) {
      double mn = 1e300, mx = 0.0, sm = 0.0; int n = 0;
      for (int b = x; b < p->tile_grid; b += 8) { mn = std::min(mn, life[b]); mx = std::max(mx, life[b]); sm += life[b]; ++n; }
      fprintf(stderr, "  XCD %d: %.1f / %.1f / %.1f\n", x, mn, sm / std::max(1, n), mx);
    }
    fprintf(stderr, "... by dispatch order (hardware workgroup id, eighths of the grid): mean\n ");
    for (int o = 0; o < 8; ++o) {
      double sm = 0.0; int n = 0;
      for (int b = o * p->tile_grid / 8; b < (o + 1) * p->tile_grid / 8; ++b) { sm += life[logical_workgroup(b, p->tile_grid)]; ++n; }
      fprintf(stderr, " %.1f", sm / std::max(1, n));
    }
    fprintf(stderr, "\n");
  }
  (void)hipFree(p->stamps);
  p->stamps = nullptr;
}
#endif

void cba_destroy(cba_problem* p) {
  if (!p) return;
  drop_plan_task(p);  // a plan thread still dealing: cancelled and joined before anything it could look at goes away
  (void)hipSetDevice(p->device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  upload_stage_release(p);  // (a set-up call that bailed out)
#ifdef CBA_PROFILING
  if (p->stamps) dump_stamps(p);
#endif
  drain_timers(p);
  for (auto& ev : p->free_events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  if (p->h_vec_sent) (void)hipEventDestroy(p->h_vec_sent);
  if (ncclComm_t c = p->comm.exchange(nullptr)) (void)ncclCommDestroy(c);  // (an aborted communicator was taken out by cba_comm_abort)
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);  // (the stream is drained: nothing of this handle is in flight)
    DevicePool& pool = g_pool[p->device];
    for (size_t i = 0; i < p->allocs.size(); ++i) {
      if (p->alloc_bytes[i] == kPoolChunk && pool.chunks.size() < kPoolKeep) pool.chunks.push_back(p->allocs[i]);
      else (void)hipFree(p->allocs[i]);
    }
    if (p->h_scal) {  // (h_cam and h_flags live in the same allocation)
      if (pool.mail.size() < kPoolKeep) pool.mail.emplace_back(p->h_scal, p->mail_doubles);
      else (void)hipHostFree(p->h_scal);
    }
    if (p->h_vec) {
      if (pool.staging.size() < kPoolKeep && p->h_vec_doubles <= kPoolStagingMax) pool.staging.emplace_back(p->h_vec, p->h_vec_doubles);
      else (void)hipHostFree(p->h_vec);
    }
    if (p->stream) {
      if (pool.streams.size() < kPoolKeep) pool.streams.push_back(p->stream);
      else (void)hipStreamDestroy(p->stream);
    }
  }
  delete p;
}

}  // extern "C"

// The dynamic-LDS ceiling of a kernel is an attribute of the function, shared by every handle of the process: it only ever
// grows, so that a handle created later for a smaller problem (fewer cameras) cannot lower it under a live handle that
// launches the same kernel with more LDS (two CaptureVolumes of different rigs, a worker thread next to the main thread).
static int raise_lds_ceiling(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> ceiling;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = ceiling[{dev, fn}];
  if (bytes <= have) return CBA_OK;
  HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = bytes;
  return CBA_OK;
}

template <typename K>
static int allow_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) return fail(CBA_ERR_UNSUPPORTED, "kernel needs %zu bytes of LDS (> 160 KiB)", bytes);
  return raise_lds_ceiling(reinterpret_cast<const void*>(kernel), bytes);
}

// Workgroups of `kernel` (256 threads, `lds` bytes of dynamic LDS) one CU holds at once, as the runtime computes it from the code object's registers
// and LDS.  Grid-stride kernels are launched with cus x this many workgroups: a grid beyond what is resident runs as a second, partly filled batch
// (device stamps, round 6: k_jv's 1024 workgroups entered in two batches 24.5 us apart — its real allocation is 150 registers, not the 76 the
// kernel trace shows — and took 43.5 us where one batch takes ~30).
template <typename K>
static int resident_per_cu(K kernel, size_t lds, int fallback) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, BLOCK, lds) != hipSuccess || n <= 0) { (void)hipGetLastError(); return fallback; }
  return std::min(n, 8);
}

static size_t lds_tab(const cba_problem* p) { return p->tab_global ? 0 : (size_t)p->C * CAMTAB_LDS; }  // doubles of the LDS copy of the camera table
static size_t lds_cost(const cba_problem* p) { return (lds_tab(p) + 8) * 8; }
// k_build<NC, 0, true> reads the camera table from global memory: chosen when the table is what keeps a second workgroup off the CU
template <int NC> static size_t lds_build_camg(const cba_problem* p) { return ((size_t)p->C * UPack<NC>::STRIDE + 9 * CHUNK + 8) * 8; }
template <int NC> static bool build_camg(const cba_problem* p) {
  if (p->tab_global) return true;
  const size_t with_tab = ((size_t)p->C * CAMTAB_LDS + (size_t)p->C * UPack<NC>::STRIDE + 9 * CHUNK + 8) * 8;
  if (const char* e = std::getenv("CBA_BUILD_CAMG")) return std::atoi(e) != 0 && !p->det_m;
  // (a second workgroup per CU when the table is what keeps it off; no choice at all when table + accumulators exceed the LDS)
  return !p->det_m && ((with_tab > 80 * 1024 && lds_build_camg<NC>(p) <= 80 * 1024) || with_tab > 160 * 1024);
}
template <int NC> static size_t lds_build_cs(const cba_problem* p, bool camg, bool uglob = false) {
  return ((camg ? 0 : (size_t)p->C * CAMTAB_LDS) + (uglob ? 0 : (size_t)p->C * UPack<NC>::STRIDE) + 12 * (size_t)p->cs.pmax + 8) * 8;
}
// beyond ~650 six- / ~320 nine-parameter cameras the packed camera blocks do not fit the LDS: one global copy, FP64 global atomics (k_build_cs<.., UGLOB>)
template <int NC> static bool build_cs_uglob(const cba_problem* p) { return lds_build_cs<NC>(p, true) > 160 * 1024; }
// k_build_cs keeps the camera table in LDS while that leaves room for two workgroups per CU
template <int NC> static bool build_cs_camg(const cba_problem* p) { return p->tab_global || lds_build_cs<NC>(p, false) > 80 * 1024; }
template <int NC> static size_t lds_build(const cba_problem* p) {
  if (build_camg<NC>(p)) return lds_build_camg<NC>(p);
  if (p->det_m)  // parking area of the fixed-order sums instead of the packed blocks, + the chunk's camera order (ints)
    return ((size_t)p->C * CAMTAB_LDS + (size_t)DET_ROUND * DET_LD + 9 * CHUNK + 8) * 8 + ((size_t)CHUNK + p->C + 1) * 4;
  return ((size_t)p->C * CAMTAB_LDS + (size_t)p->C * UPack<NC>::STRIDE + 9 * CHUNK + 8) * 8;
}
static size_t lds_jv(const cba_problem* p, int nv) { return (lds_tab(p) + (size_t)nv * p->lay.ncp_pad + 8) * 8; }
// Register-accumulating Schur kernel per camera width: rows of a camera-pair block per thread = NC / SPLIT, minimum waves
// per SIMD the kernel is compiled for, resident workgroups per CU the plan sizes its grid for (see k_schur_reg3).
template <int NC> struct RegCfg;
template <> struct RegCfg<6> { static constexpr int SPLIT = 1, MINW = 2, PER_CU = 2; };
template <> struct RegCfg<9> { static constexpr int SPLIT = 3, MINW = 3, PER_CU = 1; };
template <int NC> static size_t lds_tprep(const cba_problem* p) {
  if (p->det_m) return ((size_t)SchurRec<NC>::STAGE_WAVES * WAVE * SchurRec<NC>::STAGE * 2 + (size_t)p->C * CAMTAB_LDS + (size_t)DET_ROUND * DET_LD) * 8 + ((size_t)CHUNK + p->C + 1) * 4;
  return ((size_t)SchurRec<NC>::STAGE_WAVES * WAVE * SchurRec<NC>::STAGE * 2 + lds_tab(p) + p->lay.ncp_pad) * 8;
}
constexpr int kSchurRegMaxGroup = 16;  // g*g blocks <= 256 threads
constexpr size_t kSmallSolveLds = ((size_t)(SMALL_N + 1) * SMALL_LD + (size_t)2 * NB * (NB + 1) + (size_t)((SMALL_N + NB - 1) / NB) * NB * (NB + 1) + 2 * SMALL_N) * 8;  // k_small_solve
static size_t lds_backsub(const cba_problem* p) { return (lds_tab(p) + p->lay.ncp_pad + 3 * CHUNK) * 8; }
template <int NC> static size_t lds_backsub_rec(const cba_problem* p) { return BsrCfg<NC>::lds_bytes(p->C); }


// Plan of the pair kernel (schur_plan.h builds it on the host).  The dealing is ~1 us of host work per observation and needs nothing but the
// sorted camera indices, so cba_create starts it on a thread of its own as soon as those exist (PlanTask) and does its uploads, the camera-sorted
// copy and the allocations meanwhile; finish_reg2_tile_plan then binds the workgroups and uploads the plan.
static bool plan_timing_on() {  // CBA_PLAN_TIMING=1: the phases of cba_create and of the plan builder on stderr (tools/create_timing.py)
  static const bool on = std::getenv("CBA_PLAN_TIMING") != nullptr;
  return on;
}

template <int NC, typename KCfg>
static Reg2Params reg2_params(const cba_problem* p) {
  Reg2Params prm;
  const int g = p->gsz;
  prm.C = p->C; prm.P = p->P; prm.G = p->G; prm.g = g;
  constexpr int CT = KCfg::CODE_THREADS;
  prm.rep = (g * g <= CT / 2) ? CT / (g * g) : 1;  // small groups: several threads per block
  prm.n_waves = KCfg::CODE_WAVES;
  prm.pair_cap = KCfg::PAIR_CAP;
  prm.chunk_cap = KCfg::SCHUNK;
  prm.slots_per_wave = KCfg::EPW; prm.wave_pieces = KCfg::WAVE_PIECES; prm.rec_pieces = KCfg::LST;
  prm.zero_piece = KCfg::ZERO_PIECE;
  prm.heavy_obs = p->n_heavy ? HEAVY_OBS : 0;
  if (const char* e = std::getenv("CBA_PLAN_REGION")) prm.region_chunks = std::max(1, std::min(std::atoi(e), 1024));  // (sweeps: chunks dealt together, schur_plan.h)
  return prm;
}

// A Schur plan bound to workgroups and resident on the device, ready to be made the handle's current one (apply_install: a few pointer copies).
// Foreground (cba_create, the cheap or the only plan): device memory from the handle's arena.  Background (the dealt plan of a two-stage handle):
// the PLAN THREAD allocates and uploads while the solve runs on the cheap plan — done on the solver's thread, in the middle of an iteration, the
// allocations and the 46 MB upload of cfg4's plan held a 3 ms solve up for 30-60 ms.
struct PlanInstall {
  TilePlan tp{};
  int tile_grid = 0, n_tile_chunks = 0, reg_reduce_y = 4;
  long tile_stream_len = 0, n_pairs = 0;
  double lane_util = 0.0;
  int* tile_wg_begin = nullptr;
  std::vector<int> h_wgb;
  std::vector<std::pair<void*, size_t>> owned;  // background: its own hipMalloc'ed buffers (handed to the handle's allocation list by apply_install)
  int rc = CBA_OK;
};

template <int NC, typename KCfg, typename Upload>
static int prepare_install(const cba_problem* p, Reg2Plan& plan, const Reg2Params& prm, int max_blocks, Upload&& upload, PlanInstall& out) {
  const int G = p->G, g = p->gsz, C = p->C;
  const int nT = p->n_tiles;
  const bool plan_timing = plan_timing_on();
  auto t_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = t_now();
  auto lap = [&](const char* what) { if (plan_timing) { const double t = t_now(); fprintf(stderr, "  plan: %-34s %.3f s\n", what, t - t_mark); t_mark = t; } };
  constexpr int CT = KCfg::CODE_THREADS;
  out.n_tile_chunks = plan.tile_chunk_begin[nT];
  out.tile_stream_len = (long)plan.obs.size() - 2 * KCfg::SCHUNK;
  out.n_pairs = plan.n_pairs;
  out.lane_util = plan.lane_iters > 0 ? (double)plan.n_pairs / (double)plan.lane_iters : 0.0;
  if (plan_timing)
    fprintf(stderr, "  plan: %d tiles x %d regions, %d chunks (%.1f slots each), %ld pairs, lane utilisation %.3f, LDS cycles per 16-lane read group %.2f (arrival order %.2f)\n",
            nT, plan.n_regions, out.n_tile_chunks, out.n_tile_chunks ? (double)out.tile_stream_len / out.n_tile_chunks : 0.0, plan.n_pairs, out.lane_util,
            plan.lds_groups ? (double)plan.lds_cycles / plan.lds_groups : 0.0, plan.lds_groups ? (double)plan.lds_cycles_arrival / plan.lds_groups : 0.0);
  // cost of a chunk: gather + barrier (in units of one pair iteration; phase clocks on cfg4: ~2200 against ~830 clocks) + the pair iterations of its
  // slowest wave
  const double cost_a = 2.0;
  std::vector<float> chunk_cost;
  const std::vector<double> tile_cost = cba::tile_costs(plan.nit, plan.tile_chunk_begin, nT, KCfg::CODE_WAVES, KCfg::REG_BLOCK / KCfg::SPLIT / WAVE, cost_a, &chunk_cost);
  // CBA_BIND=fine: workgroup counts per tile to ONE workgroup and cost-proportional XCD slices (wg_binding.h).  Measured in round 6 and not the default:
  // it narrows the PLANNED cost per workgroup from 193 / 226 / 247 to 218 / 226 / 233 (min / mean / max), but the pair kernel got no faster (stamps 134
  // against 128 us, HIP-event timer 140 against 132): the slices of different tiles no longer cover the same point ranges on an XCD, and what spreads
  // the lifetimes is not the plan — inside EVERY tile they run from ~100 to ~125 us, and the workgroups dispatched second to a CU take 120 us where the
  // first take 104 (profiles/r06_pair_lifetimes.txt)
  const char* bind_env = std::getenv("CBA_BIND");
  const WgBinding bind = cba::bind_workgroups(plan.tile_chunk_begin, nT, max_blocks, true, &tile_cost, &chunk_cost, bind_env && bind_env[0] == 'f');
  out.tile_grid = bind.grid;
  lap("workgroup binding");
  std::vector<int> gcam(G + 1), gpar(G + 1), ta(nT), tb(nT);
  for (int a = 0; a <= G; ++a) {
    gcam[a] = std::min(a * g, C);
    gpar[a] = (gcam[a] < C) ? p->h_cam_off[gcam[a]] : p->ncp;
  }
  {
    int t = 0;
    for (int a = 0; a < G; ++a)
      for (int b = a; b < G; ++b, ++t) { ta[t] = a; tb[t] = b; }
  }
  int rc;
  int *dob = nullptr, *dcs = nullptr, *dcode = nullptr, *dwf = nullptr, *dwt = nullptr, *dwe = nullptr, *dws = nullptr, *dta = nullptr, *dtb = nullptr,
      *dgc = nullptr, *dgp = nullptr;
  unsigned *dcodes = nullptr, *dnit = nullptr;
#define TRYP(e) do { rc = (e); if (rc) return rc; } while (0)
  TRYP(upload(&dob, plan.obs)); TRYP(upload(&dcs, plan.chunk_start)); TRYP(upload(&dcode, plan.code_start));
  TRYP(upload(&dcodes, plan.codes)); TRYP(upload(&dnit, plan.nit));
  TRYP(upload(&dwf, bind.wfirst)); TRYP(upload(&dwt, bind.wt)); TRYP(upload(&dwe, bind.wend)); TRYP(upload(&dws, bind.wstride));
  TRYP(upload(&dta, ta)); TRYP(upload(&dtb, tb)); TRYP(upload(&dgc, gcam)); TRYP(upload(&dgp, gpar));
  TRYP(upload(&out.tile_wg_begin, bind.wgb));
#undef TRYP
  out.h_wgb = bind.wgb;
  {  // k_reg_reduce splits a tile's partial rows 4 or 16 ways
    int rows = 0;
    for (int t = 0; t < nT; ++t) rows = std::max(rows, (bind.wgb[t + 1] - bind.wgb[t]) * std::max(prm.rep, 1));
    out.reg_reduce_y = rows > 96 ? REG_REDUCE_Y_MAX : 4;
  }
  lap("upload");
  TilePlan tp{};
  tp.chunk_start = dcs; tp.wg_first = dwf; tp.wg_end = dwe; tp.wg_tile = dwt; tp.wg_stride = dws; tp.tile_a = dta; tp.tile_b = dtb;
  tp.group_cam_begin = dgc; tp.group_par_begin = dgp; tp.g = g;
  tp.tile_elems = CT * p->nct * p->nct; tp.obs = dob; tp.rep = prm.rep;  // stride of a workgroup's partial row (>= g^2 blocks)
  tp.codes = dcodes; tp.code_start = dcode; tp.nit = dnit;
  // six-parameter cameras (two workgroups per CU): the halves of the dispatch order take the higher wave priority in alternate trips — cfg4 127 -> 120 us,
  // the means of the two halves 100 / 118 -> 105 / 111 us; the nine-parameter kernel (one 12-wave workgroup per CU) measured 4 % SLOWER with it: off
  tp.prio_shift = (NC == 6) ? 0 : -1;
  if (const char* e = std::getenv("CBA_PAIR_PRIO")) tp.prio_shift = std::atoi(e);  // (-1: off; k >= 0: alternate every 2^k trips)
  out.tp = tp;
  return CBA_OK;
}

// makes a prepared plan the handle's current one (the thread that drives the handle, between two iterations)
static void apply_install(cba_problem* p, PlanInstall& in) {
  p->tp = in.tp;
  p->tile_grid = in.tile_grid; p->n_tile_chunks = in.n_tile_chunks; p->reg_reduce_y = in.reg_reduce_y;
  p->tile_stream_len = in.tile_stream_len; p->n_pairs = in.n_pairs; p->plan_lane_util = in.lane_util;
  p->tile_wg_begin = in.tile_wg_begin;
  p->h_tile_wg_begin = std::move(in.h_wgb);
  for (auto& o : in.owned) { p->allocs.push_back(o.first); p->alloc_bytes.push_back(o.second); p->device_bytes += (long)o.second; }
  in.owned.clear();
}

// background (the plan thread of a two-stage handle): binds the dealt plan and uploads it into buffers of its own
static int prepare_install_background(const cba_problem* p, Reg2Plan& plan, const Reg2Params& prm, PlanInstall& in) {
  if (hipSetDevice(p->device) != hipSuccess) return CBA_ERR_HIP;
  auto upload = [&](auto** out, const auto& h) -> int {
    using T = typename std::remove_pointer<typename std::remove_pointer<decltype(out)>::type>::type;
    const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    void* ptr = nullptr;
    if (hipMalloc(&ptr, bytes) != hipSuccess) return CBA_ERR_HIP;
    in.owned.emplace_back(ptr, bytes);
    if (!h.empty() && hipMemcpy(ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return CBA_ERR_HIP;
    *out = static_cast<T*>(ptr);
    return CBA_OK;
  };
  return (p->nct == 9) ? prepare_install<9, Reg3Cfg<9>>(p, plan, prm, p->plan_max_blocks, upload, in)
                       : prepare_install<6, Reg3Cfg<6>>(p, plan, prm, p->plan_max_blocks, upload, in);
}

// The dealing on its own thread.  One stage (default): cba_create waits for the dealt plan before it returns.  Two stages (CBA_PLAN=swap): the thread
// first makes the CHEAP plan (Reg2Params::cheap: a quarter of the host time), cba_create goes on with that one and returns; the dealt plan follows on
// the same thread and the first damped step that finds it ready swaps it in (maybe_swap_plan).  The task then owns the two host arrays the thread
// reads (cba_create moves them in: the buffers stay where they are).  The destructor cancels and joins.
struct PlanTask {
  Reg2Params prm;
  Reg2Plan cheap, plan;
  PlanInstall install;                 // two stages: the dealt plan, bound and resident (made by the thread itself)
  const cba_problem* handle = nullptr; // two stages: what the thread reads to bind the dealt plan (camera groups, offsets, budget: fixed before it starts)
  int rc_cheap = 0, rc = 0;
  double seconds_cheap = 0.0, seconds = 0.0;
  bool two_stage = false;
  std::atomic<int> stage{0};  // 1: the cheap plan is ready, 2: the dealt plan is ready (or has failed: rc)
  std::atomic<bool> cancel{false};
  std::mutex mu;
  std::condition_variable cv;
  std::thread th;
  HostVec<int> hcam_keep;
  std::vector<int> hps_keep;
  void start(const Reg2Params& params, const int* hcam, const int* hps, bool two, const cba_problem* h) {
    prm = params;
    handle = h;
    prm.cancel = &cancel;
    two_stage = two;
    th = std::thread([this, hcam, hps] {
      auto publish = [this](int s) { { std::lock_guard<std::mutex> lock(mu); stage.store(s, std::memory_order_release); } cv.notify_all(); };
      if (two_stage) {
        const auto t0 = std::chrono::steady_clock::now();
        Reg2Params c = prm;
        c.cheap = true;
        rc_cheap = build_reg2_plan(c, hcam, hps, cheap);
        seconds_cheap = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        publish(1);
      }
      const auto t0 = std::chrono::steady_clock::now();
      // second stage = background work beside a running solve: half of the CPUs the process may use (all of them would leave the thread that drives
      // the GPU waiting for a core — or, under a cgroup quota, frozen with the rest of the process: usable_cpus)
      if (two_stage && prm.threads == 0) prm.threads = (int)std::max(1u, std::min(32u, usable_cpus() / 2));
      rc = (two_stage && rc_cheap) ? rc_cheap : build_reg2_plan(prm, hcam, hps, plan);
      if (two_stage && !rc && !cancel.load()) {  // allocate and upload HERE: the thread that drives the handle only swaps pointers (apply_install)
        rc = prepare_install_background(handle, plan, prm, install);
        if (rc) for (auto& o : install.owned) (void)hipFree(o.first);
        if (rc) install.owned.clear();
      }
      seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      publish(2);
    });
  }
  void wait_stage(int s) {
    std::unique_lock<std::mutex> lock(mu);
    cv.wait(lock, [&] { return stage.load(std::memory_order_acquire) >= s; });
  }
  ~PlanTask() {
    cancel.store(true);
    if (th.joinable()) th.join();
    for (auto& o : install.owned) (void)hipFree(o.first);  // (a prepared plan nobody took over)
  }
};

static void drop_plan_task(cba_problem* p) {
  delete p->plan_task;
  p->plan_task = nullptr;
}

// foreground: binds `plan`, uploads it into the handle's arena and makes it current
static int install_reg2_plan(cba_problem* p, Reg2Plan& plan, const Reg2Params& prm) {
  PlanInstall in;
  auto upload = [&](auto** out, const auto& h) { return dev_upload(p, out, h); };
  const int rc = (p->nct == 9) ? prepare_install<9, Reg3Cfg<9>>(p, plan, prm, p->plan_max_blocks, upload, in)
                               : prepare_install<6, Reg3Cfg<6>>(p, plan, prm, p->plan_max_blocks, upload, in);
  if (rc) return rc;
  apply_install(p, in);
  return CBA_OK;
}

// Camera groups of the pair kernel: host-only (camera count), decided before anything touches the device so that the plan can be dealt while
// cba_create uploads.
static void choose_schur_groups(cba_problem* p) {
  const int gmax = std::min(p->C, kSchurRegMaxGroup);
  p->G = (p->C + gmax - 1) / gmax;
  p->gsz = (p->C + p->G - 1) / p->G;
  p->n_tiles = p->G * (p->G + 1) / 2;
}

template <int NC>
static int configure_kernels(cba_problem* p) {
  int rc;
  // The camera table in LDS (296 B per camera) next to the largest other LDS user: beyond the budget every per-observation kernel switches to
  // its CAMG variant (table through the vector cache).  What then bounds the camera count is the per-camera accumulators of the linearisation
  // (nc (nc + 3) / 2 doubles per camera in LDS: ~650 six-parameter or ~320 nine-parameter cameras).  CBA_CAMTAB_GLOBAL=0/1 forces the choice (tests).
  {
    p->tab_global = false;
    const size_t worst = std::max(std::max(lds_jv(p, 2), lds_backsub(p)), lds_tprep<NC>(p));  // (the linearisation has its own switch: build_camg)
    p->tab_global = !p->det_m && worst > 150 * 1024;
    if (const char* e = std::getenv("CBA_CAMTAB_GLOBAL")) p->tab_global = std::atoi(e) != 0 && !p->det_m;
  }
  if ((rc = allow_lds(k_cost<false>, lds_cost(p)))) return rc;
  if ((rc = allow_lds(k_cost<true>, lds_cost(p)))) return rc;
  if ((rc = allow_lds(k_cost<false, true>, lds_cost(p)))) return rc;
  if ((rc = allow_lds(k_cost<true, true>, lds_cost(p)))) return rc;
  const bool use_cs = p->cs.n_sc && !p->det_m && !p->n_heavy;  // (run_build_into's condition)
  if (use_cs) {
    if (build_cs_uglob<NC>(p)) { rc = allow_lds(k_build_cs<NC, true, true>, lds_build_cs<NC>(p, true, true)); if (!rc) rc = allow_lds(k_build_cs<NC, true, true, true>, lds_build_cs<NC>(p, true, true)); }
    else if (build_cs_camg<NC>(p)) { rc = allow_lds(k_build_cs<NC, true>, lds_build_cs<NC>(p, true)); if (!rc) rc = allow_lds(k_build_cs<NC, true, false, true>, lds_build_cs<NC>(p, true)); }
    else { rc = allow_lds(k_build_cs<NC, false>, lds_build_cs<NC>(p, false)); if (!rc) rc = allow_lds(k_build_cs<NC, false, false, true>, lds_build_cs<NC>(p, false)); }
    if (rc) return rc;
  }
  if (!use_cs || !build_cs_uglob<NC>(p)) {  // the point-ordered kernel keeps the packed blocks in LDS: not with that many cameras
    if ((rc = allow_lds(k_build<NC>, lds_build<NC>(p)))) return rc;
    if ((rc = allow_lds(k_build<NC, 0, true>, lds_build<NC>(p)))) return rc;
  } else if (p->eval_only) {
    // (nothing: an evaluation-only handle runs k_cost only)
  }
  if (p->det_m) {
    if ((rc = allow_lds(k_build<NC, 3>, lds_build<NC>(p)))) return rc;
    if ((rc = allow_lds(k_build<NC, 5>, lds_build<NC>(p)))) return rc;
    if ((rc = allow_lds(k_build<NC, 8>, lds_build<NC>(p)))) return rc;
    if ((rc = allow_lds(k_tprep<NC, 3>, lds_tprep<NC>(p)))) return rc;
    if ((rc = allow_lds(k_tprep<NC, 5>, lds_tprep<NC>(p)))) return rc;
    if ((rc = allow_lds(k_tprep<NC, 8>, lds_tprep<NC>(p)))) return rc;
    if constexpr (NC == 6) {
      if ((rc = allow_lds(k_build<NC, 16>, lds_build<NC>(p)))) return rc;
      if ((rc = allow_lds(k_tprep<NC, 16>, lds_tprep<NC>(p)))) return rc;
    }
  }
  if ((rc = allow_lds(k_jv<NC, 1>, lds_jv(p, 1)))) return rc;
  if ((rc = allow_lds(k_jv<NC, 2>, lds_jv(p, 2)))) return rc;
  if ((rc = allow_lds(k_jv<NC, 1, true>, lds_jv(p, 1)))) return rc;
  if ((rc = allow_lds(k_jv<NC, 2, true>, lds_jv(p, 2)))) return rc;
  if ((rc = allow_lds(k_schur_reg3<NC, RegCfg<NC>::SPLIT, RegCfg<NC>::MINW>, Reg3Cfg<NC>::LDS_BYTES))) return rc;
  if ((rc = allow_lds(k_tprep<NC>, lds_tprep<NC>(p)))) return rc;
  if ((rc = allow_lds(k_tprep<NC, 0, true>, lds_tprep<NC>(p)))) return rc;
  if ((rc = allow_lds(k_tprep<NC, 0, false, true>, lds_tprep<NC>(p)))) return rc;
  if ((rc = allow_lds(k_tprep<NC, 0, true, true>, lds_tprep<NC>(p)))) return rc;
  if ((rc = allow_lds(k_backsub<NC>, lds_backsub(p)))) return rc;
  if ((rc = allow_lds(k_backsub<NC, true>, lds_backsub(p)))) return rc;
  if ((rc = allow_lds(k_backsub<NC, false, true>, lds_backsub(p)))) return rc;
  if ((rc = allow_lds(k_backsub<NC, true, true>, lds_backsub(p)))) return rc;
  // back-substitution from the T records: whenever no chunk is a fragment of a very large point (those add their sums by atomics in k_backsub) and the
  // per-camera step entries fit the LDS next to the record buffer (~2400 six- / ~1000 nine-parameter cameras); CBA_BACKSUB_REC=0: the old kernel (A/B)
  {
    const char* e = std::getenv("CBA_BACKSUB_REC");
    // nine-parameter cameras keep k_backsub: their records are 176 bytes, and streaming them costs more than linearising again (cfg5, round 6: 394 us
    // against 326; six-parameter cameras, cfg4: 48 against 64)
    p->backsub_rec = (NC == 6 || (e && e[0] == '1')) && !p->eval_only && !p->has_fragments && lds_backsub_rec<NC>(p) <= 150 * 1024 && !(e && e[0] == '0');
    if (p->backsub_rec) {
      if ((rc = allow_lds(k_backsub_rec<NC, false>, lds_backsub_rec<NC>(p)))) return rc;
      if ((rc = allow_lds(k_backsub_rec<NC, true>, lds_backsub_rec<NC>(p)))) return rc;
      if (!std::getenv("CBA_BACKSUB_WGS"))  // as many workgroups as are resident at once (cfg4: four per CU, 40 000 bytes of LDS each: 48 us against 53 at two)
        p->grid_backsub = std::max(1, std::min(p->n_chunks, p->cus * resident_per_cu(k_backsub_rec<NC, true>, lds_backsub_rec<NC>(p), 2)));
    }
  }
  {
    int per_cu = p->tab_global ? resident_per_cu(k_jv<NC, 1, true>, lds_jv(p, 1), 2) : resident_per_cu(k_jv<NC, 1>, lds_jv(p, 1), 2);
    if (const char* e = std::getenv("CBA_JV_WGS")) per_cu = std::max(1, std::min(std::atoi(e), 8));
    p->jv_grid = (int)std::max<long>(1, std::min<long>((p->N + BLOCK - 1) / BLOCK, (long)p->cus * per_cu));
    p->jv_grid = std::min(p->jv_grid, 2048);  // (rows of partial4)
  }
  if (p->n_heavy && (rc = allow_lds(k_heavy_schur<NC>, (size_t)p->ncp * 3 * sizeof(double) + (size_t)p->ncp * sizeof(int)))) return rc;
  if ((rc = allow_lds(k_chol_apply, (size_t)p->ncp * 8))) return rc;
  if ((rc = allow_lds(k_step_cam, (size_t)p->lay.ncp_pad * 8))) return rc;
  if (p->ncp <= SMALL_N && (rc = allow_lds(k_small_solve<NC>, kSmallSolveLds))) return rc;
  return CBA_OK;
}

extern "C" {

int cba_create(const cba_problem_desc* d, const cba_options* opt, cba_problem** out) {
  if (!d || !out) return fail(CBA_ERR_INVALID, "cba_create: null argument");
  *out = nullptr;
  if (d->n_cams <= 0 || d->n_points <= 0 || d->n_obs <= 0) return fail(CBA_ERR_INVALID, "cba_create: empty problem (cams=%d points=%d obs=%lld)", d->n_cams, d->n_points, (long long)d->n_obs);
  if (d->n_obs >= (1LL << 31)) return fail(CBA_ERR_UNSUPPORTED, "cba_create: more than 2^31 observations");
  if (!d->cam_n_params || !d->cam_model || !d->cam_const || !d->obs_cam || !d->obs_pt || !d->obs_uv) return fail(CBA_ERR_INVALID, "cba_create: null array");
  if (d->loss < CBA_LOSS_LINEAR || d->loss > CBA_LOSS_ARCTAN) return fail(CBA_ERR_INVALID, "cba_create: unknown loss %d", d->loss);
  if (d->loss != CBA_LOSS_LINEAR && !(d->f_scale > 0.0)) return fail(CBA_ERR_INVALID, "cba_create: f_scale must be positive");
  const double t_enter = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(CBA_ERR_NO_DEVICE, "no HIP device available: the MI355X engine has no CPU fallback");
  int dev = (opt && opt->device_id >= 0) ? opt->device_id : 0;
  if (!(opt && opt->device_id >= 0)) (void)hipGetDevice(&dev);
  if (dev >= ndev) return fail(CBA_ERR_INVALID, "device %d requested, %d available", dev, ndev);
  HIPCHK(hipSetDevice(dev));
  int n_cus = 0;  // (hipGetDeviceProperties is a millisecond per call: a visible share of creating a handle for a small session)
  if (hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cus = 0;

  cba_problem* p = new cba_problem();
  p->device = dev;
  p->C = d->n_cams; p->P = d->n_points; p->N = d->n_obs;
  p->loss = d->loss; p->f_scale = d->f_scale;
  if (const char* e = std::getenv("CBA_REG_FINALIZE")) p->fuse_reg_finalize = e[0] != '0';
#ifdef CBA_PROFILING
  p->schur_clock = std::getenv("CBA_SCHUR_CLOCK") != nullptr;
  p->want_chol_trace = std::getenv("CBA_CHOL_TRACE") != nullptr;
  if (std::getenv("CBA_STAMPS")) {
    const size_t n = (size_t)STAMP_SLOTS * STAMP_BLOCKS * STAMP_ROW;
    if (hipMalloc((void**)&p->stamps, n * sizeof(long long)) == hipSuccess) {
      (void)hipMemset(p->stamps, 0, n * sizeof(long long));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cba_stamps), &p->stamps, sizeof(p->stamps));
    }
  }
#endif
  int rc = CBA_OK;
  auto bail = [&](int code) { cba_destroy(p); return code; };
  // a HIP failure after the handle owns resources goes through bail(): the handle, its arena chunks and the pooled stream are released
#define HIPBAIL(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(fail(CBA_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__)); } while (0)

  // camera tables
  std::vector<int> np(p->C), model(p->C), off(p->C);
  int ncp = 0, nct = 6;
  for (int c = 0; c < p->C; ++c) {
    np[c] = d->cam_n_params[c]; model[c] = d->cam_model[c];
    if (np[c] != 6 && np[c] != 9) return bail(fail(CBA_ERR_INVALID, "camera %d: n_params must be 6 or 9, got %d", c, np[c]));
    if (model[c] != CBA_MODEL_PINHOLE_BC5 && model[c] != CBA_MODEL_FISHEYE4) return bail(fail(CBA_ERR_INVALID, "camera %d: unknown model %d", c, model[c]));
    if (model[c] == CBA_MODEL_FISHEYE4 && np[c] != 6) return bail(fail(CBA_ERR_INVALID, "camera %d: fisheye cameras are always locked (6 params)", c));
    if (!(d->cam_const[c * 12] > 0.0)) return bail(fail(CBA_ERR_INVALID, "camera %d: fx_initial must be positive", c));
    off[c] = ncp; ncp += np[c];
    if (np[c] == 9) nct = 9;
  }
  p->ncp = ncp; p->nct = nct;
  p->h_cam_off = off; p->h_cam_np = np;
  p->lay.ncp = ncp; p->lay.ncp_pad = (ncp + 31) / 32 * 32;
  p->lay.P = p->P; p->lay.Ppad = (p->P + 31) / 32 * 32;
  std::vector<int> pcam(p->lay.ncp_pad, 0), ploc(p->lay.ncp_pad, 0);
  for (int c = 0; c < p->C; ++c)
    for (int r = 0; r < np[c]; ++r) { pcam[off[c] + r] = c; ploc[off[c] + r] = r; }

  const bool plan_timing = plan_timing_on();
  auto t_now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = t_now();
  auto lap = [&](const char* what) { if (plan_timing) { const double t = t_now(); fprintf(stderr, "cba_create: %-28s %.6f s\n", what, t - t_mark); t_mark = t; } };
  // plan: sort by point, chunk table
  HostVec<int64_t> order(p->N), pstart((size_t)p->P + 1), cstart((size_t)p->N + 2);  // (written by cba_host_plan: order and pstart in full, cstart up to the chunk count)
  // (cba_host_plan checks the point and camera indices of every observation)
  int64_t nch = cba_host_plan(p->P, p->N, d->obs_pt, d->obs_cam, p->C, CHUNK, order.data(), pstart.data(), cstart.data());
  if (nch < 0) return bail((int)nch);
  p->n_chunks = (int)nch;
  lap("sort by point, chunk table");
  HostVec<int> hcam(p->N), hpt(p->N), hord(p->N);
  std::vector<int> hps((size_t)p->P + 1), hcs((size_t)nch + 1);
  {  // gather into the sorted order, by a few host threads (1M observations: 6 ms on one)
    auto gather = [&](int64_t lo, int64_t hi) {
      for (int64_t i = lo; i < hi; ++i) {
        const int64_t o = order[i];
        hcam[i] = d->obs_cam[o]; hpt[i] = d->obs_pt[o]; hord[i] = (int)o;  // (the coordinates are gathered on the device: k_gather_uv)
      }
    };
    const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, usable_cpus()), p->N / 65536));
    std::vector<std::thread> pool;
    for (int t = 1; t < nth; ++t) pool.emplace_back(gather, p->N * t / nth, p->N * (t + 1) / nth);
    gather(0, p->N / nth);
    for (auto& th : pool) th.join();
  }
  int maxk = 0;
  for (int q = 0; q <= p->P; ++q) { hps[q] = (int)pstart[q]; if (q) maxk = std::max<int>(maxk, (int)(pstart[q] - pstart[q - 1])); }
  for (int64_t q = 0; q <= nch; ++q) hcs[q] = (int)cstart[q];
  p->max_obs_per_point = maxk;
  // heavy points (static markers observed again in every frame): per-camera Schur sums instead of observation pairs
  std::vector<int> heavy, heavy_frag;
  for (int q = 0; q < p->P && maxk > HEAVY_OBS; ++q)
    if (hps[q + 1] - hps[q] > HEAVY_OBS) { heavy.push_back(q); heavy_frag.push_back(hps[q + 1] - hps[q] > CHUNK ? 1 : 0); }
  if ((long)heavy.size() > std::max<long>(64, p->P / 64)) {
    // not a few static points but a dense problem (every point seen by > HEAVY_OBS cameras): the per-point workgroup of
    // k_heavy_schur is the wrong tool; keep the pair plan (a point with more observations in one tile than a chunk holds is refused below)
    if (maxk > CHUNK) return bail(fail(CBA_ERR_UNSUPPORTED, "%zu world points have more than %d observations (one has %d); at most %ld such points are supported",
                                       heavy.size(), HEAVY_OBS, maxk, std::max<long>(64, p->P / 64)));
    heavy.clear(); heavy_frag.clear();
  }
  p->n_heavy = (int)heavy.size();
  p->h_heavy_pts = heavy;
  // the Schur plan of the register kernels is dealt on its own thread from here on (declared after the vectors it reads: joined before they go)
  p->eval_only = opt && opt->evaluation_only != 0;
  choose_schur_groups(p);
  // Two-stage plan: start with the CHEAP plan (a third of the host time, pair kernel 1.5-1.75x slower) and swap the dealt one in when its thread is
  // done — the handle is ready 10 ms (cfg4) to half a second (cfg5) earlier and a solve of a handful of iterations may be over before the dealt plan
  // would have been.  Default from kTwoStageObs observations on (below, the dealt plan is ready before the uploads are); CBA_PLAN=full / swap force
  // one way, cba_plan_wait makes a handle final (benchmarks).  Not with fixed-order sums (the iteration the swap lands on would vary from run to run)
  // nor with the profiling build (it wants the plan it profiles).
  constexpr long kTwoStageObs = 500000;
  const char* plan_env = std::getenv("CBA_PLAN");
  bool plan_two_stage = p->N >= kTwoStageObs;
  if (plan_env && std::strcmp(plan_env, "swap") == 0) plan_two_stage = true;
  if (plan_env && (std::strcmp(plan_env, "full") == 0 || std::strcmp(plan_env, "cheap") == 0)) plan_two_stage = false;
  if ((opt && opt->deterministic) || p->schur_clock) plan_two_stage = false;
  if (!p->eval_only) {
    Reg2Params prm = (nct == 9) ? reg2_params<9, Reg3Cfg<9>>(p) : reg2_params<6, Reg3Cfg<6>>(p);
    if (plan_env) prm.cheap = std::strcmp(plan_env, "cheap") == 0;  // (measurements: the cheap plan for good)
    p->plan_task = new PlanTask();  // owned by the handle: cba_destroy (also through bail) cancels and joins it while the arrays it reads are alive
    {  // the workgroup budget of the pair kernel, fixed before the plan thread may want it (two stages: it binds the dealt plan itself)
      const int cus0 = n_cus > 0 ? n_cus : 256;
      const size_t tile_lds = (nct == 9) ? Reg3Cfg<9>::LDS_BYTES : Reg3Cfg<6>::LDS_BYTES;
      const int per_cu = std::min(std::max<int>(1, (int)((160 * 1024) / tile_lds)), (nct == 9) ? RegCfg<9>::PER_CU : RegCfg<6>::PER_CU);  // LDS, register budget
      const int mb0 = (opt && opt->max_blocks > 0) ? opt->max_blocks : 2 * cus0;
      p->plan_max_blocks = std::min(cus0 * per_cu, std::max(mb0, cus0));  // no partial last round
    }
    p->plan_task->start(prm, hcam.data(), hps.data(), plan_two_stage, p);
  }

  {
    // one mapped host allocation for the three mailboxes: scalars (64 doubles), camera blocks (3 ncp + 8 doubles), flags (4 ints)
    const size_t n_mail = 64 + ((size_t)3 * ncp + 8) + 2 + (size_t)4 * ncp;  // scalars | three camera blocks + sequence | flags | four camera blocks
    {
      std::lock_guard<std::mutex> lock(g_pool_mu);
      DevicePool& pool = g_pool[dev];
      if (!pool.streams.empty()) { p->stream = pool.streams.back(); pool.streams.pop_back(); }
      for (size_t i = 0; i < pool.mail.size(); ++i)
        if (pool.mail[i].second >= n_mail) {
          p->h_scal = pool.mail[i].first; p->mail_doubles = pool.mail[i].second;
          pool.mail.erase(pool.mail.begin() + (long)i);
          break;
        }
    }
    if (!p->stream) HIPBAIL(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    if (!p->h_scal) {
      p->mail_doubles = std::max<size_t>(n_mail, 64 + 7 * 96 + 10);  // (room for 96 camera parameters: small rigs share mailboxes)
      HIPBAIL(hipHostMalloc((void**)&p->h_scal, p->mail_doubles * sizeof(double), hipHostMallocMapped));
    }
    HIPBAIL(hipHostGetDevicePointer((void**)&p->d_hscal, p->h_scal, 0));
    p->h_cam = p->h_scal + 64; p->d_hcam = p->d_hscal + 64;
    p->h_flags = reinterpret_cast<int*>(p->h_cam + ((size_t)3 * ncp + 8)); p->d_hflags = reinterpret_cast<int*>(p->d_hcam + ((size_t)3 * ncp + 8));
    p->h_bcam = p->h_cam + ((size_t)3 * ncp + 8) + 2; p->d_hbcam = p->d_hcam + ((size_t)3 * ncp + 8) + 2;
  }
  std::memset(p->h_scal, 0, (64 + ((size_t)3 * ncp + 8) + 2 + (size_t)4 * ncp) * sizeof(double));

  const int cus = n_cus > 0 ? n_cus : 256;
  p->cus = cus;
  const int max_blocks = (opt && opt->max_blocks > 0) ? opt->max_blocks : 2 * cus;  // two persistent workgroups per CU for the per-observation kernels
  p->grid = std::max(1, std::min(p->n_chunks, max_blocks));
  // (2 per CU also for k_backsub, whose 28 KB of LDS would admit five: swept in round 5, profiles/r05_occupancy_sweeps.txt — 64 us at 2, 69-83 at 3-5)
  p->grid_backsub = std::max(1, std::min(p->n_chunks, 2 * cus));
  if (const char* e = std::getenv("CBA_BACKSUB_WGS")) p->grid_backsub = std::max(1, std::min(p->n_chunks, std::min(std::max(std::atoi(e), 1), 8) * cus));  // occupancy sweeps

#define TRY(e) do { rc = (e); if (rc) return bail(rc); } while (0)
  lap("reorder on host");
  upload_stage_acquire(p);
  TRY(dev_upload(p, &p->obs_cam, hcam)); TRY(dev_upload(p, &p->obs_pt, hpt));
  TRY(dev_upload(p, &p->order, hord)); TRY(dev_upload(p, &p->pt_start, hps)); TRY(dev_upload(p, &p->chunk_start, hcs));
  {  // the caller's (u, v) pairs as they are, sorted on the device (the raw copy is scratch: v2's memory is not big enough, it stays in the arena)
    double* uv_raw = nullptr;
    TRY(dev_alloc(p, &uv_raw, (size_t)2 * p->N)); TRY(dev_alloc(p, &p->obs_u, (size_t)p->N)); TRY(dev_alloc(p, &p->obs_v, (size_t)p->N));
    {
      const size_t bytes = (size_t)2 * p->N * sizeof(double);
      if (p->up_stage && bytes <= kUpStageMax && p->up_used + bytes <= p->up_cap) {
        std::memcpy(p->up_stage + p->up_used, d->obs_uv, bytes);
        HIPBAIL(hipMemcpyAsync(uv_raw, p->up_stage + p->up_used, bytes, hipMemcpyHostToDevice, p->stream));
        p->up_used += (bytes + 63) & ~(size_t)63;
      } else {
        HIPBAIL(hipMemcpy(uv_raw, d->obs_uv, bytes, hipMemcpyHostToDevice));
      }
    }
    hipLaunchKernelGGL(k_gather_uv, dim3((int)std::min<long>((p->N + 255) / 256, 2048)), dim3(256), 0, p->stream, (const double*)uv_raw, (const int*)p->order, p->N, p->obs_u, p->obs_v);
  }
  if (opt && opt->deterministic) {
    // fixed-order per-camera sums (k_build / k_tprep, det_round): per chunk the observation order by camera and the camera offsets
    const int need = (p->C * DET_ROUND + BLOCK - 1) / BLOCK;
    // tasks per thread of the fixed-order sums: 3, 5, 8 (<= 227 cameras) and, six-parameter cameras only, 16 (<= 455 by the task count; the LDS copy of
    // the camera table next to k_tprep's staging and parking areas admits 385: configure_kernels reports the bytes beyond that).  Nine-parameter cameras stop at 227: six
    // rounds of 16 running sums are 96 doubles per thread.
    p->det_m = need <= 3 ? 3 : need <= 5 ? 5 : need <= 8 ? 8 : (nct == 6 && need <= 16) ? 16 : -1;
    if (p->det_m < 0) return bail(fail(CBA_ERR_UNSUPPORTED, "deterministic sums support up to %d %s-parameter cameras, the problem has %d",
                                       (nct == 6 ? 16 : 8) * BLOCK / DET_ROUND, nct == 6 ? "six" : "nine", p->C));
    std::vector<unsigned char> perm((size_t)std::max<int64_t>(nch, 1) * CHUNK, 0);
    std::vector<unsigned short> cst((size_t)std::max<int64_t>(nch, 1) * (p->C + 1), 0);
    for (int64_t c = 0; c < nch; ++c) {
      const int o0 = hcs[c], n = hcs[c + 1] - o0;
      unsigned short* cs = &cst[(size_t)c * (p->C + 1)];
      for (int k = 0; k < n; ++k) cs[hcam[o0 + k] + 1]++;
      for (int q = 0; q < p->C; ++q) cs[q + 1] += cs[q];
      std::vector<unsigned short> cur(cs, cs + p->C);
      for (int k = 0; k < n; ++k) perm[(size_t)c * CHUNK + cur[hcam[o0 + k]]++] = (unsigned char)k;  // stable: observation order inside a camera
    }
    unsigned char* dperm = nullptr; unsigned short* dcst = nullptr;
    TRY(dev_upload(p, &dperm, perm)); TRY(dev_upload(p, &dcst, cst));
    p->det = DetPlan{dperm, dcst};
  }
  {
    std::vector<int> hcp((size_t)std::max<int64_t>(nch, 1) * 2, 0);
    for (int64_t q = 0; q < nch; ++q) {
      hcp[2 * q] = hpt[hcs[q]];
      hcp[2 * q + 1] = hpt[hcs[q + 1] - 1] - hpt[hcs[q]] + 1;
      if (hps[hpt[hcs[q]] + 1] - hps[hpt[hcs[q]]] > CHUNK) { hcp[2 * q + 1] = -1; p->has_fragments = true; }  // fragment of a point larger than a chunk
    }
    TRY(dev_upload(p, &p->chunk_pts, hcp));
    // camera-sorted super-chunks for k_build_cs: consecutive chunks while observations <= CS_MAX_OBS and points <= CS_MAX_PTS, their observations
    // a second time in (super-chunk, camera, point) order.  Not for the fixed-order sums (their own per-chunk order) nor with fragments of
    // points larger than a chunk (those add to V / g by global atomics in k_build).
    const char* cs_env = std::getenv("CBA_BUILD_CS");
    bool cs_ok = !(opt && opt->deterministic) && !(cs_env && cs_env[0] == '0') && nch > 0;
    for (int64_t q = 0; q < nch && cs_ok; ++q)
      if (hcp[2 * q + 1] < 0 || hcp[2 * q + 1] > CS_MAX_PTS) cs_ok = false;
    if (cs_ok) {
      // Size: every workgroup of the launch (p->grid persistent ones) should get the same number of super-chunks — with 1.3 per workgroup a
      // quarter of the pass is a tail.  Cap ~2000 observations for six-parameter cameras (two workgroups per CU), ~4000 for nine-parameter
      // ones (54 values per camera change: longer runs pay; measured on cfg4 / cfg5: 59 / 72 / 120 us at 2048 / 3072 / 4096, 545 / 530 / 508 us).
      const int64_t s_cap = (nct == 9) ? 4096 : 2048;
      const int64_t wgs = std::max(1, p->grid);  // the persistent workgroups of the launch
      int64_t rounds = std::max<int64_t>(1, (p->N + wgs * s_cap - 1) / (wgs * s_cap));
      std::vector<int> sc_chunk, sc_obs, sc_p0, sc_np;
      int pmax = 0;
      // Round 6: the super-chunks are cut at the chunk boundaries nearest to k N / (rounds * workgroups), so that there are EXACTLY rounds * workgroups of
      // them (fewer on a small problem) and workgroup w, which takes super-chunks w, w + grid, ..., gets `rounds` of about the same size.  Until round 5 they
      // were filled greedily up to N / (rounds * workgroups) observations: whole chunks leave each a little short of that, cfg4 ended with 1143 super-chunks
      // of 1750 observations for 1024 slots, and 119 of the 512 workgroups walked three of them while the others walked two (device stamps, round 6:
      // workgroup lifetimes 42 / 53 / 63 us min / mean / max; cfg5 236 / 273 / 337).  A cut that would exceed the caps (observations, points of the LDS
      // stage) asks for one more round.
      for (int attempt = 0; attempt < 8; ++attempt, ++rounds) {
        const int64_t n_target = std::min<int64_t>(std::max<int64_t>(1, wgs * rounds), nch);
        sc_chunk.assign(1, 0); sc_obs.assign(1, 0); sc_p0.clear(); sc_np.clear();
        pmax = 0;
        bool fits = true;
        int64_t q = 0;
        for (int64_t k = 0; k < n_target && q < nch; ++k) {
          const int64_t goal = (p->N * (k + 1) + n_target - 1) / n_target;  // observations behind super-chunk k
          int64_t e = q + 1;
          while (e < nch && (k + 1 == n_target || hcs[e + 1] <= goal || (hcs[e] < goal && goal - hcs[e] > hcs[e + 1] - goal))) ++e;  // nearest boundary
          if (k + 1 == n_target) e = nch;
          const int np_here = hcp[2 * (e - 1)] + hcp[2 * (e - 1) + 1] - hcp[2 * q];
          if (hcs[e] - hcs[q] > s_cap + CHUNK || np_here > CS_MAX_PTS) { fits = false; break; }
          sc_chunk.push_back((int)e); sc_obs.push_back(hcs[e]);
          sc_p0.push_back(hcp[2 * q]); sc_np.push_back(np_here);
          pmax = std::max(pmax, np_here);
          q = e;
        }
        if (fits && q == nch) break;
        if (attempt == 7) {  // (irregular point sizes: the greedy fill of rounds 3-5, which always fits)
          const int64_t s_target = std::max<int64_t>(CHUNK, (p->N + wgs * rounds - 1) / (wgs * rounds));
          sc_chunk.assign(1, 0); sc_obs.assign(1, 0); sc_p0.clear(); sc_np.clear();
          pmax = 0;
          for (int64_t qq = 0; qq < nch;) {
            int64_t e = qq + 1;
            while (e < nch && hcs[e + 1] - hcs[qq] <= s_target && hcp[2 * e] + hcp[2 * e + 1] - hcp[2 * qq] <= CS_MAX_PTS) ++e;
            sc_chunk.push_back((int)e); sc_obs.push_back(hcs[e]);
            sc_p0.push_back(hcp[2 * qq]); sc_np.push_back(hcp[2 * (e - 1)] + hcp[2 * (e - 1) + 1] - hcp[2 * qq]);
            pmax = std::max(pmax, sc_np.back());
            qq = e;
          }
        }
      }
      const int n_sc = (int)sc_p0.size();
      HostVec<int> cperm(p->N);  // position in (super-chunk, camera, point) order -> sorted observation; the copy itself is made on the device (k_cs_fill)
      auto fill = [&](int s0, int s1) {
        std::vector<int> start((size_t)p->C + 1);
        for (int sidx = s0; sidx < s1; ++sidx) {
          const int o0 = sc_obs[sidx], o1 = sc_obs[sidx + 1];
          std::fill(start.begin(), start.end(), 0);
          for (int i = o0; i < o1; ++i) start[(size_t)hcam[i] + 1]++;
          for (int c = 0; c < p->C; ++c) start[(size_t)c + 1] += start[c];
          for (int i = o0; i < o1; ++i) {  // stable: point order inside a camera
            cperm[o0 + start[hcam[i]]++] = i;
          }
        }
      };
      {
        const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, usable_cpus()), n_sc / 64));
        std::vector<std::thread> pool;
        for (int t = 1; t < nth; ++t) pool.emplace_back(fill, (int)((int64_t)n_sc * t / nth), (int)((int64_t)n_sc * (t + 1) / nth));
        fill(0, (int)((int64_t)n_sc / nth));
        for (auto& th : pool) th.join();
      }
      double *dcu = nullptr, *dcv = nullptr;
      int *dcc = nullptr, *dcp = nullptr, *dso = nullptr, *dp0 = nullptr, *dnp = nullptr;
      int* dperm = nullptr;
      TRY(dev_upload(p, &dperm, cperm));
      TRY(dev_alloc(p, &dcu, (size_t)p->N)); TRY(dev_alloc(p, &dcv, (size_t)p->N)); TRY(dev_alloc(p, &dcc, (size_t)p->N)); TRY(dev_alloc(p, &dcp, (size_t)p->N));
      TRY(dev_upload(p, &dso, sc_obs)); TRY(dev_upload(p, &dp0, sc_p0)); TRY(dev_upload(p, &dnp, sc_np));
      hipLaunchKernelGGL(k_cs_fill, dim3(n_sc), dim3(256), 0, p->stream, (const int*)dperm, (const int*)dso, (const int*)dp0, (const double*)p->obs_u, (const double*)p->obs_v,
                         (const int*)p->obs_cam, (const int*)p->obs_pt, dcu, dcv, dcc, dcp);
      p->cs = CsPlan{dcu, dcv, dcc, dcp, dso, dp0, dnp, n_sc, (pmax + 31) / 32 * 32};
    }
  }
  if (p->n_heavy) {
    TRY(dev_upload(p, &p->heavy_pts, heavy)); TRY(dev_upload(p, &p->heavy_frag, heavy_frag));
    TRY(dev_alloc(p, &p->heavy_W, (size_t)p->n_heavy * ncp * 3));
  }
  std::vector<double> cc(d->cam_const, d->cam_const + (size_t)p->C * 12);
  TRY(dev_upload(p, &p->cam_const, cc));
  TRY(dev_upload(p, &p->cam_model, model)); TRY(dev_upload(p, &p->cam_np, np)); TRY(dev_upload(p, &p->cam_off, off));
  TRY(dev_upload(p, &p->param_cam, pcam)); TRY(dev_upload(p, &p->param_loc, ploc));
  lap("upload observations");
  TRY(dev_alloc(p, &p->tab, (size_t)p->C * CAMTAB_DOUBLES)); TRY(dev_alloc(p, &p->tab_new, (size_t)p->C * CAMTAB_DOUBLES));
  const long tot = p->lay.total();
  for (double** v : {&p->x0, &p->x, &p->x_new, &p->g, &p->s, &p->sinv, &p->v1, &p->v2, &p->sinv2}) {
    TRY(dev_alloc(p, v, (size_t)tot));
    if (hipMemsetAsync(*v, 0, tot * sizeof(double), p->stream) != hipSuccess) return bail(fail(CBA_ERR_HIP, "hipMemset failed"));
  }
  for (double** v : {&p->sinv_state_c, &p->cam_diag, &p->cam_over1, &p->cam_over2, &p->lb_dev, &p->ub_dev, &p->sinv_state_c2, &p->cam_diag2}) {
    TRY(dev_alloc(p, v, (size_t)p->lay.ncp_pad));
    if (hipMemsetAsync(*v, 0, (size_t)p->lay.ncp_pad * sizeof(double), p->stream) != hipSuccess) return bail(fail(CBA_ERR_HIP, "hipMemset failed"));
  }
  TRY(dev_alloc(p, &p->V, (size_t)6 * p->lay.Ppad));
  HIPBAIL(hipMemsetAsync(p->V, 0, (size_t)6 * p->lay.Ppad * sizeof(double), p->stream));
  const int ustride = (nct == 9) ? UPack<9>::STRIDE : UPack<6>::STRIDE;
  TRY(dev_alloc(p, &p->Upacked, (size_t)p->C * ustride + 128));  // + room for the scalars that ride with the blocks (exchange_at)
  lap("allocate vectors");
  if (nct == 9) TRY(configure_kernels<9>(p)); else TRY(configure_kernels<6>(p));
  lap("reorder, upload, allocate");
  if (!p->eval_only) {
    PlanTask& task = *p->plan_task;
    const double t_wait = t_now();
    task.wait_stage(task.two_stage ? 1 : 2);
    if (plan_timing)
      fprintf(stderr, "  plan: %s took %.3f s on its own threads, started before the uploads; waited %.3f s for it\n", task.two_stage ? "the cheap plan" : "dealt streams and codes",
              task.two_stage ? task.seconds_cheap : task.seconds, t_now() - t_wait);
    if (plan_timing) {
      const Reg2Plan& made = task.two_stage ? task.cheap : task.plan;
      fprintf(stderr, "  plan: its phases: point runs %.4f s, jobs %.4f s, concatenation %.4f s (%.0f MB)\n", made.seconds_runs, made.seconds_jobs, made.seconds_concat,
              (made.obs.size() + made.codes.size()) * 4e-6);
    }
    if (task.two_stage ? task.rc_cheap : task.rc)
      return bail(fail(CBA_ERR_UNSUPPORTED, "a world point has more observations inside one camera-group tile than a chunk of the pair plan holds (%d records)",
                       (nct == 9) ? Reg3Cfg<9>::SCHUNK : Reg3Cfg<6>::SCHUNK));
    rc = install_reg2_plan(p, task.two_stage ? task.cheap : task.plan, task.prm);
    p->plan_is_cheap = task.two_stage || task.prm.cheap;
    if (rc) return bail(rc);
  }
  lap("Schur plan (streams, pairs, upload)");
  const long w_build = (long)p->C * ustride;
  p->partial_width = w_build;
  // (+ 1/8: the dealt plan of a two-stage handle has a few chunks — on small problems: workgroups — more or fewer than the cheap one)
  p->partial_capacity = (size_t)std::max<long>((long)p->grid * w_build, (long)(p->tile_grid + p->tile_grid / 8 + 1) * std::max(p->tp.rep, 1) * p->tp.tile_elems);
  TRY(dev_alloc(p, &p->partial, p->partial_capacity));
  TRY(dev_alloc(p, &p->partial4, (size_t)2048 * 4)); TRY(dev_alloc(p, &p->partial1, (size_t)2048)); TRY(dev_alloc(p, &p->partial4b, (size_t)2048 * 4));  // obs rows + constraint rows
  TRY(dev_alloc(p, &p->Sacc, (size_t)ncp * ncp + (size_t)B_SLICES * p->lay.ncp_pad));  // + the rhs accumulator b, B_SLICES rows (k_reg_reduce)
  TRY(dev_alloc(p, &p->tri, (size_t)ncp * (ncp + 1) / 2 + p->lay.ncp_pad));
  if (!p->eval_only) {
    TRY(dev_alloc(p, &p->red, (size_t)p->G * p->tp.tile_elems));
    TRY(dev_alloc(p, &p->Trec, (size_t)std::max<long>(p->N, 1) * ((nct == 9) ? SchurRec<9>::HREC : SchurRec<6>::HREC)));
    TRY(dev_alloc(p, &p->partial_b, (size_t)p->grid * p->lay.ncp_pad));
  }
  TRY(dev_alloc(p, &p->S, (size_t)ncp * ncp)); p->ldw = (ncp + 3) & ~3;
  if (p->want_chol_trace) TRY(dev_alloc(p, &p->chol_trace, (size_t)((ncp + NB - 1) / NB + 2) * 8));
  TRY(dev_alloc(p, &p->Lbuf, (size_t)(ncp + 1) * p->ldw));
  TRY(dev_alloc(p, &p->Xinv, (size_t)((ncp + NB - 1) / NB + 1) * NB * NB));
  TRY(dev_alloc(p, &p->Tinv, (size_t)ncp * p->ldw));
  HIPBAIL(hipMemsetAsync(p->Tinv, 0, (size_t)ncp * p->ldw * sizeof(double), p->stream));
  TRY(dev_alloc(p, &p->rhs, (size_t)p->lay.ncp_pad));
  TRY(dev_alloc(p, &p->scal, 64)); TRY(dev_alloc(p, &p->flags, 4)); TRY(dev_alloc(p, &p->xbuf, 128));
  TRY(dev_alloc(p, &p->fz, 8)); TRY(dev_alloc(p, &p->V2, (size_t)6 * p->lay.Ppad)); TRY(dev_alloc(p, &p->g2, (size_t)tot));
  TRY(dev_alloc(p, &p->U2, (size_t)p->C * ustride + 128));
  HIPBAIL(hipMemsetAsync(p->V2, 0, (size_t)6 * p->lay.Ppad * sizeof(double), p->stream)); HIPBAIL(hipMemsetAsync(p->g2, 0, (size_t)tot * sizeof(double), p->stream));
  HIPBAIL(hipMemsetAsync(p->scal, 0, 64 * sizeof(double), p->stream));
  HIPBAIL(hipMemsetAsync(p->flags, 0, 4 * sizeof(int), p->stream));
  HIPBAIL(hipMemsetAsync(p->Sacc, 0, ((size_t)ncp * ncp + (size_t)B_SLICES * p->lay.ncp_pad) * sizeof(double), p->stream));
#undef TRY
  {
    std::lock_guard<std::mutex> lock(g_pool_mu);
    DevicePool& pool = g_pool[dev];
    for (size_t i = 0; i < pool.staging.size(); ++i)
      if (pool.staging[i].second >= (size_t)tot && pool.staging[i].second <= 4 * (size_t)tot + 4096) {
        p->h_vec = pool.staging[i].first; p->h_vec_doubles = pool.staging[i].second;
        pool.staging.erase(pool.staging.begin() + (long)i);
        break;
      }
  }
  HIPBAIL(hipEventCreateWithFlags(&p->h_vec_sent, hipEventDisableTiming));
  if (!p->h_vec) {
    p->h_vec_doubles = (size_t)tot;
    HIPBAIL(hipHostMalloc((void**)&p->h_vec, p->h_vec_doubles * sizeof(double), hipHostMallocDefault));
  }
  lap("solver buffers");
  HIPBAIL(hipDeviceSynchronize());
  upload_stage_release(p);
  lap("device synchronize");
  if (p->plan_task) {
    if (p->plan_task->two_stage) {  // the thread goes on dealing: it reads these two arrays
      p->plan_task->hcam_keep = std::move(hcam);
      p->plan_task->hps_keep = std::move(hps);
    } else {
      drop_plan_task(p);
    }
  }
  if (plan_timing) fprintf(stderr, "cba_create: %.3f s in total\n", t_now() - t_enter);
  *out = p;
  return CBA_OK;
}

static int maybe_swap_plan(cba_problem* p, bool wait);

int cba_plan_wait(cba_problem* p) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_plan_wait: null problem");
  HIPCHK(hipSetDevice(p->device));
  return maybe_swap_plan(p, true);
}

int cba_set_loss(cba_problem* p, int32_t loss, double f_scale) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_set_loss: null problem");
  if (loss < CBA_LOSS_LINEAR || loss > CBA_LOSS_ARCTAN) return fail(CBA_ERR_INVALID, "cba_set_loss: unknown loss %d", loss);
  if (loss != CBA_LOSS_LINEAR && !(f_scale > 0.0)) return fail(CBA_ERR_INVALID, "cba_set_loss: f_scale must be positive");
  p->loss = loss; p->f_scale = f_scale;
  p->have_build = false;  // blocks and gradient of the current point belong to the old loss
  p->spec_valid = false; p->spec_enqueued = false;
  return CBA_OK;
}

int cba_get_info(cba_problem* p, cba_info* o) {
  if (!p || !o) return fail(CBA_ERR_INVALID, "null argument");
  o->n_cams = p->C; o->n_points = p->P; o->n_cam_params = p->ncp; o->n_params = p->ncp + 3 * p->P; o->n_obs = p->N;
  o->n_chunks = p->n_chunks; o->grid_blocks = p->grid;
  o->plan_state = !p->plan_is_cheap ? 0 : (p->plan_task ? 1 : 2); o->plan_error = p->plan_error;
  o->schur_groups = p->G; o->schur_tiles = p->n_tiles; o->schur_grid = p->tile_grid; o->schur_stream_len = p->tile_stream_len; o->schur_pairs = p->n_pairs;
  o->max_obs_per_point = p->max_obs_per_point; o->device_bytes = p->device_bytes; o->n_heavy_points = p->n_heavy;
  {
    const bool cs = p->cs.n_sc && !p->det_m && !p->n_heavy;
    const bool camg = cs ? (p->nct == 6 ? build_cs_camg<6>(p) : build_cs_camg<9>(p)) : (p->nct == 6 ? build_camg<6>(p) : build_camg<9>(p));
    const bool uglob = cs && (p->nct == 6 ? build_cs_uglob<6>(p) : build_cs_uglob<9>(p));
    o->build_camg = (camg ? 1 : 0) | (p->tab_global ? 2 : 0) | (cs ? 4 : 0) | (uglob ? 8 : 0);
  }
  return CBA_OK;
}

int cba_enable_timers(cba_problem* p, int32_t on) { if (!p) return fail(CBA_ERR_INVALID, "null"); p->timers_on = on != 0; return CBA_OK; }
int cba_reset_timers(cba_problem* p) {
  if (!p) return fail(CBA_ERR_INVALID, "null");
  (void)hipSetDevice(p->device);
  drain_timers(p);
  for (int t = 0; t < T_COUNT; ++t) { p->t_ms[t] = 0; p->t_calls[t] = 0; }
  return CBA_OK;
}
int cba_get_timers(cba_problem* p, double* ms, int64_t* calls) {
  if (!p) return fail(CBA_ERR_INVALID, "null");
  (void)hipSetDevice(p->device);
  drain_timers(p);
  for (int t = 0; t < T_COUNT; ++t) { if (ms) ms[t] = p->t_ms[t]; if (calls) calls[t] = p->t_calls[t]; }
  return CBA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// layout conversion between the reference parameter vector and the device vector (points SoA)
static void pack_host(const cba_problem* p, const double* x_ref, double* v, double pad_value) {
  const VecLayout& L = p->lay;
  std::fill(v, v + L.total(), pad_value);
  for (int i = 0; i < L.ncp; ++i) v[i] = x_ref[i];
  const double* pts = x_ref + L.ncp;
  double* vx = v + L.ncp_pad;
  for (int q = 0; q < L.P; ++q) {
    vx[q] = pts[3 * q]; vx[L.Ppad + q] = pts[3 * q + 1]; vx[2 * L.Ppad + q] = pts[3 * q + 2];
  }
}
static void unpack_host(const cba_problem* p, const double* v, double* x_ref) {
  const VecLayout& L = p->lay;
  for (int i = 0; i < L.ncp; ++i) x_ref[i] = v[i];
  double* pts = x_ref + L.ncp;
  const double* vx = v + L.ncp_pad;
  for (int q = 0; q < L.P; ++q) {
    pts[3 * q] = vx[q]; pts[3 * q + 1] = vx[L.Ppad + q]; pts[3 * q + 2] = vx[2 * L.Ppad + q];
  }
}

static int launch_cam_prep(cba_problem* p, const double* xvec, double* tab) {
  ScopedTimer t(p, T_CAM_PREP);
  hipLaunchKernelGGL(k_cam_prep, dim3((p->C + 63) / 64), dim3(64), 0, p->stream, xvec, p->cam_const, p->cam_model,
                     p->cam_np, p->cam_off, p->C, tab);
  return CBA_OK;
}

// cost at (xvec, tab) -> scal[slot]; flags[0] set when a residual is not finite
static int launch_cost(cba_problem* p, const double* xvec, const double* tab, int slot, double* r_out, int* rows_out = nullptr) {
  const int grid = (int)std::min<long>((p->N + BLOCK - 1) / BLOCK, 1024);
  {
    ScopedTimer t(p, T_COST);
    auto launch = [&](auto kernel, double* r, const int* ord) {
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds_cost(p), p->stream, p->obs_u, p->obs_v, p->obs_cam,
                         p->obs_pt, p->N, xvec, p->lay, tab, p->C, p->loss, p->f_scale, p->partial1, p->flags, r, ord);
    };
    if (r_out) { if (p->tab_global) launch(k_cost<true, true>, r_out, (const int*)p->order); else launch(k_cost<true>, r_out, (const int*)p->order); }
    else { if (p->tab_global) launch(k_cost<false, true>, (double*)nullptr, (const int*)nullptr); else launch(k_cost<false>, (double*)nullptr, (const int*)nullptr); }
  }
  ScopedTimer t(p, T_VECTOR);
  int rows = grid;
  if (p->con.n_con) {  // constraint rows: their cost lands in extra partial rows (and their residuals after the 2N reprojection rows)
    hipLaunchKernelGGL(k_con_eval<false>, dim3(p->con_grid), dim3(BLOCK), 0, p->stream, p->con, xvec, p->lay, p->loss, p->f_scale,
                       (double*)nullptr, p->partial1 + grid, p->flags, r_out ? r_out + 2 * p->N : (double*)nullptr);
    rows += p->con_grid;
  }
  if (rows_out) { *rows_out = rows; return CBA_OK; }  // single rank: k_publish sums the rows
  hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial1, rows, 1, p->scal + slot);
  return CBA_OK;  // sharded solves: the caller's exchange() sums scal[slot] and the flags over the ranks
}

// Single-rank fused iteration over camera-sorted super-chunks: the step scalars of the point block come out of k_backsub, one workgroup
// (k_step_cam) adds the camera block, takes the subspace step and prepares the trial point's cameras, and the trial build forms the point
// entries of the trial point itself — k_step_scalars, k_step_finish / k_step_small and k_trial_update are not launched.
static bool fused_trial(const cba_problem* p, bool compact) { return compact && p->cs.n_sc && !p->det_m && !p->n_heavy && !p->con.n_con; }

// Build pass at xvec (camera table tab) into the given outputs; the rho sum lands in scal[cost_slot].
// trial != nullptr (fused_trial): k_build_cs<.., TRIAL> forms the point entries of xvec = x_new from *trial while it stages them.
template <int NC>
static int run_build_into(cba_problem* p, const double* xvec, const double* tab, double* V, double* g, double* Upacked, int cost_slot,
                          const double* skip = nullptr, bool defer_exchange = false, bool compact = false, int flag_slot = 0,
                          const TrialSrc* trial = nullptr, const PubArgs* pub = nullptr) {
  RoctxRange range("cba:build");
  {
    ScopedTimer t(p, T_BUILD);
    if (p->n_heavy)  // fragments of heavy points add their sums by atomics
      hipLaunchKernelGGL(k_zero_heavy, dim3((p->n_heavy + 63) / 64), dim3(64), 0, p->stream, p->heavy_pts, p->n_heavy, p->lay, V, 6,
                         g + p->lay.ncp_pad, 3);
    auto launch_build = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(p->grid), dim3(BLOCK), lds_build<NC>(p), p->stream, p->obs_u, p->obs_v, p->obs_cam,
                         p->obs_pt, p->pt_start, p->chunk_start, p->chunk_pts, p->n_chunks, xvec, p->lay, tab, p->C, p->loss, p->f_scale,
                         V, g, p->partial, p->partial1, p->flags + flag_slot, skip, p->det);
    };
    if (p->cs.n_sc && !p->det_m && !p->n_heavy) {  // camera-sorted super-chunks: the camera blocks accumulate in registers
      const TrialSrc tsrc = trial ? *trial : TrialSrc{};
      auto launch_cs = [&](auto kernel, size_t lds) {
        hipLaunchKernelGGL(kernel, dim3(p->grid), dim3(BLOCK), lds, p->stream, p->cs, xvec, p->lay, tab, p->C, p->loss, p->f_scale,
                           V, g, p->partial, p->partial1, p->flags + flag_slot, skip, tsrc);
      };
      if (build_cs_uglob<NC>(p)) {
        (void)hipMemsetAsync(p->partial, 0, (size_t)p->C * UPack<NC>::STRIDE * sizeof(double), p->stream);  // the ONE copy the atomics add to
        if (trial) launch_cs(k_build_cs<NC, true, true, true>, lds_build_cs<NC>(p, true, true));
        else launch_cs(k_build_cs<NC, true, true>, lds_build_cs<NC>(p, true, true));
      } else if (build_cs_camg<NC>(p)) {
        if (trial) launch_cs(k_build_cs<NC, true, false, true>, lds_build_cs<NC>(p, true));
        else launch_cs(k_build_cs<NC, true>, lds_build_cs<NC>(p, true));
      } else {
        if (trial) launch_cs(k_build_cs<NC, false, false, true>, lds_build_cs<NC>(p, false));
        else launch_cs(k_build_cs<NC, false>, lds_build_cs<NC>(p, false));
      }
    } else
    switch (p->det_m) {  // deterministic: fixed-order camera sums (k_build<NC, tasks per thread>)
      case 3: launch_build(k_build<NC, 3>); break;
      case 5: launch_build(k_build<NC, 5>); break;
      case 8: launch_build(k_build<NC, 8>); break;
      case 16: { if constexpr (NC == 6) launch_build(k_build<NC, 16>); } break;
      default:
        if (build_camg<NC>(p)) launch_build(k_build<NC, 0, true>); else launch_build(k_build<NC, 0>);
        break;
    }
  }
  {
    ScopedTimer t(p, T_BUILD_REDUCE);
    const int w = p->C * UPack<NC>::STRIDE;
    // rows of per-workgroup partial blocks to sum; the global-atomics form of k_build_cs leaves ONE row
    const int u_rows = (p->cs.n_sc && !p->det_m && !p->n_heavy && build_cs_uglob<NC>(p)) ? 1 : p->grid;
    if (compact) {  // single-rank fused step: gradient entries written by the row reduction, rho sum by k_publish
      if (pub)  // ... whose packet leaves with one more workgroup of this launch
        hipLaunchKernelGGL(k_reduce_rows_pub, dim3((w + 63) / 64 + 1), dim3(64, REDUCE_RY), 0, p->stream, (const double*)p->partial, u_rows, w, Upacked, g,
                           (const int*)p->cam_off, (const int*)p->cam_np, (int)UPack<NC>::STRIDE, (int)UPack<NC>::TRI, *pub);
      else
        hipLaunchKernelGGL(k_reduce_rows, dim3((w + 63) / 64), dim3(64, REDUCE_RY), 0, p->stream, p->partial, u_rows, w, Upacked, g,
                           (const int*)p->cam_off, (const int*)p->cam_np, (int)UPack<NC>::STRIDE, (int)UPack<NC>::TRI);
      return CBA_OK;
    }
    hipLaunchKernelGGL(k_reduce_rows, dim3((w + 63) / 64), dim3(64, REDUCE_RY), 0, p->stream, p->partial, u_rows, w, Upacked,
                       (double*)nullptr, (const int*)nullptr, (const int*)nullptr, 1, 0);
    int rho_rows = p->grid;
    if (p->con.n_con) {  // constraint rows: f, u, their share of g_p and of the squared column norms
      HIPCHK(hipMemsetAsync(p->con.cdiag, 0, (size_t)3 * p->lay.Ppad * sizeof(double), p->stream));
      hipLaunchKernelGGL(k_con_eval<true>, dim3(p->con_grid), dim3(BLOCK), 0, p->stream, p->con, xvec, p->lay, p->loss, p->f_scale, g,
                         p->partial1 + p->grid, p->flags, (double*)nullptr);
      rho_rows += p->con_grid;
    }
    hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial1, rho_rows, 1, p->scal + cost_slot);  // rho sum
    if (defer_exchange) return CBA_OK;  // the caller sums the blocks together with its scalars and unpacks the gradient
    int rc = allreduce_sum(p, Upacked, (size_t)w);  // camera blocks U_c and g_c: sum over the point shards
    if (rc) return rc;                               // (the rho sum rides with the caller's scalar exchange)
    hipLaunchKernelGGL((k_unpack_camera_grad<NC>), dim3((p->C * NC + 255) / 256), dim3(256), 0, p->stream, Upacked,
                       p->cam_off, p->cam_np, p->C, g);
  }
  return CBA_OK;
}

template <int NC>
// the build at the current x flags a non-finite residual in flags[3] (flags[0] belongs to trial points)
static int run_build(cba_problem* p) { return run_build_into<NC>(p, p->x, p->tab, p->V, p->g, p->Upacked, 8, nullptr, false, false, 3); }

template <int NC>
static int run_jv(cba_problem* p, int nv, int* rows_out = nullptr, const double* xvec = nullptr, const double* tab = nullptr) {
  ScopedTimer t(p, T_JV);
  if (!xvec) { xvec = p->x; tab = p->tab; }  // (the speculative linearisation evaluates the trial point: x_new, tab_new)
  const int grid = nv == 1 ? p->jv_grid : (int)std::min<long>((p->N + BLOCK - 1) / BLOCK, 1024);
  auto launch_jv = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lds_jv(p, nv), p->stream, p->obs_u, p->obs_v, p->obs_cam, p->obs_pt,
                       p->N, xvec, p->lay, tab, p->cam_off, p->C, p->loss, p->f_scale, p->v1, p->v2, p->partial4);
  };
  if (nv == 1) { if (p->tab_global) launch_jv(k_jv<NC, 1, true>); else launch_jv(k_jv<NC, 1>); }
  else { if (p->tab_global) launch_jv(k_jv<NC, 2, true>); else launch_jv(k_jv<NC, 2>); }
  int rows = grid;
  if (p->con.n_con) {
    if (nv == 1) hipLaunchKernelGGL(k_con_jv<1>, dim3(p->con_grid), dim3(BLOCK), 0, p->stream, p->con, p->lay, p->v1, p->v2, p->partial4 + 4 * grid);
    else hipLaunchKernelGGL(k_con_jv<2>, dim3(p->con_grid), dim3(BLOCK), 0, p->stream, p->con, p->lay, p->v1, p->v2, p->partial4 + 4 * grid);
    rows += p->con_grid;
  }
  if (rows_out) { *rows_out = rows; return CBA_OK; }  // compact fused step: k_lin_finish sums the rows
  hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, rows, 4, p->scal + 12);
  return CBA_OK;  // scal[12..15] are summed over the ranks by the caller's exchange()
}

// device part of the linearisation (no host synchronisation): build unless the accepted trial brought its own, Jacobi
// scale, scalars, ||J_h g_h||^2, the scalar exchange of a sharded solve
template <int NC>
static int run_lin_chain(cba_problem* p, bool scalars = true, bool compact = false, double radius = 0.0, bool defer_finish = false) {
  RoctxRange range("cba:linearize");
  if (!p->have_build) {
    int rcb = run_build<NC>(p);
    if (rcb) return rcb;
    p->cost_pending = true;
    p->have_build = true;  // V, g, Upacked belong to the current x until it changes
    p->spec_valid = false;
  }
  if (compact && p->spec_valid) {
    // the accepted trial was linearised speculatively behind the previous iteration's k_publish (run_step): scale, vector sums and
    // ||J_h g_h||^2 are there, only the damping depends on the radius the host has just decided
    p->spec_valid = false;
    if (defer_finish) {  // (k_tprep sums the partial rows itself: no launch here)
      p->lf = LinFin{p->partial4b, p->partial1, p->partial4, vec_grid(p->lay.total()), p->spec_rows_jv, radius, p->scal, p->fz};
      p->lf_pending = true;
      return CBA_OK;
    }
    ScopedTimer t(p, T_SCALE_SCALARS);
    hipLaunchKernelGGL(k_lin_finish, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4b, p->partial1, vec_grid(p->lay.total()), p->partial4, p->spec_rows_jv, radius,
                       p->scal, p->fz);
    return CBA_OK;
  }
  p->spec_valid = false;
  const long tot = p->lay.total();
  const int vg = vec_grid(tot);
  const bool bnd = defer_finish && p->bounds_on;  // bounded fused iteration: the Coleman-Li scaling of the camera block happens inside k_scale_lin
  // (cam_scaled: sinv holds the EFFECTIVE scale of the camera block and sinv_state_c its Jacobi state — after cba_set_camera_scaling and after a
  // bounded fused linearisation alike, so the two routes can follow each other)
  if (bnd && !p->first_scale && !p->cam_scaled)  // first bounded fused step after linearisations that never rescaled: sinv IS the state
    HIPCHK(hipMemcpyAsync(p->sinv_state_c, p->sinv, (size_t)p->lay.ncp_pad * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  if (p->cam_scaled && !bnd)  // the monotone-max rule of the Jacobi scale runs on the unmodified state of the camera block
    HIPCHK(hipMemcpyAsync(p->sinv, p->sinv_state_c, (size_t)p->lay.ncp_pad * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  p->cam_state_saved = false;
  {
    ScopedTimer t(p, T_SCALE_SCALARS);
    if (compact) {
      // single-rank fused step: scale + scalars in one vector pass, the three reductions and the damping in one launch
      // (bounded: the camera block's Jacobi state is read and written in place — sinv_state_c — and sinv takes the effective scale)
      const BoundArgs ba = bnd ? BoundArgs{p->lb_dev, p->ub_dev, p->sinv_state_c, p->sinv_state_c, p->cam_diag} : BoundArgs{};
      if (bnd) p->cam_scaled = true;
      auto launch_sl = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(vg), dim3(BLOCK), 0, p->stream, p->Upacked, p->V, p->param_cam, p->param_loc, p->lay,
                           p->first_scale ? 1 : 0, p->sinv, p->con.n_con ? (const double*)p->con.cdiag : (const double*)nullptr, p->x, p->g, p->v1,
                           p->partial4b, p->partial1, (const double*)nullptr, ba);
      };
      if (bnd) launch_sl(k_scale_lin<NC, true>); else launch_sl(k_scale_lin<NC, false>);
      p->first_scale = false;
      int rows_jv = 0;
      int rcj = run_jv<NC>(p, 1, &rows_jv);
      if (rcj) return rcj;
      if (defer_finish) {
        p->lf = LinFin{p->partial4b, p->partial1, p->partial4, vg, rows_jv, radius, p->scal, p->fz};
        p->lf_pending = true;
        return CBA_OK;
      }
      hipLaunchKernelGGL(k_lin_finish, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4b, p->partial1, vg, p->partial4, rows_jv, radius,
                         p->scal, p->fz);
      return CBA_OK;
    }
    hipLaunchKernelGGL((k_scale_update<NC>), dim3(vg), dim3(BLOCK), 0, p->stream, p->Upacked, p->V, p->param_cam, p->param_loc,
                       p->lay, p->first_scale ? 1 : 0, p->sinv, p->con.n_con ? (const double*)p->con.cdiag : (const double*)nullptr);
    p->first_scale = false;
    if (!scalars) return exchange(p, SLOT(8), false);  // cba_linearize_build: the caller rescales first (cba_set_camera_scaling)
    hipLaunchKernelGGL(k_lin_scalars, dim3(vg), dim3(BLOCK), 0, p->stream, p->x, p->g, p->sinv, tot, p->lay.ncp_pad,
                       p->rank == 0 ? 1 : 0, 0, p->v1, p->partial4, p->partial1);
    hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, vg, 4, p->scal + 0);
    hipLaunchKernelGGL(k_reduce_narrow<true>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial1, vg, 1, p->scal + 4);
  }
  {
    int rcj = run_jv<NC>(p, 1);
    if (rcj) return rcj;
  }
  // sums of the linearisation (scal[0..3]), rho sum (8), ||J v||^2 terms (12..15), max |g| (4): one all-reduce
  return exchange(p, SLOT(0) | SLOT(1) | SLOT(2) | SLOT(3) | SLOT(8) | SLOT(12) | SLOT(13) | SLOT(14) | SLOT(15), true);
}

static void read_linearization(cba_problem* p, cba_linearization* out) {
  if (p->cost_pending) { p->cost_x = p->h_flags[3] ? NAN : 0.5 * p->h_scal[8]; p->cost_pending = false; }  // else: the cost of the accepted trial
  p->gh_sq = p->h_scal[0];
  out->gh_sq = p->h_scal[0];
  out->x_scaled_norm = std::sqrt(p->h_scal[1]);
  out->x_norm = std::sqrt(p->h_scal[2]);
  out->g_norm_inf = p->h_scal[4];
  out->cost = p->cost_x;
  out->jg_sq = p->h_scal[12];
}

template <int NC>
static int run_linearize(cba_problem* p, cba_linearization* out) {
  int rc = run_lin_chain<NC>(p, true, !p->sharded(), 1.0);  // single rank: folded launches (the damping k_lin_finish derives is not used here)
  if (rc) return rc;
  rc = sync_scalars(p, 16);
  if (rc) return rc;
  read_linearization(p, out);
  return CBA_OK;
}

// The dense solve is ncp / 32 + 2 small dependent launches.  Rounds 1-2 replayed them from a hipGraph recorded at create time; since the fused
// iteration keeps the host a whole iteration ahead of the device, plain launches are the faster form (round 3, per iteration: cfg2 139 against
// 149 us, cfg4 645 against 654 — the kernel behind a graph launch starts ~10 us late), and the graph is gone (round 4).
static int enqueue_cholesky(cba_problem* p) {
  const int n = p->ncp, nbk = (n + NB - 1) / NB;
  for (int k = -1; k < nbk; ++k) {
    // step k: panel k solved for the blocks below it (and the rhs row), D_k+1 factored; extra workgroups apply the
    // rank-NB update of panel k - 1 to the blocks right of the current panel
    const int n_panel = k < 0 ? 1 : nbk - k;
    const int x = nbk - k - 1, n_trailing = k < 1 ? 0 : x * (x + 1) / 2;
    const int n_inverse = (k >= 1) ? (nbk - k) * k : 0;  // blocks (i >= k, j < k) of T = L^-T take the term of panel k - 1
    hipLaunchKernelGGL(k_chol_step, dim3(n_panel + n_trailing + n_inverse), dim3(CHOL_THREADS), 0, p->stream, p->Lbuf, n, p->ldw, k, p->flags, p->chol_trace, p->Xinv, p->Tinv);
  }
  hipLaunchKernelGGL(k_chol_apply, dim3(nbk), dim3(APPLY_THREADS), (size_t)n * 8, p->stream, (const double*)p->Tinv, (const double*)p->Lbuf, n, p->ldw, p->s);
  return CBA_OK;
}

static int run_cholesky(cba_problem* p) {
#ifdef CBA_PROFILING
  if (p->chol_trace) {  // traced run: dump the stamps of the critical workgroups
    const int nbk = (p->ncp + NB - 1) / NB;
    HIPCHK(hipMemsetAsync(p->chol_trace, 0, (size_t)(nbk + 2) * 8 * sizeof(long long), p->stream));
    int rc = enqueue_cholesky(p);
    if (rc) return rc;
    std::vector<long long> h((size_t)(nbk + 2) * 8);
    HIPCHK(hipMemcpyAsync(h.data(), p->chol_trace, h.size() * sizeof(long long), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    for (int s = 0; s <= nbk; ++s) {
      fprintf(stderr, "chol step %2d:", s - 1);
      for (int ph = 1; ph < 7; ++ph)
        if (h[s * 8 + ph] && h[s * 8 + ph - 1]) fprintf(stderr, " ph%d %6.2f us", ph, (h[s * 8 + ph] - h[s * 8 + ph - 1]) * 0.01); else if (h[s * 8 + ph]) fprintf(stderr, " ph%d (%6.2f)", ph, (h[s * 8 + ph] - h[s * 8]) * 0.01);
      if (s < nbk && h[(s + 1) * 8] && h[s * 8]) fprintf(stderr, "  | to next step start %6.2f us", (h[(s + 1) * 8] - h[s * 8]) * 0.01);
      fprintf(stderr, "\n");
    }
    return CBA_OK;
  }
#endif
  ScopedTimer t(p, T_CHOLESKY);
  return enqueue_cholesky(p);
}

// scalars of the damped step s: ||p||^2 and <g_h, p> (scal[16], [17]) and ||w||^2, w = p - (<g_h,p> / ||g_h||^2) g_h (scal[20]).
// formula_w: the fused step derives ||w||^2 from the first two on the device (k_fused_subspace) and spends one
// collective; otherwise it is measured by a pass of its own, which stays accurate when w is tiny against p.
static int run_step_scalars(cba_problem* p, bool formula_w, bool compact = false) {
  ScopedTimer t(p, T_VECTOR);
  const long tot = p->lay.total();
  const int vg = vec_grid(tot);
  const long first = (p->rank == 0) ? 0 : p->lay.ncp_pad;  // replicated camera entries are counted on rank 0 only
  if (compact && first == 0 && tot <= STEP_SMALL_MAX) {  // small problems: sums and subspace step by one workgroup, one launch
    hipLaunchKernelGGL(k_step_small, dim3(1), dim3(STEP_SMALL_THREADS), 0, p->stream, p->g, p->sinv, p->s, tot, p->scal, (const int*)p->flags, p->fz);
    return CBA_OK;
  }
  hipLaunchKernelGGL(k_step_scalars, dim3(vg), dim3(BLOCK), 0, p->stream, p->g, p->sinv, p->s, tot, first, p->partial4);
  if (compact) {  // single-rank fused step: the reduction and the subspace step in one launch
    hipLaunchKernelGGL(k_step_finish, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, vg, p->scal, (const int*)p->flags, p->fz);
    return CBA_OK;
  }
  hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, vg, 4, p->scal + 16);
  if (formula_w) return exchange(p, SLOT(16) | SLOT(17), false);
  int rcv = allreduce_sum(p, p->scal + 16, 2);
  if (rcv) return rcv;
  hipLaunchKernelGGL(k_w_scalar, dim3(vg), dim3(BLOCK), 0, p->stream, p->g, p->sinv, p->s, tot, first, p->scal + 17, p->gh_sq,
                     (const double*)nullptr, p->partial1);
  hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial1, vg, 1, p->scal + 20);
  return exchange(p, SLOT(20), false);  // ||w||^2 and the flags
}

// Two-stage plan: the handle started with the cheap Schur plan; once the plan thread has the dealt one, the next damped step uploads it and goes on
// with it (the cheap plan's buffers stay in the arena until the handle goes).  Called between iterations by the thread that drives the handle;
// `wait`: block until the dealt plan is there (cba_plan_wait).
static int maybe_swap_plan(cba_problem* p, bool wait) {
  PlanTask* task = p->plan_task;
  if (!task) return CBA_OK;
  if (wait) task->wait_stage(2);
  if (task->stage.load(std::memory_order_acquire) < 2) return CBA_OK;
  int rc = CBA_OK;
  if (task->rc != 0 && task->two_stage) {
    // the background build (dealing, its hipMalloc / upload beside the running solve) failed: the handle keeps the quick plan — same sums to
    // rounding, pair kernel 1.5-1.75x slower — and says so: cba_info.plan_state = 2, plan_error; cba_plan_wait returns the error
    p->plan_error = task->rc;
    fprintf(stderr, "caliscope_ba: the balanced Schur plan could not be built (error %d, %.3f s on its thread): the handle keeps the quick plan\n", task->rc, task->seconds);
    if (wait) rc = fail(task->rc, "cba_plan_wait: the balanced Schur plan could not be built in the background (error %d); the handle keeps the quick plan", task->rc);
  }
  if (task->rc == 0) {
    const size_t need = (size_t)task->install.tile_grid * std::max(task->install.tp.rep, 1) * task->install.tp.tile_elems;
    if (need > p->partial_capacity) {  // (the dealt plan has a few chunks more or fewer than the cheap one; cba_create leaves headroom: rare)
      rc = dev_alloc(p, &p->partial, need);
      if (!rc) p->partial_capacity = need;
    }
    if (!rc) {
      apply_install(p, task->install);
      p->plan_is_cheap = false;
      if (plan_timing_on()) fprintf(stderr, "  plan: the dealt plan (%.3f s on its own threads, upload included) swapped in\n", task->seconds);
    }
  }
  drop_plan_task(p);
  return rc;
}

#ifdef CBA_PROFILING
// profiling build of the pair kernel (six-parameter cameras): phase clocks per wave, printed per launch
template <int NC>
static int run_pairs_clocked(cba_problem* p) {
  if constexpr (NC != 6) return fail(CBA_ERR_UNSUPPORTED, "CBA_SCHUR_CLOCK: six-parameter cameras only");
  else {
    const int nw = Reg3Cfg<6>::NWAVES;
    const size_t n = (size_t)p->tile_grid * nw * 8;
    long long* d = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(long long)) != hipSuccess) return fail(CBA_ERR_HIP, "debug buffer");
    (void)hipMemsetAsync(d, 0, n * sizeof(long long), p->stream);
    if (raise_lds_ceiling((const void*)k_schur_reg3_clk<6, 1, 2>, Reg3Cfg<6>::LDS_BYTES)) return CBA_ERR_HIP;
    hipLaunchKernelGGL((k_schur_reg3_clk<6, 1, 2>), dim3(p->tile_grid), dim3(Reg3Cfg<6>::LAUNCH_THREADS), Reg3Cfg<6>::LDS_BYTES, p->stream, p->tp, p->Trec, p->partial, d, (const double*)p->tab);
    std::vector<long long> h(n);
    (void)hipMemcpyAsync(h.data(), d, n * sizeof(long long), hipMemcpyDeviceToHost, p->stream);
    (void)hipStreamSynchronize(p->stream);
    (void)hipFree(d);
    static const char* names[4] = {"wait for loads", "barrier", "issue", "pairs"};
    auto report = [&](const char* who, auto pick) {
      double sum[8] = {0}, tmax = 0.0;
      size_t waves = 0;
      for (size_t w = 0; w < n / 8; ++w) {
        if (!h[w * 8 + 4] || !pick(w / nw)) continue;
        ++waves;
        for (int k = 0; k < 7; ++k) sum[k] += (double)h[w * 8 + k];
        tmax = std::max(tmax, (double)h[w * 8 + 6]);
      }
      const double wv = (double)std::max<size_t>(waves, 1);
      fprintf(stderr, "%s phases, mean clocks per wave (%zu waves, %.1f trips, %.1f pair iterations each):", who, waves, sum[4] / wv, sum[5] / wv);
      for (int k = 0; k < 4; ++k) fprintf(stderr, "  %s %.0f", names[k], sum[k] / wv);
      fprintf(stderr, "  | in-loop total %.0f, wave lifetime mean %.0f max %.0f\n", (sum[0] + sum[1] + sum[2] + sum[3]) / wv, sum[6] / wv, tmax);
    };
    report("k_schur_reg3", [](size_t) { return true; });
    return CBA_OK;
  }
}
#endif

// device part of the damped step; lam_dev != nullptr: the damping is read from device memory (fused step)
template <int NC>
static int run_newton_chain(cba_problem* p, double lam, const double* lam_dev, bool compact = false) {
  const int ncp = p->ncp;
  if (p->plan_task) {
    const int rcs = maybe_swap_plan(p, false);
    if (rcs) return rcs;
  }
  RoctxRange range("cba:damped_step");
  {
    RoctxRange r2("cba:schur");
    ScopedTimer t(p, T_SCHUR);
    {
      const bool linf = p->lf_pending;  // (set by run_lin_chain(defer_finish) of the same cba_step: never in fixed-order mode)
      p->lf_pending = false;
      auto launch_tprep = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(p->grid), dim3(BLOCK), lds_tprep<NC>(p), p->stream, p->obs_u, p->obs_v, p->obs_cam, p->obs_pt,
                           p->chunk_start, p->n_chunks, p->x, p->lay, p->tab, p->cam_off, p->C, p->loss, p->f_scale, lam, lam_dev, p->V, p->g,
                           p->sinv, p->Trec, p->partial_b, p->flags, p->det, linf ? p->lf : LinFin{});
      };
      switch (p->det_m) {
        case 3: launch_tprep(k_tprep<NC, 3>); break;
        case 5: launch_tprep(k_tprep<NC, 5>); break;
        case 8: launch_tprep(k_tprep<NC, 8>); break;
        case 16: { if constexpr (NC == 6) launch_tprep(k_tprep<NC, 16>); } break;
        default:
          if (linf) { if (p->tab_global) launch_tprep(k_tprep<NC, 0, true, true>); else launch_tprep(k_tprep<NC, 0, false, true>); }
          else if (p->tab_global) launch_tprep(k_tprep<NC, 0, true>); else launch_tprep(k_tprep<NC, 0>);
          break;
      }
      // (the per-workgroup rhs rows k_tprep leaves in partial_b are summed by an extra grid row of k_reg_reduce, behind the pair kernel)
      ScopedTimer tpairs(p, T_SCHUR_PAIRS);  // nested in "schur": the pair kernel alone
      constexpr int SP = RegCfg<NC>::SPLIT, MW = RegCfg<NC>::MINW;
#ifdef CBA_PROFILING
      if (p->schur_clock) {  // phase clocks per wave, printed per launch (tools/build_profiling_lib.sh)
        int rcc = run_pairs_clocked<NC>(p);
        if (rcc) return rcc;
      } else
#endif
      hipLaunchKernelGGL((k_schur_reg3<NC, SP, MW>), dim3(p->tile_grid), dim3(Reg3Cfg<NC>::LAUNCH_THREADS), Reg3Cfg<NC>::LDS_BYTES, p->stream, p->tp, p->Trec, p->partial, (const double*)p->tab);
    }
  }
  bool small_solve = false;
  // single rank, nothing else adds to the diagonal camera blocks (heavy points, constraint rows) and the pair kernel has unprimed its sums: k_schur_finalize
  // folds the helper-thread sums of the diagonal blocks itself, k_reg_fold is not launched
  const bool fold_in_finalize = !p->n_heavy && !p->con.n_con && !p->sharded();
  {
    ScopedTimer t(p, T_SCHUR_REDUCE);
    // (fold + unprime + finalize as ONE kernel, a thread per camera pair, measured slower than the three launches: 42 instead of 31 us — 2080 threads
    // with 36 entries each against 147k threads with one)
    // one launch for the reduction and the finalisation where nothing comes between them (k_reg_finalize; CBA_REG_FINALIZE=0: the two launches)
    const bool fused_finalize = fold_in_finalize && !(ncp <= SMALL_N && !p->chol_trace) && p->fuse_reg_finalize;
    if (fused_finalize) {
      const int per = REG_REDUCE_Y_MAX / p->reg_reduce_y;
      const int tile_x = (p->gsz * p->gsz * NC * NC + 64 * per - 1) / (64 * per), fold_x = (p->gsz * NC * NC + 63) / 64, rhs_x = (p->lay.ncp_pad + 63) / 64;
      hipLaunchKernelGGL((k_reg_finalize<NC>), dim3(p->G * fold_x + rhs_x + p->n_tiles * tile_x), dim3(64, REG_REDUCE_Y_MAX), 0, p->stream, p->tp, p->tile_wg_begin, p->partial,
                         p->cam_off, p->cam_np, ncp, p->n_tiles, p->G, p->reg_reduce_y, fold_x, rhs_x, tile_x, (const double*)p->partial_b, p->grid, p->lay.ncp_pad, p->Upacked, p->g,
                         p->sinv, lam, lam_dev, p->cam_diag, p->S, p->rhs, p->Lbuf, p->ldw);
    } else {
      hipLaunchKernelGGL(k_reg_reduce, dim3(std::max((p->tp.tile_elems + 63) / 64, (p->lay.ncp_pad + 63) / 64), p->n_tiles + B_SLICES), dim3(64, p->reg_reduce_y), 0,
                         p->stream, p->tp, p->tile_wg_begin, p->partial, p->cam_off, p->cam_np, NC, ncp, p->Sacc, p->red, p->n_tiles,
                         (const double*)p->partial_b, p->grid, p->lay.ncp_pad, p->Sacc + (size_t)ncp * ncp);
      if (!fold_in_finalize)
        hipLaunchKernelGGL(k_reg_fold, dim3((p->gsz * NC * NC + 63) / 64, p->G), dim3(64), 0, p->stream, p->tp, p->red,
                           p->cam_off, p->cam_np, NC, ncp, p->Sacc);
      if (p->n_heavy)  // per-camera sums of the heavy points, one workgroup each (the pair plan skips them)
        hipLaunchKernelGGL((k_heavy_schur<NC>), dim3(p->n_heavy), dim3(BLOCK), (size_t)ncp * 3 * sizeof(double) + (size_t)ncp * sizeof(int), p->stream,
                           p->heavy_pts, p->pt_start, p->obs_cam, p->cam_off, p->cam_np, ncp, p->Trec, p->tab, p->heavy_W, p->Sacc);
      if (p->con.n_con) {  // Woodbury correction of S and b for the constraint rows, one workgroup per component
        if (p->con.small)
          hipLaunchKernelGGL((k_con_schur_small<NC>), dim3(p->con.n_comp), dim3(BLOCK), (size_t)p->con.small_lds, p->stream, p->con, p->lay, lam, p->V, p->g,
                             p->sinv, p->Trec, p->tab, p->pt_start, p->obs_cam, p->cam_off, p->cam_np, ncp, p->Sacc, p->Sacc + (size_t)ncp * ncp, p->flags);
        else
          hipLaunchKernelGGL((k_con_schur<NC>), dim3(p->con.n_comp), dim3(BLOCK), 0, p->stream, p->con, p->lay, lam, p->V, p->g, p->sinv,
                             p->Trec, p->tab, p->pt_start, p->obs_cam, p->cam_off, p->cam_np, ncp, p->Sacc, p->Sacc + (size_t)ncp * ncp, p->flags);
      }
    }
    if (p->sharded()) {  // reduced camera system: the one real exchange step — upper triangle and b, packed
      const size_t ntri = (size_t)ncp * (ncp + 1) / 2 + ncp;
      const int tg = (int)std::min<size_t>(((size_t)ncp * ncp + ncp + 255) / 256, 1024);
      hipLaunchKernelGGL(k_tri_pack, dim3(tg), dim3(256), 0, p->stream, p->Sacc, p->tri, ncp, 0, p->lay.ncp_pad);
      int rcs = allreduce_sum(p, p->tri, ntri);
      if (rcs) return rcs;
      hipLaunchKernelGGL(k_tri_pack, dim3(tg), dim3(256), 0, p->stream, p->Sacc, p->tri, ncp, 1, p->lay.ncp_pad);
    }
    const long nn = (long)ncp * ncp;
    small_solve = ncp <= SMALL_N && !p->sharded() && !p->chol_trace;
    if (!small_solve && !fused_finalize) {
      hipLaunchKernelGGL((k_schur_finalize<NC>), dim3((int)((nn + 255) / 256)), dim3(256), 0, p->stream, p->Sacc,
                         p->Sacc + (size_t)ncp * ncp, p->Upacked, p->g, p->sinv, p->param_cam, p->param_loc, ncp, lam, lam_dev, p->cam_diag, p->S, p->rhs, p->Lbuf, p->ldw,
                         fold_in_finalize ? (const double*)p->red : (const double*)nullptr, p->tp.g, (long)p->tp.tile_elems, (const int*)p->tp.group_cam_begin,
                         p->lay.ncp_pad);
    }
  }
  int rc = CBA_OK;
  if (small_solve) {  // small rigs: reduced system, factorisation and both substitutions in one workgroup (k_small_solve)
    RoctxRange r2("cba:cholesky");
    ScopedTimer t(p, T_CHOLESKY);
    hipLaunchKernelGGL((k_small_solve<NC>), dim3(1), dim3(SMALL_THREADS), kSmallSolveLds, p->stream, p->Sacc, p->Sacc + (size_t)ncp * ncp, p->Upacked, p->g, p->sinv,
                       p->param_cam, p->param_loc, ncp, lam, lam_dev, p->cam_diag, p->S, p->rhs, fold_in_finalize ? (const double*)p->red : (const double*)nullptr,
                       p->tp.g, (long)p->tp.tile_elems, (const int*)p->tp.group_cam_begin, p->flags, p->s, p->lay.ncp_pad);
  } else {
    RoctxRange r2("cba:cholesky");
    rc = run_cholesky(p);
  }
  if (rc) return rc;
  {
    RoctxRange r2("cba:backsub");
    ScopedTimer t(p, T_BACKSUB);
    if (p->n_heavy)  // fragments add sum_i W_i^T dc of a heavy point into s by atomics; k_heavy_finish solves for dp
      hipLaunchKernelGGL(k_zero_heavy, dim3((p->n_heavy + 63) / 64), dim3(64), 0, p->stream, p->heavy_pts, p->n_heavy, p->lay,
                         p->s + p->lay.ncp_pad, 3, p->s + p->lay.ncp_pad, 0);
    auto launch_backsub = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(p->grid_backsub), dim3(BLOCK), lds_backsub(p), p->stream, p->obs_u, p->obs_v, p->obs_cam,
                         p->obs_pt, p->pt_start, p->chunk_start, p->chunk_pts, p->n_chunks, p->x, p->lay, p->tab, p->cam_off, p->C, p->loss,
                         p->f_scale, lam, lam_dev, p->V, p->g, p->sinv, p->s, p->partial4);
    };
    auto launch_backsub_rec = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(p->grid_backsub), dim3(BLOCK), lds_backsub_rec<NC>(p), p->stream, (const double*)p->Trec, (const int*)p->obs_cam,
                         (const int*)p->pt_start, (const int*)p->chunk_start, (const int*)p->chunk_pts, p->n_chunks, p->lay, (const double*)p->tab,
                         (const int*)p->cam_off, (const int*)p->cam_np, p->C, lam, lam_dev, (const double*)p->V, (const double*)p->g,
                         (const double*)p->sinv, p->s, p->partial4);
    };
    if (p->backsub_rec) { if (fused_trial(p, compact)) launch_backsub_rec(k_backsub_rec<NC, true>); else launch_backsub_rec(k_backsub_rec<NC, false>); }
    else if (fused_trial(p, compact)) { if (p->tab_global) launch_backsub(k_backsub<NC, true, true>); else launch_backsub(k_backsub<NC, false, true>); }
    else if (p->tab_global) launch_backsub(k_backsub<NC, true>); else launch_backsub(k_backsub<NC>);
    if (p->n_heavy)
      hipLaunchKernelGGL(k_heavy_finish, dim3((p->n_heavy + 63) / 64), dim3(64), 0, p->stream, p->heavy_pts, p->heavy_frag, p->n_heavy, p->lay, lam,
                         p->V, p->g, p->sinv, p->s);
    if (p->con.n_con) {
      if (p->con.small)
        hipLaunchKernelGGL(k_con_backsub_small, dim3(p->con.n_comp), dim3(BLOCK), (size_t)(9 * p->con.max_np + 2 * p->con.max_m) * sizeof(double), p->stream,
                           p->con, p->lay, lam, p->V, p->sinv, p->s);
      else
        hipLaunchKernelGGL(k_con_backsub, dim3(p->con.n_comp), dim3(BLOCK), 0, p->stream, p->con, p->lay, lam, p->V, p->sinv, p->s);
    }
  }
  if (fused_trial(p, compact)) {
    // step scalars (the point block's share came out of k_backsub), subspace step, camera entries and camera table of the trial point: one workgroup
    ScopedTimer t(p, T_VECTOR);
    hipLaunchKernelGGL(k_step_cam, dim3(1), dim3(BLOCK), (size_t)p->lay.ncp_pad * sizeof(double), p->stream, (const double*)p->partial4, p->grid_backsub,
                       (const double*)p->x, (const double*)p->g, (const double*)p->sinv, (const double*)p->s, p->lay.ncp_pad, p->scal, (const int*)p->flags, p->fz,
                       p->x_new, p->partial4 + p->grid, p->tab_new, (const double*)p->cam_const, (const int*)p->cam_model, (const int*)p->cam_np,
                       (const int*)p->cam_off, p->C, p->bounds_on ? (const double*)p->lb_dev : (const double*)nullptr,
                       p->bounds_on ? (const double*)p->ub_dev : (const double*)nullptr, p->ncp);
    return CBA_OK;
  }
  return run_step_scalars(p, lam_dev != nullptr, compact);
}

static void read_newton(cba_problem* p, cba_newton_info* out) {
  out->ok = (p->h_flags[1] == 0 && p->h_flags[2] == 0) ? 1 : 0;
  out->p_sq = p->h_scal[16];
  out->gh_dot_p = p->h_scal[17];
  out->w_sq = p->h_scal[20];
  if (out->ok && !(std::isfinite(out->p_sq) && std::isfinite(out->gh_dot_p) && std::isfinite(out->w_sq))) out->ok = 0;
}

template <int NC>
static int run_newton(cba_problem* p, double lam, cba_newton_info* out) {
  int rc = run_newton_chain<NC>(p, lam, nullptr);
  if (rc) return rc;
  rc = sync_scalars(p, 24);
  if (rc) return rc;
  read_newton(p, out);
  return CBA_OK;
}

// One whole trust-region iteration behind a single host synchronisation: linearisation at the current x (the build is
// skipped when the accepted trial brought its own), damping and 2-D subspace step decided on the device
// (k_fused_lam / k_fused_subspace, the same code the host driver runs: csrc/trf_math.h), damped step, and the trial
// point evaluated by a full build pass into the second set of buffers, so that accepting it costs nothing more.
// everything an iteration enqueues, up to and including the speculative linearisation behind k_publish; *seq_out: what k_publish will answer with
// (compact), *waited: the non-compact route has already synchronised
template <int NC>
static int step_enqueue(cba_problem* p, double radius, bool compact, unsigned long long* seq_out) {
  int rc = run_lin_chain<NC>(p, true, compact, radius, compact && !p->det_m);  // (single rank: k_tprep finishes the linearisation's sums itself)
  if (rc) return rc;
  if (!compact) hipLaunchKernelGGL(k_fused_lam, dim3(1), dim3(1), 0, p->stream, p->scal, radius, p->fz);
  rc = run_newton_chain<NC>(p, 0.0, p->fz, compact);
  if (rc) return rc;
  if (!compact) hipLaunchKernelGGL(k_fused_subspace, dim3(1), dim3(1), 0, p->stream, p->scal, p->flags, p->fz);
  const long tot = p->lay.total();
  const int vg = vec_grid(tot);
  const bool ft = fused_trial(p, compact);
  if (!ft) {
    ScopedTimer t(p, T_VECTOR);
    // (the camera table of the trial point is prepared by workgroup 0 of the same launch)
    hipLaunchKernelGGL(k_trial_update, dim3(vg), dim3(BLOCK), 0, p->stream, p->x, p->g, p->sinv, p->s, 0.0, 0.0, tot, p->lay.ncp_pad,
                       p->rank == 0 ? 1 : 0, (const double*)nullptr, 0, (const double*)(p->fz + 2), p->x_new, p->partial4, p->tab_new,
                       (const double*)p->cam_const, (const int*)p->cam_model, (const int*)p->cam_np, (const int*)p->cam_off, p->C);
    if (!compact) hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, vg, 1, p->scal + 28);
  }
  // fused_trial: k_step_cam (end of the damped step) has placed the camera entries and the camera table of the trial point; the build forms the rest
  const TrialSrc tsrc{p->x, p->g, p->sinv, p->s, p->fz + 2, p->x_new, p->partial4};
  // compact: the iteration's packet (scalars, trial cost rows -> slot 24, step-norm rows -> slot 28: one per workgroup of the trial build + the camera
  // block's from k_step_cam, or k_trial_update's) leaves with the launch that reduces the trial build's camera blocks
  const unsigned long long seq = compact ? ++p->publish_seq : 0;
  const bool bnd = compact && p->bounds_on;
  const PubArgs pub{p->scal, 48, p->flags, p->d_hscal, p->d_hflags, seq, p->partial1, p->grid, 24, p->partial4, ft ? p->grid + 1 : vg, 28,
                    {p->x, p->g, p->sinv_state_c, p->s}, bnd ? p->d_hbcam : (double*)nullptr, p->ncp};
  rc = run_build_into<NC>(p, p->x_new, p->tab_new, p->V2, p->g2, p->U2, 24, p->scal + 42, true, compact, 0, ft ? &tsrc : nullptr,
                          compact ? &pub : nullptr);  // (the build is skipped when need_host; the reduction and the packet are not)
  if (rc) return rc;
  if (!compact) return CBA_OK;
  *seq_out = seq;
  // speculative linearisation of the trial point, enqueued BEHIND the publish: the device works on it while the host reads the packet,
  // decides and enqueues the next iteration (k_publish -> host -> first kernel of the next cba_step used to be an idle gap per iteration)
  p->spec_enqueued = false;
  {
    {
      ScopedTimer t(p, T_SCALE_SCALARS);
      const BoundArgs ba2 = bnd ? BoundArgs{p->lb_dev, p->ub_dev, p->sinv_state_c, p->sinv_state_c2, p->cam_diag2} : BoundArgs{};
      auto launch_sl2 = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(vg), dim3(BLOCK), 0, p->stream, p->U2, p->V2, p->param_cam, p->param_loc, p->lay, 0, p->sinv2,
                           (const double*)nullptr, p->x_new, p->g2, p->v1, p->partial4b, p->partial1, (const double*)p->sinv, ba2);
      };
      if (bnd) launch_sl2(k_scale_lin<NC, true>); else launch_sl2(k_scale_lin<NC, false>);
      int rows_jv = 0;
      rc = run_jv<NC>(p, 1, &rows_jv, p->x_new, p->tab_new);
      p->spec_rows_jv = rows_jv;
    }
    if (rc) return rc;
    p->spec_enqueued = true;
  }
  return CBA_OK;
}

template <int NC>
static int run_step(cba_problem* p, double radius, cba_step_info* out) {
  RoctxRange range("cba:step");  // one fused trust-region iteration
  // single rank: no exchange steps in between, so neighbouring small kernels are folded together (k_scale_lin, k_lin_finish,
  // k_step_finish, gradient written by k_reduce_rows, last sums taken by k_publish): 8 launches fewer per iteration
  const bool compact = !p->sharded();
  int rc;
  // (replaying the ~30 launches of a steady-state iteration from a hipGraph was measured no faster than enqueueing them — round 3: cfg4 0.693
  // against 0.690 ms, cfg2 0.150 against 0.157 — and cost 0.7 ms to record: removed in round 4)
  if (compact) {
    unsigned long long seq = 0;
    rc = step_enqueue<NC>(p, radius, true, &seq);
    if (rc) return rc;
    rc = publish_wait(p, seq, !p->spec_enqueued);
  } else {
    unsigned long long seq = 0;
    rc = step_enqueue<NC>(p, radius, false, &seq);
    if (rc) return rc;
    // one collective for the trial's camera blocks, its cost, the step norm and the flags
    rc = exchange_at(p, SLOT(24) | SLOT(28), false, p->U2, (size_t)p->C * UPack<NC>::STRIDE);
    if (rc) return rc;
    hipLaunchKernelGGL((k_unpack_camera_grad<NC>), dim3((p->C * NC + 255) / 256), dim3(256), 0, p->stream, p->U2, p->cam_off, p->cam_np, p->C, p->g2);
    rc = sync_scalars(p, 48);
  }
  if (rc) return rc;
  const int bad_residual = p->h_flags[0];
  read_linearization(p, &out->lin);
  read_newton(p, &out->newton);
  out->lam = p->h_scal[40]; out->radius = p->h_scal[41];
  out->need_host = (int)p->h_scal[42];  // 0; 1: failed factorisation / collinear step; 2: a bounded trial point that is not strictly inside its box
  out->p_s[0] = p->h_scal[43]; out->p_s[1] = p->h_scal[44]; out->predicted = p->h_scal[45];
  out->alpha = p->h_scal[46]; out->beta = p->h_scal[47];
  // test hook (tests/test_gpu_parity.py): CBA_TEST_FAIL_FUSED_STEP=k reports the k-th fused step of the process as one whose factorisation failed and
  // poisons the step's camera block in the packet, so that the driver's retry route (cba_solve.cpp: larger damping through cba_newton_step, then the
  // step's camera block fetched again) can be exercised — rounding on a gauge-singular problem is the only other way to get there
  if (const char* e = std::getenv("CBA_TEST_FAIL_FUSED_STEP")) {
    static std::atomic<long> n_fused{0};
    if (++n_fused == std::atol(e)) {
      out->newton.ok = 0; out->need_host = 1;
      if (p->bounds_on) for (int i = 0; i < p->ncp; ++i) p->h_bcam[(size_t)3 * p->ncp + i] = NAN;
    }
  }
  const double c = 0.5 * p->h_scal[24];
  out->trial.finite = (bad_residual == 0 && std::isfinite(c)) ? 1 : 0;
  out->trial.cost = out->trial.finite ? c : NAN;
  out->trial.step_norm = std::sqrt(p->h_scal[28]);
  out->trial.reserved = 0;
  p->trial_cost = c;
  return CBA_OK;
}

#define DISPATCH_NC(p, call6, call9) ((p)->nct == 9 ? (call9) : (call6))

extern "C" {

static int begin_common(cba_problem* p, double* cost_out, bool evaluate = true);

int cba_set_constraints(cba_problem* p, int32_t n_con, const int32_t* groups_a, const int32_t* groups_b, const double* distances,
                        const double* weights) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_set_constraints: null problem");
  if (p->begun) return fail(CBA_ERR_INVALID, "cba_set_constraints: call it before cba_begin");
  if (p->con.n_con) return fail(CBA_ERR_INVALID, "cba_set_constraints: constraints are already set");
  if (n_con <= 0) return CBA_OK;
  if (!groups_a || !groups_b || !distances || !weights) return fail(CBA_ERR_INVALID, "cba_set_constraints: null array");
  HIPCHK(hipSetDevice(p->device));
  const auto t_enter = std::chrono::steady_clock::now();
  auto lap_ms = [&](std::chrono::steady_clock::time_point since) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - since).count(); };
  const int P = p->P;
  for (long e = 0; e < (long)n_con * 4; ++e)
    if (groups_a[e] < 0 || groups_a[e] >= P || groups_b[e] < 0 || groups_b[e] >= P)
      return fail(CBA_ERR_INVALID, "cba_set_constraints: point index out of range in constraint %ld", e / 4);
  // connected components of the constraint graph (union-find over world points)
  std::vector<int> parent(P);
  for (int q = 0; q < P; ++q) parent[q] = q;
  auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
  for (int c = 0; c < n_con; ++c) {
    const int r0 = find(groups_a[4 * c]);
    for (int s = 0; s < 8; ++s) {
      const int q = (s < 4) ? groups_a[4 * c + s] : groups_b[4 * c + s - 4];
      const int r = find(q);
      if (r != r0) parent[r] = r0;
    }
  }
  std::vector<int> comp_of_root(P, -1), con_comp(n_con);
  int K = 0;
  for (int c = 0; c < n_con; ++c) {
    const int r = find(groups_a[4 * c]);
    if (comp_of_root[r] < 0) comp_of_root[r] = K++;
    con_comp[c] = comp_of_root[r];
  }
  std::vector<int> comp_con(K + 1, 0), order(n_con);
  for (int c = 0; c < n_con; ++c) comp_con[con_comp[c] + 1]++;
  for (int k = 0; k < K; ++k) comp_con[k + 1] += comp_con[k];
  {
    std::vector<int> cur(comp_con.begin(), comp_con.end() - 1);
    for (int c = 0; c < n_con; ++c) order[cur[con_comp[c]]++] = c;  // order[i] = caller's row of the i-th constraint here
  }
  std::vector<int> pt((size_t)n_con * 8), lp((size_t)n_con * 8), comp_pt(K + 1, 0), comp_pts;
  std::vector<double> dist(n_con), wgt(n_con);
  std::vector<long> comp_m(K + 1, 0);
  std::vector<int> local(P, -1);
  long max_m = 0;
  int max_pts = 0;  // beyond CON_LDS_POINTS the per-point factors of a component live in global scratch (ConPlan::big)
  for (int k = 0; k < K; ++k) {
    const int first = (int)comp_pts.size();
    for (int i = comp_con[k]; i < comp_con[k + 1]; ++i) {
      const int c = order[i];
      dist[i] = distances[c]; wgt[i] = weights[c];
      for (int s = 0; s < 8; ++s) {
        const int q = (s < 4) ? groups_a[4 * c + s] : groups_b[4 * c + s - 4];
        if (local[q] < 0) { local[q] = (int)comp_pts.size() - first; comp_pts.push_back(q); }
        pt[(size_t)i * 8 + s] = q; lp[(size_t)i * 8 + s] = local[q];
      }
    }
    for (size_t j = first; j < comp_pts.size(); ++j) local[comp_pts[j]] = -1;
    comp_pt[k + 1] = (int)comp_pts.size();
    const long m = comp_con[k + 1] - comp_con[k];
    comp_m[k + 1] = comp_m[k] + m * m;
    max_m = std::max(max_m, m);
    max_pts = std::max(max_pts, (int)comp_pts.size() - first);
  }
  // memory of the Woodbury correction: M (sum of m^2 over the components) and G (n_con x (ncp + 1)), doubles
  if (comp_m[K] > (1L << 28) || (long)n_con * (p->ncp + 1) > (1L << 29))
    return fail(CBA_ERR_UNSUPPORTED, "cba_set_constraints: the constraint rows need %.1f GB for the per-component matrices (sum of m^2 = %ld) and %.1f GB for their "
                "camera coupling (%d rows x %d camera parameters); the limits are 2 GB and 4 GB - use fewer rows per object and frame (DESIGN.md 2.2)",
                comp_m[K] * 8e-9, comp_m[K], (double)n_con * (p->ncp + 1) * 8e-9, n_con, p->ncp);
  int rc;
  const double ms_graph = lap_ms(t_enter);
  const auto t_up = std::chrono::steady_clock::now();
  upload_stage_acquire(p);
  int *dpt = nullptr, *dlp = nullptr, *dorder = nullptr, *dcc = nullptr, *dcp = nullptr, *dcps = nullptr;
  long* dcm = nullptr;
  double *ddist = nullptr, *dw = nullptr;
#define TRYC(e) do { rc = (e); if (rc) return rc; } while (0)
  TRYC(dev_upload(p, &dpt, pt)); TRYC(dev_upload(p, &dlp, lp)); TRYC(dev_upload(p, &dorder, order)); TRYC(dev_upload(p, &dcc, comp_con));
  TRYC(dev_upload(p, &dcp, comp_pt)); TRYC(dev_upload(p, &dcps, comp_pts)); TRYC(dev_upload(p, &dcm, comp_m));
  TRYC(dev_upload(p, &ddist, dist)); TRYC(dev_upload(p, &dw, wgt));
  ConPlan cp{};
  cp.n_con = n_con; cp.n_comp = K; cp.pt = dpt; cp.lp = dlp; cp.dist = ddist; cp.weight = dw; cp.order = dorder;
  cp.comp_con = dcc; cp.comp_pt = dcp; cp.comp_pts = dcps; cp.comp_m = dcm;
  cp.heavy_pts = p->heavy_pts; cp.heavy_W = p->heavy_W; cp.n_heavy = p->n_heavy;
  TRYC(dev_alloc(p, &cp.f, (size_t)n_con)); TRYC(dev_alloc(p, &cp.u, (size_t)n_con * 3)); TRYC(dev_alloc(p, &cp.z, (size_t)n_con * 24));
  TRYC(dev_alloc(p, &cp.M, (size_t)std::max<long>(comp_m[K], 1))); TRYC(dev_alloc(p, &cp.G, (size_t)n_con * (p->ncp + 1)));
  TRYC(dev_alloc(p, &cp.cdiag, (size_t)3 * p->lay.Ppad)); TRYC(dev_alloc(p, &cp.w, (size_t)n_con));
  if (max_pts > CON_LDS_POINTS) TRYC(dev_alloc(p, &cp.big, comp_pts.size() * 9));
#undef TRYC
  const double ms_upload = lap_ms(t_up);
  const auto t_attr = std::chrono::steady_clock::now();
  {
    // small components (one board in one frame: the reference's own sessions): dense blocks in LDS, k_con_schur_small / k_con_backsub_small
    cp.max_m = (int)max_m; cp.max_np = max_pts;
    const ConSmallLayout lo((int)max_m, max_pts, p->ncp);
    cp.small_lds = lo.total * (int)sizeof(double);
    const char* e = std::getenv("CBA_CON_SMALL");
    cp.small = (max_m <= CON_SMALL_M && !p->n_heavy && cp.small_lds <= 120 * 1024 && !(e && e[0] == '0')) ? 1 : 0;
    if (cp.small) {
      int rcl = (p->nct == 9) ? allow_lds(k_con_schur_small<9>, (size_t)cp.small_lds) : allow_lds(k_con_schur_small<6>, (size_t)cp.small_lds);
      if (!rcl) rcl = allow_lds(k_con_backsub_small, (size_t)(9 * max_pts + 2 * max_m) * sizeof(double));
      if (rcl) return rcl;
    }
  }
  const double ms_attr = lap_ms(t_attr);
  const auto t_sync = std::chrono::steady_clock::now();
  HIPCHK(hipMemsetAsync(cp.cdiag, 0, (size_t)3 * p->lay.Ppad * sizeof(double), p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));  // (the staged uploads: the caller's arrays and the pinned buffer are free again)
  upload_stage_release(p);
  if (plan_timing_on())
    fprintf(stderr, "  cba_set_constraints: %d rows, %d components: graph %.3f ms, uploads + allocations %.3f, kernel attributes %.3f, memset + sync %.3f\n", n_con, K,
            ms_graph, ms_upload, ms_attr, lap_ms(t_sync));
  p->con = cp;
  p->con_grid = std::max(1, std::min((n_con + BLOCK - 1) / BLOCK, 1024));
  return CBA_OK;
}

int cba_begin(cba_problem* p, const double* x0, double* cost_out) {
  if (!p || !x0 || !cost_out) return fail(CBA_ERR_INVALID, "cba_begin: null argument");
  HIPCHK(hipSetDevice(p->device));
  { const int rcw = stage_wait(p); if (rcw) return rcw; }
  pack_host(p, x0, p->h_vec, 0.0);
  HIPCHK(hipMemcpyAsync(p->x0, p->h_vec, p->lay.total() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  { const int rcs = stage_sent(p); if (rcs) return rcs; }
  p->have_x0 = true;
  return begin_common(p, cost_out);
}

int cba_restart(cba_problem* p, double* cost_out) {
  if (!p || !cost_out) return fail(CBA_ERR_INVALID, "cba_restart: null argument");
  if (!p->have_x0) return fail(CBA_ERR_INVALID, "cba_restart: call cba_begin first");
  HIPCHK(hipSetDevice(p->device));
  return begin_common(p, cost_out);
}

int cba_begin_deferred(cba_problem* p, const double* x0) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_begin_deferred: null argument");
  if (!x0 && !p->have_x0) return fail(CBA_ERR_INVALID, "cba_begin_deferred: no x0 on the device yet");
  HIPCHK(hipSetDevice(p->device));
  if (x0) {
    { const int rcw = stage_wait(p); if (rcw) return rcw; }
    pack_host(p, x0, p->h_vec, 0.0);
    HIPCHK(hipMemcpyAsync(p->x0, p->h_vec, p->lay.total() * sizeof(double), hipMemcpyHostToDevice, p->stream));
    { const int rcs = stage_sent(p); if (rcs) return rcs; }
    p->have_x0 = true;
  }
  return begin_common(p, nullptr, false);
}

static int begin_common(cba_problem* p, double* cost_out, bool evaluate) {
  // x <- x0, the trial buffer too where the fused trial build is used (it writes the points of its super-chunks only: points no observation refers to
  // keep their x0 entries in BOTH buffers), scale 1, bound scaling and flags cleared, the camera table of x0: one launch (k_begin)
  p->cam_scaled = false; p->cam_state_saved = false;
  p->have_build = false; p->trial_built = false; p->cost_pending = false; p->lf_pending = false;
  {
    ScopedTimer t(p, T_CAM_PREP);
    double* trial_buf = (!p->eval_only && fused_trial(p, true)) ? p->x_new : nullptr;
    hipLaunchKernelGGL(k_begin, dim3(std::max(vec_grid(p->lay.total()), (p->C + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, p->stream, (const double*)p->x0, p->x, trial_buf,
                       p->sinv, p->cam_diag, p->flags, p->lay.total(), p->lay.ncp_pad, (const double*)p->cam_const, (const int*)p->cam_model, (const int*)p->cam_np,
                       (const int*)p->cam_off, p->C, p->tab);
  }
  p->first_scale = true;
  p->begun = true; p->linearized = false; p->stepped = false; p->have_trial = false;
  if (!evaluate) {  // the first linearisation (cba_step / cba_linearize) evaluates x0 with its build pass: no pass, no wait here
    p->cost_x = NAN;
    return CBA_OK;
  }
  int rc = launch_cost(p, p->x, p->tab, 24, nullptr);
  if (rc) return rc;
  rc = exchange(p, SLOT(24), false);
  if (rc) return rc;
  rc = sync_scalars(p, 32);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  *cost_out = p->h_flags[0] ? NAN : 0.5 * p->h_scal[24];
  p->cost_x = *cost_out;
  return CBA_OK;
}

int cba_linearize(cba_problem* p, cba_linearization* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_linearize: null argument");
  if (p->eval_only) return fail(CBA_ERR_INVALID, "cba_linearize: the problem was created with evaluation_only");
  if (!p->begun) return fail(CBA_ERR_INVALID, "cba_linearize: call cba_begin first");
  HIPCHK(hipSetDevice(p->device));
  int rc = DISPATCH_NC(p, run_linearize<6>(p, out), run_linearize<9>(p, out));
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  p->linearized = true; p->stepped = false;
  return CBA_OK;
}

int cba_linearize_build(cba_problem* p) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_linearize_build: null argument");
  if (!p->begun) return fail(CBA_ERR_INVALID, "cba_linearize_build: call cba_begin first");
  if (p->eval_only) return fail(CBA_ERR_INVALID, "cba_linearize_build: the problem was created with evaluation_only");
  HIPCHK(hipSetDevice(p->device));
  int rc = DISPATCH_NC(p, run_lin_chain<6>(p, false), run_lin_chain<9>(p, false));
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  p->linearized = true; p->stepped = false;
  return CBA_OK;
}

int cba_newton_step(cba_problem* p, double lam, cba_newton_info* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_newton_step: null argument");
  if (!p->linearized) return fail(CBA_ERR_INVALID, "cba_newton_step: call cba_linearize first");
  if (!(lam >= 0.0) || !std::isfinite(lam)) return fail(CBA_ERR_INVALID, "cba_newton_step: lam must be finite and >= 0");
  HIPCHK(hipSetDevice(p->device));
  int rc = DISPATCH_NC(p, run_newton<6>(p, lam, out), run_newton<9>(p, lam, out));
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  p->stepped = true;
  return CBA_OK;
}

int cba_step(cba_problem* p, double radius, cba_step_info* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_step: null argument");
  if (!p->begun) return fail(CBA_ERR_INVALID, "cba_step: call cba_begin first");
  if (!cba_step_supported(p)) return fail(CBA_ERR_UNSUPPORTED, "cba_step: not available for this problem (constraint rows, heavy points, or bound scaling without cba_set_bounds): use the primitives");
  HIPCHK(hipSetDevice(p->device));
  int rc = DISPATCH_NC(p, run_step<6>(p, radius, out), run_step<9>(p, radius, out));
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  p->linearized = true; p->stepped = true;
  p->have_trial = out->need_host == 0;
  p->trial_built = p->have_trial;
  return CBA_OK;
}

int cba_set_bounds(cba_problem* p, const double* lb, const double* ub) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_set_bounds: null argument");
  if (!lb || !ub) { p->bounds_on = false; return 0; }
  // the fused bounded iteration exists for the single-rank route over camera-sorted super-chunks (what every handle without constraint rows, heavy
  // points or fixed-order sums runs); everything else keeps scipy's bounded loop on the primitives
  if (p->eval_only || p->sharded() || !fused_trial(p, true)) { p->bounds_on = false; return 0; }
  HIPCHK(hipSetDevice(p->device));
  HIPCHK(hipMemcpyAsync(p->lb_dev, lb, (size_t)p->ncp * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIPCHK(hipMemcpyAsync(p->ub_dev, ub, (size_t)p->ncp * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));  // (the caller's arrays may go away)
  p->bounds_on = true;
  return 1;
}

int cba_step_camera_state(cba_problem* p, double* x_c, double* g_c, double* scale_inv_c, double* step_c) {
  if (!p || !x_c || !g_c || !scale_inv_c || !step_c) return fail(CBA_ERR_INVALID, "cba_step_camera_state: null argument");
  if (!p->bounds_on || !p->stepped) return fail(CBA_ERR_INVALID, "cba_step_camera_state: no bounded cba_step to report on");
  const size_t n = (size_t)p->ncp;
  std::memcpy(x_c, p->h_bcam, n * sizeof(double)); std::memcpy(g_c, p->h_bcam + n, n * sizeof(double));
  std::memcpy(scale_inv_c, p->h_bcam + 2 * n, n * sizeof(double)); std::memcpy(step_c, p->h_bcam + 3 * n, n * sizeof(double));
  return CBA_OK;
}

int cba_refresh_step_scalars(cba_problem* p, cba_newton_info* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_refresh_step_scalars: null argument");
  if (!p->stepped) return fail(CBA_ERR_INVALID, "cba_refresh_step_scalars: no damped step to measure");
  HIPCHK(hipSetDevice(p->device));
  const int f1 = p->h_flags[1], f2 = p->h_flags[2];  // the factorisation flags of the step were published (and cleared) already
  int rc = run_step_scalars(p, false);
  if (!rc) rc = sync_scalars(p, 24);
  if (rc) return rc;
  p->h_flags[1] = f1; p->h_flags[2] = f2;
  read_newton(p, out);
  return CBA_OK;
}

int cba_step_supported(cba_problem* p) {
  return (p && !p->eval_only && !p->con.n_con && !p->n_heavy && (!p->cam_scaled || (p->bounds_on && !p->sharded())) && !p->peer_needs_primitives) ? 1 : 0;
}

// camera-block override of a device vector: `host` [ncp] -> dev [ncp_pad] (padding stays zero)
static int upload_cam(cba_problem* p, const double* host, double* dev) {
  HIPCHK(hipMemcpyAsync(dev, host, (size_t)p->ncp * sizeof(double), hipMemcpyHostToDevice, p->stream));
  return CBA_OK;
}

int cba_subspace_gram_ex(cba_problem* p, double a1, double b1, const double* cam1, double a2, double b2, const double* cam2,
                         double* gram_out) {
  if (!p || !gram_out) return fail(CBA_ERR_INVALID, "cba_subspace_gram: null argument");
  if (!p->stepped) return fail(CBA_ERR_INVALID, "cba_subspace_gram: call cba_newton_step first");
  HIPCHK(hipSetDevice(p->device));
  const long tot = p->lay.total();
  if (cam1) { int rc = upload_cam(p, cam1, p->cam_over1); if (rc) return rc; }
  if (cam2) { int rc = upload_cam(p, cam2, p->cam_over2); if (rc) return rc; }
  {
    ScopedTimer t(p, T_VECTOR);
    hipLaunchKernelGGL(k_combine, dim3(vec_grid(tot)), dim3(BLOCK), 0, p->stream, p->g, p->sinv, p->s, a1, b1, tot,
                       (const double*)p->cam_over1, cam1 ? p->lay.ncp_pad : 0, p->v1);
    hipLaunchKernelGGL(k_combine, dim3(vec_grid(tot)), dim3(BLOCK), 0, p->stream, p->g, p->sinv, p->s, a2, b2, tot,
                       (const double*)p->cam_over2, cam2 ? p->lay.ncp_pad : 0, p->v2);
  }
  DISPATCH_NC(p, run_jv<6>(p, 2), run_jv<9>(p, 2));
  int rc = exchange(p, SLOT(12) | SLOT(13) | SLOT(14) | SLOT(15), false);
  if (!rc) rc = sync_scalars(p, 16);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  gram_out[0] = p->h_scal[12]; gram_out[1] = p->h_scal[13]; gram_out[2] = p->h_scal[14];
  return CBA_OK;
}

int cba_subspace_gram(cba_problem* p, double a1, double b1, double a2, double b2, double* gram_out) {
  return cba_subspace_gram_ex(p, a1, b1, nullptr, a2, b2, nullptr, gram_out);
}

int cba_set_camera_scaling(cba_problem* p, const double* mult, const double* diag_h, cba_linearization* out) {
  if (!p || !mult || !diag_h || !out) return fail(CBA_ERR_INVALID, "cba_set_camera_scaling: null argument");
  if (!p->linearized) return fail(CBA_ERR_INVALID, "cba_set_camera_scaling: call cba_linearize first");
  for (int i = 0; i < p->ncp; ++i)
    if (!(mult[i] > 0.0) || !std::isfinite(mult[i]) || !(diag_h[i] >= 0.0) || !std::isfinite(diag_h[i]))
      return fail(CBA_ERR_INVALID, "cba_set_camera_scaling: entry %d: mult must be positive and diag_h non-negative (finite)", i);
  HIPCHK(hipSetDevice(p->device));
  const int npad = p->lay.ncp_pad;
  if (!p->cam_state_saved) {  // first call after this linearisation: sinv still holds the Jacobi scale
    HIPCHK(hipMemcpyAsync(p->sinv_state_c, p->sinv, (size_t)npad * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
    p->cam_state_saved = true;
  }
  int rc = upload_cam(p, mult, p->cam_over1);
  if (!rc) rc = upload_cam(p, diag_h, p->cam_over2);
  if (rc) return rc;
  const long tot = p->lay.total();
  const int vg = vec_grid(tot);
  {
    ScopedTimer t(p, T_SCALE_SCALARS);
    hipLaunchKernelGGL(k_cam_rescale, dim3((p->ncp + 255) / 256), dim3(256), 0, p->stream, p->sinv_state_c, p->cam_over1, p->cam_over2, p->ncp,
                       p->sinv, p->cam_diag);
    hipLaunchKernelGGL(k_lin_scalars, dim3(vg), dim3(BLOCK), 0, p->stream, p->x, p->g, p->sinv, tot, npad, p->rank == 0 ? 1 : 0, npad,
                       p->v1, p->partial4, p->partial1);
    hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, vg, 4, p->scal + 0);
    hipLaunchKernelGGL(k_reduce_narrow<true>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial1, vg, 1, p->scal + 4);
  }
  rc = DISPATCH_NC(p, run_jv<6>(p, 1), run_jv<9>(p, 1));
  if (rc) return rc;
  rc = exchange(p, SLOT(0) | SLOT(1) | SLOT(2) | SLOT(3) | SLOT(12) | SLOT(13) | SLOT(14) | SLOT(15), true);
  if (!rc) rc = sync_scalars(p, 16);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  p->cam_scaled = true; p->stepped = false;
  read_linearization(p, out);  // g_norm_inf: point block only
  return CBA_OK;
}

int cba_trial(cba_problem* p, double alpha, double beta, cba_trial_info* out) { return cba_trial_ex(p, alpha, beta, nullptr, out); }

int cba_trial_ex(cba_problem* p, double alpha, double beta, const double* cam_x_new, cba_trial_info* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_trial: null argument");
  if (!p->stepped) return fail(CBA_ERR_INVALID, "cba_trial: call cba_newton_step first");
  HIPCHK(hipSetDevice(p->device));
  const long tot = p->lay.total();
  const int vg = vec_grid(tot);
  if (cam_x_new) { int rcu = upload_cam(p, cam_x_new, p->cam_over1); if (rcu) return rcu; }
  {
    ScopedTimer t(p, T_VECTOR);
    hipLaunchKernelGGL(k_trial_update, dim3(vg), dim3(BLOCK), 0, p->stream, p->x, p->g, p->sinv, p->s, alpha, beta, tot, p->lay.ncp_pad,
                       p->rank == 0 ? 1 : 0, (const double*)p->cam_over1, cam_x_new ? p->ncp : 0, (const double*)nullptr, p->x_new, p->partial4);
    if (p->sharded()) hipLaunchKernelGGL(k_reduce_narrow<false>, dim3(1), dim3(BLOCK), 0, p->stream, p->partial4, vg, 1, p->scal + 28);
  }
  launch_cam_prep(p, p->x_new, p->tab_new);
  int rc;
  if (p->sharded()) {
    rc = launch_cost(p, p->x_new, p->tab_new, 24, nullptr);
    if (rc) return rc;
    rc = exchange(p, SLOT(24) | SLOT(28), false);  // trial cost, ||step||^2, flags
    if (rc) return rc;
    rc = sync_scalars(p, 32);
  } else {  // single rank: the cost rows and the step-norm rows are summed by k_publish
    int cost_rows = 0;
    rc = launch_cost(p, p->x_new, p->tab_new, 24, nullptr, &cost_rows);
    if (rc) return rc;
    rc = sync_scalars(p, 32, p->partial1, cost_rows, 24, p->partial4, vg, 28);
  }
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  const double c = 0.5 * p->h_scal[24];
  out->finite = (p->h_flags[0] == 0 && std::isfinite(c)) ? 1 : 0;
  out->cost = out->finite ? c : NAN;
  out->step_norm = std::sqrt(p->h_scal[28]);
  p->have_trial = true; p->trial_built = false; p->trial_cost = c;
  return CBA_OK;
}

int cba_accept(cba_problem* p) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_accept: null argument");
  if (!p->have_trial) return fail(CBA_ERR_INVALID, "cba_accept: no trial point");
  std::swap(p->x, p->x_new);
  std::swap(p->tab, p->tab_new);
  p->spec_valid = false;
  if (p->trial_built) {  // the trial point came with its own build (cba_step): it becomes the linearisation point as it is
    std::swap(p->V, p->V2); std::swap(p->g, p->g2); std::swap(p->Upacked, p->U2);
    p->cost_x = p->trial_cost;
    if (p->spec_enqueued) {  // ... and so does its speculative linearisation
      std::swap(p->sinv, p->sinv2); p->spec_valid = true;
      if (p->bounds_on) { std::swap(p->sinv_state_c, p->sinv_state_c2); std::swap(p->cam_diag, p->cam_diag2); }
    }
  }
  p->spec_enqueued = false;
  p->have_build = p->trial_built;
  p->trial_built = false;
  p->have_trial = false; p->linearized = false; p->stepped = false;
  return CBA_OK;
}

int cba_comm_unique_id(char* out128) {
  if (!out128) return fail(CBA_ERR_INVALID, "cba_comm_unique_id: null argument");
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId larger than 128 bytes");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  std::memset(out128, 0, 128);
  std::memcpy(out128, &id, sizeof(id));
  return CBA_OK;
}

int cba_comm_init(cba_problem* p, const char* id128, int32_t rank, int32_t world) {
  if (!p || !id128) return fail(CBA_ERR_INVALID, "cba_comm_init: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(CBA_ERR_INVALID, "cba_comm_init: rank %d of %d", rank, world);
  if (p->comm.load()) return fail(CBA_ERR_INVALID, "cba_comm_init: communicator already initialised");
  if (p->comm_aborted.load()) return fail(CBA_ERR_COMM, "cba_comm_init: a peer rank failed before the communicator was created");
  HIPCHK(hipSetDevice(p->device));
  p->rank = rank; p->world = world;
  // A one-rank communicator is pointless in production but exercises every RCCL call site on a single GPU
  // (tests set CBA_FORCE_COMM=1); without it world == 1 stays collective-free.
  const char* force = std::getenv("CBA_FORCE_COMM");
  const bool forced = force && force[0] == '1';
  if (world == 1 && !forced) return CBA_OK;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  {
    ncclComm_t c = nullptr;
    NCCLCHK(ncclCommInitRank(&c, world, id, rank));
    p->comm.store(c);
  }
  // cba_step issues one collective more than the primitives (the camera blocks of the trial build): every rank must
  // take the same route.  A rank whose shard needs the primitives (constraint rows, heavy points)
  // switches the fused route off for all of them.
  {
    const double mine = cba_step_supported(p) ? 0.0 : 1.0;
    HIPCHK(hipMemcpyAsync(p->xbuf, &mine, sizeof(double), hipMemcpyHostToDevice, p->stream));
    NCCLCHK(ncclAllReduce(p->xbuf, p->xbuf, 1, ncclDouble, ncclSum, p->comm.load(), p->stream));
    double total = 0.0;
    HIPCHK(hipMemcpyAsync(&total, p->xbuf, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->peer_needs_primitives = total > 0.0;
  }
  return CBA_OK;
}

int cba_comm_abort(cba_problem* p) {
  // no api guard: this is called from another thread while the owner may be blocked inside a collective
  if (!p) return fail(CBA_ERR_INVALID, "cba_comm_abort: null problem");
  // Sticky: from here on every collective of the handle fails with CBA_ERR_COMM (the owner, if it was not blocked inside one, must not carry on
  // as a single-rank solve of its shard), and the communicator is taken out atomically so that cba_destroy does not destroy what was aborted.
  p->comm_aborted.store(true, std::memory_order_release);
  if (ncclComm_t c = p->comm.exchange(nullptr)) (void)ncclCommAbort(c);
  return CBA_OK;
}

int cba_group_create(int32_t world, cba_group** out) {
  if (!out) return fail(CBA_ERR_INVALID, "cba_group_create: null argument");
  *out = nullptr;
  if (world < 1 || world > GROUP_MAX) return fail(CBA_ERR_INVALID, "cba_group_create: world %d (1..%d)", world, GROUP_MAX);
  cba_group* g = new cba_group();
  g->world = world;
  if (const char* e = std::getenv("CBA_GROUP_TIMEOUT_S")) g->timeout_s = std::max(1, std::atoi(e));
  g->member.assign(world, nullptr);
  for (int par = 0; par < 2; ++par) {
    g->stage[par].assign(world, nullptr);
    g->ready[par].assign(world, nullptr);
    g->done[par].assign(world, nullptr);
  }
  *out = g;
  return CBA_OK;
}

// Called concurrently by the `world` member threads (one per handle); returns when every rank has joined.
int cba_group_join(cba_problem* p, cba_group* g, int32_t rank) {
  if (!p || !g) return fail(CBA_ERR_INVALID, "cba_group_join: null argument");
  if (rank < 0 || rank >= g->world) return fail(CBA_ERR_INVALID, "cba_group_join: rank %d of %d", rank, g->world);
  if (p->comm.load() || p->group) return fail(CBA_ERR_INVALID, "cba_group_join: the handle already belongs to a communicator");
  int rc = CBA_OK;
  {
    if (hipSetDevice(p->device) != hipSuccess) rc = fail(CBA_ERR_HIP, "hipSetDevice(%d) failed", p->device);
    // staging: the largest payload is the reduced camera system with its right-hand side, or the camera blocks with the
    // packed scalars behind them
    const size_t ustride = (p->nct == 9) ? UPack<9>::STRIDE : UPack<6>::STRIDE;
    const size_t cap = std::max<size_t>((size_t)p->ncp * p->ncp + p->lay.ncp_pad, (size_t)p->C * ustride + 128) + 128;
    for (int par = 0; par < 2 && !rc; ++par) {
      if (hipMalloc((void**)&g->stage[par][rank], cap * sizeof(double)) != hipSuccess) { rc = fail(CBA_ERR_HIP, "staging allocation failed"); break; }
      if (hipEventCreateWithFlags(&g->ready[par][rank], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&g->done[par][rank], hipEventDisableTiming) != hipSuccess) { rc = fail(CBA_ERR_HIP, "hipEventCreate failed"); break; }
      // a first record, so that the waits of the first two collectives find completed events
      (void)hipEventRecord(g->ready[par][rank], p->stream);
      (void)hipEventRecord(g->done[par][rank], p->stream);
    }
    if (rank == 0) g->capacity = cap;
    g->member[rank] = p;
  }
  if (rc) g->failed.store(1);
  if (!g->barrier()) return fail(CBA_ERR_INVALID, "cba_group_join: the group was aborted (another rank failed before joining)");  // every rank has allocated and published its staging buffers
  if (g->failed.load()) return rc ? rc : fail(CBA_ERR_HIP, "cba_group_join: another rank failed to join");
  {
    for (int q = 0; q < g->world && !rc; ++q) {
      const int dq = g->member[q]->device;
      if (dq == p->device) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, p->device, dq) != hipSuccess || !can) { rc = fail(CBA_ERR_HIP, "device %d cannot access device %d", p->device, dq); break; }
      const hipError_t e = hipDeviceEnablePeerAccess(dq, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { rc = fail(CBA_ERR_HIP, "hipDeviceEnablePeerAccess(%d) failed: %s", dq, hipGetErrorString(e)); break; }
      (void)hipGetLastError();
    }
  }
  if (rc) g->failed.store(1);
  if (!g->barrier()) return fail(CBA_ERR_INVALID, "cba_group_join: the group was aborted");  // peer access settled on every rank
  if (g->failed.load()) return rc ? rc : fail(CBA_ERR_HIP, "cba_group_join: another rank failed to join");
  p->group = g; p->rank = rank; p->world = g->world; p->group_generation = 0;
  // every rank must take the same route through an iteration (see cba_comm_init)
  {
    const double mine = cba_step_supported(p) ? 0.0 : 1.0;
    double total = 0.0;
    HIPCHK(hipMemcpyAsync(p->xbuf, &mine, sizeof(double), hipMemcpyHostToDevice, p->stream));
    int rca = allreduce_sum(p, p->xbuf, 1);
    if (rca) return rca;
    HIPCHK(hipMemcpyAsync(&total, p->xbuf, sizeof(double), hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->peer_needs_primitives = total > 0.0;
  }
  if (p->world == 1) { p->group = nullptr; }  // nothing to exchange; keep the single-rank fast paths
  return CBA_OK;
}

// A member that failed outside the library releases the ranks spinning in a group barrier; their calls return an error.
void cba_group_abort(cba_group* g) { if (g) g->aborted.store(1, std::memory_order_release); }

// After the member handles are destroyed (or at least idle).
void cba_group_destroy(cba_group* g) {
  if (!g) return;
  for (int par = 0; par < 2; ++par)
    for (int q = 0; q < g->world; ++q) {
      if (g->stage[par][q]) (void)hipFree(g->stage[par][q]);
      if (g->ready[par][q]) (void)hipEventDestroy(g->ready[par][q]);
      if (g->done[par][q]) (void)hipEventDestroy(g->done[par][q]);
    }
  delete g;
}

// camera blocks (first n_cam_params entries) of up to three device vectors in one kernel and one wait
static int fetch_camera_blocks(cba_problem* p, const double* const* srcs, double* const* outs) {
  const int ncp = p->ncp;
  unsigned long long* hseq = reinterpret_cast<unsigned long long*>(p->h_cam + (size_t)3 * ncp);
  unsigned long long* dseq = reinterpret_cast<unsigned long long*>(p->d_hcam + (size_t)3 * ncp);
  const unsigned long long seq = ++p->publish_seq;
  hipLaunchKernelGGL(k_publish_cam, dim3(1), dim3(BLOCK), 0, p->stream, srcs[0], srcs[1], srcs[2], ncp, p->d_hcam, dseq, seq);
  if (p->spin_wait) {
    volatile unsigned long long* flag = hseq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *flag != seq; ++spins)
      if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
        HIPCHK(hipStreamSynchronize(p->stream));
        if (*flag != seq) return fail(CBA_ERR_HIP, "k_publish_cam did not complete");
        break;
      }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    HIPCHK(hipStreamSynchronize(p->stream));
  }
  for (int v = 0; v < 3; ++v)
    if (outs[v]) std::memcpy(outs[v], p->h_cam + (size_t)v * ncp, (size_t)ncp * sizeof(double));
  return CBA_OK;
}

int cba_get_vector(cba_problem* p, int32_t which, double* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_get_vector: null argument");
  HIPCHK(hipSetDevice(p->device));
  const double* src = nullptr;
  switch (which) {
    case CBA_VEC_X: src = p->x; break;
    case CBA_VEC_X_NEW: src = p->x_new; break;
    case CBA_VEC_GRAD: src = p->g; break;
    case CBA_VEC_STEP: src = p->s; break;
    case CBA_VEC_SCALE_INV: src = p->sinv; break;
    default: return fail(CBA_ERR_INVALID, "cba_get_vector: unknown vector %d", which);
  }
  HIPCHK(hipStreamSynchronize(p->stream));
  { const int rcw = stage_wait(p); if (rcw) return rcw; }
  HIPCHK(hipMemcpy(p->h_vec, src, p->lay.total() * sizeof(double), hipMemcpyDeviceToHost));
  unpack_host(p, p->h_vec, out);
  return CBA_OK;
}

int cba_get_camera_params(cba_problem* p, int32_t which, double* out) {
  if (!p || !out) return fail(CBA_ERR_INVALID, "cba_get_camera_params: null argument");
  HIPCHK(hipSetDevice(p->device));
  const double* src = nullptr;
  switch (which) {
    case CBA_VEC_X: src = p->x; break;
    case CBA_VEC_X_NEW: src = p->x_new; break;
    case CBA_VEC_GRAD: src = p->g; break;
    case CBA_VEC_STEP: src = p->s; break;
    case CBA_VEC_SCALE_INV: src = p->sinv; break;
    default: return fail(CBA_ERR_INVALID, "cba_get_camera_params: unknown vector %d", which);
  }
  const double* srcs[3] = {src, nullptr, nullptr};
  double* outs[3] = {out, nullptr, nullptr};
  return fetch_camera_blocks(p, srcs, outs);
}

int cba_get_camera_state(cba_problem* p, double* x_c, double* g_c, double* scale_inv_c) {
  if (!p || !x_c || !g_c || !scale_inv_c) return fail(CBA_ERR_INVALID, "cba_get_camera_state: null argument");
  HIPCHK(hipSetDevice(p->device));
  const double* srcs[3] = {p->x, p->g, p->sinv};
  double* outs[3] = {x_c, g_c, scale_inv_c};
  return fetch_camera_blocks(p, srcs, outs);
}

int cba_residuals(cba_problem* p, const double* x, double* r_out, double* cost_out) {
  if (!p || !x || !r_out) return fail(CBA_ERR_INVALID, "cba_residuals: null argument");
  HIPCHK(hipSetDevice(p->device));
  // scratch: v2 holds the vector, tab_new the camera table (both are dead between solver calls)
  double* d_r = nullptr;
  const size_t n_rows = (size_t)2 * p->N + (size_t)p->con.n_con;  // reprojection rows, then the constraint rows
  HIPCHK(hipMalloc((void**)&d_r, n_rows * sizeof(double)));
  { const int rcw = stage_wait(p); if (rcw) { (void)hipFree(d_r); return rcw; } }
  pack_host(p, x, p->h_vec, 0.0);
  p->have_trial = false; p->trial_built = false;  // the evaluation borrows the trial point's camera table: a pending trial is gone
  hipError_t e = hipMemcpyAsync(p->v2, p->h_vec, p->lay.total() * sizeof(double), hipMemcpyHostToDevice, p->stream);
  if (e == hipSuccess) { e = hipEventRecord(p->h_vec_sent, p->stream); p->h_vec_in_flight = (e == hipSuccess); }
  if (e == hipSuccess) {
    launch_cam_prep(p, p->v2, p->tab_new);
    launch_cost(p, p->v2, p->tab_new, 24, d_r);
    e = hipMemcpyAsync(r_out, d_r, n_rows * sizeof(double), hipMemcpyDeviceToHost, p->stream);
  }
  int rc = (e == hipSuccess) ? exchange(p, SLOT(24), false) : fail(CBA_ERR_HIP, "cba_residuals: %s", hipGetErrorString(e));
  if (!rc) rc = sync_scalars(p, 32);
  (void)hipFree(d_r);
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  p->have_trial = false;  // tab_new was overwritten
  if (cost_out) *cost_out = 0.5 * p->h_scal[24];
  return CBA_OK;
}

int cba_normal_blocks(cba_problem* p, const double* x, double* U, double* V, double* gc, double* gp) {
  if (!p || !x) return fail(CBA_ERR_INVALID, "cba_normal_blocks: null argument");
  HIPCHK(hipSetDevice(p->device));
  if (p->eval_only) return fail(CBA_ERR_INVALID, "cba_normal_blocks: the problem was created with evaluation_only");
  // Runs the real build pass on x: swap it in as the current point, then restore.
  p->have_build = false;  // V, g, Upacked are overwritten below
  double cost;
  std::vector<double> saved((size_t)p->lay.total());
  HIPCHK(hipStreamSynchronize(p->stream));
  HIPCHK(hipMemcpy(saved.data(), p->x, saved.size() * sizeof(double), hipMemcpyDeviceToHost));
  const bool was_begun = p->begun;
  const bool first = p->first_scale;
  { const int rcw = stage_wait(p); if (rcw) return rcw; }
  pack_host(p, x, p->h_vec, 0.0);
  HIPCHK(hipMemcpy(p->x, p->h_vec, p->lay.total() * sizeof(double), hipMemcpyHostToDevice));
  launch_cam_prep(p, p->x, p->tab);
  DISPATCH_NC(p, run_build<6>(p), run_build<9>(p));
  HIPCHK(hipMemsetAsync(p->flags, 0, 4 * sizeof(int), p->stream));  // parity hook: no publish here, leave no flag behind
  HIPCHK(hipStreamSynchronize(p->stream));
  HIPCHK(hipGetLastError());
  (void)cost;
  const int ustride = (p->nct == 9) ? UPack<9>::STRIDE : UPack<6>::STRIDE;
  const int tri = (p->nct == 9) ? UPack<9>::TRI : UPack<6>::TRI;
  std::vector<double> hU((size_t)p->C * ustride);
  HIPCHK(hipMemcpy(hU.data(), p->Upacked, hU.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (U) {
    std::fill(U, U + (size_t)p->C * 81, 0.0);
    for (int c = 0; c < p->C; ++c) {
      const int np = p->h_cam_np[c];
      for (int r = 0; r < np; ++r)
        for (int k = r; k < np; ++k) {
          const int idx = r * p->nct - r * (r - 1) / 2 + (k - r);
          const double v = hU[(size_t)c * ustride + idx];
          U[(size_t)c * 81 + r * 9 + k] = v;
          U[(size_t)c * 81 + k * 9 + r] = v;
        }
    }
  }
  if (gc)
    for (int c = 0; c < p->C; ++c)
      for (int r = 0; r < p->h_cam_np[c]; ++r) gc[p->h_cam_off[c] + r] = hU[(size_t)c * ustride + tri + r];
  if (V) {
    std::vector<double> hV((size_t)6 * p->lay.Ppad);
    HIPCHK(hipMemcpy(hV.data(), p->V, hV.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int q = 0; q < p->P; ++q)
      for (int k = 0; k < 6; ++k) V[(size_t)q * 6 + k] = hV[(size_t)k * p->lay.Ppad + q];
  }
  if (gp) {
    { const int rcw = stage_wait(p); if (rcw) return rcw; }
    HIPCHK(hipMemcpy(p->h_vec, p->g, p->lay.total() * sizeof(double), hipMemcpyDeviceToHost));
    const double* vx = p->h_vec + p->lay.ncp_pad;
    for (int q = 0; q < p->P; ++q) {
      gp[3 * q] = vx[q]; gp[3 * q + 1] = vx[p->lay.Ppad + q]; gp[3 * q + 2] = vx[2 * p->lay.Ppad + q];
    }
  }
  // restore the solver's current point (its blocks must be rebuilt by the next cba_linearize)
  HIPCHK(hipMemcpy(p->x, saved.data(), saved.size() * sizeof(double), hipMemcpyHostToDevice));
  launch_cam_prep(p, p->x, p->tab);
  HIPCHK(hipStreamSynchronize(p->stream));
  p->begun = was_begun; p->first_scale = first; p->linearized = false; p->stepped = false;
  return CBA_OK;
}

int cba_reduced_system(cba_problem* p, double* S, double* rhs) {
  if (!p) return fail(CBA_ERR_INVALID, "cba_reduced_system: null argument");
  if (!p->stepped) return fail(CBA_ERR_INVALID, "cba_reduced_system: call cba_newton_step first");
  HIPCHK(hipSetDevice(p->device));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (S) HIPCHK(hipMemcpy(S, p->S, (size_t)p->ncp * p->ncp * sizeof(double), hipMemcpyDeviceToHost));
  if (rhs) HIPCHK(hipMemcpy(rhs, p->rhs, (size_t)p->ncp * sizeof(double), hipMemcpyDeviceToHost));
  return CBA_OK;
}

}  // extern "C"
