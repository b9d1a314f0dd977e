// CPU TEST BUILD of the C ABI (include/caliscope_ba.h) — test infrastructure, never a product fallback: the product library
// (caliscope_amd/csrc/cba_lib.hip) has no CPU path and `caliscope_amd._lib` loads this file only when a test points
// CALISCOPE_BA_LIB at it.  It lets a GPU-less CI drive the real boundary — ctypes structs, HipEngine marshalling, the
// least_squares seam, CaptureVolume.optimize(), csrc/cba_solve.cpp — end to end (last sentence).
//
// Arithmetic: the per-observation code the kernels inline (csrc/ba_math.h: projection, Jacobian blocks, robust-loss
// scaling) on DENSE matrices; the trust-region primitives are those of dense_engine.cpp (included below), the damped step
// eliminates the points of an envelope Cholesky of J^T J + lam D^2 and solves the reduced camera system.  Sizes: test scenes
// (n <~ 3000 parameters).  No device and no RCCL; cba_group_* joins handles of one process (one host thread each) into a
// point-sharded solve with a host all-reduce, the protocol of the device build, and cba_comm_* does the same between processes over a
// unix-domain socket (the launcher route of bench.py).  cba_triangulate is not available.
#define CBA_CPU_LIBRARY 1
#include "dense_engine.cpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <sys/socket.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>

#include "../../caliscope_amd/csrc/ba_math.h"
#include "../../caliscope_amd/csrc/host_plan.h"

using namespace cba;

struct BaModel {
  int C = 0, P = 0, ncp = 0, n = 0, m = 0, n_con = 0, loss = 0;
  long N = 0;
  double f_scale = 1.0;
  std::vector<int> np, model, off, ocam, opt;
  std::vector<double> cconst, ouv;
  std::vector<int> ga, gb;      // [n_con][4]
  std::vector<double> dist, wgt;
  int max_obs_per_point = 0;

  // raw (unscaled) residual rows at x: 2 per observation in the caller's order, then the constraint rows
  void residuals(const double* x, double* r) const {
    std::vector<CamTab> tab(C);
    for (int c = 0; c < C; ++c) {
      double xc[MAX_NC] = {0};
      for (int k = 0; k < np[c]; ++k) xc[k] = x[off[c] + k];
      cam_prepare(xc, &cconst[(size_t)c * CAM_CONST_STRIDE], model[c], np[c], &tab[c]);
    }
    for (long i = 0; i < N; ++i) {
      const double* X = x + ncp + 3L * opt[i];
      project_residual(tab[ocam[i]], X[0], X[1], X[2], ouv[2 * i], ouv[2 * i + 1], r + 2 * i);
    }
    for (int k = 0; k < n_con; ++k) {
      double d[3];
      con_delta(x, k, d);
      r[2 * N + k] = wgt[k] * (std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) - dist[k]);
    }
  }
  void con_delta(const double* x, int k, double* d) const {
    d[0] = d[1] = d[2] = 0.0;
    for (int s = 0; s < 4; ++s)
      for (int a = 0; a < 3; ++a) d[a] += 0.25 * (x[ncp + 3L * ga[4 * k + s] + a] - x[ncp + 3L * gb[4 * k + s] + a]);
  }
  // scipy's scaled rows (least_squares.py:169-237, common.py:720-731): f <- f rho' / sqrt(rho' + 2 rho'' f^2), returns 0.5 sum rho
  double eval(const double* x, double* f) const {
    residuals(x, f);
    double cost = 0.0;
    for (int i = 0; i < m; ++i) {
      double rs, er;
      cost += robust_one(loss, f_scale, f[i], &rs, &er);
      f[i] = er;
    }
    return 0.5 * cost;
  }
  void jac(const double* x, double* J) const {
    std::fill(J, J + (size_t)m * n, 0.0);
    std::vector<CamTab> tab(C);
    for (int c = 0; c < C; ++c) {
      double xc[MAX_NC] = {0};
      for (int k = 0; k < np[c]; ++k) xc[k] = x[off[c] + k];
      cam_prepare(xc, &cconst[(size_t)c * CAM_CONST_STRIDE], model[c], np[c], &tab[c]);
    }
    for (long i = 0; i < N; ++i) {
      const int c = ocam[i];
      const double* X = x + ncp + 3L * opt[i];
      double e[2], A[2][MAX_NC], B[2][3];
      project_full(tab[c], X[0], X[1], X[2], ouv[2 * i], ouv[2 * i + 1], e, A, B);
      for (int r = 0; r < 2; ++r) {
        double rs, er;
        robust_one(loss, f_scale, e[r], &rs, &er);
        double* row = J + (size_t)(2 * i + r) * n;
        for (int k = 0; k < np[c]; ++k) row[off[c] + k] = A[r][k] * rs;
        for (int a = 0; a < 3; ++a) row[ncp + 3L * opt[i] + a] = B[r][a] * rs;
      }
    }
    for (int k = 0; k < n_con; ++k) {
      double d[3];
      con_delta(x, k, d);
      const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      double rs, er;
      robust_one(loss, f_scale, wgt[k] * (len - dist[k]), &rs, &er);
      double* row = J + (size_t)(2 * N + k) * n;
      for (int s = 0; s < 4; ++s)
        for (int a = 0; a < 3; ++a) {
          const double u = len > 0.0 ? d[a] / len : 0.0;
          row[ncp + 3L * ga[4 * k + s] + a] += 0.25 * wgt[k] * u * rs;
          row[ncp + 3L * gb[4 * k + s] + a] -= 0.25 * wgt[k] * u * rs;
        }
    }
  }
};

static int failf(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

static void bind_model(cba_problem* p) {
  BaModel* md = p->model;
  md->m = (int)(2 * md->N) + md->n_con;
  p->fun = [md](const double* x, double* f) { return md->eval(x, f); };
  p->jac = [md](const double* x, double* J) { md->jac(x, J); };
  de_size(p, md->m, md->n, md->ncp);
}

extern "C" {

int cba_version(void) { return CBA_VERSION; }
int cba_device_count(void) { return 1; }  // "the host": this build exists for machines without a GPU

int cba_create(const cba_problem_desc* d, const cba_options*, cba_problem** out) {
  if (!d || !out) return failf(CBA_ERR_INVALID, "cba_create: null argument");
  *out = nullptr;
  if (d->n_cams <= 0 || d->n_points <= 0 || d->n_obs <= 0) return failf(CBA_ERR_INVALID, "cba_create: empty problem (cams=%d points=%d obs=%lld)", d->n_cams, d->n_points, (long long)d->n_obs);
  BaModel* md = new BaModel;
  md->C = d->n_cams; md->P = d->n_points; md->N = d->n_obs; md->loss = d->loss; md->f_scale = d->f_scale;
  for (int c = 0; c < md->C; ++c) {
    const int np = d->cam_n_params[c], mo = d->cam_model[c];
    if ((np != 6 && np != 9) || (mo != CBA_MODEL_PINHOLE_BC5 && mo != CBA_MODEL_FISHEYE4) || (mo == CBA_MODEL_FISHEYE4 && np != 6) ||
        !(d->cam_const[c * 12] > 0.0)) {
      delete md;
      return failf(CBA_ERR_INVALID, "camera %d: invalid description (n_params %d, model %d)", c, np, mo);
    }
    md->np.push_back(np); md->model.push_back(mo); md->off.push_back(md->ncp);
    md->ncp += np;
  }
  md->n = md->ncp + 3 * md->P;
  md->cconst.assign(d->cam_const, d->cam_const + (size_t)md->C * 12);
  md->ocam.assign(d->obs_cam, d->obs_cam + md->N); md->opt.assign(d->obs_pt, d->obs_pt + md->N);
  md->ouv.assign(d->obs_uv, d->obs_uv + 2 * md->N);
  std::vector<int> cnt(md->P, 0);
  for (long i = 0; i < md->N; ++i) {
    if (md->ocam[i] < 0 || md->ocam[i] >= md->C) { delete md; return failf(CBA_ERR_INVALID, "observation %ld: camera index %d out of range", i, d->obs_cam[i]); }
    if (md->opt[i] < 0 || md->opt[i] >= md->P) { delete md; return failf(CBA_ERR_INVALID, "observation %ld: world-point index %d out of range", i, d->obs_pt[i]); }
    md->max_obs_per_point = std::max(md->max_obs_per_point, ++cnt[md->opt[i]]);
  }
  if ((double)(2 * md->N) * md->n > 4e8) { delete md; return failf(CBA_ERR_UNSUPPORTED, "the CPU test build holds a dense Jacobian: %ld x %d is too large", 2 * md->N, md->n); }
  cba_problem* p = new cba_problem;
  p->model = md;
  bind_model(p);
  *out = p;
  return CBA_OK;
}

void cba_destroy(cba_problem* p) {
  if (!p) return;
  delete p->model;
  delete p;
}

int cba_set_constraints(cba_problem* p, int32_t n_con, const int32_t* groups_a, const int32_t* groups_b, const double* distances,
                        const double* weights) {
  if (!p || !p->model) return failf(CBA_ERR_INVALID, "cba_set_constraints: null problem");
  BaModel* md = p->model;
  if (md->n_con) return failf(CBA_ERR_INVALID, "cba_set_constraints: constraints are already set");
  if (n_con <= 0) return CBA_OK;
  for (long e = 0; e < 4L * n_con; ++e)
    if (groups_a[e] < 0 || groups_a[e] >= md->P || groups_b[e] < 0 || groups_b[e] >= md->P)
      return failf(CBA_ERR_INVALID, "cba_set_constraints: point index out of range in constraint %ld", e / 4);
  md->n_con = n_con;
  md->ga.assign(groups_a, groups_a + 4L * n_con); md->gb.assign(groups_b, groups_b + 4L * n_con);
  md->dist.assign(distances, distances + n_con); md->wgt.assign(weights, weights + n_con);
  bind_model(p);
  return CBA_OK;
}

int cba_plan_wait(cba_problem* p) { return p ? CBA_OK : failf(CBA_ERR_INVALID, "cba_plan_wait: null problem"); }  // (no plan in the dense test build)

int cba_set_loss(cba_problem* p, int32_t loss, double f_scale) {
  if (!p || !p->model) return failf(CBA_ERR_INVALID, "cba_set_loss: null problem");
  if (loss < CBA_LOSS_LINEAR || loss > CBA_LOSS_ARCTAN || (loss != CBA_LOSS_LINEAR && !(f_scale > 0.0))) return failf(CBA_ERR_INVALID, "cba_set_loss: bad loss / f_scale");
  p->model->loss = loss; p->model->f_scale = f_scale;
  return CBA_OK;
}

int64_t cba_trim(void) { return 0; }
// (cba_set_bounds / cba_step_camera_state: dense_engine.cpp — the dense test build has no fused bounded iteration)  // (the dense test build keeps nothing between handles)

int cba_get_info(cba_problem* p, cba_info* out) {
  std::memset(out, 0, sizeof(*out));
  const BaModel* md = p->model;
  out->n_params = p->n; out->n_cam_params = p->ncp;
  if (md) { out->n_cams = md->C; out->n_points = md->P; out->n_obs = md->N; out->max_obs_per_point = md->max_obs_per_point; }
  return CBA_OK;
}

int cba_residuals(cba_problem* p, const double* x, double* r_out, double* cost_out) {
  if (!p || !p->model || !x || !r_out) return failf(CBA_ERR_INVALID, "cba_residuals: null argument");
  const BaModel* md = p->model;
  md->residuals(x, r_out);
  if (cost_out) {
    double c = 0.0;
    for (int i = 0; i < md->m; ++i) c += robust_cost_one(md->loss, md->f_scale, r_out[i]);
    *cost_out = 0.5 * c;
  }
  return CBA_OK;
}

// (U [C][9][9], V [P][6] = xx xy xz yy yz zz, g_c [ncp], g_p [P][3]) of the robust-scaled J^T J and J^T f at x (reprojection rows only,
// like the device hook)
int cba_normal_blocks(cba_problem* p, const double* x, double* U, double* V, double* gc, double* gp) {
  if (!p || !p->model) return failf(CBA_ERR_INVALID, "cba_normal_blocks: null problem");
  const BaModel* md = p->model;
  const int n = md->n, m2 = (int)(2 * md->N);
  std::vector<double> J((size_t)md->m * n), f(md->m);
  md->eval(x, f.data());
  md->jac(x, J.data());
  std::fill(U, U + (size_t)md->C * 81, 0.0); std::fill(V, V + (size_t)md->P * 6, 0.0);
  std::fill(gc, gc + md->ncp, 0.0); std::fill(gp, gp + (size_t)md->P * 3, 0.0);
  for (int i = 0; i < m2; ++i) {
    const double* row = &J[(size_t)i * n];
    const int c = md->ocam[i / 2], pt = md->opt[i / 2], o = md->off[c];
    for (int a = 0; a < md->np[c]; ++a) {
      gc[o + a] += row[o + a] * f[i];
      for (int b = 0; b < md->np[c]; ++b) U[(size_t)c * 81 + a * 9 + b] += row[o + a] * row[o + b];
    }
    const double* rp = row + md->ncp + 3L * pt;
    double* v = V + (size_t)pt * 6;
    v[0] += rp[0] * rp[0]; v[1] += rp[0] * rp[1]; v[2] += rp[0] * rp[2]; v[3] += rp[1] * rp[1]; v[4] += rp[1] * rp[2]; v[5] += rp[2] * rp[2];
    for (int a = 0; a < 3; ++a) gp[(size_t)pt * 3 + a] += rp[a] * f[i];
  }
  return CBA_OK;
}

// reduced camera system of the last damped step: S = H_cc + lam D_c^2 + cam_diag - H_cp (H_pp + lam D_p^2)^-1 H_pc, rhs likewise
int cba_reduced_system(cba_problem* p, double* S, double* rhs) {
  const int n = p->n, ncp = p->ncp, m = p->m, np3 = n - ncp;
  std::vector<double> H((size_t)n * n, 0.0);
  for (int a = 0; a < n; ++a)
    for (int b = a; b < n; ++b) {
      double v = 0.0;
      for (int i = 0; i < m; ++i) v += p->J[(size_t)i * n + a] * p->J[(size_t)i * n + b];
      H[(size_t)a * n + b] = H[(size_t)b * n + a] = v;
    }
  for (int a = 0; a < n; ++a) H[(size_t)a * n + a] += p->last_lam * p->sinv[a] * p->sinv[a] + (a < ncp ? p->cam_diag[a] : 0.0);
  // X = H_pp^-1 [H_pc | g_p] by Cholesky of H_pp
  std::vector<double> L((size_t)np3 * np3), X((size_t)np3 * (ncp + 1));
  for (int i = 0; i < np3; ++i) {
    for (int j = 0; j <= i; ++j) {
      double v = H[(size_t)(ncp + i) * n + ncp + j];
      for (int k = 0; k < j; ++k) v -= L[(size_t)i * np3 + k] * L[(size_t)j * np3 + k];
      if (i == j) { if (!(v > 0.0)) v = 1.0; L[(size_t)i * np3 + i] = std::sqrt(v); }  // unobserved point: decoupled
      else L[(size_t)i * np3 + j] = v / L[(size_t)j * np3 + j];
    }
    for (int c = 0; c < ncp; ++c) X[(size_t)i * (ncp + 1) + c] = H[(size_t)(ncp + i) * n + c];
    X[(size_t)i * (ncp + 1) + ncp] = p->g[ncp + i];
  }
  for (int c = 0; c <= ncp; ++c) {
    for (int i = 0; i < np3; ++i) { double v = X[(size_t)i * (ncp + 1) + c]; for (int k = 0; k < i; ++k) v -= L[(size_t)i * np3 + k] * X[(size_t)k * (ncp + 1) + c]; X[(size_t)i * (ncp + 1) + c] = v / L[(size_t)i * np3 + i]; }
    for (int i = np3 - 1; i >= 0; --i) { double v = X[(size_t)i * (ncp + 1) + c]; for (int k = i + 1; k < np3; ++k) v -= L[(size_t)k * np3 + i] * X[(size_t)k * (ncp + 1) + c]; X[(size_t)i * (ncp + 1) + c] = v / L[(size_t)i * np3 + i]; }
  }
  for (int a = 0; a < ncp; ++a) {
    for (int b = 0; b < ncp; ++b) {
      double v = H[(size_t)a * n + b];
      for (int i = 0; i < np3; ++i) v -= H[(size_t)a * n + ncp + i] * X[(size_t)i * (ncp + 1) + b];
      S[(size_t)a * ncp + b] = v;
    }
    double v = -p->g[a];
    for (int i = 0; i < np3; ++i) v += H[(size_t)a * n + ncp + i] * X[(size_t)i * (ncp + 1) + ncp];
    rhs[a] = v;
  }
  return CBA_OK;
}

int64_t cba_host_plan(int32_t n_points, int64_t n_obs, const int32_t* obs_pt, const int32_t* obs_cam, int32_t n_cams,
                      int32_t chunk_cap, int64_t* order_out, int64_t* pt_start_out, int64_t* chunk_start_out) {
  return host_plan_impl([](int code, const char* fmt, auto... args) { return failf(code, fmt, args...); }, n_points, n_obs, obs_pt, obs_cam, n_cams, chunk_cap,
                        order_out, pt_start_out, chunk_start_out);
}

// timers: names of the device build, no time (there is no device)
static const char* kTimers[] = {"cam_prep", "cost", "build", "build_reduce", "scale_scalars", "jv", "schur", "schur_reduce_finalize",
                                "cholesky_solve", "backsub", "vector_ops", "schur_pairs", "exchange"};
int cba_timer_count(void) { return 13; }
const char* cba_timer_name(int32_t i) { return (i >= 0 && i < 13) ? kTimers[i] : ""; }
int cba_get_timers(cba_problem*, double* ms, int64_t* calls) { for (int i = 0; i < 13; ++i) { ms[i] = 0.0; calls[i] = 0; } return CBA_OK; }
int cba_reset_timers(cba_problem*) { return CBA_OK; }
int cba_enable_timers(cba_problem*, int32_t) { return CBA_OK; }

// RCCL has no CPU counterpart.  Between PROCESSES (what `python -m torch.distributed.run` starts) the test build offers the same call shape instead
// — rank 0 makes a 128-byte id, every rank calls cba_comm_init with it — over a unix-domain socket the id names: rank 0 listens, folds the ranks'
// buffers in rank order and answers.  It exists so that bench.py's launcher branch (control plane, id broadcast, sharding per rank, max over ranks)
// runs end to end on a GPU-less machine; it is test plumbing, not a transport.
struct ProcComm {
  int rank = 0, world = 1, listener = -1;
  std::vector<int> fd;  // rank 0: fd[r] of rank r; other ranks: fd[0] to rank 0
  std::string path;
  std::vector<double> tmp;
  ~ProcComm() {
    for (int f : fd) if (f >= 0) ::close(f);
    if (listener >= 0) { ::close(listener); ::unlink(path.c_str()); }
  }
  static bool send_all(int f, const void* buf, size_t n) {
    const char* c = static_cast<const char*>(buf);
    while (n) { const ssize_t k = ::send(f, c, n, MSG_NOSIGNAL); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
  }
  static bool recv_all(int f, void* buf, size_t n) {
    char* c = static_cast<char*>(buf);
    while (n) { const ssize_t k = ::recv(f, c, n, 0); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
  }
  int all_reduce(double* buf, int n_sum, int n_max) {
    const int n = n_sum + n_max;
    if (rank != 0) {
      const int32_t hdr[2] = {n_sum, n_max};
      if (!send_all(fd[0], hdr, sizeof hdr) || !send_all(fd[0], buf, sizeof(double) * n) || !recv_all(fd[0], buf, sizeof(double) * n)) return 1;
      return 0;
    }
    tmp.resize((size_t)n);
    for (int r = 1; r < world; ++r) {  // rank order: every replica gets the same bits
      int32_t hdr[2];
      if (!recv_all(fd[r], hdr, sizeof hdr) || hdr[0] != n_sum || hdr[1] != n_max || !recv_all(fd[r], tmp.data(), sizeof(double) * n)) return 1;
      for (int j = 0; j < n; ++j) buf[j] = j < n_sum ? buf[j] + tmp[j] : std::fmax(buf[j], tmp[j]);
    }
    for (int r = 1; r < world; ++r)
      if (!send_all(fd[r], buf, sizeof(double) * n)) return 1;
    return 0;
  }
};

int cba_comm_unique_id(char* out128) {
  static std::atomic<int> counter{0};
  std::memset(out128, 0, 128);
  std::snprintf(out128, 100, "/tmp/cba_cpu_comm_%d_%d_%lld", (int)::getpid(), counter++,
                (long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return CBA_OK;
}
int cba_comm_init(cba_problem* p, const char* id128, int32_t rank, int32_t world) {
  if (!p || !id128 || world < 1 || rank < 0 || rank >= world) return failf(CBA_ERR_INVALID, "cba_comm_init: bad arguments");
  if (world == 1) return CBA_OK;
  auto comm = std::make_shared<ProcComm>();
  comm->rank = rank; comm->world = world; comm->path.assign(id128, strnlen(id128, 100));
  sockaddr_un addr{};
  addr.sun_family = AF_UNIX;
  std::snprintf(addr.sun_path, sizeof addr.sun_path, "%s", comm->path.c_str());
  const timeval tv{120, 0};
  if (rank == 0) {
    comm->listener = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (comm->listener < 0 || ::bind(comm->listener, reinterpret_cast<sockaddr*>(&addr), sizeof addr) || ::listen(comm->listener, world))
      return failf(CBA_ERR_INVALID, "cba_comm_init (CPU test build): cannot listen on %s", comm->path.c_str());
    ::setsockopt(comm->listener, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);  // (accept gives up after two minutes: a rank that never came)
    comm->fd.assign((size_t)world, -1);
    for (int k = 1; k < world; ++k) {
      const int f = ::accept(comm->listener, nullptr, nullptr);
      int32_t r = -1;
      if (f < 0) return failf(CBA_ERR_INVALID, "cba_comm_init (CPU test build): a rank did not connect");
      ::setsockopt(f, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
      if (!ProcComm::recv_all(f, &r, sizeof r) || r < 1 || r >= world || comm->fd[(size_t)r] >= 0) { ::close(f); return failf(CBA_ERR_INVALID, "cba_comm_init (CPU test build): bad rank on the wire"); }
      comm->fd[(size_t)r] = f;
    }
  } else {
    const int f = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (f < 0) return failf(CBA_ERR_INVALID, "cba_comm_init (CPU test build): socket");
    comm->fd.assign(1, f);
    bool ok = false;
    for (int attempt = 0; attempt < 6000 && !ok; ++attempt) {  // rank 0 may not be listening yet
      ok = ::connect(f, reinterpret_cast<sockaddr*>(&addr), sizeof addr) == 0;
      if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    ::setsockopt(f, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    const int32_t r = rank;
    if (!ok || !ProcComm::send_all(f, &r, sizeof r)) return failf(CBA_ERR_INVALID, "cba_comm_init (CPU test build): rank %d could not reach rank 0 at %s", rank, comm->path.c_str());
  }
  p->reduce = [comm](double* buf, int n_sum, int n_max) { return comm->all_reduce(buf, n_sum, n_max); };
  p->lead = rank == 0;
  return CBA_OK;
}

int cba_comm_abort(cba_problem*) { return CBA_OK; }

// ... but the in-process group does: one host thread per member handle, a mutex/condvar barrier and one staging slot per rank.
// Every rank folds the slots in rank order, so the replicas stay bit-identical.
struct cba_group {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long generation = 0;
  bool aborted = false;
  std::vector<std::vector<double>> slot;

  bool barrier() {  // false: the group was aborted
    std::unique_lock<std::mutex> lk(mu);
    if (aborted) return false;
    const unsigned long gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
    else if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || aborted; })) aborted = true;  // a rank never arrived
    if (aborted) { cv.notify_all(); return false; }
    return true;
  }
  int all_reduce(int rank, double* buf, int n_sum, int n_max) {
    slot[rank].assign(buf, buf + n_sum + n_max);
    if (!barrier()) return 1;
    for (int r = 0; r < world; ++r)
      if ((int)slot[r].size() != n_sum + n_max) { std::lock_guard<std::mutex> lk(mu); aborted = true; cv.notify_all(); return 1; }  // the ranks disagree about the collective
    for (int j = 0; j < n_sum + n_max; ++j) {
      double acc = slot[0][j];
      for (int r = 1; r < world; ++r) acc = j < n_sum ? acc + slot[r][j] : std::fmax(acc, slot[r][j]);
      buf[j] = acc;
    }
    return barrier() ? 0 : 1;  // the slots are free again
  }
};

int cba_group_create(int32_t world, cba_group** out) {
  if (!out || world < 1 || world > 16) return failf(CBA_ERR_INVALID, "cba_group_create: world %d outside 1..16", world);
  cba_group* g = new cba_group;
  g->world = world; g->slot.resize(world);
  *out = g;
  return CBA_OK;
}
int cba_group_join(cba_problem* p, cba_group* g, int32_t rank) {
  if (!p || !g || rank < 0 || rank >= g->world) return failf(CBA_ERR_INVALID, "cba_group_join: bad arguments");
  if (!g->barrier()) return failf(CBA_ERR_INVALID, "cba_group_join: the group was aborted (another rank failed before joining)");
  if (g->world > 1) {
    p->reduce = [g, rank](double* buf, int n_sum, int n_max) { return g->all_reduce(rank, buf, n_sum, n_max); };
    p->lead = rank == 0;
  }
  return CBA_OK;
}
void cba_group_abort(cba_group* g) {
  if (!g) return;
  std::lock_guard<std::mutex> lk(g->mu);
  g->aborted = true;
  g->cv.notify_all();
}
void cba_group_destroy(cba_group* g) { delete g; }

int cba_triangulate(const cba_triangulate_desc*, int32_t, double*, double*) {
  return failf(CBA_ERR_UNSUPPORTED, "cba_triangulate is a device kernel: not part of the CPU test build");
}

}  // extern "C"
