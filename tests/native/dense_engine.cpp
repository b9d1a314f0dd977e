// Test double of the device engine: the primitives cba_solve drives (include/caliscope_ba.h), implemented for a small
// DENSE least-squares problem.  The (robust-scaled) residual rows, their Jacobian and the cost come from a pluggable model:
// callbacks of the test (de_create: linear loss, cost = 0.5 |f|^2) or the bundle-adjustment model of cpu_library.cpp, which
// includes this file and adds the rest of the C ABI.  It lets the CPU suite run csrc/cba_solve.cpp (compiled by g++ together
// with this file) without a GPU.
//
// Sharded by point: a handle that joined a group (cpu_library.cpp) holds the replicated camera block and ITS points
// and rows; `reduce` is the group's all-reduce.  Sums over parameters count the camera block on the lead rank only, sums over rows
// are local; the damped step eliminates the local points, all-reduces the reduced camera system and solves it on every rank —
// the exchange pattern of the device engine (csrc/cba_lib.hip, exchange_at), run by host threads.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/caliscope_ba.h"
#include "../../caliscope_amd/csrc/trf_math.h"

typedef void (*fun_cb)(const double* x, double* r);
typedef void (*jac_cb)(const double* x, double* J);  // row-major m x n

struct BaModel;  // cpu_library.cpp
struct cba_problem {
  int m = 0, n = 0, ncp = 0;
  std::function<double(const double* x, double* f)> fun;   // fills the scaled residual rows, returns the cost 0.5 sum rho
  std::function<void(const double* x, double* J)> jac;     // scaled rows, row-major m x n
  std::vector<double> x0, x, x_new, f, f_new, J, g, s, sinv, sinv_state, cam_diag;
  double cost = 0.0, cost_new = 0.0, last_lam = 0.0;
  bool first_scale = true, cam_scaled = false;
  BaModel* model = nullptr;
  // all-reduce over the group: the first n_sum entries are summed, the next n_max maximised; non-zero = the group was aborted
  std::function<int(double* buf, int n_sum, int n_max)> reduce;
  bool lead = true;  // this rank counts the replicated camera block in sums over the parameters
  long n_damped_steps = 0, n_failed_factorisations = 0;  // since creation (de_counters)
  int exchange(double* buf, int n_sum, int n_max = 0) {
    return (reduce && reduce(buf, n_sum, n_max)) ? cba_set_error(CBA_ERR_INVALID, "group aborted: another rank failed or never arrived") : 0;
  }
};

static thread_local std::string g_err;

extern "C" {

static void de_size(cba_problem* p, int m, int n, int ncp);
cba_problem* de_create(int m, int n, int ncp, fun_cb fun, jac_cb jac) {
  cba_problem* p = new cba_problem;
  p->fun = [fun, m](const double* x, double* f) { fun(x, f); double c = 0.0; for (int i = 0; i < m; ++i) c += f[i] * f[i]; return 0.5 * c; };
  p->jac = [jac](const double* x, double* J) { jac(x, J); };
  de_size(p, m, n, ncp);
  return p;
}
static void de_size(cba_problem* p, int m, int n, int ncp) {
  p->m = m; p->n = n; p->ncp = ncp;
  p->x0.assign(n, 0.0); p->x.assign(n, 0.0); p->x_new.assign(n, 0.0); p->f.assign(m, 0.0); p->f_new.assign(m, 0.0);
  p->J.assign((size_t)m * n, 0.0); p->g.assign(n, 0.0); p->s.assign(n, 0.0); p->sinv.assign(n, 1.0); p->sinv_state.assign(n, 1.0); p->cam_diag.assign(n, 0.0);
}
void de_destroy(cba_problem* p) { delete p; }

// the 2-D trust-region solve and the real-root finder of csrc/trf_math.h, for direct tests
void de_subspace(double b00, double b01, double b11, double g0, double g1, double radius, double* p) {
  trf::solve_subspace_2d(b00, b01, b11, g0, g1, radius, p);
}
int de_real_roots(const double* c, int n_coef, double* out) { return trf::real_roots(c, n_coef, out); }
double de_damping(double H_gg, double gh_sq, double radius) { return trf::damping(H_gg, gh_sq, radius); }
void de_counters(cba_problem* p, long* out) { out[0] = p->n_damped_steps; out[1] = p->n_failed_factorisations; }

int cba_set_error(int32_t code, const char* message) { g_err = message ? message : ""; return code; }
const char* cba_last_error(void) { return g_err.c_str(); }

#ifndef CBA_CPU_LIBRARY  // cpu_library.cpp has its own
int cba_get_info(cba_problem* p, cba_info* out) {
  std::memset(out, 0, sizeof(*out));
  out->n_params = p->n; out->n_cam_params = p->ncp;
  return CBA_OK;
}
#endif

static int begin_common(cba_problem* p, double* cost_out) {
  p->x = p->x0;
  p->cost = p->fun(p->x.data(), p->f.data());
  if (p->exchange(&p->cost, 1)) return CBA_ERR_INVALID;
  p->first_scale = true; p->cam_scaled = false;
  p->cam_diag.assign(p->n, 0.0);
  *cost_out = p->cost;
  return CBA_OK;
}
int cba_begin(cba_problem* p, const double* x0, double* cost_out) {
  p->x0.assign(x0, x0 + p->n);
  return begin_common(p, cost_out);
}
int cba_restart(cba_problem* p, double* cost_out) { return begin_common(p, cost_out); }
int cba_begin_deferred(cba_problem* p, const double* x0) {  // the double evaluates anyway: same state as cba_begin
  double unused;
  if (x0) p->x0.assign(x0, x0 + p->n);
  return begin_common(p, &unused);
}

static int lin_scalars(cba_problem* p, int max_from, cba_linearization* out) {
  const int m = p->m, n = p->n;
  double ginf = 0.0, gh_sq = 0.0, xs = 0.0, xn = 0.0;
  for (int j = p->lead ? 0 : p->ncp; j < n; ++j) {
    const double g = p->g[j], sc = p->sinv[j];
    if (j >= max_from) ginf = std::fmax(ginf, std::fabs(g));
    gh_sq += (g / sc) * (g / sc);
    xs += (p->x[j] * sc) * (p->x[j] * sc);
    xn += p->x[j] * p->x[j];
  }
  double jg_sq = 0.0;
  for (int i = 0; i < m; ++i) {
    double r = 0.0;
    for (int j = 0; j < n; ++j) r += p->J[(size_t)i * n + j] * p->g[j] / (p->sinv[j] * p->sinv[j]);
    jg_sq += r * r;
  }
  double v[5] = {gh_sq, xs, xn, jg_sq, ginf};
  if (p->exchange(v, 4, 1)) return CBA_ERR_INVALID;
  out->g_norm_inf = v[4]; out->gh_sq = v[0]; out->jg_sq = v[3]; out->x_scaled_norm = std::sqrt(v[1]); out->x_norm = std::sqrt(v[2]);
  out->cost = p->cost;
  return CBA_OK;
}

int cba_linearize(cba_problem* p, cba_linearization* out) {
  const int m = p->m, n = p->n;
  p->jac(p->x.data(), p->J.data());
  std::vector<double> col2(n);
  for (int j = 0; j < n; ++j) {
    double g = 0.0, c2 = 0.0;
    for (int i = 0; i < m; ++i) { g += p->J[(size_t)i * n + j] * p->f[i]; c2 += p->J[(size_t)i * n + j] * p->J[(size_t)i * n + j]; }
    p->g[j] = g; col2[j] = c2;
  }
  if (p->reduce) {  // the camera columns have rows on every rank
    std::vector<double> v(2 * p->ncp);
    for (int j = 0; j < p->ncp; ++j) { v[j] = p->g[j]; v[p->ncp + j] = col2[j]; }
    if (p->exchange(v.data(), 2 * p->ncp)) return CBA_ERR_INVALID;
    for (int j = 0; j < p->ncp; ++j) { p->g[j] = v[j]; col2[j] = v[p->ncp + j]; }
  }
  for (int j = 0; j < n; ++j) {
    double sc = std::sqrt(col2[j]);
    if (p->first_scale) { if (sc == 0.0) sc = 1.0; } else sc = std::fmax(sc, p->sinv_state[j]);  // monotone max (common.py:598-610)
    p->sinv_state[j] = sc;
    p->sinv[j] = sc;
  }
  p->first_scale = false;
  return lin_scalars(p, 0, out);
}

int cba_linearize_build(cba_problem* p) { cba_linearization unused; return cba_linearize(p, &unused); }

int cba_set_camera_scaling(cba_problem* p, const double* mult, const double* diag_h, cba_linearization* out) {
  for (int j = 0; j < p->ncp; ++j) {
    p->sinv[j] = p->sinv_state[j] * mult[j];
    p->cam_diag[j] = diag_h[j] * p->sinv[j] * p->sinv[j];
  }
  p->cam_scaled = true;
  return lin_scalars(p, p->ncp, out);
}

// ||p_h||^2, g_h.p_h and ||w||^2 (w = p_h - (g_h.p_h / ||g_h||^2) g_h) of the step in p->s, over the whole parameter vector
static int step_scalars(cba_problem* p, cba_newton_info* out) {
  const int n = p->n, j0 = p->lead ? 0 : p->ncp;
  double v[3] = {0.0, 0.0, 0.0};
  for (int j = j0; j < n; ++j) { const double pj = p->s[j] * p->sinv[j], gh = p->g[j] / p->sinv[j]; v[0] += pj * pj; v[1] += gh * pj; v[2] += gh * gh; }
  if (p->exchange(v, 3)) return CBA_ERR_INVALID;
  double w_sq = 0.0;
  for (int j = j0; j < n; ++j) { const double w = p->s[j] * p->sinv[j] - (v[1] / v[2]) * p->g[j] / p->sinv[j]; w_sq += w * w; }
  if (p->exchange(&w_sq, 1)) return CBA_ERR_INVALID;
  out->p_sq = v[0]; out->gh_dot_p = v[1]; out->w_sq = w_sq;
  return CBA_OK;
}

// Damped normal equations (J^T J + lam D^2 + cam_diag) s = -g by an envelope ("skyline") Cholesky in REVERSED parameter order:
// bundle-adjustment rows touch one camera and a few points, so with the points first the factor of the arrow matrix has no
// fill outside the camera rows and a few-thousand-parameter session factors in milliseconds; a dense test problem simply has a
// full envelope.  The rows of J are scanned for their nonzeros once.  The point rows are eliminated first; what is left of the
// camera rows is the reduced camera system of THIS rank's rows, which is summed over the group before the cameras are solved
// (identically on every rank) and the points substituted back.
int cba_newton_step(cba_problem* p, double lam, cba_newton_info* out) {
  const int m = p->m, n = p->n, ncp = p->ncp, np3 = n - ncp;  // positions [0, np3): points, [np3, n): cameras
  p->last_lam = lam;
  ++p->n_damped_steps;
  std::vector<double> A((size_t)n * n, 0.0), b(n, 0.0);
  std::vector<int> first(n), nz;
  for (int a = 0; a < n; ++a) first[a] = a;
  auto q = [n](int j) { return n - 1 - j; };  // position of parameter j in the factorisation order
  for (int i = 0; i < m; ++i) {
    const double* row = &p->J[(size_t)i * n];
    nz.clear();
    for (int j = n - 1; j >= 0; --j) if (row[j] != 0.0) nz.push_back(j);  // q() ascending
    for (size_t u = 0; u < nz.size(); ++u) {
      const int r = q(nz[u]);
      first[r] = std::min(first[r], q(nz[0]));
      for (size_t v = 0; v <= u; ++v) A[(size_t)r * n + q(nz[v])] += row[nz[u]] * row[nz[v]];
    }
  }
  for (int a = ncp; a < n; ++a) {  // the camera block's damping and gradient enter once, after the sum over the ranks
    A[(size_t)q(a) * n + q(a)] += lam * p->sinv[a] * p->sinv[a] + p->cam_diag[a];
    b[q(a)] = -p->g[a];
  }
  double failed = 0.0;
  for (int i = 0; i < n && failed == 0.0; ++i) {  // row-oriented Cholesky inside the envelope; camera rows stop at the point columns
    double* Li = &A[(size_t)i * n];
    const int j_end = std::min(i, np3 - 1);
    for (int j = first[i]; j <= j_end; ++j) {
      const double* Lj = &A[(size_t)j * n];
      double v = Li[j];
      for (int k = std::max(first[i], first[j]); k < j; ++k) v -= Li[k] * Lj[k];
      if (j < i) Li[j] = v / Lj[j];
      else if (!(v > 0.0)) { failed = 1.0; break; }
      else Li[i] = std::sqrt(v);
    }
  }
  // reduced system of this rank: [S (ncp x ncp, camera positions, lower) | y_c | failed]
  std::vector<double> R((size_t)ncp * ncp + ncp + 1, 0.0);
  if (failed == 0.0) {
    for (int i = 0; i < np3; ++i) { double v = b[i]; for (int k = first[i]; k < i; ++k) v -= A[(size_t)i * n + k] * b[k]; b[i] = v / A[(size_t)i * n + i]; }
    for (int i = np3; i < n; ++i) {
      const double* Li = &A[(size_t)i * n];
      for (int j = np3; j <= i; ++j) {
        const double* Lj = &A[(size_t)j * n];
        double v = Li[j];
        for (int k = std::max(first[i], first[j]); k < np3; ++k) v -= Li[k] * Lj[k];
        R[(size_t)(i - np3) * ncp + (j - np3)] = v;
      }
      double y = 0.0;
      for (int k = first[i]; k < np3; ++k) y -= Li[k] * b[k];
      R[(size_t)ncp * ncp + (i - np3)] = y;
    }
  }
  R.back() = failed;
  if (p->exchange(R.data(), (int)R.size())) return CBA_ERR_INVALID;
  out->ok = 1; out->reserved = 0;
  bool ok = R.back() == 0.0;
  std::vector<double> xc(ncp);
  if (ok) {
    for (int a = 0; a < ncp; ++a) {
      const int i = q(a) - np3;
      R[(size_t)i * ncp + i] += lam * p->sinv[a] * p->sinv[a] + p->cam_diag[a];
      xc[i] = R[(size_t)ncp * ncp + i] - p->g[a];
    }
    for (int i = 0; i < ncp && ok; ++i)
      for (int j = 0; j <= i; ++j) {
        double v = R[(size_t)i * ncp + j];
        for (int k = 0; k < j; ++k) v -= R[(size_t)i * ncp + k] * R[(size_t)j * ncp + k];
        if (j < i) R[(size_t)i * ncp + j] = v / R[(size_t)j * ncp + j];
        else if (!(v > 0.0)) { ok = false; break; }
        else R[(size_t)i * ncp + i] = std::sqrt(v);
      }
  }
  if (!ok) { ++p->n_failed_factorisations; out->ok = 0; out->p_sq = out->gh_dot_p = out->w_sq = 0.0; return CBA_OK; }
  for (int i = 0; i < ncp; ++i) { double v = xc[i]; for (int k = 0; k < i; ++k) v -= R[(size_t)i * ncp + k] * xc[k]; xc[i] = v / R[(size_t)i * ncp + i]; }
  for (int i = ncp - 1; i >= 0; --i) { xc[i] /= R[(size_t)i * ncp + i]; for (int k = 0; k < i; ++k) xc[k] -= R[(size_t)i * ncp + k] * xc[i]; }
  // back-substitution of the points: the camera unknowns first leave the point rows' right-hand sides
  for (int i = n - 1; i >= np3; --i) { b[i] = xc[i - np3]; for (int k = first[i]; k < np3; ++k) b[k] -= A[(size_t)i * n + k] * b[i]; }
  for (int i = np3 - 1; i >= 0; --i) { b[i] /= A[(size_t)i * n + i]; for (int k = first[i]; k < i; ++k) b[k] -= A[(size_t)i * n + k] * b[i]; }
  for (int a = 0; a < n / 2; ++a) std::swap(b[a], b[q(a)]);  // back to parameter order
  p->s = b;
  return step_scalars(p, out);
}

int cba_subspace_gram_ex(cba_problem* p, double a1, double b1, const double* cam1, double a2, double b2, const double* cam2, double* gram) {
  const int m = p->m, n = p->n;
  double s11 = 0.0, s12 = 0.0, s22 = 0.0;
  for (int i = 0; i < m; ++i) {
    double r1 = 0.0, r2 = 0.0;
    for (int j = 0; j < n; ++j) {
      const double d2g = p->g[j] / (p->sinv[j] * p->sinv[j]), Jij = p->J[(size_t)i * n + j];
      r1 += Jij * ((cam1 && j < p->ncp) ? cam1[j] : a1 * d2g + b1 * p->s[j]);
      r2 += Jij * ((cam2 && j < p->ncp) ? cam2[j] : a2 * d2g + b2 * p->s[j]);
    }
    s11 += r1 * r1; s12 += r1 * r2; s22 += r2 * r2;
  }
  gram[0] = s11; gram[1] = s12; gram[2] = s22;
  return p->exchange(gram, 3) ? CBA_ERR_INVALID : CBA_OK;
}

int cba_subspace_gram(cba_problem* p, double a1, double b1, double a2, double b2, double* gram) {
  return cba_subspace_gram_ex(p, a1, b1, nullptr, a2, b2, nullptr, gram);
}

int cba_trial_ex(cba_problem* p, double alpha, double beta, const double* cam_x_new, cba_trial_info* out) {
  double sn = 0.0;
  for (int j = 0; j < p->n; ++j) {
    double step = alpha * p->g[j] / (p->sinv[j] * p->sinv[j]) + beta * p->s[j];
    p->x_new[j] = p->x[j] + step;
    if (cam_x_new && j < p->ncp) { p->x_new[j] = cam_x_new[j]; step = p->x_new[j] - p->x[j]; }
    if (p->lead || j >= p->ncp) sn += step * step;
  }
  p->cost_new = p->fun(p->x_new.data(), p->f_new.data());
  double v[2] = {sn, p->cost_new};
  if (p->exchange(v, 2)) return CBA_ERR_INVALID;
  sn = v[0]; p->cost_new = v[1];
  out->cost = p->cost_new; out->step_norm = std::sqrt(sn); out->finite = std::isfinite(out->cost) ? 1 : 0; out->reserved = 0;
  if (!out->finite) out->cost = NAN;
  return CBA_OK;
}

int cba_trial(cba_problem* p, double alpha, double beta, cba_trial_info* out) { return cba_trial_ex(p, alpha, beta, nullptr, out); }

// the fused iteration of the device engine, emulated with the primitives above and the same scalar code (trf_math.h)
int cba_step_supported(cba_problem* p) { return p->cam_scaled ? 0 : 1; }
// no fused bounded iteration in the dense test double: cba_solve keeps scipy's bounded loop on the primitives (the route these tests pin)
int cba_set_bounds(cba_problem*, const double*, const double*) { return 0; }
int cba_step_camera_state(cba_problem*, double*, double*, double*, double*) { g_err = "cba_step_camera_state: not available in the dense test build"; return CBA_ERR_UNSUPPORTED; }
int cba_step(cba_problem* p, double radius_in, cba_step_info* out) {
  std::memset(out, 0, sizeof(*out));
  int rc = cba_linearize(p, &out->lin);
  if (rc) return rc;
  const double gh_sq = out->lin.gh_sq, gh_norm = std::sqrt(gh_sq);
  const double radius = radius_in > 0.0 ? radius_in : (out->lin.x_scaled_norm > 0.0 ? out->lin.x_scaled_norm : 1.0);
  const double lam = trf::damping(out->lin.jg_sq, gh_sq, radius);
  out->lam = lam; out->radius = radius;
  if ((rc = cba_newton_step(p, lam, &out->newton))) return rc;
  const double p_sq = out->newton.p_sq, ghp = out->newton.gh_dot_p;
  const double w_sq = p_sq - ghp * ghp / gh_sq;  // derived, as on the device
  out->newton.w_sq = w_sq;
  const bool two_d = true;
  if (!out->newton.ok || !(gh_sq > 0.0) || !(w_sq > 1e-3 * p_sq)) { out->need_host = 1; return CBA_OK; }
  double b00, b01 = 0.0, b11, pS[2];
  const double w_norm = two_d ? std::sqrt(w_sq) : 1.0, c = ghp / gh_sq;
  if (two_d) trf::subspace_model(out->lin.jg_sq, gh_sq, lam, ghp, p_sq, w_sq, &b00, &b01, &b11);
  else { b00 = out->lin.jg_sq / gh_sq; b11 = 1.0; }
  trf::solve_subspace_2d(b00, b01, b11, gh_norm, 0.0, radius, pS);
  if (!two_d) pS[1] = 0.0;
  out->p_s[0] = pS[0]; out->p_s[1] = pS[1];
  out->predicted = -(0.5 * (pS[0] * (b00 * pS[0] + b01 * pS[1]) + pS[1] * (b01 * pS[0] + b11 * pS[1])) + gh_norm * pS[0]);
  out->beta = two_d ? pS[1] / w_norm : 0.0;
  out->alpha = pS[0] / gh_norm - out->beta * c;
  return cba_trial(p, out->alpha, out->beta, &out->trial);
}

int cba_refresh_step_scalars(cba_problem* p, cba_newton_info* out) {
  out->ok = 1; out->reserved = 0;
  return step_scalars(p, out);
}

int cba_accept(cba_problem* p) { p->x = p->x_new; p->f = p->f_new; p->cost = p->cost_new; return CBA_OK; }

static const std::vector<double>& vec_of(cba_problem* p, int32_t which) {
  return which == CBA_VEC_X ? p->x : which == CBA_VEC_X_NEW ? p->x_new : which == CBA_VEC_GRAD ? p->g : which == CBA_VEC_STEP ? p->s : p->sinv;
}
int cba_get_vector(cba_problem* p, int32_t which, double* out) {
  std::memcpy(out, vec_of(p, which).data(), sizeof(double) * p->n);
  return CBA_OK;
}
int cba_get_camera_state(cba_problem* p, double* x_c, double* g_c, double* sinv_c) {
  std::memcpy(x_c, p->x.data(), sizeof(double) * p->ncp);
  std::memcpy(g_c, p->g.data(), sizeof(double) * p->ncp);
  std::memcpy(sinv_c, p->sinv.data(), sizeof(double) * p->ncp);
  return CBA_OK;
}
int cba_get_camera_params(cba_problem* p, int32_t which, double* out) {
  std::memcpy(out, vec_of(p, which).data(), sizeof(double) * p->ncp);
  return CBA_OK;
}

}  // extern "C"
