// Test double of the device engine: the primitives cba_solve drives (include/caliscope_ba.h), implemented for a small
// DENSE least-squares problem.  The (robust-scaled) residual rows, their Jacobian and the cost come from a pluggable model:
// callbacks of the test (de_create: linear loss, cost = 0.5 |f|^2) or the bundle-adjustment model of cpu_library.cpp, which
// includes this file and adds the rest of the C ABI.  It lets the CPU suite run csrc/cba_solve.cpp (compiled by g++ together
// with this file) without a GPU.
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

static void lin_scalars(cba_problem* p, int max_from, cba_linearization* out) {
  const int m = p->m, n = p->n;
  double ginf = 0.0, gh_sq = 0.0, xs = 0.0, xn = 0.0;
  for (int j = 0; j < n; ++j) {
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
  out->g_norm_inf = ginf; out->gh_sq = gh_sq; out->jg_sq = jg_sq; out->x_scaled_norm = std::sqrt(xs); out->x_norm = std::sqrt(xn);
  out->cost = p->cost;
}

int cba_linearize(cba_problem* p, cba_linearization* out) {
  const int m = p->m, n = p->n;
  p->jac(p->x.data(), p->J.data());
  for (int j = 0; j < n; ++j) {
    double g = 0.0, c2 = 0.0;
    for (int i = 0; i < m; ++i) { g += p->J[(size_t)i * n + j] * p->f[i]; c2 += p->J[(size_t)i * n + j] * p->J[(size_t)i * n + j]; }
    p->g[j] = g;
    double sc = std::sqrt(c2);
    if (p->first_scale) { if (sc == 0.0) sc = 1.0; } else sc = std::fmax(sc, p->sinv_state[j]);  // monotone max (common.py:598-610)
    p->sinv_state[j] = sc;
    p->sinv[j] = sc;
  }
  p->first_scale = false;
  lin_scalars(p, 0, out);
  return CBA_OK;
}

int cba_linearize_build(cba_problem* p) { cba_linearization unused; return cba_linearize(p, &unused); }

int cba_set_camera_scaling(cba_problem* p, const double* mult, const double* diag_h, cba_linearization* out) {
  for (int j = 0; j < p->ncp; ++j) {
    p->sinv[j] = p->sinv_state[j] * mult[j];
    p->cam_diag[j] = diag_h[j] * p->sinv[j] * p->sinv[j];
  }
  p->cam_scaled = true;
  lin_scalars(p, p->ncp, out);
  return CBA_OK;
}

// Damped normal equations (J^T J + lam D^2 + cam_diag) s = -g by an envelope ("skyline") Cholesky in REVERSED parameter order:
// bundle-adjustment rows touch one camera and a few points, so with the points first the factor of the arrow matrix has no
// fill outside the camera rows and a few-thousand-parameter session factors in milliseconds; a dense test problem simply has a
// full envelope.  The rows of J are scanned for their nonzeros once.
int cba_newton_step(cba_problem* p, double lam, cba_newton_info* out) {
  const int m = p->m, n = p->n;
  p->last_lam = lam;
  std::vector<double> A((size_t)n * n, 0.0), b(n);
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
  for (int a = 0; a < n; ++a) {
    A[(size_t)q(a) * n + q(a)] += lam * p->sinv[a] * p->sinv[a] + p->cam_diag[a];
    b[q(a)] = -p->g[a];
  }
  out->ok = 1; out->reserved = 0;
  for (int i = 0; i < n; ++i) {  // row-oriented Cholesky inside the envelope
    double* Li = &A[(size_t)i * n];
    for (int j = first[i]; j <= i; ++j) {
      const double* Lj = &A[(size_t)j * n];
      double v = Li[j];
      for (int k = std::max(first[i], first[j]); k < j; ++k) v -= Li[k] * Lj[k];
      if (j < i) Li[j] = v / Lj[j];
      else {
        if (!(v > 0.0)) { out->ok = 0; out->p_sq = out->gh_dot_p = out->w_sq = 0.0; return CBA_OK; }
        Li[i] = std::sqrt(v);
      }
    }
  }
  for (int i = 0; i < n; ++i) { double v = b[i]; for (int k = first[i]; k < i; ++k) v -= A[(size_t)i * n + k] * b[k]; b[i] = v / A[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { b[i] /= A[(size_t)i * n + i]; for (int k = first[i]; k < i; ++k) b[k] -= A[(size_t)i * n + k] * b[i]; }
  for (int a = 0; a < n / 2; ++a) std::swap(b[a], b[q(a)]);  // back to parameter order
  p->s = b;
  double p_sq = 0.0, ghp = 0.0, gh_sq = 0.0;
  for (int j = 0; j < n; ++j) { const double pj = b[j] * p->sinv[j], gh = p->g[j] / p->sinv[j]; p_sq += pj * pj; ghp += gh * pj; gh_sq += gh * gh; }
  double w_sq = 0.0;
  for (int j = 0; j < n; ++j) { const double w = b[j] * p->sinv[j] - (ghp / gh_sq) * p->g[j] / p->sinv[j]; w_sq += w * w; }
  out->p_sq = p_sq; out->gh_dot_p = ghp; out->w_sq = w_sq;
  return CBA_OK;
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
  return CBA_OK;
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
    sn += step * step;
  }
  p->cost_new = p->fun(p->x_new.data(), p->f_new.data());
  out->cost = p->cost_new; out->step_norm = std::sqrt(sn); out->finite = std::isfinite(out->cost) ? 1 : 0; out->reserved = 0;
  if (!out->finite) out->cost = NAN;
  return CBA_OK;
}

int cba_trial(cba_problem* p, double alpha, double beta, cba_trial_info* out) { return cba_trial_ex(p, alpha, beta, nullptr, out); }

// the fused iteration of the device engine, emulated with the primitives above and the same scalar code (trf_math.h)
int cba_step_supported(cba_problem* p) { return p->cam_scaled ? 0 : 1; }
int cba_step(cba_problem* p, double radius_in, cba_step_info* out) {
  std::memset(out, 0, sizeof(*out));
  cba_linearize(p, &out->lin);
  const double gh_sq = out->lin.gh_sq, gh_norm = std::sqrt(gh_sq);
  const double radius = radius_in > 0.0 ? radius_in : (out->lin.x_scaled_norm > 0.0 ? out->lin.x_scaled_norm : 1.0);
  const double lam = -trf::min_quadratic_on_segment(0.5 * out->lin.jg_sq, -gh_sq, radius / gh_norm) / (radius * radius);
  out->lam = lam; out->radius = radius;
  cba_newton_step(p, lam, &out->newton);
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
  const int n = p->n;
  double p_sq = 0.0, ghp = 0.0, gh_sq = 0.0, w_sq = 0.0;
  for (int j = 0; j < n; ++j) { const double pj = p->s[j] * p->sinv[j], gh = p->g[j] / p->sinv[j]; p_sq += pj * pj; ghp += gh * pj; gh_sq += gh * gh; }
  for (int j = 0; j < n; ++j) { const double w = p->s[j] * p->sinv[j] - (ghp / gh_sq) * p->g[j] / p->sinv[j]; w_sq += w * w; }
  out->ok = 1; out->reserved = 0; out->p_sq = p_sq; out->gh_dot_p = ghp; out->w_sq = w_sq;
  return CBA_OK;
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
