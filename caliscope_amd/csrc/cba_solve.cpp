// cba_solve: the whole trust-region solve behind one C call.
//
// Host-side driver written against the public primitives of include/caliscope_ba.h only (cba_begin, cba_linearize,
// cba_newton_step, cba_subspace_gram, cba_trial, cba_accept): a dozen scalars per iteration cross from the device,
// nothing O(n) lives here.  It is the loop of scipy's `trf_no_bounds` (scipy 1.15.3 optimize/_lsq/trf.py:401-560) —
// what the reference's `least_squares(..., method="trf", x_scale="jac")` call (core/capture_volume.py:387-411) runs
// — with the same regularisation rule, 2-D subspace span{g_h, p}, radius update, termination codes and nfev
// accounting, and with the LSMR step replaced by the exact Marquardt-damped step of cba_newton_step.
// caliscope_amd/trf.py is the same loop in Python (used with the numpy engine in the CPU tests); the two are kept in
// step by tests/test_gpu_parity.py::test_cba_solve_matches_python_driver.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/caliscope_ba.h"
#include "trf_math.h"

namespace {

// ||w||^2 / ||p||^2 below which the subspace model is built from explicit J.v products (trf.py, same constant)
constexpr double SUBSPACE_EXPLICIT_BELOW = 1e-6;

using trf::solve_subspace_2d;

int termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
  const bool f_ok = dF < ftol * F && ratio > 0.25;
  const bool x_ok = dx_norm < xtol * (xtol + x_norm);
  if (f_ok && x_ok) return 4;
  if (f_ok) return 2;
  if (x_ok) return 3;
  return -100;  // none
}


// ---- bounded camera block (scipy trf_bounds, trf.py:205-398; helpers of common.py) -------------------------------
struct CamBlock {
  int n = 0;
  const double *lb = nullptr, *ub = nullptr;
  std::vector<double> x, g, sinv, s, mult, diag_h, d, gh, p, ph, r, step, x_new;
  std::vector<char> hit;
  void resize(int ncp) {
    n = ncp;
    for (std::vector<double>* v : {&x, &g, &sinv, &s, &mult, &diag_h, &d, &gh, &p, &ph, &r, &step, &x_new}) v->assign(n, 0.0);
    hit.assign(n, 0);
  }
};

// common.py step_size_to_bound: largest t with lb <= x + t s <= ub; hit[i] marks the variables that reach a bound at t
double step_size_to_bound(const CamBlock& cb, const double* x, const double* s, char* hit) {
  double best = INFINITY;
  for (int i = 0; i < cb.n; ++i)
    if (s[i] != 0.0) best = std::min(best, std::max((cb.lb[i] - x[i]) / s[i], (cb.ub[i] - x[i]) / s[i]));
  if (hit)
    for (int i = 0; i < cb.n; ++i) hit[i] = (s[i] != 0.0 && std::max((cb.lb[i] - x[i]) / s[i], (cb.ub[i] - x[i]) / s[i]) == best) ? 1 : 0;
  return best;
}

// common.py minimize_quadratic_1d: min over t in [lo, hi] of a t^2 + b t + c
void minimize_quadratic_1d(double a, double b, double lo, double hi, double c, double* t_best, double* y_best) {
  double t[3] = {lo, hi, 0.0};
  int nt = 2;
  if (a != 0.0) {
    const double ext = -0.5 * b / a;
    if (lo < ext && ext < hi) t[nt++] = ext;
  }
  *y_best = INFINITY;
  for (int i = 0; i < nt; ++i) {
    const double y = t[i] * (a * t[i] + b) + c;
    if (y < *y_best) { *y_best = y; *t_best = t[i]; }
  }
}

// CBA_SOLVE_TRACE=1: host wall-clock per primitive (enqueue + whatever the call waits for), printed to stderr at the end of the solve
struct CallTrace {
  struct Row { const char* name; double s; long n; double longest; };
  bool on = std::getenv("CBA_SOLVE_TRACE") != nullptr;
  Row rows[24];
  int used = 0;
  template <class F> int run(const char* name, F&& f) {
    if (!on) return f();
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = f();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int i = 0;
    while (i < used && std::strcmp(rows[i].name, name) != 0) ++i;
    if (i == used) { if (used == 24) return rc; rows[used++] = Row{name, 0.0, 0, 0.0}; }
    rows[i].s += dt; ++rows[i].n;
    if (dt > rows[i].longest) rows[i].longest = dt;
    return rc;
  }
  void print(double total_s, long iterations) const {
    if (!on) return;
    double sum = 0.0;
    for (int i = 0; i < used; ++i) sum += rows[i].s;
    std::fprintf(stderr, "cba_solve trace: %ld iterations, %.3f ms, of which %.3f ms inside the primitives\n", iterations, total_s * 1e3, sum * 1e3);
    for (int i = 0; i < used; ++i) std::fprintf(stderr, "  %-22s %5ld calls %9.3f ms  %8.1f us each, longest %8.1f us\n", rows[i].name, rows[i].n, rows[i].s * 1e3, rows[i].s * 1e6 / rows[i].n, rows[i].longest * 1e6);
  }
};

}  // namespace

extern "C" int cba_solve(cba_problem* p, const double* x0, const cba_solve_options* opt_in, double* x_out, cba_result* out) {
  if (!p || !out) return cba_set_error(CBA_ERR_INVALID, "cba_solve: null argument");
  cba_solve_options opt;
  if (opt_in) opt = *opt_in;
  else { opt.ftol = opt.xtol = opt.gtol = 1e-8; opt.max_nfev = 0; opt.lb = opt.ub = nullptr; opt.verbose = 0; opt.max_damping_retries = 12; }
  const double eps = 2.220446049250313e-16;
  if (opt.ftol < eps && opt.xtol < eps && opt.gtol < eps)
    return cba_set_error(CBA_ERR_INVALID, "cba_solve: at least one of the tolerances must be higher than machine epsilon");
  cba_info info;
  int rc = cba_get_info(p, &info);
  if (rc) return rc;
  const long n_params = info.n_params;
  const int ncp = info.n_cam_params;
  const long max_nfev = opt.max_nfev > 0 ? opt.max_nfev : 100 * n_params;  // scipy: max_nfev=None -> 100 n
  const int max_retries = opt.max_damping_retries > 0 ? opt.max_damping_retries : 12;
  bool bounded = false;  // scipy: trf_no_bounds when every bound is infinite (trf.py:116-126)
  if (opt.lb && opt.ub)
    for (int i = 0; i < ncp; ++i) {
      if (!(opt.lb[i] < opt.ub[i])) return cba_set_error(CBA_ERR_INVALID, "cba_solve: each lower bound must be strictly less than each upper bound");
      bounded = bounded || std::isfinite(opt.lb[i]) || std::isfinite(opt.ub[i]);
    }
  CamBlock cb;
  if (bounded) { cb.resize(ncp); cb.lb = opt.lb; cb.ub = opt.ub; }
  const auto t_begin = std::chrono::steady_clock::now();
  CallTrace calls;

  auto not_finite_at_x0 = [&](double c) {  // scipy raises "Residuals are not finite in the initial point": status -1, nothing solved
    out->status = -1; out->reserved = 0; out->nfev = 1; out->njev = 0; out->n_iterations = 0; out->cost = c; out->optimality = NAN;
    out->t_rejected_s = 0.0; out->n_rejected_timed = 0;
    out->t_total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    return CBA_OK;
  };
  double cost = NAN;
  // Without bounds the evaluation of x0 is left to the first linearisation (its build pass computes the cost anyway): one
  // pass over the observations and one wait less per solve.
  const bool deferred = !bounded;
  if (deferred) rc = calls.run("begin_deferred", [&] { return cba_begin_deferred(p, x0); });
  else rc = x0 ? calls.run("begin", [&] { return cba_begin(p, x0, &cost); }) : calls.run("restart", [&] { return cba_restart(p, &cost); });
  if (rc) return rc;
  if (!deferred && !std::isfinite(cost)) return not_finite_at_x0(cost);
  if (bounded) {  // scipy: "`x0` is infeasible." (callers nudge on-bound entries inside first, least_squares.py:820-821)
    if ((rc = calls.run("get_camera_params", [&] { return cba_get_camera_params(p, CBA_VEC_X, cb.x.data()); }))) return rc;
    for (int i = 0; i < ncp; ++i)
      if (!(cb.x[i] > cb.lb[i] && cb.x[i] < cb.ub[i])) return cba_set_error(CBA_ERR_INVALID, "cba_solve: x0 is not strictly inside the bounds");
  }
  long nfev = 1, njev = 1, iteration = 0;
  double t_rejected = 0.0;   // wall time of the separate trial evaluations that were rejected (bench.py: ms per rejected trial)
  long n_rejected_timed = 0;
  int n_truncated = 0, n_reflected = 0, n_gradient = 0;  // which candidate select_step took when the step left the bounds
  // Fused iterations (cba_step): linearisation, damping, damped step, subspace step and first trial behind ONE host
  // synchronisation; the trial is evaluated by a build pass, so an accepted step needs no further pass.  A rejected
  // first trial costs a build instead of a cost pass: after one, the next iteration goes through the primitives.
  // Bounded solves (round 5): cba_set_bounds hands the box to the device; where the engine can, cba_step then runs the bounded iteration behind
  // one synchronisation too (Coleman-Li scaling on the device; a first trial point that leaves the box comes back as need_host = 2).
  bool bounds_on_device = false;
  if (bounded) {
    const int rb = cba_set_bounds(p, cb.lb, cb.ub);
    if (rb < 0) return rb;
    bounds_on_device = rb == 1;
  } else {
    (void)cba_set_bounds(p, nullptr, nullptr);
  }
  bool fused = (!bounded || bounds_on_device) && cba_step_supported(p);
  bool fuse_next = fused;
  int n_outside = 0;  // bounded fused steps whose first trial point left the box
  cba_linearization lin;
  cba_step_info si;
  bool lin_valid = false;    // `lin` describes the current x
  double radius = NAN;  // set from ||x0 * scale_inv (/ sqrt(v))|| on the first pass
  int status = -100;
  double g_norm = NAN, step_norm = NAN, actual = NAN;
  if (opt.verbose == 2) std::printf("%15s%15s%15s%15s%15s%15s\n", "Iteration", "Total nfev", "Cost", "Cost reduction", "Step norm", "Optimality");

  for (;;) {
    double C_gg = 0.0;  // g_h^T C g_h, C = diag_h: the Coleman-Li term of the model Hessian (zero without bounds)
    bool have_step = false;  // `si` holds this iteration's damped step and first trial
    if (!lin_valid) {
      if (status == -100 && fuse_next && nfev < max_nfev) {
        if ((rc = calls.run("step", [&] { return cba_step(p, std::isnan(radius) ? -1.0 : radius, &si); }))) return rc;
        lin = si.lin; have_step = true;
      } else if (bounded) {
        fuse_next = false;  // (THIS iteration runs on the primitives — the two routes keep the same scaling state; the first trial decides below
                            // whether the next one is fused again, and after two trial points on a bound the solve stays here: n_outside)
        if ((rc = calls.run("linearize_build", [&] { return cba_linearize_build(p); }))) return rc;  // the scalars follow from cba_set_camera_scaling below
      } else if ((rc = calls.run("linearize", [&] { return cba_linearize(p, &lin); }))) return rc;
      lin_valid = true;
      if (std::isnan(cost)) {  // deferred begin: this was the evaluation of x0
        cost = lin.cost;
        if (!std::isfinite(cost)) return not_finite_at_x0(cost);
      }
    }
    if (bounded && have_step && si.need_host == 2) {
      // the fused step's first trial point was not strictly inside the box: this iteration again, through the primitives (select_step).  A solve
      // whose solution lies ON a bound would pay a wasted Schur pass per iteration: after the second time it stays on the primitives
      lin_valid = false; fuse_next = false;
      if (++n_outside >= 2) fused = false;
      continue;
    }
    if (bounded) {
      // Coleman-Li scaling vector of the camera block (common.py CL_scaling_vector) and what follows from it
      if (have_step) {  // the camera blocks came with the step's packet: x, g, the Jacobi scale, the damped step
        if ((rc = cba_step_camera_state(p, cb.x.data(), cb.g.data(), cb.sinv.data(), cb.s.data()))) return rc;
      } else if ((rc = calls.run("get_camera_state", [&] { return cba_get_camera_state(p, cb.x.data(), cb.g.data(), cb.sinv.data()); }))) return rc;  // sinv: the Jacobi scale (restored by the linearisation)
      double gv_max = 0.0;
      for (int i = 0; i < ncp; ++i) {
        double v = 1.0, dv = 0.0;
        if (cb.g[i] < 0 && std::isfinite(cb.ub[i])) { v = cb.ub[i] - cb.x[i]; dv = -1.0; }
        else if (cb.g[i] > 0 && std::isfinite(cb.lb[i])) { v = cb.x[i] - cb.lb[i]; dv = 1.0; }
        gv_max = std::max(gv_max, std::fabs(cb.g[i] * v));
        if (dv != 0.0) v *= cb.sinv[i];                       // v[dv != 0] *= scale_inv
        cb.mult[i] = 1.0 / std::sqrt(v);                      // d = sqrt(v) * scale  ->  effective scale_inv = scale_inv / sqrt(v)
        cb.diag_h[i] = cb.g[i] * dv / cb.sinv[i];             // diag_h = g * dv * scale  (>= 0)
        cb.d[i] = 1.0 / (cb.sinv[i] * cb.mult[i]);
        cb.gh[i] = cb.g[i] * cb.d[i];
        C_gg += cb.diag_h[i] * cb.gh[i] * cb.gh[i];
      }
      if (!have_step && (rc = calls.run("set_camera_scaling", [&] { return cba_set_camera_scaling(p, cb.mult.data(), cb.diag_h.data(), &lin); }))) return rc;
      g_norm = std::max(gv_max, lin.g_norm_inf);               // ||g * v||_inf: lin.g_norm_inf covers the point block (v = 1; a fused step: both)
    } else {
      g_norm = lin.g_norm_inf;
    }
    if (std::isnan(radius)) radius = lin.x_scaled_norm > 0 ? lin.x_scaled_norm : 1.0;
    if (g_norm < opt.gtol) status = 1;
    if (opt.verbose == 2) {
      std::printf("%15ld%15ld%15.4e", iteration, nfev, cost);
      if (std::isnan(actual)) std::printf("%15s%15s", "", ""); else std::printf("%15.2e%15.2e", actual, step_norm);
      std::printf("%15.2e\n", g_norm);
    }
    if (status != -100 || nfev >= max_nfev) break;
    const double theta = std::max(0.995, 1.0 - g_norm);  // how far a step stops short of a bound (trf.py:319)

    const double gh_sq = lin.gh_sq, gh_norm = std::sqrt(gh_sq);
    const double H_gg = lin.jg_sq + C_gg;
    // regularisation = model decrease along -g_h inside the region, per unit radius^2 (trf.py:303-309 / :477-483)
    double lam = trf::damping(H_gg, gh_sq, radius);
    cba_newton_info st;
    if (have_step) {
      lam = si.lam; st = si.newton;
      // collinear step: the explicit-model branch below needs ||w||^2 measured, not derived (cba_step's shortcut)
      if (si.need_host && st.ok && (rc = calls.run("refresh_step_scalars", [&] { return cba_refresh_step_scalars(p, &st); }))) return rc;
    } else if ((rc = calls.run("newton_step", [&] { return cba_newton_step(p, lam, &st); }))) return rc;
    bool first_trial_ready = have_step && st.ok && !si.need_host;
    bool refetch_step = !have_step;  // bounded: the camera block of the damped step is needed on the host (a fused step's came with the packet)
    if (calls.on) std::fprintf(stderr, "  iteration %ld: lam %.3e radius %.3e ok %d%s\n", iteration, lam, radius, st.ok, have_step ? " (fused)" : "");
    for (int retries = 0; !st.ok;) {
      // positive definite in exact arithmetic; rounding on a gauge-singular problem can still break the factorisation
      if (++retries > max_retries) return cba_set_error(CBA_ERR_NUMERIC, "cba_solve: normal equations could not be factorised even with heavy damping");
      lam = std::max(lam * 10.0, 1e-14 * std::pow(10.0, retries));
      if ((rc = calls.run("newton_step", [&] { return cba_newton_step(p, lam, &st); }))) return rc;
      refetch_step = true;  // (the packet's step is the one whose factorisation failed: possibly not finite, and not the one on the device now)
      if (calls.on) std::fprintf(stderr, "    retry %d: lam %.3e ok %d\n", retries, lam, st.ok);
    }
    if (bounded && refetch_step && (rc = calls.run("get_camera_params", [&] { return cba_get_camera_params(p, CBA_VEC_STEP, cb.s.data()); }))) return rc;
    // orthonormal basis of span{g_h, p}: q1 = g_h / ||g_h||, q2 = w / ||w||, w = p - c g_h
    const double c = st.gh_dot_p / gh_sq, w_sq = st.w_sq;
    const bool two_d = w_sq > 0.0 && w_sq > 1e-30 * st.p_sq;
    const double w_norm = two_d ? std::sqrt(w_sq) : 1.0;
    double b00, b01 = 0.0, b11;
    if (!two_d) {
      b00 = H_gg / gh_sq; b11 = 1.0;
    } else if (w_sq > SUBSPACE_EXPLICIT_BELOW * st.p_sq) {
      // from the step equation (H + C + lam I) p = -g_h: no further pass over the observations
      trf::subspace_model(H_gg, gh_sq, lam, st.gh_dot_p, st.p_sq, w_sq, &b00, &b01, &b11);
    } else {
      // p nearly collinear with g_h (heavy damping): the identities cancel, form J_h q1, J_h q2 explicitly
      double gram[3];
      if ((rc = calls.run("subspace_gram", [&] { return cba_subspace_gram(p, 1.0 / gh_norm, 0.0, -c / w_norm, 1.0 / w_norm, gram); }))) return rc;
      b00 = gram[0]; b01 = gram[1]; b11 = gram[2];
      for (int i = 0; i < cb.n; ++i) {  // + S^T C S
        const double q1 = cb.gh[i] / gh_norm, q2 = (cb.s[i] / cb.d[i] - c * cb.gh[i]) / w_norm;
        b00 += cb.diag_h[i] * q1 * q1; b01 += cb.diag_h[i] * q1 * q2; b11 += cb.diag_h[i] * q2 * q2;
      }
    }
    actual = -1.0;
    double cost_new = cost;
    bool first_pass = true;
    while (actual <= 0 && nfev < max_nfev) {
      double pS[2];
      if (first_pass && first_trial_ready) { pS[0] = si.p_s[0]; pS[1] = si.p_s[1]; }  // decided on the device (same code)
      else {
        solve_subspace_2d(b00, b01, b11, gh_norm, 0.0, radius, pS);
        if (!two_d) pS[1] = 0.0;
      }
      const double quad_p = 0.5 * (pS[0] * (b00 * pS[0] + b01 * pS[1]) + pS[1] * (b01 * pS[0] + b11 * pS[1]));  // 0.5 p^T (H + C) p
      const double lin_p = gh_norm * pS[0];                                                                   // g_h^T p
      // step_h = pS0 q1 + pS1 q2 = alpha g_h + beta p;  step = alpha g / scale_inv^2 + beta s
      const double beta = two_d ? pS[1] / w_norm : 0.0;
      const double alpha = pS[0] / gh_norm - beta * c;
      double predicted = -(quad_p + lin_p), step_h_norm = std::hypot(pS[0], pS[1]);
      cba_trial_info tr;
      const auto t_trial = std::chrono::steady_clock::now();
      bool own_call = true;  // this trial is evaluated by a call of its own (cba_step brought the first one)
      if (first_pass && first_trial_ready) {
        own_call = false;
        tr = si.trial; predicted = si.predicted;  // the trial cba_step already evaluated
      } else if (!bounded) {
        if ((rc = calls.run("trial", [&] { return cba_trial(p, alpha, beta, &tr); }))) return rc;
      } else {
        // select_step (trf.py:129-202): the trust-region step if it stays inside the bounds, else the best of the step
        // truncated at the bound, its reflection from the bound, and the scaled-gradient step
        double pt_alpha = alpha, pt_beta = beta;  // coefficients of the point block of the chosen step
        bool inside = true;
        for (int i = 0; i < ncp; ++i) {
          cb.p[i] = alpha * cb.g[i] * cb.d[i] * cb.d[i] + beta * cb.s[i];
          cb.step[i] = cb.p[i];
          const double xn = cb.x[i] + cb.p[i];
          inside = inside && xn >= cb.lb[i] && xn <= cb.ub[i];
        }
        if (!inside) {
          const double p_norm2 = pS[0] * pS[0] + pS[1] * pS[1];
          const double p_stride = step_size_to_bound(cb, cb.x.data(), cb.p.data(), cb.hit.data());
          double hit_ph2 = 0.0, hit_gh_ph = 0.0, C_pp = 0.0, C_hit = 0.0;
          for (int i = 0; i < ncp; ++i) {
            cb.ph[i] = cb.p[i] / cb.d[i];
            cb.r[i] = cb.hit[i] ? -cb.p[i] : cb.p[i];  // reflected direction, x-space camera block
            C_pp += cb.diag_h[i] * cb.ph[i] * cb.ph[i];
            if (cb.hit[i]) { hit_ph2 += cb.ph[i] * cb.ph[i]; hit_gh_ph += cb.gh[i] * cb.ph[i]; C_hit += cb.diag_h[i] * cb.ph[i] * cb.ph[i]; }
          }
          const double ph_dot_rh = p_norm2 - 2.0 * hit_ph2, C_pr = C_pp - 2.0 * C_hit, C_rr = C_pp;
          // the reflected ray leaves either the feasible region or the trust region first (intersect_trust_region)
          double to_tr = 0.0;
          {
            const double a = p_norm2, b = p_stride * ph_dot_rh, cq = p_stride * p_stride * p_norm2 - radius * radius;
            const double disc = std::sqrt(std::max(0.0, b * b - a * cq));
            const double q = -(b + std::copysign(disc, b));
            if (a > 0.0 && q != 0.0) to_tr = std::max(q / a, cq / q);
          }
          for (int i = 0; i < ncp; ++i) cb.x_new[i] = cb.x[i] + p_stride * cb.p[i];  // x on the bound
          const double to_bound = step_size_to_bound(cb, cb.x_new.data(), cb.r.data(), nullptr);
          double r_stride = std::min(to_bound, to_tr), r_lo, r_hi, r_value = INFINITY;
          if (r_stride > 0) { r_lo = (1.0 - theta) * p_stride / r_stride; r_hi = (r_stride == to_bound) ? theta * to_bound : to_tr; }
          else { r_lo = 0.0; r_hi = -1.0; }
          if (r_lo <= r_hi) {
            // quadratic along r_h from s0 = p_stride p_h (build_quadratic_1d with s0): needs J_h p_h, J_h r_h
            double gram[3];
            if ((rc = calls.run("subspace_gram_ex", [&] { return cba_subspace_gram_ex(p, alpha, beta, nullptr, alpha, beta, cb.r.data(), gram); }))) return rc;
            const double qa = 0.5 * (gram[2] + C_rr);
            const double qb = (lin_p - 2.0 * hit_gh_ph) + p_stride * (gram[1] + C_pr);
            const double qc = 0.5 * p_stride * p_stride * (gram[0] + C_pp) + p_stride * lin_p;
            minimize_quadratic_1d(qa, qb, r_lo, r_hi, qc, &r_stride, &r_value);
          }
          const double kappa = theta * p_stride;  // truncated step, pulled strictly inside
          const double p_value = kappa * kappa * quad_p + kappa * lin_p;
          // scaled anti-gradient
          for (int i = 0; i < ncp; ++i) cb.x_new[i] = -cb.g[i] * cb.d[i] * cb.d[i];
          const double ag_to_tr = radius / gh_norm, ag_to_bound = step_size_to_bound(cb, cb.x.data(), cb.x_new.data(), nullptr);
          double ag_stride, ag_value;
          minimize_quadratic_1d(0.5 * H_gg, -gh_sq, 0.0, ag_to_bound < ag_to_tr ? theta * ag_to_bound : ag_to_tr, 0.0, &ag_stride, &ag_value);
          if (p_value < r_value && p_value < ag_value) {
            ++n_truncated;
            pt_alpha = kappa * alpha; pt_beta = kappa * beta;
            for (int i = 0; i < ncp; ++i) cb.step[i] = kappa * cb.p[i];
            predicted = -p_value; step_h_norm = kappa * std::sqrt(p_norm2);
          } else if (r_value < p_value && r_value < ag_value) {
            ++n_reflected;
            pt_alpha = (p_stride + r_stride) * alpha; pt_beta = (p_stride + r_stride) * beta;
            for (int i = 0; i < ncp; ++i) cb.step[i] = p_stride * cb.p[i] + r_stride * cb.r[i];
            predicted = -r_value;
            step_h_norm = std::sqrt(std::max(0.0, (p_stride * p_stride + r_stride * r_stride) * p_norm2 + 2.0 * p_stride * r_stride * ph_dot_rh));
          } else {
            ++n_gradient;
            pt_alpha = -ag_stride; pt_beta = 0.0;
            for (int i = 0; i < ncp; ++i) cb.step[i] = ag_stride * cb.x_new[i];
            predicted = -ag_value; step_h_norm = ag_stride * gh_norm;
          }
        }
        for (int i = 0; i < ncp; ++i) {  // make_strictly_feasible(x + step, rstep = 0)
          double xn = cb.x[i] + cb.step[i];
          if (xn <= cb.lb[i]) xn = std::nextafter(cb.lb[i], cb.ub[i]);
          else if (xn >= cb.ub[i]) xn = std::nextafter(cb.ub[i], cb.lb[i]);
          cb.x_new[i] = xn;
        }
        if ((rc = calls.run("trial_ex", [&] { return cba_trial_ex(p, pt_alpha, pt_beta, cb.x_new.data(), &tr); }))) return rc;
      }
      ++nfev;
      const bool was_first = first_pass;
      first_pass = false;
      const double dt_trial = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_trial).count();
      if (!tr.finite) { radius = 0.25 * step_h_norm; if (was_first) fuse_next = false; if (own_call) { t_rejected += dt_trial; ++n_rejected_timed; } continue; }
      cost_new = tr.cost;
      actual = cost - cost_new;
      if (own_call && !(actual > 0)) { t_rejected += dt_trial; ++n_rejected_timed; }
      // speculate again only after an accepted first trial of a well-conditioned subspace (a collinear step needs the
      // explicit J.v model, which cba_step leaves to the host)
      if (was_first) fuse_next = fused && actual > 0 && w_sq > 1e-2 * st.p_sq;  // (cba_step needs w_sq > 1e-3 p_sq: hysteresis)
      double ratio;
      if (predicted > 0) ratio = actual / predicted;
      else if (predicted == 0 && actual == 0) ratio = 1.0;
      else ratio = 0.0;
      double radius_new = radius;
      if (ratio < 0.25) radius_new = 0.25 * step_h_norm;
      else if (ratio > 0.75 && step_h_norm > 0.95 * radius) radius_new = 2.0 * radius;
      step_norm = tr.step_norm;
      status = termination(actual, cost, step_norm, lin.x_norm, ratio, opt.ftol, opt.xtol);
      if (status != -100) break;
      radius = radius_new;
    }
    if (actual > 0) {
      if ((rc = calls.run("accept", [&] { return cba_accept(p); }))) return rc;
      cost = cost_new;
      lin_valid = false;  // linearised at the top of the next pass (by cba_step when fused)
      ++njev;
    } else {
      step_norm = 0.0; actual = 0.0;
    }
    ++iteration;
  }
  if (status == -100) status = 0;
  if (x_out && (rc = calls.run("get_vector", [&] { return cba_get_vector(p, CBA_VEC_X, x_out); }))) return rc;
  out->status = std::isfinite(cost) ? status : -1;
  out->reserved = std::min(n_truncated, 1023) | (std::min(n_reflected, 1023) << 10) | (std::min(n_gradient, 1023) << 20);
  out->nfev = nfev; out->njev = njev; out->n_iterations = iteration;
  out->cost = cost; out->optimality = g_norm;
  out->t_total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  out->t_rejected_s = t_rejected; out->n_rejected_timed = n_rejected_timed;
  calls.print(out->t_total_s, iteration);
  return CBA_OK;
}
