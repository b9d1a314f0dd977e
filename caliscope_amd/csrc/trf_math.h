// Scalar pieces of the trust-region loop shared by the host driver (cba_solve.cpp) and the device-side step of the
// fused iteration (cba_kernels.h): the 1-D model minimum behind the regularisation rule and the 2-D subspace
// trust-region solve (scipy common.py:171-219).  Real arithmetic only, so the same code runs on both sides.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define TRF_HD __host__ __device__
#else
#define TRF_HD
#endif

namespace trf {

// min over t in [0, hi] of a t^2 + b t
TRF_HD inline double min_quadratic_on_segment(double a, double b, double hi) {
  double best = fmin(0.0, hi * (a * hi + b));
  if (a != 0.0) {
    const double t = -0.5 * b / a;
    if (t > 0.0 && t < hi) best = fmin(best, t * (a * t + b));
  }
  return best;
}

// Marquardt damping of an iteration = the model decrease along -g_h inside the trust region, per unit radius^2 (scipy trf.py:303-309 /
// :477-483; H_gg = |J_h g_h|^2 (+ g_h^T C g_h with bounds)).  Near a minimum it falls like the squared gradient norm — 1e-17 .. 1e-20 in
// the last iterations of a solve — while the reduced camera system of a gauge-free problem is singular without it: below ~n eps the
// damped matrix is not positive definite in double precision, the factorisation fails and the step has to be formed again (a whole
// Schur pass; measured on the 128-camera workload: every iteration after the fifth paid twice).  The floor is the value such a retry
// used anyway; against a step equation solved exactly it moves the step by O(floor / sigma_min^2) outside the gauge directions, orders
// of magnitude below the 1e-6 tolerance scipy's LSMR solves the same equation to.
constexpr double DAMPING_FLOOR = 1e-13;
TRF_HD inline double damping(double H_gg, double gh_sq, double radius) {
  const double lam = -min_quadratic_on_segment(0.5 * H_gg, -gh_sq, radius / sqrt(gh_sq)) / (radius * radius);
  return lam > DAMPING_FLOOR ? lam : DAMPING_FLOOR;
}

TRF_HD inline double poly_eval(const double* c, int deg, double t) {
  double p = c[0];
  for (int i = 1; i <= deg; ++i) p = p * t + c[i];
  return p;
}

// Real roots of c[0] t^DEG + ... + c[DEG], DEG <= 4, c[0] != 0: the real roots of the derivative split the line into
// monotone pieces; a sign change inside a piece is closed in by bisection (to the last bit) — no complex arithmetic.
// Callers rank the roots by a model value, so a duplicate is harmless.  A root of even multiplicity (the graph touches
// zero without crossing) is reported when a critical point itself is a root to rounding.
// The degree is a template parameter: the derivative's roots come from the instance one degree lower, so nothing recurses at run time.  (Until
// round 5 this was one function calling itself on `deg - 1`.  On the device a recursive function gives the kernel a DYNAMIC stack: the HIP runtime
// then reserves hipLimitStackSize = 1 KB per lane for every wave slot of the device — 512 MB of HBM on an MI355X, allocated inside the first
// launch of the one-workgroup step kernel and kept by the runtime for the life of the process; tools/device_memory_probe.py shows it.)
template <int DEG>
TRF_HD inline int real_roots_of_degree(const double* c, double* out) {
  static_assert(DEG >= 1 && DEG <= 4, "quartics at most");
  if constexpr (DEG == 1) {
    out[0] = -c[1] / c[0];
    return 1;
  } else if constexpr (DEG == 2) {
    const double disc = c[1] * c[1] - 4.0 * c[0] * c[2];
    if (disc < 0.0) return 0;
    const double q = -0.5 * (c[1] + copysign(sqrt(disc), c[1]));
    int n = 0;
    if (q != 0.0) { out[n++] = q / c[0]; out[n++] = c[2] / q; }
    else { out[n++] = 0.0; }
    return n;
  } else {
    double d[DEG], crit[DEG - 1];
    for (int i = 0; i < DEG; ++i) d[i] = c[i] * (DEG - i);
    int nc = real_roots_of_degree<DEG - 1>(d, crit);
    for (int i = 1; i < nc; ++i)  // sort (at most 3 values)
      for (int j = i; j > 0 && crit[j] < crit[j - 1]; --j) { const double t = crit[j]; crit[j] = crit[j - 1]; crit[j - 1] = t; }
    double bound = 0.0;  // Cauchy bound on |root|
    for (int i = 1; i <= DEG; ++i) bound = fmax(bound, fabs(c[i] / c[0]));
    bound += 1.0;
    double knots[DEG + 1];
    int nk = 0;
    knots[nk++] = -bound;
    for (int i = 0; i < nc; ++i)
      if (crit[i] > -bound && crit[i] < bound) knots[nk++] = crit[i];
    knots[nk++] = bound;
    int n = 0;
    double scale = 0.0;
    for (int i = 0; i <= DEG; ++i) scale = fmax(scale, fabs(c[i]));
    for (int k = 0; k + 1 < nk; ++k) {
      double a = knots[k], b = knots[k + 1];
      double fa = poly_eval(c, DEG, a), fb = poly_eval(c, DEG, b);
      if (fa == 0.0) { out[n++] = a; continue; }
      if (k + 2 == nk && fb == 0.0) { out[n++] = b; continue; }
      if ((fa < 0.0) == (fb < 0.0)) {
        // no crossing; a touching root at an interior knot shows as a tiny |f| there
        if (k > 0 && fabs(fa) <= 1e-14 * scale * fmax(1.0, pow(fabs(a), (double)DEG))) out[n++] = a;
        continue;
      }
      for (int it = 0; it < 200 && b - a > 0.0; ++it) {
        const double m = 0.5 * (a + b);
        if (m == a || m == b) break;
        const double fm = poly_eval(c, DEG, m);
        if (fm == 0.0) { a = b = m; break; }
        if ((fm < 0.0) == (fa < 0.0)) { a = m; fa = fm; } else { b = m; }
      }
      out[n++] = 0.5 * (a + b);
      if (n >= DEG) break;
    }
    return n;
  }
}

// c_in[0 .. n_coef - 1], n_coef <= 5, leading zeros allowed: the degree is what is left behind them
TRF_HD inline int real_roots(const double* c_in, int n_coef, double* out) {
  int lead = 0;
  while (lead < n_coef && c_in[lead] == 0.0) ++lead;
  switch (n_coef - 1 - lead) {
    case 4: return real_roots_of_degree<4>(c_in + lead, out);
    case 3: return real_roots_of_degree<3>(c_in + lead, out);
    case 2: return real_roots_of_degree<2>(c_in + lead, out);
    case 1: return real_roots_of_degree<1>(c_in + lead, out);
    default: return 0;
  }
}

// argmin 0.5 p^T B p + g^T p  s.t. ||p|| <= radius in two dimensions: the interior Newton point if B is positive
// definite and the point is inside, else the boundary p = radius (2t, 1 - t^2) / (1 + t^2) whose stationarity condition
// is a quartic in t; candidates (and t -> infinity) are ranked by model value.
TRF_HD inline void solve_subspace_2d(double b00, double b01, double b11, double g0, double g1, double radius, double* p) {
  if (b00 > 0.0) {
    const double schur = b11 - b01 * b01 / b00;
    if (schur > 0.0) {
      const double det = b00 * schur;
      const double p0 = -(b11 * g0 - b01 * g1) / det, p1 = -(b00 * g1 - b01 * g0) / det;
      if (p0 * p0 + p1 * p1 <= radius * radius) { p[0] = p0; p[1] = p1; return; }
    }
  }
  const double r2 = radius * radius;
  const double a = b00 * r2, b = b01 * r2, c = b11 * r2, d = g0 * radius, f = g1 * radius;
  const double coef[5] = {-b + d, 2.0 * (a - c + f), 6.0 * b, 2.0 * (-a + c + f), -b - d};
  double t[4];
  const int nt = real_roots(coef, 5, t);
  if (nt == 0) {  // degenerate quartic: steepest-descent boundary point
    const double n = hypot(g0, g1);
    p[0] = n > 0 ? -radius * g0 / n : 0.0;
    p[1] = n > 0 ? -radius * g1 / n : 0.0;
    return;
  }
  double best = INFINITY;
  for (int i = 0; i <= nt; ++i) {  // i == nt: t -> infinity, p = (0, -radius)
    double c0, c1;
    if (i < nt) { const double q = 1.0 + t[i] * t[i]; c0 = radius * 2.0 * t[i] / q; c1 = radius * (1.0 - t[i] * t[i]) / q; }
    else { c0 = 0.0; c1 = -radius; }
    const double val = 0.5 * (c0 * (b00 * c0 + b01 * c1) + c1 * (b01 * c0 + b11 * c1)) + g0 * c0 + g1 * c1;
    if (val < best) { best = val; p[0] = c0; p[1] = c1; }
  }
}

// The 2 x 2 model in the orthonormal basis q1 = g_h / ||g_h||, q2 = w / ||w|| of span{g_h, p}, from the step equation
// (H + lam I) p = -g_h (no pass over the observations); H_gg = ||J_h g_h||^2 (+ g_h^T C g_h with bounds).
TRF_HD inline void subspace_model(double H_gg, double gh_sq, double lam, double gh_dot_p, double p_sq, double w_sq, double* b00,
                                  double* b01, double* b11) {
  const double c = gh_dot_p / gh_sq, gh_norm = sqrt(gh_sq), w_norm = sqrt(w_sq);
  const double H_gp = -gh_sq - lam * gh_dot_p, H_pp = -gh_dot_p - lam * p_sq;
  *b00 = H_gg / gh_sq;
  *b01 = (H_gp - c * H_gg) / (gh_norm * w_norm);
  *b11 = (H_pp - 2.0 * c * H_gp + c * c * H_gg) / w_sq;
}

}  // namespace trf
