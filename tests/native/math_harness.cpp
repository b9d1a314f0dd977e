// CPU harness around caliscope_amd/csrc/ba_math.h — TEST INFRASTRUCTURE (built by g++ in
// tests/test_native_math.py).  It lets the non-GPU test-suite compare the exact per-observation
// arithmetic the HIP kernels inline (projection, Jacobian blocks, robust-loss scaling, 3x3 Cholesky)
// with the numpy oracle.  It is not a CPU fallback: nothing in caliscope_amd/ loads it.
#include "ba_math.h"

// cam_prepare reads MAX_NC entries of the camera's slice (the intrinsic ones are SELECTED, not branched on): callers hand over nparams
static void prepare(const double* x_cam, const double* cconst, int model, int nparams, cba::CamTab* tab) {
  double xc[cba::MAX_NC] = {0};
  for (int i = 0; i < nparams && i < cba::MAX_NC; ++i) xc[i] = x_cam[i];
  cba::cam_prepare(xc, cconst, model, nparams, tab);
}

extern "C" {

// A_out: [2][9] row-major (unused columns zero), B_out: [2][3]
void mh_project_full(const double* x_cam, const double* cconst, int model, int nparams, const double* X,
                     const double* uv, double* e_out, double* A_out, double* B_out) {
  cba::CamTab tab;
  prepare(x_cam, cconst, model, nparams, &tab);
  double A[2][cba::MAX_NC] = {{0}}, B[2][3];
  cba::project_full(tab, X[0], X[1], X[2], uv[0], uv[1], e_out, A, B);
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < cba::MAX_NC; ++c) A_out[r * cba::MAX_NC + c] = (c < nparams) ? A[r][c] : 0.0;
    for (int c = 0; c < 3; ++c) B_out[r * 3 + c] = B[r][c];
  }
}

// The factored form the per-observation kernels use since round 6 (ba_math.h project_factors): A is rebuilt from the factors — rotation columns
// (Y x G_r)^T J_l, translation columns G_r, intrinsic columns A_intr — so that the test can hold it against project_full and the oracle.
void mh_project_factors(const double* x_cam, const double* cconst, int model, int nparams, const double* X,
                        const double* uv, double* e_out, double* A_out, double* B_out) {
  cba::CamTab tab;
  prepare(x_cam, cconst, model, nparams, &tab);
  double G[2][3], Yr[3], Aint[2][3], B[2][3];
  cba::project_factors(tab, X[0], X[1], X[2], uv[0], uv[1], e_out, G, Yr, Aint, B);
  for (int r = 0; r < 2; ++r) {
    const double p[3] = {Yr[1] * G[r][2] - Yr[2] * G[r][1], Yr[2] * G[r][0] - Yr[0] * G[r][2], Yr[0] * G[r][1] - Yr[1] * G[r][0]};
    for (int c = 0; c < cba::MAX_NC; ++c) A_out[r * cba::MAX_NC + c] = 0.0;
    for (int j = 0; j < 3; ++j) A_out[r * cba::MAX_NC + j] = tab.Jl[j] * p[0] + tab.Jl[3 + j] * p[1] + tab.Jl[6 + j] * p[2];
    for (int j = 0; j < 3; ++j) A_out[r * cba::MAX_NC + 3 + j] = G[r][j];
    for (int j = 0; j < 3 && nparams == 9; ++j) A_out[r * cba::MAX_NC + 6 + j] = Aint[r][j];
    for (int c = 0; c < 3; ++c) B_out[r * 3 + c] = B[r][c];
  }
}

void mh_project_residual(const double* x_cam, const double* cconst, int model, int nparams, const double* X,
                         const double* uv, double* e_out) {
  cba::CamTab tab;
  prepare(x_cam, cconst, model, nparams, &tab);
  cba::project_residual(tab, X[0], X[1], X[2], uv[0], uv[1], e_out);
}

void mh_cam_table(const double* x_cam, const double* cconst, int model, int nparams, double* out48) {
  cba::CamTab tab;
  prepare(x_cam, cconst, model, nparams, &tab);
  const double* p = reinterpret_cast<const double*>(&tab);
  for (int i = 0; i < cba::CAMTAB_DOUBLES; ++i) out48[i] = p[i];
}

// out: rho0*fs^2, row_scale, r_scaled, cost_only
void mh_robust(int loss, double f_scale, double r, double* out) {
  double rs, rr;
  out[0] = cba::robust_one(loss, f_scale, r, &rs, &rr);
  out[1] = rs;
  out[2] = rr;
  out[3] = cba::robust_cost_one(loss, f_scale, r);
}

// solves V x = b via chol3; returns 1 on success
int mh_chol3_solve(const double* v6, const double* b, double* x) {
  double L[6], y[3];
  if (!cba::chol3(v6, L)) return 0;
  cba::chol3_fwd(L, b, y);
  cba::chol3_bwd(L, y, x);
  return 1;
}
}
