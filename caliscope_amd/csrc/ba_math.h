// Per-observation arithmetic of the bundle-adjustment hot path (host + device inline functions).
//
// What the reference obtains from cv2.Rodrigues / cv2.projectPoints / cv2.fisheye.projectPoints
// and then slices and rescales in core/reprojection.py:96-110 (residuals) and :163-187 (Jacobian
// blocks) is computed here per observation, in registers, with nothing stored:
//
//   e  (2)      residual  (proj - obs) / fx_initial                       reprojection.py:108
//   A  (2 x nc) camera block [d/drvec(3), d/dtvec(3) (, d/ds, d/dk1, d/dk2)] / fx_initial   :177-184
//   B  (2 x 3)  point block  (d proj/d tvec) @ R / fx_initial             :186-187
//
// Formulas: SURVEY.md Appendix A.1-A.3.  The rotation derivative uses the left Jacobian of SO(3):
// d(R X)/dr = -[R X]x J_l(r), so column j is  J_l[:,j] x (R X)  (J_l is a per-camera constant).
// The file is compiled by hipcc into the kernels and by g++ into the CPU math harness that the
// non-GPU tests compare with the oracle (tests/native/).
#pragma once
#include <cstddef>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define CBA_HD __host__ __device__ __forceinline__
#else
#define CBA_HD inline
#endif

namespace cba {

constexpr int MODEL_PINHOLE_BC5 = 0;
constexpr int MODEL_FISHEYE4 = 1;
constexpr int CAM_CONST_STRIDE = 12;  // fx0 fy0 cx cy d0..d4 pad pad pad
constexpr int MAX_NC = 9;
constexpr double EPS_F64 = 2.220446049250313e-16;

// 1/sqrt(x): v_rsq_f64 seed + two Newton steps on the device (an IEEE sqrt followed by an IEEE divide is ~35 FP64
// instructions), the plain quotient on the host.
CBA_HD double inv_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
#else
  return 1.0 / sqrt(x);
#endif
}

enum Loss : int { LOSS_LINEAR = 0, LOSS_HUBER = 1, LOSS_SOFT_L1 = 2, LOSS_CAUCHY = 3, LOSS_ARCTAN = 4 };

// Per-camera constants of one evaluation point (48 doubles; staged in LDS by the kernels).
struct CamTab {
  double R[9];    // row-major world->camera rotation
  double t[3];
  double Jl[9];   // left Jacobian of SO(3) at rvec, row-major
  double fx, fy, cx, cy;
  double d[5];    // k1 k2 p1 p2 k3 | k1..k4 0
  double fx0, fy0, inv_fx0;
  double model;   // 0 pinhole, 1 fisheye   (stored as double to keep the table homogeneous)
  double nparams; // 6 or 9
  double pad[13]; // pad[0]: index of the camera's first parameter in the parameter vector (as a double: the table is homogeneous); the rest unused
};
static_assert(sizeof(CamTab) == 48 * sizeof(double), "CamTab must be 48 doubles");
static_assert(offsetof(CamTab, pad) == 35 * sizeof(double), "35 doubles + pad[0] are live (CAMTAB_LIVE = 36 in cba_kernels.h)");
constexpr int CAMTAB_DOUBLES = 48;

// x_cam: this camera's slice of the parameter vector in an array of MAX_NC entries (all nine are read; those behind nparams are not used);
// cconst: cam_const row.
CBA_HD void cam_prepare(const double* x_cam, const double* cconst, int model, int nparams, CamTab* o, int param_off = 0) {
  const double rx = x_cam[0], ry = x_cam[1], rz = x_cam[2];
  const double th2 = rx * rx + ry * ry + rz * rz;
  const double th = sqrt(th2);
  double ca, sa, a, b;  // cos, sin, (1-cos)/th^2, (th-sin)/th^3
  if (th < 1e-4) {
    const double th4 = th2 * th2;
    a = 0.5 - th2 / 24.0 + th4 / 720.0;
    b = 1.0 / 6.0 - th2 / 120.0 + th4 / 5040.0;
    ca = 1.0 - th2 * a;
    sa = th - th2 * th * b;
  } else {
    ca = cos(th);
    sa = sin(th);
    a = (1.0 - ca) / th2;
    b = (th - sa) / (th2 * th);
  }
  const double sinc = (th < 1e-4) ? (1.0 - th2 / 6.0 + th2 * th2 / 120.0) : sa / th;
  // R = I + sinc [r]x + a [r]x^2 ;  J_l = I + a [r]x + b [r]x^2 ;  [r]x^2 = r r^T - th2 I
  const double xx = rx * rx, yy = ry * ry, zz = rz * rz, xy = rx * ry, xz = rx * rz, yz = ry * rz;
  o->R[0] = 1.0 + a * (xx - th2);  o->R[1] = -sinc * rz + a * xy;    o->R[2] = sinc * ry + a * xz;
  o->R[3] = sinc * rz + a * xy;    o->R[4] = 1.0 + a * (yy - th2);   o->R[5] = -sinc * rx + a * yz;
  o->R[6] = -sinc * ry + a * xz;   o->R[7] = sinc * rx + a * yz;     o->R[8] = 1.0 + a * (zz - th2);
  o->Jl[0] = 1.0 + b * (xx - th2); o->Jl[1] = -a * rz + b * xy;      o->Jl[2] = a * ry + b * xz;
  o->Jl[3] = a * rz + b * xy;      o->Jl[4] = 1.0 + b * (yy - th2);  o->Jl[5] = -a * rx + b * yz;
  o->Jl[6] = -a * ry + b * xz;     o->Jl[7] = a * rx + b * yz;       o->Jl[8] = 1.0 + b * (zz - th2);
  (void)ca;
  o->t[0] = x_cam[3]; o->t[1] = x_cam[4]; o->t[2] = x_cam[5];
  const double fx0 = cconst[0], fy0 = cconst[1];
  o->fx0 = fx0; o->fy0 = fy0; o->inv_fx0 = 1.0 / fx0;
  o->cx = cconst[2]; o->cy = cconst[3];
  // free intrinsics (nparams == 9): fx = s fx0, fy = s fy0, k1, k2 from x.  Selected by VALUE: written as stores under an `if`, the compiler merged
  // the two branches' stores into one store through a selected address — 24 bytes of scratch per lane in every kernel that prepares a camera.
  const bool free_intr = nparams == 9;
  o->fx = free_intr ? x_cam[6] * fx0 : fx0;
  o->fy = free_intr ? x_cam[6] * fy0 : fy0;
  o->d[0] = free_intr ? x_cam[7] : cconst[4];
  o->d[1] = free_intr ? x_cam[8] : cconst[5];
  for (int i = 2; i < 5; ++i) o->d[i] = cconst[4 + i];
  o->model = (double)model;
  o->nparams = (double)nparams;
  o->pad[0] = (double)param_off;  // (the per-observation kernels take a camera's slice of a vector from here: an index table in global memory cost them a
                                  // dependent load per observation, in the middle of their software pipelines)
  for (int i = 1; i < 13; ++i) o->pad[i] = 0.0;
}

// Lens model: normalised (x, y) -> distorted (xd, yd) and its 2x2 derivative dd = d(xd,yd)/d(x,y).
// Also r2-power terms for the k1,k2 columns of a free pinhole camera.
struct Lens {
  double xd, yd;
  double dxx, dxy, dyx, dyy;
  double xr2, yr2, xr4, yr4;  // x r^2, y r^2, x r^4, y r^4 (pinhole only)
};

CBA_HD void lens_pinhole(const double* d, double x, double y, Lens* L) {
  const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3], k3 = d[4];
  // written addend-first / Horner so that every step contracts to one FMA
  const double r2 = x * x + y * y, r4 = r2 * r2;
  const double cd = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
  const double dcd2 = 2.0 * (k1 + r2 * (2.0 * k2 + r2 * (3.0 * k3)));  // 2 d(cd)/d(r2)
  const double x2 = x + x, y2 = y + y;
  const double a1 = x2 * y, a2 = r2 + x2 * x, a3 = r2 + y2 * y;
  L->xd = x * cd + p1 * a1 + p2 * a2;
  L->yd = y * cd + p1 * a3 + p2 * a1;
  L->dxx = cd + x * x * dcd2 + p1 * y2 + 3.0 * p2 * x2;
  L->dxy = x * y * dcd2 + p1 * x2 + p2 * y2;
  L->dyx = L->dxy;
  L->dyy = cd + y * y * dcd2 + 3.0 * p1 * y2 + p2 * x2;
  L->xr2 = x * r2; L->yr2 = y * r2; L->xr4 = x * r4; L->yr4 = y * r4;
}

CBA_HD void lens_fisheye(const double* d, double x, double y, Lens* L) {
  const double r2 = x * x + y * y;
  const double r = sqrt(r2);
  if (r > 1e-8) {
    const double th = atan(r), t2 = th * th;
    const double poly = 1.0 + t2 * (d[0] + t2 * (d[1] + t2 * (d[2] + t2 * d[3])));
    const double dpoly = 1.0 + t2 * (3.0 * d[0] + t2 * (5.0 * d[1] + t2 * (7.0 * d[2] + t2 * 9.0 * d[3])));
    const double thd = th * poly;
    const double inv_r = 1.0 / r;
    const double cd = thd * inv_r;
    const double dcd_dr = (dpoly / (1.0 + r2) * r - thd) * inv_r * inv_r;
    const double g = dcd_dr * inv_r;  // d cd / dr * (1/r): multiply by x or y for the chain rule
    L->xd = x * cd; L->yd = y * cd;
    L->dxx = cd + x * x * g; L->dxy = x * y * g; L->dyx = L->dxy; L->dyy = cd + y * y * g;
  } else {
    L->xd = x; L->yd = y;
    L->dxx = 1.0; L->dxy = 0.0; L->dyx = 0.0; L->dyy = 1.0;
  }
  L->xr2 = L->yr2 = L->xr4 = L->yr4 = 0.0;
}

// Residual only (trial evaluations).  Returns false if the projection is not finite.
CBA_HD void project_residual(const CamTab& c, double X, double Y, double Z, double u, double v, double* e) {
  const double Xc = c.t[0] + c.R[0] * X + c.R[1] * Y + c.R[2] * Z;  // addend first: three FMAs
  const double Yc = c.t[1] + c.R[3] * X + c.R[4] * Y + c.R[5] * Z;
  const double Zc = c.t[2] + c.R[6] * X + c.R[7] * Y + c.R[8] * Z;
  const double iz = 1.0 / Zc;
  const double x = Xc * iz, y = Yc * iz;
  Lens L;
  if (c.model != 0.0) lens_fisheye(c.d, x, y, &L); else lens_pinhole(c.d, x, y, &L);
  e[0] = ((c.cx - u) + c.fx * L.xd) * c.inv_fx0;
  e[1] = ((c.cy - v) + c.fy * L.yd) * c.inv_fx0;
}

// Residual + Jacobian blocks.  A is [2][MAX_NC] (only the first nparams columns are written),
// B is [2][3].  Everything is already divided by fx_initial.
CBA_HD void project_full(const CamTab& c, double X, double Y, double Z, double u, double v,
                         double* e, double (*A)[MAX_NC], double (*B)[3]) {
  const double Y0 = c.R[0] * X + c.R[1] * Y + c.R[2] * Z;  // R X (world point rotated, not translated)
  const double Y1 = c.R[3] * X + c.R[4] * Y + c.R[5] * Z;
  const double Y2 = c.R[6] * X + c.R[7] * Y + c.R[8] * Z;
  const double Zc = Y2 + c.t[2];
  const double iz = 1.0 / Zc;
  const double x = (Y0 + c.t[0]) * iz, y = (Y1 + c.t[1]) * iz;
  Lens L;
  if (c.model != 0.0) lens_fisheye(c.d, x, y, &L); else lens_pinhole(c.d, x, y, &L);
  const double s = c.inv_fx0;
  e[0] = ((c.cx - u) + c.fx * L.xd) * s;
  e[1] = ((c.cy - v) + c.fy * L.yd) * s;
  // G = d(pixel)/dXc / fx0  (2x3)
  const double fxs = c.fx * s, fys = c.fy * s;
  const double g00 = fxs * L.dxx * iz, g01 = fxs * L.dxy * iz;
  const double g10 = fys * L.dyx * iz, g11 = fys * L.dyy * iz;
  const double G[2][3] = {{g00, g01, -(g00 * x + g01 * y)}, {g10, g11, -(g10 * x + g11 * y)}};
  for (int r = 0; r < 2; ++r) {
    // tvec columns
    A[r][3] = G[r][0]; A[r][4] = G[r][1]; A[r][5] = G[r][2];
    // point block  B = G R
    B[r][0] = G[r][0] * c.R[0] + G[r][1] * c.R[3] + G[r][2] * c.R[6];
    B[r][1] = G[r][0] * c.R[1] + G[r][1] * c.R[4] + G[r][2] * c.R[7];
    B[r][2] = G[r][0] * c.R[2] + G[r][1] * c.R[5] + G[r][2] * c.R[8];
  }
  // rvec columns: dXc/dr_j = Jl[:,j] x (R X)
  for (int j = 0; j < 3; ++j) {
    const double m0 = c.Jl[j], m1 = c.Jl[3 + j], m2 = c.Jl[6 + j];
    const double c0 = m1 * Y2 - m2 * Y1, c1 = m2 * Y0 - m0 * Y2, c2 = m0 * Y1 - m1 * Y0;
    A[0][j] = G[0][0] * c0 + G[0][1] * c1 + G[0][2] * c2;
    A[1][j] = G[1][0] * c0 + G[1][1] * c1 + G[1][2] * c2;
  }
  if (c.nparams == 9.0) {
    // d/ds = fx0 d/dfx + fy0 d/dfy ; d/dk1, d/dk2   (all / fx0)
    A[0][6] = c.fx0 * L.xd * s; A[1][6] = c.fy0 * L.yd * s;
    A[0][7] = fxs * L.xr2;      A[1][7] = fys * L.yr2;
    A[0][8] = fxs * L.xr4;      A[1][8] = fys * L.yr4;
  }
}

// Residual + the FACTORS of the Jacobian blocks (round 6).  The camera block is A = G [C | I | A_intr] with C = -[Y]x J_l (Y = R X), so its rotation
// columns are A_rot = (Y x G_r)^T J_l per row r: the per-camera constant J_l can be applied once per camera to whatever the rows are summed into
// (normal-equation blocks, right-hand sides, J v) instead of once per observation — the per-observation kernels then never read J_l (9 of the 36
// doubles of a camera row, all of which come out of LDS per observation) and skip the 36 flops of the rotation columns.
//   G (2 x 3) = d(pixel)/dX_c / fx0,   Yr (3) = R X,   B (2 x 3) = G R,   Aint (2 x 3): the free-intrinsics columns (zero unless nparams == 9)
CBA_HD void project_factors(const CamTab& c, double X, double Y, double Z, double u, double v,
                            double* e, double (*G)[3], double* Yr, double (*Aint)[3], double (*B)[3]) {
  const double Y0 = c.R[0] * X + c.R[1] * Y + c.R[2] * Z;
  const double Y1 = c.R[3] * X + c.R[4] * Y + c.R[5] * Z;
  const double Y2 = c.R[6] * X + c.R[7] * Y + c.R[8] * Z;
  Yr[0] = Y0; Yr[1] = Y1; Yr[2] = Y2;
  const double Zc = Y2 + c.t[2];
  const double iz = 1.0 / Zc;
  const double x = (Y0 + c.t[0]) * iz, y = (Y1 + c.t[1]) * iz;
  Lens L;
  if (c.model != 0.0) lens_fisheye(c.d, x, y, &L); else lens_pinhole(c.d, x, y, &L);
  const double s = c.inv_fx0;
  e[0] = ((c.cx - u) + c.fx * L.xd) * s;
  e[1] = ((c.cy - v) + c.fy * L.yd) * s;
  const double fxs = c.fx * s, fys = c.fy * s;
  const double g00 = fxs * L.dxx * iz, g01 = fxs * L.dxy * iz;
  const double g10 = fys * L.dyx * iz, g11 = fys * L.dyy * iz;
  G[0][0] = g00; G[0][1] = g01; G[0][2] = -(g00 * x + g01 * y);
  G[1][0] = g10; G[1][1] = g11; G[1][2] = -(g10 * x + g11 * y);
  for (int r = 0; r < 2; ++r) {
    B[r][0] = G[r][0] * c.R[0] + G[r][1] * c.R[3] + G[r][2] * c.R[6];
    B[r][1] = G[r][0] * c.R[1] + G[r][1] * c.R[4] + G[r][2] * c.R[7];
    B[r][2] = G[r][0] * c.R[2] + G[r][1] * c.R[5] + G[r][2] * c.R[8];
  }
  const bool free_intr = c.nparams == 9.0;
  Aint[0][0] = free_intr ? c.fx0 * L.xd * s : 0.0; Aint[1][0] = free_intr ? c.fy0 * L.yd * s : 0.0;
  Aint[0][1] = free_intr ? fxs * L.xr2 : 0.0;      Aint[1][1] = free_intr ? fys * L.yr2 : 0.0;
  Aint[0][2] = free_intr ? fxs * L.xr4 : 0.0;      Aint[1][2] = free_intr ? fys * L.yr4 : 0.0;
}

// scipy's robust-loss treatment of ONE scalar residual r (least_squares.py:169-237, common.py:720-731).
// Returns rho0 * f_scale^2 (so that cost = 0.5 * sum); *row_scale multiplies the Jacobian row,
// *r_scaled replaces the residual.
CBA_HD double robust_one(int loss, double f_scale, double r, double* row_scale, double* r_scaled) {
  if (loss == LOSS_LINEAR) { *row_scale = 1.0; *r_scaled = r; return r * r; }
  const double fs2 = f_scale * f_scale;
  const double z = (r * r) / fs2;
  double rho0, rho1, rho2;
  switch (loss) {
    case LOSS_HUBER:
      if (z <= 1.0) { rho0 = z; rho1 = 1.0; rho2 = 0.0; }
      else { const double sz = sqrt(z); rho0 = 2.0 * sz - 1.0; rho1 = 1.0 / sz; rho2 = -0.5 / (z * sz); }
      break;
    case LOSS_SOFT_L1: {
      const double t = 1.0 + z, st = sqrt(t);
      rho0 = 2.0 * (st - 1.0); rho1 = 1.0 / st; rho2 = -0.5 / (t * st);
    } break;
    case LOSS_CAUCHY: {
      const double t = 1.0 + z;
      rho0 = log1p(z); rho1 = 1.0 / t; rho2 = -1.0 / (t * t);
    } break;
    default: {  // arctan
      const double t = 1.0 + z * z;
      rho0 = atan(z); rho1 = 1.0 / t; rho2 = -2.0 * z / (t * t);
    } break;
  }
  // scipy: rho[2] /= f_scale^2 ; J_scale = rho1 + 2 rho2 f^2  ==  rho1 + 2 rho2_unscaled z
  double js = rho1 + 2.0 * rho2 * z;
  if (js < EPS_F64) js = EPS_F64;
  js = sqrt(js);
  *row_scale = js;
  *r_scaled = r * rho1 / js;
  return rho0 * fs2;
}

// cost-only variant (trial evaluations)
CBA_HD double robust_cost_one(int loss, double f_scale, double r) {
  if (loss == LOSS_LINEAR) return r * r;
  const double fs2 = f_scale * f_scale;
  const double z = (r * r) / fs2;
  switch (loss) {
    case LOSS_HUBER: return fs2 * (z <= 1.0 ? z : 2.0 * sqrt(z) - 1.0);
    case LOSS_SOFT_L1: return fs2 * 2.0 * (sqrt(1.0 + z) - 1.0);
    case LOSS_CAUCHY: return fs2 * log1p(z);
    default: return fs2 * atan(z);
  }
}

// Cholesky of a symmetric 3x3 given as (xx, xy, xz, yy, yz, zz).  L = [l00; l10 l11; l20 l21 l22] is returned with
// RECIPROCAL diagonal entries: out = (1/l00, l10, 1/l11, l20, l21, 1/l22) — every use below multiplies by them, so
// the factorisation and the solves contain no division and no square root, only three inv_sqrt.
// Returns false when a pivot is not safely positive.
CBA_HD bool chol3(const double* v, double* L) {
  const double tiny = 16.0 * EPS_F64;
  if (!(v[0] > 0.0)) return false;
  const double i0 = inv_sqrt(v[0]);
  const double l10 = v[1] * i0, l20 = v[2] * i0;
  const double d1 = v[3] - l10 * l10;
  if (!(d1 > tiny * v[3])) return false;
  const double i1 = inv_sqrt(d1);
  const double l21 = (v[4] - l20 * l10) * i1;
  const double d2 = v[5] - l20 * l20 - l21 * l21;
  if (!(d2 > tiny * v[5])) return false;
  L[0] = i0; L[1] = l10; L[2] = i1; L[3] = l20; L[4] = l21; L[5] = inv_sqrt(d2);
  return true;
}
// y = L^{-1} b
CBA_HD void chol3_fwd(const double* L, const double* b, double* y) {
  y[0] = b[0] * L[0];
  y[1] = (b[1] - L[1] * y[0]) * L[2];
  y[2] = (b[2] - L[3] * y[0] - L[4] * y[1]) * L[5];
}
// x = L^{-T} y
CBA_HD void chol3_bwd(const double* L, const double* y, double* x) {
  x[2] = y[2] * L[5];
  x[1] = (y[1] - L[4] * x[2]) * L[2];
  x[0] = (y[0] - L[1] * x[1] - L[3] * x[2]) * L[0];
}

}  // namespace cba
