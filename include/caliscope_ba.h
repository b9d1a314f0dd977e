/* caliscope_ba.h — C ABI of the MI355X-native bundle-adjustment engine (libcaliscope_ba.so).
 *
 * The reference has no FFI for this path: the seam is the single Python call
 *     scipy.optimize.least_squares(joint_residuals, x0, args=(parameterization, camera_indices,
 *         image_coords, image_to_world_indices, ...), jac=joint_jacobian, x_scale="jac", loss=...,
 *         f_scale=..., ftol=..., max_nfev=..., method="trf", bounds=...)
 * at /root/reference/src/caliscope/core/capture_volume.py:387-411.  The entry points below are what a
 * ctypes binding for that seam needs: a problem object built from exactly the arrays that call
 * receives, cba_solve — that whole call as one entry point —, the evaluation primitives of one trust-region
 * iteration it is built from (the arithmetic of core/reprojection.py:75-119 `joint_residuals` and :128-234
 * `joint_jacobian`, of scipy's `scale_for_robust_loss_function`, `compute_grad`, `compute_jac_scale` and of the
 * regularised Gauss-Newton step that scipy gets from LSMR), and parity hooks that expose intermediate results.
 * The trust-region control flow runs on the host inside the library (csrc/cba_solve.cpp — the only driver of the package; oracle/trf_driver.py,
 * test infrastructure, restates its unbounded loop in Python on the primitives) and sees scalars only.
 *
 * Conventions
 *  - all pointers are caller-owned host memory, C-contiguous, valid for the duration of the call only;
 *  - every function returns 0 on success, a negative code on error; cba_last_error() (thread-local)
 *    describes the last failure; no C++ exception crosses this boundary;
 *  - a cba_problem is not re-entrant (serialise calls per handle); different handles may be used from
 *    different threads; every entry point selects the handle's device itself;
 *  - parameter vectors `x` use the reference layout (core/bundle_parameterization.py:36-51,114-149):
 *    camera i at x[off_i .. off_i+n_i) = [rvec(3), tvec(3) (, s, k1, k2)], then points row-major (P,3);
 *  - there is NO CPU fallback: without a HIP device cba_create fails with CBA_ERR_NO_DEVICE.
 */
#ifndef CALISCOPE_BA_H
#define CALISCOPE_BA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CBA_VERSION 100 /* 0.1.0 */

enum {
  CBA_OK = 0,
  CBA_ERR_INVALID = -1,    /* bad argument / inconsistent sizes */
  CBA_ERR_NO_DEVICE = -2,  /* no usable HIP device */
  CBA_ERR_HIP = -3,        /* a HIP runtime call failed */
  CBA_ERR_UNSUPPORTED = -4,/* valid input the engine does not handle yet */
  CBA_ERR_COMM = -5,       /* RCCL failure */
  CBA_ERR_NUMERIC = -6     /* the damped normal equations could not be factorised */
};

enum { CBA_MODEL_PINHOLE_BC5 = 0, CBA_MODEL_FISHEYE4 = 1 };
enum { CBA_LOSS_LINEAR = 0, CBA_LOSS_HUBER = 1, CBA_LOSS_SOFT_L1 = 2, CBA_LOSS_CAUCHY = 3, CBA_LOSS_ARCTAN = 4 };

typedef struct cba_problem cba_problem; /* opaque; owns all device memory */

/* The arrays least_squares receives through `args` (capture_volume.py:389-399), flattened.
 * Replaces: BundleParameterization (blocks -> cam_* tables), camera_indices (int16 in the reference,
 * widened), image_coords (N,2 AoS float64), image_to_world_indices (int32). */
typedef struct {
  int32_t n_cams;
  int32_t n_points;
  int64_t n_obs;
  const int32_t* cam_n_params; /* [C] 6 (locked) or 9 (free intrinsics: + s, k1, k2)          */
  const int32_t* cam_model;    /* [C] CBA_MODEL_*                                              */
  const double* cam_const;     /* [C][12] fx0 fy0 cx cy | k1 k2 p1 p2 k3 (pinhole) or k1..k4 0 (fisheye) | 0 0 0 */
  const int32_t* obs_cam;      /* [N] camera index of each observation                         */
  const int32_t* obs_pt;       /* [N] world-point row of each observation                      */
  const double* obs_uv;        /* [N][2] observed pixel (img_loc_x, img_loc_y)                 */
  int32_t loss;                /* CBA_LOSS_* — scipy's `loss`                                  */
  double f_scale;              /* scipy's `f_scale`                                            */
} cba_problem_desc;

typedef struct {
  int32_t device_id;     /* HIP device ordinal; -1 = current device                                  */
  int32_t max_blocks;    /* 0 = default (4 workgroups per CU); upper bound on persistent grid size   */
  int32_t deterministic; /* 1: the per-camera sums of the build and Schur passes are formed in a fixed order (parked in LDS, summed per
                            camera in the chunk's camera-sorted order) instead of by FP64 LDS atomics: two solves of one problem return
                            identical bits, as the reference's single-threaded scipy does (capture_volume.py:387).  Constraint rows and
                            heavy points still add their few sums atomically.  Up to 227 nine-parameter or 385 six-parameter cameras
                            (tasks per thread; the LDS copy of the camera table): cba_create returns CBA_ERR_UNSUPPORTED beyond. */
  int32_t evaluation_only; /* 1: residuals / costs only (cba_residuals, cba_begin): the Schur plan and the solver's buffers are
                              not built — the reprojection report needs no more */
} cba_options;

/* ---- lifetime ------------------------------------------------------------------------------- */
int cba_create(const cba_problem_desc* desc, const cba_options* opt, cba_problem** out);
void cba_destroy(cba_problem* p);

/* ---- sharded solves (one process per GPU) --------------------------------------------------------
 * The world points (with ALL their observations) are partitioned over the ranks by the host
 * (caliscope_amd/sharding.py); every rank creates its problem from its own shard and the full camera set.
 * After cba_comm_init the primitives below all-reduce — over RCCL/xGMI, on the engine's stream — exactly
 * the camera blocks U_c, g_c, the reduced camera system (S, b) and the scalar sums; the dense solve is
 * repeated identically on every rank.  Rank 0 calls cba_comm_unique_id and distributes the 128 bytes by any
 * host-side channel (caliscope_amd/distributed.py: a shared variable between the threads of one process, TCP on the
 * loopback between processes).  cba_comm_abort (ncclCommAbort) is for the failure path: when one rank fails, the others
 * may already sit in a collective that will never complete; aborting their communicators from another thread makes the
 * pending call return an error (CBA_ERR_COMM / CBA_ERR_HIP) instead of hanging.  The handle is only good for cba_destroy afterwards. */
int cba_comm_unique_id(char* out128);
int cba_comm_init(cba_problem* p, const char* id128, int32_t rank, int32_t world);
int cba_comm_abort(cba_problem* p);

/* ---- sharded solves inside ONE process (one host thread per GPU) ----------------------------------
 * `CaptureVolume.optimize()` is a single in-process call in the reference (core/capture_volume.py:322-334), so a
 * caller cannot be asked for a launcher: caliscope_amd.distributed.solve_multi_device starts one thread per device,
 * each creates the handle of its shard, and the handles join a group.  With RCCL the threads call cba_comm_init with a
 * shared unique id; cba_group_* is the direct alternative: every rank stages its contribution on its own device and sums
 * the staging buffers of all ranks in rank order through peer access over xGMI (bit-identical replicas, no ring).  The
 * group also accepts several handles on the SAME device, which is how a 1-GPU box runs the multi-rank protocol in the tests.
 * cba_group_join is called concurrently by the `world` member threads and returns when all have joined; destroy the
 * member handles before the group. */
typedef struct cba_group cba_group;
int cba_group_create(int32_t world, cba_group** out);
int cba_group_join(cba_problem* p, cba_group* g, int32_t rank);
void cba_group_abort(cba_group* g);   /* a member failed: ranks waiting in the group's barrier return an error instead of spinning */
void cba_group_destroy(cba_group* g);

/* ---- one trust-region iteration, as primitives (scalars out, vectors stay on the device) ------ */

/* Upload x0 (reference layout, length n = sum(cam_n_params) + 3 P), evaluate the residuals there.
 * cost_out = 0.5 * sum rho(f) — scipy's `cost` (least_squares.py:838-864).  Resets the Jacobi scaling. */
int cba_begin(cba_problem* p, const double* x0, double* cost_out);

/* Same as cba_begin with the x0 of the last cba_begin, which is kept on the device (no host transfer). */
int cba_restart(cba_problem* p, double* cost_out);

/* cba_begin / cba_restart (x0 == NULL) without the evaluation: the first cba_step or cba_linearize evaluates x0 with the
 * build pass it runs anyway and reports the cost (NaN if a residual is not finite) — one pass and one wait less per solve. */
int cba_begin_deferred(cba_problem* p, const double* x0);

typedef struct {
  double g_norm_inf;    /* ||J^T f||_inf                      (trf.py:459)                       */
  double gh_sq;         /* ||g_h||^2,  g_h = g / scale_inv    (trf.py:469)                       */
  double jg_sq;         /* ||J_h g_h||^2                      (build_quadratic_1d, trf.py:480)   */
  double x_scaled_norm; /* ||x * scale_inv||                  (trf.py:428, initial Delta)        */
  double x_norm;        /* ||x||                              (check_termination)                */
  double cost;          /* cost at the current x (re-evaluated with the build pass)              */
} cba_linearization;

/* Residuals + Jacobian blocks at the current x, robust-loss scaling (common.py:720-731), g = J^T f,
 * U_c = sum A^T A, V_p = sum B^T B, Jacobi scale with the monotone-max rule (common.py:598-610). */
int cba_linearize(cba_problem* p, cba_linearization* out);

typedef struct {
  int32_t ok;       /* 0: a Cholesky pivot was not positive (raise `lam` and retry)              */
  int32_t reserved;
  double p_sq;      /* ||p||^2,  p = scale_inv * s   (scaled step)                                */
  double gh_dot_p;  /* <g_h, p> = <g, s>                                                          */
  double w_sq;      /* ||p - (<g_h,p>/||g_h||^2) g_h||^2                                          */
} cba_newton_info;

/* Solve (J^T J + lam * diag(scale_inv^2)) s = -g by Schur complement on the reduced camera system:
 * replaces lsmr(J_h, f, damp=sqrt(reg_term)) at trf.py:485 with an exact damped step. */
int cba_newton_step(cba_problem* p, double lam, cba_newton_info* out);

/* gram_out = { |J v1|^2, <J v1, J v2>, |J v2|^2 },  v_i = a_i * g / scale_inv^2 + b_i * s
 * (explicit J_h @ S of trf.py:488-489; only used when the subspace basis is ill-conditioned). */
int cba_subspace_gram(cba_problem* p, double a1, double b1, double a2, double b2, double* gram_out);

typedef struct {
  double cost;      /* 0.5 * sum rho(f(x_new)); NaN if any residual is not finite                 */
  double step_norm; /* ||x_new - x||                                                              */
  int32_t finite;
  int32_t reserved;
} cba_trial_info;

/* x_new = x + alpha * g / scale_inv^2 + beta * s ; evaluate the cost there (trf.py:496-511). */
int cba_trial(cba_problem* p, double alpha, double beta, cba_trial_info* out);

/* x <- x_new (trf.py:525-526). */
int cba_accept(cba_problem* p);

/* ---- one iteration, one synchronisation ----------------------------------------------------------------
 * cba_linearize + the regularisation rule + cba_newton_step + the 2-D subspace solve + cba_trial of the first trial
 * point in ONE call: the two scalar decisions in between (reg_term of trf.py:477-483, solve_trust_region_2d of
 * trf.py:491-493) are taken on the device by the same code the host driver uses (csrc/trf_math.h), and the trial point
 * is evaluated by a full build pass into a second set of buffers — accepting it (cba_accept) makes it the next
 * linearisation point without another pass.  radius <= 0: the initial Delta = ||x0 * scale_inv|| (trf.py:428-430).
 * need_host = 1: the factorisation failed or the step is nearly collinear with the gradient (explicit J.v model): no
 * trial was made, continue with the primitives (cba_newton_step / cba_subspace_gram / cba_trial) for this iteration.
 * A rejected first trial is retried with cba_trial as usual.  Not available (cba_step_supported() == 0) with constraint
 * rows, heavy points, or after cba_set_camera_scaling on a handle without cba_set_bounds. */
typedef struct {
  cba_linearization lin;
  cba_newton_info newton;
  cba_trial_info trial;
  double lam, radius;        /* damping used, radius used                                          */
  double p_s[2], predicted;  /* subspace step in the basis (g_h / |g_h|, w / |w|), model decrease  */
  double alpha, beta;        /* trial step = alpha * g / scale_inv^2 + beta * s                    */
  int32_t need_host, reserved; /* need_host 1: failed factorisation / collinear step; 2: bounded trial point outside its box */
} cba_step_info;
int cba_step(cba_problem* p, double radius, cba_step_info* out);
/* p_sq, gh_dot_p and w_sq of the current damped step measured again, w_sq by a pass of its own: cba_step derives w_sq
 * from the other two (one collective less), which loses accuracy when the step is nearly collinear with the gradient —
 * exactly the need_host case, whose explicit-model branch needs it accurately. */
int cba_refresh_step_scalars(cba_problem* p, cba_newton_info* out);
int cba_step_supported(cba_problem* p);

/* ---- bounded camera parameters (scipy trf_bounds, trf.py:205-398) ------------------------------------
 * With finite bounds (free intrinsics: s, k1, k2 of BundleParameterization.bounds()) scipy rescales every bounded
 * variable by the Coleman-Li vector v (distance to the bound the gradient points at) and adds the diagonal
 * C = diag(g * dv * scale) to the model Hessian.  Only entries of the camera block can be bounded, so the driver
 * computes v on the host from the camera parts of x, g and scale_inv and hands two small vectors to the engine:
 *   mult   [n_cam_params]  effective scale_inv of the camera block = Jacobi scale_inv * mult   (1 / sqrt(v) ; 1 if unbounded)
 *   diag_h [n_cam_params]  C in the scaled space, >= 0                                           (0 if unbounded)
 * The damped system of cba_newton_step becomes (J^T J + (lam + diag_h) * scale_inv_eff^2) s = -g, and every scaled
 * quantity (gh_sq, jg_sq, x_scaled_norm, p_sq, w_sq, the trial step) uses the effective scale.  `out` is the
 * linearisation in the new scaling, with g_norm_inf = max |g| over the POINT block only (the caller folds the camera
 * block in with its v).  Call after every cba_linearize of a bounded solve; the Jacobi scale's monotone-max state is
 * kept apart and is not disturbed. */
int cba_set_camera_scaling(cba_problem* p, const double* mult, const double* diag_h, cba_linearization* out);

/* The same loop with ONE host synchronisation per iteration (round 5): hand the bounds of the camera block (lb, ub [n_cam_params]; +-inf where a
 * parameter is free) to the device once per solve; cba_step then computes the Coleman-Li scaling itself from x, g and the Jacobi scale (what
 * cba_get_camera_state + the loop of trf.py:283-296 + cba_set_camera_scaling do in three round trips), forms the damped step with C = diag_h in the
 * model, and evaluates the first trial point IF it lies strictly inside the box; a trial point on or beyond a bound comes back as need_host = 2 and the
 * caller runs this iteration through the primitives (select_step of trf.py:129-202: truncated step, reflection, scaled anti-gradient).
 * Returns 1 when bounded cba_step iterations are available on this handle, 0 when not (sharded solves, constraint rows, heavy points, fixed-order
 * sums: use the primitives), < 0 on error.  lb == NULL switches it off.  In a bounded cba_step, lin.g_norm_inf is ||g v||_inf (trf.py:298).
 * cba_step_camera_state: the camera blocks of x, g, the JACOBI scale (before the Coleman-Li factor) and the damped step of the last bounded cba_step,
 * each [n_cam_params] — they travelled with the step's packet, no synchronisation: what the caller needs for the later trials of the iteration. */
int cba_set_bounds(cba_problem* p, const double* lb, const double* ub);
int cba_step_camera_state(cba_problem* p, double* x_c, double* g_c, double* scale_inv_c, double* step_c);

/* cba_linearize without the scalars: build pass and Jacobi scale only, no host synchronisation.  For bounded solves,
 * which rescale the camera block before any scaled quantity is meaningful: cba_linearize_build, read the camera parts
 * of x / g / scale_inv, cba_set_camera_scaling (which returns the linearisation). */
int cba_linearize_build(cba_problem* p);

/* cba_subspace_gram with the camera block of v1 / v2 replaced by cam1 / cam2 ([n_cam_params], x-space; NULL: not
 * replaced) — the reflected direction of select_step (trf.py:129-202) differs from the trust-region step in the
 * camera entries that hit a bound. */
int cba_subspace_gram_ex(cba_problem* p, double a1, double b1, const double* cam1, double a2, double b2, const double* cam2,
                         double* gram_out);

/* cba_trial with the camera block of x_new given by the caller (cam_x_new [n_cam_params], NULL: as cba_trial): the
 * driver places bounded parameters itself (truncated / reflected steps, make_strictly_feasible). */
int cba_trial_ex(cba_problem* p, double alpha, double beta, const double* cam_x_new, cba_trial_info* out);

/* ---- the whole solve behind one call ---------------------------------------------------------------
 * Replaces the reference's `scipy.optimize.least_squares(joint_residuals, x0, args=(...), jac=joint_jacobian,
 * x_scale="jac", loss=..., f_scale=..., ftol=..., max_nfev=..., method="trf", bounds=...)` call
 * (core/capture_volume.py:387-411) for a problem created with cba_create (loss, f_scale and the observation arrays
 * live in the handle; constraint rows via cba_set_constraints): the trust-region loop of scipy's trf.py:401-560 run
 * on the primitives above (csrc/cba_solve.cpp).  What the reference consumes of scipy's result (:413-432) is in
 * cba_result: status (-1..4, scipy's codes), x, nfev, cost. */
typedef struct {
  double ftol, xtol, gtol;     /* scipy's tolerances (defaults 1e-8)                                           */
  int64_t max_nfev;            /* <= 0: 100 * n, scipy's default for max_nfev=None                             */
  const double* lb;            /* [n_cam_params] bounds of the camera block of x (BundleParameterization.bounds */
  const double* ub;            /*   restricted to the cameras), or NULL: iterates are kept strictly inside      */
  int32_t verbose;             /* 2: scipy's per-iteration table on stdout                                     */
  int32_t max_damping_retries; /* <= 0: 12                                                                     */
} cba_solve_options;

typedef struct {
  int32_t status;              /* -1 improper input, 0 max_nfev reached, 1 gtol, 2 ftol, 3 xtol, 4 ftol and xtol */
  int32_t reserved;            /* bounded solves: trial steps that left the box and were replaced by the truncated step (bits 0-9),
                                  its reflection (10-19), the scaled anti-gradient (20-29) — select_step's three candidates */
  int64_t nfev, njev, n_iterations;
  double cost;                 /* 0.5 * sum rho(f) at the returned x                                           */
  double optimality;           /* ||J^T f||_inf at the returned x                                              */
  double t_total_s;            /* wall time of the call                                                        */
  double t_rejected_s;         /* of it: wall time of the cba_trial / cba_trial_ex calls whose trial point was rejected (a cost pass  */
  int64_t n_rejected_timed;    /*   each) and their number; a rejected FIRST trial of a fused iteration is not in here (it was built  */
                               /*   inside cba_step): nfev - njev - n_rejected_timed of those                                          */
} cba_result;

/* x0 [n] (NULL: restart from the x0 of the previous cba_begin / cba_solve, kept on the device), opt (NULL: defaults),
 * x_out [n] or NULL (the solution stays on the device, cba_get_vector(CBA_VEC_X)).  In a sharded solve every rank
 * calls it with its own handle; the scalars steering the loop are identical on all ranks. */
int cba_solve(cba_problem* p, const double* x0, const cba_solve_options* opt, double* x_out, cba_result* out);

/* ---- parity hooks / readback -------------------------------------------------------------------- */
enum { CBA_VEC_X = 0, CBA_VEC_X_NEW = 1, CBA_VEC_GRAD = 2, CBA_VEC_STEP = 3, CBA_VEC_SCALE_INV = 4 };

/* Download a device vector in the reference layout (length n). */
int cba_get_vector(cba_problem* p, int32_t which, double* out);

/* Download only the camera part (first n_cam_params entries) of a device vector — cheap (<= 9 KB);
 * used by the host to test trial points against the intrinsic bounds. */
int cba_get_camera_params(cba_problem* p, int32_t which, double* out);

/* The camera blocks of x, g and scale_inv in one call (what the Coleman-Li scaling of a bounded solve reads after every
 * linearisation); each [n_cam_params]. */
int cba_get_camera_state(cba_problem* p, double* x_c, double* g_c, double* scale_inv_c);

/* joint_residuals(x) in the caller's observation order, interleaved (x0,y0,x1,y1,...) [2N], followed by the
 * constraint rows [n_con] when cba_set_constraints was called, and the (robust) cost.  The linearisation and the damped
 * step stay valid; a pending trial point (cba_trial without cba_accept) is discarded. */
int cba_residuals(cba_problem* p, const double* x, double* r_out, double* cost_out);

/* Blocks of J^T J and J^T f at x (robust-scaled like scipy's J, f):
 *   U [C][9][9] dense symmetric (upper-left n_c x n_c used), V [P][6] = xx xy xz yy yz zz,
 *   gc [sum n_c], gp [P][3].  Does not disturb the solver state except the scratch blocks. */
int cba_normal_blocks(cba_problem* p, const double* x, double* U, double* V, double* gc, double* gp);

/* The reduced camera system of the LAST cba_newton_step: S [ncp][ncp] symmetric (full), rhs [ncp]. */
int cba_reduced_system(cba_problem* p, double* S, double* rhs);

/* Change the robust loss of an existing problem (same observations, same camera tables): what a caller does between the stages of
 * calibrate_extrinsics (linear, then soft_l1 on the same data: reference core/calibrate_extrinsics.py:206,230-238) without paying for the
 * Schur plan again.  The next cba_begin / cba_solve starts from scratch with the new loss. */
int cba_set_loss(cba_problem* p, int32_t loss, double f_scale);

/* A handle of 500k observations or more starts with a quickly made Schur plan and swaps the balanced one in when the host thread dealing it is done
 * (a solve in progress picks it up between two iterations).  The two plans add the same pair products in a DIFFERENT ORDER: the reduced system agrees
 * to rounding (1e-13 relative), not bit for bit, and which iteration the swap lands on depends on host timing — a large default solve is therefore
 * reproducible to summation order only (now and then one evaluation more or fewer); cba_options.deterministic = 1 builds the balanced plan inside
 * cba_create and fixes every order.  cba_plan_wait blocks until the balanced plan is installed (what a benchmark calls before its timed region; no-op
 * on a final handle) and returns the error of a background build that failed — the handle then stays usable on the quick plan (cba_info.plan_state 2).
 * (No counterpart in the reference: its solver has no set-up phase, core/capture_volume.py:387.) */
int cba_plan_wait(cba_problem* p);

/* ---- introspection ------------------------------------------------------------------------------ */
typedef struct {
  int32_t n_cams, n_points, n_cam_params, n_params;
  int64_t n_obs;
  int32_t n_chunks, grid_blocks;
  int32_t plan_state;     /* Schur plan of the pair kernel: 0 = the balanced ("dealt") plan, final; 1 = the quickly made plan of a two-stage handle, the
                             balanced one still being built on its thread (cba_plan_wait / the next damped step installs it); 2 = the quick plan for
                             good: CBA_PLAN=cheap, or the background build failed (plan_error) */
  int32_t max_obs_per_point;
  int64_t device_bytes;
  int32_t schur_groups;   /* G camera groups -> G(G+1)/2 tiles */
  int32_t schur_tiles;
  int32_t schur_grid;     /* workgroups of the tiled Schur kernel */
  int32_t n_heavy_points; /* points with > 40 observations (static markers): per-camera Schur sums, split over chunks beyond 256 */
  int64_t schur_stream_len; /* total observations over all tile streams (recompute factor = this / n_obs) */
  int64_t schur_pairs;      /* observation pairs (blocks of W V^-1 W^T) formed per Schur pass */
  int32_t plan_error;       /* CBA_ERR_* of a failed background plan build (e.g. hipMalloc beside a large solve), else 0: the handle then keeps the
                               quick plan (pair kernel 1.5-1.75x slower, same sums to rounding); cba_plan_wait returns it as an error */
  int32_t build_camg;       /* bit 0: the linearisation kernel reads the camera table through the vector cache instead of LDS (chosen when the
                               table is what keeps a second workgroup off the CU, ~100+ nine-parameter cameras; CBA_BUILD_CAMG=0/1 forces);
                               bit 1: EVERY per-observation kernel does (the LDS copy of the table would not fit: beyond ~230 six- / ~170
                               nine-parameter cameras; CBA_CAMTAB_GLOBAL=0/1 forces).  The camera count is then bounded by the per-camera
                               accumulators of the linearisation: ~650 six- / ~320 nine-parameter cameras;
                               bit 2: the linearisation runs over camera-sorted super-chunks (k_build_cs: camera blocks accumulated in registers,
                               the default; CBA_BUILD_CS=0, deterministic sums or fragments of very large points fall back to k_build). */
} cba_info;
int cba_get_info(cba_problem* p, cba_info* out);

/* Accumulated device time per kernel family since the last reset, measured with HIP events on the
 * engine's stream; names/ms arrays of length >= cba_timer_count(). */
int cba_timer_count(void);
const char* cba_timer_name(int32_t i);
int cba_get_timers(cba_problem* p, double* ms_out, int64_t* calls_out);
int cba_reset_timers(cba_problem* p);
int cba_enable_timers(cba_problem* p, int32_t on);

/* Host-side planning only (no device needed): observation order sorted by (point, camera) — obs_cam may be
 * NULL to sort by point only — stable, CSR offsets
 * per point and the chunk table (whole points per chunk, at most `chunk_cap` observations).
 * order_out [N], pt_start_out [P+1], chunk_start_out [N+1 worst case]; returns the number of chunks
 * (>= 0) or a negative error.  A point with more than chunk_cap observations gets chunks of its own (fragments of at
 * most chunk_cap observations, no other point in them). */
int64_t cba_host_plan(int32_t n_points, int64_t n_obs, const int32_t* obs_pt, const int32_t* obs_cam, int32_t n_cams,
                      int32_t chunk_cap, int64_t* order_out, int64_t* pt_start_out, int64_t* chunk_start_out);

/* Rigid-distance constraint rows appended after the 2 n_obs reprojection rows (reference core/reprojection.py:112-117
 * residual, :207-226 Jacobian; group arrays as built by capture_volume.py:446-531):
 *   r_c = weights[c] * (|| mean(X[groups_a[c][0..3]]) - mean(X[groups_b[c][0..3]]) || - distances[c]).
 * A corner endpoint repeats one point index four times.  The robust loss applies to these rows as to the others.
 * Call once, after cba_create and before cba_begin.  A connected component of the constraint graph (one board in one
 * frame; all static markers together) may couple any number of points; its m rows cost m^2 doubles (2 GB over all
 * components is the limit, CBA_ERR_UNSUPPORTED beyond).  In a sharded solve every rank passes the rows of its own points (the host
 * sharder keeps a component on one rank, caliscope_amd/sharding.py). */
int cba_set_constraints(cba_problem* p, int32_t n_con, const int32_t* groups_a, const int32_t* groups_b, const double* distances,
                        const double* weights);

/* ---- the step before the path: undistortion + batched DLT triangulation (x0 of the world points) ----
 *
 * Replaces, for one batch of 3-D points, `CameraData.undistort_points(points, output="normalized")`
 * (reference cameras/camera_array.py:135-174: cv2.undistortPoints / cv2.fisheye.undistortPoints with P = I on
 * float32 arrays) and `triangulate_image_points` (core/point_data.py:121-229: per point the 2k x 4 DLT matrix with
 * rows x P[2] - P[0], y P[2] - P[1], smallest right-singular vector).  One GPU thread per point; the singular
 * vector is the eigenvector of the smallest eigenvalue of A^T A (4 x 4, cyclic Jacobi).
 * Observations of a point are contiguous: point q owns obs_cam/obs_xy[pt_start[q] .. pt_start[q+1]).
 * cam_intr == NULL: obs_xy already holds undistorted normalised coordinates (the reference function's own input).
 * Points with fewer than two observations get NaN. */
typedef struct {
  int32_t n_cams;
  const int32_t* cam_model;  /* [n_cams] 0 = pinhole + Brown-Conrady 5, 1 = fisheye 4; ignored when cam_intr == NULL */
  const double* cam_intr;    /* [n_cams][9]  fx fy cx cy d0 d1 d2 d3 d4, or NULL */
  const double* cam_P;       /* [n_cams][12] normalised projection matrix [R | t], row-major */
  int64_t n_points;
  const int64_t* pt_start;   /* [n_points + 1] */
  const int32_t* obs_cam;    /* [n_obs] camera index in [0, n_cams) */
  const double* obs_xy;      /* [n_obs][2] */
  int32_t float32_io;        /* 1: round the pixel input and the normalised output to float32 like the reference's cv2 calls */
} cba_triangulate_desc;

/* xyz_out [n_points][3]; undistorted_out [n_obs][2] or NULL.  device = HIP device ordinal. */
int cba_triangulate(const cba_triangulate_desc* d, int32_t device, double* xyz_out, double* undistorted_out);

/* Give back what the library keeps between handles: per device up to four 4 MB arena chunks, streams, mapped mailboxes and pinned staging buffers of
 * destroyed handles, and the process-wide pool of huge-page host blocks the set-up's large arrays come from (up to 3 GB after a 10M-observation
 * handle).  Live handles are not touched.  A long-lived host process (the reference's GUI session) calls it when a calibration is done:
 * caliscope_amd.engine_cache.clear() does.  Returns the bytes released (host + device). */
int64_t cba_trim(void);

const char* cba_last_error(void);
/* Sets the calling thread's error message and returns `code` (for drivers layered on the primitives, cba_solve). */
int cba_set_error(int32_t code, const char* message);
int cba_version(void);
int cba_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* CALISCOPE_BA_H */
