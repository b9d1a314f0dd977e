// HIP kernels of the bundle-adjustment engine for gfx950 (MI355X).  Included by cba_lib.hip only.
//
// Data layout in HBM:
//   observations, sorted by world point:  obs_u[N], obs_v[N] (f64), obs_cam[N], obs_pt[N] (i32)   SoA
//   chunk table: chunk_start[n_chunks+1] — contiguous observation ranges that contain whole points and at
//                most CHUNK (=256) observations; one 256-thread workgroup processes one chunk at a time
//   vectors (x, x_new, g, s, scale_inv, v1, v2): [ncp_pad | X[Ppad] | Y[Ppad] | Z[Ppad]]  (points SoA)
//   V blocks: 6 arrays of Ppad (xx xy xz yy yz zz)
//   camera table: C x 48 doubles, recomputed per evaluation point, staged in LDS by every workgroup
//
// No MFMA: the blocks are 2x6 / 2x9 / 2x3, contraction depth 2-3.  The kernels are FP64 VALU + LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "ba_math.h"
#include "trf_math.h"
#include "wg_binding.h"

namespace cba {

constexpr int BLOCK = 256;
constexpr int CHUNK = 256;
constexpr int WAVE = 64;

struct VecLayout {
  int ncp;      // number of camera parameters
  int ncp_pad;  // padded to a multiple of 32 doubles
  int P;        // number of world points
  int Ppad;     // padded to a multiple of 32
  __host__ __device__ long total() const { return (long)ncp_pad + 3L * Ppad; }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = WAVE / 2; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, WAVE));
  return v;
}
// Device-side stamps of the kernels of one fused iteration (profiling build only: -DCBA_PROFILING, CBA_STAMPS=1).  A kernel trace under rocprofv3
// perturbs exactly what is asked about — the time BETWEEN dependent launches — so every workgroup of an instrumented kernel leaves the 100 MHz
// wall clock at its entry and every wave at its exit; the host takes min / max per kernel when the handle goes (cba_lib.hip: dump_stamps).
#ifdef CBA_PROFILING
constexpr int STAMP_SLOTS = 96, STAMP_BLOCKS = 1024, STAMP_WAVES = 16, STAMP_ROW = 1 + STAMP_WAVES;
__device__ long long* g_cba_stamps = nullptr;  // [STAMP_SLOTS][STAMP_BLOCKS][STAMP_ROW]
struct StampScope {
  long long* row;
  __device__ __forceinline__ explicit StampScope(int slot) : row(nullptr) {
    long long* base = g_cba_stamps;
    const int b = (int)(blockIdx.y * gridDim.x + blockIdx.x);
    if (base && slot < STAMP_SLOTS && b < STAMP_BLOCKS) {
      row = base + ((long)slot * STAMP_BLOCKS + b) * STAMP_ROW;
      if (threadIdx.x == 0 && threadIdx.y == 0) row[0] = wall_clock64();
    }
  }
  __device__ __forceinline__ ~StampScope() {
    const int t = (int)(threadIdx.y * blockDim.x + threadIdx.x);
    if (row && (t & (WAVE - 1)) == 0 && t / WAVE < STAMP_WAVES) row[1 + t / WAVE] = wall_clock64();
  }
};
#define CBA_STAMP(slot) StampScope cba_stamp_scope_(slot)
#else
#define CBA_STAMP(slot) ((void)0)
#endif
enum StampId { ST_TPREP = 0, ST_PAIRS, ST_REG_REDUCE, ST_FINALIZE, ST_CHOL_APPLY, ST_BACKSUB, ST_STEP_CAM, ST_BUILD, ST_REDUCE_PUB, ST_SCALE_LIN, ST_JV,
               ST_SMALL_SOLVE, ST_CHOL_STEP /* + k + 1 */ };

// Sum over the workgroup; result valid in thread 0.  `sh` holds BLOCK/WAVE doubles.
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < BLOCK / WAVE; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ double block_max(double v, double* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < BLOCK / WAVE; ++i) r = fmax(r, sh[i]);
  return r;
}

__device__ __forceinline__ void lds_add(double* addr, double v) { unsafeAtomicAdd(addr, v); }

// LDS copy of the camera table.  Rows are padded to 49 doubles: with the natural stride of 48 doubles
// (96 dwords == 32 mod 64 banks) the lanes of a wave, each reading the same field of a different camera,
// would land on two bank pairs (32-way conflict); an odd stride spreads 32 cameras over all bank pairs.
constexpr int CAMTAB_LIVE = 36;               // doubles of a CamTab row in use (everything up to and including pad[0], the camera's parameter offset)
constexpr int CAMTAB_LDS = CAMTAB_LIVE + 1;    // 37 (odd): the padding stays in HBM (49 doubles per camera cost k_build / k_tprep their third workgroup per CU)
// The per-observation kernels walk their chunks with the NEXT chunk's observation record already in flight:
// registers for (u, v, camera, point) of chunk n + 1 are loaded (unconditionally, index clamped) before chunk n is
// processed, so each workgroup sees the streaming-load latency once instead of once per chunk.
struct ObsRec { double u, v; int cam, pt; };
__device__ __forceinline__ ObsRec load_obs(const double* __restrict__ obs_u, const double* __restrict__ obs_v,
                                           const int* __restrict__ obs_cam, const int* __restrict__ obs_pt, int i) {
  ObsRec r;
  r.u = obs_u[i]; r.v = obs_v[i]; r.cam = obs_cam[i]; r.pt = obs_pt[i];
  return r;
}

__device__ __forceinline__ void stage_camtab(double* sh_tab, const double* tab, int n_cams) {
  for (int i = threadIdx.x; i < n_cams * CAMTAB_LIVE; i += BLOCK)
    sh_tab[(i / CAMTAB_LIVE) * CAMTAB_LDS + (i % CAMTAB_LIVE)] = tab[(i / CAMTAB_LIVE) * CAMTAB_DOUBLES + (i % CAMTAB_LIVE)];
}
__device__ __forceinline__ const CamTab& cam_at(const double* sh_tab, int cam) {
  return *reinterpret_cast<const CamTab*>(sh_tab + cam * CAMTAB_LDS);
}
// CAMG variants of the per-observation kernels read the camera table from global memory (through the vector cache) instead of staging it in
// LDS: chosen by cba_create when the LDS copy (296 B per camera) would not fit next to the kernel's other LDS data, so that the camera count is
// bounded by the per-camera accumulators of the linearisation alone (reference: no limit, core/reprojection.py:75-119 loops over cameras).
template <bool CAMG>
__device__ __forceinline__ const CamTab& cam_of(const double* sh_tab, const double* __restrict__ tab, int cam) {
  if constexpr (CAMG) return *reinterpret_cast<const CamTab*>(tab + (long)cam * CAMTAB_DOUBLES);
  else return cam_at(sh_tab, cam);
}

// ------------------------------------------------------------------------------------------------
// per-camera constants of an evaluation point
__global__ void k_cam_prep(const double* __restrict__ xvec, const double* __restrict__ cam_const,
                           const int* __restrict__ cam_model, const int* __restrict__ cam_np,
                           const int* __restrict__ cam_off, int n_cams, double* __restrict__ tab) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  double xc[MAX_NC];
  const int np = cam_np[c];
  for (int i = 0; i < MAX_NC; ++i) xc[i] = (i < np) ? xvec[cam_off[c] + i] : 0.0;
  CamTab t;
  cam_prepare(xc, cam_const + c * CAM_CONST_STRIDE, cam_model[c], np, &t, cam_off[c]);
  const double* src = reinterpret_cast<const double*>(&t);
  for (int i = 0; i < CAMTAB_DOUBLES; ++i) tab[c * CAMTAB_DOUBLES + i] = src[i];
}

// Start of a solve (cba_begin / cba_restart, round 6): x <- x0 (and the trial buffer, which the fused trial build only writes where it has
// observations), scale 1, bound scaling and flags cleared, and the camera table of x0 — ONE launch where two copies, two fills, k_fill and k_cam_prep
// were six (~22 us of a solve that is 0.5 ms on the reference's own sessions).  The table is prepared from x0 itself by the first threads.
__global__ void __launch_bounds__(BLOCK)
k_begin(const double* __restrict__ x0, double* __restrict__ x, double* __restrict__ x_new, double* __restrict__ sinv, double* __restrict__ cam_diag,
        int* __restrict__ flags, long total, int ncp_pad, const double* __restrict__ cam_const, const int* __restrict__ cam_model,
        const int* __restrict__ cam_np, const int* __restrict__ cam_off, int n_cams, double* __restrict__ tab) {
  const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = t0; i < total; i += (long)gridDim.x * blockDim.x) {
    const double v = x0[i];
    x[i] = v;
    if (x_new) x_new[i] = v;
    sinv[i] = 1.0;
    if (i < ncp_pad) cam_diag[i] = 0.0;
    if (i < 4) flags[i] = 0;
  }
  if (t0 < n_cams) {
    const int c = (int)t0;
    double xc[MAX_NC];
    const int np = cam_np[c];
    for (int i = 0; i < MAX_NC; ++i) xc[i] = (i < np) ? x0[cam_off[c] + i] : 0.0;
    CamTab t;
    cam_prepare(xc, cam_const + c * CAM_CONST_STRIDE, cam_model[c], np, &t, cam_off[c]);
    const double* src = reinterpret_cast<const double*>(&t);
    for (int i = 0; i < CAMTAB_DOUBLES; ++i) tab[c * CAMTAB_DOUBLES + i] = src[i];
  }
}

// ------------------------------------------------------------------------------------------------
// cost (and optionally the residual vector) at xvec:   0.5 * sum rho  is formed by the caller
template <bool WRITE_R, bool CAMG = false>
__global__ void __launch_bounds__(BLOCK)
k_cost(const double* __restrict__ obs_u, const double* __restrict__ obs_v, const int* __restrict__ obs_cam,
       const int* __restrict__ obs_pt, long n_obs, const double* __restrict__ xvec, VecLayout lay,
       const double* __restrict__ tab, int n_cams, int loss, double f_scale, double* __restrict__ partial,
       int* __restrict__ flags, double* __restrict__ r_out, const int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* sh_tab = sh;
  double* sh_red = sh + (CAMG ? 0 : n_cams * CAMTAB_LDS);
  if (!CAMG) stage_camtab(sh_tab, tab, n_cams);
  __syncthreads();
  const double* px = xvec + lay.ncp_pad;
  double acc = 0.0;
  bool bad = false;
  // two-stage software pipeline, as in k_jv: the record of trip i + 2 and the point of trip i + 1 are in flight while trip i is evaluated (the residual
  // hook's stores, WRITE_R, are waited for with them: that variant runs once per report, not per iteration)
  struct Rec { int cam, pt; double u, v; long o; };
  const long stride = (long)gridDim.x * BLOCK, last = n_obs - 1;
  auto load_rec = [&](long i) {
    Rec r; const long k = min(i, last);
    r.cam = obs_cam[k]; r.pt = obs_pt[k]; r.u = obs_u[k]; r.v = obs_v[k]; r.o = WRITE_R ? (long)order[k] : 0;
    return r;
  };
  long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  Rec rc = load_rec(i), rn = load_rec(i + stride);
  double X = px[rc.pt], Y = px[lay.Ppad + rc.pt], Z = px[2 * lay.Ppad + rc.pt];
  for (; i < n_obs; i += stride) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(rc.cam), "+v"(rc.pt), "+v"(rc.u), "+v"(rc.v), "+v"(rc.o));
    asm volatile("" : "+v"(rn.cam), "+v"(rn.pt), "+v"(rn.u), "+v"(rn.v), "+v"(rn.o));
    asm volatile("" : "+v"(X), "+v"(Y), "+v"(Z));
    const double Xn = px[rn.pt], Yn = px[lay.Ppad + rn.pt], Zn = px[2 * lay.Ppad + rn.pt];
    const Rec rnn = load_rec(i + 2 * stride);
    __builtin_amdgcn_sched_barrier(0);
    double e[2];
    project_residual(cam_of<CAMG>(sh_tab, tab, rc.cam), X, Y, Z, rc.u, rc.v, e);
    if (!(isfinite(e[0]) && isfinite(e[1]))) bad = true;
    acc += robust_cost_one(loss, f_scale, e[0]) + robust_cost_one(loss, f_scale, e[1]);
    if (WRITE_R) {
      r_out[2 * rc.o] = e[0];
      r_out[2 * rc.o + 1] = e[1];
    }
    rc = rn; rn = rnn; X = Xn; Y = Yn; Z = Zn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (bad) flags[0] = 1;
  const double tot = block_sum(acc, sh_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// out[j] = sum_b partial[b*width + j]  (fixed order => deterministic).  Launch with blockDim = (64, REDUCE_RY):
// x indexes columns (coalesced), y splits the rows REDUCE_RY ways (two interleaved chains each) so that the chain of
// dependent loads per thread stays short (512 rows: 16 loads deep); partial sums meet in LDS in a fixed order.
constexpr int REDUCE_RY = 16;
__device__ __forceinline__ void reduce_rows_block(const double* __restrict__ partial, int nrow, int width, double* __restrict__ out,
                                                  double* __restrict__ grad_out, const int* __restrict__ cam_off, const int* __restrict__ cam_np,
                                                  int stride, int tri) {
  __shared__ double sh[REDUCE_RY][64];
  const int j = blockIdx.x * 64 + threadIdx.x;
  // eight independent chains per thread (round 6: with two, 512 rows were sixteen dependent load-add steps, ~8 us of latency for 7 MB)
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (j < width) {
    int b = threadIdx.y;
    for (; b + 7 * REDUCE_RY < nrow; b += 8 * REDUCE_RY) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += partial[(long)(b + q * REDUCE_RY) * width + j];
    }
    for (; b < nrow; b += REDUCE_RY) acc[0] += partial[(long)b * width + j];
  }
  sh[threadIdx.y][threadIdx.x] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (threadIdx.y == 0 && j < width) {
    double tot = 0.0;
#pragma unroll
    for (int y = 0; y < REDUCE_RY; ++y) tot += sh[y][threadIdx.x];
    out[j] = tot;
    if (grad_out) {  // packed camera blocks: the gradient entries go to the vector as well (k_unpack_camera_grad, single-rank solves)
      const int c = j / stride, r = j % stride - tri;
      if (r >= 0 && r < cam_np[c]) grad_out[cam_off[c] + r] = tot;
    }
  }
}
__global__ void __launch_bounds__(64 * REDUCE_RY)
k_reduce_rows(const double* __restrict__ partial, int nrow, int width, double* __restrict__ out,
              double* __restrict__ grad_out = nullptr, const int* __restrict__ cam_off = nullptr, const int* __restrict__ cam_np = nullptr,
              int stride = 1, int tri = 0) {
  reduce_rows_block(partial, nrow, width, out, grad_out, cam_off, cam_np, stride, tri);
}
// Single-rank fused iteration (round 5): the end-of-iteration packet (k_publish: one workgroup, 4-5 us + a launch gap) leaves with the LAST workgroup
// of the launch that reduces the trial build's camera blocks — the packet needs none of that reduction's results (scalars, the trial cost rows and
// the step-norm rows are all there before the launch), so the two run side by side.  Same packet as k_publish; the two column sums are taken by
// 1024 threads in this kernel's own fixed order.
struct PubArgs {
  double* scal; int n_scal; int* flags; double* host_scal; int* host_flags; unsigned long long seq;
  const double* part_a; int rows_a, slot_a; const double* part_b; int rows_b, slot_b;
  // bounded fused iteration: the camera blocks of x, g, the Jacobi scale and the damped step travel with the packet (the host driver keeps scipy's
  // select_step and the later trials of the iteration; four vectors of <= 1152 doubles instead of three host round trips for them)
  const double* cam_src[4]; double* cam_dst; int ncp;
};
__global__ void __launch_bounds__(64 * REDUCE_RY)
k_reduce_rows_pub(const double* __restrict__ partial, int nrow, int width, double* __restrict__ out, double* __restrict__ grad_out,
                  const int* __restrict__ cam_off, const int* __restrict__ cam_np, int stride, int tri, PubArgs pub) {
  CBA_STAMP(ST_REDUCE_PUB);
  if (blockIdx.x + 1 < gridDim.x) { reduce_rows_block(partial, nrow, width, out, grad_out, cam_off, cam_np, stride, tri); return; }
  __shared__ double sh_w[2][REDUCE_RY];
  const int t = threadIdx.y * 64 + threadIdx.x;
  const double own = (t < pub.n_scal) ? pub.scal[t] : 0.0;  // (requested with the partial rows, not behind their sums)
  double a = 0.0, b = 0.0;
  for (int r = t; r < pub.rows_a; r += 64 * REDUCE_RY) a += pub.part_a[r];
  for (int r = t; r < pub.rows_b; r += 64 * REDUCE_RY) b += pub.part_b[r];
  a = wave_sum(a); b = wave_sum(b);
  if (threadIdx.x == 0) { sh_w[0][threadIdx.y] = a; sh_w[1][threadIdx.y] = b; }
  __syncthreads();
  __shared__ double sh_tot[2];
  if (t == 0) {
    double ta = 0.0, tb = 0.0;
    for (int i = 0; i < REDUCE_RY; ++i) { ta += sh_w[0][i]; tb += sh_w[1][i]; }
    pub.scal[pub.slot_a] = ta; pub.scal[pub.slot_b] = tb;
    sh_tot[0] = ta; sh_tot[1] = tb;
  }
  __syncthreads();
  if (t < pub.n_scal) pub.host_scal[t] = (t == pub.slot_a) ? sh_tot[0] : (t == pub.slot_b) ? sh_tot[1] : own;
  if (t < 4) { pub.host_flags[t] = pub.flags[t]; pub.flags[t] = 0; }
  if (pub.cam_dst)
    for (int e = t; e < 4 * pub.ncp; e += 64 * REDUCE_RY) pub.cam_dst[e] = pub.cam_src[e / pub.ncp][e % pub.ncp];
  __threadfence_system();
  __syncthreads();
  if (t == 0) reinterpret_cast<volatile unsigned long long*>(pub.host_scal)[63] = pub.seq;
}
// Sharded solves: every host-visible primitive ends with ONE all-reduce of the scalars it produced.  k_xpack gathers
// the scal slots named by `mask`, the four flags (as 0/1 doubles) and, when asked, this rank's max |g| (scal[4]) in
// its own slot of a world-sized tail, so that a single sum-reduction carries sums, "any rank" flags and a maximum;
// k_xunpack writes the results back where the single-rank code expects them.
__global__ void k_xpack(const double* __restrict__ scal, const int* __restrict__ flags, unsigned long long mask, int with_max,
                        int rank, int world, double* __restrict__ xbuf) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0;
  for (int s2 = 0; s2 < 32; ++s2)
    if ((mask >> s2) & 1ull) xbuf[n++] = scal[s2];
  for (int k = 0; k < 4; ++k) xbuf[n++] = flags[k] ? 1.0 : 0.0;
  if (with_max)
    for (int r = 0; r < world; ++r) xbuf[n++] = (r == rank) ? scal[4] : 0.0;
}
__global__ void k_xunpack(const double* __restrict__ xbuf, unsigned long long mask, int with_max, int world,
                          double* __restrict__ scal, int* __restrict__ flags) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0;
  for (int s2 = 0; s2 < 32; ++s2)
    if ((mask >> s2) & 1ull) scal[s2] = xbuf[n++];
  for (int k = 0; k < 4; ++k) flags[k] = xbuf[n++] > 0.5 ? 1 : 0;
  if (with_max) {
    double m = 0.0;
    for (int r = 0; r < world; ++r) m = fmax(m, xbuf[n++]);
    scal[4] = m;
  }
}

// narrow case (width <= 4): one workgroup, every thread strides over the rows
template <bool MAX>
__global__ void __launch_bounds__(BLOCK)
k_reduce_narrow(const double* __restrict__ partial, int nrow, int width, double* __restrict__ out) {
  __shared__ double sh_red[BLOCK / WAVE];
  for (int j = 0; j < width; ++j) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nrow; b += BLOCK) {
      const double v = partial[(long)b * width + j];
      s = MAX ? fmax(s, v) : s + v;
    }
    const double r = MAX ? block_max(s, sh_red) : block_sum(s, sh_red);
    if (threadIdx.x == 0) out[j] = r;
  }
}

// Shared front end of the per-observation passes: project, differentiate, apply the robust-loss scaling.
// After the call e, A, B are the rows of scipy's scaled (J, f); returns this observation's rho-sum.
template <int NC>
__device__ __forceinline__ double obs_linearize(const CamTab& c_lds, double X, double Y, double Z, double u, double v,
                                                int loss, double f_scale, double* e, double (*A)[MAX_NC],
                                                double (*B)[3]) {
  // The camera row lives in LDS.  Copy it to registers in ONE batch of reads: left alone, the compiler issues each
  // ds_read right before its first use and waits for it, ~20 exposed LDS round trips per observation.
  const CamTab c = c_lds;
  __builtin_amdgcn_sched_barrier(0);
  project_full(c, X, Y, Z, u, v, e, A, B);
  double rho = 0.0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double rs, er;
    rho += robust_one(loss, f_scale, e[r], &rs, &er);
    e[r] = er;
    if (loss != LOSS_LINEAR) {
#pragma unroll
      for (int k = 0; k < NC; ++k) A[r][k] *= rs;
      B[r][0] *= rs; B[r][1] *= rs; B[r][2] *= rs;
    }
  }
  return rho;
}

// The same front end in FACTORED form (ba_math.h: project_factors): G, Y = R X, B = G R and the intrinsic columns, robust-loss scaling applied to the
// rows; the camera row is copied from LDS WITHOUT its J_l (27 of 36 doubles: the table reads of the per-observation kernels are random gathers into the
// LDS banks, ~2.5 cycles per access group, and were 19 of k_jv's and 38 of k_tprep's microseconds of LDS time on cfg4 — r05 SQ counters).
__device__ __forceinline__ double obs_factors(const CamTab& c_lds, double X, double Y, double Z, double u, double v, int loss, double f_scale, double* e,
                                              double (*G)[3], double* Yr, double (*Aint)[3], double (*B)[3]) {
  CamTab c;
  {
    const double* src = reinterpret_cast<const double*>(&c_lds);
    double* dst = reinterpret_cast<double*>(&c);
#pragma unroll
    for (int i = 0; i < 12; ++i) dst[i] = src[i];          // R, t
#pragma unroll
    for (int i = 21; i < 35; ++i) dst[i] = src[i];         // fx .. nparams (J_l, entries 12..20, is not read)
  }
  __builtin_amdgcn_sched_barrier(0);
  project_factors(c, X, Y, Z, u, v, e, G, Yr, Aint, B);
  double rho = 0.0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double rs, er;
    rho += robust_one(loss, f_scale, e[r], &rs, &er);
    e[r] = er;
    if (loss != LOSS_LINEAR) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { G[r][k] *= rs; B[r][k] *= rs; Aint[r][k] *= rs; }
    }
  }
  return rho;
}

template <int NC> struct UPack {
  static constexpr int TRI = NC * (NC + 1) / 2;
  static constexpr int STRIDE = TRI + NC;  // upper triangle + gradient
  __host__ __device__ static constexpr int idx(int r, int c) { return r * NC - r * (r - 1) / 2 + (c - r); }
};

// ------------------------------------------------------------------------------------------------
// Deterministic per-camera sums (cba_options.deterministic).  The default kernels add an observation's share of U_c, g_c
// (k_build) and of the rhs (k_tprep) with FP64 LDS atomics, whose order — and with it the last bits of the sums, and now
// and then the evaluation count of a solve — changes from run to run; the reference is bit-reproducible (single-threaded
// scipy, capture_volume.py:387).  Here a chunk's values are parked in LDS, nine per observation and round, and summed per
// camera in the chunk's FIXED camera-sorted order (cba_create lays out, per chunk, the observation order by camera and the
// camera offsets); every thread owns the running sums of its (camera, value) tasks for the whole kernel, so the
// per-workgroup partials and everything downstream (k_reduce_rows, fixed order) are reproducible bit for bit.
constexpr int DET_ROUND = 9;              // values per observation and round
constexpr int DET_LD = CHUNK + 1;         // row stride of the parking area (odd: the nine rows of a task group hit nine banks)
struct DetPlan {
  const unsigned char* perm;    // [n_chunks][CHUNK] chunk-local observation indices, sorted by camera
  const unsigned short* cstart; // [n_chunks][C + 1] offsets into perm
};
// one round: `val[q]` of this thread's observation (zeros when it has none) -> per-camera sums into acc[m], task = tid + BLOCK * m,
// camera = task / DET_ROUND, value = task % DET_ROUND.  sh_cv: DET_ROUND * DET_LD doubles; sh_perm / sh_cs: the chunk's tables.
template <int M>
__device__ __forceinline__ void det_round(const double* val, double* sh_cv, const int* sh_perm, const int* sh_cs, int n_cams, double* acc) {
#pragma unroll
  for (int q = 0; q < DET_ROUND; ++q) sh_cv[q * DET_LD + threadIdx.x] = val[q];
  __syncthreads();
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int task = (int)threadIdx.x + BLOCK * m;
    if (task < n_cams * DET_ROUND) {
      const int c = task / DET_ROUND, q = task % DET_ROUND;
      double s = 0.0;
      for (int j = sh_cs[c]; j < sh_cs[c + 1]; ++j) s += sh_cv[q * DET_LD + sh_perm[j]];
      acc[m] += s;
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// build pass: residuals + Jacobian blocks -> per-point V_p, g_p (segmented sums through LDS), per-camera
// U_c, g_c (LDS atomics, flushed as per-workgroup partials), cost partials.
// value v (position in the packed camera block: upper triangle row by row, then the gradient) of one observation
template <int NC>
__device__ __forceinline__ double cam_block_value(int v, const double (*A)[MAX_NC], const double* e, int np) {
  using UP = UPack<NC>;
  int idx = 0;
#pragma unroll
  for (int r = 0; r < NC; ++r)
#pragma unroll
    for (int c = r; c < NC; ++c, ++idx)
      if (idx == v) return (r < np && c < np) ? A[0][r] * A[0][c] + A[1][r] * A[1][c] : 0.0;
#pragma unroll
  for (int r = 0; r < NC; ++r)
    if (UP::TRI + r == v) return (r < np) ? A[0][r] * e[0] + A[1][r] * e[1] : 0.0;
  return 0.0;
}

// CAMG: the camera table stays in global memory (48 doubles per camera, read through the vector cache) instead of LDS: for nine-parameter cameras
// and more than ~64 of them the table, the per-camera accumulators and the point stage together exceed half of the LDS, and the kernel ran one
// workgroup per CU.
template <int NC, int DETM = 0, bool CAMG = false>  // DETM > 0: deterministic variant, DETM tasks per thread and round (cameras <= DETM * 256 / 9)
__global__ void __launch_bounds__(BLOCK)
k_build(const double* __restrict__ obs_u, const double* __restrict__ obs_v, const int* __restrict__ obs_cam,
        const int* __restrict__ obs_pt, const int* __restrict__ pt_start, const int* __restrict__ chunk_start,
        const int* __restrict__ chunk_pts, int n_chunks, const double* __restrict__ xvec, VecLayout lay, const double* __restrict__ tab, int n_cams,
        int loss, double f_scale, double* __restrict__ Vblk, double* __restrict__ gvec,
        double* __restrict__ partialU, double* __restrict__ partial_cost, int* __restrict__ flags, const double* __restrict__ skip,
        DetPlan det = DetPlan{nullptr, nullptr}) {
  using UP = UPack<NC>;
  constexpr bool DET = DETM > 0;
  if (skip && *skip != 0.0) return;  // fused step without a trial (k_fused_subspace handed the iteration to the host)
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* sh_tab = sh;
  double* sh_U = sh_tab + (CAMG ? 0 : n_cams * CAMTAB_LDS);            // DET: the parking area of det_round instead
  auto cam_of = [&](int cam) -> const CamTab& {
    return CAMG ? *reinterpret_cast<const CamTab*>(tab + (long)cam * CAMTAB_DOUBLES) : cam_at(sh_tab, cam);
  };
  double* sh_pt = sh_U + (DET ? DET_ROUND * DET_LD : n_cams * UP::STRIDE);
  double* sh_red = sh_pt + 9 * CHUNK;
  int* sh_perm = reinterpret_cast<int*>(sh_red + 8);                   // DET: [CHUNK] + [n_cams + 1]
  int* sh_cs = sh_perm + CHUNK;
  constexpr int DROUNDS = (UP::STRIDE + DET_ROUND - 1) / DET_ROUND;    // 3 / 6
  constexpr int DM = DET ? DETM : 1;                                   // tasks per thread and round: ceil(C * 9 / 256)
  double dacc[DROUNDS][DM];
#pragma unroll
  for (int r = 0; r < DROUNDS; ++r)
#pragma unroll
    for (int m = 0; m < DM; ++m) dacc[r][m] = 0.0;
  if (!CAMG) stage_camtab(sh_tab, tab, n_cams);
  if (!DET)
    for (int i = threadIdx.x; i < n_cams * UP::STRIDE; i += BLOCK) sh_U[i] = 0.0;
  __syncthreads();
  const double* px = xvec + lay.ncp_pad;
  double* gp = gvec + lay.ncp_pad;
  double cost = 0.0;
  bool bad = false;
  const int last_obs = max(chunk_start[n_chunks] - 1, 0);
  int ch = blockIdx.x;
  int o0 = 0, o1 = 0;
  ObsRec cur = {0.0, 0.0, 0, 0};
  if (ch < n_chunks) {
    o0 = chunk_start[ch]; o1 = chunk_start[ch + 1];
    cur = load_obs(obs_u, obs_v, obs_cam, obs_pt, min(o0 + (int)threadIdx.x, last_obs));
  }
  while (ch < n_chunks) {
    const int nxt = ch + gridDim.x, nc = min(nxt, n_chunks - 1);
    const int no0 = chunk_start[nc], no1 = chunk_start[nc + 1];
    const ObsRec nx = load_obs(obs_u, obs_v, obs_cam, obs_pt, min(no0 + (int)threadIdx.x, last_obs));
    // per-point phase inputs of point cp0 + tid, fetched ahead of the barrier they are used behind
    const int cp0 = chunk_pts[2 * ch], npts = chunk_pts[2 * ch + 1];
    const int pp = min(cp0 + (int)threadIdx.x, lay.P - 1);
    const int pa = pt_start[pp] - o0, pb = pt_start[pp + 1] - o0;
    const int i = o0 + threadIdx.x;
    double pv[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) pv[q] = 0.0;
    double e[2] = {0.0, 0.0}, A[2][MAX_NC];
    int np_det = 0;
    if (DET) {
#pragma unroll
      for (int k = 0; k < MAX_NC; ++k) { A[0][k] = 0.0; A[1][k] = 0.0; }
      sh_perm[threadIdx.x] = det.perm[(long)ch * CHUNK + threadIdx.x];
      for (int q = threadIdx.x; q <= n_cams; q += BLOCK) sh_cs[q] = det.cstart[(long)ch * (n_cams + 1) + q];
    }
    if (i < o1) {
      const int cam = cur.cam, pt = cur.pt;
      double B[2][3];
      cost += obs_linearize<NC>(cam_of(cam), px[pt], px[lay.Ppad + pt], px[2 * lay.Ppad + pt], cur.u, cur.v, loss,
                                f_scale, e, A, B);
      if (!isfinite(e[0] + e[1])) bad = true;  // a trial point built directly by this pass (fused step): scipy's isfinite(f_new) test
      pv[0] = B[0][0] * B[0][0] + B[1][0] * B[1][0];
      pv[1] = B[0][0] * B[0][1] + B[1][0] * B[1][1];
      pv[2] = B[0][0] * B[0][2] + B[1][0] * B[1][2];
      pv[3] = B[0][1] * B[0][1] + B[1][1] * B[1][1];
      pv[4] = B[0][1] * B[0][2] + B[1][1] * B[1][2];
      pv[5] = B[0][2] * B[0][2] + B[1][2] * B[1][2];
      pv[6] = B[0][0] * e[0] + B[1][0] * e[1];
      pv[7] = B[0][1] * e[0] + B[1][1] * e[1];
      pv[8] = B[0][2] * e[0] + B[1][2] * e[1];
      const int np = (int)cam_of(cam).nparams;
      np_det = np;
      if (!DET) {
        double* Uc = sh_U + cam * UP::STRIDE;
#pragma unroll
        for (int r = 0; r < NC; ++r) {
          if (r < np) {
#pragma unroll
            for (int c = r; c < NC; ++c)
              if (c < np) lds_add(&Uc[UP::idx(r, c)], A[0][r] * A[0][c] + A[1][r] * A[1][c]);
            lds_add(&Uc[UP::TRI + r], A[0][r] * e[0] + A[1][r] * e[1]);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) sh_pt[q * CHUNK + threadIdx.x] = pv[q];
    __syncthreads();
    auto reduce_point = [&](int p, int a, int b) {
      double acc[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[q] = 0.0;
      for (int j = a; j < b; ++j) {
#pragma unroll
        for (int q = 0; q < 9; ++q) acc[q] += sh_pt[q * CHUNK + j];
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) Vblk[(long)q * lay.Ppad + p] = acc[q];
      gp[p] = acc[6];
      gp[lay.Ppad + p] = acc[7];
      gp[2 * lay.Ppad + p] = acc[8];
    };
    if (npts < 0 && threadIdx.x < 9) {
      // fragment of a heavy point (more observations than a chunk holds): the whole chunk is point cp0; its sums go
      // to V / g by atomics (zeroed by k_zero_heavy before the pass)
      double acc = 0.0;
      for (int j = 0; j < o1 - o0; ++j) acc += sh_pt[threadIdx.x * CHUNK + j];
      if (threadIdx.x < 6) unsafeAtomicAdd(&Vblk[(long)threadIdx.x * lay.Ppad + cp0], acc);
      else unsafeAtomicAdd(&gp[(long)(threadIdx.x - 6) * lay.Ppad + cp0], acc);
    }
    if ((int)threadIdx.x < npts && pb > pa) reduce_point(pp, pa, pb);
    for (int lp = threadIdx.x + BLOCK; lp < npts; lp += BLOCK) {  // a range padded by unobserved points: rare
      const int p = cp0 + lp;
      const int a = pt_start[p] - o0, b = pt_start[p + 1] - o0;
      if (b > a) reduce_point(p, a, b);
    }
    __syncthreads();
    if (DET) {  // camera blocks: fixed-order sums, nine values per round (np_det == 0: no observation, all values zero)
#pragma unroll
      for (int rd = 0; rd < DROUNDS; ++rd) {
        double val[DET_ROUND];
#pragma unroll
        for (int q = 0; q < DET_ROUND; ++q) val[q] = cam_block_value<NC>(rd * DET_ROUND + q, A, e, np_det);
        det_round<DM>(val, sh_U, sh_perm, sh_cs, n_cams, dacc[rd]);
      }
    }
    cur = nx; o0 = no0; o1 = no1; ch = nxt;
  }
  double* dst = partialU + (long)blockIdx.x * n_cams * UP::STRIDE;
  if (DET) {
#pragma unroll
    for (int rd = 0; rd < DROUNDS; ++rd)
#pragma unroll
      for (int m = 0; m < DM; ++m) {
        const int task = (int)threadIdx.x + BLOCK * m;
        const int c = task / DET_ROUND, v = rd * DET_ROUND + task % DET_ROUND;
        if (task < n_cams * DET_ROUND && v < UP::STRIDE) dst[c * UP::STRIDE + v] = dacc[rd][m];
      }
  } else {
    for (int i = threadIdx.x; i < n_cams * UP::STRIDE; i += BLOCK) dst[i] = sh_U[i];
  }
  const double tot = block_sum(cost, sh_red);
  if (threadIdx.x == 0) partial_cost[blockIdx.x] = tot;
  if (bad) flags[0] = 1;
}

// ------------------------------------------------------------------------------------------------
// k_build_cs: the build pass over CAMERA-SORTED super-chunks.  k_build adds an observation's share of U_c, g_c with nc (nc + 3) / 2 FP64 LDS
// atomics (27 / 54 per observation: ~28 clocks per wave instruction, 40 of the 90 us of the pass on cfg4, over half of it for nine-parameter
// cameras).  Here a workgroup takes a SUPER-CHUNK — a run of consecutive chunks, i.e. of whole points; cba_create sizes them so that every
// workgroup gets the same number of them (~2000 observations for six-, ~4000 for nine-parameter cameras, at most CS_MAX_PTS points) — whose
// observations it has laid out a second time sorted by camera.  A thread walks R = ceil(n / 256) CONSECUTIVE
// observations of that order: they belong to one camera (at most a few), so U_c, g_c accumulate in REGISTERS and go to the workgroup's LDS copy
// once per camera change: ~2 atomics per observation instead of 27.  What the point order gave for free now costs atomics: V_p, g_p (9 values per
// observation) are added to per-point LDS slots of the super-chunk and written out once — 11 atomics per observation instead of 27 (19 instead of 54).
// Same outputs as k_build (V, g point part, per-workgroup partials of the packed camera blocks, cost partials, flags), same arithmetic per
// observation; the order of the sums differs, as it does between two runs of k_build.
constexpr int CS_MAX_PTS = 512;    // points per super-chunk at most (9 + 3 doubles of LDS each; the launch sizes its LDS for the largest super-chunk: CsPlan::pmax)
struct CsPlan {
  const double* u;        // [N] observations in super-chunk / camera order
  const double* v;
  const int* cam;
  const int* ptl;         // point index local to the super-chunk
  const int* obs_start;   // [n_sc + 1]
  const int* pt_first;    // [n_sc]
  const int* pt_count;    // [n_sc]
  int n_sc;
  int pmax;               // points of the largest super-chunk, rounded up to a multiple of 32: row stride of the per-point LDS arrays
};
// UGLOB (more cameras than the LDS holds packed blocks for: > ~650 six- / ~320 nine-parameter cameras; the reference has no limit,
// core/reprojection.py:75-119): a thread's register sums go to ONE global copy of the blocks by FP64 global atomics (the caller zeroes it; a thread
// flushes once per camera change, ~13 atomics per observation at 1000 cameras) instead of to the workgroup's LDS copy; partialU is that copy.
// TRIAL (single-rank fused iteration): the pass evaluates the TRIAL point and forms its point entries itself while staging a super-chunk's points —
// x_new = x + alpha g / sinv^2 + beta s with (alpha, beta) from k_step_cam — writes them to x_new for the passes behind it and leaves its share of
// ||step||^2 in step_partial[workgroup]: k_trial_update's pass over five vectors is gone (xvec is not read then).
struct TrialSrc {
  const double *x, *g, *sinv, *s;  // current point, gradient, scale, damped step
  const double* ab;                // device: alpha, beta
  double* x_new;
  double* step_partial;            // [grid]
};
template <int NC, bool CAMG = false, bool UGLOB = false, bool TRIAL = false>
__global__ void __launch_bounds__(BLOCK)
k_build_cs(CsPlan cs, const double* __restrict__ xvec, VecLayout lay, const double* __restrict__ tab, int n_cams, int loss, double f_scale,
           double* __restrict__ Vblk, double* __restrict__ gvec, double* __restrict__ partialU, double* __restrict__ partial_cost,
           int* __restrict__ flags, const double* __restrict__ skip, TrialSrc trial = TrialSrc{}) {
  CBA_STAMP(ST_BUILD);
  using UP = UPack<NC>;
  static_assert(!UGLOB || CAMG, "a camera count beyond the LDS copy of the blocks is beyond the LDS copy of the table as well");
  if (skip && *skip != 0.0) return;  // fused step without a trial (k_fused_subspace handed the iteration to the host)
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* sh_tab = sh;
  double* sh_U = sh_tab + (CAMG ? 0 : n_cams * CAMTAB_LDS);
  const int PM = cs.pmax;
  double* sh_vg = sh_U + (UGLOB ? 0 : n_cams * UP::STRIDE);        // [9][PM]
  double* sh_x = sh_vg + 9 * PM;                     // [3][PM]
  double* sh_red = sh_x + 3 * PM;
  if (!CAMG) stage_camtab(sh_tab, tab, n_cams);
  if (!UGLOB)
    for (int i = threadIdx.x; i < n_cams * UP::STRIDE; i += BLOCK) sh_U[i] = 0.0;
  const double* px = xvec + lay.ncp_pad;
  double* gp = gvec + lay.ncp_pad;
  double cost = 0.0, step_sq = 0.0;
  const double t_alpha = TRIAL ? trial.ab[0] : 0.0, t_beta = TRIAL ? trial.ab[1] : 0.0;
  bool bad = false;
  for (int s = blockIdx.x; s < cs.n_sc; s += gridDim.x) {
    const int o0 = cs.obs_start[s], o1 = cs.obs_start[s + 1], p0 = cs.pt_first[s], npts = cs.pt_count[s];
    __syncthreads();  // the previous super-chunk's sums have been written out (and, first pass, the table / U zeroing is done)
    for (int i = threadIdx.x; i < 9 * PM; i += BLOCK) sh_vg[i] = 0.0;
    if (TRIAL) {  // the point entries of the trial point are formed here (k_trial_update's work) and written out for the passes behind this one
      for (int i = threadIdx.x; i < npts; i += BLOCK) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const long e = (long)lay.ncp_pad + (long)k * lay.Ppad + p0 + i;
          const double si = trial.sinv[e];
          const double st = t_alpha * trial.g[e] / (si * si) + t_beta * trial.s[e];
          const double xn = trial.x[e] + st;
          trial.x_new[e] = xn;
          sh_x[k * PM + i] = xn;
          step_sq = fma(st, st, step_sq);
        }
      }
    } else {
      for (int i = threadIdx.x; i < npts; i += BLOCK) {
        sh_x[i] = px[p0 + i]; sh_x[PM + i] = px[lay.Ppad + p0 + i]; sh_x[2 * PM + i] = px[2 * lay.Ppad + p0 + i];
      }
    }
    __syncthreads();
    const int R = (o1 - o0 + BLOCK - 1) / BLOCK;
    const int j0 = o0 + (int)threadIdx.x * R, j1 = min(j0 + R, o1);
    double acc[UP::STRIDE];
#pragma unroll
    for (int q = 0; q < UP::STRIDE; ++q) acc[q] = 0.0;
    int cur_cam = -1, cur_np = 0;
    auto flush = [&]() {
      if (cur_cam < 0) return;
      double* Uc = (UGLOB ? partialU : sh_U) + (long)cur_cam * UP::STRIDE;
#pragma unroll
      for (int r = 0; r < NC; ++r) {
        if (r < cur_np) {
#pragma unroll
          for (int c = r; c < NC; ++c)
            if (c < cur_np) { if (UGLOB) atomicAdd(&Uc[UP::idx(r, c)], acc[UP::idx(r, c)]); else lds_add(&Uc[UP::idx(r, c)], acc[UP::idx(r, c)]); }
          if (UGLOB) atomicAdd(&Uc[UP::TRI + r], acc[UP::TRI + r]); else lds_add(&Uc[UP::TRI + r], acc[UP::TRI + r]);
        }
      }
#pragma unroll
      for (int q = 0; q < UP::STRIDE; ++q) acc[q] = 0.0;
    };
    // the record of observation j + 1 is in flight while j is linearised (round 6: read at the head of every trip, its ~1 us round trip was serial time
    // eight times per super-chunk and thread); one explicit wait per trip, the loaded values laundered for the reason given in k_jv
    const int jl = max(o1 - 1, 0);
    int n_cam = cs.cam[min(j0, jl)], n_pl = cs.ptl[min(j0, jl)];
    double n_u = cs.u[min(j0, jl)], n_v = cs.v[min(j0, jl)];
    for (int j = j0; j < j1; ++j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(n_cam), "+v"(n_pl), "+v"(n_u), "+v"(n_v));
      const int cam = n_cam, pl = n_pl;
      const double u = n_u, v = n_v;
      n_cam = cs.cam[min(j + 1, jl)]; n_pl = cs.ptl[min(j + 1, jl)]; n_u = cs.u[min(j + 1, jl)]; n_v = cs.v[min(j + 1, jl)];
      __builtin_amdgcn_sched_barrier(0);
      if (cam != cur_cam) { flush(); cur_cam = cam; cur_np = (int)cam_of<CAMG>(sh_tab, tab, cam).nparams; }
      // factored linearisation (round 6): the rows of A' = [Y x G_r | G_r | A_intr,r] are summed; A = A' P with P = blockdiag(J_l, I) per camera, so
      // U = P^T U' P and g = P^T g' are formed ONCE per camera when the workgroup writes its blocks out (UGLOB adds to a global copy: true rows there)
      double e[2], A[2][MAX_NC], B[2][3];
      {
        double G[2][3], Yr[3], Aint[2][3];
        const CamTab& ctc = cam_of<CAMG>(sh_tab, tab, cam);
        cost += obs_factors(ctc, sh_x[pl], sh_x[PM + pl], sh_x[2 * PM + pl], u, v, loss, f_scale, e, G, Yr, Aint, B);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const double p0 = Yr[1] * G[r][2] - Yr[2] * G[r][1], p1 = Yr[2] * G[r][0] - Yr[0] * G[r][2], p2 = Yr[0] * G[r][1] - Yr[1] * G[r][0];
          if constexpr (UGLOB) {
            A[r][0] = ctc.Jl[0] * p0 + ctc.Jl[3] * p1 + ctc.Jl[6] * p2;
            A[r][1] = ctc.Jl[1] * p0 + ctc.Jl[4] * p1 + ctc.Jl[7] * p2;
            A[r][2] = ctc.Jl[2] * p0 + ctc.Jl[5] * p1 + ctc.Jl[8] * p2;
          } else {
            A[r][0] = p0; A[r][1] = p1; A[r][2] = p2;
          }
          A[r][3] = G[r][0]; A[r][4] = G[r][1]; A[r][5] = G[r][2];
          A[r][6] = Aint[r][0]; A[r][7] = Aint[r][1]; A[r][8] = Aint[r][2];
        }
      }
      if (!isfinite(e[0] + e[1])) bad = true;
#pragma unroll
      for (int r = 0; r < NC; ++r) {
#pragma unroll
        for (int c = r; c < NC; ++c) acc[UP::idx(r, c)] = fma(A[1][r], A[1][c], fma(A[0][r], A[0][c], acc[UP::idx(r, c)]));
        acc[UP::TRI + r] = fma(A[1][r], e[1], fma(A[0][r], e[0], acc[UP::TRI + r]));
      }
      lds_add(&sh_vg[0 * PM + pl], B[0][0] * B[0][0] + B[1][0] * B[1][0]);
      lds_add(&sh_vg[1 * PM + pl], B[0][0] * B[0][1] + B[1][0] * B[1][1]);
      lds_add(&sh_vg[2 * PM + pl], B[0][0] * B[0][2] + B[1][0] * B[1][2]);
      lds_add(&sh_vg[3 * PM + pl], B[0][1] * B[0][1] + B[1][1] * B[1][1]);
      lds_add(&sh_vg[4 * PM + pl], B[0][1] * B[0][2] + B[1][1] * B[1][2]);
      lds_add(&sh_vg[5 * PM + pl], B[0][2] * B[0][2] + B[1][2] * B[1][2]);
      lds_add(&sh_vg[6 * PM + pl], B[0][0] * e[0] + B[1][0] * e[1]);
      lds_add(&sh_vg[7 * PM + pl], B[0][1] * e[0] + B[1][1] * e[1]);
      lds_add(&sh_vg[8 * PM + pl], B[0][2] * e[0] + B[1][2] * e[1]);
    }
    flush();
    __syncthreads();
    for (int i = threadIdx.x; i < npts; i += BLOCK) {
#pragma unroll
      for (int q = 0; q < 6; ++q) Vblk[(long)q * lay.Ppad + p0 + i] = sh_vg[q * PM + i];
      gp[p0 + i] = sh_vg[6 * PM + i];
      gp[lay.Ppad + p0 + i] = sh_vg[7 * PM + i];
      gp[2 * lay.Ppad + p0 + i] = sh_vg[8 * PM + i];
    }
  }
  __syncthreads();
  if (!UGLOB) {
    // U = P^T U' P, g = P^T g' per camera (P = blockdiag(J_l, I)): a thread per camera, in place
    for (int c = threadIdx.x; c < n_cams; c += BLOCK) {
      const double* Jl = tab + (long)c * CAMTAB_DOUBLES + 12;  // row-major
      double* Uc = sh_U + (long)c * UP::STRIDE;
      double M[3][3], T[3][3];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) M[r][q] = Uc[UP::idx(min(r, q), max(r, q))];
#pragma unroll
      for (int r = 0; r < 3; ++r)  // T = J_l^T M
#pragma unroll
        for (int q = 0; q < 3; ++q) T[r][q] = Jl[r] * M[0][q] + Jl[3 + r] * M[1][q] + Jl[6 + r] * M[2][q];
#pragma unroll
      for (int r = 0; r < 3; ++r)  // upper triangle of T J_l
#pragma unroll
        for (int q = r; q < 3; ++q) Uc[UP::idx(r, q)] = T[r][0] * Jl[q] + T[r][1] * Jl[3 + q] + T[r][2] * Jl[6 + q];
#pragma unroll
      for (int q = 3; q <= NC; ++q) {  // columns 3 .. NC - 1 of the top rows, then (q == NC) the gradient's top entries
        double* v0 = (q < NC) ? &Uc[UP::idx(0, q)] : &Uc[UP::TRI + 0];
        double* v1 = (q < NC) ? &Uc[UP::idx(1, q)] : &Uc[UP::TRI + 1];
        double* v2 = (q < NC) ? &Uc[UP::idx(2, q)] : &Uc[UP::TRI + 2];
        const double a0 = *v0, a1 = *v1, a2 = *v2;
        *v0 = Jl[0] * a0 + Jl[3] * a1 + Jl[6] * a2;
        *v1 = Jl[1] * a0 + Jl[4] * a1 + Jl[7] * a2;
        *v2 = Jl[2] * a0 + Jl[5] * a1 + Jl[8] * a2;
      }
    }
    __syncthreads();
    double* dst = partialU + (long)blockIdx.x * n_cams * UP::STRIDE;
    for (int i = threadIdx.x; i < n_cams * UP::STRIDE; i += BLOCK) dst[i] = sh_U[i];
  }
  const double tot = block_sum(cost, sh_red);
  if (threadIdx.x == 0) partial_cost[blockIdx.x] = tot;
  if (TRIAL) {
    const double st = block_sum(step_sq, sh_red);
    if (threadIdx.x == 0) trial.step_partial[blockIdx.x] = st;
  }
  if (bad) flags[0] = 1;
}

// set-up on the device (cba_create): the sorted observation coordinates and the camera-sorted copy of k_build_cs are GATHERED here from the caller's
// array and a 4-byte permutation instead of being permuted on the host and uploaded (40 bytes per observation less over PCIe, no random 16-byte
// reads on the host)
__global__ void __launch_bounds__(256)
k_gather_uv(const double* __restrict__ uv_raw, const int* __restrict__ order, long n, double* __restrict__ u, double* __restrict__ v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const double2 w = reinterpret_cast<const double2*>(uv_raw)[order[i]];
    u[i] = w.x; v[i] = w.y;
  }
}
__global__ void __launch_bounds__(256)
k_cs_fill(const int* __restrict__ perm, const int* __restrict__ sc_obs, const int* __restrict__ sc_p0, const double* __restrict__ u,
          const double* __restrict__ v, const int* __restrict__ cam, const int* __restrict__ pt, double* __restrict__ cu, double* __restrict__ cv,
          int* __restrict__ ccam, int* __restrict__ cptl) {
  const int s = blockIdx.x, o0 = sc_obs[s], o1 = sc_obs[s + 1], p0 = sc_p0[s];
  for (int j = o0 + threadIdx.x; j < o1; j += 256) {
    const int i = perm[j];
    cu[j] = u[i]; cv[j] = v[i]; ccam[j] = cam[i]; cptl[j] = pt[i] - p0;
  }
}

// unpack the reduced camera blocks: gradient -> gvec camera part
template <int NC>
__global__ void k_unpack_camera_grad(const double* __restrict__ Upacked, const int* __restrict__ cam_off,
                                     const int* __restrict__ cam_np, int n_cams, double* __restrict__ gvec) {
  using UP = UPack<NC>;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = t / NC, r = t % NC;
  if (c >= n_cams || r >= cam_np[c]) return;
  gvec[cam_off[c] + r] = Upacked[c * UP::STRIDE + UP::TRI + r];
}

// Jacobi scaling  scale_inv = sqrt(diag(J^T J))  with scipy's rules: zeros -> 1 on the first call,
// monotone max with the previous scale afterwards (common.py:598-610).
template <int NC>
__global__ void k_scale_update(const double* __restrict__ Upacked, const double* __restrict__ Vblk,
                               const int* __restrict__ param_cam, const int* __restrict__ param_loc, VecLayout lay,
                               int first, double* __restrict__ sinv, const double* __restrict__ cdiag) {
  using UP = UPack<NC>;
  const long total = lay.total();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    double v;
    if (i < lay.ncp_pad) {
      if (i >= lay.ncp) continue;  // padding keeps scale 1
      const int r = param_loc[i];
      v = Upacked[param_cam[i] * UP::STRIDE + UP::idx(r, r)];
    } else {
      const long k = (i - lay.ncp_pad) / lay.Ppad, p = (i - lay.ncp_pad) % lay.Ppad;
      if (p >= lay.P) continue;
      const int q = (k == 0) ? 0 : (k == 1 ? 3 : 5);
      v = Vblk[(long)q * lay.Ppad + p];
      if (cdiag) v += cdiag[k * lay.Ppad + p];  // constraint rows are rows of J too (x_scale = 'jac')
    }
    v = sqrt(v);
    if (first) { if (v == 0.0) v = 1.0; }
    else v = fmax(v, sinv[i]);
    sinv[i] = v;
  }
}

// Coleman-Li scaling of the bounded camera parameters (cba_set_camera_scaling): effective scale of the camera block
// sinv = state * mult, extra diagonal of the damped system in x-space = diag_h * sinv^2.
__global__ void k_cam_rescale(const double* __restrict__ state, const double* __restrict__ mult, const double* __restrict__ diag_h,
                              int ncp, double* __restrict__ sinv, double* __restrict__ cam_diag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncp) return;
  const double si = state[i] * mult[i];
  sinv[i] = si;
  cam_diag[i] = diag_h[i] * si * si;
}

// scalars of the linearisation + v1 = g / scale_inv^2 (the direction of the gradient in scaled space)
//   partial[b][0..3] = sum (g/sinv)^2, sum (x sinv)^2, sum x^2, (unused) ; partial_max[b] = max |g|
// `cam_end` = ncp_pad; `count_cams` = 0 on ranks > 0 of a sharded solve: camera entries are replicated
// on every rank and must enter the all-reduced sums once.
__global__ void __launch_bounds__(BLOCK)
k_lin_scalars(const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ sinv,
              long total, int cam_end, int count_cams, int max_from, double* __restrict__ v1, double* __restrict__ partial,
              double* __restrict__ partial_max) {
  __shared__ double sh_red[BLOCK / WAVE];
  double s0 = 0, s1 = 0, s2 = 0, m = 0;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) {
    const double gi = g[i], si = sinv[i], xi = x[i];
    const double gh = gi / si;
    v1[i] = gh / si;
    if (i >= max_from) m = fmax(m, fabs(gi));  // max_from = cam_end: point block only (bounded solves weigh the camera block on the host)
    if (i < cam_end && !count_cams) continue;
    s0 += gh * gh;
    s1 += (xi * si) * (xi * si);
    s2 += xi * xi;
  }
  double r;
  r = block_sum(s0, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 0] = r;
  r = block_sum(s1, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 1] = r;
  r = block_sum(s2, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 2] = r;
  if (threadIdx.x == 0) partial[blockIdx.x * 4 + 3] = 0.0;
  r = block_max(m, sh_red); if (threadIdx.x == 0) partial_max[blockIdx.x] = r;
}

// out = a * g / sinv^2 + b * s ; entries below n_over are taken from `over` instead (camera block given by the caller)
__global__ void k_combine(const double* __restrict__ g, const double* __restrict__ sinv, const double* __restrict__ s,
                          double a, double b, long total, const double* __restrict__ over, int n_over, double* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const double si = sinv[i];
    out[i] = (i < n_over) ? over[i] : a * g[i] / (si * si) + b * s[i];
  }
}

// ------------------------------------------------------------------------------------------------
// J.v for one or two vectors: partial[b][0..2] = sum |Jv1|^2, <Jv1,Jv2>, |Jv2|^2
template <int NC, int NV, bool CAMG = false>
__global__ void __launch_bounds__(BLOCK)
k_jv(const double* __restrict__ obs_u, const double* __restrict__ obs_v, const int* __restrict__ obs_cam,
     const int* __restrict__ obs_pt, long n_obs, const double* __restrict__ xvec, VecLayout lay,
     const double* __restrict__ tab, const int* __restrict__ cam_off, int n_cams, int loss, double f_scale,
     const double* __restrict__ v1, const double* __restrict__ v2, double* __restrict__ partial) {
  CBA_STAMP(ST_JV);
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* sh_tab = sh;
  double* sh_v = sh_tab + (CAMG ? 0 : n_cams * CAMTAB_LDS);  // NV * ncp_pad
  double* sh_red = sh_v + NV * lay.ncp_pad;
  if (!CAMG) stage_camtab(sh_tab, tab, n_cams);
  for (int i = threadIdx.x; i < lay.ncp_pad; i += BLOCK) {
    sh_v[i] = v1[i];
    if (NV == 2) sh_v[lay.ncp_pad + i] = v2[i];
  }
  __syncthreads();
  // (round 6) J v in factored form: a camera's share is G (w x Y + v_t) + A_intr v_i with w = J_l v_r — formed HERE, once per camera, in the place of v_r
  for (int c = threadIdx.x; c < n_cams; c += BLOCK) {
    const double* row = tab + (long)c * CAMTAB_DOUBLES;
    const int off = (int)row[35];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      double* vq = sh_v + q * lay.ncp_pad + off;
      const double r0 = vq[0], r1 = vq[1], r2 = vq[2];
      vq[0] = row[12] * r0 + row[13] * r1 + row[14] * r2;
      vq[1] = row[15] * r0 + row[16] * r1 + row[17] * r2;
      vq[2] = row[18] * r0 + row[19] * r1 + row[20] * r2;
    }
  }
  __syncthreads();
  const double* px = xvec + lay.ncp_pad;
  const double* p1 = v1 + lay.ncp_pad;
  const double* p2 = (NV == 2) ? v2 + lay.ncp_pad : nullptr;
  double s11 = 0, s12 = 0, s22 = 0;
  // Software pipeline (round 6).  An observation needs its record (camera, point, u, v) and then six gathers keyed by the point (the point and its
  // entries of v): two dependent memory round trips, ~2 us, in front of ~0.8 us of arithmetic — written as a plain grid-stride loop a workgroup spent its
  // 8 trips mostly parked (SQ_WAIT_ANY 59 %, lifetime 22 us at 1024 workgroups).  Here trip i issues the RECORD of trip i + 4 and the GATHERS of trip
  // i + 2 (whose record was issued two trips ago) and only then works on trip i: every load has TWO trips to land (with one, 33 us; the kernel's 174
  // registers leave two workgroups per CU, so the depth has to come from the pipeline).  One explicit wait per trip, at its top: vmcnt(LOADS) lets the
  // loads of the previous trip stay in flight (vector-memory operations return in order, and the loop issues nothing else); the values that become due
  // pass through an empty asm so that the compiler, which cannot see that wait, does not put a vmcnt(0) of its own at their first use — behind the
  // loads the trip has just issued (the rule found at schur_reg3_body).
  struct Rec { int cam, pt; double u, v; };
  struct Gat { double X, Y, Z, a, b, c, d, e, f; };
  constexpr int LOADS = 4 + (NV == 2 ? 9 : 6);  // vector-memory instructions a trip issues
  const long stride = (long)gridDim.x * BLOCK, last = n_obs - 1;
  auto load_rec = [&](long i) { Rec r; const long k = min(i, last); r.cam = obs_cam[k]; r.pt = obs_pt[k]; r.u = obs_u[k]; r.v = obs_v[k]; return r; };
  auto load_gat = [&](int pt) {
    Gat g;
    g.X = px[pt]; g.Y = px[lay.Ppad + pt]; g.Z = px[2 * lay.Ppad + pt];
    g.a = p1[pt]; g.b = p1[lay.Ppad + pt]; g.c = p1[2 * lay.Ppad + pt];
    if (NV == 2) { g.d = p2[pt]; g.e = p2[lay.Ppad + pt]; g.f = p2[2 * lay.Ppad + pt]; } else { g.d = g.e = g.f = 0.0; }
    return g;
  };
  auto launder_rec = [](Rec& r) { asm volatile("" : "+v"(r.cam), "+v"(r.pt), "+v"(r.u), "+v"(r.v)); };
  auto launder_gat = [](Gat& g) {
    asm volatile("" : "+v"(g.X), "+v"(g.Y), "+v"(g.Z), "+v"(g.a), "+v"(g.b), "+v"(g.c));
    if (NV == 2) asm volatile("" : "+v"(g.d), "+v"(g.e), "+v"(g.f));
  };
  long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  Rec rc = load_rec(i), r1 = load_rec(i + stride), r2 = load_rec(i + 2 * stride), r3 = load_rec(i + 3 * stride);
  Gat gc = load_gat(rc.pt), g1 = load_gat(r1.pt);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  launder_rec(rc); launder_rec(r1); launder_rec(r2); launder_rec(r3); launder_gat(gc); launder_gat(g1);
  for (; i < n_obs; i += stride) {
    // everything but the previous trip's loads has landed: the gathers of this trip (issued two trips ago) and the record of trip i + 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    launder_gat(gc); launder_rec(r2);
    const Gat g2 = load_gat(r2.pt);            // gathers of trip i + 2
    const Rec r4 = load_rec(i + 4 * stride);   // record of trip i + 4
    __builtin_amdgcn_sched_barrier(0);
    const int cam = rc.cam;
    double e[2], G[2][3], Yr[3], Aint[2][3], B[2][3];
    const CamTab& ctj = cam_of<CAMG>(sh_tab, tab, cam);
    obs_factors(ctj, gc.X, gc.Y, gc.Z, rc.u, rc.v, loss, f_scale, e, G, Yr, Aint, B);
    const double* vc = sh_v + (int)ctj.pad[0];  // (= cam_off[cam], from the table row that is being read anyway): w (3), v_t (3), v_i (3, nine-parameter cameras)
    auto jv = [&](const double* q, double px0, double px1, double px2, double* o0, double* o1) {
      // m = w x Y + v_t;  row r: G_r . m + B_r . v_p (+ A_intr,r . v_i)
      const double m0 = fma(q[1], Yr[2], fma(-q[2], Yr[1], q[3]));
      const double m1 = fma(q[2], Yr[0], fma(-q[0], Yr[2], q[4]));
      const double m2 = fma(q[0], Yr[1], fma(-q[1], Yr[0], q[5]));
      double a0 = B[0][0] * px0 + B[0][1] * px1 + B[0][2] * px2, a1 = B[1][0] * px0 + B[1][1] * px1 + B[1][2] * px2;
      a0 = fma(G[0][2], m2, fma(G[0][1], m1, fma(G[0][0], m0, a0)));
      a1 = fma(G[1][2], m2, fma(G[1][1], m1, fma(G[1][0], m0, a1)));
      if (NC == 9 && ctj.nparams == 9.0) {
        a0 = fma(Aint[0][2], q[8], fma(Aint[0][1], q[7], fma(Aint[0][0], q[6], a0)));
        a1 = fma(Aint[1][2], q[8], fma(Aint[1][1], q[7], fma(Aint[1][0], q[6], a1)));
      }
      *o0 = a0; *o1 = a1;
    };
    double a0, a1;
    jv(vc, gc.a, gc.b, gc.c, &a0, &a1);
    s11 += a0 * a0 + a1 * a1;
    if (NV == 2) {
      double b0, b1;
      jv(vc + lay.ncp_pad, gc.d, gc.e, gc.f, &b0, &b1);
      s12 += a0 * b0 + a1 * b1;
      s22 += b0 * b0 + b1 * b1;
    }
    rc = r1; r1 = r2; r2 = r3; r3 = r4; gc = g1; g1 = g2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the loads of the trips that do not exist)
  double r;
  r = block_sum(s11, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 0] = r;
  r = block_sum(s12, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 1] = r;
  r = block_sum(s22, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 2] = r;
  if (threadIdx.x == 0) partial[blockIdx.x * 4 + 3] = 0.0;
}

// ------------------------------------------------------------------------------------------------
// Schur pass.  Per observation i of point p (V'_p = V_p + lam D_p^2 = L L^T):
//     Z_i = B_i L^-T (2x3),   y_p = L^-1 g_p,   b_{c_i} += A_i^T (Z_i y_p)
// and per pair (i <= j) of observations of the same point
//     Sacc[c_i, c_j] += A_i^T (Z_i Z_j^T) A_j          ( = W_i V'^-1 W_j^T )
// The reduced system is S = U + lam D_c^2 - Sacc, rhs = -g_c + b   (SURVEY.md Appendix A.4).
//
// Scatter-reducing Sacc is the expensive part (k(k+1)/2 blocks of nc^2 per point).  FP64 global atomics manage ~2e10 updates/s on this chip
// and an LDS-atomic tile kernel 1.0 ms per pass on cfg4 (rounds 1-3 kept it as a fallback; deleted in round 4), so the sums are formed in
// REGISTERS: the cameras are cut into G groups of g <= 16, the block upper triangle of Sacc into G(G+1)/2 tiles (a <= b), a workgroup is
// bound to one tile and every thread owns one camera-pair block of it (k_schur_reg3 below; plan: schur_plan.h).
struct TilePlan {
  const int* chunk_start;        // [n_tile_chunks + 1] offsets into `obs`
  const int* wg_first;           // [grid] first chunk of the workgroup
  const int* wg_end;             // [grid] end of the chunk range the workgroup strides through
  const int* wg_tile;            // [grid] tile of each workgroup
  const int* wg_stride;          // [grid] chunk stride (workgroups sharing the range)
  const int* tile_a;             // [n_tiles] group ids
  const int* tile_b;
  const int* group_cam_begin;    // [G + 1]
  const int* group_par_begin;    // [G + 1]
  int g;                         // cameras per group (max)
  int tile_elems;                // width of one workgroup's partial row: 256 blocks of nc^2
  const int* obs;                // chunk slot -> observation (index into the T records)
  int rep;                       // threads per camera-pair block (256 / g^2 when the group is small), each takes every rep-th pair
  const unsigned* codes;         // per chunk and wave: nit iterations x 64 lanes of (i_addr | j_addr << 16), LDS addresses in 16-byte pieces
  const int* code_start;         // [n_tile_chunks + 1] offsets into `codes`
  const unsigned* nit;           // [n_tile_chunks] iterations of waves 0..3, one byte each
  int prio_shift;                // >= 0: the workgroups of the two halves of the dispatch order raise their wave priority in alternate groups of 2^prio_shift trips
};

// ------------------------------------------------------------------------------------------------
// Schur pass: k_tprep + k_schur_reg3 (+ k_reg_reduce, k_reg_fold).
//
// With Z_i = B_i L^-T (L L^T = V + lam D^2 of the point) the pair term A_i^T (Z_i Z_j^T) A_j is T_i T_j^T for the
// per-observation NC x 3 matrix T_i = A_i^T Z_i, and the rhs term is T_i y with y = L^-1 g_point.
//
// k_tprep evaluates every observation ONCE, stores a record per observation in HBM and reduces the rhs per camera (LDS atomics,
// per-workgroup partials, k_reduce_rows).
//
// The record is COMPACT.  The camera block factors (ba_math.h, project_full): A = G [C | I | A_intr] with G = d(pixel)/dX_c
// (2 x 3), C = -[Y]x J_l the derivative of the rotated point Y = R X by the rotation vector, so
//     T = [ J_l^T [Y]x Q ;  Q ;  T_intr ],     Q = G^T Z  (3 x 3: rows 3..5 of T),   T_intr = A_intr^T Z  (rows 6..8, NC = 9).
// The record holds Y (3), Q (9) and T_intr (9): 12 doubles = 96 B (NC = 6), or 21 -> 22 doubles = 176 B (NC = 9) instead of 144 / 240 B
// (in LDS: 112 / 176 B apart, an odd stride in 16-byte pieces); the pair kernel gathers, stages and reads a third less, and it never forms the top rows: with D = Q_i Q_j^T
//     [Y_i]x D [Y_j]x^T | [Y_i]x D | D [Y_j]x^T | D
// are the four 3 x 3 quarters of the PRIMED block T'_i T'_j^T, T' = [[Y]x Q ; Q ; T_intr] (99 FP64 operations per pair
// instead of 108 on 18 + 18 doubles).  The per-camera factor J_l^T is applied once per block in the pair kernel's epilogue.
// The rhs needs the true rows 0..2, which k_tprep has in registers anyway.
template <int NC> struct SchurRec {
  static constexpr int NVAL = (NC == 9) ? 21 : 12;        // doubles that carry data
  static constexpr int NPH = (NVAL + 1) / 2;              // 16-byte pieces that carry data: 6 / 11
  static constexpr int LST = (NPH & 1) ? NPH : NPH + 1;   // record stride in LDS, 16-byte pieces, odd: 7 / 11 (bank spread)
  static constexpr int REC = 2 * LST;                     // record stride in LDS, doubles: 14 / 22
  // record stride in HBM, doubles: the pieces that carry data, 12 / 22 (96 / 176 bytes).  NC = 6: the seventh piece of the LDS stride is padding
  // for the bank spread only; the pair kernel's load lanes that land on it fetch the sixth piece again.  (128-byte records, one line each, were
  // measured: the pair kernel's issue phase shrank by 7 %, k_tprep grew by as much, and the pass moved 1.15 GB instead of 0.93 GB through HBM.)
  static constexpr int HREC = 2 * NPH;
  static constexpr int STAGE = (HREC / 2) | 1;            // k_tprep's transposing LDS stage: record stride in pieces, odd (bank spread)
  // waves of a k_tprep workgroup that stage at the same time.  NC = 9: two, in two turns — with a stage for all four (45 KB) next to the camera
  // table the kernel had 92 KB of LDS and ran one workgroup per CU
  static constexpr int STAGE_WAVES = (NC == 9) ? 2 : 4;
};
static_assert(SchurRec<6>::REC == 14 && SchurRec<6>::LST == 7 && SchurRec<9>::REC == 22 && SchurRec<9>::LST == 11, "record sizes");

// true T (NC x 3, row-major) from a compact record and the camera's J_l (row-major): k_heavy_schur, k_con_schur
template <int NC>
__device__ __forceinline__ void expand_record(const double* __restrict__ rec, const double* __restrict__ Jl, double* __restrict__ T) {
  const double Y0 = rec[0], Y1 = rec[1], Y2 = rec[2];
  const double* Q = rec + 3;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const double q0 = Q[m], q1 = Q[3 + m], q2 = Q[6 + m];
    const double c0 = Y1 * q2 - Y2 * q1, c1 = Y2 * q0 - Y0 * q2, c2 = Y0 * q1 - Y1 * q0;  // Y x Q[:, m]
#pragma unroll
    for (int r = 0; r < 3; ++r) T[3 * r + m] = Jl[r] * c0 + Jl[3 + r] * c1 + Jl[6 + r] * c2;  // (J_l^T c)_r
    T[9 + m] = q0; T[12 + m] = q1; T[15 + m] = q2;
  }
  if (NC == 9) {
#pragma unroll
    for (int k = 0; k < 9; ++k) T[18 + k] = rec[12 + k];
  }
}

// Single-rank fused iteration (round 5): the three reductions of the linearisation and the damping (k_lin_finish: one workgroup, 6 us + a launch
// gap at the head of every iteration) are done by k_tprep's workgroups themselves — each sums the ~2 x 1024 partial rows (the same additions in the
// same order everywhere: every workgroup gets the same bits), takes the damping from the radius the host has just chosen, and workgroup 0 leaves
// the scalars where k_lin_finish left them (scal[0..4], [12..15], [40], [41], fz[0], fz[1]) for the kernels behind it and for k_publish.
struct LinFin {
  const double* partial_lin;  // [rows_lin][4]   k_scale_lin
  const double* partial_max;  // [rows_lin]
  const double* partial_jv;   // [rows_jv][4]    k_jv
  int rows_lin, rows_jv;
  double radius;
  double* scal;
  double* fz;
};
// returns the damping; every thread of the workgroup calls it (two barriers inside)
__device__ __forceinline__ double lin_finish_block(const LinFin& lf, double (*sh_red)[BLOCK / WAVE], double* sh_out) {
  double v[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) v[q] = 0.0;
  for (int b = threadIdx.x; b < lf.rows_lin; b += BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += lf.partial_lin[(long)b * 4 + j];
    v[4] = fmax(v[4], lf.partial_max[b]);
  }
  for (int b = threadIdx.x; b < lf.rows_jv; b += BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[5 + j] += lf.partial_jv[(long)b * 4 + j];
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double r = (q == 4) ? wave_max(v[q]) : wave_sum(v[q]);
    if (lane == 0) sh_red[q][w] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      double r = 0.0;
      for (int i = 0; i < BLOCK / WAVE; ++i) r = (q == 4) ? fmax(r, sh_red[q][i]) : r + sh_red[q][i];
      tot[q] = r;
    }
    const double gh_sq = tot[0], jg_sq = tot[5] + tot[3], xs = sqrt(tot[1]);  // (+ C_gg, the Coleman-Li term of a bounded solve: zero otherwise)
    const double radius = lf.radius > 0.0 ? lf.radius : (xs > 0.0 ? xs : 1.0);  // (fused_lam's rule)
    const double lam = trf::damping(jg_sq, gh_sq, radius);
    sh_out[0] = lam;
    if (blockIdx.x == 0) {
#pragma unroll
      for (int q = 0; q < 9; ++q) lf.scal[q < 5 ? q : 12 + (q - 5)] = tot[q];
      lf.fz[0] = lam; lf.fz[1] = radius;
      lf.scal[40] = lam; lf.scal[41] = radius;
    }
  }
  __syncthreads();
  return sh_out[0];
}

template <int NC, int DETM = 0, bool CAMG = false, bool LINF = false>
__global__ void __launch_bounds__(BLOCK, DETM == 0 ? 2 : 1)  // (two workgroups per CU = two waves per SIMD: 256 registers; the fixed-order variants take more and run one)
k_tprep(const double* __restrict__ obs_u, const double* __restrict__ obs_v, const int* __restrict__ obs_cam,
        const int* __restrict__ obs_pt, const int* __restrict__ chunk_start, int n_chunks,
        const double* __restrict__ xvec, VecLayout lay, const double* __restrict__ tab, const int* __restrict__ cam_off,
        int n_cams, int loss, double f_scale, double lam, const double* __restrict__ lam_dev, const double* __restrict__ Vblk,
        const double* __restrict__ gvec, const double* __restrict__ sinv, double* __restrict__ Trec,
        double* __restrict__ partial_b, int* __restrict__ flags, DetPlan det = DetPlan{nullptr, nullptr}, LinFin lf = LinFin{}) {
  CBA_STAMP(ST_TPREP);
  constexpr int REC = SchurRec<NC>::HREC, NP = REC / 2, SP = SchurRec<NC>::STAGE, SW = SchurRec<NC>::STAGE_WAVES;  // the records as they lie in HBM
  static_assert((BLOCK / WAVE) % SW == 0, "stage turns");
  constexpr bool DET = DETM > 0;
  static_assert(!(DET && CAMG), "the fixed-order sums keep the camera table in LDS");
  if (LINF) {
    __shared__ double sh_lf_red[10][BLOCK / WAVE];
    __shared__ double sh_lf_out[2];
    lam = lin_finish_block(lf, sh_lf_red, sh_lf_out);
  } else if (lam_dev) lam = *lam_dev;  // fused step: the damping was computed on the device (k_fused_lam / k_lin_finish)
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double2* sh_stage = reinterpret_cast<double2*>(sh);        // [SW][WAVE * SP]  record transpose, per wave (of a turn)
  double* sh_tab = sh + (size_t)SW * WAVE * SP * 2;
  double* sh_b = sh_tab + (CAMG ? 0 : n_cams * CAMTAB_LDS);  // ncp_pad (DET: the parking area of det_round, then the chunk's camera order)
  int* sh_perm = reinterpret_cast<int*>(sh_b + DET_ROUND * DET_LD);
  int* sh_cs = sh_perm + CHUNK;
  constexpr int DM = DET ? DETM : 1;
  double dacc[DM];
#pragma unroll
  for (int m = 0; m < DM; ++m) dacc[m] = 0.0;
  stage_camtab(sh_tab, tab, n_cams);
  if (!DET)
    for (int i = threadIdx.x; i < lay.ncp_pad; i += BLOCK) sh_b[i] = 0.0;
  __syncthreads();
  const double* px = xvec + lay.ncp_pad;
  const double* gp = gvec + lay.ncp_pad;
  const double* dp = sinv + lay.ncp_pad;
  const int wv = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
  double2* stage = sh_stage + (wv % SW) * WAVE * SP;
  bool fail = false;
  const int last_obs = max(chunk_start[n_chunks] - 1, 0);
  // Two-stage software pipeline (round 6).  An observation's operands are its record (u, v, camera, point) and fifteen gathers keyed by the point (the
  // point, its V block, scale and gradient): two dependent round trips in front of the arithmetic.  Through round 5 only the record of the next chunk was
  // in flight while a chunk was processed; the gathers were issued and waited for at the head of every chunk (SQ_WAIT_ANY 54 %, 16 chunks of ~5 us per
  // workgroup).  Now chunk n issues the RECORD of chunk n + 2 and the GATHERS of chunk n + 1 before it computes, and waits for them only in front of its
  // own record stores (what the wait covers was issued before the previous chunk's stores, which are therefore never waited for).
  // Not with fixed-order sums (PIPE: those variants read their per-chunk tables in between).  Nine-parameter cameras send only the POINT ahead (!FULL), and
  // of the record two chunks ahead only its point index: the body has no 30 registers to spare at two workgroups per CU (with all fifteen values in
  // flight it asked for 280 and spilled when capped).  V, scale and gradient are then issued at the head of the chunk, in FRONT of the prefetches — the
  // wait for them leaves those in flight — and are not needed before the ~250 instructions of the linearisation.
  constexpr bool PIPE = DETM == 0, FULL = true;  // (the factored linearisation freed the registers: all fifteen values travel ahead for nine-parameter cameras too; the light form stays for reference)
  struct PtOps { double X, Y, Z, V[6], d[3], g[3]; };
  auto load_rest = [&](PtOps& q, int pt) {
#pragma unroll
    for (int k = 0; k < 6; ++k) q.V[k] = Vblk[(long)k * lay.Ppad + pt];
#pragma unroll
    for (int k = 0; k < 3; ++k) { q.d[k] = dp[(long)k * lay.Ppad + pt]; q.g[k] = gp[(long)k * lay.Ppad + pt]; }
  };
  auto load_xyz = [&](PtOps& q, int pt) { q.X = px[pt]; q.Y = px[lay.Ppad + pt]; q.Z = px[2 * lay.Ppad + pt]; };
  auto load_ops = [&](int pt) {  // what travels a chunk ahead
    PtOps q;
    load_xyz(q, pt);
    if constexpr (FULL) load_rest(q, pt);
    return q;
  };
  auto launder_obs = [](ObsRec& r) { asm volatile("" : "+v"(r.u), "+v"(r.v), "+v"(r.cam), "+v"(r.pt)); };
  auto launder_ops = [](PtOps& q) {
    asm volatile("" : "+v"(q.X), "+v"(q.Y), "+v"(q.Z));
    if constexpr (FULL) {
#pragma unroll
      for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(q.V[k]));
#pragma unroll
      for (int k = 0; k < 3; ++k) asm volatile("" : "+v"(q.d[k]), "+v"(q.g[k]));
    }
  };
  const int stride = (int)gridDim.x, last_chunk = n_chunks - 1;
  int ch = blockIdx.x;
  int o0 = 0, o1 = 0, no0 = 0, no1 = 0;
  ObsRec cur = {0.0, 0.0, 0, 0}, nx = cur;
  PtOps ops;
  if (ch < n_chunks) {
    o0 = chunk_start[ch]; o1 = chunk_start[ch + 1];
    const int c1 = min(ch + stride, last_chunk);
    no0 = chunk_start[c1]; no1 = chunk_start[c1 + 1];
    cur = load_obs(obs_u, obs_v, obs_cam, obs_pt, min(o0 + (int)threadIdx.x, last_obs));
    if constexpr (PIPE) {
      if constexpr (FULL) nx = load_obs(obs_u, obs_v, obs_cam, obs_pt, min(no0 + (int)threadIdx.x, last_obs));
      else nx.pt = obs_pt[min(no0 + (int)threadIdx.x, last_obs)];
      ops = load_ops(cur.pt);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      launder_obs(cur); launder_obs(nx); launder_ops(ops);
    }
  }
  while (ch < n_chunks) {
    const int nxt = ch + stride, c2 = min(ch + 2 * stride, last_chunk);
    const int nno0 = chunk_start[c2], nno1 = chunk_start[c2 + 1];
    PtOps ops_n;
    if constexpr (!PIPE) { load_xyz(ops, cur.pt); load_rest(ops, cur.pt); }  // this chunk's own, waited for where they are used
    else {
      if constexpr (!FULL) load_rest(ops, cur.pt);  // (in front of the prefetches: the wait for them leaves those in flight)
      ops_n = load_ops(nx.pt);                      // gathers of the next chunk
    }
    ObsRec nn = {0.0, 0.0, 0, 0};
    if constexpr (PIPE && !FULL) {  // the rest of the next chunk's record, and the point index of the chunk after next
      const int k1 = min(no0 + (int)threadIdx.x, last_obs);
      nx.u = obs_u[k1]; nx.v = obs_v[k1]; nx.cam = obs_cam[k1];
      nn.pt = obs_pt[min(nno0 + (int)threadIdx.x, last_obs)];
    } else {
      nn = load_obs(obs_u, obs_v, obs_cam, obs_pt, min((PIPE ? nno0 : no0) + (int)threadIdx.x, last_obs));  // record of the chunk after next (PIPE) / of the next
    }
    if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
    const int i = o0 + threadIdx.x;
    double rec[REC];
#pragma unroll
    for (int k = 0; k < REC; ++k) rec[k] = 0.0;
    double bval[DET_ROUND];
#pragma unroll
    for (int q = 0; q < DET_ROUND; ++q) bval[q] = 0.0;
    if (DET) {
      sh_perm[threadIdx.x] = det.perm[(long)ch * CHUNK + threadIdx.x];
      for (int q = threadIdx.x; q <= n_cams; q += BLOCK) sh_cs[q] = det.cstart[(long)ch * (n_cams + 1) + q];
    }
    if (i < o1) {
      const int cam = cur.cam;
      const CamTab& ct = cam_of<CAMG>(sh_tab, tab, cam);
      const double X = ops.X, Yw = ops.Y, Zw = ops.Z;
      // factored linearisation (round 6): T = A^T Z = [J_l^T [Y]x Q ; Q ; T_intr] with Q = G^T Z — the record never held the top rows, and their share
      // of the right-hand side, J_l^T (Y x (Q y)), is accumulated WITHOUT its J_l^T (applied once per camera when the workgroup writes its row out)
      double e[2], G[2][3], Yr[3], Aint[2][3], B[2][3], Z[2][3];
      obs_factors(ct, X, Yw, Zw, cur.u, cur.v, loss, f_scale, e, G, Yr, Aint, B);
      const int np = (int)ct.nparams;
      double Vd[6], L[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) Vd[q] = ops.V[q];
      const double d0 = ops.d[0], d1 = ops.d[1], d2 = ops.d[2];
      Vd[0] += lam * d0 * d0; Vd[3] += lam * d1 * d1; Vd[5] += lam * d2 * d2;
      if (!chol3(Vd, L)) {
        fail = true;
        L[0] = L[2] = L[5] = 1.0; L[1] = L[3] = L[4] = 0.0;
      }
      chol3_fwd(L, B[0], Z[0]);
      chol3_fwd(L, B[1], Z[1]);
      const double gpt[3] = {ops.g[0], ops.g[1], ops.g[2]};
      double y[3];
      chol3_fwd(L, gpt, y);
      double* bc = sh_b + (int)ct.pad[0];  // (= cam_off[cam])
      // rotated point Y = R X (the record's first three entries)
      rec[0] = Yr[0]; rec[1] = Yr[1]; rec[2] = Yr[2];
      double qy[3];  // Q y
#pragma unroll
      for (int r = 3; r < NC; ++r) {  // rows 3..5: Q = G^T Z; rows 6..8: T_intr = A_intr^T Z
        const bool live = r < np;
        const double a0 = (r < 6) ? G[0][r - 3] : Aint[0][(r - 6) % 3], a1 = (r < 6) ? G[1][r - 3] : Aint[1][(r - 6) % 3];
        double t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = live ? a0 * Z[0][k] + a1 * Z[1][k] : 0.0;
        const double ty = t[0] * y[0] + t[1] * y[1] + t[2] * y[2];
        if (r < 6) qy[r - 3] = ty;
        if (DET) bval[r] = live ? ty : 0.0;
        else if (live) lds_add(&bc[r], ty);
        rec[3 * r - 6] = t[0]; rec[3 * r - 5] = t[1]; rec[3 * r - 4] = t[2];
      }
      {  // rows 0..2 of the right-hand side: J_l^T c, c = Y x (Q y)
        const double c0 = Yr[1] * qy[2] - Yr[2] * qy[1], c1 = Yr[2] * qy[0] - Yr[0] * qy[2], c2 = Yr[0] * qy[1] - Yr[1] * qy[0];
        if (DET) {  // (the fixed-order sums are formed per (camera, row) task: they take the true rows)
#pragma unroll
          for (int r = 0; r < 3; ++r) bval[r] = ct.Jl[r] * c0 + ct.Jl[3 + r] * c1 + ct.Jl[6 + r] * c2;
        } else {
          lds_add(&bc[0], c0); lds_add(&bc[1], c1); lds_add(&bc[2], c2);
        }
      }
    }
    // The 64 records of a wave are one contiguous run of Trec.  A lane storing its own record issues 16-byte stores one
    // record apart (64 cache lines per instruction: the store path stalled, 45 % issue-stall cycles); so the wave transposes
    // through LDS and every store instruction writes 1 KB contiguous.
    // what this chunk issued ahead has had the chunk's arithmetic to land; behind this wait come the record stores, which nothing waits for
    if constexpr (PIPE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (FULL) launder_obs(nn); else { launder_obs(nx); asm volatile("" : "+v"(nn.pt)); }
      launder_ops(ops_n);
    }
    const int w0 = o0 + wv * WAVE;                       // first observation of this wave
    const int n_pieces = max(0, min(WAVE, o1 - w0)) * NP;  // live pieces of this wave
    double2* dst = reinterpret_cast<double2*>(Trec + (long)w0 * REC);
#pragma unroll
    for (int turn = 0; turn < (BLOCK / WAVE) / SW; ++turn) {
      if (wv / SW == turn) {
#pragma unroll
        for (int k = 0; k < NP; ++k) stage[lane * SP + k] = make_double2(rec[2 * k], rec[2 * k + 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int e = k * WAVE + lane;
          const double2 v = stage[(e / NP) * SP + e % NP];
          if (e < n_pieces) dst[e] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      if (SW < BLOCK / WAVE) __syncthreads();  // the next turn's waves reuse the stage
    }
    if (DET) {  // rhs terms: fixed-order sums per camera (one round: NC <= DET_ROUND values per observation)
      __syncthreads();
      det_round<DM>(bval, sh_b, sh_perm, sh_cs, n_cams, dacc);
    }
    if constexpr (PIPE && FULL) { cur = nx; nx = nn; ops = ops_n; }
    else if constexpr (PIPE) { cur = nx; nx.pt = nn.pt; ops.X = ops_n.X; ops.Y = ops_n.Y; ops.Z = ops_n.Z; }
    else { cur = nn; }
    o0 = no0; o1 = no1; no0 = nno0; no1 = nno1; ch = nxt;
  }
  if (fail) flags[1] = 1;
  __syncthreads();
  double* brow = partial_b + (long)blockIdx.x * lay.ncp_pad;
  if (DET) {
    for (int i = lay.ncp + threadIdx.x; i < lay.ncp_pad; i += BLOCK) brow[i] = 0.0;
#pragma unroll
    for (int m = 0; m < DM; ++m) {
      const int task = (int)threadIdx.x + BLOCK * m;
      const int c = task / DET_ROUND, q = task % DET_ROUND;
      if (task < n_cams * DET_ROUND && q < (int)cam_at(sh_tab, c).nparams) brow[cam_off[c] + q] = dacc[m];
    }
  } else {
    // the rotation entries were summed without their camera's J_l^T: applied here, once per camera and workgroup
    for (int c = threadIdx.x; c < n_cams; c += BLOCK) {
      const double* row = tab + (long)c * CAMTAB_DOUBLES;
      double* b3 = sh_b + (int)row[35];
      const double c0 = b3[0], c1 = b3[1], c2 = b3[2];
#pragma unroll
      for (int r = 0; r < 3; ++r) b3[r] = row[12 + r] * c0 + row[15 + r] * c1 + row[18 + r] * c2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < lay.ncp_pad; i += BLOCK) brow[i] = sh_b[i];
  }
}

// Round 6: the three threads of a nine-parameter block split it by SUB-BLOCK instead of by rows, so that no product is formed twice.  With
// T' = [[Y]x Q ; Q ; I] (I = T_intr) and D = Q_i Q_j^T, E = Q_i I_j^T, F = I_i Q_j^T, H = I_i I_j^T the primed block is
//     [ [Y_i]x D [Y_j]x^T   [Y_i]x D   [Y_i]x E ]        role 0: the 6 x 6 corner — pair6 on the (Y, Q) parts, 99 FP64 instructions
//     [       D [Y_j]x^T         D          E    ]        role 1: columns 6..8 — E, [Y_i]x E, H: 81
//     [       F [Y_j]x^T         F          H    ]        role 2: rows 6..8 of columns 0..5 — F, F [Y_j]x^T: 54
// 234 instructions per pair where the split by rows took 324 (each row third formed D or F, its product with [Y_j]x and its E or H), and 39
// instead of 50 16-byte LDS reads.  Accumulators: 36 / 27 / 18 doubles.
// role 0: pair6's arithmetic column by column (D = Q_i Q_j^T, then three values per column), so that 18 instead of 36 doubles sit beside the accumulators:
// the nine-parameter kernel has 168 registers per thread
__device__ __forceinline__ void pair9_corner(double (&acc)[6][6], const double2* __restrict__ Ri, const double2* __restrict__ Rj) {
  double D[3][3], Yi[3], Yj[3];
  {
    const double2 i0 = Ri[0], i1 = Ri[1], i2 = Ri[2], i3 = Ri[3], i4 = Ri[4], i5 = Ri[5];
    const double2 j0 = Rj[0], j1 = Rj[1], j2 = Rj[2], j3 = Rj[3], j4 = Rj[4], j5 = Rj[5];
    Yi[0] = i0.x; Yi[1] = i0.y; Yi[2] = i1.x; Yj[0] = j0.x; Yj[1] = j0.y; Yj[2] = j1.x;
    const double Qi[9] = {i1.y, i2.x, i2.y, i3.x, i3.y, i4.x, i4.y, i5.x, i5.y};
    const double Qj[9] = {j1.y, j2.x, j2.y, j3.x, j3.y, j4.x, j4.y, j5.x, j5.y};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) D[a][b] = fma(Qi[3 * a + 2], Qj[3 * b + 2], fma(Qi[3 * a + 1], Qj[3 * b + 1], Qi[3 * a] * Qj[3 * b]));
  }
  auto put = [&](int c, double m0, double m1, double m2) {
    acc[0][c] = fma(Yi[1], m2, fma(-Yi[2], m1, acc[0][c]));
    acc[1][c] = fma(Yi[2], m0, fma(-Yi[0], m2, acc[1][c]));
    acc[2][c] = fma(Yi[0], m1, fma(-Yi[1], m0, acc[2][c]));
    acc[3][c] += m0; acc[4][c] += m1; acc[5][c] += m2;
  };
#pragma unroll
  for (int b = 0; b < 3; ++b) put(3 + b, D[0][b], D[1][b], D[2][b]);
  put(0, fma(Yj[1], D[0][2], -(Yj[2] * D[0][1])), fma(Yj[1], D[1][2], -(Yj[2] * D[1][1])), fma(Yj[1], D[2][2], -(Yj[2] * D[2][1])));
  put(1, fma(Yj[2], D[0][0], -(Yj[0] * D[0][2])), fma(Yj[2], D[1][0], -(Yj[0] * D[1][2])), fma(Yj[2], D[2][0], -(Yj[0] * D[2][2])));
  put(2, fma(Yj[0], D[0][1], -(Yj[1] * D[0][0])), fma(Yj[0], D[1][1], -(Yj[1] * D[1][0])), fma(Yj[0], D[2][1], -(Yj[1] * D[2][0])));
}
__device__ __forceinline__ void pair9_cols(double* __restrict__ acc /* [9][3] */, const double2* __restrict__ Ri, const double2* __restrict__ Rj) {
  const double2 i0 = Ri[0], i1 = Ri[1], i2 = Ri[2], i3 = Ri[3], i4 = Ri[4], i5 = Ri[5], i6 = Ri[6], i7 = Ri[7], i8 = Ri[8], i9 = Ri[9], i10 = Ri[10];
  const double2 j6 = Rj[6], j7 = Rj[7], j8 = Rj[8], j9 = Rj[9], j10 = Rj[10];
  const double Yi[3] = {i0.x, i0.y, i1.x};
  const double Qi[9] = {i1.y, i2.x, i2.y, i3.x, i3.y, i4.x, i4.y, i5.x, i5.y};
  const double Ii[9] = {i6.x, i6.y, i7.x, i7.y, i8.x, i8.y, i9.x, i9.y, i10.x};
  const double Ij[9] = {j6.x, j6.y, j7.x, j7.y, j8.x, j8.y, j9.x, j9.y, j10.x};
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    const double e0 = fma(Qi[2], Ij[3 * b + 2], fma(Qi[1], Ij[3 * b + 1], Qi[0] * Ij[3 * b]));
    const double e1 = fma(Qi[5], Ij[3 * b + 2], fma(Qi[4], Ij[3 * b + 1], Qi[3] * Ij[3 * b]));
    const double e2 = fma(Qi[8], Ij[3 * b + 2], fma(Qi[7], Ij[3 * b + 1], Qi[6] * Ij[3 * b]));
    acc[0 * 3 + b] = fma(Yi[1], e2, fma(-Yi[2], e1, acc[0 * 3 + b]));
    acc[1 * 3 + b] = fma(Yi[2], e0, fma(-Yi[0], e2, acc[1 * 3 + b]));
    acc[2 * 3 + b] = fma(Yi[0], e1, fma(-Yi[1], e0, acc[2 * 3 + b]));
    acc[3 * 3 + b] += e0; acc[4 * 3 + b] += e1; acc[5 * 3 + b] += e2;
#pragma unroll
    for (int a = 0; a < 3; ++a)
      acc[(6 + a) * 3 + b] = fma(Ii[3 * a + 2], Ij[3 * b + 2], fma(Ii[3 * a + 1], Ij[3 * b + 1], fma(Ii[3 * a], Ij[3 * b], acc[(6 + a) * 3 + b])));
  }
}
__device__ __forceinline__ void pair9_rows(double* __restrict__ acc /* [3][6] */, const double2* __restrict__ Ri, const double2* __restrict__ Rj) {
  const double2 i6 = Ri[6], i7 = Ri[7], i8 = Ri[8], i9 = Ri[9], i10 = Ri[10];
  const double2 j0 = Rj[0], j1 = Rj[1], j2 = Rj[2], j3 = Rj[3], j4 = Rj[4], j5 = Rj[5];
  const double Ii[9] = {i6.x, i6.y, i7.x, i7.y, i8.x, i8.y, i9.x, i9.y, i10.x};
  const double Yj[3] = {j0.x, j0.y, j1.x};
  const double Qj[9] = {j1.y, j2.x, j2.y, j3.x, j3.y, j4.x, j4.y, j5.x, j5.y};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double f0 = fma(Ii[3 * a + 2], Qj[2], fma(Ii[3 * a + 1], Qj[1], Ii[3 * a] * Qj[0]));
    const double f1 = fma(Ii[3 * a + 2], Qj[5], fma(Ii[3 * a + 1], Qj[4], Ii[3 * a] * Qj[3]));
    const double f2 = fma(Ii[3 * a + 2], Qj[8], fma(Ii[3 * a + 1], Qj[7], Ii[3 * a] * Qj[6]));
    acc[a * 6 + 3] += f0; acc[a * 6 + 4] += f1; acc[a * 6 + 5] += f2;
    acc[a * 6 + 0] = fma(Yj[1], f2, fma(-Yj[2], f1, acc[a * 6 + 0]));  // (F [Y_j]x^T)[a][c] = (Y_j x F[a, :])_c
    acc[a * 6 + 1] = fma(Yj[2], f0, fma(-Yj[0], f2, acc[a * 6 + 1]));
    acc[a * 6 + 2] = fma(Yj[0], f1, fma(-Yj[1], f0, acc[a * 6 + 2]));
  }
}

// k_schur_reg3.  A workgroup is bound to one tile (camera group a x camera group b) of the reduced camera system; every THREAD owns one
// camera-pair block of the tile for the whole kernel and keeps its NC x NC accumulators in registers (NC = 9: three threads per block, three
// rows each) — no atomics, no S tile in LDS, a fixed summation order.  A tile's work is a stream of chunks (schur_plan.h):
//   * the plan equalises the pairs per block of every chunk (cap t, first-fit dealing over a region of chunks), so a wave spends ~75 % of its
//     lane-iterations on real pairs;
//   * the pair list is transposed: iteration `it` of wave w reads 64 consecutive codes, one per lane; a code holds the LDS addresses of the two
//     records, idle lanes get the address of an all-zero record.  No per-thread slice table, no pair list in LDS, no divergence: the trip
//     count is wave-uniform and the body is straight-line code;
//   * the plan also picks the slot of every record inside the chunk so that the 16 lanes the LDS serves together read from 16 different bank
//     groups (schur_plan.h, "LDS bank conflicts");
//   * the records go straight from HBM into LDS (global_load_lds_dwordx4: 64 lanes, 64 consecutive 16-byte pieces of LDS, any global
//     addresses) into the buffer that is NOT being read (two buffers), so a trip is: wait for the loads issued a trip ago, one barrier, issue
//     the next chunk's loads, multiply.  No staging registers.
//
template <int NC> struct Reg3Cfg {
  static constexpr int REC = SchurRec<NC>::REC, LST = SchurRec<NC>::LST;
  static constexpr int SPLIT = (NC == 9) ? 3 : 1;
  static constexpr int CODE_THREADS = BLOCK, CODE_WAVES = BLOCK / WAVE;             // blocks of a tile = pair-code streams
  static constexpr int REG_BLOCK = BLOCK * SPLIT, NWAVES = REG_BLOCK / WAVE;        // threads / waves of a workgroup
  static constexpr int GROUP = 16;                                                  // cameras per group: GROUP^2 blocks <= CODE_THREADS
  // pair codes per block and chunk that travel in registers (a code beyond them is loaded inside the pair loop: a vmcnt(0) behind the record loads
  // in flight)
  static constexpr int NCD = (NC == 6) ? 8 : 4;
  static constexpr int PAIR_CAP = 0;
  static constexpr int SCHUNK = (NC == 9) ? 384 : 320;                              // slots per chunk (per LDS buffer)
  static constexpr int EPW = SCHUNK / NWAVES;                                       // slots loaded by one wave (80 / 32)
  static constexpr int NLD = (EPW * LST + WAVE - 1) / WAVE;                         // load instructions per wave and chunk (9 / 6)
  static constexpr int WAVE_PIECES = NLD * WAVE;                                    // LDS pieces of one wave's run, padded to whole loads
  static constexpr int ZERO_PIECE = NWAVES * WAVE_PIECES;                           // all-zero record behind the chunk, in each buffer
  static constexpr int BUF_PIECES = ZERO_PIECE + LST + 1;                           // (+1: keeps the second buffer 32-byte aligned)
  static constexpr int NBUF = 2;                                                    // chunk buffers: a gather is issued one trip before it is read
  static constexpr int SET_PIECES = NBUF * BUF_PIECES;
  // NC = 9: the per-lane constants of the gather (record of the wave's run and piece of the record a lane fetches with load k) live in a table behind
  // the buffers and are read back every trip: as loop invariants they are twelve 64-bit base addresses and six shuffle addresses per lane, and with
  // the 36 accumulators of the corner role (pair9_corner) the kernel's 168 registers no longer held them (spilled, and reloaded in the middle of the
  // issue sequence behind a vmcnt(0))
  static constexpr bool LANE_TABLE = (NC == 9);
  static constexpr int TABLE_PIECES = LANE_TABLE ? (NLD * WAVE * 4 + 15) / 16 : 0;
  static constexpr size_t LDS_BYTES = (size_t)(SET_PIECES + TABLE_PIECES) * 16;
  static constexpr int LAUNCH_THREADS = REG_BLOCK;
  static_assert(SCHUNK % NWAVES == 0 && EPW % 16 == 0 && EPW <= 2 * WAVE && (LST & 1) == 1, "staging layout");
  static_assert(ZERO_PIECE + LST < 65536, "piece addresses are 16 bit");
  static_assert(CODE_WAVES % 4 == 0, "iteration counts: four waves per word");
};

// primed pair product of six-parameter cameras, T'_i T'_j^T added to the thread's 6 x 6 accumulators: 99 FP64 instructions on 12 + 12 doubles
__device__ __forceinline__ void pair6(double (&acc)[6][6], const double2* __restrict__ Ri, const double2* __restrict__ Rj) {
  const double2 i0 = Ri[0], i1 = Ri[1], i2 = Ri[2], i3 = Ri[3], i4 = Ri[4], i5 = Ri[5];
  const double2 j0 = Rj[0], j1 = Rj[1], j2 = Rj[2], j3 = Rj[3], j4 = Rj[4], j5 = Rj[5];
  const double Yi[3] = {i0.x, i0.y, i1.x}, Yj[3] = {j0.x, j0.y, j1.x};
  const double Qi[9] = {i1.y, i2.x, i2.y, i3.x, i3.y, i4.x, i4.y, i5.x, i5.y};
  const double Qj[9] = {j1.y, j2.x, j2.y, j3.x, j3.y, j4.x, j4.y, j5.x, j5.y};
  double M[3][6];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int b2 = 0; b2 < 3; ++b2)
      M[a][3 + b2] = fma(Qi[3 * a + 2], Qj[3 * b2 + 2], fma(Qi[3 * a + 1], Qj[3 * b2 + 1], Qi[3 * a] * Qj[3 * b2]));
    M[a][0] = fma(Yj[1], M[a][5], -(Yj[2] * M[a][4]));
    M[a][1] = fma(Yj[2], M[a][3], -(Yj[0] * M[a][5]));
    M[a][2] = fma(Yj[0], M[a][4], -(Yj[1] * M[a][3]));
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    acc[0][c] = fma(Yi[1], M[2][c], fma(-Yi[2], M[1][c], acc[0][c]));
    acc[1][c] = fma(Yi[2], M[0][c], fma(-Yi[0], M[2][c], acc[1][c]));
    acc[2][c] = fma(Yi[0], M[1][c], fma(-Yi[1], M[0][c], acc[2][c]));
    acc[3][c] += M[0][c]; acc[4][c] += M[1][c]; acc[5][c] += M[2][c];
  }
}

// applies P_i^T (.) P_j, P = blockdiag(J_l, I), to the rows r0 .. r0 + RH - 1 a thread holds of a primed block (epilogue of the pair kernels)
template <int NC, int RH>
__device__ __forceinline__ void unprime_rows(double (&acc)[RH][NC], int r0, const double* __restrict__ Ji, const double* __restrict__ Jj) {
  if (r0 == 0) {  // rows 0..2 <- J_i^T rows 0..2 (all columns)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double b0 = acc[0][c], b1 = acc[1][c], b2 = acc[2][c];
#pragma unroll
      for (int r = 0; r < 3; ++r) acc[r][c] = Ji[r] * b0 + Ji[3 + r] * b1 + Ji[6 + r] * b2;
    }
  }
#pragma unroll
  for (int r = 0; r < RH; ++r) {  // columns 0..2 <- (.) J_j, every row the thread owns
    const double a0 = acc[r][0], a1 = acc[r][1], a2 = acc[r][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[r][c] = a0 * Jj[c] + a1 * Jj[3 + c] + a2 * Jj[6 + c];
  }
}

// The body is a function of its own with `Trec` as a __restrict__ PARAMETER: inlined into the kernel, every access in it carries
// alias-scope metadata, and only with that does the compiler's wait-count insertion let a ds_read pass a pending LDS-DMA load (with
// the body written directly in the kernel it put a vmcnt(0) in front of the first record read of every trip: the wave sat out the
// gather it had just issued).
// CLK (profiling build, -DCBA_PROFILING + CBA_SCHUR_CLOCK=1): per wave the shader clocks spent waiting for loads, in barriers, issuing, multiplying.
template <int NC, int SPLIT, bool CLK = false>
__device__ __forceinline__ void schur_reg3_body(const TilePlan& tp, const unsigned* __restrict__ p_nit, const int* __restrict__ p_code_start,
                                                const int* __restrict__ p_chunk_start, const unsigned* __restrict__ p_codes, const int* __restrict__ p_obs,
                                                const double* __restrict__ Trec, double* __restrict__ partial, double* sh, long long* __restrict__ clk = nullptr,
                                                const double* __restrict__ tab = nullptr) {
  using Cfg = Reg3Cfg<NC>;
  static_assert(SPLIT == Cfg::SPLIT, "split");
  constexpr int REG_BLOCK = Cfg::REG_BLOCK, LST = Cfg::LST, NLD = Cfg::NLD, EPW = Cfg::EPW, NWORD = Cfg::CODE_WAVES / 4;
  constexpr int NCD = Cfg::NCD;                 // codes of a chunk (and block) that travel in registers

  const int nblk = tp.g * tp.g;
  const int rep = tp.rep;  // (round 6: nine-parameter cameras too — a 4-camera rig with free intrinsics ran its 300 pairs per block on 16 code threads: 53 us)
  const int tid = (int)threadIdx.x;
  const int wg = logical_workgroup((int)blockIdx.x, (int)gridDim.x);  // csrc/wg_binding.h: interleaved over the dispatch order
  double2* sh_p = reinterpret_cast<double2*>(sh);  // two chunk buffers, in 16-byte pieces
  const int ct = tid % BLOCK;                   // code thread: the SPLIT parts of a block multiply the same pairs
  const int pw = __builtin_amdgcn_readfirstlane(ct / WAVE), lane = ct % WAVE;
  const int sw = __builtin_amdgcn_readfirstlane(tid / WAVE);  // loading wave of the set
  const int half = tid / BLOCK;  // role of the thread in its block (NC = 9: sub-block 0, 1 or 2, see pair9_cols; 0 when SPLIT == 1)
  double acc[6][6];              // NC = 6 and role 0: the 6 x 6 block; role 1: [9][3], role 2: [3][6] in the same registers
  double* const accf = &acc[0][0];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[r][c] = 0.0;

  // this thread's block: virtual thread vt of the plan; rep > 1: vt = slot * nblk + block
  const int vt = (SPLIT == 1) ? tid : ct;
  const int blk = (rep > 1) ? vt % nblk : vt, slot = (rep > 1) ? vt / nblk : 0;
  const bool owner = slot < rep && blk < nblk;  // threads beyond the tile's blocks (ragged groups) only help to load
  double* dst = partial + ((long)wg * rep + min(slot, rep - 1)) * tp.tile_elems + (long)min(blk, nblk - 1) * NC * NC;

  // chunk range of this workgroup (one without chunks — more workgroups than chunks in the range — gathers somebody's valid chunk and multiplies nothing)
  const int stride = tp.wg_stride[wg], ch_end = tp.wg_end[wg];
  const int own_trips = tp.wg_first[wg] < ch_end ? (ch_end - 1 - tp.wg_first[wg]) / stride + 1 : 0;
  const int first = own_trips ? tp.wg_first[wg] : max(ch_end - 1, 0);
  const int trips = max(own_trips, 1);
  for (int k = tid; k < Cfg::NBUF * LST; k += REG_BLOCK) sh_p[(k / LST) * Cfg::BUF_PIECES + Cfg::ZERO_PIECE + k % LST] = make_double2(0.0, 0.0);
  int* const lane_table = reinterpret_cast<int*>(sh_p + Cfg::SET_PIECES);  // [NLD][WAVE]: el | piece << 16 (Cfg::LANE_TABLE)
  if constexpr (Cfg::LANE_TABLE) {
    constexpr int Q = WAVE / LST, RM = WAVE % LST;
    for (int e = tid; e < NLD * WAVE; e += REG_BLOCK) {
      const int k = e / WAVE, l = e % WAVE;
      int piece = k * RM + l % LST;
      const int el = min(k * Q + l / LST + piece / LST, EPW - 1);
      piece = min(piece % LST, SchurRec<NC>::NPH - 1);
      lane_table[e] = el | (piece << 16);
    }
    __syncthreads();
  }
  const int last = first + (own_trips ? (own_trips - 1) * stride : 0);  // last chunk of this workgroup
  // Per trip every wave issues, in this order and WITHOUT waiting in between: the codes of the next chunk and the record indices
  // of the chunk after next (their addresses come from the iteration counts / offsets loaded a trip earlier), the counts and
  // offsets of the chunks behind those, the records of the next chunk; then it multiplies the current chunk while all of that
  // is in flight.  (Until round 2's last revision addresses were computed from values loaded in the same trip: the wait for them
  // was a vmcnt(0) behind the record loads, i.e. every wave sat out its own gather before it multiplied.)
  struct Raw { unsigned nit[NWORD]; int code_start; int obs_start; };  // counts / code offset of one chunk, stream offset of its successor
  auto load_raw = [&](int chunk, int successor) {
    Raw r;
#pragma unroll
    for (int q = 0; q < NWORD; ++q) r.nit[q] = p_nit[(long)chunk * NWORD + q];
    r.code_start = p_code_start[chunk];
    r.obs_start = p_chunk_start[successor];
    return r;
  };
  auto load_indices = [&](int obs_start, int* iA, int* iB) {
    const int* src = p_obs + obs_start + sw * EPW;
    *iA = src[lane];
    if (EPW > WAVE) *iB = src[WAVE + (lane & (EPW - WAVE - 1))];
  };
  static_assert(EPW <= WAVE || ((EPW - WAVE) & (EPW - WAVE - 1)) == 0, "second index register");
  unsigned cd[NCD];
  int n_nx = 0;
  long code_nx = 0;
  auto load_codes = [&](const Raw& r) {
    int pre = 0, mine = 0;
#pragma unroll
    for (int q = 0; q < NWORD; ++q) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int n = (int)((r.nit[q] >> (8 * w)) & 0xffu);
        pre += (4 * q + w < pw) ? n : 0;
        mine = (4 * q + w == pw) ? n : mine;
      }
    }
    n_nx = __builtin_amdgcn_readfirstlane(mine);
    code_nx = (long)r.code_start + (long)pre * WAVE + lane;
#pragma unroll
    for (int k = 0; k < NCD; ++k) cd[k] = p_codes[code_nx + k * WAVE];  // past the wave's last iteration: somebody else's codes, never used
  };
  // load k of this wave fills LDS pieces [k * 64, k * 64 + 64) of the wave's run: piece (k * 64 + lane) % LST of slot
  // (k * 64 + lane) / LST; the record index of the slot comes from the wave's index registers (ds_bpermute)
  auto issue = [&](int buf, int idxA, int idxB) {
    constexpr int Q = WAVE / LST, RM = WAVE % LST;
    double2* wbase = sh_p + buf * Cfg::BUF_PIECES + sw * Cfg::WAVE_PIECES;
    // all index exchanges first, then the loads: written load by load the sequence was bpermute - wait - load, NLD LDS round trips in
    // a row under the pair loops' LDS traffic (phase clocks: 40 % of a wave's time went into issuing)
    const double2* g[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      int piece, el;
      if constexpr (Cfg::LANE_TABLE) {
        const int e = lane_table[k * WAVE + lane];
        el = e & 0xffff; piece = e >> 16;
      } else {
        piece = k * RM + lane % LST;
        el = k * Q + lane / LST + piece / LST;
        piece = min(piece % LST, SchurRec<NC>::NPH - 1);
        el = min(el, EPW - 1);  // tail lanes of the last load: a valid record, landing in the padding of the wave's run
      }
      const bool useB = EPW > WAVE && k * WAVE >= WAVE * LST;
      const int idx = useB ? __shfl(idxB, el - WAVE, WAVE) : __shfl(idxA, el, WAVE);
      g[k] = reinterpret_cast<const double2*>(Trec + (long)idx * SchurRec<NC>::HREC) + piece;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NLD; ++k)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g[k], (void __attribute__((address_space(3)))*)(wbase + k * WAVE), 16, 0, 0);
  };
  auto pair = [&](const double2* bufp, unsigned code) {
    const double2* Ri = bufp + (code & 0xffffu);
    const double2* Rj = bufp + (code >> 16);
    if constexpr (NC == 6) {
      pair6(acc, Ri, Rj);
    } else {
      if (half == 0) pair9_corner(acc, Ri, Rj);
      else if (half == 1) pair9_cols(accf, Ri, Rj);
      else pair9_rows(accf, Ri, Rj);
    }
  };

  long long clk_sum[6] = {0, 0, 0, 0, 0, 0};
  const long long t_start = CLK ? clock64() : 0;
  int idxA = 0, idxB = 0;
  const int second = min(first + stride, last);
  Raw raw = load_raw(first, second);
  load_indices(p_chunk_start[first], &idxA, &idxB);
  issue(0, idxA, idxB);
  load_codes(raw);                           // the first chunk's codes
  load_indices(raw.obs_start, &idxA, &idxB); // the second chunk's indices
  raw = load_raw(second, min(second + stride, last));
  int buf = 0;
  // The two workgroups of a CU are arbitrated by priority, then AGE: the one dispatched first wins every contested issue slot, finishes at ~104 us and
  // leaves the other, alone and at the one-wave issue rate, to finish at ~120 (device stamps, round 6: profiles/r06_pair_lifetimes.txt).  With
  // prio_shift >= 0 the halves of the dispatch order take the higher priority in alternate groups of trips, so that both make the same progress.
  const int young = ((int)blockIdx.x >= (int)gridDim.x / 2) ? 1 : 0;
  for (int trip = 0; trip < trips; ++trip) {
    if (tp.prio_shift >= 0) {
      if (((trip >> tp.prio_shift) + young) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    const int cur = min(first + trip * stride, last);
    // everything issued a trip ago has landed (the records of `cur` in `buf`, its codes, the counts and indices of the next
    // chunk), and every wave is done reading the other buffer
    long long tA = 0, tB = 0, tC = 0, tD = 0;
    if (CLK) tA = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (CLK) tB = clock64();
    // The values loaded a trip ago pass through an empty asm: to the compiler they are no longer results of loads that are
    // still pending (it cannot see the wait above, and for registers whose load sits behind the loop's back edge it inserted
    // its own vmcnt(0) at their first use — in the middle of the pair loop, behind the record loads just issued).
#pragma unroll
    for (int k = 0; k < NCD; ++k) asm volatile("" : "+v"(cd[k]));
    // (raw is wave-uniform: left alone the compiler moves it to SGPRs right behind its load — v_readfirstlane, i.e. a wait in the
    // middle of the issue sequence; scalar loads are not used inside the loop, the LDS-DMA intrinsic counts as a clobber)
#pragma unroll
    for (int q = 0; q < NWORD; ++q) asm volatile("" : "+v"(raw.nit[q]));
    asm volatile("" : "+v"(raw.code_start));
    asm volatile("" : "+v"(raw.obs_start));
    asm volatile("" : "+v"(idxA));
    if (EPW > WAVE) asm volatile("" : "+v"(idxB));
    __syncthreads();
    if (CLK) tC = clock64();
    unsigned cc[NCD];
#pragma unroll
    for (int k = 0; k < NCD; ++k) cc[k] = cd[k];
    const int n_cur = (trip < own_trips) ? n_nx : 0;
    const long code_cur = code_nx;
    const int nxt2 = min(min(cur + stride, last) + stride, last);
    load_codes(raw);                        // codes of the next chunk: addresses from registers, no wait
    int idxA_n = 0, idxB_n = 0;
    load_indices(raw.obs_start, &idxA_n, &idxB_n);  // indices of the chunk after next
    const Raw raw_n = load_raw(nxt2, min(nxt2 + stride, last));
    issue(buf ^ 1, idxA, idxB);             // records of the next chunk
    __builtin_amdgcn_sched_barrier(0);
    if (CLK) { tD = clock64(); __builtin_amdgcn_sched_barrier(0); }
    const double2* bufp = sh_p + buf * Cfg::BUF_PIECES;
    // (reading the records of pair it + 1 while pair it is multiplied — there would be registers for it in the NC = 6 kernel — measured
    // slower: 112k instead of 98k clocks per wave in the pair phase)
#pragma unroll
    for (int it = 0; it < NCD; ++it)
      if (it < n_cur) pair(bufp, cc[it]);
    for (int it = NCD; it < n_cur; ++it) pair(bufp, p_codes[code_cur + (long)it * WAVE]);  // more pairs of one block in a chunk than travel in registers
    if (CLK) {
      __builtin_amdgcn_sched_barrier(0);
      const long long tE = clock64();
      clk_sum[0] += tB - tA; clk_sum[1] += tC - tB; clk_sum[2] += tD - tC; clk_sum[3] += tE - tD; clk_sum[4] += 1; clk_sum[5] += n_cur;
    }
    raw = raw_n; idxA = idxA_n; idxB = idxB_n;
    buf ^= 1;
  }
  if (CLK && lane == 0 && clk) {
    long long* o = clk + ((long)wg * Cfg::NWAVES + sw) * 8;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = clk_sum[k];
    o[6] = clock64() - t_start;
    o[7] = ((long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 40) | (wall_clock64() & 0xffffffffffLL);  // HW_ID, start stamp (100 MHz)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last trip's loads target LDS: let them land before the workgroup retires
  // The accumulators hold PRIMED blocks T'_i T'_j^T (T' = [[Y]x Q ; Q ; T_intr]); the true block is P_i^T (.) P_j with P = blockdiag(J_l, I).  That
  // map is linear, so every thread applies it to its own partial block here — ~100 FMAs once per kernel — and no separate pass over the reduced
  // matrix is needed.
  if (blk < nblk) {
    const int tile = tp.wg_tile[wg];
    const int ga = tp.tile_a[tile], gb = tp.tile_b[tile];
    const int ca0 = tp.group_cam_begin[ga], na = tp.group_cam_begin[ga + 1] - ca0;
    const int cb0 = tp.group_cam_begin[gb], nb = tp.group_cam_begin[gb + 1] - cb0;
    const int li = blk / tp.g, lj = blk % tp.g;
    int ci = -1, cj = -1;
    if (ga == gb && lj <= li) {  // helper thread of a diagonal tile: the (i, i) items of one camera (schur_plan.h)
      const int hk = li * (li + 1) / 2 + lj;
      ci = cj = ca0 + hk % max(na, 1);
    } else if (li < na && lj < nb) {  // (else a ragged group: no such camera, nothing was accumulated)
      ci = ca0 + li; cj = cb0 + lj;
    }
    if (ci >= 0) {
      const double* Ji = tab + (long)ci * CAMTAB_DOUBLES + 12;  // CamTab::Jl, row-major
      const double* Jj = tab + (long)cj * CAMTAB_DOUBLES + 12;
      if constexpr (NC == 6) {
        unprime_rows<6, 6>(acc, 0, Ji, Jj);
      } else {
        // (the second matrix is read after the first is done with: all eighteen entries beside 36 accumulators do not fit the kernel's registers)
        if (half != 2) {  // rows 0..2 <- J_i^T rows 0..2
          const double j0 = Ji[0], j1 = Ji[1], j2 = Ji[2], j3 = Ji[3], j4 = Ji[4], j5 = Ji[5], j6 = Ji[6], j7 = Ji[7], j8 = Ji[8];
          if (half == 0) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              const double b0 = acc[0][c], b1 = acc[1][c], b2 = acc[2][c];
              acc[0][c] = j0 * b0 + j3 * b1 + j6 * b2; acc[1][c] = j1 * b0 + j4 * b1 + j7 * b2; acc[2][c] = j2 * b0 + j5 * b1 + j8 * b2;
            }
            asm volatile("; role 0 rows" ::: "memory");
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const double b0 = accf[c], b1 = accf[3 + c], b2 = accf[6 + c];
              accf[c] = j0 * b0 + j3 * b1 + j6 * b2; accf[3 + c] = j1 * b0 + j4 * b1 + j7 * b2; accf[6 + c] = j2 * b0 + j5 * b1 + j8 * b2;
            }
            asm volatile("; role 1 rows" ::: "memory");
          }
        }
        if (half != 1) {  // columns 0..2 <- (.) J_j
          const double j0 = Jj[0], j1 = Jj[1], j2 = Jj[2], j3 = Jj[3], j4 = Jj[4], j5 = Jj[5], j6 = Jj[6], j7 = Jj[7], j8 = Jj[8];
#pragma unroll
          for (int r = 0; r < 6; ++r)
            if (r < 3 || half == 0) {
              const double a0 = acc[r][0], a1 = acc[r][1], a2 = acc[r][2];
              acc[r][0] = a0 * j0 + a1 * j3 + a2 * j6; acc[r][1] = a0 * j1 + a1 * j4 + a2 * j7; acc[r][2] = a0 * j2 + a1 * j5 + a2 * j8;
            }
        }
      }
    }
  }
  if (!owner) return;
  if (NC == 6 || half == 0) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[r * NC + c] = acc[r][c];
    asm volatile("; corner stored" ::: "memory");  // (the three store sequences stay three: merged by the compiler they become one with 36 selected offsets in registers)
  } else if (half == 1) {
#pragma unroll
    for (int r = 0; r < 9; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) dst[r * NC + 6 + c] = accf[r * 3 + c];
    asm volatile("; columns stored" ::: "memory");
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) dst[(6 + r) * NC + c] = accf[r * 6 + c];
    asm volatile("; rows stored" ::: "memory");
  }
}

template <int NC, int SPLIT, int MINW>
__global__ void __launch_bounds__((Reg3Cfg<NC>::LAUNCH_THREADS), MINW)
k_schur_reg3(TilePlan tp, const double* __restrict__ Trec, double* __restrict__ partial, const double* __restrict__ tab) {
  CBA_STAMP(ST_PAIRS);
  extern __shared__ __attribute__((aligned(16))) double sh[];
  schur_reg3_body<NC, SPLIT, false>(tp, tp.nit, tp.code_start, tp.chunk_start, tp.codes, tp.obs, Trec, partial, sh, nullptr, tab);
}

#ifdef CBA_PROFILING  // tools/build_profiling_lib.sh: the phase-clock build of the pair kernel is not part of the product library
template <int NC, int SPLIT, int MINW>
__global__ void __launch_bounds__((Reg3Cfg<NC>::LAUNCH_THREADS), MINW)
k_schur_reg3_clk(TilePlan tp, const double* __restrict__ Trec, double* __restrict__ partial, long long* __restrict__ clk, const double* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  schur_reg3_body<NC, SPLIT, true>(tp, tp.nit, tp.code_start, tp.chunk_start, tp.codes, tp.obs, Trec, partial, sh, clk, tab);
}
#endif

// Reduce the partials of k_schur_reg over the workgroups of each tile (fixed order).  blockDim = (64, Y): x walks the per-thread partial entries,
// y splits the partial rows of the tile Y ways, Y = 4 or REG_REDUCE_Y_MAX = 16 (the caller's choice: a small rig has ONE tile and up to 500 partial
// rows — with four ways that is 60 dependent load-add steps per thread, 13 us of cfg2's iteration, 7 with sixteen; cfg4 has 51 rows per tile and
// is 3 us faster with four).  Off-diagonal blocks go straight into Sacc; the helper-thread entries of diagonal tiles are parked in `red` and
// folded per camera by k_reg_fold.
constexpr int kSchurRegMaxGroupK = 16;  // cameras per group of the pair kernel (Reg3Cfg::GROUP)
// sum of src[w * stride], w = w0 + y, w0 + y + Y, ... < w1, in a fixed order with eight loads in flight (round 6: the row sums of the reduction kernels were
// two chains of dependent load-add steps per thread — 38 deep for the diagonal camera blocks of cfg3, 17 for cfg2's one tile: 19 and 9 us of latency)
__device__ __forceinline__ double strided_sum(const double* __restrict__ src, long stride, int w0, int w1, int y, int Y) {
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int w = w0 + y;
  for (; w + 7 * Y < w1; w += 8 * Y) {
#pragma unroll
    for (int q = 0; q < 8; ++q) a[q] += src[(long)(w + q * Y) * stride];
  }
  if (w + 3 * Y < w1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] += src[(long)(w + q * Y) * stride];
    w += 4 * Y;
  }
  for (; w < w1; w += Y) a[4] += src[(long)w * stride];  // (at most three)
  return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}
constexpr int REG_REDUCE_Y_MAX = 16;
constexpr int B_SLICES = 8;  // the rhs accumulator b is kept as B_SLICES rows of ncp_pad entries (k_reg_reduce); an entry is their sum, slice order
__device__ __forceinline__ double b_entry(const double* __restrict__ bacc, int b_width, int i) {
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < B_SLICES; ++q) s += bacc[(long)q * b_width + i];
  return s;
}
__global__ void __launch_bounds__(64 * REG_REDUCE_Y_MAX)
k_reg_reduce(TilePlan tp, const int* __restrict__ tile_wg_begin, const double* __restrict__ partial,
             const int* __restrict__ cam_off, const int* __restrict__ cam_np, int NCt, int ncp,
             double* __restrict__ Sacc, double* __restrict__ red, int n_tiles, const double* __restrict__ partial_b = nullptr, int b_rows = 0,
             int b_width = 0, double* __restrict__ b_out = nullptr) {
  CBA_STAMP(ST_REG_REDUCE);
  const int Y = (int)blockDim.y;  // 4 or 16
  __shared__ double sh[REG_REDUCE_Y_MAX][64];
  if ((int)blockIdx.y >= n_tiles) {
    // grid rows n_tiles .. n_tiles + B_SLICES - 1 (round 5): the rhs rows k_tprep left per workgroup are summed HERE (fixed order), beside the tiles'
    // partials, instead of by a k_reduce_rows launch of six workgroups between k_tprep and the pair kernel.  Slice s takes the rows
    // [s R / B_SLICES, (s + 1) R / B_SLICES) and leaves ITS sum in b_out[s][.]: the consumers (k_schur_finalize, k_small_solve, k_tri_pack) add the
    // B_SLICES values of an entry in slice order (one workgroup per column block summing all 512 rows took 22 us, three times the tiles' reduction).
    const int sl = (int)blockIdx.y - n_tiles;
    const int r0 = (int)((long)b_rows * sl / B_SLICES), r1 = (int)((long)b_rows * (sl + 1) / B_SLICES);
    const int j = blockIdx.x * 64 + threadIdx.x;
    sh[threadIdx.y][threadIdx.x] = (j < b_width) ? strided_sum(partial_b + j, b_width, r0, r1, threadIdx.y, Y) : 0.0;
    __syncthreads();
    if (threadIdx.y == 0 && j < b_width) {
      double tot = 0.0;
      for (int y = 0; y < Y; ++y) tot += sh[y][threadIdx.x];
      b_out[(long)sl * b_width + j] = tot;
    }
    return;
  }
  const int t = blockIdx.y;
  const int ga = tp.tile_a[t], gb = tp.tile_b[t];
  const int ca0 = tp.group_cam_begin[ga], na = tp.group_cam_begin[ga + 1] - ca0;
  const int cb0 = tp.group_cam_begin[gb], nb = tp.group_cam_begin[gb + 1] - cb0;
  const int g = tp.g, bsz = NCt * NCt;
  const int e = blockIdx.x * 64 + threadIdx.x;
  const int w0 = tile_wg_begin[t] * tp.rep, w1 = tile_wg_begin[t + 1] * tp.rep;  // rep partial rows per workgroup
  const bool diag = (ga == gb);
  long dst = -1;       // >= 0: Sacc index;  -2: park in red
  if (e < g * g * bsz) {
    const int beta = e / bsz, rc = e % bsz;
    const int li = beta / g, lj = beta % g, r = rc / NCt, c = rc % NCt;
    if (diag && lj <= li) dst = -2;
    else if (li < na && lj < nb && r < cam_np[ca0 + li] && c < cam_np[cb0 + lj])
      dst = (long)(cam_off[ca0 + li] + r) * ncp + cam_off[cb0 + lj] + c;
  }
  sh[threadIdx.y][threadIdx.x] = (dst != -1) ? strided_sum(partial + e, tp.tile_elems, w0, w1, threadIdx.y, Y) : 0.0;
  __syncthreads();
  if (threadIdx.y != 0 || dst == -1) return;
  double tot = 0.0;
  for (int y = 0; y < Y; y += 4) tot += (sh[y][threadIdx.x] + sh[y + 1][threadIdx.x]) + (sh[y + 2][threadIdx.x] + sh[y + 3][threadIdx.x]);
  if (dst >= 0) Sacc[dst] = tot;
  else red[(long)ga * tp.tile_elems + e] = tot;
}

// Diagonal blocks of the diagonal tiles: sum the (already workgroup-reduced) accumulators of each camera's
// helper threads.  Helper k (k-th thread with tid % g <= tid / g, in tid order) serves camera k mod na.
// grid = (ceil(g * NC*NC / 64), G), block = 64.
__global__ void __launch_bounds__(64)
k_reg_fold(TilePlan tp, const double* __restrict__ red, const int* __restrict__ cam_off, const int* __restrict__ cam_np,
           int NCt, int ncp, double* __restrict__ Sacc) {
  const int ga = blockIdx.y;
  const int ca0 = tp.group_cam_begin[ga], na = tp.group_cam_begin[ga + 1] - ca0;
  const int g = tp.g, bsz = NCt * NCt;
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= g * bsz) return;
  const double* src = red + (long)ga * tp.tile_elems;
  const int cam = q / bsz, r = (q % bsz) / NCt, c = (q % bsz) % NCt;
  if (cam >= na || c < r || c >= cam_np[ca0 + cam]) return;
  double s = 0.0;
  for (int k = cam; k < g * (g + 1) / 2; k += na) {  // helper k = (li, lj) with k = li (li + 1) / 2 + lj
    int li = (int)((sqrtf(8.0f * k + 1.0f) - 1.0f) * 0.5f);
    while (li * (li + 1) / 2 > k) --li;
    while ((li + 1) * (li + 2) / 2 <= k) ++li;
    const int lj = k - li * (li + 1) / 2;
    s += src[(long)(li * g + lj) * bsz + r * NCt + c];
  }
  Sacc[(long)(cam_off[ca0 + cam] + r) * ncp + cam_off[ca0 + cam] + c] = s;
}

// Sharded solves exchange the reduced camera system: only the upper triangle of Sacc carries data, so the all-reduce moves
// ncp (ncp + 1) / 2 + ncp doubles (0.59 MB for cfg4) instead of ncp^2 + ncp.  dir = 0 packs [triangle | b], dir = 1 unpacks.
__global__ void __launch_bounds__(256)
k_tri_pack(double* __restrict__ Sacc, double* __restrict__ tri, int ncp, int dir, int b_width) {
  const long ntri = (long)ncp * (ncp + 1) / 2;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < (long)ncp * ncp + ncp; t += (long)gridDim.x * 256) {
    if (t >= (long)ncp * ncp) {  // b: the sum of its slices travels; the all-reduced value comes back as slice 0, the others are cleared
      const long k = t - (long)ncp * ncp;
      double* bacc = Sacc + (long)ncp * ncp;
      if (dir == 0) tri[ntri + k] = b_entry(bacc, b_width, (int)k);
      else {
        bacc[k] = tri[ntri + k];
#pragma unroll
        for (int q = 1; q < B_SLICES; ++q) bacc[(long)q * b_width + k] = 0.0;
      }
      continue;
    }
    const int row = (int)(t / ncp), col = (int)(t % ncp);
    if (col < row) continue;
    const long q = (long)row * ncp - (long)row * (row - 1) / 2 + (col - row);
    if (dir == 0) tri[q] = Sacc[t]; else Sacc[t] = tri[q];
  }
}

// ------------------------------------------------------------------------------------------------
// Dense solve of the reduced camera system  S dc = rhs  (n <= 1152) by blocked Cholesky, NB = 32.
// The work matrix is (n+1) x ldw, row-major: rows 0..n-1 hold S, row n holds rhs^T.  Row n is "below" every
// diagonal block, so the panel solves of the factorisation turn it into y^T = (L^-1 rhs)^T for free — the forward
// substitution needs no kernel of its own; L^T x = y becomes x = T y with T = L^-T built beside the factorisation (k_chol_apply).
constexpr int NB = 32;

// broadcast lane `src` (wave-uniform) of a double through SGPRs: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Factor the NB x NB block parked in `D` (LDS, 2 NB rows of stride NB + 1, `nb` live rows, identity-padded) with wave 0; the factor stays
// in LDS: rows 0..NB-1 of D hold L (zeros above the diagonal), row NB + c holds column c of X = L^-1 (chol_factor_store copies both out).
// Left-looking by columns, lane r < NB keeps row r of L in registers:  v_r = D_rj - sum_{t<j} L_rt L_jt, L_jj = sqrt(v_j), L_rj = v_r / L_jj.
// Lanes NB.. run the SAME instruction stream on a column of X (lane NB + c keeps column c):  X_jc = (delta_jc - sum_{t<j} L_jt X_tc) / L_jj
// is the row recurrence with the lane's own values X_tc in the place of L_rt and delta_jc in the place of D_rj; the broadcast row L_j,: and
// the pivot are shared.  The inverse costs nothing (the lanes were idle) and turns the panel solves of the next step and the backward
// substitution into matrix products (k_chol_step, k_chol_apply).
// Row j + 1 of L is read back from LDS as broadcasts (every lane stores its new entry D[lane][j] at pivot j; a wave executes its DS
// instructions in order, wave-scope fences only pin the compiler) at the top of pivot j, and the sum over its final columns t < j is formed
// between the instructions of pivot j's chain; the newest entry L_j+1,j travels by v_readlane.
// A pivot is a chain of DEPENDENT FP64 instructions, ~300 cycles, and 32 of them are the critical path of a panel step, so the chain is short:
//   * the pivot d_j = a_j - L_j,j-1^2 is formed by lane j from its own registers (no broadcast of L_j,j-1 in front of it);
//   * 1/sqrt(d) is v_rsq_f64 and ONE third-order correction folded into the product with the lane's value:
//       u = v y0, e = 1 - (d y0) y0, L_rj = u + (u e)(1/2 + 3/8 e)            (rsq, d y0, e, u e, fma: five levels);
//   * the checks of the pivot (positive, finite) and the zeros above the diagonal are applied beside the chain.  A failed pivot raises
//     flags[2] (the caller discards the step) and the block is replaced by the identity, so nothing non-finite leaves it.
// History (tools/chol_factor_bench, us per block): right-looking updates of 32 live row registers 20 (1.3 KB of scratch per lane);
// left-looking, IEEE sqrt and divide 9; v_rsq + two Newton steps, pivot replaced by 1.0 on the chain, each lane storing its row and its
// column of X to global memory 7.3; 2 x 2 pivots (two independent rsqrt chains) 6.8; two 16-wide stages joined by FP64 MFMA products 5.9
// (a 16-wide pivot takes 300 cycles, a 32-wide one 360: the chain, not the instruction count); this form 5.5, of which the copy-out 0.35.
__device__ __forceinline__ void chol_factor_block(double (*D)[NB + 1], int nb, int* __restrict__ flags) {
  int lane = threadIdx.x;
  asm volatile("" : "+v"(lane));  // (a caller's loop must not hoist the lane masks of all 32 pivots out of it: they do not fit the SGPRs)
  const int r = lane & (NB - 1);
  const bool isX = lane >= NB;
  double row[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) row[c] = isX ? (c == r ? 1.0 : 0.0) : D[r][c];
  int bad = 0;
  double s_prev = 0.0, raw_prev = 0.0;  // L_j,j-1 (broadcast) and this lane's own unmasked entry of column j - 1
  double asum = row[0];                 // D_rj - sum_{t<j-1} L_rt L_jt of the pivot at hand, formed during the previous pivot's chain
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double nxt[NB];  // row j + 1 of L, entries t < j (final since pivot j - 1)
#pragma unroll
    for (int t = 0; t < NB; ++t) nxt[t] = (j + 1 < NB && t < j) ? D[(j + 1) & (NB - 1)][t] : 0.0;
    __builtin_amdgcn_sched_barrier(0);  // the reads go first
    const double d = readlane_f64(fma(-raw_prev, raw_prev, asum), j);        // lane j: its own L_j,j-1 twice
    const double acc = (j >= 1) ? fma(-row[j - 1], s_prev, asum) : asum;     // every lane (lane j: bit-identical to d)
    const double y0 = __builtin_amdgcn_rsq(d);
    const double u = acc * y0;
    const double e = fma(-(d * y0), y0, 1.0);
    const double raw = fma(u * e, fma(0.375, e, 0.5), u);
    // independent of the chain above and issued between its instructions: the next pivot's sum over the columns that are final
    double a[4] = {j + 1 < NB ? row[(j + 1) & (NB - 1)] : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < j; ++t) a[t & 3] -= row[t] * nxt[t];
    asum = (a[0] + a[1]) + (a[2] + a[3]);
    if (j + 1 < NB) s_prev = readlane_f64(raw, j + 1);
    raw_prev = raw;
    const double l = (!isX && r < j) ? 0.0 : raw;
    row[j] = l;
    D[lane][j] = l;
    if (j < nb && (!(d > 0.0) || !isfinite(d))) bad = 1;
    asm volatile("" : "+v"(bad));  // settled per pivot (left alone, the compiler keeps every pivot's comparison masks to the end and spills them)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (bad) {  // wave-uniform (d is)
    if (lane == 0) flags[2] = 1;
#pragma unroll
    for (int c = 0; c < NB; ++c) D[lane][c] = (c == r) ? 1.0 : 0.0;
  }
}

// every thread of the workgroup, behind a barrier: L (lower triangle, live rows) to the global block at `out` (row stride ldw) and
// X = L^-1 (NB x NB, row-major, identity-padded beyond nb) to `xinv`.  Three stores per thread instead of 32 per lane of the factoring wave.
// tdiag != nullptr: the diagonal block of T = L^-T (k_chol_apply) receives X^T as well.
__device__ __forceinline__ void chol_factor_store(double (*D)[NB + 1], int nb, double* __restrict__ out, int ldw, double* __restrict__ xinv,
                                                  int tid, int nthreads, double* __restrict__ tdiag = nullptr) {
  for (int e = tid; e < NB * NB; e += nthreads) {
    const int i = e / NB, c = e % NB;
    const double l = D[i][c], x = D[NB + c][i];
    if (i < nb && c <= i) out[(long)i * ldw + c] = l;
    xinv[e] = x;
    if (tdiag && i < nb && c < nb) tdiag[(long)c * ldw + i] = x;  // T_kk[c][i] = X[i][c]
  }
}

// S = U + lam D_c^2 + cam_diag - Sacc (symmetric, both triangles written), rhs = -g_c + b
// entry (row, col), col >= row, of S
template <int NC>
__device__ __forceinline__ double schur_entry(int row, int col, const double* __restrict__ Sacc, const double* __restrict__ Upacked,
                                              const double* __restrict__ sinv, const int* __restrict__ param_cam, const int* __restrict__ param_loc,
                                              int ncp, double lam, const double* __restrict__ cam_diag, const double* __restrict__ red, int g,
                                              long tile_elems, const int* __restrict__ group_cam_begin) {
  using UP = UPack<NC>;
  const bool same_cam = param_cam[row] == param_cam[col];
  double v;
  if (red && same_cam) {
    // red != nullptr (single rank, no heavy points, no constraint rows): the diagonal camera blocks are folded HERE from the helper-thread sums
    // k_reg_reduce parked in `red` (what k_reg_fold does — one launch less; Sacc's diagonal blocks are then never written nor read)
    const int cam = param_cam[row], ga = cam / g, ca0 = group_cam_begin[ga], na = group_cam_begin[ga + 1] - ca0;
    const double* src = red + (long)ga * tile_elems + param_loc[row] * NC + param_loc[col];
    double sum = 0.0;
    for (int k = cam - ca0; k < g * (g + 1) / 2; k += na) {  // helper k = (li, lj) with k = li (li + 1) / 2 + lj
      int li = (int)((sqrtf(8.0f * k + 1.0f) - 1.0f) * 0.5f);
      while (li * (li + 1) / 2 > k) --li;
      while ((li + 1) * (li + 2) / 2 <= k) ++li;
      const int lj = k - li * (li + 1) / 2;
      sum += src[(long)(li * g + lj) * (NC * NC)];
    }
    v = -sum;
  } else {
    v = -Sacc[(long)row * ncp + col];
  }
  if (same_cam) v += Upacked[param_cam[row] * UP::STRIDE + UP::idx(param_loc[row], param_loc[col])];
  if (row == col) v += lam * sinv[row] * sinv[row] + cam_diag[row];  // cam_diag: zero unless the caller set a bound scaling
  return v;
}
// (Round 6 measured ONE MORE workgroup here that formed the first diagonal block itself and factored it — step k = -1 of the dense solve without its
// launch: 31 us against 9.5 + 3.5, its 1024 entries being chains of dependent loads; removed again, profiles/r06_experiments.txt.)
template <int NC>
__global__ void __launch_bounds__(256)
k_schur_finalize(const double* __restrict__ Sacc, const double* __restrict__ bacc,
                 const double* __restrict__ Upacked, const double* __restrict__ gvec,
                 const double* __restrict__ sinv, const int* __restrict__ param_cam,
                 const int* __restrict__ param_loc, int ncp, double lam, const double* __restrict__ lam_dev,
                 const double* __restrict__ cam_diag, double* __restrict__ S, double* __restrict__ rhs,
                 double* __restrict__ W, int ldw, const double* __restrict__ red = nullptr, int g = 0, long tile_elems = 0,
                 const int* __restrict__ group_cam_begin = nullptr, int b_width = 0) {
  CBA_STAMP(ST_FINALIZE);
  if (lam_dev) lam = *lam_dev;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long)ncp * ncp) return;
  const int row = (int)(t / ncp), col = (int)(t % ncp);
  if (col < row) return;
  const double v = schur_entry<NC>(row, col, Sacc, Upacked, sinv, param_cam, param_loc, ncp, lam, cam_diag, red, g, tile_elems, group_cam_begin);
  if (row == col) {
    const double rv = -gvec[row] + b_entry(bacc, b_width, row);
    rhs[row] = rv;
    W[(long)ncp * ldw + row] = rv;  // rhs^T: last row of the Cholesky work matrix
  }
  S[(long)row * ncp + col] = v;
  S[(long)col * ncp + row] = v;
  W[(long)row * ldw + col] = v;
  W[(long)col * ldw + row] = v;
}

// Round 6: k_reg_reduce and k_schur_finalize as ONE launch for the route on which nothing else adds to the reduced system between them (single rank, no
// heavy points, no constraint rows; not the one-workgroup solve of small rigs): the sums over a tile's partial rows leave as entries of S and of the
// Cholesky work matrix at once — a launch boundary and a pass over Sacc less per iteration; Sacc and `red` are not written.  blockDim = (64, 16), a
// one-dimensional grid; roles in dispatch order, the ones with the longest chains first:
//   G x fold_x workgroups   the diagonal camera blocks of group ga: entry (r <= c) of camera cam = U - sum over the camera's helpers AND the diagonal tile's
//                           partial rows (+ lam D^2 + cam_diag on the diagonal), sixteen ways (k_reg_reduce parked the helper entries for
//                           k_schur_finalize's fold);
//   rhs_x workgroups        the right-hand side: -g + the rhs rows k_tprep left per workgroup, sixteen ways; also row n of the work matrix;
//   n_tiles x tile_x        a tile: `ysplit` (4 or 16, k_reg_reduce's choice) threads share an entry's partial rows, 16 / ysplit blocks of 64 entries per
//                           workgroup.  Entries of two different cameras are final: S_rc = S_cr = -sum.
// The pass reads the pair kernel's 38 MB of partial blocks (cfg4) once: ~9 us at the rate HBM delivers them is its floor.
// Fixed summation order in every role (not the order of the two-launch route: partial rows before helpers there, the rhs in eight slices).
template <int NC>
__global__ void __launch_bounds__(64 * REG_REDUCE_Y_MAX)
k_reg_finalize(TilePlan tp, const int* __restrict__ tile_wg_begin, const double* __restrict__ partial, const int* __restrict__ cam_off,
               const int* __restrict__ cam_np, int ncp, int n_tiles, int G, int ysplit, int fold_x, int rhs_x, int tile_x,
               const double* __restrict__ partial_b, int b_rows, int b_width,
               const double* __restrict__ Upacked, const double* __restrict__ gvec, const double* __restrict__ sinv, double lam,
               const double* __restrict__ lam_dev, const double* __restrict__ cam_diag, double* __restrict__ S, double* __restrict__ rhs,
               double* __restrict__ W, int ldw) {
  CBA_STAMP(ST_REG_REDUCE);
  using UP = UPack<NC>;
  constexpr int YM = REG_REDUCE_Y_MAX;
  __shared__ double sh[YM][64];
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int g = tp.g, bsz = NC * NC;
  auto put = [&](int row, int col, double v) {
    S[(long)row * ncp + col] = v; S[(long)col * ncp + row] = v;
    W[(long)row * ldw + col] = v; W[(long)col * ldw + row] = v;
  };
  const int bid = (int)blockIdx.x;
  if (bid >= G * fold_x && bid < G * fold_x + rhs_x) {  // right-hand side
    const int j = (bid - G * fold_x) * 64 + tx;
    sh[ty][tx] = (j < b_width) ? strided_sum(partial_b + j, b_width, 0, b_rows, ty, YM) : 0.0;
    __syncthreads();
    if (ty == 0 && j < ncp) {
      double tot = 0.0;
#pragma unroll
      for (int y = 0; y < YM; ++y) tot += sh[y][tx];
      const double rv = -gvec[j] + tot;
      rhs[j] = rv;
      W[(long)ncp * ldw + j] = rv;
    }
    return;
  }
  if (bid < G * fold_x) {  // diagonal camera blocks of one group
    if (lam_dev) lam = *lam_dev;
    const int ga = bid / fold_x, bx = bid % fold_x;
    const int t = ga * G - ga * (ga - 1) / 2;  // tiles are numbered (a, b >= a), a outermost: the diagonal tile of group ga
    const int ca0 = tp.group_cam_begin[ga], na = tp.group_cam_begin[ga + 1] - ca0;
    const int q = bx * 64 + tx;
    const int cl = q / bsz, r = (q % bsz) / NC, c = (q % bsz) % NC;
    const bool live = q < g * bsz && cl < na && c >= r && c < cam_np[ca0 + min(cl, max(na - 1, 0))];
    // the camera's helpers x the tile's partial rows as ONE list of items, item m = (helper m / R, row m % R), thread ty takes m = ty, ty + 16, ..:
    // eight loads in flight whatever the split between helpers and rows is (helper by helper the sums were 27-76 dependent steps per thread).  The
    // helpers' offsets come from a table the workgroup makes first (its 64 entries belong to at most three cameras); (helper, row) advance incrementally.
    constexpr int HMAX = kSchurRegMaxGroupK * (kSchurRegMaxGroupK + 1) / 2;
    __shared__ int sh_hoff[3][HMAX];
    const int cl_first = (bx * 64) / bsz;
    for (int e = ty * 64 + tx; e < 3 * HMAX; e += 64 * YM) {
      const int cam_l = cl_first + e / HMAX, h = e % HMAX, k = cam_l + h * max(na, 1);
      int off = 0;
      if (cam_l < na && k < g * (g + 1) / 2) {  // helper k = (li, lj), k = li (li + 1) / 2 + lj (schur_entry)
        int li = (int)((sqrtf(8.0f * k + 1.0f) - 1.0f) * 0.5f);
        while (li * (li + 1) / 2 > k) --li;
        while ((li + 1) * (li + 2) / 2 <= k) ++li;
        off = (li * g + (k - li * (li + 1) / 2)) * bsz;
      }
      sh_hoff[e / HMAX][h] = off;
    }
    __syncthreads();
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int w0 = tile_wg_begin[t] * tp.rep, R = tile_wg_begin[t + 1] * tp.rep - w0;
    if (live && R > 0) {
      const int H = (g * (g + 1) / 2 - cl + na - 1) / na;  // helpers k = cl, cl + na, .. of this camera
      const int* hoff = sh_hoff[cl - cl_first];
      const double* base = partial + (long)w0 * tp.tile_elems + r * NC + c;
      int h = ty / R, w = ty % R;  // item ty
      auto next = [&]() {  // the item's address; then on to item + 16
        const double* ptr = base + (long)w * tp.tile_elems + hoff[min(h, H - 1)];
        w += YM;
        while (w >= R) { w -= R; ++h; }
        return ptr;
      };
      const int total = H * R;
      int m = ty;
      for (; m + 7 * YM < total; m += 8 * YM) {
        const double* ptr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ptr[q] = next();
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += *ptr[q];
      }
      for (; m < total; m += YM) a[0] += *next();
    }
    sh[ty][tx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (ty != 0 || !live) return;
    double tot = 0.0;
#pragma unroll
    for (int y = 0; y < YM; ++y) tot += sh[y][tx];
    const int cam = ca0 + cl, row = cam_off[cam] + r, col = cam_off[cam] + c;
    double v = Upacked[cam * UP::STRIDE + UP::idx(r, c)] - tot;
    if (r == c) v += lam * sinv[row] * sinv[row] + cam_diag[row];
    put(row, col, v);
    return;
  }
  // a tile
  const int t = (bid - G * fold_x - rhs_x) / tile_x, bx = (bid - G * fold_x - rhs_x) % tile_x;
  const int ga = tp.tile_a[t], gb = tp.tile_b[t];
  const int ca0 = tp.group_cam_begin[ga], na = tp.group_cam_begin[ga + 1] - ca0;
  const int cb0 = tp.group_cam_begin[gb], nb = tp.group_cam_begin[gb + 1] - cb0;
  const int per = YM / ysplit, sub = ty / ysplit, yy = ty % ysplit;
  const int e = (bx * per + sub) * 64 + tx;
  const int w0 = tile_wg_begin[t] * tp.rep, w1 = tile_wg_begin[t + 1] * tp.rep;
  int row = -1, col = -1;
  if (e < g * g * bsz) {
    const int beta = e / bsz, rc = e % bsz;
    const int li = beta / g, lj = beta % g, r = rc / NC, c = rc % NC;
    if (!(ga == gb && lj <= li) && li < na && lj < nb && r < cam_np[ca0 + li] && c < cam_np[cb0 + lj]) {
      row = cam_off[ca0 + li] + r; col = cam_off[cb0 + lj] + c;
    }
  }
  sh[ty][tx] = (row >= 0) ? strided_sum(partial + e, tp.tile_elems, w0, w1, yy, ysplit) : 0.0;
  __syncthreads();
  if (yy != 0 || row < 0) return;
  double tot = 0.0;
  for (int y = 0; y < ysplit; y += 4) tot += (sh[ty + y][tx] + sh[ty + y + 1][tx]) + (sh[ty + y + 2][tx] + sh[ty + y + 3][tx]);
  put(row, col, -tot);
}

// One launch per panel (right-looking with look-ahead).  Work matrix W: (n + 1) rows, row stride ldw (multiple of
// 4), rows 0..n-1 = S (both triangles on entry), row n = rhs^T.  Row blocks of NB rows; the rhs row is a block of
// its own.  Step k >= 0 (panel columns k0 = NB k .., L_kk already factored by step k - 1) has two kinds of workgroup:
//
//   panel workgroup, row block b = k + 1 .. nbk:
//     1. U    = W_bk - L_b,k-1 L_k,k-1^T      the one update its panel columns still miss (rank NB, FP64 MFMA);
//     2. L_bk = U L_kk^-T                     panel solve, one thread per row;
//     3. D_b -= L_bk L_bk^T                   its own diagonal block, kept up to date every step;
//     4. b == k + 1: factor D_b (wave 0), so that the next step finds L_k+1,k+1 ready (look-ahead);
//   trailing workgroup, block (i, j) with k + 1 <= j < i:
//        W_ij -= L_i,k-1 L_j,k-1^T            the update of the previous panel, off the critical path.
//
// Step k = -1 is one workgroup that factors D_0.  The critical chain per panel is one launch with one global round
// trip of three 8 KB blocks.  History: (a) panel-solve + trailing-update kernels, 2 launches and ~28 us per panel;
// (b) one launch with a left-looking update of depth k0 - its operands (up to 160 KB) all pass through the one CU
// that runs the critical workgroup, 6-10 us per step at the memory-level parallelism of a single CU.
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int CHOL_THREADS = 512;   // 8 waves; 1024 threads measured slower (register budget halves, the panel solve spills)

// red[tile] = A B^T for two NB x NB panels (depth NB), four 16 x 16 tiles, one per wave 0..3 (v_mfma_f64_16x16x4:
// eight per tile).  A, B point at element (first row, first panel column) of the operands in the work matrix; rows
// are clamped to the live ones (callers ignore the padding rows).  A lane loads 4 consecutive doubles of row
// (lane & 15) of its tile per 16-deep slab; the K index of an MFMA is lane >> 4, and step t pairs the t-th of each
// lane's 4 doubles: the same permutation of K on both operands, so the sum is unchanged.
__device__ __forceinline__ void chol_rank_nb(const double* __restrict__ A, int rcA, const double* __restrict__ B, int rcB,
                                             int ldw, double (*red)[16][17], int wv, int lane) {
  if (wv >= 4) return;
  const int ti = wv >> 1, tj = wv & 1, lr = lane & 15, kq = 4 * (lane >> 4);
  const double* pa = A + (long)min(ti * 16 + lr, rcA - 1) * ldw + kq;
  const double* pb = B + (long)min(tj * 16 + lr, rcB - 1) * ldw + kq;
  double2 a[NB / 16][2], b[NB / 16][2];
#pragma unroll
  for (int s = 0; s < NB / 16; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a[s][h] = *reinterpret_cast<const double2*>(pa + 16 * s + 2 * h);
      b[s][h] = *reinterpret_cast<const double2*>(pb + 16 * s + 2 * h);
    }
  v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < NB / 16; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][h].x, b[s][h].x, c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][h].y, b[s][h].y, c, 0, 0, 0);
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][(lane >> 4) + 4 * r][lr] = c[r];
}

__global__ void __launch_bounds__(CHOL_THREADS)
k_chol_step(double* __restrict__ W, int n, int ldw, int k, int* __restrict__ flags, long long* __restrict__ trace, double* __restrict__ Xinv,
            double* __restrict__ Tinv = nullptr) {
  CBA_STAMP(ST_CHOL_STEP + k + 1);
  __shared__ double sh_red[4][16][17];
  __shared__ double sh_U[NB][NB + 1], sh_X[NB][NB + 1], sh_D[2 * NB][NB + 1];
  __shared__ __attribute__((aligned(16))) double sh_L[NB][NB + 2];  // even row stride: pairs of coefficients are 16-byte aligned
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int nbk = (n + NB - 1) / NB;
  constexpr int EPT = NB * NB / CHOL_THREADS, ISTEP = CHOL_THREADS / NB;  // elements of a 32 x 32 block per thread
  const int j = tid & 31, i0 = tid >> 5;                  // (i0 + h * ISTEP, j), h < EPT
  const int n_panel = (k < 0) ? 1 : nbk - k;
  const int n_trailing = (k < 1) ? 0 : (nbk - k - 1) * (nbk - k) / 2;

  if ((int)blockIdx.x >= n_panel + n_trailing) {
    // inverse role (Tinv != nullptr, k >= 1): T = L^-T is built next to the factorisation, off its critical path, so that the backward substitution
    // — a serial chain of one workgroup, 2.3 us per block — becomes one matrix-vector product (k_chol_apply).  With M = L^-1:
    //     M_ij = -X_i sum_{m=j}^{i-1} L_im M_mj   (i > j),   M_jj = X_j;    T_ji = M_ij^T is what is stored (upper block triangle).
    // The sum is accumulated right-looking: launch k adds the term m = k - 1 to every block (i >= k, j < k) — L_i,k-1 comes from the panel solves
    // of launch k - 1, T_j,k-1 was finalised there — and finalises row i = k with X_k (factored by launch k - 1's look-ahead):
    //     acc^T_ji += T_j,k-1 L_i,k-1^T ;      i == k:  T_jk = -acc^T_jk X_k^T.
    const int t2 = blockIdx.x - n_panel - n_trailing;
    const int bi = k + t2 / k, bj = t2 % k, m = k - 1;
    const int ri = bi * NB, rci = min(NB, n - ri), rj = bj * NB, rcj = min(NB, n - rj);
    double* Tb = Tinv + (long)rj * ldw + ri;  // block (bj, bi)
    double old[EPT];
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      old[h] = (m != bj && i < rcj && j < rci) ? Tb[(long)i * ldw + j] : 0.0;  // m == bj: the first term
    }
    chol_rank_nb(Tinv + (long)rj * ldw + m * NB, rcj, W + (long)ri * ldw + m * NB, rci, ldw, sh_red, wv, lane);
    __syncthreads();
    if (bi != k) {
#pragma unroll
      for (int h = 0; h < EPT; ++h) {
        const int i = i0 + h * ISTEP;
        if (i < rcj && j < rci) Tb[(long)i * ldw + j] = old[h] + sh_red[(i >> 4) * 2 + (j >> 4)][i & 15][j & 15];
      }
      return;
    }
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      sh_U[i][j] = (i < rcj && j < rci) ? old[h] + sh_red[(i >> 4) * 2 + (j >> 4)][i & 15][j & 15] : 0.0;
      sh_L[i][j] = Xinv[(long)k * NB * NB + i * NB + j];  // X_k, identity-padded
    }
    __syncthreads();
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
      v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < NB / 4; ++t) {
        const int q = 4 * t + (lane >> 4);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(sh_U[ti * 16 + (lane & 15)][q], sh_L[tj * 16 + (lane & 15)][q], c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 16 + (lane >> 4) + 4 * r, jj = tj * 16 + (lane & 15);
        if (i < rcj && jj < rci) Tb[(long)i * ldw + jj] = -c[r];
      }
    }
    return;
  }

  if ((int)blockIdx.x >= n_panel) {
    // trailing role: block (bi, bj), k + 1 <= bj < bi <= nbk, takes the update of panel k - 1
    int t = blockIdx.x - n_panel, bj = k + 1;
    while (t >= nbk - bj) { t -= nbk - bj; ++bj; }
    const int bi = bj + 1 + t;
    const int ri = (bi < nbk) ? bi * NB : n, rci = (bi < nbk) ? min(NB, n - ri) : 1;
    const int rj = bj * NB, rcj = min(NB, n - rj);
    const int m0 = (k - 1) * NB;
    double* Wi = W + (long)ri * ldw;
    double w_ij[EPT];
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      w_ij[h] = (i < rci && j < rcj) ? Wi[(long)i * ldw + rj + j] : 0.0;
    }
    chol_rank_nb(Wi + m0, rci, W + (long)rj * ldw + m0, rcj, ldw, sh_red, wv, lane);
    __syncthreads();
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      if (i < rci && j < rcj) Wi[(long)i * ldw + rj + j] = w_ij[h] - sh_red[(i >> 4) * 2 + (j >> 4)][i & 15][j & 15];
    }
    return;
  }

  const int b = k + 1 + blockIdx.x;                       // row block; nbk = the rhs row
  const int rb = (b < nbk) ? b * NB : n;                  // first row
  const int rc = (b < nbk) ? min(NB, n - rb) : 1;         // live rows
  const bool has_diag = b < nbk;
  double* Wb = W + (long)rb * ldw;
  // optional phase stamps of the critical workgroup (b == k + 1), 100 MHz wall clock: CBA_CHOL_TRACE=1
  const bool stamp = trace != nullptr && blockIdx.x == 0 && tid == 0;
#define CHOL_STAMP(ph) do { if (stamp) trace[(k + 1) * 8 + (ph)] = wall_clock64(); } while (0)
  CHOL_STAMP(0);

  // everything this thread needs from global, in one round trip
  double d_ij[EPT], m_ij[EPT], l_ij[EPT];
#pragma unroll
  for (int h = 0; h < EPT; ++h) {
    const int i = i0 + h * ISTEP;
    d_ij[h] = (has_diag && i < rc && j < rc) ? Wb[(long)i * ldw + rb + j] : (i == j ? 1.0 : 0.0);
  }
  // Step k = -1 (one workgroup: D_0 as loaded) and the look-ahead of every later step run the SAME factorisation code below: two inlined copies were
  // two cold instruction streams per iteration (the first pivot chain through a copy took 8.6 us instead of 5.6)
  if (k < 0) {
#pragma unroll
    for (int h = 0; h < EPT; ++h) sh_D[i0 + h * ISTEP][j] = d_ij[h];
  } else {
    const int k0 = k * NB, nbp = min(NB, n - k0);
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      m_ij[h] = (i < rc && j < nbp) ? Wb[(long)i * ldw + k0 + j] : 0.0;
      l_ij[h] = Xinv[(long)k * NB * NB + i * NB + j];  // X_k = L_kk^-1 (identity-padded), written by the factorisation of D_k
    }

    CHOL_STAMP(1);
    // 1. pending update of this block's panel columns by panel k - 1 (all earlier panels were applied by the trailing
    // workgroups of earlier steps)
    if (k >= 1) chol_rank_nb(Wb + (k0 - NB), rc, W + (long)k0 * ldw + (k0 - NB), nbp, ldw, sh_red, wv, lane);
    __syncthreads();
    CHOL_STAMP(2);
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      const double upd = (k >= 1) ? sh_red[(i >> 4) * 2 + (j >> 4)][i & 15][j & 15] : 0.0;
      sh_U[i][j] = m_ij[h] - upd;
      sh_L[i][j] = l_ij[h];
    }
    __syncthreads();
    CHOL_STAMP(3);
    // 2. panel solve  L_bk = U L_kk^-T = U X_k^T: a 32 x 32 x 32 product on the matrix cores (four 16 x 16 tiles, one per
    // wave 0..3).  History: a triangular solve by wave 0, one row of U per thread, was 2.8 us of the 13 us of a step.
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
      v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < NB / 4; ++t) {
        const int q = 4 * t + (lane >> 4);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(sh_U[ti * 16 + (lane & 15)][q], sh_L[tj * 16 + (lane & 15)][q], c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 16 + (lane >> 4) + 4 * r, jj = tj * 16 + (lane & 15);
        sh_X[i][jj] = (jj < nbp && i < rc) ? c[r] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      if (i < rc && j < nbp) Wb[(long)i * ldw + k0 + j] = sh_X[i][j];
    }
    CHOL_STAMP(4);
    if (!has_diag) return;
    // 3. own diagonal block, rank-NB update
    if (wv < 4) {
      const int ti = wv >> 1, tj = wv & 1;
      v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int t = 0; t < NB / 4; ++t) {
        const int q = 4 * t + (lane >> 4);
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(sh_X[ti * 16 + (lane & 15)][q], sh_X[tj * 16 + (lane & 15)][q], c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sh_red[wv][(lane >> 4) + 4 * r][lane & 15] = c[r];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < EPT; ++h) {
      const int i = i0 + h * ISTEP;
      const double d_new = d_ij[h] - sh_red[(i >> 4) * 2 + (j >> 4)][i & 15][j & 15];
      if (b != k + 1) {
        if (i < rc && j < rc) Wb[(long)i * ldw + rb + j] = d_new;
      } else {
        sh_D[i][j] = (i < rc && j < rc) ? d_new : (i == j ? 1.0 : 0.0);
      }
    }
    if (b != k + 1) return;
  }
  // 4. look-ahead: the next panel's diagonal block
  __syncthreads();
  CHOL_STAMP(5);
  if (tid < WAVE) chol_factor_block(sh_D, rc, flags);
  __syncthreads();
  chol_factor_store(sh_D, rc, Wb + rb, ldw, Xinv + (long)(k + 1) * NB * NB, tid, CHOL_THREADS, Tinv ? Tinv + (long)rb * ldw + rb : nullptr);
  CHOL_STAMP(6);
#undef CHOL_STAMP
}

// Small rigs (ncp <= SMALL_N: sixteen six-parameter cameras — every rig the reference's users have): the whole dense solve behind the pair
// kernel in ONE workgroup with the matrix in LDS: S = U + lam D_c^2 + cam_diag - Sacc and the right-hand side formed as k_schur_finalize does,
// blocked Cholesky (the 32-pivot wave factorisation of chol_factor_block per diagonal block, panel solves with its inverse, trailing updates; the
// rhs rides along as row n), backward substitution with the blocks' inverses, x to `out`.  Replaces k_schur_finalize + ncp / 32 + 1 k_chol_step +
// k_chol_apply: five launches of 4-9 us each on cfg2, where every launch — however small — occupies the device for ~4.5 us.
constexpr int SMALL_N = 96;
constexpr int SMALL_THREADS = 512;
constexpr int SMALL_LD = SMALL_N + 1;  // odd row stride
template <int NC>
__global__ void __launch_bounds__(SMALL_THREADS)
k_small_solve(const double* __restrict__ Sacc, const double* __restrict__ bacc, const double* __restrict__ Upacked, const double* __restrict__ gvec,
              const double* __restrict__ sinv, const int* __restrict__ param_cam, const int* __restrict__ param_loc, int n, double lam,
              const double* __restrict__ lam_dev, const double* __restrict__ cam_diag, double* __restrict__ S, double* __restrict__ rhs,
              const double* __restrict__ red, int g, long tile_elems, const int* __restrict__ group_cam_begin, int* __restrict__ flags,
              double* __restrict__ out, int b_width) {
  CBA_STAMP(ST_SMALL_SOLVE);
  using UP = UPack<NC>;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double (*W)[SMALL_LD] = reinterpret_cast<double (*)[SMALL_LD]>(sh);                       // [n + 1][SMALL_LD]: S, then rhs^T
  double (*D)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(sh + (SMALL_N + 1) * SMALL_LD);  // [2 NB][NB + 1]: chol_factor_block's block
  double (*X)[NB][NB + 1] = reinterpret_cast<double (*)[NB][NB + 1]>(&D[2 * NB][0]);          // [nbk][NB][NB + 1]: inverses of the diagonal blocks
  double* y = &X[(SMALL_N + NB - 1) / NB][0][0];                                              // [SMALL_N]
  if (lam_dev) lam = *lam_dev;
  const int tid = threadIdx.x, nbk = (n + NB - 1) / NB;
  // 1. the reduced system (k_schur_finalize's arithmetic).  A thread handles several entries one after the other, so nothing a load's address
  // depends on may come from global memory inside the loops (a chain of three dependent loads per entry was 25 of this kernel's first 42 us): the
  // per-parameter tables go to LDS first, the off-block entries are four independent loads per trip, the few entries inside a camera's own block
  // (they carry U_c, the damping and the folded helper sums) have a loop of their own.
  int* pc = reinterpret_cast<int*>(y + SMALL_N);  // [n] camera of a parameter
  int* pl = pc + SMALL_N;                         // [n] its index inside the camera's block
  for (int i = tid; i < n; i += SMALL_THREADS) {
    pc[i] = param_cam[i]; pl[i] = param_loc[i];
    const double rv = -gvec[i] + b_entry(bacc, b_width, i);
    rhs[i] = rv;
    W[n][i] = rv;
  }
  __syncthreads();
  for (int base = tid; base < n * n; base += 4 * SMALL_THREADS) {
    double v4[4];
    int r4[4], c4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = base + u * SMALL_THREADS;
      const int row = min(t, n * n - 1) / n, col = min(t, n * n - 1) % n;
      const bool live = t < n * n && col > row && pc[row] != pc[col];
      r4[u] = live ? row : -1; c4[u] = col;
      v4[u] = -Sacc[(long)row * n + col];  // (unconditional, clamped: the four loads are in flight together)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r4[u] < 0) continue;
      S[(long)r4[u] * n + c4[u]] = v4[u]; S[(long)c4[u] * n + r4[u]] = v4[u];
      W[r4[u]][c4[u]] = v4[u]; W[c4[u]][r4[u]] = v4[u];
    }
  }
  for (int q = tid; q < n * NC; q += SMALL_THREADS) {
    const int row = q / NC, col = row - pl[row] + q % NC;
    if (col < row || col >= n || pc[col] != pc[row]) continue;
    const int cam = pc[row];
    double v;
    if (red) {
      const int ga = cam / g, ca0 = group_cam_begin[ga], na = group_cam_begin[ga + 1] - ca0;
      const double* src = red + (long)ga * tile_elems + pl[row] * NC + pl[col];
      double sum = 0.0;
      for (int k = cam - ca0; k < g * (g + 1) / 2; k += na) {  // helper k = (li, lj) with k = li (li + 1) / 2 + lj
        int li = (int)((sqrtf(8.0f * k + 1.0f) - 1.0f) * 0.5f);
        while (li * (li + 1) / 2 > k) --li;
        while ((li + 1) * (li + 2) / 2 <= k) ++li;
        const int lj = k - li * (li + 1) / 2;
        sum += src[(long)(li * g + lj) * (NC * NC)];
      }
      v = -sum;
    } else {
      v = -Sacc[(long)row * n + col];
    }
    v += Upacked[cam * UP::STRIDE + UP::idx(pl[row], pl[col])];
    if (row == col) v += lam * sinv[row] * sinv[row] + cam_diag[row];
    S[(long)row * n + col] = v; S[(long)col * n + row] = v;
    W[row][col] = v; W[col][row] = v;
  }
  __syncthreads();
  // 2. blocked Cholesky in place (lower triangle), the rhs row as one more row of every panel
  for (int k = 0; k < nbk; ++k) {
    const int k0 = k * NB, nb = min(NB, n - k0);
    for (int e = tid; e < NB * NB; e += SMALL_THREADS) {  // diagonal block, identity-padded
      const int i = e / NB, j = e % NB;
      D[i][j] = (i < nb && j < nb) ? W[k0 + max(i, j)][k0 + min(i, j)] : (i == j ? 1.0 : 0.0);  // (the trailing updates keep the lower triangle)
    }
    __syncthreads();
    if (tid < WAVE) chol_factor_block(D, nb, flags);
    __syncthreads();
    for (int e = tid; e < NB * NB; e += SMALL_THREADS) {
      const int i = e / NB, j = e % NB;
      if (i < nb && j <= i) W[k0 + i][k0 + j] = D[i][j];
      X[k][i][j] = D[NB + j][i];  // X_k = L_kk^-1, row-major (row NB + c of D holds column c)
    }
    __syncthreads();
    // panel: L_bk = W_bk X_k^T for the rows below the block (and the rhs row): row r, column j <- sum_t W[r][k0 + t] X_k[j][t]
    const int r_lo = k0 + nb, n_rows = n + 1 - r_lo;
    double acc4[(SMALL_N + 1) * NB / SMALL_THREADS + 1];
    {
      int q = 0;
      for (int e = tid; e < n_rows * NB; e += SMALL_THREADS, ++q) {
        const int r = r_lo + e / NB, j = e % NB;
        double a = 0.0;
        if (j < nb)
          for (int t = 0; t <= j; ++t) a = fma(W[r][k0 + t], X[k][j][t], a);  // (X_k = L_kk^-1 is lower triangular)
        acc4[q] = a;
      }
    }
    __syncthreads();
    {
      int q = 0;
      for (int e = tid; e < n_rows * NB; e += SMALL_THREADS, ++q) {
        const int r = r_lo + e / NB, j = e % NB;
        if (j < nb) W[r][k0 + j] = acc4[q];
      }
    }
    __syncthreads();
    // trailing update: W[i][j] -= sum_t L[i][k0 + t] L[j][k0 + t] for r_lo <= j <= i (i up to the rhs row n)
    const int m = n - r_lo;  // columns left
    for (int e = tid; e < (m + 1) * m; e += SMALL_THREADS) {
      const int i = r_lo + e / m, j = r_lo + e % m;
      if (j > i) continue;
      double a = W[i][j];
      for (int t = 0; t < nb; ++t) a = fma(-W[i][k0 + t], W[j][k0 + t], a);
      W[i][j] = a;
    }
    __syncthreads();
  }
  // 3. backward substitution L^T x = y (y^T = row n) with the blocks' inverses
  for (int i = tid; i < n; i += SMALL_THREADS) y[i] = W[n][i];
  __syncthreads();
  for (int k = nbk - 1; k >= 0; --k) {
    const int k0 = k * NB, nb = min(NB, n - k0);
    double xv = 0.0;
    if (tid < nb) {  // x_k = X_k^T y_k
      for (int t = tid; t < nb; ++t) xv = fma(X[k][t][tid], y[k0 + t], xv);
    }
    __syncthreads();
    if (tid < nb) { y[k0 + tid] = xv; out[k0 + tid] = xv; }
    __syncthreads();
    for (int i = tid; i < k0; i += SMALL_THREADS) {  // y_i -= sum_t L[k0 + t][i] x_t
      double a = y[i];
      for (int t = 0; t < nb; ++t) a = fma(-W[k0 + t][i], y[k0 + t], a);
      y[i] = a;
    }
    __syncthreads();
  }
}

// x = L^-T y = T y with the explicit T of the inverse role above: one workgroup per block row, 16 threads per row (coalesced 128-byte reads),
// fixed summation order.  Replaces the serial chain of a backward substitution (33 us at n = 384, 139 us at n = 1152 in round 2) by a launch of a few microseconds.
constexpr int APPLY_THREADS = 512;
__global__ void __launch_bounds__(APPLY_THREADS)
k_chol_apply(const double* __restrict__ Tinv, const double* __restrict__ W, int n, int ldw, double* __restrict__ out) {
  CBA_STAMP(ST_CHOL_APPLY);
  extern __shared__ __attribute__((aligned(16))) double y[];  // n
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += APPLY_THREADS) y[i] = W[(long)n * ldw + i];
  __syncthreads();
  const int r = tid >> 4, cl = tid & 15, row = blockIdx.x * NB + r;
  double a0 = 0.0, a1 = 0.0;
  if (row < n) {
    const double* Tr = Tinv + (long)row * ldw;
    int c = blockIdx.x * NB + cl;
    for (; c + 16 < n; c += 32) { a0 = fma(Tr[c], y[c], a0); a1 = fma(Tr[c + 16], y[c + 16], a1); }
    if (c < n) a0 = fma(Tr[c], y[c], a0);
  }
  double s = a0 + a1;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off, 16);
  if (cl == 0 && row < n) out[row] = s;
}

// ------------------------------------------------------------------------------------------------
// back-substitution:  dp = -V'^-1 (g_p + sum_i W_i^T dc_{c_i}),  W_i^T dc = B_i^T (A_i dc)
// SCAL (single-rank fused iteration): the point block's share of the step scalars — sum (s sinv)^2 and sum g s, what k_step_scalars would
// read s, g and sinv a second time for — is accumulated by the threads that solve the points and leaves as one partial row per workgroup
// (step_partial[b][0..3]); k_step_cam adds the camera block.
template <int NC, bool CAMG = false, bool SCAL = false>
__global__ void __launch_bounds__(BLOCK)
k_backsub(const double* __restrict__ obs_u, const double* __restrict__ obs_v, const int* __restrict__ obs_cam,
          const int* __restrict__ obs_pt, const int* __restrict__ pt_start, const int* __restrict__ chunk_start,
          const int* __restrict__ chunk_pts, int n_chunks, const double* __restrict__ xvec, VecLayout lay,
          const double* __restrict__ tab, const int* __restrict__ cam_off, int n_cams, int loss, double f_scale, double lam,
          const double* __restrict__ lam_dev, const double* __restrict__ Vblk, const double* __restrict__ gvec,
          const double* __restrict__ sinv, double* __restrict__ svec, double* __restrict__ step_partial = nullptr) {
  CBA_STAMP(ST_BACKSUB);
  extern __shared__ __attribute__((aligned(16))) double sh[];
  if (lam_dev) lam = *lam_dev;
  double* sh_tab = sh;
  double* sh_dc = sh_tab + (CAMG ? 0 : n_cams * CAMTAB_LDS);  // ncp_pad
  double* sh_pt = sh_dc + lay.ncp_pad;               // [3][CHUNK]
  if (!CAMG) stage_camtab(sh_tab, tab, n_cams);
  for (int i = threadIdx.x; i < lay.ncp_pad; i += BLOCK) sh_dc[i] = svec[i];
  __syncthreads();
  // (round 6) factored form, as in k_jv: A dc = G (w x Y + dt) + A_intr di with w = J_l dc_r, formed once per camera in the place of dc_r
  for (int c = threadIdx.x; c < n_cams; c += BLOCK) {
    const double* row = tab + (long)c * CAMTAB_DOUBLES;
    double* d3 = sh_dc + (int)row[35];
    const double r0 = d3[0], r1 = d3[1], r2 = d3[2];
    d3[0] = row[12] * r0 + row[13] * r1 + row[14] * r2;
    d3[1] = row[15] * r0 + row[16] * r1 + row[17] * r2;
    d3[2] = row[18] * r0 + row[19] * r1 + row[20] * r2;
  }
  __syncthreads();
  const double* px = xvec + lay.ncp_pad;
  const double* gp = gvec + lay.ncp_pad;
  const double* dp = sinv + lay.ncp_pad;
  double* sp = svec + lay.ncp_pad;
  // per-point solve of point p given the summed observation terms q (on top of g_p)
  double sc0 = 0.0, sc1 = 0.0;  // SCAL: sum (s sinv)^2, sum g s over the points this thread solves
  auto solve_point = [&](int p, double* q, const double* Vin, const double* din, const double* gin) {
    double Vd[6], L[6], y[3], x[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Vd[k] = Vin[k];
    Vd[0] += lam * din[0] * din[0]; Vd[3] += lam * din[1] * din[1]; Vd[5] += lam * din[2] * din[2];
    if (chol3(Vd, L)) {
      chol3_fwd(L, q, y);
      chol3_bwd(L, y, x);
    } else {
      x[0] = x[1] = x[2] = 0.0;  // flagged by the Schur pass already
    }
    sp[p] = -x[0];
    sp[lay.Ppad + p] = -x[1];
    sp[2 * lay.Ppad + p] = -x[2];
    if (SCAL) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double ps = x[k] * din[k]; sc0 = fma(ps, ps, sc0); sc1 = fma(-gin[k], x[k], sc1); }
    }
  };
  const int last_obs = max(chunk_start[n_chunks] - 1, 0);
  int ch = blockIdx.x;
  int o0 = 0, o1 = 0;
  ObsRec cur = {0.0, 0.0, 0, 0};
  // (round 6) the point of the NEXT chunk's observation travels one chunk ahead as well, keyed by a point index loaded two chunks ahead: the gather
  // used to be issued and waited for at the head of every chunk, behind the record it depends on
  int pt_next = 0;
  double X = 0.0, Yw = 0.0, Zw = 0.0;
  if (ch < n_chunks) {
    o0 = chunk_start[ch]; o1 = chunk_start[ch + 1];
    cur = load_obs(obs_u, obs_v, obs_cam, obs_pt, min(o0 + (int)threadIdx.x, last_obs));
    pt_next = obs_pt[min(chunk_start[min(ch + (int)gridDim.x, n_chunks - 1)] + (int)threadIdx.x, last_obs)];
    X = px[cur.pt]; Yw = px[lay.Ppad + cur.pt]; Zw = px[2 * lay.Ppad + cur.pt];
  }
  while (ch < n_chunks) {
    const int nxt = ch + gridDim.x, nc = min(nxt, n_chunks - 1);
    const int no0 = chunk_start[nc], no1 = chunk_start[nc + 1];
    ObsRec nx;
    {
      const int k1 = min(no0 + (int)threadIdx.x, last_obs);
      nx.u = obs_u[k1]; nx.v = obs_v[k1]; nx.cam = obs_cam[k1]; nx.pt = pt_next;
    }
    const double Xn = px[pt_next], Yn = px[lay.Ppad + pt_next], Zn = px[2 * lay.Ppad + pt_next];
    const int pt_nn = obs_pt[min(chunk_start[min(nxt + (int)gridDim.x, n_chunks - 1)] + (int)threadIdx.x, last_obs)];
    const int cp0 = chunk_pts[2 * ch], npts = chunk_pts[2 * ch + 1];
    const int i = o0 + threadIdx.x;
    // Inputs of the per-point phase, fetched now and consumed after the barrier: wave 0 takes points cp0 .. cp0 + 63
    // (a chunk of 10-observation points has ~25), unconditionally with clamped indices inside a wave-uniform branch.
    // Fetched after the barrier, their latency - three dependent global loads - was serial time with one wave
    // working and three parked: 45 of the kernel's 96 us.
    const int pp = min(cp0 + (int)threadIdx.x, lay.P - 1);
    int pa = 0, pb = 0;
    double Vp[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, dpp[3] = {0.0, 0.0, 0.0}, q[3] = {0.0, 0.0, 0.0}, gq[3] = {0.0, 0.0, 0.0};
    if (threadIdx.x < WAVE) {
      pa = pt_start[pp] - o0; pb = pt_start[pp + 1] - o0;
#pragma unroll
      for (int k = 0; k < 6; ++k) Vp[k] = Vblk[(long)k * lay.Ppad + pp];
#pragma unroll
      for (int k = 0; k < 3; ++k) { dpp[k] = dp[(long)k * lay.Ppad + pp]; q[k] = gp[(long)k * lay.Ppad + pp]; gq[k] = q[k]; }
    }
    double t[3] = {0.0, 0.0, 0.0};
    if (i < o1) {
      const int cam = cur.cam;
      double e[2], G[2][3], Yr[3], Aint[2][3], B[2][3];
      const CamTab& ctb = cam_of<CAMG>(sh_tab, tab, cam);
      obs_factors(ctb, X, Yw, Zw, cur.u, cur.v, loss, f_scale, e, G, Yr, Aint, B);
      const double* dc = sh_dc + (int)ctb.pad[0];  // (= cam_off[cam]): w (3), dt (3), di (3)
      const double m0 = fma(dc[1], Yr[2], fma(-dc[2], Yr[1], dc[3]));
      const double m1 = fma(dc[2], Yr[0], fma(-dc[0], Yr[2], dc[4]));
      const double m2 = fma(dc[0], Yr[1], fma(-dc[1], Yr[0], dc[5]));
      double a0 = fma(G[0][2], m2, fma(G[0][1], m1, G[0][0] * m0)), a1 = fma(G[1][2], m2, fma(G[1][1], m1, G[1][0] * m0));
      if (NC == 9 && ctb.nparams == 9.0) {
        a0 = fma(Aint[0][2], dc[8], fma(Aint[0][1], dc[7], fma(Aint[0][0], dc[6], a0)));
        a1 = fma(Aint[1][2], dc[8], fma(Aint[1][1], dc[7], fma(Aint[1][0], dc[6], a1)));
      }
      t[0] = B[0][0] * a0 + B[1][0] * a1;
      t[1] = B[0][1] * a0 + B[1][1] * a1;
      t[2] = B[0][2] * a0 + B[1][2] * a1;
    }
    sh_pt[threadIdx.x] = t[0];
    sh_pt[CHUNK + threadIdx.x] = t[1];
    sh_pt[2 * CHUNK + threadIdx.x] = t[2];
    __syncthreads();
    if (npts < 0 && threadIdx.x < 3) {  // fragment of a heavy point: partial sum of W^T dc into svec (zeroed before), k_heavy_finish solves
      double acc = 0.0;
      for (int j = 0; j < o1 - o0; ++j) acc += sh_pt[threadIdx.x * CHUNK + j];
      unsafeAtomicAdd(&sp[(long)threadIdx.x * lay.Ppad + cp0], acc);
    }
    if (threadIdx.x < WAVE && (int)threadIdx.x < npts && pb > pa) {
      for (int j = pa; j < pb; ++j) { q[0] += sh_pt[j]; q[1] += sh_pt[CHUNK + j]; q[2] += sh_pt[2 * CHUNK + j]; }
      solve_point(pp, q, Vp, dpp, gq);
    }
    for (int lp = (threadIdx.x < WAVE ? threadIdx.x + BLOCK : threadIdx.x); lp < npts; lp += BLOCK) {  // points beyond the first 64
      const int p = cp0 + lp;
      const int a = pt_start[p] - o0, b = pt_start[p + 1] - o0;
      if (b > a) {
        double q2[3] = {gp[p], gp[lay.Ppad + p], gp[2 * lay.Ppad + p]}, V2[6], d2[3];
        const double g2[3] = {q2[0], q2[1], q2[2]};
        for (int j = a; j < b; ++j) { q2[0] += sh_pt[j]; q2[1] += sh_pt[CHUNK + j]; q2[2] += sh_pt[2 * CHUNK + j]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) V2[k] = Vblk[(long)k * lay.Ppad + p];
#pragma unroll
        for (int k = 0; k < 3; ++k) d2[k] = dp[(long)k * lay.Ppad + p];
        solve_point(p, q2, V2, d2, g2);
      }
    }
    __syncthreads();
    cur = nx; o0 = no0; o1 = no1; ch = nxt;
    pt_next = pt_nn; X = Xn; Yw = Yn; Zw = Zn;
  }
  if (SCAL) {  // (sh_pt is free: the loop ended behind a barrier)
    double r;
    r = block_sum(sc0, sh_pt); if (threadIdx.x == 0) step_partial[blockIdx.x * 4 + 0] = r;
    r = block_sum(sc1, sh_pt); if (threadIdx.x == 0) { step_partial[blockIdx.x * 4 + 1] = r; step_partial[blockIdx.x * 4 + 2] = 0.0; step_partial[blockIdx.x * 4 + 3] = 0.0; }
  }
}

// ------------------------------------------------------------------------------------------------
// Back-substitution from the T records (round 6).  k_tprep has just written T_i = A_i^T B_i L^-T per observation, in the very order this pass
// walks, so nothing is linearised again (reference arithmetic replaced: core/reprojection.py:186-187, the point block by the chain rule):
//     dp = -L^-T ( L^-1 g_p + sum_i T_i^T dc_{c_i} ),      T_i^T dc = Q_i^T (w_c x Y_i + dt_c) + T_intr,i^T di_c,     w_c = J_l,c dc_r
// (Y = R X, Q = rows 3..5 of T, T_intr = rows 6..8: the compact record of SchurRec; dc = (dc_r, dt, di) the camera's step).  Per observation
// 18 (27) FP64 instructions on a 96 (176) byte record instead of obs_linearize's ~250, no camera table, no point gather.
//   * per camera (w, dt, di) once per workgroup into LDS (6 / 9 doubles per camera);
//   * the 64 records of a wave are one contiguous run of Trec: the wave fetches them HBM -> LDS by LDS-DMA, 1 KB per instruction, into its own
//     quarter of ONE buffer — a lane copies its record to registers, the wave issues the next chunk's DMA into the same place and computes
//     while that is in flight (no workgroup barrier guards the records: a wave only ever reads what it loaded itself);
//   * the point phase runs ONE CHUNK BEHIND: the per-observation terms go to one of two LDS buffers, the points of the previous chunk are solved
//     from the other — by threads spread over all four waves — in the same barrier interval, so one barrier per chunk and no wave waits for
//     a serial point phase (k_backsub: one wave solved the points of a chunk while three were parked: SQ_WAIT_ANY 68 %).
// The body is a function with Trec as a __restrict__ PARAMETER for the reason given at schur_reg3_body (LDS reads may pass the pending DMA).
constexpr int BSR_PT = 3 * CHUNK;  // doubles of one buffer of per-observation terms
template <int NC> struct BsrCfg {
  static constexpr int NP = SchurRec<NC>::NPH;       // 16-byte pieces of a record in HBM (6 / 11)
  static constexpr int CS = (NC == 9) ? 9 : 6;       // doubles of a camera's step entry
  static constexpr int CSS = CS;                     // its stride in LDS (an odd stride would spread the banks; 40 000 bytes per workgroup let four share a CU)
  static constexpr size_t lds_bytes(int n_cams) { return ((size_t)BLOCK * NP * 2 + 2 * BSR_PT + (size_t)n_cams * CSS + 8) * sizeof(double); }
};
template <int NC, bool SCAL>
__device__ __forceinline__ void backsub_rec_body(const double* __restrict__ Trec, const int* __restrict__ obs_cam, const int* __restrict__ pt_start,
                                                 const int* __restrict__ chunk_start, const int* __restrict__ chunk_pts, int n_chunks, VecLayout lay,
                                                 const double* __restrict__ tab, const int* __restrict__ cam_off, const int* __restrict__ cam_np,
                                                 int n_cams, double lam, const double* __restrict__ Vblk, const double* __restrict__ gp,
                                                 const double* __restrict__ dp, const double* __restrict__ s_cam, double* __restrict__ sp,
                                                 double* __restrict__ step_partial, double* sh) {
  using Cfg = BsrCfg<NC>;
  constexpr int NP = Cfg::NP, CSS = Cfg::CSS;
  double2* sh_rec = reinterpret_cast<double2*>(sh);          // [BLOCK][NP] pieces, wave w owns [w * 64 * NP, (w + 1) * 64 * NP)
  double* sh_pt = sh + (size_t)BLOCK * NP * 2;               // [2][3][CHUNK]
  double* sh_step = sh_pt + 2 * BSR_PT;                      // [n_cams][CSS]
  const int tid = (int)threadIdx.x, lane = tid % WAVE;
  const int wv = __builtin_amdgcn_readfirstlane(tid / WAVE);
  for (int idx = tid; idx < n_cams * 3; idx += BLOCK) {
    const int c = idx / 3, a = idx % 3, off = cam_off[c];
    const double* Jl = tab + (long)c * CAMTAB_DOUBLES + 12;
    sh_step[c * CSS + a] = fma(Jl[3 * a + 2], s_cam[off + 2], fma(Jl[3 * a + 1], s_cam[off + 1], Jl[3 * a] * s_cam[off]));
    sh_step[c * CSS + 3 + a] = s_cam[off + 3 + a];
    if (NC == 9) sh_step[c * CSS + 6 + a] = (cam_np[c] == 9) ? s_cam[off + 6 + a] : 0.0;
  }
  const int n_obs = chunk_start[n_chunks];
  const int last_obs = max(n_obs - 1, 0);
  const long last_piece = (long)max(n_obs, 1) * NP - 1;
  const double2* rec_g = reinterpret_cast<const double2*>(Trec);
  double2* wrec = sh_rec + wv * (WAVE * NP);
  auto issue = [&](int o0) {  // this wave's 64 records of the chunk that starts at observation o0
    const long base = ((long)o0 + wv * WAVE) * NP + lane;
#pragma unroll
    for (int k = 0; k < NP; ++k)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rec_g + min(base + k * WAVE, last_piece)),
                                       (void __attribute__((address_space(3)))*)(wrec + k * WAVE), 16, 0, 0);
  };
  // inputs of the point phase of one chunk: thread t takes local point (t % 64) * 4 + t / 64 — consecutive points alternate between the waves
  // (a, b are kept as loaded — offsets into the sorted observations — and turned into chunk-local positions where they are used, an iteration
  // later: arithmetic on a loaded value in the iteration that loads it is a wait behind the DMA just issued)
  struct PtIn { int p, a, b, on; double V[6], d[3], g[3]; };
  const int lp0 = lane * 4 + wv;
  auto load_point = [&](int cp0, int npts) {
    PtIn q;
    q.on = lp0 < npts;
    q.p = min(cp0 + min(lp0, max(npts - 1, 0)), lay.P - 1);  // (clamped: the loads are unconditional)
    q.a = pt_start[q.p]; q.b = pt_start[q.p + 1];
#pragma unroll
    for (int k = 0; k < 6; ++k) q.V[k] = Vblk[(long)k * lay.Ppad + q.p];
#pragma unroll
    for (int k = 0; k < 3; ++k) { q.d[k] = dp[(long)k * lay.Ppad + q.p]; q.g[k] = gp[(long)k * lay.Ppad + q.p]; }
    return q;
  };
  double sc0 = 0.0, sc1 = 0.0;
  auto solve_point = [&](int p, const double* Vin, const double* din, const double* gin, const double* q) {
    double Vd[6], L[6], y[3], x[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) Vd[k] = Vin[k];
    Vd[0] = fma(lam * din[0], din[0], Vd[0]); Vd[3] = fma(lam * din[1], din[1], Vd[3]); Vd[5] = fma(lam * din[2], din[2], Vd[5]);
    if (chol3(Vd, L)) {
      chol3_fwd(L, gin, y);
      y[0] += q[0]; y[1] += q[1]; y[2] += q[2];
      chol3_bwd(L, y, x);
    } else {
      x[0] = x[1] = x[2] = 0.0;  // flagged by the Schur pass already
    }
    sp[p] = -x[0];
    sp[lay.Ppad + p] = -x[1];
    sp[2 * lay.Ppad + p] = -x[2];
    if (SCAL) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { const double ps = x[k] * din[k]; sc0 = fma(ps, ps, sc0); sc1 = fma(-gin[k], x[k], sc1); }
    }
  };
  auto point_phase = [&](const PtIn& q, const double* pt, int cp0, int npts, int o0) {
    if (q.on && q.b > q.a) {
      double acc[3] = {0.0, 0.0, 0.0};
      for (int j = q.a - o0; j < q.b - o0; ++j) { acc[0] += pt[j]; acc[1] += pt[CHUNK + j]; acc[2] += pt[2 * CHUNK + j]; }
      solve_point(q.p, q.V, q.d, q.g, acc);
    }
    for (int lp = tid + BLOCK; lp < npts; lp += BLOCK) {  // a chunk's range with more than 256 points (most of them unobserved): rare
      const int p = cp0 + lp, a = pt_start[p] - o0, b = pt_start[p + 1] - o0;
      if (b > a) {
        double acc[3] = {0.0, 0.0, 0.0}, V2[6], d2[3], g2[3];
        for (int j = a; j < b; ++j) { acc[0] += pt[j]; acc[1] += pt[CHUNK + j]; acc[2] += pt[2 * CHUNK + j]; }
#pragma unroll
        for (int k = 0; k < 6; ++k) V2[k] = Vblk[(long)k * lay.Ppad + p];
#pragma unroll
        for (int k = 0; k < 3; ++k) { d2[k] = dp[(long)k * lay.Ppad + p]; g2[k] = gp[(long)k * lay.Ppad + p]; }
        solve_point(p, V2, d2, g2, acc);
      }
    }
  };

  int ch = blockIdx.x;
  if (ch >= n_chunks) {
    if (SCAL && tid == 0) { step_partial[blockIdx.x * 4 + 0] = 0.0; step_partial[blockIdx.x * 4 + 1] = 0.0; step_partial[blockIdx.x * 4 + 2] = 0.0; step_partial[blockIdx.x * 4 + 3] = 0.0; }
    return;
  }
  // table entries of a chunk: (first observation, end, first point, points).  Nothing loaded in an iteration feeds an address of the same
  // iteration (the wait for it would be a vmcnt(0) behind the DMA just issued): the entries of chunk ch + 2 grid are loaded while chunk ch is
  // processed, the camera index of an observation and the inputs of a chunk's points one iteration ahead.
  struct Ent { int o0, o1, cp0, npts; };
  auto load_ent = [&](int c) { Ent e; e.o0 = chunk_start[c]; e.o1 = chunk_start[c + 1]; e.cp0 = chunk_pts[2 * c]; e.npts = chunk_pts[2 * c + 1]; return e; };
  const int stride = (int)gridDim.x, last_chunk = n_chunks - 1;
  Ent cur = load_ent(ch), nx = load_ent(min(ch + stride, last_chunk));
  int cam = obs_cam[min(cur.o0 + tid, last_obs)];
  issue(cur.o0);
  PtIn pprev;  // the chunk whose points are still to be solved: none yet
  pprev.p = 0; pprev.a = 0; pprev.b = 0; pprev.on = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) pprev.V[k] = 1.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { pprev.d[k] = 0.0; pprev.g[k] = 0.0; }
  Ent eprev{0, 0, 0, 0};
  int buf = 0;
  __syncthreads();  // sh_step
  while (true) {
    // everything issued an iteration ago has landed: this wave's records of `ch`, the table entries, the camera index, the inputs of the previous chunk's points
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (values whose loads sit behind the loop's back edge pass through an empty asm: the compiler cannot see the wait above and would put a
    // vmcnt(0) of its own at their first use — behind the DMA this iteration issues)
    asm volatile("" : "+v"(cam));
    asm volatile("" : "+v"(nx.o0)); asm volatile("" : "+v"(nx.o1)); asm volatile("" : "+v"(nx.cp0)); asm volatile("" : "+v"(nx.npts));
    asm volatile("" : "+v"(pprev.a)); asm volatile("" : "+v"(pprev.b));
#pragma unroll
    for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(pprev.V[k]));
#pragma unroll
    for (int k = 0; k < 3; ++k) { asm volatile("" : "+v"(pprev.d[k])); asm volatile("" : "+v"(pprev.g[k])); }
    double2 rec[NP];
    const double2* mine = wrec + lane * NP;
#pragma unroll
    for (int k = 0; k < NP; ++k) rec[k] = mine[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the record is in registers before the next DMA may overwrite it
#pragma unroll
    for (int k = 0; k < NP; ++k) asm volatile("" : "+v"(rec[k].x), "+v"(rec[k].y));
    const bool more = ch + stride < n_chunks;
    if (more) issue(nx.o0);
    const int ncam = obs_cam[min(nx.o0 + tid, last_obs)];
    const Ent nn = load_ent(min(ch + 2 * stride, last_chunk));
    const PtIn pcur = load_point(cur.cp0, cur.npts);
    __builtin_amdgcn_sched_barrier(0);
    // this observation's term T_i^T dc
    double t0, t1, t2;
    {
      const double* cs = sh_step + cam * CSS;
      const double w0 = cs[0], w1 = cs[1], w2 = cs[2];
      const double Y0 = rec[0].x, Y1 = rec[0].y, Y2 = rec[1].x;
      const double m0 = fma(w1, Y2, fma(-w2, Y1, cs[3]));
      const double m1 = fma(w2, Y0, fma(-w0, Y2, cs[4]));
      const double m2 = fma(w0, Y1, fma(-w1, Y0, cs[5]));
      // Q row-major behind Y: entries 3..11 of the record
      t0 = fma(rec[4].y, m2, fma(rec[3].x, m1, rec[1].y * m0));
      t1 = fma(rec[5].x, m2, fma(rec[3].y, m1, rec[2].x * m0));
      t2 = fma(rec[5].y, m2, fma(rec[4].x, m1, rec[2].y * m0));
      if constexpr (NC == 9) {  // T_intr row-major: entries 12..20
        const double i0 = cs[6], i1 = cs[7], i2 = cs[8];
        t0 = fma(rec[9].x, i2, fma(rec[7].y, i1, fma(rec[6].x, i0, t0)));
        t1 = fma(rec[9].y, i2, fma(rec[8].x, i1, fma(rec[6].y, i0, t1)));
        t2 = fma(rec[10].x, i2, fma(rec[8].y, i1, fma(rec[7].x, i0, t2)));
      }
      if (cur.o0 + tid >= cur.o1) { t0 = 0.0; t1 = 0.0; t2 = 0.0; }  // (a lane without an observation read somebody's record: select, do not multiply)
    }
    double* pt = sh_pt + buf * BSR_PT;
    pt[tid] = t0; pt[CHUNK + tid] = t1; pt[2 * CHUNK + tid] = t2;
    point_phase(pprev, sh_pt + (buf ^ 1) * BSR_PT, eprev.cp0, eprev.npts, eprev.o0);
    __syncthreads();
    pprev = pcur; eprev = cur;
    buf ^= 1;
    if (!more) break;
    cam = ncam; cur = nx; nx = nn; ch += stride;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  point_phase(pprev, sh_pt + (buf ^ 1) * BSR_PT, eprev.cp0, eprev.npts, eprev.o0);
  if (SCAL) {
    __syncthreads();  // (sh_pt is free behind it)
    double r;
    r = block_sum(sc0, sh_pt); if (tid == 0) step_partial[blockIdx.x * 4 + 0] = r;
    r = block_sum(sc1, sh_pt); if (tid == 0) { step_partial[blockIdx.x * 4 + 1] = r; step_partial[blockIdx.x * 4 + 2] = 0.0; step_partial[blockIdx.x * 4 + 3] = 0.0; }
  }
}
template <int NC, bool SCAL = false>
__global__ void __launch_bounds__(BLOCK)
k_backsub_rec(const double* __restrict__ Trec, const int* __restrict__ obs_cam, const int* __restrict__ pt_start, const int* __restrict__ chunk_start,
              const int* __restrict__ chunk_pts, int n_chunks, VecLayout lay, const double* __restrict__ tab, const int* __restrict__ cam_off,
              const int* __restrict__ cam_np, int n_cams, double lam, const double* __restrict__ lam_dev, const double* __restrict__ Vblk,
              const double* __restrict__ gvec, const double* __restrict__ sinv, double* __restrict__ svec, double* __restrict__ step_partial) {
  CBA_STAMP(ST_BACKSUB);
  extern __shared__ __attribute__((aligned(16))) double sh[];
  if (lam_dev) lam = *lam_dev;
  backsub_rec_body<NC, SCAL>(Trec, obs_cam, pt_start, chunk_start, chunk_pts, n_chunks, lay, tab, cam_off, cam_np, n_cams, lam, Vblk,
                             gvec + lay.ncp_pad, sinv + lay.ncp_pad, svec, svec + lay.ncp_pad, step_partial, sh);
}

// scalars of the Newton step: partial[b][0] = sum (s sinv)^2, [1] = sum g s
__global__ void __launch_bounds__(BLOCK)
k_step_scalars(const double* __restrict__ g, const double* __restrict__ sinv, const double* __restrict__ s, long total,
               long first, double* __restrict__ partial) {
  __shared__ double sh_red[BLOCK / WAVE];
  double s0 = 0, s1 = 0;
  for (long i = first + (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) {
    const double p = s[i] * sinv[i];
    s0 += p * p;
    s1 += g[i] * s[i];
  }
  double r;
  r = block_sum(s0, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 0] = r;
  r = block_sum(s1, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 1] = r;
  if (threadIdx.x == 0) { partial[blockIdx.x * 4 + 2] = 0.0; partial[blockIdx.x * 4 + 3] = 0.0; }
}
// w_sq = sum (s sinv - c g / sinv)^2 with c = scal[idx_gdot] / gh_sq   (device-side scalars, no host round trip)
__global__ void __launch_bounds__(BLOCK)
k_w_scalar(const double* __restrict__ g, const double* __restrict__ sinv, const double* __restrict__ s, long total,
           long first, const double* __restrict__ scal_gdot, double gh_sq, const double* __restrict__ gh_sq_dev,
           double* __restrict__ partial) {
  __shared__ double sh_red[BLOCK / WAVE];
  if (gh_sq_dev) gh_sq = *gh_sq_dev;
  const double c = scal_gdot[0] / gh_sq;
  double s0 = 0;
  for (long i = first + (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) {
    const double w = s[i] * sinv[i] - c * g[i] / sinv[i];
    s0 += w * w;
  }
  const double r = block_sum(s0, sh_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

// x_new = x + alpha g / sinv^2 + beta s (entries below n_over: x_new = cam_x_new) ; partial[b] = sum step^2
// tab_out != nullptr: workgroup 0 takes the whole camera block and then prepares the camera table of the new point (k_cam_prep's work: one
// launch fewer per trial point; the other workgroups share the point block).
__global__ void __launch_bounds__(BLOCK)
k_trial_update(const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ sinv,
               const double* __restrict__ s, double alpha, double beta, long total, int cam_end, int count_cams,
               const double* __restrict__ cam_x_new, int n_over, const double* __restrict__ ab_dev, double* __restrict__ x_new,
               double* __restrict__ partial, double* __restrict__ tab_out = nullptr, const double* __restrict__ cam_const = nullptr,
               const int* __restrict__ cam_model = nullptr, const int* __restrict__ cam_np = nullptr, const int* __restrict__ cam_off = nullptr,
               int n_cams = 0) {
  __shared__ double sh_red[BLOCK / WAVE];
  if (ab_dev) { alpha = ab_dev[0]; beta = ab_dev[1]; }  // fused step: coefficients from k_fused_subspace
  double s0 = 0;
  auto entry = [&](long i) {
    const double si = sinv[i];
    double st = alpha * g[i] / (si * si) + beta * s[i];
    double xn = x[i] + st;
    if (i < n_over) { xn = cam_x_new[i]; st = xn - x[i]; }  // camera block placed by the caller (bounded solves)
    x_new[i] = xn;
    if (i >= cam_end || count_cams) s0 += st * st;
  };
  if (tab_out) {
    if (blockIdx.x == 0)
      for (long i = threadIdx.x; i < cam_end; i += BLOCK) entry(i);
    for (long i = cam_end + (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) entry(i);
  } else {
    for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) entry(i);
  }
  const double r = block_sum(s0, sh_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
  if (tab_out && blockIdx.x == 0) {
    __threadfence();   // the camera block of x_new, written by this workgroup above, is read back below
    __syncthreads();
    for (int c = threadIdx.x; c < n_cams; c += BLOCK) {
      double xc[MAX_NC];
      const int np = cam_np[c];
      for (int i = 0; i < MAX_NC; ++i) xc[i] = (i < np) ? __builtin_nontemporal_load(&x_new[cam_off[c] + i]) : 0.0;
      CamTab t;
      cam_prepare(xc, cam_const + c * CAM_CONST_STRIDE, cam_model[c], np, &t, cam_off[c]);
      const double* src = reinterpret_cast<const double*>(&t);
      for (int i = 0; i < CAMTAB_DOUBLES; ++i) tab_out[c * CAMTAB_DOUBLES + i] = src[i];
    }
  }
}

// ---- fused iteration (cba_step): the two scalar decisions of an iteration made on the device, so that one iteration
// needs one host synchronisation instead of three.  scal slots: 0 gh_sq, 1 |x D|^2, 12 |J_h g_h|^2, 16 p_sq, 17 <g_h,p>,
// 20 w_sq; outputs 40 lam, 41 radius, 42 need_host, 43/44 p_S, 45 predicted, 46 alpha, 47 beta.  fz: [0] lam, [2] alpha, [3] beta.
__device__ __forceinline__ void fused_lam(double* __restrict__ scal, double radius_in, double* __restrict__ fz);
struct SubspaceIn;
__device__ __forceinline__ void fused_subspace(double* __restrict__ scal, const int* __restrict__ flags, double* __restrict__ fz, const SubspaceIn* pre = nullptr);
__global__ void k_fused_lam(double* __restrict__ scal, double radius_in, double* __restrict__ fz) { fused_lam(scal, radius_in, fz); }
__global__ void k_fused_subspace(double* __restrict__ scal, const int* __restrict__ flags, double* __restrict__ fz) {
  fused_subspace(scal, flags, fz);
}

__device__ __forceinline__ void fused_lam(double* __restrict__ scal, double radius_in, double* __restrict__ fz) {
  const double gh_sq = scal[0], jg_sq = scal[12] + scal[3], xs = sqrt(scal[1]);  // (scal[3]: C_gg of a bounded fused iteration, zero otherwise)
  const double radius = radius_in > 0.0 ? radius_in : (xs > 0.0 ? xs : 1.0);  // first iteration: Delta = ||x0 * scale_inv||
  const double lam = trf::damping(jg_sq, gh_sq, radius);
  fz[0] = lam; fz[1] = radius;
  scal[40] = lam; scal[41] = radius;
}

// (the inputs that do not come from the calling kernel's own sums, loaded ahead of them by k_step_cam)
struct SubspaceIn { double gh_sq, c_gg, jg, lam, radius; int f1, f2; };
__device__ __forceinline__ SubspaceIn subspace_inputs(const double* __restrict__ scal, const int* __restrict__ flags, const double* __restrict__ fz) {
  return SubspaceIn{scal[0], scal[3], scal[12], fz[0], fz[1], flags[1], flags[2]};
}
__device__ __forceinline__ void fused_subspace(double* __restrict__ scal, const int* __restrict__ flags, double* __restrict__ fz, const SubspaceIn* pre) {
  const SubspaceIn in = pre ? *pre : subspace_inputs(scal, flags, fz);
  const double gh_sq = in.gh_sq, jg_sq = in.jg + in.c_gg, p_sq = scal[16], ghp = scal[17];  // H_gg = ||J_h g_h||^2 + C_gg
  const double lam = in.lam, radius = in.radius, gh_norm = sqrt(gh_sq);
  const double c = ghp / gh_sq;
  // ||w||^2 = ||p - c g_h||^2 = ||p||^2 - <g_h, p>^2 / ||g_h||^2: relative error ~ eps ||p||^2 / ||w||^2, so it is used
  // only while w is not small against p (the primitives measure ||w||^2 by a pass of its own, k_w_scalar)
  const double w_sq = p_sq - ghp * ghp / gh_sq;
  scal[20] = w_sq;
  const bool two_d = true;
  double need_host = 0.0, pS[2] = {0.0, 0.0}, alpha = 0.0, beta = 0.0, predicted = 0.0;
  const bool ok = in.f1 == 0 && in.f2 == 0 && isfinite(p_sq) && isfinite(ghp) && gh_sq > 0.0;
  if (!ok || !(w_sq > 1e-3 * p_sq)) {
    need_host = 1.0;  // failed factorisation, or p nearly collinear with g_h: the host takes over with the primitives
  } else {
    double b00, b01 = 0.0, b11;
    const double w_norm = two_d ? sqrt(w_sq) : 1.0;
    if (two_d) trf::subspace_model(jg_sq, gh_sq, lam, ghp, p_sq, w_sq, &b00, &b01, &b11);
    else { b00 = jg_sq / gh_sq; b11 = 1.0; }
    trf::solve_subspace_2d(b00, b01, b11, gh_norm, 0.0, radius, pS);
    if (!two_d) pS[1] = 0.0;
    predicted = -(0.5 * (pS[0] * (b00 * pS[0] + b01 * pS[1]) + pS[1] * (b01 * pS[0] + b11 * pS[1])) + gh_norm * pS[0]);
    beta = two_d ? pS[1] / w_norm : 0.0;
    alpha = pS[0] / gh_norm - beta * c;
  }
  fz[2] = alpha; fz[3] = beta;
  scal[42] = need_host; scal[43] = pS[0]; scal[44] = pS[1]; scal[45] = predicted; scal[46] = alpha; scal[47] = beta;
}

// ---- single-rank fused iterations: fewer, fatter launches (a small problem is bound by launch latency: cfg2 runs
// ~40 kernels of a few microseconds per iteration).  Same arithmetic and summation order as the separate kernels.

// k_scale_update + k_lin_scalars in one pass over the vector
// Bounded camera parameters inside the fused iteration (round 5; scipy trf_bounds, trf.py:283-296, common.py CL_scaling_vector): what
// cba_get_camera_state + the host's Coleman-Li loop + cba_set_camera_scaling did in three host round trips.  For a camera entry the Jacobi scale
// (monotone-max state, kept apart in state_out) is multiplied by 1 / sqrt(v), v = distance to the bound the gradient points at (times the scale),
// cam_diag = diag_h scale_eff^2 with diag_h = g dv / scale is what k_schur_finalize adds to the diagonal of S, the sums are formed with the
// effective scale, max |g| becomes max |g v| and the fourth partial column carries C_gg = sum diag_h g_h^2 (the Coleman-Li term of g_h^T (H + C) g_h).
struct BoundArgs {
  const double* lb;         // [ncp] or nullptr: no bounds, plain k_scale_lin
  const double* ub;
  const double* state_in;   // [ncp_pad] Jacobi scale of the camera block before this linearisation
  double* state_out;        // [ncp_pad] ... after it
  double* cam_diag_out;     // [ncp_pad]
};
template <int NC, bool BND = false>  // BND: the bounded variant (a kernel of its own: the plain pass measured 8.7 us without the branch, 10.0 with it)
__global__ void __launch_bounds__(BLOCK)
k_scale_lin(const double* __restrict__ Upacked, const double* __restrict__ Vblk, const int* __restrict__ param_cam,
            const int* __restrict__ param_loc, VecLayout lay, int first, double* __restrict__ sinv, const double* __restrict__ cdiag,
            const double* __restrict__ x, const double* __restrict__ g, double* __restrict__ v1, double* __restrict__ partial,
            double* __restrict__ partial_max, const double* __restrict__ sinv_in = nullptr, BoundArgs bnd = BoundArgs{}) {
  CBA_STAMP(ST_SCALE_LIN);
  // sinv_in != nullptr: the scale is read there and written to `sinv` (the speculative linearisation of a trial point leaves the current one alone)
  using UP = UPack<NC>;
  __shared__ double sh_red[BLOCK / WAVE];
  const long total = lay.total();
  const double* sin = sinv_in ? sinv_in : sinv;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, m = 0;
  for (long i = (long)blockIdx.x * BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * BLOCK) {
    double si = sin[i];
    bool live = true;
    double v = 0.0;
    if (BND && i < lay.ncp) {  // bounded camera entry: Jacobi scale from its own state, then the Coleman-Li factor
      const int r = param_loc[i];
      double sj = sqrt(Upacked[param_cam[i] * UP::STRIDE + UP::idx(r, r)]);
      if (first) { if (sj == 0.0) sj = 1.0; } else sj = fmax(sj, bnd.state_in[i]);
      bnd.state_out[i] = sj;
      const double gi = g[i], xi = x[i], lo = bnd.lb[i], hi = bnd.ub[i];
      double vv = 1.0, dv = 0.0;
      if (gi < 0.0 && isfinite(hi)) { vv = hi - xi; dv = -1.0; }
      else if (gi > 0.0 && isfinite(lo)) { vv = xi - lo; dv = 1.0; }
      m = fmax(m, fabs(gi * vv));              // ||g v||_inf (trf.py:298)
      if (dv != 0.0) vv *= sj;                 // v[dv != 0] *= scale_inv
      const double se = sj / sqrt(vv);         // effective scale_inv = scale_inv / sqrt(v)
      const double dh = gi * dv / sj;          // diag_h = g dv scale  (>= 0)
      sinv[i] = se;
      bnd.cam_diag_out[i] = dh * se * se;
      const double gh = gi / se;
      v1[i] = gh / se;
      s0 += gh * gh;
      s1 += (xi * se) * (xi * se);
      s2 += xi * xi;
      s3 += dh * gh * gh;
      continue;
    }
    if (i < lay.ncp_pad) {
      if (i >= lay.ncp) live = false;  // padding keeps scale 1
      else { const int r = param_loc[i]; v = Upacked[param_cam[i] * UP::STRIDE + UP::idx(r, r)]; }
    } else {
      const long k = (i - lay.ncp_pad) / lay.Ppad, p = (i - lay.ncp_pad) % lay.Ppad;
      if (p >= lay.P) live = false;
      else {
        const int q = (k == 0) ? 0 : (k == 1 ? 3 : 5);
        v = Vblk[(long)q * lay.Ppad + p];
        if (cdiag) v += cdiag[k * lay.Ppad + p];
      }
    }
    if (live) {
      v = sqrt(v);
      if (first) { if (v == 0.0) v = 1.0; } else v = fmax(v, si);
      si = v;
      sinv[i] = si;
    } else if (sinv_in) {
      sinv[i] = si;  // padding entries: carried over
    }
    const double gi = g[i], xi = x[i];
    const double gh = gi / si;
    v1[i] = gh / si;
    m = fmax(m, fabs(gi));
    s0 += gh * gh;
    s1 += (xi * si) * (xi * si);
    s2 += xi * xi;
  }
  double r;
  r = block_sum(s0, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 0] = r;
  r = block_sum(s1, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 1] = r;
  r = block_sum(s2, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 2] = r;
  r = block_sum(s3, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 3] = r;  // C_gg (zero without bounds)
  r = block_max(m, sh_red); if (threadIdx.x == 0) partial_max[blockIdx.x] = r;
  if (BND && blockIdx.x == 0)  // padding entries of the state carry over
    for (int i = lay.ncp + threadIdx.x; i < lay.ncp_pad; i += BLOCK) { bnd.state_out[i] = 1.0; bnd.cam_diag_out[i] = 0.0; }
}

__device__ __forceinline__ double column_sum(const double* __restrict__ partial, int nrow, int width, int col, double* sh) {
  double s = 0.0;
  for (int b = threadIdx.x; b < nrow; b += BLOCK) s += partial[(long)b * width + col];
  return block_sum(s, sh);
}

// the three reductions of the linearisation (sums, max |g|, ||J v||^2) and the damping in one launch
__global__ void __launch_bounds__(BLOCK)
k_lin_finish(const double* __restrict__ partial_lin, const double* __restrict__ partial_max, int rows_lin,
             const double* __restrict__ partial_jv, int rows_jv, double radius, double* __restrict__ scal, double* __restrict__ fz) {
  // nine sums and a maximum in ONE pass and one barrier (column by column — ten block reductions in a row — the kernel took 10.6 us);
  // per value the order of the additions is the one of column_sum / block_sum
  __shared__ double sh_red[10][BLOCK / WAVE];
  double v[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) v[q] = 0.0;
  for (int b = threadIdx.x; b < rows_lin; b += BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += partial_lin[(long)b * 4 + j];
    v[4] = fmax(v[4], partial_max[b]);
  }
  for (int b = threadIdx.x; b < rows_jv; b += BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[5 + j] += partial_jv[(long)b * 4 + j];
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double r = (q == 4) ? wave_max(v[q]) : wave_sum(v[q]);
    if (lane == 0) sh_red[q][w] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      double r = 0.0;
      for (int i = 0; i < BLOCK / WAVE; ++i) r = (q == 4) ? fmax(r, sh_red[q][i]) : r + sh_red[q][i];
      scal[q < 5 ? q : 12 + (q - 5)] = r;
    }
    fused_lam(scal, radius, fz);
  }
}

// reduction of the step scalars and the subspace step in one launch
__global__ void __launch_bounds__(BLOCK)
k_step_finish(const double* __restrict__ partial, int rows, double* __restrict__ scal, const int* __restrict__ flags, double* __restrict__ fz) {
  __shared__ double sh_red[4][BLOCK / WAVE];  // four sums, one pass, one barrier (as k_lin_finish)
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < rows; b += BLOCK) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += partial[(long)b * 4 + j];
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double r = wave_sum(v[j]);
    if (lane == 0) sh_red[j][w] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double r = 0.0;
      for (int i = 0; i < BLOCK / WAVE; ++i) r += sh_red[j][i];
      scal[16 + j] = r;
    }
    fused_subspace(scal, flags, fz);
  }
}

// Small problems (a few thousand parameters: the reference's own sessions, cfg2): k_step_scalars and k_step_finish as ONE workgroup — the sums
// over the whole vector by 1024 threads, then the subspace step — one launch (~4 us) less per iteration.
constexpr int STEP_SMALL_THREADS = 1024;
constexpr long STEP_SMALL_MAX = 32768;  // parameters (32 per thread)
__global__ void __launch_bounds__(STEP_SMALL_THREADS)
k_step_small(const double* __restrict__ g, const double* __restrict__ sinv, const double* __restrict__ s, long total, double* __restrict__ scal,
             const int* __restrict__ flags, double* __restrict__ fz) {
  __shared__ double sh_red[2][STEP_SMALL_THREADS / WAVE];
  double s0 = 0.0, s1 = 0.0;
  for (long i = threadIdx.x; i < total; i += STEP_SMALL_THREADS) {
    const double p = s[i] * sinv[i];
    s0 += p * p;
    s1 += g[i] * s[i];
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
  const double r0 = wave_sum(s0), r1 = wave_sum(s1);
  if (lane == 0) { sh_red[0][w] = r0; sh_red[1][w] = r1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int i = 0; i < STEP_SMALL_THREADS / WAVE; ++i) { a += sh_red[0][i]; b += sh_red[1][i]; }
    scal[16] = a; scal[17] = b; scal[18] = 0.0; scal[19] = 0.0;
    fused_subspace(scal, flags, fz);
  }
}

// Single-rank fused iteration, camera-sorted build (round 5): what used to be k_step_scalars + k_step_finish + k_trial_update's camera share in ONE
// workgroup.  The point block's step scalars arrive as per-workgroup partials of k_backsub<.., SCAL>; this kernel adds the camera block, takes the
// subspace step (fused_subspace: alpha, beta -> fz[2], fz[3]), forms the camera entries of the trial point and its camera table (k_cam_prep's work),
// and leaves the camera block's share of ||step||^2 in step_cam[0].  The point entries of the trial point are formed by the build pass that evaluates
// it (k_build_cs<.., TRIAL>), which stages every point of its super-chunk anyway: no vector pass over x, g, sinv, s in between.
// Dynamic LDS: ncp_pad doubles (the camera block of the trial point).
__global__ void __launch_bounds__(BLOCK)
k_step_cam(const double* __restrict__ partial, int rows, const double* __restrict__ x, const double* __restrict__ g, const double* __restrict__ sinv,
           const double* __restrict__ s, int ncp_pad, double* __restrict__ scal, const int* __restrict__ flags, double* __restrict__ fz,
           double* __restrict__ x_new, double* __restrict__ step_cam, double* __restrict__ tab_out, const double* __restrict__ cam_const,
           const int* __restrict__ cam_model, const int* __restrict__ cam_np, const int* __restrict__ cam_off, int n_cams,
           const double* __restrict__ lb = nullptr, const double* __restrict__ ub = nullptr, int ncp = 0) {
  CBA_STAMP(ST_STEP_CAM);
  extern __shared__ __attribute__((aligned(16))) double sh_xc[];  // [ncp_pad]
  __shared__ double sh_red[4][BLOCK / WAVE];
  __shared__ double sh_ab[3];
  __shared__ int sh_outside;
  if (threadIdx.x == 0) sh_outside = 0;
  // (round 6) everything the later phases read that does not depend on the earlier ones is requested HERE, in the first round trip: the subspace
  // step's inputs (thread 0), a thread's first two camera entries (all of them up to 512 camera parameters), the constants of the camera whose
  // table the thread prepares at the end — the kernel was five dependent round trips of one workgroup, 10 us
  SubspaceIn pre{};
  if (threadIdx.x == 0) pre = subspace_inputs(scal, flags, fz);
  constexpr int KEEP = 2;
  double kx[KEEP], kg[KEEP], ksi[KEEP], ks[KEEP];
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const int i = threadIdx.x + q * BLOCK;
    const bool in = i < ncp_pad;
    kx[q] = in ? x[i] : 0.0; kg[q] = in ? g[i] : 0.0; ksi[q] = in ? sinv[i] : 1.0; ks[q] = in ? s[i] : 0.0;
  }
  const int my_cam = threadIdx.x < n_cams ? threadIdx.x : -1;
  int my_np = 0, my_off = 0, my_model = 0;
  double my_const[CAM_CONST_STRIDE];
  if (my_cam >= 0) {
    my_np = cam_np[my_cam]; my_off = cam_off[my_cam]; my_model = cam_model[my_cam];
#pragma unroll
    for (int i = 0; i < CAM_CONST_STRIDE; ++i) my_const[i] = cam_const[my_cam * CAM_CONST_STRIDE + i];
  }
  double v[2] = {0.0, 0.0};
  for (int b = threadIdx.x; b < rows; b += BLOCK) { v[0] += partial[(long)b * 4 + 0]; v[1] += partial[(long)b * 4 + 1]; }
#pragma unroll
  for (int q = 0; q < KEEP; ++q) {
    const double ps = ks[q] * ksi[q];
    v[0] = fma(ps, ps, v[0]);
    v[1] = fma(kg[q], ks[q], v[1]);
  }
  for (int i = threadIdx.x + KEEP * BLOCK; i < ncp_pad; i += BLOCK) {
    const double si = s[i], ps = si * sinv[i];
    v[0] = fma(ps, ps, v[0]);
    v[1] = fma(g[i], si, v[1]);
  }
  const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const double r = wave_sum(v[j]);
    if (lane == 0) sh_red[j][w] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double r = 0.0;
      for (int i = 0; i < BLOCK / WAVE; ++i) r += sh_red[j][i];
      scal[16 + j] = r;
    }
    scal[18] = 0.0; scal[19] = 0.0;
    fused_subspace(scal, flags, fz, &pre);
    sh_ab[0] = fz[2]; sh_ab[1] = fz[3]; sh_ab[2] = scal[42];
  }
  __syncthreads();
  if (sh_ab[2] != 0.0) return;  // need_host: no trial point (the build pass behind this kernel skips itself)
  const double alpha = sh_ab[0], beta = sh_ab[1];
  double s0 = 0.0;
  auto entry = [&](int i, double xi, double gi, double si, double stp) {
    const double st = alpha * gi / (si * si) + beta * stp;
    const double xn = xi + st;
    x_new[i] = xn;
    sh_xc[i] = xn;
    s0 = fma(st, st, s0);
    // bounded solve (trf.py:129-202, select_step): a trial point that is not STRICTLY inside the box goes back to the host, which chooses between the
    // truncated step, its reflection and the scaled anti-gradient with the primitives — rare (the bounds of the reference are far from its solutions)
    if (lb && i < ncp && !(xn > lb[i] && xn < ub[i])) sh_outside = 1;
  };
#pragma unroll
  for (int q = 0; q < KEEP; ++q)
    if (threadIdx.x + q * BLOCK < ncp_pad) entry(threadIdx.x + q * BLOCK, kx[q], kg[q], ksi[q], ks[q]);
  for (int i = threadIdx.x + KEEP * BLOCK; i < ncp_pad; i += BLOCK) entry(i, x[i], g[i], sinv[i], s[i]);
  {
    const double r = wave_sum(s0);
    if (lane == 0) sh_red[2][w] = r;
  }
  __syncthreads();
  if (sh_outside) {
    // need_host (2: the step leaves the bounds); the build pass behind this kernel skips itself.  x_new's camera entries are undefined from here on;
    // the step-norm row is zeroed so that the packet's slot 28 does not sum a stale value (the host ignores trial and step_norm when need_host != 0)
    if (threadIdx.x == 0) { scal[42] = 2.0; step_cam[0] = 0.0; }
    return;
  }
  if (threadIdx.x == 0) {
    double r = 0.0;
    for (int i = 0; i < BLOCK / WAVE; ++i) r += sh_red[2][i];
    step_cam[0] = r;
  }
  for (int c = threadIdx.x; c < n_cams; c += BLOCK) {
    double xc[MAX_NC];
    const bool mine = c == my_cam;  // (the first BLOCK cameras: constants loaded at the top)
    const int np = mine ? my_np : cam_np[c], off = mine ? my_off : cam_off[c];
    for (int i = 0; i < MAX_NC; ++i) xc[i] = (i < np) ? sh_xc[off + i] : 0.0;
    CamTab t;
    if (mine) cam_prepare(xc, my_const, my_model, np, &t, off);
    else cam_prepare(xc, cam_const + c * CAM_CONST_STRIDE, cam_model[c], np, &t, off);
    const double* src = reinterpret_cast<const double*>(&t);
    for (int i = 0; i < CAMTAB_DOUBLES; ++i) tab_out[c * CAMTAB_DOUBLES + i] = src[i];
  }
}

// End of a primitive: the host-visible scalars and flags go straight to pinned host memory (mapped into the device's
// address space) — no copy-engine round trip — and the flags are cleared for the next primitive.
__global__ void __launch_bounds__(BLOCK)
k_publish(double* __restrict__ scal, int n_scal, int* __restrict__ flags, double* __restrict__ host_scal,
          int* __restrict__ host_flags, unsigned long long seq, const double* __restrict__ part_a, int rows_a, int slot_a,
          const double* __restrict__ part_b, int rows_b, int slot_b) {
  __shared__ double sh_red[BLOCK / WAVE];
  const int t = threadIdx.x;
  // single-rank fused step: the last two per-workgroup partial columns (trial cost, step norm) are summed here
  if (part_a) { const double r = column_sum(part_a, rows_a, 1, 0, sh_red); if (t == 0) scal[slot_a] = r; }
  if (part_b) { const double r = column_sum(part_b, rows_b, 1, 0, sh_red); if (t == 0) scal[slot_b] = r; }
  __syncthreads();
  if (t < n_scal) host_scal[t] = scal[t];
  if (t < 4) { host_flags[t] = flags[t]; flags[t] = 0; }
  // the sequence number goes last: the host spins on it instead of sleeping in hipStreamSynchronize (host_scal[63])
  __threadfence_system();
  __syncthreads();
  if (t == 0) reinterpret_cast<volatile unsigned long long*>(host_scal)[63] = seq;
}

// camera blocks of up to three device vectors to mapped host memory, sequence number last (as k_publish)
__global__ void k_publish_cam(const double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c, int ncp,
                              double* __restrict__ host, unsigned long long* __restrict__ host_seq, unsigned long long seq) {
  for (int i = threadIdx.x; i < ncp; i += blockDim.x) {
    host[i] = a[i];
    if (b) host[ncp + i] = b[i];
    if (c) host[2 * ncp + i] = c[i];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned long long*>(host_seq) = seq;
}

__global__ void k_fill(double* __restrict__ p, double v, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}


// ------------------------------------------------------------------------------------------------
// Undistortion + DLT triangulation, one thread per world point (include/caliscope_ba.h: cba_triangulate).
// Pinhole: OpenCV's five fixed-point iterations  x <- (x0 - delta(x)) / cdist(x).  Fisheye: Newton on
// theta_d = theta (1 + k1 theta^2 + ...), at most 10 steps, stop below 1e-8, theta_d clipped to [-pi/2, pi/2].
__device__ __forceinline__ void undistort_one(int model, const double* __restrict__ in9, double u, double v, int f32, double* xo,
                                              double* yo) {
  if (f32) { u = (double)(float)u; v = (double)(float)v; }
  const double fx = in9[0], fy = in9[1], cx = in9[2], cy = in9[3];
  const double x0 = (u - cx) / fx, y0 = (v - cy) / fy;
  double x = x0, y = y0;
  if (model == 0) {
    const double k1 = in9[4], k2 = in9[5], p1 = in9[6], p2 = in9[7], k3 = in9[8];
    for (int it = 0; it < 5; ++it) {
      const double r2 = x * x + y * y;
      const double icd = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2);
      const double dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
      const double dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
      x = (x0 - dx) * icd;
      y = (y0 - dy) * icd;
    }
  } else {
    const double k1 = in9[4], k2 = in9[5], k3 = in9[6], k4 = in9[7];
    const double hp = 1.5707963267948966;
    double td = sqrt(x0 * x0 + y0 * y0);
    td = fmin(fmax(td, -hp), hp);
    double scale = 1.0;
    if (td > 1e-8) {
      double th = td;
      for (int it = 0; it < 10; ++it) {
        const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const double fix = (th * (1.0 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8) - td) /
                           (1.0 + 3.0 * k1 * t2 + 5.0 * k2 * t4 + 7.0 * k3 * t6 + 9.0 * k4 * t8);
        th -= fix;
        if (fabs(fix) < 1e-8) break;
      }
      scale = tan(th) / td;
    }
    x = x0 * scale;
    y = y0 * scale;
  }
  if (f32) { x = (double)(float)x; y = (double)(float)y; }
  *xo = x; *yo = y;
}

__global__ void __launch_bounds__(BLOCK)
k_triangulate(long n_points, const long* __restrict__ pt_start, const int* __restrict__ obs_cam, const double* __restrict__ obs_xy,
              const int* __restrict__ cam_model, const double* __restrict__ cam_intr, const double* __restrict__ cam_P, int f32,
              double* __restrict__ xyz, double* __restrict__ undist) {
  const long q = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (q >= n_points) return;
  const long a = pt_start[q], b = pt_start[q + 1];
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  double M[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) M[r][c] = 0.0;
  for (long i = a; i < b; ++i) {
    const int cam = obs_cam[i];
    double x = obs_xy[2 * i], y = obs_xy[2 * i + 1];
    if (cam_intr) undistort_one(cam_model[cam], cam_intr + 9 * cam, x, y, f32, &x, &y);
    if (undist) { undist[2 * i] = x; undist[2 * i + 1] = y; }
    const double* P = cam_P + 12 * cam;
    double r0[4], r1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { r0[c] = x * P[8 + c] - P[c]; r1[c] = y * P[8 + c] - P[4 + c]; }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = r; c < 4; ++c) M[r][c] += r0[r] * r0[c] + r1[r] * r1[c];
  }
  if (b - a < 2) { xyz[3 * q] = xyz[3 * q + 1] = xyz[3 * q + 2] = nan; return; }
#pragma unroll
  for (int r = 1; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) M[r][c] = M[c][r];
  // cyclic Jacobi on the symmetric 4 x 4; V accumulates the rotations (columns = eigenvectors)
  double V[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) V[r][c] = (r == c) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) off += M[p][r] * M[p][r];
    const double diag = M[0][0] * M[0][0] + M[1][1] * M[1][1] + M[2][2] * M[2][2] + M[3][3] * M[3][3];
    if (off <= 1e-34 * diag) break;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) {
        const double apq = M[p][r];
        if (apq != 0.0) {
          const double theta = (M[r][r] - M[p][p]) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // columns p, r of M and V
            const double mkp = M[k][p], mkr = M[k][r];
            M[k][p] = c * mkp - s * mkr; M[k][r] = s * mkp + c * mkr;
            const double vkp = V[k][p], vkr = V[k][r];
            V[k][p] = c * vkp - s * vkr; V[k][r] = s * vkp + c * vkr;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {  // rows p, r of M
            const double mpk = M[p][k], mrk = M[r][k];
            M[p][k] = c * mpk - s * mrk; M[r][k] = s * mpk + c * mrk;
          }
        }
      }
  }
  int best = 0;
  double smallest = M[0][0];  // (kept beside `best`: M[best][best] is a dynamic index and sent the whole matrix to scratch)
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (M[k][k] < smallest) { smallest = M[k][k]; best = k; }
  double w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = (best == 0) ? V[k][0] : (best == 1) ? V[k][1] : (best == 2) ? V[k][2] : V[k][3];
  xyz[3 * q] = w[0] / w[3];
  xyz[3 * q + 1] = w[1] / w[3];
  xyz[3 * q + 2] = w[2] / w[3];
}


// ------------------------------------------------------------------------------------------------
// Heavy points: world points with more observations than the pair plan takes (HEAVY_OBS) — static markers seen again in
// every frame.  Their share of the Schur complement is formed per CAMERA, not per observation pair:
//   W_pc = sum_{i in (p, c)} T_i   (NC x 3),    Sacc += W_p W_p^T   over the cameras that see p,
// one workgroup per heavy point; the pair plan skips them.  Points with more than CHUNK observations are also split
// over several chunks ("fragments": chunk_pts = (point, -1)); k_build / k_backsub add a fragment's sums by atomics.
constexpr int HEAVY_OBS = 40;  // more observations than this: per-camera sums (k_heavy_schur) instead of 40^2 / 2 pair codes per point

__global__ void k_zero_heavy(const int* __restrict__ heavy_pts, int n_heavy, VecLayout lay, double* __restrict__ a, int rows_a,
                             double* __restrict__ b, int rows_b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_heavy) return;
  const int p = heavy_pts[t];
  for (int k = 0; k < rows_a; ++k) a[(long)k * lay.Ppad + p] = 0.0;
  for (int k = 0; k < rows_b; ++k) b[(long)k * lay.Ppad + p] = 0.0;
}

// dp of the fragmented heavy points: svec holds sum_i W_i^T dc (k_backsub fragments), on top of g_p
__global__ void k_heavy_finish(const int* __restrict__ heavy_pts, const int* __restrict__ heavy_frag, int n_heavy, VecLayout lay, double lam,
                               const double* __restrict__ Vblk, const double* __restrict__ gvec, const double* __restrict__ sinv,
                               double* __restrict__ svec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_heavy || !heavy_frag[t]) return;
  const int p = heavy_pts[t];
  const double* gp = gvec + lay.ncp_pad;
  const double* dp = sinv + lay.ncp_pad;
  double* sp = svec + lay.ncp_pad;
  double Vd[6], L[6], q[3], y[3], x[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 6; ++k) Vd[k] = Vblk[(long)k * lay.Ppad + p];
  const double d0 = dp[p], d1 = dp[lay.Ppad + p], d2 = dp[2 * lay.Ppad + p];
  Vd[0] += lam * d0 * d0; Vd[3] += lam * d1 * d1; Vd[5] += lam * d2 * d2;
#pragma unroll
  for (int k = 0; k < 3; ++k) q[k] = gp[(long)k * lay.Ppad + p] + sp[(long)k * lay.Ppad + p];
  if (chol3(Vd, L)) { chol3_fwd(L, q, y); chol3_bwd(L, y, x); }
#pragma unroll
  for (int k = 0; k < 3; ++k) sp[(long)k * lay.Ppad + p] = -x[k];
}

template <int NC>
__global__ void __launch_bounds__(BLOCK)
k_heavy_schur(const int* __restrict__ heavy_pts, const int* __restrict__ pt_start, const int* __restrict__ obs_cam,
              const int* __restrict__ cam_off, const int* __restrict__ cam_np, int ncp,
              const double* __restrict__ Trec, const double* __restrict__ tab, double* __restrict__ heavy_W, double* __restrict__ Sacc) {
  constexpr int REC = SchurRec<NC>::HREC;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* W = sh;                                  // [ncp][3]
  int* seen = reinterpret_cast<int*>(W + (size_t)ncp * 3);  // [ncp] 1 if the parameter's camera sees the point
  const int h = blockIdx.x, p = heavy_pts[h];
  for (int e = threadIdx.x; e < ncp * 3; e += BLOCK) W[e] = 0.0;
  for (int e = threadIdx.x; e < ncp; e += BLOCK) seen[e] = 0;
  __syncthreads();
  const int o0 = pt_start[p], o1 = pt_start[p + 1];
  for (int i = o0 + threadIdx.x; i < o1; i += BLOCK) {
    const int cam = obs_cam[i], off = cam_off[cam], np = cam_np[cam];
    double T[3 * NC];
    expand_record<NC>(Trec + (long)i * REC, tab + (long)cam * CAMTAB_DOUBLES + 12, T);  // compact record -> true T (J_l of the camera)
#pragma unroll
    for (int r = 0; r < NC; ++r) {  // (constant trip count: T stays in registers; `r < np` as a loop bound indexed it dynamically: 160 / 224 bytes of scratch)
      if (r < np) {
        lds_add(&W[(off + r) * 3 + 0], T[3 * r]); lds_add(&W[(off + r) * 3 + 1], T[3 * r + 1]); lds_add(&W[(off + r) * 3 + 2], T[3 * r + 2]);
        seen[off + r] = 1;
      }
    }
  }
  __syncthreads();
  double* Wg = heavy_W + (long)h * ncp * 3;
  for (int e = threadIdx.x; e < ncp * 3; e += BLOCK) Wg[e] = W[e];
  // upper triangle of W W^T, rows / columns of cameras that see the point only
  for (long e = threadIdx.x; e < (long)ncp * ncp; e += BLOCK) {
    const int r = (int)(e / ncp), c = (int)(e % ncp);
    if (c < r || !seen[r] || !seen[c]) continue;
    const double acc = W[r * 3] * W[c * 3] + W[r * 3 + 1] * W[c * 3 + 1] + W[r * 3 + 2] * W[c * 3 + 2];
    unsafeAtomicAdd(&Sacc[(long)r * ncp + c], acc);
  }
}

// ------------------------------------------------------------------------------------------------
// Rigid-distance constraint rows (reference core/reprojection.py:112-117 residual, :207-226 Jacobian;
// capture_volume.py:446-531 group arrays):  r_c = w_c (|| mean X[a_c] - mean X[b_c] || - d_c), one row per constraint
// after the 2N reprojection rows, derivative +-1/4 w_c unit_c on each of the 8 group slots (a repeated point index
// collects its slots).  The rows touch points only, so H_pp = V~ + J_c^T J_c stops being block diagonal; the damped
// step uses  H_pp^-1 = V~^-1 - V~^-1 J_c^T M^-1 J_c V~^-1,  M = I + J_c V~^-1 J_c^T  (Woodbury): everything the
// unconstrained path computes stays as it is and a correction is added per CONNECTED COMPONENT of the constraint
// graph (the corners of one board in one frame), one workgroup per component.
struct ConPlan {
  int n_con, n_comp;
  const int* pt;          // [n_con][8] world point of every group slot (0-3 group a, 4-7 group b), component order
  const int* lp;          // [n_con][8] index of that point inside its component
  const double* dist;     // [n_con]
  const double* weight;   // [n_con]
  const int* order;       // [n_con] caller's row index of constraint c (residual hook)
  const int* comp_con;    // [n_comp + 1] constraint range of a component
  const int* comp_pt;     // [n_comp + 1] range into comp_pts
  const int* comp_pts;    // world points of each component
  const long* comp_m;     // [n_comp + 1] offset of the component's M (m x m) in Mbuf
  double* f;              // [n_con] robust-scaled residual at the linearisation point
  double* u;              // [n_con][3] robust-scaled row direction  w rs unit
  double* z;              // [n_con][8][3]  L_p^-1 (slot coefficient)
  double* M;              // sum m_k^2
  double* G;              // [n_con][ncp + 1]
  double* cdiag;          // [3][Ppad] squared column norms of the constraint rows
  double* w;              // [n_con] scratch of k_con_backsub (J_c V~^-1 q, then M^-1 of it)
  const int* heavy_pts;   // sorted heavy points (k_heavy_schur) and their per-camera sums W [n_heavy][ncp][3]
  const double* heavy_W;
  int n_heavy;
  double* big;            // 9 doubles per component point (offset 9 comp_pt[k]): per-point scratch of the components that do not fit the LDS copy; else nullptr
  // small components (round 6): every component has at most CON_SMALL_M rows and its dense blocks fit `small_lds` bytes of LDS — k_con_schur_small /
  // k_con_backsub_small run, and `M` holds the INVERSE of each component's Cholesky factor (row-major m x m) instead of the factor
  int small;
  int small_lds;          // dynamic LDS of k_con_schur_small (bytes)
  int max_m, max_np;      // rows / points of the largest component
};

__device__ __forceinline__ void con_geometry(const ConPlan& cp, int c, const double* __restrict__ px, VecLayout lay, double* unit,
                                             double* nrm) {
  double a[3] = {0.0, 0.0, 0.0}, b[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int pa = cp.pt[c * 8 + s], pb = cp.pt[c * 8 + 4 + s];
#pragma unroll
    for (int k = 0; k < 3; ++k) { a[k] += px[(long)k * lay.Ppad + pa]; b[k] += px[(long)k * lay.Ppad + pb]; }
  }
  double d[3], n2 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { d[k] = 0.25 * (a[k] - b[k]); n2 += d[k] * d[k]; }
  const double n = sqrt(n2);
  const double inv = n > 0.0 ? 1.0 / n : 0.0;  // zero subgradient at coincident endpoints (reprojection.py:213-214)
#pragma unroll
  for (int k = 0; k < 3; ++k) unit[k] = d[k] * inv;
  *nrm = n;
}

// LIN = false: cost of the rows at xvec (trial points, residual hook).  LIN = true: also f, u, and the rows' share of
// the gradient and of the squared column norms (global FP64 atomics: a few per constraint).
template <bool LIN>
__global__ void __launch_bounds__(BLOCK)
k_con_eval(ConPlan cp, const double* __restrict__ xvec, VecLayout lay, int loss, double f_scale, double* __restrict__ gvec,
           double* __restrict__ partial, int* __restrict__ flags, double* __restrict__ r_out) {
  __shared__ double sh_red[BLOCK / WAVE];
  const double* px = xvec + lay.ncp_pad;
  double acc = 0.0;
  bool bad = false;
  for (int c = blockIdx.x * BLOCK + threadIdx.x; c < cp.n_con; c += gridDim.x * BLOCK) {
    double unit[3], nrm;
    con_geometry(cp, c, px, lay, unit, &nrm);
    const double r = (nrm - cp.dist[c]) * cp.weight[c];
    if (!isfinite(r)) bad = true;
    if (r_out) r_out[cp.order[c]] = r;
    if (!LIN) {
      acc += robust_cost_one(loss, f_scale, r);
    } else {
      double rs, er;
      acc += robust_one(loss, f_scale, r, &rs, &er);
      cp.f[c] = er;
      const double w = cp.weight[c] * rs;
      const double u[3] = {unit[0] * w, unit[1] * w, unit[2] * w};
      cp.u[c * 3] = u[0]; cp.u[c * 3 + 1] = u[1]; cp.u[c * 3 + 2] = u[2];
      double* gp = gvec + lay.ncp_pad;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int p = cp.pt[c * 8 + s];
        bool first = true;
        double mult = 0.0;  // signed multiplicity of p among the slots, in quarters
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2)
          if (cp.pt[c * 8 + s2] == p) { if (s2 < s) first = false; mult += (s2 < 4) ? 0.25 : -0.25; }
        if (first && mult != 0.0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double jk = mult * u[k];
            unsafeAtomicAdd(&gp[(long)k * lay.Ppad + p], jk * er);
            unsafeAtomicAdd(&cp.cdiag[(long)k * lay.Ppad + p], jk * jk);
          }
        }
      }
    }
  }
  if (bad) flags[0] = 1;
  const double tot = block_sum(acc, sh_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// || J v ||^2 share of the constraint rows for one or two vectors: partial[b][0..2] = sum r1^2, r1 r2, r2^2
template <int NV>
__global__ void __launch_bounds__(BLOCK)
k_con_jv(ConPlan cp, VecLayout lay, const double* __restrict__ v1, const double* __restrict__ v2, double* __restrict__ partial) {
  __shared__ double sh_red[BLOCK / WAVE];
  double s11 = 0.0, s12 = 0.0, s22 = 0.0;
  for (int c = blockIdx.x * BLOCK + threadIdx.x; c < cp.n_con; c += gridDim.x * BLOCK) {
    double r1 = 0.0, r2 = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int p = cp.pt[c * 8 + s];
      const double q = (s < 4) ? 0.25 : -0.25;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double jk = q * cp.u[c * 3 + k];
        r1 += jk * v1[lay.ncp_pad + (long)k * lay.Ppad + p];
        if (NV == 2) r2 += jk * v2[lay.ncp_pad + (long)k * lay.Ppad + p];
      }
    }
    s11 += r1 * r1; s12 += r1 * r2; s22 += r2 * r2;
  }
  double r;
  r = block_sum(s11, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 0] = r;
  r = block_sum(s12, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 1] = r;
  r = block_sum(s22, sh_red); if (threadIdx.x == 0) partial[blockIdx.x * 4 + 2] = r;
  if (threadIdx.x == 0) partial[blockIdx.x * 4 + 3] = 0.0;
}

// Points of one constraint component whose per-point factors fit the workgroup's LDS copy (one board in one frame: tens).  A larger component — all
// static markers of a room are ONE component, core/capture_volume.py:446-531 — keeps them in global scratch (ConPlan::big) instead: same code, the
// pointers differ.
constexpr int CON_LDS_POINTS = 256;

// L^T x for the 3 x 3 factor in chol3's convention (reciprocal diagonal entries)
__device__ __forceinline__ void chol3_lt_mul(const double* L, const double* x, double* y) {
  y[0] = x[0] / L[0] + L[1] * x[1] + L[3] * x[2];
  y[1] = x[1] / L[2] + L[4] * x[2];
  y[2] = x[2] / L[5];
}

__device__ __forceinline__ bool con_point_factor(const double* __restrict__ Vblk, const double* __restrict__ dp, VecLayout lay, int p,
                                                 double lam, double* L) {
  double Vd[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) Vd[q] = Vblk[(long)q * lay.Ppad + p];
  const double d0 = dp[p], d1 = dp[lay.Ppad + p], d2 = dp[2 * lay.Ppad + p];
  Vd[0] += lam * d0 * d0; Vd[3] += lam * d1 * d1; Vd[5] += lam * d2 * d2;
  if (chol3(Vd, L)) return true;
  L[0] = L[2] = L[5] = 1.0; L[1] = L[3] = L[4] = 0.0;
  return false;
}

// Woodbury correction of the reduced camera system, one workgroup per component:
//   Sacc -= G^T M^-1 G,  bacc -= G^T M^-1 h,   G = J_c V~^-1 W^T (rows: sum_i (T_i z)^T),  h = J_c V~^-1 g_p,
//   z = L_p^-1 (slot coefficient), M = I + Z Z^T factored in place (lower) and kept for k_con_backsub.
// Runs after the T records exist (k_tprep) and after the unconstrained Sacc / bacc are in place.
template <int NC>
__global__ void __launch_bounds__(BLOCK)
k_con_schur(ConPlan cp, VecLayout lay, double lam, const double* __restrict__ Vblk, const double* __restrict__ gvec,
            const double* __restrict__ sinv, const double* __restrict__ Trec, const double* __restrict__ tab, const int* __restrict__ pt_start,
            const int* __restrict__ obs_cam, const int* __restrict__ cam_off, const int* __restrict__ cam_np, int ncp,
            double* __restrict__ Sacc, double* __restrict__ bacc, int* __restrict__ flags) {
  constexpr int REC = SchurRec<NC>::HREC;
  __shared__ double sh_Ly[CON_LDS_POINTS * 9];
  __shared__ double sh_piv;
  const int k = blockIdx.x;
  const int c0 = cp.comp_con[k], m = cp.comp_con[k + 1] - c0;
  const int p0 = cp.comp_pt[k], np = cp.comp_pt[k + 1] - p0;
  double* pL = (np <= CON_LDS_POINTS) ? sh_Ly : cp.big + (long)p0 * 9;  // [np][6] factors, then [np][3] y
  double* py = pL + (long)np * 6;
  double* M = cp.M + cp.comp_m[k];
  const int gw = ncp + 1;
  double* G = cp.G + (long)c0 * gw;
  const double* gp = gvec + lay.ncp_pad;
  const double* dp = sinv + lay.ncp_pad;
  // 1. per point: factor of V~_p, y_p = L^-1 g_p
  for (int lp = threadIdx.x; lp < np; lp += BLOCK) {
    const int p = cp.comp_pts[p0 + lp];
    double L[6];
    if (!con_point_factor(Vblk, dp, lay, p, lam, L)) flags[1] = 1;
    const double g3[3] = {gp[p], gp[lay.Ppad + p], gp[2 * lay.Ppad + p]};
    double y[3];
    chol3_fwd(L, g3, y);
#pragma unroll
    for (int q = 0; q < 6; ++q) pL[lp * 6 + q] = L[q];
    py[lp * 3] = y[0]; py[lp * 3 + 1] = y[1]; py[lp * 3 + 2] = y[2];
  }
  __syncthreads();
  // 2. z for every (constraint, slot); rows of G zeroed
  for (int e = threadIdx.x; e < m * 8; e += BLOCK) {
    const int c = c0 + e / 8, s = e % 8;
    const double q = (s < 4) ? 0.25 : -0.25;
    const double j3[3] = {q * cp.u[c * 3], q * cp.u[c * 3 + 1], q * cp.u[c * 3 + 2]};
    double z[3];
    chol3_fwd(pL + (long)cp.lp[c * 8 + s] * 6, j3, z);
    cp.z[(long)(c * 8 + s) * 3] = z[0]; cp.z[(long)(c * 8 + s) * 3 + 1] = z[1]; cp.z[(long)(c * 8 + s) * 3 + 2] = z[2];
  }
  for (long e = threadIdx.x; e < (long)m * gw; e += BLOCK) G[e] = 0.0;
  __syncthreads();
  // 3. M = I + Z Z^T (lower triangle), h in the last column of G
  for (long e = threadIdx.x; e < (long)m * m; e += BLOCK) {
    const int a = (int)(e / m), b = (int)(e % m);
    if (b > a) continue;
    double acc = (a == b) ? 1.0 : 0.0;
    for (int s = 0; s < 8; ++s) {
      const int la = cp.lp[(c0 + a) * 8 + s];
      const double* za = cp.z + (long)((c0 + a) * 8 + s) * 3;
      for (int s2 = 0; s2 < 8; ++s2)
        if (cp.lp[(c0 + b) * 8 + s2] == la) {
          const double* zb = cp.z + (long)((c0 + b) * 8 + s2) * 3;
          acc += za[0] * zb[0] + za[1] * zb[1] + za[2] * zb[2];
        }
    }
    M[(long)a * m + b] = acc;
  }
  for (int c = threadIdx.x; c < m; c += BLOCK) {
    double h = 0.0;
    for (int s = 0; s < 8; ++s) {
      const double* z = cp.z + (long)((c0 + c) * 8 + s) * 3;
      const double* y = py + (long)cp.lp[(c0 + c) * 8 + s] * 3;
      h += z[0] * y[0] + z[1] * y[1] + z[2] * y[2];
    }
    G[(long)c * gw + ncp] = h;
    // 4. G rows: sum over the observations of the slot points of (T_i z)^T at the camera's columns
    for (int s = 0; s < 8; ++s) {
      const int p = cp.pt[(c0 + c) * 8 + s];
      const double* z = cp.z + (long)((c0 + c) * 8 + s) * 3;
      int hidx = -1;  // heavy point: its per-camera sums are already formed (k_heavy_schur), binary search in the sorted list
      for (int lo = 0, hi = cp.n_heavy - 1; lo <= hi;) {
        const int mid = (lo + hi) >> 1, q = cp.heavy_pts[mid];
        if (q == p) { hidx = mid; break; }
        if (q < p) lo = mid + 1; else hi = mid - 1;
      }
      if (hidx >= 0) {
        const double* Wh = cp.heavy_W + (long)hidx * ncp * 3;
        for (int r = 0; r < ncp; ++r) G[(long)c * gw + r] += Wh[3 * r] * z[0] + Wh[3 * r + 1] * z[1] + Wh[3 * r + 2] * z[2];
        continue;
      }
      for (int i = pt_start[p]; i < pt_start[p + 1]; ++i) {
        const int cam = obs_cam[i];
        double T[3 * NC];
        expand_record<NC>(Trec + (long)i * REC, tab + (long)cam * CAMTAB_DOUBLES + 12, T);
        const int off = cam_off[cam], npar = cam_np[cam];
#pragma unroll
        for (int r = 0; r < NC; ++r)  // (constant trip count: T stays in registers)
          if (r < npar) G[(long)c * gw + off + r] += T[3 * r] * z[0] + T[3 * r + 1] * z[1] + T[3 * r + 2] * z[2];
      }
    }
  }
  __syncthreads();
  // 5. Cholesky of M, left-looking by columns: v_i = M_ij - sum_{t<j} L_it L_jt
  for (int j = 0; j < m; ++j) {
    for (int i = j + threadIdx.x; i < m; i += BLOCK) {
      double v = M[(long)i * m + j];
      for (int t = 0; t < j; ++t) v -= M[(long)i * m + t] * M[(long)j * m + t];
      M[(long)i * m + j] = v;
      if (i == j) sh_piv = v;
    }
    __syncthreads();
    const double piv = sh_piv;
    if (!(piv > 0.0)) { if (threadIdx.x == 0) flags[2] = 1; }
    const double inv = piv > 0.0 ? 1.0 / sqrt(piv) : 1.0;
    for (int i = j + threadIdx.x; i < m; i += BLOCK) M[(long)i * m + j] = (i == j) ? (piv > 0.0 ? sqrt(piv) : 1.0) : M[(long)i * m + j] * inv;
    __syncthreads();
  }
  // 6. Y = L_M^-1 [G | h] in place, one thread per column
  for (int col = threadIdx.x; col < gw; col += BLOCK) {
    for (int c = 0; c < m; ++c) {
      double v = G[(long)c * gw + col];
      for (int t = 0; t < c; ++t) v -= M[(long)c * m + t] * G[(long)t * gw + col];
      G[(long)c * gw + col] = v / M[(long)c * m + c];
    }
  }
  __syncthreads();
  // 7. Sacc -= Y^T Y (upper triangle), bacc -= Y^T y_h
  for (long e = threadIdx.x; e < (long)ncp * gw; e += BLOCK) {
    const int r = (int)(e / gw), c2 = (int)(e % gw);
    if (c2 < r) continue;
    double acc = 0.0;
    for (int c = 0; c < m; ++c) acc += G[(long)c * gw + r] * G[(long)c * gw + c2];
    if (acc != 0.0) {
      if (c2 < ncp) unsafeAtomicAdd(&Sacc[(long)r * ncp + c2], -acc);
      else unsafeAtomicAdd(&bacc[r], -acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Small components (round 6): the rows of ONE board in ONE frame — 35 rows on 12 points in the reference's own sessions (core/constraints.py:68-81,
// docs/scripting.md:168-174).  k_con_schur above walks global memory (z, the slot tables, M, G) and factors M column by column with two barriers per
// column: 215-238 us per launch, 41 % of the kernel time of an optimize() call on that session.  Here everything of a component is a dense block in
// LDS and every step is a small matrix product of all 256 threads:
//     Z  (m x 3 np)   Z[c][3 p + k] = sum over the slots of c on point p of  L_p^-1 (+-1/4 u_c)      (a repeated point collects its slots)
//     M  = I + Z Z^T,   h = Z y   (y = L^-1 g_p stacked),   Wst (3 np x ncp) = the points' sum_i T_i^T at their cameras' columns,   G = Z Wst
//     M  = L L^T by 32-pivot blocks (chol_factor_block: the dense solve's wave factorisation, which also leaves X = L^-1 of a block), the inverse of
//          the whole factor assembled from the blocks' inverses (m <= 64: two blocks);   Y = L^-1 [G | h] = X [G | h]: a product, not a substitution
//     Sacc -= Y^T Y,  bacc -= Y^T y_h   as before.
// X goes to global memory in M's place: k_con_backsub_small needs w = M^-1 u = X^T (X u), two matrix-vector products instead of 2 m barrier pairs.
constexpr int CON_SMALL_M = 2 * NB;
struct ConSmallLayout {  // offsets in doubles into the dynamic LDS; ZL / GW / ML: odd row strides
  int ZL, GW, ML, oL, oY, oZ, oW, oG, oM, oX, oT, oR, total;
  __host__ __device__ ConSmallLayout(int m, int np, int ncp) {
    ZL = (3 * np) | 1; GW = (ncp + 1) | 1; ML = m | 1;
    oL = 2 * NB * (NB + 1); oY = oL + 6 * np; oZ = oY + 3 * np; oW = oZ + m * ZL; oG = oW + 3 * np * GW; oM = oG + m * GW; oX = oM + m * ML;
    oT = oX + m * ML; oR = oT + NB * NB;  // a block of scratch for the panel / inverse products, and Y = X [G | h]
    total = oR + m * GW;
  }
};
template <int NC>
__global__ void __launch_bounds__(BLOCK)
k_con_schur_small(ConPlan cp, VecLayout lay, double lam, const double* __restrict__ Vblk, const double* __restrict__ gvec,
                  const double* __restrict__ sinv, const double* __restrict__ Trec, const double* __restrict__ tab, const int* __restrict__ pt_start,
                  const int* __restrict__ obs_cam, const int* __restrict__ cam_off, const int* __restrict__ cam_np, int ncp,
                  double* __restrict__ Sacc, double* __restrict__ bacc, int* __restrict__ flags) {
  constexpr int REC = SchurRec<NC>::HREC;
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int c0 = cp.comp_con[k], m = cp.comp_con[k + 1] - c0;
  const int p0 = cp.comp_pt[k], np = cp.comp_pt[k + 1] - p0;
  const ConSmallLayout lo(cp.max_m, cp.max_np, ncp);
  double (*sh_D)[NB + 1] = reinterpret_cast<double (*)[NB + 1]>(sh);
  double *pL = sh + lo.oL, *py = sh + lo.oY, *Z = sh + lo.oZ, *Wst = sh + lo.oW, *G = sh + lo.oG, *M = sh + lo.oM, *X = sh + lo.oX, *tmp = sh + lo.oT;
  const int ZL = lo.ZL, GW = lo.GW, ML = lo.ML, gw = ncp + 1, nz = 3 * np;
  const double* gp = gvec + lay.ncp_pad;
  const double* dp = sinv + lay.ncp_pad;
  // 1. per point: factor of V~_p, y_p = L^-1 g_p; zero Z, Wst
  for (int lp = tid; lp < np; lp += BLOCK) {
    const int p = cp.comp_pts[p0 + lp];
    double L[6];
    if (!con_point_factor(Vblk, dp, lay, p, lam, L)) flags[1] = 1;
    const double g3[3] = {gp[p], gp[lay.Ppad + p], gp[2 * lay.Ppad + p]};
    double y[3];
    chol3_fwd(L, g3, y);
#pragma unroll
    for (int q = 0; q < 6; ++q) pL[lp * 6 + q] = L[q];
    py[lp * 3] = y[0]; py[lp * 3 + 1] = y[1]; py[lp * 3 + 2] = y[2];
  }
  for (int e = tid; e < m * ZL; e += BLOCK) Z[e] = 0.0;
  for (int e = tid; e < nz * GW; e += BLOCK) Wst[e] = 0.0;
  __syncthreads();
  // 2. z of every (constraint, slot) — kept in global memory for k_con_backsub_small — summed into Z; the points' T^T blocks into Wst
  for (int e = tid; e < m * 8; e += BLOCK) {
    const int c = c0 + e / 8, sl = e % 8;
    const double q = (sl < 4) ? 0.25 : -0.25;
    const double j3[3] = {q * cp.u[c * 3], q * cp.u[c * 3 + 1], q * cp.u[c * 3 + 2]};
    const int lp = cp.lp[c * 8 + sl];
    double z[3];
    chol3_fwd(pL + lp * 6, j3, z);
    cp.z[(long)(c * 8 + sl) * 3] = z[0]; cp.z[(long)(c * 8 + sl) * 3 + 1] = z[1]; cp.z[(long)(c * 8 + sl) * 3 + 2] = z[2];
    double* zr = Z + (e / 8) * ZL + 3 * lp;
    lds_add(&zr[0], z[0]); lds_add(&zr[1], z[1]); lds_add(&zr[2], z[2]);
  }
  // (point, observation) pairs: sixteen observation lanes per point at a time
  for (int lp = tid / 16; lp < np; lp += BLOCK / 16) {
    const int p = cp.comp_pts[p0 + lp];
    for (int i = pt_start[p] + (tid % 16); i < pt_start[p + 1]; i += 16) {
      const int cam = obs_cam[i];
      double T[3 * NC];
      expand_record<NC>(Trec + (long)i * REC, tab + (long)cam * CAMTAB_DOUBLES + 12, T);
      const int off = cam_off[cam], npar = cam_np[cam];
#pragma unroll
      for (int r = 0; r < NC; ++r)  // (constant trip count: T stays in registers)
        if (r < npar) {
          lds_add(&Wst[(3 * lp + 0) * GW + off + r], T[3 * r]);
          lds_add(&Wst[(3 * lp + 1) * GW + off + r], T[3 * r + 1]);
          lds_add(&Wst[(3 * lp + 2) * GW + off + r], T[3 * r + 2]);
        }
    }
  }
  __syncthreads();
  // 3. M = I + Z Z^T (both triangles), G = Z Wst, h = Z y in G's last column
  for (int e = tid; e < m * m; e += BLOCK) {
    const int a = e / m, b = e % m;
    const double *za = Z + a * ZL, *zb = Z + b * ZL;
    double acc = (a == b) ? 1.0 : 0.0;
    for (int q = 0; q < nz; ++q) acc = fma(za[q], zb[q], acc);
    M[a * ML + b] = acc;
  }
  for (int e = tid; e < m * gw; e += BLOCK) {
    const int c = e / gw, col = e % gw;
    const double* zc = Z + c * ZL;
    double acc = 0.0;
    if (col < ncp) for (int q = 0; q < nz; ++q) acc = fma(zc[q], Wst[q * GW + col], acc);
    else for (int q = 0; q < nz; ++q) acc = fma(zc[q], py[q], acc);
    G[c * GW + col] = acc;
  }
  for (int e = tid; e < m * m; e += BLOCK) X[(e / m) * ML + e % m] = 0.0;
  __syncthreads();
  // 4. Cholesky of M by 32-pivot blocks, X = L^-1 assembled block by block: X_bb = X_b;  X_bj = -X_b sum_{j <= t < b} L_bt X_tj  (j < b)
  const int nblk = (m + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    const int r0 = b * NB, rc = min(NB, m - r0);
    // D_b = M_bb - sum_{t < b} L_bt L_bt^T (L_bt kept in M's lower blocks)
    for (int e = tid; e < NB * NB; e += BLOCK) {
      const int i = e / NB, j = e % NB;
      double v = (i == j) ? 1.0 : 0.0;
      if (i < rc && j < rc) {
        v = M[(r0 + i) * ML + r0 + j];
        for (int t = 0; t < r0; ++t) v = fma(-M[(r0 + i) * ML + t], M[(r0 + j) * ML + t], v);
      }
      sh_D[i][j] = v;
    }
    __syncthreads();
    if (tid < WAVE) chol_factor_block(sh_D, rc, flags);
    __syncthreads();
    // L_bb and X_bb out of the factoring buffer (row NB + c of sh_D holds column c of X_b)
    for (int e = tid; e < rc * rc; e += BLOCK) {
      const int i = e / rc, j = e % rc;
      M[(r0 + i) * ML + r0 + j] = (j <= i) ? sh_D[i][j] : 0.0;
      X[(r0 + i) * ML + r0 + j] = sh_D[NB + j][i];
    }
    __syncthreads();
    // panel below: L_ib = (M_ib - sum_{t < b} L_it L_bt^T) X_b^T for the rows i behind this block
    const int rest = m - (r0 + rc);
    for (int e = tid; e < rest * rc; e += BLOCK) {
      const int i = r0 + rc + e / rc, j = e % rc;  // entry (i, r0 + j) of the panel, before the multiplication with X_b^T
      double v = M[i * ML + r0 + j];
      for (int t = 0; t < r0; ++t) v = fma(-M[i * ML + t], M[(r0 + j) * ML + t], v);
      tmp[(e / rc) * rc + j] = v;
    }
    __syncthreads();
    for (int e = tid; e < rest * rc; e += BLOCK) {
      const int i = r0 + rc + e / rc, j = e % rc;
      double v = 0.0;
      for (int t = 0; t <= j; ++t) v = fma(tmp[(e / rc) * rc + t], X[(r0 + j) * ML + r0 + t], v);  // (U X_b^T)_ij = sum_t U_it X_jt, X lower triangular
      M[i * ML + r0 + j] = v;
    }
    __syncthreads();
    // X_bj for the column blocks j < b:  -X_b (sum_{t} L_bt X_tj), t over the rows before this block
    for (int e = tid; e < rc * r0; e += BLOCK) {
      const int i = e / r0, j = e % r0;
      double v = 0.0;
      for (int t = j; t < r0; ++t) v = fma(M[(r0 + i) * ML + t], X[t * ML + j], v);
      tmp[i * r0 + j] = v;
    }
    __syncthreads();
    for (int e = tid; e < rc * r0; e += BLOCK) {
      const int i = e / r0, j = e % r0;
      double v = 0.0;
      for (int t = 0; t <= i; ++t) v = fma(X[(r0 + i) * ML + r0 + t], tmp[t * r0 + j], v);
      X[(r0 + i) * ML + j] = -v;
    }
    __syncthreads();
  }
  // 5. X to global memory in M's place (k_con_backsub_small), Y = X [G | h], then the update
  double* Mg = cp.M + cp.comp_m[k];
  for (int e = tid; e < m * m; e += BLOCK) Mg[e] = X[(e / m) * ML + e % m];
  double* Y = sh + lo.oR;  // m x GW
  for (int e = tid; e < m * gw; e += BLOCK) {
    const int c = e / gw, col = e % gw;
    double acc = 0.0;
    for (int t = 0; t <= c; ++t) acc = fma(X[c * ML + t], G[t * GW + col], acc);
    Y[c * GW + col] = acc;
  }
  __syncthreads();
  for (int e = tid; e < ncp * gw; e += BLOCK) {
    const int r = e / gw, c2 = e % gw;
    if (c2 < r) continue;
    double acc = 0.0;
    for (int c = 0; c < m; ++c) acc = fma(Y[c * GW + r], Y[c * GW + c2], acc);
    if (acc != 0.0) {
      if (c2 < ncp) unsafeAtomicAdd(&Sacc[(long)r * ncp + c2], -acc);
      else unsafeAtomicAdd(&bacc[r], -acc);
    }
  }
}

// point steps of a small component: as k_con_backsub, with w = M^-1 u = X^T (X u) from the inverse factor k_con_schur_small left in M's place
__global__ void __launch_bounds__(BLOCK)
k_con_backsub_small(ConPlan cp, VecLayout lay, double lam, const double* __restrict__ Vblk, const double* __restrict__ sinv, double* __restrict__ svec) {
  extern __shared__ __attribute__((aligned(16))) double sh[];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int c0 = cp.comp_con[k], m = cp.comp_con[k + 1] - c0;
  const int p0 = cp.comp_pt[k], np = cp.comp_pt[k + 1] - p0;
  double *pL = sh, *pq = pL + 6 * cp.max_np, *u = pq + 3 * cp.max_np, *t1 = u + cp.max_m;  // [np][6], [np][3], [m], [m]
  const double* Xg = cp.M + cp.comp_m[k];
  const double* dp = sinv + lay.ncp_pad;
  double* sp = svec + lay.ncp_pad;
  for (int lp = tid; lp < np; lp += BLOCK) {
    const int p = cp.comp_pts[p0 + lp];
    double L[6];
    con_point_factor(Vblk, dp, lay, p, lam, L);
    const double d0[3] = {-sp[p], -sp[lay.Ppad + p], -sp[2 * lay.Ppad + p]};
    double yq[3];
    chol3_lt_mul(L, d0, yq);
#pragma unroll
    for (int q = 0; q < 6; ++q) pL[lp * 6 + q] = L[q];
    pq[lp * 3] = yq[0]; pq[lp * 3 + 1] = yq[1]; pq[lp * 3 + 2] = yq[2];
  }
  __syncthreads();
  for (int c = tid; c < m; c += BLOCK) {
    double acc = 0.0;
    for (int sl = 0; sl < 8; ++sl) {
      const double* z = cp.z + (long)((c0 + c) * 8 + sl) * 3;
      const double* y = pq + (long)cp.lp[(c0 + c) * 8 + sl] * 3;
      acc += z[0] * y[0] + z[1] * y[1] + z[2] * y[2];
    }
    u[c] = acc;
  }
  __syncthreads();
  for (int c = tid; c < m; c += BLOCK) {  // t1 = X u (X lower triangular)
    double acc = 0.0;
    for (int t = 0; t <= c; ++t) acc = fma(Xg[(long)c * m + t], u[t], acc);
    t1[c] = acc;
  }
  __syncthreads();
  for (int c = tid; c < m; c += BLOCK) {  // w = X^T t1
    double acc = 0.0;
    for (int t = c; t < m; ++t) acc = fma(Xg[(long)t * m + c], t1[t], acc);
    u[c] = acc;
  }
  for (int lp = tid; lp < np; lp += BLOCK) { pq[lp * 3] = 0.0; pq[lp * 3 + 1] = 0.0; pq[lp * 3 + 2] = 0.0; }
  __syncthreads();
  for (int e = tid; e < m * 8; e += BLOCK) {
    const int c = e / 8, sl = e % 8;
    const double* z = cp.z + (long)((c0 + c) * 8 + sl) * 3;
    const double w = u[c];
    double* a = pq + (long)cp.lp[(c0 + c) * 8 + sl] * 3;
    lds_add(&a[0], z[0] * w); lds_add(&a[1], z[1] * w); lds_add(&a[2], z[2] * w);
  }
  __syncthreads();
  for (int lp = tid; lp < np; lp += BLOCK) {
    const int p = cp.comp_pts[p0 + lp];
    double t[3];
    chol3_bwd(pL + lp * 6, pq + lp * 3, t);
    sp[p] += t[0]; sp[lay.Ppad + p] += t[1]; sp[2 * lay.Ppad + p] += t[2];
  }
}

// Point steps of a component:  dp += V~^-1 J_c^T M^-1 (J_c V~^-1 q)  on top of the unconstrained dp0 = -V~^-1 q
// that k_backsub left in svec (L^-1 q = -L^T dp0 is recovered from it).
__global__ void __launch_bounds__(BLOCK)
k_con_backsub(ConPlan cp, VecLayout lay, double lam, const double* __restrict__ Vblk, const double* __restrict__ sinv,
              double* __restrict__ svec) {
  __shared__ double sh_Lq[CON_LDS_POINTS * 9];
  __shared__ double sh_w;
  const int k = blockIdx.x;
  const int c0 = cp.comp_con[k], m = cp.comp_con[k + 1] - c0;
  const int p0 = cp.comp_pt[k], np = cp.comp_pt[k + 1] - p0;
  double* pL = (np <= CON_LDS_POINTS) ? sh_Lq : cp.big + (long)p0 * 9;  // [np][6] factors, then [np][3] q (k_con_schur's layout)
  double* pq = pL + (long)np * 6;
  const double* M = cp.M + cp.comp_m[k];
  double* u = cp.w + c0;
  const double* dp = sinv + lay.ncp_pad;
  double* sp = svec + lay.ncp_pad;
  for (int lp = threadIdx.x; lp < np; lp += BLOCK) {
    const int p = cp.comp_pts[p0 + lp];
    double L[6];
    con_point_factor(Vblk, dp, lay, p, lam, L);
    const double d0[3] = {-sp[p], -sp[lay.Ppad + p], -sp[2 * lay.Ppad + p]};
    double yq[3];
    chol3_lt_mul(L, d0, yq);
#pragma unroll
    for (int q = 0; q < 6; ++q) pL[lp * 6 + q] = L[q];
    pq[lp * 3] = yq[0]; pq[lp * 3 + 1] = yq[1]; pq[lp * 3 + 2] = yq[2];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < m; c += BLOCK) {
    double acc = 0.0;
    for (int s = 0; s < 8; ++s) {
      const double* z = cp.z + (long)((c0 + c) * 8 + s) * 3;
      const double* y = pq + (long)cp.lp[(c0 + c) * 8 + s] * 3;
      acc += z[0] * y[0] + z[1] * y[1] + z[2] * y[2];
    }
    u[c] = acc;
  }
  __syncthreads();
  // M w = u with the factor left by k_con_schur: forward then backward, column oriented
  for (int j = 0; j < m; ++j) {
    if (threadIdx.x == 0) { const double w = u[j] / M[(long)j * m + j]; u[j] = w; sh_w = w; }
    __syncthreads();
    const double w = sh_w;
    for (int i = j + 1 + threadIdx.x; i < m; i += BLOCK) u[i] -= M[(long)i * m + j] * w;
    __syncthreads();
  }
  for (int j = m - 1; j >= 0; --j) {
    if (threadIdx.x == 0) { const double w = u[j] / M[(long)j * m + j]; u[j] = w; sh_w = w; }
    __syncthreads();
    const double w = sh_w;
    for (int i = threadIdx.x; i < j; i += BLOCK) u[i] -= M[(long)j * m + i] * w;
    __syncthreads();
  }
  // scatter: acc_p = sum_{(c, s) -> p} z w_c, then dp_p += L_p^-T acc_p
  for (int lp = threadIdx.x; lp < np; lp += BLOCK) { pq[lp * 3] = 0.0; pq[lp * 3 + 1] = 0.0; pq[lp * 3 + 2] = 0.0; }
  __syncthreads();
  for (int e = threadIdx.x; e < m * 8; e += BLOCK) {
    const int c = e / 8, s = e % 8;
    const double* z = cp.z + (long)((c0 + c) * 8 + s) * 3;
    const double w = u[c];
    double* a = pq + (long)cp.lp[(c0 + c) * 8 + s] * 3;  // (LDS or global: a flat atomic either way)
    lds_add(&a[0], z[0] * w); lds_add(&a[1], z[1] * w); lds_add(&a[2], z[2] * w);
  }
  __syncthreads();
  for (int lp = threadIdx.x; lp < np; lp += BLOCK) {
    const int p = cp.comp_pts[p0 + lp];
    double t[3];
    chol3_bwd(pL + lp * 6, pq + lp * 3, t);
    sp[p] += t[0]; sp[lay.Ppad + p] += t[1]; sp[2 * lay.Ppad + p] += t[2];
  }
}

}  // namespace cba
