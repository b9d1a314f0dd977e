// Factor the NB x NB block parked in `D` (LDS, row stride NB + 1, `nb` live rows, identity-padded, BOTH triangles) with wave 0, as a
// 2 x 2 arrangement of 16 x 16 blocks:
//   1. L00, X00 = L00^-1        16 pivots (chol_factor_sub: lanes 0..15 keep a row of L, lanes 16..31 a column of X, same instructions)
//   2. L10 = D10 X00^T          four v_mfma_f64_16x16x4
//   3. D11 -= L10 L10^T         four
//   4. L11, X11                 16 pivots
//   5. X10 = -X11 (L10 X00)     eight
// A pivot of the 32-wide version issued ~80 instructions (a 31-term dot product and 32 broadcast reads), 525 cycles, and the single wave
// is bound by what it issues in order (7.0 us per block); a 16-wide pivot issues about half.
// `T` (NB x NB, row stride NB + 1) is scratch and ends up holding X.
__device__ long long g_clk[8];
#define STAMP(i) do { if (threadIdx.x == 0) g_clk[i] = clock64(); } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int HB = NB / 2;

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Left-looking by columns over the B x B diagonal block at (off, off):  v_r = D_rj - sum_{t<j} L_rt L_jt, L_jj = sqrt(v_j),
// L_rj = v_r / L_jj.  Row j of L is read back from LDS as broadcasts (every lane stores its new entry at pivot j; a wave executes its
// DS instructions in order), prefetched one pivot ahead; the newest entry L_j,j-1 travels by v_readlane.  The X lanes (B .. 2B-1) run the
// row recurrence X_jc = (delta_jc - sum_{t<j} L_jt X_tc) / L_jj with their own X_tc in the place of L_rt.  Lanes 2B.. repeat the first
// 2B; whatever is not a row of L is stored into the scratch rows NB.. of D.
// A pivot is a chain of DEPENDENT FP64 instructions (~20 cycles each), so the chain is kept short:
//   * the pivot d_j = a_j - L_j,j-1^2 is formed by lane j from its own registers (no broadcast of L_j,j-1 in front of it);
//   * 1/sqrt(d) is v_rsq_f64 and ONE third-order correction folded into the product with the lane's value:
//       u = v y0, e = 1 - (d y0) y0, L_rj = u + (u e)(1/2 + 3/8 e)            (rsq, d y0, e, u e, fma: five levels);
//   * the checks of the pivot (positive, finite) and the zeros above the diagonal are applied beside the chain.  A failed pivot
//     raises `bad` and leaves NaN/Inf in the factor: the caller discards the factorisation.
template <int B>
__device__ __forceinline__ void chol_factor_sub(double (*D)[NB + 1], int off, int nb, int lane, int& bad, double (&row)[B]) {
  asm volatile("" : "+v"(lane));  // the lane masks of this stage's pivots are formed here, not together with another stage's (SGPR budget)
  const int r = lane & (B - 1);
  const bool isX = (lane & B) != 0;
  const int srow = lane < B ? off + lane : NB + (lane & (NB - 1));
#pragma unroll
  for (int c = 0; c < B; ++c) row[c] = isX ? (c == r ? 1.0 : 0.0) : D[off + r][off + c];
  double s_prev = 0.0, raw_prev = 0.0;  // L_j,j-1 (broadcast) and this lane's own unmasked entry of column j - 1
  double asum = row[0];                 // D_rj - sum_{t<j-1} L_rt L_jt of the pivot at hand, formed DURING the previous pivot's chain
#pragma unroll
  for (int j = 0; j < B; ++j) {
    double nxt[B];  // row j + 1 of L, entries t < j (final since pivot j - 1)
#pragma unroll
    for (int t = 0; t < B; ++t) nxt[t] = (j + 1 < B && t < j) ? D[off + ((j + 1) & (B - 1))][off + t] : 0.0;
    __builtin_amdgcn_sched_barrier(0);  // the reads go first
    const double d = readlane_f64(fma(-raw_prev, raw_prev, asum), j);        // lane j: its own L_j,j-1 twice
    const double acc = (j >= 1) ? fma(-row[j - 1], s_prev, asum) : asum;     // every lane (lane j: bit-identical to d)
    const double y0 = __builtin_amdgcn_rsq(d);
    const double u = acc * y0;
    const double e = fma(-(d * y0), y0, 1.0);
    const double raw = fma(u * e, fma(0.375, e, 0.5), u);
    // independent of the chain above and issued between its instructions: the next pivot's sum over the columns that are final
    double a[4] = {j + 1 < B ? row[(j + 1) & (B - 1)] : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < j; ++t) a[t & 3] -= row[t] * nxt[t];
    asum = (a[0] + a[1]) + (a[2] + a[3]);
    if (j + 1 < B) s_prev = readlane_f64(raw, j + 1);
    raw_prev = raw;
    const double l = (!isX && r < j) ? 0.0 : raw;
    row[j] = l;
    D[srow][off + j] = l;
    if (off + j < nb && (!(d > 0.0) || !isfinite(d))) bad = 1;
    asm volatile("" : "+v"(bad));  // settled per pivot (left alone, the compiler keeps every pivot's comparison masks to the end and spills them)
    wave_sync_lds();
  }
}

__device__ __forceinline__ void chol_factor_block(double (*D)[NB + 1], double (*T)[NB + 1], int nb, double* __restrict__ out, int ldw,
                                                  int* __restrict__ flags, double* __restrict__ xinv) {
  int lane = threadIdx.x;
  asm volatile("" : "+v"(lane));  // (a caller's loop must not hoist the lane-dependent masks of all pivots out of it: they do not fit the SGPRs)
  asm volatile("" : "+s"(nb));
  const int lr = lane & 15, kq = lane >> 4;
  const bool x_lane = (lane & 48) == HB;  // lanes 16..31
  int bad = 0;
  double row[HB];
  STAMP(0);
  chol_factor_sub<HB>(D, 0, nb, lane, bad, row);
  STAMP(1);
  if (x_lane) {
#pragma unroll
    for (int t = 0; t < HB; ++t) T[t][lr] = row[t];  // X00, row-major
  }
  wave_sync_lds();
  {  // L10 = D10 X00^T  (operand lane: row lane & 15, depth 4 t + (lane >> 4); result lane: rows (lane >> 4) + 4 r, column lane & 15)
    v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < HB / 4; ++t) {
      const int q = 4 * t + kq;
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(D[HB + lr][q], T[lr][q], c, 0, 0, 0);
    }
    wave_sync_lds();
#pragma unroll
    for (int r = 0; r < 4; ++r) D[HB + kq + 4 * r][lr] = c[r];
  }
  wave_sync_lds();
  {  // D11 -= L10 L10^T
    v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < HB / 4; ++t) {
      const double l = D[HB + lr][4 * t + kq];
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(l, l, c, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) D[HB + kq + 4 * r][HB + lr] -= c[r];
  }
  wave_sync_lds();
  STAMP(2);
  chol_factor_sub<HB>(D, HB, nb, lane, bad, row);
  STAMP(3);
  if (x_lane) {
#pragma unroll
    for (int t = 0; t < HB; ++t) T[HB + t][HB + lr] = row[t];  // X11
  }
  wave_sync_lds();
  {  // M = L10 X00 into the upper right quadrant of T, X10 = -X11 M
    v4f64 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < HB / 4; ++t) {
      const int q = 4 * t + kq;
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(D[HB + lr][q], T[q][lr], c, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) T[kq + 4 * r][HB + lr] = c[r];
    wave_sync_lds();
    v4f64 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < HB / 4; ++t) {
      const int q = 4 * t + kq;
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(T[HB + lr][HB + q], T[q][HB + lr], x, 0, 0, 0);
    }
    wave_sync_lds();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T[HB + kq + 4 * r][lr] = -x[r];
      T[kq + 4 * r][HB + lr] = 0.0;  // X01
    }
  }
  wave_sync_lds();
  STAMP(4);
  if (bad && lane == 0) flags[2] = 1;
}

// every thread of the workgroup, behind a barrier: L (lower triangle, live rows) and X = L^-1 from LDS to the work matrix / the inverse array
__device__ __forceinline__ void chol_factor_store(double (*D)[NB + 1], double (*T)[NB + 1], int nb, double* __restrict__ out, int ldw,
                                                  double* __restrict__ xinv, int tid, int nthreads) {
  for (int e = tid; e < NB * NB; e += nthreads) {
    const int i = e / NB, c = e % NB;
    const double l = D[i][c], x = T[i][c];
    if (i < nb && c <= i) out[(long)i * ldw + c] = l;
    xinv[e] = x;
  }
}
