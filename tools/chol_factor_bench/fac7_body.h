// Factor the NB x NB block parked in `D` (LDS, row stride NB + 1, `nb` live rows, identity-padded) with wave 0, TWO pivots at a time.
// Left-looking by column pairs (j, j + 1); lane r keeps row r of L in registers (lanes 32..63: column c of X = L^-1, same instruction stream).
//   v0_r = D_rj    - sum_{t<j} L_rt L_jt        v1_r = D_r,j+1 - sum_{t<j} L_rt L_j+1,t
//   pivot block [a b; b c] = [v0_j v0_j+1; . v1_j+1]:  s0 = rsqrt(a), s1 = rsqrt(a c - b^2)  — two INDEPENDENT chains —
//   L_rj = v0_r s0,   L_r,j+1 = (v1_r - q v0_r) i11   with q = b s0^2 = b / a,  i11 = a s0 s1 = 1 / L_j+1,j+1
// (one serial rsqrt per pivot was 215 ns x 32; a pair costs one rsqrt latency instead of two).  The X lanes run the same formulas with
// the lane's own X_tc in the place of L_rt and delta in the place of D (X_jc = (delta_jc - sum_t L_jt X_tc) / L_jj is the row recurrence).
__device__ __forceinline__ void chol_factor_block(double (*D)[NB + 1], int nb, double* __restrict__ out, int ldw,
                                                  int* __restrict__ flags, double* __restrict__ xinv) {
  int lane = threadIdx.x;
  asm volatile("" : "+v"(lane));  // (keeps a caller's loop from hoisting the lane-dependent constants of all 16 pairs out of it)
  const int r = lane & (NB - 1);
  const bool isX = lane >= NB;
  double row[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) row[c] = isX ? (c == r ? 1.0 : 0.0) : D[r][c];
  bool bad = false;
  double q0 = 0.0, q1 = 0.0;                            // sum_{t < j-2} L_rt L_jt and ... L_j+1,t: accumulated during the previous pair
  double n00 = 0.0, n01 = 0.0, n10 = 0.0, n11 = 0.0;    // the four newest entries L_j,j-2  L_j,j-1  L_j+1,j-2  L_j+1,j-1 (by readlane)
#pragma unroll
  for (int j = 0; j < NB; j += 2) {
    // look-ahead, off the pivot chain: the sums of the NEXT pair over the columns that are final already (t < j); rows j + 2, j + 3 of
    // L are read from LDS as broadcasts and consumed at once (holding two pairs of rows in registers spilled 190 of them)
    double la0[2] = {0.0, 0.0}, la1[2] = {0.0, 0.0};
    if (j + 2 < NB) {
#pragma unroll
      for (int t0 = 0; t0 < j; t0 += 8) {  // eight columns at a time: unbounded, the scheduler hoists all 2 j loads and spills
#pragma unroll
        for (int t = t0; t < t0 + 8 && t < j; ++t) {
          la0[t & 1] = fma(row[t], D[(j + 2) & (NB - 1)][t], la0[t & 1]);
          la1[t & 1] = fma(row[t], D[(j + 3) & (NB - 1)][t], la1[t & 1]);
        }
      }
    }
    double v0 = row[j] - q0, v1 = row[j + 1] - q1;
    if (j >= 2) {  // the two newest columns
      v0 = fma(-row[j - 1], n01, fma(-row[j - 2], n00, v0));
      v1 = fma(-row[j - 1], n11, fma(-row[j - 2], n10, v1));
    }
    double a = readlane_f64(v0, j), b = readlane_f64(v0, j + 1), c = readlane_f64(v1, j + 1);
    double det = fma(a, c, -(b * b));
    const bool okp = (a > 0.0) && isfinite(a) && (det > 0.0) && isfinite(det);
    if (!okp && j < nb) bad = true;
    if (!okp) { a = 1.0; b = 0.0; det = 1.0; }
    const double s0 = fast_rsqrt(a), s1 = fast_rsqrt(det);
    const double q = b * s0 * s0, i11 = a * s0 * s1;
    double l0 = v0 * s0, l1 = (v1 - q * v0) * i11;
    if (!isX) {  // the strict upper triangle of L is zero (selects on values: no divergent branch on the pivot chain)
      l0 = (r < j) ? 0.0 : l0;
      l1 = (r < j + 1) ? 0.0 : l1;
    }
    row[j] = l0; row[j + 1] = l1;
    D[lane][j] = l0; D[lane][j + 1] = l1;  // rows NB.. of D take the X lanes' values (never read)
    if (j + 2 < NB) { n00 = readlane_f64(l0, j + 2); n01 = readlane_f64(l1, j + 2); }
    if (j + 3 < NB) { n10 = readlane_f64(l0, j + 3); n11 = readlane_f64(l1, j + 3); }
    q0 = la0[0] + la0[1]; q1 = la1[0] + la1[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (bad && lane == 0) flags[2] = 1;
  if (!isX) {
    if (lane < nb) {
#pragma unroll
      for (int c = 0; c < NB; ++c)
        if (c <= lane) out[(long)lane * ldw + c] = row[c];
    }
  } else {  // X (NB x NB, row-major, identity-padded beyond nb): lane 32 + c holds column c
#pragma unroll
    for (int t = 0; t < NB; ++t) xinv[t * NB + r] = row[t];
  }
}
