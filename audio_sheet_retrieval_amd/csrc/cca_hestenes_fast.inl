// Device-only fast path of cca_hestenes (cca_solve.inl): the same one-sided Jacobi sweeps with ONE barrier per
// round instead of three.  The first 256 threads work: 16 lanes per column pair; lane `ln` owns rows ln and ln+16
// of W and V, the three dot products are reduced with xor-shuffles inside the 16-lane group, every lane derives the
// same (c, s) and rotates its own rows.  Pairs of one round touch disjoint columns, so no intra-round hazards.
// Requires blockDim.x >= 256, CCA_DIM == 32.  Returns the number of sweeps.
#ifndef ASR_CCA_WAVE
#define ASR_CCA_WAVE 1
#endif
// The same iteration on ONE wave with the matrices in registers: no workgroup barrier and no LDS round trip per round
// (the 31 x ~10 rounds of a decomposition were ~1.2 us each - barrier, LDS reads, shuffle reductions, LDS writes; an
// eigen-decomposition took 0.35-0.45 ms and the training step runs two of them back to back).  Lane 2c + h holds rows
// 16h .. 16h + 15 of column c of W and of V.  Round r pairs column i < 31 with (r - i) mod 31 (with column 31 where
// that is i itself) - every lane fetches its partner's 2 x 16 values with lane shuffles, both lanes of a pair form the
// same three dot products in the same order (bitwise equal), derive the same rotation and update their OWN column.
// A sweep ends with a wave ballot instead of a flag in LDS.  Requires blockDim.x >= 64 and CCA_DIM == 32; all threads
// of the workgroup must call it.  Returns the number of sweeps.
// the iteration itself, run by ONE wave (lane = 0..63) on row-major Wm, Vm in LDS; no workgroup barrier inside
// WITH_V = false (round 6; the training step's decompositions): V is not carried.  For a symmetric positive definite M
// the iteration ends with W = M V = V diag(lambda), so the eigenvectors are the columns of W divided by their norms
// (eigh_spd does that): the partner's half column of V is not fetched (32 of the 64 lane shuffles of a round) and not
// rotated (32 of its 112 float64 multiply-adds).  The division costs eps * ||M|| / lambda_min in accuracy - every matrix
// decomposed there carries a +1e-3 regulariser on its diagonal against eigenvalues of order one: ~1e-13.
template <bool WITH_V = true>
__device__ inline int cca_hestenes_wave_on(double *Wm, double *Vm, int lane) {
    const int N = CCA_DIM;
    const double eps = 1e-15;
    int sweep = 0;
    {
        const int tid = lane;
        const int col = tid >> 1, half = tid & 1;
        double w[16], v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            w[i] = Wm[(half * 16 + i) * N + col];
            v[i] = WITH_V ? Vm[(half * 16 + i) * N + col] : 0.0;
        }
        for (; sweep < 40; ++sweep) {
            bool rotated = false;
            bool big = false;                                              // some ga^2 / (al be) of this sweep was above 1e-16
#pragma unroll 1
            for (int r = 0; r < N - 1; ++r) {
                int pc;
                if (col == N - 1) pc = (r * 16) % (N - 1);                 // the i with 2 i = r (mod 31)
                else {
                    pc = r - col;
                    pc += pc < 0 ? N - 1 : 0;
                    if (pc == col) pc = N - 1;
                }
                const int pl = 2 * pc + half;
                double pw[16], pv[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    pw[i] = __shfl(w[i], pl);
                    if (WITH_V) pv[i] = __shfl(v[i], pl);
                }
                double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    al += w[i] * w[i];
                    be += pw[i] * pw[i];
                    ga += w[i] * pw[i];
                }
                al += __shfl_xor(al, 1);
                be += __shfl_xor(be, 1);
                ga += __shfl_xor(ga, 1);
                // |ga| > eps sqrt(al be), without the square root
                const double g2 = ga * ga, ab = al * be;
                if (g2 > eps * eps * ab && fabs(ga) > 1e-300) {
                    big |= g2 > 1e-16 * ab;                                // (relative off-diagonal above 1e-8)
                    const bool isp = col < pc;                             // this lane holds the lower column of the pair
                    const double ap = isp ? al : be, aq = isp ? be : al;
                    // The rotation without a float64 division (a ~25-instruction sequence on the critical path of every
                    // round; there were two, and a third for the convergence measure).  With d = aq - ap,
                    // h = sqrt(d^2 + 4 ga^2), m = |d| + h, n = sgn(d) 2 ga (sgn(0) = +1):
                    //   t = sgn(zeta) / (|zeta| + sqrt(1 + zeta^2)) = n / m   for zeta = d / (2 ga),
                    //   c = 1 / sqrt(1 + t^2) = m / sqrt(m^2 + n^2),   s = c t = n / sqrt(m^2 + n^2)
                    // - one square root and one reciprocal square root; c^2 + s^2 = 1 to the rounding of rsqrt.
                    const double d = aq - ap;
                    const double m = fabs(d) + sqrt(d * d + 4.0 * g2);
                    const double n = (d >= 0 ? 2.0 : -2.0) * ga;
                    const double r = rsqrt(m * m + 4.0 * g2);
                    const double c = m * r;
                    const double s = n * r;
                    const double so = isp ? -s : s;                        // p' = c p - s q ; q' = s p + c q
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        w[i] = c * w[i] + so * pw[i];
                        if (WITH_V) v[i] = c * v[i] + so * pv[i];
                    }
                    rotated = true;
                }
            }
            if (__ballot(rotated) == 0) { ++sweep; break; }
            // Jacobi converges quadratically: when the largest relative off-diagonal of this sweep was below 1e-8
            // (its square: 1e-16), the rotations just applied have left the columns orthogonal to ~1e-16 - the sweep
            // that would only confirm it (a tenth to a quarter of the whole solve when warm-started) is skipped
            if (__ballot(big) == 0) { ++sweep; break; }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            Wm[(half * 16 + i) * N + col] = w[i];
            if (WITH_V) Vm[(half * 16 + i) * N + col] = v[i];
        }
    }
    return sweep;
}

template <bool WITH_V = true>
__device__ inline int cca_hestenes_wave(CcaScratch &S, int tid) {
    if (tid < 64) {
        const int sweeps = cca_hestenes_wave_on<WITH_V>(S.W, S.V, tid);
        if (tid == 0) S.rotated = sweeps;
    }
    __syncthreads();
    return S.rotated;
}

// S11^-1/2 and S22^-1/2 of cca_solve (cca_solve.inl: two cca_inv_sqrt_spd calls in a row) with the two independent
// decompositions on waves 0 and 1 AT THE SAME TIME - each is a chain of ~300 dependent rounds on one wave, and the
// second wave was idle.  S22's matrices live in S.T / S.tmp (free until T is formed), its eigenvalue factors in S.rot.
// The same arithmetic on the same data in the same order: bit-identical to the sequential form.  blockDim.x >= 128.
__device__ inline void cca_inv_sqrt_pair_fast(CcaScratch &S, const double *S11, const double *S22, int tid, int nt) {
    const int N = CCA_DIM;
    double *W2 = S.T, *V2 = S.tmp, *sv2 = S.rot;
    for (int e = tid; e < N * N; e += nt) { S.W[e] = S11[e]; W2[e] = S22[e]; }
    cca_set_identity(S.V, tid, nt);
    cca_set_identity(V2, tid, nt);
    __syncthreads();
    if (tid < 64) (void)cca_hestenes_wave_on(S.W, S.V, tid);
    else if (tid < 128) (void)cca_hestenes_wave_on(W2, V2, tid - 64);
    __syncthreads();
    for (int j = tid; j < 2 * N; j += nt) {
        const double *Wm = j < N ? S.W : W2;
        const int c = j < N ? j : j - N;
        double n2 = 0;
        for (int i = 0; i < N; ++i) n2 += Wm[i * N + c] * Wm[i * N + c];
        (j < N ? S.sv : sv2)[c] = 1.0 / sqrt(sqrt(n2));        // eigenvalue = ||W_j||; want l^-1/2
    }
    __syncthreads();
    for (int e = tid; e < 2 * N * N; e += nt) {
        const bool second = e >= N * N;
        const int ee = second ? e - N * N : e;
        const int i = ee / N, j = ee - i * N;
        const double *Vm = second ? V2 : S.V, *sv = second ? sv2 : S.sv;
        double acc = 0;
        for (int k = 0; k < N; ++k) acc += Vm[i * N + k] * sv[k] * Vm[j * N + k];
        (second ? S.B : S.A)[ee] = acc;
    }
    __syncthreads();
}

__device__ inline int cca_hestenes_fast(CcaScratch &S, int tid) {
    if (ASR_CCA_WAVE) return cca_hestenes_wave(S, tid);
    const int N = CCA_DIM;
    const double eps = 1e-15;
    const int grp = tid >> 4, ln = tid & 15;
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        if (tid == 0) S.rotated = 0;
        __syncthreads();
        for (int r = 0; r < N - 1; ++r) {
            if (tid < 256) {
                int p, q;
                cca_pair(r, grp, &p, &q);
                double x0 = S.W[ln * N + p], y0 = S.W[ln * N + q];
                double x1 = S.W[(ln + 16) * N + p], y1 = S.W[(ln + 16) * N + q];
                double al = x0 * x0 + x1 * x1, be = y0 * y0 + y1 * y1, ga = x0 * y0 + x1 * y1;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    al += __shfl_xor(al, m);
                    be += __shfl_xor(be, m);
                    ga += __shfl_xor(ga, m);
                }
                const double lim = eps * sqrt(al * be);
                if (fabs(ga) > lim && fabs(ga) > 1e-300) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t);
                    const double s = c * t;
                    S.W[ln * N + p] = c * x0 - s * y0;
                    S.W[ln * N + q] = s * x0 + c * y0;
                    S.W[(ln + 16) * N + p] = c * x1 - s * y1;
                    S.W[(ln + 16) * N + q] = s * x1 + c * y1;
                    const double v0 = S.V[ln * N + p], w0 = S.V[ln * N + q];
                    const double v1 = S.V[(ln + 16) * N + p], w1 = S.V[(ln + 16) * N + q];
                    S.V[ln * N + p] = c * v0 - s * w0;
                    S.V[ln * N + q] = s * v0 + c * w0;
                    S.V[(ln + 16) * N + p] = c * v1 - s * w1;
                    S.V[(ln + 16) * N + q] = s * v1 + c * w1;
                    if (ln == 0) S.rotated = 1;
                }
            }
            __syncthreads();
        }
        const int any = S.rotated;
        __syncthreads();
        if (!any) { ++sweep; break; }
    }
    return sweep;
}
