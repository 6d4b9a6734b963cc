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
__device__ inline int cca_hestenes_wave(CcaScratch &S, int tid) {
    const int N = CCA_DIM;
    const double eps = 1e-15;
    int sweep = 0;
    if (tid < 64) {
        const int col = tid >> 1, half = tid & 1;
        double w[16], v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            w[i] = S.W[(half * 16 + i) * N + col];
            v[i] = S.V[(half * 16 + i) * N + col];
        }
        for (; sweep < 40; ++sweep) {
            bool rotated = false;
            double worst = 0.0;                                            // largest ga^2 / (al be) met in this sweep
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
                    pv[i] = __shfl(v[i], pl);
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
                    worst = fmax(worst, g2 / ab);
                    const bool isp = col < pc;                             // this lane holds the lower column of the pair
                    const double ap = isp ? al : be, aq = isp ? be : al;
                    const double zeta = (aq - ap) / (2.0 * ga);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = rsqrt(1.0 + t * t);
                    const double s = c * t;
                    const double so = isp ? -s : s;                        // p' = c p - s q ; q' = s p + c q
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        w[i] = c * w[i] + so * pw[i];
                        v[i] = c * v[i] + so * pv[i];
                    }
                    rotated = true;
                }
            }
            if (__ballot(rotated) == 0) { ++sweep; break; }
            // Jacobi converges quadratically: when the largest relative off-diagonal of this sweep was below 1e-8
            // (worst = its square), the rotations just applied have left the columns orthogonal to ~1e-16 - the sweep
            // that would only confirm it (a tenth to a quarter of the whole solve when warm-started) is skipped
            if (__ballot(worst > 1e-16) == 0) { ++sweep; break; }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            S.W[(half * 16 + i) * N + col] = w[i];
            S.V[(half * 16 + i) * N + col] = v[i];
        }
        if (tid == 0) S.rotated = sweep;
    }
    __syncthreads();
    return S.rotated;
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
