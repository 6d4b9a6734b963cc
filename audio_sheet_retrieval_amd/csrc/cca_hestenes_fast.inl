// Device-only fast path of cca_hestenes (cca_solve.inl): the same one-sided Jacobi sweeps with ONE barrier per
// round instead of three.  The first 256 threads work: 16 lanes per column pair; lane `ln` owns rows ln and ln+16
// of W and V, the three dot products are reduced with xor-shuffles inside the 16-lane group, every lane derives the
// same (c, s) and rotates its own rows.  Pairs of one round touch disjoint columns, so no intra-round hazards.
// Requires blockDim.x >= 256, CCA_DIM == 32.  Returns the number of sweeps.
__device__ inline int cca_hestenes_fast(CcaScratch &S, int tid) {
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
