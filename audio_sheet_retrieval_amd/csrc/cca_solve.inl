// 32x32 float64 algebra of CCA('svd') (reference: utils/cca.py:199-211:
//   S11^-1/2 = inv(sqrtm(S11)), S22^-1/2 likewise, T = S11^-1/2 S12 S22^-1/2,
//   U,s,Vt = svd(T), U <- S11^-1/2 U, V <- S22^-1/2 V)
// as one-sided Jacobi (Hestenes) iterations, written once for
//   - the gfx950 single-workgroup kernel (cca_kernels.hip: CCA_NT threads,
//     CCA_SYNC = __syncthreads), and
//   - a serial host build used only by tests/ to check the numerics on CPU
//     (tests/cca_host_harness.cpp: CCA_NT = 1, CCA_SYNC = nothing).
// Every phase writes disjoint data and phases are separated by CCA_SYNC().
//
// The including file defines: CCA_FN (function qualifiers), CCA_SYNC().
#ifndef CCA_DIM
#define CCA_DIM 32
#endif

struct CcaScratch {                 // lives in LDS on the device
    double W[CCA_DIM * CCA_DIM];
    double V[CCA_DIM * CCA_DIM];
    double A[CCA_DIM * CCA_DIM];    // S11^-1/2
    double B[CCA_DIM * CCA_DIM];    // S22^-1/2
    double T[CCA_DIM * CCA_DIM];
    double tmp[CCA_DIM * CCA_DIM];
    double red[(CCA_DIM / 2) * 16 * 3];
    double rot[(CCA_DIM / 2) * 2];  // (c, s) per pair
    double sv[CCA_DIM];
    int pq[(CCA_DIM / 2) * 2];
    int order[CCA_DIM];
    int rotated;
};

// round-robin ("circle") pairing: round r in [0, N-1), pair k in [0, N/2)
CCA_FN inline void cca_pair(int r, int k, int *p, int *q) {
    const int N = CCA_DIM;
    int a, b;
    if (k == 0) { a = N - 1; b = r % (N - 1); }
    else { a = (r + k) % (N - 1); b = (r - k + (N - 1)) % (N - 1); }
    *p = a < b ? a : b;
    *q = a < b ? b : a;
}

// One-sided Jacobi on the columns of S.W (row-major N x N); S.V accumulates the
// rotations (must hold the identity on entry).  On exit the columns of W are
// mutually orthogonal: W = W0 * V.  Returns the number of sweeps used.
CCA_FN inline int cca_hestenes(CcaScratch &S, int tid, int nt) {
    const int N = CCA_DIM, NP = CCA_DIM / 2;
    const double eps = 1e-15;
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        if (tid == 0) S.rotated = 0;
        CCA_SYNC();
        for (int r = 0; r < N - 1; ++r) {
            // phase A1: partial dot products, 16 slices per pair
            for (int w = tid; w < NP * 16; w += nt) {
                const int k = w >> 4, sl = w & 15;
                int p, q;
                cca_pair(r, k, &p, &q);
                double al = 0, be = 0, ga = 0;
                for (int i = sl; i < N; i += 16) {
                    const double x = S.W[i * N + p], y = S.W[i * N + q];
                    al += x * x; be += y * y; ga += x * y;
                }
                S.red[w * 3] = al; S.red[w * 3 + 1] = be; S.red[w * 3 + 2] = ga;
            }
            CCA_SYNC();
            // phase A2: rotation per pair
            for (int k = tid; k < NP; k += nt) {
                int p, q;
                cca_pair(r, k, &p, &q);
                double al = 0, be = 0, ga = 0;
                for (int sl = 0; sl < 16; ++sl) {
                    al += S.red[(k * 16 + sl) * 3];
                    be += S.red[(k * 16 + sl) * 3 + 1];
                    ga += S.red[(k * 16 + sl) * 3 + 2];
                }
                double c = 1.0, s = 0.0;
                const double lim = eps * sqrt(al * be);
                if (fabs(ga) > lim && fabs(ga) > 1e-300) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = c * t;
                    S.rotated = 1;
                }
                S.rot[2 * k] = c; S.rot[2 * k + 1] = s;
                S.pq[2 * k] = p; S.pq[2 * k + 1] = q;
            }
            CCA_SYNC();
            // phase B: apply the NP disjoint plane rotations to W and V
            for (int w = tid; w < NP * N * 2; w += nt) {
                const int which = w / (NP * N);
                const int rem = w - which * NP * N;
                const int k = rem / N, i = rem - k * N;
                const double c = S.rot[2 * k], s = S.rot[2 * k + 1];
                if (s == 0.0) continue;
                const int p = S.pq[2 * k], q = S.pq[2 * k + 1];
                double *M = which ? S.V : S.W;
                const double x = M[i * N + p], y = M[i * N + q];
                M[i * N + p] = c * x - s * y;
                M[i * N + q] = s * x + c * y;
            }
            CCA_SYNC();
        }
        const int any = S.rotated;
        CCA_SYNC();
        if (!any) { ++sweep; break; }
    }
    return sweep;
}

// The device builds define CCA_HESTENES as the shuffle-based fast path (cca_hestenes_fast.inl, declared before use
// through this forward declaration); the serial host build uses the generic sweeps above.
#ifndef CCA_HESTENES
#define CCA_HESTENES(S, tid, nt) cca_hestenes(S, tid, nt)
#endif

CCA_FN inline void cca_set_identity(double *M, int tid, int nt) {
    for (int e = tid; e < CCA_DIM * CCA_DIM; e += nt) M[e] = (e / CCA_DIM == e % CCA_DIM) ? 1.0 : 0.0;
}

// out = X * Y (row-major N x N)
CCA_FN inline void cca_matmul(const double *X, const double *Y, double *out, int tid, int nt) {
    const int N = CCA_DIM;
    for (int e = tid; e < N * N; e += nt) {
        const int i = e / N, j = e - i * N;
        double acc = 0;
        for (int k = 0; k < N; ++k) acc += X[i * N + k] * Y[k * N + j];
        out[e] = acc;
    }
}

// S^-1/2 of a symmetric positive definite S: Hestenes on S gives S V = V diag(l),
// column norms = eigenvalues; out = V diag(l^-1/2) V^T.
CCA_FN inline void cca_inv_sqrt_spd(CcaScratch &S, const double *Sin, double *out, int tid, int nt) {
    const int N = CCA_DIM;
    for (int e = tid; e < N * N; e += nt) S.W[e] = Sin[e];
    cca_set_identity(S.V, tid, nt);
    CCA_SYNC();
    CCA_HESTENES(S, tid, nt);
    for (int j = tid; j < N; j += nt) {
        double n2 = 0;
        for (int i = 0; i < N; ++i) n2 += S.W[i * N + j] * S.W[i * N + j];
        S.sv[j] = 1.0 / sqrt(sqrt(n2));        // eigenvalue = ||W_j||; want l^-1/2
    }
    CCA_SYNC();
    for (int e = tid; e < N * N; e += nt) {
        const int i = e / N, j = e - i * N;
        double acc = 0;
        for (int k = 0; k < N; ++k) acc += S.V[i * N + k] * S.sv[k] * S.V[j * N + k];
        out[e] = acc;
    }
    CCA_SYNC();
}

// S11, S22 (regularised, symmetric PD) and S12, all float64 row-major 32x32.
// Outputs: U, V (float64, columns ordered by descending canonical correlation),
// coeffs (the singular values).  utils/cca.py:201-211.
CCA_FN inline void cca_solve(CcaScratch &S, const double *S11, const double *S22, const double *S12,
                             double *Uout, double *Vout, double *coeffs, int tid, int nt) {
    const int N = CCA_DIM;
#ifdef CCA_INV_SQRT_PAIR
    CCA_INV_SQRT_PAIR(S, S11, S22, tid, nt);               // :201-202, the two decompositions side by side (device)
#else
    cca_inv_sqrt_spd(S, S11, S.A, tid, nt);               // :201
    cca_inv_sqrt_spd(S, S22, S.B, tid, nt);               // :202
#endif
    cca_matmul(S.A, S12, S.tmp, tid, nt);
    CCA_SYNC();
    cca_matmul(S.tmp, S.B, S.T, tid, nt);                 // :204  T = S11^-1/2 S12 S22^-1/2
    CCA_SYNC();
    for (int e = tid; e < N * N; e += nt) S.W[e] = S.T[e];
    cca_set_identity(S.V, tid, nt);
    CCA_SYNC();
    CCA_HESTENES(S, tid, nt);                             // :206  T V = U diag(s)
    for (int j = tid; j < N; j += nt) {
        double n2 = 0;
        for (int i = 0; i < N; ++i) n2 += S.W[i * N + j] * S.W[i * N + j];
        S.sv[j] = sqrt(n2);
    }
    CCA_SYNC();
    for (int j = tid; j < N; j += nt) {                    // descending order, stable
        int rank = 0;
        for (int k = 0; k < N; ++k) rank += (S.sv[k] > S.sv[j]) || (S.sv[k] == S.sv[j] && k < j);
        S.order[rank] = j;
    }
    CCA_SYNC();
    // tmp <- left singular vectors (normalised columns of W, reordered); T <- V reordered
    for (int e = tid; e < N * N; e += nt) {
        const int i = e / N, jj = e - i * N;
        const int j = S.order[jj];
        const double sj = S.sv[j];
        S.tmp[e] = sj > 0 ? S.W[i * N + j] / sj : 0.0;
        S.T[e] = S.V[i * N + j];
    }
    for (int jj = tid; jj < N; jj += nt) coeffs[jj] = S.sv[S.order[jj]];     // :208
    CCA_SYNC();
    cca_matmul(S.A, S.tmp, Uout, tid, nt);                // :210  U = S11^-1/2 U
    cca_matmul(S.B, S.T, Vout, tid, nt);                  // :211  V = S22^-1/2 V
    CCA_SYNC();
}
