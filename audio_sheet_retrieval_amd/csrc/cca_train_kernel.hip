// CCALayer train branch + LengthNormLayer + pairwise ranking loss, forward AND
// backward, as ONE single-workgroup gfx950 kernel (1024 threads, float64).
//
// Reference: models/lasagne_extensions/layers/cca.py:91-182,198-201 (forward;
// gradients by theano.grad incl. nlinalg.EighGrad), cca.py:39-40
// (LengthNormLayer), models/objectives.py:30-69 (contrastive cos loss).
// The reference runs the four eigh's on the HOST (LAPACK via NumPy, float32) and
// their gradients in pure Python; here the 32x32 eigenproblems are solved by
// one-sided Jacobi in LDS (cca_solve.inl) and everything stays on the device.
// Arithmetic is float64 throughout (inputs/outputs float32): the reference is
// float32, the parity budget is 1e-4, and the 1/(w_n - w_m) terms of the eigh
// gradient are the sensitive part.
//
// Eigenvector conventions: eigh fixes eigenvectors only up to sign; LAPACK's
// choice is not reproducible.  U is sign-fixed against V by the reference itself
// (cca.py:172-173), so the only freedom left is a JOINT sign per canonical
// dimension (U[:,j], V[:,j]) -> (-U[:,j], -V[:,j]).  Loss, lv1.lv2^T and every
// gradient wrt H1/H2 are invariant under it.
#include "asr_kernels.h"
#include <cmath>

#define CCA_FN __device__
#define CCA_SYNC() __syncthreads()
#include "cca_solve.inl"

namespace asr {

constexpr int CT_THREADS = 1024;
constexpr int D = 32;
constexpr int DD = D * D;

// float64 workspace layout (offsets in doubles)
struct CcaTrainWs {
    // 32x32 matrices
    enum { S11 = 0, S22, S12, S11si, S22si, T, M, E, F, U0, U, V, dU, dV, dS11si, dS22si, dE, dF, dM1, dM2, dT, dS12,
           dS11, dS22, tmpA, tmpB, tmpC, NMAT };
    // vectors (32)
    enum { mean1 = 0, mean2, d1, d2, E1, F1, sgn, vtmp, cmean1, cmean2, NVEC };
};

__device__ __forceinline__ double *mat(double *ws, int id) { return ws + (size_t)id * DD; }
__device__ __forceinline__ double *vec(double *ws, int id) { return ws + (size_t)CcaTrainWs::NMAT * DD + (size_t)id * D; }

// out = X^T (transpose flags) helpers on 32x32 row-major
__device__ void mm(const double *X, bool tx, const double *Y, bool ty, double *out, int tid, int nt) {
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, j = e - i * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += (tx ? X[k * D + i] : X[i * D + k]) * (ty ? Y[j * D + k] : Y[k * D + j]);
        out[e] = acc;
    }
    __syncthreads();
}

// eigen-decomposition of a symmetric positive definite matrix: ascending eigenvalues w, eigenvectors in the
// columns of Vout (one-sided Jacobi on Min: Min*V = V*diag(w), column norms are the eigenvalues).
__device__ void eigh_spd(CcaScratch &S, const double *Min, double *w, double *Vout, int tid, int nt) {
    for (int e = tid; e < DD; e += nt) S.W[e] = Min[e];
    cca_set_identity(S.V, tid, nt);
    __syncthreads();
    cca_hestenes(S, tid, nt);
    for (int j = tid; j < D; j += nt) {
        double n2 = 0;
        for (int i = 0; i < D; ++i) n2 += S.W[i * D + j] * S.W[i * D + j];
        S.sv[j] = sqrt(n2);
    }
    __syncthreads();
    for (int j = tid; j < D; j += nt) {
        int rank = 0;
        for (int k = 0; k < D; ++k) rank += (S.sv[k] < S.sv[j]) || (S.sv[k] == S.sv[j] && k < j);
        S.order[rank] = j;
    }
    __syncthreads();
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, jj = e - i * D;
        Vout[e] = S.V[i * D + S.order[jj]];
    }
    for (int jj = tid; jj < D; jj += nt) w[jj] = S.sv[S.order[jj]];
    __syncthreads();
}

// out = V diag(w^-1/2) V^T
__device__ void inv_sqrt_from_eig(const double *w, const double *V, double *out, int tid, int nt) {
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, j = e - i * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += V[i * D + k] * V[j * D + k] / sqrt(w[k]);
        out[e] = acc;
    }
    __syncthreads();
}

// Theano EighGrad: g = v (diag(gw) + K) v^T, K[n,m] = (v^T gv)[m,n]/(w[n]-w[m]); out = tril(g)+triu(g,1)^T
// gw may be null (zeros).  Uses tmp1, tmp2 (32x32).
__device__ void eigh_grad(const double *w, const double *v, const double *gw, const double *gv, double *out,
                          double *tmp1, double *tmp2, int tid, int nt) {
    mm(v, true, gv, false, tmp1, tid, nt);                       // tmp1[m][n] = v_m . gv_n
    for (int e = tid; e < DD; e += nt) {
        const int n = e / D, m = e - n * D;
        tmp2[e] = (n == m) ? (gw ? gw[n] : 0.0) : tmp1[m * D + n] / (w[n] - w[m]);     // diag(gw) + K
    }
    __syncthreads();
    mm(v, false, tmp2, false, tmp1, tid, nt);                    // v (diag + K)
    mm(tmp1, false, v, true, tmp2, tid, nt);                     // g
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, j = e - i * D;
        out[e] = (i > j) ? tmp2[e] + tmp2[j * D + i] : (i == j ? tmp2[e] : 0.0);
    }
    __syncthreads();
}

// gradient through S^-1/2 = (A*w) A^T, (d,A) = eigh(S), w = d^-1/2; Q = grad wrt S^-1/2; out = grad wrt S
__device__ void inv_sqrt_bwd(const double *d, const double *A, const double *Q, double *out, double *dd, double *t1,
                             double *t2, double *t3, int tid, int nt) {
    // dA = (Q + Q^T) (A * w)
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, j = e - i * D;
        t1[e] = Q[e] + Q[j * D + i];
    }
    __syncthreads();
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, k = e - i * D;
        double acc = 0.0;
        for (int j = 0; j < D; ++j) acc += t1[i * D + j] * A[j * D + k];
        t2[e] = acc / sqrt(d[k]);
    }
    // dw_k = a_k^T Q a_k ; dd_k = dw_k * (-1/2) d_k^-3/2
    for (int k = tid; k < D; k += nt) {
        double acc = 0.0;
        for (int i = 0; i < D; ++i) {
            double r = 0.0;
            for (int j = 0; j < D; ++j) r += Q[i * D + j] * A[j * D + k];
            acc += A[i * D + k] * r;
        }
        dd[k] = acc * (-0.5) / (d[k] * sqrt(d[k]));
    }
    __syncthreads();
    eigh_grad(d, A, dd, t2, out, t1, t3, tid, nt);
}

struct CcaTrainArgs {
    const float *H1, *H2;        // (B,32)
    const float *cca_in;         // running values: U V mean1 mean2 S12 S11 S22 (5184 floats, reference order)
    float *cca_out;              // new values, same layout (may alias cca_in)
    float *dH1, *dH2;            // (B,32) gradients (may be null: forward only)
    float *lv1, *lv2;            // (B,32) train-mode outputs (may be null)
    float *loss_out;             // [0] ranking loss, [1..32] corr
    double *ws;                  // CcaTrainWs matrices + vectors, then B-sized arrays
    int B;
    float r1, r2, rT, alpha, gamma;
};

__global__ __launch_bounds__(CT_THREADS) void cca_train_kernel(CcaTrainArgs a) {
    __shared__ CcaScratch S;
    double *red = S.tmp;          // 1024 doubles, free outside the Jacobi solver
    const int tid = threadIdx.x, nt = CT_THREADS;
    const int B = a.B;
    double *ws = a.ws;
    typedef CcaTrainWs W;
    // B-sized float64 arrays after the fixed part
    double *bs = ws + (size_t)W::NMAT * DD + (size_t)W::NVEC * D;
    double *Hb1 = bs, *Hb2 = Hb1 + (size_t)B * D, *o1 = Hb2 + (size_t)B * D, *o2 = o1 + (size_t)B * D;
    double *l1 = o2 + (size_t)B * D, *l2 = l1 + (size_t)B * D, *g1 = l2 + (size_t)B * D, *g2 = g1 + (size_t)B * D;
    double *nrm1 = g2 + (size_t)B * D, *nrm2 = nrm1 + B, *rowsum = nrm2 + B, *diag = rowsum + B;
    const double al = (double)a.alpha, oma = 1.0 - al;
    const double cinv = 1.0 / ((double)B - 1.0);
    const float *Uin = a.cca_in, *m1in = a.cca_in + 2 * DD, *m2in = m1in + D;
    const float *S12in = m2in + D, *S11in = S12in + DD, *S22in = S11in + DD;
    (void)Uin;

    // ---- means (cca.py:94-106)
    for (int c = tid; c < 2 * D; c += nt) {
        const float *H = c < D ? a.H1 : a.H2;
        const int cc = c & (D - 1);
        double s = 0.0;
        for (int n = 0; n < B; ++n) s += (double)H[(size_t)n * D + cc];
        const double run = (double)(c < D ? m1in[cc] : m2in[cc]);
        vec(ws, c < D ? W::mean1 : W::mean2)[cc] = oma * run + al * (s / (double)B);
    }
    __syncthreads();
    for (int e = tid; e < B * D; e += nt) {                       // :109-110
        const int c = e & (D - 1);
        Hb1[e] = (double)a.H1[e] - vec(ws, W::mean1)[c];
        Hb2[e] = (double)a.H2[e] - vec(ws, W::mean2)[c];
    }
    __syncthreads();
    // ---- covariances (:117-141)
    for (int e = tid; e < 3 * DD; e += nt) {
        const int m = e / DD, idx = e - m * DD, i = idx / D, j = idx - i * D;
        const double *X = (m == 1) ? Hb2 : Hb1, *Y = (m == 0) ? Hb1 : Hb2;      // S11: 1,1  S22: 2,2  S12: 1,2
        double s = 0.0;
        for (int n = 0; n < B; ++n) s += X[(size_t)n * D + i] * Y[(size_t)n * D + j];
        s *= cinv;
        if (m == 0 && i == j) s += (double)a.r1;
        if (m == 1 && i == j) s += (double)a.r2;
        const float *run = (m == 0) ? S11in : (m == 1 ? S22in : S12in);
        mat(ws, m == 0 ? W::S11 : (m == 1 ? W::S22 : W::S12))[idx] = oma * (double)run[idx] + al * s;
    }
    __syncthreads();
    // ---- S11^-1/2, S22^-1/2 (:144-147)
    eigh_spd(S, mat(ws, W::S11), vec(ws, W::d1), mat(ws, W::tmpA), tid, nt);       // tmpA = A1
    inv_sqrt_from_eig(vec(ws, W::d1), mat(ws, W::tmpA), mat(ws, W::S11si), tid, nt);
    eigh_spd(S, mat(ws, W::S22), vec(ws, W::d2), mat(ws, W::tmpB), tid, nt);       // tmpB = A2
    inv_sqrt_from_eig(vec(ws, W::d2), mat(ws, W::tmpB), mat(ws, W::S22si), tid, nt);
    // ---- T, M1, M2, eigh (:150-158)
    mm(mat(ws, W::S11si), false, mat(ws, W::S12), false, mat(ws, W::tmpC), tid, nt);
    mm(mat(ws, W::tmpC), false, mat(ws, W::S22si), false, mat(ws, W::T), tid, nt);
    mm(mat(ws, W::T), false, mat(ws, W::T), true, mat(ws, W::M), tid, nt);
    for (int i = tid; i < D; i += nt) mat(ws, W::M)[i * D + i] += (double)a.rT;
    __syncthreads();
    eigh_spd(S, mat(ws, W::M), vec(ws, W::E1), mat(ws, W::E), tid, nt);
    mm(mat(ws, W::T), true, mat(ws, W::T), false, mat(ws, W::M), tid, nt);
    for (int i = tid; i < D; i += nt) mat(ws, W::M)[i * D + i] += (double)a.rT;
    __syncthreads();
    eigh_spd(S, mat(ws, W::M), vec(ws, W::F1), mat(ws, W::F), tid, nt);
    // ---- U, V, sign fix (:167-173)
    mm(mat(ws, W::S11si), false, mat(ws, W::E), false, mat(ws, W::U0), tid, nt);
    mm(mat(ws, W::S22si), false, mat(ws, W::F), false, mat(ws, W::V), tid, nt);
    mm(mat(ws, W::S12), false, mat(ws, W::V), false, mat(ws, W::tmpC), tid, nt);
    for (int j = tid; j < D; j += nt) {
        double s = 0.0;
        for (int i = 0; i < D; ++i) s += mat(ws, W::U0)[i * D + j] * mat(ws, W::tmpC)[i * D + j];
        vec(ws, W::sgn)[j] = s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0);
    }
    __syncthreads();
    for (int e = tid; e < DD; e += nt) mat(ws, W::U)[e] = mat(ws, W::U0)[e] * vec(ws, W::sgn)[e & (D - 1)];
    __syncthreads();
    // ---- write the new running values (:98-141, 176-182) and corr (:161-164)
    for (int e = tid; e < DD; e += nt) {
        a.cca_out[e] = (float)mat(ws, W::U)[e];
        a.cca_out[DD + e] = (float)mat(ws, W::V)[e];
        a.cca_out[2 * DD + 2 * D + e] = (float)mat(ws, W::S12)[e];
        a.cca_out[3 * DD + 2 * D + e] = (float)mat(ws, W::S11)[e];
        a.cca_out[4 * DD + 2 * D + e] = (float)mat(ws, W::S22)[e];
    }
    for (int c = tid; c < D; c += nt) {
        a.cca_out[2 * DD + c] = (float)vec(ws, W::mean1)[c];
        a.cca_out[2 * DD + D + c] = (float)vec(ws, W::mean2)[c];
        double e1 = vec(ws, W::E1)[c];
        e1 = e1 < 1e-7 ? 1e-7 : (e1 > 1.0 ? 1.0 : e1);
        a.loss_out[1 + c] = (float)sqrt(e1);
    }
    // ---- projections + length norm (:198-201, 39-40)
    for (int e = tid; e < B * D; e += nt) {
        const int n = e / D, j = e - n * D;
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < D; ++k) {
            s1 += Hb1[(size_t)n * D + k] * mat(ws, W::U)[k * D + j];
            s2 += Hb2[(size_t)n * D + k] * mat(ws, W::V)[k * D + j];
        }
        o1[e] = s1; o2[e] = s2;
    }
    __syncthreads();
    for (int n = tid; n < 2 * B; n += nt) {
        const double *o = n < B ? o1 + (size_t)n * D : o2 + (size_t)(n - B) * D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) s += o[k] * o[k];
        (n < B ? nrm1 : nrm2)[n < B ? n : n - B] = sqrt(s);
    }
    __syncthreads();
    for (int e = tid; e < B * D; e += nt) {
        const int n = e / D;
        l1[e] = o1[e] / nrm1[n];
        l2[e] = o2[e] / nrm2[n];
        if (a.lv1) a.lv1[e] = (float)l1[e];
        if (a.lv2) a.lv2[e] = (float)l2[e];
    }
    __syncthreads();
    // ---- ranking loss (objectives.py:36-50).  A 32-lane group owns one row: lane k holds component k,
    // dot products by xor-shuffles inside the half-wave (no per-thread 32-vectors, no spills).
    const double gam = (double)a.gamma, wpair = 1.0 / ((double)B * ((double)B - 1.0));
    const int grp = tid >> 5, k = tid & 31, ngrp = nt >> 5;
    auto hsum = [](double v) {
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        return v;
    };
    double lpart = 0.0;
    for (int i = grp; i < B; i += ngrp) {                          // row pass: dlv1, row sums, loss
        const double lik = l1[(size_t)i * D + k];
        const double l2ik = l2[(size_t)i * D + k];
        const double dii = hsum(lik * l2ik);
        double acc = 0.0, rs = 0.0;
        for (int j = 0; j < B; ++j) {
            const double l2jk = l2[(size_t)j * D + k];
            const double dij = hsum(lik * l2jk);
            if (j == i) continue;
            const double L = gam - dii + dij;
            if (L >= 0.0 && L <= 1000.0) { rs += 1.0; acc += l2jk; }
            if (k == 0) lpart += L < 0.0 ? 0.0 : (L > 1000.0 ? 1000.0 : L);
        }
        g1[(size_t)i * D + k] = wpair * (acc - rs * l2ik);
        if (k == 0) { rowsum[i] = rs; diag[i] = dii; }
    }
    red[tid] = lpart;
    __syncthreads();
    for (int st = CT_THREADS / 2; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) a.loss_out[0] = (float)(red[0] * wpair);
    if (a.dH1 == nullptr) return;                                 // forward only (uniform branch)
    // ---- column pass: dlv2_j = wpair * ( sum_{i != j} M_ij lv1_i - rowsum_j lv1_j )
    for (int j = grp; j < B; j += ngrp) {
        const double ljk = l2[(size_t)j * D + k];
        double acc = 0.0;
        for (int i = 0; i < B; ++i) {
            const double l1ik = l1[(size_t)i * D + k];
            const double dij = hsum(l1ik * ljk);
            if (i == j) continue;
            const double L = gam - diag[i] + dij;
            if (L >= 0.0 && L <= 1000.0) acc += l1ik;
        }
        g2[(size_t)j * D + k] = wpair * (acc - rowsum[j] * l1[(size_t)j * D + k]);
    }
    __syncthreads();
    // ---- length-norm backward: dout = (dlv - lv (lv.dlv)) / ||out||   (g1,g2 in place)
    for (int n = tid; n < 2 * B; n += nt) {
        const bool first = n < B;
        const int r = first ? n : n - B;
        double *g = (first ? g1 : g2) + (size_t)r * D;
        const double *l = (first ? l1 : l2) + (size_t)r * D;
        double dot = 0.0;
        for (int k = 0; k < D; ++k) dot += l[k] * g[k];
        const double inv = 1.0 / (first ? nrm1 : nrm2)[r];
        for (int k = 0; k < D; ++k) g[k] = (g[k] - l[k] * dot) * inv;
    }
    __syncthreads();
    // ---- dU = Hb1^T dout1, dV = Hb2^T dout2
    for (int e = tid; e < 2 * DD; e += nt) {
        const int m = e / DD, idx = e - m * DD, i = idx / D, j = idx - i * D;
        const double *Hb = m ? Hb2 : Hb1, *g = m ? g2 : g1;
        double s = 0.0;
        for (int n = 0; n < B; ++n) s += Hb[(size_t)n * D + i] * g[(size_t)n * D + j];
        mat(ws, m ? W::dV : W::dU)[idx] = s;
    }
    __syncthreads();
    // dU0 = dU * s (in place)
    for (int e = tid; e < DD; e += nt) mat(ws, W::dU)[e] *= vec(ws, W::sgn)[e & (D - 1)];
    __syncthreads();
    mm(mat(ws, W::dU), false, mat(ws, W::E), true, mat(ws, W::dS11si), tid, nt);       // dU0 E^T
    mm(mat(ws, W::S11si), true, mat(ws, W::dU), false, mat(ws, W::dE), tid, nt);       // S11si^T dU0
    mm(mat(ws, W::dV), false, mat(ws, W::F), true, mat(ws, W::dS22si), tid, nt);
    mm(mat(ws, W::S22si), true, mat(ws, W::dV), false, mat(ws, W::dF), tid, nt);
    eigh_grad(vec(ws, W::E1), mat(ws, W::E), nullptr, mat(ws, W::dE), mat(ws, W::dM1), mat(ws, W::tmpC),
              mat(ws, W::M), tid, nt);
    eigh_grad(vec(ws, W::F1), mat(ws, W::F), nullptr, mat(ws, W::dF), mat(ws, W::dM2), mat(ws, W::tmpC),
              mat(ws, W::M), tid, nt);
    // dT = (dM1 + dM1^T) T + T (dM2 + dM2^T)
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, j = e - i * D;
        double s = 0.0;
        for (int k = 0; k < D; ++k) {
            s += (mat(ws, W::dM1)[i * D + k] + mat(ws, W::dM1)[k * D + i]) * mat(ws, W::T)[k * D + j];
            s += mat(ws, W::T)[i * D + k] * (mat(ws, W::dM2)[k * D + j] + mat(ws, W::dM2)[j * D + k]);
        }
        mat(ws, W::dT)[e] = s;
    }
    __syncthreads();
    // dS11si += dT (S12 S22si)^T ; dS12 = S11si^T dT S22si^T ; dS22si += (S11si S12)^T dT
    mm(mat(ws, W::S12), false, mat(ws, W::S22si), false, mat(ws, W::tmpC), tid, nt);
    mm(mat(ws, W::dT), false, mat(ws, W::tmpC), true, mat(ws, W::M), tid, nt);
    for (int e = tid; e < DD; e += nt) mat(ws, W::dS11si)[e] += mat(ws, W::M)[e];
    __syncthreads();
    mm(mat(ws, W::S11si), true, mat(ws, W::dT), false, mat(ws, W::tmpC), tid, nt);
    mm(mat(ws, W::tmpC), false, mat(ws, W::S22si), true, mat(ws, W::dS12), tid, nt);
    mm(mat(ws, W::S11si), false, mat(ws, W::S12), false, mat(ws, W::tmpC), tid, nt);
    mm(mat(ws, W::tmpC), true, mat(ws, W::dT), false, mat(ws, W::M), tid, nt);
    for (int e = tid; e < DD; e += nt) mat(ws, W::dS22si)[e] += mat(ws, W::M)[e];
    __syncthreads();
    inv_sqrt_bwd(vec(ws, W::d1), mat(ws, W::tmpA), mat(ws, W::dS11si), mat(ws, W::dS11), vec(ws, W::vtmp),
                 mat(ws, W::tmpC), mat(ws, W::M), mat(ws, W::dM1), tid, nt);
    inv_sqrt_bwd(vec(ws, W::d2), mat(ws, W::tmpB), mat(ws, W::dS22si), mat(ws, W::dS22), vec(ws, W::vtmp),
                 mat(ws, W::tmpC), mat(ws, W::M), mat(ws, W::dM1), tid, nt);
    // ---- dHb = dout U^T + alpha c ( Hb (dS + dS^T) + Hb_other dS12(^T) )   (written over o1/o2)
    const double ac = al * cinv;
    for (int e = tid; e < B * D; e += nt) {
        const int n = e / D, i = e - n * D;
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < D; ++k) {
            s1 += g1[(size_t)n * D + k] * mat(ws, W::U)[i * D + k];
            s2 += g2[(size_t)n * D + k] * mat(ws, W::V)[i * D + k];
            s1 += ac * (Hb1[(size_t)n * D + k] * (mat(ws, W::dS11)[k * D + i] + mat(ws, W::dS11)[i * D + k]) +
                        Hb2[(size_t)n * D + k] * mat(ws, W::dS12)[i * D + k]);
            s2 += ac * (Hb2[(size_t)n * D + k] * (mat(ws, W::dS22)[k * D + i] + mat(ws, W::dS22)[i * D + k]) +
                        Hb1[(size_t)n * D + k] * mat(ws, W::dS12)[k * D + i]);
        }
        o1[e] = s1; o2[e] = s2;
    }
    __syncthreads();
    for (int c = tid; c < 2 * D; c += nt) {
        const double *o = c < D ? o1 : o2;
        const int cc = c & (D - 1);
        double s = 0.0;
        for (int n = 0; n < B; ++n) s += o[(size_t)n * D + cc];
        vec(ws, c < D ? W::cmean1 : W::cmean2)[cc] = s / (double)B;
    }
    __syncthreads();
    for (int e = tid; e < B * D; e += nt) {
        const int c = e & (D - 1);
        a.dH1[e] = (float)(o1[e] - al * vec(ws, W::cmean1)[c]);
        a.dH2[e] = (float)(o2[e] - al * vec(ws, W::cmean2)[c]);
    }
}

// iter_funcs['valid'] (utils/train_dcca_pool.py:155): ranking loss of deterministic outputs, no gradients
__global__ __launch_bounds__(1024) void rank_loss_kernel(const float *__restrict__ lv1, const float *__restrict__ lv2, int B,
                                                         float gamma, float *__restrict__ loss_out) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, grp = tid >> 5, k = tid & 31;
    auto hsum = [](double v) {
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        return v;
    };
    double lpart = 0.0;
    for (int i = grp; i < B; i += 32) {
        const double lik = (double)lv1[(size_t)i * D + k];
        const double dii = hsum(lik * (double)lv2[(size_t)i * D + k]);
        for (int j = 0; j < B; ++j) {
            const double dij = hsum(lik * (double)lv2[(size_t)j * D + k]);
            if (j == i || k != 0) continue;
            const double L = (double)gamma - dii + dij;
            lpart += L < 0.0 ? 0.0 : (L > 1000.0 ? 1000.0 : L);
        }
    }
    red[tid] = lpart;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) loss_out[0] = (float)(red[0] / ((double)B * ((double)B - 1.0)));
}

hipError_t launch_rank_loss(hipStream_t s, const float *lv1, const float *lv2, int B, float gamma, float *loss_out) {
    rank_loss_kernel<<<1, 1024, 0, s>>>(lv1, lv2, B, gamma, loss_out);
    return hipGetLastError();
}

size_t cca_train_ws_bytes(int B) {
    return ((size_t)CcaTrainWs::NMAT * DD + (size_t)CcaTrainWs::NVEC * D + (size_t)8 * B * D + (size_t)4 * B) *
           sizeof(double);
}

hipError_t launch_cca_train(hipStream_t s, const float *H1, const float *H2, int B, const float *cca_in,
                            float *cca_out, float r1, float r2, float rT, float alpha, float gamma, void *ws,
                            float *loss_out, float *lv1, float *lv2, float *dH1, float *dH2) {
    CcaTrainArgs a;
    a.H1 = H1; a.H2 = H2; a.cca_in = cca_in; a.cca_out = cca_out; a.dH1 = dH1; a.dH2 = dH2;
    a.lv1 = lv1; a.lv2 = lv2; a.loss_out = loss_out; a.ws = (double *)ws; a.B = B;
    a.r1 = r1; a.r2 = r2; a.rT = rT; a.alpha = alpha; a.gamma = gamma;
    cca_train_kernel<<<1, CT_THREADS, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace asr
