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
#include <cstdlib>
#include <algorithm>

#define CCA_FN __device__
#define CCA_SYNC() __syncthreads()
#include "cca_solve.inl"
#include "cca_hestenes_fast.inl"

namespace asr {

constexpr int CT_THREADS = 1024;           // launch bound
constexpr int CT_THREADS_DEFAULT = 1024;
constexpr int D = 32;
constexpr int DD = D * D;

// float64 workspace layout (offsets in doubles)
struct CcaTrainWs {
    // 32x32 matrices
    enum { S11 = 0, S22, S12, S11si, S22si, T, M, E, F, U0, U, V, dU, dV, dS11si, dS22si, dE, dF, dM1, dM2, dT, dS12,
           dS11, dS22, tmpA, tmpB, tmpC, T2, tmpC2, M2, NMAT };
    // vectors (32)
    enum { mean1 = 0, mean2, d1, d2, E1, F1, sgn, vtmp, cmean1, cmean2, sdout1, sdout2, shb1, shb2, warm, vtmp2, NVEC };
};

__device__ __forceinline__ double *mat(double *ws, int id) { return ws + (size_t)id * DD; }
__device__ __forceinline__ double *vec(double *ws, int id) { return ws + (size_t)CcaTrainWs::NMAT * DD + (size_t)id * D; }

// out = X^T (transpose flags) helpers on 32x32 row-major
// Both operands are staged in LDS first (row pitch 33: the transposed reads of 32 lanes hit 32 banks): the matrices
// live in the global workspace, and 64 loads per thread straight from there made every product a ~6 us chain of L2
// round trips - the backward phases of the CCALayer are ~20 such products.  Same products, same order of summation.
__device__ void mm(const double *X, bool tx, const double *Y, bool ty, double *out, int tid, int nt) {
    constexpr int P = D + 1;
    __shared__ double xs[D * P], ys[D * P];
    for (int e = tid; e < DD; e += nt) {
        const int r = e / D, c = e - r * D;
        xs[r * P + c] = X[e];
        ys[r * P + c] = Y[e];
    }
    __syncthreads();
    for (int e = tid; e < DD; e += nt) {
        const int i = e / D, j = e - i * D;
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += (tx ? xs[k * P + i] : xs[i * P + k]) * (ty ? ys[j * P + k] : ys[k * P + j]);
        out[e] = acc;
    }
    __syncthreads();
}

// eigen-decomposition of a symmetric positive definite matrix: ascending eigenvalues w, eigenvectors in the
// columns of Vout (one-sided Jacobi on Min: Min*V = V*diag(w), column norms are the eigenvalues).
// warm: Vout holds the eigenvectors of the PREVIOUS step's matrix (the covariances are exponential averages and T moves
// with them: consecutive matrices are close).  The iteration then starts from V = V_prev, W = Min V_prev - columns that
// are almost orthogonal already - and needs 2-4 sweeps instead of ~10; any orthogonal start converges to the same
// decomposition up to rounding, and the loss and every gradient are invariant under the remaining sign freedom.
// reg: the regulariser the caller put on Min's diagonal (r1 / r2 / rT of asr_config).  From 1e-6 upwards the iteration
// runs without V (see below); a caller that switches the regulariser off gets the V-carrying iteration, whose vectors
// stay orthonormal whatever the spectrum.
__device__ void eigh_spd(CcaScratch &S, const double *Min, double *w, double *Vout, int tid, int nt, bool warm, float reg) {
    if (warm) {
        for (int e = tid; e < DD; e += nt) { S.V[e] = Vout[e]; S.tmp[e] = Min[e]; }
        __syncthreads();
        for (int e = tid; e < DD; e += nt) {
            const int i = e / D, j = e - i * D;
            double acc = 0.0;
            for (int k = 0; k < D; ++k) acc += S.tmp[i * D + k] * S.V[k * D + j];
            S.W[e] = acc;
        }
    } else {
        for (int e = tid; e < DD; e += nt) S.W[e] = Min[e];
        cca_set_identity(S.V, tid, nt);
    }
    __syncthreads();
    // ASR_CCA_NOV (default 1, round 6): the Jacobi iteration does not carry V - see cca_hestenes_wave_on<false>; the
    // eigenvectors are W's columns over their norms.  0: V accumulated by the rotations as in rounds 2-5.
#ifndef ASR_CCA_NOV
#define ASR_CCA_NOV 1
#endif
    const bool nov = ASR_CCA_NOV && ASR_CCA_WAVE && reg >= 1e-6f;      // uniform
    if (nov) cca_hestenes_wave<false>(S, tid);
    else cca_hestenes_fast(S, tid);
    for (int j = tid; j < D; j += nt) {
        double n2 = 0;
        for (int i = 0; i < D; ++i) n2 += S.W[i * D + j] * S.W[i * D + j];
        S.sv[j] = sqrt(n2);
    }
    __syncthreads();
    if (nov) {
        for (int e = tid; e < DD; e += nt) S.V[e] = S.W[e] / S.sv[e & (D - 1)];
        __syncthreads();
    }
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
    // dw_k = a_k^T Q a_k ; dd_k = dw_k * (-1/2) d_k^-3/2.  Q A as one product of all threads (t3 is free until eigh_grad)
    // instead of 32 threads walking 1024 terms each through global memory; the same sums in the same order
    __syncthreads();
    mm(Q, false, A, false, t3, tid, nt);
    for (int k = tid; k < D; k += nt) {
        double acc = 0.0;
        for (int i = 0; i < D; ++i) acc += A[i * D + k] * t3[i * D + k];
        dd[k] = acc * (-0.5) / (d[k] * sqrt(d[k]));
    }
    __syncthreads();
    eigh_grad(d, A, dd, t2, out, t1, t3, tid, nt);
}

__device__ __forceinline__ double hsum32(double v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
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
    float loss_weight;           // get_contrastive_cos_loss(weight, ...): the loss and its gradients are scaled by it
    int phase;                   // 0: means; 1: S11^-1/2 | S22^-1/2 (2 workgroups); 3: eigh(TT') | eigh(T'T) (2 workgroups);
                                 // 4: U, V, sign fix, outputs; backward 32x32 chain: 2 (loss, dU/dV, dE/dF), 5 (EighGrad of
                                 // E | F, 2 workgroups), 6 (dT, dS12, dS11si/dS22si), 7 (S^-1/2 backward, 2 workgroups), 8 (means)
    int loss_blocks;             // partial loss sums written by loss_rows_kernel
    int row_blocks;              // 32-row blocks of ct_cov_kernel / ct_bwd_partial_kernel
    int warm_ok;                 // warm-started Jacobi allowed (ASR_CCA_WARM=0: always from the identity)
};

// Phases 1 and 3 - the four eigen-decompositions - as a kernel of their own with a 256-thread launch bound (round 6).
// cca_hestenes_wave_on keeps 64 float64 values per lane in registers (its own and its partner's halves of a column of W
// and of V: 128 VGPRs) plus the rotation's temporaries; inside cca_train_kernel, whose bound of 1024 threads caps a
// wave at 128 VGPRs, eigh_spd was compiled with 344 bytes of scratch and every one of the ~250 rounds of a
// decomposition went through 34 scratch loads / stores and their vmcnt(0) waits (round 5: 219 + 201 us for the two
// launches with warm starts; cca_fit's solver, bound 256, needs ~130 us per COLD decomposition).  Here a wave may use 512.
__global__ __launch_bounds__(256) void cca_eigh_kernel(CcaTrainArgs a) {
    __shared__ CcaScratch S;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int B = a.B;
    double *ws = a.ws;
    typedef CcaTrainWs W;
    double *bs = ws + (size_t)W::NMAT * DD + (size_t)W::NVEC * D;
    double *rowsum = bs + 8 * (size_t)B * D + 2 * (size_t)B;       // (cca_train_kernel's layout: 8 B x D arrays, nrm1, nrm2)
    const double al = (double)a.alpha, oma = 1.0 - al;
    const double cinv = 1.0 / ((double)B - 1.0);
    const float *m1in = a.cca_in + 2 * DD, *m2in = m1in + D;
    const float *S12in = m2in + D, *S11in = S12in + DD, *S22in = S11in + DD;
    const bool warm = a.warm_ok && vec(ws, W::warm)[0] == 1.0;      // uniform; set at the end of phase 4
    double *covp = rowsum + 2 * (size_t)B + (size_t)a.loss_blocks;          // [row_blocks][3*DD + 2*D]
    // The four eigen-decompositions are latency-bound Jacobi sweeps (one barrier per rotation round) and pairwise
    // independent: S11 | S22, then TT' | T'T run as two workgroups each.
    if (a.phase == 1) {
        // ---- covariances (:117-141): block-ordered reduction of ct_cov_kernel's partials.  Workgroup 0: S11 and S12,
        // workgroup 1: S22
        const int side = blockIdx.x;
        for (int e = tid; e < 3 * DD; e += nt) {
            const int m = e / DD, idx = e - m * DD, i = idx / D, j = idx - i * D;
            if ((side == 0) != (m != 1)) continue;
            // sixteen partials in flight, then added in block order (the same sum as a plain loop, which waited for one
            // L2 round trip per partial: with 256 threads - eight elements each - that was a third of this launch)
            double s = 0.0;
            for (int b0 = 0; b0 < a.row_blocks; b0 += 16) {
                double pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    pv[u] = b0 + u < a.row_blocks ? covp[(size_t)(b0 + u) * (3 * DD + 2 * D) + e] : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) s += pv[u];
            }
            s *= cinv;
            if (m == 0 && i == j) s += (double)a.r1;
            if (m == 1 && i == j) s += (double)a.r2;
            const float *run = (m == 0) ? S11in : (m == 1 ? S22in : S12in);
            mat(ws, m == 0 ? W::S11 : (m == 1 ? W::S22 : W::S12))[idx] = oma * (double)run[idx] + al * s;
        }
        __syncthreads();
        // ---- S11^-1/2 | S22^-1/2 (:144-147)
        if (side == 0) {
            eigh_spd(S, mat(ws, W::S11), vec(ws, W::d1), mat(ws, W::tmpA), tid, nt, warm, a.r1);       // tmpA = A1
            inv_sqrt_from_eig(vec(ws, W::d1), mat(ws, W::tmpA), mat(ws, W::S11si), tid, nt);
        } else {
            eigh_spd(S, mat(ws, W::S22), vec(ws, W::d2), mat(ws, W::tmpB), tid, nt, warm, a.r2);       // tmpB = A2
            inv_sqrt_from_eig(vec(ws, W::d2), mat(ws, W::tmpB), mat(ws, W::S22si), tid, nt);
        }
        return;
    }
    if (a.phase == 3) {
        // ---- T, then M1 = TT' + rT | M2 = T'T + rT and their eigen-decompositions (:150-158); each workgroup
        // computes its own copy of T (workgroup 0's is the one the backward pass reads)
        const int side = blockIdx.x;
        double *Tm = mat(ws, side == 0 ? W::T : W::T2), *tc = mat(ws, side == 0 ? W::tmpC : W::tmpC2);
        double *Mm = mat(ws, side == 0 ? W::M : W::M2);
        mm(mat(ws, W::S11si), false, mat(ws, W::S12), false, tc, tid, nt);
        mm(tc, false, mat(ws, W::S22si), false, Tm, tid, nt);
        if (side == 0) mm(Tm, false, Tm, true, Mm, tid, nt);
        else mm(Tm, true, Tm, false, Mm, tid, nt);
        for (int i = tid; i < D; i += nt) Mm[i * D + i] += (double)a.rT;
        __syncthreads();
        if (side == 0) eigh_spd(S, Mm, vec(ws, W::E1), mat(ws, W::E), tid, nt, warm, a.rT);
        else eigh_spd(S, Mm, vec(ws, W::F1), mat(ws, W::F), tid, nt, warm, a.rT);
        return;
    }
}

__global__ __launch_bounds__(CT_THREADS) void cca_train_kernel(CcaTrainArgs a) {
    __shared__ CcaScratch S;
    double *red = S.tmp;          // 1024 doubles, free outside the Jacobi solver
    const int tid = threadIdx.x, nt = blockDim.x;      // 1024, or fewer (barriers over fewer waves)
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
    const bool warm = a.warm_ok && vec(ws, W::warm)[0] == 1.0;      // uniform; set at the end of phase 4

    // partial buffers written by the multi-workgroup kernels (after lpart)
    double *covp = rowsum + 2 * (size_t)B + (size_t)a.loss_blocks;          // [row_blocks][3*DD + 2*D]
    double *duvp = covp + (size_t)a.row_blocks * (3 * DD + 2 * D);          // [row_blocks][2*DD + 2*D]
    if (a.phase == 0) {
        // ---- means (cca.py:94-106): nt / 64 interleaved row subsets per column, summed through LDS in a fixed order
        // (one thread per column walking all B rows was 117 us of dependent loads)
        {
            const int c = tid & 63, part = tid >> 6, parts = nt >> 6;
            const float *H = c < D ? a.H1 : a.H2;
            const int cc = c & (D - 1);
            double s = 0.0;
            for (int n = part; n < B; n += parts) s += (double)H[(size_t)n * D + cc];
            red[tid] = s;
        }
        __syncthreads();
        for (int c = tid; c < 2 * D; c += nt) {
            const int cc = c & (D - 1);
            double s = 0.0;
            for (int q = 0; q < (nt >> 6); ++q) s += red[q * 64 + c];
            const double run = (double)(c < D ? m1in[cc] : m2in[cc]);
            vec(ws, c < D ? W::mean1 : W::mean2)[cc] = oma * run + al * (s / (double)B);
        }
        return;
    }
    if (a.phase == 4) {
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
    if (tid == 0) vec(ws, W::warm)[0] = 1.0;                  // tmpA, tmpB, E, F hold eigenvectors from now on
        return;
    }   // phase 4
    if (a.phase == 2) {
    // ---- phase 2.  The pair passes ran as multi-workgroup kernels (loss_rows_kernel / loss_cols_kernel): finish the loss
    {
        double *lpart = rowsum + 2 * (size_t)B;            // after rowsum[B], diag[B]
        double sacc = 0.0;
        for (int b = tid; b < a.loss_blocks; b += nt) sacc += lpart[b];
        red[tid] = sacc;
        __syncthreads();
        for (int st = nt / 2; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) a.loss_out[0] = (float)((double)a.loss_weight * red[0] / ((double)B * ((double)B - 1.0)));
        __syncthreads();
    }
    if (a.dH1 == nullptr) return;                                 // forward only (uniform branch)
    // ---- dU = Hb1^T dout1, dV = Hb2^T dout2: reduce ct_bwd_partial_kernel's partials (it also did the
    // length-norm backward in place); column sums of dout / Hb for the mean term of dH
    // (partials fetched sixteen at a time, added in block order: see cca_eigh_kernel's covariance reduction)
    for (int e = tid; e < 2 * DD; e += nt) {
        double sacc = 0.0;
        for (int b0 = 0; b0 < a.row_blocks; b0 += 16) {
            double pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) pv[u] = b0 + u < a.row_blocks ? duvp[(size_t)(b0 + u) * (2 * DD + 2 * D) + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u) sacc += pv[u];
        }
        mat(ws, e < DD ? W::dU : W::dV)[e & (DD - 1)] = sacc;
    }
    for (int c = tid; c < 2 * D; c += nt) {
        double sd = 0.0, sh = 0.0;
        for (int b0 = 0; b0 < a.row_blocks; b0 += 16) {
            double pd[16], ph[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool in = b0 + u < a.row_blocks;
                pd[u] = in ? duvp[(size_t)(b0 + u) * (2 * DD + 2 * D) + 2 * DD + c] : 0.0;
                ph[u] = in ? covp[(size_t)(b0 + u) * (3 * DD + 2 * D) + 3 * DD + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { sd += pd[u]; sh += ph[u]; }
        }
        vec(ws, c < D ? W::sdout1 : W::sdout2)[c & (D - 1)] = sd;
        vec(ws, c < D ? W::shb1 : W::shb2)[c & (D - 1)] = sh;
    }
    __syncthreads();
    // dU0 = dU * s (in place)
    for (int e = tid; e < DD; e += nt) mat(ws, W::dU)[e] *= vec(ws, W::sgn)[e & (D - 1)];
    __syncthreads();
    mm(mat(ws, W::dU), false, mat(ws, W::E), true, mat(ws, W::dS11si), tid, nt);       // dU0 E^T
    mm(mat(ws, W::S11si), true, mat(ws, W::dU), false, mat(ws, W::dE), tid, nt);       // S11si^T dU0
    mm(mat(ws, W::dV), false, mat(ws, W::F), true, mat(ws, W::dS22si), tid, nt);
    mm(mat(ws, W::S22si), true, mat(ws, W::dV), false, mat(ws, W::dF), tid, nt);
    return;
    }   // phase 2
    // The two EighGrad's and the two S^-1/2 backward passes are pairwise independent chains of 32x32 products: each
    // pair runs as two workgroups (with temporaries of their own), like the eigen-decompositions of the forward pass.
    if (a.phase == 5) {
        if (blockIdx.x == 0)
            eigh_grad(vec(ws, W::E1), mat(ws, W::E), nullptr, mat(ws, W::dE), mat(ws, W::dM1), mat(ws, W::tmpC),
                      mat(ws, W::M), tid, nt);
        else
            eigh_grad(vec(ws, W::F1), mat(ws, W::F), nullptr, mat(ws, W::dF), mat(ws, W::dM2), mat(ws, W::tmpC2),
                      mat(ws, W::M2), tid, nt);
        return;
    }
    if (a.phase == 6) {
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
    return;
    }   // phase 6
    if (a.phase == 7) {
        if (blockIdx.x == 0)
            inv_sqrt_bwd(vec(ws, W::d1), mat(ws, W::tmpA), mat(ws, W::dS11si), mat(ws, W::dS11), vec(ws, W::vtmp),
                         mat(ws, W::tmpC), mat(ws, W::M), mat(ws, W::dM1), tid, nt);
        else
            inv_sqrt_bwd(vec(ws, W::d2), mat(ws, W::tmpB), mat(ws, W::dS22si), mat(ws, W::dS22), vec(ws, W::vtmp2),
                         mat(ws, W::tmpC2), mat(ws, W::M2), mat(ws, W::dM2), tid, nt);
        return;
    }
    // ---- phase 8
    // ---- column means of dHb = dout U^T + alpha c ( Hb (dS + dS^T) + Hb_other dS12(^T) ), from the column sums
    const double ac = al * cinv;
    for (int c = tid; c < 2 * D; c += nt) {
        const int i = c & (D - 1);
        const bool first = c < D;
        double sacc = 0.0;
        for (int k = 0; k < D; ++k) {
            if (first)
                sacc += vec(ws, W::sdout1)[k] * mat(ws, W::U)[i * D + k] +
                        ac * (vec(ws, W::shb1)[k] * (mat(ws, W::dS11)[k * D + i] + mat(ws, W::dS11)[i * D + k]) +
                              vec(ws, W::shb2)[k] * mat(ws, W::dS12)[i * D + k]);
            else
                sacc += vec(ws, W::sdout2)[k] * mat(ws, W::V)[i * D + k] +
                        ac * (vec(ws, W::shb2)[k] * (mat(ws, W::dS22)[k * D + i] + mat(ws, W::dS22)[i * D + k]) +
                              vec(ws, W::shb1)[k] * mat(ws, W::dS12)[k * D + i]);
        }
        vec(ws, first ? W::cmean1 : W::cmean2)[i] = sacc / (double)B;
    }
}

// ---- multi-workgroup row kernels of the CCALayer stage (32 rows per workgroup) --------------------------------
struct RowPtrs {
    double *Hb1, *Hb2, *o1, *o2, *l1, *l2, *g1, *g2, *nrm1, *nrm2, *covp, *duvp;
};
__device__ __forceinline__ RowPtrs row_ptrs(double *ws, int B, int loss_blocks, int row_blocks) {
    double *bs = ws + (size_t)CcaTrainWs::NMAT * DD + (size_t)CcaTrainWs::NVEC * D;
    RowPtrs p;
    p.Hb1 = bs; p.Hb2 = bs + (size_t)B * D; p.o1 = p.Hb2 + (size_t)B * D; p.o2 = p.o1 + (size_t)B * D;
    p.l1 = p.o2 + (size_t)B * D; p.l2 = p.l1 + (size_t)B * D; p.g1 = p.l2 + (size_t)B * D; p.g2 = p.g1 + (size_t)B * D;
    p.nrm1 = p.g2 + (size_t)B * D; p.nrm2 = p.nrm1 + B;
    p.covp = p.nrm2 + B + 2 * (size_t)B + loss_blocks;
    p.duvp = p.covp + (size_t)row_blocks * (3 * DD + 2 * D);
    return p;
}

// centre the batch (cca.py:109-110) and per-block second moments + column sums
__global__ __launch_bounds__(256) void ct_cov_kernel(const float *__restrict__ H1, const float *__restrict__ H2, double *ws,
                                                     int B, int loss_blocks, int row_blocks) {
    __shared__ double a[32][33], b[32][33];
    const RowPtrs p = row_ptrs(ws, B, loss_blocks, row_blocks);
    const double *m1 = vec(ws, CcaTrainWs::mean1), *m2 = vec(ws, CcaTrainWs::mean2);
    const int tid = threadIdx.x, r0 = blockIdx.x * 32;
    for (int e = tid; e < 32 * D; e += 256) {
        const int r = e >> 5, c = e & 31, n = r0 + r;
        double va = 0.0, vb = 0.0;
        if (n < B) {
            va = (double)H1[(size_t)n * D + c] - m1[c];
            vb = (double)H2[(size_t)n * D + c] - m2[c];
            p.Hb1[(size_t)n * D + c] = va;
            p.Hb2[(size_t)n * D + c] = vb;
        }
        a[r][c] = va; b[r][c] = vb;
    }
    __syncthreads();
    double *out = p.covp + (size_t)blockIdx.x * (3 * DD + 2 * D);
    for (int e = tid; e < 3 * DD; e += 256) {
        const int m = e / DD, idx = e - m * DD, i = idx / D, j = idx - i * D;
        double sacc = 0.0;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) sacc += (m == 1 ? b[r][i] : a[r][i]) * (m == 0 ? a[r][j] : b[r][j]);
        out[e] = sacc;
    }
    if (tid < 2 * D) {
        double sacc = 0.0;
        for (int r = 0; r < 32; ++r) sacc += tid < D ? a[r][tid] : b[r][tid - D];
        out[3 * DD + tid] = sacc;
    }
}

// projections + length norm (cca.py:198-201, 39-40): 32-lane group per row, lane j = output dimension
__global__ __launch_bounds__(256) void ct_project_kernel(double *ws, int B, int loss_blocks, int row_blocks,
                                                         float *__restrict__ lv1, float *__restrict__ lv2) {
    const RowPtrs p = row_ptrs(ws, B, loss_blocks, row_blocks);
    const double *U = mat(ws, CcaTrainWs::U), *V = mat(ws, CcaTrainWs::V);
    const int grp = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int n = blockIdx.x * 8 + grp;
    if (n >= B) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < D; ++k) {
        s1 += p.Hb1[(size_t)n * D + k] * U[k * D + j];
        s2 += p.Hb2[(size_t)n * D + k] * V[k * D + j];
    }
    const double n1 = sqrt(hsum32(s1 * s1)), n2 = sqrt(hsum32(s2 * s2));
    p.o1[(size_t)n * D + j] = s1; p.o2[(size_t)n * D + j] = s2;
    p.l1[(size_t)n * D + j] = s1 / n1; p.l2[(size_t)n * D + j] = s2 / n2;
    if (j == 0) { p.nrm1[n] = n1; p.nrm2[n] = n2; }
    if (lv1) lv1[(size_t)n * D + j] = (float)(s1 / n1);
    if (lv2) lv2[(size_t)n * D + j] = (float)(s2 / n2);
}

// length-norm backward in place (g <- dout) and per-block Hb^T dout + column sums of dout
__global__ __launch_bounds__(256) void ct_bwd_partial_kernel(double *ws, int B, int loss_blocks, int row_blocks) {
    __shared__ double ha[32][33], hb[32][33], ga[32][33], gb[32][33];
    const RowPtrs p = row_ptrs(ws, B, loss_blocks, row_blocks);
    const int tid = threadIdx.x, r0 = blockIdx.x * 32;
    const int grp = tid >> 5, k = tid & 31;
    for (int rr = grp; rr < 32; rr += 8) {            // 8 rows at a time, lane k = component
        const int n = r0 + rr;
        double d1 = 0.0, d2 = 0.0, h1 = 0.0, h2 = 0.0;
        if (n < B) {
            const double g1k = p.g1[(size_t)n * D + k], g2k = p.g2[(size_t)n * D + k];
            const double l1k = p.l1[(size_t)n * D + k], l2k = p.l2[(size_t)n * D + k];
            const double dot1 = hsum32(l1k * g1k), dot2 = hsum32(l2k * g2k);
            d1 = (g1k - l1k * dot1) / p.nrm1[n];
            d2 = (g2k - l2k * dot2) / p.nrm2[n];
            p.g1[(size_t)n * D + k] = d1;
            p.g2[(size_t)n * D + k] = d2;
            h1 = p.Hb1[(size_t)n * D + k]; h2 = p.Hb2[(size_t)n * D + k];
        }
        ha[rr][k] = h1; hb[rr][k] = h2; ga[rr][k] = d1; gb[rr][k] = d2;
    }
    __syncthreads();
    double *out = p.duvp + (size_t)blockIdx.x * (2 * DD + 2 * D);
    for (int e = tid; e < 2 * DD; e += 256) {
        const int m = e / DD, idx = e - m * DD, i = idx / D, j = idx - i * D;
        double sacc = 0.0;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) sacc += (m ? hb[r][i] * gb[r][j] : ha[r][i] * ga[r][j]);
        out[e] = sacc;
    }
    if (tid < 2 * D) {
        double sacc = 0.0;
        for (int r = 0; r < 32; ++r) sacc += tid < D ? ga[r][tid] : gb[r][tid - D];
        out[2 * DD + tid] = sacc;
    }
}

// dH = dout U^T + alpha c ( Hb (dS + dS^T) + Hb_other dS12(^T) ) - alpha * column mean
__global__ __launch_bounds__(256) void ct_dH_kernel(double *ws, int B, int loss_blocks, int row_blocks, float alpha,
                                                    float *__restrict__ dH1, float *__restrict__ dH2) {
    const RowPtrs p = row_ptrs(ws, B, loss_blocks, row_blocks);
    typedef CcaTrainWs W;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= B * D) return;
    const int n = e / D, i = e - n * D;
    const double al = (double)alpha, ac = al / ((double)B - 1.0);
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < D; ++k) {
        s1 += p.g1[(size_t)n * D + k] * mat(ws, W::U)[i * D + k];
        s2 += p.g2[(size_t)n * D + k] * mat(ws, W::V)[i * D + k];
        s1 += ac * (p.Hb1[(size_t)n * D + k] * (mat(ws, W::dS11)[k * D + i] + mat(ws, W::dS11)[i * D + k]) +
                    p.Hb2[(size_t)n * D + k] * mat(ws, W::dS12)[i * D + k]);
        s2 += ac * (p.Hb2[(size_t)n * D + k] * (mat(ws, W::dS22)[k * D + i] + mat(ws, W::dS22)[i * D + k]) +
                    p.Hb1[(size_t)n * D + k] * mat(ws, W::dS12)[k * D + i]);
    }
    dH1[e] = (float)(s1 - al * vec(ws, W::cmean1)[i]);
    dH2[e] = (float)(s2 - al * vec(ws, W::cmean2)[i]);
}

// ---- pair passes of the ranking loss as multi-workgroup kernels (objectives.py:36-50) --------------------------
// A 32-lane group owns one row (lane k = component k; dot products by xor-shuffles inside the half-wave); 8 rows
// per 256-thread workgroup.  Same float64 workspace layout as cca_train_kernel.
struct PairPtrs {
    double *l1, *l2, *g1, *g2, *rowsum, *diag, *lpart;
};
__device__ __forceinline__ PairPtrs pair_ptrs(double *ws, int B) {
    double *bs = ws + (size_t)CcaTrainWs::NMAT * DD + (size_t)CcaTrainWs::NVEC * D;
    PairPtrs p;
    p.l1 = bs + (size_t)4 * B * D; p.l2 = p.l1 + (size_t)B * D; p.g1 = p.l2 + (size_t)B * D; p.g2 = p.g1 + (size_t)B * D;
    p.rowsum = p.g2 + (size_t)B * D + 2 * (size_t)B; p.diag = p.rowsum + B; p.lpart = p.diag + B;
    return p;
}

// One 32-lane group per row i; lane t walks the columns j = t, t + 32, ... and evaluates each pair on its own: a
// 32-term float64 dot product in registers, then - for the pairs inside the margin - 32 adds into its private copy of
// the row's gradient.  The 32 private copies are summed once per row through LDS, in lane order (deterministic).
// (The first version gave lane k component k and reduced every pair's dot product with five float64 xor-shuffles:
// 512 dependent shuffle chains per row, 0.23 ms per pass at batch 512; this form runs the same pass in ~20 us.)
// second = true: the transposed direction of get_contrastive_cos_loss(symmetric=True) (objectives.py:53-65: D = lv2 lv1^T) -
// the same pass with the two views' roles swapped, its gradients and loss partials ADDED to the first direction's.
// add_grad: the second direction ADDS its row gradient to what the first direction's loss_cols_kernel pass stored in the
// same buffer; in a forward-only call (no loss_cols passes ran) that buffer is uninitialised workspace, so the value is
// stored plainly there (it is not consumed) instead of read-modified-written.
__global__ __launch_bounds__(256) void loss_rows_kernel(double *ws, int B, float gamma, float weight, bool second, bool add_grad) {
    __shared__ double red[8];
    __shared__ double part[8][32][33];
    PairPtrs p = pair_ptrs(ws, B);
    if (second) { double *t_ = p.l1; p.l1 = p.l2; p.l2 = t_; t_ = p.g1; p.g1 = p.g2; p.g2 = t_; }
    const int grp = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int i = blockIdx.x * 8 + grp;
    const double gam = (double)gamma, wpair = (double)weight / ((double)B * ((double)B - 1.0));
    double lpart = 0.0, rs = 0.0;
    double acc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] = 0.0;
    double dii = 0.0;
    __shared__ double chunk[64][33];
    double li[D];
#pragma unroll
    for (int k = 0; k < D; ++k) li[k] = 0.0;
    if (i < B) {
#pragma unroll
        for (int k = 0; k < D; ++k) li[k] = p.l1[(size_t)i * D + k];
#pragma unroll
        for (int k = 0; k < D; ++k) dii += li[k] * p.l2[(size_t)i * D + k];
    }
    // the other view's rows travel through LDS in chunks of 64 (coalesced loads, row pitch 33): a lane reading ITS row
    // from global memory made every load instruction touch 32 different lines - 33 us for a 512 x 512 pass.  Lane t
    // still takes the rows t, t + 32, ... in ascending order: the same sums.
    for (int j0 = 0; j0 < B; j0 += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * D; e += 256) {
            const int r = e / D, c = e - r * D;
            chunk[r][c] = j0 + r < B ? p.l2[(size_t)(j0 + r) * D + c] : 0.0;
        }
        __syncthreads();
        if (i >= B) continue;
        for (int jj = t; jj < 64 && j0 + jj < B; jj += 32) {
            const int j = j0 + jj;
            double lj[D];
#pragma unroll
            for (int k = 0; k < D; ++k) lj[k] = chunk[jj][k];
            double dij = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) dij += li[k] * lj[k];
            if (j == i) continue;
            const double L = gam - dii + dij;
            if (L >= 0.0 && L <= 1000.0) {
                rs += 1.0;
#pragma unroll
                for (int k = 0; k < D; ++k) acc[k] += lj[k];
            }
            lpart += L < 0.0 ? 0.0 : (L > 1000.0 ? 1000.0 : L);
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) part[grp][t][k] = acc[k];
    // the scalars travel in the padding column
    part[grp][t][32] = rs;
    __syncthreads();
    if (i < B) {
        double a = 0.0, r = 0.0;
        for (int q = 0; q < 32; ++q) { a += part[grp][q][t]; r += part[grp][q][32]; }      // lane t = component t
        const double gi = wpair * (a - r * p.l2[(size_t)i * D + t]);
        p.g1[(size_t)i * D + t] = add_grad ? p.g1[(size_t)i * D + t] + gi : gi;
        if (t == 0) { p.rowsum[i] = r; p.diag[i] = dii; }
    }
    lpart = hsum32(lpart);
    if (t == 0) red[grp] = lpart;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int q = 0; q < 8; ++q) s += red[q];
        p.lpart[blockIdx.x] = second ? p.lpart[blockIdx.x] + s : s;
    }
}

// dlv2_j = wpair * ( sum_{i != j} M_ij lv1_i - rowsum_j lv1_j ); same decomposition with the roles of rows and columns
// swapped: group = column j, lane t walks the rows i = t, t + 32, ...
__global__ __launch_bounds__(256) void loss_cols_kernel(double *ws, int B, float gamma, float weight, bool second) {
    __shared__ double part[8][32][33];
    PairPtrs p = pair_ptrs(ws, B);
    if (second) { double *t_ = p.l1; p.l1 = p.l2; p.l2 = t_; t_ = p.g1; p.g1 = p.g2; p.g2 = t_; }
    const int grp = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int j = blockIdx.x * 8 + grp;
    const double gam = (double)gamma, wpair = (double)weight / ((double)B * ((double)B - 1.0));
    double acc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) acc[k] = 0.0;
    __shared__ double chunk[64][33];
    __shared__ double dchunk[64];
    double lj[D];
#pragma unroll
    for (int k = 0; k < D; ++k) lj[k] = j < B ? p.l2[(size_t)j * D + k] : 0.0;
    for (int i0 = 0; i0 < B; i0 += 64) {                   // (rows through LDS in chunks of 64: see loss_rows_kernel)
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * D; e += 256) {
            const int r = e / D, c = e - r * D;
            chunk[r][c] = i0 + r < B ? p.l1[(size_t)(i0 + r) * D + c] : 0.0;
        }
        if (threadIdx.x < 64) dchunk[threadIdx.x] = i0 + threadIdx.x < B ? p.diag[i0 + threadIdx.x] : 0.0;
        __syncthreads();
        if (j >= B) continue;
        for (int ii = t; ii < 64 && i0 + ii < B; ii += 32) {
            const int i = i0 + ii;
            double li[D];
#pragma unroll
            for (int k = 0; k < D; ++k) li[k] = chunk[ii][k];
            double dij = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) dij += li[k] * lj[k];
            if (i == j) continue;
            const double L = gam - dchunk[ii] + dij;
            if (L >= 0.0 && L <= 1000.0) {
#pragma unroll
                for (int k = 0; k < D; ++k) acc[k] += li[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < D; ++k) part[grp][t][k] = acc[k];
    __syncthreads();
    if (j < B) {
        double a = 0.0;
        for (int q = 0; q < 32; ++q) a += part[grp][q][t];
        const double gj = wpair * (a - p.rowsum[j] * p.l1[(size_t)j * D + t]);
        p.g2[(size_t)j * D + t] = second ? p.g2[(size_t)j * D + t] + gj : gj;
    }
}

// iter_funcs['valid'] (utils/train_dcca_pool.py:155): ranking loss of deterministic outputs, no gradients
__global__ __launch_bounds__(1024) void rank_loss_kernel(const float *__restrict__ lv1, const float *__restrict__ lv2, int B,
                                                         float gamma, float weight, int symmetric,
                                                         float *__restrict__ loss_out) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, grp = tid >> 5, k = tid & 31;
    auto hsum = [](double v) {
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        return v;
    };
    double lpart = 0.0;
    for (int dir = 0; dir < (symmetric ? 2 : 1); ++dir) {       // direction 2 (objectives.py:53-65): the views swapped
        const float *a = dir ? lv2 : lv1, *b = dir ? lv1 : lv2;
        for (int i = grp; i < B; i += 32) {
            const double lik = (double)a[(size_t)i * D + k];
            const double dii = hsum(lik * (double)b[(size_t)i * D + k]);
            for (int j = 0; j < B; ++j) {
                const double dij = hsum(lik * (double)b[(size_t)j * D + k]);
                if (j == i || k != 0) continue;
                const double L = (double)gamma - dii + dij;
                lpart += L < 0.0 ? 0.0 : (L > 1000.0 ? 1000.0 : L);
            }
        }
    }
    red[tid] = lpart;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) loss_out[0] = (float)((double)weight * red[0] / ((double)B * ((double)B - 1.0)));
}

hipError_t launch_rank_loss(hipStream_t s, const float *lv1, const float *lv2, int B, float gamma, float *loss_out,
                            float weight, int symmetric) {
    rank_loss_kernel<<<1, 1024, 0, s>>>(lv1, lv2, B, gamma, weight, symmetric, loss_out);
    return hipGetLastError();
}

size_t cca_train_ws_bytes(int B) {
    return ((size_t)CcaTrainWs::NMAT * DD + (size_t)CcaTrainWs::NVEC * D + (size_t)8 * B * D + (size_t)4 * B + (size_t)(B + 7) / 8 + (size_t)((B + 31) / 32) * (5 * DD + 4 * D)) *
           sizeof(double);
}

hipError_t launch_cca_train(hipStream_t s, const float *H1, const float *H2, int B, const float *cca_in,
                            float *cca_out, float r1, float r2, float rT, float alpha, float gamma, void *ws,
                            float *loss_out, float *lv1, float *lv2, float *dH1, float *dH2, float weight, int symmetric) {
    CcaTrainArgs a;
    a.loss_weight = weight;
    a.H1 = H1; a.H2 = H2; a.cca_in = cca_in; a.cca_out = cca_out; a.dH1 = dH1; a.dH2 = dH2;
    a.lv1 = lv1; a.lv2 = lv2; a.loss_out = loss_out; a.ws = (double *)ws; a.B = B;
    a.r1 = r1; a.r2 = r2; a.rT = rT; a.alpha = alpha; a.gamma = gamma;
    const int lb = (B + 7) / 8, rb = (B + 31) / 32;
    a.loss_blocks = lb;
    a.row_blocks = rb;
    static const int warm_ok = !(getenv("ASR_CCA_WARM") && getenv("ASR_CCA_WARM")[0] == '0');
    a.warm_ok = warm_ok;
    double *w = (double *)ws;
    a.phase = 0;
    // threads per workgroup: the 32x32 float64 algebra is a chain of short loops separated by barriers, and the
    // Jacobi solver uses 256 threads - a barrier over 4 waves costs less than one over 16
    static const int cth = getenv("ASR_CCA_THREADS") ? std::max(256, std::min(1024, atoi(getenv("ASR_CCA_THREADS")) / 64 * 64)) : CT_THREADS_DEFAULT;
    cca_train_kernel<<<1, cth, 0, s>>>(a);                         // batch means
    ct_cov_kernel<<<rb, 256, 0, s>>>(H1, H2, w, B, lb, rb);               // centring + second-moment partials
    a.phase = 1;
    cca_eigh_kernel<<<2, 256, 0, s>>>(a);                          // covariance reduction, S11^-1/2 | S22^-1/2
    a.phase = 3;
    cca_eigh_kernel<<<2, 256, 0, s>>>(a);                          // T, eigh(TT') | eigh(T'T)
    a.phase = 4;
    cca_train_kernel<<<1, cth, 0, s>>>(a);                         // U, V, sign fix, running values, corr
    ct_project_kernel<<<lb, 256, 0, s>>>(w, B, lb, rb, lv1, lv2);         // projections + length norm
    loss_rows_kernel<<<lb, 256, 0, s>>>(w, B, gamma, weight, false, false);
    if (dH1 != nullptr) loss_cols_kernel<<<lb, 256, 0, s>>>(w, B, gamma, weight, false);
    if (symmetric) {                                                      // direction 2: roles swapped, results added
        loss_rows_kernel<<<lb, 256, 0, s>>>(w, B, gamma, weight, true, dH1 != nullptr);
        if (dH1 != nullptr) loss_cols_kernel<<<lb, 256, 0, s>>>(w, B, gamma, weight, true);
    }
    if (dH1 != nullptr) ct_bwd_partial_kernel<<<rb, 256, 0, s>>>(w, B, lb, rb);   // length-norm backward, dU/dV partials
    a.phase = 2;
    cca_train_kernel<<<1, cth, 0, s>>>(a);                         // loss sum; dU, dV, dE, dF
    if (dH1 != nullptr) {
        a.phase = 5;
        cca_train_kernel<<<2, cth, 0, s>>>(a);                     // EighGrad of E | F
        a.phase = 6;
        cca_train_kernel<<<1, cth, 0, s>>>(a);                     // dT, dS12, dS11si, dS22si
        a.phase = 7;
        cca_train_kernel<<<2, cth, 0, s>>>(a);                     // S11^-1/2 | S22^-1/2 backward
        a.phase = 8;
        cca_train_kernel<<<1, cth, 0, s>>>(a);                     // column means of dHb
        ct_dH_kernel<<<(B * D + 255) / 256, 256, 0, s>>>(w, B, lb, rb, alpha, dH1, dH2);
    }
    return hipGetLastError();
}

}  // namespace asr
