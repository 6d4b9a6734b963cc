// gfx950 kernels for the CCA re-estimation of refine_cca.py:95-107
// (CCA('svd').fit, utils/cca.py:25-53,199-211) - BASELINE config 4.
//
//   cca_colsum_kernel  : per-block column sums of H1, H2 (float64 partials)
//   cca_means_kernel   : block-ordered reduction -> float32 means (utils/cca.py:31-32)
//   cca_cov_kernel     : per-block centred second moments S11, S22, S12 of a row
//                        chunk staged in LDS (float32 centring like :35-36, exact
//                        float64 accumulation of the float32 products)
//   cca_solve_kernel   : one workgroup: reduces the partials in block order
//                        (deterministic, no atomics), scales/regularises like
//                        :43-52, then the float64 32x32 algebra of :201-211 as
//                        one-sided Jacobi in LDS (cca_solve.inl).
// Memory-bound on 2*N*32*4 bytes read twice; HBM roofline.
#include "asr_kernels.h"

#define CCA_FN __device__
#define CCA_SYNC() __syncthreads()
struct CcaScratch;
__device__ inline int cca_hestenes_fast(CcaScratch &S, int tid);
#define CCA_HESTENES(S, tid, nt) cca_hestenes_fast(S, tid)
__device__ inline void cca_inv_sqrt_pair_fast(CcaScratch &S, const double *S11, const double *S22, int tid, int nt);
#define CCA_INV_SQRT_PAIR(S, S11, S22, tid, nt) cca_inv_sqrt_pair_fast(S, S11, S22, tid, nt)
#include "cca_solve.inl"
#include "cca_hestenes_fast.inl"

namespace asr {

constexpr int CCA_THREADS = 256;
constexpr int CCA_ROWS = 128;          // rows per block in the covariance pass

__global__ __launch_bounds__(CCA_THREADS) void cca_colsum_kernel(const float *__restrict__ H1,
                                                                 const float *__restrict__ H2, int64_t n,
                                                                 int64_t rows_per_block,
                                                                 double *__restrict__ partial) {
    __shared__ double red[8][64];
    const int tid = threadIdx.x, c = tid & 31, r = tid >> 5;
    const int64_t lo = (int64_t)blockIdx.x * rows_per_block;
    const int64_t hi = lo + rows_per_block < n ? lo + rows_per_block : n;
    double a1 = 0.0, a2 = 0.0;
    for (int64_t i = lo + r; i < hi; i += 8) {
        a1 += (double)H1[i * 32 + c];
        a2 += (double)H2[i * 32 + c];
    }
    red[r][c] = a1;
    red[r][32 + c] = a2;
    __syncthreads();
    if (tid < 64) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][tid];
        partial[(size_t)blockIdx.x * 64 + tid] = s;
    }
}

// 64 values x 16 interleaved subsets of the blocks, fixed-order finish (one thread per value walking all partials is
// a chain of dependent global loads)
__global__ __launch_bounds__(1024) void cca_means_kernel(const double *__restrict__ partial, int nblocks, int64_t n,
                                                         float *__restrict__ means /*[64]: m1 | m2*/) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, o = tid & 63, part = tid >> 6;
    double s = 0.0;
    for (int b = part; b < nblocks; b += 16) s += partial[(size_t)b * 64 + o];
    red[tid] = s;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 64 + o];
        means[o] = (float)(t / (double)n);
    }
}

// partial [nblocks][3][1024] -> reduced [3][1024]: 64 values per workgroup x 16 interleaved subsets of the blocks
__global__ __launch_bounds__(1024) void cca_cov_reduce_kernel(const double *__restrict__ partial, int nblocks,
                                                              double *__restrict__ reduced) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, o = tid & 63, part = tid >> 6;
    const int e = blockIdx.x * 64 + o;
    double s = 0.0;
#pragma unroll 4
    for (int b = part; b < nblocks; b += 16) s += partial[(size_t)b * 3072 + e];
    red[tid] = s;
    __syncthreads();
    if (part == 0) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 64 + o];
        reduced[e] = t;
    }
}

__global__ __launch_bounds__(CCA_THREADS) void cca_cov_kernel(const float *__restrict__ H1,
                                                              const float *__restrict__ H2, int64_t n,
                                                              const float *__restrict__ means,
                                                              double *__restrict__ partial /*[blocks][3][1024]*/) {
    __shared__ float a[CCA_ROWS][33];       // +1 pad: column reads hit distinct banks
    __shared__ float b[CCA_ROWS][33];
    const int tid = threadIdx.x;
    const int64_t lo = (int64_t)blockIdx.x * CCA_ROWS;
    const int rows = (int)((lo + CCA_ROWS < n ? lo + CCA_ROWS : n) - lo);
    for (int e = tid; e < CCA_ROWS * 32; e += CCA_THREADS) {
        const int r = e >> 5, c = e & 31;
        float va = 0.f, vb = 0.f;
        if (r < rows) {
            va = H1[(lo + r) * 32 + c] - means[c];          // float32 centring (utils/cca.py:35-36)
            vb = H2[(lo + r) * 32 + c] - means[32 + c];
        }
        a[r][c] = va;
        b[r][c] = vb;
    }
    __syncthreads();
    const int i = tid >> 3, j0 = (tid & 7) * 4;
    double s11[4] = {0, 0, 0, 0}, s22[4] = {0, 0, 0, 0}, s12[4] = {0, 0, 0, 0};
    for (int r = 0; r < rows; ++r) {
        const double ai = (double)a[r][i], bi = (double)b[r][i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double aj = (double)a[r][j0 + q], bj = (double)b[r][j0 + q];
            s11[q] += ai * aj;
            s22[q] += bi * bj;
            s12[q] += ai * bj;
        }
    }
    double *out = partial + (size_t)blockIdx.x * 3 * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        out[i * 32 + j0 + q] = s11[q];
        out[1024 + i * 32 + j0 + q] = s22[q];
        out[2048 + i * 32 + j0 + q] = s12[q];
    }
}

__global__ __launch_bounds__(CCA_THREADS) void cca_solve_kernel(const double *__restrict__ partial, int nblocks,
                                                                int64_t n, float r1, float r2,
                                                                double *__restrict__ work /*[3][1024] + [2][1024]*/,
                                                                float *__restrict__ U, float *__restrict__ V,
                                                                double *__restrict__ coeffs) {
    __shared__ CcaScratch S;
    const int tid = threadIdx.x, nt = CCA_THREADS;
    double *S11 = work, *S22 = work + 1024, *S12 = work + 2048, *Ud = work + 3072, *Vd = work + 4096;
    const float inv_m1 = (float)(1.0 / (double)(n - 1));       // python float -> float32 (weak scalar)
    for (int e = tid; e < 3 * 1024; e += nt) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * 3072 + e];
        // (1.0 / (m - 1)) * np.dot(...) stays float32 (utils/cca.py:43,47,51) ...
        const float sf = (float)s * inv_m1;
        double v = (double)sf;
        // ... "+ r * np.identity" promotes to float64 (:48,52)
        const int m = e >> 10, idx = e & 1023;
        if (m < 2 && (idx >> 5) == (idx & 31)) v += (double)(m == 0 ? r1 : r2);
        work[e] = v;
    }
    __syncthreads();
    cca_solve(S, S11, S22, S12, Ud, Vd, coeffs, tid, nt);
    for (int e = tid; e < 1024; e += nt) {                     // refine_cca.py:106-107 astype(float32)
        U[e] = (float)Ud[e];
        V[e] = (float)Vd[e];
    }
}

size_t cca_workspace_bytes(int64_t n) {
    const int64_t nb_cov = (n + CCA_ROWS - 1) / CCA_ROWS;
    const int64_t nb_sum = 256;
    return (size_t)(nb_sum * 64 + nb_cov * 3072 + 3072 + 5 * 1024) * sizeof(double);
}

hipError_t launch_cca_fit(hipStream_t s, const float *H1, const float *H2, int64_t n, float r1, float r2,
                          void *workspace, float *U, float *V, float *means, double *coeffs) {
    const int nb_sum = 256;
    const int64_t rows_per_block = (n + nb_sum - 1) / nb_sum;
    const int nb_cov = (int)((n + CCA_ROWS - 1) / CCA_ROWS);
    double *p_sum = (double *)workspace;
    double *p_cov = p_sum + (size_t)nb_sum * 64;
    double *reduced = p_cov + (size_t)nb_cov * 3072;
    double *work = reduced + 3072;
    cca_colsum_kernel<<<nb_sum, CCA_THREADS, 0, s>>>(H1, H2, n, rows_per_block, p_sum);
    cca_means_kernel<<<1, 1024, 0, s>>>(p_sum, nb_sum, n, means);
    cca_cov_kernel<<<nb_cov, CCA_THREADS, 0, s>>>(H1, H2, n, means, p_cov);
    cca_cov_reduce_kernel<<<3072 / 64, 1024, 0, s>>>(p_cov, nb_cov, reduced);
    cca_solve_kernel<<<1, CCA_THREADS, 0, s>>>(reduced, 1, n, r1, r2, work, U, V, coeffs);
    return hipGetLastError();
}

}  // namespace asr
