// gfx950 kernels for the end of the towers and for retrieval ranking.
//
//   tail_kernel : block 9 (1x1 conv + BN, identity), GlobalPoolLayer,
//                 CCALayer deterministic branch, LengthNormLayer
//                 (models/mutopia_ccal_cont.py:93-97,128-138;
//                  models/lasagne_extensions/layers/cca.py:185-201, 39-40)
//   row_norms_kernel / rank_kernel : eval_retrieval's cdist + argsort
//                 (utils/train_dcca_pool.py:28-82) as exact float64 distances
//                 and rank-by-counting; bit-exact against oracle/retrieval.py.
#include "asr_kernels.h"
#include <algorithm>

namespace asr {

// ---------------------------------------------------------------------------
// tail: one 256-thread block per sample.  thread = (pixel group pg, output o):
// o = 0..31, the 8 pixel groups stride the pixels; partial sums meet in LDS.
// ---------------------------------------------------------------------------
template <int C8>
__global__ __launch_bounds__(256) void tail_kernel(const float *__restrict__ a8, int N, int npix,
                                                   const float *__restrict__ w9, const float *__restrict__ bnp9,
                                                   const float *__restrict__ cca_mean,
                                                   const float *__restrict__ cca_proj,
                                                   float *__restrict__ features, float *__restrict__ latent) {
    __shared__ float part[8][32];
    const int n = blockIdx.x;
    if (n >= N) return;
    const int tid = threadIdx.x;
    const int o = tid & 31, pg = tid >> 5;
    float wrow[C8];
#pragma unroll
    for (int c = 0; c < C8; ++c) wrow[c] = w9[o * C8 + c];
    const float mean9 = bnp9[o], scale9 = bnp9[32 + o], beta9 = bnp9[64 + o];
    const float *img = a8 + (size_t)n * npix * C8;
    float sum = 0.0f;
    for (int p = pg; p < npix; p += 8) {
        const float4 *px = reinterpret_cast<const float4 *>(img + (size_t)p * C8);
        float z = 0.0f;
#pragma unroll
        for (int c4 = 0; c4 < C8 / 4; ++c4) {
            const float4 v = px[c4];
            z = fmaf(v.x, wrow[4 * c4], z);
            z = fmaf(v.y, wrow[4 * c4 + 1], z);
            z = fmaf(v.z, wrow[4 * c4 + 2], z);
            z = fmaf(v.w, wrow[4 * c4 + 3], z);
        }
        sum += (z - mean9) * scale9 + beta9;          // BatchNormLayer, identity nonlinearity
    }
    part[pg][o] = sum;
    __syncthreads();
    if (tid >= 32) return;                            // one half-wave finishes the sample
    float tot = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) tot += part[q][o];
    const float hfeat = tot / (float)npix;            // GlobalPoolLayer: mean over H*W
    if (features != nullptr) features[(size_t)n * 32 + o] = hfeat;
    if (latent == nullptr) return;
    // CCALayer deterministic: (H - mean) . U ; LengthNormLayer: x / ||x||_2
    const float hc = hfeat - cca_mean[o];
    float e = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) e = fmaf(__shfl(hc, k), cca_proj[k * 32 + o], e);   // source lanes 0..31: active
    float ss = e * e;
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
    latent[(size_t)n * 32 + o] = e / sqrtf(ss);
}

hipError_t launch_tail(hipStream_t s, const float *a8, int N, int h, int w, int c8, const float *w9,
                       const float *bnp9, const float *cca_mean, const float *cca_proj, float *features,
                       float *latent) {
    if (N == 0) return hipSuccess;
    const int npix = h * w;
    if (c8 == 48)
        tail_kernel<48><<<N, 256, 0, s>>>(a8, N, npix, w9, bnp9, cca_mean, cca_proj, features, latent);
    else if (c8 == 96)
        tail_kernel<96><<<N, 256, 0, s>>>(a8, N, npix, w9, bnp9, cca_mean, cca_proj, features, latent);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ranking: float64, scipy cdist_cosine operation order (two accumulators:
// even k / odd k, summed at the end; odd tail element last).  Products of
// float32 values are exact in float64, so fma vs mul+add cannot differ; the
// explicit __dadd_rn/__dmul_rn only keep the compiler from re-associating.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double dot2acc(const float *__restrict__ u, const float *__restrict__ v, int dim) {
    double a0 = 0.0, a1 = 0.0;
    const int m = dim & ~1;
    for (int k = 0; k < m; k += 2) {
        a0 = __dadd_rn(a0, __dmul_rn((double)u[k], (double)v[k]));
        a1 = __dadd_rn(a1, __dmul_rn((double)u[k + 1], (double)v[k + 1]));
    }
    double sacc = __dadd_rn(a0, a1);
    if (dim & 1) sacc = __dadd_rn(sacc, __dmul_rn((double)u[dim - 1], (double)v[dim - 1]));
    return sacc;
}

__global__ __launch_bounds__(256) void row_norms_kernel(const float *__restrict__ x, int64_t n, int64_t ld, int dim,
                                                        double *__restrict__ norms) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = x + i * ld;
    norms[i] = __dsqrt_rn(dot2acc(r, r, dim));
}

__device__ __forceinline__ double cos_dist(double dot, double na, double nb) {
    double c = __ddiv_rn(dot, __dmul_rn(na, nb));
    if (fabs(c) > 1.0) c = copysign(1.0, c);
    return __dsub_rn(1.0, c);
}

constexpr int RANK_THREADS = 256;
constexpr int RANK_MAXD = 64;

// One block per query.  Pass 1: d* and j* over the correct candidates
// (lane-strided, then reduced); pass 2: count d < d*, d == d* (before j* / all).
__global__ __launch_bounds__(RANK_THREADS) void rank_kernel(
    const float *__restrict__ lv1, const double *__restrict__ norm1, int64_t n1, int64_t ld1,
    const float *__restrict__ lv2, const double *__restrict__ norm2, int64_t n2, int64_t ld2, int dim,
    int64_t query_offset, int64_t kk, int64_t hh, int32_t *__restrict__ ranks, double *__restrict__ dstar_out,
    int32_t *__restrict__ ties_out) {
    __shared__ float q[RANK_MAXD];
    __shared__ double s_d[RANK_THREADS];
    __shared__ long long s_j[RANK_THREADS];
    __shared__ int s_less[RANK_THREADS], s_eqb[RANK_THREADS], s_eq[RANK_THREADS];
    const int64_t i = blockIdx.x;
    if (i >= n1) return;
    const int tid = threadIdx.x;
    for (int k = tid; k < dim; k += RANK_THREADS) q[k] = lv1[i * ld1 + k];
    __syncthreads();
    const double nq = norm1[i];
    const int64_t i_fixed = (i + query_offset) / hh;
    const int64_t lo = i_fixed * kk;
    const int64_t hi = (lo + kk < n2) ? lo + kk : n2;

    // pass 1: first minimum over the correct candidates
    double best = 1e300;
    long long bj = 0x7fffffffffffffffLL;
    for (int64_t j = lo + tid; j < hi; j += RANK_THREADS) {
        const double d = cos_dist(dot2acc(q, lv2 + j * ld2, dim), nq, norm2[j]);
        if (d < best) { best = d; bj = j; }      // j ascending per thread: keeps the first
    }
    s_d[tid] = best;
    s_j[tid] = bj;
    __syncthreads();
    for (int st = RANK_THREADS / 2; st > 0; st >>= 1) {
        if (tid < st) {
            const double d2 = s_d[tid + st];
            const long long j2 = s_j[tid + st];
            if (d2 < s_d[tid] || (d2 == s_d[tid] && j2 < s_j[tid])) { s_d[tid] = d2; s_j[tid] = j2; }
        }
        __syncthreads();
    }
    const double dstar = s_d[0];
    const long long jstar = s_j[0];
    __syncthreads();

    // pass 2: counts over all candidates
    int less = 0, eqb = 0, eq = 0;
    for (int64_t j = tid; j < n2; j += RANK_THREADS) {
        const double d = cos_dist(dot2acc(q, lv2 + j * ld2, dim), nq, norm2[j]);
        less += d < dstar;
        const int e = d == dstar;
        eq += e;
        eqb += e && (j < jstar);
    }
    s_less[tid] = less; s_eqb[tid] = eqb; s_eq[tid] = eq;
    __syncthreads();
    for (int st = RANK_THREADS / 2; st > 0; st >>= 1) {
        if (tid < st) {
            s_less[tid] += s_less[tid + st];
            s_eqb[tid] += s_eqb[tid + st];
            s_eq[tid] += s_eq[tid + st];
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (ranks) ranks[i] = 1 + s_less[0] + s_eqb[0];
        if (dstar_out) dstar_out[i] = dstar;
        if (ties_out) ties_out[i] = s_eq[0] - 1;
    }
}

hipError_t launch_row_norms(hipStream_t s, const float *x, int64_t n, int64_t ld, int dim, double *norms) {
    if (n == 0) return hipSuccess;
    row_norms_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, n, ld, dim, norms);
    return hipGetLastError();
}

hipError_t launch_rank(hipStream_t s, const float *lv1, const double *norm1, int64_t n1, int64_t ld1,
                       const float *lv2, const double *norm2, int64_t n2, int64_t ld2, int dim,
                       int64_t query_offset, int64_t k, int64_t h, int32_t *ranks, double *dstar, int32_t *ties) {
    if (n1 == 0) return hipSuccess;
    if (dim > RANK_MAXD) return hipErrorInvalidValue;
    rank_kernel<<<(unsigned)n1, RANK_THREADS, 0, s>>>(lv1, norm1, n1, ld1, lv2, norm2, n2, ld2, dim, query_offset, k,
                                                      h, ranks, dstar, ties);
    return hipGetLastError();
}

}  // namespace asr

// ---------------------------------------------------------------------------
// top-k retrieval against a code database (audio_sheet_server.py:530-563:
// cdist(DB, q, "cosine") -> argsort[:n_candidates]).  Exact float64 distances
// (same arithmetic as the rank kernel); (distance, index) keys ordered
// lexicographically = NumPy's stable argsort.  One workgroup per query keeps a
// sorted best-list in LDS and a candidate buffer that is filtered by the current
// k-th best key, so that after the first few chunks almost nothing is appended.
// ---------------------------------------------------------------------------
namespace asr {

constexpr int TOPK_THREADS = 256;
constexpr int TOPK_KMAX = 128;                 // k <= 128
constexpr int TOPK_CAP = 1024;                 // candidate buffer
constexpr int TOPK_SORT = 2048;                // >= TOPK_CAP + TOPK_KMAX, power of two
constexpr int TOPK_PER_THREAD = 2;             // candidates per thread per step

struct TopkKey {
    unsigned long long d;       // bits of the non-negative float64 distance (monotone as unsigned)
    long long j;                // global candidate index
};
__device__ __forceinline__ bool key_less(const TopkKey &a, const TopkKey &b) {
    return a.d < b.d || (a.d == b.d && a.j < b.j);
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(
    const float *__restrict__ db, const double *__restrict__ norm_db, int64_t n_db, int64_t ld_db,
    const float *__restrict__ qs, const double *__restrict__ norm_q, int64_t ld_q, int dim, int k,
    int64_t idx_offset, int32_t *__restrict__ idx_out, double *__restrict__ dist_out) {
    __shared__ float q[RANK_MAXD];
    __shared__ TopkKey keys[TOPK_SORT];        // [0, KMAX): best list, [KMAX, KMAX+CAP): candidates
    __shared__ int ncand;
    __shared__ TopkKey thr;                    // current k-th best key
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    for (int c = tid; c < dim; c += TOPK_THREADS) q[c] = qs[qi * ld_q + c];
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    for (int e = tid; e < TOPK_SORT; e += TOPK_THREADS) keys[e] = inf;
    if (tid == 0) { ncand = 0; thr = inf; }
    __syncthreads();
    const double nq = norm_q[qi];

    const int64_t step = (int64_t)TOPK_THREADS * TOPK_PER_THREAD;
    for (int64_t base = 0; base < n_db; base += step) {
        const TopkKey t = thr;
#pragma unroll
        for (int u = 0; u < TOPK_PER_THREAD; ++u) {
            const int64_t j = base + (int64_t)u * TOPK_THREADS + tid;
            if (j < n_db) {
                const double d = cos_dist(dot2acc(q, db + j * ld_db, dim), nq, norm_db[j]);
                TopkKey kk;
                kk.d = (unsigned long long)__double_as_longlong(d + 0.0);     // +0.0: never the -0.0 pattern
                kk.j = j + idx_offset;
                if (key_less(kk, t)) {
                    const int pos = atomicAdd(&ncand, 1);
                    keys[TOPK_KMAX + pos] = kk;        // pos < CAP: merged whenever fewer than `step` slots remain
                }
            }
        }
        __syncthreads();
        const bool last = base + step >= n_db;
        if (ncand > TOPK_CAP - (int)step || last) {
            // bitonic sort of the whole key array (best list + candidates + padding)
            for (int size = 2; size <= TOPK_SORT; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int e = tid; e < TOPK_SORT / 2; e += TOPK_THREADS) {
                        const int lo = 2 * e - (e & (stride - 1));
                        const int hi = lo + stride;
                        const bool up = (lo & size) == 0;
                        const TopkKey a = keys[lo], b = keys[hi];
                        if (key_less(b, a) == up) { keys[lo] = b; keys[hi] = a; }
                    }
                    __syncthreads();
                }
            // keep the k best, reset the rest
            for (int e = tid; e < TOPK_SORT; e += TOPK_THREADS)
                if (e >= k) keys[e] = inf;
            if (tid == 0) { ncand = 0; thr = keys[k - 1]; }
            __syncthreads();
        }
    }
    for (int e = tid; e < k; e += TOPK_THREADS) {
        const TopkKey kk = keys[e];
        const bool valid = e < n_db;
        idx_out[qi * k + e] = valid ? (int32_t)kk.j : -1;
        dist_out[qi * k + e] = valid ? __longlong_as_double((long long)kk.d) : __longlong_as_double(0x7ff0000000000000LL);
    }
}

hipError_t launch_topk(hipStream_t s, const float *db, const double *norm_db, int64_t n_db, int64_t ld_db,
                       const float *q, const double *norm_q, int64_t n_q, int64_t ld_q, int dim, int k,
                       int64_t idx_offset, int32_t *idx_out, double *dist_out) {
    if (n_q == 0) return hipSuccess;
    if (dim > RANK_MAXD || k < 1 || k > TOPK_KMAX) return hipErrorInvalidValue;
    topk_kernel<<<(unsigned)n_q, TOPK_THREADS, 0, s>>>(db, norm_db, n_db, ld_db, q, norm_q, ld_q, dim, k, idx_offset,
                                                       idx_out, dist_out);
    return hipGetLastError();
}

}  // namespace asr

// ---------------------------------------------------------------------------
// Alignment: distance matrix + DTW
// reference: utils/alignment.py:143-186 compute_alignment (cdist cosine), utils/dtw_by_dist.py:5-34 dtw_by_dist
// (accumulated cost D1[i,j] += min(D0[i,j], D0[i,j+1], D0[i+1,j])), :76-91 _traceback (argmin over
// (diagonal, up, left), first minimum wins).  float64 throughout: the distances are the SciPy-order cosine
// distances of the ranking path, the recurrence is one min3 and one add per cell - bit-exact.
// ---------------------------------------------------------------------------
namespace asr {

// D[(i+1)*(C+1) + (j+1)] = dist(a_i, b_j); first row / column: inf, D[0] = 0   (D0 of dtw_by_dist.py:17-22)
__global__ __launch_bounds__(256) void dtw_dist_kernel(const float *__restrict__ a, const double *__restrict__ na, int64_t R,
                                                       int64_t lda, const float *__restrict__ b,
                                                       const double *__restrict__ nb, int64_t C, int64_t ldb, int dim,
                                                       double *__restrict__ D, double *__restrict__ dist_out) {
    const int64_t total = (R + 1) * (C + 1);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / (C + 1), j = e - i * (C + 1);
        double v;
        if (i == 0 || j == 0) v = (i == 0 && j == 0) ? 0.0 : (double)INFINITY;
        else {
            v = cos_dist(dot2acc(a + (i - 1) * lda, b + (j - 1) * ldb, dim), na[i - 1], nb[j - 1]);
            if (dist_out) dist_out[(i - 1) * C + (j - 1)] = v;
        }
        D[e] = v;
    }
}

// anti-diagonal wavefront of the accumulated cost, one workgroup (cells of a diagonal are independent), then the
// traceback by one thread.  path_*: reversed order, *path_len entries.
__global__ __launch_bounds__(1024) void dtw_accumulate_kernel(double *__restrict__ D, int R, int C,
                                                              int32_t *__restrict__ path_i, int32_t *__restrict__ path_j,
                                                              int32_t *__restrict__ path_len, double *__restrict__ min_dist) {
    const int tid = threadIdx.x;
    const int64_t W = C + 1;
    for (int d = 0; d < R + C - 1; ++d) {
        const int ilo = d - (C - 1) > 0 ? d - (C - 1) : 0;
        const int ihi = d < R - 1 ? d : R - 1;
        for (int i = ilo + tid; i <= ihi; i += 1024) {
            const int j = d - i;
            const double m0 = D[(int64_t)i * W + j], m1 = D[(int64_t)i * W + j + 1], m2 = D[(int64_t)(i + 1) * W + j];
            const double m = fmin(m0, fmin(m1, m2));
            D[(int64_t)(i + 1) * W + j + 1] += m;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) {
        *min_dist = D[(int64_t)R * W + C] / (double)(R + C);
        int i = R - 1, j = C - 1, n = 0;
        path_i[n] = i; path_j[n] = j; ++n;
        while (i > 0 || j > 0) {
            const double v0 = D[(int64_t)i * W + j], v1 = D[(int64_t)i * W + j + 1], v2 = D[(int64_t)(i + 1) * W + j];
            int tb = 0;
            double best = v0;
            if (v1 < best) { best = v1; tb = 1; }
            if (v2 < best) { best = v2; tb = 2; }
            if (tb == 0) { --i; --j; } else if (tb == 1) --i; else --j;
            path_i[n] = i; path_j[n] = j; ++n;
        }
        *path_len = n;
    }
}

hipError_t launch_dtw(hipStream_t s, const float *a, const double *na, int64_t R, int64_t lda, const float *b,
                      const double *nb, int64_t C, int64_t ldb, int dim, double *D, double *dist_out, int32_t *path_i,
                      int32_t *path_j, int32_t *path_len, double *min_dist) {
    const int64_t total = (R + 1) * (C + 1);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    dtw_dist_kernel<<<blocks, 256, 0, s>>>(a, na, R, lda, b, nb, C, ldb, dim, D, dist_out);
    dtw_accumulate_kernel<<<1, 1024, 0, s>>>(D, (int)R, (int)C, path_i, path_j, path_len, min_dist);
    return hipGetLastError();
}

}  // namespace asr
