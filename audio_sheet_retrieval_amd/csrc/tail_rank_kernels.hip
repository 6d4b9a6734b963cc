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
#include <type_traits>

namespace asr {

// ---------------------------------------------------------------------------
// tail: one 256-thread block per sample.  thread = (pixel group, four channels) for
// the channel means; partial sums meet in LDS; one half-wave finishes the sample.
// ---------------------------------------------------------------------------
template <int C8>
__global__ __launch_bounds__(256) void tail_kernel(const float *__restrict__ a8, int N, int npix,
                                                   const float *__restrict__ w9, const float *__restrict__ bnp9,
                                                   const float *__restrict__ cca_mean,
                                                   const float *__restrict__ cca_proj,
                                                   float *__restrict__ features, float *__restrict__ latent) {
    // Block 9 has the identity nonlinearity: 1x1 conv, BatchNorm (deterministic) and the mean over H x W are all affine,
    // so GlobalPool(BN(conv(a8))) = BN(conv(mean over pixels of a8)) - the channel means first (one pass over the
    // sample's 23 KB), then a C8 x 32 product, instead of a 1x1 convolution at every pixel (32 x the multiply-adds; the
    // per-pixel form took 43 us per 1000 sheets).  float32 summation order differs from the per-pixel form at 1e-7.
    constexpr int C4 = C8 / 4, G = 256 / C4;                  // channel groups of four, pixel groups
    __shared__ float4 part[G][C4];
    __shared__ float cmean[C8];
    const int n = blockIdx.x;
    if (n >= N) return;
    const int tid = threadIdx.x;
    const int c4 = tid % C4, pg = tid / C4;
    const float4 *img = reinterpret_cast<const float4 *>(a8 + (size_t)n * npix * C8);
    if (pg < G) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int p = pg; p < npix; p += G) {
            const float4 v = img[(size_t)p * C4 + c4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        part[pg][c4] = acc;
    }
    __syncthreads();
    if (tid < C8) {
        const int cc4 = tid >> 2, k = tid & 3;
        float t = 0.0f;
        for (int q = 0; q < G; ++q) {
            const float4 v = part[q][cc4];
            t += k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
        }
        cmean[tid] = t / (float)npix;
    }
    __syncthreads();
    if (tid >= 32) return;                            // one half-wave finishes the sample
    const int o = tid;
    float z = 0.0f;
#pragma unroll 8
    for (int c = 0; c < C8; ++c) z = fmaf(cmean[c], w9[o * C8 + c], z);
    const float hfeat = (z - bnp9[o]) * bnp9[32 + o] + bnp9[64 + o];      // BatchNormLayer, identity nonlinearity
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

// Packed 32-d rows (the embeddings): a workgroup's 256 rows are 32 contiguous KiB - fetched with coalesced 16-byte
// loads into LDS (row stride 33 floats: the per-row walks below then hit 32 different banks), after which every thread
// sums ITS row in dot2acc's order (the float64 norm is part of the bit-exact distance).  One thread per row reading its
// 128 bytes straight from global memory ran at 0.5 TB/s: 0.54 ms for the 2 M-code pool of configs[4], more than the
// top-k filter itself.
__global__ __launch_bounds__(256) void row_norms32_kernel(const float *__restrict__ x, int64_t n, double *__restrict__ norms) {
    __shared__ float rows[256 * 33];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int64_t total4 = (n - r0 < 256 ? n - r0 : 256) * 8;      // float4 chunks of this workgroup's rows
    const float4 *src = reinterpret_cast<const float4 *>(x + r0 * 32);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = it * 256 + tid;
        if (c < total4) {
            const float4 v = src[c];
            float *d = rows + (c >> 3) * 33 + (c & 7) * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    if (r0 + tid < n) {
        const float *r = rows + tid * 33;
        norms[r0 + tid] = __dsqrt_rn(dot2acc(r, r, 32));
    }
}

hipError_t launch_row_norms(hipStream_t s, const float *x, int64_t n, int64_t ld, int dim, double *norms) {
    if (n == 0) return hipSuccess;
    if (dim == 32 && ld == 32 && (reinterpret_cast<uintptr_t>(x) & 15) == 0)
        row_norms32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, n, norms);
    else
        row_norms_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, n, ld, dim, norms);
    return hipGetLastError();
}

// Resident code data base (asr_db_create): everything about the pool that does not depend on the queries, computed once -
// float64 row norms (part of the bit-exact distance), their fp32 reciprocals (the filters' scale when they read the
// raw rows) and a UNIT-LENGTH fp32 copy of the rows: against it the MFMA accumulator IS the cosine, so the filters
// compare it with per-query constants and the per-distance scaling (one multiply + one fused multiply-add) disappears.
// |<q^, x^> - cos| <= 2.2e-6: 1.2e-7 per operand from the two roundings of x * (float)(1/|x|), 32 * 2^-24 from the
// accumulation, inside the 3e-6 the filters' proofs assume.
__global__ __launch_bounds__(256) void db_prepare32_kernel(const float *__restrict__ x, int64_t n, double *__restrict__ norms,
                                                           float *__restrict__ rn, float *__restrict__ unit) {
    __shared__ float rows[256 * 33];
    __shared__ float rnl[256];
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int64_t total4 = (n - r0 < 256 ? n - r0 : 256) * 8;
    const float4 *src = reinterpret_cast<const float4 *>(x + r0 * 32);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = it * 256 + tid;
        if (c < total4) {
            const float4 v = src[c];
            float *d = rows + (c >> 3) * 33 + (c & 7) * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
    if (r0 + tid < n) {
        const float *r = rows + tid * 33;
        const double nr = __dsqrt_rn(dot2acc(r, r, 32));
        norms[r0 + tid] = nr;
        const float rf = (float)(1.0 / nr);
        rn[r0 + tid] = rf;
        rnl[tid] = rf;
    }
    __syncthreads();
    float4 *dst = reinterpret_cast<float4 *>(unit + r0 * 32);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = it * 256 + tid;
        if (c < total4) {
            const float *d = rows + (c >> 3) * 33 + (c & 7) * 4;
            const float rf = rnl[c >> 3];
            dst[c] = make_float4(d[0] * rf, d[1] * rf, d[2] * rf, d[3] * rf);
        }
    }
}

hipError_t launch_db_prepare(hipStream_t s, const float *x, int64_t n, double *norms, float *rn, float *unit) {
    if (n == 0) return hipSuccess;
    if (reinterpret_cast<uintptr_t>(x) & 15) return hipErrorInvalidValue;
    const int64_t n_pad = (n + 3) & ~(int64_t)3;
    if (n_pad > n) {
        hipError_t e = hipMemsetAsync(rn + n, 0, (size_t)(n_pad - n) * sizeof(float), s);
        if (e != hipSuccess) return e;
    }
    db_prepare32_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(x, n, norms, rn, unit);
    return hipGetLastError();
}

// ---- ranking of large candidate sets: counts on the fp32 MFMA, exact arithmetic only inside the +-2e-5 band -------
// rank_i = 1 + #{j: d_ij < d*_i} + #{j < j*_i: d_ij == d*_i} needs, for almost every pair, only the SIDE of d*_i the
// distance falls on.  d~ (fp32, |d~ - d| <= 3e-6, see topk_filter_kernel) decides it whenever |d~ - d*| > 2e-5; the
// handful of pairs inside the band are evaluated in float64 exactly as rank_kernel does.  Integer counters are
// accumulated with atomics (order-independent), so the ranks, d* and tie counts are bit-identical to rank_kernel's.
constexpr float RF_BAND = 2e-5f;
constexpr float RF_BAND_BF2 = 5.5e-5f;         // ... and with two bf16 planes: |d~ - d| <= 4.9e-5 (TF_EPS_BF2's derivation)

// d*, j* of every query: first minimum over its kk correct candidates (utils/train_dcca_pool.py:52-55).  One WAVE per
// query, lanes strided over the candidates, (distance, index) minimum by shuffles: one thread per query walking 512
// candidates (4096 queries against a 2^21-code pool) took 0.8 ms - 6 % of the fused retrieval pass.
__global__ __launch_bounds__(256) void rank_dstar_kernel(const float *__restrict__ lv1, const double *__restrict__ norm1,
                                                         int64_t n1, const float *__restrict__ lv2,
                                                         const double *__restrict__ norm2, int64_t n2,
                                                         int64_t query_offset, int64_t kk, int64_t hh,
                                                         double *__restrict__ dstar, int64_t *__restrict__ jstar,
                                                         int64_t item_offset, int64_t n2_global) {
    // item_offset / n2_global: lv2 is rows [item_offset, item_offset + n2) of a pool of n2_global rows; the caller
    // passes only queries whose correct candidates lie inside it.  j* is a global index.
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n1) return;                                           // wave-uniform
    const int64_t glo = ((i + query_offset) / hh) * kk;
    const int64_t ghi = (glo + kk < n2_global) ? glo + kk : n2_global;
    const int64_t lo = glo - item_offset, hi = ghi - item_offset;
    double best = 1e300;
    int64_t bj = 0x7fffffffffffffffLL;
    for (int64_t j = lo + lane; j < hi; j += 64) {
        const double d = cos_dist(dot2acc(lv1 + i * 32, lv2 + j * 32, 32), norm1[i], norm2[j]);
        if (d < best) { best = d; bj = j; }                        // j ascending per lane: keeps the first
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double d2 = __shfl_xor(best, o);
        const int64_t j2 = __shfl_xor(bj, o);
        if (d2 < best || (d2 == best && j2 < bj)) { best = d2; bj = j2; }
    }
    if (lane == 0) { dstar[i] = best; jstar[i] = bj == 0x7fffffffffffffffLL ? bj : bj + item_offset; }
}

typedef float floatx4_r __attribute__((ext_vector_type(4)));

// Unit-length rows on the bf16 MFMA (round 4b).  The fp32 MFMA runs at the fp32 VECTOR rate (64 FLOP per clock and
// SIMD, 1/16 of the bf16 rate) and its cycles add to the VALU's - the few-queries filter spent 2.6x the time the pool
// takes to stream from HBM on eight v_mfma_f32_16x16x4_f32 per (tile, query group).  A float32 splits EXACTLY into three
// bf16 planes, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2: 8 + 8 + <= 8 significant bits,
// round-to-nearest, both differences exact), and
//     <x, y> = x3.y1 + x1.y3 + x2.y2 + x2.y1 + x1.y2 + x1.y1  +  (x2.y3 + x3.y2 + x3.y3 <= 2^-26 sum |x_i y_i|)
// is six v_mfma_f32_16x16x32_bf16 (K = 32: the whole row, ~16 cycles each, products exact, fp32 accumulate, smallest
// terms first) instead of eight fp32 ones of 32 cycles: 96 cycles instead of 256, and the VALU keeps issuing under them.
// The rows are split on the fly (44 vector instructions per tile, shared by the workgroup's query groups), the queries
// once per workgroup.  Accuracy, tools/mfma_bf16x3_probe.hip on the MI355X (16.7 M pairs of unit vectors: one sign - sum
// |x_i y_i| = 1 -, alternating signs, near neighbours, one-hot, random): max |dot - float64| 2.1e-7, the fp32 MFMA's
// on the same pairs 3.2e-7 - the 3e-6 that the filters' proofs assume holds with the same margin.  NaN rows stay NaN
// (a quiet NaN keeps its top mantissa bit in bf16).  ASR_TF_BF3=0 at compile time: the fp32 MFMAs.
#ifndef ASR_TF_BF3
#define ASR_TF_BF3 1
#endif
// ASR_TF_BF2: two planes instead of three where a wider error bound is affordable (see TF_EPS_BF2 at the top-k filter):
// 1 = the top-k filter, 2 = also the counting ranking (fused and stand-alone, band RF_BAND_BF2), 0 = three planes
#ifndef ASR_TF_BF2
#define ASR_TF_BF2 2
#endif
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float floatx2_t __attribute__((ext_vector_type(2)));
typedef unsigned uintx4_t __attribute__((ext_vector_type(4)));
struct Bf3 { bf16x8_t p1, p2, p3; };
__device__ __forceinline__ Bf3 split_bf3(const float (&x)[8]) {
    uintx4_t w1, w2, w3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const floatx2_t v = {x[2 * i], x[2 * i + 1]};
        const unsigned u1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
        const floatx2_t r = {v[0] - __uint_as_float(u1 << 16), v[1] - __uint_as_float(u1 & 0xFFFF0000u)};
        const unsigned u2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
        const floatx2_t t = {r[0] - __uint_as_float(u2 << 16), r[1] - __uint_as_float(u2 & 0xFFFF0000u)};
        const unsigned u3 = __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2_t));
        w1[i] = u1; w2[i] = u2; w3[i] = u3;
    }
    Bf3 o;
    o.p1 = __builtin_bit_cast(bf16x8_t, w1);
    o.p2 = __builtin_bit_cast(bf16x8_t, w2);
    o.p3 = __builtin_bit_cast(bf16x8_t, w3);
    return o;
}


// QG query groups of 16 per workgroup: the A fragment of an item tile (its 2 KB come from L2 / HBM) is multiplied with
// QG B fragments held in registers.  With one group per workgroup 4096 queries against a 2^21-code pool re-streamed the
// 256 MB pool 256 times - 65 GB through the L2s in 16 ms, 4.1 TB/s: the kernel was L2-bound at 20 % of the MFMA peak.
__global__ __launch_bounds__(256) void rnorm_f32_kernel(const double *__restrict__ norms, int64_t n, float *__restrict__ rn);

// rn2: (float)(1.0 / norm2[j]) of every candidate, computed ONCE (rnorm_f32_kernel) - the kernel used to evaluate that
// float64 reciprocal per item, lane and query group inside its tile loop (four per tile: more vector cycles than the
// tile's eight MFMAs)
template <int QG>
__global__ __launch_bounds__(256) void rank_count_kernel(
    const float *__restrict__ lv1, const double *__restrict__ norm1, int64_t n1, const float *__restrict__ lv2,
    const double *__restrict__ norm2, const float *__restrict__ rn2, int64_t n2, const double *__restrict__ dstar,
    const int64_t *__restrict__ jstar, int n_slices, int32_t *__restrict__ counts /*[n1][3]: less, eq, eq before j* */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nn = lane & 15;
    const int grp = blockIdx.x / n_slices, slice = blockIdx.x - grp * n_slices;
    const int64_t q0 = (int64_t)grp * 16 * QG;
    const int64_t tiles = (n2 + 15) / 16;
    const int64_t t_lo = tiles * slice / n_slices, t_hi = tiles * (slice + 1) / n_slices;
    float bq[QG][8], rq[QG], lo_t[QG], hi_t[QG];
    double nq[QG], ds[QG];
    int64_t js[QG], qi[QG];
    int less[QG], eq[QG], eqb[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        qi[u] = q0 + 16 * u + nn < n1 ? q0 + 16 * u + nn : n1 - 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) bq[u][j] = lv1[qi[u] * 32 + 8 * g + j];
        nq[u] = norm1[qi[u]]; ds[u] = dstar[qi[u]]; js[u] = jstar[qi[u]];
        rq[u] = (float)(1.0 / nq[u]);
        constexpr float band = (ASR_TF_BF3 && ASR_TF_BF2 > 1) ? RF_BAND_BF2 : RF_BAND;
        lo_t[u] = (float)ds[u] - band; hi_t[u] = (float)ds[u] + band;
        less[u] = 0; eq[u] = 0; eqb[u] = 0;
    }
    // the products on the bf16 MFMA as an exact three-plane split (split_bf3 above; raw rows: the error is relative to
    // |q| |x| like the fp32 MFMA's and is scaled by the same reciprocal norms)
    Bf3 qb[ASR_TF_BF3 ? QG : 1];
    if (ASR_TF_BF3) {
#pragma unroll
        for (int u = 0; u < QG; ++u) qb[u] = split_bf3(bq[u]);
    }
    for (int64_t tb = t_lo + wave; tb < t_hi; tb += 8) {          // two tiles per wave and iteration
        float4 a0[2], a1[2];
        float rn[2][4];
        int lim[2];                                                // valid items among this lane's four of the tile
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t tile = tb + 4 * r;
            const int64_t item = tile * 16 + nn;
            const int64_t left = n2 - (tile * 16 + 4 * g);
            lim[r] = tile < t_hi ? (left >= 4 ? 4 : (left > 0 ? (int)left : 0)) : 0;
            a0[r] = make_float4(0.f, 0.f, 0.f, 0.f); a1[r] = a0[r];
            if (tile < t_hi && item < n2) {
                const float4 *p = reinterpret_cast<const float4 *>(lv2 + item * 32 + 8 * g);
                a0[r] = p[0]; a1[r] = p[1];
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int64_t it = tile * 16 + 4 * g + rr;
                rn[r][rr] = (tile < t_hi && it < n2) ? rn2[it] : 0.0f;
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t tile = tb + 4 * r;
            if (tile >= t_hi) continue;
            const float af[8] = {a0[r].x, a0[r].y, a0[r].z, a0[r].w, a1[r].x, a1[r].y, a1[r].z, a1[r].w};
            Bf3 ab;
            if (ASR_TF_BF3) ab = split_bf3(af);
#pragma unroll
            for (int u = 0; u < QG; ++u) {
                floatx4_r acc = {0.f, 0.f, 0.f, 0.f};
                if (ASR_TF_BF3 && ASR_TF_BF2 > 1) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p1, acc, 0, 0, 0);
                } else if (ASR_TF_BF3) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p3, qb[u].p1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p3, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p1, acc, 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bq[u][j], acc, 0, 0, 0);
                }
                // per distance: one multiply, one fused multiply-add, two compares; the items past the end of the pool
                // (last tile only) are cut off by `lim`, computed once per tile, and the exact path is entered only when
                // some lane of the wave is inside the band (the 64-bit index tests and three-way branches per distance
                // cost the SIMD as much as the tile's MFMAs)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const float d = fmaf(-(acc[rr] * rn[r][rr]), rq[u], 1.0f);
                    const bool ok = rr < lim[r];
                    less[u] += (ok && d < lo_t[u]) ? 1 : 0;
                    const bool band = ok && d >= lo_t[u] && d <= hi_t[u];     // (NaN: never counted, like d < d*)
                    if (__ballot(band) == 0) continue;                         // wave-uniform
                    if (band) {
                        const int64_t it = tile * 16 + 4 * g + rr;
                        const double de = cos_dist(dot2acc(lv1 + qi[u] * 32, lv2 + it * 32, 32), nq[u], norm2[it]);
                        less[u] += de < ds[u];
                        const int e = de == ds[u];
                        eq[u] += e;
                        eqb[u] += e && (it < js[u]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QG; ++u)
        if (q0 + 16 * u + nn < n1) {
            if (less[u]) atomicAdd(&counts[qi[u] * 3], less[u]);
            if (eq[u]) atomicAdd(&counts[qi[u] * 3 + 1], eq[u]);
            if (eqb[u]) atomicAdd(&counts[qi[u] * 3 + 2], eqb[u]);
        }
}

__global__ __launch_bounds__(256) void rank_finish_kernel(const int32_t *__restrict__ counts, const double *__restrict__ dstar,
                                                          int64_t n1, int32_t *__restrict__ ranks,
                                                          double *__restrict__ dstar_out, int32_t *__restrict__ ties_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    if (ranks) ranks[i] = 1 + counts[i * 3] + counts[i * 3 + 2];
    if (dstar_out) dstar_out[i] = dstar[i];
    if (ties_out) ties_out[i] = counts[i * 3 + 1] - 1;
}

size_t rank_workspace_bytes(int64_t n1, int64_t n2) {
    return (size_t)n1 * (sizeof(double) + sizeof(int64_t)) + (size_t)((n1 + 3) & ~(int64_t)3) * 3 * sizeof(int32_t) +
           (size_t)((n2 + 3) & ~(int64_t)3) * sizeof(float);
}

hipError_t launch_rank(hipStream_t s, const float *lv1, const double *norm1, int64_t n1, int64_t ld1,
                       const float *lv2, const double *norm2, int64_t n2, int64_t ld2, int dim,
                       int64_t query_offset, int64_t k, int64_t h, int32_t *ranks, double *dstar, int32_t *ties,
                       void *workspace, const float *rn2_pre) {
    if (n1 == 0) return hipSuccess;
    if (dim > RANK_MAXD) return hipErrorInvalidValue;
    static const int use_filter = getenv("ASR_RANK_FILTER") ? atoi(getenv("ASR_RANK_FILTER")) : 1;
    // k = n2 / n1_global correct candidates per query (utils/train_dcca_pool.py:35): rank_dstar_kernel walks them in one
    // thread per query - 512 for 4096 queries against a 2^21-code pool (configs[4]), tens of microseconds
    if (!use_filter || !workspace || dim != 32 || ld1 != 32 || ld2 != 32 || n2 < 2048 || k > 8192) {
        rank_kernel<<<(unsigned)n1, RANK_THREADS, 0, s>>>(lv1, norm1, n1, ld1, lv2, norm2, n2, ld2, dim, query_offset, k,
                                                          h, ranks, dstar, ties);
        return hipGetLastError();
    }
    double *ds = (double *)workspace;
    int64_t *js = (int64_t *)(ds + n1);
    int32_t *counts = (int32_t *)(js + n1);
    const float *rn2 = rn2_pre;                                           // a resident data base brings them along
    if (!rn2) {
        float *w = (float *)(counts + 3 * ((n1 + 3) & ~(int64_t)3));     // (behind the counters, 16-byte aligned)
        rnorm_f32_kernel<<<(unsigned)((n2 + 255) / 256), 256, 0, s>>>(norm2, n2, w);
        rn2 = w;
    }
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)n1 * 3 * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    rank_dstar_kernel<<<(unsigned)((n1 + 3) / 4), 256, 0, s>>>(lv1, norm1, n1, lv2, norm2, n2, query_offset, k, h, ds, js, 0, n2);
    // query groups per workgroup: as many as still leave >= 1024 workgroups with at least one slice each
    static const int qg_env = getenv("ASR_RANK_QG") ? atoi(getenv("ASR_RANK_QG")) : 0;
    const int64_t groups16 = (n1 + 15) / 16;
    // (measured, 4096 queries x 2^21 codes: one group 16.0 ms, two 12.8, four 20.4 - the fourfold epilogue spills)
    const int qg = (qg_env == 1 || qg_env == 2) ? qg_env : (groups16 >= 128 ? 2 : 1);
    const int64_t groups = (groups16 + qg - 1) / qg;
    int S = (int)std::max<int64_t>(1, std::min<int64_t>(64, (2048 + groups - 1) / groups));
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, n2 / 512));
    if (qg == 2)
        rank_count_kernel<2><<<(unsigned)(groups * S), 256, 0, s>>>(lv1, norm1, n1, lv2, norm2, rn2, n2, ds, js, S, counts);
    else
        rank_count_kernel<1><<<(unsigned)(groups * S), 256, 0, s>>>(lv1, norm1, n1, lv2, norm2, rn2, n2, ds, js, S, counts);
    rank_finish_kernel<<<(unsigned)((n1 + 255) / 256), 256, 0, s>>>(counts, ds, n1, ranks, dstar, ties);
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
constexpr int TOPK_TICKETS = 512;              // last-arriver counters per use (refine merge | threshold select): 2 x 512
constexpr int TSEL_CAP = 16384;                // sort-free refine: most exact keys per query between its two kernels
// ... what a call really reserves per query: 64 MB for all of them together, 4096 .. TSEL_CAP keys each (the survivors of a
// seeded filter are ~k n_db / sample rows: ~3000 +- 40 % for 2 M rows, ~760 for 250 k)
static inline int tsel_cap(int64_t n_q) {
    int64_t c = ((int64_t)64 << 20) / (16 * (n_q > 0 ? n_q : 1));
    c = c < 4096 ? 4096 : (c > TSEL_CAP ? TSEL_CAP : c);
    return (int)(c & ~(int64_t)255);
}
constexpr int TSEL_NQ_MAX = 1024;           // ... and queries (counts / flags live in the context's 4096-int state block)
constexpr int TSEL_THREADS = 1024;           // threads of topk_select_kernel

struct TopkKey {
    unsigned long long d;       // bits of the non-negative float64 distance (monotone as unsigned)
    long long j;                // global candidate index
};
__device__ __forceinline__ bool key_less(const TopkKey &a, const TopkKey &b) {
    return a.d < b.d || (a.d == b.d && a.j < b.j);
}

// The exact scan of one query over a (virtual) row space by the NT threads of a workgroup: every row's float64 key, the
// ones below the current k-th best collected in LDS (keys[TOPK_KMAX ..), <= CAP of them) and merged into the best list
// (keys[0, k), ascending) by a bitonic network whenever fewer than a step's worth of slots remain.  On entry keys[0,
// TOPK_SORT) hold the +inf key, *ncand = 0, *thr = +inf (and a barrier has passed); on return keys[0, k) are the k best.
// use_lists: the row space is lists [l0, ..) x slot_cap slots of the filter's candidate lists; otherwise rows 0 .. n_db-1
// (times row_stride).  Shared by topk_kernel and by topk_select_kernel's fallback.
template <int NT, int PT, int CAP>
__device__ __forceinline__ void topk_scan_rows(const float *__restrict__ db, const double *__restrict__ norm_db, int64_t ld_db,
                                               int64_t row_stride, const float *q, double nq, int dim, int64_t idx_offset, int k,
                                               int64_t n_db, bool use_lists, int64_t qi, int n_lists, int l0, int slot_cap,
                                               int list_cap, const int32_t *__restrict__ cand_idx,
                                               const int32_t *__restrict__ cand_cnt, TopkKey *keys, int *ncand, TopkKey *thr) {
    static_assert(CAP + TOPK_KMAX <= TOPK_SORT && NT * PT < CAP, "candidate buffer");
    const int tid = threadIdx.x;
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    const int64_t step = (int64_t)NT * PT;
    for (int64_t base = 0; base < n_db; base += step) {
        const TopkKey t = *thr;
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            int64_t j = base + (int64_t)u * NT + tid;
            bool have = j < n_db;
            if (have && use_lists) {
                const int l = l0 + (int)(j / slot_cap), e = (int)(j % slot_cap);
                have = e < cand_cnt[qi * n_lists + l];
                if (have) j = cand_idx[(qi * n_lists + l) * list_cap + e];
            }
            if (have) {
                const double d = cos_dist(dot2acc(q, db + j * row_stride * ld_db, dim), nq, norm_db[j * row_stride]);
                TopkKey kk;
                kk.d = (unsigned long long)__double_as_longlong(d + 0.0);     // +0.0: never the -0.0 pattern
                kk.j = j + idx_offset;
                if (key_less(kk, t)) {
                    const int pos = atomicAdd(ncand, 1);
                    keys[TOPK_KMAX + pos] = kk;        // pos < CAP: merged whenever fewer than `step` slots remain
                }
            }
        }
        __syncthreads();
        const bool last = base + step >= n_db;
        if (*ncand > CAP - (int)step || last) {
            // (round 5, measured and dropped: ordering <= 512 keys by counting ranks instead of the bitonic network, plus an
            // early first threshold - 64 queries x 2 M codes 0.151 -> 0.211 ms: 300-500 dependent 16-byte LDS reads per
            // key cost more than the 36-45 barrier rounds they replace)
            // bitonic sort of the key array (best list + candidates + padding): only the power of two that covers the
            // occupied slots - everything behind them holds the +inf key already
            int sort_n = 256;
            while (sort_n < TOPK_KMAX + *ncand) sort_n <<= 1;
            for (int size = 2; size <= sort_n; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int e = tid; e < sort_n / 2; e += NT) {
                        const int lo = 2 * e - (e & (stride - 1));
                        const int hi = lo + stride;
                        const bool up = (lo & size) == 0;
                        const TopkKey a = keys[lo], b = keys[hi];
                        if (key_less(b, a) == up) { keys[lo] = b; keys[hi] = a; }
                    }
                    __syncthreads();
                }
            // keep the k best, reset the rest
            for (int e = tid; e < sort_n; e += NT)
                if (e >= k) keys[e] = inf;
            if (tid == 0) { *ncand = 0; *thr = keys[k - 1]; }
            __syncthreads();
        }
    }
}

// cand_idx / cand_cnt (may be null): per query `n_lists` lists of `list_cap` data-base indices produced by
// topk_filter_kernel (a superset of the query's top-k); cand_cnt < 0 marks a list that overflowed - then, and when no
// lists are given, the whole data base is scanned.
// grid = (queries, chunks).  One chunk: the block writes the query's result.  Several (few queries against a large pool:
// 64 queries used to be 64 workgroups on 256 CUs walking 512 lists each): block (q, c) orders the survivors of ITS lists
// and writes k (index, distance) keys to part_idx / part_dist [q][c][k]; topk_merge_kernel picks the k smallest keys of
// the query's chunks - keys are exact (float64 distance, index), so the merge of partial top-k lists is the top-k.  A
// query with an overflowed list is scanned exactly by its chunk 0 alone.
__device__ __forceinline__ void topk_merge_body(const int32_t *__restrict__ part_idx, const double *__restrict__ part_dist,
                                                int n_chunks, int k, int64_t n_db_full, int32_t *__restrict__ idx_out,
                                                double *__restrict__ dist_out, int64_t q_stride, int64_t c_stride,
                                                int64_t qi_in, int64_t qi_out, TopkKey *keys);

// tickets (several chunks, may be null): one zero-initialised counter per query - the LAST chunk of a query to finish
// merges the query's partial lists itself (topk_merge_body) and resets the counter: topk_merge_kernel's launch and the gap
// in front of it disappear from the few-queries call (13 us + the gap of a 0.166 ms call).
__global__ __launch_bounds__(TOPK_THREADS) void topk_kernel(
    const float *__restrict__ db, const double *__restrict__ norm_db, int64_t n_db_full, int64_t ld_db,
    const float *__restrict__ qs, const double *__restrict__ norm_q, int64_t ld_q, int dim, int k,
    int64_t idx_offset, int32_t *__restrict__ idx_out, double *__restrict__ dist_out,
    const int32_t *__restrict__ cand_idx, const int32_t *__restrict__ cand_cnt, int n_lists, int list_cap,
    int32_t *__restrict__ part_idx, double *__restrict__ part_dist, int64_t row_stride, unsigned *__restrict__ tickets) {
    // row_stride > 1: the "data base" is a strided sample of the rows (virtual row j = row j * row_stride)
    __shared__ float q[RANK_MAXD];
    __shared__ TopkKey keys[TOPK_SORT];        // [0, KMAX): best list, [KMAX, KMAX+CAP): candidates
    __shared__ int ncand;
    __shared__ TopkKey thr;                    // current k-th best key
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int chunk = blockIdx.y, n_chunks = gridDim.y;
    for (int c = tid; c < dim; c += TOPK_THREADS) q[c] = qs[qi * ld_q + c];
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    for (int e = tid; e < TOPK_SORT; e += TOPK_THREADS) keys[e] = inf;
    if (tid == 0) { ncand = 0; thr = inf; }
    __syncthreads();
    const double nq = norm_q[qi];
    // candidate mode: the virtual index space is this chunk's lists x list_cap slots, slot (l, e) valid when e < cnt[l]
    bool use_lists = cand_idx != nullptr;
    if (use_lists) {
        int over = 0;
        for (int l = tid; l < n_lists; l += TOPK_THREADS) over |= cand_cnt[qi * n_lists + l] < 0;
        if (__syncthreads_or(over)) use_lists = false;                    // block-uniform
    }
    const int per = (n_lists + n_chunks - 1) / n_chunks;
    const int l0 = use_lists ? chunk * per : 0;
    const int l1 = use_lists ? (l0 + per < n_lists ? l0 + per : n_lists) : 0;
    // the virtual slot space is lists x (longest list of this chunk), not lists x list_cap: with seeded thresholds a list
    // holds a handful of its 192 slots, and hundreds of lists per query would otherwise be walked slot by empty slot
    __shared__ int s_mc;
    if (tid == 0) s_mc = 0;
    __syncthreads();
    if (use_lists) {
        int m = 0;
        for (int l = l0 + tid; l < l1; l += TOPK_THREADS) m = max(m, cand_cnt[qi * n_lists + l]);
        if (m > 0) atomicMax(&s_mc, m);
    }
    __syncthreads();
    const int slot_cap = use_lists ? s_mc : 0;
    const int64_t n_db = use_lists ? (int64_t)(l1 > l0 ? l1 - l0 : 0) * slot_cap : (chunk == 0 ? n_db_full : 0);

    topk_scan_rows<TOPK_THREADS, TOPK_PER_THREAD, TOPK_CAP>(db, norm_db, ld_db, row_stride, q, nq, dim, idx_offset, k, n_db, use_lists, qi,
                                                            n_lists, l0, slot_cap, list_cap, cand_idx, cand_cnt, keys, &ncand, &thr);
    if (n_chunks > 1) {                        // partial list of this chunk (unfilled slots: the +inf key)
        for (int e = tid; e < k; e += TOPK_THREADS) {
            const TopkKey kk = keys[e];
            const bool real = kk.j != inf.j;
            part_idx[(qi * n_chunks + chunk) * k + e] = real ? (int32_t)kk.j : -1;
            part_dist[(qi * n_chunks + chunk) * k + e] = __longlong_as_double((long long)kk.d);
        }
        if (!tickets) return;                  // (topk_merge_kernel follows)
        __shared__ int last_chunk;
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const unsigned t = atomicAdd(&tickets[qi], 1u);
            last_chunk = (t == (unsigned)n_chunks - 1u) ? 1 : 0;
            if (last_chunk) tickets[qi] = 0u;
        }
        __syncthreads();
        if (!last_chunk) return;
        __threadfence();
        topk_merge_body(part_idx, part_dist, n_chunks, k, n_db_full, idx_out, dist_out, (int64_t)n_chunks * k, k, qi, qi, keys);
        return;
    }
    for (int e = tid; e < k; e += TOPK_THREADS) {
        const TopkKey kk = keys[e];
        const bool valid = e < n_db_full;
        idx_out[qi * k + e] = valid ? (int32_t)kk.j : -1;
        dist_out[qi * k + e] = valid ? __longlong_as_double((long long)kk.d) : __longlong_as_double(0x7ff0000000000000LL);
    }
}

// the k smallest (distance, index) keys among a query's n_chunks partial lists (n_chunks * k <= TOPK_SORT)
// q_stride / c_stride: elements between two queries / two chunks of a query in part_* ([q][chunk][k]: n_chunks k and k;
// lists gathered from several ranks, [rank][q][k]: k and n_q_total k); qi_in / qi_out: the query's row in part_* / in the
// outputs.  keys: TOPK_SORT entries of LDS; all threads of the workgroup call it.
__device__ __forceinline__ void topk_merge_body(const int32_t *__restrict__ part_idx, const double *__restrict__ part_dist,
                                                int n_chunks, int k, int64_t n_db_full, int32_t *__restrict__ idx_out,
                                                double *__restrict__ dist_out, int64_t q_stride, int64_t c_stride,
                                                int64_t qi_in, int64_t qi_out, TopkKey *keys) {
    const int tid = threadIdx.x;
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    const int n = n_chunks * k;
    int sort_n = 64;                           // the next power of two: 16 chunks x 25 keys sort as 512, not 2048
    while (sort_n < n) sort_n <<= 1;
    for (int e = tid; e < sort_n; e += TOPK_THREADS) {
        TopkKey kk = inf;
        if (e < n) {
            const int64_t src = qi_in * q_stride + (int64_t)(e / k) * c_stride + e % k;
            const int32_t j = part_idx[src];
            if (j >= 0) { kk.d = (unsigned long long)__double_as_longlong(part_dist[src]); kk.j = j; }
        }
        keys[e] = kk;
    }
    __syncthreads();
    for (int size = 2; size <= sort_n; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = tid; e < sort_n / 2; e += TOPK_THREADS) {
                const int lo = 2 * e - (e & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const TopkKey a = keys[lo], b = keys[hi];
                if (key_less(b, a) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    for (int e = tid; e < k; e += TOPK_THREADS) {
        const TopkKey kk = e < sort_n ? keys[e] : inf;
        const bool valid = e < n_db_full && kk.j != inf.j;
        idx_out[qi_out * k + e] = valid ? (int32_t)kk.j : -1;
        dist_out[qi_out * k + e] = valid ? __longlong_as_double((long long)kk.d) : __longlong_as_double(0x7ff0000000000000LL);
    }
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_merge_kernel(const int32_t *__restrict__ part_idx,
                                                                  const double *__restrict__ part_dist, int n_chunks,
                                                                  int k, int64_t n_db_full, int32_t *__restrict__ idx_out,
                                                                  double *__restrict__ dist_out, int64_t q_stride,
                                                                  int64_t c_stride, int64_t q_in0) {
    __shared__ TopkKey keys[TOPK_SORT];
    topk_merge_body(part_idx, part_dist, n_chunks, k, n_db_full, idx_out, dist_out, q_stride, c_stride,
                    q_in0 + (int64_t)blockIdx.x, (int64_t)blockIdx.x, keys);
}

// ---- filter stage on the fp32 MFMA -------------------------------------------------------------------------------
// A workgroup takes 16 queries and one slice of the data base: d~ = 1 - <q, x> / (|q| |x|) for 16 x 16 (item, query)
// pairs per eight v_mfma_f32_16x16x4_f32, instead of 64 float64 FLOP per pair on the VALU with the data base
// re-read for every query.  |d~ - d| <= 3e-6 (fp32 dot of 32 terms + two roundings), EPS = 1e-5 is used:
//   * the k-th smallest d~ seen so far, t~_k, never undercuts d_k - EPS (d_k: the true k-th distance), so
//   * every true top-k item satisfies d~ <= d_k + EPS <= t~_k + 2 EPS  -> keeping {d~ <= t~_k + 2 EPS} keeps a
//     SUPERSET of the top-k (ties included), per slice as well as globally.
// The survivors (a few more than k) go to topk_kernel, which computes their exact float64 distances and orders them
// exactly as before.
// Buffer discipline: entries are appended to a per-query LDS buffer (TF_CAP entries: 256 for k <= 32, 512 for
// k <= 128) with an LDS atomic; after a round every buffer above TF_CAP / 2 is compacted to {d~ <= d~_k + 2 EPS},
// which also tightens its threshold.  No candidate is ever lost silently: an append that finds the buffer full,
// survivors that alone exceed TF_CAP / 2 (masses of near-ties) or more than TF_OUT survivors at the end mark the
// query, and a marked query falls back to the exact scan.
constexpr int TF_THREADS = 256;
constexpr int TF_OUT = 192;                  // survivors handed over per (query, slice); >= TOPK_KMAX
constexpr float TF_EPS = 1e-5f;
// Two bf16 planes instead of three (round 5, top-k without the fused ranking): x = x1 + x2 + x3 with |x2| <= 2^-8 |x|,
// |x3| <= 2^-16 |x| (round to nearest), and <x, y> ~ x2.y1 + x1.y2 + x1.y1 drops x2.y2 + x1.y3 + x3.y1 + (2^-24 terms)
// <= 3.02 2^-16 sum |x_i y_i| <= 4.6e-5 for unit-length rows.  With the 3e-6 of the fp32 accumulation |d~ - d| <= 4.9e-5:
// the same proof with EPS = 5.5e-5 (the threshold's own rounding to float, 1.2e-7, included).  What the wider margin costs is the density of distances at the threshold - the
// k / sample quantile, three sigma out on random codes: 0.2 % more survivors per 1e-4 - and what it buys is half the
// MFMAs (three per tile and query group instead of six) and a third of the split's vector instructions in a kernel
// that is bound by instruction issue, not by memory (two workgroups on a CU each run at half the speed of one alone).
constexpr float TF_EPS_BF2 = 5.5e-5f;
static_assert(RF_BAND_BF2 >= TF_EPS_BF2, "the ranking band covers the two-plane error bound");

__global__ __launch_bounds__(256) void rnorm_f32_kernel(const double *__restrict__ norms, int64_t n, float *__restrict__ rn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rn[i] = (float)(1.0 / norms[i]);
}

typedef float floatx4_t __attribute__((ext_vector_type(4)));

// Which 8 of a row's 32 dimensions lane (item nn, k group g) of an A / B fragment holds.  The contraction index may be
// permuted freely as long as both operands use the same map.  ASR_TF_HALFROW=1 (round 6): dims 4g..4g+3 and 16+4g..16+4g+3 -
// the first float4 load of a wave then covers the FIRST 64 bytes of each of its 16 rows and the second load the other
// 64, i.e. 16 whole 64-byte requests per instruction; with dims 8g..8g+7 (=0, rounds 3-5) each instruction took the
// even / odd 16-byte pieces of all 32 half-lines - twice the requests for the same bytes.
#ifndef ASR_TF_HALFROW
#define ASR_TF_HALFROW 1
#endif
__device__ __forceinline__ int tf_dim(int g, int j) { return ASR_TF_HALFROW ? (j < 4 ? 4 * g + j : 12 + 4 * g + j) : 8 * g + j; }
constexpr int TF_A0 = ASR_TF_HALFROW ? 4 : 8;      // float offset of a lane's first float4 = TF_A0 * g
constexpr int TF_A1 = ASR_TF_HALFROW ? 4 : 1;      // its second float4, in float4 units from the first

#if defined(ASR_TF_ABL) && (ASR_TF_ABL & 4)          // trace build (tools/ab_topk_abl.sh 4): per-workgroup time stamps
__device__ unsigned long long g_tf_trace[8192 * 8];
#define TF_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_tf_trace[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define TF_NOTE(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_tf_trace[blockIdx.x * 8 + (i)] = (unsigned long long)(v); } while (0)
#else
#define TF_STAMP(i) do { } while (0)
#define TF_NOTE(i, v) do { } while (0)
#endif

// RANK: the counting ranking of rank_count_kernel rides the same item tiles (asr_topk_rank_db_dev: the reference computes
// ONE distance row per query and uses it for the top-k and for the rank, audio_sheet_server.py:534-537,
// utils/train_dcca_pool.py:40-74).  What the rare exact evaluations inside the +-RF_BAND band need:
struct RankFuse {
    const float *db_raw;           // the original rows (the filter itself reads the unit-length copy)
    const double *norm_db, *norm_q;
    const double *dstar;           // per query: distance to its first correct candidate (rank_dstar_kernel) ...
    const int64_t *jstar;          // ... and that candidate's (global) index
    int32_t *counts;               // [n_q][3]: less, equal, equal before j*
    int64_t item_offset;           // global index of the data base's row 0 (a shard of a larger pool; j* is global)
};

// QG query groups of 16 per workgroup share every loaded item tile (see rank_count_kernel): NQ = 16 QG queries, each
// with its own candidate buffer.  QG = 4 serves <= 64 queries from ONE workgroup per slice: the pool is streamed once.
// NORM: `db` holds unit-length rows (db_prepare32_kernel) and the queries are normalised as they are loaded, so the
// accumulator s = <q^, x^> is the cosine itself: d~ = 1 - s, and "d~ <= t" is "s >= 1 - t" - one compare per distance
// against a per-query constant; rn_db is not read.  (Rounding 1 - t to float moves a threshold by <= 1.2e-7, far
// inside the slack between the 3e-6 error bound and the EPS = 1e-5 the thresholds are widened by.)
template <int TF_CAP, int QG, bool NORM, bool RANK>
__global__ __launch_bounds__(TF_THREADS, 2) void topk_filter_kernel(
    const float *__restrict__ db, const float *__restrict__ rn_db, int64_t n_db, const float *__restrict__ qs,
    const float *__restrict__ rn_q, int64_t n_q, int k, int n_slices, int32_t *__restrict__ cand_idx,
    int32_t *__restrict__ cand_cnt, RankFuse R, int64_t row_stride, const float *__restrict__ thr_init) {
    // Seeded thresholds (unit rows).  Before the pool is walked, the same filter + exact refine run on a SAMPLE of it -
    // every row_stride-th row, 16384 of them (this kernel with row_stride > 1, then topk_kernel) - and give each
    // query the exact k-th distance within the sample, d_k(sample).  The k-th smallest over ANY subset of the pool is
    // >= the k-th smallest over the pool, so every true top-k item has d~ <= d_k(sample) + EPS: the main launch starts
    // from that threshold (thr_init) instead of +inf - no warm-up rounds in which everything is appended, a handful of
    // appends per slice afterwards (the sample's quantile is k / 16384), hardly any compaction, which is what the
    // few-queries-large-pool shape spent its time on (64 queries x 512 slices x several compactions of a wave each) -
    // and candidate buffers small enough (TF_CAP = 128) to keep four query groups per workgroup at two workgroups per
    // CU.
    static_assert(!RANK || NORM, "the fused ranking reads the unit-length copy");
    constexpr int NQ = 16 * QG;
    __shared__ float cd[NQ][TF_CAP];
    __shared__ int32_t ci[NQ][TF_CAP];
    __shared__ float thr[NQ];
    __shared__ int cnt[NQ];
    __shared__ int bad[NQ];
    __shared__ unsigned long long mask;        // bit q: query q is compacted in this pass
    __shared__ int prev[NQ], grow;             // entries at the start of the round; round-length decision
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nn = lane & 15;
    // Block -> (query group, slice).  Every group has to see the whole pool, and what that costs is decided by who runs
    // next to whom: group-major order (all slices of group 0, then group 1 ...) made every group stream the pool from
    // HBM / the infinity cache by itself - 4096 queries x 2^21 codes with 32 queries per workgroup were 128 passes
    // over 256 MB = 34 GB at 4.1 TB/s: 8.3 ms, the whole kernel.  Slice-major order lets the groups of a slice run
    // together and share it in L2 - and since workgroups go to the XCDs round-robin by block index, each XCD (its
    // own 4 MB L2) takes every 8th slice and runs that slice's groups back to back.
    const int n_groups = gridDim.x / n_slices;
    int grp, slice;
    if (row_stride < 0) {                      // (host switch ASR_TOPK_SLICE_MAJOR=0: round 3's group-major order, for A/B runs)
        row_stride = -row_stride;
        grp = blockIdx.x / n_slices;
        slice = blockIdx.x - grp * n_slices;
    } else if ((n_slices & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        slice = (j / n_groups) * 8 + xcd;
        grp = j % n_groups;
    } else {
        slice = blockIdx.x / n_groups;
        grp = blockIdx.x - slice * n_groups;
    }
    TF_STAMP(0);
    const int64_t q0 = (int64_t)grp * NQ;
    const int64_t tiles = (n_db + 15) / 16;
    const int64_t t_lo = tiles * slice / n_slices, t_hi = tiles * (slice + 1) / n_slices;
    const int64_t n_db_pad = (n_db + 3) & ~(int64_t)3;           // rn_db is allocated (and zero-filled) up to here
    // B fragment: lane (k group g, query nn) holds dims 8g .. 8g+7 of its query; MFMA step j pairs dim 8g + j of both
    // RANK: cosine of d* + band per query - s >= chi: the pair is inside the band or closer (counted or examined);
    // the band's other edge is chi + 2 band.  Kept in LDS with the thresholds: the tile loop holds ONE constant per
    // query group and lane in registers (tq_r, ch_r below).
    __shared__ float chi_s[NQ];
    float bq[QG][8], rq[QG];
    int less[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const bool qvalid = q0 + 16 * u + nn < n_q;
        const int64_t qi = qvalid ? q0 + 16 * u + nn : n_q - 1;
        rq[u] = rn_q[qi];
#pragma unroll
        for (int j = 0; j < 8; ++j) bq[u][j] = NORM ? qs[qi * 32 + tf_dim(g, j)] * rq[u] : qs[qi * 32 + tf_dim(g, j)];
        less[u] = 0;
    }
    constexpr bool BF3 = NORM && ASR_TF_BF3;
    constexpr bool BF2 = BF3 && ASR_TF_BF2 && (!RANK || ASR_TF_BF2 > 1);     // (ASR_TF_BF2=2: the fused ranking too, band 5.5e-5)
    constexpr float RFB = BF2 ? RF_BAND_BF2 : RF_BAND;        // the fused ranking's band: >= the error bound of d~
    constexpr float EPSF = BF2 ? TF_EPS_BF2 : TF_EPS;
    Bf3 qb[BF3 ? QG : 1];
    if constexpr (BF3) {
#pragma unroll
        for (int u = 0; u < QG; ++u) qb[u] = split_bf3(bq[u]);
    }
    for (int e = tid; e < NQ; e += TF_THREADS) {
        const bool qvalid = q0 + e < n_q;
        thr[e] = thr_init ? thr_init[qvalid ? q0 + e : n_q - 1] : INFINITY;
        cnt[e] = 0; bad[e] = 0; prev[e] = 0;
        chi_s[e] = (RANK && qvalid) ? 1.0f - ((float)R.dstar[q0 + e] + RFB) : INFINITY;   // a padding lane never counts
    }
    __syncthreads();

    // compact the buffers selected by `m`: wave w takes queries w, w+4, w+8, w+12 - one wave per query, no workgroup
    // barrier inside (callers put one before and one after).  The k-th smallest d~ by a 4 x 8-bit radix select
    // (the wave's own 256-bin histogram in LDS, scan by shuffles), then {d~ <= d~_k + 2 EPS} is kept in place, in
    // order (a write position never passes the read position).  A bitonic sort of the 512 keys cost 30x this, and
    // one-query-at-a-time compaction through a dozen workgroup barriers still left 70 % of the wave cycles idle.
    __shared__ int hist[4][256];
    auto sortable = [](float v) -> unsigned {        // monotone map float -> unsigned (handles the -1e-7 of d~(x, x))
        const unsigned u = __float_as_uint(v);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    };
    auto wave_sync = []() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    auto compact = [&](unsigned long long m) {
        int *hw = hist[wave];
        for (int q = wave; q < NQ; q += 4) {
            if (!(m >> q & 1ull)) continue;           // wave-uniform
            const int n = cnt[q];
            float lim = INFINITY;
            if (n >= k) {
                unsigned prefix = 0;
                int rank = k;
                for (int pass = 3; pass >= 0; --pass) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) hw[4 * lane + j] = 0;
                    wave_sync();
                    const unsigned himask = pass == 3 ? 0u : (0xFFFFFFFFu << (8 * (pass + 1)));
                    for (int e = lane; e < n; e += 64) {
                        const unsigned u = sortable(cd[q][e]);
                        if ((u & himask) == prefix) atomicAdd(&hw[(u >> (8 * pass)) & 255], 1);
                    }
                    wave_sync();
                    const int h0 = hw[4 * lane], h1 = hw[4 * lane + 1], h2 = hw[4 * lane + 2], h3 = hw[4 * lane + 3];
                    const int tot = h0 + h1 + h2 + h3;
                    int incl = tot;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int v = __shfl_up(incl, o);
                        if (lane >= o) incl += v;
                    }
                    const int excl = incl - tot;
                    const bool hit = excl < rank && rank <= incl;        // exactly one lane
                    int bin = 4 * lane, c = excl;
                    if (rank > c + h0) { c += h0; ++bin; if (rank > c + h1) { c += h1; ++bin; if (rank > c + h2) { c += h2; ++bin; } } }
                    const int src = __ffsll((long long)__ballot(hit)) - 1;
                    const int fb = __shfl(bin, src), nr = __shfl(rank - c, src);
                    prefix |= (unsigned)fb << (8 * pass);
                    rank = nr;
                    wave_sync();
                }
                lim = __uint_as_float((prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix) + 2.0f * EPSF;
            }
            // keep {d~ <= lim} in place, 64 entries at a time
            int kept_n = 0;
            for (int base = 0; base < n; base += 64) {
                const int e = base + lane;
                const float dv = e < n ? cd[q][e] : INFINITY;
                const int iv = e < n ? ci[q][e] : 0;
                const bool keep_it = e < n && dv <= lim;
                const unsigned long long bal = __ballot(keep_it);
                const int before = __popcll(bal & ((1ull << lane) - 1ull));
                wave_sync();                                          // all reads of this chunk precede its writes
                if (keep_it) { cd[q][kept_n + before] = dv; ci[q][kept_n + before] = iv; }
                kept_n += __popcll(bal);
                wave_sync();
            }
            if (lane == 0) {
                int keep = kept_n;
                if (n >= k) thr[q] = lim;
                if (keep > TF_CAP / 2) { bad[q] = 1; keep = 0; thr[q] = -INFINITY; }     // near-tie mass: exact scan instead
                cnt[q] = keep;
            }
        }
    };

    // A round = L groups of 4 tiles per wave between two workgroup barriers.  L starts at 1 (everything passes the
    // +inf threshold) and doubles after a round that hardly appended anything, up to 16: once the thresholds are
    // tight the barriers + checks of a short round would dominate.  A long round is speculative: an append that finds
    // its buffer full marks the query bad (exact scan instead).
    // The reciprocal norms of a tile's items travel with its A fragment: a load issued after the MFMAs would have to
    // wait for every older load (vmcnt counts in order), i.e. for the prefetched next group as well.
    // tiles per wave and group.  (Round 4: two for four query groups - "they hold a tile's registers four times as long".
    // Round 5, with half the MFMAs per tile: 2 / 3 / 4 tiles - 64 x 2 M 0.115 / 0.111 / 0.113 ms, 512 x 2 M 0.529 / 0.510 /
    // 0.501, 4096 x 2 M fused 4.04 / - / 3.90, 1024 x 250 k 0.221 / 0.219 / 0.221; -DASR_TF_TPW4=2 builds the old form)
    // Round 6, after the straight-line tile loop: 4 / 6 / 8 tiles - 64 x 2 M 0.1077-0.1100 / 0.1060-0.1070 / 0.125 ms,
    // 512 x 2 M 0.485 / 0.471 / 0.512, 1024 x 250 k 0.215 / 0.212 / 0.234 (eight: the register budget of two workgroups
    // per CU is gone) - six.
#ifndef ASR_TF_TPW4
#define ASR_TF_TPW4 6
#endif
    // (the fused-ranking build keeps four: its triggered path exists once per unrolled tile, and with six copies the
    // 4096 x 2 M fused call went from 3.61 to 6.45 ms - instruction cache)
    constexpr int TPW = QG == 4 ? (RANK ? 4 : ASR_TF_TPW4) : 4, GT = 4 * TPW;
    auto load_group = [&](int64_t tg, float4 (&a0)[TPW], float4 (&a1)[TPW], float4 (&rn)[TPW]) {
#pragma unroll
        for (int r = 0; r < TPW; ++r) {
            const int64_t tile = tg + r * 4 + wave;
            const int64_t item = tile * 16 + nn;                 // A fragment: lane (item nn, k group g)
            // (items past the pool's end: NaN rows - see score_tile; tiles past the slice are never scored)
            a0[r] = make_float4(NAN, NAN, NAN, NAN); a1[r] = a0[r]; rn[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tile < t_hi && item < n_db) {
                const float4 *p = reinterpret_cast<const float4 *>(db + item * row_stride * 32 + TF_A0 * g);
                a0[r] = p[0]; a1[r] = p[TF_A1];
            }
            const int64_t it0 = tile * 16 + 4 * g;               // C rows of this lane
            if (!NORM && tile < t_hi && it0 + 3 < n_db_pad) rn[r] = *reinterpret_cast<const float4 *>(rn_db + it0);
        }
    };
    // one (tile, query group): eight MFMAs, then per distance the top-k test (+ rare append) and, RANK, the side of d*.
    // ONE copy of this body per query group in the kernel's code: the tile loops around it are real loops over rotating
    // registers (score_group) and the pool's last, partial tile is handled by NaN rows, not by a second instantiation.
    // Unrolled over the two prefetch buffers, the tiles of a group and full / partial tiles the triggered path existed 32
    // times per query group - 147 KB of code for the fused-ranking build, which takes that path on nearly every tile,
    // against an instruction cache of 64 KB per pair of CUs.
    auto score_tile = [&](int64_t tile, const float (&af)[8], const float (&rn4)[4], const float (&tq_r)[QG],
                          const float (&ch_r)[RANK ? QG : 1]) {
        const int64_t it0 = tile * 16 + 4 * g;                   // C: lane (g, nn) holds items it0 + rr against query 16u + nn
        Bf3 ab;
        if constexpr (BF3) ab = split_bf3(af);                  // once per tile, shared by the query groups
        // the accumulators of all query groups first: QG independent chains of MFMAs issue back to back (inside one
        // chain every MFMA waits for the previous one's result), then the epilogues
        floatx4_t accs[QG];
#pragma unroll
        for (int u = 0; u < QG; ++u) accs[u] = floatx4_t{0.f, 0.f, 0.f, 0.f};
#if defined(ASR_TF_ABL) && (ASR_TF_ABL & 2)          // timing experiment: no MFMAs (wrong results)
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            accs[u][0] = af[0] * bq[u][0]; accs[u][1] = af[1] * bq[u][1]; accs[u][2] = af[2] * bq[u][2]; accs[u][3] = af[3] * bq[u][3];
        }
#else
        if constexpr (BF2) {
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p1, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p2, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p1, accs[u], 0, 0, 0);
        } else if constexpr (BF3) {
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p3, qb[u].p1, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p3, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p2, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p1, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p2, accs[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p1, accs[u], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int u = 0; u < QG; ++u) accs[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bq[u][j], accs[u], 0, 0, 0);
        }
#endif
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            const floatx4_t acc = accs[u];
            const int qn = 16 * u + nn;
#if defined(ASR_TF_ABL) && (ASR_TF_ABL & 1)          // timing experiment: the epilogue never triggers (wrong results)
            if (NORM) { if (__ballot(acc[0] + acc[1] + acc[2] + acc[3] == 12345.0f) == 0) continue; }
#endif
            // Items past the pool's end (its last tile only) were loaded as NaN rows: their cosines are NaN and fail
            // every comparison below, like the NaN of a zero-norm row - no per-distance bound check anywhere.
            // For almost every (item, query) pair NOTHING happens - it is neither among the k best so far nor within
            // reach of d*.  One test per four distances decides that: the best of the lane's four against the per-query
            // constants (top-k threshold; RANK: lower cosine edge of the d* band - everything closer than that is
            // counted or examined), two v_max, one compare, one wave-uniform branch.  The constants sit in registers
            // (tq_r, ch_r: one per query group and lane, refreshed per round).
            float sc[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) sc[rr] = NORM ? acc[rr] : fmaf(-(acc[rr] * rn4[rr]), rq[u], 1.0f);
            const float tq = tq_r[u];
            // best of four: the largest cosine (unit rows) / the smallest distance (raw rows); NaN never wins
            const float b4 = NORM ? fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])) : fminf(fminf(sc[0], sc[1]), fminf(sc[2], sc[3]));
            const bool cand = NORM ? b4 >= tq : b4 <= tq;         // a top-k candidate among the four
            const float c_hi = RANK ? ch_r[RANK ? u : 0] : 0.0f, c_lo = c_hi + 2.0f * RFB;
            if (__ballot(cand || (RANK && b4 >= c_hi)) == 0) continue;
            bool band = false;
            if (RANK) {
                // closer than d* - band: counted; inside the band <=> (>= c_hi) but not (> c_lo): two compares and two
                // conditional adds per distance, one comparison of the two counts per lane
                int gt = 0, ge = 0;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    gt += sc[rr] > c_lo ? 1 : 0;
                    ge += sc[rr] >= c_hi ? 1 : 0;
                }
                less[u] += gt;
                band = ge != gt;
            }
            // the top-k side: ONE LDS atomic per lane that has anything to append.  RANK: the pairs that got here are
            // mostly the counted ones, a candidate among them is rare - one more wave-uniform test skips the flags
            if (!RANK || __ballot(cand) != 0) {
                unsigned pm = 0;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) pm |= (NORM ? sc[rr] >= tq : sc[rr] <= tq) ? 1u << rr : 0u;
                if (pm) {
                    int pos = atomicAdd(&cnt[qn], __popc(pm));
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (pm >> rr & 1u) {
                            if (pos < TF_CAP) { cd[qn][pos] = NORM ? 1.0f - sc[rr] : sc[rr]; ci[qn][pos] = (int32_t)(it0 + rr); }
                            else bad[qn] = 1;                    // speculative round overflowed: exact scan for this query
                            ++pos;
                        }
                }
            }
            if (RANK && __ballot(band) != 0 && band) {
                const int64_t qi = q0 + qn;
                const double ds = R.dstar[qi], nqd = R.norm_q[qi];
                const int64_t js = R.jstar[qi];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (sc[rr] >= c_hi && sc[rr] <= c_lo) {      // (NaN: never counted, like d < d*)
                        const int64_t it = it0 + rr;
                        const double de = cos_dist(dot2acc(qs + qi * 32, R.db_raw + it * 32, 32), nqd, R.norm_db[it]);
                        if (de < ds) atomicAdd(&R.counts[qi * 3], 1);
                        if (de == ds) {
                            atomicAdd(&R.counts[qi * 3 + 1], 1);
                            if (it + R.item_offset < js) atomicAdd(&R.counts[qi * 3 + 2], 1);
                        }
                    }
            }
        }
    };
    // the tiles of a group one after the other through register set 0 (the sets rotate: 8 moves per tile)
    auto score_group = [&](int64_t tg, float4 (&a0)[TPW], float4 (&a1)[TPW], float4 (&rn)[TPW], const float (&tq_r)[QG],
                           const float (&ch_r)[RANK ? QG : 1]) {
        // The tiles of a group as straight-line code (round 5): in the rolled loop below the compiler cannot count which
        // of the rotating registers' loads are outstanding and waits with vmcnt(0) after EVERY tile - i.e. for the whole
        // prefetched next group as soon as the first tile of this one is scored.  Unrolled, tile r waits for its own two
        // loads only.  Rolled / unrolled: 64 x 2 M 0.111 / 0.107 ms, 1024 x 250 k 0.221 / 0.214, 512 x 2 M 0.501 / 0.473,
        // 1024 x 65 k 0.119 / 0.114, fused 4096 x 2 M 3.90 / 3.66 (the fused build's triggered path exists four times now:
        // still inside the instruction cache).  -DASR_TF_UNROLL=0: the rolled loop everywhere, =1: only without the ranking.
#ifndef ASR_TF_UNROLL
#define ASR_TF_UNROLL 2
#endif
#if ASR_TF_UNROLL
        if constexpr ((!RANK || ASR_TF_UNROLL > 1) && NORM) {
#pragma unroll
            for (int r = 0; r < TPW; ++r) {
                const int64_t tile = tg + r * 4 + wave;
                if (tile < t_hi) {
                    const float af[8] = {a0[r].x, a0[r].y, a0[r].z, a0[r].w, a1[r].x, a1[r].y, a1[r].z, a1[r].w};
                    const float rn4[4] = {0.f, 0.f, 0.f, 0.f};
                    score_tile(tile, af, rn4, tq_r, ch_r);
                }
            }
            return;
        }
#endif
#pragma unroll 1
        for (int r = 0; r < TPW; ++r) {
            const int64_t tile = tg + r * 4 + wave;
            if (tile < t_hi) {
                const float af[8] = {a0[0].x, a0[0].y, a0[0].z, a0[0].w, a1[0].x, a1[0].y, a1[0].z, a1[0].w};
                const float rn4[4] = {rn[0].x, rn[0].y, rn[0].z, rn[0].w};
                score_tile(tile, af, rn4, tq_r, ch_r);
            }
#pragma unroll
            for (int i = 0; i + 1 < TPW; ++i) {
                a0[i] = a0[i + 1]; a1[i] = a1[i + 1];
                if (!NORM) rn[i] = rn[i + 1];
            }
        }
    };
    TF_STAMP(1);
    int n_rounds = 0, n_compact = 0;
    (void)n_rounds; (void)n_compact;
    int L = thr_init ? 4 : 1;
    for (int64_t tb = t_lo; tb < t_hi;) {
        // raw rows: the threshold on d~.  Unit rows: the cosine below which a pair is of no interest to anybody - the
        // smaller of the top-k threshold (1 - thr) and, RANK, the lower edge of the d* band
        float tq_r[QG], ch_r[RANK ? QG : 1];
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            const float th = thr[16 * u + nn];
            tq_r[u] = NORM ? 1.0f - th : th;
            if (RANK) ch_r[u] = chi_s[16 * u + nn];
        }
        float4 c0[TPW], c1[TPW], cr[TPW];
        load_group(tb, c0, c1, cr);
#pragma unroll 1
        for (int gI = 0; gI < L; ++gI) {
            const int64_t tg = tb + (int64_t)gI * GT;
            if (tg >= t_hi) break;
            float4 n0[TPW], n1[TPW], nr[TPW];
            const bool more = gI + 1 < L;
            if (more) load_group(tg + GT, n0, n1, nr);          // next group in flight during this one
            score_group(tg, c0, c1, cr, tq_r, ch_r);
            if (more) {
#pragma unroll
                for (int r = 0; r < TPW; ++r) { c0[r] = n0[r]; c1[r] = n1[r]; cr[r] = nr[r]; }
            }
        }
        tb += (int64_t)L * GT;
        __syncthreads();
        if (wave == 0) {                       // lanes 0..NQ-1: one query each
            int need = 0, lse = 0, app = 0;
            if (lane < NQ) {
                const int q = lane;
                if (bad[q]) { cnt[q] = 0; thr[q] = -INFINITY; }
                else if (cnt[q] > TF_CAP) { bad[q] = 1; cnt[q] = 0; thr[q] = -INFINITY; }
                app = cnt[q] - prev[q];
                // compact: buffer more than half full, or the first k entries are in (first finite threshold)
                const bool first = thr[q] == INFINITY && cnt[q] >= k;
                need = (cnt[q] > TF_CAP / 2) || first;
                lse = thr[q] == INFINITY && !first;
            }
            const unsigned long long m = __ballot(need != 0);        // (lanes >= NQ contribute zeros)
            const bool loose = __ballot(lse != 0) != 0;
            const bool many = __ballot(app > 16) != 0, flood = __ballot(app > 128) != 0;
            if (lane == 0) {
                mask = m;
                // the next round may be twice as long when this one hardly appended anything and every threshold is
                // finite; half as long when it appended a lot
                grow = (m == 0 && !loose && !many) ? 1 : (flood ? -1 : 0);
            }
        }
        __syncthreads();
        const unsigned long long m = mask;
        const int gr = grow;                   // wave-uniform
        ++n_rounds;
        if (m) { compact(m); __syncthreads(); ++n_compact; }
        for (int e = tid; e < NQ; e += TF_THREADS) prev[e] = cnt[e];
        if (gr > 0 && L < 16) L *= 2;
        else if (gr < 0 && L > 1) L >>= 1;
        __syncthreads();
    }
    // final compaction: only buffers that would not fit a candidate list, or that never had a threshold - every other
    // entry was appended under a finite threshold and is a legitimate survivor (a few more rows for the exact kernel
    // cost less than a radix select per query and slice: 64 queries x 488 slices of the few-queries shape)
    TF_STAMP(2);
    {
        // one lane per query decides (every thread walking the 64 counters and thresholds in turn cost 6 us per workgroup)
        if (wave == 0) {
            const bool need = lane < NQ && (cnt[lane < NQ ? lane : 0] > (TF_OUT < TF_CAP ? TF_OUT : TF_CAP) / 2 ||
                                            !(thr[lane < NQ ? lane : 0] < INFINITY));
            const unsigned long long m = __ballot(need);
            if (lane == 0) mask = m;
        }
        __syncthreads();
        const unsigned long long fm = mask;
        if (fm) { compact(fm); __syncthreads(); }
    }
    TF_STAMP(3);
    // hand-over: wave w writes the lists of queries w, w + 4, ... (one query after the other through all 256 threads: 9 us)
    for (int q = wave; q < NQ; q += 4) {
        if (q0 + q >= n_q) break;                              // wave-uniform
        const int n = cnt[q];
        const int64_t list = (q0 + q) * n_slices + slice;
        const bool over = bad[q] || n > TF_OUT;
        if (lane == 0) cand_cnt[list] = over ? -1 : n;
        if (!over)
            for (int e = lane; e < n; e += 64) cand_idx[list * TF_OUT + e] = ci[q][e];
    }
    if (RANK) {
#pragma unroll
        for (int u = 0; u < QG; ++u)
            if (less[u]) atomicAdd(&R.counts[(q0 + 16 * u + nn) * 3], less[u]);      // (padding lanes hold zero)
    }
    TF_STAMP(4);
    TF_NOTE(5, n_rounds);
    TF_NOTE(6, n_compact);
    TF_NOTE(7, t_hi - t_lo);
}

// thr0[q] = exact k-th distance of query q within the sample, widened by EPS + EPS_BF2 (covers the filter's error bound and the
// rounding to float); +inf when the sample gave fewer than k finite distances
__global__ __launch_bounds__(256) void seed_threshold_kernel(const int32_t *__restrict__ idx, const double *__restrict__ dist,
                                                             int64_t n_q, int k, float *__restrict__ thr0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_q) return;
    const double d = dist[i * k + k - 1];
    thr0[i] = (idx[i * k + k - 1] >= 0 && d < 1e30) ? (float)d + (TF_EPS + TF_EPS_BF2) : INFINITY;   // (>= the filter's error bound)
}

// The seeding pass without the filter machinery (round 4b; before: the filter over the sample, the exact refine of its
// survivors and seed_threshold_kernel - three dependent launches whose workgroups each live ~50 us whatever the sample's
// size: 100 us for 64 queries, 190 us for 1024, 660 us for 4096).  A threshold does not have to be exact, only an upper
// bound of d_k(sample) about as tight as the filter's own error bound:
//   sample_keys_kernel    a workgroup takes QB queries and 256 sampled unit-length rows (eight lanes per row: one 16-byte
//                         load each - a wave's load is eight whole 128-byte lines - and a three-step lane reduction),
//                         all loads of a thread in flight at once; the grid covers (row blocks) x (query blocks), so one
//                         query alone still spreads over 64 workgroups.  Key = floor(d~ * 2^15), 16 bits.
//   sample_select_kernel  one workgroup per query: the k-th smallest key by a 2 x 8-bit radix select, keys in registers.
// thr0 = (key_k + 1) 2^-15 + EPS + EPS_BF2 >= d~_k + 3e-6 + the filter's error bound (4.9e-5 with two bf16 planes).
// |d~ - d| <= 3e-6 for this summation order as for the MFMA's (32
// products in fp32), so the k-th smallest d~ of the sample is within 3e-6 of its k-th smallest exact distance, which is
// >= d_k of the pool; a true top-k row has filter distance <= d_k + 3e-6 <= d~_k + 6e-6 < thr0.  The 16-bit key loosens
// the threshold by <= 3.1e-5 (a few per cent more survivors at worst).  NaN cosines (a zero-norm row or query) take
// the last key: far away, which is what the exact kernel does with them, and a query whose k-th key is one of the last
// two gets +inf (its small buffers overflow: exact scan, as before).
constexpr int SS_ROWS_MAX = 32768;
__device__ __forceinline__ void sample_select_body(const uint16_t *__restrict__ keys, int64_t rows, int k,
                                                   float *__restrict__ thr0, int64_t qi, int *hist, int *sel);
// tickets (QB = 1, may be null): one zero-initialised counter per query - the LAST row block of a query to store its keys
// selects the query's threshold itself (sample_select_body): sample_select_kernel's launch disappears from the
// few-queries call
template <int QB>
__global__ __launch_bounds__(256) void sample_keys_kernel(const float *__restrict__ unit, int64_t rows, int64_t stride,
                                                          const float *__restrict__ qs, const double *__restrict__ norm_q,
                                                          int64_t n_q, uint16_t *__restrict__ keys,
                                                          double *__restrict__ norm_q_out, float *__restrict__ rn_q_out,
                                                          unsigned *__restrict__ tickets, int k, float *__restrict__ thr0) {
    // norm_q_out (may be null): the float64 query norms do not exist yet - every workgroup forms the ones it needs
    // (row_norms32_kernel's arithmetic: the norm is part of the bit-exact distance) and row block 0 stores them, with
    // their fp32 reciprocals, for the filter and the exact kernel behind this launch: two launches less per call
    __shared__ uint16_t lk[QB][256];
    const int tid = threadIdx.x, sub = tid & 7, rg = tid >> 3;
    const int64_t r0 = (int64_t)blockIdx.x * 256, q0 = (int64_t)blockIdx.y * QB;
    float4 x[8];
    const float4 *src = reinterpret_cast<const float4 *>(unit) + sub;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[(r0 + rg + 32 * j) * stride * 8];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const int64_t qi = q0 + u < n_q ? q0 + u : n_q - 1;
        const double nq = norm_q_out ? __dsqrt_rn(dot2acc(qs + qi * 32, qs + qi * 32, 32)) : norm_q[qi];
        const float rq = (float)(1.0 / nq);                          // (the filter's rn_q: rnorm_f32_kernel)
        if (norm_q_out && blockIdx.x == 0 && tid == 0 && q0 + u < n_q) { norm_q_out[qi] = nq; rn_q_out[qi] = rq; }
        float4 qn = *reinterpret_cast<const float4 *>(qs + qi * 32 + 4 * sub);
        qn.x *= rq; qn.y *= rq; qn.z *= rq; qn.w *= rq;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float sdot = x[j].x * qn.x;
            sdot = fmaf(x[j].y, qn.y, sdot); sdot = fmaf(x[j].z, qn.z, sdot); sdot = fmaf(x[j].w, qn.w, sdot);
            sdot += __shfl_xor(sdot, 1); sdot += __shfl_xor(sdot, 2); sdot += __shfl_xor(sdot, 4);
            const float d = 1.0f - sdot;
            // d >= 0: floor(d 2^15), d slightly negative (a row against itself): 0, NaN: the last key
            const unsigned key = d >= 0.0f ? (unsigned)fminf(d * 32768.0f, 65534.0f) : (d < 0.0f ? 0u : 65535u);
            if (sub == 0) lk[u][rg + 32 * j] = (uint16_t)key;
        }
    }
    __syncthreads();
    for (int e = tid; e < QB * 128; e += 256) {                      // two keys per store
        const int u = e >> 7, c = e & 127;
        if (q0 + u < n_q)
            reinterpret_cast<uint32_t *>(keys + (q0 + u) * rows + r0)[c] = reinterpret_cast<const uint32_t *>(lk[u])[c];
    }
    if constexpr (QB == 1) {
        if (!tickets) return;
        __shared__ int last_block;
        __shared__ int hist[256];
        __shared__ int sel[2];
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const unsigned t = atomicAdd(&tickets[q0], 1u);
            last_block = (t == gridDim.x - 1u) ? 1 : 0;
            if (last_block) tickets[q0] = 0u;
        }
        __syncthreads();
        if (!last_block) return;
        __threadfence();
        sample_select_body(keys, rows, k, thr0, q0, hist, sel);
    }
}

// k-th smallest 16-bit key of query qi's sample -> its starting threshold (all 256 threads of a workgroup; hist: 256
// ints, sel: 2 ints of LDS)
__device__ __forceinline__ void sample_select_body(const uint16_t *__restrict__ keys, int64_t rows, int k,
                                                   float *__restrict__ thr0, int64_t qi, int *hist, int *sel) {
    const int tid = threadIdx.x, lane = tid & 63;
    const uint4 *src = reinterpret_cast<const uint4 *>(keys + qi * rows);
    uint4 kv[SS_ROWS_MAX / 2048];
#pragma unroll
    for (int j = 0; j < SS_ROWS_MAX / 2048; ++j)                      // eight keys per load; rows: a multiple of 4096
        kv[j] = (int64_t)(j * 256 + tid) * 8 < rows ? src[j * 256 + tid] : make_uint4(~0u, ~0u, ~0u, ~0u);
    unsigned prefix = 0;
    int rank = k;
    for (int pass = 1; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SS_ROWS_MAX / 2048; ++j) {
            if ((int64_t)(j * 256 + tid) * 8 >= rows) continue;
            const unsigned w[4] = {kv[j].x, kv[j].y, kv[j].z, kv[j].w};
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const unsigned u = (w[h >> 1] >> (16 * (h & 1))) & 0xFFFFu;
                if (pass == 1) atomicAdd(&hist[u >> 8], 1);
                else if ((u >> 8) == prefix) atomicAdd(&hist[u & 255u], 1);
            }
        }
        __syncthreads();
        if (tid < 64) {                                              // four bins per lane, scan by shuffles
            const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            const int tot = h0 + h1 + h2 + h3;
            int incl = tot;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (lane >= o) incl += v;
            }
            const int excl = incl - tot;
            if (excl < rank && rank <= incl) {                       // exactly one lane (rows >= k)
                int bin = 4 * lane, c = excl;
                if (rank > c + h0) { c += h0; ++bin; if (rank > c + h1) { c += h1; ++bin; if (rank > c + h2) { c += h2; ++bin; } } }
                sel[0] = bin; sel[1] = rank - c;
            }
        }
        __syncthreads();
        if (pass == 1) prefix = (unsigned)sel[0];
        else prefix = (prefix << 8) | (unsigned)sel[0];
        rank = sel[1];
    }
    // (+ the sample keys' own 3e-6 and the filter's error bound, 4.9e-5 with two bf16 planes: TF_EPS_BF2)
    if (tid == 0) thr0[qi] = prefix >= 65534u ? INFINITY : (float)(prefix + 1u) * (1.0f / 32768.0f) + (TF_EPS + TF_EPS_BF2);
}

// The sample keys on the matrix cores (round 5, second half).  sample_keys_kernel spends eight lanes and ~12 vector
// instructions on ONE (row, query) cosine: with the seed sample grown to 32 768 rows it was the second largest kernel of
// the few-queries call (64 x 2 M: 21.9 us of 125) and 14-23 % of the many-queries ones (1024 x 250 k: 59.5 us of 254;
// 4096 x 2 M fused: 0.65 ms of 4.55).  Here a wave takes a tile of 16 sampled rows (the filter's A fragment, split into
// three bf16 planes) against QG groups of 16 normalised queries (B fragments, split once per workgroup): six
// v_mfma_f32_16x16x32_bf16 per group give 256 cosines - the exact three-plane form, |d~ - d| <= 3e-6 as thr0's
// derivation assumes (NOT the two-plane form of the main filter pass: the sample is 1/64 of the pool, its MFMAs do
// not matter).  Keys as in sample_keys_kernel; they pass through LDS so that a query's 128 keys leave as one 256-byte
// segment.  Grid: (rows / 128) x (query blocks of 16 QG).
constexpr int SKM_ROWS = 128;
template <int QG>
__global__ __launch_bounds__(256) void sample_keys_mfma_kernel(const float *__restrict__ unit, int64_t rows, int64_t stride,
                                                               const float *__restrict__ qs, const double *__restrict__ norm_q,
                                                               int64_t n_q, uint16_t *__restrict__ keys,
                                                               double *__restrict__ norm_q_out, float *__restrict__ rn_q_out) {
    constexpr int NQ = 16 * QG;
    __shared__ uint16_t lk[NQ][SKM_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nn = lane & 15;
    const int64_t r0 = (int64_t)blockIdx.x * SKM_ROWS, q0 = (int64_t)blockIdx.y * NQ;
    // the rows first: two tiles per wave, both loads in flight while the queries are prepared
    float4 a0[2], a1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t row = r0 + (2 * wave + t) * 16 + nn;         // A fragment: lane (item nn, k group g)
        const float4 *p = reinterpret_cast<const float4 *>(unit + row * stride * 32 + TF_A0 * g);
        a0[t] = p[0]; a1[t] = p[TF_A1];
    }
    Bf3 qb[QG];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const bool qvalid = q0 + 16 * u + nn < n_q;
        const int64_t qi = qvalid ? q0 + 16 * u + nn : n_q - 1;
        // norm_q_out (may be null): the float64 query norms do not exist yet (see sample_keys_kernel)
        const double nq = norm_q_out ? __dsqrt_rn(dot2acc(qs + qi * 32, qs + qi * 32, 32)) : norm_q[qi];
        const float rq = (float)(1.0 / nq);                          // (the filter's rn_q: rnorm_f32_kernel)
        if (norm_q_out && blockIdx.x == 0 && wave == 0 && g == 0 && qvalid) { norm_q_out[qi] = nq; rn_q_out[qi] = rq; }
        float bq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bq[j] = qs[qi * 32 + tf_dim(g, j)] * rq;
        qb[u] = split_bf3(bq);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float af[8] = {a0[t].x, a0[t].y, a0[t].z, a0[t].w, a1[t].x, a1[t].y, a1[t].z, a1[t].w};
        const Bf3 ab = split_bf3(af);
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            floatx4_r acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p3, qb[u].p1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p3, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p2, qb[u].p1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab.p1, qb[u].p1, acc, 0, 0, 0);
            // C: lane (g, nn) holds rows 4g + rr of the tile against query 16u + nn
            unsigned kq[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float d = 1.0f - acc[rr];
                // d >= 0: floor(d 2^15), d slightly negative (a row against itself): 0, NaN: the last key
                kq[rr] = d >= 0.0f ? (unsigned)fminf(d * 32768.0f, 65534.0f) : (d < 0.0f ? 0u : 65535u);
            }
            uint2 w;
            w.x = kq[0] | (kq[1] << 16); w.y = kq[2] | (kq[3] << 16);
            *reinterpret_cast<uint2 *>(&lk[16 * u + nn][(2 * wave + t) * 16 + 4 * g]) = w;
        }
    }
    __syncthreads();
    for (int e = tid; e < NQ * (SKM_ROWS / 2); e += 256) {          // two keys per store, 256 bytes per query
        const int u = e / (SKM_ROWS / 2), c = e % (SKM_ROWS / 2);
        if (q0 + u < n_q)
            reinterpret_cast<uint32_t *>(keys + (q0 + u) * rows + r0)[c] = reinterpret_cast<const uint32_t *>(lk[u])[c];
    }
}

__global__ __launch_bounds__(256) void sample_select_kernel(const uint16_t *__restrict__ keys, int64_t rows, int k,
                                                            float *__restrict__ thr0) {
    __shared__ int hist[256];
    __shared__ int sel[2];
    sample_select_body(keys, rows, k, thr0, (int64_t)blockIdx.x, hist, sel);
}

// ---- ONE query (up to four) against a large resident pool: a single streaming pass (round 5) ---------------------------
// The reference's server asks exactly this: cdist(DB, q (1,32)) -> argsort[:n_candidates] per incoming frame
// (audio_sheet_server.py:530-563).  The general path - threshold sample, MFMA filter for a group of 16 queries, exact
// refine, merge: five dependent launches - costs 0.12 ms for it; 256 MB at the HBM rate are ~45 us.  Here workgroup
// (c, q) owns slice c of the pool (<= SCAN_ROWS rows) for query q and does everything about that slice by itself:
//   1. fp32 cosines against the unit-length rows (8 lanes per row, coalesced float4 loads) -> 16-bit keys
//      floor(d~ 2^15) in LDS (sample_keys_kernel's arithmetic and key);
//   2. the k-th smallest key of the slice by a two-pass radix select (sample_select_body's rule): every row whose key is
//      <= that key + 1 survives - |d~ - d| <= 3e-6 and a key bin is 3.05e-5 wide, so the survivors are a SUPERSET of the
//      slice's k nearest rows, ties included (the argument of the seeded filter thresholds, per slice);
//   3. the survivors' exact float64 distances (topk_kernel's arithmetic: dot2acc + cos_dist on the raw rows and the
//      float64 norms), a bitonic sort of the few of them, k (distance, index) keys to part_idx / part_dist [q][c][k].
// topk_merge_kernel then merges the slices' lists in a tree (keys are exact, so merging partial top-k lists is exact).
// A slice whose survivors do not fit (masses of near-ties: > SCAN_SURV) orders ALL its rows exactly, batch by batch -
// slow and correct.  norm_q_out (may be null): the query norms do not exist yet; workgroup (0, q) stores it.
constexpr int SCAN_ROWS = 8192;              // rows per slice (16-bit keys in LDS: 16 KB)
constexpr int SCAN_SURV = 1024;              // survivors ordered in one sort (16 KB of keys)
constexpr int SCAN_NQ_MAX = 128;            // queries one call may send down this path
__global__ __launch_bounds__(256) void topk_scan_kernel(const float *__restrict__ unit, const float *__restrict__ db,
                                                        const double *__restrict__ norm_db, int64_t n_db,
                                                        const float *__restrict__ qs, const double *__restrict__ norm_q,
                                                        double *__restrict__ norm_q_out, int k, int64_t idx_offset,
                                                        int32_t *__restrict__ part_idx, double *__restrict__ part_dist) {
    __shared__ uint16_t skey[SCAN_ROWS];
    __shared__ TopkKey keys[SCAN_SURV];
    __shared__ int32_t surv[SCAN_SURV];
    __shared__ int hist[256];
    __shared__ int sel[2];
    __shared__ int nsurv;
    __shared__ __align__(16) float ql[32];          // read as float4 (ds_read_b128)
    const int tid = threadIdx.x, lane = tid & 63;
    const int chunk = blockIdx.x, n_chunks = gridDim.x;
    const int64_t qi = blockIdx.y;
    const int64_t r0 = n_db * chunk / n_chunks, r1 = n_db * (chunk + 1) / n_chunks;
    const int nr = (int)(r1 - r0);
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    if (tid < 32) ql[tid] = qs[qi * 32 + tid];
    if (tid == 0) nsurv = 0;
    __syncthreads();
    const double nq = norm_q_out ? __dsqrt_rn(dot2acc(ql, ql, 32)) : norm_q[qi];
    if (norm_q_out && chunk == 0 && tid == 0) norm_q_out[qi] = nq;
    // ---- 1. keys
    {
        const int sub = tid & 7, rg = tid >> 3;
        const float rq = (float)(1.0 / nq);
        float4 qn = *reinterpret_cast<const float4 *>(ql + 4 * sub);
        qn.x *= rq; qn.y *= rq; qn.z *= rq; qn.w *= rq;
        const float4 *src = reinterpret_cast<const float4 *>(unit + r0 * 32) + sub;
        for (int base = 0; base < nr; base += 256) {                 // 8 x 32 rows in flight per iteration
            float4 x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = base + rg + 32 * j;
                x[j] = r < nr ? src[(size_t)r * 8] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float sdot = x[j].x * qn.x;
                sdot = fmaf(x[j].y, qn.y, sdot); sdot = fmaf(x[j].z, qn.z, sdot); sdot = fmaf(x[j].w, qn.w, sdot);
                sdot += __shfl_xor(sdot, 1); sdot += __shfl_xor(sdot, 2); sdot += __shfl_xor(sdot, 4);
                const float d = 1.0f - sdot;
                const unsigned key = d >= 0.0f ? (unsigned)fminf(d * 32768.0f, 65534.0f) : (d < 0.0f ? 0u : 65535u);
                const int r = base + rg + 32 * j;
                if (sub == 0 && r < nr) skey[r] = (uint16_t)key;
            }
        }
    }
    __syncthreads();
    // ---- 2. the k-th smallest key of the slice (all rows survive when the slice has no more than k)
    unsigned cut = 65536u;                                           // survivors: key <= cut
    if (nr > k) {
        unsigned prefix = 0;
        int rank = k;
        for (int pass = 1; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            for (int e = tid; e < nr; e += 256) {
                const unsigned u = skey[e];
                if (pass == 1) atomicAdd(&hist[u >> 8], 1);
                else if ((u >> 8) == prefix) atomicAdd(&hist[u & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {                                          // four bins per lane, scan by shuffles
                const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const int tot = h0 + h1 + h2 + h3;
                int incl = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                const int excl = incl - tot;
                if (excl < rank && rank <= incl) {                   // exactly one lane
                    int bin = 4 * lane, c = excl;
                    if (rank > c + h0) { c += h0; ++bin; if (rank > c + h1) { c += h1; ++bin; if (rank > c + h2) { c += h2; ++bin; } } }
                    sel[0] = bin; sel[1] = rank - c;
                }
            }
            __syncthreads();
            if (pass == 1) prefix = (unsigned)sel[0];
            else prefix = (prefix << 8) | (unsigned)sel[0];
            rank = sel[1];
            __syncthreads();
        }
        // d~ <= (key_k + 1) / 2^15 + 2 EPS (the seeded thresholds' rule) <=> key <= key_k + 1: 2 EPS = 2e-5 is 0.66 of a bin
        cut = prefix >= 65534u ? 65536u : prefix + 1u;
    }
    // ---- 3. survivors -> exact keys
    for (int e = tid; e < nr; e += 256)
        if ((unsigned)skey[e] <= cut) {
            const int pos = atomicAdd(&nsurv, 1);
            if (pos < SCAN_SURV) surv[pos] = e;
        }
    __syncthreads();
    const int ns = nsurv;
    const int64_t out = (qi * n_chunks + chunk) * k;
    auto exact_key = [&](int64_t j) {
        const double d = cos_dist(dot2acc(ql, db + j * 32, 32), nq, norm_db[j]);
        TopkKey kk;
        kk.d = (unsigned long long)__double_as_longlong(d + 0.0);     // +0.0: never the -0.0 pattern
        kk.j = j + idx_offset;
        return kk;
    };
    auto sort_keys = [&](int sort_n) {                               // bitonic, ascending (key_less)
        for (int size = 2; size <= sort_n; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int e = tid; e < sort_n / 2; e += 256) {
                    const int lo = 2 * e - (e & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const TopkKey a = keys[lo], b = keys[hi];
                    if (key_less(b, a) == up) { keys[lo] = b; keys[hi] = a; }
                }
                __syncthreads();
            }
    };
    int filled = SCAN_SURV;                                          // entries of keys[] that hold a key or +inf
    if (ns <= SCAN_SURV) {
        int sort_n = 32;
        while (sort_n < ns) sort_n <<= 1;
        filled = sort_n;
        for (int e = tid; e < sort_n; e += 256) keys[e] = e < ns ? exact_key(r0 + surv[e]) : inf;
        __syncthreads();
        sort_keys(sort_n);
    } else {
        // masses of near-ties: every row of the slice, exactly, SCAN_SURV - k at a time behind the best k so far
        for (int e = tid; e < SCAN_SURV; e += 256) keys[e] = inf;
        __syncthreads();
        const int step = SCAN_SURV - TOPK_KMAX;
        for (int base = 0; base < nr; base += step) {
            for (int e = tid; e < step; e += 256) keys[TOPK_KMAX + e] = base + e < nr ? exact_key(r0 + base + e) : inf;
            __syncthreads();
            sort_keys(SCAN_SURV);
            for (int e = tid; e < SCAN_SURV; e += 256)
                if (e >= k) keys[e] = inf;
            __syncthreads();
        }
    }
    for (int e = tid; e < k; e += 256) {
        const TopkKey kk = e < filled ? keys[e] : inf;
        const bool real = kk.j != inf.j;
        part_idx[out + e] = real ? (int32_t)kk.j : -1;
        part_dist[out + e] = __longlong_as_double((long long)kk.d);
    }
}

// Merge of MANY sorted partial lists of one query (up to 512 lists of k <= 128 keys from topk_scan_kernel) in one small
// launch.  Only the lists whose HEAD is among the k smallest heads can contribute: an entry x of any other list L has
// k keys below it - those k heads are all <= h_k < head(L) <= x (keys are distinct) - and of those lists only entries
// with d <= d(h_k), since h_k itself bounds the k-th key of the union from above.
//   fast path: the k-th smallest head is located to within one 16-bit bin of its distance (two-pass radix select over the
//   <= 512 heads, no sort); every entry below that bin's upper edge, taken from the lists whose head lies at or below the
//   bin, is a superset of the answer - typically k .. 2k keys - and is ordered by one small bitonic network;
//   exact path (more than 1024 such entries: masses of near-ties): order the heads, take the k-th, and order the entries of
//   the <= k lists at or below it behind the best k so far, 896 / k lists per sort.
// The merge tree this replaces sorted 2048 keys per group of 64 lists: 50 of the single-query call's 110 us.
__global__ __launch_bounds__(TOPK_THREADS) void topk_merge_heads_kernel(const int32_t *__restrict__ part_idx,
                                                                        const double *__restrict__ part_dist, int n_lists,
                                                                        int k, int64_t n_db_full, int32_t *__restrict__ idx_out,
                                                                        double *__restrict__ dist_out) {
    __shared__ TopkKey heads[512];
    __shared__ TopkKey keys[1024];
    __shared__ uint16_t hkey[512];
    __shared__ int hist[256];
    __shared__ int sel[2];
    __shared__ int ngath, nkeys, nreal;
    __shared__ int picked[512];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t qi = blockIdx.x;
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    const int32_t *pi = part_idx + qi * (int64_t)n_lists * k;
    const double *pd = part_dist + qi * (int64_t)n_lists * k;
    auto load = [&](int e) {
        TopkKey kk = inf;
        const int32_t j = pi[e];
        if (j >= 0) { kk.d = (unsigned long long)__double_as_longlong(pd[e]); kk.j = j; }
        return kk;
    };
    auto sort_keys = [&](TopkKey *a, int sort_n) {
        for (int size = 2; size <= sort_n; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int e = tid; e < sort_n / 2; e += TOPK_THREADS) {
                    const int lo = 2 * e - (e & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const TopkKey x = a[lo], y = a[hi];
                    if (key_less(y, x) == up) { a[lo] = y; a[hi] = x; }
                }
                __syncthreads();
            }
    };
    auto write_out = [&]() {
        for (int e = tid; e < k; e += TOPK_THREADS) {
            const TopkKey kk = keys[e];
            const bool valid = e < n_db_full && kk.j != inf.j;
            idx_out[qi * k + e] = valid ? (int32_t)kk.j : -1;
            dist_out[qi * k + e] = valid ? __longlong_as_double((long long)kk.d) : __longlong_as_double(0x7ff0000000000000LL);
        }
    };
    int hn = 32;
    while (hn < n_lists) hn <<= 1;
    if (tid == 0) { ngath = 0; nkeys = 0; nreal = 0; }
    for (int e = tid; e < 1024; e += TOPK_THREADS) keys[e] = inf;
    __syncthreads();
    for (int l = tid; l < hn; l += TOPK_THREADS) {
        const TopkKey h = l < n_lists ? load(l * k) : inf;
        heads[l] = h;
        unsigned key = 65535u;                                       // empty lists / NaN distances: the last bin
        if (h.j != inf.j) {
            const double d = __longlong_as_double((long long)h.d);
            if (d == d) key = (unsigned)fmin(d * 32768.0, 65534.0);  // distances are >= 0
            atomicAdd(&nreal, 1);
        }
        hkey[l] = (uint16_t)key;
    }
    __syncthreads();
    // ---- fast path: the bin of the k-th smallest head
    unsigned kb = 65535u;
    if (nreal > k) {
        unsigned prefix = 0;
        int rank = k;
        for (int pass = 1; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            for (int l = tid; l < n_lists; l += TOPK_THREADS) {
                const unsigned u = hkey[l];
                if (pass == 1) atomicAdd(&hist[u >> 8], 1);
                else if ((u >> 8) == prefix) atomicAdd(&hist[u & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {
                const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                const int tot = h0 + h1 + h2 + h3;
                int incl = tot;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                const int excl = incl - tot;
                if (excl < rank && rank <= incl) {
                    int bin = 4 * lane, c = excl;
                    if (rank > c + h0) { c += h0; ++bin; if (rank > c + h1) { c += h1; ++bin; if (rank > c + h2) { c += h2; ++bin; } } }
                    sel[0] = bin; sel[1] = rank - c;
                }
            }
            __syncthreads();
            if (pass == 1) prefix = (unsigned)sel[0];
            else prefix = (prefix << 8) | (unsigned)sel[0];
            rank = sel[1];
            __syncthreads();
        }
        kb = prefix;
    }
    // every entry with d < (kb + 1) / 2^15 (all entries when that bin is the last one), from the lists whose head is there
    const bool all = kb >= 65534u;
    const double edge = (double)(kb + 1u) * (1.0 / 32768.0);
    for (int l = tid; l < n_lists; l += TOPK_THREADS)
        if (heads[l].j != inf.j && (unsigned)hkey[l] <= kb) picked[atomicAdd(&ngath, 1)] = l;
    __syncthreads();
    const int ng = ngath;
    for (int e = tid; e < ng * k; e += TOPK_THREADS) {
        const TopkKey kk = load(picked[e / k] * k + e % k);
        if (kk.j == inf.j) continue;
        const double d = __longlong_as_double((long long)kk.d);
        if (all || d < edge) {
            const int pos = atomicAdd(&nkeys, 1);
            if (pos < 1024) keys[pos] = kk;
        }
    }
    __syncthreads();
    if (nkeys <= 1024) {
        int sort_n = 32;
        while (sort_n < nkeys) sort_n <<= 1;
        sort_keys(keys, sort_n);
        write_out();
        return;
    }
    // ---- exact path: the heads in order, the <= k lists at or below the k-th, (1024 - 128) / k lists per sort behind the
    // best k so far (k <= 32: one sort)
    __syncthreads();
    for (int e = tid; e < 1024; e += TOPK_THREADS) keys[e] = inf;
    if (tid == 0) ngath = 0;
    __syncthreads();
    sort_keys(heads, hn);
    const TopkKey hk = heads[(k <= n_lists ? k : n_lists) - 1];      // the k-th smallest head (+inf: fewer than k real lists)
    for (int l = tid; l < n_lists; l += TOPK_THREADS) {
        const TopkKey h = load(l * k);
        if (h.j != inf.j && !key_less(hk, h)) picked[atomicAdd(&ngath, 1)] = l;      // head <= h_k: at most k such lists
    }
    __syncthreads();
    const int per = (1024 - TOPK_KMAX) / k, nl = ngath;
    for (int l0 = 0; l0 < nl; l0 += per) {
        const int cnt = (nl - l0 < per ? nl - l0 : per) * k;
        for (int e = tid; e < 1024 - TOPK_KMAX; e += TOPK_THREADS)
            keys[TOPK_KMAX + e] = e < cnt ? load(picked[l0 + e / k] * k + e % k) : inf;
        __syncthreads();
        sort_keys(keys, 1024);
        for (int e = tid; e < 1024; e += TOPK_THREADS)
            if (e >= k) keys[e] = inf;
        __syncthreads();
    }
    write_out();
}

// query groups of 16 per filter workgroup: two (32 queries, 64 KB of candidate buffers) once there are enough queries to
// fill the chip that way - every item tile then serves twice the queries per trip through L2; four (128 KB, unit-length
// data base only) when ALL queries fit one workgroup - 64 queries, the live server's shape
// (audio_sheet_server.py:530-563): the pool is then streamed exactly once.
static int topk_query_groups(int64_t n_q, int64_t n_db, int k, bool unit, bool seeded) {
    static const int qg_env = getenv("ASR_TOPK_QG") ? atoi(getenv("ASR_TOPK_QG")) : 0;
    if (k > 32) return 1;
    if (qg_env == 1 || qg_env == 2 || (qg_env == 4 && unit && seeded)) return qg_env;
    // (fp32 MFMAs, round 4 first half - 1024 queries x 250 k codes: two groups 0.475 ms, four 0.487)
    // With the products on the bf16 MFMA what a call costs beyond its fixed part is the pool re-read once per query
    // group of a workgroup from L2 / the memory-side cache: four groups halve that and win once there is enough work
    // (measured, two / four groups: 512 x 250 k 0.232 / 0.200 ms, 1024 x 250 k 0.355 / 0.326, 128 x 2 M 0.324 / 0.285,
    // 512 x 2 M 0.886 / 0.736, 2000 x 1 M 1.77 / 1.40; below ~1e8 pairs the halved workgroup count costs more:
    // 128 x 250 k 0.111 / 0.138, 256 x 250 k 0.138 / 0.150, 1024 x 65 k 0.165 / 0.183)
    if (unit && seeded)
        return n_q <= 16 ? 1 : n_q <= 32 ? 2 : n_q <= 64 ? 4 : (double)n_q * (double)n_db >= 1e8 ? 4 : 2;
    return n_q >= 2048 ? 2 : 1;
}

// rows of the pool the threshold-seeding pass looks at, and the smallest pool it pays for
// rows of the seeding sample (slices of 1024).  Measured, 64 queries x 2 M codes: 16 384 rows 0.30 ms, 65 536 rows 0.34,
// 131 072 rows 0.38 - a four times tighter threshold cuts the main pass's triggered tiles from a third to a tenth, but
// the sample pass (filter + exact refine + threshold: three dependent launches) grows by more than that saves.
static int64_t topk_sample_rows(int64_t n_q, int64_t n_db) {
    static const int64_t v = getenv("ASR_TOPK_SAMPLE") ? atoll(getenv("ASR_TOPK_SAMPLE")) : 0;
    (void)n_q;
    // ... and for a pool of 250 k rows 16 384 are 6.5 % of it - a third of the call's time: 1/32 of the pool, 4096 to 16 384
    // (1024 queries x 250 k codes: 0.53 ms with 16 384 rows, 0.49 with 8192, 0.51 with 4096)
    // ... and 1/64 of a pool beyond 1 M rows, up to 32 768 (round 5, sample keys + select instead of the three-launch
    // form; 16 384 / 32 768 rows: 64 x 2 M 0.137 / 0.132 ms, 512 x 2 M 0.686 / 0.656, 16 x 2 M 0.092 / 0.092 - the filter's
    // survivors halve; 256 x 1 M 0.224 / 0.231, 1024 x 1 M 0.712 / 0.734: a 1 M pool keeps 16 384)
    // ... and with the sample keys on the matrix cores (sample_keys_mfma_kernel) the sample costs a third of what it
    // did: 1/16 of the pool, up to 32 768 rows (half / full: 256 x 1 M 0.191 / 0.182 ms, 1024 x 1 M 0.588 / 0.560,
    // 1024 x 250 k 0.230 / 0.221, 128 x 250 k 0.085 / 0.079, 4096 x 250 k 0.841 / 0.763, 512 x 500 k 0.218 / 0.198,
    // 2000 x 100 k 0.291 / 0.251; 1024 x 65 k 0.119 / 0.120 keeps 4096)
    int64_t rows = v > 0 ? std::max<int64_t>(4096, v & ~(int64_t)4095)
                         : std::min<int64_t>(n_db >= 1000000 ? SS_ROWS_MAX : 16384,      // (512 x 500 k: 16 384 rows 0.199 ms, 32 768 0.204)
                                             std::max<int64_t>(4096, (n_db / 16 + 2048) & ~(int64_t)4095));
    while (rows > 4096 && n_db < 8 * rows) rows >>= 1;
    return rows;
}
static bool topk_seeded(int64_t n_q, int64_t n_db, bool unit) {
    static const int on = getenv("ASR_TOPK_SEED") ? atoi(getenv("ASR_TOPK_SEED")) : 1;
    return on && unit && n_db >= 8 * topk_sample_rows(n_q, n_db);
}

// slices of the single-pass scan (topk_scan_kernel): >= 1024 rows each, 16 .. 512 of them (<= SCAN_ROWS rows at 512)
static int scan_chunks(int64_t n_db) {
    int chunks = 512;
    while (chunks > 16 && n_db < (int64_t)chunks * 1024) chunks >>= 1;
    return chunks;
}

// Layout of the scratch buffer of one top-k (+ fused ranking) call
struct TopkPlan {
    int qg, S, chunks;
    bool seeded;
    int64_t sample_rows;            // seeding pass: rows of the strided sample, in slices of 1024
    int sample_slices;
    size_t off_rn_db, off_rn_q, off_cnt, off_idx, off_pidx, off_pdist, off_ds, off_js, off_counts, off_thr0, off_scnt,
        off_sidx, off_soidx, off_sodist, off_skeys, off_scan_idx, off_scan_dist, off_gkeys, bytes;
};

static TopkPlan plan_topk(int64_t n_db, int64_t n_q, int k, bool unit, bool fuse_rank) {
    TopkPlan P{};
    P.seeded = topk_seeded(n_q, n_db, unit);
    P.sample_rows = topk_sample_rows(n_q, n_db);
    P.sample_slices = (int)(P.sample_rows / 1024);
    P.qg = topk_query_groups(n_q, n_db, k, unit, P.seeded);
    const int64_t groups = (n_q + 16 * P.qg - 1) / (16 * P.qg);
    // slices per query group: enough workgroups to fill the chip also when there are few queries (64 queries against a
    // 2 M-code pool are 4 groups - with at most 16 slices that was 64 workgroups on 256 CUs, 140 GB/s of a stream
    // that should run at the HBM rate); every slice hands <= TF_OUT survivors per query to the exact kernel
    // measured on the MI355X (256 CUs, two resident filter workgroups each): 1024 queries x 250 k codes / 64 x 2 M take
    // 0.87 / 0.65 ms at 256 workgroups, 0.66 / 0.60 at 512, 0.90 / 0.83 at 640 (a second, nearly empty round), 0.81 / 0.82
    // at 1024; the two-group form (4096 x 2 M) 12.2 ms at 384, 9.7 at 1024
    static const int wgs_env = getenv("ASR_TOPK_WGS") ? atoi(getenv("ASR_TOPK_WGS")) : 0;
    const int target_wgs = wgs_env ? wgs_env : (P.qg == 4 ? 1024 : 512 * P.qg);
    static const int max_slices = getenv("ASR_TOPK_SLICES") ? atoi(getenv("ASR_TOPK_SLICES")) : 1024;
    int S = (int)std::max<int64_t>(1, std::min<int64_t>(max_slices, (target_wgs + groups - 1) / groups));
    // the groups of a slice run side by side on one XCD and share the slice in its 4 MB L2 (topk_filter_kernel's block
    // mapping): with many groups a slice is at most ~1 MB (8192 rows), whatever the workgroup count that gives
    // (measured, 4096 queries x 2^21 codes fused with the ranking: 8192-row slices 9.2 ms, 131072-row slices 8.0 - a
    // slice that long tightens its own thresholds far below the seed, and the groups still walk it together)
    static const int slice_items = getenv("ASR_TOPK_SLICE_ITEMS") ? atoi(getenv("ASR_TOPK_SLICE_ITEMS")) : 131072;
    if (unit && groups >= 8) S = (int)std::max<int64_t>(S, std::min<int64_t>(max_slices, (n_db + slice_items - 1) / slice_items));
    // a slice should hold >= 4096 items - 2048 when many four-group workgroups share it (measured, 4096 / 2048:
    // 1024 x 250 k 0.289 / 0.274 ms, 2000 x 100 k 0.374 / 0.329 - 1024 workgroups instead of 896 / 768: two full rounds;
    // 512 x 250 k 0.160 / 0.165, 1024 x 65 k (two groups) 0.139 / 0.146, 64 x 2 M 0.134 / 0.143; larger shapes unchanged)
    static const int slice_min_env = getenv("ASR_TOPK_SLICE_MIN") ? std::max(256, atoi(getenv("ASR_TOPK_SLICE_MIN"))) : 0;
    const int slice_min = slice_min_env ? slice_min_env : (P.qg == 4 && groups >= 16 ? 2048 : 4096);
    S = (int)std::min<int64_t>(S, std::max<int64_t>(1, n_db / slice_min));
    if (S >= 8) S &= ~7;                                                     // one eighth of the slices per XCD
    P.S = S;
    // exact refine: one workgroup per query walks all S lists - with few queries that leaves most of the chip idle, so
    // the lists are cut into chunks (one workgroup each, partial top-k lists merged by topk_merge_kernel)
    static const int chunks_env = getenv("ASR_TOPK_CHUNKS") ? atoi(getenv("ASR_TOPK_CHUNKS")) : 0;
    int chunks = n_q >= 512 ? 1 : (int)std::min<int64_t>((1024 + n_q - 1) / n_q, S);
    if (chunks_env > 0) chunks = std::min(chunks_env, S);
    chunks = std::max(1, std::min(chunks, TOPK_SORT / std::max(1, k)));
    P.chunks = chunks;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    P.off_rn_db = o; o = al(o + (unit ? 0 : (size_t)((n_db + 3) & ~(int64_t)3) * sizeof(float)));
    P.off_rn_q = o; o = al(o + (size_t)n_q * sizeof(float));
    P.off_cnt = o; o = al(o + (size_t)n_q * S * sizeof(int32_t));
    P.off_idx = o; o = al(o + (size_t)n_q * S * TF_OUT * sizeof(int32_t));
    P.off_pidx = o; o = al(o + (chunks > 1 ? (size_t)n_q * chunks * k * sizeof(int32_t) : 0));
    P.off_pdist = o; o = al(o + (chunks > 1 ? (size_t)n_q * chunks * k * sizeof(double) : 0));
    P.off_ds = o; o = al(o + (fuse_rank ? (size_t)n_q * sizeof(double) : 0));
    P.off_js = o; o = al(o + (fuse_rank ? (size_t)n_q * sizeof(int64_t) : 0));
    P.off_counts = o; o = al(o + (fuse_rank ? (size_t)n_q * 3 * sizeof(int32_t) : 0));
    P.off_thr0 = o; o = al(o + (P.seeded ? (size_t)n_q * sizeof(float) : 0));
    P.off_scnt = o; o = al(o + (P.seeded ? (size_t)n_q * P.sample_slices * sizeof(int32_t) : 0));
    P.off_sidx = o; o = al(o + (P.seeded ? (size_t)n_q * P.sample_slices * TF_OUT * sizeof(int32_t) : 0));
    P.off_soidx = o; o = al(o + (P.seeded ? (size_t)n_q * k * sizeof(int32_t) : 0));
    P.off_sodist = o; o = al(o + (P.seeded ? (size_t)n_q * k * sizeof(double) : 0));
    P.off_skeys = o; o = al(o + (P.seeded ? (size_t)n_q * P.sample_rows * sizeof(uint16_t) : 0));
    // the few-queries scan (topk_scan_kernel): per-slice lists [q][slices][k]
    {
        const size_t lists = (unit && n_q <= SCAN_NQ_MAX && n_db <= (int64_t)512 * SCAN_ROWS)
                                 ? (size_t)n_q * (size_t)scan_chunks(n_db) * (size_t)k : 0;
        P.off_scan_idx = o; o = al(o + lists * sizeof(int32_t));
        P.off_scan_dist = o; o = al(o + lists * sizeof(double));
    }
    // the sort-free exact refine (topk_collect_kernel / topk_select_kernel): tsel_cap(n_q) exact keys per query
    P.off_gkeys = o; o = al(o + ((n_q <= TSEL_NQ_MAX && k <= 32) ? (size_t)n_q * tsel_cap(n_q) * sizeof(TopkKey) : 0));
    P.bytes = o;
    return P;
}

size_t topk_workspace_bytes(int64_t n_db, int64_t n_q, int k, bool unit, bool fuse_rank) {
    return plan_topk(n_db, n_q, std::max(1, std::min(k, TOPK_KMAX)), unit, fuse_rank).bytes;
}

// thresholds from a strided sample of the pool (see topk_filter_kernel): filter + exact refine on 16384 virtual rows
// ASR_TOPK_SEED=2: the three-launch form (filter, exact refine, threshold) for A/B runs
static bool seed_in_two_launches(const TopkPlan &P, int k) {
    static const int three = getenv("ASR_TOPK_SEED") && atoi(getenv("ASR_TOPK_SEED")) == 2;
    return P.seeded && !three && P.sample_rows <= SS_ROWS_MAX && k <= P.sample_rows;
}

static void seed_thresholds(hipStream_t s, const TopkPlan &P, char *ws, const float *unit, const float *db,
                            const double *norm_db, int64_t n_db, const float *q, const double *norm_q, const float *rn_q,
                            int64_t n_q, int k, float *thr0, double *norm_q_out = nullptr, float *rn_q_out = nullptr,
                            unsigned *tickets = nullptr) {
    const int64_t rows = P.sample_rows, stride = n_db / rows;
    const int sl = P.sample_slices;
    if (seed_in_two_launches(P, k)) {
        uint16_t *keys = (uint16_t *)(ws + P.off_skeys);
        static const bool mfma_keys = !(getenv("ASR_TOPK_SAMPLE_MFMA") && getenv("ASR_TOPK_SAMPLE_MFMA")[0] == '0');
        unsigned *tk_fold = tickets ? tickets + TOPK_TICKETS : nullptr;
        if (mfma_keys && !(tk_fold && n_q < 256)) {                  // (the last-arriver experiment keeps the vector form)
            const unsigned rb = (unsigned)(rows / SKM_ROWS);
            if (n_q <= 16)
                sample_keys_mfma_kernel<1><<<dim3(rb, 1), 256, 0, s>>>(unit, rows, stride, q, norm_q, n_q, keys, norm_q_out, rn_q_out);
            else if (n_q <= 32)
                sample_keys_mfma_kernel<2><<<dim3(rb, 1), 256, 0, s>>>(unit, rows, stride, q, norm_q, n_q, keys, norm_q_out, rn_q_out);
            else
                sample_keys_mfma_kernel<4><<<dim3(rb, (unsigned)((n_q + 63) / 64)), 256, 0, s>>>(unit, rows, stride, q, norm_q, n_q, keys,
                                                                                              norm_q_out, rn_q_out);
            sample_select_kernel<<<(unsigned)n_q, 256, 0, s>>>(keys, rows, k, thr0);
            return;
        }
        static const int64_t qb4_from = getenv("ASR_TOPK_SAMPLE_QB4") ? atoll(getenv("ASR_TOPK_SAMPLE_QB4")) : 256;
        const bool qb4 = n_q >= qb4_from;
        const dim3 grid((unsigned)(rows / 256), (unsigned)((n_q + (qb4 ? 3 : 0)) / (qb4 ? 4 : 1)));
        // (round 5, measured and dropped: 16 queries per workgroup instead of 4 - the sample is read 4x less often, the
        // call got slower: 1024 x 250 k 0.316 -> 0.325 ms, 512 x 250 k 0.187 -> 0.200)
        if (qb4) {
            sample_keys_kernel<4><<<grid, 256, 0, s>>>(unit, rows, stride, q, norm_q, n_q, keys, norm_q_out, rn_q_out, nullptr, k, thr0);
            sample_select_kernel<<<(unsigned)n_q, 256, 0, s>>>(keys, rows, k, thr0);
        } else {
            // few queries: the last row block of each query selects its threshold (one launch instead of two)
            unsigned *tk = tickets ? tickets + TOPK_TICKETS : nullptr;
            sample_keys_kernel<1><<<grid, 256, 0, s>>>(unit, rows, stride, q, norm_q, n_q, keys, norm_q_out, rn_q_out, tk, k, thr0);
            if (!tk) sample_select_kernel<<<(unsigned)n_q, 256, 0, s>>>(keys, rows, k, thr0);
        }
        return;
    }
    int32_t *scnt = (int32_t *)(ws + P.off_scnt), *sidx = (int32_t *)(ws + P.off_sidx);
    int32_t *oidx = (int32_t *)(ws + P.off_soidx);
    double *odist = (double *)(ws + P.off_sodist);
    const unsigned grid = (unsigned)((n_q + 15) / 16) * sl;
    RankFuse none{};
    if (k > 32)
        topk_filter_kernel<512, 1, true, false><<<grid, TF_THREADS, 0, s>>>(unit, nullptr, rows, q, rn_q, n_q, k, sl, sidx, scnt, none,
                                                                           stride, nullptr);
    else
        topk_filter_kernel<256, 1, true, false><<<grid, TF_THREADS, 0, s>>>(unit, nullptr, rows, q, rn_q, n_q, k, sl, sidx, scnt, none,
                                                                           stride, nullptr);
    topk_kernel<<<dim3((unsigned)n_q, 1), TOPK_THREADS, 0, s>>>(db, norm_db, rows, 32, q, norm_q, 32, 32, k, 0, oidx, odist, sidx, scnt,
                                                                sl, TF_OUT, nullptr, nullptr, stride, nullptr);
    seed_threshold_kernel<<<(unsigned)((n_q + 255) / 256), 256, 0, s>>>(oidx, odist, n_q, k, thr0);
}

// ---- exact refine without per-chunk sorts (round 5; k <= 32, <= 1024 queries) ------------------------------------------
// topk_kernel orders every chunk's survivors with a bitonic network (36-45 barrier rounds per workgroup) and
// topk_merge_kernel orders the chunks' lists again: 32 + 13 us of the 0.151 ms of a 64-query call.  Here the chunks only
// EVALUATE: workgroup (q, c) computes the exact float64 keys of its lists' survivors (the same arithmetic) and appends
// them, unordered, to the query's global key array (one atomic per workgroup reserves the range); topk_select_kernel
// then finds each query's k smallest among its ~k n_db / sample keys the way topk_merge_heads_kernel does: the k-th
// smallest key to within one 16-bit distance bin by a two-pass radix select, the keys below that bin's edge (k .. 2k of
// them) ordered by one small network.
// A query with an overflowed candidate list, with more survivors than its key array holds, or with more than 1024 keys in
// that last bin (masses of near-ties) is scanned exactly (flag in qbad) -
// topk_kernel's scan mode, launched behind and skipping every other query.
__global__ __launch_bounds__(TOPK_THREADS) void topk_collect_kernel(
    const float *__restrict__ db, const double *__restrict__ norm_db, int64_t ld_db, const float *__restrict__ qs,
    const double *__restrict__ norm_q, int64_t ld_q, int dim, int64_t idx_offset, const int32_t *__restrict__ cand_idx,
    const int32_t *__restrict__ cand_cnt, int n_lists, int list_cap, TopkKey *__restrict__ gkeys, int *__restrict__ gcount,
    int *__restrict__ qbad, int cap) {
    __shared__ float q[RANK_MAXD];
    __shared__ int pre[TOPK_THREADS + 1];
    __shared__ int s_base, s_bad;
    const int tid = threadIdx.x;
    const int64_t qi = blockIdx.x;
    const int chunk = blockIdx.y, n_chunks = gridDim.y;
    for (int c = tid; c < dim; c += TOPK_THREADS) q[c] = qs[qi * ld_q + c];
    const int per = (n_lists + n_chunks - 1) / n_chunks;           // (<= TOPK_THREADS: the launcher sees to it)
    const int l0 = chunk * per, l1 = l0 + per < n_lists ? l0 + per : n_lists;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    int c_mine = 0;
    if (l0 + tid < l1) {
        c_mine = cand_cnt[qi * n_lists + l0 + tid];
        if (c_mine < 0) { c_mine = 0; s_bad = 1; }
    }
    pre[tid + 1] = c_mine;
    if (tid == 0) pre[0] = 0;
    __syncthreads();
    if (tid < 64) {                                                // inclusive scan of the <= 256 counts: four per lane
        const int lane = tid;
        const int v0 = pre[4 * lane + 1], v1 = pre[4 * lane + 2], v2 = pre[4 * lane + 3], v3 = pre[4 * lane + 4];
        const int tot = v0 + v1 + v2 + v3;
        int incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int excl = incl - tot;
        pre[4 * lane + 1] = excl + v0; pre[4 * lane + 2] = excl + v0 + v1;
        pre[4 * lane + 3] = excl + v0 + v1 + v2; pre[4 * lane + 4] = incl;
    }
    __syncthreads();
    if (s_bad) {                                                   // (the whole query is scanned exactly)
        if (tid == 0) atomicExch(&qbad[qi], 1);
        return;
    }
    const int nl = l1 > l0 ? l1 - l0 : 0;
    const int total = pre[nl];
    // the LAST thread reserves the range (it rarely has an entry of its own): the atomic's round trip runs beside the
    // others' loads and arithmetic - the range's base is needed only for the store
    if (tid == TOPK_THREADS - 1) {
        int base = 0;
        if (total > 0) {
            base = atomicAdd(&gcount[qi], total);
            if (base + total > cap) atomicExch(&qbad[qi], 1);
        }
        s_base = base;
    }
    const double nq = norm_q[qi];
    // entry t of this chunk: list = the l with pre[l] <= t < pre[l + 1] (binary search over <= TOPK_THREADS + 1 offsets)
    auto entry = [&](int t) {
        int lo = 0, hi = nl;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= t) lo = mid; else hi = mid;
        }
        const int64_t j = cand_idx[(qi * n_lists + l0 + lo) * list_cap + (t - pre[lo])];
        const double d = cos_dist(dot2acc(q, db + j * ld_db, dim), nq, norm_db[j]);
        TopkKey kk;
        kk.d = (unsigned long long)__double_as_longlong(d + 0.0);     // +0.0: never the -0.0 pattern
        kk.j = j + idx_offset;
        return kk;
    };
    TopkKey first = {0, 0};
    if (tid < total) first = entry(tid);
    __syncthreads();
    const int base = s_base;
    if (base + total > cap) return;
    TopkKey *dst = gkeys + qi * cap + base;
    if (tid < total) dst[tid] = first;
    for (int t = tid + TOPK_THREADS; t < total; t += TOPK_THREADS) dst[t] = entry(t);
}

__global__ __launch_bounds__(TSEL_THREADS) void topk_select_kernel(
    const TopkKey *__restrict__ gkeys, int *__restrict__ gcount, int *__restrict__ qbad, int k, int64_t n_db_full,
    int32_t *__restrict__ idx_out, double *__restrict__ dist_out, int cap, const float *__restrict__ db,
    const double *__restrict__ norm_db, int64_t ld_db, const float *__restrict__ qs, const double *__restrict__ norm_q,
    int64_t ld_q, int dim, int64_t idx_offset) {
    // one block of LDS, two uses: [0, 1024) the gathered keys, behind them the 16-bit keys of all survivors (select) or
    // the TOPK_SORT keys of topk_scan_rows (fallback)
    __shared__ TopkKey pool[1024 + TOPK_SORT];
    static_assert(TSEL_CAP * sizeof(uint16_t) <= TOPK_SORT * sizeof(TopkKey), "hkey fits behind the gathered keys");
    __shared__ int hist[256];
    __shared__ int sel[2];
    __shared__ int nkeys;
    TopkKey *keys = pool;
    uint16_t *hkey = reinterpret_cast<uint16_t *>(pool + 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t qi = blockIdx.x;
    const TopkKey inf = {0x7ff0000000000000ULL, 0x7fffffffffffffffLL};
    const int n = gcount[qi];
    const int flagged = qbad[qi];                                  // (set by topk_collect_kernel: overflowed list / key array)
    __syncthreads();
    if (tid == 0) { gcount[qi] = 0; qbad[qi] = 0; }                // ready for the next call (stream order)
    const TopkKey *src = gkeys + qi * cap;
    auto sort_keys = [&](int sort_n) {
        for (int size = 2; size <= sort_n; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int e = tid; e < sort_n / 2; e += TSEL_THREADS) {
                    const int lo = 2 * e - (e & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const TopkKey x = keys[lo], y = keys[hi];
                    if (key_less(y, x) == up) { keys[lo] = y; keys[hi] = x; }
                }
                __syncthreads();
            }
    };
    auto write_out = [&](const TopkKey *from, int filled) {
        for (int e = tid; e < k; e += TSEL_THREADS) {
            const TopkKey kk = e < filled ? from[e] : inf;
            const bool valid = e < n_db_full && kk.j != inf.j;
            idx_out[qi * k + e] = valid ? (int32_t)kk.j : -1;
            dist_out[qi * k + e] = valid ? __longlong_as_double((long long)kk.d) : __longlong_as_double(0x7ff0000000000000LL);
        }
    };
    if (!flagged) {                                                // block-uniform
        if (tid == 0) nkeys = 0;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        // (1024 threads: the ~3000 keys of a query are three loads per thread in flight, not twelve dependent round trips)
        for (int e = tid; e < n; e += TSEL_THREADS) {
            const double d = __longlong_as_double((long long)src[e].d);
            const unsigned u = d == d ? (unsigned)fmin(d * 32768.0, 65534.0) : 65535u;
            hkey[e] = (uint16_t)u;
            atomicAdd(&hist[u >> 8], 1);                           // the first histogram pass rides on the load
        }
        __syncthreads();
        unsigned kb = 65535u;
        if (n > k) {
            unsigned prefix = 0;
            int rank = k;
            for (int pass = 1; pass >= 0; --pass) {
                if (pass == 0) {
                    if (tid < 256) hist[tid] = 0;
                    __syncthreads();
                    for (int e = tid; e < n; e += TSEL_THREADS) {
                        const unsigned u = hkey[e];
                        if ((u >> 8) == prefix) atomicAdd(&hist[u & 255u], 1);
                    }
                    __syncthreads();
                }
                if (tid < 64) {
                    const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
                    const int tot = h0 + h1 + h2 + h3;
                    int incl = tot;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const int v = __shfl_up(incl, o);
                        if (lane >= o) incl += v;
                    }
                    const int excl = incl - tot;
                    if (excl < rank && rank <= incl) {
                        int bin = 4 * lane, c = excl;
                        if (rank > c + h0) { c += h0; ++bin; if (rank > c + h1) { c += h1; ++bin; if (rank > c + h2) { c += h2; ++bin; } } }
                        sel[0] = bin; sel[1] = rank - c;
                    }
                }
                __syncthreads();
                if (pass == 1) prefix = (unsigned)sel[0];
                else prefix = (prefix << 8) | (unsigned)sel[0];
                rank = sel[1];
                __syncthreads();
            }
            kb = prefix;
        }
        // the keys at or below the bin of the k-th: a superset of the k smallest (the k-th key itself lies in that bin)
        for (int e = tid; e < n; e += TSEL_THREADS)
            if ((unsigned)hkey[e] <= kb) {
                const int pos = atomicAdd(&nkeys, 1);
                if (pos < 1024) keys[pos] = src[e];
            }
        __syncthreads();
        if (nkeys <= 1024) {
            int sort_n = 32;
            while (sort_n < nkeys) sort_n <<= 1;
            for (int e = nkeys + tid; e < sort_n; e += TSEL_THREADS) keys[e] = inf;
            __syncthreads();
            sort_keys(sort_n);
            write_out(keys, sort_n);
            return;
        }
        __syncthreads();                                           // (everybody has read nkeys and hkey)
    }
    // The exact scan of the whole pool for this query: an overflowed candidate list, more survivors than its key array
    // holds, or more than 1024 keys at or below the bin of the k-th (masses of near-ties).  topk_kernel's arithmetic and
    // selection (topk_scan_rows), in this workgroup - a launch of its own behind this one cost 4 us per call to skip.
    __shared__ float qv[RANK_MAXD];
    __shared__ int ncand;
    __shared__ TopkKey thr;
    TopkKey *skeys = pool + 1024;
    for (int c = tid; c < dim; c += TSEL_THREADS) qv[c] = qs[qi * ld_q + c];
    for (int e = tid; e < TOPK_SORT; e += TSEL_THREADS) skeys[e] = inf;
    if (tid == 0) { ncand = 0; thr = inf; }
    __syncthreads();
    topk_scan_rows<TSEL_THREADS, 1, TOPK_SORT - TOPK_KMAX>(db, norm_db, ld_db, 1, qv, norm_q[qi], dim, idx_offset, k, n_db_full, false, qi,
                                                           0, 0, 0, 0, nullptr, nullptr, skeys, &ncand, &thr);
    write_out(skeys, k);
}

template <bool NORM, bool RANK>
static void launch_filter(hipStream_t s, const TopkPlan &P, const float *rows, const float *rn_db, int64_t n_db,
                          const float *q, const float *rn_q, int64_t n_q, int k, int32_t *cand_idx, int32_t *cand_cnt,
                          const RankFuse &R, const float *seed) {
    const int64_t groups = (n_q + 16 * P.qg - 1) / (16 * P.qg);
    const unsigned grid = (unsigned)(groups * P.S);
    static const int sm_env = getenv("ASR_TOPK_SLICE_MAJOR") ? atoi(getenv("ASR_TOPK_SLICE_MAJOR")) : -1;
    const int64_t rs = (sm_env == 0 || (sm_env < 0 && !NORM)) ? -1 : 1;      // (negative: group-major block order)
    if (k > 32)
        topk_filter_kernel<512, 1, NORM, RANK><<<grid, TF_THREADS, 0, s>>>(rows, rn_db, n_db, q, rn_q, n_q, k, P.S, cand_idx, cand_cnt, R, rs, seed);
    else if (P.qg == 2)
        topk_filter_kernel<256, 2, NORM, RANK><<<grid, TF_THREADS, 0, s>>>(rows, rn_db, n_db, q, rn_q, n_q, k, P.S, cand_idx, cand_cnt, R, rs, seed);
    else if (P.qg == 4) {
        if constexpr (NORM)         // (only chosen with seeded thresholds: the small buffers would overflow in a +inf warm-up round)
            topk_filter_kernel<128, 4, true, RANK><<<grid, TF_THREADS, 0, s>>>(rows, rn_db, n_db, q, rn_q, n_q, k, P.S, cand_idx, cand_cnt, R, rs, seed);
    } else
        topk_filter_kernel<256, 1, NORM, RANK><<<grid, TF_THREADS, 0, s>>>(rows, rn_db, n_db, q, rn_q, n_q, k, P.S, cand_idx, cand_cnt, R, rs, seed);
}

static void launch_refine(hipStream_t s, const TopkPlan &P, char *ws, const float *db, const double *norm_db, int64_t n_db,
                          int64_t ld_db, const float *q, const double *norm_q, int64_t n_q, int64_t ld_q, int dim, int k,
                          int64_t idx_offset, int32_t *idx_out, double *dist_out, unsigned *tickets = nullptr,
                          int *state = nullptr) {
    int32_t *cand_cnt = (int32_t *)(ws + P.off_cnt), *cand_idx = (int32_t *)(ws + P.off_idx);
    // state (the context's zero-between-calls block, may be null): [0, 1024) survivor counts, [1024, 2048) "scan this query
    // exactly" flags of the sort-free refine.  ASR_TOPK_SELECT=0: topk_kernel + topk_merge_kernel as in round 4.
    static const bool use_select = !(getenv("ASR_TOPK_SELECT") && getenv("ASR_TOPK_SELECT")[0] == '0');
    // (without seeded thresholds a slice hands over up to TF_OUT / 2 uncompacted survivors: only while they fit the key array)
    if (state && use_select && k <= 32 && n_q <= TSEL_NQ_MAX && (P.seeded || (int64_t)P.S * (TF_OUT / 2) <= tsel_cap(n_q))) {
        int *gcount = state, *qbad = state + TSEL_NQ_MAX;
        TopkKey *gkeys = (TopkKey *)(ws + P.off_gkeys);
        int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(P.S, (1024 + n_q - 1) / n_q));
        chunks = std::max(chunks, (P.S + TOPK_THREADS - 1) / TOPK_THREADS);          // <= 256 lists per workgroup
        topk_collect_kernel<<<dim3((unsigned)n_q, (unsigned)chunks), TOPK_THREADS, 0, s>>>(
            db, norm_db, ld_db, q, norm_q, ld_q, dim, idx_offset, cand_idx, cand_cnt, P.S, TF_OUT, gkeys, gcount, qbad, tsel_cap(n_q));
        topk_select_kernel<<<(unsigned)n_q, TSEL_THREADS, 0, s>>>(gkeys, gcount, qbad, k, n_db, idx_out, dist_out, tsel_cap(n_q),
                                                                  db, norm_db, ld_db, q, norm_q, ld_q, dim, idx_offset);
        return;
    }
    int32_t *pidx = (int32_t *)(ws + P.off_pidx);
    double *pdist = (double *)(ws + P.off_pdist);
    topk_kernel<<<dim3((unsigned)n_q, (unsigned)P.chunks), TOPK_THREADS, 0, s>>>(
        db, norm_db, n_db, ld_db, q, norm_q, ld_q, dim, k, idx_offset, idx_out, dist_out, cand_idx, cand_cnt, P.S, TF_OUT,
        pidx, pdist, 1, (P.chunks > 1 && n_q <= TOPK_TICKETS) ? tickets : nullptr);
    if (P.chunks > 1 && !(tickets && n_q <= TOPK_TICKETS))
        topk_merge_kernel<<<(unsigned)n_q, TOPK_THREADS, 0, s>>>(pidx, pdist, P.chunks, k, n_db, idx_out, dist_out, (int64_t)P.chunks * k, k, 0);
}

// unit / rn_db_pre (may be null): the resident data base's unit-length rows and reciprocal norms (launch_db_prepare)
hipError_t launch_topk(hipStream_t s, const float *db, const double *norm_db, int64_t n_db, int64_t ld_db,
                       const float *q, const double *norm_q, int64_t n_q, int64_t ld_q, int dim, int k,
                       int64_t idx_offset, int32_t *idx_out, double *dist_out, void *workspace, const float *unit,
                       const float *rn_db_pre, double *norm_q_pending, unsigned *tickets) {
    // ASR_TOPK_FOLD: 0 = separate launches, 2 = only the merge folded, 3 = only the threshold select folded (A/B runs)
    // Default 0.  Measured, 64 queries x 2 M codes: separate launches 0.151 ms; merge folded 0.204; threshold select folded
    // 0.440; both 0.50.  A last arriver needs every workgroup to publish its writes with an agent-scope fence first, and on
    // this chip (eight XCDs, each with its own non-coherent L2) that fence is an L2 write-back + invalidate: ~50-70 ns
    // apiece, serialised - 4096 workgroups of the key pass pay 0.29 ms for the 5 us launch they save.  (The training
    // step's reductions, <= 128 workgroups per launch, do not notice it.)
    static const int fold = getenv("ASR_TOPK_FOLD") ? atoi(getenv("ASR_TOPK_FOLD")) : 0;
    int *state = tickets ? reinterpret_cast<int *>(tickets) + 2 * TOPK_TICKETS : nullptr;      // behind the 1024 fold tickets
    if (!fold) tickets = nullptr;
    unsigned *tk_seed = (fold == 2) ? nullptr : tickets, *tk_ref = (fold == 3) ? nullptr : tickets;
    if (n_q == 0) return hipSuccess;
    if (dim > RANK_MAXD || k < 1 || k > TOPK_KMAX) return hipErrorInvalidValue;
    if (norm_q_pending && norm_q_pending != norm_q) return hipErrorInvalidValue;
    static const int use_filter = getenv("ASR_TOPK_FILTER") ? atoi(getenv("ASR_TOPK_FILTER")) : 1;
    // the MFMA filter needs 32-d packed rows and a data base large enough to amortise it
    if (!use_filter || !workspace || dim != 32 || ld_db != 32 || ld_q != 32 || n_db < 16384) {
        if (norm_q_pending) {
            hipError_t e = launch_row_norms(s, q, n_q, ld_q, dim, norm_q_pending);
            if (e != hipSuccess) return e;
        }
        topk_kernel<<<(unsigned)n_q, TOPK_THREADS, 0, s>>>(db, norm_db, n_db, ld_db, q, norm_q, ld_q, dim, k, idx_offset,
                                                           idx_out, dist_out, nullptr, nullptr, 0, 0, nullptr, nullptr, 1, nullptr);
        return hipGetLastError();
    }
    const TopkPlan P = plan_topk(n_db, n_q, k, unit != nullptr, false);
    char *ws = (char *)workspace;
    // One query (ASR_TOPK_SCAN=<n>: up to n <= 4; 0: never) against a resident pool: one streaming pass + a merge tree
    // instead of sample / filter / refine / merge (topk_scan_kernel)
    // Every query streams the pool by itself there, so it pays for one or two queries against any pool and for many
    // against a pool that stays in the caches.  Measured, top-25, scan / general path (ms): 1 x 2 M 0.0625 / 0.118, 2 x 2 M
    // 0.065 / 0.120, 4 x 2 M 0.120 / 0.123, 8 x 2 M 0.268 / 0.126; 100 x 20 k 0.053 / 0.153, 100 x 50 k 0.095 / 0.118 (the
    // server's detect_score: 100 windows against a data base of tens of thousands of codes), 64 x 100 k 0.101 / 0.128,
    // 32 x 250 k 0.110 / 0.106, 16 x 500 k 0.115 / 0.115, 8 x 1 M 0.148 / 0.114, 128 x 100 k 0.183 / 0.096: taken for <= 2
    // queries, and for <= 128 when queries x rows <= 6.5e6.  ASR_TOPK_SCAN=0: never; =n: whenever there are <= n queries.
    static const int scan_env = getenv("ASR_TOPK_SCAN") ? std::max(0, std::min(SCAN_NQ_MAX, atoi(getenv("ASR_TOPK_SCAN")))) : -1;
    const bool scan_fits = scan_env >= 0 ? n_q <= scan_env
                                         : (n_q <= 2 || (n_q <= SCAN_NQ_MAX && (double)n_q * (double)n_db <= 6.5e6));
    if (unit && scan_fits && n_db <= (int64_t)512 * SCAN_ROWS && (reinterpret_cast<uintptr_t>(q) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(db) & 15) == 0) {
        const int chunks = scan_chunks(n_db);
        int32_t *bi = (int32_t *)(ws + P.off_scan_idx);
        double *bd = (double *)(ws + P.off_scan_dist);
        topk_scan_kernel<<<dim3((unsigned)chunks, (unsigned)n_q), 256, 0, s>>>(unit, db, norm_db, n_db, q, norm_q, norm_q_pending, k,
                                                                              idx_offset, bi, bd);
        topk_merge_heads_kernel<<<(unsigned)n_q, TOPK_THREADS, 0, s>>>(bi, bd, chunks, k, n_db, idx_out, dist_out);
        return hipGetLastError();
    }
    float *rn_q = (float *)(ws + P.off_rn_q);
    int32_t *cand_cnt = (int32_t *)(ws + P.off_cnt), *cand_idx = (int32_t *)(ws + P.off_idx);
    // pending query norms: the seeding kernel forms and stores them (and their fp32 reciprocals) where it runs
    const bool in_seed = norm_q_pending && unit && seed_in_two_launches(P, k) && (reinterpret_cast<uintptr_t>(q) & 15) == 0;
    if (norm_q_pending && !in_seed) {
        hipError_t e = launch_row_norms(s, q, n_q, ld_q, dim, norm_q_pending);
        if (e != hipSuccess) return e;
    }
    if (!in_seed) rnorm_f32_kernel<<<(unsigned)((n_q + 255) / 256), 256, 0, s>>>(norm_q, n_q, rn_q);
    RankFuse none{};
    if (unit) {
        float *thr0 = nullptr;
        if (P.seeded) {
            thr0 = (float *)(ws + P.off_thr0);
            seed_thresholds(s, P, ws, unit, db, norm_db, n_db, q, norm_q, rn_q, n_q, k, thr0, in_seed ? norm_q_pending : nullptr,
                            in_seed ? rn_q : nullptr, tk_seed);
        }
        launch_filter<true, false>(s, P, unit, nullptr, n_db, q, rn_q, n_q, k, cand_idx, cand_cnt, none, thr0);
    } else {
        const float *rn_db = rn_db_pre;
        if (!rn_db) {
            const int64_t n_db_pad = (n_db + 3) & ~(int64_t)3;
            float *w = (float *)(ws + P.off_rn_db);
            if (n_db_pad > n_db) (void)hipMemsetAsync(w + n_db, 0, (size_t)(n_db_pad - n_db) * sizeof(float), s);
            rnorm_f32_kernel<<<(unsigned)((n_db + 255) / 256), 256, 0, s>>>(norm_db, n_db, w);
            rn_db = w;
        }
        launch_filter<false, false>(s, P, db, rn_db, n_db, q, rn_q, n_q, k, cand_idx, cand_cnt, none, nullptr);
    }
    launch_refine(s, P, ws, db, norm_db, n_db, ld_db, q, norm_q, n_q, ld_q, dim, k, idx_offset, idx_out, dist_out, tk_ref, state);
    return hipGetLastError();
}

// Top-k AND eval_retrieval ranks of the same queries against a resident data base in ONE walk over the pool (32-d packed
// rows, n_db >= 16384, kk <= 8192 - the caller falls back to launch_topk + launch_rank otherwise).  kk, hh: correct
// candidates per query / queries per candidate block (utils/train_dcca_pool.py:35-36).
bool topk_rank_fusable(int64_t n_db, int64_t kk) {
    static const int use_filter = getenv("ASR_TOPK_FILTER") ? atoi(getenv("ASR_TOPK_FILTER")) : 1;
    static const int use_rank_filter = getenv("ASR_RANK_FILTER") ? atoi(getenv("ASR_RANK_FILTER")) : 1;
    static const int fused = getenv("ASR_TOPK_RANK_FUSED") ? atoi(getenv("ASR_TOPK_RANK_FUSED")) : 1;
    return use_filter && use_rank_filter && fused && n_db >= 16384 && kk <= 8192;
}

// d*, j* (global index) of queries whose correct candidates lie in this shard (rows [item_offset, item_offset + n_db) of a
// pool of n2_global rows)
hipError_t launch_rank_dstar(hipStream_t s, const float *q, const double *norm_q, int64_t n_q, const float *db,
                             const double *norm_db, int64_t n_db, int64_t item_offset, int64_t n2_global,
                             int64_t query_offset, int64_t kk, int64_t hh, double *dstar, int64_t *jstar) {
    if (n_q == 0) return hipSuccess;
    rank_dstar_kernel<<<(unsigned)((n_q + 3) / 4), 256, 0, s>>>(q, norm_q, n_q, db, norm_db, n_db, query_offset, kk, hh, dstar, jstar,
                                                                item_offset, n2_global);
    return hipGetLastError();
}

hipError_t launch_rank_finish(hipStream_t s, const int32_t *counts, const double *dstar, int64_t n, int32_t *ranks,
                              double *dstar_out, int32_t *ties) {
    if (n == 0) return hipSuccess;
    rank_finish_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(counts, dstar, n, ranks, dstar_out, ties);
    return hipGetLastError();
}

// the k smallest keys per query among n_parts lists laid out [part][n_q_total][k] (lists all-gathered from the ranks of a
// sharded pool): queries q_lo .. q_lo + n_q
hipError_t launch_topk_merge(hipStream_t s, const int32_t *part_idx, const double *part_dist, int n_parts, int64_t n_q_total,
                             int64_t q_lo, int64_t n_q, int k, int32_t *idx_out, double *dist_out) {
    if (n_q == 0) return hipSuccess;
    if (k < 1 || n_parts < 1 || (int64_t)n_parts * k > TOPK_SORT) return hipErrorInvalidValue;
    topk_merge_kernel<<<(unsigned)n_q, TOPK_THREADS, 0, s>>>(part_idx, part_dist, n_parts, k, (int64_t)1 << 62, idx_out, dist_out, k,
                                                             n_q_total * k, q_lo);
    return hipGetLastError();
}

// shared body of the fused pass: thresholds, filter with the rank counters riding along, exact refine.  counts [n_q][3]
// must be zero; ds / js: d* and (global) j* of every query
static hipError_t topk_count_pass(hipStream_t s, const TopkPlan &P, char *ws, const float *db, const float *unit,
                                  const double *norm_db, int64_t n_db, const float *q, const double *norm_q, int64_t n_q, int k,
                                  int64_t idx_offset, int32_t *idx_out, double *dist_out, const double *ds, const int64_t *js,
                                  int32_t *counts) {
    float *rn_q = (float *)(ws + P.off_rn_q);
    int32_t *cand_cnt = (int32_t *)(ws + P.off_cnt), *cand_idx = (int32_t *)(ws + P.off_idx);
    RankFuse R{db, norm_db, norm_q, ds, js, counts, idx_offset};
    float *thr0 = nullptr;
    if (P.seeded) {
        thr0 = (float *)(ws + P.off_thr0);
        seed_thresholds(s, P, ws, unit, db, norm_db, n_db, q, norm_q, rn_q, n_q, k, thr0);
    }
    launch_filter<true, true>(s, P, unit, nullptr, n_db, q, rn_q, n_q, k, cand_idx, cand_cnt, R, thr0);
    launch_refine(s, P, ws, db, norm_db, n_db, 32, q, norm_q, n_q, 32, 32, k, idx_offset, idx_out, dist_out);
    return hipGetLastError();
}

// top-k against this data base (a shard: global indices = local + idx_offset) and the rank COUNTERS of the same queries
// against it, d* / j* given: what one rank of a sharded pool contributes (counts are summed over the shards)
hipError_t launch_topk_count_db(hipStream_t s, const float *db, const float *unit, const double *norm_db, int64_t n_db,
                                const float *q, const double *norm_q, int64_t n_q, int k, int64_t idx_offset,
                                int32_t *idx_out, double *dist_out, const double *dstar, const int64_t *jstar,
                                int32_t *counts, void *workspace) {
    if (n_q == 0) return hipSuccess;
    if (k < 1 || k > TOPK_KMAX || !workspace || !unit) return hipErrorInvalidValue;
    const TopkPlan P = plan_topk(n_db, n_q, k, true, true);
    char *ws = (char *)workspace;
    rnorm_f32_kernel<<<(unsigned)((n_q + 255) / 256), 256, 0, s>>>(norm_q, n_q, (float *)(ws + P.off_rn_q));
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)n_q * 3 * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    return topk_count_pass(s, P, ws, db, unit, norm_db, n_db, q, norm_q, n_q, k, idx_offset, idx_out, dist_out, dstar, jstar, counts);
}

hipError_t launch_topk_rank_db(hipStream_t s, const float *db, const float *unit, const double *norm_db, int64_t n_db,
                               const float *q, const double *norm_q, int64_t n_q, int k, int64_t idx_offset,
                               int32_t *idx_out, double *dist_out, int64_t query_offset, int64_t kk, int64_t hh,
                               int32_t *ranks, double *dstar, int32_t *ties, void *workspace) {
    if (n_q == 0) return hipSuccess;
    if (k < 1 || k > TOPK_KMAX || !workspace || !unit) return hipErrorInvalidValue;
    const TopkPlan P = plan_topk(n_db, n_q, k, true, true);
    char *ws = (char *)workspace;
    float *rn_q = (float *)(ws + P.off_rn_q);
    int32_t *cand_cnt = (int32_t *)(ws + P.off_cnt), *cand_idx = (int32_t *)(ws + P.off_idx);
    double *ds = (double *)(ws + P.off_ds);
    int64_t *js = (int64_t *)(ws + P.off_js);
    int32_t *counts = (int32_t *)(ws + P.off_counts);
    rnorm_f32_kernel<<<(unsigned)((n_q + 255) / 256), 256, 0, s>>>(norm_q, n_q, rn_q);
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)n_q * 3 * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    rank_dstar_kernel<<<(unsigned)((n_q + 3) / 4), 256, 0, s>>>(q, norm_q, n_q, db, norm_db, n_db, query_offset, kk, hh, ds, js, 0, n_db);
    RankFuse R{db, norm_db, norm_q, ds, js, counts, 0};
    float *thr0 = nullptr;
    if (P.seeded) {
        thr0 = (float *)(ws + P.off_thr0);
        seed_thresholds(s, P, ws, unit, db, norm_db, n_db, q, norm_q, rn_q, n_q, k, thr0);
    }
    launch_filter<true, true>(s, P, unit, nullptr, n_db, q, rn_q, n_q, k, cand_idx, cand_cnt, R, thr0);
    launch_refine(s, P, ws, db, norm_db, n_db, 32, q, norm_q, n_q, 32, 32, k, idx_offset, idx_out, dist_out);
    rank_finish_kernel<<<(unsigned)((n_q + 255) / 256), 256, 0, s>>>(counts, ds, n_q, ranks, dstar, ties);
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

#if defined(ASR_TF_ABL) && (ASR_TF_ABL & 4)
extern "C" int asr_debug_tf_trace(unsigned long long *out, int n_words) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(asr::g_tf_trace), (size_t)n_words * sizeof(unsigned long long));
}
#endif
