// gfx950 kernels of the TRAIN-mode forward pass (deterministic=False outputs of
// utils/train_dcca_pool.py:100-101): raw convolutions come from the MFMA kernel
// (RAW epilogue); this file adds what batch statistics need.
//
//   conv1_raw_kernel        : block 1 (C_in = 1) without BN/ELU
//   bn_stats_partial/final  : per-channel batch mean and biased variance over
//                             (N,H,W), float64 accumulation, block-
//                             ordered (deterministic) reduction; also applies the
//                             EMA side effect mean <- .9 mean + .1 mu,
//                             inv_std <- .9 inv_std + .1 s to the master params
//   bn_apply_elu_pool_kernel: y = (z-mu)*(gamma*s)+beta, ELU, 2x2 max-pool
//   conv1x1_raw_kernel, bn_gpool_kernel : block 9 + GlobalPoolLayer
#include "asr_kernels.h"
#include <algorithm>

namespace asr {

__device__ __forceinline__ float elu_t(float v) { return v > 0.0f ? v : expm1f(v); }

// ---------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void conv1_raw_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float *__restrict__ z, int N, int H, int W) {
    // x: (N,H,W) prepared float32; w: [COUT][9] correlation-form taps; z: (N,H,W,COUT)
    const int64_t total = (int64_t)N * H * W;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(s % W);
        const int64_t q = s / W;
        const int y = (int)(q % H);
        const int n = (int)(q / H);
        float v[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = y - 1 + a, xb = xx - 1 + b;
                v[a * 3 + b] = (yy >= 0 && yy < H && xb >= 0 && xb < W) ? x[((size_t)n * H + yy) * W + xb] : 0.0f;
            }
        float *o = z + (size_t)s * COUT;
#pragma unroll 1
        for (int cg = 0; cg < COUT / 4; ++cg) {
            float r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(v[t], w[(cg * 4 + c) * 9 + t], acc);
                r[c] = acc;
            }
            *reinterpret_cast<float4 *>(o + cg * 4) = make_float4(r[0], r[1], r[2], r[3]);
        }
    }
}

hipError_t launch_conv1_raw(hipStream_t s, const float *x, const float *w, float *z, int N, int H, int W, int cout) {
    const int64_t total = (int64_t)N * H * W;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    if (blocks == 0) return hipSuccess;
    if (cout == 12) conv1_raw_kernel<12><<<blocks, 256, 0, s>>>(x, w, z, N, H, W);
    else if (cout == 24) conv1_raw_kernel<24><<<blocks, 256, 0, s>>>(x, w, z, N, H, W);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
constexpr int BNS_THREADS = 256;
constexpr int BNS_MAXC = 128;

// rows x C float32, row-major.  partial: [gridDim.x][2][C] float64 (sum, sum of squares).
// thread = (row lane r, float4 channel group c4): 16-B loads, float64 accumulation.
__global__ __launch_bounds__(BNS_THREADS) void bn_stats_partial_kernel(const float *__restrict__ z, int64_t rows, int C,
                                                                       int64_t rows_per_block,
                                                                       double *__restrict__ partial) {
    __shared__ double s1[BNS_THREADS * 4], s2[BNS_THREADS * 4];
    const int tid = threadIdx.x;
    const int C4 = C >> 2;
    const int rpi = BNS_THREADS / C4;               // rows per iteration
    const int r = tid / C4, c4 = tid - r * C4;
    const bool active = r < rpi;
    const int64_t lo = (int64_t)blockIdx.x * rows_per_block;
    const int64_t hi = lo + rows_per_block < rows ? lo + rows_per_block : rows;
    double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    if (active)
        for (int64_t i = lo + r; i < hi; i += rpi) {
            const float4 v = *reinterpret_cast<const float4 *>(z + i * C + c4 * 4);
            const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
            a1[0] += d0; a1[1] += d1; a1[2] += d2; a1[3] += d3;
            a2[0] += d0 * d0; a2[1] += d1 * d1; a2[2] += d2 * d2; a2[3] += d3 * d3;
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[tid * 4 + k] = a1[k]; s2[tid * 4 + k] = a2[k]; }
    __syncthreads();
    if (tid < C) {
        const int cc4 = tid >> 2, k = tid & 3;
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < rpi; ++q) { t1 += s1[(q * C4 + cc4) * 4 + k]; t2 += s2[(q * C4 + cc4) * 4 + k]; }
        partial[((size_t)blockIdx.x * 2) * C + tid] = t1;
        partial[((size_t)blockIdx.x * 2 + 1) * C + tid] = t2;
    }
}

// stats: [2][C] float32 (mu, inv_std).  run_mean / run_istd (may be null): EMA side effect.
__global__ __launch_bounds__(BNS_MAXC) void bn_stats_final_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                                  double count, float eps, float ema,
                                                                  float *__restrict__ stats,
                                                                  float *__restrict__ run_mean,
                                                                  float *__restrict__ run_istd) {
    const int c = threadIdx.x;
    if (c >= C) return;
    double t1 = 0.0, t2 = 0.0;
    for (int b = 0; b < nblocks; ++b) {
        t1 += partial[((size_t)b * 2) * C + c];
        t2 += partial[((size_t)b * 2 + 1) * C + c];
    }
    const double mu = t1 / count;
    double var = t2 / count - mu * mu;            // biased variance, float64: no cancellation issue
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu;
    const float istd = 1.0f / sqrtf((float)var + eps);
    stats[c] = muf;
    stats[C + c] = istd;
    if (run_mean) run_mean[c] = (1.0f - ema) * run_mean[c] + ema * muf;
    if (run_istd) run_istd[c] = (1.0f - ema) * run_istd[c] + ema * istd;
}

int bn_stats_blocks(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, (rows + 2047) / 2048)); }

// data-parallel form of bn_stats_final_kernel: block-ordered column sums only ...
__global__ __launch_bounds__(BNS_MAXC) void bn_stats_sum_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                                double *__restrict__ sums) {
    const int c = threadIdx.x;
    if (c >= C) return;
    double t1 = 0.0, t2 = 0.0;
    for (int b = 0; b < nblocks; ++b) {
        t1 += partial[((size_t)b * 2) * C + c];
        t2 += partial[((size_t)b * 2 + 1) * C + c];
    }
    sums[c] = t1; sums[C + c] = t2;
}
// ... and, after the all-reduce over the ranks, the statistics of the full batch (`count` = global rows)
__global__ __launch_bounds__(BNS_MAXC) void bn_stats_finish_kernel(const double *__restrict__ sums, int C, double count,
                                                                   float eps, float ema, float *__restrict__ stats,
                                                                   float *__restrict__ run_mean,
                                                                   float *__restrict__ run_istd) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const double mu = sums[c] / count;
    double var = sums[C + c] / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu;
    const float istd = 1.0f / sqrtf((float)var + eps);
    stats[c] = muf;
    stats[C + c] = istd;
    if (run_mean) run_mean[c] = (1.0f - ema) * run_mean[c] + ema * muf;
    if (run_istd) run_istd[c] = (1.0f - ema) * run_istd[c] + ema * istd;
}

hipError_t launch_bn_stats(hipStream_t s, const float *z, int64_t rows, int C, double *partial, float *stats,
                           float *run_mean, float *run_istd, float eps, float ema, const Exchange *ex, double *sums) {
    if (C > BNS_MAXC || C < 4 || C % 4) return hipErrorInvalidValue;
    const int nb = bn_stats_blocks(rows);
    const int64_t rpb = (rows + nb - 1) / nb;
    bn_stats_partial_kernel<<<nb, BNS_THREADS, 0, s>>>(z, rows, C, rpb, partial);
    if (ex) {
        if (!sums) return hipErrorInvalidValue;
        bn_stats_sum_kernel<<<1, BNS_MAXC, 0, s>>>(partial, nb, C, sums);
        if (ex->allreduce_f64(ex->self, s, sums, 2 * C) != 0) return hipErrorUnknown;
        bn_stats_finish_kernel<<<1, BNS_MAXC, 0, s>>>(sums, C, (double)rows * ex->world, eps, ema, stats, run_mean,
                                                      run_istd);
    } else {
        bn_stats_final_kernel<<<1, BNS_MAXC, 0, s>>>(partial, nb, C, (double)rows, eps, ema, stats, run_mean, run_istd);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// z: (N,H,W,C) raw conv output; out: (N,OH,OW,C) with OH = H/2 (floor) when pool.  thread = (output pixel, 4 channels)
__global__ __launch_bounds__(256) void bn_apply_elu_pool_kernel(const float *__restrict__ z, const float *__restrict__ stats,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float *__restrict__ out,
                                                                int N, int H, int W, int C, int pool, int elu) {
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C4) * 4;
        int64_t q = e / C4;
        const int ox = (int)(q % OW); q /= OW;
        const int oy = (int)(q % OH);
        const int n = (int)(q / OH);
        const float4 mu = *reinterpret_cast<const float4 *>(stats + c);
        const float4 is = *reinterpret_cast<const float4 *>(stats + C + c);
        const float4 ga = *reinterpret_cast<const float4 *>(gamma + c);
        const float4 be = *reinterpret_cast<const float4 *>(beta + c);
        const float sc[4] = {ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w};
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, b4[4] = {be.x, be.y, be.z, be.w};
        float res[4];
        if (pool) {
#pragma unroll
            for (int k = 0; k < 4; ++k) res[k] = -3.4e38f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 v4 = *reinterpret_cast<const float4 *>(
                    z + (((size_t)n * H + 2 * oy + (r >> 1)) * W + 2 * ox + (r & 1)) * C + c);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float y = (v[k] - m4[k]) * sc[k] + b4[k];
                    if (elu) y = elu_t(y);
                    res[k] = fmaxf(res[k], y);
                }
            }
        } else {
            const float4 v4 = *reinterpret_cast<const float4 *>(z + (((size_t)n * H + oy) * W + ox) * C + c);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                res[k] = (v[k] - m4[k]) * sc[k] + b4[k];
                if (elu) res[k] = elu_t(res[k]);
            }
        }
        *reinterpret_cast<float4 *>(out + e * 4) = make_float4(res[0], res[1], res[2], res[3]);
    }
}

hipError_t launch_bn_apply(hipStream_t s, const float *z, const float *stats, const float *gamma, const float *beta,
                           float *out, int N, int H, int W, int C, int pool, int elu) {
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int64_t total = (int64_t)N * OH * OW * (C / 4);
    if (total == 0) return hipSuccess;
    if (C % 4) return hipErrorInvalidValue;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 32);
    bn_apply_elu_pool_kernel<<<blocks, 256, 0, s>>>(z, stats, gamma, beta, out, N, H, W, C, pool, elu);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// block 9: z9[n,p,o] = sum_c a8[n,p,c] * w9[o,c]   (1x1 conv, no flip needed)
__global__ __launch_bounds__(256) void conv1x1_raw_kernel(const float *__restrict__ a8, const float *__restrict__ w9,
                                                          float *__restrict__ z9, int64_t rows, int C8) {
    const int64_t total = rows * 32;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(e & 31);
        const int64_t r = e >> 5;
        const float *x = a8 + r * C8;
        const float *w = w9 + (size_t)o * C8;
        float acc = 0.0f;
        for (int c = 0; c < C8; ++c) acc = fmaf(x[c], w[c], acc);
        z9[e] = acc;
    }
}

// H[n,o] = mean_p ( (z9[n,p,o]-mu)*gamma*s + beta )     (BN identity + GlobalPoolLayer)
__global__ __launch_bounds__(256) void bn_gpool_kernel(const float *__restrict__ z9, const float *__restrict__ stats,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       float *__restrict__ Hout, int N, int npix) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * 32) return;
    const int o = e & 31, n = e >> 5;
    const float mu = stats[o], sc = gamma[o] * stats[32 + o], be = beta[o];
    float sum = 0.0f;
    for (int p = 0; p < npix; ++p) sum += (z9[((size_t)n * npix + p) * 32 + o] - mu) * sc + be;
    Hout[e] = sum / (float)npix;
}

hipError_t launch_conv1x1_raw(hipStream_t s, const float *a8, const float *w9, float *z9, int64_t rows, int C8) {
    if (rows == 0) return hipSuccess;
    const int blocks = (int)std::min<int64_t>((rows * 32 + 255) / 256, 256 * 16);
    conv1x1_raw_kernel<<<blocks, 256, 0, s>>>(a8, w9, z9, rows, C8);
    return hipGetLastError();
}

hipError_t launch_bn_gpool(hipStream_t s, const float *z9, const float *stats, const float *gamma, const float *beta,
                           float *Hout, int N, int npix) {
    if (N == 0) return hipSuccess;
    bn_gpool_kernel<<<(N * 32 + 255) / 256, 256, 0, s>>>(z9, stats, gamma, beta, Hout, N, npix);
    return hipGetLastError();
}

}  // namespace asr
