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
// stats (may be null): per-workgroup [sum(COUT) | sum of squares(COUT)] of the outputs, float64 - the BatchNorm
// statistics gathered where z is produced (see wino_stats_store in conv_wino_kernels.hip)
// MODE 0: z and its statistics.  MODE 1: the statistics only (nothing is stored).  MODE 2: BatchNorm (batch statistics
// `bn` = [mu | inv_std], gamma, beta) + ELU applied to the recomputed z, block 1's OUTPUT written to `z` - the pair
// (1, 2) is the training step's block 1 without the raw tensor: the stencil costs 108 FMAs per pixel, the tensor 6 KB
// ... per pixel and pass 48 bytes written and 48 read back, and its three later readers recompute it the same way
// (bn_bwd_reduce_conv1_kernel, conv1_wgrad_kernel) - 3.1 GB of the 20 GB a batch-512 step moves on the sheet tower.
template <int COUT, int MODE>
__global__ __launch_bounds__(256) void conv1_raw_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        float *__restrict__ z, int N, int H, int W,
                                                        double *__restrict__ stats, const float *__restrict__ bn,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta) {
    // x: (N,H,W) prepared float32; w: [COUT][9] correlation-form taps; z: (N,H,W,COUT)
    // A wave's 64 pixels are one contiguous run of 64 * COUT floats of z: the results are parked in LDS (lane-major)
    // and streamed out as fully coalesced 1-KB store instructions (a lane's own float4 stores are 4 * COUT bytes
    // apart: 3.1 TB/s of writes); taps from clamped addresses, zero padding applied afterwards (no bounds branches).
    __shared__ __attribute__((aligned(16))) float wstage[4 * 64 * COUT];
    const int lane = threadIdx.x & 63;
    float *wbuf = wstage + (threadIdx.x >> 6) * 64 * COUT;
    const int64_t total = (int64_t)N * H * W;
    const bool small = total < ((int64_t)1 << 31);
    float a1[COUT], a2[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) { a1[c] = 0.f; a2[c] = 0.f; }
    // the trip count is wave-uniform: lanes past the end recompute the last pixel, their results are not stored or summed
    for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); s0 < total; s0 += (int64_t)gridDim.x * blockDim.x) {
        const bool live = s0 + lane < total;
        const int64_t s = live ? s0 + lane : total - 1;
        int xx, y, n;
        if (small) {
            const unsigned u = (unsigned)s, q = u / (unsigned)W;
            xx = (int)(u - q * (unsigned)W);
            n = (int)(q / (unsigned)H);
            y = (int)(q - (unsigned)n * (unsigned)H);
        } else {
            xx = (int)(s % W);
            const int64_t q = s / W;
            y = (int)(q % H);
            n = (int)(q / H);
        }
        const float *xn = x + (size_t)n * H * W;
        float v[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = y - 1 + a, xb = xx - 1 + b;
                const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xb < 0 ? 0 : (xb >= W ? W - 1 : xb);
                // (a select here is turned back into a conditional load + wait: nine dependent round trips per pixel)
                v[a * 3 + b] = xn[yc * W + xc] * ((yy == yc && xb == xc) ? 1.0f : 0.0f);
            }
#pragma unroll
        for (int cg = 0; cg < COUT / 4; ++cg) {
            float r[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int t = 0; t < 9; ++t) acc = fmaf(v[t], w[(cg * 4 + c) * 9 + t], acc);
                r[c] = acc;
                if (MODE != 2 && live) {
                    a1[cg * 4 + c] += acc;
                    a2[cg * 4 + c] = fmaf(acc, acc, a2[cg * 4 + c]);
                }
                if (MODE == 2) {                        // the expression of bn_apply_elu_pool_kernel (unpooled, ELU)
                    const int co = cg * 4 + c;
                    const float y = (acc - bn[co]) * (gamma[co] * bn[COUT + co]) + beta[co];
                    r[c] = y > 0.0f ? y : __expf(y) - 1.0f;
                }
            }
            if (MODE != 1) *reinterpret_cast<float4 *>(wbuf + lane * COUT + cg * 4) = make_float4(r[0], r[1], r[2], r[3]);
        }
        if (MODE == 1) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 *dst = reinterpret_cast<float4 *>(z + (size_t)s0 * COUT);
        const int64_t lim4 = (total - s0) * (COUT / 4);               // float4 still inside the tensor
#pragma unroll
        for (int k = 0; k < COUT / 4; ++k) {
            const int f = k * 64 + lane;
            const float4 v4 = *reinterpret_cast<const float4 *>(wbuf + f * 4);
            if (f < lim4) dst[f] = v4;
        }
        __builtin_amdgcn_wave_barrier();                              // the buffer is rewritten by the next iteration
    }
    if (MODE == 2 || stats == nullptr) return;
    // a thread's float32 sums cover at most a few dozen pixels; from here on float64: across the wave by shuffles,
    // across the four waves through LDS in wave order
    __shared__ double red[4][2 * COUT];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        double s1 = (double)a1[c], s2 = (double)a2[c];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
        if (lane == 0) { red[wave][c] = s1; red[wave][COUT + c] = s2; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * COUT)
        stats[(size_t)blockIdx.x * 2 * COUT + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

hipError_t launch_conv1_raw(hipStream_t s, const float *x, const float *w, float *z, int N, int H, int W, int cout,
                            double *stats, int *stats_rows, int mode, const float *bn, const float *gamma,
                            const float *beta) {
    const int64_t total = (int64_t)N * H * W;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    if (blocks == 0) return hipSuccess;
    if (stats_rows) *stats_rows = blocks;
    if (cout != 12 && cout != 24) return hipErrorInvalidValue;
#define ASR_C1R(C, M) conv1_raw_kernel<C, M><<<blocks, 256, 0, s>>>(x, w, z, N, H, W, stats, bn, gamma, beta)
    if (mode == 0) { if (cout == 12) ASR_C1R(12, 0); else ASR_C1R(24, 0); }
    else if (mode == 1) { if (cout == 12) ASR_C1R(12, 1); else ASR_C1R(24, 1); }
    else if (mode == 2) { if (cout == 12) ASR_C1R(12, 2); else ASR_C1R(24, 2); }
    else return hipErrorInvalidValue;
#undef ASR_C1R
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
constexpr int BNS_THREADS = 192;     // 8 * 24: a multiple of every C/4 in use
constexpr int BNS_MAXC = 128;

// rows x C float32, row-major.  partial: [gridDim.x][2][C] float64 (sum, sum of squares).
// The tensor is one flat run of float4; 192 = 8 * 24 threads per workgroup and chunks that are multiples of 192
// keep a thread on ONE channel group (C/4 in {3, 6, 12, 24} divides 192): no index arithmetic in the loop, four
// independent 16-B loads in flight per thread, float64 accumulation.
__global__ __launch_bounds__(BNS_THREADS) void bn_stats_partial_kernel(const float *__restrict__ z, int64_t n4, int C,
                                                                       int64_t chunk4, double *__restrict__ partial) {
    __shared__ double s1[BNS_THREADS * 4], s2[BNS_THREADS * 4];
    const int tid = threadIdx.x;
    const int C4 = C >> 2;
    const float4 *z4 = reinterpret_cast<const float4 *>(z);
    const int64_t lo = (int64_t)blockIdx.x * chunk4;
    const int64_t hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
    double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    for (int64_t i = lo + tid; i < hi; i += 4 * BNS_THREADS) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * BNS_THREADS;
            v[u] = j < hi ? z4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double d0 = v[u].x, d1 = v[u].y, d2 = v[u].z, d3 = v[u].w;
            a1[0] += d0; a1[1] += d1; a1[2] += d2; a1[3] += d3;
            a2[0] += d0 * d0; a2[1] += d1 * d1; a2[2] += d2 * d2; a2[3] += d3 * d3;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[tid * 4 + k] = a1[k]; s2[tid * 4 + k] = a2[k]; }
    __syncthreads();
    if (tid < C) {                      // channel tid = group cc4, component k: threads cc4, cc4 + C4, ... hold it
        const int cc4 = tid >> 2, k = tid & 3;
        double t1 = 0.0, t2 = 0.0;
        for (int t = cc4; t < BNS_THREADS; t += C4) { t1 += s1[t * 4 + k]; t2 += s2[t * 4 + k]; }
        partial[((size_t)blockIdx.x * 2) * C + tid] = t1;
        partial[((size_t)blockIdx.x * 2 + 1) * C + tid] = t2;
    }
}

// Block-ordered sum of nblocks partial rows [2][C] (float64) with 1024 threads: slot = one of the 2C values, the
// threads of a slot sum interleaved subsets of the blocks, then a fixed-order pass over LDS.  (A single thread per
// value walking 1024 partials is a 1024-deep chain of dependent global loads: ~0.25 ms per BatchNorm layer.)
constexpr int BNR_THREADS = 1024;
__device__ __forceinline__ double bn_reduce_partials(const double *__restrict__ partial, int nblocks, int C, int slot_of,
                                                     double *lds /*[1024]*/) {
    const int tid = threadIdx.x;
    const int slots = 2 * C;                       // <= 256
    const int groups = BNR_THREADS / slots;        // >= 4
    const int slot = tid % slots, grp = tid / slots;
    double acc = 0.0;
    if (grp < groups)
        for (int b = grp; b < nblocks; b += groups) acc += partial[(size_t)b * slots + slot];
    lds[tid] = (grp < groups) ? acc : 0.0;
    __syncthreads();
    double tot = 0.0;
    if (slot_of >= 0)
        for (int gI = 0; gI < groups; ++gI) tot += lds[gI * slots + slot_of];
    return tot;
}

// stats: [2][C] float32 (mu, inv_std).  run_mean / run_istd (may be null): EMA side effect.
__global__ __launch_bounds__(BNR_THREADS) void bn_stats_final_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                                     double count, float eps, float ema,
                                                                     float *__restrict__ stats,
                                                                     float *__restrict__ run_mean,
                                                                     float *__restrict__ run_istd) {
    __shared__ double red[BNR_THREADS], tot[256];
    const int c = threadIdx.x;
    const double t = bn_reduce_partials(partial, nblocks, C, c < 2 * C ? c : -1, red);
    if (c < 2 * C) tot[c] = t;
    __syncthreads();
    if (c >= C) return;
    const double mu = tot[c] / count;
    double var = tot[C + c] / count - mu * mu;            // biased variance, float64: no cancellation issue
    if (var < 0.0) var = 0.0;
    const float muf = (float)mu;
    const float istd = 1.0f / sqrtf((float)var + eps);
    stats[c] = muf;
    stats[C + c] = istd;
    if (run_mean) run_mean[c] = (1.0f - ema) * run_mean[c] + ema * muf;
    if (run_istd) run_istd[c] = (1.0f - ema) * run_istd[c] + ema * istd;
}

// First stage of the column sums of a long partial table [nb][cols] (float64): one workgroup per 32 rows, written
// behind the table itself ([nb .. nb + ceil(nb/32)) - the owners allocate colsum_stage_extra() more).  The
// single-workgroup finish kernels then walk <= 128 rows instead of up to 4096: one workgroup reading 3 MB was 55 us per
// BatchNorm layer, 1 ms of the batch-512 step.  Fixed partition and order: deterministic for a given nb.
constexpr int CS_ROWS = 32;
__global__ __launch_bounds__(256) void colsum_stage_kernel(const double *__restrict__ partial, int nb, int cols,
                                                           double *__restrict__ out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int groups = 256 / cols;
    const int slot = tid % cols, grp = tid / cols;
    const int r0 = blockIdx.x * CS_ROWS, r1 = min(r0 + CS_ROWS, nb);
    double acc = 0.0;
    if (grp < groups)
        for (int r = r0 + grp; r < r1; r += groups) acc += partial[(size_t)r * cols + slot];
    red[tid] = grp < groups ? acc : 0.0;
    __syncthreads();
    if (tid < cols) {
        double t = 0.0;
        for (int gI = 0; gI < groups; ++gI) t += red[gI * cols + tid];
        out[(size_t)blockIdx.x * cols + tid] = t;
    }
}
size_t colsum_stage_extra(size_t partial_doubles) { return partial_doubles / CS_ROWS + 512; }
double *colsum_stage(hipStream_t s, double *partial, int *nb, int cols) {
    if (*nb <= 96 || cols > 256) return partial;
    const int g = (*nb + CS_ROWS - 1) / CS_ROWS;
    double *out = partial + (size_t)*nb * cols;
    colsum_stage_kernel<<<g, 256, 0, s>>>(partial, *nb, cols, out);
    *nb = g;
    return out;
}

// ---- column sums + finish in ONE launch (ColsumFinalArgs, asr_kernels.h) -----------------------------------------
__global__ __launch_bounds__(256) void colsum_final_kernel(ColsumFinalArgs a) {
    __shared__ double red[256], tot[256];
    __shared__ int is_last;
    const int tid = threadIdx.x, cols = a.cols;
    const int groups = 256 / cols;                                 // >= 1 (cols <= 256)
    const int slot = tid % cols, grp = tid / cols;
    {
        const int r0 = blockIdx.x * CS_ROWS, r1 = min(r0 + CS_ROWS, a.nb);
        double acc = 0.0;
        if (grp < groups)
            for (int r = r0 + grp; r < r1; r += groups) {
                double *p = a.partial + (size_t)r * cols + slot;
                acc += *p;
                if (a.zero_rows) *p = 0.0;                         // the table is all-zero again when this launch ends
            }
        red[tid] = grp < groups ? acc : 0.0;
        __syncthreads();
        if (tid < cols) {
            double t = 0.0;
            for (int gI = 0; gI < groups; ++gI) t += red[gI * cols + tid];
            a.staged[(size_t)blockIdx.x * cols + tid] = t;
        }
    }
    // the last workgroup to get here sums the staged rows - in row order, whoever it is.  Release (L2 write-back of the
    // staged row) by the threads that wrote it, then the ticket; acquire (L2 invalidate) only in the workgroup that goes
    // on to read the others' rows: a full __threadfence() in every workgroup did both everywhere (eight XCDs, eight L2s)
    if (tid < cols) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = atomicAdd(a.ticket, 1u);
        is_last = (t == gridDim.x - 1) ? 1 : 0;
        if (is_last) *a.ticket = 0u;                               // ready for the next launch (stream order)
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        const int nst = (int)gridDim.x;
        double acc = 0.0;
        if (grp < groups)
            for (int r = grp; r < nst; r += groups) acc += __builtin_nontemporal_load(a.staged + (size_t)r * cols + slot);
        __syncthreads();
        red[tid] = grp < groups ? acc : 0.0;
        __syncthreads();
        if (tid < cols) {
            double t = 0.0;
            for (int gI = 0; gI < groups; ++gI) t += red[gI * cols + tid];
            tot[tid] = t;
        }
        __syncthreads();
    }
    const int C = a.C;
    if (a.mode == 0) {                                             // bn_stats_final_kernel's arithmetic
        if (tid >= C) return;
        const double mu = tot[tid] / a.count;
        double var = tot[C + tid] / a.count - mu * mu;             // biased variance, float64: no cancellation issue
        if (var < 0.0) var = 0.0;
        const float muf = (float)mu;
        const float istd = 1.0f / sqrtf((float)var + a.eps);
        a.stats[tid] = muf;
        a.stats[C + tid] = istd;
        if (a.run_mean) a.run_mean[tid] = (1.0f - a.ema) * a.run_mean[tid] + a.ema * muf;
        if (a.run_istd) a.run_istd[tid] = (1.0f - a.ema) * a.run_istd[tid] + a.ema * istd;
    } else {                                                       // bn_stats_sum_kernel / bn_bwd_final_kernel
        if (tid >= cols) return;
        a.sums[tid] = tot[tid];
        if (a.mode == 2) {
            if (tid < C) a.dbeta[tid] = (float)tot[tid]; else a.dgamma[tid - C] = (float)tot[tid];
        }
    }
}

hipError_t launch_colsum_final(hipStream_t s, const ColsumFinalArgs &a) {
    if (a.nb < 1 || a.cols < 1 || a.cols > 256 || !a.ticket || !a.staged) return hipErrorInvalidValue;
    colsum_final_kernel<<<(a.nb + CS_ROWS - 1) / CS_ROWS, 256, 0, s>>>(a);
    return hipGetLastError();
}

// (512 rows per workgroup: block 9's 61 440 rows were 30 workgroups - 20 us for 8 MB)
int bn_stats_blocks(int64_t rows) { return (int)std::max<int64_t>(1, std::min<int64_t>(1024, (rows + 511) / 512)); }

// data-parallel form of bn_stats_final_kernel: block-ordered column sums only ...
__global__ __launch_bounds__(BNR_THREADS) void bn_stats_sum_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                                   double *__restrict__ sums) {
    __shared__ double red[BNR_THREADS];
    const int c = threadIdx.x;
    const double t = bn_reduce_partials(partial, nblocks, C, c < 2 * C ? c : -1, red);
    if (c < 2 * C) sums[c] = t;
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
                           float *run_mean, float *run_istd, float eps, float ema, const Exchange *ex, double *sums,
                           unsigned *ticket) {
    if (C > BNS_MAXC || C < 4 || C % 4) return hipErrorInvalidValue;
    if (BNS_THREADS % (C / 4)) return hipErrorInvalidValue;
    const int nb = bn_stats_blocks(rows);
    const int64_t n4 = rows * (C / 4);
    int64_t chunk4 = (n4 + nb - 1) / nb;
    chunk4 = (chunk4 + BNS_THREADS - 1) / BNS_THREADS * BNS_THREADS;      // thread <-> channel group stays fixed
    if (!(ex && ex->phase == 2)) bn_stats_partial_kernel<<<nb, BNS_THREADS, 0, s>>>(z, n4, C, chunk4, partial);
    return launch_bn_stats_final(s, partial, nb, rows, C, stats, run_mean, run_istd, eps, ema, ex, sums, ticket, false);
}

// The partial table came from the convolution itself (conv3x3_wino / conv3x3_winog RAW epilogues, conv1_raw_kernel) or
// from bn_stats_partial_kernel: the block-ordered finish.  nb rows of [2][C] float64.
hipError_t launch_bn_stats_final(hipStream_t s, double *partial_in, int nb, int64_t rows, int C, float *stats,
                                 float *run_mean, float *run_istd, float eps, float ema, const Exchange *ex, double *sums,
                                 unsigned *ticket, bool zero_rows, double *staged) {
    if (C > BNS_MAXC || C < 4 || nb < 1) return hipErrorInvalidValue;
    static const bool fused = (getenv("ASR_TRAIN_FUSED_REDUCE") && getenv("ASR_TRAIN_FUSED_REDUCE")[0] == '1');
    if (!ex && ticket && 2 * C <= 256 && (fused || zero_rows)) {
        ColsumFinalArgs a{};
        a.partial = partial_in; a.nb = nb; a.cols = 2 * C; a.staged = staged ? staged : partial_in + (size_t)nb * 2 * C; a.ticket = ticket;
        if (zero_rows && !staged) return hipErrorInvalidValue;
        a.zero_rows = zero_rows ? 1 : 0; a.mode = 0; a.C = C; a.count = (double)rows; a.eps = eps; a.ema = ema;
        a.stats = stats; a.run_mean = run_mean; a.run_istd = run_istd;
        return launch_colsum_final(s, a);
    }
    if (ex) {
        if (!sums) return hipErrorInvalidValue;
        if (ex->phase != 2) {
            const double *partial = colsum_stage(s, partial_in, &nb, 2 * C);
            bn_stats_sum_kernel<<<1, BNR_THREADS, 0, s>>>(partial, nb, C, sums);
        }
        if (ex->phase == 0 && ex->allreduce_f64(ex->self, s, sums, 2 * C) != 0) return hipErrorUnknown;
        if (ex->phase != 1)
            bn_stats_finish_kernel<<<1, BNS_MAXC, 0, s>>>(sums, C, (double)(rows / ex->n_local) * ex->n_global, eps, ema, stats,
                                                          run_mean, run_istd);
    } else {
        const double *partial = colsum_stage(s, partial_in, &nb, 2 * C);
        bn_stats_final_kernel<<<1, BNR_THREADS, 0, s>>>(partial, nb, C, (double)rows, eps, ema, stats, run_mean, run_istd);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// z: (N,H,W,C) raw conv output; out: (N,OH,OW,C) with OH = H/2 (floor) when pool.
// grid = (chunks of one image, images); 192 threads; a thread keeps ONE channel group (its BN constants live in
// registers) and walks the image's output pixels with a constant pixel step - 32-bit indices, no divisions by C.
__device__ __forceinline__ float elu_fastt(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }
__device__ __forceinline__ int fdivt(int n, float rcp) { return (int)(((float)n + 0.5f) * rcp); }

__global__ __launch_bounds__(BNS_THREADS) void bn_apply_elu_pool_kernel(const float *__restrict__ z,
                                                                        const float *__restrict__ stats,
                                                                        const float *__restrict__ gamma,
                                                                        const float *__restrict__ beta,
                                                                        float *__restrict__ out, int N, int H, int W, int C,
                                                                        int pool, int elu, float *__restrict__ zsel,
                                                                        uint8_t *__restrict__ ztie, float *__restrict__ snap) {
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int C4 = C >> 2;
    const int tid = threadIdx.x;
    const int c4 = tid % C4, c = c4 * 4;
    const int opix = OH * OW;
    const int q0 = (blockIdx.x * BNS_THREADS + tid) / C4;         // first output pixel of this thread
    const int qstep = gridDim.x * (BNS_THREADS / C4);
    const float rcpOW = 1.0f / (float)OW;
    float m4[4], sc[4], b4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m4[k] = stats[c + k];
        sc[k] = gamma[c + k] * stats[C + c + k];
        b4[k] = beta[c + k];
    }
    // the scale and shift this pass compares with, kept for launch_pool_mask (gamma / beta move with the next update)
    if (snap && blockIdx.x == 0 && blockIdx.y == 0 && tid < C4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { snap[c + k] = sc[k]; snap[C + c + k] = b4[k]; }
    }
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const float *zn = z + (size_t)n * H * W * C + c;
        float *on = out + (size_t)n * opix * C + c;
        float *sn = zsel ? zsel + (size_t)n * opix * C + c : nullptr;
        uint8_t *tn = ztie ? ztie + (size_t)n * opix * C4 + c4 : nullptr;
#pragma unroll 4
        for (int q = q0; q < opix; q += qstep) {
            float res[4];
            if (pool) {
                const int oy = fdivt(q, rcpOW), ox = q - oy * OW;
                const float *zp = zn + ((size_t)(2 * oy) * W + 2 * ox) * C;
                const float4 v0 = *reinterpret_cast<const float4 *>(zp);
                const float4 v1 = *reinterpret_cast<const float4 *>(zp + C);
                const float4 v2 = *reinterpret_cast<const float4 *>(zp + (size_t)W * C);
                const float4 v3 = *reinterpret_cast<const float4 *>(zp + (size_t)W * C + C);
                const float v[4][4] = {{v0.x, v0.y, v0.z, v0.w}, {v1.x, v1.y, v1.z, v1.w},
                                       {v2.x, v2.y, v2.z, v2.w}, {v3.x, v3.y, v3.z, v3.w}};
                if (sn) {
                    // train mode: also the raw value of the FIRST window element with the largest y, so that the reduce
                    // pass of bn_bwd_* reads one value per window instead of four (z of a pooled block is 4x its
                    // output) - and, under the "every tied element" pooling gradient (tn), how many elements share that
                    // largest y: two bits per channel.  y by bn_affine: the backward pass compares the same values.
                    float vb[4];
                    unsigned tie = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float yv[4];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) yv[rr] = bn_affine(v[rr][k], m4[k], sc[k], b4[k]);
                        float yb = yv[0];
                        vb[k] = v[0][k];
#pragma unroll
                        for (int rr = 1; rr < 4; ++rr)
                            if (yv[rr] > yb) { yb = yv[rr]; vb[k] = v[rr][k]; }
                        int cnt = 0;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) cnt += (yv[rr] == yb) ? 1 : 0;
                        tie |= (unsigned)((cnt > 0 ? cnt - 1 : 0) & 3) << (2 * k);
                        res[k] = elu ? elu_fastt(yb) : yb;
                    }
                    *reinterpret_cast<float4 *>(sn + (size_t)q * C) = make_float4(vb[0], vb[1], vb[2], vb[3]);
                    if (tn) tn[(size_t)q * C4] = (uint8_t)tie;
                } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // BN is affine and ELU monotone: max over the window of ELU(BN(v)) = ELU(BN(max or min of v))
                    const float hi = fmaxf(fmaxf(v[0][k], v[1][k]), fmaxf(v[2][k], v[3][k]));
                    const float lo = fminf(fminf(v[0][k], v[1][k]), fminf(v[2][k], v[3][k]));
                    const float y = bn_affine(sc[k] >= 0.0f ? hi : lo, m4[k], sc[k], b4[k]);
                    res[k] = elu ? elu_fastt(y) : y;
                }
                }
            } else {
                const float4 v4 = *reinterpret_cast<const float4 *>(zn + (size_t)q * C);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float y = bn_affine(v[k], m4[k], sc[k], b4[k]);
                    res[k] = elu ? elu_fastt(y) : y;
                }
            }
            *reinterpret_cast<float4 *>(on + (size_t)q * C) = make_float4(res[0], res[1], res[2], res[3]);
        }
    }
}

hipError_t launch_bn_apply(hipStream_t s, const float *z, const float *stats, const float *gamma, const float *beta,
                           float *out, int N, int H, int W, int C, int pool, int elu, float *zsel, uint8_t *ztie,
                           float *snap) {
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int64_t per_img = (int64_t)OH * OW * (C / 4);
    if (per_img * N == 0) return hipSuccess;
    if (C % 4 || BNS_THREADS % (C / 4)) return hipErrorInvalidValue;
    const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((per_img + BNS_THREADS - 1) / BNS_THREADS, 64));
    const int by = std::max(1, std::min(N, 8192 / bx));
    bn_apply_elu_pool_kernel<<<dim3(bx, by), BNS_THREADS, 0, s>>>(z, stats, gamma, beta, out, N, H, W, C, pool, elu,
                                                                  pool ? zsel : nullptr, (pool && zsel) ? ztie : nullptr,
                                                                  snap);
    return hipGetLastError();
}

// debug export (asr_debug_train_tensor kind 10): the set of window elements the "every tied element" pooling gradient
// feeds, as the backward pass decides it - y by bn_affine, equality with the window maximum
__global__ __launch_bounds__(256) void pool_mask_kernel(const float *__restrict__ z, const float *__restrict__ stats,
                                                        const float *__restrict__ snap,
                                                        float *__restrict__ mask, int64_t total, int H, int W, int C) {
    const int OH = H / 2, OW = W / 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        int64_t r = e / C;
        const int ox = (int)(r % OW); r /= OW;
        const int oy = (int)(r % OH);
        const int64_t n = r / OH;
        const float mu = stats[c], sc = snap[c], be = snap[C + c];
        float yv[4], yb = -3.4e38f;
        for (int rr = 0; rr < 4; ++rr) {
            yv[rr] = bn_affine(z[((n * H + 2 * oy + (rr >> 1)) * W + 2 * ox + (rr & 1)) * C + c], mu, sc, be);
            yb = fmaxf(yb, yv[rr]);
        }
        int m = 0;
        for (int rr = 0; rr < 4; ++rr) m |= (yv[rr] == yb) ? (1 << rr) : 0;
        mask[e] = (float)m;
    }
}

hipError_t launch_pool_mask(hipStream_t s, const float *z, const float *stats, const float *snap, float *mask, int N, int H,
                            int W, int C) {
    const int64_t total = (int64_t)N * (H / 2) * (W / 2) * C;
    if (total == 0) return hipSuccess;
    pool_mask_kernel<<<(int)std::min<int64_t>((total + 255) / 256, 65536), 256, 0, s>>>(z, stats, snap, mask, total, H, W, C);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// block 9: z9[n,p,o] = sum_c a8[n,p,c] * w9[o,c]   (1x1 conv, no flip needed)
__global__ __launch_bounds__(256) void conv1x1_raw_kernel(const float *__restrict__ a8, const float *__restrict__ w9,
                                                          float *__restrict__ z9, int64_t rows, int C8) {
    // thread = (row, output channel o): the 32 threads of a row read the same 16 bytes of a8 (one broadcast
    // transaction), w9 row o stays in L1.  (One scalar load pair per FMA took 98 us per tower at 61 440 rows.)
    const int64_t total = rows * 32;
    const int c4n = C8 >> 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(e & 31);
        const int64_t r = e >> 5;
        const float4 *x = reinterpret_cast<const float4 *>(a8 + r * C8);
        const float4 *w = reinterpret_cast<const float4 *>(w9 + (size_t)o * C8);
        float acc = 0.0f;
#pragma unroll 4
        for (int c = 0; c < c4n; ++c) {
            const float4 xv = x[c], wv = w[c];
            acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc);      // (the summation order of the scalar loop)
            acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
        }
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
    if (C8 % 4) return hipErrorInvalidValue;
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
