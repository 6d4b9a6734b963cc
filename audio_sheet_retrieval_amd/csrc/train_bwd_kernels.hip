// gfx950 kernels of the backward pass and the parameter update of the training
// step (theano.grad + lasagne.updates.adam in utils/train_dcca_pool.py:148-151;
// gradient rules SURVEY A.2-A.4, A.7).
//
//   bn_bwd_reduce / bn_bwd_apply : gradient through max-pool (every element equal
//        to the maximum of its 2x2 window - Theano's CPU MaxPoolGrad, SURVEY 8a row
//        3 - or only the first: asr_config.pool_ties), ELU and train-mode BatchNorm;
//        recomputes y from the raw conv output z instead of storing it
//   wgrad_mfma_kernel  : dW[tap][ci][co] = sum_pixels x[pix+tap][ci] dz[pix][co] on
//        v_mfma_f32_16x16x4_f32 (M = ci, N = co, K = pixels), one wave per tap,
//        accumulators persistent across the workgroup's tiles, per-block partials
//   wgrad_reduce_kernel: block-ordered (deterministic) sum -> OIHW gradient
//   conv1_wgrad_kernel : block 1 (C_in = 1)
//   tail_bwd_*         : GlobalPool + BN + 1x1 conv backward
//   adam_kernel        : L2 term + Lasagne Adam
//   repack_*           : master OIHW weights -> MFMA fragment order (forward and
//        data-gradient forms), BN fold for the deterministic path
#include "asr_kernels.h"
#include "repack_elems.inl"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace asr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// BN / ELU / pool backward
// ---------------------------------------------------------------------------
constexpr int BB_THREADS = 192;      // 8 * 24: a multiple of every C/4 in use

struct BnBwdArgs {
    const float *z;        // (N,H,W,C) raw conv output
    float *dz;             // output (may alias z)
    const float *dout;     // (N,OH,OW,C) gradient wrt the block output
    const float *stats;    // [mu | inv_std]
    const float *gamma, *beta;
    double *partial;       // [blocks][2][C]
    const double *sums;    // [2][C] reduced (apply pass)
    int N, H, W, C, pool, elu;
    int Ng;                // samples of the WHOLE batch (data parallel: all ranks' shards, which may differ in size)
    const float *zsel;     // pooled blocks, may be null: (N,OH,OW,C) raw value of each window's selected element
    const uint8_t *ztie;   // with zsel, "every tied element" rule: (N,OH,OW,C/4) bytes, 2 bits per channel = ties - 1
    int ties_first;        // 1: only the first maximal element of a window receives the gradient; 0: every one
};

// y value and ELU' of one raw element
__device__ __forceinline__ void bn_y(float v, float mu, float sc, float be, int elu, float &y, float &dact) {
    y = bn_affine(v, mu, sc, be);
    dact = 1.0f;
    if (elu && y <= 0.0f) dact = __expf(y);          // ELU'(y) = exp(y) for y <= 0
}

__device__ __forceinline__ int fdivb(int n, float rcp) { return (int)(((float)n + 0.5f) * rcp); }

// reduce pass: a1 = sum dy, a2 = sum dy * xhat.  grid = (chunks of one image, images), 192 = 8 * 24 threads: a thread
// keeps ONE channel group (constants in registers) and walks the image's output pixels with a constant step.
// One instantiation per pooling mode (the shared kernel kept 140 registers - three waves per SIMD - and evaluated
// ELU' = exp(y) for all four window elements before choosing the maximum: 16 quarter-rate exponentials per float4 of
// output gradient instead of 4).
template <bool POOL>
__global__ __launch_bounds__(BB_THREADS) void bn_bwd_reduce_kernel(BnBwdArgs a) {
    __shared__ double s1[BB_THREADS * 4], s2[BB_THREADS * 4];
    const int tid = threadIdx.x, C = a.C, C4 = C >> 2;
    const int c4 = tid % C4, c = c4 * 4;
    const int OH = POOL ? a.H / 2 : a.H, OW = POOL ? a.W / 2 : a.W;
    const int opix = OH * OW;
    const int q0 = (blockIdx.x * BB_THREADS + tid) / C4;
    const int qstep = gridDim.x * (BB_THREADS / C4);
    const float rcpOW = 1.0f / (float)OW;
    double a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    float mu[4], istd[4], sc[4], be[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = a.stats[c + k]; istd[k] = a.stats[C + c + k];
        sc[k] = a.gamma[c + k] * istd[k]; be[k] = a.beta[c + k];
    }
    for (int n = blockIdx.y; n < a.N; n += gridDim.y) {
        const float *zn = a.z + (size_t)n * a.H * a.W * C + c;
        const float *gn = a.dout + (size_t)n * opix * C + c;
        // (round 5) this thread's terms of ONE image are summed in float32 (a handful: opix / qstep of them, x the tie
        // multiplicity) and folded into the float64 sums once per image: the two float64 additions + conversions per
        // element were what kept this pass at 3.5 TB/s where the apply pass streams at 6
        float f1[4] = {0.f, 0.f, 0.f, 0.f}, f2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll(POOL ? 2 : 4)
        for (int q = q0; q < opix; q += qstep) {
            const float4 g4 = *reinterpret_cast<const float4 *>(gn + (size_t)q * C);
            const float g[4] = {g4.x, g4.y, g4.z, g4.w};
            float vbest[4], ybest[4];
            float mult[4] = {1.f, 1.f, 1.f, 1.f};      // window elements that receive the gradient (equal y: equal terms)
            if (POOL && a.zsel) {
                // the forward apply pass stored the selected element: one 16-byte read instead of four
                const float4 v4 = *reinterpret_cast<const float4 *>(a.zsel + ((size_t)n * opix + q) * C + c);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vbest[k] = v[k];
                    ybest[k] = bn_affine(v[k], mu[k], sc[k], be[k]);
                }
                if (a.ztie) {                          // ... and how many elements tie with it (two bits per channel)
                    const unsigned tb = a.ztie[((size_t)n * opix + q) * C4 + c4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) mult[k] = (float)(((tb >> (2 * k)) & 3u) + 1u);
                }
            } else if (POOL) {
                const int oy = fdivb(q, rcpOW), ox = q - oy * OW;
                const float *zp = zn + ((size_t)(2 * oy) * a.W + 2 * ox) * C;
                const float4 w0 = *reinterpret_cast<const float4 *>(zp);
                const float4 w1 = *reinterpret_cast<const float4 *>(zp + C);
                const float4 w2 = *reinterpret_cast<const float4 *>(zp + (size_t)a.W * C);
                const float4 w3 = *reinterpret_cast<const float4 *>(zp + (size_t)a.W * C + C);
                const float v[4][4] = {{w0.x, w0.y, w0.z, w0.w}, {w1.x, w1.y, w1.z, w1.w},
                                       {w2.x, w2.y, w2.z, w2.w}, {w3.x, w3.y, w3.z, w3.w}};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float yv[4];
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) yv[rr] = bn_affine(v[rr][k], mu[k], sc[k], be[k]);
                    ybest[k] = yv[0];
                    vbest[k] = v[0][k];
#pragma unroll
                    for (int rr = 1; rr < 4; ++rr)
                        if (yv[rr] > ybest[k]) { ybest[k] = yv[rr]; vbest[k] = v[rr][k]; }      // strict >: first max
                    if (!a.ties_first) {
                        int cnt = 0;
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) cnt += (yv[rr] == ybest[k]) ? 1 : 0;
                        mult[k] = (float)cnt;
                    }
                }
            } else {
                const float4 v4 = *reinterpret_cast<const float4 *>(zn + (size_t)q * C);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vbest[k] = v[k];
                    ybest[k] = bn_affine(v[k], mu[k], sc[k], be[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dact = (a.elu && ybest[k] <= 0.0f) ? __expf(ybest[k]) : 1.0f;    // ELU'(y) = exp(y), y <= 0
                const float dy = g[k] * dact * mult[k];
                f1[k] += dy;
                f2[k] = fmaf(dy, (vbest[k] - mu[k]) * istd[k], f2[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] += (double)f1[k]; a2[k] += (double)f2[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { s1[tid * 4 + k] = a1[k]; s2[tid * 4 + k] = a2[k]; }
    __syncthreads();
    if (tid < C) {
        const int cc4 = tid >> 2, k = tid & 3;
        double t1 = 0.0, t2 = 0.0;
        for (int t = cc4; t < BB_THREADS; t += C4) { t1 += s1[t * 4 + k]; t2 += s2[t * 4 + k]; }
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        a.partial[(blk * 2) * C + tid] = t1;
        a.partial[(blk * 2 + 1) * C + tid] = t2;
    }
}

// sums[2][C] <- block-ordered sum of the partials; also the BN parameter gradients dbeta = sum dy, dgamma = sum dy xhat
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                            double *__restrict__ sums, float *__restrict__ dbeta,
                                                            float *__restrict__ dgamma) {
    // 1024 threads: slot = one of the 2C values, its threads sum interleaved subsets of the blocks, fixed-order finish
    __shared__ double red[1024];
    const int tid = threadIdx.x;
    const int slots = 2 * C, groups = 1024 / slots;
    const int slot = tid % slots, grp = tid / slots;
    double acc = 0.0;
    if (grp < groups)
        for (int b = grp; b < nblocks; b += groups) acc += partial[(size_t)b * slots + slot];
    red[tid] = (grp < groups) ? acc : 0.0;
    __syncthreads();
    if (tid >= slots) return;
    double t = 0.0;
    for (int gI = 0; gI < groups; ++gI) t += red[gI * slots + tid];
    sums[tid] = t;
    if (tid < C) dbeta[tid] = (float)t; else dgamma[tid - C] = (float)t;
}

// apply pass: dz = gamma s (dy - mean(dy) - xhat mean(dy xhat)).  thread = (window or pixel, 4 channels): a pooled
// block's thread owns the whole 2x2 window (reads its 4 z, writes its 4 dz), plus the odd last row / column that no
// window covers (dy = 0 there, the mean terms still apply).
__global__ __launch_bounds__(BB_THREADS) void bn_bwd_apply_kernel(BnBwdArgs a) {
    const int C = a.C, C4 = C >> 2;
    const int tid = threadIdx.x;
    const int c4 = tid % C4, c = c4 * 4;
    const int OH = a.pool ? a.H / 2 : a.H, OW = a.pool ? a.W / 2 : a.W;
    // pooled: iterate over ceil(H/2) x ceil(W/2) cells so that the uncovered border is visited too
    const int GH = a.pool ? (a.H + 1) / 2 : a.H, GW = a.pool ? (a.W + 1) / 2 : a.W;
    const int cells = GH * GW;
    const int q0 = (blockIdx.x * BB_THREADS + tid) / C4;
    const int qstep = gridDim.x * (BB_THREADS / C4);
    const float rcpGW = 1.0f / (float)GW;
    const double inv_m = 1.0 / ((double)a.Ng * a.H * a.W);
    float mu[4], istd[4], sc[4], be[4], m1[4], m2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = a.stats[c + k]; istd[k] = a.stats[C + c + k];
        sc[k] = a.gamma[c + k] * istd[k]; be[k] = a.beta[c + k];
        m1[k] = (float)(a.sums[c + k] * inv_m); m2[k] = (float)(a.sums[C + c + k] * inv_m);
    }
    for (int n = blockIdx.y; n < a.N; n += gridDim.y) {
        const float *zn = a.z + (size_t)n * a.H * a.W * C + c;
        float *dzn = a.dz + (size_t)n * a.H * a.W * C + c;
        const float *gn = a.dout + (size_t)n * OH * OW * C + c;
#pragma unroll 2
        for (int q = q0; q < cells; q += qstep) {
            if (a.pool) {
                const int gy = fdivb(q, rcpGW), gx = q - gy * GW;
                const bool has_win = gy < OH && gx < OW;
                float g[4] = {0.f, 0.f, 0.f, 0.f};
                if (has_win) {
                    const float4 g4 = *reinterpret_cast<const float4 *>(gn + ((size_t)gy * OW + gx) * C);
                    g[0] = g4.x; g[1] = g4.y; g[2] = g4.z; g[3] = g4.w;
                }
                float v[4][4], yv[4][4];
                bool valid[4];
                float ybest[4] = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int y = 2 * gy + (rr >> 1), x = 2 * gx + (rr & 1);
                    valid[rr] = y < a.H && x < a.W;
                    float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (valid[rr]) v4 = *reinterpret_cast<const float4 *>(zn + ((size_t)y * a.W + x) * C);
                    v[rr][0] = v4.x; v[rr][1] = v4.y; v[rr][2] = v4.z; v[rr][3] = v4.w;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        yv[rr][k] = bn_affine(v[rr][k], mu[k], sc[k], be[k]);
                        if (has_win) ybest[k] = fmaxf(ybest[k], yv[rr][k]);
                    }
                }
                // the window elements that receive the pooled gradient: every one whose y equals the maximum (Theano's CPU
                // MaxPoolGrad) or, ties_first, only the first of them in row-major order
                bool taken[4] = {false, false, false, false};
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    if (!valid[rr]) continue;
                    const int y = 2 * gy + (rr >> 1), x = 2 * gx + (rr & 1);
                    float o[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const bool hit = has_win && yv[rr][k] == ybest[k] && !(a.ties_first && taken[k]);
                        taken[k] = taken[k] || hit;
                        // ELU'(y) = exp(y) for y <= 0; tied elements share y, so one exponential per channel would do
                        const float dact = (a.elu && ybest[k] <= 0.0f) ? __expf(ybest[k]) : 1.0f;
                        const float dy = hit ? g[k] * dact : 0.0f;
                        const float xhat = (v[rr][k] - mu[k]) * istd[k];
                        o[k] = sc[k] * (dy - m1[k] - xhat * m2[k]);
                    }
                    *reinterpret_cast<float4 *>(dzn + ((size_t)y * a.W + x) * C) = make_float4(o[0], o[1], o[2], o[3]);
                }
            } else {
                const size_t off = (size_t)q * C;
                const float4 v4 = *reinterpret_cast<const float4 *>(zn + off);
                const float4 g4 = *reinterpret_cast<const float4 *>(gn + off);
                const float v[4] = {v4.x, v4.y, v4.z, v4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w};
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float yy, d;
                    bn_y(v[k], mu[k], sc[k], be[k], a.elu, yy, d);
                    const float xhat = (v[k] - mu[k]) * istd[k];
                    o[k] = sc[k] * (g[k] * d - m1[k] - xhat * m2[k]);
                }
                *reinterpret_cast<float4 *>(dzn + off) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// reduce grid: bx chunks of an image times by images; bn_bwd_blocks = the most partials a launch can write
// (1024 workgroups of three waves reached 3.4-3.6 TB/s on the unpooled blocks where the apply pass, 8192 workgroups,
// reaches 6; the partial table is pre-summed by colsum_stage_kernel, so its length costs nothing any more)
static const int g_bn_bwd_parts = getenv("ASR_BN_BWD_PARTS") ? std::max(64, std::min(4096, atoi(getenv("ASR_BN_BWD_PARTS")))) : 4096;
static void bn_bwd_grid(int N, int per_img4, int *bx, int *by) {
    *bx = (int)std::max(1, std::min((per_img4 + BB_THREADS - 1) / BB_THREADS, 64));
    *by = std::max(1, std::min(N, g_bn_bwd_parts / *bx));
}
int bn_bwd_blocks(int64_t) { return 4096; }

// NOTE: the apply pass of a pooled block re-reads the neighbours' z, so dz must NOT alias z for pooled blocks.
hipError_t launch_bn_bwd(hipStream_t s, const float *z, float *dz, const float *dout, const float *stats,
                         const float *gamma, const float *beta, double *partial, double *sums, float *dbeta,
                         float *dgamma, int N, int H, int W, int C, int pool, int elu, const Exchange *ex,
                         const float *zsel, const uint8_t *ztie, int ties_first, unsigned *ticket, double *pre_partial,
                         int pre_rows, double *pre_staged) {
    if (C > 128 || C < 4 || C % 4 || BB_THREADS % (C / 4)) return hipErrorInvalidValue;
    if (pool && zsel && !ties_first && !ztie) return hipErrorInvalidValue;      // the multiplicities are not derivable from zsel
    BnBwdArgs a;
    a.zsel = pool ? zsel : nullptr;
    a.ztie = (pool && zsel && !ties_first) ? ztie : nullptr;
    a.ties_first = ties_first;
    a.Ng = ex ? ex->n_global : N;
    a.z = z; a.dz = dz; a.dout = dout; a.stats = stats; a.gamma = gamma; a.beta = beta;
    a.partial = partial; a.sums = sums; a.N = N; a.H = H; a.W = W; a.C = C; a.pool = pool; a.elu = elu;
    const int OH = pool ? H / 2 : H, OW = pool ? W / 2 : W;
    const int phase = ex ? ex->phase : 0;                  // (paired exchange of the data-parallel step, see Exchange)
    if (phase != 2 && pre_partial && pre_rows > 0) {
        if (!ticket || !pre_staged || 2 * C > 256) return hipErrorInvalidValue;
        ColsumFinalArgs f{};
        f.partial = pre_partial; f.nb = pre_rows; f.cols = 2 * C; f.staged = pre_staged; f.ticket = ticket; f.zero_rows = 1;
        f.mode = 2; f.C = C; f.sums = sums; f.dbeta = dbeta; f.dgamma = dgamma;
        const hipError_t fe = launch_colsum_final(s, f);
        if (fe != hipSuccess) return fe;
    } else if (phase != 2) {
        int bx, by;
        bn_bwd_grid(N, OH * OW * (C / 4), &bx, &by);
        if (pool) bn_bwd_reduce_kernel<true><<<dim3(bx, by), BB_THREADS, 0, s>>>(a);
        else bn_bwd_reduce_kernel<false><<<dim3(bx, by), BB_THREADS, 0, s>>>(a);
        // dbeta / dgamma stay LOCAL sums (the gradient all-reduce adds the ranks); the apply pass needs the batch sums
        int nparts = bx * by;
        static const bool fused = (getenv("ASR_TRAIN_FUSED_REDUCE") && getenv("ASR_TRAIN_FUSED_REDUCE")[0] == '1');
        if (ticket && fused && 2 * C <= 256) {        // partial sums -> batch sums + dbeta / dgamma in one launch
            ColsumFinalArgs f{};
            f.partial = partial; f.nb = nparts; f.cols = 2 * C; f.staged = partial + (size_t)nparts * 2 * C; f.ticket = ticket;
            f.mode = 2; f.C = C; f.sums = sums; f.dbeta = dbeta; f.dgamma = dgamma;
            const hipError_t fe = launch_colsum_final(s, f);
            if (fe != hipSuccess) return fe;
        } else {
            const double *ptab = colsum_stage(s, partial, &nparts, 2 * C);
            bn_bwd_final_kernel<<<1, 1024, 0, s>>>(ptab, nparts, C, sums, dbeta, dgamma);
        }
    }
    if (phase == 0 && ex && ex->allreduce_f64(ex->self, s, sums, 2 * C) != 0) return hipErrorUnknown;
    if (phase == 1) return hipGetLastError();
    if (dz == nullptr) return hipGetLastError();          // the consumer applies dz itself (block 1: conv1_wgrad_kernel)
    const int GH = pool ? (H + 1) / 2 : H, GW = pool ? (W + 1) / 2 : W;
    const int cells4 = GH * GW * (C / 4);
    const int ax = (int)std::max(1, std::min((cells4 + BB_THREADS - 1) / BB_THREADS, 64));
    const int ay = std::max(1, std::min(N, 8192 / ax));
    bn_bwd_apply_kernel<<<dim3(ax, ay), BB_THREADS, 0, s>>>(a);
    return hipGetLastError();
}

// Block 1 without its raw tensor (see conv1_raw_kernel, MODE 1 / 2): the reduce pass of block 1's BatchNorm backward
// recomputes z from the input image - nine taps, the same FMA chain as the forward kernel - instead of reading it.
// thread = one pixel, all channels; float64 sums per thread, across the wave by shuffles, across the four waves through
// LDS in wave order; one partial row [2][C] per workgroup.
template <int COUT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_conv1_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                                  const float *__restrict__ dout,
                                                                  const float *__restrict__ stats,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta, int N, int H, int W,
                                                                  double *__restrict__ partial) {
    __shared__ double red[4][2 * COUT];
    __shared__ float kc[4 * COUT];                         // mu, istd, sc, be
    for (int c = threadIdx.x; c < COUT; c += 256) {
        const float mu = stats[c], istd = stats[COUT + c];
        kc[c] = mu; kc[COUT + c] = istd; kc[2 * COUT + c] = gamma[c] * istd; kc[3 * COUT + c] = beta[c];
    }
    __syncthreads();
    double a1[COUT], a2[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) { a1[c] = 0.0; a2[c] = 0.0; }
    const int64_t total = (int64_t)N * H * W;
    const bool small = total < ((int64_t)1 << 31);
#pragma unroll 1
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (int64_t)gridDim.x * blockDim.x) {
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
                v[a * 3 + b] = xn[yc * W + xc] * ((yy == yc && xb == xc) ? 1.0f : 0.0f);
            }
        const float4 *g4 = reinterpret_cast<const float4 *>(dout + (size_t)s * COUT);
#pragma unroll
        for (int o4 = 0; o4 < COUT / 4; ++o4) {
            const float4 gq = g4[o4];
            const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = o4 * 4 + j;
                float z = 0.0f;
#pragma unroll
                for (int t = 0; t < 9; ++t) z = fmaf(v[t], w[c * 9 + t], z);        // conv1_raw_kernel's chain
                const float mu = kc[c], istd = kc[COUT + c];
                const float yv = (z - mu) * kc[2 * COUT + c] + kc[3 * COUT + c];
                const float dact = yv <= 0.0f ? __expf(yv) : 1.0f;                  // ELU'(y) = exp(y), y <= 0
                const double dy = (double)(gv[j] * dact);
                a1[c] += dy;
                a2[c] += dy * (double)((z - mu) * istd);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        double s1 = a1[c], s2 = a2[c];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
        if (lane == 0) { red[wave][c] = s1; red[wave][COUT + c] = s2; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * COUT)
        partial[(size_t)blockIdx.x * 2 * COUT + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// reduce pass + batch sums (+ their all-reduce) of block 1's BatchNorm backward from the INPUT image; the apply pass
// lives in conv1_wgrad_kernel
hipError_t launch_bn_bwd_conv1(hipStream_t s, const float *x, const float *w, const float *dout, const float *stats,
                               const float *gamma, const float *beta, double *partial, double *sums, float *dbeta,
                               float *dgamma, int N, int H, int W, int C, const Exchange *ex) {
    if (ex && ex->phase == 2) return hipSuccess;           // (nothing follows this block's all-reduce here)
    const int64_t total = (int64_t)N * H * W;
    int nparts = (int)std::max<int64_t>(1, std::min<int64_t>((total + 255) / 256, 2048));
    if (C == 12) bn_bwd_reduce_conv1_kernel<12><<<nparts, 256, 0, s>>>(x, w, dout, stats, gamma, beta, N, H, W, partial);
    else if (C == 24) bn_bwd_reduce_conv1_kernel<24><<<nparts, 256, 0, s>>>(x, w, dout, stats, gamma, beta, N, H, W, partial);
    else return hipErrorInvalidValue;
    const double *ptab = colsum_stage(s, partial, &nparts, 2 * C);
    bn_bwd_final_kernel<<<1, 1024, 0, s>>>(ptab, nparts, C, sums, dbeta, dgamma);
    if (ex && ex->phase == 0 && ex->allreduce_f64(ex->self, s, sums, 2 * C) != 0) return hipErrorUnknown;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// weight gradient on MFMA
// ---------------------------------------------------------------------------
__host__ __device__ constexpr int wg_stride(int c) {     // LDS pixel stride: multiple of 16, == 16 (mod 32)
    return ((c + 15) / 16 * 16) % 32 == 0 ? (c + 15) / 16 * 16 + 16 : (c + 15) / 16 * 16;
}

struct WgradArgs {
    const float *x;      // (N,H,W,CIN)  block input
    const float *dz;     // (N,H,W,COUT) gradient wrt the raw conv output
    float *partial;      // [gridDim.x][9][CIN][COUT]
    int N, H, W;
    int TH, TW;          // tile (TW multiple of 4)
    int tiles_y, tiles_x, total_tiles;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(576) void wgrad_mfma_kernel(WgradArgs a) {
    constexpr int MI = (CIN + 15) / 16, NJ = (COUT + 15) / 16;
    constexpr int CSX = wg_stride(CIN), CSZ = wg_stride(COUT);
    constexpr int THREADS = 576;                       // 9 waves: wave = tap
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, tap = tid >> 6;
    const int g = lane >> 4, nn = lane & 15;
    const int ta = tap / 3, tb = tap % 3;
    const int LW = a.TW + 2, LH = a.TH + 2;
    float *xs = lds;                                   // [LH*LW][CSX]
    float *zs = lds + (size_t)LH * LW * CSX;           // [TH*TW][CSZ]
    floatx4 acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nxv = LH * LW * (CSX / 4), nzv = a.TH * a.TW * (CSZ / 4);
    const int kgroups = a.TW >> 2;
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        __syncthreads();                               // previous tile fully consumed
        for (int e = tid; e < nxv; e += THREADS) {
            const int c4 = e % (CSX / 4);
            const int p = e / (CSX / 4);
            const int col = p % LW, row = p / LW;
            const int gy = y0 + row - 1, gx = x0 + col - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 * 4 < CIN && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                v = *reinterpret_cast<const float4 *>(a.x + (((size_t)n * a.H + gy) * a.W + gx) * CIN + c4 * 4);
            *reinterpret_cast<float4 *>(xs + (size_t)p * CSX + c4 * 4) = v;
        }
        for (int e = tid; e < nzv; e += THREADS) {
            const int c4 = e % (CSZ / 4);
            const int p = e / (CSZ / 4);
            const int col = p % a.TW, row = p / a.TW;
            const int gy = y0 + row, gx = x0 + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 * 4 < COUT && gy < a.H && gx < a.W)
                v = *reinterpret_cast<const float4 *>(a.dz + (((size_t)n * a.H + gy) * a.W + gx) * COUT + c4 * 4);
            *reinterpret_cast<float4 *>(zs + (size_t)p * CSZ + c4 * 4) = v;
        }
        __syncthreads();
        for (int row = 0; row < a.TH; ++row) {
            const float *xr = xs + (size_t)((row + ta) * LW + tb + g) * CSX + nn;     // A: (ci = nn, pixel k = g)
            const float *zr = zs + (size_t)(row * a.TW + g) * CSZ + nn;               // B: (pixel k = g, co = nn)
            for (int kg = 0; kg < kgroups; ++kg) {
                float af[MI], bf[NJ];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) af[mi] = xr[(size_t)kg * 4 * CSX + mi * 16];
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) bf[nj] = zr[(size_t)kg * 4 * CSZ + nj * 16];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
            }
        }
    }
    // C/D layout: lane (g, nn) holds rows ci = mi*16 + 4g + r, column co = nj*16 + nn
    float *out = a.partial + ((size_t)blockIdx.x * 9 + tap) * CIN * COUT;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = mi * 16 + 4 * g + r, co = nj * 16 + nn;
                if (ci < CIN && co < COUT) out[(size_t)ci * COUT + co] = acc[mi][nj][r];
            }
}

// Small-channel form (C_in <= 24): the GEMM rows are (tap, ci) PACKED - m = tap * C_in + ci, ceil(9 C_in / 16)
// m-tiles instead of 9 padded ones (C_in = 12: 7 tiles for 108 rows, 96 % of the rows useful instead of 75 %) - and
// one B fragment (4 pixels x 16 C_out) feeds all of a wave's m-tiles: (MT + NJ) LDS reads per MT * NJ MFMAs instead
// of 2 per MFMA.  The waves of a workgroup split the tile rows (K) and keep all MT x NJ accumulators; they are summed
// through LDS once, after the persistent tile loop.  Several workgroups per CU overlap staging with the MFMA loops.
template <int CIN, int COUT, int WAVES, int RX, int RZ>
__global__ __launch_bounds__(64 * WAVES, 2) void wgrad_taps_kernel(WgradArgs a) {
    constexpr int MROWS = 9 * CIN;
    constexpr int MT = (MROWS + 15) / 16, NJ = (COUT + 15) / 16;
    constexpr int CSX = wg_stride(CIN), CSZ = wg_stride(COUT);
    constexpr int THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, nn = lane & 15;
    const int LW = a.TW + 2, LH = a.TH + 2;
    float *xs = lds;                                   // [LH*LW][CSX]
    float *zs = lds + (size_t)LH * LW * CSX;           // [TH*TW][CSZ]
    floatx4 acc[MT][NJ];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) acc[mt][nj] = floatx4{0.f, 0.f, 0.f, 0.f};
    // A fragment of m-tile mt: lane (row m = mt*16 + nn, k = pixel g) reads x[(r + dy) * LW + c + dx + g][ci]
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int m = mt * 16 + nn;
        m = m < MROWS ? m : MROWS - 1;                 // padding rows: any valid address, their results are dropped
        const int tap = m / CIN, ci = m - tap * CIN;
        aoff[mt] = ((tap / 3) * LW + (tap % 3) + g) * CSX + ci;
    }
    const int boff = g * CSZ + nn;                     // B: (pixel k = g, co = nn)
    // tile-independent staging tables (float4 elements)
    constexpr int X4 = CSX / 4, Z4 = CSZ / 4;
    const int nxv = LH * LW * X4, nzv = a.TH * a.TW * Z4;
    int sx_lds[RX], sx_g[RX], sx_meta[RX], sz_lds[RZ], sz_g[RZ], sz_meta[RZ];
#pragma unroll
    for (int r = 0; r < RX; ++r) {
        const int e = tid + r * THREADS;
        sx_lds[r] = -1; sx_g[r] = 0; sx_meta[r] = 0;
        if (e < nxv) {
            const int c4 = e % X4, p = e / X4;
            const int col = p % LW, row = p / LW;
            sx_lds[r] = p * CSX + c4 * 4;
            sx_g[r] = c4 * 4 < CIN ? (row * a.W + col) * CIN + c4 * 4 : -1;      // -1: padding channels, zero
            sx_meta[r] = row | (col << 8);
        }
    }
#pragma unroll
    for (int r = 0; r < RZ; ++r) {
        const int e = tid + r * THREADS;
        sz_lds[r] = -1; sz_g[r] = 0; sz_meta[r] = 0;
        if (e < nzv) {
            const int c4 = e % Z4, p = e / Z4;
            const int col = p % a.TW, row = p / a.TW;
            sz_lds[r] = p * CSZ + c4 * 4;
            sz_g[r] = c4 * 4 < COUT ? (row * a.W + col) * COUT + c4 * 4 : -1;
            sz_meta[r] = row | (col << 8);
        }
    }
    const int kgroups = a.TW >> 2;
    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        const float *xb = a.x + ((int64_t)((int64_t)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * CIN;
        const float *zb = a.dz + ((int64_t)((int64_t)n * a.H + y0) * a.W + x0) * COUT;
        const int ylo = 1 - y0, yhi = a.H + 1 - y0, xlo = 1 - x0, xhi = a.W + 1 - x0;
        float4 vx[RX], vz[RZ];
#pragma unroll
        for (int r = 0; r < RX; ++r) {
            const int row = sx_meta[r] & 255, col = sx_meta[r] >> 8;
            vx[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sx_lds[r] >= 0 && sx_g[r] >= 0 && row >= ylo && row < yhi && col >= xlo && col < xhi)
                vx[r] = *reinterpret_cast<const float4 *>(xb + sx_g[r]);
        }
#pragma unroll
        for (int r = 0; r < RZ; ++r) {
            const int row = sz_meta[r] & 255, col = sz_meta[r] >> 8;
            vz[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sz_lds[r] >= 0 && sz_g[r] >= 0 && row < a.H - y0 && col < a.W - x0)
                vz[r] = *reinterpret_cast<const float4 *>(zb + sz_g[r]);
        }
        __syncthreads();                               // previous tile fully consumed
#pragma unroll
        for (int r = 0; r < RX; ++r)
            if (sx_lds[r] >= 0) *reinterpret_cast<float4 *>(xs + sx_lds[r]) = vx[r];
#pragma unroll
        for (int r = 0; r < RZ; ++r)
            if (sz_lds[r] >= 0) *reinterpret_cast<float4 *>(zs + sz_lds[r]) = vz[r];
        __syncthreads();
        for (int row = wave; row < a.TH; row += WAVES) {
            const float *xr = xs + (size_t)row * LW * CSX;
            const float *zr = zs + (size_t)row * a.TW * CSZ + boff;
            for (int kg = 0; kg < kgroups; ++kg) {
                float af[MT], bf[NJ];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) af[mt] = xr[aoff[mt] + kg * 4 * CSX];
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj) bf[nj] = zr[kg * 4 * CSZ + nj * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
                        acc[mt][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nj], acc[mt][nj], 0, 0, 0);
            }
        }
    }
    // ---- sum the waves' accumulators in one LDS slab, wave after wave (fixed order), then one partial per
    // workgroup.  C/D layout: lane (g, nn) holds rows m = mt*16 + 4g + r, column co = nj*16 + nn
    __syncthreads();
    constexpr int SLAB = MT * NJ * 256;
    float *red = lds;
    for (int w = 0; w < WAVES; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *q = red + ((mt * NJ + nj) * 4 + r) * 64 + lane;
                        *q = (w == 0) ? acc[mt][nj][r] : *q + acc[mt][nj][r];
                    }
        }
        __syncthreads();
    }
    float *out = a.partial + (size_t)blockIdx.x * MROWS * COUT;
    for (int e = tid; e < SLAB; e += THREADS) {
        const int l = e & 63, r = (e >> 6) & 3, tile_id = e >> 8;
        const int mt = tile_id / NJ, nj = tile_id - mt * NJ;
        const int m = mt * 16 + 4 * (l >> 4) + r, co = nj * 16 + (l & 15);
        if (m < MROWS && co < COUT) out[(size_t)m * COUT + co] = red[e];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad_dma_kernel: the packed-(tap, ci) GEMM of wgrad_taps_kernel for every block, with the tiles DOUBLE-BUFFERED in
// LDS by LDS-DMA (global_load_lds_dwordx4: no data registers; tile i+1 is in flight while tile i multiplies; one
// workgroup barrier per tile).  The first two forms staged global -> registers -> LDS between two barriers with
// nothing in flight during the MFMA loop: the matrix pipe was busy about half of the time (36-54 % of the fp32 peak).
// The waves of a workgroup form a WM x WK grid: wave (wm, wk) owns m-tiles [wm MPW, (wm+1) MPW) of the 9 C_in packed
// rows (all n-tiles) and the tile rows wk, wk + WK, ... (K split); the WK copies are summed through LDS once, after
// the persistent tile loop, in wave order.  48 -> 48: 27 m-tiles as 4 x 7 (the wave-per-tap form put 9 waves on 4 SIMDs).
// LDS element e (16 bytes) of a buffer = (pixel p, channel group c4) of the x tile (halo 1), then of the dz tile,
// each region rounded up to a whole wave (64 elements): a wave's DMA lands 64 consecutive elements, every lane picks
// its own source; out-of-image pixels and padding channels read a zero block.
__device__ float4 g_wgrad_zero[4];

// One LDS-DMA of 16 bytes per lane: lane i's data lands at LDS byte address lds_addr + 16 i (lds_addr wave-uniform).
// Issued as inline assembly ON PURPOSE: behind the builtin the compiler cannot tell the buffer being filled from the
// buffer being read and puts s_waitcnt vmcnt(0) in front of the first LDS read after it - the copy of tile i+1 then
// completes before tile i is multiplied and nothing overlaps.  The kernel waits itself (dma_wait) before the barrier
// that publishes a buffer; copies the compiler does not know of only make its own vmcnt waits stricter.
__device__ __forceinline__ void lds_dma16(const float *src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const float *p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}

template <int CIN, int COUT, int WM, int WK, int RX, int RZ>
__global__ __launch_bounds__(64 * WM * WK) void wgrad_dma_kernel(WgradArgs a) {
    constexpr int WAVES = WM * WK, THREADS = 64 * WAVES;
    // The 9 C_in packed rows are WM units of 108 (C_in = 12 WM); wave column wm owns unit wm as SEVEN m-tiles whose rows
    // are interleaved so that one wide LDS read feeds several tiles: lane nn reads rows 4nn..4nn+3 of the unit with one
    // ds_read_b128 (tiles 0-3: tile j holds rows 4 nn + j), rows 64 + 2nn, + 1 with one ds_read_b64 (tiles 4, 5) and
    // row 96 + nn with one ds_read_b32 (tile 6, 12 rows used) - three reads and three address adds per step instead
    // of seven.  Likewise the n-tiles hold the output channels NJ nn + nj: one read of NJ consecutive floats.
    constexpr int MROWS = 9 * CIN, MPW = 7, NJ = (COUT + 15) / 16;
    static_assert(CIN == 12 * WM, "one unit of 108 packed rows per wave column");
    // pixel strides WITHOUT padding: the fragment reads (4 pixels of stride 12 / 24 / 48 floats) have two-way bank
    // conflicts on a few lanes, which costs less than copying and storing 33-100 % padding per tile
    constexpr int CSX = CIN, CSZ = COUT, X4 = CSX / 4, Z4 = CSZ / 4;
    static_assert(CIN % 4 == 0 && COUT % 4 == 0, "whole float4 channel groups");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wk = wave / WM;
    const int g = lane >> 4, nn = lane & 15;
    const int LW = a.TW + 2, LH = a.TH + 2;
    const int nxv = LH * LW * X4, nzv = a.TH * a.TW * Z4;
    const int nxp = (nxv + 63) & ~63, nzp = (nzv + 63) & ~63;
    const int buf_floats = (nxp + nzp) * 4;
    floatx4 acc[MPW][NJ];
#pragma unroll
    for (int i = 0; i < MPW; ++i)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) acc[i][nj] = floatx4{0.f, 0.f, 0.f, 0.f};
    // A operand of packed row R = tap * C_in + ci, k = pixel g: x[(r + tap / 3) * LW + c + tap % 3 + g][ci]
    auto a_off = [&](int rho) {
        const int R = 108 * wm + (rho < 108 ? rho : 107);      // rows past the unit: any valid address, results dropped
        const int tap = R / CIN, ci = R - tap * CIN;
        return ((tap / 3) * LW + (tap % 3) + g) * CSX + ci;
    };
    const int ao4 = a_off(4 * nn), ao2 = a_off(64 + 2 * nn), ao1 = a_off(96 + nn);
    const int bo = g * CSZ + (NJ * nn + NJ <= COUT ? NJ * nn : 0);   // B: (pixel k = g, channels NJ nn .. NJ nn + NJ - 1)
    // which elements this thread moves (the same for every tile): offset from the tile's first (halo) pixel in global
    // memory and row | col << 8 for the border test (-1: past the end of the region)
    int sx[RX], sxg[RX], sz[RZ], szg[RZ];
#pragma unroll
    for (int r = 0; r < RX; ++r) {
        const int e = tid + r * THREADS;
        const int c4 = e % X4, p = e / X4;
        const int col = p % LW, row = p / LW;
        sx[r] = e < nxv ? (row | (col << 8)) : -1;
        sxg[r] = e < nxv ? (row * a.W + col) * CIN + c4 * 4 : (a.W + 1) * CIN;   // (padding: the tile's first pixel)
    }
#pragma unroll
    for (int r = 0; r < RZ; ++r) {
        const int e = tid + r * THREADS;
        const int c4 = e % Z4, p = e / Z4;
        const int col = p % a.TW, row = p / a.TW;
        sz[r] = e < nzv ? (row | (col << 8)) : -1;
        szg[r] = e < nzv ? (row * a.W + col) * COUT + c4 * 4 : 0;
    }
    const float *zero = reinterpret_cast<const float *>(g_wgrad_zero);
    const unsigned lds0 = lds_addr_of(lds);
    auto stage = [&](int tile, int which) {
        const unsigned buf_addr = lds0 + (unsigned)(which * buf_floats) * 4u;
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        const float *xb = a.x + ((int64_t)((int64_t)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * CIN;
        const float *zb = a.dz + ((int64_t)((int64_t)n * a.H + y0) * a.W + x0) * COUT;
        const int ylo = 1 - y0, yhi = a.H + 1 - y0, xlo = 1 - x0, xhi = a.W + 1 - x0;
        if (y0 >= 1 && y0 + a.TH + 1 <= a.H && x0 >= 1 && x0 + a.TW + 1 <= a.W) {
            // tile and halo inside the image (most tiles): one address add per copy.  (The border tests below cost ~20
            // vector instructions per copy - as much SIMD time as a fifth of the tile's MFMAs, which they do not overlap.)
            // Elements past the end of a region land in its padding, which nothing reads: any valid source will do
#pragma unroll
            for (int r = 0; r < RX; ++r) {
                if (r * THREADS + wave * 64 >= nxp) break;             // wave-uniform
                int sg = sxg[r];
                asm volatile("" : "+v"(sg));
                lds_dma16(xb + sg, buf_addr + (unsigned)(r * THREADS + wave * 64) * 16u);
            }
#pragma unroll
            for (int r = 0; r < RZ; ++r) {
                if (r * THREADS + wave * 64 >= nzp) break;
                int sg = szg[r];
                asm volatile("" : "+v"(sg));
                lds_dma16(zb + sg, buf_addr + (unsigned)(nxp + r * THREADS + wave * 64) * 16u);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < RX; ++r) {
            if (r * THREADS + wave * 64 >= nxp) break;                 // wave-uniform
            int se = sx[r], sg = sxg[r];
            asm volatile("" : "+v"(se), "+v"(sg));   // keep the address arithmetic HERE (hoisted out of the tile loop
                                                     // it occupied ~50 registers for the whole kernel)
            const int row = se & 255, col = se >> 8;
            const bool ok = se >= 0 && row >= ylo && row < yhi && col >= xlo && col < xhi;
            const float *src = ok ? xb + sg : zero;
            lds_dma16(src, buf_addr + (unsigned)(r * THREADS + wave * 64) * 16u);
        }
#pragma unroll
        for (int r = 0; r < RZ; ++r) {
            if (r * THREADS + wave * 64 >= nzp) break;
            int se = sz[r], sg = szg[r];
            asm volatile("" : "+v"(se), "+v"(sg));
            const int row = se & 255, col = se >> 8;
            const bool ok = se >= 0 && row < a.H - y0 && col < a.W - x0;
            const float *src = ok ? zb + sg : zero;
            lds_dma16(src, buf_addr + (unsigned)(nxp + r * THREADS + wave * 64) * 16u);
        }
    };
    const int kgroups = a.TW >> 2;
    int tile = blockIdx.x;
    if (tile < a.total_tiles) stage(tile, 0);
    for (int cur = 0; tile < a.total_tiles; tile += gridDim.x, cur ^= 1) {
        dma_wait();                // this wave's copies of this tile have landed ...
        __syncthreads();           // ... and everybody's; the other buffer has been consumed
        if (tile + (int)gridDim.x < a.total_tiles) stage(tile + gridDim.x, cur ^ 1);
        const float *xs = lds + cur * buf_floats;
        const float *zs = xs + nxp * 4 + bo;
        // this wave's (row, k-group) steps as ONE software-pipelined loop without branches inside: the fragments of step
        // s + 1 are read from LDS before the MFMAs of step s are issued (two register sets, two steps per iteration; the
        // last iteration re-reads its own step instead of running past the end), so every wait is a counted lgkmcnt
        const int steps = ((a.TH - wk + WK - 1) / WK) * kgroups;
        int xo = wk * LW * CSX, zo = wk * a.TW * CSZ, kg = 0;
        auto advance_if = [&](bool c) {                        // uniform: selects, no control flow
            const bool wrap = kg + 1 == kgroups;
            const int dx = 4 * CSX + (wrap ? (WK * LW - a.TW) * CSX : 0);
            const int dzz = 4 * CSZ + (wrap ? (WK - 1) * a.TW * CSZ : 0);
            xo += c ? dx : 0; zo += c ? dzz : 0;
            kg = c ? (wrap ? 0 : kg + 1) : kg;
        };
        float af[2][MPW], bf[2][NJ];
        auto frag = [&](int set) {
            const float4 v4 = *reinterpret_cast<const float4 *>(xs + xo + ao4);
            const float2 v2 = *reinterpret_cast<const float2 *>(xs + xo + ao2);
            af[set][0] = v4.x; af[set][1] = v4.y; af[set][2] = v4.z; af[set][3] = v4.w;
            af[set][4] = v2.x; af[set][5] = v2.y;
            af[set][6] = xs[xo + ao1];
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) bf[set][nj] = zs[zo + nj];
        };
        auto mul = [&](int set) {
#pragma unroll
            for (int i = 0; i < MPW; ++i)
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
                    acc[i][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[set][i], bf[set][nj], acc[i][nj], 0, 0, 0);
        };
        if (steps > 0) {
            frag(0);
            for (int pr = 0; pr < (steps >> 1); ++pr) {
                advance_if(true);
                frag(1);
                __builtin_amdgcn_sched_barrier(0);             // (the scheduler otherwise sinks every read to its use)
                mul(0);
                __builtin_amdgcn_sched_barrier(0);
                advance_if(2 * pr + 2 < steps);
                frag(0);
                __builtin_amdgcn_sched_barrier(0);
                mul(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (steps & 1) mul(0);
        }
    }
    // ---- sum the WK copies through LDS, one m-tile row of every wm at a time (a slab of NJ x 256 floats per wm), wave
    // after wave (fixed order), then one partial per workgroup.  C/D layout: lane (g, nn) holds rows m = mt*16 + 4g + r,
    // column co = nj*16 + nn
    __syncthreads();
    constexpr int SLAB = NJ * 256;                             // floats per wm
    float *red = lds + wm * SLAB;
    float *out = a.partial + (size_t)blockIdx.x * MROWS * COUT;
#pragma unroll
    for (int i = 0; i < MPW; ++i) {
        for (int w = 0; w < WK; ++w) {
            if (wk == w) {
#pragma unroll
                for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float *q = red + (nj * 4 + r) * 64 + lane;
                        *q = (w == 0) ? acc[i][nj][r] : *q + acc[i][nj][r];
                    }
            }
            __syncthreads();
        }
        for (int e = tid; e < WM * SLAB; e += THREADS) {
            const int l = e & 63, r = (e >> 6) & 3, tile_id = e >> 8;
            const int wmm = tile_id / NJ, nj = tile_id - wmm * NJ;
            const int rr = 4 * (l >> 4) + r;                   // row of the tile; its row of the unit:
            const int rho = i < 4 ? 4 * rr + i : i < 6 ? 64 + 2 * rr + (i - 4) : 96 + rr;
            const int m = 108 * wmm + rho, co = NJ * (l & 15) + nj;
            if (rho < 108 && NJ * (l & 15) + NJ <= COUT) out[(size_t)m * COUT + co] = lds[e];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad_wino_kernel: the weight gradient as Winograd F(3x3, 2x2) - the 3x3 gradient of a 2x2 block of dz pixels and the
// 4x4 block of x under it takes 16 products per channel pair instead of 36:
//     dWc = A^T [ sum over 2x2 blocks and images of (G D G^T) .* (B^T X B) ] A
//   B^T = the forward F(2x2,3x3) input transform (rows x0-x2, x1+x2, x2-x1, x1-x3), G = [1 0; 1 1; 1 -1; 0 -1],
//   A^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]            (the F(2,3) matrices with the roles of G and A^T exchanged)
// Per transform position (u, v) this is a GEMM  S_uv[ci][co] += Xt_uv[ci][block] Dt_uv[block][co]  with K = blocks:
// the fp32 MFMA's k index is the block (lane group g = one of four horizontally adjacent blocks), m = ci, n = co, so a
// lane (g, nn) transforms ITS block's 4x4 patch of channel nn - the transforms never leave the lane, as in the forward
// Winograd kernels.  MFMAs per 16 pixels: 16 x ceil(C_in/16) x ceil(C_out/16) against 4 x ceil(9 C_in / 16) x
// ceil(C_out/16) of the packed-taps form (48 -> 48: 144 against 324).
// The waves of a workgroup form a 4 x WK grid: wave (u, wk) owns transform row u - it reads only the two patch rows and
// the dz row(s) that row u of B^T X and of G D needs, 8 + 4 LDS reads and as many vector operations per channel group
// and step - and the block rows wk, wk + WK, ... of a tile.  Tiles, their staging by LDS-DMA and the split over
// workgroups are wgrad_dma_kernel's; at the end the WK copies are summed and the 16 positions of every channel pair are
// brought together through LDS (16 KB: the K copies are added in wave order), transformed (A^T S A) and written as one
// partial in the SAME [tap][ci][co] layout, which wgrad_reduce_kernel sums.  TH even, TW a multiple of 8.
template <int CIN, int COUT, int PU, int WK, int RX, int RZ>
__global__ __launch_bounds__(64 * (4 / PU) * WK) void wgrad_wino_kernel(WgradArgs a) {
    // PU = 1: a wave owns one transform row u (4 positions); PU = 4: all sixteen positions (narrow blocks: the full
    // 4x4 transform costs half the LDS reads per position, and four accumulator rows still fit the registers)
    static_assert(PU == 1 || PU == 4, "one transform row or all four per wave");
    constexpr int UW = 4 / PU, WAVES = UW * WK, THREADS = 64 * WAVES;
    constexpr int NA = (CIN + 15) / 16, NB = (COUT + 15) / 16;
    constexpr int X4 = CIN / 4, Z4 = COUT / 4;
    static_assert(CIN % 4 == 0 && COUT % 4 == 0, "whole float4 channel groups");
    // LDS bank plan.  A ds_read_b32 is served in two groups of 32 lanes over 32 banks: lanes (g, nn) and (g + 1, nn) of a
    // group read the same channels of two pixels TWO COLUMNS apart, so the distance of those pixels in LDS, counted in
    // 16-byte chunks modulo 8, has to leave room for the 3 or 4 chunks a lane group covers.  With the pixels of a row
    // in image order that distance is 2 * C/4: fine for 24 channels (12 -> 4), a two-way conflict on every read for 12
    // (6: banks 24..35 over 0..11) and 48 (24 -> 0) - conflict share 0.50 / 0.49 of the LDS cycles in round 3's PMC
    // run, 0.33 where only one operand has such a count, 0.012 for <24, 24>.  The LDS-DMA writes lane-linear, but every
    // lane chooses its SOURCE: rows are stored with the even columns first, then the odd ones (XEO / ZEO) - two
    // columns apart becomes one pixel apart, C/4 chunks: 3 for 12 channels, 12 -> 4 for 48.  No padding, no extra traffic.
    constexpr bool XEO = (2 * X4) % 8 != 4 && (X4 % 8 >= 3 && X4 % 8 <= 5);
    constexpr bool ZEO = (2 * Z4) % 8 != 4 && (Z4 % 8 >= 3 && Z4 % 8 <= 5);
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = PU == 4 ? 0 : (wave & 3), wk = wave / UW;
    const int g = lane >> 4, nn = lane & 15;
    const int LW = a.TW + 2, LH = a.TH + 2;
    const int nxv = LH * LW * X4, nzv = a.TH * a.TW * Z4;
    const int nxp = (nxv + 63) & ~63, nzp = (nzv + 63) & ~63;
    const int buf_floats = (nxp + nzp) * 4;
    floatx4 acc[4 * PU][NA][NB];
#pragma unroll
    for (int v = 0; v < 4 * PU; ++v)
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[v][i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // row u of B^T X = X[ra] + sx X[rb]; row u of G D = d0 D[0] + d1 D[1]   (wave-uniform: no branches in the loop)
    const int ra = u == 0 ? 0 : u == 2 ? 2 : 1, rb = u == 3 ? 3 : u == 2 ? 1 : 2;
    const float sx = u == 1 ? 1.0f : -1.0f;
    const float d0 = u == 3 ? 0.0f : 1.0f, d1 = u == 0 ? 0.0f : u == 1 ? 1.0f : -1.0f;
    // this lane's channel of every channel group (clamped: the padding rows / columns of a last group compute on a
    // valid address and are dropped at the end)
    int ca[NA], cb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) ca[i] = min(16 * i + nn, CIN - 1);
#pragma unroll
    for (int j = 0; j < NB; ++j) cb[j] = min(16 * j + nn, COUT - 1);
    const int hx = LW >> 1, hz = a.TW >> 1;              // even columns first, then the odd ones (LW, TW are even)
    const int xlane = ((PU == 4 ? 0 : ra) * LW + (XEO ? g : 2 * g)) * CIN, xrow2 = (rb - ra) * LW * CIN;
    const int zlane = (ZEO ? g : 2 * g) * COUT;
    // patch column c (0..3) / dz column (0, 1) of this lane's block, relative to its first column
    int xcol[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xcol[c] = (XEO ? (c >> 1) + (c & 1) * hx : c) * CIN;
    const int zcol1 = (ZEO ? hz : 1) * COUT;
    constexpr int XSTEP = XEO ? 4 : 8, ZSTEP = ZEO ? 4 : 8;   // LDS pixels between two groups of four blocks
    // ---- staging: identical to wgrad_dma_kernel's (see there)
    int sxe[RX], sxg[RX], sze[RZ], szg[RZ];
#pragma unroll
    for (int r = 0; r < RX; ++r) {
        const int e = tid + r * THREADS;
        const int c4 = e % X4, p = e / X4;
        const int scol = p % LW, row = p / LW;             // LDS slot -> image column of the patch
        const int col = XEO ? (scol < hx ? 2 * scol : 2 * (scol - hx) + 1) : scol;
        sxe[r] = e < nxv ? (row | (col << 8)) : -1;
        sxg[r] = e < nxv ? (row * a.W + col) * CIN + c4 * 4 : (a.W + 1) * CIN;
    }
#pragma unroll
    for (int r = 0; r < RZ; ++r) {
        const int e = tid + r * THREADS;
        const int c4 = e % Z4, p = e / Z4;
        const int scol = p % a.TW, row = p / a.TW;
        const int col = ZEO ? (scol < hz ? 2 * scol : 2 * (scol - hz) + 1) : scol;
        sze[r] = e < nzv ? (row | (col << 8)) : -1;
        szg[r] = e < nzv ? (row * a.W + col) * COUT + c4 * 4 : 0;
    }
    const float *zero = reinterpret_cast<const float *>(g_wgrad_zero);
    const unsigned lds0 = lds_addr_of(lds);
    auto stage = [&](int tile, int which) {
        const unsigned buf_addr = lds0 + (unsigned)(which * buf_floats) * 4u;
        const int tx = tile % a.tiles_x;
        const int t2 = tile / a.tiles_x;
        const int ty = t2 % a.tiles_y;
        const int n = t2 / a.tiles_y;
        const int y0 = ty * a.TH, x0 = tx * a.TW;
        const float *xb = a.x + ((int64_t)((int64_t)n * a.H + (y0 - 1)) * a.W + (x0 - 1)) * CIN;
        const float *zb = a.dz + ((int64_t)((int64_t)n * a.H + y0) * a.W + x0) * COUT;
        const int ylo = 1 - y0, yhi = a.H + 1 - y0, xlo = 1 - x0, xhi = a.W + 1 - x0;
        if (y0 >= 1 && y0 + a.TH + 1 <= a.H && x0 >= 1 && x0 + a.TW + 1 <= a.W) {
#pragma unroll
            for (int r = 0; r < RX; ++r) {
                if (r * THREADS + wave * 64 >= nxp) break;             // wave-uniform
                int sg = sxg[r];
                asm volatile("" : "+v"(sg));
                lds_dma16(xb + sg, buf_addr + (unsigned)(r * THREADS + wave * 64) * 16u);
            }
#pragma unroll
            for (int r = 0; r < RZ; ++r) {
                if (r * THREADS + wave * 64 >= nzp) break;
                int sg = szg[r];
                asm volatile("" : "+v"(sg));
                lds_dma16(zb + sg, buf_addr + (unsigned)(nxp + r * THREADS + wave * 64) * 16u);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < RX; ++r) {
            if (r * THREADS + wave * 64 >= nxp) break;
            int se = sxe[r], sg = sxg[r];
            asm volatile("" : "+v"(se), "+v"(sg));
            const int row = se & 255, col = se >> 8;
            const bool ok = se >= 0 && row >= ylo && row < yhi && col >= xlo && col < xhi;
            const float *src = ok ? xb + sg : zero;
            lds_dma16(src, buf_addr + (unsigned)(r * THREADS + wave * 64) * 16u);
        }
#pragma unroll
        for (int r = 0; r < RZ; ++r) {
            if (r * THREADS + wave * 64 >= nzp) break;
            int se = sze[r], sg = szg[r];
            asm volatile("" : "+v"(se), "+v"(sg));
            const int row = se & 255, col = se >> 8;
            const bool ok = se >= 0 && row < a.H - y0 && col < a.W - x0;
            const float *src = ok ? zb + sg : zero;
            lds_dma16(src, buf_addr + (unsigned)(nxp + r * THREADS + wave * 64) * 16u);
        }
    };
    const int kgroups = a.TW >> 3, brows = a.TH >> 1;
    int tile = blockIdx.x;
    if (tile < a.total_tiles) stage(tile, 0);
    for (int cur = 0; tile < a.total_tiles; tile += gridDim.x, cur ^= 1) {
        dma_wait();
        __syncthreads();
        if (tile + (int)gridDim.x < a.total_tiles) stage(tile + gridDim.x, cur ^ 1);
        const float *xs = lds + cur * buf_floats + xlane;
        const float *zs = lds + cur * buf_floats + nxp * 4 + zlane;
        // This wave's steps (block row, group of four blocks) as one flat loop, software-pipelined at the grain of a
        // channel group: the raw values of the NEXT group (of the next step after the last one) are requested right
        // after the current group's transform has consumed its registers, and arrive while its MFMAs run - one set of
        // raw registers, no wait in front of an MFMA group but the first of a tile.
        const int nsteps = brows > wk ? ((brows - wk + WK - 1) / WK) * kgroups : 0;
        const float *xk = xs + 2 * wk * LW * CIN, *zk = zs + 2 * wk * a.TW * COUT;
        int kg = 0;
        constexpr int XR = PU == 4 ? 16 : 8;
        float xraw[XR], zraw[NB][4];
        auto load_x = [&](const float *xp, int i) {
            if constexpr (PU == 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 4; ++c) xraw[4 * r + c] = xp[r * LW * CIN + xcol[c] + ca[i]];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { xraw[c] = xp[xcol[c] + ca[i]]; xraw[4 + c] = xp[xrow2 + xcol[c] + ca[i]]; }
            }
        };
        auto load_z = [&](const float *zp) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                zraw[j][0] = zp[cb[j]]; zraw[j][1] = zp[zcol1 + cb[j]];
                zraw[j][2] = zp[a.TW * COUT + cb[j]]; zraw[j][3] = zp[a.TW * COUT + zcol1 + cb[j]];
            }
        };
        if (nsteps > 0) { load_z(zk); load_x(xk, 0); }
        for (int s2 = 0; s2 < nsteps; ++s2) {
            // the step after this one (the last step re-reads itself)
            const bool more = s2 + 1 < nsteps, wrap = kg + 1 == kgroups;
            const float *xn = xk + (more ? (wrap ? (2 * WK * LW - XSTEP * (kgroups - 1)) * CIN : XSTEP * CIN) : 0);
            const float *zn = zk + (more ? (wrap ? (2 * WK * a.TW - ZSTEP * (kgroups - 1)) * COUT : ZSTEP * COUT) : 0);
            kg = wrap ? 0 : kg + 1;
            float dt[NB][4 * PU];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (PU == 4) {
                    const float z00 = zraw[j][0], z01 = zraw[j][1], z10 = zraw[j][2], z11 = zraw[j][3];
                    const float e[4][2] = {{z00, z01}, {z00 + z10, z01 + z11}, {z00 - z10, z01 - z11}, {-z10, -z11}};
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu) {
                        dt[j][4 * uu] = e[uu][0]; dt[j][4 * uu + 1] = e[uu][0] + e[uu][1];
                        dt[j][4 * uu + 2] = e[uu][0] - e[uu][1]; dt[j][4 * uu + 3] = -e[uu][1];
                    }
                } else {
                    const float t0 = fmaf(d1, zraw[j][2], d0 * zraw[j][0]), t1 = fmaf(d1, zraw[j][3], d0 * zraw[j][1]);
                    dt[j][0] = t0; dt[j][1] = t0 + t1; dt[j][2] = t0 - t1; dt[j][3] = -t1;
                }
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                float xt[4 * PU];
                if constexpr (PU == 4) {
                    float q[4][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x0 = xraw[c], x1 = xraw[4 + c], x2 = xraw[8 + c], x3 = xraw[12 + c];
                        q[0][c] = x0 - x2; q[1][c] = x1 + x2; q[2][c] = x2 - x1; q[3][c] = x1 - x3;
                    }
#pragma unroll
                    for (int uu = 0; uu < 4; ++uu) {
                        xt[4 * uu] = q[uu][0] - q[uu][2]; xt[4 * uu + 1] = q[uu][1] + q[uu][2];
                        xt[4 * uu + 2] = q[uu][2] - q[uu][1]; xt[4 * uu + 3] = q[uu][1] - q[uu][3];
                    }
                } else {
                    float t[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] = fmaf(sx, xraw[4 + c], xraw[c]);
                    xt[0] = t[0] - t[2]; xt[1] = t[1] + t[2]; xt[2] = t[2] - t[1]; xt[3] = t[1] - t[3];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < NA) load_x(xk, i + 1);
                else { load_z(zn); load_x(xn, 0); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < 4 * PU; ++v)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[v][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[v], dt[j][v], acc[v][i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            xk = xn; zk = zn;
        }
    }
    // ---- the 16 positions of one 16 x 16 channel-pair tile at a time through LDS: [wk][position][ci_l][co_l]; thread e
    // of the first 256 sums the WK copies in order, applies A^T S A and writes the nine taps of its (ci, co)
    __syncthreads();
    float *out = a.partial + (size_t)blockIdx.x * 9 * CIN * COUT;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            for (int w = 0; w < WK; ++w) {                       // the K copies, added in wave order
                if (wk == w) {
#pragma unroll
                    for (int v = 0; v < 4 * PU; ++v)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float *q = lds + (4 * u + v) * 256 + (4 * g + r) * 16 + nn;
                            *q = (w == 0) ? acc[v][i][j][r] : *q + acc[v][i][j][r];
                        }
                }
                __syncthreads();
            }
            if (tid < 256) {                                     // (THREADS >= 256 in every build)
                const int ci = 16 * i + (tid >> 4), co = 16 * j + (tid & 15);
                float sp[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) sp[p] = lds[p * 256 + tid];
                float h[3][4];                                   // A^T S
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float s12 = 0.5f * (sp[4 + v] + sp[8 + v]);
                    h[0][v] = sp[v] + s12;
                    h[1][v] = 0.5f * (sp[4 + v] - sp[8 + v]);
                    h[2][v] = s12 + sp[12 + v];
                }
                if (ci < CIN && co < COUT) {
#pragma unroll
                    for (int aa = 0; aa < 3; ++aa) {
                        const float s12 = 0.5f * (h[aa][1] + h[aa][2]);
                        out[((size_t)(aa * 3 + 0) * CIN + ci) * COUT + co] = h[aa][0] + s12;
                        out[((size_t)(aa * 3 + 1) * CIN + ci) * COUT + co] = 0.5f * (h[aa][1] - h[aa][2]);
                        out[((size_t)(aa * 3 + 2) * CIN + ci) * COUT + co] = s12 + h[aa][3];
                    }
                }
            }
            __syncthreads();
        }
}

// dW[o][i][a][b] = sum_blocks partial[blk][(2-a)*3 + (2-b)][i][o]   (correlation form -> Lasagne's flipped filters)
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float *__restrict__ partial, int nblocks, int cin,
                                                            int cout, float *__restrict__ dW) {
    // 64 CONSECUTIVE values of the partial layout [tap][ci][co] per workgroup (16 lanes x float4 = one 256-byte read
    // per partial block) x 64 interleaved subsets of the per-workgroup partials - a single chain over ~1000 partials
    // is latency-bound - then a fixed-order finish in float64 and one scatter to the OIHW positions.  (Indexing the
    // values in OIHW order made every read of a wave hit 64 different lines: 132 us per layer; 16 subsets of scalar
    // reads: 48 us per layer, 0.67 ms of the batch-512 step.)
    __shared__ double red[64 * 64];
    const int tid = threadIdx.x, q = tid & 15, part = tid >> 4;
    const int e0 = blockIdx.x * 64 + q * 4;             // index into [tap][ci][co]; 9 cin cout is a multiple of 4
    const int total = cout * cin * 9;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (e0 < total) {
        const size_t per_block = (size_t)9 * cin * cout;
        const float *src = partial + e0;
#pragma unroll 4
        for (int blk = part; blk < nblocks; blk += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(src + blk * per_block);
            s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[part * 64 + q * 4 + k] = s[k];
    __syncthreads();
    const int e = blockIdx.x * 64 + tid;
    if (tid < 64 && e < total) {
        double t = 0.0;
#pragma unroll 8
        for (int pI = 0; pI < 64; ++pI) t += red[pI * 64 + tid];
        const int o = e % cout, i = (e / cout) % cin, tap = e / (cout * cin);
        const int a = 2 - tap / 3, b = 2 - tap % 3;
        dW[(((size_t)o * cin + i) * 3 + a) * 3 + b] = (float)t;
    }
}

struct WgradVariant { int cin, cout; void (*kernel)(WgradArgs); int taps_waves, rx, rz; int wm = 0, wk = 0; int wino = 0, pu = 1, pref = 0; };
static const WgradVariant g_wgrad[] = {
    {12, 12, wgrad_mfma_kernel<12, 12>, 0, 0, 0}, {12, 24, wgrad_mfma_kernel<12, 24>, 0, 0, 0},
    {24, 24, wgrad_mfma_kernel<24, 24>, 0, 0, 0},
    {24, 48, wgrad_mfma_kernel<24, 48>, 0, 0, 0}, {48, 48, wgrad_mfma_kernel<48, 48>, 0, 0, 0},
    {48, 96, wgrad_mfma_kernel<48, 96>, 0, 0, 0}, {96, 96, wgrad_mfma_kernel<96, 96>, 0, 0, 0},
    // packed-taps form for the small-channel blocks (preferred when it exists; ASR_WGRAD_TAPS=0 disables)
    {12, 12, wgrad_taps_kernel<12, 12, 4, 5, 4>, 4, 5, 4}, {12, 24, wgrad_taps_kernel<12, 24, 4, 5, 7>, 4, 5, 7},
    {24, 24, wgrad_taps_kernel<24, 24, 8, 5, 4>, 8, 5, 4},
    // LDS-DMA double-buffered form (wm > 0; preferred when it exists; ASR_WGRAD_DMA=0 disables)
    {12, 12, wgrad_dma_kernel<12, 12, 1, 4, 8, 8>, 0, 8, 8, 1, 4}, {12, 24, wgrad_dma_kernel<12, 24, 1, 4, 8, 8>, 0, 8, 8, 1, 4},
    {24, 24, wgrad_dma_kernel<24, 24, 2, 4, 8, 8>, 0, 8, 8, 2, 4}, {24, 48, wgrad_dma_kernel<24, 48, 2, 2, 8, 8>, 0, 8, 8, 2, 2},
    {48, 48, wgrad_dma_kernel<48, 48, 4, 2, 8, 8>, 0, 8, 8, 4, 2},
    // the _rsz model's 96-channel blocks: 7 x 6 accumulator tiles (168 registers) per wave, two waves per SIMD
    {48, 96, wgrad_dma_kernel<48, 96, 4, 2, 8, 8>, 0, 8, 8, 4, 2}, {96, 96, wgrad_dma_kernel<96, 96, 8, 1, 8, 8>, 0, 8, 8, 8, 1},
    // Winograd F(3x3, 2x2) form (wino = its waves; pu = transform rows per wave; pref = 1: the planner's choice for its
    // block without the training tuner - where the form measured faster at batch 512 on the sheet tower's maps, see
    // DESIGN.md; ASR_WGRAD_WINO=0 / 1 / 2: never / the first / the all-positions build where it exists)
    {12, 24, wgrad_wino_kernel<12, 24, 1, 2, 8, 8>, 0, 8, 8, 0, 0, 8, 1, 0}, {24, 24, wgrad_wino_kernel<24, 24, 1, 2, 8, 8>, 0, 8, 8, 0, 0, 8, 1, 0},
    {24, 48, wgrad_wino_kernel<24, 48, 1, 2, 8, 8>, 0, 8, 8, 0, 0, 8, 1, 1}, {48, 48, wgrad_wino_kernel<48, 48, 1, 2, 8, 8>, 0, 8, 8, 0, 0, 8, 1, 1},
    {12, 12, wgrad_wino_kernel<12, 12, 1, 2, 8, 8>, 0, 8, 8, 0, 0, 8, 1, 0},
    // ... four waves, one per transform row: smaller tiles, two or three workgroups per CU
    {24, 24, wgrad_wino_kernel<24, 24, 1, 1, 16, 16>, 0, 16, 16, 0, 0, 4, 1, 1}, {24, 48, wgrad_wino_kernel<24, 48, 1, 1, 16, 16>, 0, 16, 16, 0, 0, 4, 1, 0},
    {48, 48, wgrad_wino_kernel<48, 48, 1, 1, 16, 16>, 0, 16, 16, 0, 0, 4, 1, 0},
    // ... with all sixteen positions in every wave (the waves split the block rows of a tile)
    {12, 12, wgrad_wino_kernel<12, 12, 4, 8, 8, 8>, 0, 8, 8, 0, 0, 8, 4, 1}, {12, 24, wgrad_wino_kernel<12, 24, 4, 8, 8, 8>, 0, 8, 8, 0, 0, 8, 4, 0},
    {24, 24, wgrad_wino_kernel<24, 24, 4, 4, 16, 16>, 0, 16, 16, 0, 0, 4, 4, 0},
};
static int wgrad_wino_wk(const WgradVariant &v) { return v.pu == 4 ? v.wino : v.wino / 4; }   // its K split

static bool plan_wgrad_dma(int vi, int H, int W, int num_cus, WgradPlan *p, std::vector<WgradPlan> *all = nullptr) {
    const WgradVariant &v = g_wgrad[vi];
    const int cin = v.cin, cout = v.cout;
    const int csx = cin, csz = cout;                            // unpadded pixel strides (see the kernel)
    const int waves = v.wm * v.wk, threads = 64 * waves;
    const int mpw = 7, nj = (cout + 15) / 16;                   // seven m-tiles per unit of 108 packed rows
    const int red_bytes = v.wm * nj * 256 * 4;                  // the final cross-wave sum re-uses the tile LDS
    static const int budget_kb = getenv("ASR_WGRAD_LDS_KB") ? atoi(getenv("ASR_WGRAD_LDS_KB")) : 78;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    WgradPlan bp{};
    auto search = [&](int budget) {
        double best = 1e300;
        for (int TH = v.wk; TH <= std::min(std::max(H, v.wk), 64); ++TH)
            for (int TW = 4; TW <= std::min((W + 3) & ~3, 64); TW += 4) {
                const int nxp = ((TH + 2) * (TW + 2) * (csx / 4) + 63) & ~63, nzp = (TH * TW * (csz / 4) + 63) & ~63;
                if (nxp > v.rx * threads || nzp > v.rz * threads || TH + 2 > 255 || TW + 2 > 255) continue;
                const int lds = std::max(2 * (nxp + nzp) * 16, red_bytes);
                if (lds > budget * 1024) continue;
                const int ty = (H + TH - 1) / TH, tx = (W + TW - 1) / TW;
                const int rows_per_wave = (TH + v.wk - 1) / v.wk;
                // per tile and wave: MFMA issue + the copy instructions this wave issues (a few cycles each inside the
                // image, ~100 with the border tests) + barrier / pipeline fill (~800); the copies themselves overlap
                // unless they are longer
                const double mf = (double)rows_per_wave * (TW / 4) * mpw * nj * 32.0;
                const double border = 1.0 - (double)std::max(ty - 2, 0) * std::max(tx - 2, 0) / ((double)ty * tx);
                const double st = (double)((nxp + nzp) / 64 + waves - 1) / waves * (12.0 + 100.0 * border);
                const double cp = (nxp + nzp) * 16 / 32.0;
                const double cost = (std::max(mf + st, cp) + 800.0) * ty * tx;
                if (all) {                                      // tuner: every tiling with its model cost (filtered below)
                    WgradPlan c{};
                    c.cin = cin; c.cout = cout; c.H = H; c.W = W; c.variant = vi;
                    c.TH = TH; c.TW = TW; c.tiles_y = ty; c.tiles_x = tx; c.lds_bytes = lds;
                    c.grid_cap = (int)std::min(cost, 2.0e9);    // (cost parked here until the list is cut)
                    all->push_back(c);
                }
                if (cost < best) { best = cost; bp.TH = TH; bp.TW = TW; bp.tiles_y = ty; bp.tiles_x = tx; bp.lds_bytes = lds; }
            }
        return best < 1e300;
    };
    auto occupancy = [&]() {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), threads,
                                                         (size_t)bp.lds_bytes) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 1;
        }
        return nb;
    };
    if (all) return search(156);                                // tuner: the whole list, both budgets
    if (!search(budget_kb)) return false;
    int nb = occupancy();
    // registers allow one workgroup per CU only: let it have the whole LDS (bigger tiles, fewer barriers and halos)
    if (nb == 1 && budget_kb < 150 && search(150)) nb = occupancy();
    bp.cin = cin; bp.cout = cout; bp.H = H; bp.W = W; bp.variant = vi;
    bp.grid_cap = num_cus * std::min(nb, 4);
    if (getenv("ASR_DEBUG"))
        fprintf(stderr, "[asr] plan wgrad(dma) %d->%d %dx%d: tile %dx%d, tiles %dx%d, lds %d B, %d blocks/CU\n", cin, cout,
                H, W, bp.TH, bp.TW, bp.tiles_y, bp.tiles_x, bp.lds_bytes, std::min(nb, 4));
    *p = bp;
    return true;
}


// Winograd form: tiles of TH x TW dz pixels, TH even (block rows), TW a multiple of 8 (four 2x2 blocks per k-step)
static bool plan_wgrad_wino(int vi, int H, int W, int num_cus, WgradPlan *p, std::vector<WgradPlan> *all = nullptr) {
    const WgradVariant &v = g_wgrad[vi];
    const int cin = v.cin, cout = v.cout, wk = wgrad_wino_wk(v);
    const int threads = 64 * v.wino;
    const bool all16 = wk == v.wino;
    const int na = (cin + 15) / 16, nb = (cout + 15) / 16;
    const int fin_bytes = 16 * 256 * 4;                         // the final exchange re-uses the tile LDS
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    WgradPlan bp{};
    double best = 1e300;
    for (int budget : {78, 150}) {
        for (int TH = 2; TH <= std::min((H + 1) & ~1, 64); TH += 2)
            for (int TW = 8; TW <= std::min((W + 7) & ~7, 64); TW += 8) {
                const int nxp = ((TH + 2) * (TW + 2) * (cin / 4) + 63) & ~63, nzp = (TH * TW * (cout / 4) + 63) & ~63;
                if (nxp > v.rx * threads || nzp > v.rz * threads) continue;
                const int lds = std::max(2 * (nxp + nzp) * 16, fin_bytes);
                if (lds > budget * 1024) continue;
                const int ty = (H + TH - 1) / TH, tx = (W + TW - 1) / TW;
                const int rows_per_wave = (TH / 2 + wk - 1) / wk;
                // per tile and wave: MFMA issue + the transforms' reads and vector operations, the copy instructions this
                // wave issues, barrier / pipeline fill; the copies themselves overlap unless they are longer
                const double step = all16 ? 16.0 * na * nb * 32.0 + (48.0 * na + 24.0 * nb) * 6.0 + 40.0
                                          : 4.0 * na * nb * 32.0 + (16.0 * na + 8.0 * nb) * 6.0 + 40.0;
                const double mf = (double)rows_per_wave * (TW / 8) * step;
                const double border = 1.0 - (double)std::max(ty - 2, 0) * std::max(tx - 2, 0) / ((double)ty * tx);
                const double st = (double)((nxp + nzp) / 64 + v.wino - 1) / v.wino * (12.0 + 100.0 * border);
                const double cp = (nxp + nzp) * 16 / 32.0;
                // two workgroups per CU at the small budget share the SIMDs: their MFMA time adds, their fill overlaps
                const double cost = (std::max(mf * (budget <= 78 ? 1.0 : 0.5) + st, cp) + 800.0) * ty * tx;
                if (all) {
                    WgradPlan c{};
                    c.cin = cin; c.cout = cout; c.H = H; c.W = W; c.variant = vi;
                    c.TH = TH; c.TW = TW; c.tiles_y = ty; c.tiles_x = tx; c.lds_bytes = lds;
                    c.grid_cap = (int)std::min(cost, 2.0e9);
                    all->push_back(c);
                }
                if (cost < best) { best = cost; bp.TH = TH; bp.TW = TW; bp.tiles_y = ty; bp.tiles_x = tx; bp.lds_bytes = lds; }
            }
    }
    if (best >= 1e300) return false;
    if (all) return true;
    int nb_occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_occ, reinterpret_cast<const void *>(v.kernel), threads,
                                                     (size_t)bp.lds_bytes) != hipSuccess || nb_occ < 1) {
        (void)hipGetLastError();
        nb_occ = 1;
    }
    bp.cin = cin; bp.cout = cout; bp.H = H; bp.W = W; bp.variant = vi;
    bp.grid_cap = num_cus * std::min(nb_occ, 4);
    if (getenv("ASR_DEBUG"))
        fprintf(stderr, "[asr] plan wgrad(wino) %d->%d %dx%d: tile %dx%d, tiles %dx%d, lds %d B, %d blocks/CU\n", cin, cout,
                H, W, bp.TH, bp.TW, bp.tiles_y, bp.tiles_x, bp.lds_bytes, std::min(nb_occ, 4));
    *p = bp;
    return true;
}

static bool plan_wgrad_taps(int vi, int H, int W, int num_cus, WgradPlan *p) {
    const WgradVariant &v = g_wgrad[vi];
    const int cin = v.cin, cout = v.cout;
    const int csx = wg_stride(cin), csz = wg_stride(cout);
    const int threads = 64 * v.taps_waves;
    const int mt = (9 * cin + 15) / 16, nj = (cout + 15) / 16;
    const int red_bytes = mt * nj * 256 * 4;                    // the final cross-wave sum re-uses the tile LDS
    double best = 1e300;
    WgradPlan bp{};
    for (int TH = v.taps_waves; TH <= std::min(std::max(H, v.taps_waves), 64); ++TH)
        for (int TW = 4; TW <= std::min((W + 3) & ~3, 64); TW += 4) {
            const int nxv = (TH + 2) * (TW + 2) * (csx / 4), nzv = TH * TW * (csz / 4);
            if (nxv > v.rx * threads || nzv > v.rz * threads || TH + 2 > 255 || TW + 2 > 255) continue;
            const int lds = std::max(((TH + 2) * (TW + 2) * csx + TH * TW * csz) * 4, red_bytes);
            if (lds > 52 * 1024) continue;                      // >= 3 workgroups per CU
            const int ty = (H + TH - 1) / TH, tx = (W + TW - 1) / TW;
            const int rows_per_wave = (TH + v.taps_waves - 1) / v.taps_waves;
            const double cost = ((double)rows_per_wave * (TW / 4) * mt * nj * 32.0 + lds / 16.0 + 600.0) * ty * tx;
            if (cost < best) { best = cost; bp.TH = TH; bp.TW = TW; bp.tiles_y = ty; bp.tiles_x = tx; bp.lds_bytes = lds; }
        }
    if (best >= 1e300) return false;
    bp.cin = cin; bp.cout = cout; bp.H = H; bp.W = W; bp.variant = vi;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), threads,
                                                     (size_t)bp.lds_bytes) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = 2;
    }
    bp.grid_cap = num_cus * std::min(nb, 4);
    if (getenv("ASR_DEBUG"))
        fprintf(stderr, "[asr] plan wgrad(taps) %d->%d %dx%d: tile %dx%d, tiles %dx%d, lds %d B, %d blocks/CU\n", cin, cout,
                H, W, bp.TH, bp.TW, bp.tiles_y, bp.tiles_x, bp.lds_bytes, std::min(nb, 4));
    *p = bp;
    return true;
}

bool plan_wgrad(int cin, int cout, int H, int W, int num_cus, WgradPlan *p) {
    int vi = -1;
    static const int use_taps = getenv("ASR_WGRAD_TAPS") ? atoi(getenv("ASR_WGRAD_TAPS")) : 1;
    static const int use_dma = getenv("ASR_WGRAD_DMA") ? atoi(getenv("ASR_WGRAD_DMA")) : 1;
    // ASR_WGRAD_WINO=1: the Winograd form wherever it exists (otherwise only the training tuner can pick it); 0: never
    static const int use_wino = getenv("ASR_WGRAD_WINO") ? atoi(getenv("ASR_WGRAD_WINO")) : -1;
    // (2: its all-positions-per-wave build where one exists)
    for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])) && use_wino == -1; ++i)
        if (g_wgrad[i].cin == cin && g_wgrad[i].cout == cout && g_wgrad[i].wino > 0 && g_wgrad[i].pref &&
            plan_wgrad_wino(i, H, W, num_cus, p))
            return true;
    for (int pass = 0; pass < 2 && use_wino >= 1; ++pass)
        for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])); ++i)
            if (g_wgrad[i].cin == cin && g_wgrad[i].cout == cout && g_wgrad[i].wino > 0 &&
                (pass == 1 || (g_wgrad[i].pu == 4) == (use_wino == 2)) && plan_wgrad_wino(i, H, W, num_cus, p))
                return true;
    for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])); ++i)
        if (g_wgrad[i].cin == cin && g_wgrad[i].cout == cout && g_wgrad[i].wm > 0 && use_dma &&
            plan_wgrad_dma(i, H, W, num_cus, p))
            return true;
    for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])); ++i)
        if (g_wgrad[i].cin == cin && g_wgrad[i].cout == cout && g_wgrad[i].taps_waves > 0 && use_taps &&
            plan_wgrad_taps(i, H, W, num_cus, p))
            return true;
    for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])); ++i)
        if (g_wgrad[i].cin == cin && g_wgrad[i].cout == cout && g_wgrad[i].taps_waves == 0 && g_wgrad[i].wm == 0 &&
            g_wgrad[i].wino == 0)
            vi = i;
    if (vi < 0) return false;
    const int csx = wg_stride(cin), csz = wg_stride(cout);
    // two 9-wave workgroups per CU (ASR_WGRAD_BLOCKS=1: one big tile): the staging of one overlaps the MFMA loop of
    // the other
    static const int per_cu = getenv("ASR_WGRAD_BLOCKS") ? std::max(1, atoi(getenv("ASR_WGRAD_BLOCKS"))) : 2;
    const int budget = (per_cu >= 2 ? 76 : 150) * 1024;
    double best = 1e300;
    WgradPlan bp{};
    for (int TH = 1; TH <= std::min(H, 32); ++TH)
        for (int TW = 4; TW <= std::min((W + 3) & ~3, 64); TW += 4) {
            const int lds = ((TH + 2) * (TW + 2) * csx + TH * TW * csz) * 4;
            if (lds > budget) continue;
            const int ty = (H + TH - 1) / TH, tx = (W + TW - 1) / TW;
            // MFMA issue per tile (one wave per tap) + staging (~16 B/clk) + fixed overhead
            const double cost = ((double)TH * (TW / 4) * ((cin + 15) / 16) * ((cout + 15) / 16) * 32.0 + lds / 16.0 + 800.0) * ty * tx;
            if (cost < best) { best = cost; bp.TH = TH; bp.TW = TW; bp.tiles_y = ty; bp.tiles_x = tx; bp.lds_bytes = lds; }
        }
    if (best >= 1e300) return false;
    bp.cin = cin; bp.cout = cout; bp.H = H; bp.W = W; bp.variant = vi;
    bp.grid_cap = num_cus * (per_cu >= 2 ? 2 : 1);
    if (getenv("ASR_DEBUG"))
        fprintf(stderr, "[asr] plan wgrad %d->%d %dx%d: tile %dx%d, tiles %dx%d, lds %d B\n", cin, cout, H, W, bp.TH,
                bp.TW, bp.tiles_y, bp.tiles_x, bp.lds_bytes);
    // the attribute is per FUNCTION, and one instantiation serves several blocks / both towers with different tile
    // sizes: always allow the full 160 KiB instead of the size of whichever plan was made last
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(g_wgrad[vi].kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    *p = bp;
    return true;
}

// the tuner's candidates for a block's weight gradient: the planner's pick first, then the cheapest few tilings of the
// LDS-DMA form by the planner's model at both LDS budgets, with different tile shapes
void wgrad_candidates(int cin, int cout, int H, int W, int num_cus, int max_count, std::vector<WgradPlan> *out) {
    WgradPlan first;
    if (!plan_wgrad(cin, cout, H, W, num_cus, &first)) return;
    out->push_back(first);
    // the Winograd form of this block, its planner's pick and the next-cheapest different tile shapes
    static const int use_wino = getenv("ASR_WGRAD_WINO") ? atoi(getenv("ASR_WGRAD_WINO")) : -1;
    for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])) && use_wino != 0; ++i) {
        if (g_wgrad[i].cin != cin || g_wgrad[i].cout != cout || g_wgrad[i].wino == 0 || i == first.variant) continue;
        WgradPlan wp;
        if (!plan_wgrad_wino(i, H, W, num_cus, &wp)) continue;
        out->push_back(wp);
        std::vector<WgradPlan> wall;
        (void)plan_wgrad_wino(i, H, W, num_cus, &wp, &wall);
        std::sort(wall.begin(), wall.end(), [](const WgradPlan &a, const WgradPlan &b) { return a.grid_cap < b.grid_cap; });
        int added = 0;
        for (const WgradPlan &c : wall) {
            if (added >= 2) break;
            bool close = false;
            for (const WgradPlan &o : *out)
                if (o.variant == c.variant && std::abs(o.TH - c.TH) * 4 <= o.TH && std::abs(o.TW - c.TW) * 4 <= o.TW) close = true;
            if (close) continue;
            WgradPlan q = c;
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(g_wgrad[i].kernel),
                                                             64 * g_wgrad[i].wino, (size_t)q.lds_bytes) != hipSuccess || nb < 1) {
                (void)hipGetLastError();
                nb = 1;
            }
            q.grid_cap = num_cus * std::min(nb, 4);
            out->push_back(q);
            ++added;
        }
    }
    max_count += (int)out->size() - 1;
    // the packed-taps LDS-DMA form: its planner's pick (unless that is `first`) and the next-cheapest tile shapes
    int dv = -1;
    static const int use_dma = getenv("ASR_WGRAD_DMA") ? atoi(getenv("ASR_WGRAD_DMA")) : 1;
    for (int i = 0; i < (int)(sizeof(g_wgrad) / sizeof(g_wgrad[0])) && use_dma; ++i)
        if (g_wgrad[i].cin == cin && g_wgrad[i].cout == cout && g_wgrad[i].wm > 0) dv = i;
    if (dv < 0) return;                                         // no DMA form of this block: nothing else to time
    if (dv != first.variant) {
        WgradPlan dp;
        if (!plan_wgrad_dma(dv, H, W, num_cus, &dp)) return;
        out->push_back(dp);
        ++max_count;
    }
    std::vector<WgradPlan> all;
    WgradPlan dummy;
    (void)plan_wgrad_dma(dv, H, W, num_cus, &dummy, &all);
    std::sort(all.begin(), all.end(), [](const WgradPlan &a, const WgradPlan &b) { return a.grid_cap < b.grid_cap; });
    const WgradVariant &v = g_wgrad[dv];
    for (const WgradPlan &c : all) {
        if ((int)out->size() >= max_count) break;
        bool close = false;
        for (const WgradPlan &o : *out)
            if (o.variant == c.variant && std::abs(o.TH - c.TH) * 4 <= o.TH && std::abs(o.TW - c.TW) * 4 <= o.TW)
                close = true;                                                                       // within 25 %
        if (close) continue;
        WgradPlan q = c;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), 64 * v.wm * v.wk,
                                                         (size_t)q.lds_bytes) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 1;
        }
        q.grid_cap = num_cus * std::min(nb, 4);
        out->push_back(q);
    }
}

size_t wgrad_partial_floats(const WgradPlan &p) { return (size_t)p.grid_cap * 9 * p.cin * p.cout; }

hipError_t launch_wgrad(hipStream_t s, const WgradPlan &p, const float *x, const float *dz, int N, float *partial,
                        float *dW) {
    WgradArgs a;
    a.x = x; a.dz = dz; a.partial = partial; a.N = N; a.H = p.H; a.W = p.W; a.TH = p.TH; a.TW = p.TW;
    a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x; a.total_tiles = N * p.tiles_y * p.tiles_x;
    const int grid = std::max(1, std::min(a.total_tiles, p.grid_cap));
    const WgradVariant &wv = g_wgrad[p.variant];
    const int threads = wv.wino > 0 ? 64 * wv.wino : wv.wm > 0 ? 64 * wv.wm * wv.wk : wv.taps_waves > 0 ? 64 * wv.taps_waves : 576;
    hipLaunchKernelGGL(g_wgrad[p.variant].kernel, dim3(grid), dim3(threads), p.lds_bytes, s, a);
    const int total = p.cout * p.cin * 9;
    wgrad_reduce_kernel<<<(total + 63) / 64, 1024, 0, s>>>(partial, grid, p.cin, p.cout, dW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// block 1 weight gradient (C_in = 1): dW[o][0][a][b] = sum x[n, y+1-a, x+1-b] dz[n,y,x,o]
// ---------------------------------------------------------------------------
// FUSE: block 1's BatchNorm / ELU backward is applied HERE - dz = gamma s (dy ELU'(y) - mean(dy) - xhat mean(dy xhat)) from
// the raw output z and the incoming gradient, the same float32 expression as bn_bwd_apply_kernel's unpooled branch.
// Block 1 has no data gradient, so this kernel is dz's only reader: the apply pass (2.4 GB of traffic at batch 512 on
// the sheet tower, 0.42 ms) and the dz tensor itself disappear; this kernel reads z and dout instead of dz (+0.8 GB).
struct Conv1BnBwd {
    const float *z, *dout, *stats, *gamma, *beta;
    const float *w;            // z == null: block 1's raw output is recomputed from the taps with these weights [COUT][9]
    const double *sums;
    double inv_m;              // 1 / (samples of the whole batch * H * W)
};
template <int COUT, bool FUSE>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ dz,
                                                          int N, int H, int W, double *__restrict__ partial,
                                                          Conv1BnBwd f) {
    // thread accumulates COUT*9 partial sums over its pixels (correlation taps t = a'*3+b': x[y-1+a', x-1+b'])
    __shared__ double red[4][COUT * 9];
    __shared__ float kc[FUSE ? 6 * COUT : 1];              // per channel: mu, istd, sc, be, m1, m2 (broadcast reads)
    if (FUSE) {
        for (int c = threadIdx.x; c < COUT; c += 256) {
            const float mu = f.stats[c], istd = f.stats[COUT + c];
            kc[c] = mu; kc[COUT + c] = istd; kc[2 * COUT + c] = f.gamma[c] * istd; kc[3 * COUT + c] = f.beta[c];
            kc[4 * COUT + c] = (float)(f.sums[c] * f.inv_m); kc[5 * COUT + c] = (float)(f.sums[COUT + c] * f.inv_m);
        }
        __syncthreads();
    }
    float acc[COUT * 9];
#pragma unroll
    for (int i = 0; i < COUT * 9; ++i) acc[i] = 0.0f;
    const int64_t total = (int64_t)N * H * W;
    const bool small = total < ((int64_t)1 << 31);        // uniform: 32-bit index arithmetic (a fifth of the 64-bit cost)
    // One pixel per iteration, software-pipelined: the nine taps and the gradient (and z) of the NEXT pixel are requested
    // before the current one is worked on.  The accumulators leave room for two waves per SIMD only, each iteration is
    // ~350 vector instructions behind a round trip to HBM, and nothing else hid that trip: 0.325 ms for a pass whose
    // traffic takes 0.16.
    struct Px {
        float v[9], mk[9];
        float4 d[COUT / 4], z[COUT / 4];
    };
    auto fetch = [&](int64_t s, Px &p) {
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
        // nine independent loads from clamped (always valid) addresses, the zero padding applied afterwards as a factor
        // (a bounds branch per tap made them nine dependent round trips; a select is turned back into one)
        const float *xn = x + (size_t)n * H * W;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = y - 1 + a, xb = xx - 1 + b;
                const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xb < 0 ? 0 : (xb >= W ? W - 1 : xb);
                p.v[a * 3 + b] = xn[yc * W + xc];
                p.mk[a * 3 + b] = (yy == yc && xb == xc) ? 1.0f : 0.0f;
            }
        const float4 *d4 = reinterpret_cast<const float4 *>((FUSE ? f.dout : dz) + (size_t)s * COUT);
#pragma unroll
        for (int o4 = 0; o4 < COUT / 4; ++o4) p.d[o4] = d4[o4];
        if (FUSE && f.z != nullptr) {                          // uniform
            const float4 *z4 = reinterpret_cast<const float4 *>(f.z + (size_t)s * COUT);
#pragma unroll
            for (int o4 = 0; o4 < COUT / 4; ++o4) p.z[o4] = z4[o4];
        }
    };
    const int64_t sstep = (int64_t)gridDim.x * blockDim.x;
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    Px cur, nxt;
    if (s < total) fetch(s, cur);
#pragma unroll 1
    for (; s < total; s += sstep) {
        fetch(s + sstep < total ? s + sstep : s, nxt);         // (the last iteration re-reads its own pixel)
        __builtin_amdgcn_sched_barrier(0);
        float v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = cur.v[t] * cur.mk[t];
#pragma unroll
        for (int o4 = 0; o4 < COUT / 4; ++o4) {
            const float4 dq = cur.d[o4];
            float dv[4] = {dq.x, dq.y, dq.z, dq.w};
            if (FUSE) {
                float zv[4];
                if (f.z != nullptr) {                          // uniform
                    const float4 zq = cur.z[o4];
                    zv[0] = zq.x; zv[1] = zq.y; zv[2] = zq.z; zv[3] = zq.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float zc = 0.0f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) zc = fmaf(v[t], f.w[(o4 * 4 + j) * 9 + t], zc);   // conv1_raw_kernel's chain
                        zv[j] = zc;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = o4 * 4 + j;
                    const float mu = kc[c], istd = kc[COUT + c], sc = kc[2 * COUT + c], be = kc[3 * COUT + c];
                    const float yy = (zv[j] - mu) * sc + be;
                    const float dact = yy <= 0.0f ? __expf(yy) : 1.0f;
                    const float xhat = (zv[j] - mu) * istd;
                    dv[j] = sc * (dv[j] * dact - kc[4 * COUT + c] - xhat * kc[5 * COUT + c]);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 9; ++t) acc[(o4 * 4 + j) * 9 + t] = fmaf(v[t], dv[j], acc[(o4 * 4 + j) * 9 + t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    // across the wave by float32 shuffles (a thread's own sum is a float32 chain of ~125 terms already; float64 chains
    // here cost 80 registers - one workgroup less per CU for the whole kernel), float64 across the four waves through LDS
    // in wave order and across workgroups.  (One LDS tree reduction per value - 108 x 9 barriers per workgroup - was a
    // third of this kernel's time.)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < COUT * 9; ++i) {
        float sv = acc[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sv += __shfl_xor(sv, m);
        if (lane == 0) red[wave][i] = (double)sv;
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COUT * 9; i += 256)
        partial[(size_t)blockIdx.x * COUT * 9 + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}

// dW[o][0][a][b] = sum_blocks partial[blk][o][(2-a)*3 + (2-b)]
__global__ __launch_bounds__(1024) void conv1_wgrad_reduce_kernel(const double *__restrict__ partial, int nblocks, int cout,
                                                                  float *__restrict__ dW) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, oo = tid & 63, part = tid >> 6;
    const int e = blockIdx.x * 64 + oo;
    double s = 0.0;
    if (e < cout * 9) {
        const int o = e / 9, ab = e % 9, a = ab / 3, b = ab % 3;
        const int t = (2 - a) * 3 + (2 - b);
#pragma unroll 4
        for (int blk = part; blk < nblocks; blk += 16) s += partial[(size_t)blk * cout * 9 + o * 9 + t];
    }
    red[tid] = s;
    __syncthreads();
    if (part == 0 && e < cout * 9) {
        double t2 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t2 += red[q * 64 + oo];
        dW[e] = (float)t2;
    }
}

int conv1_wgrad_blocks() { return 512; }

hipError_t launch_conv1_wgrad(hipStream_t s, const float *x, const float *dz, int N, int H, int W, int cout,
                              double *partial, float *dW, const float *z, const float *dout, const float *stats,
                              const float *gamma, const float *beta, const double *sums, int n_global, const float *w1) {
    const int nb = conv1_wgrad_blocks();
    Conv1BnBwd f{};
    if (z != nullptr || w1 != nullptr) {      // fused BatchNorm / ELU backward (dz unused); w1: z recomputed
        f.z = z; f.w = w1; f.dout = dout; f.stats = stats; f.gamma = gamma; f.beta = beta; f.sums = sums;
        f.inv_m = 1.0 / ((double)(n_global > 0 ? n_global : N) * H * W);
        if (cout == 12) conv1_wgrad_kernel<12, true><<<nb, 256, 0, s>>>(x, nullptr, N, H, W, partial, f);
        else if (cout == 24) conv1_wgrad_kernel<24, true><<<nb, 256, 0, s>>>(x, nullptr, N, H, W, partial, f);
        else return hipErrorInvalidValue;
    } else if (cout == 12) conv1_wgrad_kernel<12, false><<<nb, 256, 0, s>>>(x, dz, N, H, W, partial, f);
    else if (cout == 24) conv1_wgrad_kernel<24, false><<<nb, 256, 0, s>>>(x, dz, N, H, W, partial, f);
    else return hipErrorInvalidValue;
    conv1_wgrad_reduce_kernel<<<(cout * 9 + 63) / 64, 1024, 0, s>>>(partial, nb, cout, dW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// tail backward: H = mean_p BN(z9), z9 = a8 . w9^T
// ---------------------------------------------------------------------------
// sums[o] = sum_n dH[n,o] ; sums[32+o] = sum_n (dH[n,o]/npix) sum_p xhat[n,p,o]
// stage 1: one workgroup per SAMPLE, thread = (channel o, one of 8 pixel lanes): a lane sums every eighth pixel, the
// eight lanes are added in order, partial[n][64].  (One thread per (sample, channel) walking all pixels - 120
// dependent float64 adds behind strided loads - was 37 us for 2 M elements.)
// stage 2: fixed-order reduction of the partials (one workgroup walking all of z9 alone took 0.3 ms).
__global__ __launch_bounds__(256) void tail_bwd_partial_kernel(const float *__restrict__ dH, const float *__restrict__ z9,
                                                               const float *__restrict__ stats, int N, int npix,
                                                               double *__restrict__ partial) {
    __shared__ double sl[8][32];
    const int tid = threadIdx.x, o = tid & 31, l = tid >> 5;
    const int n = blockIdx.x;
    const float mu = stats[o], istd = stats[32 + o];
    double sx = 0.0;
    for (int p = l; p < npix; p += 8) sx += (double)((z9[((size_t)n * npix + p) * 32 + o] - mu) * istd);
    sl[l][o] = sx;
    __syncthreads();
    if (l == 0) {
        double t = sl[0][o];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += sl[q][o];
        const double g = (double)dH[(size_t)n * 32 + o];
        partial[(size_t)n * 64 + o] = g;
        partial[(size_t)n * 64 + 32 + o] = g / (double)npix * t;
    }
}

__global__ __launch_bounds__(1024) void tail_bwd_reduce_kernel(const double *__restrict__ partial, int nblocks,
                                                               double *__restrict__ sums, float *__restrict__ dbeta,
                                                               float *__restrict__ dgamma) {
    __shared__ double red[1024];
    const int tid = threadIdx.x, slot = tid & 63, part = tid >> 6;
    double s = 0.0;
    for (int b = part; b < nblocks; b += 16) s += partial[(size_t)b * 64 + slot];
    red[tid] = s;
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q * 64 + tid];
        sums[tid] = t;
        if (tid < 32) dbeta[tid] = (float)t; else dgamma[tid - 32] = (float)t;
    }
}

// dz9 (in place over z9): gamma s (dH/npix - sum1/M - xhat sum2/M).  grid = (chunks of one sample, samples): no index
// division per element (a flat 64-bit index divided by npix per element made this 62 us per tower)
__global__ __launch_bounds__(256) void tail_bwd_dz_kernel(float *__restrict__ z9, const float *__restrict__ dH,
                                                          const float *__restrict__ stats, const float *__restrict__ gamma,
                                                          const double *__restrict__ sums, int N, int npix, int Ng) {
    const int n = blockIdx.y;
    const int per = npix * 32;
    const int o = threadIdx.x & 31;
    const double inv_m = 1.0 / ((double)Ng * npix);
    const float mu = stats[o], istd = stats[32 + o], gs = gamma[o] * istd;
    const double dy = (double)dH[(size_t)n * 32 + o] / (double)npix;
    const double base = dy - sums[o] * inv_m, k2 = sums[32 + o] * inv_m;
    float *zn = z9 + (size_t)n * per;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per; e += gridDim.x * 256) {     // e & 31 == o (256 % 32 == 0)
        const float xhat = (zn[e] - mu) * istd;
        zn[e] = gs * (float)(base - (double)xhat * k2);
    }
}

// da8[r,c] = sum_o dz9[r,o] w9[o,c]: w9 (32 x C8) in LDS, thread = (row, four channels)
__global__ __launch_bounds__(256) void tail_bwd_da_kernel(const float *__restrict__ dz9, const float *__restrict__ w9,
                                                          float *__restrict__ da8, int64_t rows, int C8) {
    __shared__ __attribute__((aligned(16))) float wl[32 * 96];
    for (int i = threadIdx.x; i < 32 * C8; i += 256) wl[i] = w9[i];
    __syncthreads();
    const int c4n = C8 >> 2;
    const int64_t total = rows * c4n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(e % c4n);
        const int64_t r = e / c4n;
        const float4 *d4 = reinterpret_cast<const float4 *>(dz9 + r * 32);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 dv = d4[q];
            const float d[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {                      // o = 4 q + j ascending: the order of the scalar loop
                const float4 wv = *reinterpret_cast<const float4 *>(wl + (4 * q + j) * C8 + c4 * 4);
                acc[0] = fmaf(d[j], wv.x, acc[0]); acc[1] = fmaf(d[j], wv.y, acc[1]);
                acc[2] = fmaf(d[j], wv.z, acc[2]); acc[3] = fmaf(d[j], wv.w, acc[3]);
            }
        }
        *reinterpret_cast<float4 *>(da8 + r * C8 + c4 * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// dW9[o,c] = sum_r dz9[r,o] a8[r,c] on the MFMA: M = o (two m-tiles), N = c, K = rows, four rows per
// v_mfma_f32_16x16x4_f32.  A wave owns a contiguous run of rows and reads both operands straight from global memory
// (lane (g, nn): row r + g, columns nn of each tile - 64-byte runs), partial[wave][o][c] in float32 (chains of <= 64
// products), block-ordered float64 finish in partial_sum_f32_kernel.  (The scalar float64 form - one thread per
// (o, c), 128 rows each - took 222 us per tower and its one-value-per-thread finish 114 us: 0.67 ms of the step.)
template <int NJ>
__global__ __launch_bounds__(256) void tail_bwd_dw_mfma_kernel(const float *__restrict__ dz9, const float *__restrict__ a8,
                                                               int64_t rows, int C8, int64_t rows_per_wave,
                                                               float *__restrict__ partial) {
    const int lane = threadIdx.x & 63, g = lane >> 4, nn = lane & 15;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t lo = wv * rows_per_wave;
    const int64_t hi = lo + rows_per_wave < rows ? lo + rows_per_wave : rows;
    floatx4 acc[2][NJ];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) acc[mi][nj] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int64_t r = lo; r < hi; r += 4) {
        const int64_t row = r + g;
        const bool ok = row < hi;
        float af[2], bf[NJ];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[mi] = ok ? dz9[row * 32 + mi * 16 + nn] : 0.f;
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) bf[nj] = (ok && nj * 16 + nn < C8) ? a8[row * C8 + nj * 16 + nn] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj)
                acc[mi][nj] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
    }
    // C/D layout: lane (g, nn) holds rows o = mi*16 + 4g + q, column c = nj*16 + nn.  The four waves' tables meet in LDS and
    // leave as ONE table per workgroup (fixed order w0 + w1 + w2 + w3): the finish then reads 64 tables instead of 256
    // (it took 168 us per tower for 1.5 MB - 0.34 ms of the batch-512 step)
    __shared__ float wtab[4][32 * 16 * NJ];
    const int w = threadIdx.x >> 6, ncol = 16 * NJ;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
            for (int q = 0; q < 4; ++q) wtab[w][(mi * 16 + 4 * g + q) * ncol + nj * 16 + nn] = acc[mi][nj][q];
    __syncthreads();
    float *out = partial + (size_t)blockIdx.x * 32 * C8;
    for (int e = threadIdx.x; e < 32 * ncol; e += 256) {
        const int o = e / ncol, c = e - o * ncol;
        if (c < C8) out[(size_t)o * C8 + c] = ((wtab[0][e] + wtab[1][e]) + wtab[2][e]) + wtab[3][e];
    }
}
// out[e] = sum_blocks partial[blk][e], e < n (n a multiple of 16): 16 columns per workgroup (n / 16 workgroups - the
// 1024-thread form covered 64 columns each: 24 workgroups for block 9's 1536 values), thread = (float4 column group,
// one of 64 parts of the block list), float64 sums in block order
__global__ __launch_bounds__(256) void partial_sum_f32_kernel(const float *__restrict__ partial, int nblocks, int n,
                                                              float *__restrict__ out) {
    __shared__ double red[64][16];
    const int tid = threadIdx.x, q = tid & 3, part = tid >> 2;
    const int e0 = blockIdx.x * 16 + q * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (e0 < n) {
        const float *src = partial + e0;
        for (int blk = part; blk < nblocks; blk += 64) {
            const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)blk * n);
            s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[part][q * 4 + k] = s[k];
    __syncthreads();
    const int e = blockIdx.x * 16 + tid;
    if (tid < 16 && e < n) {
        double t = 0.0;
#pragma unroll 8
        for (int pI = 0; pI < 64; ++pI) t += red[pI][tid];
        out[e] = (float)t;
    }
}

// waves (= float32 partial tables of 32 * C8) of the dW9 kernel; the table lives in the float64 `partial` buffer
int tail_dw_blocks(int64_t rows) { return (int)std::max<int64_t>(4, std::min<int64_t>(256, (rows + 31) / 32 / 4 * 4)); }

hipError_t launch_tail_bwd(hipStream_t s, const float *dH, float *z9, const float *a8, const float *w9,
                           const float *stats, const float *gamma, int N, int npix, int C8, double *sums,
                           double *partial, float *dbeta, float *dgamma, float *dW9, float *da8, const Exchange *ex) {
    const int Ng = ex ? ex->n_global : N;               // samples of the whole batch
    // stage-1 partials live at the end of `partial` (the dW9 partials below use its first tail_dw_blocks * 32 * C8)
    double *p1 = partial + (size_t)tail_dw_blocks((int64_t)N * npix) * 32 * C8;       // N rows of 64 (the caller allocates)
    const int phase = ex ? ex->phase : 0;
    if (phase != 2) {
        tail_bwd_partial_kernel<<<N, 256, 0, s>>>(dH, z9, stats, N, npix, p1);
        tail_bwd_reduce_kernel<<<1, 1024, 0, s>>>(p1, N, sums, dbeta, dgamma);
    }
    if (phase == 0 && ex && ex->allreduce_f64(ex->self, s, sums, 64) != 0) return hipErrorUnknown;
    if (phase == 1) return hipGetLastError();
    const int64_t rows = (int64_t)N * npix;
    tail_bwd_dz_kernel<<<dim3((npix * 32 + 255) / 256, N), 256, 0, s>>>(z9, dH, stats, gamma, sums, N, npix, Ng);
    const int b2 = (int)std::min<int64_t>((rows * (C8 / 4) + 255) / 256, 8192);
    tail_bwd_da_kernel<<<b2, 256, 0, s>>>(z9, w9, da8, rows, C8);
    const int nw = tail_dw_blocks(rows);                       // waves, a multiple of 4
    const int64_t rpw = ((rows + nw - 1) / nw + 3) / 4 * 4;
    float *fp = reinterpret_cast<float *>(partial);
    if (C8 % 4 || C8 > 96) return hipErrorInvalidValue;
    if (C8 <= 48) tail_bwd_dw_mfma_kernel<3><<<nw / 4, 256, 0, s>>>(z9, a8, rows, C8, rpw, fp);
    else tail_bwd_dw_mfma_kernel<6><<<nw / 4, 256, 0, s>>>(z9, a8, rows, C8, rpw, fp);
    partial_sum_f32_kernel<<<(32 * C8 + 15) / 16, 256, 0, s>>>(fp, nw / 4, 32 * C8, dW9);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// global mean pool backward helper is folded into tail_bwd; Adam
// ---------------------------------------------------------------------------
// p, g, m, v: flat arrays over ALL 97 parameter tensors; mask[i] != 0 marks trainable elements.
// g <- g + 2*l2*p (weight decay, train_dcca_pool.py:141-142); Lasagne Adam (A.7) with step size a_t.
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   const unsigned char *__restrict__ mask, int64_t n, float a_t,
                                                   float beta1, float beta2, float omb1, float omb2, float eps,
                                                   float l2x2) {
    // omb1 / omb2 = 1.0f - (float)beta, a float32 subtraction: lasagne.updates.adam writes `(one - beta2)` on graph
    // constants, and under floatX = float32 the Python floats 0.9 / 0.999 become float32 constants, so Theano folds
    // 1 - float32(0.999) = 0.00099998713 (1.3e-5 off 0.001).  Round 3 rounded the complement from the double value to
    // match its own float64 oracle; the reference's arithmetic is the float32 one - oracle/train.py and
    // the four-step test use the same complement now.
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const float pv = p[i];
        const float gv = g[i] + l2x2 * pv;
        const float mv = beta1 * m[i] + omb1 * gv;
        const float vv = beta2 * v[i] + omb2 * gv * gv;
        m[i] = mv; v[i] = vv;
        p[i] = pv - a_t * mv / (sqrtf(vv) + eps);
    }
}

hipError_t launch_adam(hipStream_t s, float *p, const float *g, float *m, float *v, const unsigned char *mask,
                       int64_t n, float a_t, double beta1, double beta2, float eps, float l2) {
    const int blocks = (int)std::min<int64_t>((n + 255) / 256, 2048);
    adam_kernel<<<blocks, 256, 0, s>>>(p, g, m, v, mask, n, a_t, (float)beta1, (float)beta2, 1.0f - (float)beta1,
                                       1.0f - (float)beta2, eps, 2.0f * l2);
    return hipGetLastError();
}

// sum of squares of the trainable elements (for the reported loss: + l2 * sum p^2), single workgroup
__global__ __launch_bounds__(1024) void l2_penalty_kernel(const float *__restrict__ p, const unsigned char *__restrict__ mask,
                                                          int64_t n, double *__restrict__ out) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        if (mask[i]) s += (double)p[i] * (double)p[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

hipError_t launch_l2_penalty(hipStream_t s, const float *p, const unsigned char *mask, int64_t n, double *out) {
    l2_penalty_kernel<<<1, 1024, 0, s>>>(p, mask, n, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// master weights -> kernel layouts (on the device, after every update)
// ---------------------------------------------------------------------------
// element bodies: repack_elems.inl
__global__ __launch_bounds__(256) void repack_conv_kernel(const float *__restrict__ W, int cin, int cout,
                                                          float *__restrict__ wfwd, float *__restrict__ wdgrad) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < repack_conv_fwd_count(cin, cout)) repack_conv_fwd_elem(e, W, cin, cout, wfwd);
    if (wdgrad != nullptr && cout % 4 == 0 && e < repack_conv_dgrad_count(cin, cout))
        repack_conv_dgrad_elem(e, W, cin, cout, wdgrad);
}

__global__ void repack_conv1_kernel(const float *__restrict__ W, int cout, float *__restrict__ w1) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= cout * 9) return;
    repack_conv1_elem(e, W, w1);
}

__global__ void bn_fold_kernel(const float *__restrict__ beta, const float *__restrict__ gamma,
                               const float *__restrict__ mean, const float *__restrict__ istd, int cout, int coutp,
                               float *__restrict__ bnp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= coutp) return;
    bn_fold_elem(c, beta, gamma, mean, istd, cout, coutp, bnp);
}

// Every layout of one conv block (blockIdx.y = entry of the table) in one launch: after each training update the
// master parameters of the 18 blocks are re-laid-out for the training kernels (direct-form fragments forward / data
// gradient, Winograd-domain copies where the step's plans use them) and for the deterministic path (BN fold, 1x1
// weights, CCALayer block).  As ~65 separate 5-us launches this was 0.4 ms at the end of every step.
__global__ __launch_bounds__(256) void repack_all_kernel(const RepackDesc *__restrict__ descs) {
    const RepackDesc d = descs[blockIdx.y];
    const int stride = gridDim.x * blockDim.x;
    const int e0 = blockIdx.x * blockDim.x + threadIdx.x;
    if (d.kind == 0) {                                   // block 1
        for (int e = e0; e < d.cout * 9; e += stride) repack_conv1_elem(e, d.W, d.wfwd);
    } else if (d.kind == 1) {                            // 3x3 block
        const int nf = repack_conv_fwd_count(d.cin, d.cout);
        for (int e = e0; e < nf; e += stride) repack_conv_fwd_elem(e, d.W, d.cin, d.cout, d.wfwd);
        if (d.wdgrad) {
            const int nd = repack_conv_dgrad_count(d.cin, d.cout);
            for (int e = e0; e < nd; e += stride) repack_conv_dgrad_elem(e, d.W, d.cin, d.cout, d.wdgrad);
        }
        if (d.wino_fwd) {
            const int nw = wino_pack_count(d.cin, d.cout, 0);
            for (int e = e0; e < nw; e += stride) wino_pack_elem(e, d.W, d.cin, d.cout, 0, d.wino_fwd);
        }
        if (d.wino_dgrad) {
            const int nw = wino_pack_count(d.cin, d.cout, 1);
            for (int e = e0; e < nw; e += stride) wino_pack_elem(e, d.W, d.cin, d.cout, 1, d.wino_dgrad);
        }
        if (d.wino4_fwd) {
            const int nw = wino4_pack_count(d.cin, d.cout, 0);
            for (int e = e0; e < nw; e += stride) wino4_pack_elem(e, d.W, d.cin, d.cout, 0, d.wino4_fwd);
        }
        if (d.wino4_dgrad) {
            const int nw = wino4_pack_count(d.cin, d.cout, 1);
            for (int e = e0; e < nw; e += stride) wino4_pack_elem(e, d.W, d.cin, d.cout, 1, d.wino4_dgrad);
        }
    } else {                                             // kind 2: plain copy of W (1x1 conv weights, CCALayer block)
        for (int e = e0; e < d.cin * d.cout; e += stride) d.wfwd[e] = d.W[e];
    }
    if (d.bnp) {
        const int coutp = (d.cout + 15) / 16 * 16;
        for (int c = e0; c < coutp; c += stride) bn_fold_elem(c, d.beta, d.gamma, d.mean, d.istd, d.cout, coutp, d.bnp);
    }
}

hipError_t launch_repack_all(hipStream_t s, const RepackDesc *descs_dev, int n_descs) {
    if (n_descs <= 0) return hipSuccess;
    repack_all_kernel<<<dim3(24, n_descs), 256, 0, s>>>(descs_dev);
    return hipGetLastError();
}

hipError_t launch_repack_conv(hipStream_t s, const float *W, int cin, int cout, float *wfwd, float *wdgrad) {
    const int nf = (cout + 15) / 16 * 9 * (cin / 4) * 64;
    const int nd = wdgrad ? (cin + 15) / 16 * 9 * (cout / 4) * 64 : 0;
    const int n = std::max(nf, nd);
    repack_conv_kernel<<<(n + 255) / 256, 256, 0, s>>>(W, cin, cout, wfwd, wdgrad);
    return hipGetLastError();
}
hipError_t launch_repack_conv1(hipStream_t s, const float *W, int cout, float *w1) {
    repack_conv1_kernel<<<1, 256, 0, s>>>(W, cout, w1);
    return hipGetLastError();
}
hipError_t launch_bn_fold(hipStream_t s, const float *beta, const float *gamma, const float *mean, const float *istd,
                          int cout, float *bnp) {
    const int coutp = (cout + 15) / 16 * 16;
    bn_fold_kernel<<<1, 128, 0, s>>>(beta, gamma, mean, istd, cout, coutp, bnp);
    return hipGetLastError();
}

}  // namespace asr
