// gfx950 kernels for the conv blocks of the two towers
// (reference: models/mutopia_ccal_cont.py:54-58,76-91 conv_bn + MaxPool2DLayer;
//  semantics SURVEY.md A.1-A.3).
//
//   conv1_kernel        : block 1 (C_in = 1): prepare() + 3x3 stencil + BN + ELU,
//                         VALU, thread = 4 pixels x all channels, HBM-write bound.
//   conv3x3_mfma_kernel : blocks 2..8: implicit GEMM on v_mfma_f32_16x16x4_f32
//                         (exact fp32), M = pixels, N = C_out, K = 9*C_in;
//                         input tile (+halo) staged in LDS, the wave's weight
//                         fragments live in VGPRs for the whole (persistent)
//                         block, BN + ELU + 2x2 max-pool fused in the epilogue.
//
// Layout: activations NHWC fp32.  M-tile = 4 pooling windows x 4 pixels, so the
// 4 accumulator registers of a lane are exactly one 2x2 window: max-pool is a
// per-lane max with no cross-lane traffic.
#include "asr_kernels.h"
#include "../../include/asr_hip.h"
#include <algorithm>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace asr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float elu_f(float v) { return v > 0.0f ? v : expm1f(v); }
// ELU on the hot epilogue: exp(v) - 1 via v_exp_f32; |error| <= ~1.5e-7 absolute (the activations are O(1),
// the parity budget is 1e-4), 4 instructions instead of the ~40 of expm1f.
__device__ __forceinline__ float elu_fast(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }
// n / d for 0 <= n < 2^20, 1 <= d <= 2^12 with rcp = 1.0f / d: (n + 0.5) * rcp is at least 0.5/d away from
// an integer, the float error is < 1e-3 of that.
__device__ __forceinline__ int fdiv(int n, float rcp) { return (int)(((float)n + 0.5f) * rcp); }

// ---------------------------------------------------------------------------
// block 1
// ---------------------------------------------------------------------------
// tab (uint8 inputs only; may be null): 256-entry table of the correctly rounded quotients v / 255.0f.  The IEEE
// division costs ~10 VALU instructions per tap - with nine taps per pixel about as much as block 1's 108 FMAs.
template <int IN_MODE>
__device__ __forceinline__ float load_prepared(const void *in, size_t img_off_raw, int Wraw,
                                               int y, int x, int H, int W, int rsz, const float *tab = nullptr) {
    // returns the prepared pixel (y,x) of the network-resolution image, 0 outside
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.0f;
    if (IN_MODE == ASR_IN_F32_PREPARED) {
        return ((const float *)in)[img_off_raw + (size_t)y * W + x];
    }
    auto rawn = [&](int yy, int xx) -> float {          // raw value / 255 (model.prepare)
        if (IN_MODE == ASR_IN_U8_RAW) {
            const unsigned char v = ((const unsigned char *)in)[img_off_raw + (size_t)yy * Wraw + xx];
            return tab ? tab[v] : (float)v / 255.0f;
        }
        return ((const float *)in)[img_off_raw + (size_t)yy * Wraw + xx] / 255.0f;
    };
    if (!rsz) return rawn(y, x);
    // rsz prepare: /255, then bilinear factor-2 = (.5,.5) horizontally, then vertically
    const float a = rawn(2 * y, 2 * x), b = rawn(2 * y, 2 * x + 1);
    const float c = rawn(2 * y + 1, 2 * x), d = rawn(2 * y + 1, 2 * x + 1);
    const float top = a * 0.5f + b * 0.5f, bot = c * 0.5f + d * 0.5f;
    return top * 0.5f + bot * 0.5f;
}

// PX pixels (along x) per thread.  PX = 1: a wave's float4 stores are 48 B apart (2.7 lanes per 128-B line) instead
// of 192 B (one lane per line) - the store path, not the arithmetic, bounds this kernel.
template <int COUT, int IN_MODE, int PX>
__global__ __launch_bounds__(256) void conv1_kernel(const void *__restrict__ in, const float *__restrict__ w,
                                                    const float *__restrict__ bnp, float *__restrict__ out,
                                                    int N, int Hraw, int Wraw, int H, int W, int rsz, int ablate) {
    constexpr int COUTP = (COUT + 15) / 16 * 16;
    __shared__ float div255[256];
    if (IN_MODE == ASR_IN_U8_RAW) {
        div255[threadIdx.x] = (float)threadIdx.x / 255.0f;          // 256 threads: one exact quotient each
        __syncthreads();
    }
    const float *tab = (IN_MODE == ASR_IN_U8_RAW) ? div255 : nullptr;
    __shared__ __attribute__((aligned(16))) float wstage[PX == 1 ? 4 * 64 * COUT : 4];
    const int lane = threadIdx.x & 63;
    float *wbuf = wstage + (PX == 1 ? (threadIdx.x >> 6) * 64 * COUT : 0);
    const int xg_per_row = (W + PX - 1) / PX;
    const int64_t total = (int64_t)N * H * xg_per_row;
    // PX == 1: the trip count is wave-uniform (a wave's 64 pixels are processed together, lanes past the end idle).
    // The taps of the NEXT iteration are loaded before this iteration's stores are issued: vmcnt counts loads and
    // stores together and in order, so a load issued after the stores could only be waited for together with them -
    // every iteration then paid a full store round trip (measured: 2.8 TB/s of writes instead of the 6.6 TB/s a plain
    // fill reaches).
    const int64_t stride_s = (int64_t)gridDim.x * blockDim.x;
    auto decode = [&](int64_t s, int &n, int &y, int &x0) {
        const int64_t sd = s < total ? s : total - 1;   // idle tail lanes recompute the last pixel; their stores are masked
        if (total < (int64_t)1 << 31) {                 // wave-uniform: 32-bit divisions cost a fifth of the 64-bit ones
            const unsigned u = (unsigned)sd;
            const unsigned q = u / (unsigned)xg_per_row;
            n = (int)(q / (unsigned)H);
            y = (int)(q - (unsigned)n * (unsigned)H);
            x0 = (int)(u - q * (unsigned)xg_per_row) * PX;
            return;
        }
        const int xg = (int)(sd % xg_per_row);
        const int64_t q = sd / xg_per_row;
        y = (int)(q % H);
        n = (int)(q / H);
        x0 = xg * PX;
    };
    auto load_taps = [&](int64_t s, float (&v)[3][PX + 2]) {
        int n, y, x0;
        decode(s, n, y, x0);
        const size_t img_off = (IN_MODE == ASR_IN_F32_PREPARED) ? (size_t)n * H * W : (size_t)n * Hraw * Wraw;
        if (!rsz && !(ablate & 16)) {
            // branch-free: all taps are loaded from clamped (always valid) addresses first, the zero padding and the
            // /255 are applied afterwards - nine independent loads in flight instead of nine dependent round trips
            // (a bounds branch + table look-up per tap serialised them)
            const int Wsrc = (IN_MODE == ASR_IN_F32_PREPARED) ? W : Wraw;
            float raw[3][PX + 2];
            bool ok[3][PX + 2];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < PX + 2; ++b) {
                    const int yy = y - 1 + a, xx = x0 - 1 + b;
                    ok[a][b] = yy >= 0 && yy < H && xx >= 0 && xx < W;
                    const int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy), xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                    const size_t off = img_off + (size_t)yc * Wsrc + xc;
                    if (IN_MODE == ASR_IN_U8_RAW) raw[a][b] = (float)((const unsigned char *)in)[off];
                    else raw[a][b] = ((const float *)in)[off];
                }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < PX + 2; ++b) {
                    float val = raw[a][b];
                    if (IN_MODE == ASR_IN_U8_RAW) val = tab[(int)val];
                    else if (IN_MODE == ASR_IN_F32_RAW) val = val / 255.0f;
                    v[a][b] = ok[a][b] ? val : 0.0f;
                }
            return;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < PX + 2; ++b)
                v[a][b] = (ablate & 16) ? 0.25f * (a + b)
                                        : load_prepared<IN_MODE>(in, img_off, Wraw, y - 1 + a, x0 - 1 + b, H, W, rsz, tab);
    };
    const int64_t s_first = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
    float v[3][PX + 2], vn[3][PX + 2];
    if (s_first < total) load_taps(s_first + lane, v);
    for (int64_t s0 = s_first; s0 < total; s0 += stride_s) {
        const int64_t s = s0 + lane;
        const bool live = s < total;
        const bool more = s0 + stride_s < total;        // wave-uniform
        if (more) load_taps(s0 + stride_s + lane, vn);
        int n, y, x0;
        decode(s, n, y, x0);
        if (PX != 1 && !live) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < PX + 2; ++b) v[a][b] = vn[a][b];
            continue;
        }
        float *orow = out + (((size_t)n * H + y) * W + x0) * COUT;
        // channel groups of 4: the 36 taps + 12 BN values of a group are wave-uniform scalar loads;
        // keeping the group loop rolled bounds the live SGPRs (a full unroll spilled > 200 of them)
#pragma unroll 1
        for (int cg = 0; cg < COUT / 4; ++cg) {
            const float *wg = w + cg * 36;
            float res[PX][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int co = cg * 4 + c;
                const float mean = bnp[co], scale = bnp[COUTP + co], beta = bnp[2 * COUTP + co];
#pragma unroll
                for (int px = 0; px < PX; ++px) {
                    float acc = 0.0f;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) acc = fmaf(v[a][px + b], wg[c * 9 + a * 3 + b], acc);
                    res[px][c] = elu_fast((acc - mean) * scale + beta);
                }
            }
            if (ablate & 8) {        // diagnostics: keep the arithmetic, drop the stores
#pragma unroll
                for (int px = 0; px < PX; ++px) asm volatile("" ::"v"(res[px][0]), "v"(res[px][1]), "v"(res[px][2]), "v"(res[px][3]));
                continue;
            }
            if constexpr (PX == 1) {
                // the wave's 64 pixels are one contiguous run of 64 * COUT floats: park the channel groups in LDS
                // (lane-major), stream them out below as fully coalesced 1-KB store instructions
                *reinterpret_cast<float4 *>(wbuf + lane * COUT + cg * 4) = make_float4(res[0][0], res[0][1], res[0][2], res[0][3]);
            } else {
#pragma unroll
                for (int px = 0; px < PX; ++px)
                    if (x0 + px < W)
                        *reinterpret_cast<float4 *>(orow + (size_t)px * COUT + cg * 4) =
                            make_float4(res[px][0], res[px][1], res[px][2], res[px][3]);
            }
        }
        if constexpr (PX == 1) {
            if (!(ablate & 8)) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int64_t wave_px0 = s - lane;                        // first pixel of this wave's run
                float4 *dst = reinterpret_cast<float4 *>(out + (size_t)wave_px0 * COUT);
                const int64_t lim4 = (total - wave_px0) * (COUT / 4);     // float4 still inside the tensor
#pragma unroll
                for (int k = 0; k < COUT / 4; ++k) {
                    const int f = k * 64 + lane;
                    const float4 v4 = *reinterpret_cast<const float4 *>(wbuf + f * 4);
                    if (f < lim4) {
                        typedef float f4nt __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(f4nt{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<f4nt *>(dst + f));
                    }
                }
                __builtin_amdgcn_wave_barrier();                          // the buffer is rewritten by the next iteration
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < PX + 2; ++b) v[a][b] = vn[a][b];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// conv1_quad_kernel: block 1 for the common case (no rsz prepare, W a multiple of 4): a thread computes FOUR
// consecutive pixels of a row for all C_out channels.  conv1_kernel (one pixel per thread) is bounded by vector issue
// before HBM (ASR_ABLATE table in DESIGN.md: 0.233 ms of arithmetic for the 0.28 ms the stores need at the fill rate):
// per pixel it issues 9 tap loads with their address clamps, 9 table look-ups, the index decode and 108 scalar FMAs.
// Here a row of the 3 x 6 window is one 4-byte (uint8) or 16-byte (float) load plus two edge loads, the decode and the
// clamps are shared by four pixels, and the taps of a channel PAIR run as v_pk_fma_f32 (two IEEE FMAs per lane and
// instruction; same products, same summation order as conv1_kernel: the outputs are bit-identical).  A wave's 256
// pixels are 256 * C_out contiguous floats of the output: parked in LDS (lane-major) and written as 1-KB stores.
typedef float f2q __attribute__((ext_vector_type(2)));
// RSZ (the _rsz model's prepare on a raw image of exactly 2H x 2W): a window value is the 2x2 mean of load_prepared -
// same expression, same bits - and a window row is two raw rows of one 8-byte (uint8) / two 16-byte (float) loads plus
// two 2-pixel edge loads each.
template <int COUT, int IN_MODE, bool RSZ>
__global__ __launch_bounds__(256) void conv1_quad_kernel(const void *__restrict__ in, const float *__restrict__ w,
                                                         const float *__restrict__ bnp, float *__restrict__ out,
                                                         int N, int H, int W) {
    static_assert(!RSZ || IN_MODE != ASR_IN_F32_PREPARED, "a prepared image is at network resolution already");
    constexpr int COUTP = (COUT + 15) / 16 * 16;
    static_assert(COUT % 4 == 0, "channel groups of four");
    extern __shared__ __attribute__((aligned(16))) float c1lds[];
    float *div255 = c1lds;                                      // uint8 inputs: the exact quotients v / 255
    // Output staging, per wave: the 64 x 4 x COUT floats of its 256 pixels as float4 chunks (pixel of the quad px,
    // channel group cg) in rows [px * CG + cg] of 64 + 1 chunks, lane = column.  Round 3 parked them lane-major
    // ((lane * 4 + px) * COUT + co, scalar stores): lanes 48 floats apart hit 2 of the 32 banks - a 16-way conflict on
    // every one of the 48 ds_write_b32 per lane and iteration, 0.62 of the kernel's LDS cycles.  Now the four channels of
    // a group leave as ONE 16-byte store to consecutive chunks of consecutive lanes (conflict-free), and the read side
    // - which needs the chunks in output order, i.e. the 4 CG chunks of one lane after the other - finds them one row
    // (65 chunks) apart: row + lane runs through 16 different residues mod 16, conflict-free as well.
    constexpr int CG = COUT / 4, CPL = 4 * CG, ROWC = 65;
    float *wstage = c1lds + 256;                                // [4 waves][CPL rows][65 chunks][4 floats]
    if (IN_MODE == ASR_IN_U8_RAW) {
        for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) div255[i] = (float)i / 255.0f;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    float *wbuf = wstage + (threadIdx.x >> 6) * CPL * ROWC * 4;
    const int wq = W >> 2;                                      // quads per row
    const unsigned total = (unsigned)N * H * wq;                // the launcher admits N H W < 2^31 only
    const unsigned stride = gridDim.x * blockDim.x;           // 4 waves per workgroup, 2 at C_out = 24 (LDS per wave)
    for (unsigned q0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); q0 < total; q0 += stride) {
        const bool live = q0 + lane < total;
        const unsigned q = live ? q0 + lane : total - 1;       // idle tail lanes recompute the last quad, nothing is stored
        const unsigned r = q / (unsigned)wq;                    // image row index n * H + y
        const int x0 = (int)(q - r * (unsigned)wq) * 4;
        const unsigned n = r / (unsigned)H;
        const int y = (int)(r - n * (unsigned)H);
        // ---- the 3 x 6 window: rows clamped (masked afterwards), centre 4 values in one load, the two edge columns
        // from clamped addresses.  All loads are issued before the first use
        float v[3][6];
        {
            const bool lok = x0 > 0, rok = x0 + 4 < W;
            const int xl = lok ? x0 - 1 : 0, xr = rok ? x0 + 4 : W - 1;
            float rowm[3];
            if constexpr (RSZ) {
                const int Wr = 2 * W;
                auto mean4 = [](float a, float b, float c, float d) {        // load_prepared's rsz expression
                    const float top = a * 0.5f + b * 0.5f, bot = c * 0.5f + d * 0.5f;
                    return top * 0.5f + bot * 0.5f;
                };
                if (IN_MODE == ASR_IN_U8_RAW) {
                    const unsigned char *img = (const unsigned char *)in + (size_t)n * (4 * H * W);
                    uint2 cw[3][2];
                    unsigned short el[3][2], er[3][2];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int yy = y - 1 + a, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                        rowm[a] = yy == yc ? 1.0f : 0.0f;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned char *rp = img + (size_t)(2 * yc + h) * Wr;
                            cw[a][h] = *reinterpret_cast<const uint2 *>(rp + 2 * x0);
                            el[a][h] = *reinterpret_cast<const unsigned short *>(rp + 2 * xl);
                            er[a][h] = *reinterpret_cast<const unsigned short *>(rp + 2 * xr);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        v[a][0] = mean4(div255[el[a][0] & 255u], div255[el[a][0] >> 8], div255[el[a][1] & 255u], div255[el[a][1] >> 8]) *
                                  (lok ? rowm[a] : 0.0f);
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const unsigned t0 = (b < 2 ? cw[a][0].x : cw[a][0].y) >> (16 * (b & 1));
                            const unsigned t1 = (b < 2 ? cw[a][1].x : cw[a][1].y) >> (16 * (b & 1));
                            v[a][1 + b] = mean4(div255[t0 & 255u], div255[(t0 >> 8) & 255u], div255[t1 & 255u], div255[(t1 >> 8) & 255u]) * rowm[a];
                        }
                        v[a][5] = mean4(div255[er[a][0] & 255u], div255[er[a][0] >> 8], div255[er[a][1] & 255u], div255[er[a][1] >> 8]) *
                                  (rok ? rowm[a] : 0.0f);
                    }
                } else {
                    const float *img = (const float *)in + (size_t)n * (4 * H * W);
                    float4 cw[3][2][2];
                    float2 el[3][2], er[3][2];
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int yy = y - 1 + a, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                        rowm[a] = yy == yc ? 1.0f : 0.0f;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float *rp = img + (size_t)(2 * yc + h) * Wr;
                            cw[a][h][0] = *reinterpret_cast<const float4 *>(rp + 2 * x0);
                            cw[a][h][1] = *reinterpret_cast<const float4 *>(rp + 2 * x0 + 4);
                            el[a][h] = *reinterpret_cast<const float2 *>(rp + 2 * xl);
                            er[a][h] = *reinterpret_cast<const float2 *>(rp + 2 * xr);
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float t[2][12];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            t[h][0] = el[a][h].x; t[h][1] = el[a][h].y;
                            t[h][2] = cw[a][h][0].x; t[h][3] = cw[a][h][0].y; t[h][4] = cw[a][h][0].z; t[h][5] = cw[a][h][0].w;
                            t[h][6] = cw[a][h][1].x; t[h][7] = cw[a][h][1].y; t[h][8] = cw[a][h][1].z; t[h][9] = cw[a][h][1].w;
                            t[h][10] = er[a][h].x; t[h][11] = er[a][h].y;
#pragma unroll
                            for (int b = 0; b < 12; ++b) t[h][b] = t[h][b] / 255.0f;
                        }
#pragma unroll
                        for (int b = 0; b < 6; ++b)
                            v[a][b] = mean4(t[0][2 * b], t[0][2 * b + 1], t[1][2 * b], t[1][2 * b + 1]) *
                                      ((b == 0 && !lok) || (b == 5 && !rok) ? 0.0f : rowm[a]);
                    }
                }
            } else if (IN_MODE == ASR_IN_U8_RAW) {
                const unsigned char *img = (const unsigned char *)in + (size_t)n * H * W;
                unsigned cw[3];
                unsigned char el[3], er[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int yy = y - 1 + a, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                    rowm[a] = yy == yc ? 1.0f : 0.0f;
                    const unsigned char *rp = img + (size_t)yc * W;
                    cw[a] = *reinterpret_cast<const unsigned *>(rp + x0);
                    el[a] = rp[xl]; er[a] = rp[xr];
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    v[a][0] = div255[el[a]] * (lok ? rowm[a] : 0.0f);
#pragma unroll
                    for (int b = 0; b < 4; ++b) v[a][1 + b] = div255[(cw[a] >> (8 * b)) & 255u] * rowm[a];
                    v[a][5] = div255[er[a]] * (rok ? rowm[a] : 0.0f);
                }
            } else {
                const float *img = (const float *)in + (size_t)n * H * W;
                float4 cw[3];
                float el[3], er[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int yy = y - 1 + a, yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
                    rowm[a] = yy == yc ? 1.0f : 0.0f;
                    const float *rp = img + (size_t)yc * W;
                    cw[a] = *reinterpret_cast<const float4 *>(rp + x0);
                    el[a] = rp[xl]; er[a] = rp[xr];
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float t[6] = {el[a], cw[a].x, cw[a].y, cw[a].z, cw[a].w, er[a]};
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        if (IN_MODE == ASR_IN_F32_RAW) t[b] = t[b] / 255.0f;
                        v[a][b] = t[b] * ((b == 0 && !lok) || (b == 5 && !rok) ? 0.0f : rowm[a]);
                    }
                }
            }
        }
        // ---- channel groups of 4 (two pairs): taps and BN constants are wave-uniform scalar loads; the group loop
        // stays rolled to bound the live SGPRs
#pragma unroll 1
        for (int cg = 0; cg < COUT / 4; ++cg) {
            const float *wg = w + cg * 36;
            float res[4][4];                                    // [pixel of the quad][channel of the group]
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                const int co = cg * 4 + cp * 2;
                const f2q mean = {bnp[co], bnp[co + 1]}, scale = {bnp[COUTP + co], bnp[COUTP + co + 1]};
                const f2q beta = {bnp[2 * COUTP + co], bnp[2 * COUTP + co + 1]};
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    f2q acc = {0.0f, 0.0f};
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            const f2q wv = {wg[(cp * 2) * 9 + a * 3 + b], wg[(cp * 2 + 1) * 9 + a * 3 + b]};
                            const f2q tv = {v[a][px + b], v[a][px + b]};
                            acc = __builtin_elementwise_fma(tv, wv, acc);
                        }
                    const f2q yv = (acc - mean) * scale + beta;
                    res[px][cp * 2] = elu_fast(yv.x);
                    res[px][cp * 2 + 1] = elu_fast(yv.y);
                }
            }
#pragma unroll
            for (int px = 0; px < 4; ++px)
                *reinterpret_cast<float4 *>(wbuf + ((px * CG + cg) * ROWC + lane) * 4) =
                    make_float4(res[px][0], res[px][1], res[px][2], res[px][3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the wave's 256 pixels = COUT * 64 float4, contiguous in the output: COUT fully coalesced 1-KB stores
        float4 *dst = reinterpret_cast<float4 *>(out + (size_t)q0 * 4 * COUT);
        const unsigned lim4 = (total - q0) * (unsigned)COUT;    // float4 still inside the tensor (COUT per quad)
#pragma unroll
        for (int k = 0; k < COUT; ++k) {
            const unsigned fi = k * 64 + lane;                   // chunk fi of the output = chunk fi % CPL of lane fi / CPL
            const float4 v4 = *reinterpret_cast<const float4 *>(wbuf + ((fi % CPL) * ROWC + fi / CPL) * 4);
            if (fi < lim4) {
                typedef float f4nt __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store(f4nt{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<f4nt *>(dst + fi));
            }
        }
        __builtin_amdgcn_wave_barrier();                        // the buffer is rewritten by the next iteration
    }
}

template <int COUT>
static hipError_t launch_conv1_quad(hipStream_t s, const void *in, int in_mode, int rsz, const float *w,
                                    const float *bnp, float *out, int N, int H, int W) {
    const unsigned total = (unsigned)((int64_t)N * H * (W / 4));
    static const int per_cu = getenv("ASR_CONV1_QBLK") ? std::max(1, atoi(getenv("ASR_CONV1_QBLK"))) : 48;   // (measured: 12 -> 0.309 ms, 24 -> 0.285, 48 -> 0.267, 96 -> 0.275)
    // a wave parks 64 x 4 x C_out floats: at C_out = 24 four-wave workgroups (99 KB) would leave one per CU - two-wave
    // ones fit three (0.27 -> see DESIGN.md)
    static const int waves24 = getenv("ASR_CONV1_QWAVES") ? std::max(1, std::min(4, atoi(getenv("ASR_CONV1_QWAVES")))) : 2;
    const int waves = COUT > 12 ? waves24 : 4;
    const unsigned T = 64u * waves;
    const int blocks = (int)std::min<unsigned>((total + T - 1) / T, 256u * per_cu * (4 / waves));
    const size_t lds = (256 + waves * (4 * (COUT / 4)) * 65 * 4) * sizeof(float);        // per wave: 4 CG rows of 65 float4 chunks
#define ASR_C1Q(MODE, RSZ)                                                                                         \
    do {                                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv1_quad_kernel<COUT, MODE, RSZ>),              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
        conv1_quad_kernel<COUT, MODE, RSZ><<<blocks, T, lds, s>>>(in, w, bnp, out, N, H, W);                       \
    } while (0)
    switch (in_mode) {
        case ASR_IN_F32_PREPARED: ASR_C1Q(ASR_IN_F32_PREPARED, false); break;
        case ASR_IN_F32_RAW:
            if (rsz) ASR_C1Q(ASR_IN_F32_RAW, true);
            else ASR_C1Q(ASR_IN_F32_RAW, false);
            break;
        case ASR_IN_U8_RAW:
            if (rsz) ASR_C1Q(ASR_IN_U8_RAW, true);
            else ASR_C1Q(ASR_IN_U8_RAW, false);
            break;
        default: return hipErrorInvalidValue;
    }
#undef ASR_C1Q
    return hipGetLastError();
}
// which kernel block 1 runs: the quad form where it applies (ASR_CONV1_QUAD=0: never)
// (the rsz form: the raw image must be exactly 2H x 2W - the 8-byte row loads rely on it - and fit 32-bit indices)
static bool conv1_use_quad(int in_mode, int rsz, int N, int H, int W, int Hraw, int Wraw) {
    static const int use = getenv("ASR_CONV1_QUAD") ? atoi(getenv("ASR_CONV1_QUAD")) : 1;
    if (in_mode == ASR_IN_F32_PREPARED) rsz = 0;
    if (rsz && (Hraw != 2 * H || Wraw != 2 * W)) return false;
    return use && W % 4 == 0 && W >= 8 && (int64_t)N * H * W * (rsz ? 4 : 1) < ((int64_t)1 << 31);
}

template <int COUT>
static hipError_t launch_conv1_t(hipStream_t s, const void *in, int in_mode, int rsz, const float *w,
                                 const float *bnp, float *out, int N, int Hraw, int Wraw, int H, int W) {
    if (conv1_use_quad(in_mode, rsz, N, H, W, Hraw, Wraw))
        return launch_conv1_quad<COUT>(s, in, in_mode, in_mode == ASR_IN_F32_PREPARED ? 0 : rsz, w, bnp, out, N, H, W);
    static const int px1 = getenv("ASR_CONV1_PX") ? atoi(getenv("ASR_CONV1_PX")) : 1;
    const int PXr = px1 == 4 ? 4 : 1;
    const int64_t total = (int64_t)N * H * ((W + PXr - 1) / PXr);
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    if (blocks == 0) return hipSuccess;
    static const int ablate = getenv("ASR_ABLATE") ? atoi(getenv("ASR_ABLATE")) : 0;
    switch (in_mode) {
        case ASR_IN_F32_PREPARED:
            if (PXr == 4) conv1_kernel<COUT, ASR_IN_F32_PREPARED, 4><<<blocks, 256, 0, s>>>(in, w, bnp, out, N, Hraw, Wraw, H, W, rsz, ablate);
            else conv1_kernel<COUT, ASR_IN_F32_PREPARED, 1><<<blocks, 256, 0, s>>>(in, w, bnp, out, N, Hraw, Wraw, H, W, rsz, ablate);
            break;
        case ASR_IN_F32_RAW:
            if (PXr == 4) conv1_kernel<COUT, ASR_IN_F32_RAW, 4><<<blocks, 256, 0, s>>>(in, w, bnp, out, N, Hraw, Wraw, H, W, rsz, ablate);
            else conv1_kernel<COUT, ASR_IN_F32_RAW, 1><<<blocks, 256, 0, s>>>(in, w, bnp, out, N, Hraw, Wraw, H, W, rsz, ablate);
            break;
        case ASR_IN_U8_RAW:
            if (PXr == 4) conv1_kernel<COUT, ASR_IN_U8_RAW, 4><<<blocks, 256, 0, s>>>(in, w, bnp, out, N, Hraw, Wraw, H, W, rsz, ablate);
            else conv1_kernel<COUT, ASR_IN_U8_RAW, 1><<<blocks, 256, 0, s>>>(in, w, bnp, out, N, Hraw, Wraw, H, W, rsz, ablate);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

const char *conv1_symbol(int cout, int in_mode, int rsz, int N, int H, int W, int Hraw, int Wraw) {
    static const int px1 = getenv("ASR_CONV1_PX") ? atoi(getenv("ASR_CONV1_PX")) : 1;
    static char names[3][2][3][176];
    static bool init = false;
    if (!init) {
        for (int c = 0; c < 2; ++c)
            for (int m = 0; m < 3; ++m) {
                snprintf(names[0][c][m], sizeof names[0][c][m],
                         "void asr::conv1_kernel<%d, %d, %d>(void const*, float const*, float const*, float*, int, int, "
                         "int, int, int, int, int)", c ? 24 : 12, m, px1 == 4 ? 4 : 1);
                for (int r = 0; r < 2; ++r)
                    snprintf(names[1 + r][c][m], sizeof names[1 + r][c][m],
                             "void asr::conv1_quad_kernel<%d, %d, %s>(void const*, float const*, float const*, float*, int, "
                             "int, int)", c ? 24 : 12, m, r ? "true" : "false");
            }
        init = true;
    }
    const int form = !conv1_use_quad(in_mode, rsz, N, H, W, Hraw, Wraw) ? 0 : (rsz && in_mode != ASR_IN_F32_PREPARED) ? 2 : 1;
    return names[form][cout == 24 ? 1 : 0][in_mode < 0 || in_mode > 2 ? 0 : in_mode];
}

hipError_t launch_conv1(hipStream_t s, const void *in, int in_mode, int rsz, const float *w, const float *bnp,
                        float *out, int N, int Hraw, int Wraw, int H, int W, int cout) {
    if (cout == 12) return launch_conv1_t<12>(s, in, in_mode, rsz, w, bnp, out, N, Hraw, Wraw, H, W);
    if (cout == 24) return launch_conv1_t<24>(s, in, in_mode, rsz, w, bnp, out, N, Hraw, Wraw, H, W);
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// blocks 2..8
// ---------------------------------------------------------------------------
struct ConvArgs {
    const float *in;
    const float *wpk;
    const float *bnp;
    float *out;
    int N, H, W, OH, OW;
    int TH, TW, NI;
    int tiles_y, tiles_x, total_tiles;
    int ablate;            // diagnostics only (ASR_ABLATE): 1 skip epilogue, 2 skip staging, 4 skip MFMA loop
    // fused block 1 (FUSE1): the tile is produced from the raw view-1/2 input instead of read from `in`
    const void *raw;       // (N,Hraw,Wraw) uint8 / float32, or prepared float32 (N,H,W)
    const float *w1;       // [CIN][9] correlation-form taps of block 1
    const float *bn1;      // [3][CIN padded to 16] of block 1
    int in_mode, rsz, Hraw, Wraw;
};

// prepared input pixel with a run-time input mode (wave-uniform switch); 0 outside the image
__device__ __forceinline__ float load_prepared_rt(const void *in, int mode, int n, int y, int x, int H, int W,
                                                  int Hraw, int Wraw, int rsz) {
    const size_t off = (mode == ASR_IN_F32_PREPARED) ? (size_t)n * H * W : (size_t)n * Hraw * Wraw;
    if (mode == ASR_IN_F32_PREPARED) return load_prepared<ASR_IN_F32_PREPARED>(in, off, Wraw, y, x, H, W, rsz);
    if (mode == ASR_IN_F32_RAW) return load_prepared<ASR_IN_F32_RAW>(in, off, Wraw, y, x, H, W, rsz);
    return load_prepared<ASR_IN_U8_RAW>(in, off, Wraw, y, x, H, W, rsz);
}

// LDS pixel stride (floats) per C_in: multiple of 4 (float4 staging), chosen with the bank model of the
// A-fragment reads (16 pixels = 2 rows x 8 columns per half-wave): 12 -> 20 (2-way on ds_read_b32; 16 would
// be 8-way), 24 -> 28, 48 -> 56, 96 -> 112 (<= 2-way on ds_read_b64/b128).
__host__ __device__ constexpr int lds_pixel_stride(int cin) {
    return cin == 12 ? 20 : cin == 24 ? 28 : cin == 48 ? 56 : cin == 96 ? 112 : cin + 4;
}

template <int KS>
__device__ __forceinline__ void load_frag(const float *p, float (&af)[KS]) {
    if constexpr (KS % 4 == 0) {
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(p)[q];
            af[4 * q] = t.x; af[4 * q + 1] = t.y; af[4 * q + 2] = t.z; af[4 * q + 3] = t.w;
        }
    } else if constexpr (KS % 2 == 0) {
#pragma unroll
        for (int q = 0; q < KS / 2; ++q) {
            const float2 t = reinterpret_cast<const float2 *>(p)[q];
            af[2 * q] = t.x; af[2 * q + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < KS; ++q) af[q] = p[q];
    }
}

// CIN, COUT: channels; POOL: fuse the 2x2 max-pool; WN x WM waves per block
// (WN splits the C_out tiles, WM the M-tiles); MTW: M-tiles in flight per wave.
// RAW: store the plain convolution output (no BN / ELU / pool): train-mode forward (the batch statistics
// are not known yet) and the data-gradient convolution of the backward pass.
// FUSE1: this is block 2 and block 1 (C_in = 1: prepare + 3x3 stencil + BN + ELU, VALU) is evaluated while staging
// the tile - its (N,H,W,nf) output, the largest activation of the network, never exists in HBM.
template <int CIN, int COUT, bool POOL, int WN, int WM, int MTW, bool RAW = false, bool FUSE1 = false>
__global__ __launch_bounds__(64 * WN * WM) void conv3x3_mfma_kernel(ConvArgs a) {
    constexpr int KS = CIN / 4;              // k-steps (of 4 channels) per tap
    constexpr int NT = (COUT + 15) / 16;     // 16-wide C_out tiles
    constexpr int NTW = NT / WN;             // ... per wave
    constexpr int CS = lds_pixel_stride(CIN); // LDS pixel stride in floats (16-B aligned, bank-conflict model)
    constexpr int THREADS = 64 * WN * WM;
    constexpr int COUTP = NT * 16;
    static_assert(NT % WN == 0, "C_out tiles must split evenly over WN");
    static_assert(CIN % 4 == 0, "C_in must be a multiple of 4");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % WN;
    const int wm = wave / WN;
    const int g = lane >> 4;     // A/B: k index inside a k-step;  C/D: window inside the M-tile
    const int nn = lane & 15;    // A: pixel row of the M-tile;     B/C/D: channel column

    // ---- this wave's weight fragments -> registers (kept for the whole kernel)
    float wreg[NTW][9][KS];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < KS; ++j)
                wreg[nt][tap][j] = a.wpk[((size_t)((wn * NTW + nt) * 9 + tap) * KS + j) * 64 + lane];
    float bmean[NTW], bscale[NTW], bbeta[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int co = (wn * NTW + nt) * 16 + nn;
        bmean[nt] = RAW ? 0.f : a.bnp[co];
        bscale[nt] = RAW ? 1.f : a.bnp[COUTP + co];
        bbeta[nt] = RAW ? 0.f : a.bnp[2 * COUTP + co];
    }

    const int LW = a.TW + 2, LH = a.TH + 2;      // LDS tile incl. halo, in pixels
    const int WX = a.TW >> 1, WY = a.TH >> 1;    // pooling windows per tile
    const int win_per_img = WX * WY;
    const int nwin = win_per_img * a.NI;
    const int n_mt = (nwin + 3) >> 2;
    const int img_lds = LH * LW * CS;
    const int npix = a.NI * LH * LW;
    const float rcp_LW = 1.0f / (float)LW, rcp_LH = 1.0f / (float)LH;
    const float rcp_WX = 1.0f / (float)WX, rcp_win = 1.0f / (float)win_per_img;
    const float rcp_tx = 1.0f / (float)a.tiles_x, rcp_ty = 1.0f / (float)a.tiles_y;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        const int t2 = (a.tiles_x == 1) ? tile : tile / a.tiles_x;      // scalar, once per tile
        const int tx = tile - t2 * a.tiles_x;
        const int grp = (a.tiles_y == 1) ? t2 : t2 / a.tiles_y;
        const int ty = t2 - grp * a.tiles_y;
        const int y0 = ty * a.TH, x0 = tx * a.TW, n0 = grp * a.NI;

        if constexpr (FUSE1) {
            // phase A: prepared input patch with a 2-pixel halo -> LDS (after the tile)
            float *rawbuf = lds + (size_t)a.NI * img_lds;
            const int RW = LW + 2, RH = LH + 2;
            const float rcp_RW = 1.0f / (float)RW, rcp_RH = 1.0f / (float)RH;
            for (int e = tid; e < a.NI * RH * RW; e += THREADS) {
                const int q = fdiv(e, rcp_RW);
                const int col = e - q * RW;
                const int img = fdiv(q, rcp_RH);
                const int row = q - img * RH;
                const int n = n0 + img;
                rawbuf[e] = n < a.N ? load_prepared_rt(a.raw, a.in_mode, n, y0 + row - 2, x0 + col - 2, a.H, a.W,
                                                       a.Hraw, a.Wraw, a.rsz)
                                    : 0.0f;
            }
            __syncthreads();
            // phase B: block 1 for every tile pixel inside the image (zeros outside = block 2's zero padding)
            constexpr int P1 = (CIN + 15) / 16 * 16;
            for (int p = tid; p < npix; p += THREADS) {
                const int q = fdiv(p, rcp_LW);
                const int col = p - q * LW;
                const int img = fdiv(q, rcp_LH);
                const int row = q - img * LH;
                const int gy = y0 + row - 1, gx = x0 + col - 1, n = n0 + img;
                const bool ok = (n < a.N) && (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
                float4 *dst = reinterpret_cast<float4 *>(lds + (size_t)p * CS);
                float v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = rawbuf[(img * RH + row + t / 3) * RW + col + t % 3];
#pragma unroll 1
                for (int cg = 0; cg < CIN / 4; ++cg) {
                    float r[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int co = cg * 4 + c;
                        float acc1 = 0.0f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc1 = fmaf(v[t], a.w1[co * 9 + t], acc1);
                        r[c] = ok ? elu_fast((acc1 - a.bn1[co]) * a.bn1[P1 + co] + a.bn1[2 * P1 + co]) : 0.0f;
                    }
                    dst[cg] = make_float4(r[0], r[1], r[2], r[3]);
                }
            }
        }
        // ---- stage the input tile (zero outside the image / batch).  SU pixels per thread are loaded before the
        // first LDS write so that a thread keeps SU * C_in/4 independent 16-byte loads in flight.
        constexpr int SU = (CIN <= 24) ? 2 : 1;
        for (int p0 = tid; p0 < ((FUSE1 || (a.ablate & 2)) ? 0 : npix); p0 += SU * THREADS) {
            float4 buf[SU][CIN / 4];
            bool ok[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int p = p0 + u * THREADS;
                const int q = fdiv(p, rcp_LW);
                const int col = p - q * LW;
                const int img = fdiv(q, rcp_LH);
                const int row = q - img * LH;
                const int gy = y0 + row - 1, gx = x0 + col - 1, n = n0 + img;
                ok[u] = (p < npix) && (n < a.N) && (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
                const float4 *src =
                    reinterpret_cast<const float4 *>(a.in + (((size_t)n * a.H + gy) * a.W + gx) * CIN);
#pragma unroll
                for (int c4 = 0; c4 < CIN / 4; ++c4)
                    buf[u][c4] = ok[u] ? src[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int p = p0 + u * THREADS;
                if (p < npix) {
                    float4 *dst = reinterpret_cast<float4 *>(lds + (size_t)p * CS);
#pragma unroll
                    for (int c4 = 0; c4 < CIN / 4; ++c4) dst[c4] = buf[u][c4];
                }
            }
        }
        __syncthreads();

        // ---- implicit GEMM over this wave's M-tiles
        auto frag_base = [&](int mt0, int (&abase)[MTW]) {
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                int wdx = (mt0 + i) * 4 + (nn >> 2);          // A row nn: window nn>>2, pixel nn&3
                wdx = wdx < nwin ? wdx : nwin - 1;
                const int img = fdiv(wdx, rcp_win);
                const int rem = wdx - img * win_per_img;
                const int wy = fdiv(rem, rcp_WX);
                const int wx = rem - wy * WX;
                const int py = 2 * wy + ((nn & 3) >> 1), px = 2 * wx + (nn & 1);
                abase[i] = img * img_lds + (py * LW + px) * CS + g * KS;
            }
        };
        // ---- epilogue: BN (deterministic) + ELU (+ 2x2 max-pool), NHWC store.
        // Pooled blocks: ELU is monotone and BN is affine per channel, so
        //   max_r elu(bn(x_r)) = elu(bn(scale >= 0 ? max_r x_r : min_r x_r))
        // - one BN + one ELU per lane instead of four, same value.
        auto epilogue = [&](int mt0, floatx4 (&acc)[MTW][NTW]) {
            if (a.ablate & 1) {
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) asm volatile("" ::"v"(acc[i][nt]));
                return;
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int wdx = (mt0 + i) * 4 + g;             // C/D rows 4g..4g+3 = window g
                if (wdx >= nwin) continue;
                const int img = fdiv(wdx, rcp_win);
                const int rem = wdx - img * win_per_img;
                const int wy = fdiv(rem, rcp_WX);
                const int wx = rem - wy * WX;
                const int n = n0 + img;
                if (n >= a.N) continue;
                if (RAW) {
                    const int yb = y0 + 2 * wy, xb = x0 + 2 * wx;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int co = (wn * NTW + nt) * 16 + nn;
                        if (co >= COUT) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int y = yb + (r >> 1), x = xb + (r & 1);
                            if (y < a.H && x < a.W)
                                a.out[(((size_t)n * a.H + y) * a.W + x) * COUT + co] = acc[i][nt][r];
                        }
                    }
                } else if (POOL) {
                    const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
                    if (oy >= a.OH || ox >= a.OW) continue;
                    float *orow = a.out + (((size_t)n * a.OH + oy) * a.OW + ox) * COUT;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int co = (wn * NTW + nt) * 16 + nn;
                        if (co >= COUT) continue;
                        const floatx4 c4 = acc[i][nt];
                        const float hi = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
                        const float lo = fminf(fminf(c4[0], c4[1]), fminf(c4[2], c4[3]));
                        const float x = bscale[nt] >= 0.0f ? hi : lo;
                        orow[co] = elu_fast((x - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    }
                } else {
                    const int yb = y0 + 2 * wy, xb = x0 + 2 * wx;
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) {
                        const int co = (wn * NTW + nt) * 16 + nn;
                        if (co >= COUT) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int y = yb + (r >> 1), x = xb + (r & 1);
                            if (y < a.H && x < a.W)
                                a.out[(((size_t)n * a.H + y) * a.W + x) * COUT + co] =
                                    elu_fast((acc[i][nt][r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        }
                    }
                }
            }
        };
        for (int mt0 = wm * MTW; mt0 < ((a.ablate & 4) ? 0 : n_mt); mt0 += WM * MTW) {
            floatx4 acc[MTW][NTW];
            int abase[MTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) acc[i][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
            frag_base(mt0, abase);
            if constexpr (MTW * 9 * KS <= 112) {
                // small C_in: fetch the A fragments of ALL nine taps first (<= 112 VGPRs), then issue the MFMAs
                // back to back - one LDS wait per pass instead of one per tap
                float afa[9][MTW][KS];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int toff = ((tap / 3) * LW + (tap % 3)) * CS;
#pragma unroll
                    for (int i = 0; i < MTW; ++i) load_frag<KS>(lds + abase[i] + toff, afa[tap][i]);
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int j = 0; j < KS; ++j)
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int nt = 0; nt < NTW; ++nt)
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(afa[tap][i][j], wreg[nt][tap][j],
                                                                                  acc[i][nt], 0, 0, 0);
            } else {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int toff = ((tap / 3) * LW + (tap % 3)) * CS;
                    float af[MTW][KS];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) load_frag<KS>(lds + abase[i] + toff, af[i]);
#pragma unroll
                    for (int j = 0; j < KS; ++j)
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int nt = 0; nt < NTW; ++nt)
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][j], wreg[nt][tap][j],
                                                                                  acc[i][nt], 0, 0, 0);
                }
            }
            epilogue(mt0, acc);
        }
        __syncthreads();   // LDS is re-staged by the next tile
    }
}

// ---- instantiation table ----------------------------------------------------
struct ConvVariant {
    int cin, cout, pool, wn, wm, mtw;
    void (*kernel)(ConvArgs);
    const char *symbol;        // as rocprofv3 prints it
    int raw;
    int fuse1;
};
#define ASR_BOOLSTR_0 "false"
#define ASR_BOOLSTR_1 "true"
#define ASR_CONV_VARIANT(CIN, COUT, POOL, WN, WM, MTW)                                              \
    { CIN, COUT, POOL, WN, WM, MTW, conv3x3_mfma_kernel<CIN, COUT, (POOL != 0), WN, WM, MTW>,       \
      "void asr::conv3x3_mfma_kernel<" #CIN ", " #COUT ", " ASR_BOOLSTR_##POOL ", " #WN ", " #WM ", " #MTW \
      ", false, false>(asr::ConvArgs)", 0, 0 }
#define ASR_CONV_FUSED1(CIN, COUT, POOL, WN, WM, MTW)                                               \
    { CIN, COUT, POOL, WN, WM, MTW, conv3x3_mfma_kernel<CIN, COUT, (POOL != 0), WN, WM, MTW, false, true>, \
      "void asr::conv3x3_mfma_kernel<" #CIN ", " #COUT ", " ASR_BOOLSTR_##POOL ", " #WN ", " #WM ", " #MTW \
      ", false, true>(asr::ConvArgs)", 0, 1 }
#define ASR_CONV_RAW(CIN, COUT, WN, WM, MTW)                                                        \
    { CIN, COUT, 0, WN, WM, MTW, conv3x3_mfma_kernel<CIN, COUT, false, WN, WM, MTW, true>,          \
      "void asr::conv3x3_mfma_kernel<" #CIN ", " #COUT ", false, " #WN ", " #WM ", " #MTW           \
      ", true, false>(asr::ConvArgs)", 1, 0 }
static const ConvVariant g_variants[] = {
    // mutopia_ccal_cont (num_filters 12)
    ASR_CONV_VARIANT(12, 12, 1, 1, 4, 4),
    ASR_CONV_VARIANT(12, 24, 0, 2, 2, 2),
    ASR_CONV_VARIANT(24, 24, 1, 2, 2, 2),
    ASR_CONV_VARIANT(24, 48, 0, 3, 2, 2),
    ASR_CONV_VARIANT(48, 48, 1, 3, 2, 2),
    ASR_CONV_VARIANT(48, 48, 0, 3, 2, 2),
    // mutopia_ccal_cont_rsz (num_filters 24) adds
    ASR_CONV_VARIANT(48, 96, 0, 6, 1, 2),
    ASR_CONV_VARIANT(96, 96, 1, 6, 1, 2),
    ASR_CONV_VARIANT(96, 96, 0, 6, 1, 2),
    // higher-occupancy shapes of the small-K blocks (autotuner candidates)
    ASR_CONV_VARIANT(12, 12, 1, 1, 4, 2),
    ASR_CONV_VARIANT(12, 12, 1, 1, 4, 1),
    ASR_CONV_VARIANT(12, 12, 1, 1, 8, 1),
    ASR_CONV_VARIANT(12, 12, 1, 1, 8, 2),
    ASR_CONV_VARIANT(12, 24, 0, 2, 2, 1),
    ASR_CONV_VARIANT(12, 24, 0, 2, 4, 1),
    ASR_CONV_VARIANT(12, 24, 0, 1, 4, 2),
    ASR_CONV_VARIANT(12, 24, 0, 1, 4, 1),
    ASR_CONV_VARIANT(24, 24, 1, 2, 2, 1),
    ASR_CONV_VARIANT(24, 24, 1, 2, 4, 1),
    ASR_CONV_VARIANT(24, 24, 1, 1, 4, 1),
    ASR_CONV_VARIANT(24, 24, 1, 1, 4, 2),
    // block 1 fused into block 2 (deterministic path)
    ASR_CONV_FUSED1(12, 12, 1, 1, 4, 4),
    ASR_CONV_FUSED1(24, 24, 1, 2, 2, 2),
    // RAW epilogue: train-mode forward convolutions and data gradients (C_in/C_out swapped)
    ASR_CONV_RAW(12, 12, 1, 4, 4),
    ASR_CONV_RAW(12, 24, 2, 2, 2),
    ASR_CONV_RAW(24, 12, 1, 4, 2),
    ASR_CONV_RAW(24, 24, 2, 2, 2),
    ASR_CONV_RAW(24, 48, 3, 2, 2),
    ASR_CONV_RAW(48, 24, 2, 2, 2),
    ASR_CONV_RAW(48, 48, 3, 2, 2),
    ASR_CONV_RAW(48, 96, 6, 1, 2),
    ASR_CONV_RAW(96, 48, 3, 2, 1),
    ASR_CONV_RAW(96, 96, 6, 1, 2),
};
static const int g_num_variants = (int)(sizeof(g_variants) / sizeof(g_variants[0]));

static const int kLdsBudget = 64 * 1024;   // per block: >= 2 blocks per CU of the 160 KiB

// all feasible (TH, TW, NI) tilings of one block under an LDS budget, cheapest first by the issue/staging model
static void enumerate_v1(int vi, int H, int W, int lds_budget, std::vector<ConvPlan> &out) {
    const ConvVariant &v = g_variants[vi];
    const int cin = v.cin, cout = v.cout;
    const int cs = lds_pixel_stride(cin);
    const int slots = v.wm * v.mtw;               // M-tiles one pass of the block covers
    const int ktot = 9 * cin / 4 * ((cout + 15) / 16);   // MFMAs per M-tile over all waves' n-tiles
    const int He = (H + 1) & ~1, We = (W + 1) & ~1;
    for (int TH = 2; TH <= std::min(He, 64); TH += 2) {
        for (int TW = 2; TW <= std::min(We, 128); TW += 2) {
            const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
            const int per_img_lds = (TH + 2) * (TW + 2) * cs * 4 + (v.fuse1 ? (TH + 4) * (TW + 4) * 4 : 0);
            if (per_img_lds > lds_budget) continue;
            const int ni_max = (tiles_y == 1 && tiles_x == 1) ? std::min(16, lds_budget / per_img_lds) : 1;
            for (int NI = 1; NI <= ni_max; ++NI) {
                const int nwin = (TH / 2) * (TW / 2) * NI;
                const int n_mt = (nwin + 3) / 4;
                const int passes = (n_mt + slots - 1) / slots;
                // cost per image: MFMA issue slots (per-wave serial work) + staging traffic
                const double mfma = (double)passes * v.mtw * ktot / v.wn * 32.0;   // SIMD cycles per wave
                const double stage = v.fuse1 ? (double)NI * (TH + 2) * (TW + 2) * cin * 12.0 / (64.0 * v.wn * v.wm) * 4.0
                                             : (double)NI * (TH + 2) * (TW + 2) * cin * 4 / 24.0;   // ~24 B/clk/CU
                ConvPlan bp{};
                bp.cost = (mfma + stage + 600.0) * tiles_y * tiles_x / NI;
                bp.TH = TH; bp.TW = TW; bp.NI = NI;
                bp.tiles_y = tiles_y; bp.tiles_x = tiles_x;
                bp.lds_bytes = per_img_lds * NI;
                bp.cin = cin; bp.cout = cout; bp.pool = v.pool;
                bp.H = H; bp.W = W;
                bp.OH = v.pool ? H / 2 : H;
                bp.OW = v.pool ? W / 2 : W;
                bp.threads = 64 * v.wn * v.wm;
                bp.variant = vi;
                bp.symbol = v.symbol;
                bp.fuse1 = v.fuse1;
                out.push_back(bp);
            }
        }
    }
    std::sort(out.begin(), out.end(), [](const ConvPlan &x, const ConvPlan &y) { return x.cost < y.cost; });
}

static void finish_v1(ConvPlan &bp) {
    const ConvVariant &v = g_variants[bp.variant];
    // persistent grid = what is actually resident (registers, LDS, waves)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), bp.threads,
                                                     (size_t)bp.lds_bytes) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = std::max(1, std::min(4, (160 * 1024) / std::max(1, bp.lds_bytes)));
    }
    bp.blocks_per_cu = std::min(nb, 8);
}

static int find_v1(int cin, int cout, int pool, int raw, int fuse1 = 0) {
    if (raw) pool = 0;
    for (int i = 0; i < g_num_variants; ++i)
        if (g_variants[i].cin == cin && g_variants[i].cout == cout && g_variants[i].pool == pool &&
            g_variants[i].raw == raw && g_variants[i].fuse1 == fuse1)
            return i;
    return -1;
}

bool plan_conv(int cin, int cout, int pool, int H, int W, ConvPlan *plan, int raw, int fuse1) {
    const int vi = find_v1(cin, cout, pool, raw, fuse1);
    if (vi < 0) return false;
    std::vector<ConvPlan> c;
    enumerate_v1(vi, H, W, kLdsBudget, c);
    if (c.empty()) return false;
    ConvPlan bp = c[0];
    finish_v1(bp);
    if (getenv("ASR_DEBUG"))
        fprintf(stderr, "[asr] plan conv %d->%d pool=%d %dx%d: tile %dx%d x%d img, tiles %dx%d, lds %d B, %d thr, "
                        "%d blocks/CU\n", cin, cout, pool, H, W, bp.TH, bp.TW, bp.NI, bp.tiles_y, bp.tiles_x,
                bp.lds_bytes, bp.threads, bp.blocks_per_cu);
    *plan = bp;
    return true;
}

// candidates for the run-time autotuner (asr_api.hip): the cheapest few by the model under two LDS budgets
// (>= 2 resident workgroups per CU, and one big one), with distinct tile shapes
void conv_candidates_v1(int cin, int cout, int pool, int H, int W, int raw, int max_count, std::vector<ConvPlan> *out,
                        int fuse1) {
    if (raw) pool = 0;
    for (int vi = 0; vi < g_num_variants; ++vi) {
        const ConvVariant &v = g_variants[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool || v.raw != raw || v.fuse1 != fuse1) continue;
        for (int budget : {kLdsBudget, 40 * 1024, 150 * 1024}) {
            std::vector<ConvPlan> c;
            enumerate_v1(vi, H, W, budget, c);
            int taken = 0;
            for (auto &cand : c) {
                bool dup = false;
                for (auto &o : *out)
                    if (o.variant == cand.variant && o.TH == cand.TH && o.TW == cand.TW && o.NI == cand.NI) dup = true;
                if (dup) continue;
                finish_v1(cand);
                out->push_back(cand);
                if (++taken >= max_count) break;
            }
        }
    }
}

size_t conv_wpack_floats(int cin, int cout) { return (size_t)((cout + 15) / 16) * 9 * (cin / 4) * 64; }

void pack_conv_weights(const float *wcorr, int cin, int cout, float *wpk) {
    const int KS = cin / 4, NT = (cout + 15) / 16;
    for (int nt = 0; nt < NT; ++nt)
        for (int tap = 0; tap < 9; ++tap)
            for (int j = 0; j < KS; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, n = lane & 15;
                    const int ci = g * KS + j, co = nt * 16 + n;
                    wpk[((size_t)(nt * 9 + tap) * KS + j) * 64 + lane] =
                        co < cout ? wcorr[((size_t)tap * cin + ci) * cout + co] : 0.0f;
                }
}

hipError_t launch_conv(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk, const float *bnp,
                       float *out, int N, int num_cus, const Fuse1Args *f1) {
    const ConvVariant &v = g_variants[p.variant];
    ConvArgs a;
    a.raw = nullptr; a.w1 = nullptr; a.bn1 = nullptr; a.in_mode = 0; a.rsz = 0; a.Hraw = 0; a.Wraw = 0;
    if (v.fuse1) {
        if (!f1) return hipErrorInvalidValue;
        a.raw = f1->raw; a.w1 = f1->w1; a.bn1 = f1->bn1; a.in_mode = f1->in_mode; a.rsz = f1->rsz;
        a.Hraw = f1->Hraw; a.Wraw = f1->Wraw;
    }
    a.in = in; a.wpk = wpk; a.bnp = bnp; a.out = out;
    a.N = N; a.H = p.H; a.W = p.W; a.OH = p.OH; a.OW = p.OW;
    a.TH = p.TH; a.TW = p.TW; a.NI = p.NI;
    a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
    const int groups = (N + p.NI - 1) / p.NI;
    a.total_tiles = groups * p.tiles_y * p.tiles_x;
    if (a.total_tiles == 0) return hipSuccess;
    static const int ablate = getenv("ASR_ABLATE") ? atoi(getenv("ASR_ABLATE")) : 0;
    a.ablate = ablate;
    const int grid = std::min(a.total_tiles, num_cus * std::max(1, p.blocks_per_cu));
    hipLaunchKernelGGL(v.kernel, dim3(grid), dim3(p.threads), p.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace asr
