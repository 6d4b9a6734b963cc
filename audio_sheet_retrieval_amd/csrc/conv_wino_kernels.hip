// conv3x3_wino: conv block (3x3 conv + BN + ELU [+ 2x2 max-pool]) as Winograd F(2x2,3x3) on the fp32 MFMA (gfx950)
//
// Why: v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate, so the direct implicit-GEMM schedules (conv_kernels.hip,
// conv_v2/v3) are bounded by 9*C_in/4 MFMAs per 16 output pixels and n-tile.  F(2x2,3x3) needs 16 products per 2x2 output tile instead of 36: 16*C_in/4 MFMAs per 64 output pixels,
// 2.25x fewer, paid for with ~32 adds per input channel (input transform) and 24 adds per output channel (output
// transform) per tile - both lane-local in the MFMA operand / accumulator layouts used here:
//   M = 16 winograd tiles (an MY x MX arrangement of 2x2-pixel output tiles), N = 16 output channels, K = C_in;
//   one accumulator set per transform position p = 4*xi + nu:  acc[p] += V_p (16 tiles x C_in) * U_p (C_in x 16)
//   A operand: lane (m = lane%16, g = lane/16) holds V_p[tile m][channel c(s,g)] for k-step s.  Channels are taken
//              in blocks of 8: k-steps 2t and 2t+1 use channels 8t+2g and 8t+2g+1, so that ONE ds_read_b64 per patch
//              pixel fetches a lane's operands of two k-steps as a register pair, and the input transform of both
//              runs as packed fp32 adds (a 4-channel remainder, C_in = 12, is one more k-step read as b32);
//   D layout : lane (g, n) holds tiles 4g..4g+3, channel n, for every p: the output transform (and the 2x2 max-pool,
//              whose window IS the tile) never leaves the lane.
// Numerics: fp32 throughout (weights transformed once, in float64, rounded to fp32).  This is a different fp32
// summation order than the direct form, not a reduced precision: measured embedding differences stay at the 1e-7
// level of the direct kernels themselves (tolerance 1e-4).
//
// LDS layout of a patch (planned for conflict-free b64 reads, see wino_lds_layout): 16-byte chunks (4 channels of a
// pixel); pixel stride C4P chunks (odd), row pitch RP chunks, and rows whose index has bit 1 set start one chunk
// later.  Neighbouring winograd tiles are two pixels apart, so every linear layout puts the 16 tiles of an M-tile
// on even chunk slots only; the odd pixel stride spreads a tile row over the 8 even slots of the 64 banks and the
// one-chunk shift moves every second tile row to the odd ones.
//
// Work decomposition: persistent workgroups.  A workgroup keeps the transformed weights of its group of NTW n-tiles in
// LDS for its whole life and walks a contiguous range of regions (RY x RX output pixels of one image).  The input
// patch (+ one-pixel halo) of a region is double-buffered in LDS: the LDS-DMA of region k+1 is issued before the
// M-tiles of region k are computed, so HBM/L2 latency hides under the MFMA work; the waves split a region's M-tiles.
// The transforms do NOT hide under other waves' MFMAs: on gfx950 the fp32 MFMA issues at the packed-fp32 vector rate, so
// a SIMD's MFMA and VALU cycles add up (round-3 instruction accounting, DESIGN.md section 4: conv2 runs within 10 % of
// that sum); three waves per SIMD cover LDS and memory latency, not the transform arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "asr_kernels.h"
#include "repack_elems.inl"

#ifndef ASR_WINOG_ABL
#define ASR_WINOG_ABL 0      // timing experiments only (wrong results).  LDS form: 1 = no patch DMA after the first region,
                             // 16 = no B reads, 32 = no A (patch) reads, 64 = no stores; global-A form: 2 = no input loads
                             // after the first block, 4 = no B reads, 8 = no input transform, 128 = no stores,
                             // 256 = no input loads at all (constant patch)
#endif

namespace asr {

typedef float floatx4w __attribute__((ext_vector_type(4)));
typedef float float2w __attribute__((ext_vector_type(2)));

#ifndef ASR_WINO_ASMREAD
#define ASR_WINO_ASMREAD 1     // 0: the patch reads as plain loads (the compiler pairs them into ds_read2_b64)
#endif
#ifndef ASR_WINO_BUFDMA
#define ASR_WINO_BUFDMA 1      // 0: round 2's pointer form of the LDS-DMA (A/B timing: tools/ab_build.sh)
#endif

struct WinoArgs {
    const float *in;       // (N,H,W,CIN)
    const float *wpk;      // [CIN/4][16][4][coutp]: k-step s, position p, lane group g: U_p[c(s,g)][n]
    const float *bnp;      // [3][coutp]
    float *out;            // (N,OH,OW,COUT)
    int N, H, W, OH, OW;
    int RY, RX;            // output pixels per workgroup region (even)
    int MY, MX;            // winograd tiles per M-tile, MY*MX == 16
    int nmy, nmx;          // M-tiles per region
    int tiles_y, tiles_x;  // regions per image
    int coutp;             // 16 * n-tiles
    int C4P, RP;           // LDS patch layout: chunks (16 B) per pixel and per row
    int total;             // regions in the launch
    int per;               // regions per workgroup (contiguous range)
    double *stats;         // RAW only, may be null: per-wave [sum | sum of squares] of the outputs, [row][2][C_out]
    BnBwdFuse bf;          // RAW + stats, bf.z != null: the sums are those of a BatchNorm backward instead (asr_kernels.h)
    // producer-wave builds (PW > 0): block 1 is evaluated into the LDS patch, `in` is unused
    const void *raw;       // raw view-1 / view-2 input (N,1,Hraw,Wraw): uint8 or float32
    const float *w1;       // block 1: [C][9] correlation-form taps
    const float *bn1;      // block 1: [3][16-padded C] mean | gamma*inv_std | beta
};

// a - b on vector types.  (Tried: spelling it as v_pk_add_f32 with neg modifiers in inline asm, because the compiler
// expands a packed fsub into scalar v_sub_f32 - 55 fewer VALU instructions per M-tile, no measurable time, dropped.)
#ifndef ASR_WINO_PKADD
#define ASR_WINO_PKADD 1     // 0: spell the transform adds as scalar v_add_f32 / v_sub_f32 (measured: 1 % SLOWER, see padd)
#endif
__device__ __forceinline__ float ssub(float a, float b) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float2w psub(float2w a, float2w b) {
    if (ASR_WINO_PKADD) return a - b;
    return float2w{ssub(a[0], b[0]), ssub(a[1], b[1])};
}
__device__ __forceinline__ floatx4w psub(floatx4w a, floatx4w b) {
    if (ASR_WINO_PKADD) return a - b;
    return floatx4w{ssub(a[0], b[0]), ssub(a[1], b[1]), ssub(a[2], b[2]), ssub(a[3], b[3])};
}
// a + b on vector types.  The compiler lowers vector fadd / fsub to v_pk_add_f32, which MI355X_MICROARCH.md lists as
// an anti-lever beside bf16 MFMAs (2 v_pk_add_f32 per MFMA gap: +26 cycles against 2 scalar adds).  Measured here
// (round 2, tools/ab_flags.sh, -DASR_WINO_PKADD=0: every transform add as a scalar instruction the SLP vectoriser
// cannot re-pack): the scalar form is 1 % SLOWER on every Winograd layer (conv2 0.725 -> 0.732 ms, conv6 0.474 ->
// 0.485) - the fp32 MFMA occupies the vector pipe either way, so what counts is the instruction count.  Packed stays.
__device__ __forceinline__ float sadd(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float2w padd(float2w a, float2w b) {
    if (ASR_WINO_PKADD) return a + b;
    return float2w{sadd(a[0], b[0]), sadd(a[1], b[1])};
}
__device__ __forceinline__ floatx4w padd(floatx4w a, floatx4w b) {
    if (ASR_WINO_PKADD) return a + b;
    return floatx4w{sadd(a[0], b[0]), sadd(a[1], b[1]), sadd(a[2], b[2]), sadd(a[3], b[3])};
}

__device__ float4 g_wino_zero[4];       // zero block the border lanes of the LDS-DMA read

// BatchNorm statistics of a RAW convolution, gathered where the outputs are produced (train-mode forward,
// train_fwd_kernels.hip): every lane (g, n) keeps float64 sums of the values and squares it wrote for channel
// nt*16 + n; at the end of the kernel the four lane groups are added and the wave writes ONE row
// [sum(C_out) | sum of squares(C_out)] of the partial table that bn_stats_final_kernel reduces in row order - the
// separate pass that re-read every raw output (0.85 ms of the batch-512 step) is gone.  Which M-tiles a wave
// computes is a fixed function of the launch, so the result is deterministic.
template <int NTV>
__device__ __forceinline__ void wino_stats_store(double *stats, int row, int cout, int col0, int lane,
                                                 const double (&s1)[NTV], const double (&s2)[NTV]) {
    const int g = lane >> 4, n = lane & 15;
#pragma unroll
    for (int nt = 0; nt < NTV; ++nt) {
        double a = s1[nt], b = s2[nt];
        a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
        a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
        const int ch = col0 + nt * 16 + n;
        if (g == 0 && ch < cout) {
            stats[((size_t)row * 2) * cout + ch] = a;
            stats[((size_t)row * 2 + 1) * cout + ch] = b;
        }
    }
}

__device__ __forceinline__ float elu_fastw(float y) { return y > 0.0f ? y : __expf(y) - 1.0f; }

// CIN, COUT: channels; POOL: fused 2x2 max-pool; NTW: n-tiles per workgroup (grid.y walks the groups);
// KB: 8-channel blocks transformed and multiplied per chunk (bounds the registers of the transformed patch);
// WAVES per workgroup; MINW: waves per SIMD the register budget is set for; RMAX: 16-byte chunks of a region's patch
// each thread moves from global memory to LDS (the planner guarantees it suffices); RAW: plain convolution output.
// PW > 0: "block 1 inside block 2" by wave specialisation.  The workgroup has PW more waves, the PRODUCERS: while the
// WAVES consumer waves multiply region k out of one patch buffer, the producers evaluate block 1 (C_in = 1: nine taps
// of the raw image, 9*CIN FMAs, BN, ELU per pixel - pure VALU work) for the patch of region k+1 straight into the
// other buffer, in the planned (padded, shifted) layout; the region barrier that used to wait for the LDS-DMA hands
// the buffer over.  Block 1's activation - the largest tensor of the network, 1.5 GB per 1000 sheets written and
// read back - never exists in HBM, and its arithmetic runs in the issue slots the consumers leave idle (the fp32
// MFMA pipe is ~45 % busy in this kernel; a producer wave never issues an MFMA).  IN_MODE: ASR_IN_* of the raw input.
template <int CIN, int COUT, bool POOL, int NTW, int KB, int WAVES, int MINW, int RMAX, bool RAW, int PW = 0,
          int IN_MODE = 0>
__global__ __launch_bounds__(64 * (WAVES + PW), MINW) void conv3x3_wino(WinoArgs a) {
    constexpr int KS = CIN / 4;            // k-steps
    constexpr int NB = CIN / 8;            // 8-channel blocks (two k-steps each)
    constexpr bool REM = (CIN % 8) != 0;   // one more k-step on the last 4 channels
    constexpr int NCH = NB / KB;
    constexpr int WROW = NTW * 16;
    constexpr int WS = NTW == 2 ? 48 : WROW;     // LDS row stride of the weights (32 would alias lane groups on banks)
    constexpr int C4 = CIN / 4;
    constexpr int T = 64 * (WAVES + PW);
    static_assert(NB % KB == 0, "chunking must divide the channel blocks");
    static_assert(PW == 0 || (!RAW && NTW * 16 >= COUT), "producer builds: deterministic path, one n-group");
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: M-tile index math stays scalar
    const int LW = a.RX + 2, LH = a.RY + 2;
    constexpr int C4P = (C4 & 1) ? C4 : C4 + 1;                    // chunks per pixel in LDS (odd; == a.C4P)
    constexpr int CSf = C4P * 4;                                   // pixel stride in floats: ds_read immediates
    const int RPf = a.RP * 4;                                      // row pitch in floats
    const int buf_floats = (LH * RPf + 255) & ~255;                // whole 1-KiB wave-instructions
    float *w_lds = lds + 2 * buf_floats;
    const int ng = blockIdx.y;
    const int first = blockIdx.x * a.per;
    const int last = min(first + a.per, a.total);
    if (first >= last) return;

    // ---- the transformed weights of this n-group, once: 16*KS*4 rows of WROW floats
    for (int i = tid; i < 16 * KS * 4 * (WROW / 4); i += T) {
        const int row = i / (WROW / 4), q = i - row * (WROW / 4);
        const int col = ng * WROW + q * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < a.coutp) v = *reinterpret_cast<const float4 *>(a.wpk + (size_t)row * a.coutp + col);
        *reinterpret_cast<float4 *>(w_lds + row * WS + q * 4) = v;
    }
    // ---- which chunks of a region's patch this thread moves (the same for every region): offset from the region's
    // first output pixel in global memory and (row, column) for the border test; -1 marks padding chunks.  The copy
    // itself is an LDS-DMA (global_load_lds_dwordx4): no data registers, the wave's 64 x 16 B land contiguously in
    // LDS, and every lane picks its own source - that is how the padded, shifted layout is produced.  Lanes outside
    // the image (and padding) read a zero block
    constexpr int RMAXE = PW > 0 ? 1 : RMAX;
    int st_g[RMAXE], st_rc[RMAXE];
    if constexpr (PW == 0) {
#pragma unroll
        for (int k = 0; k < RMAX; ++k) {
            const int f = tid + k * T;
            const int r = f / a.RP;
            const int xs = f - r * a.RP - ((r >> 1) & 1);
            const int px = xs / C4P, c = xs - px * C4P;
            const bool real = r < LH && xs >= 0 && px < LW && c < C4;
#if ASR_WINO_BUFDMA
            st_g[k] = real ? (((r - 1) * a.W + (px - 1)) * CIN + c * 4) * 4 : (int)0x80000000;      // BYTES; padding: out of range
#else
            st_g[k] = ((r - 1) * a.W + (px - 1)) * CIN + c * 4;
#endif
            st_rc[k] = real ? (r << 16) | px : -1;
        }
    }
    auto fetch = [&](int region, float *buf) {
        if constexpr (PW == 0) {
        const int tx = region % a.tiles_x;
        const int rest = region / a.tiles_x;
        const int ty = rest % a.tiles_y;
        const int img = rest / a.tiles_y;
        const int Y0 = ty * a.RY, X0 = tx * a.RX;
#if ASR_WINO_BUFDMA
        // Round 3: the copy as BUFFER loads (buffer_load_dwordx4 ... lds) through a descriptor of THIS image: a chunk's
        // address is "region offset (scalar) + this thread's constant" - one v_add per copy instead of a 64-bit
        // pointer, two selects and the border tests - and the hardware's range check supplies the zeros: rows above
        // the image give a negative (= huge unsigned) offset, rows below it an offset past num_records, padding chunks
        // carry 0x80000000.  Only columns left / right of the image need a per-lane test, and only in the first and
        // last region column.  (With 32x8-pixel regions 45 % of a 160x200 map's regions touch its border; the pointer
        // form spent ~70 vector instructions per border region and wave, ~25 per interior one.)
        {                                       // (the launcher admits images below 2 GiB only)
            // the descriptor by hand and the load as inline assembly: with __builtin_amdgcn_raw_ptr_buffer_load_lds in
            // this function template the HOST pass of hipcc (ROCm 7.2) silently emits no kernel stub for the
            // instantiation - an unresolved symbol when the library is loaded.  M0 = LDS address of the wave's 1 KiB;
            // one wait state between the scalar write of M0 and the LDS-DMA that reads it.
            typedef int int4v __attribute__((ext_vector_type(4)));
            const uint64_t ibase = reinterpret_cast<uint64_t>(a.in + (int64_t)img * a.H * a.W * CIN);
            int4v rsrc;
            rsrc.x = (int)(uint32_t)ibase;
            rsrc.y = (int)(uint32_t)(ibase >> 32);                                // stride 0: a raw buffer
            rsrc.z = a.H * a.W * CIN * 4;                                         // num_records in bytes
            rsrc.w = 0x00020000;
            const int roff = (Y0 * a.W + X0) * CIN * 4;
            const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)buf;
            const bool xedge = X0 < 1 || X0 + LW - 1 > a.W;                       // wave-uniform
            auto dma = [&](int k, int vo) {
                const unsigned m0v = lds0 + (unsigned)((k * T + wave * 64) * 16);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                             ::"v"(vo), "s"(rsrc), "s"(m0v)
                             : "memory", "m0");
            };
            if (!__builtin_amdgcn_readfirstlane((int)xedge)) {                    // a real branch: one v_add per copy
#pragma unroll
                for (int k = 0; k < RMAX; ++k) {
                    if ((k * T + wave * 64) * 4 >= buf_floats) break;             // wave-uniform: nothing left for this wave
                    dma(k, st_g[k] + roff);
                }
                return;
            }
#pragma unroll
            for (int k = 0; k < RMAX; ++k) {
                if ((k * T + wave * 64) * 4 >= buf_floats) break;
                const int x = X0 - 1 + (st_rc[k] & 0xffff);
                dma(k, (st_rc[k] >= 0 && (unsigned)x < (unsigned)a.W) ? st_g[k] + roff : (int)0x80000000);
            }
            return;
        }
#endif
        const float *gbase = a.in + (((int64_t)img * a.H + Y0) * a.W + X0) * CIN;
        if (!(ASR_WINOG_ABL & 512) && Y0 >= 1 && Y0 + LH - 1 <= a.H && X0 >= 1 && X0 + LW - 1 <= a.W) {
            // patch and halo inside the image (most regions): no border tests - they are ~15 vector instructions per
            // copy, which the fp32 MFMAs of the same SIMD do not overlap
#pragma unroll
            for (int k = 0; k < RMAX; ++k) {
                if ((k * T + wave * 64) * 4 >= buf_floats) break;          // wave-uniform: nothing left for this wave
                const float *src = st_rc[k] >= 0 ? gbase + st_g[k] : reinterpret_cast<const float *>(g_wino_zero);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(buf + (k * T + wave * 64) * 4),
                                                 16, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < RMAX; ++k) {
            if ((k * T + wave * 64) * 4 >= buf_floats) break;              // wave-uniform: nothing left for this wave
            const int y = Y0 - 1 + (st_rc[k] >> 16), x = X0 - 1 + (st_rc[k] & 0xffff);
            const bool ok = st_rc[k] >= 0 && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const float *src = ok ? gbase + st_g[k] : reinterpret_cast<const float *>(g_wino_zero);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(buf + (k * T + wave * 64) * 4),
                                             16, 0, 0);
        }
        }
    };
    // ---- producer waves: block 1 of one region's patch (output pixels [-1, R+1) of the region) into `buf`.  A lane
    // takes patch pixels ptid, ptid + 64 PW, ...; pixels outside the image are block 2's zero padding.
    float *tab255 = w_lds + 16 * KS * 4 * WS;                      // uint8 input: the exact quotients v / 255
    auto produce = [&](int region, float *buf) {
        if constexpr (PW > 0) {
        const int ptid = tid - 64 * WAVES;
        const int tx = region % a.tiles_x;
        const int rest = region / a.tiles_x;
        const int ty = rest % a.tiles_y;
        const int img = rest / a.tiles_y;
        const int Y0 = ty * a.RY, X0 = tx * a.RX;
        const size_t ioff = (size_t)img * a.H * a.W;
        const int npx = LH * LW;
        const float rcpLW = 1.0f / (float)LW;
        for (int f = ptid; f < npx; f += 64 * PW) {
            const int r = (int)(((float)f + 0.5f) * rcpLW), px = f - r * LW;
            const int y = Y0 - 1 + r, x = X0 - 1 + px;
            float *dst = buf + (r * a.RP + ((r >> 1) & 1) + px * C4P) * 4;
            if (y < 0 || y >= a.H || x < 0 || x >= a.W) {
#pragma unroll
                for (int c = 0; c < C4; ++c) *reinterpret_cast<float4 *>(dst + c * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            // nine taps from clamped (always valid) addresses, zero padding and /255 applied afterwards
            float v[9];
#pragma unroll
            for (int aa = 0; aa < 3; ++aa)
#pragma unroll
                for (int bb = 0; bb < 3; ++bb) {
                    const int yy = y - 1 + aa, xx = x - 1 + bb;
                    const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                    const int yc = min(max(yy, 0), a.H - 1), xc = min(max(xx, 0), a.W - 1);
                    const size_t off = ioff + (size_t)yc * a.W + xc;
                    float val;
                    if (IN_MODE == 2) val = tab255[((const unsigned char *)a.raw)[off]];
                    else if (IN_MODE == 1) val = ((const float *)a.raw)[off] / 255.0f;
                    else val = ((const float *)a.raw)[off];
                    v[aa * 3 + bb] = ok ? val : 0.0f;
                }
            constexpr int C1P = (CIN + 15) / 16 * 16;
#pragma unroll 1
            for (int cg = 0; cg < C4; ++cg) {                  // rolled: the 36 taps + 12 BN values are scalar loads
                const float *wg = a.w1 + cg * 36;
                float res[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int co = cg * 4 + c;
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) acc = fmaf(v[t], wg[c * 9 + t], acc);
                    res[c] = elu_fastw((acc - a.bn1[co]) * a.bn1[C1P + co] + a.bn1[2 * C1P + co]);
                }
                *reinterpret_cast<float4 *>(dst + cg * 4) = make_float4(res[0], res[1], res[2], res[3]);
            }
        }
        }
    };
    const bool producer = PW > 0 && wave >= WAVES;                 // wave-uniform
    if constexpr (PW > 0) {
        if (IN_MODE == 2)
            for (int i = tid; i < 256; i += T) tab255[i] = (float)i / 255.0f;
        __syncthreads();                                           // table and weights in place
        if (producer) produce(first, lds);
    } else {
        fetch(first, lds);
#if ASR_WINO_BUFDMA
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    __syncthreads();

    // ---- per-lane constants
    const int m = lane & 15, g = lane >> 4, n = lane & 15;
    const int tyy = m / a.MX, txx = m - tyy * a.MX;
    // patch row i of this lane's tile sits in LDS row (M-tile row origin, a multiple of 4) + 2*tyy + i, whose
    // one-chunk shift is ((tyy + (i >> 1)) & 1)
    int a_row[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        a_row[i] = (2 * tyy + i) * RPf + 2 * txx * CSf + ((tyy + (i >> 1)) & 1) * 4;
    const int a_pair = (g >> 1) * 4 + (g & 1) * 2;       // block t: channels 8t + 2g, 8t + 2g + 1
    const int a_rem = NB * 8 + g;                        // remainder k-step: channel 8*NB + g
    const float *w_lane = w_lds + g * WS + n;
    // B operand q of a run of k-steps: k-step q / (16 NTW), position (q / NTW) % 16, n-tile q % NTW
    auto wq_off = [](int q) { return ((q / (16 * NTW)) * 16 + (q / NTW) % 16) * 4 * WS + (q % NTW) * 16; };
    // accumulator element r of this lane belongs to tile 4g + r of the M-tile: its coordinates (in tiles) and the
    // offset of its output (pooled: one pixel; else the tile's top-left pixel) from the M-tile's output origin
    int ey[4], ex[4], eoff[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int me = 4 * g + r;
        ey[r] = me / a.MX;
        ex[r] = me - ey[r] * a.MX;
        eoff[r] = (POOL ? ey[r] * a.OW + ex[r] : 2 * ey[r] * a.W + 2 * ex[r]) * COUT + ng * WROW + n;
    }
    float bmean[NTW], bscale[NTW], bbeta[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int ch = ng * WROW + nt * 16 + n;
        const bool ok = !RAW && ch < COUT;
        bmean[nt] = ok ? a.bnp[ch] : 0.f;
        bscale[nt] = ok ? a.bnp[a.coutp + ch] : 1.f;
        bbeta[nt] = ok ? a.bnp[2 * a.coutp + ch] : 0.f;
    }
    float bistd[NTW];                        // BatchNorm-backward sums (a.bf): mu, gamma*inv_std, beta, inv_std per channel
    const bool bnb = ASR_BNB_FUSE_BUILD && RAW && a.stats && a.bf.z != nullptr;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int ch = ng * WROW + nt * 16 + n;
        bistd[nt] = 0.f;
        if (bnb && ch < COUT) {
            bmean[nt] = a.bf.cst[ch]; bistd[nt] = a.bf.cst[COUT + ch];
            bscale[nt] = a.bf.cst[2 * COUT + ch]; bbeta[nt] = a.bf.cst[3 * COUT + ch];
        }
    }

    double st1[NTW], st2[NTW];               // RAW + a.stats: running sums of this lane's channel(s)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { st1[nt] = 0.0; st2[nt] = 0.0; }
    for (int region = first; region < last; ++region) {
    const float *in_lds = lds + ((region - first) & 1) * buf_floats;
    if (region + 1 < last && !(ASR_WINOG_ABL & 1)) {
        if constexpr (PW > 0) {
            if (producer) produce(region + 1, lds + ((region + 1 - first) & 1) * buf_floats);
        } else {
            fetch(region + 1, lds + ((region + 1 - first) & 1) * buf_floats);
        }
    }
    const int tx = region % a.tiles_x;
    const int rest = region / a.tiles_x;
    const int ty = rest % a.tiles_y;
    const int img = rest / a.tiles_y;
    const int Y0 = ty * a.RY, X0 = tx * a.RX;
    for (int mt = producer ? a.nmy * a.nmx : wave; mt < a.nmy * a.nmx; mt += WAVES) {      // consumers only
        const int mty = mt / a.nmx, mtx = mt - mty * a.nmx;
        const int oy = mty * a.MY * 2, ox = mtx * a.MX * 2;
        if (Y0 + oy >= a.H || X0 + ox >= a.W) continue;          // wave-uniform: M-tile outside the image
        const float *ap = in_lds + oy * RPf + ox * CSf;
        floatx4w acc[16][NTW];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[p][nt] = floatx4w{0.f, 0.f, 0.f, 0.f};

#pragma unroll
        for (int ch = 0; ch < NCH + (REM ? 1 : 0); ++ch) {
            if (ch < NCH) {
                // KB channel blocks: per patch pixel one b64 read = the operands of two k-steps, transformed together
                float2w dp[4][4][KB];
#if ASR_WINO_ASMREAD
                // The patch reads as explicit ds_read_b64: the compiler pairs neighbouring reads into ds_read2_b64, which
                // this LDS serves at half the rate of two ds_read_b64 (8 cycles against 2 + 2, MI355X_MICROARCH.md) and
                // in 16-lane groups on 32 banks - not the 32-lane / 64-bank pattern the patch layout is planned for
                // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of this kernel: 0.33 with the paired form).  The wait is
                // part of the data flow: the values leave the second statement, so no use can be scheduled above it.
                if (ASR_WINOG_ABL & 32) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int c = 0; c < KB; ++c) dp[i][j][c] = float2w{(float)i, (float)j};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const unsigned la = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)(ap + a_row[i] + a_pair);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int c = 0; c < KB; ++c)
                                asm volatile("ds_read_b64 %0, %1 offset:%2"
                                             : "=v"(dp[i][j][c])
                                             : "v"(la), "n"((j * CSf + (ch * KB + c) * 8) * 4));
                    }
#pragma unroll
                    for (int c = 0; c < KB; ++c)
                        asm volatile("s_waitcnt lgkmcnt(0)"
                                     : "+v"(dp[0][0][c]), "+v"(dp[0][1][c]), "+v"(dp[0][2][c]), "+v"(dp[0][3][c]),
                                       "+v"(dp[1][0][c]), "+v"(dp[1][1][c]), "+v"(dp[1][2][c]), "+v"(dp[1][3][c]),
                                       "+v"(dp[2][0][c]), "+v"(dp[2][1][c]), "+v"(dp[2][2][c]), "+v"(dp[2][3][c]),
                                       "+v"(dp[3][0][c]), "+v"(dp[3][1][c]), "+v"(dp[3][2][c]), "+v"(dp[3][3][c]));
                }
#else
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int c = 0; c < KB; ++c)
                            dp[i][j][c] = (ASR_WINOG_ABL & 32) ? float2w{(float)i, (float)j}
                                                               : *reinterpret_cast<const float2w *>(ap + a_row[i] + j * CSf + (ch * KB + c) * 8 + a_pair);
#endif
#pragma unroll
                for (int c = 0; c < KB; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {              // columns: B^T d
                        const float2w d0 = dp[0][j][c], d1 = dp[1][j][c], d2 = dp[2][j][c], d3 = dp[3][j][c];
                        dp[0][j][c] = psub(d0, d2); dp[1][j][c] = padd(d1, d2); dp[2][j][c] = psub(d2, d1); dp[3][j][c] = psub(d1, d3);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {              // rows: (B^T d) B
                        const float2w t0 = dp[i][0][c], t1 = dp[i][1][c], t2 = dp[i][2][c], t3 = dp[i][3][c];
                        dp[i][0][c] = psub(t0, t2); dp[i][1][c] = padd(t1, t2); dp[i][2][c] = psub(t2, t1); dp[i][3][c] = psub(t1, t3);
                    }
                }
                // B operands in a rolling window WD ahead of their MFMA, pinned against the compiler's sinking
                // (see conv3x3_winog)
                constexpr int WQ = 2 * KB * 16 * NTW, WD = 4 * NTW;
                const float *wk = w_lane + (ch * 2 * KB) * (16 * 4 * WS);
                float wv[WQ + WD];
#pragma unroll
                for (int q = 0; q < WD; ++q) wv[q] = wk[wq_off(q)];
#pragma unroll
                for (int q0 = 0; q0 < WQ; q0 += NTW) {
#pragma unroll
                    for (int q = q0; q < q0 + NTW; ++q)
                        wv[q + WD] = (ASR_WINOG_ABL & 16) ? wv[q] : (q + WD < WQ) ? wk[wq_off(q + WD)] : 0.f;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = q0; q < q0 + NTW; ++q) {
                        const int p = (q / NTW) % 16, c = q / (16 * NTW), nt = q % NTW;
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[p >> 2][p & 3][c >> 1][c & 1], wv[q],
                                                                          acc[p][nt], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                float dsg[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dsg[i][j] = (ASR_WINOG_ABL & 32) ? (float)(i + j) : ap[a_row[i] + j * CSf + a_rem];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d0 = dsg[0][j], d1 = dsg[1][j], d2 = dsg[2][j], d3 = dsg[3][j];
                    dsg[0][j] = d0 - d2; dsg[1][j] = d1 + d2; dsg[2][j] = d2 - d1; dsg[3][j] = d1 - d3;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t0 = dsg[i][0], t1 = dsg[i][1], t2 = dsg[i][2], t3 = dsg[i][3];
                    dsg[i][0] = t0 - t2; dsg[i][1] = t1 + t2; dsg[i][2] = t2 - t1; dsg[i][3] = t1 - t3;
                }
#pragma unroll
                for (int p = 0; p < 16; ++p)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt)
                        acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            dsg[p >> 2][p & 3], w_lane[((2 * NB * 16 + p) * 4) * WS + nt * 16], acc[p][nt], 0, 0, 0);
            }
        }

        // ---- output transform Y = A^T M A for the lane's four tiles at once (float4 = tiles r = 0..3), then
        // BN + ELU (+ pool) and the store
        const int py0 = Y0 + oy, px0 = X0 + ox;              // pixel origin of this M-tile (wave-uniform)
        float *obase = POOL ? a.out + (((int64_t)img * a.OH + (py0 >> 1)) * a.OW + (px0 >> 1)) * COUT
                            : a.out + (((int64_t)img * a.H + py0) * a.W + px0) * COUT;
        const int ly = POOL ? a.OH - (py0 >> 1) : (a.H - py0 + 1) >> 1;      // tiles with at least one row inside
        const int lx = POOL ? a.OW - (px0 >> 1) : (a.W - px0 + 1) >> 1;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            if (ng * WROW + nt * 16 + n >= COUT) continue;
            floatx4w s0[4], s1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                s0[nu] = padd(padd(acc[nu][nt], acc[4 + nu][nt]), acc[8 + nu][nt]);
                s1[nu] = psub(psub(acc[4 + nu][nt], acc[8 + nu][nt]), acc[12 + nu][nt]);
            }
            const floatx4w y00 = padd(padd(s0[0], s0[1]), s0[2]), y01 = psub(psub(s0[1], s0[2]), s0[3]);
            const floatx4w y10 = padd(padd(s1[0], s1[1]), s1[2]), y11 = psub(psub(s1[1], s1[2]), s1[3]);
            // M-tiles that lie inside the image completely (wave-uniform; all but those of the last region row / column)
            // store without per-lane edge tests: each test is a compare, an exec-mask update and a branch around ONE
            // store, 4-5 vector / scalar instructions for every value written (0.09 of conv3's 0.45 ms went to its stores)
            const bool full = POOL ? (a.MY <= ly && a.MX <= lx) : (py0 + 2 * a.MY <= a.H && px0 + 2 * a.MX <= a.W);
            if (full && !(ASR_WINOG_ABL & 64)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *o = obase + eoff[r] + nt * 16;
                    if (POOL) {
                        const float hi = fmaxf(fmaxf(y00[r], y01[r]), fmaxf(y10[r], y11[r]));
                        const float lo = fminf(fminf(y00[r], y01[r]), fminf(y10[r], y11[r]));
                        const float x = bscale[nt] >= 0.0f ? hi : lo;  // max commutes with the monotone BN + ELU
                        o[0] = elu_fastw((x - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    } else {
                        const int rstride = a.W * COUT;
                        const float v00 = RAW ? y00[r] : elu_fastw((y00[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        const float v01 = RAW ? y01[r] : elu_fastw((y01[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        const float v10 = RAW ? y10[r] : elu_fastw((y10[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        const float v11 = RAW ? y11[r] : elu_fastw((y11[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        o[0] = v00; o[COUT] = v01; o[rstride] = v10; o[rstride + COUT] = v11;
                        if (RAW && a.stats) {
                            if (bnb) {
                                const size_t e = (size_t)(o - a.out);
                                const float *zp = a.bf.z + e;
                                float t1 = 0.f, t2 = 0.f;
                                bnb_acc(v00, zp[0], bnb_mult(a.bf.tie, e), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                bnb_acc(v01, zp[COUT], bnb_mult(a.bf.tie, e + COUT), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                bnb_acc(v10, zp[rstride], bnb_mult(a.bf.tie, e + rstride), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                bnb_acc(v11, zp[rstride + COUT], bnb_mult(a.bf.tie, e + rstride + COUT), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                st1[nt] += (double)t1; st2[nt] += (double)t2;
                            } else {
                            st1[nt] += (double)((v00 + v01) + (v10 + v11));
                            st2[nt] += (double)((v00 * v00 + v01 * v01) + (v10 * v10 + v11 * v11));
                            }
                        }
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ey[r] >= ly || ex[r] >= lx) continue;
                float *o = obase + eoff[r] + nt * 16;
                if (ASR_WINOG_ABL & 64) {          // keep the arithmetic, drop the stores
                    asm volatile("" ::"v"(y00[r]), "v"(y01[r]), "v"(y10[r]), "v"(y11[r]));
                    continue;
                }
                if (POOL) {
                    const float hi = fmaxf(fmaxf(y00[r], y01[r]), fmaxf(y10[r], y11[r]));
                    const float lo = fminf(fminf(y00[r], y01[r]), fminf(y10[r], y11[r]));
                    const float x = bscale[nt] >= 0.0f ? hi : lo;      // max commutes with the monotone BN + ELU
                    o[0] = elu_fastw((x - bmean[nt]) * bscale[nt] + bbeta[nt]);
                } else {
                    const bool y1 = 2 * ey[r] + 1 < a.H - py0, x1 = 2 * ex[r] + 1 < a.W - px0;
                    const int rstride = a.W * COUT;
                    const float v00 = RAW ? y00[r] : elu_fastw((y00[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    const float v01 = RAW ? y01[r] : elu_fastw((y01[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    const float v10 = RAW ? y10[r] : elu_fastw((y10[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    const float v11 = RAW ? y11[r] : elu_fastw((y11[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    o[0] = v00;
                    if (x1) o[COUT] = v01;
                    if (y1) o[rstride] = v10;
                    if (y1 && x1) o[rstride + COUT] = v11;
                    if (RAW && a.stats) {
                        if (bnb) {
                            const size_t e = (size_t)(o - a.out);
                            const float *zp = a.bf.z + e;
                            float t1 = 0.f, t2 = 0.f;
                            bnb_acc(v00, zp[0], bnb_mult(a.bf.tie, e), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                            if (x1) bnb_acc(v01, zp[COUT], bnb_mult(a.bf.tie, e + COUT), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                            if (y1) bnb_acc(v10, zp[rstride], bnb_mult(a.bf.tie, e + rstride), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                            if (y1 && x1) bnb_acc(v11, zp[rstride + COUT], bnb_mult(a.bf.tie, e + rstride + COUT), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                            st1[nt] += (double)t1; st2[nt] += (double)t2;
                        } else {
                        const float w01 = x1 ? v01 : 0.f, w10 = y1 ? v10 : 0.f, w11 = (y1 && x1) ? v11 : 0.f;
                        st1[nt] += (double)((v00 + w01) + (w10 + w11));
                        st2[nt] += (double)((v00 * v00 + w01 * w01) + (w10 * w10 + w11 * w11));
                        }
                    }
                }
            }
        }
    }
#if ASR_WINO_BUFDMA
    if constexpr (PW == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the copies above are invisible to the compiler's counters
#endif
    __syncthreads();      // the next region's patch is in place (LDS-DMA drained / producers done), this buffer is free
    }
    if (RAW && a.stats)
        wino_stats_store<NTW>(a.stats, (int)blockIdx.x * WAVES + wave, COUT, ng * WROW, lane, st1, st2);
}

// ---------------------------------------------------------------------------------------------------------------
// conv3x3_winog: the same algorithm for C_in >= 24, with the A operand read straight from global memory.
// For wide layers the LDS cannot hold both the transformed weights (16*C_in*C_out floats: 144 KiB at 48 -> 48) and
// input patches, and splitting the output channels over workgroups repeats the input transform per group.  Here a
// persistent workgroup keeps ALL transformed weights in LDS, every wave owns whole M-tiles (16 winograd tiles x all
// n-tiles: the input transform is paid once per C_out), and each lane fetches its tile's 4x4 patch with 8-byte
// loads (two channels = two k-steps per load; the 4 lane groups of a tile read 32 contiguous bytes, neighbouring
// tiles and the waves of a workgroup reuse each other's lines in L1/L2).  No input staging, no barriers after the
// weights are in place; the next channel block's loads are issued before the current block's MFMAs.
// An M-tile is 16 CONSECUTIVE tiles of the batch's tile list (listed strip by strip of two tile rows, column-major
// inside a strip, so that 16 consecutive tiles form an 8 x 2 block) - it may wrap around a strip end or run into the
// next image, every lane addresses its own tile - so no slot is wasted on maps whose size is not a multiple of an
// M-tile shape (20x25 output pixels = 130 tiles: 8.1 M-tiles instead of the 10-12 of 2-D arrangements).
struct WinoGArgs {
    const float *in, *wpk, *bnp;
    float *out;
    int N, H, W, OH, OW;
    int ty_img, tx_img;    // winograd tiles per image (rows, columns)
    int coutp;
    int tiles;             // winograd tiles in the launch = N * ty_img * tx_img
    int total;             // M-tiles in the launch = ceil(tiles / 16)
    int strips;            // tile list order: strips of this many tile rows (1, 2, 4 or 8), column-major inside
    int strip_shift;       // log2(strips)
    double *stats;         // RAW only, may be null: per-wave [sum | sum of squares] of the outputs, [row][2][C_out]
    BnBwdFuse bf;          // RAW + stats, bf.z != null: the sums are those of a BatchNorm backward instead (asr_kernels.h)
};

template <int CIN, int COUT, bool POOL, int NT, int WAVES, int MINW, bool RAW>
__global__ __launch_bounds__(64 * WAVES, MINW) void conv3x3_winog(WinoGArgs a) {
    constexpr int KS = CIN / 4, NB = CIN / 8, WROW = NT * 16, T = 64 * WAVES;
    constexpr int WS = NT == 2 ? 48 : WROW;      // LDS row stride: 32 would put lane groups g and g+1 on the same banks
    constexpr int WD = 4 * NT;                   // B operands in flight ahead of the MFMA that uses them (4 positions)
    constexpr int WQ = 32 * NT;                  // B operands per channel block (2 k-steps x 16 positions x NT)
    constexpr bool REM = (CIN % 8) != 0;         // one more k-step on the last 4 channels (C_in = 12)
    // software pipeline across M-tiles (next M-tile's first loads issued before this one's epilogue, stores deferred
    // by one M-tile): needs ~50 more live registers - the one-wave-per-SIMD builds have them, the two-wave builds
    // (256 registers) would spill and rely on the second wave to cover memory latency instead
    constexpr bool PIPE = (MINW == 1);
    static_assert(CIN % 4 == 0, "k-steps of 4 channels");
    extern __shared__ __align__(16) float w_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // NT n-tiles of this workgroup's group (blockIdx.y; one group unless the weights of all n-tiles exceed the LDS)
    const int ng = blockIdx.y;
    for (int i = tid; i < 16 * CIN * WROW / 4; i += T) {
        const int row = i / (WROW / 4), q = i - row * (WROW / 4);
        const int col = ng * WROW + q * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < a.coutp) v = *reinterpret_cast<const float4 *>(a.wpk + (size_t)row * a.coutp + col);
        reinterpret_cast<float4 *>(w_lds + row * WS)[q] = v;
    }
    __syncthreads();

    const int m = lane & 15, g = lane >> 4, n = lane & 15;
    const float *w_lane = w_lds + g * WS + n;
    // B operand q of a channel block: k-step q / (16 NT), position (q / NT) % 16, n-tile q % NT
    auto wq_off = [](int q) { return ((q / (16 * NT)) * 16 + (q / NT) % 16) * 4 * WS + (q % NT) * 16; };
    float bmean[NT], bscale[NT], bbeta[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = ng * WROW + nt * 16 + n;
        const bool ok = !RAW && ch < COUT;
        bmean[nt] = ok ? a.bnp[ch] : 0.f;
        bscale[nt] = ok ? a.bnp[a.coutp + ch] : 1.f;
        bbeta[nt] = ok ? a.bnp[2 * a.coutp + ch] : 0.f;
    }
    float bistd[NT];                         // BatchNorm-backward sums (a.bf): mu, gamma*inv_std, beta, inv_std per channel
    const bool bnb = ASR_BNB_FUSE_BUILD && RAW && a.stats && a.bf.z != nullptr;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = ng * WROW + nt * 16 + n;
        bistd[nt] = 0.f;
        if (bnb && ch < COUT) {
            bmean[nt] = a.bf.cst[ch]; bistd[nt] = a.bf.cst[COUT + ch];
            bscale[nt] = a.bf.cst[2 * COUT + ch]; bbeta[nt] = a.bf.cst[3 * COUT + ch];
        }
    }
    const int per_img = a.ty_img * a.tx_img;

    // Which M-tiles this wave owns.  Blocks b and b + 8 share an XCD (observed placement; correctness does not depend
    // on it): every XCD walks ONE contiguous eighth of the tile list, consecutive M-tiles going to the waves of one
    // workgroup - the rows a tile shares with its vertical neighbours (two of its four patch rows) are then re-read
    // from that XCD's L2 / the CU's L1 instead of from HBM by a workgroup on another XCD (1.7x the input otherwise).
    int mt, mt_end, mt_stride;
    if ((gridDim.x & 7) == 0) {
        const int per_x = (a.total + 7) >> 3;
        const int r0 = (int)(blockIdx.x & 7) * per_x;
        mt_end = min(r0 + per_x, a.total);
        mt_stride = (int)(gridDim.x >> 3) * WAVES;
        mt = r0 + (int)(blockIdx.x >> 3) * WAVES + wave;
    } else {
        mt_end = a.total;
        mt_stride = (int)gridDim.x * WAVES;
        mt = (int)blockIdx.x * WAVES + wave;
    }
    if (mt >= mt_end) {                   // no barrier below this point
        // a wave without M-tiles still owns a row of the statistics table: it writes zeros, so that EVERY row of the
        // table is written by every launch and no memset has to precede it (round 6: 14 fills per training step, each a
        // launch + a gap on the forward chain of its tower)
        if (RAW && a.stats && (lane >> 4) == 0) {
            const int row = (int)blockIdx.x * WAVES + wave;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int ch = ng * WROW + nt * 16 + (lane & 15);
                if (ch < COUT) {
                    a.stats[((size_t)row * 2) * COUT + ch] = 0.0;
                    a.stats[((size_t)row * 2 + 1) * COUT + ch] = 0.0;
                }
            }
        }
        return;
    }

    // per-tile addressing of an M-tile: the lane's 4x4 patch as clamped element offsets (always loadable), which of
    // its pixels lie inside the image, and where the tile's output goes
    int off[4][4];
    unsigned okm, my_off;
    int my_flags;
    const float *ibase;
    auto setup = [&](int mtile) {
        // this lane's tile: number 16*mtile + m of the batch's tile list (lanes past the end compute on clamped
        // addresses and store nothing)
        const int tnum = mtile * 16 + m;
        const bool tvalid = tnum < a.tiles;
        const int tcl = min(tnum, a.tiles - 1);
        const int img = tcl / per_img;
        const int trest = tcl - img * per_img;
        // a.strips = S: tiles are listed strip by strip (S tile rows), column-major inside a strip, so 16 consecutive
        // tiles form a (16/S) x S block: S = 2 shares 6 x 18 patch pixels, S = 1 (plain row-major) 4 x 34 - and a strip
        // re-reads only 2 of its 2S + 2 input rows from the strip above (S = 1: two of four, from another CU when a
        // tile row is longer than a workgroup's M-tiles; conv4: 1.37 GB read per 1000 sheets at S = 1, 1.01 GB at
        // S = 2).  A last strip with fewer rows is listed the same way with its own height.  Which S is fastest
        // depends on the layer and the build (measured +-4 %): the tuner times S = 1 and 2 (4 and 8 never won).
        int tty, ttx;
        {
            const int S = a.strips, strip_tiles = S * a.tx_img;
            const int sidx = trest / strip_tiles, q = trest - sidx * strip_tiles;
            const int rows_here = min(S, a.ty_img - sidx * S);            // height of this (possibly last) strip
            if (rows_here == S) {
                // (q mod S without the mask S - 1: held in a VGPR across the M-tile loop the mask was spilled to scratch
                // and reloaded - with a vmcnt(0) wait - once per M-tile; the shift count sits in an SGPR)
                ttx = q >> a.strip_shift;
                tty = sidx * S + (q - (ttx << a.strip_shift));
            } else {
                ttx = q / rows_here;
                tty = sidx * S + (q - ttx * rows_here);
            }
        }
        const int py = 2 * tty, px = 2 * ttx;                          // top-left output pixel of the tile
        ibase = a.in + (int64_t)img * a.H * a.W * CIN;           // (+ 2g: in the element offsets below)
        int yo[4], xo[4];
        unsigned oy_m = 0, ox_m = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = py - 1 + i, x = px - 1 + i;
            oy_m |= (unsigned)(y >= 0 && y < a.H) << i;
            ox_m |= (unsigned)(x >= 0 && x < a.W) << i;
            yo[i] = min(max(y, 0), a.H - 1) * a.W;
            xo[i] = min(max(x, 0), a.W - 1);
        }
        okm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                off[i][j] = (yo[i] + xo[j]) * CIN + 2 * g;
                okm |= (((oy_m >> i) & (ox_m >> j)) & 1u) << (i * 4 + j);
            }
        if (!tvalid) okm = 0;
        // (unsigned 32-bit element offsets: output buffers of up to 16 GiB)
        if (POOL) {
            my_off = (((unsigned)img * a.OH + tty) * a.OW + ttx) * COUT;
            my_flags = (tvalid && tty < a.OH && ttx < a.OW) ? 1 : 0;
        } else {
            my_off = (((unsigned)img * a.H + py) * a.W + px) * COUT;
            my_flags = tvalid ? (1 | ((px + 1 < a.W) ? 2 : 0) | ((py + 1 < a.H) ? 4 : 0)) : 0;
        }
    };
    float2w nxt[4][4];
    float drem[4][4];                        // remainder k-step: channel 8*NB + g of every patch pixel
    auto load_first = [&]() {                // channel block 0 (and the remainder channels) of the tile `setup` described
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                nxt[i][j] = (ASR_WINOG_ABL & 256) ? float2w{(float)i, (float)j}
                                                  : *reinterpret_cast<const float2w *>(ibase + off[i][j]);
    };
    auto load_rem = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) drem[i][j] = ibase[off[i][j] + 8 * NB - g];       // off carries + 2g
    };
    setup(mt);
    load_first();
    if (REM) load_rem();
    // B operands travel WD MFMAs ahead of their use (one wave per SIMD: nothing else hides the LDS latency, and
    // the 4-bit lgkmcnt cannot express "the older half of 96 reads")
    float wpre[WD];
#pragma unroll
    for (int q = 0; q < WD; ++q) wpre[q] = w_lane[wq_off(q)];

    // Stores are DEFERRED by one M-tile.  vmcnt counts loads and stores together, in issue order, and the number of
    // store instructions an epilogue issues is not static (edge tiles), so the compiler can only wait for "everything"
    // at the next use of loaded data: with the stores issued right behind the epilogue every M-tile paid a full store
    // round trip before its first MFMA (measured by ablation: 0.14 of conv4's 0.77 ms).  The finished values of
    // M-tile k are kept in registers and written at the top of M-tile k+1, right AFTER the wait for k+1's first
    // channel block (whose loads were issued during k's last block): they then complete under a block of MFMAs.
    constexpr int PV = POOL ? 1 : 4;
    float pend[NT][4][PV];
    unsigned pend_off[4];
    int pend_flags[4];
    bool have_pend = false;                                            // wave-uniform
    auto flush = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (ng * WROW + nt * 16 + n >= COUT) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!(pend_flags[r] & 1)) continue;
                float *o = a.out + (size_t)pend_off[r] + ng * WROW + nt * 16 + n;
                if (ASR_WINOG_ABL & 128) {         // keep the arithmetic, drop the stores
                    asm volatile("" ::"v"(pend[nt][r][0]));
                    continue;
                }
                if constexpr (POOL) {
                    o[0] = pend[nt][r][0];
                } else {
                    const bool x1 = (pend_flags[r] & 2) != 0, y1 = (pend_flags[r] & 4) != 0;
                    const int rstride = a.W * COUT;
                    o[0] = pend[nt][r][0];
                    if (x1) o[COUT] = pend[nt][r][PV - 3];
                    if (y1) o[rstride] = pend[nt][r][PV - 2];
                    if (y1 && x1) o[rstride + COUT] = pend[nt][r][PV - 1];
                }
            }
        }
    };

    double gst1[NT], gst2[NT];               // RAW + a.stats: running sums of this lane's channel(s)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { gst1[nt] = 0.0; gst2[nt] = 0.0; }
    for (; mt < mt_end; mt += mt_stride) {
        const bool more = PIPE && mt + mt_stride < mt_end;             // wave-uniform
        const unsigned okm_cur = okm, off_cur = my_off;
        const int flags_cur = my_flags;
        // every patch of the M-tile inside its image: no border selects (wave-uniform)
        const bool interior = __builtin_amdgcn_ballot_w64(okm_cur != 0xffffu) == 0;
        floatx4w acc[16][NT];
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[p][nt] = floatx4w{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int t = 0; t < NB; ++t) {
            float2w dp[4][4];
            if (interior) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dp[i][j] = nxt[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        dp[i][j] = ((okm_cur >> (i * 4 + j)) & 1u) ? nxt[i][j] : float2w{0.f, 0.f};
            }
            if (t == 0 && have_pend) {
                // dp holds block 0: its loads have landed, nothing else is outstanding - the previous M-tile's
                // stores go out now and retire under this block's MFMAs
                asm volatile("" ::"v"(dp[3][3]));
                flush();
            }
            if (t + 1 < NB) {
                if (!(ASR_WINOG_ABL & (2 | 256))) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            nxt[i][j] = *reinterpret_cast<const float2w *>(ibase + off[i][j] + 8 * (t + 1));
                }
            } else if (more && !REM) {
                setup(mt + mt_stride);
                load_first();
            }
            if (!(ASR_WINOG_ABL & 8)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {              // columns: B^T d
                const float2w d0 = dp[0][j], d1 = dp[1][j], d2 = dp[2][j], d3 = dp[3][j];
                dp[0][j] = psub(d0, d2); dp[1][j] = padd(d1, d2); dp[2][j] = psub(d2, d1); dp[3][j] = psub(d1, d3);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // rows: (B^T d) B
                const float2w t0 = dp[i][0], t1 = dp[i][1], t2 = dp[i][2], t3 = dp[i][3];
                dp[i][0] = psub(t0, t2); dp[i][1] = padd(t1, t2); dp[i][2] = psub(t2, t1); dp[i][3] = psub(t1, t3);
            }
            }
            const float *wk = w_lane + (2 * t) * (16 * 4 * WS);
            // next block; after the last one the remainder k-step's rows (or any valid rows)
            const float *wn = (t + 1 < NB || REM) ? wk + 2 * 16 * 4 * WS : w_lane;
            float wv[WQ + WD];
#pragma unroll
            for (int q = 0; q < WD; ++q) wv[q] = wpre[q];
#pragma unroll
            for (int q0 = 0; q0 < WQ; q0 += NT) {
                // the operands WD steps ahead - inside this block, or the first ones of the next block - are issued
                // before this position's MFMAs; the scheduling barriers keep the compiler from sinking them again
#pragma unroll
                for (int q = q0; q < q0 + NT; ++q)
                    wv[q + WD] = (ASR_WINOG_ABL & 4) ? wv[q] : (q + WD < WQ) ? wk[wq_off(q + WD)] : wn[wq_off(q + WD - WQ)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = q0; q < q0 + NT; ++q) {
                    const int p = (q / NT) % 16, c = q / (16 * NT), nt = q % NT;
                    acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[p >> 2][p & 3][c], wv[q], acc[p][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < WD; ++q) wpre[q] = wv[WQ + q];
        }
        if (REM) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (!((okm_cur >> (i * 4 + j)) & 1u)) drem[i][j] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d0 = drem[0][j], d1 = drem[1][j], d2 = drem[2][j], d3 = drem[3][j];
                drem[0][j] = d0 - d2; drem[1][j] = d1 + d2; drem[2][j] = d2 - d1; drem[3][j] = d1 - d3;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t0 = drem[i][0], t1 = drem[i][1], t2 = drem[i][2], t3 = drem[i][3];
                drem[i][0] = t0 - t2; drem[i][1] = t1 + t2; drem[i][2] = t2 - t1; drem[i][3] = t1 - t3;
            }
            float dr[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) dr[i][j] = drem[i][j];
            if (more) {                       // the transformed remainder lives in dr: drem, nxt are free for the next M-tile
                setup(mt + mt_stride);
                load_first();
                load_rem();
            }
            const float *wk = w_lane + (2 * NB) * (16 * 4 * WS);
            constexpr int WQR = 16 * NT;
            float wv[WQR + WD];
#pragma unroll
            for (int q = 0; q < WD; ++q) wv[q] = wpre[q];
#pragma unroll
            for (int q0 = 0; q0 < WQR; q0 += NT) {
#pragma unroll
                for (int q = q0; q < q0 + NT; ++q) wv[q + WD] = (q + WD < WQR) ? wk[wq_off(q + WD)] : w_lane[wq_off(q + WD - WQR)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = q0; q < q0 + NT; ++q) {
                    const int p = (q / NT) % 16, nt = q % NT;
                    acc[p][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dr[p >> 2][p & 3], wv[q], acc[p][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < WD; ++q) wpre[q] = wv[WQR + q];
        }

        // accumulator element r of this lane belongs to tile 4g + r of the M-tile, whose owner is lane 4g + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pend_off[r] = (unsigned)__shfl((int)off_cur, 4 * g + r);
            pend_flags[r] = __shfl(flags_cur, 4 * g + r);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            floatx4w s0[4], s1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                s0[nu] = padd(padd(acc[nu][nt], acc[4 + nu][nt]), acc[8 + nu][nt]);
                s1[nu] = psub(psub(acc[4 + nu][nt], acc[8 + nu][nt]), acc[12 + nu][nt]);
            }
            const floatx4w y00 = padd(padd(s0[0], s0[1]), s0[2]), y01 = psub(psub(s0[1], s0[2]), s0[3]);
            const floatx4w y10 = padd(padd(s1[0], s1[1]), s1[2]), y11 = psub(psub(s1[1], s1[2]), s1[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (POOL) {
                    const float hi = fmaxf(fmaxf(y00[r], y01[r]), fmaxf(y10[r], y11[r]));
                    const float lo = fminf(fminf(y00[r], y01[r]), fminf(y10[r], y11[r]));
                    const float x = bscale[nt] >= 0.0f ? hi : lo;
                    pend[nt][r][0] = elu_fastw((x - bmean[nt]) * bscale[nt] + bbeta[nt]);
                } else {
                    pend[nt][r][0] = RAW ? y00[r] : elu_fastw((y00[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    pend[nt][r][PV - 3] = RAW ? y01[r] : elu_fastw((y01[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    pend[nt][r][PV - 2] = RAW ? y10[r] : elu_fastw((y10[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    pend[nt][r][PV - 1] = RAW ? y11[r] : elu_fastw((y11[r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    if (RAW && a.stats) {
                        const int f = pend_flags[r];
                        if (bnb) {
                            if ((f & 1) && ng * WROW + nt * 16 + n < COUT) {
                                const size_t e = (size_t)pend_off[r] + ng * WROW + nt * 16 + n;
                                const float *zp = a.bf.z + e;
                                const int rstride = a.W * COUT;
                                float t1 = 0.f, t2 = 0.f;
                                bnb_acc(y00[r], zp[0], bnb_mult(a.bf.tie, e), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                if ((f & 3) == 3) bnb_acc(y01[r], zp[COUT], bnb_mult(a.bf.tie, e + COUT), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                if ((f & 5) == 5) bnb_acc(y10[r], zp[rstride], bnb_mult(a.bf.tie, e + rstride), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                if ((f & 7) == 7) bnb_acc(y11[r], zp[rstride + COUT], bnb_mult(a.bf.tie, e + rstride + COUT), bmean[nt], bscale[nt], bbeta[nt], bistd[nt], t1, t2);
                                gst1[nt] += (double)t1; gst2[nt] += (double)t2;
                            }
                        } else {
                        const float w00 = (f & 1) ? y00[r] : 0.f, w01 = ((f & 3) == 3) ? y01[r] : 0.f;
                        const float w10 = ((f & 5) == 5) ? y10[r] : 0.f, w11 = ((f & 7) == 7) ? y11[r] : 0.f;
                        gst1[nt] += (double)((w00 + w01) + (w10 + w11));
                        gst2[nt] += (double)((w00 * w00 + w01 * w01) + (w10 * w10 + w11 * w11));
                        }
                    }
                }
            }
        }
        if (PIPE) {
            have_pend = true;
        } else {
            flush();
            if (mt + mt_stride < mt_end) {
                setup(mt + mt_stride);
                load_first();
                if (REM) load_rem();
            }
        }
    }
    if (PIPE) flush();         // the wave's last M-tile (every wave that gets here has computed at least one)
    if (RAW && a.stats)
        wino_stats_store<NT>(a.stats, (int)blockIdx.x * WAVES + wave, COUT, ng * WROW, lane, gst1, gst2);
}

// ---- weight transform (repack_elems.inl: wino_pack_elem) -------------------------------------------------------
__global__ void wino_pack_kernel(const float *W, int cin, int cout, int dgrad, float *wpk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= wino_pack_count(cin, cout, dgrad)) return;
    wino_pack_elem(idx, W, cin, cout, dgrad, wpk);
}

size_t wino_wpack_floats(int cin, int cout) { return (size_t)16 * cin * ((cout + 15) / 16 * 16); }

hipError_t launch_wino_pack(hipStream_t s, const float *W, int cin, int cout, float *wpk, int dgrad) {
    const int kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
    const int total = kdim * ((ndim + 15) / 16 * 16);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, W, cin, cout, dgrad, wpk);
    return hipGetLastError();
}

// ---- instantiation table ---------------------------------------------------------------------------------------
struct WinoVariant {
    int cin, cout, pool, ntw, kb, waves, minw, rmax, raw;
    void (*kernel)(WinoArgs);
    const char *symbol;
    int pw, in_mode;          // producer waves (block 1 evaluated inside, see conv3x3_wino) and their raw input mode
};
#define ASR_BOOLSTRW_0 "false"
#define ASR_BOOLSTRW_1 "true"
#define ASR_WINO(CIN, COUT, POOL, NTW, KB, WAVES, MINW, RMAX)                                                     \
    { CIN, COUT, POOL, NTW, KB, WAVES, MINW, RMAX, 0,                                                             \
      conv3x3_wino<CIN, COUT, (POOL != 0), NTW, KB, WAVES, MINW, RMAX, false>,                                    \
      "void asr::conv3x3_wino<" #CIN ", " #COUT ", " ASR_BOOLSTRW_##POOL ", " #NTW ", " #KB ", " #WAVES ", " #MINW \
      ", " #RMAX ", false, 0, 0>(asr::WinoArgs)" }
#define ASR_WINOR(CIN, COUT, NTW, KB, WAVES, MINW, RMAX)                                                          \
    { CIN, COUT, 0, NTW, KB, WAVES, MINW, RMAX, 1,                                                                \
      conv3x3_wino<CIN, COUT, false, NTW, KB, WAVES, MINW, RMAX, true>,                                           \
      "void asr::conv3x3_wino<" #CIN ", " #COUT ", false, " #NTW ", " #KB ", " #WAVES ", " #MINW                  \
      ", " #RMAX ", true, 0, 0>(asr::WinoArgs)" }
// producer-wave builds: three consecutive entries per shape, one per raw input mode (ASR_IN_F32_PREPARED / _F32_RAW /
// _U8_RAW); the planner enumerates the first, the launcher adds the call's in_mode
#define ASR_WINOF1(CIN, COUT, WAVES, MINW, PW, MODE)                                                              \
    { CIN, COUT, 1, 1, 1, WAVES, MINW, 1, 0,                                                                      \
      conv3x3_wino<CIN, COUT, true, 1, 1, WAVES, MINW, 1, false, PW, MODE>,                                       \
      "void asr::conv3x3_wino<" #CIN ", " #COUT ", true, 1, 1, " #WAVES ", " #MINW ", 1, false, " #PW ", " #MODE  \
      ">(asr::WinoArgs)", PW, MODE }
#define ASR_WINOF(CIN, COUT, WAVES, MINW, PW)                                                                     \
    ASR_WINOF1(CIN, COUT, WAVES, MINW, PW, 0), ASR_WINOF1(CIN, COUT, WAVES, MINW, PW, 1),                         \
        ASR_WINOF1(CIN, COUT, WAVES, MINW, PW, 2)
static const WinoVariant g_wino[] = {
    ASR_WINO(12, 12, 1, 1, 1, 4, 3, 8),
    ASR_WINO(12, 12, 1, 1, 1, 8, 3, 6),
    ASR_WINO(12, 24, 0, 2, 1, 4, 2, 8),
    ASR_WINO(12, 24, 0, 2, 1, 8, 2, 6),
    ASR_WINO(12, 24, 0, 1, 1, 4, 3, 8),
    ASR_WINO(24, 24, 1, 2, 1, 4, 2, 8),
    ASR_WINO(24, 24, 1, 2, 1, 8, 2, 6),
    ASR_WINO(24, 24, 1, 1, 3, 4, 2, 10),
    ASR_WINO(24, 24, 1, 1, 1, 8, 2, 6),
    ASR_WINO(24, 48, 0, 1, 3, 4, 2, 10),
    ASR_WINO(24, 48, 0, 1, 1, 8, 2, 8),
    ASR_WINO(24, 48, 0, 2, 1, 8, 2, 6),
    ASR_WINO(48, 48, 1, 1, 2, 4, 2, 12),
    ASR_WINO(48, 48, 1, 1, 2, 8, 2, 10),
    ASR_WINO(48, 48, 1, 1, 1, 8, 2, 8),
    ASR_WINO(48, 48, 0, 1, 2, 4, 2, 12),
    ASR_WINO(48, 48, 0, 1, 2, 8, 2, 10),
    ASR_WINO(48, 48, 0, 1, 1, 8, 2, 8),
    // RAW epilogue (no BN / ELU / pool): train-mode forward convolutions and data gradients with C_in = 12
    ASR_WINOR(12, 12, 1, 1, 8, 3, 6),
    ASR_WINOR(12, 24, 2, 1, 8, 2, 6),
    // (four-wave workgroups, the deterministic path's winners: candidates of the training step's tuner)
    ASR_WINOR(12, 12, 1, 1, 4, 3, 8),
    ASR_WINOR(12, 24, 2, 1, 4, 2, 8),
    ASR_WINOR(12, 24, 1, 1, 4, 3, 8),
    // block 1 evaluated by producer waves (C_in of block 2 = 12: the `cont` model, both towers)
    ASR_WINOF(12, 12, 4, 4, 4), ASR_WINOF(12, 12, 4, 3, 2), ASR_WINOF(12, 12, 8, 3, 4), ASR_WINOF(12, 12, 8, 4, 8),
    ASR_WINOF(12, 12, 4, 3, 4),
};
static const int g_num_wino = (int)(sizeof(g_wino) / sizeof(g_wino[0]));

// Patch layout in LDS for an M-tile arrangement: pixel stride C4P chunks (the smallest odd count that holds the
// channels) and the row pitch RP >= LW*C4P + 1 (one spare chunk for the shifted rows) for which the lanes of a
// ds_read_b64 half-wave (32 lanes: two lane groups x 16 tiles, 8 bytes each) fall on the fewest common banks.
// Returns the worst number of LDS cycles of one such read (1 = conflict-free).
static int wino_lds_layout(int cin, int LW, int MY, int MX, int *c4p_out, int *rp_out) {
    const int c4 = cin / 4;
    const int c4p = (c4 & 1) ? c4 : c4 + 1;
    int best_rp = 0, best = 1 << 30;
    for (int extra = 1; extra <= 8; ++extra) {
        const int rp = LW * c4p + extra;
        int worst = 0;
        for (int half = 0; half < 2; ++half)
            for (int i = 0; i < 4; ++i) {
                // per bank: the distinct dword addresses its lanes ask for; the read takes max-over-banks cycles
                int seen[64][32], nseen[64] = {0};
                int cycles = 1;
                for (int l = 0; l < 32; ++l) {
                    const int lane = half * 32 + l;
                    const int m = lane & 15, g = lane >> 4;
                    const int tyy = m / MX, txx = m % MX;
                    const int fl = ((2 * tyy + i) * rp + 2 * txx * c4p + ((tyy + (i >> 1)) & 1)) * 4 + (g >> 1) * 4 + (g & 1) * 2;
                    for (int w = 0; w < 2; ++w) {
                        const int bank = (fl + w) & 63;
                        bool dup = false;
                        for (int q = 0; q < nseen[bank]; ++q) dup = dup || seen[bank][q] == fl + w;
                        if (!dup) seen[bank][nseen[bank]++] = fl + w;
                        cycles = std::max(cycles, nseen[bank]);
                    }
                }
                worst = std::max(worst, cycles);
            }
        if (worst < best) { best = worst; best_rp = rp; }
    }
    *c4p_out = c4p;
    *rp_out = best_rp;
    return best;
}

static void enumerate_wino(int vi, int H, int W, int lds_budget, std::vector<ConvPlan> &out) {
    const WinoVariant &v = g_wino[vi];
    const int nt = (v.cout + 15) / 16;
    const int ngroups = (nt + v.ntw - 1) / v.ntw;
    const int w_bytes = 16 * v.cin * (v.ntw == 2 ? 48 : v.ntw * 16) * 4;
    static const int shapes[4][2] = {{2, 8}, {4, 4}, {8, 2}, {16, 1}};      // MY even: row origins are multiples of 4
    const int ty_img = (H + 1) / 2, tx_img = (W + 1) / 2;       // winograd tiles covering the image
    for (const auto &sh : shapes) {
        const int MY = sh[0], MX = sh[1];
        const int max_my = (ty_img + MY - 1) / MY, max_mx = (tx_img + MX - 1) / MX;
        for (int nmy = 1; nmy <= max_my; ++nmy) {
            for (int nmx = 1; nmx <= max_mx; ++nmx) {
                const int RY = nmy * MY * 2, RX = nmx * MX * 2;
                int c4p = 0, rp = 0;
                const int rd_cycles = wino_lds_layout(v.cin, RX + 2, MY, MX, &c4p, &rp);
                const int patch_f4 = (RY + 2) * rp;
                const int lds = 2 * ((patch_f4 * 16 + 1023) & ~1023) + w_bytes + (v.pw ? 1024 : 0);
                if (lds > lds_budget) continue;
                if (!v.pw && patch_f4 > v.rmax * 64 * v.waves) continue;
                const int n_mt = nmy * nmx;
                if (n_mt > 16 * v.waves) continue;
                const int tiles_y = (H + RY - 1) / RY, tiles_x = (W + RX - 1) / RX;
                const double per_wave = (double)((n_mt + v.waves - 1) / v.waves);
                const double mfma = 16.0 * (v.cin / 4) * v.ntw * 32.0;
                const double valu = (16.0 * (v.cin / 4) + 110.0 * v.ntw) * 4.0 + (rd_cycles - 1) * 8.0 * v.cin;
                // staging per region: address + issue + LDS write of the DMA form; the producers' block-1 arithmetic
                // (about 330 vector instructions per 64 pixels) runs beside the consumers and only counts when it is
                // the longer of the two
                double stage = (double)v.rmax * 16.0 * 4.0;
                if (v.pw) {
                    const double prod = std::ceil((RY + 2.0) * (RX + 2.0) / (64.0 * v.pw)) * 330.0 * 4.0;
                    const double cons = (16.0 * (v.cin / 4) * v.ntw * 32.0 + (16.0 * (v.cin / 4) + 110.0 * v.ntw) * 4.0) *
                                        (double)((n_mt + v.waves - 1) / v.waves);
                    stage = prod > cons ? prod - cons : 0.0;
                }
                ConvPlan bp{};
                // regions of one image x (work of the slowest wave + staging); partially filled edge regions cost
                // the same as full ones, so the model charges tiles_y*tiles_x full regions
                bp.cost = ((mfma + valu) * per_wave + stage + 800.0) * tiles_y * tiles_x * ngroups;
                bp.TH = RY; bp.TW = RX; bp.NI = MY;
                bp.tiles_y = tiles_y; bp.tiles_x = tiles_x;
                bp.lds_bytes = lds;
                bp.tile_floats = nmy * 1000 + nmx;              // (nmy, nmx) for the launcher
                bp.c4p = c4p; bp.rp = rp;
                bp.cin = v.cin; bp.cout = v.cout; bp.pool = v.pool;
                bp.H = H; bp.W = W;
                bp.OH = v.pool ? H / 2 : H;
                bp.OW = v.pool ? W / 2 : W;
                bp.threads = 64 * (v.waves + v.pw);
                bp.variant = 3000 + vi;
                bp.symbol = v.symbol;
                bp.fuse1 = v.pw ? 1 : 0;
                out.push_back(bp);
            }
        }
    }
    std::sort(out.begin(), out.end(), [](const ConvPlan &x, const ConvPlan &y) { return x.cost < y.cost; });
}

static void finish_wino(ConvPlan &bp) {
    const WinoVariant &v = g_wino[bp.variant - 3000];
    for (int mode = 0; mode < (v.pw ? 3 : 1); ++mode)        // producer builds: one instantiation per raw input mode
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(g_wino[bp.variant - 3000 + mode].kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), bp.threads,
                                                     (size_t)bp.lds_bytes) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = std::max(1, std::min(4, (160 * 1024) / std::max(1, bp.lds_bytes)));
    }
    bp.blocks_per_cu = std::min(nb, 8);
}

struct WinoGVariant {
    int cin, cout, pool, nt, waves, minw, raw;
    void (*kernel)(WinoGArgs);
    const char *symbol;
};
#define ASR_WINOG(CIN, COUT, POOL, NT, WAVES, MINW)                                                               \
    { CIN, COUT, POOL, NT, WAVES, MINW, 0, conv3x3_winog<CIN, COUT, (POOL != 0), NT, WAVES, MINW, false>,         \
      "void asr::conv3x3_winog<" #CIN ", " #COUT ", " ASR_BOOLSTRW_##POOL ", " #NT ", " #WAVES ", " #MINW         \
      ", false>(asr::WinoGArgs)" }
#define ASR_WINOGR(CIN, COUT, NT, WAVES, MINW)                                                                    \
    { CIN, COUT, 0, NT, WAVES, MINW, 1, conv3x3_winog<CIN, COUT, false, NT, WAVES, MINW, true>,                   \
      "void asr::conv3x3_winog<" #CIN ", " #COUT ", false, " #NT ", " #WAVES ", " #MINW ", true>(asr::WinoGArgs)" }
static const WinoGVariant g_winog[] = {
    ASR_WINOG(12, 12, 1, 1, 4, 2), ASR_WINOG(12, 12, 1, 1, 8, 2), ASR_WINOG(12, 24, 0, 2, 4, 2), ASR_WINOG(12, 24, 0, 2, 4, 1),
    ASR_WINOG(24, 24, 1, 2, 4, 1), ASR_WINOG(24, 24, 1, 2, 4, 2), ASR_WINOG(24, 24, 1, 2, 8, 2),
    ASR_WINOG(24, 48, 0, 3, 4, 1),
    ASR_WINOG(48, 48, 1, 3, 4, 1), ASR_WINOG(48, 48, 0, 3, 4, 1),
    // the 96-channel blocks of the _rsz model: the weights of all n-tiles do not fit the LDS, so the n-tiles are split
    // into groups (grid.y) and the input transform is repeated per group
    ASR_WINOG(48, 96, 0, 3, 4, 1), ASR_WINOG(96, 96, 1, 1, 8, 2), ASR_WINOG(96, 96, 0, 1, 8, 2),
    ASR_WINOG(96, 96, 1, 1, 4, 2), ASR_WINOG(96, 96, 0, 1, 4, 2),
    // RAW: train-mode forward convolutions and data gradients (C_in / C_out swapped)
    ASR_WINOGR(24, 24, 2, 4, 2), ASR_WINOGR(24, 48, 3, 4, 1), ASR_WINOGR(48, 48, 3, 4, 1),
    ASR_WINOGR(24, 12, 1, 4, 2), ASR_WINOGR(48, 24, 2, 4, 2),
    // (more builds for the training step's tuner: eight waves / the one-wave-per-SIMD pipeline where the other exists)
    ASR_WINOGR(24, 24, 2, 8, 2), ASR_WINOGR(24, 24, 2, 4, 1), ASR_WINOGR(48, 24, 2, 8, 2), ASR_WINOGR(48, 24, 2, 4, 1),
    ASR_WINOGR(24, 12, 1, 8, 2),
    ASR_WINOGR(48, 96, 3, 4, 1), ASR_WINOGR(96, 96, 1, 8, 2), ASR_WINOGR(96, 48, 1, 8, 2),
};
static const int g_num_winog = (int)(sizeof(g_winog) / sizeof(g_winog[0]));

// plan.variant in [3500, 4000): the global-A form; tiles_y/tiles_x = winograd tiles per image
static void candidates_winog(int cin, int cout, int pool, int H, int W, std::vector<ConvPlan> *out, int raw = 0) {
    for (int vi = 0; vi < g_num_winog; ++vi) {
        const WinoGVariant &v = g_winog[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool || v.raw != raw) continue;
        const int lds = 16 * cin * (v.nt == 2 ? 48 : v.nt * 16) * 4;
        if (lds > 160 * 1024) continue;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), 64 * v.waves,
                                                         (size_t)lds) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 1;
        }
        ConvPlan bp{};
        bp.cin = cin; bp.cout = cout; bp.pool = pool;
        bp.H = H; bp.W = W; bp.OH = pool ? H / 2 : H; bp.OW = pool ? W / 2 : W;
        bp.TH = 2; bp.TW = 32; bp.NI = 16;           // an M-tile: 16 consecutive tiles of the batch's tile list,
                                                     // listed in strips of TH tile rows (1 = row-major)
        bp.tiles_y = (H + 1) / 2; bp.tiles_x = (W + 1) / 2;
        bp.threads = 64 * v.waves;
        bp.lds_bytes = lds;
        bp.blocks_per_cu = std::min(nb, 4);
        bp.cost = (double)bp.tiles_y * bp.tiles_x / 16.0 * ((cout + 15) / 16 + v.nt - 1) / v.nt *
                  (16.0 * (cin / 4) * v.nt * 32.0 + 1500.0);
        bp.variant = 3500 + vi;
        bp.symbol = v.symbol;
        out->push_back(bp);
        if (!raw)                                    // the deterministic path's tuner times the other strip heights too
            for (int S : {1}) {                      // (S = 4, 8 measured: equal or slower everywhere, conv4 +14 %)
                bp.TH = S;
                bp.cost *= 1.01;
                out->push_back(bp);
            }
    }
}

// Winograd candidates for the autotuner; plan.variant >= 3000 marks them.  LDS form: plan.TH x TW = the region, plan.NI
// = MY (MX = 16 / MY), plan.tile_floats = nmy * 1000 + nmx; global-A form (>= 3500): tiles_y / tiles_x = tiles per image
void conv_candidates_wino(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out) {
    static const int use = getenv("ASR_CONV_WINO") ? atoi(getenv("ASR_CONV_WINO")) : 1;
    if (!use) return;
    candidates_winog(cin, cout, pool, H, W, out);
    for (int vi = 0; vi < g_num_wino; ++vi) {
        const WinoVariant &v = g_wino[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool || v.raw || v.pw) continue;
        for (int budget : {52 * 1024, 78 * 1024, 158 * 1024}) {
            std::vector<ConvPlan> c;
            enumerate_wino(vi, H, W, budget, c);
            int taken = 0;
            for (auto &cand : c) {
                bool dup = false;
                for (auto &o : *out)
                    if (o.variant == cand.variant && o.TH == cand.TH && o.TW == cand.TW && o.NI == cand.NI) dup = true;
                if (dup) continue;
                finish_wino(cand);
                out->push_back(cand);
                if (++taken >= max_count) break;
            }
        }
    }
}

// block 2 with block 1 evaluated inside by producer waves (plan.fuse1 = 1): candidates for the autotuner
void conv_candidates_wino_fused(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out) {
    static const int use = getenv("ASR_CONV_WINO") ? atoi(getenv("ASR_CONV_WINO")) : 1;
    if (!use) return;
    for (int vi = 0; vi < g_num_wino; ++vi) {
        const WinoVariant &v = g_wino[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool || v.raw || !v.pw || v.in_mode != 0) continue;
        for (int budget : {52 * 1024, 78 * 1024, 158 * 1024}) {
            std::vector<ConvPlan> c;
            enumerate_wino(vi, H, W, budget, c);
            int taken = 0;
            for (auto &cand : c) {
                bool dup = false;
                for (auto &o : *out)
                    if (o.variant == cand.variant && o.TH == cand.TH && o.TW == cand.TW && o.NI == cand.NI) dup = true;
                if (dup) continue;
                finish_wino(cand);
                out->push_back(cand);
                if (++taken >= max_count) break;
            }
        }
    }
}

// kernel symbol of a producer-wave plan for the call's raw input mode (the plan carries the in_mode-0 instantiation)
const char *conv_wino_symbol(const ConvPlan &p, int in_mode) {
    if (p.variant < 3000 || p.variant >= 3500) return p.symbol;
    const WinoVariant &v = g_wino[p.variant - 3000];
    if (!v.pw || in_mode < 0 || in_mode > 2) return p.symbol;
    return g_wino[p.variant - 3000 + in_mode].symbol;
}

// model-chosen Winograd plan for the RAW (training) form: the global-A kernel where it exists (least tile waste),
// else the LDS form's cheapest tiling that leaves room for two workgroups per CU
bool plan_conv_wino_raw(int cin, int cout, int H, int W, ConvPlan *plan) {
    static const int use = getenv("ASR_TRAIN_WINO") ? atoi(getenv("ASR_TRAIN_WINO")) : 1;
    if (!use) return false;
    std::vector<ConvPlan> c;
    candidates_winog(cin, cout, 0, H, W, &c, 1);
    if (!c.empty()) { *plan = c[0]; return true; }
    for (int vi = 0; vi < g_num_wino; ++vi) {
        const WinoVariant &v = g_wino[vi];
        if (!v.raw || v.pw || v.cin != cin || v.cout != cout) continue;
        enumerate_wino(vi, H, W, 78 * 1024, c);
        if (c.empty()) continue;
        *plan = c[0];
        finish_wino(*plan);
        return true;
    }
    return false;
}

// every RAW (training) Winograd schedule of a block, for the training step's tuner: the global-A variants with both
// tile orders and the LDS form's best tilings at three LDS budgets
void conv_candidates_wino_raw(int cin, int cout, int H, int W, int max_count, std::vector<ConvPlan> *out) {
    static const int use = getenv("ASR_TRAIN_WINO") ? atoi(getenv("ASR_TRAIN_WINO")) : 1;
    if (!use) return;
    std::vector<ConvPlan> g;
    candidates_winog(cin, cout, 0, H, W, &g, 1);
    for (auto &c : g) {
        out->push_back(c);
        ConvPlan r = c;
        r.TH = 1;                                    // row-major tile list
        out->push_back(r);
    }
    for (int vi = 0; vi < g_num_wino; ++vi) {
        const WinoVariant &v = g_wino[vi];
        if (!v.raw || v.pw || v.cin != cin || v.cout != cout) continue;
        for (int budget : {52 * 1024, 78 * 1024, 158 * 1024}) {
            std::vector<ConvPlan> c;
            enumerate_wino(vi, H, W, budget, c);
            int taken = 0;
            for (auto &cand : c) {
                bool dup = false;
                for (auto &o : *out)
                    if (o.variant == cand.variant && o.TH == cand.TH && o.TW == cand.TW && o.NI == cand.NI) dup = true;
                if (dup) continue;
                finish_wino(cand);
                out->push_back(cand);
                if (++taken >= max_count) break;
            }
        }
    }
}

// stats (RAW plans only, may be null): partial table of the outputs' per-channel sums, [rows][2][C_out] float64, zeroed
// here (waves without work leave their row untouched); *stats_rows receives the number of rows
hipError_t launch_conv_wino(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk, const float *bnp,
                            float *out, int N, int num_cus, double *stats, int *stats_rows, const Fuse1Args *f1,
                            bool stats_clean, const BnBwdFuse *bf) {
    if (p.variant >= 3500) {
        const WinoGVariant &v = g_winog[p.variant - 3500];
        WinoGArgs a;
        a.in = in; a.wpk = wpk; a.bnp = bnp; a.out = out;
        a.N = N; a.H = p.H; a.W = p.W; a.OH = p.OH; a.OW = p.OW;
        a.ty_img = p.tiles_y; a.tx_img = p.tiles_x;
        a.coutp = (p.cout + 15) / 16 * 16;
        a.tiles = N * a.ty_img * a.tx_img;
        a.total = (a.tiles + 15) / 16;
        a.strips = (p.TH == 1 || p.TH == 4 || p.TH == 8) ? p.TH : 2;
        a.strip_shift = a.strips == 8 ? 3 : a.strips == 4 ? 2 : a.strips == 2 ? 1 : 0;
        if (a.total == 0) return hipSuccess;
        const int waves = p.threads / 64;
        const int ngroups = (a.coutp / 16 + v.nt - 1) / v.nt;
        const int slots = std::max(1, num_cus * std::max(1, p.blocks_per_cu) / ngroups);
        int grid = std::min((a.total + waves - 1) / waves, slots);
        if (grid >= 8) grid &= ~7;           // a multiple of 8: the kernel's XCD-aware walk (blocks b, b + 8 share an XCD)
        a.stats = v.raw ? stats : nullptr;
        a.bf = (bf && a.stats) ? *bf : BnBwdFuse{nullptr, nullptr, nullptr};
        if (a.stats) {
            (void)stats_clean;               // every wave of the launch writes its row (zeros without work): no memset
            if (stats_rows) *stats_rows = grid * waves;
        }
        hipLaunchKernelGGL(v.kernel, dim3(grid, ngroups), dim3(p.threads), p.lds_bytes, s, a);
        return hipGetLastError();
    }
    const WinoVariant &v0 = g_wino[p.variant - 3000];
    if (v0.pw && (!f1 || f1->rsz || f1->in_mode < 0 || f1->in_mode > 2)) return hipErrorInvalidValue;
    if ((double)p.H * p.W * p.cin * 4.0 >= 2147483648.0) return hipErrorInvalidValue;     // per-image buffer descriptor, 32-bit offsets
    const WinoVariant &v = v0.pw ? g_wino[p.variant - 3000 + f1->in_mode] : v0;
    WinoArgs a;
    a.raw = nullptr; a.w1 = nullptr; a.bn1 = nullptr;
    if (v.pw) { a.raw = f1->raw; a.w1 = f1->w1; a.bn1 = f1->bn1; }
    a.in = in; a.wpk = wpk; a.bnp = bnp; a.out = out;
    a.N = N; a.H = p.H; a.W = p.W; a.OH = p.OH; a.OW = p.OW;
    a.RY = p.TH; a.RX = p.TW;
    a.MY = p.NI; a.MX = 16 / p.NI;
    a.nmy = p.tile_floats / 1000; a.nmx = p.tile_floats % 1000;
    a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
    a.coutp = (p.cout + 15) / 16 * 16;
    a.C4P = p.c4p; a.RP = p.rp;
    a.total = N * p.tiles_y * p.tiles_x;
    if (a.total == 0) return hipSuccess;
    const int nt = a.coutp / 16;
    const int ngroups = (nt + v.ntw - 1) / v.ntw;
    // persistent: as many workgroups as fit on the chip at once, each with a contiguous range of regions
    const int slots = std::max(1, num_cus * std::max(1, p.blocks_per_cu) / ngroups);
    a.per = (a.total + slots - 1) / slots;
    const int grid_x = (a.total + a.per - 1) / a.per;
    a.stats = v.raw ? stats : nullptr;
    a.bf = (bf && a.stats) ? *bf : BnBwdFuse{nullptr, nullptr, nullptr};
    if (a.stats) {
        const int rows = grid_x * (p.threads / 64);
        // (no memset: every consumer wave of every workgroup stores its row at the end of the kernel, and grid_x holds
        // only workgroups that have regions)
        if (stats_rows) *stats_rows = rows;
    }
    hipLaunchKernelGGL(v.kernel, dim3(grid_x, ngroups), dim3(p.threads), p.lds_bytes, s, a);
    return hipGetLastError();
}

// upper bound of the rows launch_conv_wino writes into a statistics table: workgroups on the chip x waves
int conv_wino_stats_rows_max(int num_cus) { return num_cus * 8 * 8; }

}  // namespace asr
