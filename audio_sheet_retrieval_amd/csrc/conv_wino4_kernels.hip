// conv3x3_wino4g: conv block (3x3 conv + BN + ELU [+ 2x2 max-pool]) as Winograd F(4x4,3x3) on the fp32 MFMA (gfx950)
//
// F(4x4,3x3) needs 36 products per 4x4 output tile: 2.25 multiply-adds per output pixel and channel pair, against 4
// for F(2x2,3x3) (conv_wino_kernels.hip) and 9 for the direct form.  On an MFMA that runs at the fp32 vector rate the
// product count IS the time, so this is the next 1.8x - paid for with 6x6 transforms (12 packed fp32 operations per
// 6-vector on the way in, 10 on the way out) that, like the F(2x2) ones, never leave the lane:
//   M = 16 tiles (16 consecutive 4x4-pixel tiles of the batch's row-major tile list), N = 16 output channels,
//   K = C_in, one accumulator per transform position p = 6*xi + nu: 36 x 4 = 144 registers (AGPRs), which is why a
//   workgroup handles ONE n-tile (blockIdx.y) - its 36*C_in*16 transformed weights (110 KiB at C_in = 48) are all the
//   LDS holds - and the input transform is repeated per n-tile.  The A operand is read straight from global memory
//   (8-byte loads, two channels = two k-steps each), prefetched one channel block ahead; B operands travel a few MFMAs
//   ahead of their use (see conv3x3_winog).
// STATUS (round 1): correct, but not selected - measured on MI355X it loses to conv3x3_winog (48->48, 40x50 map: 0.63 ms
// against 0.47).  Its MFMA + transform + epilogue part is where the model puts it (0.35 ms with the input loads
// ablated, -DASR_WINO4_ABL=2); the patch loads are the problem: 36 gather loads per 8-channel block, each touching 16
// pixels x 32 bytes, repeated for every n-tile, saturate the texture-address path (one wave per SIMD, four waves per
// CU share it).  The fix is a channel-slab-streaming LDS stage with coalesced loads ("what comes next");
// until then the tuner only sees these candidates with ASR_CONV_WINO4=1.
// Numerics: fp32, weights transformed once in float64.  The transform constants (4, 5, 8, 1/24 ...) make the error
// about 3x that of F(2x2,3x3) per layer; with every block on this kernel the embeddings move by 2-5e-7 (tolerance
// 1e-4; tests/test_gpu_embed_parity.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "asr_kernels.h"
#include "repack_elems.inl"

#ifndef ASR_WINO4_ABL
#define ASR_WINO4_ABL 0      // timing experiments only (wrong results): 2 = no input loads after the first block;
                             // conv3x3_wino4s also: 4 = no weight loads after the first, 8 = no input transform,
                             // 16 = no LDS writes of V, 32 = no output stores
#endif

namespace asr {

typedef float floatx4q __attribute__((ext_vector_type(4)));
typedef float float2q __attribute__((ext_vector_type(2)));

struct Wino4Args {
    const float *in, *wpk, *bnp;
    float *out;
    int N, H, W, OH, OW;
    int ty_img, tx_img;    // 4x4 tiles per image (rows, columns)
    int coutp;
    int tiles;             // tiles in the launch
    int total;             // M-tiles in the launch = ceil(tiles / 16)
    double *stats;         // RAW builds of conv3x3_wino4s, may be null: per-workgroup [sum | sum of squares] of the outputs
    BnBwdFuse bf;          // RAW + stats, bf.z != null: the sums are those of a BatchNorm backward instead (asr_kernels.h)
};

__device__ __forceinline__ float elu_fastq(float y) { return y > 0.0f ? y : __expf(y) - 1.0f; }

template <typename T> __device__ __forceinline__ T splatq(float c);
template <> __device__ __forceinline__ float2q splatq<float2q>(float c) { return float2q{c, c}; }
template <> __device__ __forceinline__ floatx4q splatq<floatx4q>(float c) { return floatx4q{c, c, c, c}; }
template <typename T> __device__ __forceinline__ T fmaq(float c, T x, T y) {
    return __builtin_elementwise_fma(splatq<T>(c), x, y);
}

// B^T d for one 6-vector (Lavin & Gray's F(4,3) matrices), 12 operations
template <typename T> __device__ __forceinline__ void in6(T &d0, T &d1, T &d2, T &d3, T &d4, T &d5) {
    const T r0 = fmaq(4.f, d0, fmaq(-5.f, d2, d4));
    const T t1 = fmaq(-4.f, d2, d4), t2 = fmaq(-4.f, d1, d3);
    const T t3 = d4 - d2, t4 = d3 - d1;
    const T r5 = fmaq(4.f, d1, fmaq(-5.f, d3, d5));
    d0 = r0; d1 = t1 + t2; d2 = t1 - t2; d3 = fmaq(2.f, t4, t3); d4 = fmaq(-2.f, t4, t3); d5 = r5;
}
// A^T m for one 6-vector -> 4 values, 10 operations
template <typename T> __device__ __forceinline__ void out6(T m0, T m1, T m2, T m3, T m4, T m5, T &y0, T &y1, T &y2, T &y3) {
    const T s1 = m1 + m2, s2 = m1 - m2, s3 = m3 + m4, s4 = m3 - m4;
    y0 = (m0 + s1) + s3;
    y1 = fmaq(2.f, s4, s2);
    y2 = fmaq(4.f, s3, s1);
    y3 = fmaq(8.f, s4, s2) + m5;
}

template <int CIN, int COUT, bool POOL, int WAVES, bool RAW>
__global__ __launch_bounds__(64 * WAVES, 1) void conv3x3_wino4g(Wino4Args a) {
    constexpr int KS = CIN / 4, NB = CIN / 8, T = 64 * WAVES;
    constexpr int WS = 16;                       // LDS row: the 16 output channels of this n-tile
    constexpr int WD = 6;                        // B operands in flight ahead of their MFMA
    constexpr int WQ = 72;                       // B operands per channel block (2 k-steps x 36 positions)
    static_assert(CIN % 8 == 0, "channel blocks of 8");
    extern __shared__ __align__(16) float w_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt0 = blockIdx.y;
    // this n-tile's transformed weights: KS*36*4 rows of 16 floats out of rows of coutp
    // (LDS rows ((ks * 36 + p) * 4 + g) of 16 floats, gathered from the packed layout of wino4_pack_kernel)
    for (int i = tid; i < KS * 36 * 4 * 16; i += T) {
        const int row = i >> 4, nn = i & 15;
        const int gg = row & 3, kp = row >> 2, ks = kp / 36, p = kp - ks * 36;
        const int qq = (ks & 1) * 36 + p;
        w_lds[i] = a.wpk[((((size_t)(ks >> 1) * 18 + (qq >> 2)) * 4 + gg) * a.coutp + nt0 * 16 + nn) * 4 + (qq & 3)];
    }
    __syncthreads();

    const int m = lane & 15, g = lane >> 4, n = lane & 15;
    const float *w_lane = w_lds + g * WS + n;
    const int chn = nt0 * 16 + n;
    const bool ch_ok = chn < COUT;
    const float bmean = (!RAW && ch_ok) ? a.bnp[chn] : 0.f;
    const float bscale = (!RAW && ch_ok) ? a.bnp[a.coutp + chn] : 1.f;
    const float bbeta = (!RAW && ch_ok) ? a.bnp[2 * a.coutp + chn] : 0.f;
    const int per_img = a.ty_img * a.tx_img;

    for (int mt = blockIdx.x * WAVES + wave; mt < a.total; mt += gridDim.x * WAVES) {
        // this lane's tile: number 16*mt + m of the batch's tile list
        const int tnum = mt * 16 + m;
        const bool tvalid = tnum < a.tiles;
        const int tcl = min(tnum, a.tiles - 1);
        const int img = tcl / per_img;
        const int trest = tcl - img * per_img;
        const int tty = trest / a.tx_img, ttx = trest - tty * a.tx_img;
        const int py = 4 * tty, px = 4 * ttx;
        const float *ibase = a.in + (int64_t)img * a.H * a.W * CIN + 2 * g;
        // 6x6 patch: clamped row / column element offsets (always loadable) and which rows / columns are inside
        int rowb[6], colb[6];
        unsigned oy_m = 0, ox_m = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = py - 1 + i, x = px - 1 + i;
            oy_m |= (unsigned)(y >= 0 && y < a.H) << i;
            ox_m |= (unsigned)(x >= 0 && x < a.W) << i;
            rowb[i] = min(max(y, 0), a.H - 1) * a.W * CIN;
            colb[i] = min(max(x, 0), a.W - 1) * CIN;
        }
        if (!tvalid) oy_m = 0;
        const bool interior = __builtin_amdgcn_ballot_w64(oy_m != 0x3fu || ox_m != 0x3fu) == 0;
        // where the tile's outputs go: element offset of its top-left (pooled) output pixel, rows / columns that exist
        unsigned my_off;
        int my_ext;                               // rows | cols << 8 (0 rows: nothing to store)
        if (POOL) {
            my_off = (((unsigned)img * a.OH + 2 * tty) * a.OW + 2 * ttx) * COUT;
            const int nr = min(2, a.OH - 2 * tty), nc = min(2, a.OW - 2 * ttx);
            my_ext = (tvalid && nr > 0 && nc > 0) ? (nr | (nc << 8)) : 0;
        } else {
            my_off = (((unsigned)img * a.H + py) * a.W + px) * COUT;
            const int nr = min(4, a.H - py), nc = min(4, a.W - px);
            my_ext = tvalid ? (nr | (nc << 8)) : 0;
        }

        floatx4q acc[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) acc[p] = floatx4q{0.f, 0.f, 0.f, 0.f};

        float2q nxt[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) nxt[i][j] = *reinterpret_cast<const float2q *>(ibase + rowb[i] + colb[j]);
        float wpre[WD];
#pragma unroll
        for (int q = 0; q < WD; ++q) wpre[q] = w_lane[q * 4 * WS];
#pragma unroll 1
        for (int t = 0; t < NB; ++t) {
            float2q dp[6][6];
            if (interior) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) dp[i][j] = nxt[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        dp[i][j] = (((oy_m >> i) & (ox_m >> j)) & 1u) ? nxt[i][j] : float2q{0.f, 0.f};
            }
            if (t + 1 < NB && !(ASR_WINO4_ABL & 2)) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        nxt[i][j] = *reinterpret_cast<const float2q *>(ibase + rowb[i] + colb[j] + 8 * (t + 1));
            }
            // V = B^T d B in place: columns, then rows
#pragma unroll
            for (int j = 0; j < 6; ++j) in6(dp[0][j], dp[1][j], dp[2][j], dp[3][j], dp[4][j], dp[5][j]);
#pragma unroll
            for (int i = 0; i < 6; ++i) in6(dp[i][0], dp[i][1], dp[i][2], dp[i][3], dp[i][4], dp[i][5]);
            // 72 MFMAs: k-steps 2t, 2t+1 x 36 positions; B operand q = (k-step q / 36, position q % 36)
            const float *wk = w_lane + (2 * t) * (36 * 4 * WS);
            const float *wn = (t + 1 < NB) ? wk + 2 * 36 * 4 * WS : w_lane;
            float wv[WQ + WD];
#pragma unroll
            for (int q = 0; q < WD; ++q) wv[q] = wpre[q];
#pragma unroll
            for (int q0 = 0; q0 < WQ; q0 += 2) {
#pragma unroll
                for (int q = q0; q < q0 + 2; ++q)
                    wv[q + WD] = (q + WD < WQ) ? wk[(q + WD) * 4 * WS] : wn[(q + WD - WQ) * 4 * WS];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = q0; q < q0 + 2; ++q) {
                    const int c = q / 36, p = q % 36;
                    acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[p / 6][p % 6][c], wv[q], acc[p], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < WD; ++q) wpre[q] = wv[WQ + q];
        }

        // ---- output transform Y = A^T M A for the lane's four tiles at once (float4 = tiles r = 0..3): columns
        // first (36 -> 24 values), then row by row with the epilogue of that row (pair)
        unsigned eo[4];
        int ee[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            eo[r] = (unsigned)__shfl((int)my_off, 4 * g + r);
            ee[r] = __shfl(my_ext, 4 * g + r);
        }
        if (!ch_ok) continue;
        floatx4q tc[4][6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
            out6(acc[nu], acc[6 + nu], acc[12 + nu], acc[18 + nu], acc[24 + nu], acc[30 + nu], tc[0][nu], tc[1][nu], tc[2][nu],
                 tc[3][nu]);
        if (POOL) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {                 // pooled row: output rows 2pr, 2pr+1 of the tile
                floatx4q ya[4], yb[4];
                out6(tc[2 * pr][0], tc[2 * pr][1], tc[2 * pr][2], tc[2 * pr][3], tc[2 * pr][4], tc[2 * pr][5], ya[0], ya[1],
                     ya[2], ya[3]);
                out6(tc[2 * pr + 1][0], tc[2 * pr + 1][1], tc[2 * pr + 1][2], tc[2 * pr + 1][3], tc[2 * pr + 1][4],
                     tc[2 * pr + 1][5], yb[0], yb[1], yb[2], yb[3]);
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (pr >= (ee[r] & 0xff) || pc >= (ee[r] >> 8)) continue;
                        const float v0 = ya[2 * pc][r], v1 = ya[2 * pc + 1][r], v2 = yb[2 * pc][r], v3 = yb[2 * pc + 1][r];
                        const float hi = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                        const float lo = fminf(fminf(v0, v1), fminf(v2, v3));
                        const float x = bscale >= 0.0f ? hi : lo;      // max commutes with the monotone BN + ELU
                        a.out[(size_t)eo[r] + (size_t)(pr * a.OW + pc) * COUT + chn] =
                            elu_fastq((x - bmean) * bscale + bbeta);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                floatx4q y[4];
                out6(tc[i][0], tc[i][1], tc[i][2], tc[i][3], tc[i][4], tc[i][5], y[0], y[1], y[2], y[3]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (i >= (ee[r] & 0xff) || j >= (ee[r] >> 8)) continue;
                        const float v = RAW ? y[j][r] : elu_fastq((y[j][r] - bmean) * bscale + bbeta);
                        a.out[(size_t)eo[r] + (size_t)(i * a.W + j) * COUT + chn] = v;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv3x3_wino4s: F(4x4,3x3) with WAVE SPECIALISATION - the form the tuner actually selects.
// What stopped conv3x3_wino4g: its transformed weights (36 positions x C_in x 16 floats = 110 KiB per n-tile at
// C_in = 48) fill the LDS, so a workgroup owns one n-tile and the 6x6 input transform - and its 36 gather loads per
// channel block - is repeated per n-tile.  Here a workgroup is ONE producer wave plus one consumer wave per n-tile
// (four waves for 48 output channels: one per SIMD):
//   producer : gathers the 6x6 patches of an M-tile (16 tiles) for one 8-channel block, forms V = B^T d B in
//              registers (the lane layout IS the A-operand layout) and writes it to one of two LDS buffers
//              (36 x 16 x 8 floats); it never issues an MFMA and the consumers never issue a transform instruction -
//              the vector work that the fp32 MFMA cannot overlap inside one wave runs on another SIMD;
//   consumers: read V as A operands (one conflict-free ds_read_b64 per position = two k-steps), multiply by their
//              n-tile's transformed weights - kept entirely in registers when they fit (C_in = 24: 216 values),
//              otherwise streamed from global memory (L2-resident, 331 KiB per layer at 48 -> 48) one channel block
//              ahead into a ping-pong register set - and run the output transform / BN / ELU / pool of their n-tile.
// One workgroup barrier per (M-tile, channel block) step hands a V buffer over.  The input transform is paid once per
// M-tile, three of the four SIMDs run MFMAs only, and there are 1.78x fewer of them than in F(2x2,3x3).
// SLICE < C_out (96 output channels): blockIdx.y picks SLICE of them, so a workgroup stays at one wave per SIMD with
// the two 72-register weight sets; the workgroups (x, 0) and (x, 1) walk the same M-tiles at the same time on the same
// XCD (gridDim.x is a multiple of 8), the second reads its patches from L2.
// PW = 2 (two n-tiles at most, so that the workgroup stays at four waves): TWO producer waves.  One producer needs
// ~3600 cycles per (M-tile, channel block) item whatever the number of output channels, a consumer 72 MFMAs = 2304;
// with three consumers the two sides are balanced, with two (24 output channels) the producer is the critical path
// (24 -> 24 at 80x100: 0.76 ms against 0.66 for the F(2x2) global-A kernel - round 2 did not select it).  The
// producers take alternate steps of the flattened (M-tile, channel block) sequence - producer p owns V buffer p - so
// each has two step times per item; every wave still passes one workgroup barrier per step.
// RAW (training step, un-pooled blocks): the output is the raw convolution z, and the consumers also gather the
// BatchNorm statistics of what they store - float64 sums per lane, one row [sum(C_out) | sum of squares(C_out)] per
// workgroup in a.stats (see wino_stats_store in conv_wino_kernels.hip).
template <int CIN, int COUT, bool POOL, int SLICE = COUT, int PW = 1, bool RAW = false>
__global__ __launch_bounds__(64 * (PW + (SLICE + 15) / 16), 1) void conv3x3_wino4s(Wino4Args a) {
    static_assert(!(RAW && POOL), "the raw form has no pooled epilogue");
    constexpr int NT = (SLICE + 15) / 16, NB = CIN / 8, KS = CIN / 4;
    static_assert(PW == 1 || PW == 2, "one or two producer waves");
    static_assert(PW + NT <= 4, "one wave per SIMD");
    static_assert(SLICE == COUT || (SLICE % 16 == 0 && COUT % SLICE == 0), "slices are whole n-tiles");
    constexpr int VB = 36 * 16 * 8;                       // floats per V buffer
    static_assert(CIN % 8 == 0, "channel blocks of 8");
    extern __shared__ __align__(16) float vbuf[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave < PW;
    const int m = lane & 15, g = lane >> 4, n = lane & 15;
    const int per_img = a.ty_img * a.tx_img;

    // M-tiles of this workgroup: XCD-aware contiguous walk (blocks b, b + 8 share an XCD; see conv3x3_winog)
    int mt, mt_end, mt_stride;
    if ((gridDim.x & 7) == 0) {
        const int per_x = (a.total + 7) >> 3;
        const int r0 = (int)(blockIdx.x & 7) * per_x;
        mt_end = min(r0 + per_x, a.total);
        mt_stride = (int)(gridDim.x >> 3);
        mt = r0 + (int)(blockIdx.x >> 3);
    } else {
        mt_end = a.total; mt_stride = (int)gridDim.x; mt = (int)blockIdx.x;
    }
    if (mt >= mt_end) {                                   // the whole workgroup (uniform)
        if constexpr (RAW) {                                  // its row of the statistics table: zeros (no memset before the launch)
            if (a.stats)
                for (int c = threadIdx.x; c < 2 * COUT; c += blockDim.x) a.stats[(size_t)blockIdx.x * 2 * COUT + c] = 0.0;
        }
        return;
    }

    // lane -> tile of an M-tile (row-major list of 4x4 tiles; lanes past the end work on clamped addresses)
    auto tile_of = [&](int mtile, int &img, int &tty, int &ttx, bool &tvalid) {
        const int tnum = mtile * 16 + m;
        tvalid = tnum < a.tiles;
        const int tcl = min(tnum, a.tiles - 1);
        img = tcl / per_img;
        const int trest = tcl - img * per_img;
        tty = trest / a.tx_img;
        ttx = trest - tty * a.tx_img;
    };
    // this lane's 8-byte slot of a position's 16 x 8 block, [k-group g][tile m][2]: the producer's ds_write_b64 is
    // served in groups of 16 consecutive lanes (one g, banks modulo 32 dwords) and the consumers' ds_read_b64 in
    // 32-lane halves (two g's, banks modulo 64) - with the tile index fastest both see every bank exactly once.
    // (Round 2 had [m][g][2]: the 16 lanes of a write group sat 32 bytes apart on 4 bank pairs, a 4-way conflict, 16
    // LDS cycles per store instead of 6; the reads 2-way.  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.74.)
    float *vlane = vbuf + (g * 32 + 2 * m);
#ifdef ASR_WINO4_OLD_V
    vlane = vbuf + (m * 8 + 2 * g);
#endif

    if (producer) {
        // the 36 patch pixels of this lane's tile as 32-bit BYTE offsets from the tensor base (the launcher admits
        // tensors below 4 GiB only): every load is "uniform base + channel block (scalar) + lane offset" - no vector
        // address arithmetic per load
        unsigned poff[6][6];
        unsigned oy_m = 0, ox_m = 0;
        auto setup = [&](int mtile) {
            int img, tty, ttx;
            bool tvalid;
            tile_of(mtile, img, tty, ttx, tvalid);
            const int py = 4 * tty, px = 4 * ttx;
            const unsigned ib = ((unsigned)img * a.H * a.W * CIN + 2 * g) * 4u;
            unsigned rowb[6], colb[6];
            oy_m = 0; ox_m = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int y = py - 1 + i, x = px - 1 + i;
                oy_m |= (unsigned)(y >= 0 && y < a.H) << i;
                ox_m |= (unsigned)(x >= 0 && x < a.W) << i;
                rowb[i] = (unsigned)(min(max(y, 0), a.H - 1) * a.W * CIN) * 4u;
                colb[i] = (unsigned)(min(max(x, 0), a.W - 1) * CIN) * 4u;
            }
            if (!tvalid) oy_m = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) poff[i][j] = ib + rowb[i] + colb[j];
        };
        float2q nxt[6][6];
        auto load = [&](int t) {
            const char *cb = reinterpret_cast<const char *>(a.in + 8 * t);       // uniform
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) nxt[i][j] = *reinterpret_cast<const float2q *>(cb + poff[i][j]);
        };
        // the flattened step sequence s = 0, 1, ... (M-tile by M-tile, NB channel blocks each); this producer's items
        // are s = wave, wave + PW, ...  Barrier b (every wave passes it) says "V(b) is written and everybody has
        // finished reading V(b - 1)": V(s) goes into buffer s & 1, last read in step s - 2, i.e. it may be written
        // once barrier s - 1 is behind this wave, and must be written before barrier s.
        const int n_mt = (mt_end - mt + mt_stride - 1) / mt_stride;
        const int n_steps = n_mt * NB;
        int passed = 0;                                       // barriers this wave has passed
        int t = wave;                                         // channel block of this producer's current item
        while (t >= NB) { t -= NB; mt += mt_stride; }
        if (mt < mt_end) { setup(mt); load(t); }
        for (int step = wave; step < n_steps; step += PW) {
            float2q dp[6][6];
            const unsigned oy = oy_m, ox = ox_m;
            const bool interior = __builtin_amdgcn_ballot_w64(oy != 0x3fu || ox != 0x3fu) == 0;
            if (interior) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) dp[i][j] = nxt[i][j];
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) dp[i][j] = (((oy >> i) & (ox >> j)) & 1u) ? nxt[i][j] : float2q{0.f, 0.f};
            }
            // this producer's next item is requested before the current one is transformed
            {
                int tn = t + PW, mtn = mt;
                while (tn >= NB) { tn -= NB; mtn += mt_stride; }
                if (!(ASR_WINO4_ABL & 2) && step + PW < n_steps) {
                    if (mtn != mt) setup(mtn);
                    load(tn);
                }
                t = tn; mt = mtn;
            }
            if (!(ASR_WINO4_ABL & 8)) {
#pragma unroll
            for (int j = 0; j < 6; ++j) in6(dp[0][j], dp[1][j], dp[2][j], dp[3][j], dp[4][j], dp[5][j]);
#pragma unroll
            for (int i = 0; i < 6; ++i) in6(dp[i][0], dp[i][1], dp[i][2], dp[i][3], dp[i][4], dp[i][5]);
            }
            while (passed < step) { __syncthreads(); ++passed; }      // PW = 2: barrier step - 1 (none for PW = 1)
            float *vb = vlane + (step & 1) * VB;
            if (!(ASR_WINO4_ABL & 16)) {
#pragma unroll
            for (int p = 0; p < 36; ++p) *reinterpret_cast<float2q *>(vb + p * 128) = dp[p / 6][p % 6];
            } else {
#pragma unroll
            for (int p = 0; p < 36; ++p) asm volatile("" ::"v"(dp[p / 6][p % 6]));
            }
            __syncthreads();                                  // barrier `step`: hands buffer (step & 1) to the consumers
            ++passed;
        }
        // the steps after this producer's last item, and the consumers' barrier after the last step
        while (passed < n_steps + 1) { __syncthreads(); ++passed; }
        return;
    }

    // ---------------------------------------------------------------------------------------------- consumers
    const int nt = wave - PW;
    const int chn = (int)blockIdx.y * SLICE + nt * 16 + n;
    const bool ch_ok = chn < COUT;
    float bmean = (!RAW && ch_ok) ? a.bnp[chn] : 0.f;
    float bscale = (!RAW && ch_ok) ? a.bnp[a.coutp + chn] : 1.f;
    float bbeta = (!RAW && ch_ok) ? a.bnp[2 * a.coutp + chn] : 0.f;
    float bistd = 0.f;                                        // BatchNorm-backward sums (a.bf): mu, gamma*inv_std, beta, inv_std
    const bool bnb = ASR_BNB_FUSE_BUILD && RAW && a.stats && a.bf.z != nullptr;
    if (bnb && ch_ok) {
        bmean = a.bf.cst[chn]; bistd = a.bf.cst[COUT + chn];
        bscale = a.bf.cst[2 * COUT + chn]; bbeta = a.bf.cst[3 * COUT + chn];
    }
    double st1 = 0.0, st2 = 0.0;                              // RAW + a.stats: sums of this lane's channel
    // B operands of a channel block: 18 float4 per lane (row = k-step parity * 36 + position, four rows per float4; see
    // wino4_pack_kernel), at [block][row / 4][g][coutp][4] - a uniform block pointer (scalar arithmetic) + this lane's
    // 32-bit byte offset + a compile-time row-group offset
    const unsigned wlane = ((unsigned)g * a.coutp + chn) * 16u;
    const float *wl = a.wpk;                                  // uniform
    constexpr int wstep = 4 * ((COUT + 15) / 16 * 16);        // floats per (row, g, channel) plane (a.coutp);
                                                              // a row group of four is wstep * 16 bytes
    // streamed weights in two register sets of 72 that alternate per step: the weights of the NEXT channel block (the
    // first block of the next M-tile after the last one) are requested right after the step's barrier, as 18 dwordx4
    // loads, and have the whole step to arrive from L2.  Measured alternatives (conv6, 0.404 ms with 72 dword loads
    // after the barrier): the same loads before the barrier 0.440; in three bursts between the MFMA groups 0.473; one
    // register set reloaded in place after each group 0.449 - vector-memory instructions issued between MFMAs cost
    // more than they hide
    floatx4q bw[2][18];
#pragma unroll
    for (int j = 0; j < 18; ++j)
        bw[0][j] = *reinterpret_cast<const floatx4q *>(reinterpret_cast<const char *>(wl) + (size_t)j * (wstep * 16) + wlane);
    int step = 0;
    for (; mt < mt_end; mt += mt_stride) {
        unsigned my_off;
        int my_ext;                                           // rows | cols << 8 (0 rows: nothing to store)
        {
            int img, tty, ttx;
            bool tvalid;
            tile_of(mt, img, tty, ttx, tvalid);
            if (POOL) {
                my_off = (((unsigned)img * a.OH + 2 * tty) * a.OW + 2 * ttx) * COUT;
                const int nr = min(2, a.OH - 2 * tty), nc = min(2, a.OW - 2 * ttx);
                my_ext = (tvalid && nr > 0 && nc > 0) ? (nr | (nc << 8)) : 0;
            } else {
                my_off = (((unsigned)img * a.H + 4 * tty) * a.W + 4 * ttx) * COUT;
                const int nr = min(4, a.H - 4 * tty), nc = min(4, a.W - 4 * ttx);
                my_ext = tvalid ? (nr | (nc << 8)) : 0;
            }
        }
        // byte offsets of this lane in row groups 0, 2, 4 ... 16 of a channel block (the odd ones are an immediate
        // offset away); opaque, so that the weight loads - which do not depend on the M-tile - are neither hoisted out
        // of this loop (a layer's B operands do not fit the register file; it spilled them) nor folded into 64-bit
        // vector pointers
        unsigned voff[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            voff[k] = wlane + (unsigned)k * (2 * wstep * 16);
            asm volatile("" : "+v"(voff[k]));
        }
        floatx4q acc[36];
        // one step: 72 MFMAs on V buffer vb in three groups of 12 positions; the A operands of group g + 1 are read
        // from LDS before the MFMAs of group g are issued.  FIRST: the accumulators start from the MFMA's inline zero
        // C operand instead of 144 register writes per M-tile
        // reload(wn, set): request the 72 weights of the channel block at wn into register set `set`.  Scalar base +
        // one of nine fixed 32-bit lane offsets + immediate: no vector address arithmetic (the offsets are made opaque
        // HERE so that their zero-extension stays in this block, where instruction selection can fold it into the
        // scalar-base addressing mode)
        auto reload = [&](const float *wn, auto set_c) {
            constexpr int SET = decltype(set_c)::value;
            if (ASR_WINO4_ABL & 4) return;
            const char *wnb = reinterpret_cast<const char *>(wn);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                unsigned vo = voff[k];
                asm volatile("" : "+v"(vo));
                bw[SET][2 * k] = *reinterpret_cast<const floatx4q *>(wnb + vo);
                bw[SET][2 * k + 1] = *reinterpret_cast<const floatx4q *>(wnb + (wstep * 16) + vo);
            }
        };
        auto run_step = [&](const float *vb, const float *wn, auto cur_c, auto first) {
            constexpr bool FIRST = decltype(first)::value;
            constexpr int CUR = decltype(cur_c)::value;
            using NXT = std::integral_constant<int, CUR ^ 1>;
            __syncthreads();                                  // V of this step is in buffer (step & 1)
            reload(wn, NXT{});
            __builtin_amdgcn_sched_barrier(0);
            float2q dp[2][12];
#pragma unroll
            for (int i = 0; i < 12; ++i) dp[0][i] = *reinterpret_cast<const float2q *>(vb + i * 128);
#pragma unroll
            for (int grp = 0; grp < 3; ++grp) {
                if (grp < 2) {
#pragma unroll
                    for (int i = 0; i < 12; ++i)
                        dp[(grp + 1) & 1][i] = *reinterpret_cast<const float2q *>(vb + ((grp + 1) * 12 + i) * 128);
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const int p = grp * 12 + i;
                        if (FIRST && c == 0)
                            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[grp & 1][i][c], bw[CUR][(c * 36 + p) >> 2][(c * 36 + p) & 3],
                                                                          floatx4q{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[grp & 1][i][c], bw[CUR][(c * 36 + p) >> 2][(c * 36 + p) & 3], acc[p],
                                                                          0, 0,
                                                                          0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using C0 = std::integral_constant<int, 0>;
        using C1 = std::integral_constant<int, 1>;
        auto wblock = [&](int t) {                            // the channel block after step t's, as a SCALAR pointer
            unsigned blk = (unsigned)(((t + 1) % NB) * 72) * wstep;
            asm volatile("" : "+s"(blk));                     // (a constant here would be folded into the lane offsets)
            return wl + blk;
        };
        run_step(vlane + (step & 1) * VB, wblock(0), C0{}, std::true_type{});
        ++step;
#pragma unroll 1
        for (int t = 1; t + 1 < NB; t += 2) {
            run_step(vlane + (step & 1) * VB, wblock(t), C1{}, std::false_type{});
            ++step;
            run_step(vlane + (step & 1) * VB, wblock(t + 1), C0{}, std::false_type{});
            ++step;
        }
        if constexpr ((NB & 1) == 0) {                        // an odd last step fills set 0 for the next M-tile
            run_step(vlane + (step & 1) * VB, wblock(NB - 1), C1{}, std::false_type{});
            ++step;
        } else {                                              // the last step filled set 1: 72 moves per M-tile
#pragma unroll
            for (int j = 0; j < 18; ++j) bw[0][j] = bw[1][j];
        }

        // ---- output transform Y = A^T M A for the lane's four tiles at once (float4 = tiles r = 0..3), BN + ELU
        // (+ pool), store: this consumer's n-tile only
        unsigned eo[4];
        int ee[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            eo[r] = (unsigned)__shfl((int)my_off, 4 * g + r);
            ee[r] = __shfl(my_ext, 4 * g + r);
        }
        if (!ch_ok) continue;
        floatx4q tc[4][6];
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
            out6(acc[nu], acc[6 + nu], acc[12 + nu], acc[18 + nu], acc[24 + nu], acc[30 + nu], tc[0][nu], tc[1][nu], tc[2][nu],
                 tc[3][nu]);
        if (POOL) {
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {                 // pooled row: output rows 2pr, 2pr+1 of the tile
                floatx4q ya[4], yb[4];
                out6(tc[2 * pr][0], tc[2 * pr][1], tc[2 * pr][2], tc[2 * pr][3], tc[2 * pr][4], tc[2 * pr][5], ya[0], ya[1],
                     ya[2], ya[3]);
                out6(tc[2 * pr + 1][0], tc[2 * pr + 1][1], tc[2 * pr + 1][2], tc[2 * pr + 1][3], tc[2 * pr + 1][4],
                     tc[2 * pr + 1][5], yb[0], yb[1], yb[2], yb[3]);
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (pr >= (ee[r] & 0xff) || pc >= (ee[r] >> 8)) continue;
                        const float v0 = ya[2 * pc][r], v1 = ya[2 * pc + 1][r], v2 = yb[2 * pc][r], v3 = yb[2 * pc + 1][r];
                        const float hi = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                        const float lo = fminf(fminf(v0, v1), fminf(v2, v3));
                        const float x = bscale >= 0.0f ? hi : lo;      // max commutes with the monotone BN + ELU
                        const float res = elu_fastq((x - bmean) * bscale + bbeta);
                        if (ASR_WINO4_ABL & 32) asm volatile("" ::"v"(res));
                        else a.out[(size_t)eo[r] + (size_t)(pr * a.OW + pc) * COUT + chn] = res;
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                floatx4q y[4];
                out6(tc[i][0], tc[i][1], tc[i][2], tc[i][3], tc[i][4], tc[i][5], y[0], y[1], y[2], y[3]);
                float rs1 = 0.f, rs2 = 0.f;                   // RAW: this output row's float32 partial sums (<= 16 terms)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (i >= (ee[r] & 0xff) || j >= (ee[r] >> 8)) continue;
                        const float res = RAW ? y[j][r] : elu_fastq((y[j][r] - bmean) * bscale + bbeta);
                        if (RAW) {
                            if (bnb) {
                                const size_t e = (size_t)eo[r] + (size_t)(i * a.W + j) * COUT + chn;
                                bnb_acc(res, a.bf.z[e], bnb_mult(a.bf.tie, e), bmean, bscale, bbeta, bistd, rs1, rs2);
                            } else { rs1 += res; rs2 = fmaf(res, res, rs2); }
                        }
                        if (ASR_WINO4_ABL & 32) asm volatile("" ::"v"(res));
                        else a.out[(size_t)eo[r] + (size_t)(i * a.W + j) * COUT + chn] = res;
                    }
                }
                if (RAW) { st1 += (double)rs1; st2 += (double)rs2; }
            }
        }
    }
    __syncthreads();                                          // pairs with the producer's final barrier
    if constexpr (RAW) {
        if (a.stats) {                                        // lanes (g, n): the four g hold the same channel
            st1 += __shfl_xor(st1, 16); st2 += __shfl_xor(st2, 16);
            st1 += __shfl_xor(st1, 32); st2 += __shfl_xor(st2, 32);
            if (g == 0 && ch_ok) {
                a.stats[((size_t)blockIdx.x * 2) * COUT + chn] = st1;
                a.stats[((size_t)blockIdx.x * 2 + 1) * COUT + chn] = st2;
            }
        }
    }
}

// ---- weight transform: U = G g G^T (6x6 per channel pair) in float64, stored per channel block t of 8 channels as
// [t][row / 4][g][coutp][row % 4] with row = (k-step parity) * 36 + (p = 6 xi + nu): a lane (g, channel) finds FOUR B
// operands in 16 contiguous bytes, a wave's dwordx4 load covers 4 x 256 contiguous bytes (conv3x3_wino4s streams the
// weights with 18 such loads per step instead of 72 dword loads).  Channel order of the kernels: k-steps 2t, 2t+1 of
// lane group g <-> contraction channels 8t+2g, 8t+2g+1.
// forward / data-gradient roles as in wino_pack_kernel (conv_wino_kernels.hip)
__global__ void wino4_pack_kernel(const float *W, int cin, int cout, int dgrad, float *wpk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= wino4_pack_count(cin, cout, dgrad)) return;
    wino4_pack_elem(idx, W, cin, cout, dgrad, wpk);            // (repack_elems.inl: shared with repack_all_kernel)
}

size_t wino4_wpack_floats(int cin, int cout) { return (cin % 8) ? 0 : (size_t)36 * cin * ((cout + 15) / 16 * 16); }

hipError_t launch_wino4_pack(hipStream_t s, const float *W, int cin, int cout, float *wpk, int dgrad) {
    const int kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
    if (kdim % 8) return hipSuccess;                 // no F(4x4) variant for this block
    const int total = kdim * ((ndim + 15) / 16 * 16);
    hipLaunchKernelGGL(wino4_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, W, cin, cout, dgrad, wpk);
    return hipGetLastError();
}

// ---- instantiation table ---------------------------------------------------------------------------------------
struct Wino4Variant {
    int cin, cout, pool, waves, raw;
    void (*kernel)(Wino4Args);
    const char *symbol;
    int spec;                 // 1: conv3x3_wino4s (one producer wave + one consumer wave per n-tile)
    int slice;                // conv3x3_wino4s: output channels per workgroup (grid.y = cout / slice); 0 = all
};
#define ASR_BOOLSTRQ_0 "false"
#define ASR_BOOLSTRQ_1 "true"
#define ASR_WINO4(CIN, COUT, POOL, WAVES)                                                                         \
    { CIN, COUT, POOL, WAVES, 0, conv3x3_wino4g<CIN, COUT, (POOL != 0), WAVES, false>,                            \
      "void asr::conv3x3_wino4g<" #CIN ", " #COUT ", " ASR_BOOLSTRQ_##POOL ", " #WAVES ", false>(asr::Wino4Args)" }
#define ASR_WINO4S(CIN, COUT, POOL)                                                                               \
    { CIN, COUT, POOL, 1 + (COUT + 15) / 16, 0, conv3x3_wino4s<CIN, COUT, (POOL != 0)>,                           \
      "void asr::conv3x3_wino4s<" #CIN ", " #COUT ", " ASR_BOOLSTRQ_##POOL ", " #COUT ", 1, false>(asr::Wino4Args)", 1, 0 }
#define ASR_WINO4S2(CIN, COUT, POOL)                                                                              \
    { CIN, COUT, POOL, 2 + (COUT + 15) / 16, 0, conv3x3_wino4s<CIN, COUT, (POOL != 0), COUT, 2>,                  \
      "void asr::conv3x3_wino4s<" #CIN ", " #COUT ", " ASR_BOOLSTRQ_##POOL ", " #COUT ", 2, false>(asr::Wino4Args)", 1, 0 }
#define ASR_WINO4SR(CIN, COUT)                                                                                    \
    { CIN, COUT, 0, 1 + (COUT + 15) / 16, 1, conv3x3_wino4s<CIN, COUT, false, COUT, 1, true>,                     \
      "void asr::conv3x3_wino4s<" #CIN ", " #COUT ", false, " #COUT ", 1, true>(asr::Wino4Args)", 1, 0 }
#define ASR_WINO4SL(CIN, COUT, POOL, SLICE)                                                                       \
    { CIN, COUT, POOL, 1 + SLICE / 16, 0, conv3x3_wino4s<CIN, COUT, (POOL != 0), SLICE>,                          \
      "void asr::conv3x3_wino4s<" #CIN ", " #COUT ", " ASR_BOOLSTRQ_##POOL ", " #SLICE ", 1, false>(asr::Wino4Args)", 1, SLICE }
static const Wino4Variant g_wino4[] = {
    ASR_WINO4(24, 24, 1, 4), ASR_WINO4(24, 48, 0, 4), ASR_WINO4(48, 48, 1, 4), ASR_WINO4(48, 48, 0, 4),
    ASR_WINO4S(24, 24, 1), ASR_WINO4S(24, 48, 0), ASR_WINO4S(48, 48, 1), ASR_WINO4S(48, 48, 0),
    ASR_WINO4S2(24, 24, 1),                               // two producer waves + two consumers
    // the 96-channel blocks of the _rsz model: two workgroups of 48 output channels each per M-tile
    ASR_WINO4SL(48, 96, 0, 48), ASR_WINO4SL(96, 96, 1, 48), ASR_WINO4SL(96, 96, 0, 48),
    // RAW builds for the training step (forward and data gradient of the 48-channel blocks)
    ASR_WINO4SR(24, 48), ASR_WINO4SR(48, 48), ASR_WINO4SR(48, 24),
};
static const int g_num_wino4 = (int)(sizeof(g_wino4) / sizeof(g_wino4[0]));

// plan.variant >= 4000: F(4x4,3x3), global-A form; tiles_y / tiles_x = 4x4 tiles per image
void conv_candidates_wino4(int cin, int cout, int pool, int H, int W, std::vector<ConvPlan> *out) {
    // the wave-specialised form is a regular candidate; the one-n-tile-per-workgroup form (slower everywhere) only
    // with ASR_CONV_WINO4=1; ASR_CONV_WINO4=0 removes both
    static const int use = getenv("ASR_CONV_WINO4") ? atoi(getenv("ASR_CONV_WINO4")) : -1;
    if (use == 0) return;
    for (int vi = 0; vi < g_num_wino4; ++vi) {
        const Wino4Variant &v = g_wino4[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool || v.raw) continue;
        if (!v.spec && use != 1) continue;
        const int lds = v.spec ? 2 * 36 * 16 * 8 * 4 : 36 * cin * 16 * 4;
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
        bp.TH = 4; bp.TW = 64; bp.NI = 16;
        bp.tiles_y = (H + 3) / 4; bp.tiles_x = (W + 3) / 4;
        bp.threads = 64 * v.waves;
        bp.lds_bytes = lds;
        bp.blocks_per_cu = v.spec ? 1 : std::min(nb, 4);     // specialised: one wave per SIMD (its registers need it)
        bp.cost = (double)bp.tiles_y * bp.tiles_x / 16.0 * ((cout + 15) / 16) * (36.0 * (cin / 4) * 32.0 + 3000.0) *
                  (v.spec ? 0.4 : 1.0);
        bp.variant = 4000 + vi;
        bp.symbol = v.symbol;
        out->push_back(bp);
    }
}

// the RAW (training) builds of a block, for the training step's tuner: forward convolutions and data gradients.
// Round 3 kept them out of the forward pass: against the FREE float64 oracle the median relative error of the 54
// gradient tensors rose from 1.3e-5 to 1.3e-3 (F(4x4)'s float32 rounding - transform constants up to 8 and 1/24, z moves
// by 3e-6 of its maximum, ten times F(2x2)'s - flips the arg-max of ten times as many 2x2 pooling windows whose two
// largest values nearly tie, and each flip routes a gradient to another pixel).  Round 4 measures what is left once the
// device's own pooling selection is imposed on the oracle (tests/test_gpu_train_routed.py): every gradient tensor within
// 7.7e-5 of its maximum at batch 512 with F(4x4) forced for forward and backward (6.3e-5 without it), under the 1e-4
// bar - the flips were the whole difference, and a flip is a tie broken the other way, not an error.  So the forward
// builds compete too (conv6 0.265 -> 0.235 ms, conv7 / conv8 0.092 -> 0.068 per direction).
// ASR_TRAIN_WINO4=0: none; 5: data gradients only (round 3's default); 2 / 3 / 4: forced for both / forward only /
// data gradients only (asr_api_train.hip).
void conv_candidates_wino4_raw(int cin, int cout, int H, int W, std::vector<ConvPlan> *out, int dgrad) {
    static const int use = getenv("ASR_TRAIN_WINO4") ? atoi(getenv("ASR_TRAIN_WINO4")) : -1;
    if (use == 0 || (use == 5 && !dgrad)) return;
    for (int vi = 0; vi < g_num_wino4; ++vi) {
        const Wino4Variant &v = g_wino4[vi];
        if (v.cin != cin || v.cout != cout || !v.raw) continue;
        const int lds = 2 * 36 * 16 * 8 * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        ConvPlan bp{};
        bp.cin = cin; bp.cout = cout; bp.pool = 0;
        bp.H = H; bp.W = W; bp.OH = H; bp.OW = W;
        bp.TH = 4; bp.TW = 64; bp.NI = 16;
        bp.tiles_y = (H + 3) / 4; bp.tiles_x = (W + 3) / 4;
        bp.threads = 64 * v.waves;
        bp.lds_bytes = lds;
        bp.blocks_per_cu = 1;
        bp.cost = (double)bp.tiles_y * bp.tiles_x / 16.0 * ((cout + 15) / 16) * (36.0 * (cin / 4) * 32.0 + 3000.0) * 0.4;
        bp.variant = 4000 + vi;
        bp.symbol = v.symbol;
        out->push_back(bp);
    }
}
bool conv_wino4_is_raw(const ConvPlan &p) { return p.variant >= 4000 && g_wino4[p.variant - 4000].raw != 0; }
int conv_wino4_stats_rows_max(int num_cus) { return num_cus; }

// stats / stats_rows: RAW builds only (see launch_conv_wino)
hipError_t launch_conv_wino4(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk, const float *bnp,
                             float *out, int N, int num_cus, double *stats, int *stats_rows, bool stats_clean,
                             const BnBwdFuse *bf) {
    const Wino4Variant &v = g_wino4[p.variant - 4000];
    Wino4Args a;
    a.stats = nullptr;
    a.bf = BnBwdFuse{nullptr, nullptr, nullptr};
    a.in = in; a.wpk = wpk; a.bnp = bnp; a.out = out;
    a.N = N; a.H = p.H; a.W = p.W; a.OH = p.OH; a.OW = p.OW;
    a.ty_img = p.tiles_y; a.tx_img = p.tiles_x;
    a.coutp = (p.cout + 15) / 16 * 16;
    a.tiles = N * a.ty_img * a.tx_img;
    a.total = (a.tiles + 15) / 16;
    if (a.total == 0) return hipSuccess;
    if (v.spec) {
        if ((double)N * p.H * p.W * p.cin * 4.0 >= 4294967296.0) return hipErrorInvalidValue;   // 32-bit byte offsets
        // one workgroup (producer + a consumer per n-tile) per CU, an M-tile at a time each; a multiple of 8 workgroups
        // for the XCD-aware walk
        const int slices = v.slice ? p.cout / v.slice : 1;
        int grid = std::min(a.total, std::max(1, num_cus * std::max(1, p.blocks_per_cu) / slices));
        if (grid >= 8) grid &= ~7;
        if (v.raw && stats) {                                 // (workgroups without M-tiles write zeros into their row: no memset)
            a.stats = stats;
            if (bf) a.bf = *bf;
            (void)stats_clean;
            if (stats_rows) *stats_rows = grid;
        }
        hipLaunchKernelGGL(v.kernel, dim3(grid, slices), dim3(p.threads), p.lds_bytes, s, a);
        return hipGetLastError();
    }
    const int ntiles = a.coutp / 16;
    const int waves = p.threads / 64;
    // persistent: the chip's workgroup slots are shared by the n-tile groups
    const int slots = std::max(1, num_cus * std::max(1, p.blocks_per_cu) / ntiles);
    const int grid = std::min((a.total + waves - 1) / waves, slots);
    hipLaunchKernelGGL(v.kernel, dim3(grid, ntiles), dim3(p.threads), p.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace asr
