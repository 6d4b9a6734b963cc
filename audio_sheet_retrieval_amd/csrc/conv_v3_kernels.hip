// conv3x3_mfma_v3: "lean" implicit-GEMM conv block for the small-K blocks (C_in <= 24) on gfx950
// (conv_bn + optional MaxPool2D of models/mutopia_ccal_cont.py:54-58,76-91).
//
// Why a third schedule: v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD,
// MI355X_MICROARCH.md "Matrix cores") and, measured here, ordinary VALU instructions of the same SIMD do not hide
// under it - every VALU instruction in the M-tile loop costs its 4 (v_mul_lo_u32, v_mad_u64: 16) cycles on top of
// the 32 cycles per MFMA.  conv3x3_mfma_kernel spends ~75 VALU instructions per 27-MFMA M-tile on index
// arithmetic (window -> image/row/column decode, 64-bit addresses, bounds); all of it is tile-independent.
// Here every per-lane index (A-fragment LDS offsets, output offsets, window coordinates, staging offsets) is
// computed ONCE per kernel for the <= PMAX passes a wave makes over a tile; a tile only adds wave-uniform
// (scalar) bases.  Per M-tile the VALU work left is the epilogue arithmetic itself.
//
// Same math, M-tile layout (4 pooling windows x 4 pixels), LDS layout and weight fragment order as
// conv3x3_mfma_kernel (conv_kernels.hip); weights live in VGPRs, all waves split the M-tiles.
#include "asr_kernels.h"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>

namespace asr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float elu_fast3(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }
__device__ __forceinline__ int fdiv3(int n, float rcp) { return (int)(((float)n + 0.5f) * rcp); }

__host__ __device__ constexpr int lds_pixel_stride3(int cin) {
    return cin == 12 ? 20 : cin == 24 ? 28 : cin == 48 ? 56 : cin == 96 ? 112 : cin + 4;
}

struct ConvArgs3 {
    const float *in;
    const float *wpk;       // fragment order [nt][tap][j][lane]
    const float *bnp;
    float *out;
    int N, H, W, OH, OW;
    int TH, TW, NI;
    int tiles_y, tiles_x, total_tiles;
    // FUSE1: block 1 (prepare + 3x3 stencil + BN + ELU, C_in = 1) is evaluated while staging the tile of block 2
    const void *raw;       // (N,Hraw,Wraw) uint8 / float32, or prepared float32 (N,H,W)
    const float *w1;       // [CIN][9] correlation-form taps of block 1
    const float *bn1;      // [3][CIN padded to 16]: mean | gamma * inv_std | beta
    int in_mode, rsz, Hraw, Wraw;
    int PP;                // tile pixels rounded up to a multiple of 64 (FUSE1 element order: channel group major)
};

// prepared input pixel (model.prepare: / 255, rsz: 2x2 mean) of the network-resolution image; 0 outside
__device__ __forceinline__ float prepared3(const void *in, int mode, size_t img_off, int Wraw, int y, int x, int H, int W,
                                           int rsz) {
    if (y < 0 || y >= H || x < 0 || x >= W) return 0.0f;
    if (mode == 0) return ((const float *)in)[img_off + (size_t)y * W + x];
    auto rawv = [&](int yy, int xx) -> float {
        if (mode == 2) return (float)((const unsigned char *)in)[img_off + (size_t)yy * Wraw + xx];
        return ((const float *)in)[img_off + (size_t)yy * Wraw + xx];
    };
    if (!rsz) return rawv(y, x) / 255.0f;
    const float a = rawv(2 * y, 2 * x) / 255.0f, b = rawv(2 * y, 2 * x + 1) / 255.0f;
    const float c = rawv(2 * y + 1, 2 * x) / 255.0f, d = rawv(2 * y + 1, 2 * x + 1) / 255.0f;
    const float top = a * 0.5f + b * 0.5f, bot = c * 0.5f + d * 0.5f;
    return top * 0.5f + bot * 0.5f;
}

template <int KS>
__device__ __forceinline__ void load_frag3(const float *p, float (&af)[KS]) {
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

// WAVES: waves per workgroup (all split M); MTW: M-tiles in flight per wave; PMAX: passes per tile the planner
// guarantees not to exceed; RMAX: staged float4 per thread; MINW: waves per SIMD the register budget allows.
// RAW: store the plain convolution (no BN / ELU / pool): train-mode forward and the data-gradient convolution.
template <int CIN, int COUT, bool POOL, int WAVES, int MTW, int PMAX, int RMAX, int MINW, bool FUSE1 = false,
          bool RAW = false>
__global__ __launch_bounds__(64 * WAVES, MINW) void conv3x3_mfma_v3(ConvArgs3 a) {
    static_assert(!(RAW && (POOL || FUSE1)), "RAW stores the un-pooled convolution of a materialised input");
    constexpr int KS = CIN / 4;
    constexpr int NT = (COUT + 15) / 16;
    constexpr int CS = lds_pixel_stride3(CIN);
    constexpr int C4 = CIN / 4;
    constexpr int THREADS = 64 * WAVES;
    constexpr int COUTP = NT * 16;

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;
    const int nn = lane & 15;

    float wreg[NT][9][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < KS; ++j) wreg[nt][tap][j] = a.wpk[((size_t)(nt * 9 + tap) * KS + j) * 64 + lane];
    float bmean[NT], bscale[NT], bbeta[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + nn;
        bmean[nt] = RAW ? 0.f : a.bnp[co];
        bscale[nt] = RAW ? 1.f : a.bnp[COUTP + co];
        bbeta[nt] = RAW ? 0.f : a.bnp[2 * COUTP + co];
    }

    const int LW = a.TW + 2, LH = a.TH + 2;
    const int WX = a.TW >> 1, WY = a.TH >> 1;
    const int win_per_img = WX * WY;
    const int nwin = win_per_img * a.NI;
    const int n_mt = (nwin + 3) >> 2;
    const int img_lds = LH * LW * CS;
    const int nvec = a.NI * LH * LW * C4;
    const int row_lds = LW * CS;                     // LDS floats between tile rows
    const float rcp_LW = 1.0f / (float)LW, rcp_LH = 1.0f / (float)LH;
    const float rcp_WX = 1.0f / (float)WX, rcp_win = 1.0f / (float)win_per_img;
    // passes this wave makes over a tile (wave-uniform)
    const int np_wave = (n_mt - wave * MTW + WAVES * MTW - 1) / (WAVES * MTW);

    // ---- tile-independent staging table of this thread
    int st_lds[RMAX], st_goff[RMAX], st_meta[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int e = tid + r * THREADS;
        st_lds[r] = -1; st_goff[r] = 0; st_meta[r] = 0;
        if (!FUSE1 && e < nvec) {
            const int p = e / C4;
            const int c4 = e - p * C4;
            const int q = fdiv3(p, rcp_LW);
            const int col = p - q * LW;
            const int img = fdiv3(q, rcp_LH);
            const int row = q - img * LH;
            st_lds[r] = p * CS + c4 * 4;
            st_goff[r] = ((img * a.H + row) * a.W + col) * CIN + c4 * 4;
            st_meta[r] = row | (col << 8) | (img << 16);
        }
    }
    // ---- FUSE1 tables.  Phase A copies the prepared raw patch (tile + 2-pixel halo, zeros outside the image) into
    // LDS behind the tile; phase B evaluates block 1 for every tile pixel from it (thread = pixel, all channels).
    constexpr int RA = FUSE1 ? 4 : 1, RB = FUSE1 ? 3 : 1;
    const int RW = LW + 2, RH = LH + 2;
    float *rawbuf = lds + (size_t)a.NI * img_lds;
    int fa_idx[RA], fa_meta[RA];       // raw-patch element: LDS index, row | col << 8 | img << 16 (patch-local)
    int fb_lds[RB], fb_raw[RB], fb_meta[RB];   // tile pixel: LDS float offset, top-left tap in rawbuf, row|col|img
    if constexpr (FUSE1) {
        const float rcp_RW = 1.0f / (float)RW, rcp_RH = 1.0f / (float)RH;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = tid + r * THREADS;
            fa_idx[r] = -1; fa_meta[r] = 0;
            if (e < a.NI * RH * RW) {
                const int q = fdiv3(e, rcp_RW);
                const int col = e - q * RW;
                const int img = fdiv3(q, rcp_RH);
                const int row = q - img * RH;
                fa_idx[r] = e;
                fa_meta[r] = row | (col << 8) | (img << 16);
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int p = tid + r * THREADS;
            fb_lds[r] = -1; fb_raw[r] = 0; fb_meta[r] = 0;
            if (p < a.NI * LH * LW) {
                const int q = fdiv3(p, rcp_LW);
                const int col = p - q * LW;
                const int img = fdiv3(q, rcp_LH);
                const int row = q - img * LH;
                fb_lds[r] = p * CS;
                fb_raw[r] = (img * RH + row) * RW + col;
                fb_meta[r] = row | (col << 8) | (img << 16);
            }
        }
    }
    // ---- tile-independent M-tile tables: A-fragment LDS offsets, output offsets, window coordinates
    int abase[PMAX][MTW], eoff[PMAX][MTW], ewy[PMAX][MTW], ewx[PMAX][MTW], eimg[PMAX][MTW];
#pragma unroll
    for (int k = 0; k < PMAX; ++k)
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
            const int mt = (k * WAVES + wave) * MTW + i;
            {
                int wdx = mt * 4 + (nn >> 2);                 // A row nn: window nn>>2, pixel nn&3
                wdx = wdx < nwin ? wdx : nwin - 1;
                const int img = fdiv3(wdx, rcp_win);
                const int rem = wdx - img * win_per_img;
                const int wy = fdiv3(rem, rcp_WX);
                const int wx = rem - wy * WX;
                const int py = 2 * wy + ((nn & 3) >> 1), px = 2 * wx + (nn & 1);
                abase[k][i] = img * img_lds + (py * LW + px) * CS + g * KS;
            }
            {
                const int wdx = mt * 4 + g;                   // C/D rows 4g..4g+3 = window g
                const bool valid = wdx < nwin;
                const int wc = valid ? wdx : 0;
                const int img = fdiv3(wc, rcp_win);
                const int rem = wc - img * win_per_img;
                const int wy = fdiv3(rem, rcp_WX);
                const int wx = rem - wy * WX;
                // a window past the tile gets coordinates that fail every bounds test
                ewy[k][i] = valid ? wy : (1 << 20);
                ewx[k][i] = wx;
                eimg[k][i] = img;
                eoff[k][i] = POOL ? ((img * a.OH + wy) * a.OW + wx) * COUT + nn
                                  : ((img * a.H + 2 * wy) * a.W + 2 * wx) * COUT + nn;
            }
        }

    // XCD-aware tile order: workgroups with the same blockIdx % 8 share an XCD (and its L2); each such group walks
    // one contiguous eighth of the tile list, so the tiles that re-read each other's halo meet in the same L2
    const int nx = gridDim.x >= 8 ? 8 : 1;
    const int xg = blockIdx.x % nx, xslot = blockIdx.x / nx;
    const int xper = (a.total_tiles + nx - 1) / nx;
    const int xslots = (gridDim.x - xg + nx - 1) / nx;
    for (int tk = xslot; tk < xper; tk += xslots) {
        const int tile = xg * xper + tk;
        if (tile >= a.total_tiles) break;
        const int t2 = (a.tiles_x == 1) ? tile : tile / a.tiles_x;
        const int tx = tile - t2 * a.tiles_x;
        const int grp = (a.tiles_y == 1) ? t2 : t2 / a.tiles_y;
        const int ty = t2 - grp * a.tiles_y;
        const int y0 = ty * a.TH, x0 = tx * a.TW, n0 = grp * a.NI;

        if constexpr (FUSE1) {
            // ---- block 1 on the fly: the (N,H,W,nf) activation - the largest of the network - never exists in HBM
            constexpr int P1 = (CIN + 15) / 16 * 16;
            const int nlim = a.N - n0;
            // phase A: prepared input patch, 2-pixel halo, zeros outside the image (block 1's zero padding)
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                if (fa_idx[r] < 0) continue;
                const int m = fa_meta[r];
                const int row = m & 255, col = (m >> 8) & 255, img = m >> 16;
                const int n = n0 + img;
                float v = 0.0f;
                if (img < nlim) {
                    const size_t img_off = (a.in_mode == 0) ? (size_t)n * a.H * a.W : (size_t)n * a.Hraw * a.Wraw;
                    v = prepared3(a.raw, a.in_mode, img_off, a.Wraw, y0 + row - 2, x0 + col - 2, a.H, a.W, a.rsz);
                }
                rawbuf[fa_idx[r]] = v;
            }
            __syncthreads();
            // phase B: every tile pixel inside the image; zeros outside (block 2's zero padding)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (fb_lds[r] < 0) continue;
                const int m = fb_meta[r];
                const int row = m & 255, col = (m >> 8) & 255, img = m >> 16;
                const int gy = y0 + row - 1, gx = x0 + col - 1;
                const bool ok = (img < nlim) && (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
                const float *rp = rawbuf + fb_raw[r];
                float v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = rp[(t / 3) * RW + (t % 3)];
                float4 *dst = reinterpret_cast<float4 *>(lds + fb_lds[r]);
                // channel groups of 4: the taps + BN values of a group are wave-uniform scalar loads (rolled loop:
                // bounded SGPR pressure, like conv1_kernel)
#pragma unroll 1
                for (int cg = 0; cg < C4; ++cg) {
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int co = cg * 4 + c;
                        float acc1 = 0.0f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc1 = fmaf(v[t], a.w1[co * 9 + t], acc1);
                        o[c] = ok ? elu_fast3((acc1 - a.bn1[co]) * a.bn1[P1 + co] + a.bn1[2 * P1 + co]) : 0.0f;
                    }
                    dst[cg] = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        } else
        // ---- stage the input tile: all loads of this thread first, then the LDS writes
        {
            const float *gbase = a.in + ((int64_t)((int64_t)n0 * a.H + (y0 - 1)) * a.W + (x0 - 1)) * CIN;
            const int ylo = 1 - y0, yhi = a.H + 1 - y0;        // valid tile rows: ylo <= row < yhi
            const int xlo = 1 - x0, xhi = a.W + 1 - x0;
            const int nlim = a.N - n0;
            float4 stage[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int m = st_meta[r];
                const int row = m & 255, col = (m >> 8) & 255, img = m >> 16;
                const bool ok = (st_lds[r] >= 0) && (row >= ylo) && (row < yhi) && (col >= xlo) && (col < xhi) &&
                                (img < nlim);
                stage[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) stage[r] = *reinterpret_cast<const float4 *>(gbase + st_goff[r]);
            }
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if (st_lds[r] >= 0) *reinterpret_cast<float4 *>(lds + st_lds[r]) = stage[r];
        }
        __syncthreads();

        // wave-uniform output base and bounds of this tile
        float *obase = POOL ? a.out + ((int64_t)((int64_t)n0 * a.OH + (y0 >> 1)) * a.OW + (x0 >> 1)) * COUT
                            : a.out + ((int64_t)((int64_t)n0 * a.H + y0) * a.W + x0) * COUT;
        const int wy_lim = POOL ? a.OH - (y0 >> 1) : (a.H - y0 + 1) >> 1;   // windows with at least one row inside
        const int wx_lim = POOL ? a.OW - (x0 >> 1) : (a.W - x0 + 1) >> 1;
        const int n_lim = a.N - n0;

#pragma unroll
        for (int k = 0; k < PMAX; ++k) {
            if (k >= np_wave) break;
            floatx4 acc[MTW][NT];
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                float af[3][MTW][KS];
#pragma unroll
                for (int i = 0; i < MTW; ++i) {
                    const float *rowp = lds + abase[k][i] + dy * row_lds;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) load_frag3<KS>(rowp + dx * CS, af[dx][i]);
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int j = 0; j < KS; ++j)
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[dx][i][j], wreg[nt][dy * 3 + dx][j],
                                                                                  acc[i][nt], 0, 0, 0);
            }
            // ---- epilogue (see conv_kernels.hip for the pooled-max identity)
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const bool ok = (ewy[k][i] < wy_lim) && (ewx[k][i] < wx_lim) && (eimg[k][i] < n_lim);
                if (POOL) {
                    if (!ok) continue;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (nt * 16 + nn >= COUT) continue;
                        const floatx4 c4 = acc[i][nt];
                        const float hi = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
                        const float lo = fminf(fminf(c4[0], c4[1]), fminf(c4[2], c4[3]));
                        const float x = bscale[nt] >= 0.0f ? hi : lo;
                        obase[eoff[k][i] + nt * 16] = elu_fast3((x - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    }
                } else {
                    if (!ok) continue;
                    // rows / columns of the 2x2 window that are inside the image
                    const bool y1 = 2 * ewy[k][i] + 1 < a.H - y0, x1 = 2 * ewx[k][i] + 1 < a.W - x0;
                    const int rstride = a.W * COUT;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if (nt * 16 + nn >= COUT) continue;
                        float *o = obase + eoff[k][i] + nt * 16;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            v[r] = RAW ? acc[i][nt][r] : elu_fast3((acc[i][nt][r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        o[0] = v[0];
                        if (x1) o[COUT] = v[1];
                        if (y1) o[rstride] = v[2];
                        if (y1 && x1) o[rstride + COUT] = v[3];
                    }
                }
            }
        }
        __syncthreads();   // LDS is re-staged by the next tile
    }
}

// ---- instantiation table ----------------------------------------------------
struct ConvVariant3 {
    int cin, cout, pool, waves, mtw, pmax, rmax, fuse1, raw;
    void (*kernel)(ConvArgs3);
    const char *symbol;
};
#define ASR_BOOLSTR3_0 "false"
#define ASR_BOOLSTR3_1 "true"
#define ASR_CONV3(CIN, COUT, POOL, WAVES, MTW, PMAX, RMAX, MINW)                                                 \
    { CIN, COUT, POOL, WAVES, MTW, PMAX, RMAX, 0, 0,                                                             \
      conv3x3_mfma_v3<CIN, COUT, (POOL != 0), WAVES, MTW, PMAX, RMAX, MINW>,                                     \
      "void asr::conv3x3_mfma_v3<" #CIN ", " #COUT ", " ASR_BOOLSTR3_##POOL ", " #WAVES ", " #MTW ", " #PMAX    \
      ", " #RMAX ", " #MINW ", false, false>(asr::ConvArgs3)" }
#define ASR_CONV3R(CIN, COUT, WAVES, MTW, PMAX, RMAX, MINW)                                                      \
    { CIN, COUT, 0, WAVES, MTW, PMAX, RMAX, 0, 1,                                                                \
      conv3x3_mfma_v3<CIN, COUT, false, WAVES, MTW, PMAX, RMAX, MINW, false, true>,                              \
      "void asr::conv3x3_mfma_v3<" #CIN ", " #COUT ", false, " #WAVES ", " #MTW ", " #PMAX                      \
      ", " #RMAX ", " #MINW ", false, true>(asr::ConvArgs3)" }
#define ASR_CONV3F(CIN, COUT, POOL, WAVES, MTW, PMAX, RMAX, MINW)                                                \
    { CIN, COUT, POOL, WAVES, MTW, PMAX, RMAX, 1, 0,                                                             \
      conv3x3_mfma_v3<CIN, COUT, (POOL != 0), WAVES, MTW, PMAX, RMAX, MINW, true>,                               \
      "void asr::conv3x3_mfma_v3<" #CIN ", " #COUT ", " ASR_BOOLSTR3_##POOL ", " #WAVES ", " #MTW ", " #PMAX    \
      ", " #RMAX ", " #MINW ", true, false>(asr::ConvArgs3)" }
static const ConvVariant3 g_variants3[] = {
    ASR_CONV3(12, 12, 1, 4, 1, 6, 6, 4),
    ASR_CONV3(12, 12, 1, 4, 2, 3, 6, 3),
    ASR_CONV3(12, 12, 1, 8, 1, 4, 4, 4),
    ASR_CONV3(12, 24, 0, 4, 1, 6, 6, 3),
    ASR_CONV3(12, 24, 0, 4, 2, 3, 6, 2),
    ASR_CONV3(12, 24, 0, 8, 1, 4, 4, 3),
    ASR_CONV3(24, 24, 1, 4, 1, 6, 10, 2),
    ASR_CONV3(24, 24, 1, 4, 2, 3, 10, 2),
    ASR_CONV3(24, 24, 1, 8, 1, 4, 6, 2),
    // block 1 fused into block 2 (its activation never reaches HBM)
    ASR_CONV3F(12, 12, 1, 4, 1, 6, 6, 4),
    ASR_CONV3F(12, 12, 1, 4, 2, 3, 6, 3),
    ASR_CONV3F(12, 12, 1, 8, 1, 4, 4, 4),
    ASR_CONV3F(24, 24, 1, 4, 1, 6, 10, 2),
    ASR_CONV3F(24, 24, 1, 8, 1, 4, 6, 2),
    // RAW epilogue: train-mode forward convolutions and data gradients (C_in / C_out swapped) of the small-K blocks
    ASR_CONV3R(12, 12, 4, 2, 3, 6, 3),
    ASR_CONV3R(12, 24, 4, 2, 3, 6, 2),
    ASR_CONV3R(24, 12, 4, 2, 3, 10, 2),
    ASR_CONV3R(24, 24, 4, 1, 6, 10, 2),
};
static const int g_num_variants3 = (int)(sizeof(g_variants3) / sizeof(g_variants3[0]));

static void enumerate_v3(int vi, int H, int W, int lds_budget, std::vector<ConvPlan> &out) {
    const ConvVariant3 &v = g_variants3[vi];
    const int cin = v.cin, cout = v.cout;
    const int cs = lds_pixel_stride3(cin);
    const int threads = 64 * v.waves;
    const int nt = (cout + 15) / 16;
    const int slots = v.waves * v.mtw;
    const int vec_budget = v.rmax * threads;
    const int He = (H + 1) & ~1, We = (W + 1) & ~1;
    for (int TH = 2; TH <= std::min(He, 64); TH += 2) {
        for (int TW = 2; TW <= std::min(We, 128); TW += 2) {
            const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
            const int px = (TH + 2) * (TW + 2);
            const int per_img_lds = px * cs * 4;
            const int per_img_vec = v.fuse1 ? 0 : px * (cin / 4);
            const int raw_px = (TH + 4) * (TW + 4);
            if (v.fuse1 && (px > 3 * threads || raw_px > 4 * threads)) continue;
            if (per_img_lds + (v.fuse1 ? raw_px * 4 : 0) > lds_budget || per_img_vec > vec_budget) continue;
            if (TH + 2 > 255 || TW + 2 > 255) continue;
            int ni_max = 1;
            if (tiles_y == 1 && tiles_x == 1)
                ni_max = v.fuse1 ? 1 : std::max(1, std::min(16, std::min(lds_budget / per_img_lds,
                                                                          vec_budget / std::max(1, per_img_vec))));
            for (int NI = 1; NI <= ni_max; ++NI) {
                const int nwin = (TH / 2) * (TW / 2) * NI;
                const int n_mt = (nwin + 3) / 4;
                const int passes = (n_mt + slots - 1) / slots;
                if (passes > v.pmax) continue;
                const double mfma = (double)passes * v.mtw * 9.0 * (cin / 4) * nt * 32.0;
                const double stage = v.fuse1 ? (double)NI * px * cin * 18.0 / (64.0 * v.waves) * 4.0
                                             : (double)NI * px * cin * 4 / 24.0;
                ConvPlan bp{};
                bp.cost = (mfma + stage + 600.0) * tiles_y * tiles_x / NI;
                bp.TH = TH; bp.TW = TW; bp.NI = NI;
                bp.tiles_y = tiles_y; bp.tiles_x = tiles_x;
                bp.lds_bytes = per_img_lds * NI + (v.fuse1 ? raw_px * 4 * NI : 0);
                bp.tile_floats = per_img_lds * NI / 4;
                bp.cin = cin; bp.cout = cout; bp.pool = v.pool;
                bp.H = H; bp.W = W;
                bp.OH = v.pool ? H / 2 : H;
                bp.OW = v.pool ? W / 2 : W;
                bp.threads = threads;
                bp.variant = 2000 + vi;
                bp.symbol = v.symbol;
                bp.fuse1 = v.fuse1;
                out.push_back(bp);
            }
        }
    }
    std::sort(out.begin(), out.end(), [](const ConvPlan &x, const ConvPlan &y) { return x.cost < y.cost; });
}

static void finish_v3(ConvPlan &bp) {
    const ConvVariant3 &v = g_variants3[bp.variant - 2000];
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

void conv_candidates_v3(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out,
                        int fuse1) {
    static const int use_v3 = getenv("ASR_CONV_V3") ? atoi(getenv("ASR_CONV_V3")) : 1;
    if (!use_v3) return;
    for (int vi = 0; vi < g_num_variants3; ++vi) {
        const ConvVariant3 &v = g_variants3[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool || v.fuse1 != fuse1 || v.raw) continue;
        for (int budget : {30 * 1024, 50 * 1024, 76 * 1024}) {
            std::vector<ConvPlan> c;
            enumerate_v3(vi, H, W, budget, c);
            int taken = 0;
            for (auto &cand : c) {
                bool dup = false;
                for (auto &o : *out)
                    if (o.variant == cand.variant && o.TH == cand.TH && o.TW == cand.TW && o.NI == cand.NI) dup = true;
                if (dup) continue;
                finish_v3(cand);
                out->push_back(cand);
                if (++taken >= max_count) break;
            }
        }
    }
}

// model-chosen plan for the RAW (training) form: the cheapest tiling that allows >= 3 workgroups per CU
bool plan_conv_v3_raw(int cin, int cout, int H, int W, ConvPlan *plan) {
    static const int use_v3 = getenv("ASR_CONV_V3") ? atoi(getenv("ASR_CONV_V3")) : 1;
    if (!use_v3) return false;
    for (int vi = 0; vi < g_num_variants3; ++vi) {
        const ConvVariant3 &v = g_variants3[vi];
        if (!v.raw || v.cin != cin || v.cout != cout) continue;
        std::vector<ConvPlan> c;
        enumerate_v3(vi, H, W, 50 * 1024, c);
        if (c.empty()) continue;
        *plan = c[0];
        finish_v3(*plan);
        return true;
    }
    return false;
}

hipError_t launch_conv_v3(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk, const float *bnp,
                          float *out, int N, int num_cus, const Fuse1Args *f1) {
    const ConvVariant3 &v = g_variants3[p.variant - 2000];
    ConvArgs3 a;
    a.raw = nullptr; a.w1 = nullptr; a.bn1 = nullptr; a.in_mode = 0; a.rsz = 0; a.Hraw = 0; a.Wraw = 0; a.PP = 0;
    if (v.fuse1) {
        if (!f1) return hipErrorInvalidValue;
        a.raw = f1->raw; a.w1 = f1->w1; a.bn1 = f1->bn1; a.in_mode = f1->in_mode; a.rsz = f1->rsz;
        a.Hraw = f1->Hraw; a.Wraw = f1->Wraw;
        a.PP = (p.NI * (p.TH + 2) * (p.TW + 2) + 63) / 64 * 64;
    }
    a.in = in; a.wpk = wpk; a.bnp = bnp; a.out = out;
    a.N = N; a.H = p.H; a.W = p.W; a.OH = p.OH; a.OW = p.OW;
    a.TH = p.TH; a.TW = p.TW; a.NI = p.NI;
    a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
    const int groups = (N + p.NI - 1) / p.NI;
    a.total_tiles = groups * p.tiles_y * p.tiles_x;
    if (a.total_tiles == 0) return hipSuccess;
    const int grid = std::min(a.total_tiles, num_cus * std::max(1, p.blocks_per_cu));
    hipLaunchKernelGGL(v.kernel, dim3(grid), dim3(p.threads), p.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace asr
