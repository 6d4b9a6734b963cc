// conv3x3_mfma_v2: second-generation implicit-GEMM conv block for gfx950
// (conv_bn + optional MaxPool2D of models/mutopia_ccal_cont.py:54-58,76-91).
//
// Same math and M-tile layout as conv3x3_mfma_kernel (conv_kernels.hip); what
// changes is the schedule:
//   * double-buffered input tiles: the global loads of tile t+1 are issued into
//     registers BEFORE the MFMA loop of tile t and written to the other LDS
//     buffer after it - one barrier per tile, staging latency hidden behind
//     the matrix work of the same workgroup (no reliance on a second resident
//     workgroup);
//   * all waves of the workgroup split the M-tiles only (balanced 2 or 4 waves
//     per SIMD); the weight fragments either stay in VGPRs (small K) or are
//     read from an LDS copy in fragment order (48->48: 83 KB, conflict-free
//     ds_read_b128) which frees ~110 VGPRs per wave for occupancy.
#include "asr_kernels.h"
#include <algorithm>
#include <vector>
#include <cstdio>
#include <cstdlib>

namespace asr {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float elu_fast2(float v) { return v > 0.0f ? v : __expf(v) - 1.0f; }
__device__ __forceinline__ int fdiv2(int n, float rcp) { return (int)(((float)n + 0.5f) * rcp); }

__host__ __device__ constexpr int lds_pixel_stride2(int cin) {
    return cin == 12 ? 20 : cin == 24 ? 28 : cin == 48 ? 56 : cin == 96 ? 112 : cin + 4;
}

struct ConvArgs2 {
    const float *in;
    const float *wpk;       // fragment order [nt][tap][j][lane]  (same as v1)
    const float *bnp;
    float *out;
    int N, H, W, OH, OW;
    int TH, TW, NI;
    int tiles_y, tiles_x, total_tiles;
    int tile_floats;        // LDS floats of one tile buffer
};

template <int N>
__device__ __forceinline__ void lds_read_vec(const float *p, float (&v)[N]) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(p)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int q = 0; q < N / 2; ++q) {
            const float2 t = reinterpret_cast<const float2 *>(p)[q];
            v[2 * q] = t.x; v[2 * q + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) v[q] = p[q];
    }
}

// WAVES: waves per workgroup (all split M); MTW: M-tiles in flight per wave;
// WLDS: weights in LDS (true) or VGPRs (false); RMAX: staged float4 per thread.
// MINW: minimum waves per SIMD the register allocation must allow (4 -> <= 128 VGPRs, 2 -> <= 256).
// WN: wave columns splitting the C_out tiles (a workgroup is (WAVES / WN) x WN waves): with tiles of only a few
// M-tiles this keeps every wave busy - the A fragments are read WN times, each wave's B reads shrink to NT / WN.
template <int CIN, int COUT, bool POOL, int WAVES, int MTW, bool WLDS, int RMAX, int MINW = 4, int WN = 1>
__global__ __launch_bounds__(64 * WAVES, MINW) void conv3x3_mfma_v2(ConvArgs2 a) {
    constexpr int KS = CIN / 4;
    constexpr int NTALL = (COUT + 15) / 16;         // 16-wide C_out tiles of the layer
    constexpr int NT = NTALL / WN;                  // ... of one wave
    constexpr int WROWS = WAVES / WN;               // wave rows splitting the M-tiles
    static_assert(NTALL % WN == 0 && WAVES % WN == 0, "wave grid must divide the tiles");
    constexpr int CS = lds_pixel_stride2(CIN);
    constexpr int C4 = CIN / 4;                     // float4 per pixel
    constexpr int THREADS = 64 * WAVES;
    constexpr int COUTP = NTALL * 16;
    constexpr int JS = (KS % 4 == 0) ? 4 : ((KS % 2 == 0) ? 2 : KS);   // k-steps per register sub-block
    constexpr int NSB = KS / JS;
    constexpr int WFLOATS = WLDS ? NTALL * 9 * KS * 64 : 0;
    static_assert(KS % JS == 0, "");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;                                 // [nt][tap][lane][KS] when WLDS
    float *tile0 = lds + WFLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = (tid >> 6) / WN;              // wave row: which M-tiles
    const int nt0 = ((tid >> 6) % WN) * NT;        // first C_out tile of this wave
    const int g = lane >> 4;
    const int nn = lane & 15;

    // ---- weights: VGPR copy or LDS copy (lane-major so that a lane's KS values are contiguous)
    float wreg[WLDS ? 1 : NT][WLDS ? 1 : 9][WLDS ? 1 : KS];
    if constexpr (WLDS) {
        for (int e = tid; e < NTALL * 9 * KS * 64; e += THREADS) {
            const int l = e & 63;
            const int j = (e >> 6) % KS;
            const int nt_tap = (e >> 6) / KS;
            wl[((size_t)nt_tap * 64 + l) * KS + j] = a.wpk[e];
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < KS; ++j)
                    wreg[nt][tap][j] = a.wpk[((size_t)((nt0 + nt) * 9 + tap) * KS + j) * 64 + lane];
    }
    float bmean[NT], bscale[NT], bbeta[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = (nt0 + nt) * 16 + nn;
        bmean[nt] = a.bnp[co];
        bscale[nt] = a.bnp[COUTP + co];
        bbeta[nt] = a.bnp[2 * COUTP + co];
    }

    const int LW = a.TW + 2, LH = a.TH + 2;
    const int WX = a.TW >> 1, WY = a.TH >> 1;
    const int win_per_img = WX * WY;
    const int nwin = win_per_img * a.NI;
    const int n_mt = (nwin + 3) >> 2;
    const int img_lds = LH * LW * CS;
    const int nvec = a.NI * LH * LW * C4;            // float4 elements of one tile
    const float rcp_LW = 1.0f / (float)LW, rcp_LH = 1.0f / (float)LH;
    const float rcp_WX = 1.0f / (float)WX, rcp_win = 1.0f / (float)win_per_img;

    // tile-independent part of the staging addresses of this thread
    int st_lds[RMAX];          // LDS float offset inside a tile buffer, -1: nothing to stage
    int st_meta[RMAX];         // row | col << 8 | img << 16 | c4 << 24 (tile-local, halo included)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        const int e = tid + r * THREADS;
        st_lds[r] = -1;
        st_meta[r] = 0;
        if (e < nvec) {
            const int p = e / C4;
            const int c4 = e - p * C4;
            const int q = fdiv2(p, rcp_LW);
            const int col = p - q * LW;
            const int img = fdiv2(q, rcp_LH);
            const int row = q - img * LH;
            st_lds[r] = p * CS + c4 * 4;
            st_meta[r] = row | (col << 8) | (img << 16) | (c4 << 24);
        }
    }

    float4 stage[RMAX];
    auto tile_origin = [&](int tile, int &y0, int &x0, int &n0) {
        const int t2 = (a.tiles_x == 1) ? tile : tile / a.tiles_x;
        const int tx = tile - t2 * a.tiles_x;
        const int grp = (a.tiles_y == 1) ? t2 : t2 / a.tiles_y;
        const int ty = t2 - grp * a.tiles_y;
        y0 = ty * a.TH; x0 = tx * a.TW; n0 = grp * a.NI;
    };
    auto issue_loads = [&](int tile) {
        int y0, x0, n0;
        tile_origin(tile, y0, x0, n0);
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int m = st_meta[r];
            const int gy = y0 + (m & 255) - 1, gx = x0 + ((m >> 8) & 255) - 1, n = n0 + ((m >> 16) & 255);
            const bool ok = (st_lds[r] >= 0) && (n < a.N) && (gy >= 0) && (gy < a.H) && (gx >= 0) && (gx < a.W);
            stage[r] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok)
                stage[r] = *reinterpret_cast<const float4 *>(a.in + (((size_t)n * a.H + gy) * a.W + gx) * CIN +
                                                              ((m >> 24) & 255) * 4);
        }
    };
    auto write_stage = [&](float *buf) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
            if (st_lds[r] >= 0) *reinterpret_cast<float4 *>(buf + st_lds[r]) = stage[r];
    };

    // XCD-aware tile order (see conv_v3_kernels.hip): the workgroups of one XCD walk one contiguous eighth of the tiles
    const int nx = gridDim.x >= 8 ? 8 : 1;
    const int xg = blockIdx.x % nx, xslot = blockIdx.x / nx;
    const int xper = (a.total_tiles + nx - 1) / nx;
    const int xslots = (gridDim.x - xg + nx - 1) / nx;
    auto tile_at = [&](int k) {
        const int tk = xslot + k * xslots;
        const int t = xg * xper + tk;
        return (tk < xper && t < a.total_tiles) ? t : a.total_tiles;
    };
    int tile = tile_at(0);
    if (tile < a.total_tiles) {
        issue_loads(tile);
        write_stage(tile0);
    }
    __syncthreads();

    for (int it = 0; tile < a.total_tiles; ++it) {
        const float *buf = tile0 + (size_t)(it & 1) * a.tile_floats;
        float *nbuf = tile0 + (size_t)((it + 1) & 1) * a.tile_floats;
        const int next = tile_at(it + 1);
        if (next < a.total_tiles) issue_loads(next);          // in flight during the MFMA loop below

        int y0, x0, n0;
        tile_origin(tile, y0, x0, n0);

        for (int mt0 = wave * MTW; mt0 < n_mt; mt0 += WROWS * MTW) {
            // the LDS weight reads are loop-invariant; without this opaque offset LICM hoists all
            // NT*9*KS of them out of the loop and spills
            int woff = 0;
            asm volatile("" : "+v"(woff));
            floatx4 acc[MTW][NT];
            int abase[MTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[i][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
                int wdx = (mt0 + i) * 4 + (nn >> 2);
                wdx = wdx < nwin ? wdx : nwin - 1;
                const int img = fdiv2(wdx, rcp_win);
                const int rem = wdx - img * win_per_img;
                const int wy = fdiv2(rem, rcp_WX);
                const int wx = rem - wy * WX;
                const int py = 2 * wy + ((nn & 3) >> 1), px = 2 * wx + (nn & 1);
                abase[i] = img * img_lds + (py * LW + px) * CS + g * KS;
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int toff = ((tap / 3) * LW + (tap % 3)) * CS;
#pragma unroll
                for (int sb = 0; sb < NSB; ++sb) {
                    float af[MTW][JS];
                    float bf[NT][JS];
#pragma unroll
                    for (int i = 0; i < MTW; ++i) lds_read_vec<JS>(buf + abase[i] + toff + sb * JS, af[i]);
                    if constexpr (WLDS) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            lds_read_vec<JS>(wl + woff + (((nt0 + nt) * 9 + tap) * 64 + lane) * KS + sb * JS, bf[nt]);
                    } else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int j = 0; j < JS; ++j) bf[nt][j] = wreg[nt][tap][sb * JS + j];
                    }
#pragma unroll
                    for (int j = 0; j < JS; ++j)
#pragma unroll
                        for (int i = 0; i < MTW; ++i)
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][j], bf[nt][j], acc[i][nt], 0, 0, 0);
                }
            }
            // ---- epilogue (see conv_kernels.hip for the pooled-max identity)
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int wdx = (mt0 + i) * 4 + g;
                if (wdx >= nwin) continue;
                const int img = fdiv2(wdx, rcp_win);
                const int rem = wdx - img * win_per_img;
                const int wy = fdiv2(rem, rcp_WX);
                const int wx = rem - wy * WX;
                const int n = n0 + img;
                if (n >= a.N) continue;
                if (POOL) {
                    const int oy = (y0 >> 1) + wy, ox = (x0 >> 1) + wx;
                    if (oy >= a.OH || ox >= a.OW) continue;
                    float *orow = a.out + (((size_t)n * a.OH + oy) * a.OW + ox) * COUT;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = (nt0 + nt) * 16 + nn;
                        if (co >= COUT) continue;
                        const floatx4 c4 = acc[i][nt];
                        const float hi = fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3]));
                        const float lo = fminf(fminf(c4[0], c4[1]), fminf(c4[2], c4[3]));
                        const float x = bscale[nt] >= 0.0f ? hi : lo;
                        orow[co] = elu_fast2((x - bmean[nt]) * bscale[nt] + bbeta[nt]);
                    }
                } else {
                    const int yb = y0 + 2 * wy, xb = x0 + 2 * wx;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int co = (nt0 + nt) * 16 + nn;
                        if (co >= COUT) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int y = yb + (r >> 1), x = xb + (r & 1);
                            if (y < a.H && x < a.W)
                                a.out[(((size_t)n * a.H + y) * a.W + x) * COUT + co] =
                                    elu_fast2((acc[i][nt][r] - bmean[nt]) * bscale[nt] + bbeta[nt]);
                        }
                    }
                }
            }
        }
        if (next < a.total_tiles) write_stage(nbuf);
        __syncthreads();
        tile = next;
    }
}

// ---- instantiation table ----------------------------------------------------
struct ConvVariant2 {
    int cin, cout, pool, waves, mtw, wlds, rmax, wn;
    void (*kernel)(ConvArgs2);
    const char *symbol;        // as rocprofv3 prints it
};
#define ASR_BOOLSTR2_0 "false"
#define ASR_BOOLSTR2_1 "true"
#define ASR_CONV2(CIN, COUT, POOL, WAVES, MTW, WLDS, RMAX)                                                       \
    { CIN, COUT, POOL, WAVES, MTW, WLDS, RMAX, 1,                                                                \
      conv3x3_mfma_v2<CIN, COUT, (POOL != 0), WAVES, MTW, (WLDS != 0), RMAX>,                                    \
      "void asr::conv3x3_mfma_v2<" #CIN ", " #COUT ", " ASR_BOOLSTR2_##POOL ", " #WAVES ", " #MTW ", "          \
      ASR_BOOLSTR2_##WLDS ", " #RMAX ", 4, 1>(asr::ConvArgs2)" }
#define ASR_CONV2N(CIN, COUT, POOL, WAVES, MTW, WLDS, RMAX, MINW, WN)                                            \
    { CIN, COUT, POOL, WAVES, MTW, WLDS, RMAX, WN,                                                               \
      conv3x3_mfma_v2<CIN, COUT, (POOL != 0), WAVES, MTW, (WLDS != 0), RMAX, MINW, WN>,                          \
      "void asr::conv3x3_mfma_v2<" #CIN ", " #COUT ", " ASR_BOOLSTR2_##POOL ", " #WAVES ", " #MTW ", "          \
      ASR_BOOLSTR2_##WLDS ", " #RMAX ", " #MINW ", " #WN ">(asr::ConvArgs2)" }
// Measured on MI355X (chunk 250, 160x200 tower): v2 beats the v1 schedule on the 48-channel blocks
// (conv6 0.272 -> 0.199 ms, conv7/8 0.084 -> 0.064 ms) and loses on the small-K blocks (conv2 0.318 -> 0.380,
// conv4 0.244 -> 0.318: one pass per tile leaves the per-tile barrier + staging exposed), so only the
// former are routed here; ASR_CONV_V2_ALL=1 enables every variant for experiments.
#define ASR_CONV2W(CIN, COUT, POOL, WAVES, MTW, WLDS, RMAX, MINW)                                                \
    { CIN, COUT, POOL, WAVES, MTW, WLDS, RMAX, 1,                                                                \
      conv3x3_mfma_v2<CIN, COUT, (POOL != 0), WAVES, MTW, (WLDS != 0), RMAX, MINW>,                              \
      "void asr::conv3x3_mfma_v2<" #CIN ", " #COUT ", " ASR_BOOLSTR2_##POOL ", " #WAVES ", " #MTW ", "          \
      ASR_BOOLSTR2_##WLDS ", " #RMAX ", " #MINW ", 1>(asr::ConvArgs2)" }
static const ConvVariant2 g_variants2[] = {
    // 4-wave workgroups with the weights in VGPRs: several double-buffered workgroups per CU keep 2-3 tiles of
    // global loads in flight under the MFMA loops (the small-K blocks sit near both roofs)
    ASR_CONV2W(12, 12, 1, 4, 4, 0, 4, 2),
    ASR_CONV2W(12, 12, 1, 4, 2, 0, 4, 3),
    ASR_CONV2W(12, 24, 0, 4, 2, 0, 4, 2),
    ASR_CONV2W(24, 24, 1, 4, 2, 0, 6, 2),
    ASR_CONV2W(24, 24, 1, 4, 1, 0, 6, 2),
    ASR_CONV2(12, 12, 1, 8, 4, 0, 4),
    ASR_CONV2(12, 24, 0, 8, 2, 1, 4),
    ASR_CONV2(24, 24, 1, 8, 2, 1, 4),
    ASR_CONV2(24, 48, 0, 8, 2, 1, 3),
    ASR_CONV2(48, 48, 1, 16, 2, 1, 3),
    ASR_CONV2(48, 48, 0, 16, 2, 1, 3),
    // wave grid (rows x 3 columns of C_out tiles): every wave has MFMA work on the small double-buffered tiles
    ASR_CONV2N(24, 48, 0, 12, 1, 1, 3, 4, 3),
    ASR_CONV2N(24, 48, 0, 12, 2, 1, 3, 4, 3),
    ASR_CONV2N(48, 48, 1, 12, 1, 1, 4, 4, 3),
    ASR_CONV2N(48, 48, 1, 12, 2, 1, 4, 4, 3),
    ASR_CONV2N(48, 48, 0, 12, 1, 1, 4, 4, 3),
    ASR_CONV2N(48, 48, 0, 12, 2, 1, 4, 4, 3),
    ASR_CONV2N(48, 48, 1, 6, 2, 1, 8, 4, 3),
    ASR_CONV2N(48, 48, 0, 6, 2, 1, 8, 4, 3),
};
static const int g_num_variants2 = (int)(sizeof(g_variants2) / sizeof(g_variants2[0]));

static int find_v2(int cin, int cout, int pool) {
    for (int i = 0; i < g_num_variants2; ++i)
        if (g_variants2[i].cin == cin && g_variants2[i].cout == cout && g_variants2[i].pool == pool) return i;
    return -1;
}

static void enumerate_v2(int vi, int H, int W, int target_blocks, std::vector<ConvPlan> &out) {
    const ConvVariant2 &v = g_variants2[vi];
    const int cin = v.cin, cout = v.cout;
    const int cs = lds_pixel_stride2(cin);
    const int threads = 64 * v.waves;
    const int nt = (cout + 15) / 16;
    const int wbytes = v.wlds ? nt * 9 * (cin / 4) * 64 * 4 : 0;
    const int lds_total = (160 * 1024) / target_blocks - 1024;
    const int tile_budget = (lds_total - wbytes) / 2;                  // two tile buffers
    const int vec_budget = v.rmax * threads;                           // float4 a workgroup can stage
    const int slots = (v.waves / v.wn) * v.mtw;
    const int He = (H + 1) & ~1, We = (W + 1) & ~1;
    if (tile_budget <= 0) return;
    for (int TH = 2; TH <= std::min(He, 96); TH += 2) {
        for (int TW = 2; TW <= std::min(We, 128); TW += 2) {
            const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + TW - 1) / TW;
            const int px = (TH + 2) * (TW + 2);
            const int per_img_lds = px * cs * 4;
            const int per_img_vec = px * (cin / 4);
            if (per_img_lds > tile_budget || per_img_vec > vec_budget) continue;
            int ni_max = 1;
            if (tiles_y == 1 && tiles_x == 1)
                ni_max = std::max(1, std::min(32, std::min(tile_budget / per_img_lds, vec_budget / per_img_vec)));
            for (int NI = 1; NI <= ni_max; ++NI) {
                const int nwin = (TH / 2) * (TW / 2) * NI;
                const int n_mt = (nwin + 3) / 4;
                const int passes = (n_mt + slots - 1) / slots;
                // per-wave serial MFMA issue time of a tile (cycles) + exposed per-tile overhead
                const double mfma = (double)passes * v.mtw * 9.0 * (cin / 4) * (nt / v.wn) * 32.0;
                ConvPlan bp{};
                bp.cost = (mfma + 1500.0) * tiles_y * tiles_x / NI;
                bp.TH = TH; bp.TW = TW; bp.NI = NI;
                bp.tiles_y = tiles_y; bp.tiles_x = tiles_x;
                bp.lds_bytes = wbytes + 2 * per_img_lds * NI;
                bp.tile_floats = per_img_lds * NI / 4;
                bp.cin = cin; bp.cout = cout; bp.pool = v.pool;
                bp.H = H; bp.W = W;
                bp.OH = v.pool ? H / 2 : H;
                bp.OW = v.pool ? W / 2 : W;
                bp.threads = threads;
                bp.variant = 1000 + vi;
                bp.symbol = v.symbol;
                out.push_back(bp);
            }
        }
    }
    std::sort(out.begin(), out.end(), [](const ConvPlan &x, const ConvPlan &y) { return x.cost < y.cost; });
}

static void finish_v2(ConvPlan &bp, int target_blocks) {
    const ConvVariant2 &v = g_variants2[bp.variant - 1000];
    // per-function attribute shared by both towers' plans: allow the full 160 KiB
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(v.kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(v.kernel), bp.threads,
                                                     (size_t)bp.lds_bytes) != hipSuccess || nb < 1) {
        (void)hipGetLastError();
        nb = target_blocks;
    }
    bp.blocks_per_cu = std::min(nb, 4);
}

bool plan_conv_v2(int cin, int cout, int pool, int H, int W, ConvPlan *plan) {
    if (getenv("ASR_CONV_V1")) return false;
    const int vi = find_v2(cin, cout, pool);
    if (vi < 0) return false;
    if (cout < 48 && !getenv("ASR_CONV_V2_ALL")) return false;
    const int target_blocks = g_variants2[vi].waves >= 16 ? 1 : 2;     // one workgroup per CU for 16 waves
    std::vector<ConvPlan> c;
    enumerate_v2(vi, H, W, target_blocks, c);
    if (c.empty()) return false;
    ConvPlan bp = c[0];
    finish_v2(bp, target_blocks);
    if (getenv("ASR_DEBUG"))
        fprintf(stderr, "[asr] plan v2 conv %d->%d pool=%d %dx%d: tile %dx%d x%d img, tiles %dx%d, lds %d B, %d thr, "
                        "%d blocks/CU\n", cin, cout, pool, H, W, bp.TH, bp.TW, bp.NI, bp.tiles_y, bp.tiles_x,
                bp.lds_bytes, bp.threads, bp.blocks_per_cu);
    *plan = bp;
    return true;
}

void conv_candidates_v2(int cin, int cout, int pool, int H, int W, int max_count, std::vector<ConvPlan> *out) {
    if (getenv("ASR_CONV_V1")) return;
    for (int vi = 0; vi < g_num_variants2; ++vi) {
        const ConvVariant2 &v = g_variants2[vi];
        if (v.cin != cin || v.cout != cout || v.pool != pool) continue;
        for (int target_blocks : {1, 2, 3, 4}) {
            if (v.waves >= 16 && target_blocks >= 2) continue;
            if (v.waves >= 8 && target_blocks >= 3) continue;
            if (v.waves <= 4 && target_blocks == 1) continue;
            std::vector<ConvPlan> c;
            enumerate_v2(vi, H, W, target_blocks, c);
            int taken = 0;
            for (auto &cand : c) {
                bool dup = false;
                for (auto &o : *out)
                    if (o.variant == cand.variant && o.TH == cand.TH && o.TW == cand.TW && o.NI == cand.NI) dup = true;
                if (dup) continue;
                finish_v2(cand, target_blocks);
                out->push_back(cand);
                if (++taken >= max_count) break;
            }
        }
    }
}

hipError_t launch_conv_v2(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk, const float *bnp,
                          float *out, int N, int num_cus) {
    const ConvVariant2 &v = g_variants2[p.variant - 1000];
    ConvArgs2 a;
    a.in = in; a.wpk = wpk; a.bnp = bnp; a.out = out;
    a.N = N; a.H = p.H; a.W = p.W; a.OH = p.OH; a.OW = p.OW;
    a.TH = p.TH; a.TW = p.TW; a.NI = p.NI;
    a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
    a.tile_floats = p.tile_floats;
    const int groups = (N + p.NI - 1) / p.NI;
    a.total_tiles = groups * p.tiles_y * p.tiles_x;
    if (a.total_tiles == 0) return hipSuccess;
    const int grid = std::min(a.total_tiles, num_cus * std::max(1, p.blocks_per_cu));
    hipLaunchKernelGGL(v.kernel, dim3(grid), dim3(p.threads), p.lds_bytes, s, a);
    return hipGetLastError();
}

}  // namespace asr
