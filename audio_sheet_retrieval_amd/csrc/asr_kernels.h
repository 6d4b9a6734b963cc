// Internal launcher interface between the C-ABI layer (asr_api.hip) and the
// gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asr {

// ---- first block: conv3x3 (C_in = 1) + BN + ELU, prepare() folded in -------
// in_mode: ASR_IN_* of asr_hip.h.  Hraw/Wraw: raw sheet size; H/W: network
// resolution (= raw, or raw/2 when rsz).  w: [COUT][9] correlation-form taps,
// bnp: [3][COUTP] (mean, gamma*inv_std, beta).
hipError_t launch_conv1(hipStream_t s, const void *in, int in_mode, int rsz,
                        const float *w, const float *bnp, float *out,
                        int N, int Hraw, int Wraw, int H, int W, int cout);

const char *conv1_symbol(int cout, int in_mode);

// ---- blocks 2..8: conv3x3 (C_in >= 12) as implicit GEMM on fp32 MFMA -------
struct ConvPlan {           // chosen on the host per layer geometry
    int cin, cout, pool;
    int H, W, OH, OW;
    int TH, TW, NI;         // tile: NI images x TH x TW output pixels (pre-pool)
    int tiles_y, tiles_x;
    int threads, lds_bytes, blocks_per_cu;
    int tile_floats;        // v2: floats of one LDS tile buffer
    int variant;            // index into the instantiation table
    const char *symbol;     // kernel symbol as rocprofv3 prints it
};
// Returns false when no instantiation exists for (cin, cout, pool).
bool plan_conv(int cin, int cout, int pool, int H, int W, ConvPlan *plan);
// second-generation schedule (conv_v2_kernels.hip); plan.variant >= 1000 marks a v2 plan
bool plan_conv_v2(int cin, int cout, int pool, int H, int W, ConvPlan *plan);
hipError_t launch_conv_v2(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk,
                          const float *bnp, float *out, int N, int num_cus);
size_t conv_wpack_floats(int cin, int cout);
// Wcorr: [9][cin][cout] correlation-form taps -> MFMA fragment order.
void pack_conv_weights(const float *wcorr, int cin, int cout, float *wpk);
hipError_t launch_conv(hipStream_t s, const ConvPlan &p, const float *in, const float *wpk,
                       const float *bnp, float *out, int N, int num_cus);

// ---- tail: 1x1 conv + BN + global mean (+ CCA projection + length norm) ----
// a8: [N, h, w, c8] NHWC; w9: [32][c8]; bnp9: [3][32]; cca_mean: [32];
// cca_proj: [32][32] (U or V, row-major, rows = input dim).
// features (N,32) and/or latent (N,32) may be null.
hipError_t launch_tail(hipStream_t s, const float *a8, int N, int h, int w, int c8,
                       const float *w9, const float *bnp9, const float *cca_mean,
                       const float *cca_proj, float *features, float *latent);

// ---- ranking ---------------------------------------------------------------
hipError_t launch_row_norms(hipStream_t s, const float *x, int64_t n, int64_t ld, int dim, double *norms);
hipError_t launch_rank(hipStream_t s, const float *lv1, const double *norm1, int64_t n1, int64_t ld1,
                       const float *lv2, const double *norm2, int64_t n2, int64_t ld2, int dim,
                       int64_t query_offset, int64_t k, int64_t h,
                       int32_t *ranks, double *dstar, int32_t *ties);

// top-k smallest cosine distances per query (exact float64, stable index order); k <= 128
hipError_t launch_topk(hipStream_t s, const float *db, const double *norm_db, int64_t n_db, int64_t ld_db,
                       const float *q, const double *norm_q, int64_t n_q, int64_t ld_q, int dim, int k,
                       int64_t idx_offset, int32_t *idx_out, double *dist_out);

// ---- CCA re-estimation (refine_cca.py / utils/cca.py 'svd') ------------------
size_t cca_workspace_bytes(int64_t n);
// H1,H2: [n][32] fp32 device; outputs device: U,V [32][32] fp32, means [64] fp32
// (m1 | m2), coeffs [32] fp64.
hipError_t launch_cca_fit(hipStream_t s, const float *H1, const float *H2, int64_t n, float r1, float r2,
                          void *workspace, float *U, float *V, float *means, double *coeffs);

}  // namespace asr
