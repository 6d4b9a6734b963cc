// Per-element bodies of the weight re-layout kernels (master OIHW parameters -> what the conv kernels read), shared by
// the stand-alone kernels (asr_set_params, refresh after training) and the table-driven repack_all_kernel that
// re-derives every layout of both towers in ONE launch after each training update (train_bwd_kernels.hip).
#pragma once

namespace asr {

// W: OIHW (cout, cin, 3, 3).  fwd: fragment order of Wcorr[tap][ci][co] = W[co][ci][2-a'][2-b']
__device__ __forceinline__ void repack_conv_fwd_elem(int e, const float *__restrict__ W, int cin, int cout,
                                                      float *__restrict__ wfwd) {
    const int KSf = cin / 4;
    const int lane = e & 63;
    const int j = (e >> 6) % KSf;
    const int tap = ((e >> 6) / KSf) % 9;
    const int nt = (e >> 6) / KSf / 9;
    const int g = lane >> 4, n = lane & 15;
    const int ci = g * KSf + j, co = nt * 16 + n;
    const int a = tap / 3, b = tap % 3;
    wfwd[e] = co < cout ? W[((size_t)co * cin + ci) * 9 + (2 - a) * 3 + (2 - b)] : 0.0f;
}
__device__ __forceinline__ int repack_conv_fwd_count(int cin, int cout) { return (cout + 15) / 16 * 9 * (cin / 4) * 64; }

// dgrad: fragment order of Wd[tap'][co][ci] = Wcorr[8-tap'][ci][co] (roles of ci/co swapped)
__device__ __forceinline__ void repack_conv_dgrad_elem(int e, const float *__restrict__ W, int cin, int cout,
                                                        float *__restrict__ wdgrad) {
    const int KSd = cout / 4;
    const int lane = e & 63;
    const int j = (e >> 6) % KSd;
    const int tap = ((e >> 6) / KSd) % 9;
    const int nt = (e >> 6) / KSd / 9;
    const int g = lane >> 4, n = lane & 15;
    const int co = g * KSd + j;          // contraction index of the data gradient
    const int ci = nt * 16 + n;          // its output channel
    // Wd[tap][co][ci] = Wcorr[8-tap][ci][co] = W[co][ci][2-a''][2-b''] with (a'',b'') = taps of 8-tap
    const int t2 = 8 - tap, a = t2 / 3, b = t2 % 3;
    wdgrad[e] = ci < cin ? W[((size_t)co * cin + ci) * 9 + (2 - a) * 3 + (2 - b)] : 0.0f;
}
__device__ __forceinline__ int repack_conv_dgrad_count(int cin, int cout) { return (cin + 15) / 16 * 9 * (cout / 4) * 64; }

// block 1 (C_in = 1): [co][9] correlation-form taps
__device__ __forceinline__ void repack_conv1_elem(int e, const float *__restrict__ W, float *__restrict__ w1) {
    const int co = e / 9, t = e % 9, a = t / 3, b = t % 3;
    w1[e] = W[(size_t)co * 9 + (2 - a) * 3 + (2 - b)];
}

// deterministic-path BN fold: [mean | gamma*inv_std | beta] padded to coutp
__device__ __forceinline__ void bn_fold_elem(int c, const float *__restrict__ beta, const float *__restrict__ gamma,
                                             const float *__restrict__ mean, const float *__restrict__ istd, int cout,
                                             int coutp, float *__restrict__ bnp) {
    bnp[c] = c < cout ? mean[c] : 0.0f;
    bnp[coutp + c] = c < cout ? gamma[c] * istd[c] : 0.0f;
    bnp[2 * coutp + c] = c < cout ? beta[c] : 0.0f;
}

// Winograd F(2x2,3x3) weight transform: master W (Lasagne layout [co][ci][3][3], convolution form) -> U = G g G^T in
// float64, stored [k-step][p][g][coutp] (zero padded) with the kernels' channel order: k-steps 2t, 2t+1 of lane group
// g <-> contraction channels 8t+2g, 8t+2g+1; remainder 8*NB + g.
//   forward:        contraction over ci, outputs co, correlation taps g[a][b] = W[co][ci][2-a][2-b]
//   data gradient:  contraction over co, outputs ci, taps g'[a][b] = W[co][ci][a][b]
// idx in [0, kdim * coutp)
__device__ __forceinline__ void wino_pack_elem(int idx, const float *__restrict__ W, int cin, int cout, int dgrad,
                                               float *__restrict__ wpk) {
    const int kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
    const int coutp = (ndim + 15) / 16 * 16;
    const int n = idx % coutp, k = idx / coutp;
    const int nb = kdim / 8;
    int ks, g;
    if (k < nb * 8) {
        const int t = k >> 3, w = k & 7;
        g = w >> 1;
        ks = 2 * t + (w & 1);
    } else {
        g = k - nb * 8;
        ks = 2 * nb;
    }
    double gm[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            if (n < ndim)
                v = dgrad ? (double)W[((size_t)k * cin + n) * 9 + i * 3 + j]
                          : (double)W[((size_t)n * cin + k) * 9 + (2 - i) * 3 + (2 - j)];
            gm[i][j] = v;
        }
    double t[4][3];                                            // G g
    for (int j = 0; j < 3; ++j) {
        t[0][j] = gm[0][j];
        t[1][j] = 0.5 * (gm[0][j] + gm[1][j] + gm[2][j]);
        t[2][j] = 0.5 * (gm[0][j] - gm[1][j] + gm[2][j]);
        t[3][j] = gm[2][j];
    }
    // rows are ordered [k-step][position][lane group]: a k-step's 16 positions sit within ds_read immediate reach
    for (int i = 0; i < 4; ++i) {                              // (G g) G^T
        const double u[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
        for (int j = 0; j < 4; ++j) wpk[((size_t)(ks * 16 + i * 4 + j) * 4 + g) * coutp + n] = (float)u[j];
    }
}
__device__ __forceinline__ int wino_pack_count(int cin, int cout, int dgrad) {
    const int kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
    return kdim * ((ndim + 15) / 16 * 16);
}

// Winograd F(4x4,3x3) weight transform (conv_wino4_kernels.hip): U = G g G^T (6x6 per channel pair) in float64, stored per
// channel block t of 8 contraction channels as [t][row / 4][g][coutp][row % 4], row = (k-step parity) * 36 + 6 xi + nu.
// idx in [0, kdim * coutp); forward / data-gradient roles as in wino_pack_elem
__device__ __forceinline__ void wino4_pack_elem(int idx, const float *__restrict__ W, int cin, int cout, int dgrad,
                                                float *__restrict__ wpk) {
    const int kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
    const int coutp = (ndim + 15) / 16 * 16;
    const int n = idx % coutp, k = idx / coutp;
    const int t = k >> 3, w = k & 7;
    const int g = w >> 1, ks = 2 * t + (w & 1);
    (void)kdim;
    double gm[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            if (n < ndim)
                v = dgrad ? (double)W[((size_t)k * cin + n) * 9 + i * 3 + j]
                          : (double)W[((size_t)n * cin + k) * 9 + (2 - i) * 3 + (2 - j)];
            gm[i][j] = v;
        }
    const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double tg[6][3];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) tg[i][j] = G[i][0] * gm[0][j] + G[i][1] * gm[1][j] + G[i][2] * gm[2][j];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const double u = tg[i][0] * G[j][0] + tg[i][1] * G[j][1] + tg[i][2] * G[j][2];
            const int qq = (ks & 1) * 36 + i * 6 + j;        // row of the channel block: k-step parity x position
            wpk[((((size_t)t * 18 + (qq >> 2)) * 4 + g) * coutp + n) * 4 + (qq & 3)] = (float)u;
        }
}
__device__ __forceinline__ int wino4_pack_count(int cin, int cout, int dgrad) {
    const int kdim = dgrad ? cout : cin, ndim = dgrad ? cin : cout;
    return (kdim % 8) ? 0 : kdim * ((ndim + 15) / 16 * 16);
}

}  // namespace asr
