/* Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): plain-C restatement of
 * the convolution the reference reaches through Lasagne's Conv2DLayer
 * (audio_sheet_retrieval/models/mutopia_ccal_cont.py:54-58,93,118;
 * flip_filters=True, pad='same' for 3x3, stride 1, no bias - SURVEY A.1,
 * "third-party semantic, unverified offline").
 *
 *   y[n,h,w,o] = sum_{a,b,i} W[o,i,a,b] * x[n, h+p-a, w+p-b, i],  p=(k-1)/2
 *
 * The caller passes the taps already in correlation form
 *   wt[a'][b'][i][o] = W[o][i][k-1-a'][k-1-b']
 * so that  y[n,h,w,o] = sum_{a',b',i} wt[a'][b'][i][o] * x[n,h-p+a',w-p+b',i].
 * float32 accumulation in the order a', b', i (outer -> inner); NHWC layout.
 * It exists only because the NumPy form of the same sum is ~20x slower; the
 * NumPy form (oracle/network.py:conv2d_flip_nhwc_numpy) is kept and the two
 * are compared in tests/test_oracle_network.py.
 */
#include <stddef.h>
#include <string.h>

#define MAXCO 128

/* one output row (n,h,*) */
__attribute__((target_clones("arch=haswell", "default")))
static void conv_row(const float *x, const float *wt, float *y,
                     int n, int h, int H, int W, int CI, int CO, int K)
{
    const int p = (K - 1) / 2;
    float acc[MAXCO];
    for (int w = 0; w < W; ++w) {
        for (int o = 0; o < CO; ++o) acc[o] = 0.0f;
        for (int a = 0; a < K; ++a) {
            const int hh = h - p + a;
            if (hh < 0 || hh >= H) continue;
            for (int b = 0; b < K; ++b) {
                const int ww = w - p + b;
                if (ww < 0 || ww >= W) continue;
                const float *xp = x + (((size_t)n * H + hh) * W + ww) * CI;
                const float *wp = wt + ((size_t)(a * K + b) * CI) * CO;
                for (int i = 0; i < CI; ++i) {
                    const float xv = xp[i];
                    const float *wr = wp + (size_t)i * CO;
                    for (int o = 0; o < CO; ++o) acc[o] += xv * wr[o];
                }
            }
        }
        memcpy(y + (((size_t)n * H + h) * W + w) * CO, acc, sizeof(float) * CO);
    }
}

void conv2d_corr_nhwc_f32(const float *x, const float *wt, float *y,
                          int N, int H, int W, int CI, int CO, int K)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            conv_row(x, wt, y, n, h, H, W, CI, CO, K);
}

/* The same sum in float64 (the float64 oracle of the training step, oracle/train.py: the NumPy form needs nine
 * strided copies + GEMMs per block, minutes at batch 512).  Accumulation order a', b', i as above. */
__attribute__((target_clones("arch=haswell", "default")))
static void conv_row_f64(const double *x, const double *wt, double *y,
                         int n, int h, int H, int W, int CI, int CO, int K)
{
    const int p = (K - 1) / 2;
    double acc[MAXCO];
    for (int w = 0; w < W; ++w) {
        for (int o = 0; o < CO; ++o) acc[o] = 0.0;
        for (int a = 0; a < K; ++a) {
            const int hh = h - p + a;
            if (hh < 0 || hh >= H) continue;
            for (int b = 0; b < K; ++b) {
                const int ww = w - p + b;
                if (ww < 0 || ww >= W) continue;
                const double *xp = x + (((size_t)n * H + hh) * W + ww) * CI;
                const double *wp = wt + ((size_t)(a * K + b) * CI) * CO;
                for (int i = 0; i < CI; ++i) {
                    const double xv = xp[i];
                    const double *wr = wp + (size_t)i * CO;
                    for (int o = 0; o < CO; ++o) acc[o] += xv * wr[o];
                }
            }
        }
        memcpy(y + (((size_t)n * H + h) * W + w) * CO, acc, sizeof(double) * CO);
    }
}

void conv2d_corr_nhwc_f64(const double *x, const double *wt, double *y,
                          int N, int H, int W, int CI, int CO, int K)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            conv_row_f64(x, wt, y, n, h, H, W, CI, CO, K);
}

/* ELU, lasagne.nonlinearities.elu = switch(x > 0, x, expm1(x)),
 * float32; in place over `count` values.  (np.expm1 over whole feature maps
 * dominated the NumPy oracle's run time.) */
#include <math.h>
void elu_f32(float *x, size_t count)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < count; ++i) {
        const float v = x[i];
        x[i] = v > 0.0f ? v : expm1f(v);
    }
}

/* Deterministic BatchNormLayer:
 *   y = (x - mean[c]) * (gamma[c] * inv_std[c]) + beta[c],  NHWC rows x C. */
void bn_det_nhwc_f32(const float *x, const float *beta, const float *gamma,
                     const float *mean, const float *inv_std, float *y,
                     size_t rows, int C)
{
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; ++r) {
        const float *xr = x + r * (size_t)C;
        float *yr = y + r * (size_t)C;
        for (int c = 0; c < C; ++c)
            yr[c] = (xr[c] - mean[c]) * (gamma[c] * inv_std[c]) + beta[c];
    }
}
