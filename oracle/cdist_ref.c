/* Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): plain-C form of
 * oracle/retrieval.py:cdist_cosine64, i.e. of scipy.spatial.distance.cdist(A, B,
 * "cosine") as the reference calls it (audio_sheet_retrieval/utils/train_dcca_pool.py:40,
 * audio_sheet_server.py:534,553) - float64 from float32 inputs, SciPy's summation
 * order (two accumulators over even / odd k, added at the end, odd tail last).
 * It exists because the NumPy form takes ~0.3 s per query row against 2 M
 * candidates; tests/test_oracle_retrieval.py checks the two bit-for-bit.
 * Every product of two float32 values is exact in float64, so contraction of
 * mul+add into fma cannot change a result; only the order matters.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static double sum2_prod(const float *u, const float *v, int dim)
{
    double a0 = 0.0, a1 = 0.0;
    const int m = dim / 2 * 2;
    for (int k = 0; k < m; k += 2) {
        a0 = a0 + (double)u[k] * (double)v[k];
        a1 = a1 + (double)u[k + 1] * (double)v[k + 1];
    }
    double s = a0 + a1;
    for (int k = m; k < dim; ++k) s = s + (double)u[k] * (double)v[k];
    return s;
}

/* norms[i] = sqrt(sum2(x_i * x_i)) */
void row_norms_f64(const float *X, int64_t n, int64_t ld, int dim, double *norms)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) norms[i] = sqrt(sum2_prod(X + i * ld, X + i * ld, dim));
}

/* out[i][j] = 1 - clip(dot(a_i, b_j) / (|a_i| * |b_j|)) for rows i of A (n1 x dim) against all of B (n2 x dim) */
void cdist_cosine_f64(const float *A, int64_t n1, const float *B, int64_t n2, int dim,
                      const double *na, const double *nb, double *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t i = 0; i < n1; ++i)
        for (int64_t j0 = 0; j0 < n2; j0 += 4096) {
            const int64_t j1 = j0 + 4096 < n2 ? j0 + 4096 : n2;
            for (int64_t j = j0; j < j1; ++j) {
                double c = sum2_prod(A + i * dim, B + j * dim, dim) / (na[i] * nb[j]);
                if (fabs(c) > 1.0) c = copysign(1.0, c);
                out[i * n2 + j] = 1.0 - c;
            }
        }
}
