// Serial host build of csrc/cca_solve.inl - TEST ONLY (tests/test_cca_solver_host.py): the 32x32 float64 algebra of
// CCA('svd') (reference utils/cca.py:199-211) exactly as the gfx950 kernel runs it, compiled for the CPU with
// AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool), checked against the
// NumPy / SciPy oracle.  Not linked into libasr_hip.so.
//   usage: cca_host_harness <in.bin> <out.bin>
//     in : n_cases x [S11 | S22 | S12] (3 x 1024 float64 each);  out: n_cases x [U | V | coeffs | S11^-1/2] (3104 float64)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CCA_FN
#define CCA_SYNC() do {} while (0)
#include "../audio_sheet_retrieval_amd/csrc/cca_solve.inl"

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror("in"); return 2; }
    fseek(fi, 0, SEEK_END);
    const long bytes = ftell(fi);
    fseek(fi, 0, SEEK_SET);
    const size_t per_case = 3 * 1024;
    const size_t n = (size_t)bytes / (per_case * sizeof(double));
    std::vector<double> in(n * per_case), out(n * 3104);
    if (fread(in.data(), sizeof(double), in.size(), fi) != in.size()) { fprintf(stderr, "short read\n"); return 2; }
    fclose(fi);
    // heap-allocated scratch: out-of-bounds accesses in the solver are caught by ASan's red zones
    CcaScratch *S = new CcaScratch();
    for (size_t c = 0; c < n; ++c) {
        const double *S11 = in.data() + c * per_case, *S22 = S11 + 1024, *S12 = S22 + 1024;
        double *U = out.data() + c * 3104, *V = U + 1024, *coeffs = V + 1024, *A = coeffs + 32;
        cca_solve(*S, S11, S22, S12, U, V, coeffs, 0, 1);
        cca_inv_sqrt_spd(*S, S11, A, 0, 1);
    }
    delete S;
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror("out"); return 2; }
    fwrite(out.data(), sizeof(double), out.size(), fo);
    fclose(fo);
    return 0;
}
