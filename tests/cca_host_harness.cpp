// Serial host build of csrc/cca_solve.inl - TEST ONLY (tests/test_cca_solver_host.py):
// lets the CPU suite check the Jacobi numerics that the gfx950 kernel runs with
// 256 threads.  Not linked into libasr_hip.so.
#include <cmath>
#define CCA_FN
#define CCA_SYNC() do {} while (0)
#include "../audio_sheet_retrieval_amd/csrc/cca_solve.inl"

extern "C" void cca_solve_host(const double *S11, const double *S22, const double *S12, double *U, double *V,
                               double *coeffs) {
    static CcaScratch S;
    cca_solve(S, S11, S22, S12, U, V, coeffs, 0, 1);
}

extern "C" void cca_inv_sqrt_host(const double *Sin, double *out) {
    static CcaScratch S;
    cca_inv_sqrt_spd(S, Sin, out, 0, 1);
}
