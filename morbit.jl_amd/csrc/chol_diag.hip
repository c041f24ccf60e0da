// D: Cholesky of one 128 x 128 diagonal block + inverse of its factor -- MFMA-tiled version (stand-alone launch).
// The algorithm lives in chol_diag_core.hpp.
#include "chol_diag_core.hpp"

namespace mrbf {

__global__ __launch_bounds__(256, 1) void chol_diag_v4_kernel(double *__restrict__ A, int64_t lda, double *__restrict__ Linv,
                                                              int *__restrict__ info, int col0, int dbg) {
    __shared__ __attribute__((aligned(16))) diagcore::DiagV4Shared sh;
    if (*info != 0) return;
    __builtin_amdgcn_s_setprio(3);  // latency-critical: outrank the bulk update's waves sharing this CU's SIMDs
    diagcore::v4d acc[diagcore::NSLOT];
    const int bad = diagcore::diag_v4_core<false, false, false>(A, lda, Linv, sh, acc, nullptr, nullptr, dbg);
    if (bad && threadIdx.x == 0) *info = col0 + bad;
}

// the one-barrier-per-panel core (chol_diag_core.hpp, v6) as a stand-alone launch; scr: 8 x 256 doubles for the leaf inverses (the
// core assembles the inverse from them) followed by one progress word
__global__ __launch_bounds__(256, 1) void chol_diag_v6_kernel(double *__restrict__ A, int64_t lda, double *__restrict__ Linv,
                                                              int *__restrict__ info, int col0, double *__restrict__ scr) {
    __shared__ __attribute__((aligned(16))) diagcore::DiagV6Shared sh;
    if (*info != 0) return;
    __builtin_amdgcn_s_setprio(3);
    diagcore::v4d acc[diagcore::NSLOT6];
    diagcore::diag_v6_load(A, lda, acc);
    const int bad = diagcore::diag_v6_core(A, lda, Linv, sh, acc, scr, reinterpret_cast<unsigned *>(scr + 8 * 256));
    if (bad && threadIdx.x == 0) *info = col0 + bad;
}

int launch_diag_v4(mrbf_ctx *ctx, hipStream_t st, double *Ajj, int64_t lda, double *Linv, int *dinfo, int col0) {
    static const int dbg = mrbf_env("MRBF_DIAG_DBG") ? atoi(mrbf_env("MRBF_DIAG_DBG")) : 0;
    static const int impl = mrbf_env("MRBF_DIAG_KERNEL") ? atoi(mrbf_env("MRBF_DIAG_KERNEL")) : 6;
    if (impl == 6 && dbg == 0) {
        double *scr;
        MRBF_TRY(get_buf(ctx, S_DIAG_SCR, (size_t)8 * 256 + 16, &scr));
        hipLaunchKernelGGL(chol_diag_v6_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, Linv, dinfo, col0, scr);
        return 0;
    }
    hipLaunchKernelGGL(chol_diag_v4_kernel, dim3(1), dim3(256), 0, st, Ajj, lda, Linv, dinfo, col0, dbg);
    return 0;
}

}  // namespace mrbf
