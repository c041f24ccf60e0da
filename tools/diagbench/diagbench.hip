// Stand-alone timing of the 128 x 128 diagonal-block core (chol_diag_core.hpp), one workgroup.  (Round 2 also timed a wave-specialised
// variant with LDS flags instead of workgroup barriers here: 54-59 us per block against 55-57 us, not kept -- DESIGN.md section 3.)  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I/opt/rocm/include -I../../morbit.jl_amd/csrc -o diagbench diagbench.hip
#include "chol_diag_core.hpp"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace mrbf;

template <int V>
__global__ __launch_bounds__(256, 2) void k(double *A, double *Linv, int *info, unsigned long long *ts, int reps, int dbg) {
    __shared__ __attribute__((aligned(16))) struct {
        diagcore::DiagV4Shared s4;
    } sh;
    __builtin_amdgcn_s_setprio(3);
    if ((threadIdx.x & 63) == 0) ts[16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
    for (int r = 0; r < reps; ++r) {
        double *Ar = A + (size_t)r * 128 * 128;
        const unsigned long long t0 = __builtin_readcyclecounter();
        int bad;
        diagcore::v4d acc[diagcore::NSLOT];
        bad = diagcore::diag_v4_core<true, false, false>(Ar, 128, Linv, sh.s4, acc, nullptr, nullptr, dbg);
        __syncthreads();
        if (threadIdx.x == 0) {
            ts[0] += __builtin_readcyclecounter() - t0;
            if (bad) *info = bad;
        }
    }
}

int main(int argc, char **argv) {
    const int dbg = argc > 1 ? atoi(argv[1]) : 0;  // 4: per-segment cycles of the leaf wave (no inverse check then)
    const int reps = dbg ? 1 : 64;
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    std::vector<double> G(128 * 160), A(128 * 128), all((size_t)reps * 128 * 128), L(128 * 128), Li(128 * 128);
    for (auto &x : G) x = nd(g);
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
            double s = 0;
            for (int k2 = 0; k2 < 160; ++k2) s += G[i * 160 + k2] * G[j * 160 + k2];
            A[i + j * 128] = s / 128 + (i == j);
        }
    double *dA, *dL;
    int *dinfo;
    unsigned long long *dts;
    hipMalloc(&dA, all.size() * 8);
    hipMalloc(&dL, 128 * 128 * 8);
    hipMalloc(&dinfo, 4);
    hipMalloc(&dts, 64 * 8);
    for (int v = 4; v <= 4; ++v) {
        for (int r = 0; r < reps; ++r) std::copy(A.begin(), A.end(), all.begin() + (size_t)r * 128 * 128);
        hipMemcpy(dA, all.data(), all.size() * 8, hipMemcpyHostToDevice);
        hipMemset(dinfo, 0, 4);
        hipMemset(dts, 0, 64 * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(256), 0, 0, dA, dL, dinfo, dts, reps, dbg);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long ts[64];
        int info;
        hipMemcpy(ts, dts, sizeof(ts), hipMemcpyDeviceToHost);
        hipMemcpy(&info, dinfo, 4, hipMemcpyDeviceToHost);
        hipMemcpy(L.data(), dA, 128 * 128 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(Li.data(), dL, 128 * 128 * 8, hipMemcpyDeviceToHost);
        // check: L L' = A on the lower triangle, Linv L = I
        double e = 0, ei = 0;
        for (int i = 0; i < 128; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0, t = 0;
                for (int k2 = 0; k2 <= j; ++k2) s += L[i + k2 * 128] * L[j + k2 * 128];
                for (int k2 = j; k2 <= i; ++k2) t += Li[i + k2 * 128] * L[k2 + j * 128];
                e = std::max(e, std::fabs(s - A[i + j * 128]));
                ei = std::max(ei, std::fabs(t - (i == j)));
            }
        if (dbg & 4)
            printf("leaf wave cycles over 8 panels: pre-leaf %.0f, leaf %.0f, post-leaf LDS %.0f, barrier X %.0f, look-ahead work %.0f, barrier Y %.0f\n", Li[0], Li[1], Li[2], Li[3], Li[4], Li[5]);
        printf("wave -> simd:");
        for (int w = 0; w < 4; ++w) printf(" %llu(cu %llu)", (ts[16 + w] >> 4) & 3, (ts[16 + w] >> 8) & 15);
        printf("\n");
        printf("v%d: %.2f us per block (events), %.0f cycles per block (shader clock), info %d, |LL'-A| %.1e |Linv L - I| %.1e\n", v,
               ms * 1e3 / reps, (double)ts[0] / reps, info, e, ei);
    }
    return 0;
}
