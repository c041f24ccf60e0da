// Stand-alone timing of the round-4 decision kernels (morbit.jl_amd/csrc/walk_kernel.hpp) on a synthetic block in which every candidate is
// accepted: S = 4 I + small symmetric noise, Ginv = I, pi small random.  Prints the kernel time and, per wave, the cycles spent in the
// four phases of a candidate step (0: publish / G pi, 1: barrier, 2: reads + pi' g + decision, 3: updates).
//   tools/walklab/build.sh && ./tools/walklab/walklab <q> <8: every wave both halves | 2: S waves and G waves> [reps]
#include "../../morbit.jl_amd/csrc/walk_kernel.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mrbf::r4;
#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

struct Bufs {
    double *dS, *dG, *dP, *dL;
    int *dacc, *dcnt, *dbi;
    unsigned long long *dprof;
};
constexpr int MAXACC = 4096;

template <class Launch>
static int run_generic(int q, int reps, const char *label, Launch launch) {
    std::vector<double> S((size_t)SB * SB), G((size_t)q * q, 0.0), P((size_t)SB * q);
    unsigned long long seed = 12345;
    auto rnd = [&]() {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(seed >> 11) / 9007199254740992.0 - 0.5;
    };
    for (int c = 0; c < SB; ++c)
        for (int r = 0; r <= c; ++r) {
            const double v = (r == c ? 4.0 : 0.0) + 0.01 * rnd();
            S[r + (size_t)c * SB] = v;
            S[c + (size_t)r * SB] = v;
        }
    for (int t = 0; t < q; ++t) G[t + (size_t)t * q] = 1.0;
    for (auto &x : P) x = 0.1 * rnd();
    Bufs B;
    CK(hipMalloc(&B.dS, S.size() * 8));
    CK(hipMalloc(&B.dG, G.size() * 8));
    CK(hipMalloc(&B.dP, P.size() * 8));
    CK(hipMalloc(&B.dL, (size_t)SB * SB * 8));
    CK(hipMalloc(&B.dacc, (MAXACC + 1) * 4));
    CK(hipMalloc(&B.dcnt, 8 * 4));
    CK(hipMalloc(&B.dbi, SB * 4));
    CK(hipMalloc(&B.dprof, 16 * 4 * 8));
    CK(hipMemcpy(B.dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    int nacc = 0;
    std::vector<double> Gout((size_t)q * q), Lout((size_t)SB * SB);
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipMemcpy(B.dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(B.dcnt, 0, 32));
        CK(hipEventRecord(e0, 0));
        if (launch(B, it == reps) != 0) return 1;
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it < reps && ms < best) best = ms;
        int hc[2];
        CK(hipMemcpy(hc, B.dcnt, 8, hipMemcpyDeviceToHost));
        nacc = hc[0];
    }
    CK(hipMemcpy(Gout.data(), B.dG, Gout.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(Lout.data(), B.dL, Lout.size() * 8, hipMemcpyDeviceToHost));
    double gsum = 0.0, lsum = 0.0;
    for (double x : Gout) gsum += x;
    for (double x : Lout) lsum += x;
    unsigned long long hp[64];
    CK(hipMemcpy(hp, B.dprof, 8 * 4 * 8, hipMemcpyDeviceToHost));
    printf("q=%d %s: %.1f us per block (best of %d), %d accepted -> %.2f us per candidate   [checksums: Ginv %.12g, L %.12g]\n", q, label, best * 1e3, reps,
           nacc, best * 1e3 / SB, gsum, lsum);
    for (int w = 0; w < 8; w += 2)
        printf("  wave %2d cycles per candidate: phase 0 %5.0f  barrier %5.0f  reads+decision %5.0f  updates %5.0f\n", w, hp[w * 4] / (double)SB,
               hp[w * 4 + 1] / (double)SB, hp[w * 4 + 2] / (double)SB, hp[w * 4 + 3] / (double)SB);
    return 0;
}

template <int TW, int NA, int NB_>
static int run(int q, int reps) {
    const size_t pis = (size_t)SB * 16 * NB_ * sizeof(double);
    CK(hipFuncSetAttribute((const void *)select_block_walk_kernel<TW, NA, NB_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pis));
    CK(hipFuncSetAttribute((const void *)select_block_walk_kernel<TW, NA, NB_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pis));
    return run_generic(q, reps, "every wave both halves", [&](const Bufs &B, bool prof) {
        if (!prof)
            hipLaunchKernelGGL((select_block_walk_kernel<TW, NA, NB_, false>), dim3(1), dim3(64 * TW), pis, 0, B.dS, SB, (int64_t)0, 0, q, 1 << 30, MAXACC, 1e-12,
                               B.dP, B.dG, B.dacc, B.dcnt, B.dL, B.dbi, (unsigned long long *)nullptr);
        else
            hipLaunchKernelGGL((select_block_walk_kernel<TW, NA, NB_, true>), dim3(1), dim3(64 * TW), pis, 0, B.dS, SB, (int64_t)0, 0, q, 1 << 30, MAXACC, 1e-12,
                               B.dP, B.dG, B.dacc, B.dcnt, B.dL, B.dbi, B.dprof);
        return 0;
    });
}

template <int N>
static int run_duo(int q, int reps) {
    const size_t pis = (size_t)SB * 16 * N * sizeof(double);
    CK(hipFuncSetAttribute((const void *)select_block_duo_kernel<N, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pis));
    CK(hipFuncSetAttribute((const void *)select_block_duo_kernel<N, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pis));
    return run_generic(q, reps, "S waves + G waves", [&](const Bufs &B, bool prof) {
        if (!prof)
            hipLaunchKernelGGL((select_block_duo_kernel<N, false>), dim3(1), dim3(512), pis, 0, B.dS, SB, (int64_t)0, 0, q, 1 << 30, MAXACC, 1e-12, B.dP, B.dG,
                               B.dacc, B.dcnt, B.dL, B.dbi, (unsigned long long *)nullptr);
        else
            hipLaunchKernelGGL((select_block_duo_kernel<N, true>), dim3(1), dim3(512), pis, 0, B.dS, SB, (int64_t)0, 0, q, 1 << 30, MAXACC, 1e-12, B.dP, B.dG,
                               B.dacc, B.dcnt, B.dL, B.dbi, B.dprof);
        return 0;
    });
}

int main(int argc, char **argv) {
    const int q = argc > 1 ? atoi(argv[1]) : 65, tw = argc > 2 ? atoi(argv[2]) : 8, reps = argc > 3 ? atoi(argv[3]) : 5;
    if (tw == 2) {
        if (q <= 32) return run_duo<2>(q, reps);
        if (q <= 64) return run_duo<4>(q, reps);
        if (q <= 80) return run_duo<5>(q, reps);
        if (q <= 128) return run_duo<8>(q, reps);
        if (q <= 144) return run_duo<9>(q, reps);
        return 2;
    }
    if (q <= 32) return run<8, 1, 2>(q, reps);
    if (q <= 64) return run<8, 2, 4>(q, reps);
    if (q <= 80) return run<8, 3, 5>(q, reps);
    if (q <= 128) return run<8, 4, 8>(q, reps);
    if (q <= 144) return run<8, 5, 9>(q, reps);
    fprintf(stderr, "q <= 144\n");
    return 2;
}
