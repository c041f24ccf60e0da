// Stand-alone timing of the round-4 decision kernel (morbit.jl_amd/csrc/walk_kernel.hpp) on a synthetic block in which every candidate is
// accepted: S = 4 I + small symmetric noise, Ginv = I, pi small random.  Prints the kernel time and, per wave, the cycles spent in the
// four phases of a candidate step (0: column publish + G pi, 1: barrier, 2: reads + pi' g + decision, 3: updates).
//   tools/walklab/build.sh && ./tools/walklab/walklab <q> 8 [reps]
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

template <int TW, int NA, int NB_>
static int run(int q, int reps) {
    const int b = SB, maxacc = 4096;
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
    double *dS, *dG, *dP, *dL;
    int *dacc, *dcnt, *dbi;
    unsigned long long *dprof;
    CK(hipMalloc(&dS, S.size() * 8));
    CK(hipMalloc(&dG, G.size() * 8));
    CK(hipMalloc(&dP, P.size() * 8));
    CK(hipMalloc(&dL, (size_t)SB * SB * 8));
    CK(hipMalloc(&dacc, (maxacc + 1) * 4));
    CK(hipMalloc(&dcnt, 8 * 4));
    CK(hipMalloc(&dbi, SB * 4));
    CK(hipMalloc(&dprof, 16 * 4 * 8));
    CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    const size_t pis = (size_t)SB * 16 * NB_ * sizeof(double);
    CK(hipFuncSetAttribute((const void *)select_block_walk_kernel<TW, NA, NB_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pis));
    CK(hipFuncSetAttribute((const void *)select_block_walk_kernel<TW, NA, NB_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pis));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    int nacc = 0;
    for (int it = 0; it < reps + 1; ++it) {
        CK(hipMemcpy(dG, G.data(), G.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(dcnt, 0, 32));
        CK(hipEventRecord(e0, 0));
        if (it < reps)
            hipLaunchKernelGGL((select_block_walk_kernel<TW, NA, NB_, false>), dim3(1), dim3(64 * TW), pis, 0, dS, b, (int64_t)0, 0, q, 1 << 30, maxacc, 1e-12,
                               dP, dG, dacc, dcnt, dL, dbi, (unsigned long long *)nullptr);
        else
            hipLaunchKernelGGL((select_block_walk_kernel<TW, NA, NB_, true>), dim3(1), dim3(64 * TW), pis, 0, dS, b, (int64_t)0, 0, q, 1 << 30, maxacc, 1e-12,
                               dP, dG, dacc, dcnt, dL, dbi, dprof);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it < reps && ms < best) best = ms;
        int hc[2];
        CK(hipMemcpy(hc, dcnt, 8, hipMemcpyDeviceToHost));
        nacc = hc[0];
    }
    unsigned long long hp[64];
    CK(hipMemcpy(hp, dprof, TW * 4 * 8, hipMemcpyDeviceToHost));
    printf("q=%d waves=%d <%d,%d>: %.1f us per block (best of %d), %d accepted -> %.2f us per candidate\n", q, TW, NA, NB_, best * 1e3, reps, nacc,
           best * 1e3 / SB);
    for (int w = 0; w < TW; w += TW / 4)
        printf("  wave %2d cycles per candidate: publish+Gpi %5.0f  barrier %5.0f  reads+decision %5.0f  updates %5.0f\n", w, hp[w * 4] / (double)SB,
               hp[w * 4 + 1] / (double)SB, hp[w * 4 + 2] / (double)SB, hp[w * 4 + 3] / (double)SB);
    return 0;
}

int main(int argc, char **argv) {
    const int q = argc > 1 ? atoi(argv[1]) : 65, tw = argc > 2 ? atoi(argv[2]) : 8, reps = argc > 3 ? atoi(argv[3]) : 5;
    if (q <= 32) return run<8, 1, 2>(q, reps);
    if (q <= 64) return run<8, 2, 4>(q, reps);
    if (q <= 80) return run<8, 3, 5>(q, reps);
    if (q <= 128) return run<8, 4, 8>(q, reps);
    if (q <= 144) return run<8, 5, 9>(q, reps);
    fprintf(stderr, "q <= 144\n");
    return 2;
}
