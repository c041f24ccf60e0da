// Stand-alone check of mega::gemm_acc (csrc/mega_gemm.hpp): one workgroup per tile computes acc = sum_k B(j,k) A(i,k) over K columns
// and stores it; the host compares with a plain double loop.  Also times a launch that keeps every CU busy with `wgs` workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../morbit.jl_amd/csrc -I../../include -o gemmtest gemmtest.hip && ./gemmtest
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mega_gemm.hpp"

using namespace mrbf::mega;

// mode 0: A rows distinct per workgroup, B shared by all (B from L2); 1: both shared by the workgroups of a group of 8 (everything from L2);
// 2: B distinct per group of 8 workgroups too (both operands streamed from beyond L2: what a bulk job of the factorisation sees)
template <int TM>
__global__ __launch_bounds__(256, 2) void k_gemm(const double *A, int64_t lda, const double *B, int64_t ldb, int K, double *C, int reps, int mode = 0, int nbt = 1) {
    __shared__ __attribute__((aligned(16))) double smem[4 * 2 * 8 * LDS_LD];
    v4d acc[TM / 32][4];
    for (int j = 0; j < TM / 32; ++j)
        for (int i = 0; i < 4; ++i) acc[j][i] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double *Ab = A + (size_t)(mode == 1 ? blockIdx.x % 8 : blockIdx.x) * TM;  // tile rows of this workgroup
    if (mode == 2) B += (size_t)((blockIdx.x / 8) % nbt) * 128;
#ifdef TEST_V1
    for (int r = 0; r < reps; ++r) gemm_acc_v1<TM>(Ab, lda, B, ldb, K, acc, smem);
#else
    for (int r = 0; r < reps; ++r) gemm_acc_v2<TM>(Ab, lda, B, ldb, K, acc, smem);
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int ioff = (TM == 128) ? (wave >> 1) * 64 : 0, joff = (TM == 128) ? (wave & 1) * 64 : wave * 32;
    double *Cb = C + (size_t)blockIdx.x * TM * 128;
    for (int j = 0; j < TM / 32; ++j)
        for (int r = 0; r < 4; ++r)
            for (int i = 0; i < 4; ++i) Cb[(ioff + i * 16 + l15) + (size_t)(joff + j * 16 + l4 + 4 * r) * TM] = acc[j][i][r];
}

template <int TM>
int run(int K, int tiles, int reps) {
    const int64_t rowsA = (int64_t)tiles * TM, lda = rowsA + 2, ldb = 128 + 4;
    std::vector<double> A((size_t)lda * K), B((size_t)ldb * K), C((size_t)tiles * TM * 128);
    srand(K + TM);
    for (auto &x : A) x = (rand() % 2001 - 1000) / 1000.0;
    for (auto &x : B) x = (rand() % 2001 - 1000) / 1000.0;
    double *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 8);
    hipMalloc(&dB, B.size() * 8);
    hipMalloc(&dC, C.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_gemm<TM>, dim3(tiles), dim3(256), 0, 0, dA, lda, dB, ldb, K, dC, 1);
    hipDeviceSynchronize();
    hipMemcpy(C.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    long bad = 0;
    for (int t = 0; t < tiles; ++t)
        for (int j = 0; j < 128; ++j)
            for (int i = 0; i < TM; ++i) {
                double ref = 0;
                for (int k = 0; k < K; ++k) ref += A[(size_t)t * TM + i + (size_t)k * lda] * B[j + (size_t)k * ldb];
                const double e = std::abs(ref - C[(size_t)t * TM * 128 + i + (size_t)j * TM]);
                if (e > worst) worst = e;
                if (e > 1e-9) {
                    if (bad < 5) printf("  TM %d K %d tile %d (i %d, j %d): got %.6f ref %.6f\n", TM, K, t, i, j, C[(size_t)t * TM * 128 + i + (size_t)j * TM], ref);
                    ++bad;
                }
            }
    // timing: every CU busy
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gemm<TM>, dim3(tiles), dim3(256), 0, 0, dA, lda, dB, ldb, K, dC, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = 2.0 * TM * 128 * K * (double)reps * tiles;
    printf("TM %3d K %5d tiles %4d: max err %.2e, %ld bad | %d reps: %.3f ms = %.1f us per call, %.1f TFLOP/s\n", TM, K, tiles, worst, bad, reps, best,
           best * 1e3 / reps, flops / best / 1e9);
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
    return bad != 0;
}

template <int TM>
void run_mode(int K, int tiles, int reps, int mode) {
    const int nbt = 64;
    const int64_t rowsA = (int64_t)tiles * TM, lda = rowsA + 2, ldb = (int64_t)nbt * 128 + 4;
    double *dA, *dB, *dC;
    hipMalloc(&dA, (size_t)lda * K * 8);
    hipMalloc(&dB, (size_t)ldb * K * 8);
    hipMalloc(&dC, (size_t)tiles * TM * 128 * 8);
    hipMemset(dA, 0, (size_t)lda * K * 8);
    hipMemset(dB, 0, (size_t)ldb * K * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gemm<TM>, dim3(tiles), dim3(256), 0, 0, dA, lda, dB, ldb, K, dC, reps, mode, nbt);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("mode %d TM %3d K %5d tiles %4d: %.1f us per call, %.1f TFLOP/s\n", mode, TM, K, tiles, best * 1e3 / reps, 2.0 * TM * 128 * K * (double)reps * tiles / best / 1e9);
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
}

int main(int argc, char **argv) {
    int rc = 0;
    rc |= run<128>(128, 3, 20);
    rc |= run<64>(128, 3, 20);
    rc |= run<128>(768, 256, 20);   // one workgroup per CU: the "alone" rate
    rc |= run<128>(768, 512, 20);   // two per CU
    rc |= run<64>(768, 512, 20);
    rc |= run<128>(256, 512, 20);
    for (int mode = 0; mode < 3; ++mode) run_mode<128>(768, 512, 20, mode);
    for (int mode = 0; mode < 3; ++mode) run_mode<128>(768, 256, 20, mode);
    return rc;
}
