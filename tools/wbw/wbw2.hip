// Store-only ceilings for the Gram kernel's write patterns: the same grid, tile-pair decoding, 64-row halves,
// 256-thread workgroups (occupancy 4) and address arithmetic as gram_mfma64_kernel, but NO loads, MFMAs or radial functions --
// each variant only issues the stores of one candidate epilogue.  n = 8192, ld = 8320 (the fit's leading dimension).
//   P0  round-1 pattern: direct tile one double per lane (4 rows x 128 B per wave-instruction), mirrored tile 16 B per lane
//       (8 rows x 128 B per wave-instruction)
//   P1  pair-mapped columns: direct tile 16 B per lane, 4 rows x 256 B; mirrored tile 16 B per lane, 4 rows x 256 B
//   P2  whole rows: direct tile 1 row x 1 KB per wave-instruction, mirrored tile 2 rows x 512 B
//   P3  lower triangle only (no mirrored tile), 16 B per lane, 4 rows x 256 B
// each with plain and non-temporal stores.  Build: hipcc --offload-arch=gfx950 -O3 -o wbw2 wbw2.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void tri_decode(int bid, int &ti, int &tj) {
    int t = (int)((sqrt(8.0 * (double)bid + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= bid) ++t;
    while (t * (t + 1) / 2 > bid) --t;
    ti = t;
    tj = bid - t * (t + 1) / 2;
}

template <bool NT>
__device__ __forceinline__ void st1(double *p, double v) {
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}
template <bool NT>
__device__ __forceinline__ void st2(double *p, v2d v) {
    if (NT)
        __builtin_nontemporal_store(v, (v2d *)p);
    else
        *(v2d *)p = v;
}

template <int PAT, bool NT>
__global__ __launch_bounds__(256, 4) void store_only(double *__restrict__ Phi, long ld, double seed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    int ti, tj;
    tri_decode(blockIdx.x >> 1, ti, tj);
    const int half = blockIdx.x & 1;
    const long I0 = (long)ti * 128 + 64 * half, J0 = (long)tj * 128;
    const double v = seed + blockIdx.x;
    const v2d vv = {v, v + 1.0};
    if (PAT == 0) {
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 4; ++r)
                for (int j = 0; j < 2; ++j) st1<NT>(Phi + (I0 + i * 16 + l4 + 4 * r) * ld + J0 + wave * 32 + j * 16 + l15, v);
        if (ti == tj) return;
        for (int i = 0; i < 4; ++i)
            for (int it = 0; it < 4; ++it) st2<NT>(Phi + (J0 + wave * 32 + it * 8 + (lane >> 3)) * ld + I0 + i * 16 + 2 * (lane & 7), vv);
    } else if (PAT == 1 || PAT == 3) {
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 4; ++r) st2<NT>(Phi + (I0 + i * 16 + l4 + 4 * r) * ld + J0 + wave * 32 + 2 * l15, vv);
        if (ti == tj || PAT == 3) return;
        for (int p = 0; p < 2; ++p)
            for (int it = 0; it < 8; ++it) st2<NT>(Phi + (J0 + wave * 32 + 4 * it + l4) * ld + I0 + 32 * p + 2 * l15, vv);
    } else if (PAT == 2) {
        for (int rr = 0; rr < 16; ++rr) st2<NT>(Phi + (I0 + wave * 16 + rr) * ld + J0 + 2 * lane, vv);
        if (ti == tj) return;
        for (int rr = 0; rr < 16; ++rr) st2<NT>(Phi + (J0 + wave * 32 + 2 * rr + (lane >> 5)) * ld + I0 + 2 * (lane & 31), vv);
    }
}

template <int PAT, bool NT>
static void run(double *A, long ld, long n, const char *name) {
    const long nt = n / 128, nb = nt * (nt + 1) / 2;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((store_only<PAT, NT>), dim3((unsigned)(2 * nb)), dim3(256), 0, 0, A, ld, 1.0);
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((store_only<PAT, NT>), dim3((unsigned)(2 * nb)), dim3(256), 0, 0, A, ld, 2.0 + r);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (PAT == 3) ? 8.0 * (n * (n + 128) / 2) : 8.0 * n * n;
    printf("%-44s %s : %7.1f us  %.2f TB/s  (%.0f MB)\n", name, NT ? "nt   " : "plain", ms * 1e3, bytes / ms / 1e9, bytes / 1e6);
}

int main() {
    const long n = 8192, ld = 8320;
    double *A;
    if (hipMalloc(&A, sizeof(double) * ld * n) != hipSuccess) return 1;
    run<0, false>(A, ld, n, "P0 round-1 (8B direct, 128-B segments)");
    run<0, true>(A, ld, n, "P0 round-1 (8B direct, 128-B segments)");
    run<1, false>(A, ld, n, "P1 pair-mapped (16B, 256-B segments)");
    run<1, true>(A, ld, n, "P1 pair-mapped (16B, 256-B segments)");
    run<2, false>(A, ld, n, "P2 whole rows (1 KB / 512 B segments)");
    run<2, true>(A, ld, n, "P2 whole rows (1 KB / 512 B segments)");
    run<3, false>(A, ld, n, "P3 lower triangle only (16B, 256-B)");
    run<3, true>(A, ld, n, "P3 lower triangle only (16B, 256-B)");
    hipMemset(A, 0, sizeof(double) * ld * n);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a, 0);
    for (int r = 0; r < 10; ++r) hipMemsetAsync(A, 0, sizeof(double) * n * n, 0);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("hipMemsetAsync of 8 n^2 bytes                              : %7.1f us  %.2f TB/s\n", ms * 100, 8.0 * n * n / (ms / 10) / 1e9);
    return 0;
}
