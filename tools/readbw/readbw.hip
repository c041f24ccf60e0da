// Read bandwidth of a column-major n x n fp64 matrix walked the way symm_panel_kernel walks Phi: workgroup (bi, s) reads rows
// [R bi, R bi + R) of the columns of split s, 16 columns per step, every thread 16 bytes per column (a wave reads 1 KB of a column).
// R = 128 (what the kernel does), 256, 512: does a longer contiguous run per column read faster?
//   hipcc --offload-arch=gfx950 -O3 -o readbw readbw.hip && ./readbw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int R, int PF>
__global__ __launch_bounds__(256, 2) void walk(const double *__restrict__ A, long ld, long n, int klen, double *out) {
    // R rows per workgroup: thread t reads rows 2 (t % (R/2)) .. +1 of column k0 + t / (R/2) + (256 / (R/2)) u
    constexpr int CPP = 256 / (R / 2);        // columns covered by one pass of the workgroup
    constexpr int NU = 16 / CPP;              // passes per 16-column step
    const long I0 = (long)blockIdx.x * R;
    const long kbeg = (long)blockIdx.y * klen, kend = kbeg + klen < n ? kbeg + klen : n;
    const int r2 = (threadIdx.x % (R / 2)) * 2, c0 = threadIdx.x / (R / 2);
    v2d acc = {0.0, 0.0};
    v2d buf[PF][NU];
    auto fetch = [&](long kb, int slot) {
#pragma unroll
        for (int f = 0; f < PF; ++f)
            if (f == slot) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const long k = kb + c0 + CPP * u;
                    buf[f][u] = k < kend ? *(const v2d *)(A + I0 + r2 + k * ld) : (v2d){0.0, 0.0};
                }
            }
    };
#pragma unroll
    for (int f = 0; f < PF; ++f) fetch(kbeg + 16 * f, f);
    for (long kb0 = kbeg; kb0 < kend; kb0 += 16 * PF) {
#pragma unroll
        for (int f = 0; f < PF; ++f) {
            const long kb = kb0 + 16 * f;
            if (kb >= kend) break;
#pragma unroll
            for (int u = 0; u < NU; ++u) acc += buf[f][u];
            if (kb + 16 * PF < kend) fetch(kb + 16 * PF, f);
        }
    }
    if (acc[0] + acc[1] == 12345.678) out[0] = acc[0];
}

template <int R, int PF>
void run(const double *A, long n, int nsplit, double *out) {
    const int klen = (int)(((n + nsplit - 1) / nsplit + 15) / 16 * 16);
    dim3 grid((unsigned)(n / R), nsplit);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((walk<R, PF>), grid, dim3(256), 0, 0, A, n, n, klen, out);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL((walk<R, PF>), grid, dim3(256), 0, 0, A, n, n, klen, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("R = %3d rows per workgroup, %d steps in flight, grid %4u x %2d: %.3f ms = %.2f TB/s\n", R, PF, grid.x, nsplit, ms, 8.0 * n * n / ms / 1e9);
}

int main() {
    const long n = 8192;
    double *A, *out;
    CK(hipMalloc(&A, n * n * sizeof(double)));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(A, 0, n * n * sizeof(double)));
    run<128, 3>(A, n, 8, out);
    run<128, 3>(A, n, 16, out);
    run<256, 3>(A, n, 8, out);
    run<256, 3>(A, n, 16, out);
    run<256, 3>(A, n, 32, out);
    run<512, 3>(A, n, 16, out);
    run<512, 3>(A, n, 32, out);
    run<128, 1>(A, n, 8, out);
    run<256, 1>(A, n, 16, out);
    run<512, 2>(A, n, 32, out);
    return 0;
}
