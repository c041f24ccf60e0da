// Write-bandwidth experiment for the Gram matrix store pattern: how fast can 8 n^2 bytes be written when every workgroup
// writes a TR x TC tile of a column-major matrix (column segments of TR * 8 bytes) and, optionally, the mirrored TC x TR tile?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
// one workgroup per (bi, bj) tile of TR rows x TC cols; 256 threads; WIDE: 16-byte stores
template <int TR, int TC, bool WIDE>
__global__ __launch_bounds__(256, 2) void write_tiles(double *A, long ld, int ntr, int ntc, int lower_only, int order) {
    int bid = blockIdx.x;
    int bi, bj;
    if (order == 0) { bi = bid / ntc; bj = bid % ntc; } else { bj = bid / ntr; bi = bid % ntr; }
    if (lower_only && (long)bj * TC > (long)bi * TR + TR - 1) return;
    double *T = A + (long)bi * TR + (long)bj * TC * ld;
    const int tid = threadIdx.x;
    if (WIDE) {
        constexpr int VPC = TR / 2;  // v2d per column
        for (int e = tid; e < VPC * TC; e += 256) {
            const int c = e / VPC, r2 = e % VPC;
            *(v2d *)(T + 2 * r2 + (long)c * ld) = (v2d){1.0 + bid, 2.0};
        }
    } else {
        for (int e = tid; e < TR * TC; e += 256) {
            const int c = e / TR, r = e % TR;
            T[r + (long)c * ld] = 1.0 + bid;
        }
    }
}
template <int TR, int TC, bool WIDE>
float run(double *A, long ld, long n, int lower_only, int order, int reps) {
    const int ntr = n / TR, ntc = n / TC;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((write_tiles<TR, TC, WIDE>), dim3(ntr * ntc), dim3(256), 0, 0, A, ld, ntr, ntc, lower_only, order);
    hipEventRecord(a, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((write_tiles<TR, TC, WIDE>), dim3(ntr * ntc), dim3(256), 0, 0, A, ld, ntr, ntc, lower_only, order);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const long n = 8192, ld = 8320;
    double *A; hipMalloc(&A, sizeof(double) * ld * n);
    const double bytes = 8.0 * n * n;
#define R(TRv, TCv, W, ord) { float ms = run<TRv, TCv, W>(A, ld, n, 0, ord, 10); printf("tile %4d x %4d  %s  order %d : %7.1f us  %.2f TB/s\n", TRv, TCv, W ? "16B" : " 8B", ord, ms * 1e3, bytes / ms / 1e9); }
    R(128, 128, false, 0) R(128, 128, false, 1) R(128, 128, true, 0) R(128, 128, true, 1)
    R(256, 64, false, 1) R(256, 64, true, 1) R(512, 32, true, 1) R(1024, 16, true, 1) R(64, 256, true, 0) R(8192, 2, true, 1)
    return 0;
}
