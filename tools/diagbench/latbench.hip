// Dependent-issue latency of the f64 instructions on the leaf's pivot chain (one wave alone on a CU), cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o latbench latbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(64) void k(double *out, unsigned long long *ts, int reps, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (V == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            if constexpr (V == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
            if constexpr (V == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
            if constexpr (V == 3) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x));
            if constexpr (V == 4) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));
            if constexpr (V == 5) {
                int lo, hi;
                asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(x)), "v"(__double2hiint(x)));
                asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x) : "s"(__hiloint2double(hi, lo)), "v"(y));
            }
            if constexpr (V == 6) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(x) : "v"(y));  // independent-ish accumulate (same dst)
            if constexpr (V == 7) {  // 4 independent chains
                static_assert(true, "");
                asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ts[V] = t1 - t0;
}
__global__ __launch_bounds__(64) void k4(double *out, unsigned long long *ts, int reps, double seed) {
    double x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, y = 1.0000001;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x0) : "v"(y));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x1) : "v"(y));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x2) : "v"(y));
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x3) : "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3;
    if (threadIdx.x == 0) ts[7] = t1 - t0;
}

// issue rate: 8 independent accumulators per instruction kind
template <int V>
__global__ __launch_bounds__(64) void kt(double *out, unsigned long long *ts, int reps, double seed) {
    double x[8], y = 1.0000001, m = seed + threadIdx.x * 1e-9;
    for (int i = 0; i < 8; ++i) x[i] = seed + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (V == 0) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(m), "v"(y));
                if constexpr (V == 1) {
                    int lo, hi;
                    asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3" : "=s"(lo), "=s"(hi) : "v"(__double2loint(m)), "v"(__double2hiint(m)));
                    asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(x[i]) : "s"(__hiloint2double(hi, lo)), "v"(y));
                }
                if constexpr (V == 2) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(y));
                if constexpr (V == 3) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(m));
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) ts[8 + V] = t1 - t0;
}
int main() {
    double *o; unsigned long long *ts;
    (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&ts, 128); (void)hipMemset(ts, 0, 128);
    const int reps = 2000;
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(kt<0>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(kt<1>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(kt<2>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    hipLaunchKernelGGL(kt<3>, dim3(1), dim3(64), 0, 0, o, ts, reps, 0.5);
    (void)hipDeviceSynchronize();
    unsigned long long h[16];
    (void)hipMemcpy(h, ts, 128, hipMemcpyDeviceToHost);
    const char *nt[4] = {"v_fmac_f64_dpp, independent", "2 v_readlane + v_fma_f64 (sgpr), independent", "v_fmac_f64, independent", "v_mov_b64_dpp, independent"};
    for (int v = 0; v < 4; ++v) printf("%-52s %.1f cycles\n", nt[v], (double)h[8 + v] / reps / 16);
    const char *nm[8] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rsq_f64 dependent", "s_nop 1 + v_mov_b64_dpp dependent", "s_nop 1 + v_fmac_f64_dpp dependent",
                         "2 v_readlane + v_mul_f64 (sgpr) dependent", "v_fmac_f64 same accumulator", "v_fma_f64, 4 independent chains (per instruction)"};
    for (int v = 0; v < 8; ++v) printf("%-52s %.1f cycles\n", nm[v], (double)h[v] / reps / 16);
    return 0;
}
