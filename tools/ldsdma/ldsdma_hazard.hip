// Reproducer for the hazard noted at chol_mega.hip (run_panel): a workgroup reads a 128 x 128 f64 tile with plain loads (the lines are
// now in its CU's vector L1), overwrites it with plain stores, and reloads it by LDS-DMA (global_load_lds_dwordx4) -- does the DMA see
// the new data?  Each workgroup owns a tile; per iteration every thread stores elements that OTHER waves reload.  Modes:
//   0  stores -> s_waitcnt vmcnt(0) -> barrier -> DMA                (what window_part + gemm_acc_v2 do)
//   1  stores -> barrier -> DMA                                      (no wait for the stores: expected to fail)
//   2  mode 0 + buffer_inv sc1 in the loading waves before the DMA
//   3  mode 0 with agent-scope (sc1, write-through) stores
//   4  mode 0, but the reload by plain global loads instead of DMA   (control)
//   5  mode 1, reload by plain global loads                          (control: is the missing wait visible to plain loads as well?)
// Prints mismatching elements per mode.  Build: hipcc --offload-arch=gfx950 -O3 -o ldsdma_hazard ldsdma_hazard.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((address_space(1))) double gf64;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) v2d gv2d;

__device__ __forceinline__ double val(int it, int wg, int idx) { return (double)(it * 131 + wg) + 1e-5 * idx; }

template <int MODE>
__global__ __launch_bounds__(256) void hazard(double *__restrict__ A, int iters, unsigned long long *bad, double *sink) {
    __shared__ __attribute__((aligned(16))) double T[128 * 130];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *tile = A + (size_t)blockIdx.x * 128 * 128;  // column-major, ld = 128
    double keep = 0.0;
    unsigned long long nbad = 0;
    for (int it = 1; it <= iters; ++it) {
        // phase A: plain loads of the whole tile (the MFMA-layout read of store_tile<SUB>): thread reads rows lane*2.., column tid>>6 + 4u
        for (int u = 0; u < 32; ++u) {
            const int col = wave + 4 * u;
            const v2d v = *(const gv2d *)(tile + (size_t)col * 128 + 2 * lane);
            keep += v.x + v.y;
        }
        // phase B: overwrite -- thread (lane, wave) writes ROW-wise pieces: row = tid & 127, columns 64 * (tid >> 7) .. + 63 (8-byte stores,
        // like the accumulator layout's epilogue), so that a column reloaded by one wave was written by lanes of all four waves... of two
        for (int c = 0; c < 64; ++c) {
            const int row = tid & 127, col = 64 * (tid >> 7) + c;
            const double v = val(it, blockIdx.x, col * 128 + row);
            if (MODE == 3)
                __hip_atomic_store((gf64 *)(tile + (size_t)col * 128 + row), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                *(gf64 *)(tile + (size_t)col * 128 + row) = v;
        }
        if (MODE != 1 && MODE != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE == 2) {
            asm volatile("buffer_inv sc1" ::: "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // phase C: reload: wave w brings columns w, w + 4, ... (lane l rows 2l, 2l+1), by DMA or by plain loads through registers
        if (MODE == 4 || MODE == 5) {
            for (int u = 0; u < 32; ++u) {
                const int col = wave + 4 * u;
                const v2d v = *(const gv2d *)(tile + (size_t)col * 128 + 2 * lane);
                *(v2d *)&T[col * 130 + 2 * lane] = v;
            }
        } else {
            for (int u = 0; u < 32; ++u) {
                const int col = wave + 4 * u;
                __builtin_amdgcn_global_load_lds((glb_void *)(tile + (size_t)col * 128 + 2 * lane), (lds_void *)&T[col * 130], 16, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int e = tid; e < 128 * 128; e += 256) {
            const int col = e >> 7, row = e & 127;
            if (T[col * 130 + row] != val(it, blockIdx.x, col * 128 + row)) ++nbad;
        }
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
    if (keep == 123.456) sink[0] = keep;
}

template <int MODE>
static void run(double *A, int grid, int iters, unsigned long long *dbad, double *sink, const char *what) {
    hipMemset(dbad, 0, 8);
    hipMemset(A, 0, (size_t)grid * 128 * 128 * 8);
    hipLaunchKernelGGL(hazard<MODE>, dim3(grid), dim3(256), 0, 0, A, iters, dbad, sink);
    unsigned long long h = 0;
    hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost);
    printf("mode %d  %-62s mismatching elements: %llu of %llu\n", MODE, what, h, (unsigned long long)grid * iters * 128 * 128);
}

int main(int argc, char **argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 200;
    double *A, *sink;
    unsigned long long *dbad;
    if (hipMalloc(&A, (size_t)grid * 128 * 128 * 8) != hipSuccess) return 1;
    hipMalloc(&sink, 8);
    hipMalloc(&dbad, 8);
    run<0>(A, grid, iters, dbad, sink, "stores, vmcnt(0), barrier, LDS-DMA");
    run<1>(A, grid, iters, dbad, sink, "stores, barrier (no wait), LDS-DMA");
    run<2>(A, grid, iters, dbad, sink, "stores, vmcnt(0), barrier, buffer_inv sc1, LDS-DMA");
    run<3>(A, grid, iters, dbad, sink, "sc1 stores, vmcnt(0), barrier, LDS-DMA");
    run<4>(A, grid, iters, dbad, sink, "stores, vmcnt(0), barrier, plain loads (control)");
    run<5>(A, grid, iters, dbad, sink, "stores, barrier (no wait), plain loads (control)");
    return 0;
}
