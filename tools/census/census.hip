// Residency census: which hardware ids do the workgroups of a 512 x 256-thread launch (76 KB LDS each) land on?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void census(unsigned *out, unsigned *arrived, int total) {
    __shared__ double pad[9500];
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, offset 0, size 32
        unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        pad[threadIdx.x] = hw;
        atomicAdd(arrived, 1u);
        // stay resident until everyone has arrived (bounded)
        for (int it = 0; it < 200000; ++it) {
            if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)total) break;
            __builtin_amdgcn_s_sleep(32);
        }
        out[2 * gridDim.x + blockIdx.x] = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (pad[0] == 12345.678) out[0] = 0;
}
int main(int argc, char **argv) {
    int grid = argc > 1 ? atoi(argv[1]) : 512;
    unsigned *d, *arr;
    hipMalloc(&d, sizeof(unsigned) * grid * 3);
    hipMalloc(&arr, 4);
    hipMemset(arr, 0, 4);
    hipLaunchKernelGGL(census, dim3(grid), dim3(256), 0, 0, d, arr, grid);
    hipDeviceSynchronize();
    std::vector<unsigned> h(grid * 3);
    hipMemcpy(h.data(), d, sizeof(unsigned) * grid * 3, hipMemcpyDeviceToHost);
    std::map<unsigned long long, int> cnt;
    int full = 0;
    for (int b = 0; b < grid; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1];
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        unsigned long long key = ((unsigned long long)(xcc & 0xf) << 16) | (se << 8) | (sh << 4) | cu;
        cnt[key]++;
        if (h[2 * grid + b] >= (unsigned)grid) ++full;
        if (b < 24) printf("block %d hw=%08x xcc=%08x -> xcc %u se %u sh %u cu %u simd %u wave %u\n", b, hw, xcc, xcc & 0xf, se, sh, cu, (hw >> 4) & 3, hw & 0xf);
    }
    std::map<int, int> hist;
    for (auto &kv : cnt) hist[kv.second]++;
    printf("grid %d: distinct (xcc,se,sh,cu) = %zu; co-resident-all %d\n", grid, cnt.size(), full);
    for (auto &kv : hist) printf("  %d CUs host %d workgroups\n", kv.second, kv.first);
    return 0;
}
