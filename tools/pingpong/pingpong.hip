// Latency of a one-word hand-off between two workgroups on gfx950, by placement (same XCD / different XCDs) and by the cache
// policy bits of the store and of the polling load.  Two single-wave workgroups bounce a counter N times; every wait is bounded
// by wall-clock time, so a combination that never becomes visible reports "stale" instead of hanging.
//   hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SV>
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) {
    if constexpr (SV == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (SV == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (SV == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (SV == 3) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (SV == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}
template <int LV>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) {
    unsigned long long v;
    if constexpr (LV == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LV == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LV == 2) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LV == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if constexpr (LV == 4 || LV == 5) {
        // scalar load (lgkmcnt, not vmcnt): the address must be wave-uniform -- lane 0's pointer
        // (readfirstlane returns int: widen through unsigned, or a set bit 31 sign-extends into the upper half of the address)
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)p);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)p >> 32));
        const unsigned long long a = ((unsigned long long)hi << 32) | (unsigned long long)lo;
        unsigned long long sv;
        if constexpr (LV == 4) asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(a) : "memory");
        if constexpr (LV == 5) asm volatile("s_dcache_inv\n s_load_dwordx2 %0, %1, 0x0\n s_waitcnt lgkmcnt(0)" : "=s"(sv) : "s"(a) : "memory");
        v = sv;
    }
    return v;
}

// lanes: how many lanes of the wave take part (each with its own 8-byte word of the line(s)): 1 or 64
template <int SV, int LV>
__global__ void pingpong(unsigned long long *buf, int a, int b, int n, int lanes, long long *out, int *xcc) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(id & 0xf);
    if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
    if ((int)threadIdx.x >= lanes) return;
    unsigned long long *ping = buf + threadIdx.x, *pong = buf + 1024 + threadIdx.x;
    const bool first = (int)blockIdx.x == a;
    const long long limit = 2000000;  // 20 ms of the 100 MHz clock
    const long long t0 = wall_clock64();
    bool stale = false;
    for (int r = 1; r <= n && !stale; ++r) {
        if (first) st<SV>(ping, (unsigned long long)r);
        const unsigned long long *w = first ? pong : ping;
        int spins = 0;
        while (ld<LV>(w) != (unsigned long long)r) {
            if ((++spins & 255) == 0 && wall_clock64() - t0 > limit) { stale = true; break; }
        }
        if (!first && !stale) st<SV>(pong, (unsigned long long)r);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[first ? 0 : 1] = stale ? -1 : (t1 - t0);
}

template <int SV, int LV>
void run(const char *sname, const char *lname, unsigned long long *buf, long long *out, int *xcc, int nblocks) {
    const int n = 2000;
    for (int lanes : {1, 64}) {
        for (int place = 0; place < 2; ++place) {
            const int a = 0, b = place == 0 ? 8 : 1;
            CK(hipMemset(buf, 0, 4096 * sizeof(unsigned long long)));
            CK(hipMemset(out, 0, 2 * sizeof(long long)));
            hipLaunchKernelGGL((pingpong<SV, LV>), dim3(nblocks), dim3(64), 0, 0, buf, a, b, n, lanes, out, xcc);
            CK(hipDeviceSynchronize());
            long long h[2];
            std::vector<int> hx(nblocks);
            CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
            CK(hipMemcpy(hx.data(), xcc, nblocks * sizeof(int), hipMemcpyDeviceToHost));
            if (h[0] < 0 || h[1] < 0)
                printf("store %-8s load %-8s lanes %2d  blocks %d,%d (XCC %d,%d): STALE (never seen within 20 ms)\n", sname, lname, lanes, a, b, hx[a], hx[b]);
            else
                printf("store %-8s load %-8s lanes %2d  blocks %d,%d (XCC %d,%d): %.0f ns per one-way hand-off\n", sname, lname, lanes, a, b, hx[a],
                       hx[b], (double)h[0] * 10.0 / (2.0 * n));
            fflush(stdout);
        }
    }
}

int main() {
    unsigned long long *buf;
    long long *out;
    int *xcc;
    const int nblocks = 64;
    CK(hipMalloc(&buf, 4096 * sizeof(unsigned long long)));
    CK(hipMalloc(&out, 2 * sizeof(long long)));
    CK(hipMalloc(&xcc, nblocks * sizeof(int)));
    run<1, 0>("sc1", "sc1", buf, out, xcc, nblocks);
    run<2, 1>("sc0sc1", "sc0sc1", buf, out, xcc, nblocks);
    run<0, 0>("plain", "sc1", buf, out, xcc, nblocks);
    run<0, 1>("plain", "sc0sc1", buf, out, xcc, nblocks);
    run<3, 0>("nt", "sc1", buf, out, xcc, nblocks);
    run<4, 0>("sc0", "sc1", buf, out, xcc, nblocks);
    run<0, 2>("plain", "nt", buf, out, xcc, nblocks);
    run<1, 2>("sc1", "nt", buf, out, xcc, nblocks);
    run<1, 1>("sc1", "sc0sc1", buf, out, xcc, nblocks);
    run<0, 3>("plain", "sc0", buf, out, xcc, nblocks);
    run<1, 4>("sc1", "s_load glc", buf, out, xcc, nblocks);
    run<0, 4>("plain", "s_load glc", buf, out, xcc, nblocks);
    run<1, 5>("sc1", "dcache_inv+s_load", buf, out, xcc, nblocks);
    run<0, 5>("plain", "dcache_inv+s_load", buf, out, xcc, nblocks);
    return 0;
}
