// Do stores and f64 MFMAs of DIFFERENT waves of one CU overlap at all?  One 512-thread workgroup per CU, no barriers, no LDS:
//   waves 0..3  "compute": NU x 64 v_mfma_f64_16x16x4 on registers (the Gram lab's unit), nothing else
//   waves 4..7  "store":   NU x 24 store instructions of 512 B / 1 KB in the Gram kernel's tile + mirrored-tile shape (64 KB per unit and
//                          workgroup), optionally with the multiquadric's f64 VALU work (16 radial functions per lane and unit) in front
// MODE bits: 1 compute waves on, 2 stores on, 4 radial VALU work in the store waves, 8 the VALU work in the COMPUTE waves instead
// Build: hipcc --offload-arch=gfx950 -O3 -o ovl ovl.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double mq_phi(double s, double a2) {
    const double t = fma(a2, s, 1.0);
    double y = __builtin_amdgcn_rsq(t);
    double g = t * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double e = fma(-g, g, t);
    return -fma(e, h, g);
}

// MODE bit 16: CU split -- even workgroups run only their compute waves (2 nu units), odd ones only their store waves (2 nu units)
// MODE bit 32: same-wave interleave -- all eight waves: per unit 32 MFMAs with 12 store instructions spread between them
// MODE bit 64: store waves at raised priority (s_setprio 3)
template <int MODE>
__global__ __launch_bounds__(512, 1) void ovl(double *__restrict__ Phi, long ld, int nu, double seed, double *sink, unsigned long long *clk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4, w = wave & 3;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    struct Stamp {
        unsigned long long *clk, c0, r0;
        int on;
        __device__ ~Stamp() {
            if (on) {
                clk[0] = __builtin_amdgcn_s_memtime() - c0;
                clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
            }
        }
    } stamp{clk, c0, r0, (blockIdx.x == 8 || blockIdx.x == 9) && (tid == 0 || tid == 256) ? 1 : 0};
    if (stamp.on) stamp.clk = clk + 2 * ((blockIdx.x & 1) * 2 + (tid >> 8));
    if (MODE & 16) {
        if (((blockIdx.x & 1) == 0) != (wave < 4)) return;
        nu *= 2;
    }
    if (MODE & 32) {
        v4d acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
        const double a = seed + lane, b = seed * 0.5 + lane;
        const long nt64 = ld / 64;
        const int hf = wave >> 2;
        const double v = seed + blockIdx.x;
        for (int u = 0; u < nu; ++u) {
            const long unit = (long)blockIdx.x * nu + u;
            const long ti = unit / nt64 % nt64, tj = unit % nt64;
            const long I0 = ti * 64, J0 = tj * 64;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                // 8 direct stores (one per k) + 4 mirrored (every other k)
                const int it = 2 * hf + (k >> 2), r = k & 3;
                Phi[(I0 + 16 * it + l4 + 4 * r) * ld + J0 + 16 * w + l15] = v + k;
                if (k & 1) {
                    const int it2 = 2 * hf + (k >> 2), t = (k >> 1) & 1;
                    const v2d vv = {v, v + k};
                    *(v2d *)(Phi + ((nt64 - 1 - tj) * 64 + 16 * w + t * 8 + (lane >> 3)) * ld + (nt64 - 1 - ti) * 64 + 16 * it2 + 2 * (lane & 7)) = vv;
                }
            }
        }
        double sacc = 0;
        for (int i = 0; i < 4; ++i) sacc += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (sacc == 123.456) sink[0] = sacc;
        return;
    }
    if (wave < 4) {
        if (!(MODE & 1)) return;
        v4d acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
        double a = seed + lane, b = seed * 0.5 + lane, extra = 0.0;
        for (int u = 0; u < nu; ++u) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (MODE & 8) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) extra += mq_phi(fabs(acc[i][r]) + u, 1.0);
            }
        }
        double s = extra;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (s == 123.456) sink[0] = s;
        return;
    }
    if (MODE & 64) __builtin_amdgcn_s_setprio(3);
    // store waves: units of this workgroup: tile row = blockIdx.x / 2 band of 64 rows, walking 64-column tiles; mirrored tile transposed
    const long nt64 = ld / 64;  // (ld = n here)
    double v = seed + blockIdx.x;
    for (int u = 0; u < nu; ++u) {
        const long unit = (long)((MODE & 16) ? (blockIdx.x >> 1) : blockIdx.x) * nu + u;
        const long ti = unit / nt64 % nt64, tj = unit % nt64;  // a plain rectangle walk: every 64 x 64 block of the matrix written twice in shape, once in bytes
        const long I0 = ti * 64, J0 = tj * 64;
        double vals[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) vals[e] = v + e;
        if (MODE & 4) {
#pragma unroll
            for (int e = 0; e < 16; ++e) vals[e] = mq_phi(vals[e] + u, 1.0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Phi[(I0 + 16 * it + l4 + 4 * r) * ld + J0 + 16 * w + l15] = vals[it * 4 + r];
            // the "mirrored" half of the bytes: 8 instructions of 16 B per lane (8 rows x 128 B), other half of the matrix rows
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const v2d vv = {vals[it * 4 + t], vals[it * 4 + t + 2]};
                    *(v2d *)(Phi + ((nt64 - 1 - tj) * 64 + 16 * w + t * 8 + (lane >> 3)) * ld + (nt64 - 1 - ti) * 64 + 16 * it + 2 * (lane & 7)) = vv;
                }
        } else {
            double s = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += vals[e];
            v += s * 1e-30;
        }
    }
    if (v == 123.456) sink[0] = v;
}

// store-only: every workgroup writes 64 KB as R rows x SEG doubles (R * SEG = 8192), consecutive workgroups side by side in a row band
template <int SEG>
__global__ __launch_bounds__(256, 4) void segstore(double *__restrict__ Phi, long n, double seed) {
    constexpr int R = 8192 / SEG;
    const long ncb = n / SEG, cb = blockIdx.x % ncb, rb = blockIdx.x / ncb;
    const v2d vv = {seed + blockIdx.x, seed};
    // thread t writes 16-byte pieces p = t, t + 256, ... of the block's 4096 pieces, piece p -> row p / (SEG/2), column pair p % (SEG/2)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int p = threadIdx.x + 256 * q, row = p / (SEG / 2), cp = p % (SEG / 2);
        *(v2d *)(Phi + (rb * R + row) * n + cb * SEG + 2 * cp) = vv;
    }
}
template <int SEG>
static void run_seg(double *Phi, long n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned grid = (unsigned)(n * n / 8192);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(segstore<SEG>, dim3(grid), dim3(256), 0, 0, Phi, n, 1.0);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(segstore<SEG>, dim3(grid), dim3(256), 0, 0, Phi, n, 2.0 + i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    printf("store-only, 64 KB per workgroup as %4d rows x %5d B segments: %8.1f us  %.2f TB/s\n", 8192 / SEG, SEG * 8, ms * 1e3, 8.0 * n * n / ms / 1e9);
    fflush(stdout);
}

template <int MODE>
static void run(double *Phi, long n, int nu, double *sink, const char *name) {
    static unsigned long long *clk = nullptr;
    if (!clk) hipMalloc(&clk, 64);
    hipMemset(clk, 0, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ovl<MODE>, dim3(256), dim3(512), 0, 0, Phi, n, nu, 1.0 + i, sink, clk);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(ovl<MODE>, dim3(256), dim3(512), 0, 0, Phi, n, nu, 2.0 + i, sink, clk);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (MODE & 2) ? 256.0 * nu * 65536.0 : 0.0, flops = (MODE & 1) ? 256.0 * 4 * nu * 64 * 2048.0 : 0.0;
    unsigned long long h[8];
    hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int q = 0; q < 4; ++q)
        if (h[2 * q + 1]) ghz = ghz > (double)h[2 * q] / h[2 * q + 1] * 0.1 ? ghz : (double)h[2 * q] / h[2 * q + 1] * 0.1;
    const bool inter = MODE & 32;
    const double by = (MODE & 2 || inter) ? 256.0 * nu * 65536.0 : 0.0, fl = (MODE & 1 || inter) ? 256.0 * 4 * nu * 64 * 2048.0 : 0.0;
    (void)bytes;
    (void)flops;
    printf("%-64s %8.1f us   stores %.2f TB/s   MFMA %.1f TFLOP/s   in-kernel clock %.2f GHz\n", name, ms * 1e3, by / ms / 1e9, fl / ms / 1e9, ghz);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const long n = 8192;
    const int nu = argc > 1 ? atoi(argv[1]) : 32;
    double *Phi, *sink;
    if (hipMalloc(&Phi, (size_t)n * n * 8) != hipSuccess) return 1;
    hipMalloc(&sink, 8);
    run<1>(Phi, n, nu, sink, "MFMA waves alone");
    run<2>(Phi, n, nu, sink, "store waves alone");
    run<3>(Phi, n, nu, sink, "MFMA waves + store waves");
    run<6>(Phi, n, nu, sink, "store waves with the radial VALU work, no MFMA waves");
    run<7>(Phi, n, nu, sink, "MFMA waves + store waves with the radial VALU work");
    run<5>(Phi, n, nu, sink, "MFMA waves + radial VALU work in the other waves, no stores");
    run<4>(Phi, n, nu, sink, "radial VALU work alone (other waves)");
    run<9>(Phi, n, nu, sink, "MFMA waves doing the radial VALU work themselves, no stores");
    run<11>(Phi, n, nu, sink, "MFMA + VALU in the compute waves, stores in the other waves");
    run<19>(Phi, n, nu, sink, "CU split: even workgroups MFMA only, odd workgroups stores only");
    run<32>(Phi, n, nu, sink, "same-wave interleave: 8 waves, a store behind every 4 MFMAs");
    run<67>(Phi, n, nu, sink, "MFMA waves + store waves at raised priority");
    run_seg<32>(Phi, n);
    run_seg<64>(Phi, n);
    run_seg<128>(Phi, n);
    run_seg<256>(Phi, n);
    run_seg<512>(Phi, n);
    run_seg<1024>(Phi, n);
    run_seg<8192>(Phi, n);
    return 0;
}
