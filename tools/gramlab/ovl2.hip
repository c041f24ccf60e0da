// Round 6: the two measurements the round-5 review found missing in profiles/r05_gram_overlap.txt (full-matrix Gram assembly,
// mrbf_gram = RBF.get_matrices, /root/reference/src/models/RbfModel.jl:374-375; n = 8192, d = 64: 8 n^2 = 537 MB, n^2 d / 2 = 2.1e9 FMAs).
//
// (a) Do f64 VALU FMAs (not MFMAs) of compute waves overlap with the stores of other waves of the same CU?
//     One 512-thread workgroup per CU as in ovl.hip; waves 4..7 issue the Gram kernel's stores (tile + mirrored tile), waves 0..3 compute
//     the tile's distance products as v_fma_f64:
//       V1  register-tiled 8 x 8 per lane (a wave = one 64 x 64 tile, k = 64), operands from LDS (conflict-free ds_read_b128 layout)
//       V2  lane = row, 16 columns at a time, the column block's coordinates as SCALAR operands (s_load -> SGPR source of v_fma_f64): no LDS
//     each alone, with the multiquadric's radial function behind it, and beside the store waves.
// (b) Why do the same bytes take 75 us in ovl's store waves and 99-109 us on the shipped grid?  Store-only kernels that differ in ONE
//     thing at a time: store waves per CU (4 / 8 / 16), persistent walk vs one workgroup per unit, rectangle walk vs the triangle's
//     tile pairs, 8-byte vs 16-byte direct stores.
// Build: hipcc --offload-arch=gfx950 -O3 -o ovl2 ovl2.hip        Run: ./ovl2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

// the store waves' unit of ovl.hip: one 64 x 64 tile (rows of 512 B by four waves) + its mirrored tile (16-byte pieces), 64 KB
__device__ __forceinline__ void store_unit(double *__restrict__ Phi, long ld, long ti, long tj, long nt64, int w, int lane, const double *vals) {
    const int l15 = lane & 15, l4 = lane >> 4;
    const long I0 = ti * 64, J0 = tj * 64;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) Phi[(I0 + 16 * it + l4 + 4 * r) * ld + J0 + 16 * w + l15] = vals[it * 4 + r];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const v2d vv = {vals[it * 4 + t], vals[it * 4 + t + 2]};
            *(v2d *)(Phi + ((nt64 - 1 - tj) * 64 + 16 * w + t * 8 + (lane >> 3)) * ld + (nt64 - 1 - ti) * 64 + 16 * it + 2 * (lane & 7)) = vv;
        }
}

// MODE bits: 1 = V1 compute waves, 2 = store waves, 4 = radial function behind the FMAs (in the compute waves), 8 = V2 compute waves
template <int MODE, int NCW = 4>
__global__ __launch_bounds__(64 * (NCW + 4), 1) void ovl2(double *__restrict__ Phi, long ld, int nu, double seed, double *sink, const double *__restrict__ cols) {
    __shared__ double sa[64 * 64], sb[64 * 64];  // [k][piece p = 0..3][r = 0..7][2]: a lane's four 16-byte reads per k are conflict-free
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (MODE & 1) {
        for (int i = tid; i < 64 * 64; i += 64 * (NCW + 4)) {
            sa[i] = seed + 1e-3 * i;
            sb[i] = seed - 1e-3 * i;
        }
        __syncthreads();
    }
    if (wave < NCW) {
        if (MODE & 1) {
            // V1: 4 waves x nu / 4 tiles (one tile per unit is all the symmetric kernel computes: the mirrored tile is the same numbers)
            const int r = lane >> 3, c = lane & 7;
            double total = 0.0;
            for (int u = wave; u < nu; u += NCW) {
                double acc[8][8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0;
#pragma unroll 4
                for (int k = 0; k < 64; ++k) {
                    double a[8], b[8];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const v2d va = *(const v2d *)(sa + k * 64 + p * 16 + r * 2), vb = *(const v2d *)(sb + k * 64 + p * 16 + c * 2);
                        a[2 * p] = va[0];
                        a[2 * p + 1] = va[1];
                        b[2 * p] = vb[0];
                        b[2 * p + 1] = vb[1];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
                }
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s += (MODE & 4) ? mq_phi(fabs(acc[i][j]) + u, 1.0) : acc[i][j];
                total += s;
            }
            if (total == 123.456) sink[0] = total;
            return;
        }
        if (MODE & 8) {
            // V2: lane = row of a 64-row band (its 64 coordinates in registers), columns in groups of 16 with scalar coordinates
            double x[64];
#pragma unroll
            for (int k = 0; k < 64; ++k) x[k] = seed + lane + 0.01 * k;
            double total = 0.0;
            for (int u = wave; u < nu; u += NCW) {
                for (int jb = 0; jb < 4; ++jb) {
                    const double *cj = cols + ((size_t)((blockIdx.x * 7 + u * 4 + jb) & 127)) * 16 * 64;  // wave-uniform: scalar loads
                    double acc[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j] = 0.0;
#pragma unroll
                    for (int k = 0; k < 64; ++k)  // ([k][j] layout: the 16 columns' k-th coordinates are one scalar load, 16 independent chains)
#pragma unroll
                        for (int j = 0; j < 16; ++j) acc[j] = fma(x[k], cj[k * 16 + j], acc[j]);
                    double s = 0.0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) s += (MODE & 4) ? mq_phi(fabs(acc[j]) + u, 1.0) : acc[j];
                    total += s;
                }
            }
            if (total == 123.456) sink[0] = total;
            return;
        }
        return;
    }
    if (!(MODE & 2)) return;
    const long nt64 = ld / 64;
    double v = seed + blockIdx.x;
    for (int u = 0; u < nu; ++u) {
        const long unit = (long)blockIdx.x * nu + u;
        double vals[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) vals[e] = v + e;
        if (MODE & 16) {  // every byte written once: tiles of the top half, "mirrors" into the bottom half (the rectangle walk of ovl.hip writes the
                          // top right quadrant twice -- a quarter of its bytes hit lines that are still in the Infinity Cache)
            const long ti = unit / nt64 % nt64, tj = unit % nt64;
            store_unit(Phi, ld, ti, tj, nt64, wave & 3, lane, vals);  // (direct part only matters; see so2<., ., 2> for the clean split)
        } else {
            store_unit(Phi, ld, unit / nt64 % nt64, unit % nt64, nt64, wave & 3, lane, vals);
        }
    }
    if (v == 123.456) sink[0] = v;
}

// (b) store-only variants.  WPC = store waves per workgroup (one workgroup per CU for PERSIST = 1), every group of four waves works on one
// unit.  PERSIST 1: 256 workgroups walk nu units each; 0: one workgroup of 256 threads per unit (grid = units), four per CU.
// WALK 0: ovl's rectangle walk; 1: the lower triangle's tile pairs (tile (i, j) and its mirror (j, i), j <= i, 64 x 64 tiles, row by row).
template <int WPC, int PERSIST, int WALK, int PART = 3>
__global__ __launch_bounds__(WPC * 64, PERSIST ? 1 : 4) void so2(double *__restrict__ Phi, long ld, long units, double seed, double *sink, long uoff = 0, int ilv = 0) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, ngrp = WPC / 4;
    const long nt64 = 128;  // (n = 8192; ld may be padded)
    double v = seed + blockIdx.x;
    const long per = PERSIST ? (units + gridDim.x - 1) / gridDim.x : 1;
    const long u0 = (long)blockIdx.x * per;
    for (long uu = grp; uu < per; uu += ngrp) {
        long unit = u0 + uu;
        if (unit >= units) break;
        unit += uoff;
        long ti, tj;
        if (WALK == 0) {
            ti = unit / nt64 % nt64;
            tj = unit % nt64;
        } else if (WALK == 10) {
            // LEVEL-ORDERED walk of the symmetric matrix: the triangle is the off-diagonal square of nt/2 x nt/2 tiles, two of nt/4 x nt/4,
            // four of nt/8 x nt/8, ..., nt/2 of 1 x 1, and the diagonal.  Level by level, squares of a level side by side, each square row
            // by row: at any time the tiles being written and their transposed tiles lie in different row bands AND column bands
            long idx = unit, sq = nt64 / 2, lvl = 0;
            while (sq >= 1 && idx >= (nt64 / 2) * sq) {
                idx -= (nt64 / 2) * sq;
                sq >>= 1;
                ++lvl;
            }
            if (sq >= 1) {
                const long nsq = (nt64 / 2) / sq;
                const long q = ilv ? idx % nsq : idx / (sq * sq), within = ilv ? idx / nsq : idx % (sq * sq);
                ti = q * 2 * sq + sq + within / sq;
                tj = q * 2 * sq + within % sq;
            } else {
                ti = tj = idx;  // the diagonal tiles
            }
        } else if (WALK == 8 || WALK == 9) {
            // WALK 8: ONE off-diagonal square of the symmetric matrix -- tiles (ti, tj), ti in [nt/2, nt), tj in [0, nt/2), row by row, and
            // their transposed tiles in the opposite corner (half of all bytes).  WALK 9: row bands of nt/2 tiles over the left half, the
            // mirror-shape stores at the transposed-like position in the right half (all bytes, rows half as long as the rectangle walk's)
            const long h = nt64 / 2;
            ti = unit / h + (WALK == 8 ? h : 0);
            tj = unit % h;
        } else if (WALK == 3) {
            // CYCLIC walk of the symmetric matrix: row band ti pairs with the 64 column blocks behind it, tj = ti - s (mod nt), s = 0..63;
            // the pairs at cyclic distance nt / 2 once (ti < nt / 2): every unordered tile pair exactly once, every row band the same
            // number of units, and the tiles written at any one time spread over the whole width of the matrix
            if (unit < nt64 * 64) {
                ti = unit / 64;
                tj = (ti - unit % 64 + nt64) % nt64;
            } else {
                ti = unit - nt64 * 64;
                tj = ti + nt64 / 2;
            }
        } else {  // unit -> (ti, tj), tj <= ti, rows in order: ti = floor((sqrt(8 u + 1) - 1) / 2)
            ti = (long)((sqrt(8.0 * (double)unit + 1.0) - 1.0) * 0.5);
            while (ti * (ti + 1) / 2 > unit) --ti;
            while ((ti + 1) * (ti + 2) / 2 <= unit) ++ti;
            tj = unit - ti * (ti + 1) / 2;
        }
        double vals[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) vals[e] = v + e;
        if (WALK == 4) {
            // the triangle's tile pairs with the mirrored tile written ONE UNIT LATER than its tile (the pair's two streams -- transposed
            // addresses on a power-of-two pitch -- are then never in flight together)
            const int l15 = lane & 15, l4 = lane >> 4, w = wave & 3;
            long t2 = (long)((sqrt(8.0 * (double)unit + 1.0) - 1.0) * 0.5);
            while (t2 * (t2 + 1) / 2 > unit) --t2;
            while ((t2 + 1) * (t2 + 2) / 2 <= unit) ++t2;
            const long tj2 = unit - t2 * (t2 + 1) / 2;
            const long I0 = t2 * 64, J0 = tj2 * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Phi[(I0 + 16 * it + l4 + 4 * r) * ld + J0 + 16 * w + l15] = vals[it * 4 + r];
            const long up = unit - ngrp >= u0 ? unit - ngrp : unit + (per / ngrp - 1) * ngrp;  // previous unit of this wave group (wraps inside the workgroup's range)
            if (up < units) {
                long t3 = (long)((sqrt(8.0 * (double)up + 1.0) - 1.0) * 0.5);
                while (t3 * (t3 + 1) / 2 > up) --t3;
                while ((t3 + 1) * (t3 + 2) / 2 <= up) ++t3;
                const long tj3 = up - t3 * (t3 + 1) / 2;
                if (t3 != tj3) {
#pragma unroll
                    for (int it = 0; it < 4; ++it)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const v2d vv = {vals[it * 4 + t], vals[it * 4 + t + 2]};
                            *(v2d *)(Phi + (tj3 * 64 + 16 * w + t * 8 + (lane >> 3)) * ld + t3 * 64 + 16 * it + 2 * (lane & 7)) = vv;
                        }
                }
            }
            continue;
        }
        if (WALK == 0) {
            store_unit(Phi, ld, ti, tj, nt64, wave & 3, lane, vals);
        } else if (WALK == 2) {
            // the rectangle walk with every byte written ONCE: tile (ti, tj) in the top half, the mirrored-shape stores into the bottom half
            const int l15 = lane & 15, l4 = lane >> 4, w = wave & 3;
            const long I0 = ti * 64, J0 = tj * 64;
            if (PART & 1) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Phi[(I0 + 16 * it + l4 + 4 * r) * ld + J0 + 16 * w + l15] = vals[it * 4 + r];
            }
            const long Ib = (64 + (tj & 63)) * 64, Jb = (ti + 64 * (tj >> 6)) * 64;
            if (PART & 2)
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const v2d vv = {vals[it * 4 + t], vals[it * 4 + t + 2]};
                    *(v2d *)(Phi + (Ib + 16 * w + t * 8 + (lane >> 3)) * ld + Jb + 16 * it + 2 * (lane & 7)) = vv;
                }
        } else {
            // tile (ti, tj) in rows of 512 B, mirrored tile (tj, ti) in 16-byte pieces: the shipped kernel's shape on the real triangle
            const int l15 = lane & 15, l4 = lane >> 4, w = wave & 3;
            const long I0 = ti * 64, J0 = tj * 64;
            if (PART & 1) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int r = 0; r < 4; ++r) Phi[(I0 + 16 * it + l4 + 4 * r) * ld + J0 + 16 * w + l15] = vals[it * 4 + r];
            }
            if ((ti != tj || WALK == 9) && (PART & 2)) {
                // WALK 5: the mirror-shape stores at the POINT-REFLECTED tile (nt-1-ti, nt-1-tj) -- upper triangle too, but marching along a row
                // band like the tiles; WALK 6: at the transposed tile of a SECOND buffer; WALK 7: transposed tile, rows walked bottom-up
                long Jm = (WALK == 5 ? (nt64 - 1 - ti) : tj) * 64, Im = (WALK == 5 ? (nt64 - 1 - tj) : ti) * 64;
                if (WALK == 9) {
                    Jm = (tj + 64 * (ti >> 6)) * 64;
                    Im = (64 + (ti & 63)) * 64;
                }
                double *P2 = WALK == 6 ? Phi + (size_t)8192 * (8192 + 64) : Phi;
#pragma unroll
                for (int it = 0; it < 4; ++it)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const v2d vv = {vals[it * 4 + t], vals[it * 4 + t + 2]};
                        *(v2d *)(P2 + (Jm + 16 * w + t * 8 + (lane >> 3)) * ld + Im + 16 * it + 2 * (lane & 7)) = vv;
                    }
            }
        }
    }
    if (v == 123.456) sink[0] = v;
}

template <int MODE, int NCW = 4>
static float run(double *Phi, long n, int nu, double *sink, const double *cols, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ovl2<MODE, NCW>), dim3(256), dim3(64 * (NCW + 4)), 0, 0, Phi, n, nu, 1.0 + i, sink, cols);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ovl2<MODE, NCW>), dim3(256), dim3(64 * (NCW + 4)), 0, 0, Phi, n, nu, 2.0 + i, sink, cols);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double by = (MODE & 2) ? 256.0 * nu * 65536.0 : 0.0, fma_ = (MODE & 9) ? 256.0 * nu * 64.0 * 64 * 64 : 0.0;
    printf("%-76s %8.1f us   stores %.2f TB/s   VALU %.1f TFLOP/s (f64 FMA)\n", name, ms * 1e3, by / ms / 1e9, 2.0 * fma_ / ms / 1e9);
    fflush(stdout);
    return ms;
}

template <int WPC, int PERSIST, int WALK, int PART = 3>
static void run_so(double *Phi, long n, double *sink, const char *name, long ld = 0) {
    const long nt64 = n / 64;
    if (ld == 0) ld = n;
    const long units = (WALK == 0 || WALK == 2 || WALK == 9) ? nt64 * nt64 / 2 : (WALK == 8 ? nt64 * nt64 / 4 : nt64 * (nt64 + 1) / 2);
    (void)0;  // 64 KB (rectangle walk) / a tile pair each: n^2 * 8 bytes in all
    const unsigned grid = PERSIST ? 256u : (unsigned)units;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((so2<WPC, PERSIST, WALK, PART>), dim3(grid), dim3(WPC * 64), 0, 0, Phi, ld, units, 1.0 + i, sink);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((so2<WPC, PERSIST, WALK, PART>), dim3(grid), dim3(WPC * 64), 0, 0, Phi, ld, units, 2.0 + i, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = 8.0 * n * n * (PART == 3 ? 1.0 : 0.5) * (WALK == 8 ? 0.5 : 1.0);
    printf("%-76s %8.1f us   %.2f TB/s  (%.3f of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
    fflush(stdout);
}

template <int WPC, int PERSIST>
static void run_level(double *Phi, long n, double *sink, long uoff, long units, double bytes, int ilv, const char *name) {
    const unsigned grid = PERSIST ? 256u : (unsigned)units;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((so2<WPC, PERSIST, 10, 3>), dim3(grid), dim3(WPC * 64), 0, 0, Phi, n, units, 1.0 + i, sink, uoff, ilv);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((so2<WPC, PERSIST, 10, 3>), dim3(grid), dim3(WPC * 64), 0, 0, Phi, n, units, 2.0 + i, sink, uoff, ilv);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("%-76s %8.1f us   %.2f TB/s  (%.3f of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const long n = 8192;
    const int nu = argc > 1 ? atoi(argv[1]) : 32;
    double *Phi, *sink, *cols;
    if (hipMalloc(&Phi, (size_t)2 * n * (n + 64) * 8) != hipSuccess) return 1;
    hipMalloc(&sink, 8);
    hipMalloc(&cols, 128 * 16 * 64 * 8);
    hipMemset(cols, 0, 128 * 16 * 64 * 8);
    printf("(a) f64 VALU compute waves beside store waves; one 512-thread workgroup per CU, n = 8192, d = 64, %d units per workgroup\n", nu);
    run<2>(Phi, n, nu, sink, cols, "store waves alone (ovl.hip's line: 75.0 us)");
    run<1>(Phi, n, nu, sink, cols, "V1 (8 x 8 per lane, operands from LDS) alone");
    run<5>(Phi, n, nu, sink, cols, "V1 + radial function alone");
    run<3>(Phi, n, nu, sink, cols, "V1 + store waves");
    run<7>(Phi, n, nu, sink, cols, "V1 + radial function + store waves");
    run<8>(Phi, n, nu, sink, cols, "V2 (lane = row, scalar column operands, no LDS) alone");
    run<12>(Phi, n, nu, sink, cols, "V2 + radial function alone");
    run<10>(Phi, n, nu, sink, cols, "V2 + store waves");
    run<14>(Phi, n, nu, sink, cols, "V2 + radial function + store waves");
    run<1, 8>(Phi, n, nu, sink, cols, "V1 on EIGHT compute waves (two per SIMD) alone");
    run<5, 8>(Phi, n, nu, sink, cols, "V1 on eight compute waves + radial function alone");
    run<3, 8>(Phi, n, nu, sink, cols, "V1 on eight compute waves + store waves");
    run<7, 8>(Phi, n, nu, sink, cols, "V1 on eight compute waves + radial function + store waves");
    printf("(b) store-only, n^2 * 8 = 537 MB: what separates ovl's 75 us from the shipped grid's 99-109 us\n");
    run_so<4, 1, 0>(Phi, n, sink, "persistent, 4 store waves / CU, rectangle walk (= ovl's store waves)");
    run_so<8, 1, 0>(Phi, n, sink, "persistent, 8 store waves / CU, rectangle walk");
    run_so<16, 1, 0>(Phi, n, sink, "persistent, 16 store waves / CU, rectangle walk");
    run_so<4, 0, 0>(Phi, n, sink, "one 256-thread workgroup per unit (grid = 8192, 4 / CU), rectangle walk");
    run_so<4, 1, 2>(Phi, n, sink, "persistent, 4 store waves / CU, rectangle walk with EVERY BYTE WRITTEN ONCE");
    run_so<4, 0, 2>(Phi, n, sink, "one workgroup per unit, rectangle walk with every byte written once");
    run_so<4, 1, 3>(Phi, n, sink, "persistent, 4 store waves / CU, CYCLIC walk of the tile pairs (tile + mirror)");
    run_so<4, 0, 3>(Phi, n, sink, "one workgroup per tile pair, CYCLIC walk (row band ti x the 64 column blocks behind it)");
    run_so<4, 1, 1>(Phi, n, sink, "persistent, 4 store waves / CU, the triangle's tile pairs (tile + mirror)");
    run_so<4, 1, 1>(Phi, n, sink, "  the same on a pitch of n + 16 doubles (no power-of-two pitch)", n + 16);
    run_so<4, 0, 1>(Phi, n, sink, "  one workgroup per tile pair on a pitch of n + 16 doubles", n + 16);
    run_so<4, 1, 0>(Phi, n, sink, "  (rectangle walk on a pitch of n + 16 doubles)", n + 16);
    run_so<4, 0, 2, 1>(Phi, n, sink, "rectangle walk, one workgroup per unit: TILES only (half the bytes)");
    run_so<4, 0, 2, 2>(Phi, n, sink, "rectangle walk, one workgroup per unit: MIRROR-shape stores only");
    run_so<4, 0, 1, 1>(Phi, n, sink, "triangle walk, one workgroup per tile pair: TILES only");
    run_so<4, 0, 1, 2>(Phi, n, sink, "triangle walk, one workgroup per tile pair: MIRRORS only");
    run_so<4, 0, 5>(Phi, n, sink, "triangle walk, mirror-shape stores at the POINT-REFLECTED tile (both streams along row bands)");
    run_so<4, 1, 5>(Phi, n, sink, "  the same, persistent");
    run_so<4, 0, 6>(Phi, n, sink, "triangle walk, mirrors at the transposed tile of a SECOND buffer");
    run_so<4, 0, 8>(Phi, n, sink, "ONE off-diagonal square (64 x 64 tiles) + its transposed tiles, one workgroup per pair (half the bytes)");
    run_so<4, 1, 8>(Phi, n, sink, "  the same, persistent");
    run_so<4, 0, 9>(Phi, n, sink, "row bands of 64 tiles over the left half + transposed-like stores into the right half");
    run_so<4, 0, 10>(Phi, n, sink, "LEVEL-ORDERED walk of all tile pairs (squares of 64, 32, 16, ... tiles), one workgroup per pair");
    run_so<4, 1, 10>(Phi, n, sink, "  the same, persistent (4 store waves / CU)");
    run_so<8, 1, 10>(Phi, n, sink, "  the same, persistent (8 store waves / CU)");
    {
        long off = 0;
        char nm[128];
        for (long sq = 64; sq >= 1; sq >>= 1) {
            const long cnt = 64 * sq;
            snprintf(nm, sizeof nm, "  level-ordered walk, ONLY the %ld squares of %ld x %ld tiles (%ld pairs)", 64 / sq, sq, sq, cnt);
            run_level<4, 0>(Phi, n, sink, off, cnt, cnt * 65536.0, 0, nm);
            if (sq < 64 && sq >= 8) {
                snprintf(nm, sizeof nm, "    the same, the squares' pairs interleaved (all %ld squares in flight together)", 64 / sq);
                run_level<4, 0>(Phi, n, sink, off, cnt, cnt * 65536.0, 1, nm);
            }
            off += cnt;
        }
    }
    run_level<4, 0>(Phi, n, sink, 0, 4096 + 2048, (4096 + 2048) * 65536.0, 0, "  levels 0 + 1 in ONE launch (39.2 + 20.9 us apart)");
    run_level<4, 0>(Phi, n, sink, 0, 4096 + 2048 + 1024, (4096 + 2048 + 1024) * 65536.0, 0, "  levels 0 + 1 + 2 in ONE launch");
    run_level<4, 0>(Phi, n, sink, 4096, 2048 + 1024, (2048 + 1024) * 65536.0, 0, "  levels 1 + 2 in ONE launch (20.9 + 11.3 us apart)");
    run_level<4, 0>(Phi, n, sink, 4096 + 2048 + 1024, 960, 960 * 65536.0, 0, "  levels 3 .. 6 in ONE launch (960 pairs)");
    run_level<4, 0>(Phi, n, sink, 4096, 4032, 4032 * 65536.0, 0, "  levels 1 .. 6 in ONE launch (the two half-size triangles, 4032 pairs)");
    {
        // the two launches ALTERNATING (what a two-launch Gram kernel would do, and no launch re-writes what the launch before it left in the
        // 256 MB Infinity Cache): level 0, then levels 1 .. 6 + diagonal
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        auto pair = [&](double sd) {
            hipLaunchKernelGGL((so2<4, 0, 10, 3>), dim3(4096), dim3(256), 0, 0, Phi, n, (long)4096, sd, sink, (long)0, 0);
            hipLaunchKernelGGL((so2<4, 0, 10, 3>), dim3(4160), dim3(256), 0, 0, Phi, n, (long)4160, sd, sink, (long)4096, 0);
        };
        for (int i = 0; i < 3; ++i) pair(1.0 + i);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) pair(2.0 + i);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        printf("%-76s %8.1f us   %.2f TB/s  (%.3f of 8 TB/s)\n", "TWO launches alternating: level 0, then levels 1 .. 6 + diagonal (all 537 MB)", ms * 1e3, 8.0 * n * n / ms / 1e9,
               8.0 * n * n / ms / 1e9 / 8.0);
        // and with a 600 MB memset between repetitions (cold Infinity Cache for every pair), timed separately
        double *scratch;
        hipMalloc(&scratch, (size_t)600 << 20);
        float tot = 0;
        for (int i = 0; i < 10; ++i) {
            hipMemsetAsync(scratch, i, (size_t)600 << 20, 0);
            hipEventRecord(e0, 0);
            pair(30.0 + i);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            tot += ms;
        }
        printf("%-76s %8.1f us   %.2f TB/s\n", "  the pair behind a 600 MB memset of another buffer, single pairs timed", tot / 10 * 1e3, 8.0 * n * n / (tot / 10) / 1e9);
        tot = 0;
        for (int i = 0; i < 10; ++i) {
            hipMemsetAsync(scratch, i, (size_t)600 << 20, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((so2<4, 0, 1, 3>), dim3(8256), dim3(256), 0, 0, Phi, n, (long)8256, 40.0 + i, sink, (long)0, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            tot += ms;
        }
        printf("%-76s %8.1f us   %.2f TB/s\n", "  the shipped shape (one launch, triangle row by row) behind the same memset", tot / 10 * 1e3, 8.0 * n * n / (tot / 10) / 1e9);
        tot = 0;
        for (int i = 0; i < 10; ++i) {
            hipMemsetAsync(scratch, i, (size_t)600 << 20, 0);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL((so2<4, 0, 2, 3>), dim3(8192), dim3(256), 0, 0, Phi, n, (long)8192, 50.0 + i, sink, (long)0, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            tot += ms;
        }
        printf("%-76s %8.1f us   %.2f TB/s\n", "  the rectangle walk (every byte once) behind the same memset", tot / 10 * 1e3, 8.0 * n * n / (tot / 10) / 1e9);
    }
    run_so<4, 1, 4>(Phi, n, sink, "persistent, triangle's tile pairs, the MIRROR WRITTEN ONE UNIT LATER than its tile");
    run_so<8, 1, 4>(Phi, n, sink, "  the same with 8 store waves / CU");
    run_so<8, 1, 1>(Phi, n, sink, "persistent, 8 store waves / CU, the triangle's tile pairs");
    run_so<16, 1, 1>(Phi, n, sink, "persistent, 16 store waves / CU, the triangle's tile pairs");
    run_so<4, 0, 1>(Phi, n, sink, "one 256-thread workgroup per tile pair (grid = 8256, 4 / CU): the shipped shape");
    return 0;
}
