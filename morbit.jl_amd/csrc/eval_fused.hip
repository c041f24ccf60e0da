// Fused batched surrogate evaluation on the fp64 matrix cores (dpad <= 128).
//
// For a block of 64 query points and a run of 64-centre tiles, per tile:
//   1. S'[c][q] = <cc_c, xc_q>                     MFMA, A operand = centre tile from LDS, B operand = query fragments (kept in
//                                                  registers for the whole kernel, or in LDS when two outputs share a pass at D = 128)
//   2. s = |xc_q|^2 + |cc_c|^2 - 2 S', phi(s), psi(s);  per output l:  v_l[q] += w_lc phi,  a_lc = w_lc psi
//   3. G_l'[t][q] += cc_c[t] a_lc                  MFMA: the a_l tile sits in the C/D register layout with the summed
//                                                  index c on its ROW axis, which is exactly the B-operand layout of the
//                                                  next MFMA, so phase 2's output feeds phase 3 with no data movement
// (the structure of a flash-attention forward pass: centres play K and V, the radial function plays softmax).
// Large models: the centre range is split over gridDim.y so that small query batches still fill the chip; a combine kernel sums
// the splits in a fixed order, adds the polynomial tail and writes  J_l = (sum_c a_lc) xc - G_l + grad p_l.
// Small models (<= 8 centre tiles, never split): the kernel's own epilogue does that (FINAL) -- no partials, no combine pass.
// Every launch takes its operands from an EvalDesc: one by value for a single mrbf_eval, an array (blockIdx.z = problem) for the
// batched entry points (mrbf_batch_run); the arithmetic per problem is the same, so a batch and single calls agree bit for bit.
#include "radial.hpp"
#ifndef MRBF_EVAL_DBG
#define MRBF_EVAL_DBG 0  // diagnostics builds only (tools/build_variant.sh): 1 no Jacobian epilogue, 2 no Jacobian MFMAs, 64 unpaired epilogue
#endif
#include "small.hpp"

namespace mrbf {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int EQ = 64;  // queries per workgroup
constexpr int EC = 64;  // centres per tile

// QLDS: the query fragments live in LDS instead of registers (frees D / 2 VGPRs per lane: two outputs per pass fit at D = 128 without
// accumulators in AGPRs, where the f64 MFMA runs at half rate -- tools/microbench.py)
template <int KID, bool FAST, int KOUT, int DT, bool JAC, bool FINAL, bool QLDS>
__global__ __launch_bounds__(256, (DT <= 4) ? 2 : 1) void eval_fused_kernel(EvalDesc one, const EvalDesc *__restrict__ many, int l0) {
    constexpr int D = DT * 16, LDC = D + 2;
    __shared__ __attribute__((aligned(16))) double Cs[EC * LDC];              // centre tile
    __shared__ double Ws[KOUT * EC];                                          // weights of the tile
    __shared__ double Sq[EC];                                                 // squared norms of the tile's centres
    __shared__ __attribute__((aligned(16))) double Xs[QLDS ? EQ * LDC : 2];   // QLDS: the query block
    const EvalDesc &E = many ? many[blockIdx.z] : one;
    if (many && ((int64_t)blockIdx.x * EQ >= E.mpad || (int)blockIdx.y >= E.nsplit)) return;
    const double *__restrict__ Xq = E.Xq;
    const double *__restrict__ Cc = E.Cc;
    const double *__restrict__ csq = E.csq;
    const double *__restrict__ Wc = E.Wc;
    const int64_t npad = E.npad, mpad = E.mpad;
    const int tiles_per_split = E.tiles_per_split;
    const KP kp = E.kp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t q0 = (int64_t)blockIdx.x * EQ;
    const int split = blockIdx.y;
    const int64_t qrow = q0 + wave * 16 + l15;

    // query fragments: B operand of phase 1, k-slice s -> Xq[qrow][4s + l4]
    double xb[QLDS ? 1 : D / 4];
    if constexpr (QLDS) {
        for (int e = tid; e < EQ * D / 2; e += 256) {
            const int row = (2 * e) / D, col = (2 * e) % D;
            *(v2d *)&Xs[row * LDC + col] = *(const v2d *)&Xq[(q0 + row) * D + col];
        }
    } else {
#pragma unroll
        for (int s = 0; s < D / 4; ++s) xb[s] = Xq[qrow * D + 4 * s + l4];
    }
    const double xs = E.xsq[qrow];

    v4d JT[KOUT][DT];
    double vsum[KOUT], sasum[KOUT];
#pragma unroll
    for (int l = 0; l < KOUT; ++l) {
        vsum[l] = 0.0;
        sasum[l] = 0.0;
#pragma unroll
        for (int t = 0; t < DT; ++t) JT[l][t] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    // centre-tile staging: 64 rows x D doubles, row-major in global (contiguous 64*D*8 bytes); DT v2d per thread
    const int64_t c_begin = (int64_t)split * tiles_per_split * EC;
    // the centre range is cut into gridDim.y nearly equal pieces (the last one may be shorter)
    const int ntiles_all = E.ntiles;
    const int nsub = __builtin_amdgcn_readfirstlane(E.nsub);  // (E may live in global memory: read inside the centre loop it was a flat load + wait per 16-centre step)
    const int my_tiles = min(tiles_per_split, ntiles_all - split * tiles_per_split);
    constexpr int NLD = 2 * DT;  // 64 * D / 2 v2d over 256 threads
    v2d stg[NLD];
    auto load_tile = [&](int64_t c0) {
        const v2d *src = reinterpret_cast<const v2d *>(Cc + c0 * D);
#pragma unroll
        for (int u = 0; u < NLD; ++u) stg[u] = src[tid + 256 * u];
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (tid + 256 * u) * 2;  // element index in the 64 x D tile
            const int row = e / D, col = e % D;
            *(v2d *)&Cs[row * LDC + col] = stg[u];
        }
    };
    // the tile's squared norms and weights travel with it, requested a tile ahead like the coordinates (fetched at the point of use
    // they were a memory round trip per 64-centre tile -- 128 per workgroup at n = 8192 -- with every wave waiting between two barriers)
    constexpr int NSC = ((KOUT + 1) * EC + 255) / 256;
    double stg_s[NSC];
    auto load_scalars = [&](int64_t c0) {
#pragma unroll
        for (int u = 0; u < NSC; ++u) {
            const int e = tid + 256 * u;
            if (e < EC)
                stg_s[u] = csq[c0 + e];
            else if (e < (KOUT + 1) * EC)
                stg_s[u] = Wc[(int64_t)(l0 + (e - EC) / EC) * npad + c0 + ((e - EC) % EC)];
        }
    };
    load_tile(c_begin);
    load_scalars(c_begin);
    for (int tile = 0; tile < my_tiles; ++tile) {
        const int64_t c0 = c_begin + (int64_t)tile * EC;
        __syncthreads();
        store_tile();
#pragma unroll
        for (int u = 0; u < NSC; ++u) {
            const int e = tid + 256 * u;
            if (e < EC)
                Sq[e] = stg_s[u];
            else if (e < (KOUT + 1) * EC)
                Ws[e - EC] = stg_s[u];
        }
        __syncthreads();
        if (tile + 1 < my_tiles) {
            load_tile(c0 + EC);
            load_scalars(c0 + EC);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if ((int)(c0 >> 4) + ct >= nsub) break;  // only padding from here on (zero weights: nothing to add)
            // ---- phase 1
            v4d S = {0.0, 0.0, 0.0, 0.0};
            const double *crow = &Cs[(16 * ct + l15) * LDC + l4];
            if constexpr (QLDS) {
                const double *xrow = &Xs[(wave * 16 + l15) * LDC + l4];
#pragma unroll
                for (int s = 0; s < D / 4; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64(crow[4 * s], xrow[4 * s], S, 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < D / 4; ++s) S = __builtin_amdgcn_mfma_f64_16x16x4f64(crow[4 * s], xb[s], S, 0, 0, 0);
            }
            // ---- phase 2: C/D layout: register r <-> centre c = 16 ct + l4 + 4 r, lane & 15 <-> query
            v4d Aw[KOUT];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + l4 + 4 * r;
                double s2 = fma(-2.0, S[r], xs + Sq[c]);
                s2 = s2 > 0.0 ? s2 : 0.0;
                double phi, psi;
                rbf_phi_psi_t<KID, FAST>(s2, kp, phi, psi);
#pragma unroll
                for (int l = 0; l < KOUT; ++l) {
                    const double w = Ws[l * EC + c];
                    vsum[l] = fma(w, phi, vsum[l]);
                    if constexpr (JAC) {  // (a pass for values only: no psi, no coefficient sums -- a quarter of the tile's vector work)
                        const double a = w * psi;
                        sasum[l] += a;
                        Aw[l][r] = a;
                    }
                }
            }
            // ---- phase 3: G_l'[t][q] += sum_c Cc[c][t] a_lc
            if (JAC) {
#pragma unroll
                for (int tt = 0; tt < DT; ++tt) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const double cop = Cs[(16 * ct + 4 * s + l4) * LDC + 16 * tt + l15];
#pragma unroll
                        for (int l = 0; l < KOUT; ++l) JT[l][tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(cop, Aw[l][s], JT[l][tt], 0, 0, 0);
                    }
                }
            }
        }
    }
    // ---- reduce the value / sum-of-a accumulators over the four lane groups that share a query (l4 = 0..3)
#pragma unroll
    for (int l = 0; l < KOUT; ++l) {
        vsum[l] += __shfl_xor(vsum[l], 16);
        vsum[l] += __shfl_xor(vsum[l], 32);
        sasum[l] += __shfl_xor(sasum[l], 16);
        sasum[l] += __shfl_xor(sasum[l], 32);
    }
    if constexpr (!FINAL) {
        double *__restrict__ vpart = E.vpart, *__restrict__ sapart = E.sapart, *__restrict__ gpart = E.gpart;
        if (l4 == 0) {
#pragma unroll
            for (int l = 0; l < KOUT; ++l) {
                vpart[((int64_t)split * mpad + qrow) * KOUT + l] = vsum[l];
                sapart[((int64_t)split * mpad + qrow) * KOUT + l] = sasum[l];
            }
        }
        if (JAC) {
            // G tiles -> gpart[split][q][l][t] through an LDS transpose (per wave 16 q x D), coalesced rows of D doubles
            __syncthreads();
            double *T = Cs + wave * 16 * LDC;
#pragma unroll
            for (int l = 0; l < KOUT; ++l) {
#pragma unroll
                for (int tt = 0; tt < DT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[l15 * LDC + 16 * tt + l4 + 4 * r] = JT[l][tt][r];
                __syncthreads();
                for (int e = lane; e < 16 * D; e += 64) {
                    const int qq = e / D, t = e % D;
                    gpart[(((int64_t)split * mpad + q0 + wave * 16 + qq) * KOUT + l) * D + t] = T[qq * LDC + t];
                }
                __syncthreads();
            }
        }
    } else {
        // ---- the whole centre range was this workgroup's: final values and Jacobians straight from the accumulators
        //      vals[p][l] = sum_c w phi + p_l(x);   jac[p][t*k + l] = (sum_c a_lc) xc[p][t] - G_l[t] + lam[t+1][l]
        const int d = E.d, k = E.k, q = E.q;
        const int64_t m = E.m;
        const double *__restrict__ lam = E.lam;
        double *__restrict__ vals = E.vals;
        double *__restrict__ jac = E.jac;
        // polynomial tail of the value, p_l(x) = lam_0l + sum_t lam_tl (xc_t + mean_t): the centred coordinates are at hand (query
        // fragments), the tail coefficients of this pass and the centroid go through LDS once per workgroup; lane (query l15, part l4)
        // takes the coordinates 4 s + l4, the four parts are added in a fixed order
        if (vals) {
            __syncthreads();  // the last centre tile is done: its LDS area holds the coefficients now
            double *Lm = Cs;  // KOUT x D tail coefficients, then D centroid entries
            for (int e = tid; e < (KOUT + 1) * D; e += 256) {
                const int l = e / D, t = e % D;
                Lm[e] = (t < d && q > 1) ? (l < KOUT ? (l0 + l < k ? lam[(int64_t)(t + 1) * k + l0 + l] : 0.0) : E.mean[t]) : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int l = 0; l < KOUT; ++l) {
                double acc = 0.0, cst = 0.0;
#pragma unroll
                for (int s = 0; s < D / 4; ++s) {
                    const double xq = QLDS ? Xs[(wave * 16 + l15) * LDC + 4 * s + l4] : xb[QLDS ? 0 : s];
                    acc = fma(Lm[l * D + 4 * s + l4], xq, acc);
                }
                for (int t = lane; t < D; t += 64) cst = fma(Lm[l * D + t], Lm[KOUT * D + t], cst);
                for (int off = 32; off > 0; off >>= 1) cst += __shfl_xor(cst, off);
                acc += __shfl_xor(acc, 16);
                acc += __shfl_xor(acc, 32);
                double v = vsum[l];
                if (q > 0 && l0 + l < k) v += lam[l0 + l];
                if (q > 1) v += acc + cst;
                if (l4 == 0 && qrow < m && l0 + l < k) vals[qrow * k + l0 + l] = v;
            }
        }
        if (JAC && jac) {
            __syncthreads();  // every wave is done with the last centre tile
            double *T = Cs + wave * 16 * LDC;
            double *SA = Ws;  // KOUT x 64 (= 4 waves x 16 queries): the weights of the last tile are not needed any more
#pragma unroll
            for (int l = 0; l < KOUT; ++l)
                if (l4 == 0) SA[l * EQ + wave * 16 + l15] = sasum[l];
#pragma unroll
            for (int l = 0; l < KOUT; ++l) {
#pragma unroll
                for (int tt = 0; tt < DT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[l15 * LDC + 16 * tt + l4 + 4 * r] = JT[l][tt][r];
                __syncthreads();
                if (l0 + l < k) {
                    for (int e = lane; e < 16 * D; e += 64) {
                        const int qq = e / D, t = e % D;
                        const int64_t row = q0 + wave * 16 + qq;
                        if (row < m && t < d) {
                            double v = fma(SA[l * EQ + wave * 16 + qq], Xq[row * D + t], -T[qq * LDC + t]);
                            if (q > 1) v += lam[(int64_t)(t + 1) * k + l0 + l];
                            jac[row * (int64_t)k * d + (int64_t)t * k + l0 + l] = v;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

// ---- D = 128 / 256: the same evaluation with the coordinates cut in two halves over two groups of four waves (512 threads) -----
// With D = 128 one wave cannot hold the Jacobian accumulators of two outputs (2 x 8 tiles = 128 VGPRs) next to its query fragments and
// the staged centre tile: the compiler moved them to AGPRs, where the f64 MFMA issues at half rate, and one wave per SIMD left
// nothing to overlap the radial function and the LDS traffic with.  Here group g (waves 4g .. 4g + 3) owns coordinates H g .. H g + H - 1
// (H = D / 2):
//   phase 1  each group sums its half of <cc_c, xc_q> (H / 4 MFMAs per 16 x 16 tile), the halves meet through LDS (one barrier per
//            16-centre step, slots double-buffered by the step's parity) and are added in the fixed order  S = S_0 + S_1;
//   phase 2  both groups apply the radial function to the same S (the VALU work is duplicated; it runs under the partner's MFMAs);
//   phase 3  each group accumulates its H Jacobian columns (KOUT x H / 16 tiles: 64 VGPRs, no AGPRs, two waves per SIMD).
// No matrix-core work is duplicated.  H = 64 (D = 128): two outputs per pass, 64-centre tiles.  H = 128 (D = 256, the C5 models): one
// output per pass, 32-centre tiles (LDS), replaces the rocBLAS pipeline of eval.hip for 128 < d <= 256.
template <int KID, bool FAST, int KOUT, int H, int ECT, bool JAC, bool FINAL>
__global__ __launch_bounds__(512, 2) void eval_fused_split_kernel(EvalDesc one, const EvalDesc *__restrict__ many, int l0) {
    constexpr int D = 2 * H, LDC = D + 2, LDT = 65, NCT = ECT / 16, HT = H / 16;
    static_assert(8 * 16 * LDT <= ECT * LDC + 2 * 8 * 256, "epilogue transpose area");
    __shared__ __attribute__((aligned(16))) double smem[ECT * LDC + 2 * 8 * 256];
    double *Cs = smem;              // centre tile; the epilogue's transpose area (8 waves x 16 x LDT, reaching into Sx)
    double *Sx = smem + ECT * LDC;  // partial S tiles of the eight waves, two parities; the epilogue's coefficient area
    __shared__ double Ws[KOUT * ECT];
    __shared__ double Sq[ECT];
    const EvalDesc &E = many ? many[blockIdx.z] : one;
    if (many && ((int64_t)blockIdx.x * EQ >= E.mpad || (int)blockIdx.y >= E.nsplit)) return;
    const double *__restrict__ Xq = E.Xq;
    const double *__restrict__ Cc = E.Cc;
    const double *__restrict__ csq = E.csq;
    const double *__restrict__ Wc = E.Wc;
    const int64_t npad = E.npad, mpad = E.mpad;
    const KP kp = E.kp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = wave >> 2, qw = wave & 3;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t q0 = (int64_t)blockIdx.x * EQ;
    const int split = blockIdx.y;
    const int64_t qrow = q0 + qw * 16 + l15;
    double xb[H / 4];  // this group's half of the query fragments: k-slice s -> Xq[qrow][H grp + 4 s + l4]
#pragma unroll
    for (int s = 0; s < H / 4; ++s) xb[s] = Xq[qrow * D + H * grp + 4 * s + l4];
    const double xs = E.xsq[qrow];
    v4d JT[KOUT][HT];
    double vsum[KOUT], sasum[KOUT];
#pragma unroll
    for (int l = 0; l < KOUT; ++l) {
        vsum[l] = 0.0;
        sasum[l] = 0.0;
#pragma unroll
        for (int t = 0; t < HT; ++t) JT[l][t] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    // the descriptor counts centre tiles of EC = 64; this kernel walks them in pieces of ECT
    const int64_t c_begin = (int64_t)split * E.tiles_per_split * EC;
    const int ntiles_all = E.ntiles;
    const int nsub = __builtin_amdgcn_readfirstlane(E.nsub);  // (E may live in global memory: read inside the centre loop it was a flat load + wait per 16-centre step)
    const int my_tiles = min(E.tiles_per_split, ntiles_all - split * E.tiles_per_split) * (EC / ECT);
    constexpr int NLD = ECT * D / 2 / 512;
    v2d stg[NLD];
    auto load_tile = [&](int64_t c0) {
        const v2d *src = reinterpret_cast<const v2d *>(Cc + c0 * D);
#pragma unroll
        for (int u = 0; u < NLD; ++u) stg[u] = src[tid + 512 * u];
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (tid + 512 * u) * 2;
            const int row = e / D, col = e % D;
            *(v2d *)&Cs[row * LDC + col] = stg[u];
        }
    };
    // the tile's squared norms and weights travel with it (one value per thread: KOUT * ECT + ECT <= 512), requested a tile ahead like
    // the coordinates -- fetched at the point of use they cost a memory round trip per tile with every wave waiting
    static_assert((KOUT + 1) * ECT <= 512, "one staged scalar per thread");
    double stg_w = 0.0;
    auto load_scalars = [&](int64_t c0) {
        if (tid < ECT)
            stg_w = csq[c0 + tid];
        else if (tid < (KOUT + 1) * ECT)
            stg_w = Wc[(int64_t)(l0 + (tid - ECT) / ECT) * npad + c0 + ((tid - ECT) % ECT)];
    };
    load_tile(c_begin);
    load_scalars(c_begin);
    for (int tile = 0; tile < my_tiles; ++tile) {
        const int64_t c0 = c_begin + (int64_t)tile * ECT;
        __syncthreads();
        store_tile();
        if (tid < ECT)
            Sq[tid] = stg_w;
        else if (tid < (KOUT + 1) * ECT)
            Ws[tid - ECT] = stg_w;
        __syncthreads();
        if (tile + 1 < my_tiles) {
            load_tile(c0 + ECT);
            load_scalars(c0 + ECT);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if ((int)(c0 >> 4) + ct >= nsub) break;  // only padding from here on (uniform: every wave leaves before the barrier)
            // ---- phase 1: this group's half of S', exchanged with the partner wave (same queries, other half)
            v4d Sp = {0.0, 0.0, 0.0, 0.0};
            const double *crow = &Cs[(16 * ct + l15) * LDC + H * grp + l4];
#pragma unroll
            for (int s = 0; s < H / 4; ++s) Sp = __builtin_amdgcn_mfma_f64_16x16x4f64(crow[4 * s], xb[s], Sp, 0, 0, 0);
            const int par = (tile * NCT + ct) & 1;
            double *mine = &Sx[(par * 8 + wave) * 256];
            const double *theirs = &Sx[(par * 8 + (wave ^ 4)) * 256];
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[r * 64 + lane] = Sp[r];
            __syncthreads();
            v4d S;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double o = theirs[r * 64 + lane];
                S[r] = grp == 0 ? Sp[r] + o : o + Sp[r];  // S_0 + S_1 in both groups
            }
            // ---- phase 2
            v4d Aw[KOUT];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + l4 + 4 * r;
                double s2 = fma(-2.0, S[r], xs + Sq[c]);
                s2 = s2 > 0.0 ? s2 : 0.0;
                double phi, psi;
                rbf_phi_psi_t<KID, FAST>(s2, kp, phi, psi);
#pragma unroll
                for (int l = 0; l < KOUT; ++l) {
                    const double w = Ws[l * ECT + c];
                    vsum[l] = fma(w, phi, vsum[l]);
                    if constexpr (JAC) {
                        const double a = w * psi;
                        sasum[l] += a;
                        Aw[l][r] = a;
                    }
                }
            }
            // ---- phase 3: this group's H Jacobian columns
            if (JAC && !(MRBF_EVAL_DBG & 2)) {
#pragma unroll
                for (int tt = 0; tt < HT; ++tt) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const double cop = Cs[(16 * ct + 4 * s + l4) * LDC + H * grp + 16 * tt + l15];
#pragma unroll
                        for (int l = 0; l < KOUT; ++l) JT[l][tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(cop, Aw[l][s], JT[l][tt], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < KOUT; ++l) {
        vsum[l] += __shfl_xor(vsum[l], 16);
        vsum[l] += __shfl_xor(vsum[l], 32);
        sasum[l] += __shfl_xor(sasum[l], 16);
        sasum[l] += __shfl_xor(sasum[l], 32);
    }
    const int d = E.d, k = E.k, q = E.q;
    const int64_t m = E.m;
    const double *__restrict__ lam = E.lam;
    double *__restrict__ vals = E.vals;
    double *__restrict__ jac = E.jac;
    if constexpr (!FINAL) {
        if (l4 == 0 && grp == 0) {
#pragma unroll
            for (int l = 0; l < KOUT; ++l) {
                E.vpart[((int64_t)split * mpad + qrow) * KOUT + l] = vsum[l];
                E.sapart[((int64_t)split * mpad + qrow) * KOUT + l] = sasum[l];
            }
        }
    } else if (vals) {
        // polynomial tail of the value, p_l(x) = lam_0l + sum_t lam_tl (xc_t + mean_t), from the query fragments (see eval_fused_kernel):
        // each group sums its H coordinates, group 1 hands its part to group 0 through LDS
        __syncthreads();
        double *Lm = Sx;                    // KOUT x D tail coefficients, then D centroid entries (the exchange slots are free now)
        double *Pp = Sx + (KOUT + 1) * D;   // KOUT x 64 partial sums of group 1
        for (int e = tid; e < (KOUT + 1) * D; e += 512) {
            const int l = e / D, t = e % D;
            Lm[e] = (t < d && q > 1) ? (l < KOUT ? (l0 + l < k ? lam[(int64_t)(t + 1) * k + l0 + l] : 0.0) : E.mean[t]) : 0.0;
        }
        __syncthreads();
        double accs[KOUT], csts[KOUT];
#pragma unroll
        for (int l = 0; l < KOUT; ++l) {
            double acc = 0.0, cst = 0.0;
#pragma unroll
            for (int s = 0; s < H / 4; ++s) acc = fma(Lm[l * D + H * grp + 4 * s + l4], xb[s], acc);
            for (int t = lane; t < D; t += 64) cst = fma(Lm[l * D + t], Lm[KOUT * D + t], cst);
            for (int off = 32; off > 0; off >>= 1) cst += __shfl_xor(cst, off);
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            accs[l] = acc;
            csts[l] = cst;
            if (grp == 1 && l4 == 0) Pp[l * EQ + qw * 16 + l15] = acc;
        }
        __syncthreads();
        if (grp == 0 && l4 == 0 && qrow < m) {
#pragma unroll
            for (int l = 0; l < KOUT; ++l) {
                if (l0 + l >= k) continue;
                double v = vsum[l];
                if (q > 0) v += lam[l0 + l];
                if (q > 1) v += (accs[l] + Pp[l * EQ + qw * 16 + l15]) + csts[l];
                vals[qrow * k + l0 + l] = v;
            }
        }
    }
    if (JAC && (!FINAL || jac) && !(MRBF_EVAL_DBG & 1)) {
        __syncthreads();  // every wave is done with the last centre tile (and with the coefficient area)
        double *T = smem + wave * 16 * LDT;
        double *SA = Ws;  // KOUT x 64 is at most KOUT x ECT only for ECT = 64: the sums go to Sq's neighbour otherwise
        __shared__ double SAs[KOUT * EQ];
        SA = SAs;
        if (FINAL && grp == 0 && l4 == 0) {
#pragma unroll
            for (int l = 0; l < KOUT; ++l) SA[l * EQ + qw * 16 + l15] = sasum[l];
        }
        // two outputs whose Jacobian entries are neighbours in memory (k even): both through the transpose area together, 32 columns
        // at a time, one 16-byte store per (site, coordinate) -- the coordinates of a site are then written as whole lines, the
        // query coordinate and the tail coefficients are fetched once (C4, 64 starts: the epilogue was 0.42 of the 2.34 ms)
        bool paired = false;
        // (the 16-byte accesses below need jac and lam 16-byte aligned: true for the arena, checked for a caller's device pointer)
        if constexpr (FINAL && KOUT == 2)
            paired = (k & 1) == 0 && (l0 & 1) == 0 && l0 + 1 < k && !(MRBF_EVAL_DBG & 64) &&
                     (((uintptr_t)jac | (uintptr_t)lam) & 15) == 0;
        if constexpr (FINAL && KOUT == 2) {
            if (paired) {
                constexpr int LDP = 33;
                // T2 of the eight waves (8 x 2 x 16 x 33 doubles) reaches past the centre tile into Sx, where the value epilogue kept the
                // tail coefficients: safe behind the __syncthreads at the top of this epilogue, and inside the block:
                static_assert(8 * 2 * 16 * LDP <= ECT * LDC + 2 * 8 * 256, "paired Jacobian transpose area");
                double *T2 = smem + wave * (2 * 16 * LDP);  // [l][qq][32 columns]
#pragma unroll
                for (int hq = 0; hq < H / 32; ++hq) {
#pragma unroll
                    for (int l = 0; l < 2; ++l)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) T2[(l * 16 + l15) * LDP + 16 * tt + l4 + 4 * r] = JT[l][2 * hq + tt][r];
                    __syncthreads();
                    for (int e = lane; e < 16 * 32; e += 64) {
                        const int qq = e >> 5, t = e & 31, col = H * grp + 32 * hq + t;
                        const int64_t row = q0 + qw * 16 + qq;
                        if (row < m && col < d) {
                            const double x = Xq[row * D + col];
                            v2d v;
                            v.x = fma(SA[qw * 16 + qq], x, -T2[qq * LDP + t]);
                            v.y = fma(SA[EQ + qw * 16 + qq], x, -T2[(16 + qq) * LDP + t]);
                            if (q > 1) {
                                const v2d lm = *(const v2d *)&lam[(int64_t)(col + 1) * k + l0];
                                v.x += lm.x;
                                v.y += lm.y;
                            }
                            *(v2d *)&jac[row * (int64_t)k * d + (int64_t)col * k + l0] = v;
                        }
                    }
                    __syncthreads();
                }
            }
        }
#pragma unroll
        for (int l = 0; l < KOUT; ++l) {
            if (paired) break;
#pragma unroll
            for (int hc = 0; hc < H / 64; ++hc) {  // 64 Jacobian columns at a time through the transpose area
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[l15 * LDT + 16 * tt + l4 + 4 * r] = JT[l][4 * hc + tt][r];
                __syncthreads();
                for (int e = lane; e < 16 * 64; e += 64) {
                    const int qq = e / 64, t = e % 64, col = H * grp + 64 * hc + t;
                    const int64_t row = q0 + qw * 16 + qq;
                    if constexpr (!FINAL) {
                        E.gpart[(((int64_t)split * mpad + row) * KOUT + l) * D + col] = T[qq * LDT + t];
                    } else {
                        if (l0 + l < k && row < m && col < d) {
                            double v = fma(SA[l * EQ + qw * 16 + qq], Xq[row * D + col], -T[qq * LDT + t]);
                            if (q > 1) v += lam[(int64_t)(col + 1) * k + l0 + l];
                            jac[row * (int64_t)k * d + (int64_t)col * k + l0 + l] = v;
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
}

// ---- D = 256, values only, split centre range: a wave owns ALL coordinates of its sixteen queries -----------------------------------
// The two-group form above exists for the Jacobian accumulators.  A pass for values has none, and what the split costs it is an LDS
// exchange and a workgroup barrier of eight waves per sixteen centres (32 MFMAs per wave): 0.42 of the fp64 MFMA peak on the PS
// solver's populations (d = 256, 5160 points, 2048 centres).  Here eight waves of one workgroup per CU take 128 queries, each wave
// keeps its queries' 256 coordinates in registers (128 VGPRs) and runs the 64 MFMAs of a 16 x 16 tile as the SAME two chains --
// coordinates 0..127 and 128..255, summed S_0 + S_1 -- so the values are bit for bit those of the two-group kernel (and of a call that
// also asks for Jacobians), with no exchange and one barrier per 32-centre tile.  Non-FINAL only (partials + combine pass).
template <int KID, bool FAST, int KOUT>
__global__ __launch_bounds__(512, 1) void eval_vals256_kernel(EvalDesc one, const EvalDesc *__restrict__ many, int l0) {
    constexpr int D = 256, H = 128, ECT = 32, LDC = D + 2, NCT = ECT / 16, EQW = 128;
    __shared__ __attribute__((aligned(16))) double Cs[ECT * LDC];
    __shared__ double Ws[KOUT * ECT];
    __shared__ double Sq[ECT];
    const EvalDesc &E = many ? many[blockIdx.z] : one;
    if (many && ((int64_t)blockIdx.x * EQW >= E.mpad || (int)blockIdx.y >= E.nsplit)) return;
    const double *__restrict__ Xq = E.Xq;
    const double *__restrict__ Cc = E.Cc;
    const double *__restrict__ csq = E.csq;
    const double *__restrict__ Wc = E.Wc;
    const int64_t npad = E.npad, mpad = E.mpad;
    const KP kp = E.kp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int64_t q0 = (int64_t)blockIdx.x * EQW;
    const int split = blockIdx.y;
    const int64_t qrow = q0 + wave * 16 + l15;
    const bool qin = qrow < mpad;  // (mpad is a multiple of 64: the second half of the last workgroup may lie beyond it)
    const int64_t qld = qin ? qrow : mpad - 1;
    double xb[D / 4];  // k-slice s -> Xq[qrow][4 s + l4]
#pragma unroll
    for (int s = 0; s < D / 4; ++s) xb[s] = Xq[qld * D + 4 * s + l4];
    const double xs = E.xsq[qld];
    double vsum[KOUT];
#pragma unroll
    for (int l = 0; l < KOUT; ++l) vsum[l] = 0.0;
    const int64_t c_begin = (int64_t)split * E.tiles_per_split * EC;
    const int nsub = __builtin_amdgcn_readfirstlane(E.nsub);
    const int my_tiles = min(E.tiles_per_split, E.ntiles - split * E.tiles_per_split) * (EC / ECT);
    constexpr int NLD = ECT * D / 2 / 512;
    v2d stg[NLD];
    auto load_tile = [&](int64_t c0) {
        const v2d *src = reinterpret_cast<const v2d *>(Cc + c0 * D);
#pragma unroll
        for (int u = 0; u < NLD; ++u) stg[u] = src[tid + 512 * u];
    };
    static_assert((KOUT + 1) * ECT <= 512, "one staged scalar per thread");
    double stg_w = 0.0;
    auto load_scalars = [&](int64_t c0) {
        if (tid < ECT)
            stg_w = csq[c0 + tid];
        else if (tid < (KOUT + 1) * ECT)
            stg_w = Wc[(int64_t)(l0 + (tid - ECT) / ECT) * npad + c0 + ((tid - ECT) % ECT)];
    };
    load_tile(c_begin);
    load_scalars(c_begin);
    for (int tile = 0; tile < my_tiles; ++tile) {
        const int64_t c0 = c_begin + (int64_t)tile * ECT;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (tid + 512 * u) * 2;
            *(v2d *)&Cs[(e / D) * LDC + (e % D)] = stg[u];
        }
        if (tid < ECT)
            Sq[tid] = stg_w;
        else if (tid < (KOUT + 1) * ECT)
            Ws[tid - ECT] = stg_w;
        __syncthreads();
        if (tile + 1 < my_tiles) {
            load_tile(c0 + ECT);
            load_scalars(c0 + ECT);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if ((int)(c0 >> 4) + ct >= nsub) break;  // only padding from here on
            v4d S0 = {0.0, 0.0, 0.0, 0.0}, S1 = {0.0, 0.0, 0.0, 0.0};
            const double *crow = &Cs[(16 * ct + l15) * LDC + l4];
#pragma unroll
            for (int s = 0; s < H / 4; ++s) {
                S0 = __builtin_amdgcn_mfma_f64_16x16x4f64(crow[4 * s], xb[s], S0, 0, 0, 0);
                S1 = __builtin_amdgcn_mfma_f64_16x16x4f64(crow[H + 4 * s], xb[H / 4 + s], S1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + l4 + 4 * r;
                double s2 = fma(-2.0, S0[r] + S1[r], xs + Sq[c]);
                s2 = s2 > 0.0 ? s2 : 0.0;
                double phi, psi;
                rbf_phi_psi_t<KID, FAST>(s2, kp, phi, psi);
#pragma unroll
                for (int l = 0; l < KOUT; ++l) vsum[l] = fma(Ws[l * ECT + c], phi, vsum[l]);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < KOUT; ++l) {
        vsum[l] += __shfl_xor(vsum[l], 16);
        vsum[l] += __shfl_xor(vsum[l], 32);
    }
    if (l4 == 0 && qin) {
#pragma unroll
        for (int l = 0; l < KOUT; ++l) {
            E.vpart[((int64_t)split * mpad + qrow) * KOUT + l] = vsum[l];
            E.sapart[((int64_t)split * mpad + qrow) * KOUT + l] = 0.0;  // (the combine pass sums it; only a Jacobian would use it)
        }
    }
}

// vals[p][l0 + l] = sum_s vpart + p_l(x);   jac[p][t*k + l0 + l] = (sum_s sa) xc[p][t] - sum_s G + lam[t+1]
// One workgroup of 128 threads per query point.  The polynomial tail of the value -- a dot product over d coordinates per output --
// is spread over the threads that also produce the Jacobian entries and summed in a fixed order through LDS (it used to be a serial
// d-term loop of one thread per output, the critical path of the whole workgroup: 54 us for 10^4 points at d = 64, 92 us at d = 128).
template <int KOUT>
__global__ __launch_bounds__(128) void eval_combine_kernel(EvalDesc one, const EvalDesc *__restrict__ many, int l0, int D) {
    const EvalDesc &E = many ? many[blockIdx.y] : one;
    const int64_t p = blockIdx.x, m = E.m, mpad = E.mpad;
    if (p >= m) return;
    const double *__restrict__ vpart = E.vpart, *__restrict__ sapart = E.sapart, *__restrict__ gpart = E.gpart;
    const double *__restrict__ Xq = E.Xq, *__restrict__ Xorig = E.X, *__restrict__ lam = E.lam;
    double *__restrict__ vals = E.vals, *__restrict__ jac = E.jac;
    const int nsplit = E.nsplit, d = E.d, k = E.k, q = E.q;
    __shared__ double sa[KOUT], sv[KOUT];
    __shared__ double pdot[KOUT][128];
    const int tid = threadIdx.x;
    if (tid < KOUT && l0 + tid < k) {
        double v = 0.0, a = 0.0;
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            double xv[8], xa[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                xv[u] = s0 + u < nsplit ? vpart[((int64_t)(s0 + u) * mpad + p) * KOUT + tid] : 0.0;
                xa[u] = s0 + u < nsplit ? sapart[((int64_t)(s0 + u) * mpad + p) * KOUT + tid] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) {
                    v += xv[u];
                    a += xa[u];
                }
        }
        sa[tid] = a;
        sv[tid] = v;
    }
    // partial dot products lam[t+1][l] * x[t] over this thread's coordinates t = tid, tid + 128, ...
    if (vals && q > 1) {
#pragma unroll
        for (int l = 0; l < KOUT; ++l) {
            double acc = 0.0;
            if (l0 + l < k)
                for (int t = tid; t < d; t += 128) acc = fma(lam[(int64_t)(t + 1) * k + l0 + l], Xorig[p * d + t], acc);
            pdot[l][tid] = acc;
        }
    }
    __syncthreads();
    if (vals && tid < KOUT && l0 + tid < k) {
        double v = sv[tid];
        if (q > 0) v += lam[l0 + tid];
        if (q > 1) {
            const int nt = d < 128 ? d : 128;
            double dsum = 0.0;
            for (int t = 0; t < nt; ++t) dsum += pdot[tid][t];  // fixed order
            v += dsum;
        }
        vals[p * k + l0 + tid] = v;
    }
    if (!jac) return;
    for (int e = tid; e < KOUT * d; e += blockDim.x) {
        const int l = e / d, t = e % d;
        if (l0 + l >= k) continue;
        double g = 0.0;
        for (int s0 = 0; s0 < nsplit; s0 += 8) {  // eight loads in flight, added in the order of the splits (a load per addition was a memory round trip each)
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = s0 + u < nsplit ? gpart[(((int64_t)(s0 + u) * mpad + p) * KOUT + l) * D + t] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) g += x[u];
        }
        double v = fma(sa[l], Xq[p * D + t], -g);
        if (q > 1) v += lam[(int64_t)(t + 1) * k + l0 + l];
        jac[p * (int64_t)k * d + (int64_t)t * k + l0 + l] = v;
    }
}

// The same for a pass without Jacobians (round 6; the PS solver's populations: one such pass per generation), a WAVE per query point:
// the coefficient sums are not read, the splits' loads are in flight together, four points per workgroup.  Every sum in the order of
// eval_combine_kernel (splits ascending; tail: the 128 partial products t, t + 128 summed t ascending) -- the values are bit for bit
// those of a call that also asks for Jacobians.
template <int KOUT>
__global__ __launch_bounds__(256) void eval_combine_vals_kernel(EvalDesc one, const EvalDesc *__restrict__ many, int l0) {
    const EvalDesc &E = many ? many[blockIdx.y] : one;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t p = (int64_t)blockIdx.x * 4 + wave, m = E.m, mpad = E.mpad;
    const bool active = p < m;
    const double *__restrict__ vpart = E.vpart, *__restrict__ Xorig = E.X, *__restrict__ lam = E.lam;
    const int nsplit = E.nsplit, d = E.d, k = E.k, q = E.q;
    __shared__ double pdot[4][KOUT][128];
    double v = 0.0;
    if (active && lane < KOUT && l0 + lane < k) {
        for (int s0 = 0; s0 < nsplit; s0 += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = s0 + u < nsplit ? vpart[((int64_t)(s0 + u) * mpad + p) * KOUT + lane] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nsplit) v += x[u];
        }
    }
    if (q > 1) {
#pragma unroll
        for (int l = 0; l < KOUT; ++l)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t0 = lane + 64 * h;
                double acc = 0.0;
                if (active && l0 + l < k)
                    for (int t = t0; t < d; t += 128) acc = fma(lam[(int64_t)(t + 1) * k + l0 + l], Xorig[p * d + t], acc);
                pdot[wave][l][t0] = acc;
            }
    }
    __syncthreads();
    if (active && lane < KOUT && l0 + lane < k) {
        if (q > 0) v += lam[l0 + lane];
        if (q > 1) {
            const int nt = d < 128 ? d : 128;
            double dsum = 0.0;
            for (int t = 0; t < nt; ++t) dsum += pdot[wave][lane][t];  // fixed order
            v += dsum;
        }
        if (E.vals) E.vals[p * k + l0 + lane] = v;
    }
}

// queries: Xq[row][t] = X[row][t] - mean[t] (zero in the padding), xsq[row] = |Xq[row]|^2 -- center_pad_kernel's arithmetic (prep.hip)
__global__ void center_pad_batch_kernel(const EvalDesc *__restrict__ many, int Dfixed) {
    const EvalDesc &E = many[blockIdx.y];
    const int D = Dfixed > 0 ? Dfixed : (E.d <= 64 ? 64 : 128);  // 0: by the descriptor (a batch of mixed dimensions)
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= E.mpad) return;
    double s = 0.0;
    for (int t = lane; t < D; t += 64) {
        double v = 0.0;
        if (row < E.m && t < E.d) v = E.X[row * E.d + t] - E.mean[t];
        E.Xq[row * D + t] = v;
        s = fma(v, v, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) E.xsq[row] = s;
}

int eval_nsplit(const mrbf_ctx *ctx, int64_t m, int ntiles, bool check_call) {
    const int64_t mpad = round_up(m, EQ);
    // Split the centre range so that the grid fills the resident workgroup slots (2 per CU) in whole rounds: the cost of a
    // split count is (rounds of workgroups) x (tiles per workgroup) plus the combine pass, which reads one partial per split.
    // (C3: 157 query tiles -> 3 splits of 43 tiles = 471 workgroups in one round, instead of 4 x 32 tiles in two rounds.)
    const int slots = 2 * ctx->ncu;
    const int64_t qtiles = mpad / EQ;
    // few centre tiles (n <= 512): never split -- a workgroup's prologue and epilogue (query fragments in, 64 x D Jacobian tile
    // out) cost more than the handful of tiles it would shed, the unsplit kernel finishes values and Jacobians itself (no partials,
    // no combine pass), and such models are evaluated in batches that fill the chip anyway (mrbf_batch_run, the PS solver)
    // -- unless the query batch is small as well (at most an eighth of the workgroup slots: the populations of a PS step at d <= 24,
    // single points): the launch then is one workgroup's walk over the tiles, 35 us at n = 512 whatever m; one tile per workgroup
    // + the combine pass: 10 + 7 us (PS step d = 12, n = 512: 7.25 -> 5.15 ms; MRBF_EVAL_NSPLIT_SMALL=1 keeps them unsplit)
    static const int small_split = mrbf_env("MRBF_EVAL_NSPLIT_SMALL") ? atoi(mrbf_env("MRBF_EVAL_NSPLIT_SMALL")) : 8;
    // (the same rule for a member of a batch -- bit-identical to the single call --: batch.hip launches split and unsplit members apart)
    // Not for the residual check (the model at its own sites: m = n): in a batch of 64 starts those launches fill the chip together and
    // splitting them cost C4 1.6 % (19 540 against 19 870 problems/s, three alternating runs); the single fit's check follows the same rule.
    const bool small_batch = !check_call && qtiles * 8 <= slots;
    if (ntiles >= 4 && ntiles <= 8 && small_split > 1 && small_batch) return std::min(small_split, ntiles);
    // the PS solver's populations (ctx->eval_population, values only): a query batch that fills less than half of the workgroup slots
    // (round 6: the ideal-point populations of a step at d = 128, 5160 points on n = 257 sites = 81 unsplit workgroups on 256 CUs, 40 us
    // per generation) takes one tile per workgroup + the combine pass, which for values reads a few bytes per point.  Only there: the
    // API's own calls keep ONE rule for values and Jacobians (a values-only call and a call with Jacobians return the same bits, and so
    // does a member of a batch: tests/test_gpu_configs.py), and C4's batches fill the chip by their number.
    static const int vals_split = mrbf_env("MRBF_EVAL_NSPLIT_VALS") ? atoi(mrbf_env("MRBF_EVAL_NSPLIT_VALS")) : 1;
    if (ctx->eval_population && vals_split && !check_call && ntiles >= 4 && ntiles <= 8 && qtiles * 2 <= slots) return (int)std::min<int64_t>(ntiles, slots / qtiles);
    if (ntiles <= 8) return 1;
    static const double comb_small = mrbf_env("MRBF_EVAL_COMB_SMALL") ? atof(mrbf_env("MRBF_EVAL_COMB_SMALL")) : 0.1;
    int nsplit = 1;
    double best_cost = 1e300;
    for (int s = 1; s <= std::min(ntiles, 32); ++s) {
        const int tps_s = (ntiles + s - 1) / s;
        if ((int64_t)(s - 1) * tps_s >= ntiles) continue;  // the last piece would be empty
        const double rounds = std::ceil((double)(qtiles * s) / slots);
        const double cost = rounds * tps_s + (small_batch ? comb_small : 0.75) * s;  // (a small query batch: the combine pass reads next to nothing)
        if (cost < best_cost) {
            best_cost = cost;
            nsplit = s;
        }
    }
    static const int force_split = mrbf_env("MRBF_EVAL_NSPLIT") ? atoi(mrbf_env("MRBF_EVAL_NSPLIT")) : 0;
    if (force_split > 0 && force_split <= ntiles && (int64_t)(force_split - 1) * ((ntiles + force_split - 1) / force_split) < ntiles)
        nsplit = force_split;
    return nsplit;
}

// outputs per pass: two when the model has several (at D = 128 with the query block in LDS)
int outputs_per_pass(int k, int D, bool want_jac) {
    static const int ko128 = mrbf_env("MRBF_EVAL_KO128") ? atoi(mrbf_env("MRBF_EVAL_KO128")) : 2;
    static const int ko256 = mrbf_env("MRBF_EVAL_KO256") ? atoi(mrbf_env("MRBF_EVAL_KO256")) : 2;
    // D = 256: the Jacobian tiles of one output fill the accumulator budget; a pass for values only carries no such tiles and takes two
    // outputs -- the distances, the radial function and the centre traffic once instead of twice (a population of the PS solver at
    // d = 256, k = 2: 2 x 161 -> 1 x ~170 us per generation)
    if (D == 256) return (want_jac || k < 2) ? 1 : ko256;
    return k >= 2 ? (D == 128 ? ko128 : 2) : 1;
}

template <int KID, bool FAST, int KOUT, int DT, bool QLDS>
static int launch_fused3(mrbf_ctx *ctx, bool want_jac, bool final_, dim3 grid, const EvalDesc &one, const EvalDesc *many, int l0) {
#define MRBF_EFL(JACV, FINV) \
    hipLaunchKernelGGL((eval_fused_kernel<KID, FAST, KOUT, DT, JACV, FINV, QLDS>), grid, dim3(256), 0, ctx->stream, one, many, l0)
    if (want_jac && final_)
        MRBF_EFL(true, true);
    else if (want_jac)
        MRBF_EFL(true, false);
    else if (final_)
        MRBF_EFL(false, true);
    else
        MRBF_EFL(false, false);
#undef MRBF_EFL
    return 0;
}
template <int KID, bool FAST, int KOUT, int H, int ECT>
static int launch_split_b(mrbf_ctx *ctx, bool want_jac, bool final_, dim3 grid, const EvalDesc &one, const EvalDesc *many, int l0) {
#define MRBF_EFL(JACV, FINV) \
    hipLaunchKernelGGL((eval_fused_split_kernel<KID, FAST, KOUT, H, ECT, JACV, FINV>), grid, dim3(512), 0, ctx->stream, one, many, l0)
    if (want_jac && final_)
        MRBF_EFL(true, true);
    else if (want_jac)
        MRBF_EFL(true, false);
    else if (final_)
        MRBF_EFL(false, true);
    else
        MRBF_EFL(false, false);
#undef MRBF_EFL
    return 0;
}
template <int KID, int KOUT, int H, int ECT>
static int launch_split(mrbf_ctx *ctx, bool want_jac, bool final_, dim3 grid, const KP &kp, const EvalDesc &one, const EvalDesc *many, int l0) {
    if (kp.fast && (KID == MRBF_MULTIQUADRIC || KID == MRBF_INV_MULTIQUADRIC || KID == MRBF_CUBIC))
        return launch_split_b<KID, true, KOUT, H, ECT>(ctx, want_jac, final_, grid, one, many, l0);
    return launch_split_b<KID, false, KOUT, H, ECT>(ctx, want_jac, final_, grid, one, many, l0);
}

template <int KID, int KOUT, int H, int ECT>
static int launch_split_vals(mrbf_ctx *ctx, bool final_, dim3 grid, const KP &kp, const EvalDesc &one, const EvalDesc *many, int l0) {
    const bool fast = kp.fast && (KID == MRBF_MULTIQUADRIC || KID == MRBF_INV_MULTIQUADRIC || KID == MRBF_CUBIC);
    static const int wide = mrbf_env("MRBF_EVAL_WIDE256") ? atoi(mrbf_env("MRBF_EVAL_WIDE256")) : 1;
    if (H == 128 && !final_ && wide) {  // values only, split centre range: one wave per sixteen queries with all 256 coordinates
        const dim3 g2((grid.x + 1) / 2, grid.y, grid.z);
        if (fast)
            hipLaunchKernelGGL((eval_vals256_kernel<KID, true, KOUT>), g2, dim3(512), 0, ctx->stream, one, many, l0);
        else
            hipLaunchKernelGGL((eval_vals256_kernel<KID, false, KOUT>), g2, dim3(512), 0, ctx->stream, one, many, l0);
        return 0;
    }
#define MRBF_EFL(FASTV, FINV) \
    hipLaunchKernelGGL((eval_fused_split_kernel<KID, FASTV, KOUT, H, ECT, false, FINV>), grid, dim3(512), 0, ctx->stream, one, many, l0)
    if (fast && final_)
        MRBF_EFL(true, true);
    else if (fast)
        MRBF_EFL(true, false);
    else if (final_)
        MRBF_EFL(false, true);
    else
        MRBF_EFL(false, false);
#undef MRBF_EFL
    return 0;
}

template <int KID, int KOUT, int DT, bool QLDS>
static int launch_fused(mrbf_ctx *ctx, bool want_jac, bool final_, dim3 grid, const KP &kp, const EvalDesc &one, const EvalDesc *many, int l0) {
    if (kp.fast && (KID == MRBF_MULTIQUADRIC || KID == MRBF_INV_MULTIQUADRIC || KID == MRBF_CUBIC))
        return launch_fused3<KID, true, KOUT, DT, QLDS>(ctx, want_jac, final_, grid, one, many, l0);
    return launch_fused3<KID, false, KOUT, DT, QLDS>(ctx, want_jac, final_, grid, one, many, l0);
}

template <int KOUT>
static void launch_combine(mrbf_ctx *ctx, bool want_jac, dim3 cgrid, const EvalDesc &one, const EvalDesc *many, int l0, int D) {
    static const int vals_combine = mrbf_env("MRBF_EVAL_COMBINE_VALS") ? atoi(mrbf_env("MRBF_EVAL_COMBINE_VALS")) : 1;
    if (!want_jac && vals_combine)
        hipLaunchKernelGGL(eval_combine_vals_kernel<KOUT>, dim3((cgrid.x + 3) / 4, cgrid.y), dim3(256), 0, ctx->stream, one, many, l0);
    else
        hipLaunchKernelGGL(eval_combine_kernel<KOUT>, cgrid, dim3(128), 0, ctx->stream, one, many, l0, D);
}

// the passes over the outputs of one evaluation (single: many == nullptr, `one` by value; batch: blockIdx.z / .y = problem)
static int run_passes(mrbf_ctx *ctx, const KP &kp, int D, int k, bool want_jac, bool final_, dim3 grid, dim3 cgrid, const EvalDesc &one,
                      const EvalDesc *many) {
    const int KO = outputs_per_pass(k, D, want_jac);
    for (int l0 = 0; l0 < k; l0 += KO) {
        const int ko = std::min(KO, k - l0);
#define MRBF_EF(KOV, DTV, QL) MRBF_DISPATCH_KID(kp.kid, MRBF_TRY((launch_fused<KID, KOV, DTV, QL>(ctx, want_jac, final_, grid, kp, one, many, l0))))
        static const int split128 = mrbf_env("MRBF_EVAL_SPLIT128") ? atoi(mrbf_env("MRBF_EVAL_SPLIT128")) : 1;
#define MRBF_EF128(KOV) MRBF_DISPATCH_KID(kp.kid, MRBF_TRY((launch_split<KID, KOV, 64, 64>(ctx, want_jac, final_, grid, kp, one, many, l0))))
#define MRBF_EF256(KOV) MRBF_DISPATCH_KID(kp.kid, MRBF_TRY((launch_split<KID, KOV, 128, 32>(ctx, want_jac, final_, grid, kp, one, many, l0))))
        if (D == 256 && ko == 2) {  // (values only)
            MRBF_DISPATCH_KID(kp.kid, MRBF_TRY((launch_split_vals<KID, 2, 128, 32>(ctx, final_, grid, kp, one, many, l0))));
            if (!final_) launch_combine<2>(ctx, want_jac, cgrid, one, many, l0, D);
        } else if (D == 256) {
            if (!want_jac) {
                MRBF_DISPATCH_KID(kp.kid, MRBF_TRY((launch_split_vals<KID, 1, 128, 32>(ctx, final_, grid, kp, one, many, l0))));
            } else {
                MRBF_EF256(1);
            }
            if (!final_) launch_combine<1>(ctx, want_jac, cgrid, one, many, l0, D);
        } else if (ko == 2) {
            if (D == 64) {
                MRBF_EF(2, 4, false);
            } else if (split128) {
                MRBF_EF128(2);
            } else {
                MRBF_EF(2, 8, true);
            }
            if (!final_) launch_combine<2>(ctx, want_jac, cgrid, one, many, l0, D);
        } else {
            if (D == 64) {
                MRBF_EF(1, 4, false);
            } else if (split128) {
                MRBF_EF128(1);
            } else {
                MRBF_EF(1, 8, false);
            }
            if (!final_) launch_combine<1>(ctx, want_jac, cgrid, one, many, l0, D);
        }
#undef MRBF_EF
#undef MRBF_EF128
#undef MRBF_EF256
    }
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

int center_pad_batch(mrbf_ctx *ctx, const EvalDesc *dev_descs, int count, int64_t max_mpad) {
    if (count <= 0 || max_mpad <= 0) return 0;
    hipLaunchKernelGGL(center_pad_batch_kernel, dim3((unsigned)((max_mpad + 3) / 4), (unsigned)count), dim3(256), 0, ctx->stream, dev_descs, 0);
    MRBF_HIP(ctx, hipGetLastError());
    return 0;
}

int eval_fused_batch(mrbf_ctx *ctx, const KP &kp, int D, int k, bool want_jac, const EvalDesc *host_descs, const EvalDesc *dev_descs, int count,
                     bool centred) {
    if (count <= 0) return 0;
    if (D != 64 && D != 128 && D != 256) return fail(ctx, MRBF_EHIP, "eval_fused_batch needs dpad in {64, 128, 256}");
    int64_t max_mpad = 0, max_m = 0;
    int max_split = 1;
    for (int p = 0; p < count; ++p) {
        max_mpad = std::max(max_mpad, host_descs[p].mpad);
        max_m = std::max(max_m, host_descs[p].m);
        max_split = std::max(max_split, host_descs[p].nsplit);
    }
    // one launch per group: either every member finishes inside the kernel (no split of the centre range anywhere) or every member
    // carries partial buffers -- a group that mixes the two would run the partial-writing kernels on members without buffers
    // (batch.hip sizes them by the member's own split and keeps the two kinds in different groups); refused otherwise
    if (max_split > 1)
        for (int p = 0; p < count; ++p)
            if (host_descs[p].nsplit == 1 && host_descs[p].m > 0)
                return fail(ctx, MRBF_EHIP, "eval_fused_batch: descriptor %d is unsplit in a group that splits the centre range", p);
    if (!centred)
        hipLaunchKernelGGL(center_pad_batch_kernel, dim3((unsigned)((max_mpad + 3) / 4), (unsigned)count), dim3(256), 0, ctx->stream, dev_descs, D);
    dim3 grid((unsigned)(max_mpad / EQ), (unsigned)max_split, (unsigned)count);
    dim3 cgrid((unsigned)max_m, (unsigned)count);
    return run_passes(ctx, kp, D, k, want_jac, max_split == 1, grid, cgrid, host_descs[0], dev_descs);
}

int eval_fused(mrbf_ctx *ctx, const mrbf_model *M, int64_t m, const double *X, double *vals, double *jac, mrbf_eval_info *info) {
    const int d = M->d, k = M->k, q = M->q;
    const int D = (M->dpad <= 64) ? 64 : (M->dpad <= 128 ? 128 : 256);
    if (M->dpad > 256) return fail(ctx, MRBF_EHIP, "eval_fused supports d <= 256");
    if (D != M->dpad) return fail(ctx, MRBF_EHIP, "eval_fused needs dpad in {64, 128, 256} (got %d)", M->dpad);
    const int64_t mpad = round_up(m, EQ);
    // (n = 2d + 1 = 257 sites are five tiles, not the six of the 128-padded storage: a sixth of a C4 evaluation)
    const int ntiles = (int)((M->n + EC - 1) / EC);
    const int nsplit = eval_nsplit(ctx, m, ntiles, ctx->eval_check_call != 0);
    const int KO = outputs_per_pass(k, D, jac != nullptr);
    EvalDesc E;
    std::memset(&E, 0, sizeof(E));
    E.X = X;
    E.mean = M->mean;
    MRBF_TRY(get_buf(ctx, S_EVAL_XC, (size_t)mpad * D, &E.Xq));
    MRBF_TRY(get_buf(ctx, S_EVAL_XSQ, (size_t)mpad, &E.xsq));
    E.Cc = M->Xc;
    E.csq = M->sq;
    E.Wc = M->Wc;
    E.lam = M->lam;
    E.npad = M->npad;
    E.mpad = mpad;
    E.m = m;
    E.d = d;
    E.k = k;
    E.q = q;
    E.nsplit = nsplit;
    E.ntiles = ntiles;
    E.nsub = (int)((M->n + 15) / 16);
    E.tiles_per_split = (ntiles + nsplit - 1) / nsplit;
    E.kp = M->kp;
    if (nsplit > 1) {
        MRBF_TRY(get_buf(ctx, S_EVAL_SA, (size_t)nsplit * mpad * KO * 2, &E.vpart));
        E.sapart = E.vpart + (size_t)nsplit * mpad * KO;
        if (jac) MRBF_TRY(get_buf(ctx, S_EVAL_J, (size_t)nsplit * mpad * KO * D, &E.gpart));
    }
    E.vals = vals;
    E.jac = jac;
    const bool timing = ctx->timing && info;
    if (timing) MRBF_HIP(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    // (the PS solver's breeding kernel writes its offspring centred and padded straight into these buffers -- same arithmetic as the
    //  centring kernel, one launch per generation less; any other caller, or a buffer that moved, takes the launch)
    if (!(ctx->eval_pre_xq && ctx->eval_pre_xq == E.Xq)) MRBF_TRY(launch_center_pad(ctx, X, m, d, M->mean, nullptr, E.Xq, mpad, D, E.xsq));
    dim3 grid((unsigned)(mpad / EQ), (unsigned)nsplit);
    MRBF_TRY(run_passes(ctx, M->kp, D, k, jac != nullptr, nsplit == 1, grid, dim3((unsigned)m), E, nullptr));
    if (timing) {
        MRBF_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
        MRBF_HIP(ctx, hipEventSynchronize(ctx->ev[7]));
        float t;
        MRBF_HIP(ctx, hipEventElapsedTime(&t, ctx->ev[4], ctx->ev[7]));
        info->ms_total = t;
        info->ms_kernel = t;
    }
    return 0;
}

}  // namespace mrbf
