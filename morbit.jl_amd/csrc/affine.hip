// Scores of the affinely-independent-point filter for ALL candidates at once -- replaces the candidate loop of
// Base.iterate(::AffinelyIndependentPointFilter, num_found) (/root/reference/src/models/AffinelyIndependentPoints.jl:71-106,
// used by _find_suitable_points, src/models/RbfModel.jl:205-238): for every candidate xi of the database box
//     val(xi) = || Z (Z' (xi - x0)) ||_p ,   p = inf (the filter's default, RbfModel.jl:226) or 2,
// and the FIRST maximiser (the reference's `>` scan keeps the earliest).  Z (d x dz, the p-normalised complement basis of the
// directions chosen so far) is recomputed by the caller after every pick -- a d x d QR, SURVEY.md section 8 row a12 -- the two
// tall products and the reduction over up to 10^5..10^6 database sites run here.
#include "common.hpp"

namespace mrbf {

// val[c] = norm_p(U[:, c]), U d x mc column-major
__global__ __launch_bounds__(256) void col_norms_kernel(const double *__restrict__ U, int d, int64_t mc, int use_inf, double *__restrict__ val) {
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= mc) return;
    double s = 0.0;
    for (int i = lane; i < d; i += 64) {
        const double u = fabs(U[i + c * d]);
        s = use_inf ? fmax(s, u) : fma(u, u, s);
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(s, off);
        s = use_inf ? fmax(s, o) : s + o;
    }
    if (lane == 0) val[c] = use_inf ? s : sqrt(s);
}

// first maximiser: out[0] = index, out[1] = value (as doubles); single workgroup, ties -> smallest index
__global__ __launch_bounds__(1024) void argmax_first_kernel(const double *__restrict__ val, int64_t mc, double *__restrict__ out) {
    __shared__ double sv[1024];
    __shared__ long long si[1024];
    double bv = -INFINITY;
    long long bi = -1;
    for (int64_t c = threadIdx.x; c < mc; c += 1024) {
        const double v = val[c];
        if (v > bv) {  // ascending c per thread: `>` keeps the first maximiser of the thread's subsequence
            bv = v;
            bi = c;
        }
    }
    sv[threadIdx.x] = bv;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const double ov = sv[threadIdx.x + w];
            const long long oi = si[threadIdx.x + w];
            if (oi >= 0 && (si[threadIdx.x] < 0 || ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (double)si[0];
        out[1] = sv[0];
    }
}


// ---- the WHOLE pick loop on the device (round 6) ---------------------------------------------------------------------------------
// After a pick the reference re-factors Y = [picked directions] from scratch (qr(Y), AffinelyIndependentPoints.jl:4-11, :93-94) and
// scans the candidates again.  Here the Householder factorisation grows by ONE reflector per pick and everything stays on the device:
//   Q   d x d  the full orthogonal factor (columns j.. = the unnormalised complement basis);  C = Q' S  the candidates in that basis
//   pick i  ->  x = C[j:, i]  gives the reflector (larfg: beta = -sign(x_0) |x|, v_0 = 1);  Q[:, j:] <- Q[:, j:] H,  C[j:, :] <- H C[j:, :]
//   scores  ->  U = Zs C[j:, :] with Zs = Q[:, j:] D^2 (D = 1 / column p-norms: the reference's normalised Z enters as Z Z'), val = |U|_p by column
// No host round trip between picks (the host only looks at the "done" word every sixteen picks); same reflectors as LAPACK's geqrf, so
// the picks are the reference's (ties aside, which need scores equal to the last bit).
struct AffSel {
    double *Q, *C, *Zs, *U, *val, *hv;  // hv: [0] tau, [1 .. d] v (over the trailing rows), [1 + d .. 2 d] w = Q[:, j:] v
    long long *picks;
    int *state;                         // [0] done, [1] picks so far, [2] ticket
    unsigned long long *stamps;         // diagnostic (MRBF_AFFINE_STAMPS=1): 100 MHz clock at the phases of the last workgroup of a launch
    int d, use_inf;
    int64_t mc;
    double pivot;
};

// val[c] = | Zs C[j:, c] |_p for eight candidates per workgroup (no rocBLAS inside the pick loop: its lazily loaded kernels cost 50-90 ms the
// first time a new shape class turns up -- three such hiccups in the twenty iterations of the rehearsal).  Thread t owns rows t, t + 256, ...
// of the eight products; Zs is read column by column (coalesced), the candidates' trailing coordinates sit in LDS.
constexpr int AFF_G = 8;
__global__ __launch_bounds__(256) void affsel_score_kernel(AffSel a, int j) {
    extern __shared__ double sx[];  // [AFF_G][dz]
    __shared__ double red[256];
    const int tid = threadIdx.x, d = a.d, dz = d - j;
    if (a.state[0]) return;
    const int64_t c0 = (int64_t)blockIdx.x * AFF_G;
    for (int e = tid; e < AFF_G * dz; e += 256) {
        const int g = e / dz, k = e % dz;
        sx[e] = c0 + g < a.mc ? a.C[(size_t)(c0 + g) * d + j + k] : 0.0;
    }
    __syncthreads();
    double nrm[AFF_G];
#pragma unroll
    for (int g = 0; g < AFF_G; ++g) nrm[g] = 0.0;
    for (int r = tid; r < d; r += 256) {
        double u[AFF_G];
#pragma unroll
        for (int g = 0; g < AFF_G; ++g) u[g] = 0.0;
        for (int k = 0; k < dz; ++k) {
            const double z = a.Zs[(size_t)k * d + r];
#pragma unroll
            for (int g = 0; g < AFF_G; ++g) u[g] = fma(z, sx[g * dz + k], u[g]);
        }
#pragma unroll
        for (int g = 0; g < AFF_G; ++g) nrm[g] = a.use_inf ? fmax(nrm[g], fabs(u[g])) : fma(u[g], u[g], nrm[g]);
    }
    for (int g = 0; g < AFF_G; ++g) {
        __syncthreads();
        red[tid] = nrm[g];
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] = a.use_inf ? fmax(red[tid], red[tid + s2]) : red[tid] + red[tid + s2];
            __syncthreads();
        }
        if (tid == 0 && c0 + g < a.mc) a.val[c0 + g] = a.use_inf ? red[0] : sqrt(red[0]);
    }
}

// C[:, c] = Q' s_c for eight candidates per workgroup (once per call): thread i owns row i of the eight products
__global__ __launch_bounds__(256) void affsel_project_kernel(AffSel a, const double *__restrict__ S) {
    extern __shared__ double sx[];  // [AFF_G][d]
    const int tid = threadIdx.x, d = a.d;
    const int64_t c0 = (int64_t)blockIdx.x * AFF_G;
    for (int e = tid; e < AFF_G * d; e += 256) {
        const int g = e / d, k = e % d;
        sx[e] = c0 + g < a.mc ? S[(size_t)(c0 + g) * d + k] : 0.0;
    }
    __syncthreads();
    for (int i = tid; i < d; i += 256) {
        double u[AFF_G];
#pragma unroll
        for (int g = 0; g < AFF_G; ++g) u[g] = 0.0;
        const double *q = a.Q + (size_t)i * d;
        for (int r = 0; r < d; ++r) {
            const double z = q[r];
#pragma unroll
            for (int g = 0; g < AFF_G; ++g) u[g] = fma(z, sx[g * d + r], u[g]);
        }
#pragma unroll
        for (int g = 0; g < AFF_G; ++g)
            if (c0 + g < a.mc) a.C[(size_t)(c0 + g) * d + i] = u[g];
    }
}

// ONE launch per pick (inf-norm, d < 1024).  Every workgroup first brings its eight candidates up to date with the reflector of the pick
// before (C[jq:, c] <- H C[jq:, c], jq = j - 1), scores them against Zs, and the LAST workgroup to finish (a ticket) takes the decision for
// the whole grid and prepares the next launch: the new pick's reflector, Q[:, j:] <- Q[:, j:] H in place (every reader of this launch has
// finished), the new Zs columns with their norms, the picked candidate zeroed.  No assumption about residency (the last-block pattern is
// safe at any occupancy); the maximum over a column is order-independent, so the step is reproducible whichever workgroup comes last.
__device__ __forceinline__ double wave_max(double v) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
constexpr int AFF_NT = 1024;  // threads of the one-launch-per-pick kernel: the last workgroup's tail (d x (d - j) elements, three passes) is
                              // latency-bound on one CU -- sixteen waves and eight independent loads per thread hide it
__global__ __launch_bounds__(AFF_NT) void affsel_pick_kernel(AffSel a, int j, int apply) {
    extern __shared__ double sx[];  // [AFF_G][dz + 1]: the candidates' coordinates from row j - 1 on (last block afterwards: column maxima)
    __shared__ double red[AFF_NT];
    __shared__ long long redi[AFF_NT];
    __shared__ double s_scal[4];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, d = a.d, dz = d - j, ldx = dz + 1;
    if (a.state[0]) return;
    unsigned long long st[12];
    int nst = 0;
#define AFF_STAMP() do { if (a.stamps && tid == 0 && nst < 12) st[nst++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    AFF_STAMP();
    const int64_t c0 = (int64_t)blockIdx.x * AFF_G;
    const int jq = j - 1;
    for (int e = tid; e < AFF_G * ldx; e += AFF_NT) {
        const int g = e / ldx, k = e % ldx;  // k = 0 is row j - 1 (only read when a reflector is pending)
        sx[e] = (c0 + g < a.mc && (k > 0 || apply)) ? a.C[(size_t)(c0 + g) * d + jq + k] : 0.0;
    }
    __syncthreads();
    if (apply) {  // x <- x - tau v (v' x) over rows jq .. d - 1, 32 lanes per candidate
        if (tid < 32 * AFF_G) {
            const double tau = a.hv[0];
            const double *v = a.hv + 1;
            const int g = tid >> 5, l = tid & 31;
            double dot = 0.0;
            for (int k8 = l; k8 < ldx; k8 += 32 * 8) {
                double vv[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) vv[t] = k8 + 32 * t < ldx ? v[k8 + 32 * t] : 0.0;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (k8 + 32 * t < ldx) dot = fma(vv[t], sx[g * ldx + k8 + 32 * t], dot);
            }
            for (int off = 16; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
            const double td = tau * dot;
            for (int k = l; k < ldx; k += 32) {
                const double xn = fma(-td, v[k], sx[g * ldx + k]);
                sx[g * ldx + k] = xn;
                if (c0 + g < a.mc) __hip_atomic_store(&a.C[(size_t)(c0 + g) * d + jq + k], xn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (write-through: the last workgroup may sit behind another L2)
            }
        }
        __syncthreads();
    }
    AFF_STAMP();  // 1: candidates loaded, pending reflector applied
    // scores: rows r by thread, the k range split over the thread groups beyond the first d threads (d = 128: eight groups of sixteen k)
    int dr = 64;
    while (dr < d) dr <<= 1;            // rows rounded up to a power of two (<= 1024): thread = (row, chunk)
    const int nch = AFF_NT / dr, row = tid & (dr - 1), ch = tid / dr;
    {
        const int kw = (dz + nch - 1) / nch, k0 = ch * kw, k1 = min(dz, k0 + kw);
        double u[AFF_G];
#pragma unroll
        for (int g = 0; g < AFF_G; ++g) u[g] = 0.0;
        if (row < d)
            for (int k8 = k0; k8 < k1; k8 += 8) {  // eight independent loads per batch (a load per iteration is a memory round trip each)
                double z[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) z[t] = k8 + t < k1 ? a.Zs[(size_t)(k8 + t) * d + row] : 0.0;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (k8 + t < k1) {
#pragma unroll
                        for (int g = 0; g < AFF_G; ++g) u[g] = fma(z[t], sx[g * ldx + 1 + k8 + t], u[g]);
                    }
            }
        AFF_STAMP();  // 2: products
        // sum over the chunks (fixed order), max over the rows -- all eight candidates in ONE round through LDS (8 x 1024 partial sums behind
        // the candidates' coordinates), wave shuffles for the maxima: three barriers (a block-wide tree per candidate was 112)
        double *ps = sx + (size_t)AFF_G * (d + 1);   // [AFF_G][AFF_NT]
        const int wave = tid >> 6;
        __syncthreads();
#pragma unroll
        for (int g = 0; g < AFF_G; ++g) ps[g * AFF_NT + tid] = u[g];
        __syncthreads();
        double nv[AFF_G];
#pragma unroll
        for (int g = 0; g < AFF_G; ++g) nv[g] = 0.0;
        if (tid < dr) {
#pragma unroll
            for (int g = 0; g < AFF_G; ++g) {
                double sum = 0.0;
                for (int c2 = 0; c2 < nch; ++c2) sum += ps[g * AFF_NT + c2 * dr + tid];
                nv[g] = tid < d ? fabs(sum) : 0.0;
            }
        }
        if (tid < dr) {  // (whole waves: dr is a multiple of 64)
#pragma unroll
            for (int g = 0; g < AFF_G; ++g) nv[g] = wave_max(nv[g]);
        }
        __syncthreads();
        if (lane == 0 && tid < dr) {
#pragma unroll
            for (int g = 0; g < AFF_G; ++g) red[g * 16 + wave] = nv[g];
        }
        __syncthreads();
        if (tid < AFF_G && c0 + tid < a.mc) {
            double m = 0.0;
            for (int w2 = 0; w2 < dr / 64; ++w2) m = fmax(m, red[tid * 16 + w2]);
            __hip_atomic_store(a.val + c0 + tid, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    AFF_STAMP();  // 3: scores reduced and stored
    // (the codebase's hand-off: write-through stores counted out of vmcnt, workgroup barrier, ONE agent-scope atomic; the reader drops its
    // own caches with an acquire fence and reads what others wrote with coherent loads -- no agent-scope release: that writes back the L2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(a.state + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    AFF_STAMP();  // 4: ticket taken (last workgroup from here on)
    if (tid == 0) a.state[2] = 0;
    double bv = -INFINITY;
    long long bi = -1;
    for (int64_t c = tid; c < a.mc; c += AFF_NT) {
        const double v = __hip_atomic_load(a.val + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v > bv) {
            bv = v;
            bi = c;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {  // first maximiser inside the wave, then over the sixteen waves
        const double ov = __shfl_xor(bv, off);
        const long long oi = __shfl_xor(bi, off);
        if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    __syncthreads();
    if (lane == 0) {
        red[tid >> 6] = bv;
        redi[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w2 = 1; w2 < AFF_NT / 64; ++w2) {
            const double ov = red[w2];
            const long long oi = redi[w2];
            if (oi >= 0 && (redi[0] < 0 || ov > red[0] || (ov == red[0] && oi < redi[0]))) {
                red[0] = ov;
                redi[0] = oi;
            }
        }
    }
    __syncthreads();
    const long long best = redi[0];
    const double bestv = red[0];
    __syncthreads();
    AFF_STAMP();  // 5: first maximiser
    if (best < 0 || !(bestv > a.pivot)) {
        if (tid == 0) a.state[0] = 1;
        return;
    }
    double *x = a.C + (size_t)best * d + j;
    double part = 0.0;
    for (int c = 1 + tid; c < dz; c += AFF_NT) {
        const double xv = __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        part = fma(xv, xv, part);
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    __syncthreads();
    if (lane == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        double ssq = 0.0;
        for (int w2 = 0; w2 < AFF_NT / 64; ++w2) ssq += red[w2];
        const double alpha = __hip_atomic_load(x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), xn = sqrt(ssq);
        double tau = 0.0, scale = 0.0;
        if (xn != 0.0) {
            const double beta = -copysign(hypot(alpha, xn), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        s_scal[0] = tau;
        s_scal[1] = scale;
        a.hv[0] = tau;
        a.picks[a.state[1]] = best;
        a.state[1] += 1;
    }
    __syncthreads();
    const double tau = s_scal[0], scale = s_scal[1];
    double *v = a.hv + 1, *w = a.hv + 1 + d;
    unsigned long long *cmax = reinterpret_cast<unsigned long long *>(sx);  // (non-negative doubles order like their bit patterns)
    double *sv = sx + dz;                                                    // v in LDS behind the column maxima (2 dz <= AFF_G (dz + 1))
    for (int c = tid; c < dz; c += AFF_NT) {
        const double vc = c == 0 ? 1.0 : __hip_atomic_load(x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * scale;
        v[c] = vc;
        sv[c] = vc;
        cmax[c] = 0ull;
    }
    __syncthreads();
    AFF_STAMP();  // 6: reflector
    // thread = (row, column chunk): w = Q[:, j:] v as partial sums over the chunks, then Q[:, j:] <- Q[:, j:] - tau w v' with the column maxima
    // of the NEW trailing columns j + 1 .., then their doubly normalised copies -- eight independent loads per thread and batch
    const int cw = (dz + nch - 1) / nch, cb = ch * cw, ce = min(dz, cb + cw);
    double acc = 0.0;
    if (row < d)
        for (int c8 = cb; c8 < ce; c8 += 8) {
            double q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = c8 + u < ce ? a.Q[(size_t)(j + c8 + u) * d + row] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma(q[u], c8 + u < ce ? sv[c8 + u] : 0.0, acc);
        }
    red[tid] = acc;
    __syncthreads();
    if (tid < dr) {
        double sum = 0.0;
        for (int c2 = 0; c2 < nch; ++c2) sum += red[c2 * dr + tid];
        if (tid < d) w[tid] = sum;
        redi[tid] = __double_as_longlong(sum);  // (w for every chunk's threads)
    }
    __syncthreads();
    AFF_STAMP();  // 7: w
    const double tw = tau * __longlong_as_double(redi[row]);
    for (int c8 = cb; c8 < ce; c8 += 8) {  // (uniform per wave: a wave holds 64 consecutive rows of ONE chunk)
        double q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = (row < d && c8 + u < ce) ? a.Q[(size_t)(j + c8 + u) * d + row] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c8 + u;
            double e = 0.0;
            if (row < d && c < ce) {
                e = fma(-tw, sv[c], q[u]);
                a.Q[(size_t)(j + c) * d + row] = e;
            }
            q[u] = e;
            if (c < ce && c > 0) {
                const double m = wave_max(fabs(e));
                if (lane == 0) atomicMax(&cmax[c], (unsigned long long)__double_as_longlong(m));
            }
        }
    }
    __syncthreads();
    AFF_STAMP();  // 8: Q updated, column maxima
    if (row < d)
        for (int c8 = max(cb, 1); c8 < ce; c8 += 8) {
            double q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = c8 + u < ce ? a.Q[(size_t)(j + c8 + u) * d + row] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (c8 + u < ce) {
                    const double nrmc = __longlong_as_double((long long)cmax[c8 + u]);
                    const double s2n = 1.0 / (nrmc * nrmc);  // (the same expression as affsel_update_kernel: Z Z' = Q2 D^2 Q2')
                    a.Zs[(size_t)(c8 + u - 1) * d + row] = q[u] * s2n;
                }
        }
    for (int r = tid; r < d; r += AFF_NT) a.C[(size_t)best * d + r] = 0.0;
    AFF_STAMP();  // 9: Zs written
    if (a.stamps && tid == 0)
        for (int q2 = 0; q2 < nst; ++q2) a.stamps[q2] = st[q2];
#undef AFF_STAMP
}

// one workgroup: first maximiser of val, pivot test, the pick's reflector and w; the picked candidate becomes the zero vector
__global__ __launch_bounds__(1024) void affsel_decide_kernel(AffSel a, int j) {
    __shared__ double sv[1024];
    __shared__ long long si[1024];
    __shared__ double s_red[1024];
    const int tid = threadIdx.x, d = a.d, dz = d - j;
    if (a.state[0]) return;
    double bv = -INFINITY;
    long long bi = -1;
    for (int64_t c = tid; c < a.mc; c += 1024) {
        const double v = a.val[c];
        if (v > bv) {
            bv = v;
            bi = c;
        }
    }
    sv[tid] = bv;
    si[tid] = bi;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) {
            const double ov = sv[tid + w];
            const long long oi = si[tid + w];
            if (oi >= 0 && (si[tid] < 0 || ov > sv[tid] || (ov == sv[tid] && oi < si[tid]))) {
                sv[tid] = ov;
                si[tid] = oi;
            }
        }
        __syncthreads();
    }
    const long long best = si[0];
    const double bestv = sv[0];
    __syncthreads();
    if (best < 0 || !(bestv > a.pivot)) {  // "No point was sufficiently linearly independent" (AffinelyIndependentPoints.jl:104)
        if (tid == 0) a.state[0] = 1;
        return;
    }
    double *x = a.C + (size_t)best * d + j;  // the new direction in the current basis, trailing part (dz entries)
    double part = 0.0;
    for (int c = 1 + tid; c < dz; c += 1024) part = fma(x[c], x[c], part);
    s_red[tid] = part;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) s_red[tid] += s_red[tid + w];
        __syncthreads();
    }
    const double alpha = x[0], xn = sqrt(s_red[0]);
    double tau = 0.0, scale = 0.0;
    if (xn != 0.0) {
        const double beta = -copysign(hypot(alpha, xn), alpha);
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
    }
    __syncthreads();
    double *v = a.hv + 1, *w = a.hv + 1 + d;
    for (int c = tid; c < dz; c += 1024) v[c] = c == 0 ? 1.0 : x[c] * scale;
    if (tid == 0) {
        a.hv[0] = tau;
        a.picks[a.state[1]] = best;
        a.state[1] += 1;
    }
    __syncthreads();
    // w = Q[:, j:] v (d x dz), row r by thread r (column-major Q: consecutive threads read consecutive rows)
    for (int r = tid; r < d; r += 1024) {
        double acc = 0.0;
        for (int c = 0; c < dz; ++c) acc = fma(a.Q[(size_t)(j + c) * d + r], v[c], acc);
        w[r] = acc;
    }
    __syncthreads();
    for (int r = tid; r < d; r += 1024) a.C[(size_t)best * d + r] = 0.0;  // chosen sites score 0 from now on (the reference drops them from the list)
}

// blocks 0 .. d - jq - 1: column jq + b of Q gets the reflector (apply != 0) and, for columns >= jz, its p-normalised-twice copy goes to Zs;
// the blocks behind them: four candidate columns of C each (C[jq:, col] <- H C[jq:, col])
__global__ __launch_bounds__(256) void affsel_update_kernel(AffSel a, int jq, int jz, int apply) {
    __shared__ double red[256];
    const int tid = threadIdx.x, d = a.d, ncol = d - jq;
    if (a.state[0]) return;
    const double tau = apply ? a.hv[0] : 0.0;
    const double *v = a.hv + 1, *w = a.hv + 1 + d;
    if ((int)blockIdx.x < ncol) {
        const int c = blockIdx.x;
        double *q = a.Q + (size_t)(jq + c) * d;
        const double tv = apply ? tau * v[c] : 0.0;
        double m = 0.0;
        for (int r = tid; r < d; r += 256) {
            double e = q[r];
            if (apply) {
                e = fma(-tv, w[r], e);
                q[r] = e;
            }
            m = a.use_inf ? fmax(m, fabs(e)) : fma(e, e, m);
        }
        if (jq + c < jz) return;
        red[tid] = m;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] = a.use_inf ? fmax(red[tid], red[tid + s2]) : red[tid] + red[tid + s2];
            __syncthreads();
        }
        const double nrm = a.use_inf ? red[0] : sqrt(red[0]);
        const double s2n = 1.0 / (nrm * nrm);  // Z Z' = Q2 D^2 Q2'
        double *z = a.Zs + (size_t)(jq + c - jz) * d;
        for (int r = tid; r < d; r += 256) z[r] = q[r] * s2n;
        return;
    }
    if (!apply) return;
    const int64_t col = ((int64_t)blockIdx.x - ncol) * 4 + (tid >> 6);
    const int lane = tid & 63;
    if (col >= a.mc) return;
    double *x = a.C + (size_t)col * d + jq;
    double dot = 0.0;
    for (int c = lane; c < ncol; c += 64) dot = fma(v[c], x[c], dot);
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    const double td = tau * dot;
    for (int c = lane; c < ncol; c += 64) x[c] = fma(-td, v[c], x[c]);
}

}  // namespace mrbf

using namespace mrbf;

extern "C" int32_t mrbf_affine_scores(mrbf_ctx *ctx, int64_t mc, int32_t d, int32_t dz, const double *shifted, const double *Z, int32_t p_is_inf,
                                      double *vals_out, int64_t *argmax, double *maxval) {
    if (!ctx) return -1;
    if (mc < 0 || mc > ((int64_t)1 << 26)) return fail(ctx, -2, "mc out of range");
    if (d < 1 || d > 4096) return fail(ctx, -3, "d out of range");
    if (dz < 0 || dz > d) return fail(ctx, -4, "dz must lie in [0, d]");
    if (argmax) *argmax = -1;
    if (maxval) *maxval = -INFINITY;
    if (mc == 0) return MRBF_OK;
    if (!shifted) return fail(ctx, -5, "shifted is NULL");
    if (dz > 0 && !Z) return fail(ctx, -6, "Z is NULL");
    (void)hipSetDevice(ctx->device);
    const double *S, *Zd = nullptr;
    double *T1, *U, *val, *res;
    MRBF_TRY(stage_in(ctx, S_STAGE_A, shifted, (size_t)mc * d, &S));
    MRBF_TRY(get_buf(ctx, S_STAGE_C, (size_t)std::max(dz, 1) * mc, &T1));
    MRBF_TRY(get_buf(ctx, S_STAGE_D, (size_t)d * mc, &U));
    MRBF_TRY(get_buf(ctx, S_EVAL_SA, (size_t)mc, &val));
    MRBF_TRY(get_buf(ctx, S_MISC, (size_t)8, &res));
    if (dz > 0) {
        MRBF_TRY(stage_in(ctx, S_STAGE_B, Z, (size_t)d * dz, &Zd));
        const double one = 1.0, zero = 0.0;
        // the candidates, mc x d row-major, are S' (d x mc) column-major:  T1 = Z' S' (dz x mc),  U = Z T1 (d x mc)
        MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, dz, (int)mc, d, &one, Zd, d, S, d, &zero, T1, dz));
        MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, d, (int)mc, dz, &one, Zd, d, T1, dz, &zero, U, d));
    } else {
        MRBF_HIP(ctx, hipMemsetAsync(U, 0, (size_t)d * mc * sizeof(double), ctx->stream));  // empty complement: every score is 0
    }
    hipLaunchKernelGGL(col_norms_kernel, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, ctx->stream, U, d, mc, p_is_inf ? 1 : 0, val);
    hipLaunchKernelGGL(argmax_first_kernel, dim3(1), dim3(1024), 0, ctx->stream, val, mc, res);
    MRBF_HIP(ctx, hipGetLastError());
    double h[2] = {-1.0, 0.0};
    MRBF_HIP(ctx, hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    if (vals_out) MRBF_HIP(ctx, hipMemcpyAsync(vals_out, val, (size_t)mc * sizeof(double), hipMemcpyDefault, ctx->stream));
    MRBF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (argmax) *argmax = (int64_t)h[0];
    if (maxval) *maxval = h[1];
    return MRBF_OK;
}


extern "C" int32_t mrbf_affine_select(mrbf_ctx *ctx, int64_t mc, int32_t d, const double *shifted, int32_t j0, const double *Q0, int32_t max_picks,
                                      double pivot_val, int32_t p_is_inf, int64_t *picked_out, int32_t *n_picked, double *Z_out) {
    if (!ctx) return -1;
    if (mc < 0 || mc > ((int64_t)1 << 26)) return fail(ctx, -2, "mc out of range");
    if (d < 1 || d > 4096) return fail(ctx, -3, "d out of range");
    if (j0 < 0 || j0 > d) return fail(ctx, -5, "j0 must lie in [0, d]");
    if (j0 > 0 && !Q0) return fail(ctx, -6, "Q0 is NULL");
    if (max_picks < 0) return fail(ctx, -7, "max_picks < 0");
    if (!n_picked) return fail(ctx, -11, "n_picked is NULL");
    *n_picked = 0;
    max_picks = std::min(max_picks, d - j0);
    if (mc > 0 && !shifted) return fail(ctx, -4, "shifted is NULL");
    if (max_picks > 0 && !picked_out) return fail(ctx, -10, "picked_out is NULL");
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    AffSel a{};
    a.d = d;
    a.mc = mc;
    a.use_inf = p_is_inf ? 1 : 0;
    a.pivot = pivot_val;
    MRBF_TRY(get_buf(ctx, S_PHI, (size_t)d * d, &a.Q));
    MRBF_TRY(get_buf(ctx, S_STAGE_B, (size_t)d * std::max<int64_t>(mc, 1), &a.C));
    MRBF_TRY(get_buf(ctx, S_G, (size_t)d * d, &a.Zs));
    MRBF_TRY(get_buf(ctx, S_STAGE_D, (size_t)d * std::max<int64_t>(mc, 1), &a.U));
    MRBF_TRY(get_buf(ctx, S_EVAL_SA, (size_t)std::max<int64_t>(mc, 1), &a.val));
    MRBF_TRY(get_buf(ctx, S_RHS, (size_t)2 * d + 8, &a.hv));
    MRBF_TRY(get_buf(ctx, S_IPIV, (size_t)std::max(max_picks, 1) + 2, &a.picks));
    MRBF_TRY(get_buf(ctx, S_INFO, (size_t)4, &a.state));
    MRBF_HIP(ctx, hipMemsetAsync(a.state, 0, 4 * sizeof(int), s));
    static const int want_stamps = mrbf_env("MRBF_AFFINE_STAMPS") ? atoi(mrbf_env("MRBF_AFFINE_STAMPS")) : 0;
    a.stamps = nullptr;
    if (want_stamps) MRBF_TRY(get_buf(ctx, S_MISC, (size_t)16, &a.stamps));
    const double one = 1.0, zero = 0.0;
    std::vector<double> hq;
    if (j0 > 0) {
        const double *Qd;
        MRBF_TRY(stage_in(ctx, S_STAGE_C, Q0, (size_t)d * d, &Qd));
        MRBF_HIP(ctx, hipMemcpyAsync(a.Q, Qd, (size_t)d * d * sizeof(double), hipMemcpyDeviceToDevice, s));
    } else {
        hq.assign((size_t)d * d, 0.0);
        for (int i = 0; i < d; ++i) hq[(size_t)i * d + i] = 1.0;
        MRBF_HIP(ctx, hipMemcpyAsync(a.Q, hq.data(), hq.size() * sizeof(double), hipMemcpyHostToDevice, s));
    }
    int npick = 0;
    if (mc > 0 && max_picks > 0) {
        const double *S;
        MRBF_TRY(stage_in(ctx, S_STAGE_A, shifted, (size_t)mc * d, &S));
        // C = Q' S' (the candidates, mc x d row-major, are S' d x mc column-major)
        const unsigned ngrp = (unsigned)((mc + AFF_G - 1) / AFF_G);
        const size_t shm = (size_t)AFF_G * d * sizeof(double);
        if (shm > 64 * 1024) {  // (d > 1024: beyond the kernels' LDS budget -- rocBLAS for the products, as before)
            MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_transpose, rocblas_operation_none, d, (int)mc, d, &one, a.Q, d, S, d, &zero, a.C, d));
        } else {
            hipLaunchKernelGGL(affsel_project_kernel, dim3(ngrp), dim3(256), shm, s, a, S);
        }
        hipLaunchKernelGGL(affsel_update_kernel, dim3((unsigned)(d - j0)), dim3(256), 0, s, a, j0, j0, 0);  // Zs of the start basis
        int hstate[2] = {0, 0};
        // one launch per pick (affsel_pick_kernel) for the filter's own norm (p = inf) while the candidates' coordinates fit the LDS budget;
        // the three-launch form otherwise (2-norm: its column sums would depend on the order of arrival)
        const size_t shm_pick = (size_t)AFF_G * (d + 1) * sizeof(double);
        static const int fused_env = mrbf_env("MRBF_AFFINE_FUSED") ? atoi(mrbf_env("MRBF_AFFINE_FUSED")) : 1;
        const bool fused = a.use_inf && shm_pick <= 64 * 1024 && fused_env != 0;
        const size_t shm_pick_total = shm_pick + (size_t)AFF_G * AFF_NT * sizeof(double);  // coordinates + the 8 x 1024 partial sums (<= 128 KB of the CU's 160)
        if (fused) MRBF_HIP(ctx, hipFuncSetAttribute((const void *)affsel_pick_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_pick_total));
        for (int t = 0; t < max_picks; ++t) {
            const int j = j0 + t, dz = d - j;
            if (fused) {
                hipLaunchKernelGGL(affsel_pick_kernel, dim3(ngrp), dim3(AFF_NT), shm_pick_total, s, a, j, t > 0 ? 1 : 0);
                if ((t & 15) == 15 && t + 1 < max_picks) {
                    MRBF_HIP(ctx, hipMemcpyAsync(hstate, a.state, sizeof(hstate), hipMemcpyDeviceToHost, s));
                    MRBF_HIP(ctx, hipStreamSynchronize(s));
                    if (hstate[0]) break;
                }
                continue;
            }
            if (shm > 64 * 1024) {
                MRBF_BLAS(ctx, rocblas_dgemm(ctx->blas, rocblas_operation_none, rocblas_operation_none, d, (int)mc, dz, &one, a.Zs, d, a.C + j, d, &zero, a.U, d));
                hipLaunchKernelGGL(col_norms_kernel, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, s, a.U, d, mc, a.use_inf, a.val);
            } else {
                hipLaunchKernelGGL(affsel_score_kernel, dim3(ngrp), dim3(256), (size_t)AFF_G * dz * sizeof(double), s, a, j);
            }
            hipLaunchKernelGGL(affsel_decide_kernel, dim3(1), dim3(1024), 0, s, a, j);
            hipLaunchKernelGGL(affsel_update_kernel, dim3((unsigned)(dz + (mc + 3) / 4)), dim3(256), 0, s, a, j, j + 1, 1);
            if ((t & 15) == 15 && t + 1 < max_picks) {  // the filter may stop early: look at the "done" word now and then
                MRBF_HIP(ctx, hipMemcpyAsync(hstate, a.state, sizeof(hstate), hipMemcpyDeviceToHost, s));
                MRBF_HIP(ctx, hipStreamSynchronize(s));
                if (hstate[0]) break;
            }
        }
        MRBF_HIP(ctx, hipGetLastError());
        MRBF_HIP(ctx, hipMemcpyAsync(hstate, a.state, sizeof(hstate), hipMemcpyDeviceToHost, s));
        MRBF_HIP(ctx, hipStreamSynchronize(s));
        npick = hstate[1];
        if (a.stamps) {  // phases of the last launch's last workgroup, us since its start
            unsigned long long hs[12];
            MRBF_HIP(ctx, hipMemcpy(hs, a.stamps, sizeof(hs), hipMemcpyDeviceToHost));
            fprintf(stderr, "affsel_pick_kernel phases (us):");
            for (int q2 = 1; q2 < 10; ++q2) fprintf(stderr, " %.2f", (double)(hs[q2] - hs[0]) * 0.01);
            fprintf(stderr, "\n");
        }
        if (npick > 0) {
            std::vector<long long> hp((size_t)npick);
            MRBF_HIP(ctx, hipMemcpy(hp.data(), a.picks, hp.size() * sizeof(long long), hipMemcpyDeviceToHost));
            for (int i = 0; i < npick; ++i) picked_out[i] = (int64_t)hp[i];
        }
    }
    *n_picked = npick;
    if (Z_out) {  // the normalised complement basis of everything chosen so far: Q[:, j0 + npick :] with unit p-norm columns
        const int jf = j0 + npick, dz = d - jf;
        if (dz > 0) {
            hq.resize((size_t)d * dz);
            MRBF_HIP(ctx, hipMemcpy(hq.data(), a.Q + (size_t)jf * d, hq.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int c = 0; c < dz; ++c) {
                double nrm = 0.0;
                for (int r = 0; r < d; ++r) nrm = p_is_inf ? std::max(nrm, std::fabs(hq[(size_t)c * d + r])) : nrm + hq[(size_t)c * d + r] * hq[(size_t)c * d + r];
                if (!p_is_inf) nrm = std::sqrt(nrm);
                for (int r = 0; r < d; ++r) hq[(size_t)c * d + r] /= nrm;
            }
            MRBF_HIP(ctx, hipMemcpy(Z_out, hq.data(), hq.size() * sizeof(double), hipMemcpyDefault));
        }
    }
    return MRBF_OK;
}
