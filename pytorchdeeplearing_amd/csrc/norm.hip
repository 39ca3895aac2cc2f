// GroupNorm(8) + channel dropout + ReLU (+ residual add), forward and backward, channels-last.
// Reference semantics: relu(dropout3d(group_norm(conv(x)))) — networks/VNet3d.py:13-15,34-43,55-59,72-80;
// networks/Unet3d.py:64-86.  Statistics are per (sample, group) over (C/8)*V elements, biased
// variance, eps 1e-5 (torch.nn.GroupNorm).  The conv epilogues deliver per-(n,c) sum / sum-of-squares
// in fp64, so the whole GN forward is: finalize (tiny) + one elementwise pass; the backward is one
// reduction pass + finalize (tiny) + one elementwise pass.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "gn_fold.h"

namespace seg {
namespace {

// GroupNorm statistics are independent per (sample, group): grid = (8 groups, N), 64 threads.  Every thread folds
// the STAT_REP replicas of (channel, replica-slice) pairs of its group; a wave reduction gives the group moments.
// (grid.z = 2: the two branches of the fused input block in one launch)
__global__ __launch_bounds__(64) void gn_finalize_kernel(GnFinArgs a0, GnFinArgs a1) {
    const GnFinArgs a = blockIdx.z ? a1 : a0;
    __shared__ double csum[64][2];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = a.C / GN_GROUPS;                    // 2..32 channels per group
    const int RG = 64 / cpg;                            // replica slices per channel
    const int cl = tid % cpg, rg = tid / cpg, c = g * cpg + cl;
    double s = 0.0, ss = 0.0;
    if (rg < RG) {
        // batches of 8 independent loads: a load -> add loop serialises one L2 round trip per replica (16 for C = 256)
        const int trips = STAT_REP / RG;
        for (int j0 = 0; j0 < trips; j0 += 8) {
            double v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rep = rg + (j0 + u) * RG;
                const double* st = a.stats + (((long long)(rep < STAT_REP ? rep : rg) * a.N + n) * a.C + c) * 2;
                v0[u] = st[0]; v1[u] = st[1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < trips) { s += v0[u]; ss += v1[u]; }
        }
    }
    const double ts = wave_sum_d(s), tss = wave_sum_d(ss);
    const double cnt = (double)cpg * (double)a.V;
    const double mean = ts / cnt;
    double var = tss / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    (void)csum;
    if (tid == 0) {
        a.mean[n * GN_GROUPS + g] = (float)mean;
        a.rstd[n * GN_GROUPS + g] = rstd;
    }
    if (tid < cpg) {
        const float mk = a.mask ? a.mask[(long long)n * a.mask_ld + c] : 1.f;
        const float ga = a.gamma[c], be = a.beta[c];
        a.scale[(long long)n * a.C + c] = mk * ga * rstd;
        a.shift[(long long)n * a.C + c] = mk * (be - ga * (float)mean * rstd);
    }
}

// elementwise; one thread = one 8-channel chunk.  FOLD: the statistics finalize runs as a per-workgroup prologue
// (gn_fold_block) instead of a launch of its own - a kernel boundary costs 4-5 us on this part, the fold 1-2.
// R2 / RES: which optional sources exist, as template flags: the loop body has no branches between its loads, so a thread issues the
// loads of UNR chunks back to back and waits ONCE (with runtime `if (r2)` tests the compiler emitted load -> s_waitcnt vmcnt(0) ->
// load -> ... : up to three serial HBM round trips per 16 bytes, memory-level parallelism left to occupancy alone - and next to the
// weight-gradient stream's register-heavy workgroups a streaming kernel gets few waves per SIMD).
// HC > 0 (C == 16: two neighbouring lanes hold a voxel): the 1^d head on this activation in the same pass.  The even lane starts head_fwd_kernel's fmaf chain
// (bias, channels 0 .. 7) on the values as stored, the odd lane continues it (channels 8 .. 15) and writes the voxel's logits / probabilities: same operations
// in the same order as the separate launch.
template <class T, bool FOLD, bool R2, bool RES, int HC = 0>
__global__ __launch_bounds__(256) void gn_act_kernel(ActArgs a) {
    __shared__ double part[FOLD ? 256 : 1][2];
    __shared__ float coef_s[FOLD ? 4 : 1][256];
    if (HC > 0 && a.zero_ptr && blockIdx.x == 0 && blockIdx.y == 0)
        for (long long i = threadIdx.x; i < a.zero_n; i += 256) a.zero_ptr[i] = 0.0;
    // grid.y = sample; chunk index inside the sample in 32 bits, C/8 a power of two: no 64-bit division per element
    const int CPR = a.C / 8, n = blockIdx.y;
    const int per_n = (int)(a.V * CPR);
    const long long base = (long long)n * per_n;
    const T* r1 = (const T*)a.r1;
    const T* r2 = (const T*)a.r2;
    const T* res = (const T*)a.res;
    T* out = (T*)a.out;
    // the grid stride is a multiple of C/8: a thread keeps its channel chunk, the coefficients live in registers
    const int c0 = ((blockIdx.x * 256 + threadIdx.x) & (CPR - 1)) * 8;
    vec<float, 8> sc, sh, sc2, sh2;
    if (FOLD) {
        gn_fold_block(a.fin1, n, blockIdx.x == 0, part, coef_s[0], coef_s[1]);
        if (R2) gn_fold_block(a.fin2, n, blockIdx.x == 0, part, coef_s[2], coef_s[3]);
        sc = *(const vec<float, 8>*)&coef_s[0][c0];
        sh = *(const vec<float, 8>*)&coef_s[1][c0];
        if (R2) { sc2 = *(const vec<float, 8>*)&coef_s[2][c0]; sh2 = *(const vec<float, 8>*)&coef_s[3][c0]; }
    } else {
        sc = *(const vec<float, 8>*)(a.scale1 + (long long)n * a.C + c0);
        sh = *(const vec<float, 8>*)(a.shift1 + (long long)n * a.C + c0);
        if (R2) {
            sc2 = *(const vec<float, 8>*)(a.scale2 + (long long)n * a.C + c0);
            sh2 = *(const vec<float, 8>*)(a.shift2 + (long long)n * a.C + c0);
        }
    }
    constexpr int UNR = (R2 || RES) ? 2 : 4;
    const int stride = gridDim.x * 256;
    float hw[HC > 0 ? HC : 1][8], hb[HC > 0 ? HC : 1];
    if (HC > 0) {
#pragma unroll
        for (int c = 0; c < HC; ++c) {
            hb[c] = a.head_b ? a.head_b[c] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) hw[c][j] = a.head_w[c * 16 + c0 + j];
        }
    }
    for (int ii = blockIdx.x * 256 + threadIdx.x; ii < per_n; ii += stride * UNR) {
        long long iu[UNR];
        bool ok[UNR];
        vec<T, 8> x[UNR], x2[UNR], rr[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {                 // out-of-range slots re-read slot 0 (no branch around a load) and store nothing
            const int t = ii + u * stride;
            ok[u] = t < per_n;
            iu[u] = base + (ok[u] ? t : ii);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            x[u] = load8(r1 + iu[u] * 8);
            if (R2) x2[u] = load8(r2 + iu[u] * 8);
            if (RES) rr[u] = load8(res + iu[u] * 8);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = fmaxf(fmaf(sc[j], to_f(x[u][j]), sh[j]), 0.f);
            if (R2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] += fmaxf(fmaf(sc2[j], to_f(x2[u][j]), sh2[j]), 0.f);
            }
            if (RES) {
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] += to_f(rr[u][j]);
            }
            vec<T, 8> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = from_f<T>(y[j]);
            if (ok[u]) store8(out + iu[u] * 8, o);
            if constexpr (HC > 0) {
                // (per_n is even and the trip stride a multiple of 256: both lanes of a voxel take the same trips)
                const bool hi = threadIdx.x & 1;
                float z[HC > 0 ? HC : 1];
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                    float s = hb[c];
#pragma unroll
                    for (int j = 0; j < 8; ++j) s = fmaf(to_f(o[j]), hw[c][j], s);          // even lane: bias + channels 0 .. 7
                    const float lo = __shfl_xor(s, 1);                                    // odd lane: the even lane's partial chain
                    float t = lo;
#pragma unroll
                    for (int j = 0; j < 8; ++j) t = fmaf(to_f(o[j]), hw[c][j], t);          // ... continued with channels 8 .. 15
                    z[c] = t;
                }
                if (hi && ok[u]) {
                    const long long v = (long long)((ok[u] ? (int)(iu[u] - base) : 0) >> 1);
                    if (HC == 1) {
                        a.logits[(long long)n * a.V + v] = z[0];
                        a.probs[(long long)n * a.V + v] = 1.f / (1.f + expf(-z[0]));
                    } else {
                        float mx = z[0];
#pragma unroll
                        for (int c = 1; c < HC; ++c) mx = fmaxf(mx, z[c]);
                        float e[HC > 0 ? HC : 1], se = 0.f;
#pragma unroll
                        for (int c = 0; c < HC; ++c) { e[c] = expf(z[c] - mx); se += e[c]; }
                        const float inv = 1.f / se;
#pragma unroll
                        for (int c = 0; c < HC; ++c) {
                            const long long o2 = ((long long)n * HC + c) * a.V + v;
                            a.logits[o2] = z[c];
                            a.probs[o2] = e[c] * inv;
                        }
                    }
                }
            }
        }
    }
}

// i = 8-channel chunk index ((n*V + v)*C/8 + c0/8); (n, v, c0) only matter for the virtual head source
template <class T>
__device__ __forceinline__ void load_dy_sum(const GnBwdArgs& a, long long i, float* g, int n = 0, long long v = 0, int c0 = 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
    if (a.vdl) {
        for (int k = 0; k < a.vK; ++k) {
            const float dl = a.vdl[((long long)n * a.vK + k) * a.V + v];
            const vec<float, 8> w = *(const vec<float, 8>*)(a.vw + (long long)k * a.C + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = fmaf(dl, w[j], g[j]);
        }
    }
    if (a.ndy > 0) {
        const vec<T, 8> d0 = load8((const T*)a.dy[0] + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += to_f(d0[j]);
    }
    if (a.ndy > 1) {
        const vec<T, 8> d1 = load8((const T*)a.dy[1] + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += to_f(d1[j]);
    }
    if (a.ndy > 2) {
        const vec<T, 8> d2 = load8((const T*)a.dy[2] + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] += to_f(d2[j]);
    }
}

// NDY = 1..3: exactly that many stored gradient sources and no virtual one - issue() is branch-free, so callers can put the loads of
// several chunks in flight before the first use; NDY = 0: anything else (the unit below the head), through load_dy_sum.
// NDY = 4 / 5: the virtual head source of a one-class head (vK == 1: dy[v][c] += dl[n][v] * w[c]) plus 0 / 1 stored sources - the two
// 96^3 units under the head, the largest GroupNorm-backward launches of the step; one scalar load per chunk, the head weights of the
// thread's channels are loop-invariant (vw8, loaded by the caller)
template <class T, int NDY> struct DySrc {
    static constexpr bool VH = NDY >= 4;
    static constexpr int NS = VH ? NDY - 4 : NDY;
    vec<T, 8> d[NS > 0 ? NS : 1];
    float g0[8];
    float dl;
    __device__ __forceinline__ void issue(const GnBwdArgs& a, long long i, int n, long long v, int c0) {
        if (NDY == 0) { load_dy_sum<T>(a, i, g0, n, v, c0); return; }
        if (VH) dl = a.vdl[(long long)n * a.V + v];
#pragma unroll
        for (int k = 0; k < NS; ++k) d[k] = load8((const T*)a.dy[k] + i * 8);
    }
    __device__ __forceinline__ void sum(float* g, const vec<float, 8>& vw8) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = NDY == 0 ? g0[j] : (VH ? fmaf(dl, vw8[j], 0.f) : 0.f);
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += to_f(d[k][j]);          // same order as load_dy_sum: ((head + d0) + d1) + d2
    }
};
__host__ inline int dy_variant(const GnBwdArgs& a) {
    if (a.vdl) return (a.vK == 1 && a.ndy >= 0 && a.ndy <= 1) ? 4 + a.ndy : 0;
    return (a.ndy >= 1 && a.ndy <= 3) ? a.ndy : 0;
}

// pass 1.  grid = (slabs, N); a block reduces `rows_per_block` voxels of one sample over all channels.
// thread = (chunk column cc, row group g); LDS tree over row groups; fp64 atomics per (n,c).
template <class T, bool DUAL, int NDY>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnBwdArgs a, int GNB_ROWS) {
    __shared__ float red[256 * 16];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int CPR = a.C / 8;                  // 2..32 (power of two)
    const int G = 256 / CPR;
    const int cc = tid % CPR, g = tid / CPR;
    const long long v0 = (long long)blockIdx.x * GNB_ROWS;
    const long long v1 = (v0 + GNB_ROWS < a.V) ? v0 + GNB_ROWS : a.V;
    const T* r = (const T*)a.r;
    const T* r2 = (const T*)a.r2;
    const vec<float, 8> sc = *(const vec<float, 8>*)(a.scale + (long long)n * a.C + cc * 8);
    const vec<float, 8> sh = *(const vec<float, 8>*)(a.shift + (long long)n * a.C + cc * 8);
    vec<float, 8> sc2 = sc, sh2 = sh;
    if (DUAL) {
        sc2 = *(const vec<float, 8>*)(a.scale2 + (long long)n * a.C + cc * 8);
        sh2 = *(const vec<float, 8>*)(a.shift2 + (long long)n * a.C + cc * 8);
    }
    float q1[8], q2[8], p1[8], p2[8];          // p*: second branch
#pragma unroll
    for (int j = 0; j < 8; ++j) { q1[j] = 0.f; q2[j] = 0.f; p1[j] = 0.f; p2[j] = 0.f; }
    // RW rows in flight per thread, every load issued before the first use (pure streaming: 2-4 x 16 B loads per row)
    constexpr int RW = (NDY == 0) ? 1 : ((DUAL || DySrc<T, NDY>::NS > 1) ? 2 : 4);
    vec<float, 8> vw8;
#pragma unroll
    for (int j = 0; j < 8; ++j) vw8[j] = 0.f;
    if (DySrc<T, NDY>::VH) vw8 = *(const vec<float, 8>*)(a.vw + cc * 8);
    long long v = v0 + g;
    for (; v < v1; v += (long long)RW * G) {
        long long iu[RW];
        bool ok[RW];
        DySrc<T, NDY> src[RW];
        vec<T, 8> x[RW], z[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) {                  // rows past the slab re-read row 0 and contribute nothing
            const long long vv = v + (long long)u * G;
            ok[u] = vv < v1;
            iu[u] = ((long long)n * a.V + (ok[u] ? vv : v)) * CPR + cc;
        }
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            src[u].issue(a, iu[u], n, ok[u] ? v + (long long)u * G : v, cc * 8);
            x[u] = load8(r + iu[u] * 8);
            if (DUAL) z[u] = load8(r2 + iu[u] * 8);
        }
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            float dy[8];
            src[u].sum(dy, vw8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = to_f(x[u][j]);
                const float d = (ok[u] && fmaf(sc[j], xv, sh[j]) > 0.f) ? dy[j] : 0.f;
                q1[j] += d;
                q2[j] = fmaf(d, xv, q2[j]);
                if (DUAL) {
                    const float zv = to_f(z[u][j]);
                    const float e = (ok[u] && fmaf(sc2[j], zv, sh2[j]) > 0.f) ? dy[j] : 0.f;
                    p1[j] += e;
                    p2[j] = fmaf(e, zv, p2[j]);
                }
            }
        }
    }
#pragma unroll                  // (unrolled: `br ? p1 : q1` under a run-time branch index sent all four accumulator arrays to scratch, 144 B per lane)
    for (int br = 0; br < (DUAL ? 2 : 1); ++br) {
        if (br) __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = br ? p1[j] : q1[j]; red[tid * 16 + 8 + j] = br ? p2[j] : q2[j]; }
        __syncthreads();
        // column (cc, j, which) summed over the G row groups by one thread each: CPR*16 <= 512 columns
        double* Q = br ? a.Q2 : a.Q;
        for (int col = tid; col < CPR * 16; col += 256) {
            const int ccx = col / 16, jj = col % 16;
            double s = 0.0;
            for (int k = 0; k < G; ++k) s += red[(k * CPR + ccx) * 16 + jj];
            const int c = ccx * 8 + (jj & 7), which = jj >> 3;
            atomicAdd(Q + (((long long)(blockIdx.x % (a.rep_q > 0 ? a.rep_q : STAT_REP)) * a.N + n) * a.C + c) * 2 + which, s);
        }
    }
}

// backward finalize: grid = (8 groups, N), 64 threads — same decomposition as the forward finalize
__global__ __launch_bounds__(64) void gn_bwd_finalize_kernel(GnBwdFinArgs a0, GnBwdFinArgs a1) {
    const GnBwdFinArgs a = blockIdx.z ? a1 : a0;
    __shared__ double sh[64][3];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = a.C / GN_GROUPS, RG = 64 / cpg;
    const int cl = tid % cpg, rg = tid / cpg, c = g * cpg + cl;
    double f1 = 0.0, f2 = 0.0, f3 = 0.0;
    if (rg < RG) {
        const int trips = STAT_REP / RG;
        for (int j0 = 0; j0 < trips; j0 += 8) {
            double v1[8], v2[8], v3[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int rep = rg + (j0 + u) * RG;
                const long long o = (((long long)(rep < STAT_REP ? rep : rg) * a.N + n) * a.C + c) * 2;
                v1[u] = a.Q[o]; v2[u] = a.Q[o + 1]; v3[u] = a.stats[o];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (j0 + u < trips) { f1 += v1[u]; f2 += v2[u]; f3 += v3[u]; }
        }
    }
    sh[tid][0] = f1; sh[tid][1] = f2; sh[tid][2] = f3;
    __syncthreads();
    double Q1 = 0.0, Q2 = 0.0, R1 = 0.0;                 // per-channel totals (valid for tid < cpg)
    if (tid < cpg)
        for (int k = 0; k < RG; ++k) { Q1 += sh[tid + k * cpg][0]; Q2 += sh[tid + k * cpg][1]; R1 += sh[tid + k * cpg][2]; }
    const double mu = a.mean[n * GN_GROUPS + g], rs = a.rstd[n * GN_GROUPS + g];
    double mk = 1.0, ga = 0.0, q1 = 0.0, qx = 0.0;
    if (tid < cpg) {
        mk = a.mask ? (double)a.mask[(long long)n * a.mask_ld + c] : 1.0;
        ga = a.gamma[c];
        q1 = mk * Q1;                                    // sum dz
        qx = (mk * Q2 - mu * q1) * rs;                   // sum dz * xhat
        atomicAdd(&a.dbeta[c], (float)q1);
        atomicAdd(&a.dgamma[c], (float)qx);
    }
    const double S1 = wave_sum_d(tid < cpg ? ga * q1 : 0.0);
    const double S2 = wave_sum_d(tid < cpg ? ga * qx : 0.0);
    if (tid < cpg) {
        const double Mg = (double)cpg * (double)a.V;
        const double A = rs * ga * mk;
        const double B = -rs * rs * S2 / Mg;
        const double Cc = -rs * S1 / Mg + rs * rs * S2 * mu / Mg;
        float* co = a.coef + ((long long)n * a.C + c) * 3;
        co[0] = (float)A; co[1] = (float)B; co[2] = (float)Cc;
        // sum_v dr = A*sum(dzr) + B*sum(r) + Cc*V   (sum(r) from the forward statistics)
        if (a.dbias) atomicAdd(&a.dbias[c], (float)(A * Q1 + B * R1 + Cc * (double)a.V));
    }
}

// The backward finalize of one sample inside a consumer workgroup (same decomposition as gn_fold_block): folds the replica
// partials of Q (sum dz, sum dz*r) and of the forward channel sums, and leaves the A / B / Cc coefficients of every channel in
// LDS (same formulas as gn_bwd_finalize_kernel).  `publish`: this workgroup adds the gamma / beta / conv-bias gradients of its
// sample.  Ends with a barrier.
__device__ __forceinline__ void gn_bwd_fold_block(const GnBwdFinArgs& f, int n, bool publish, double (*part)[3], float* A_s, float* B_s,
                                                  float* C_s) {
    const int tid = threadIdx.x, C = f.C, cpg = C / GN_GROUPS;
    const int S = 256 / C;
    const int c = tid % C, sl = tid / C;
    const int nq = f.rep_q > 0 ? f.rep_q : STAT_REP, ns = f.rep_s > 0 ? f.rep_s : STAT_REP;
    const int nrep = nq > ns ? nq : ns;
    // loads that do not depend on the folded sums are issued up front (one L2 round trip for the prologue, not two)
    double mu = 0.0, rs = 0.0, mk = 1.0, ga = 0.0;
    if (tid < C) {
        mu = f.mean[n * GN_GROUPS + tid / cpg]; rs = f.rstd[n * GN_GROUPS + tid / cpg];
        mk = f.mask ? (double)f.mask[(long long)n * f.mask_ld + tid] : 1.0;
        ga = f.gamma[tid];
    }
    double f1 = 0.0, f2 = 0.0, f3 = 0.0;
    for (int r0 = sl; r0 < nrep; r0 += 4 * S) {
        double v1[4], v2[4], v3[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rep = r0 + u * S;
            const long long o = (((long long)(rep < nrep ? rep : sl) * f.N + n) * C + c) * 2;
            v1[u] = f.Q[o]; v2[u] = f.Q[o + 1]; v3[u] = f.stats[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u * S < nrep) { f1 += v1[u]; f2 += v2[u]; f3 += v3[u]; }
    }
    part[tid][0] = f1; part[tid][1] = f2; part[tid][2] = f3;
    __syncthreads();
    double Q1 = 0.0, Q2 = 0.0, R1 = 0.0, q1 = 0.0, qx = 0.0;
    if (tid < C) {
        for (int k = 0; k < S; ++k) { Q1 += part[tid + k * C][0]; Q2 += part[tid + k * C][1]; R1 += part[tid + k * C][2]; }
        q1 = mk * Q1;                                    // sum dz
        qx = (mk * Q2 - mu * q1) * rs;                   // sum dz * xhat
        if (publish) {
            atomicAdd(&f.dbeta[c], (float)q1);
            atomicAdd(&f.dgamma[c], (float)qx);
        }
    }
    double S1 = ga * q1, S2 = ga * qx;
    for (int o = cpg >> 1; o >= 1; o >>= 1) { S1 += __shfl_xor(S1, o); S2 += __shfl_xor(S2, o); }
    if (tid < C) {
        const double Mg = (double)cpg * (double)f.V;
        const double A = rs * ga * mk;
        const double B = -rs * rs * S2 / Mg;
        const double Cc = -rs * S1 / Mg + rs * rs * S2 * mu / Mg;
        A_s[c] = (float)A; B_s[c] = (float)B; C_s[c] = (float)Cc;
        if (publish) {
            if (f.coef) { float* co = f.coef + ((long long)n * C + c) * 3; co[0] = (float)A; co[1] = (float)B; co[2] = (float)Cc; }
            // sum_v dr = A*sum(dzr) + B*sum(r) + Cc*V   (sum(r) from the forward statistics)
            if (f.dbias) atomicAdd(&f.dbias[c], (float)(A * Q1 + B * R1 + Cc * (double)f.V));
        }
    }
    __syncthreads();
}

// FOLD: the backward finalize runs as a per-workgroup prologue (gn_bwd_fold_block); fa / fb are the finalize arguments of the branches
template <class T, bool DUAL, bool FOLD, int NDY>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdArgs a, GnBwdFinArgs fa, GnBwdFinArgs fb) {
    __shared__ double part[FOLD ? 256 : 1][3];
    __shared__ float coef_s[FOLD ? (DUAL ? 6 : 3) : 1][256];
    const int CPR = a.C / 8, n = blockIdx.y;       // grid.y = sample, 32-bit chunk index inside it (C/8 is a power of two)
    const int per_n = (int)(a.V * CPR), lc = 31 - __builtin_clz(CPR);
    const long long base = (long long)n * per_n;
    const T* r = (const T*)a.r;
    T* dr = (T*)a.dr;
    // the grid stride is a multiple of C/8: a thread keeps its channel chunk, all coefficients live in registers
    const int c0 = ((blockIdx.x * 256 + threadIdx.x) & (CPR - 1)) * 8;
    float co[24], co2[24];
    if (FOLD) {
        gn_bwd_fold_block(fa, n, blockIdx.x == 0, part, coef_s[0], coef_s[1], coef_s[2]);
        if (DUAL) gn_bwd_fold_block(fb, n, blockIdx.x == 0, part, coef_s[DUAL ? 3 : 0], coef_s[DUAL ? 4 : 0], coef_s[DUAL ? 5 : 0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            co[j * 3] = coef_s[0][c0 + j]; co[j * 3 + 1] = coef_s[1][c0 + j]; co[j * 3 + 2] = coef_s[2][c0 + j];
            if (DUAL) { co2[j * 3] = coef_s[DUAL ? 3 : 0][c0 + j]; co2[j * 3 + 1] = coef_s[DUAL ? 4 : 0][c0 + j]; co2[j * 3 + 2] = coef_s[DUAL ? 5 : 0][c0 + j]; }
        }
    } else {
        // per-(n, c) coefficients as wide loads (the small levels are latency-bound)
        const vec<float, 8>* cop = (const vec<float, 8>*)(a.coef + ((long long)n * a.C + c0) * 3);
        const vec<float, 8> ca = cop[0], cb = cop[1], cc = cop[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) { co[j] = ca[j]; co[8 + j] = cb[j]; co[16 + j] = cc[j]; }
        if (DUAL) {
            const vec<float, 8>* cop2 = (const vec<float, 8>*)(a.coef2 + ((long long)n * a.C + c0) * 3);
            const vec<float, 8> da = cop2[0], db = cop2[1], dc = cop2[2];
#pragma unroll
            for (int j = 0; j < 8; ++j) { co2[j] = da[j]; co2[8 + j] = db[j]; co2[16 + j] = dc[j]; }
        }
    }
    const vec<float, 8> sc = *(const vec<float, 8>*)(a.scale + (long long)n * a.C + c0);
    const vec<float, 8> sh = *(const vec<float, 8>*)(a.shift + (long long)n * a.C + c0);
    vec<float, 8> sc2 = sc, sh2 = sh;
    if (DUAL) {
        sc2 = *(const vec<float, 8>*)(a.scale2 + (long long)n * a.C + c0);
        sh2 = *(const vec<float, 8>*)(a.shift2 + (long long)n * a.C + c0);
    }
    // UNR chunks per trip, all loads in front (see gn_act_kernel)
    constexpr int UNR = (NDY == 0) ? 1 : ((DUAL || DySrc<T, NDY>::NS > 1) ? 2 : 4);
    vec<float, 8> vw8;
#pragma unroll
    for (int j = 0; j < 8; ++j) vw8[j] = 0.f;
    if (DySrc<T, NDY>::VH) vw8 = *(const vec<float, 8>*)(a.vw + c0);
    const int stride = gridDim.x * 256;
    for (int ii = blockIdx.x * 256 + threadIdx.x; ii < per_n; ii += stride * UNR) {
        long long iu[UNR];
        int tu[UNR];
        bool ok[UNR];
        DySrc<T, NDY> src[UNR];
        vec<T, 8> x[UNR], z[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int t = ii + u * stride;
            ok[u] = t < per_n;
            tu[u] = ok[u] ? t : ii;
            iu[u] = base + tu[u];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            src[u].issue(a, iu[u], n, tu[u] >> lc, c0);
            x[u] = load8(r + iu[u] * 8);
            if (DUAL) z[u] = load8((const T*)a.r2 + iu[u] * 8);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float dy[8];
            src[u].sum(dy, vw8);
            vec<T, 8> o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = to_f(x[u][j]);
                const float d = (fmaf(sc[j], xv, sh[j]) > 0.f) ? dy[j] : 0.f;
                o[j] = from_f<T>(fmaf(co[j * 3], d, fmaf(co[j * 3 + 1], xv, co[j * 3 + 2])));
            }
            if (ok[u]) store8(dr + iu[u] * 8, o);
            if (DUAL) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float zv = to_f(z[u][j]);
                    const float d = (fmaf(sc2[j], zv, sh2[j]) > 0.f) ? dy[j] : 0.f;
                    o[j] = from_f<T>(fmaf(co2[j * 3], d, fmaf(co2[j * 3 + 1], zv, co2[j * 3 + 2])));
                }
                if (ok[u]) store8((T*)a.dr2 + iu[u] * 8, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Small-tensor GroupNorm backward (C >= 64, i.e. the 24^3 / 12^3 / 6^3 levels whose tensors sit in L2):
// GroupNorm is independent per (sample, group), so ONE 1024-thread workgroup per (n, g) does the
// reduction pass, the coefficient math (incl. gamma/beta/bias gradients) and the elementwise pass
// back to back — one launch instead of reduce + finalize + apply (3 x ~8 us of launch/latency each).
// ------------------------------------------------------------------------------------------------
struct GnBwdGroupArgs {
    GnBwdArgs e;          // dy sources, r, forward scale/shift, dr
    GnBwdFinArgs f;       // stats, gamma, mask, mean/rstd, dgamma/dbeta/dbias
};

template <class T>
__global__ __launch_bounds__(1024) void gn_bwd_group_kernel(GnBwdGroupArgs a) {
    __shared__ float wsum[16][4][16];        // [wave][chunk-in-group][q1 0..7 | q2 0..7]
    __shared__ double chan[32][3];           // per channel of the group: Q1, Q2, R1
    __shared__ float coef[32][3];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = a.e.C, cpg = C / GN_GROUPS, CG = cpg / 8, CPR = C / 8;     // CG = 16-B chunks per voxel in this group: 1, 2 or 4
    const long long V = a.e.V;
    const int cg = tid % CG, c0 = g * cpg + cg * 8;
    const T* r = (const T*)a.e.r;
    T* dr = (T*)a.e.dr;
    const vec<float, 8> sc = *(const vec<float, 8>*)(a.e.scale + (long long)n * C + c0);
    const vec<float, 8> sh = *(const vec<float, 8>*)(a.e.shift + (long long)n * C + c0);
    float q1[8], q2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { q1[j] = 0.f; q2[j] = 0.f; }
    const long long items = V * CG;
    for (long long it = tid; it < items; it += 1024) {
        const long long i = ((long long)n * V + it / CG) * CPR + g * CG + cg;
        float dy[8];
        load_dy_sum<T>(a.e, i, dy);
        const vec<T, 8> x = load8(r + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = to_f(x[j]);
            const float d = (fmaf(sc[j], xv, sh[j]) > 0.f) ? dy[j] : 0.f;
            q1[j] += d;
            q2[j] = fmaf(d, xv, q2[j]);
        }
    }
    // lanes with equal (lane % CG) hold the same channels: butterfly over the remaining lane bits
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        for (int m = 32; m >= CG; m >>= 1) { q1[j] += __shfl_xor(q1[j], m); q2[j] += __shfl_xor(q2[j], m); }
    }
    if (lane < CG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { wsum[wv][lane][j] = q1[j]; wsum[wv][lane][8 + j] = q2[j]; }
    }
    // forward sum(r) of the group's channels (bias gradient): fold the statistic replicas
    if (tid < cpg) {
        const int nrep = a.f.rep_s > 0 ? a.f.rep_s : STAT_REP;       // the replicas the forward producer wrote
        double r1 = 0.0;
        for (int rep = 0; rep < nrep; rep += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = a.f.stats[(((long long)(rep + u < nrep ? rep + u : 0) * a.f.N + n) * C + g * cpg + tid) * 2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (rep + u < nrep) r1 += v[u];
        }
        chan[tid][2] = r1;
    }
    __syncthreads();
    if (tid < cpg) {
        double s1 = 0.0, s2 = 0.0;
        for (int w = 0; w < 16; ++w) { s1 += wsum[w][tid >> 3][tid & 7]; s2 += wsum[w][tid >> 3][8 + (tid & 7)]; }
        chan[tid][0] = s1; chan[tid][1] = s2;
    }
    __syncthreads();
    if (wv == 0) {                                       // one wave finishes the group: cpg <= 32 channels
        const int c = g * cpg + (lane < cpg ? lane : 0);
        const bool act = lane < cpg;
        const double mu = a.f.mean[n * GN_GROUPS + g], rs = a.f.rstd[n * GN_GROUPS + g];
        const double mk = act ? (a.f.mask ? (double)a.f.mask[(long long)n * a.f.mask_ld + c] : 1.0) : 0.0;
        const double ga = act ? (double)a.f.gamma[c] : 0.0;
        const double Q1 = act ? chan[lane][0] : 0.0, Q2 = act ? chan[lane][1] : 0.0, R1 = act ? chan[lane][2] : 0.0;
        const double q1d = mk * Q1, qx = (mk * Q2 - mu * q1d) * rs;
        const double S1 = wave_sum_d(ga * q1d), S2 = wave_sum_d(ga * qx);
        if (act) {
            atomicAdd(&a.f.dbeta[c], (float)q1d);
            atomicAdd(&a.f.dgamma[c], (float)qx);
            const double Mg = (double)cpg * (double)V;
            const double A = rs * ga * mk, B = -rs * rs * S2 / Mg, Cc = -rs * S1 / Mg + rs * rs * S2 * mu / Mg;
            coef[lane][0] = (float)A; coef[lane][1] = (float)B; coef[lane][2] = (float)Cc;
            if (a.f.dbias) atomicAdd(&a.f.dbias[c], (float)(A * Q1 + B * R1 + Cc * (double)V));
        }
    }
    __syncthreads();
    float cA[8], cB[8], cC[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cA[j] = coef[cg * 8 + j][0]; cB[j] = coef[cg * 8 + j][1]; cC[j] = coef[cg * 8 + j][2]; }
    for (long long it = tid; it < items; it += 1024) {
        const long long i = ((long long)n * V + it / CG) * CPR + g * CG + cg;
        float dy[8];
        load_dy_sum<T>(a.e, i, dy);
        const vec<T, 8> x = load8(r + i * 8);
        vec<T, 8> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = to_f(x[j]);
            const float d = (fmaf(sc[j], xv, sh[j]) > 0.f) ? dy[j] : 0.f;
            o[j] = from_f<T>(fmaf(cA[j], d, fmaf(cB[j], xv, cC[j])));
        }
        store8(dr + i * 8, o);
    }
}

// forward twin: statistics fold -> mean/rstd/scale/shift (stored for the backward) -> y = relu(scale*r+shift) [+ residual]
struct GnFwdGroupArgs {
    GnFinArgs f;
    const void* r; const void* res; void* out;
};

template <class T>
__global__ __launch_bounds__(1024) void gn_fwd_group_kernel(GnFwdGroupArgs a) {
    __shared__ float ssc[32], ssh[32];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = a.f.C, cpg = C / GN_GROUPS, CG = cpg / 8, CPR = C / 8;
    const long long V = a.f.V;
    if (wv == 0) {
        const int RG = 64 / cpg, cl = lane % cpg, rg = lane / cpg, c = g * cpg + cl;
        // only the replicas the producer wrote (f.rep; 4 at this level, not all 32), four independent 16-B loads per trip: the first version walked
        // all 32 replicas with one dependent L2 round trip per trip - 16 of them, ~10 of this kernel's 11 us (round-4 single-queue trace)
        const int nrep = a.f.rep > 0 ? a.f.rep : STAT_REP;
        double s = 0.0, ss = 0.0;
        if (rg < RG)
            for (int rep = rg; rep < nrep; rep += 4 * RG) {
                double v0[4], v1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = rep + u * RG;
                    const double* st = a.f.stats + (((long long)(rr < nrep ? rr : rg) * a.f.N + n) * C + c) * 2;
                    v0[u] = st[0]; v1[u] = st[1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (rep + u * RG < nrep) { s += v0[u]; ss += v1[u]; }
            }
        const double ts = wave_sum_d(s), tss = wave_sum_d(ss);
        const double cnt = (double)cpg * (double)V;
        const double mean = ts / cnt;
        double var = tss / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)a.f.eps));
        if (lane == 0) { a.f.mean[n * GN_GROUPS + g] = (float)mean; a.f.rstd[n * GN_GROUPS + g] = rstd; }
        if (lane < cpg) {
            const float mk = a.f.mask ? a.f.mask[(long long)n * a.f.mask_ld + c] : 1.f;
            const float ga = a.f.gamma[c], be = a.f.beta[c];
            const float sc = mk * ga * rstd, sh = mk * (be - ga * (float)mean * rstd);
            ssc[lane] = sc; ssh[lane] = sh;
            a.f.scale[(long long)n * C + c] = sc;
            a.f.shift[(long long)n * C + c] = sh;
        }
    }
    __syncthreads();
    const int cg = tid % CG;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = ssc[cg * 8 + j]; sh[j] = ssh[cg * 8 + j]; }
    const T* r = (const T*)a.r;
    const T* res = (const T*)a.res;
    T* out = (T*)a.out;
    const long long items = V * CG;
    for (long long it = tid; it < items; it += 1024) {
        const long long i = ((long long)n * V + it / CG) * CPR + g * CG + cg;
        const vec<T, 8> x = load8(r + i * 8);
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = fmaxf(fmaf(sc[j], to_f(x[j]), sh[j]), 0.f);
        if (res) {
            const vec<T, 8> rr = load8(res + i * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] += to_f(rr[j]);
        }
        vec<T, 8> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = from_f<T>(y[j]);
        store8(out + i * 8, o);
    }
}

// ------------------------------------------------------------------------------------------------
// Deep-level GroupNorm backward in ONE launch on MANY workgroups (C >= 64: the 24^3 / 12^3 / 6^3 levels).  The two-pass form costs two latency-bound
// launches per unit there (5 - 10 us each for a tensor of <= 7 MB), the one-workgroup-per-group form above pulls a group through 1024 threads (32
// workgroups: 11 - 14 us at 6^3, 4x slower than two passes at 24^3).  Here the S workgroups of a (sample, group) each load a slice ONCE and keep it in
// registers (<= K chunks of 8 channels per thread), hand their partial sums (sum dz, sum dz*r per channel: two floats in one 8-byte word) to one another
// through memory (xwg_store / xwg_load: no fence, no atomics on data), every workgroup folds the S partials in slot order (deterministic) into the
// group's coefficients and applies them to the registers it kept.  A word is "there" when it is non-zero (+0.0 is stored as -0.0); the slot area is the
// unit's Q region, which every backward pass starts with zeros.  Workgroups of a group have consecutive ids and the dispatcher hands out ids in order, so
// the earliest unfinished group is always resident as a whole: the wait cannot starve as long as S workgroups fit on the device at once (S <= 32).
// The poll gives up after ~0.3 s and poisons the output with NaN (the gradient check of the step then reports it) instead of hanging the queue.
// PH: 0 = the kernel; 1 / 2 = its two halves as separate launches (host checker: workgroups run one after another there).
// ------------------------------------------------------------------------------------------------
struct GnBwdCoopArgs {
    GnBwdArgs e;
    GnBwdFinArgs f;
    unsigned long long* slots;      // [N][8][S][cpg] words, zero at launch
    int S, ku;                      // workgroups per (sample, group); chunks per thread (<= K)
};

template <class T, int NDY, int K, int PH>
__global__ __launch_bounds__(256, 2) void gn_bwd_coop_kernel(GnBwdCoopArgs a) {      // (two waves per SIMD: left alone hipcc takes 258 registers for <f16, 1, 8>)
    __shared__ float wsum[16][4][16];        // [DPP row of the workgroup][chunk-in-group][q1 0..7 | q2 0..7]
    __shared__ float part[256][2];           // [slot * cpg + channel]: the partial sums of every workgroup of the group
    __shared__ double chan[32][3];           // per channel of the group: Q1, Q2, R1
    __shared__ float coef[32][3];
    const int S = a.S;
    const int g = blockIdx.x / S, sl = blockIdx.x - g * S, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = a.e.C, cpg = C / GN_GROUPS, CG = cpg / 8, CPR = C / 8;     // CG = 16-B chunks per voxel in this group: 1, 2 or 4
    const int V = (int)a.e.V;
    const int cg = tid % CG, c0 = g * cpg + cg * 8;
    const int items = V * CG, it0 = sl * a.ku * 256;       // (a slice starts at a multiple of 256: a thread keeps its chunk column cg)
    const T* r = (const T*)a.e.r;
    T* dr = (T*)a.e.dr;
    unsigned long long* slot = a.slots + (long long)((n * GN_GROUPS + g) * S) * cpg;
    const vec<float, 8> sc = *(const vec<float, 8>*)(a.e.scale + (long long)n * C + c0);
    const vec<float, 8> sh = *(const vec<float, 8>*)(a.e.shift + (long long)n * C + c0);
    // the slice: every load in front of the first use (slots past the slice re-read a valid chunk and contribute / store nothing)
    unsigned okm = 0u;
    DySrc<T, NDY> src[K];
    vec<T, 8> x[K];
    vec<float, 8> vw8;
#pragma unroll
    for (int j = 0; j < 8; ++j) vw8[j] = 0.f;
    // chunk index of slot k = i0 + k * dI (consecutive slots lie 256 items = 256 / CG voxels apart): two registers instead of K across the exchange
    const int ifall = n * V * CPR + g * CG + cg;                       // (voxel 0: what a slot past the slice reads)
    const int i0 = ifall + ((it0 + tid) / CG) * CPR, dI = (256 / CG) * CPR;
    auto chunk_of = [&](int k) { return ((okm >> k) & 1u) ? i0 + k * dI : ifall; };
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int it = it0 + tid + k * 256;
        const bool ok = k < a.ku && it < items;
        okm |= ok ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int ic = chunk_of(k);
        src[k].issue(a.e, ic, n, 0, c0);
        x[k] = load8(r + (long long)ic * 8);
    }
    // what the coefficient math needs besides the sums travels with the slice (not as a round trip of its own behind the exchange)
    double f_mu = 0.0, f_rs = 0.0, f_mk = 0.0, f_ga = 0.0;
    if (wv == 0 && PH != 1) {
        const int c = g * cpg + (lane < cpg ? lane : 0);
        f_mu = a.f.mean[n * GN_GROUPS + g]; f_rs = a.f.rstd[n * GN_GROUPS + g];
        if (lane < cpg) {
            f_mk = a.f.mask ? (double)a.f.mask[(long long)n * a.f.mask_ld + c] : 1.0;
            f_ga = (double)a.f.gamma[c];
        }
    }
    // forward sum(r) of the group's channels (conv-bias gradient): only the publishing workgroup needs it
    double r1 = 0.0;
    if (sl == 0 && tid < cpg && PH != 1) {
        const int nrep = a.f.rep_s > 0 ? a.f.rep_s : STAT_REP;
        for (int rep = 0; rep < nrep; rep += 4) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = a.f.stats[(((long long)(rep + u < nrep ? rep + u : 0) * a.f.N + n) * C + g * cpg + tid) * 2];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (rep + u < nrep) r1 += v[u];
        }
    }
    if (PH != 2) {
        float q1[8], q2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { q1[j] = 0.f; q2[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float dy[8];
            src[k].sum(dy, vw8);
            const bool ok = (okm >> k) & 1u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xv = to_f(x[k][j]);
                const float d = (ok && fmaf(sc[j], xv, sh[j]) > 0.f) ? dy[j] : 0.f;
                q1[j] += d;
                q2[j] = fmaf(d, xv, q2[j]);
            }
        }
        // lanes with equal (lane % CG) hold the same channels: DPP sums inside the 16-lane rows, the 16 rows of the workgroup meet in LDS
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (CG == 1) { q1[j] = row_sum_mod<1>(q1[j]); q2[j] = row_sum_mod<1>(q2[j]); }
            else if (CG == 2) { q1[j] = row_sum_mod<2>(q1[j]); q2[j] = row_sum_mod<2>(q2[j]); }
            else { q1[j] = row_sum_mod<4>(q1[j]); q2[j] = row_sum_mod<4>(q2[j]); }
        }
        if ((lane & 15) < CG) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { wsum[tid >> 4][lane & 15][j] = q1[j]; wsum[tid >> 4][lane & 15][8 + j] = q2[j]; }
        }
        __syncthreads();
        if (tid < cpg) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) { s1 += wsum[w][tid >> 3][tid & 7]; s2 += wsum[w][tid >> 3][8 + (tid & 7)]; }
            unsigned long long w = ((unsigned long long)__builtin_bit_cast(unsigned, s2) << 32) | (unsigned long long)__builtin_bit_cast(unsigned, s1);
            if (w == 0ull) w = 0x80000000ull;                 // "there": +0.0 travels as -0.0
            xwg_store(slot + sl * cpg + tid, w);
        }
        if (PH == 1) return;
    }
    // every thread fetches one (workgroup, channel) word of the group (S * cpg <= 256)
    {
        const int lc = 31 - __builtin_clz(cpg);
        const int ps = tid >> lc, pc = tid & (cpg - 1);
        float v1 = 0.f, v2 = 0.f;
        if (ps < S) {
            unsigned long long w = xwg_load(slot + ps * cpg + pc);
            for (int spin = 0; w == 0ull && spin < (1 << 18); ++spin) { xwg_pause(); w = xwg_load(slot + ps * cpg + pc); }
            v1 = __builtin_bit_cast(float, (unsigned)(w & 0xffffffffull));
            v2 = __builtin_bit_cast(float, (unsigned)(w >> 32));
            if (w == 0ull) { v1 = __builtin_nanf(""); v2 = v1; }
        }
        part[tid][0] = v1; part[tid][1] = v2;
    }
    __syncthreads();
    if (tid < cpg) {
        double s1 = 0.0, s2 = 0.0;
        for (int s = 0; s < S; ++s) { s1 += (double)part[s * cpg + tid][0]; s2 += (double)part[s * cpg + tid][1]; }
        chan[tid][0] = s1; chan[tid][1] = s2; chan[tid][2] = r1;
    }
    __syncthreads();
    if (wv == 0) {                                       // one wave finishes the group: cpg <= 32 channels (same formulas as gn_bwd_finalize_kernel)
        const int c = g * cpg + (lane < cpg ? lane : 0);
        const bool act = lane < cpg;
        const bool publish = sl == 0;
        const double mu = f_mu, rs = f_rs, mk = act ? f_mk : 0.0, ga = act ? f_ga : 0.0;
        const double Q1 = act ? chan[lane][0] : 0.0, Q2 = act ? chan[lane][1] : 0.0, R1 = act ? chan[lane][2] : 0.0;
        const double q1d = mk * Q1, qx = (mk * Q2 - mu * q1d) * rs;
        const double S1 = wave_sum_d(ga * q1d), S2 = wave_sum_d(ga * qx);
        if (act) {
            const double Mg = (double)cpg * (double)V;
            const double A = rs * ga * mk, B = -rs * rs * S2 / Mg, Cc = -rs * S1 / Mg + rs * rs * S2 * mu / Mg;
            coef[lane][0] = (float)A; coef[lane][1] = (float)B; coef[lane][2] = (float)Cc;
            if (publish) {
                atomicAdd(&a.f.dbeta[c], (float)q1d);
                atomicAdd(&a.f.dgamma[c], (float)qx);
                if (a.f.dbias) atomicAdd(&a.f.dbias[c], (float)(A * Q1 + B * R1 + Cc * (double)V));
            }
        }
    }
    __syncthreads();
    float cA[8], cB[8], cC[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { cA[j] = coef[cg * 8 + j][0]; cB[j] = coef[cg * 8 + j][1]; cC[j] = coef[cg * 8 + j][2]; }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float dy[8];
        src[k].sum(dy, vw8);
        vec<T, 8> o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xv = to_f(x[k][j]);
            const float d = (fmaf(sc[j], xv, sh[j]) > 0.f) ? dy[j] : 0.f;
            o[j] = from_f<T>(fmaf(cA[j], d, fmaf(cB[j], xv, cC[j])));
        }
        if ((okm >> k) & 1u) store8(dr + (long long)chunk_of(k) * 8, o);
    }
}

inline int ew_blocks(long long total_threads) {
    long long b = (total_threads + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

void launch_gn_finalize(const GnFinArgs& a, hipStream_t s, const GnFinArgs* b) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(GN_GROUPS, a.N, b ? 2 : 1), dim3(64), 0, s, a, b ? *b : a);
}

void launch_gn_act(const ActArgs& a, int dtype, hipStream_t s) {
    int bx = ew_blocks(a.V * (a.C / 8));
    if (a.fold) {
        // every workgroup repeats the fold: fewer, longer-running workgroups on the big levels
        static const int cap = 2048;
        const int per_n = cap / a.N > 0 ? cap / a.N : 1;
        if (bx > per_n) bx = per_n;
    }
    dim3 grid(bx, a.N);
#define SEG_ACT2(T_, R2_, RS_) { if (a.fold) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_kernel<T_, true, R2_, RS_>), grid, dim3(256), 0, s, a); \
                                 else hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_kernel<T_, false, R2_, RS_>), grid, dim3(256), 0, s, a); }
#define SEG_ACTH(T_, RS_, HC_) { if (a.fold) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_kernel<T_, true, false, RS_, HC_>), grid, dim3(256), 0, s, a); \
                                 else hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_act_kernel<T_, false, false, RS_, HC_>), grid, dim3(256), 0, s, a); }
#define SEG_ACT(T_) { if (a.head_w && a.head_C == 1) { if (a.res) SEG_ACTH(T_, true, 1) else SEG_ACTH(T_, false, 1) }             \
                      else if (a.head_w && a.head_C == 2) { if (a.res) SEG_ACTH(T_, true, 2) else SEG_ACTH(T_, false, 2) }        \
                      else if (a.head_w) { if (a.res) SEG_ACTH(T_, true, 4) else SEG_ACTH(T_, false, 4) }                         \
                      else if (a.r2 && a.res) SEG_ACT2(T_, true, true) else if (a.r2) SEG_ACT2(T_, true, false) \
                      else if (a.res) SEG_ACT2(T_, false, true) else SEG_ACT2(T_, false, false) }
    if (a.head_w && !gn_act_head_supported(a.C, a.head_C, a.r2 != nullptr)) { fprintf(stderr, "segengine: head in the activation pass: unsupported shape (internal error)\n"); abort(); }
    if (dtype == DT_F32) SEG_ACT(float) else if (dtype == DT_F16) SEG_ACT(f16) else SEG_ACT(bf16)
#undef SEG_ACT
#undef SEG_ACTH
#undef SEG_ACT2
}

void launch_gn_bwd_reduce(const GnBwdArgs& a, int dtype, hipStream_t s) {
    // slab size: ~2048 workgroups in total, at least two rows per thread, at most 1024 rows (small tensors were
    // latency-bound with the fixed 512-row slabs: 16 workgroups x 16 serial round trips)
    const int G = 256 / (a.C / 8);
    long long rows = ((long long)a.N * a.V + 2047) / 2048;
    rows = (rows + G - 1) / G * G;
    if (rows < 2 * G) rows = 2 * G;
    static const int max_rows = 1024;   // measured: 256 -> 686, 512 -> 688, 1024..4096 -> 690.5 volumes/s
    if (rows > max_rows) rows = max_rows / G * G;
    const int GNB_ROWS = (int)rows;
    dim3 grid(cdiv(a.V, GNB_ROWS), a.N);
    const int nv = dy_variant(a);
#define SEG_GNR1(T_, D_, K_) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_reduce_kernel<T_, D_, K_>), grid, dim3(256), 0, s, a, GNB_ROWS)
#define SEG_GNR(T_, D_) { if (nv == 1) SEG_GNR1(T_, D_, 1); else if (nv == 2) SEG_GNR1(T_, D_, 2); else if (nv == 3) SEG_GNR1(T_, D_, 3); \
                          else if (nv == 4) SEG_GNR1(T_, D_, 4); else if (nv == 5) SEG_GNR1(T_, D_, 5); else SEG_GNR1(T_, D_, 0); }
    if (a.r2) { if (dtype == DT_F32) SEG_GNR(float, true) else if (dtype == DT_F16) SEG_GNR(f16, true) else SEG_GNR(bf16, true) }
    else { if (dtype == DT_F32) SEG_GNR(float, false) else if (dtype == DT_F16) SEG_GNR(f16, false) else SEG_GNR(bf16, false) }
#undef SEG_GNR
#undef SEG_GNR1
}

// measured on MI355X: one workgroup per (n,g) only wins while the per-sample tensor is <= ~128 KB (the 6^3 level:
// 12 us vs 22 us for three launches); at 12^3 it ties, at 24^3 it is 4x slower (32 workgroups cannot pull 7 MB)
bool gn_bwd_group_eligible(int C, long long V, int esz) { return C >= 64 && C <= 256 && (long long)C * V * esz <= (128ll << 10); }

void launch_gn_fwd_group(const GnFinArgs& f, const void* r, const void* res, void* out, int dtype, hipStream_t s) {
    GnFwdGroupArgs a; a.f = f; a.r = r; a.res = res; a.out = out;
    dim3 grid(GN_GROUPS, f.N);
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_fwd_group_kernel<float>), grid, dim3(1024), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_fwd_group_kernel<f16>), grid, dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_fwd_group_kernel<bf16>), grid, dim3(1024), 0, s, a);
}

void launch_gn_bwd_group(const GnBwdArgs& e, const GnBwdFinArgs& f, int dtype, hipStream_t s) {
    GnBwdGroupArgs a; a.e = e; a.f = f;
    dim3 grid(GN_GROUPS, e.N);
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_group_kernel<float>), grid, dim3(1024), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_group_kernel<f16>), grid, dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_group_kernel<bf16>), grid, dim3(1024), 0, s, a);
}

// S workgroups of 256 threads per (sample, group), ku <= K chunks per thread: about one workgroup per CU in all, more where a slice would not fit the
// registers (K = 8 chunks of 16-bit data, 4 of f32), never more than 1024 workgroups or 32 per group, S * cpg <= 256 (one polled word per thread).
// Only tensors of a few MB: the reduce + apply launches are latency-bound there; on a larger tensor they stream at the copy rate and one launch that holds the
// tensor in registers is slower (MI355X, round 6, in-call A/B of VNet3d 4 x 96^3: every >= 64-channel level 1052 - 1055 volumes/s, only <= 4 MB 1048 - 1050, none
// 1042 - 1044 - 7 MB at 24^3, 1.8 MB at 12^3 11 us against 13, 0.44 MB at 6^3 8 us against 12.5; 512 instead of 256 workgroups 1030 - 1044; VNet2d 16 x 512^2 with
// its 8 - 34 MB tensors 6.58 ms with the one-launch form against 6.33 ms without)
bool gn_bwd_coop_plan(int C, long long V, int N, int esz, int* S_out, int* ku_out) {
    const int cpg = C / GN_GROUPS;
    if (C % 64 || cpg < 8 || cpg > 32 || N < 1 || V < 1) return false;
    constexpr long long max_bytes = 8000000;
    constexpr int target = 256;
    if ((long long)N * V * C * esz > max_bytes) return false;
    const int CG = cpg / 8, kmax = esz == 2 ? 8 : 4;
    const long long items = V * CG;
    int smax = 256 / cpg;
    if (smax > 32) smax = 32;
    while (smax > 1 && GN_GROUPS * N * smax > 1024) smax >>= 1;
    int S = target / (GN_GROUPS * N);
    if (S < 1) S = 1;
    if (S > smax) S = smax;
    int ku = (int)((items + (long long)S * 256 - 1) / ((long long)S * 256));
    while (ku > kmax && S < smax) { S *= 2; if (S > smax) S = smax; ku = (int)((items + (long long)S * 256 - 1) / ((long long)S * 256)); }
    if (ku > kmax) return false;
    S = (int)((items + (long long)ku * 256 - 1) / ((long long)ku * 256));      // (no empty slices)
    if (S_out) *S_out = S;
    if (ku_out) *ku_out = ku;
    return true;
}
bool gn_bwd_coop_eligible(const GnBwdArgs& e, int esz) {
    return !e.vdl && !e.r2 && e.ndy >= 1 && e.ndy <= 3 && gn_bwd_coop_plan(e.C, e.V, e.N, esz, nullptr, nullptr);
}

void launch_gn_bwd_coop(const GnBwdArgs& e, const GnBwdFinArgs& f, int dtype, hipStream_t s) {
    GnBwdCoopArgs a; a.e = e; a.f = f;
    a.slots = (unsigned long long*)e.Q;
    gn_bwd_coop_plan(e.C, e.V, e.N, dtype == DT_F32 ? 4 : 2, &a.S, &a.ku);
    dim3 grid(GN_GROUPS * a.S, e.N);
#ifdef SEG_EMU
#define SEG_GNC2(T_, D_, K_) { hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_coop_kernel<T_, D_, K_, 1>), grid, dim3(256), 0, s, a); \
                               hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_coop_kernel<T_, D_, K_, 2>), grid, dim3(256), 0, s, a); }
#else
#define SEG_GNC2(T_, D_, K_) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_coop_kernel<T_, D_, K_, 0>), grid, dim3(256), 0, s, a);
#endif
#define SEG_GNC1(T_, D_) { if (a.ku <= 2) SEG_GNC2(T_, D_, 2) else if (a.ku <= 4) SEG_GNC2(T_, D_, 4) else SEG_GNC2(T_, D_, (sizeof(T_) == 2 ? 8 : 4)) }
#define SEG_GNC(T_) { if (e.ndy == 1) SEG_GNC1(T_, 1) else if (e.ndy == 2) SEG_GNC1(T_, 2) else SEG_GNC1(T_, 3) }
    if (dtype == DT_F32) SEG_GNC(float) else if (dtype == DT_F16) SEG_GNC(f16) else SEG_GNC(bf16)
#undef SEG_GNC
#undef SEG_GNC1
#undef SEG_GNC2
}

void launch_gn_bwd_finalize(const GnBwdFinArgs& a, hipStream_t s, const GnBwdFinArgs* b) {
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(GN_GROUPS, a.N, b ? 2 : 1), dim3(64), 0, s, a, b ? *b : a);
}


void launch_gn_bwd_apply(const GnBwdArgs& a, int dtype, hipStream_t s, const GnBwdFinArgs* fa, const GnBwdFinArgs* fb) {
    int bx = ew_blocks(a.V * (a.C / 8));
    const bool fold = fa != nullptr;
    if (fold) {
        static const int cap = 2048;
        const int per_n = cap / a.N > 0 ? cap / a.N : 1;
        if (bx > per_n) bx = per_n;
    }
    dim3 grid(bx, a.N);
    const GnBwdFinArgs za = fa ? *fa : GnBwdFinArgs{}, zb = fb ? *fb : GnBwdFinArgs{};
    const int nv = dy_variant(a);
#define SEG_GNA1(T_, D_, K_) { if (fold) hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_apply_kernel<T_, D_, true, K_>), grid, dim3(256), 0, s, a, za, zb); \
                               else hipLaunchKernelGGL(HIP_KERNEL_NAME(gn_bwd_apply_kernel<T_, D_, false, K_>), grid, dim3(256), 0, s, a, za, zb); }
#define SEG_GNA(T_, D_) { if (nv == 1) SEG_GNA1(T_, D_, 1) else if (nv == 2) SEG_GNA1(T_, D_, 2) else if (nv == 3) SEG_GNA1(T_, D_, 3) \
                          else if (nv == 4) SEG_GNA1(T_, D_, 4) else if (nv == 5) SEG_GNA1(T_, D_, 5) else SEG_GNA1(T_, D_, 0) }
    if (a.r2) { if (dtype == DT_F32) SEG_GNA(float, true) else if (dtype == DT_F16) SEG_GNA(f16, true) else SEG_GNA(bf16, true) }
    else { if (dtype == DT_F32) SEG_GNA(float, false) else if (dtype == DT_F16) SEG_GNA(f16, false) else SEG_GNA(bf16, false) }
#undef SEG_GNA
#undef SEG_GNA1
}

}  // namespace seg
