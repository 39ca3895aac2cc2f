// Generic weight-gradient GEMM for gfx950:  dW[p][tap][q] = sum_m dR[m][p] * X[vox(m,tap)][q]
// (conv 2^d s2, conv 1^d on a virtual concat, ConvTranspose 2^d s2, the image stem, and 3^d convs
// whose input is a concat; the plain 3^d convs use the halo kernel in conv3.hip).
// Both operands are reduced along the voxel axis m, the SLOW axis of the channels-last tensors, so
// both MFMA fragments need a transpose: tiles are staged row-major [m][c] in LDS and read with
// ds_read_b64_tr_b16 (f16/bf16) or element-wise (f32).  A workgroup owns a (p-tile, q-tile) pair for
// up to 8 taps (all taps of the 2^d kernels: their windows do not overlap, so every voxel is read
// once) and a slice of the voxel axis; global loads of step s+1 are in flight while step s runs on
// the matrix cores.  Each workgroup writes ONE partial tile; wgrad_reduce_kernel sums the slices
// (deterministic, no same-address atomics).
// Pinned by autograd of the conv call sites in networks/VNet3d.py / Unet3d.py (reference).
#include "kernels.h"
#include <cstdlib>

namespace seg {
namespace {

constexpr int WM = 32;        // voxel rows per step (one MFMA K-step)
constexpr int WT = 64;        // max tile extent along p and q
constexpr int LDW = WT + 8;
constexpr int MAXTB = 8;      // taps handled inside one workgroup

struct RowCoord { int n, d, h, w; };
__device__ __forceinline__ RowCoord decode_row(long long m64, int D, int H, int W) {
    int m = (int)m64;                                   // row counts stay far below 2^31 (launcher checks); 32-bit division
    RowCoord r;
    r.w = m % W; m /= W;
    r.h = m % H; m /= H;
    r.d = m % D;
    r.n = m / D;
    return r;
}

template <class T> struct TFrag {
    static __device__ __forceinline__ typename Mma<T>::frag load(const T* tile, int col0, int lane) {
        const int t = lane & 15, q = lane >> 4;
        const T* p0 = tile + (4 * q + (t >> 2)) * LDW + col0 + (t & 3) * 4;
        const s16x4 lo = lds_read_tr16(p0);
        const s16x4 hi = lds_read_tr16(p0 + 16 * LDW);
        vec<short, 8> v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        return __builtin_bit_cast(typename Mma<T>::frag, v);
    }
};
template <> struct TFrag<float> {
    static __device__ __forceinline__ Mma<float>::frag load(const float* tile, int col0, int lane) {
        const int t = lane & 15, q = lane >> 4;
        Mma<float>::frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = tile[(4 * j + q) * LDW + col0 + t];
        return f;
    }
};

struct WgPlan { int TP, TQ, ntq, ntile, TB, ntg; long long Mc; int parts; int direct; };   // direct: 1^d stride-1 conv, voxel == row

// partial layout per voxel slice: [P][T][Qc]  (Qc = Q, or the (tap,ci) column count of the stem)
#ifndef SEG_EMU
#define SEG_WG_WAVES(x) __attribute__((amdgpu_waves_per_eu(x)))     // lower bound on resident waves per SIMD = upper bound on VGPRs
#else
#define SEG_WG_WAVES(x)
#endif
// (the 8-tap form allocated 184 VGPRs - two waves per SIMD - where its 42 KB of LDS allow three workgroups per CU)
template <class T, bool STEM, int NTB>
__global__ __launch_bounds__(256) SEG_WG_WAVES(NTB == MAXTB && sizeof(T) == 2 ? 3 : 1) void wgrad_kernel(WgradArgs a, WgPlan pl, float* partial) {
    // voxel rows per staging step: 128 for the single-tap kernels (4 MFMA K-steps per barrier pair), 32 otherwise
    constexpr int WMT = (NTB == 1 && !STEM) ? 128 : WM;
    constexpr int NPC = WMT * 8 / 256;                  // staging pieces (16 B) per thread, upper bound
    __shared__ T Ds[WMT * LDW];
    __shared__ T Xs[NTB * WMT * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tg = (int)(blockIdx.x / pl.ntile), tile = (int)(blockIdx.x % pl.ntile);
    const int tap0 = tg * NTB;
    const int TP = pl.TP, TQ = pl.TQ;
    const int p0 = (tile / pl.ntq) * TP, q0 = (tile % pl.ntq) * TQ;
    const long long M = (long long)a.N * a.OD * a.OH * a.OW;
    const long long mbeg = (long long)blockIdx.y * pl.Mc;
    const long long mend = (mbeg + pl.Mc < M) ? mbeg + pl.Mc : M;
    const T* dr = (const T*)a.dr;
    const T* x0 = (const T*)a.x0;
    const T* x1 = (const T*)a.x1;
    const int ntaps = STEM ? 1 : a.taps.n;
    const int ntb = (ntaps - tap0 < NTB) ? ntaps - tap0 : NTB;      // taps of this workgroup

    const int nt_p = TP / 16, nt_q = (TQ + 15) / 16, n16 = nt_p * nt_q;   // 16x16 tiles per tap
    const int nwork = ntb * n16;                                          // (tap, tile) work items, split over the 4 waves
    constexpr int MAXW = NTB == 1 ? 4 : 8;
    f32x4 acc[MAXW];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: piece = (row, 8-channel chunk); thread handles pieces tid, tid+256, ...
    const int cpr_p = TP / 8, cpr_q = STEM ? 8 : TQ / 8;
    const int npd = WMT * cpr_p, npx = WMT * cpr_q;

    vec<T, 8> dreg[NPC], xreg[NTB][NPC];
    unsigned xok[NTB];                                  // bit u: piece u of tap t holds data (else zeros go to LDS)
    float sreg[4];
    // tap offsets of this workgroup, read ONCE (a dynamic index into the argument segment is a vector-memory load; inside gload each of
    // them carried its own s_waitcnt vmcnt(0))
    int tpd[NTB], tph[NTB], tpw[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
        const int tp = STEM ? 0 : tap0 + (t < ntb ? t : 0);
        tpd[t] = a.taps.d[tp]; tph[t] = a.taps.h[tp]; tpw[t] = a.taps.w[tp];
        settle(tpd[t]); settle(tph[t]); settle(tpw[t]);
        xok[t] = 0u;
    }
    unsigned dok = 0u;                                  // bit u: piece u of the dR tile holds data
    auto gload = [&](long long ms) {
        if (STEM) {
#pragma unroll
            for (int u = 0; u < NPC; ++u) {
                const int pc = u * 256 + tid;
                if (pc < npd) {
                    const long long m = ms + pc / cpr_p;
                    dreg[u] = (m < mend) ? load8(dr + m * a.P + p0 + (pc % cpr_p) * 8) : zero8<T>();
                    dok |= 1u << u;
                }
            }
            const int xrow_ = tid >> 3, xcc = tid & 7;
            const long long m = ms + xrow_;
            const bool mv = m < mend;
            const RowCoord r = decode_row(mv ? m : 0, a.OD, a.OH, a.OW);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = xcc * 4 + j;
                float xv = 0.f;
                if (mv && col < a.Q) {
                    const int tp = col / a.C0, ci = col % a.C0;
                    const int id = r.d + a.taps.d[tp], ih = r.h + a.taps.h[tp], iw = r.w + a.taps.w[tp];
                    if ((unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
                        xv = to_f(x0[((((long long)r.n * a.ID + id) * a.IH + ih) * a.IW + iw) * a.C0 + ci]);
                }
                sreg[j] = xv;
            }
            return;
        }
        // Two phases with a scheduling fence between them: every ADDRESS first, then every load back to back.  Rows past the slice and taps
        // outside the source read a valid (clamped) address and are zeroed when the piece goes to LDS (dok / xok); a concat source is a
        // pointer select in front of ONE load.  What this replaces: `cond ? load8(a) : load8(b)` and zero-or-load merges made hipcc put an
        // s_waitcnt vmcnt(0) behind every piece, and address temporaries allocated on top of load destinations added vmcnt(3..1) waits
        // between the loads - the "loads in flight while the MFMAs run" arrived one or two at a time.
        const T* dsrc[NPC];
        const T* xsrc[NTB][NPC];
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int pc = u * 256 + tid;
            const int pcd = pc < npd ? pc : 0;
            const long long md = ms + pcd / cpr_p;
            const bool dv = pc < npd && md < mend;
            dsrc[u] = dr + (dv ? md : mbeg) * a.P + p0 + (pcd % cpr_p) * 8;
            dok = dv ? (dok | (1u << u)) : (dok & ~(1u << u));
            const int pcx = pc < npx ? pc : 0;
            const long long m = ms + pcx / cpr_q;
            const bool mv = pc < npx && m < mend;
            const long long mc = mv ? m : mbeg;
            const int qc = q0 + (pcx % cpr_q) * 8;
            const bool from0 = qc < a.C0;
            if (NTB == 1 && pl.direct) {                // 1^d stride-1 conv: the gathered voxel is the row itself
                xsrc[0][u] = from0 ? x0 + mc * a.C0 + qc : x1 + mc * a.C1 + (qc - a.C0);
                xok[0] = mv ? (xok[0] | (1u << u)) : (xok[0] & ~(1u << u));
                continue;
            }
            const RowCoord r = decode_row(mc, a.OD, a.OH, a.OW);
#pragma unroll
            for (int t = 0; t < NTB; ++t) {
                const int id = r.d * a.sd + tpd[t], ih = r.h * a.sh + tph[t], iw = r.w * a.sw + tpw[t];
                const bool inb = mv && t < ntb && (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW;
                const long long vox = inb ? (((long long)r.n * a.ID + id) * a.IH + ih) * a.IW + iw : 0;
                xsrc[t][u] = from0 ? x0 + vox * a.C0 + qc : x1 + vox * a.C1 + (qc - a.C0);
                xok[t] = inb ? (xok[t] | (1u << u)) : (xok[t] & ~(1u << u));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            dreg[u] = load8(dsrc[u]);
#pragma unroll
            for (int t = 0; t < NTB; ++t) xreg[t][u] = load8(xsrc[t][u]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sstore = [&]() {
#pragma unroll
        for (int u = 0; u < NPC; ++u) {
            const int pc = u * 256 + tid;
            if (pc < npd) store8(&Ds[(pc / cpr_p) * LDW + (pc % cpr_p) * 8], ((dok >> u) & 1u) ? dreg[u] : zero8<T>());
        }
        if (STEM) {
            const int xrow_ = tid >> 3, xcc = tid & 7;
#pragma unroll
            for (int j = 0; j < 4; ++j) Xs[xrow_ * LDW + xcc * 4 + j] = from_f<T>(sreg[j]);
        } else {
#pragma unroll
            for (int u = 0; u < NPC; ++u) {
                const int pc = u * 256 + tid;
                if (pc < npx) {
#pragma unroll
                    for (int t = 0; t < NTB; ++t)
                        store8(&Xs[(t * WMT + pc / cpr_q) * LDW + (pc % cpr_q) * 8], ((xok[t] >> u) & 1u) ? xreg[t][u] : zero8<T>());
                }
            }
        }
    };

    if (mbeg < mend) gload(mbeg);
    for (long long ms = mbeg; ms < mend; ms += WMT) {
        sstore();
        __syncthreads();
        if (ms + WMT < mend) gload(ms + WMT);            // next step's loads fly while the MFMAs run
#pragma unroll
        for (int i = 0; i < MAXW; ++i) {
            const int wi = wv + 4 * i;
            if (wi < nwork) {
                const int t = wi / n16, tt = wi % n16;
                const int pi = tt / nt_q, qi = tt % nt_q;
#pragma unroll
                for (int kk = 0; kk < WMT / 32; ++kk) {
                    const typename Mma<T>::frag af = TFrag<T>::load(Ds + kk * 32 * LDW, pi * 16, lane);
                    const typename Mma<T>::frag bf = TFrag<T>::load(Xs + (t * WMT + kk * 32) * LDW, qi * 16, lane);
                    acc[i] = Mma<T>::run(af, bf, acc[i]);
                }
            }
        }
        __syncthreads();
    }
    // ---- partial tile of this voxel slice
    const int l15 = lane & 15, q = lane >> 4;
    const int Tn = STEM ? 1 : a.taps.n, Qc = a.Q;
    float* dst = partial + (long long)blockIdx.y * a.P * Tn * Qc;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        const int wi = wv + 4 * i;
        if (wi < nwork) {
            const int t = wi / n16, tt = wi % n16;
            const int pi = tt / nt_q, qi = tt % nt_q;
            const int qq = q0 + qi * 16 + l15;
            if (qq < Qc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = p0 + pi * 16 + 4 * q + r;
                    dst[((long long)p * Tn + tap0 + t) * Qc + qq] = acc[i][r];
                }
            }
        }
    }
}

// The 1^d stride-1 convolutions on 16-bit tensors (the concat -> 1^d conv of every decoder level, networks/VNet3d.py:66-77): the gathered voxel
// is the row itself, so the kernel is a pure stream of two channels-last tensors - 340 MB at 96^3 - and wgrad_kernel above ran it at
// 1.7 TB/s: ONE staging step (128 rows = 12 KB per workgroup) in flight, i.e. 48 KB per CU against a loaded-memory latency of several
// microseconds (profiles/r04_rocprofv3_kernel_stats.txt: 204 us).  Here DEPTH steps travel at once: each thread keeps DEPTH register sets of
// its NPD + NPX 16-byte pieces, statically indexed (the step loop is unrolled DEPTH times), so the compiler's vmcnt waits for the OLDEST set
// only.  Tile extents are template parameters (TP = 16 NPD, TQ = 16 NPX: exactly NPD / NPX pieces per thread and step).  Same partial-tile
// layout, same order of the fp32 sums over the voxel axis as wgrad_kernel: bit-identical results.
// ACT: x0 holds the RAW output of a conv + GroupNorm unit; its pieces become relu(scale * r + shift) rounded to T on their way into LDS (see
// conv_stream_kernel<..., ACT>).  The coefficients of the (at most two) samples a workgroup's voxel slice touches are parked in LDS.
template <class T, int NPD, int NPX, int DEPTH, bool ACT = false>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WgradArgs a, WgPlan pl, float* partial) {
    constexpr int WMT = 128, TP = 16 * NPD, TQ = 16 * NPX;
    constexpr int cpr_p = TP / 8, cpr_q = TQ / 8;                // 16-byte pieces per row
    __shared__ T Ds[WMT * LDW];
    __shared__ T Xs[WMT * LDW];
    __shared__ __attribute__((aligned(32))) float coef_s[ACT ? 2 : 1][2][ACT ? 64 : 8];       // [sample of the slice][scale | shift][channel of x0]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = (int)blockIdx.x;
    const int p0 = (tile / pl.ntq) * TP, q0 = (tile % pl.ntq) * TQ;
    const long long M = (long long)a.N * a.OD * a.OH * a.OW;
    const long long mbeg = (long long)blockIdx.y * pl.Mc;
    const long long mend = (mbeg + pl.Mc < M) ? mbeg + pl.Mc : M;
    const T* dr = (const T*)a.dr;
    constexpr int nt_q = NPX, n16 = NPD * NPX;                   // 16x16 output tiles, split over the 4 waves
    constexpr int MAXW = (n16 + 3) / 4;
    f32x4 acc[MAXW];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this thread's pieces: (row inside the step, byte-free element offset of the 8-channel chunk); the x chunk comes from one of the two
    // concat sources (pointer select once, outside the loop)
    int drow[NPD], xrow[NPX];
    const T* dbase[NPD];
    const T* xbase[NPX];
    long long xstride[NPX];
    int xact[NPX];                                               // ACT: channel offset of the piece inside x0, -1: a piece of x1
    const long long Vs = (long long)a.OD * a.OH * a.OW;          // rows per sample
    const long long nb = (mbeg / Vs + 1) * Vs;                   // first row of the slice's second sample
    if (ACT) {
        if (tid < 2 * a.C0) {
            const int set = tid / a.C0, c = tid % a.C0;
            long long n = mbeg / Vs + set;
            if (n > a.N - 1) n = a.N - 1;
            coef_s[set][0][c] = a.act_scale[n * a.C0 + c];
            coef_s[set][1][c] = a.act_shift[n * a.C0 + c];
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NPD; ++u) {
        const int pc = u * 256 + tid;
        drow[u] = pc / cpr_p;
        dbase[u] = dr + p0 + (pc % cpr_p) * 8;
    }
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
        const int pc = u * 256 + tid;
        xrow[u] = pc / cpr_q;
        const int qc = q0 + (pc % cpr_q) * 8;
        const bool from0 = qc < a.C0;
        xbase[u] = from0 ? (const T*)a.x0 + qc : (const T*)a.x1 + (qc - a.C0);
        xstride[u] = from0 ? a.C0 : a.C1;
        xact[u] = from0 ? qc : -1;
    }
    vec<T, 8> dq[DEPTH][NPD], xq[DEPTH][NPX];
    unsigned okq[DEPTH];                                         // bit u: d piece u, bit 8 + u: x piece u of that set holds data
    auto issue = [&](int slot, long long ms) {                   // rows past the slice read a valid (clamped) row and are zeroed on the way to LDS
        unsigned ok = 0u;
#pragma unroll
        for (int u = 0; u < NPD; ++u) {
            const long long m = ms + drow[u];
            const bool v = m < mend;
            ok |= v ? (1u << u) : 0u;
            dq[slot][u] = load8(dbase[u] + (v ? m : mbeg) * a.P);
        }
#pragma unroll
        for (int u = 0; u < NPX; ++u) {
            const long long m = ms + xrow[u];
            const bool v = m < mend;
            ok |= v ? (1u << (8 + u)) : 0u;
            xq[slot][u] = load8(xbase[u] + (v ? m : mbeg) * xstride[u]);
        }
        okq[slot] = ok;
    };
#pragma unroll
    for (int k = 0; k < DEPTH - 1; ++k) issue(k, mbeg + (long long)k * WMT);
    for (long long ms = mbeg; ms < mend; ms += (long long)DEPTH * WMT) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            // every body runs, also past the end of the slice (zero tiles: exact zeros are added): a conditional around the loads makes the
            // compiler's vmcnt bookkeeping give up at the join and wait for ALL older sets at once
            const long long m0 = ms + (long long)k * WMT;
            {
                issue((k + DEPTH - 1) % DEPTH, m0 + (long long)(DEPTH - 1) * WMT);
#pragma unroll
                for (int u = 0; u < NPD; ++u)
                    store8(&Ds[drow[u] * LDW + ((u * 256 + tid) % cpr_p) * 8], ((okq[k] >> u) & 1u) ? dq[k][u] : zero8<T>());
#pragma unroll
                for (int u = 0; u < NPX; ++u) {
                    vec<T, 8> xv = xq[k][u];
                    if (ACT) {
                        const int set = (m0 + xrow[u]) >= nb ? 1 : 0;
                        const int c = xact[u] >= 0 ? xact[u] : 0;
                        const vec<float, 8> sc = *(const vec<float, 8>*)&coef_s[set][0][c], sh = *(const vec<float, 8>*)&coef_s[set][1][c];
                        vec<T, 8> av;
#pragma unroll
                        for (int e = 0; e < 8; ++e) av[e] = from_f<T>(fmaxf(fmaf(sc[e], to_f(xv[e]), sh[e]), 0.f));
                        if (xact[u] >= 0) xv = av;
                    }
                    store8(&Xs[xrow[u] * LDW + ((u * 256 + tid) % cpr_q) * 8], ((okq[k] >> (8 + u)) & 1u) ? xv : zero8<T>());
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < MAXW; ++i) {
                    const int wi = wv + 4 * i;
                    if (wi < n16) {
                        const int pi = wi / nt_q, qi = wi % nt_q;
#pragma unroll
                        for (int kk = 0; kk < WMT / 32; ++kk) {
                            const typename Mma<T>::frag af = TFrag<T>::load(Ds + kk * 32 * LDW, pi * 16, lane);
                            const typename Mma<T>::frag bf = TFrag<T>::load(Xs + kk * 32 * LDW, qi * 16, lane);
                            acc[i] = Mma<T>::run(af, bf, acc[i]);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    const int l15 = lane & 15, q = lane >> 4;
    const int Qc = a.Q;
    float* dst = partial + (long long)blockIdx.y * a.P * Qc;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        const int wi = wv + 4 * i;
        if (wi < n16) {
            const int pi = wi / nt_q, qi = wi % nt_q;
            const int qq = q0 + qi * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(long long)(p0 + pi * 16 + 4 * q + r) * Qc + qq] = acc[i][r];
        }
    }
}

// dw[p*sP + q*sQ + tap*sT] += sum_slices partial[slice][p][tap][q]    (stem: column q = (tap, ci))
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* partial, float* dw, int P, int Tn, int Qc, int parts, long long sP,
                                                           long long sQ, long long sT, int stem_cimg, int qreal) {
    // grid.y slices the partial list (32 per slice); slices meet in dw through one atomic each
    const long long total = (long long)P * Tn * Qc;
    const int b0 = blockIdx.y * 32, b1 = (b0 + 32 < parts) ? b0 + 32 : parts;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = b0;
        for (; b + 4 <= b1; b += 4) {
            s0 += partial[(long long)b * total + i]; s1 += partial[(long long)(b + 1) * total + i];
            s2 += partial[(long long)(b + 2) * total + i]; s3 += partial[(long long)(b + 3) * total + i];
        }
        for (; b < b1; ++b) s0 += partial[(long long)b * total + i];
        const int qq = (int)(i % Qc), tap = (int)((i / Qc) % Tn), p = (int)(i / ((long long)Qc * Tn));
        if (!stem_cimg && qq >= qreal) continue;        // zero-padded input channels: no such weight
        long long o;
        if (stem_cimg) o = p * sP + (qq % stem_cimg) * sQ + (qq / stem_cimg) * sT;
        else o = p * sP + qq * sQ + tap * sT;
        const float tot = (s0 + s1) + (s2 + s3);
        if (gridDim.y == 1) dw[o] += tot;
        else atomicAdd(&dw[o], tot);
    }
}

WgPlan make_plan(const WgradArgs& a) {
    WgPlan pl;
    const long long M = (long long)a.N * a.OD * a.OH * a.OW;
    const int T = a.stem ? 1 : a.taps.n;
    pl.TB = (T <= MAXTB) ? T : 1;                      // taps per workgroup
    int cap = (pl.TB > 1) ? 32 : WT;                   // keep the accumulator count per wave <= 8
    pl.TP = a.P < cap ? a.P : cap;
    if (a.stem) { pl.TQ = 32; pl.ntq = 1; }
    else { pl.TQ = a.Q < cap ? a.Q : cap; pl.ntq = a.Q / pl.TQ; }
    pl.ntile = (a.P / pl.TP) * pl.ntq;
    pl.ntg = (T + pl.TB - 1) / pl.TB;
    const int bx = pl.ntg * pl.ntile;
    // voxel-axis split: ~2048 workgroups in total, at most 1024 slices and 16 MB of partial tiles
    static const long long total = 2048;        // tuning knobs
    static const long long pcap = ((4ll << 20));
    long long parts = total / bx;
    if (parts < 1) parts = 1;
    if (parts > 1024) parts = 1024;
    const long long elems = (long long)a.P * T * a.Q;
    if (parts * elems > pcap) parts = pcap / elems;
    if (parts < 1) parts = 1;
    const long long maxp = (M + WM - 1) / WM;
    if (parts > maxp) parts = maxp;
    const int step = (pl.TB == 1 && !a.stem) ? 128 : WM;
    long long Mc = (M + parts - 1) / parts;
    Mc = (Mc + step - 1) / step * step;
    pl.Mc = Mc;
    pl.direct = (!a.stem && T == 1 && a.sd == 1 && a.sh == 1 && a.sw == 1 && a.taps.d[0] == 0 && a.taps.h[0] == 0 && a.taps.w[0] == 0 &&
                 a.ID == a.OD && a.IH == a.OH && a.IW == a.OW) ? 1 : 0;
    pl.parts = (int)((M + Mc - 1) / Mc);
    return pl;
}

template <class T>
void wgrad_dispatch(const WgradArgs& a, float* partial, hipStream_t s, int qreal) {
    const WgPlan pl = make_plan(a);
    dim3 grid(pl.ntg * pl.ntile, pl.parts);
    static const bool direct_on = (knob_i("SEG_WG_DIRECT", 1) != 0);       // A/B switch
    bool done = false;
    if constexpr (sizeof(T) == 2) {
        if ((direct_on || a.act_scale) && !a.stem && pl.TB == 1 && pl.direct && pl.ntg == 1 && a.C0 % 8 == 0) {
#define SEG_WGD(ND, NX, DP) if (!done && pl.TP == 16 * ND && pl.TQ == 16 * NX) {                                                                     \
                if (a.act_scale) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_direct_kernel<T, ND, NX, DP, true>), grid, dim3(256), 0, s, a, pl, partial);  \
                else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_direct_kernel<T, ND, NX, DP>), grid, dim3(256), 0, s, a, pl, partial);                   \
                done = true; }
            SEG_WGD(1, 2, 4) SEG_WGD(2, 4, 3) SEG_WGD(4, 4, 3) SEG_WGD(1, 1, 4) SEG_WGD(2, 2, 4)
#undef SEG_WGD
        }
    }
    if (a.act_scale && !done) { fprintf(stderr, "segengine: activation on load needs the direct weight-gradient kernel (internal error)\n"); abort(); }
    if (done) {}
    else if (a.stem) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<T, true, 1>), grid, dim3(256), 0, s, a, pl, partial);
    else if (pl.TB == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<T, false, 1>), grid, dim3(256), 0, s, a, pl, partial);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<T, false, MAXTB>), grid, dim3(256), 0, s, a, pl, partial);
    const int Tn = a.stem ? 1 : a.taps.n;
    const long long total = (long long)a.P * Tn * a.Q;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks, (pl.parts + 31) / 32), dim3(256), 0, s, (const float*)partial, a.dw, a.P, Tn, a.Q, pl.parts, a.sP, a.sQ,
                       a.sT, a.stem ? a.C0 : 0, (qreal > 0 && qreal < a.Q) ? qreal : a.Q);
}

}  // namespace

// the direct kernel with activation on load applies: 16-bit tensors (checked by the caller), a 1^d stride-1 conv whose tile extents have an instantiation,
// x0 of at most 64 channels, voxel slices that touch at most two samples
bool wgrad_act_supported(const WgradArgs& a) {
    if (a.stem || a.C0 % 8 || a.C0 > 64) return false;
    const WgPlan pl = make_plan(a);
    if (!(pl.TB == 1 && pl.direct && pl.ntg == 1)) return false;
    if (pl.Mc > (long long)a.OD * a.OH * a.OW && a.N > 2) return false;       // (a slice may then touch more than two samples)
    const int np = pl.TP / 16, nq = pl.TQ / 16;
    return pl.TP % 16 == 0 && pl.TQ % 16 == 0 &&
           ((np == 1 && nq == 2) || (np == 2 && nq == 4) || (np == 4 && nq == 4) || (np == 1 && nq == 1) || (np == 2 && nq == 2));
}

size_t wgrad_partial_bytes(const WgradArgs& a) {
    const WgPlan pl = make_plan(a);
    return (size_t)pl.parts * a.P * (a.stem ? 1 : a.taps.n) * a.Q * sizeof(float);
}

void launch_wgrad(const WgradArgs& a, float* partial, int dtype, hipStream_t s, int qreal) {
    if (dtype == DT_F32) wgrad_dispatch<float>(a, partial, s, qreal);
    else if (dtype == DT_F16) wgrad_dispatch<f16>(a, partial, s, qreal);
    else wgrad_dispatch<bf16>(a, partial, s, qreal);
}

}  // namespace seg
