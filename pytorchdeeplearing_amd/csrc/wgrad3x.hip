// Weight gradient of the 3^d / 3^2 stride-1 pad-1 convolutions on 16-bit tensors, double-buffered:
//     dW[p][q][tap] = sum_v dR[v][p] * X[v + tap][q]            (autograd of networks/VNet3d.py:8, networks/Unet3d.py:66-80)
// Same decomposition as wgrad3_kernel (conv3.hip): a workgroup owns a [CP x taps x CQ] tile, walks a strided list of boxes and
// writes ONE partial tile (wgrad3_reduce_kernel sums them).  What changes is how a box reaches the MFMAs.  The round-1 kernel
// staged dR and the X halo through registers and spent 3 - 4 us per box waiting for that (1 us of MFMA work per box), one box
// after the other.  Here both tiles are copied global -> LDS directly (common.h dma16_async) into one of TWO buffers: the copies of box
// i + 1 are in flight while box i is multiplied, and the only wait is the one barrier per box.  Rows are unpadded (64 B for a
// 32-channel tile) with the 32-B halves swapped on rows 4..7 of every 8, which makes the transposing reads
// (ds_read_b64_tr_b16: 32 lanes x 8 B per pass) conflict-free without the 96-B row padding a lane-linear copy cannot produce.
// 8 waves per workgroup, each owning one 16-channel q sub-tile and a group of taps (7 + 7 + 7 + 6 for a 32-channel q-tile), two
// waves per SIMD covering each other's LDS latency.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace seg {
namespace {

template <int TD_, int TH_, int TW_, int KD_> struct WBox {
    static constexpr int TD = TD_, TH = TH_, TW = TW_, KD = KD_;
    static constexpr int V = TD * TH * TW;
    static constexpr int HD = TD + KD - 1, HH = TH + 2, HW = TW + 2, HV = HD * HH * HW;
    static constexpr int PD = (KD - 1) / 2, NTAP = KD * 9;
    static __device__ __forceinline__ int halo_base(int v) {
        const int vx = v % TW, vy = (v / TW) % TH, vz = v / (TW * TH);
        return (vz * HH + vy) * HW + vx;
    }
    static __device__ __forceinline__ int tap_off(int t) {
        const int kw = t % 3, kh = (t / 3) % 3, kd = t / 9;
        return (kd * HH + kh) * HW + kw;
    }
};

struct Wgrad3xArgs {
    const void* dr; const void* x0; const void* x1; int C0;     // x1: second source of a virtual concat (channels C0..Q-1) or null
    float* partial;
    int N, D, H, W, P, Q;
    int nb;                                                     // workgroups per (p-tile, q-tile) combo
};

constexpr int W3X_WAVES = 8;

// Swizzle of the 64-B rows of a 32-channel tile (none for 32-B rows): the 16-B piece `slot` of a row is stored at
// slot ^ 2*s, s = (v >> 2) & 1 for row v of the dR tile and s = (hx >> 2) & 1 for the halo voxel at x position hx.  One pass of a
// transposing read covers 8 consecutive voxels of one x row: rows R and R + 4 share their banks, s tells them apart.  Keying
// the halo swizzle on hx (not on the linear row index) makes the reader's bit a function of (vx + kw) only: three values per
// lane, so every tap offset stays an immediate of the ds_read.
template <int C> __device__ __forceinline__ int piece_swz(int s) { return C == 32 ? ((s & 1) << 1) : 0; }

// One box multiplied by one wave: the wave owns q sub-tile jw and the taps [T0, T0 + NTG) (compile-time), both p sub-tiles.
template <class T, class B, int CP, int CQ, int T0, int NTG, int NACC>
__device__ __forceinline__ void w3x_box(const T* Ds, const T* Xs, const int (&da)[B::V / 32][2][CP / 16], const int (&xb)[B::V / 32][2][3],
                                        f32x4 (&acc)[NACC][CP / 16]) {
    constexpr int PT = CP / 16, KS = B::V / 32;
    auto frag2 = [](s16x4 lo, s16x4 hi) -> typename Mma<T>::frag {
        vec<short, 8> v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        return __builtin_bit_cast(typename Mma<T>::frag, v);
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        typename Mma<T>::frag af[PT];
#pragma unroll
        for (int i = 0; i < PT; ++i) af[i] = frag2(lds_read_tr16(Ds + da[ks][0][i]), lds_read_tr16(Ds + da[ks][1][i]));
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            constexpr int dummy = 0; (void)dummy;
            const int tap = T0 + t;
            if (tap < B::NTAP) {
                const int kw = tap % 3, toff = B::tap_off(tap) * CQ;          // immediates
                const typename Mma<T>::frag bf = frag2(lds_read_tr16(Xs + toff + xb[ks][0][kw]), lds_read_tr16(Xs + toff + xb[ks][1][kw]));
#pragma unroll
                for (int i = 0; i < PT; ++i) acc[t][i] = Mma<T>::run(af[i], bf, acc[t][i]);
            }
        }
    }
}

template <class T, class B, int CP, int CQ>
__global__ __launch_bounds__(W3X_WAVES * 64, 1) void wgrad3x_kernel(Wgrad3xArgs a) {
    static_assert(sizeof(T) == 2, "16-bit run dtypes only");
    static_assert(B::V % 32 == 0, "box must hold a multiple of 32 voxels");
    static_assert(B::TW == 16 || B::TW == 8, "a transposing-read pass must stay inside one x row");
    constexpr int NW = W3X_WAVES, NT = NW * 64;
    constexpr int PT = CP / 16, QT = CQ / 16, CPV = CP / 8, CQV = CQ / 8;
    constexpr int DI = (B::V * CPV + NT - 1) / NT, XI = (B::HV * CQV + NT - 1) / NT;   // copy instructions per wave (D tile, X halo)
    constexpr int DELEMS = DI * NT * 8, XELEMS = XI * NT * 8, BUF = DELEMS + XELEMS;
    constexpr int NG = NW / QT;                          // tap groups: a wave owns (q sub-tile, tap group)
    constexpr int NTG = (B::NTAP + NG - 1) / NG, KS = B::V / 32;
    __shared__ __attribute__((aligned(16))) T S[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int jw = wv % QT, grp = __builtin_amdgcn_readfirstlane(wv / QT);    // wave-uniform: the switch below is a scalar branch
    const int nqt = a.Q / CQ;
    const int combo = blockIdx.y, p0 = (combo / nqt) * CP, q0 = (combo % nqt) * CQ;
    const int nbx = (a.W + B::TW - 1) / B::TW, nby = (a.H + B::TH - 1) / B::TH, nbz = (a.D + B::TD - 1) / B::TD;
    const int nbox = a.N * nbz * nby * nbx;
    const long long vol = (long long)a.D * a.H * a.W;

    // sources: dR [N][vol][P]; X is x0 ([..][C0]) or x1 ([..][Q - C0]) for this q-tile (tiles never straddle: CQ divides C0)
    const bool second = a.x1 != nullptr && q0 >= a.C0;
    const int xC = a.x1 ? (second ? a.Q - a.C0 : a.C0) : a.Q, xq0 = second ? q0 - a.C0 : q0;
    const i32x4 rd = make_rsrc(a.dr, (unsigned)(a.N * vol * a.P * 2));
    const i32x4 rx = make_rsrc(second ? a.x1 : a.x0, (unsigned)(a.N * vol * xC * 2));

    // ---- per-lane description of the granules this lane copies (box-independent): packed (z, y, x) offsets inside the box /
    //      halo and the channel piece; -1: padding of the copy grid
    int dsrc[DI], xsrc[XI];
#pragma unroll
    for (int u = 0; u < DI; ++u) {
        const int g = (u * NW + wv) * 64 + lane, v = g / CPV, slot = g % CPV;
        const int vx = v % B::TW, vy = (v / B::TW) % B::TH, vz = v / (B::TW * B::TH);
        dsrc[u] = v < B::V ? (((vz << 8 | vy) << 8 | vx) << 2) | (slot ^ piece_swz<CP>(v >> 2)) : -1;
    }
#pragma unroll
    for (int u = 0; u < XI; ++u) {
        const int g = (u * NW + wv) * 64 + lane, hv = g / CQV, slot = g % CQV;
        const int hx = hv % B::HW, hy = (hv / B::HW) % B::HH, hz = hv / (B::HW * B::HH);
        xsrc[u] = hv < B::HV ? (((hz << 8 | hy) << 8 | hx) << 2) | (slot ^ piece_swz<CQ>(hx >> 2)) : -1;
    }
    auto issue = [&](int b, int buf) {
        int bb = b;
        const int x0 = (bb % nbx) * B::TW; bb /= nbx;
        const int y0 = (bb % nby) * B::TH; bb /= nby;
        const int z0 = (bb % nbz) * B::TD;
        const int n = bb / nbz;
        T* Ds = S + buf * BUF;
        T* Xs = Ds + DELEMS;
#pragma unroll
        for (int u = 0; u < DI; ++u) {
            const int s = dsrc[u];
            const int z = z0 + (s >> 18), y = y0 + ((s >> 10) & 255), x = x0 + ((s >> 2) & 255);
            const bool ok = s >= 0 && z < a.D && y < a.H && x < a.W;
            const unsigned off = ok ? (unsigned)((((n * a.D + z) * a.H + y) * a.W + x) * a.P + p0 + (s & 3) * 8) * 2u : DMA_OOB;
            dma16_async(rd, Ds + (u * NW + wv) * 512, off);
        }
#pragma unroll
        for (int u = 0; u < XI; ++u) {
            const int s = xsrc[u];
            const int z = z0 + (s >> 18) - B::PD, y = y0 + ((s >> 10) & 255) - 1, x = x0 + ((s >> 2) & 255) - 1;
            const bool ok = s >= 0 && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            const unsigned off = ok ? (unsigned)((((n * a.D + z) * a.H + y) * a.W + x) * xC + xq0 + (s & 3) * 8) * 2u : DMA_OOB;
            dma16_async(rx, Xs + (u * NW + wv) * 512, off);
        }
    };

    // ---- element offsets of this lane's transposing reads.  Step ks reduces box voxels 32 ks .. 32 ks + 31; the two reads of a
    //      fragment fetch rows r = 16 jj + 4 q + (l15 >> 2), jj = 0, 1 (the same row permutation for both operands).
    //      da[ks][jj][i]: dR tile, p sub-tile i;  xb[ks][jj][kw]: halo, this wave's q sub-tile, tap column kw (tap row/plane = immediate)
    int da[KS][2][PT], xb[KS][2][3];
    const int c4 = (l15 & 3) * 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int v = ks * 32 + 16 * jj + 4 * q + (l15 >> 2);
#pragma unroll
            for (int i = 0; i < PT; ++i) da[ks][jj][i] = v * CP + (CP == 32 ? ((i ^ ((v >> 2) & 1)) << 4) : 0) + c4;
            const int vx = v % B::TW;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                xb[ks][jj][kw] = B::halo_base(v) * CQ + (CQ == 32 ? ((jw ^ (((vx + kw) >> 2) & 1)) << 4) : 0) + c4;
        }

    f32x4 acc[NTG][PT];
#pragma unroll
    for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int i = 0; i < PT; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    int cur = 0;
    if ((int)blockIdx.x < nbox) issue(blockIdx.x, 0);
    for (int b = blockIdx.x; b < nbox; b += gridDim.x) {
        wait_vmem();                                      // this wave's copies of box b have landed ...
        __syncthreads();                                  // ... and everyone's; every wave is done with the other buffer
        if (b + (int)gridDim.x < nbox) issue(b + gridDim.x, cur ^ 1);      // in flight while box b is multiplied
        const T* Ds = S + cur * BUF;
        const T* Xs = Ds + DELEMS;
        // the tap group is wave-uniform: one specialised copy of the loop per group keeps every tap offset an immediate
        switch (grp) {
            case 0: w3x_box<T, B, CP, CQ, 0 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            case 1: w3x_box<T, B, CP, CQ, 1 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            case 2: w3x_box<T, B, CP, CQ, 2 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            case 3: w3x_box<T, B, CP, CQ, 3 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            case 4: if (NG > 4) w3x_box<T, B, CP, CQ, 4 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            case 5: if (NG > 4) w3x_box<T, B, CP, CQ, 5 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            case 6: if (NG > 4) w3x_box<T, B, CP, CQ, 6 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
            default: if (NG > 4) w3x_box<T, B, CP, CQ, 7 * NTG, NTG, NTG>(Ds, Xs, da, xb, acc); break;
        }
        cur ^= 1;
    }
    // partial tile [p][tap][q] of this workgroup (layout of wgrad3_reduce_kernel)
    float* dst = a.partial + ((long long)combo * a.nb + blockIdx.x) * (CP * B::NTAP * CQ);
#pragma unroll
    for (int t = 0; t < NTG; ++t) {
        const int tap = grp * NTG + t;
        if (tap < B::NTAP) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[((i * 16 + 4 * q + r) * B::NTAP + tap) * CQ + jw * 16 + l15] = acc[t][i][r];
        }
    }
}

template <class T, int TD, int TH, int TW, int KD>
void launch_shape(const Wgrad3xArgs& a, int CP, int CQ, hipStream_t s) {
    typedef WBox<TD, TH, TW, KD> B;
    const int combos = (a.P / CP) * (a.Q / CQ);
    dim3 grid(a.nb, combos), block(W3X_WAVES * 64);
#define SEG_W3X(CPv, CQv) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3x_kernel<T, B, CPv, CQv>), grid, block, 0, s, a)
    if (CP == 32 && CQ == 32) SEG_W3X(32, 32);
    else if (CP == 32) SEG_W3X(32, 16);
    else if (CQ == 32) SEG_W3X(16, 32);
    else SEG_W3X(16, 16);
#undef SEG_W3X
}

template <class T>
void dispatch(const Wgrad3xArgs& a, int ndim, int CP, int CQ, bool wide, hipStream_t s) {
    if (ndim == 3) {
        if (wide) launch_shape<T, 3, 4, 16, 3>(a, CP, CQ, s);
        else launch_shape<T, 3, 8, 8, 3>(a, CP, CQ, s);
    } else {
        if (wide) launch_shape<T, 1, 8, 16, 1>(a, CP, CQ, s);
        else launch_shape<T, 1, 8, 8, 1>(a, CP, CQ, s);
    }
}

}  // namespace

bool wgrad3x_supported(int dtype, int N, int D, int H, int W, int P, int Q, int C0, bool has_x1) {
    if (dtype == DT_F32) return false;
    if (P % 16 || Q % 16) return false;
    if (has_x1 && (C0 % 16 || C0 <= 0 || C0 >= Q)) return false;
    const long long vox = (long long)N * D * H * W;
    const int cmax = P > Q ? P : Q;
    if (vox * cmax * 2 >= (1ll << 31)) return false;                  // buffer ranges / 32-bit element offsets
    return true;
}

// the (p-tile, q-tile) widths: 32 wherever the channel counts allow; a q-tile never straddles the two concat sources
void wgrad3x_tiles(int P, int Q, int C0, bool has_x1, int* CP, int* CQ) {
    *CP = P % 32 == 0 ? 32 : 16;
    *CQ = (Q % 32 == 0 && (!has_x1 || C0 % 32 == 0)) ? 32 : 16;
}

bool launch_wgrad3x(const void* dr, const void* x0, const void* x1, int C0, float* partial, int nb, int N, int D, int H, int W, int P, int Q,
                    int ndim, int dtype, bool wide, hipStream_t s) {
    if (!wgrad3x_supported(dtype, N, ndim == 3 ? D : 1, H, W, P, Q, C0, x1 != nullptr)) return false;
    Wgrad3xArgs a;
    a.dr = dr; a.x0 = x0; a.x1 = x1; a.C0 = x1 ? C0 : Q; a.partial = partial;
    a.N = N; a.D = ndim == 3 ? D : 1; a.H = H; a.W = W; a.P = P; a.Q = Q; a.nb = nb;
    int CP, CQ;
    wgrad3x_tiles(P, Q, C0, x1 != nullptr, &CP, &CQ);
    if (dtype == DT_F16) dispatch<f16>(a, ndim, CP, CQ, wide, s);
    else dispatch<bf16>(a, ndim, CP, CQ, wide, s);
    return true;
}

}  // namespace seg
