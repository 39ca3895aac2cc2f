// LDS halo-tile kernels for the 3^d (or 3^2) stride-1 pad-1 convolutions — the LUConv / _block layers
// that carry >90 % of the FLOPs (networks/VNet3d.py:8, networks/Unet3d.py:66-80).
//
//   conv3_kernel  : forward and data-gradient (same kernel, flipped packed weights).  A workgroup
//                   owns a TD x TH x TW box of output voxels; the (TD+2)(TH+2)(TW+2) input halo of one
//                   32- (or 16-) channel chunk is staged ONCE in LDS and reused by all 27 taps, so
//                   HBM/L2 traffic is ~1x the tensor instead of 27x.  MFMA 16x16x32 (f16/bf16) or
//                   8 x 16x16x4 (f32), weights streamed from L2 with one-tap-ahead prefetch, fused
//                   bias + GroupNorm partial sums + coalesced channels-last stores.
//   wgrad3_kernel : weight gradient.  Both operands need the voxel axis as the MFMA K dimension, so
//                   dR and the X halo are staged row-major and read with ds_read_b64_tr_b16.  Each
//                   workgroup walks a strided list of boxes accumulating a [CP x taps x CQ] tile in
//                   registers and writes ONE partial tile; wgrad3_reduce_kernel sums the partials
//                   (deterministic; no fp32 atomics).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "kernels.h"

// waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument); tuning knobs of
// tools/build_variant.py, the defaults are the measured choice
#ifndef SEG_W3_OCC
#define SEG_W3_OCC 2
#endif
#ifndef SEG_STEM_OCC
#define SEG_STEM_OCC 2
#endif

namespace seg {
namespace {

template <int TD, int TH, int TW, int KD> struct Box {
    static constexpr int V = TD * TH * TW;            // output voxels per box
    static constexpr int HD = TD + KD - 1, HH = TH + 2, HW = TW + 2;
    static constexpr int HV = HD * HH * HW;           // halo voxels
    static constexpr int PD = (KD - 1) / 2;           // pad along depth (0 for 2-D)
    static constexpr int NTAP = KD * 9;
    static __device__ __forceinline__ int halo_base(int v) {      // box voxel -> halo index of tap (0,0,0)
        const int vx = v % TW, vy = (v / TW) % TH, vz = v / (TW * TH);
        return (vz * HH + vy) * HW + vx;
    }
    static __device__ __forceinline__ int tap_off(int t) {        // tap -> halo index offset
        const int kw = t % 3, kh = (t / 3) % 3, kd = t / 9;
        return (kd * HH + kh) * HW + kw;
    }
};

struct BoxPos { int n, z0, y0, x0; };
template <class B, int TD, int TH, int TW>
__device__ __forceinline__ BoxPos box_pos(long long b, int D, int H, int W) {
    const int nbx = (W + TW - 1) / TW, nby = (H + TH - 1) / TH, nbz = (D + TD - 1) / TD;
    BoxPos p;
    p.x0 = (int)(b % nbx) * TW; b /= nbx;
    p.y0 = (int)(b % nby) * TH; b /= nby;
    p.z0 = (int)(b % nbz) * TD;
    p.n = (int)(b / nbz);
    return p;
}

// stage the halo of channels [c0, c0+CH) of tensor `in` ([N][D][H][W][C]) into LDS rows of LD elements
// `in` may be a virtual channel concat: channels [0, C0) come from `in` (row length C0), the rest from `in1` (row length C - C0)
// XOR swizzle of the 16-B pieces of a 64-B halo row (16-bit types, 32-channel chunks, unpadded rows).  ds_read_b128 is
// serviced in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS table): with
// the row pitch a multiple of 4 voxels, piece' = piece ^ 2*(row&1) ^ 3*((x>>2)&1) makes the A-fragment reads of both
// tile shapes (16 voxels of one row / 8+8 voxels of two rows) hit all 64 banks exactly once for every tap shift;
// the padded 80-B rows cost 2x (16-wide tiles) to 3x (8x8 tiles) the LDS cycles.
__device__ __forceinline__ int halo_swz(int row, int x) { return ((row & 1) << 1) ^ (((x >> 2) & 1) * 3); }

template <class T, class B, int CH, int LD, int PITCH = B::HW, bool SWZ = false>
__device__ __forceinline__ void stage_halo(T* Xs, const T* in, int C, int c0, const BoxPos& p, int D, int H, int W,
                                           const T* in1 = nullptr, int C0 = 0) {
    if (!in1) C0 = C;
    // All global loads of a batch are issued before the first LDS store: a load->wait->store loop would
    // serialise ~9 HBM round trips per box (measured: the dominant cost of the first version).
    constexpr int CPV = CH / 8, TOTAL = B::HV * CPV, NIT = (TOTAL + 255) / 256;
    constexpr int UN = sizeof(T) == 2 ? NIT : (NIT + 1) / 2;        // f32 fragments are twice as wide: two batches
#pragma unroll
    for (int b0 = 0; b0 < NIT; b0 += UN) {
        vec<T, 8> v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = (b0 + u) * 256 + threadIdx.x;
            const int hv = i / CPV, c8 = i % CPV;
            const int hx = hv % B::HW, hy = (hv / B::HW) % B::HH, hz = hv / (B::HW * B::HH);
            const int z = p.z0 + hz - B::PD, y = p.y0 + hy - 1, x = p.x0 + hx - 1;
            v[u] = zero8<T>();
            if (b0 + u < NIT && i < TOTAL && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                const long long vox = (((long long)p.n * D + z) * H + y) * W + x;
                const int ch = c0 + c8 * 8;
                v[u] = ch < C0 ? load8(in + vox * C0 + ch) : load8(in1 + vox * (C - C0) + (ch - C0));
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int i = (b0 + u) * 256 + threadIdx.x;
            if (b0 + u < NIT && i < TOTAL) {
                if (SWZ) {
                    const int hv = i / CPV, c8 = i % CPV;
                    const int hx = hv % B::HW, row = hv / B::HW;
                    store8(&Xs[((row * PITCH + hx) * CPV + (c8 ^ halo_swz(row, hx))) * 8], v[u]);
                } else {
                    store8(&Xs[(i / CPV) * LD + (i % CPV) * 8], v[u]);
                }
            }
        }
    }
}

// shared epilogue of the box kernels: bias -> LDS tile [voxel][co] (aliases the halo buffer) -> coalesced
// channels-last stores + per-channel sum / sum-of-squares into this workgroup's statistics replica
template <class T, class B, int TW, int TH, int MT, int NT>
__device__ __forceinline__ void box_epilogue(f32x4 (&acc)[MT][NT], T* Os, float* red, const float* bias, T* out, double* stats,
                                             const BoxPos& bp, int co0, int N, int D, int H, int W, int Cout) {
    constexpr int BN = NT * 16, OLD = BN + 8;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = j * 16 + l15;
        const float bsv = bias ? bias[co0 + col] : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[((wv * MT + m) * 16 + q * 4 + r) * OLD + col] = from_f<T>(acc[m][j][r] + bsv);
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
    for (int i = tid; i < B::V * CPR; i += 256) {
        const int v = i / CPR, c8 = i % CPR;
        const int x = bp.x0 + v % TW, y = bp.y0 + (v / TW) % TH, z = bp.z0 + v / (TW * TH);
        if (x < W && y < H && z < D)
            store8(out + ((((long long)bp.n * D + z) * H + y) * W + x) * Cout + co0 + c8 * 8, load8(&Os[v * OLD + c8 * 8]));
    }
    if (stats) {
        constexpr int G = 256 / BN;
        const int col = tid % BN, g = tid / BN;
        float s = 0.f, ss = 0.f;
        for (int v = g; v < B::V; v += G) {
            const int x = bp.x0 + v % TW, y = bp.y0 + (v / TW) % TH, z = bp.z0 + v / (TW * TH);
            if (x < W && y < H && z < D) {
                const float f = to_f(Os[v * OLD + col]);
                s += f; ss += f * f;
            }
        }
        red[(g * BN + col) * 2] = s;
        red[(g * BN + col) * 2 + 1] = ss;
        __syncthreads();
        if (tid < BN) {
            double ts = 0.0, tss = 0.0;
            for (int k = 0; k < G; ++k) { ts += red[(k * BN + col) * 2]; tss += red[(k * BN + col) * 2 + 1]; }
            double* dst = stats + ((long long)(blockIdx.x % STAT_REP) * N * Cout + (long long)bp.n * Cout + co0 + col) * 2;
            atomicAdd(dst, ts);
            atomicAdd(dst + 1, tss);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient
// ------------------------------------------------------------------------------------------------
struct Conv3Args {
    const void* in1; int C0;     // optional second concat source (channels C0..Cin-1)
    const void* in; const void* w; const float* bias; void* out; double* stats;
    int N, D, H, W, Cin, Cout, Kpad;
    int dbg;   // ablation mask (SEG_CONV3_DBG): 1 no halo loads, 2 no weight loads, 4 no MFMA loop, 8 no epilogue
    unsigned long long* trace;   // SEG_CONV3_TRACE: 6 wall-clock stamps (10 ns ticks) per workgroup, else null
};
#define SEG_STAMP(k) do { if (a.trace && threadIdx.x == 0) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 6 + (k)] = wall_clock64(); } while (0)

// Waves per SIMD the register allocator leaves room for, per tiling (measured, profiles/r01_conv3_occupancy_step24.log):
// the 16-channel NT=1 tiling fits four workgroups per CU in LDS (40.5 KB each) once it is held to 128 VGPRs (151 -> 135 us at
// 96^3); the 8x8-box 32-channel NT=1 tiling gains from 184 instead of 198 VGPRs (24^3: 36.5 -> 31 us); the NT=2 tilings lose
// when squeezed (48^3: 51 -> 77 us) and stay at 2.  SEG_C3_OCC overrides all of them (tools/build_variant.py).
template <int CH, int NT, int TH>
constexpr int c3_occ() {
#ifdef SEG_C3_OCC
    return SEG_C3_OCC;
#else
    return (CH == 16 && NT == 1) ? 4 : (CH == 32 && NT == 1 && TH == 8) ? 3 : 2;
#endif
}

// WL = true (NT == 1 tiles): the [16][taps x CH] weight slab of the current channel chunk is staged in LDS with the
// halo, so the tap loop never waits on L2 (the small deep levels run ~1.5 workgroups per CU and were bound by that
// latency); WL = false: weights stream from L2 through a register ring.
template <class T, int TD, int TH, int TW, int KD, int CH, int NT, bool WL>
__global__ __launch_bounds__(256, (c3_occ<CH, NT, TH>())) void conv3_kernel(Conv3Args a) {
    typedef Box<TD, TH, TW, KD> B;
    constexpr bool SWZ = sizeof(T) == 2 && CH == 32;    // swizzled unpadded 64-B rows (see halo_swz); else padded rows
    constexpr int XLD = SWZ ? CH : CH + 8;              // padded: 48 B (CH=16) / 160 B (f32) rows
    constexpr int HWP = SWZ ? (B::HW + 3) / 4 * 4 : B::HW;   // halo row pitch in voxels
    constexpr int MT = B::V / 64;                       // 16-voxel M tiles per wave
    constexpr int BN = NT * 16, OLD = BN + 8;
    constexpr int XS_ELEMS = B::HD * B::HH * HWP * XLD, OS_ELEMS = B::V * OLD;
    // epilogue scratch (output tile + 2 KB of reduction slots) aliases the halo buffer: 38.4 KB per workgroup for the
    // 16-bit 3x4x16 box -> four workgroups per CU instead of three
    constexpr int RED_ELEMS = 2048 / sizeof(T);
    __shared__ __attribute__((aligned(16))) T Xs[XS_ELEMS > OS_ELEMS + RED_ELEMS ? XS_ELEMS : OS_ELEMS + RED_ELEMS];
    float* red = (float*)(Xs + OS_ELEMS);
    static_assert(B::V % 64 == 0, "box must hold a multiple of 64 voxels");
    constexpr int NSTEP_ = CH == 32 ? B::NTAP : (B::NTAP + 1) / 2;
    constexpr bool WSWZ = sizeof(T) == 2;                 // weight slab as [step][16 co][32 k], pieces swizzled by 3*((co>>2)&1)
    constexpr int WLD = NSTEP_ * 32 + 8;                  // f32: [co][step x 32 + pad] rows
    __shared__ T Ws[WL ? 16 * WLD : 8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    SEG_STAMP(0);
    const BoxPos bp = box_pos<B, TD, TH, TW>(blockIdx.x, a.D, a.H, a.W);
    const int co0 = blockIdx.y * BN;
    const T* in = (const T*)a.in;
    const T* wp = (const T*)a.w;

    int hb[MT];                                          // halo base index of this lane's voxel per M tile
    int pq[MT][3];                                       // swizzled element offset of this lane's piece per kw shift
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int v = (wv * MT + m) * 16 + l15;
        const int vx = v % TW, vy = (v / TW) % TH, vz = v / (TW * TH);
        hb[m] = (vz * B::HH + vy) * HWP + vx;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) pq[m][kw] = (q ^ halo_swz(vz * B::HH + vy, vx + kw)) * 8;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* wrow[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wrow[j] = wp + (long long)(co0 + j * 16 + l15) * a.Kpad + q * 8;

    // K order of the packed weights is (tap, ci).  CH == 32: step s = (tap s) x (32 channels of chunk cc).
    // CH == 16 (Cin == 16): step s covers taps 2s, 2s+1; lanes q = 0,1 -> tap 2s, q = 2,3 -> tap 2s+1.
    constexpr int NSTEP = CH == 32 ? B::NTAP : (B::NTAP + 1) / 2;
    const int nchunk = a.Cin / CH;
    for (int cc = 0; cc < nchunk; ++cc) {
        if (cc) __syncthreads();
        if (!(a.dbg & 1)) stage_halo<T, B, CH, XLD, HWP, SWZ>(Xs, in, a.Cin, cc * CH, bp, a.D, a.H, a.W, (const T*)a.in1, a.C0);
        // weight slab of ONE 16-channel output tile and this channel chunk: global -> registers -> LDS
        constexpr int PIECES = 16 * NSTEP_ * 4, WN = (PIECES + 255) / 256;
        vec<T, 8> wv_[WN];
        auto wload = [&](int j) {
#pragma unroll
            for (int u = 0; u < WN; ++u) {
                const int i = u * 256 + tid;
                const int c4 = i & 3, st_ = (i >> 2) % NSTEP_, co = i / (4 * NSTEP_);
                wv_[u] = zero8<T>();
                if (i < PIECES)
                    wv_[u] = load8(wp + (long long)(co0 + j * 16 + co) * a.Kpad + (CH == 32 ? st_ * a.Cin + cc * 32 : st_ * 32) + c4 * 8);
            }
        };
        auto wstore = [&]() {
#pragma unroll
            for (int u = 0; u < WN; ++u) {
                const int i = u * 256 + tid;
                const int c4 = i & 3, st_ = (i >> 2) % NSTEP_, co = i / (4 * NSTEP_);
                if (i < PIECES) {
                    if (WSWZ) store8(&Ws[(st_ * 16 + co) * 32 + ((c4 ^ (((co >> 2) & 1) * 3)) * 8)], wv_[u]);
                    else store8(&Ws[co * WLD + st_ * 32 + c4 * 8], wv_[u]);
                }
            }
        };
        if (WL) { wload(0); wstore(); }
        if (cc == 0) SEG_STAMP(1);
        __syncthreads();
        if (cc == 0) SEG_STAMP(2);
        if (a.dbg & 4) continue;
        // A-fragment of M tile m for reduction step s (compile-time s)
        auto afrag = [&](int s, int m) -> typename Mma<T>::frag {
            if (CH == 32) {
                const int skw = s % 3, srow = (s / 9) * B::HH + (s / 3) % 3;      // step = tap (kd, kh, kw)
                const int toff = srow * HWP + skw;
                return SWZ ? load8(&Xs[(hb[m] + toff) * XLD + (pq[m][skw] ^ ((srow & 1) << 4))]) : load8(&Xs[(hb[m] + toff) * XLD + q * 8]);
            }
            // CH == 16: lanes q = 0,1 take tap 2s, q = 2,3 tap 2s+1: both offsets are compile-time constants
            const int t1 = 2 * s + 1;
            const int off0 = B::tap_off(2 * s), off1 = B::tap_off(t1 < B::NTAP ? t1 : 0);
            const bool tvalid = (q < 2) || t1 < B::NTAP;
            typename Mma<T>::frag af = load8(&Xs[(hb[m] + ((q < 2) ? off0 : off1)) * XLD + (q & 1) * 8]);
            if (!tvalid) af = zero8<T>();
            return af;
        };
        if (WL) {
            // output tiles one after the other over the resident halo: every weight byte crosses L2 -> CU once per
            // workgroup (streaming them per wave made the 48^3 level L2-bound: 4 waves x 27 taps x 2 KB per box)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j + 1 < NT) wload(j + 1);
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    const typename Mma<T>::frag bf = WSWZ ? load8(&Ws[(s * 16 + l15) * 32 + ((q ^ (((l15 >> 2) & 1) * 3)) * 8)])
                                                          : load8(&Ws[l15 * WLD + s * 32 + q * 8]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][j] = Mma<T>::run(afrag(s, m), bf, acc[m][j]);
                }
                if (j + 1 < NT) { __syncthreads(); wstore(); __syncthreads(); }
            }
        } else {
            // weights straight from L2 through a PF-deep register ring
            constexpr int PF = NT == 2 ? 6 : 3;
            typename Mma<T>::frag bq[PF + 1][NT];
            auto wofs = [&](int s) { return CH == 32 ? s * a.Cin + cc * 32 : s * 32; };
#pragma unroll
            for (int s = 0; s < PF && s < NSTEP; ++s)
#pragma unroll
                for (int j = 0; j < NT; ++j) bq[s][j] = load8(wrow[j] + ((a.dbg & 2) ? 0 : wofs(s)));
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s + PF < NSTEP && !(a.dbg & 2)) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) bq[(s + PF) % (PF + 1)][j] = load8(wrow[j] + wofs(s + PF));
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const typename Mma<T>::frag af = afrag(s, m);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[m][j] = Mma<T>::run(af, bq[s % (PF + 1)][j], acc[m][j]);
                }
            }
        }
    }
    SEG_STAMP(3);
    __syncthreads();
    SEG_STAMP(4);
    if (a.dbg & 8) { if (acc[0][0][0] == 123.f) ((T*)a.out)[0] = from_f<T>(1.f); return; }
    box_epilogue<T, B, TW, TH, MT, NT>(acc, Xs, red, a.bias, (T*)a.out, a.stats, bp, co0, a.N, a.D, a.H, a.W, a.Cout);
    SEG_STAMP(5);
}

template <int TD, int TH, int TW>
inline long long num_boxes(int N, int D, int H, int W) {
    return (long long)N * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

template <class T, int TD, int TH, int TW, int KD>
void conv3_launch_shape(const Conv3Args& a, hipStream_t s) {
    int nt = (a.Cout % 64 == 0) ? 4 : (a.Cout % 32 == 0) ? 2 : 1;
    if (a.Cin == 16 && nt == 4) nt = 2;
    const long long nbox = num_boxes<TD, TH, TW>(a.N, a.D, a.H, a.W);
    // measured on MI355X: on the small levels more, narrower workgroups (several resident per CU) beat NT = 4
    // tiles (conv3 class 2.3 ms vs 3.4 ms per step) - the per-workgroup tap loop is latency-bound, so occupancy wins
    while (nt > 1 && nbox * (a.Cout / (16 * nt)) < 1024) nt /= 2;
    dim3 grid((unsigned)nbox, a.Cout / (16 * nt));
    // LDS weight slab (one 16-channel output tile at a time over the resident halo) for every 16-bit tiling; the f32
    // halo is too large to share the LDS with it
    const bool wl = nt == 1 || sizeof(T) == 2;
#define SEG_C3(CH, NT, WLV) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3_kernel<T, TD, TH, TW, KD, CH, NT, WLV>), grid, dim3(256), 0, s, a)
    if (a.Cin == 16) { if (nt == 1) SEG_C3(16, 1, true); else if (wl) SEG_C3(16, 2, true); else SEG_C3(16, 2, false); }
    else if (nt == 1) SEG_C3(32, 1, true);
    else if (nt == 2) { if (wl) SEG_C3(32, 2, true); else SEG_C3(32, 2, false); }
    else { if (wl) SEG_C3(32, 4, true); else SEG_C3(32, 4, false); }
#undef SEG_C3
}

// box shape per level: x extent 16 when the row length suits it, else 8x8 tiles (24^3, 6^3 levels)
inline bool wide_box(int W) { return W % 16 == 0 || W == 12; }

template <class T>
void conv3_dispatch(const Conv3Args& a, int ndim, hipStream_t s) {
    if (ndim == 3) {
        if (wide_box(a.W)) conv3_launch_shape<T, 3, 4, 16, 3>(a, s);
        else conv3_launch_shape<T, 3, 8, 8, 3>(a, s);
    } else {
        if (wide_box(a.W)) conv3_launch_shape<T, 1, 8, 16, 1>(a, s);
        else conv3_launch_shape<T, 1, 8, 8, 1>(a, s);
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
// (p-tile, q-tile) widths of the halo weight gradient.  Round 4: the q-tile is 16 channels wherever the tensors are 16-bit - the 32 x 32 tile keeps
// 112 accumulator registers per lane next to 48 prefetch registers, which left no room to pipeline the LDS reads (every variant of the pipelined
// sweep spilled: the compiler's reloads sat between the prefetch loads and serialised them); 32 x 16 needs 56, reads its B fragments a whole K
// step ahead, writes partial tiles half the size (the reduce pass re-reads half as much) and its 44 KB of LDS / 184 registers leave a halo-conv
// workgroup of the main stream room on the same CU.  SEG_W3_CQ=32 restores the wide tile.
// The 2-D nets keep the wide tile (9 taps: 36 accumulator registers per 16 x 16 sub-tile set, the pressure argument does not apply, and
// C2 VNet2d 16 x 512^2 measures 6.45 ms per step with it against 6.66 with the narrow one; C4 / C5, 3-D: 4.43 / 4.45 against 4.56 / 4.68,
// profiles/r04_configs_tile_ab.log).
inline void wgrad3_tile(int P, int Q, int esz, int ndim, int* CP, int* CQ) {
    static const int cq = 0;          // 0: by dimensionality
    const int want = cq ? cq : (ndim == 3 ? 16 : 32);
    *CP = P >= 32 ? 32 : 16;
    *CQ = (Q >= 32 && (want >= 32 || esz == 4)) ? 32 : 16;
}

struct Wgrad3Args {
    const void* x1; int C0;      // optional second concat source of x
    const void* dr; const void* x; float* partial;
    int N, D, H, W, P, Q;       // channel counts of dr / x
    int nb;                     // workgroups per (p-tile, q-tile) combo
#ifdef SEG_W3_TRACE
    unsigned long long* trace;            // diagnostic build (tools/build_variant.py ... -DSEG_W3_TRACE): 8 phase sums per workgroup
#endif
};
#ifdef SEG_W3_TRACE
#define SEG_W3T(k) do { const unsigned long long t_ = wall_clock64(); ph[k] += t_ - tl; tl = t_; } while (0)
#else
#define SEG_W3T(k)
#endif

template <class T, int C> struct WLd { static constexpr int v = C == 32 ? 48 : 16; };   // 96 B / 32 B rows: conflict-free tr reads
template <int C> struct WLd<float, C> { static constexpr int v = C + 4; };

template <class T, int LD> struct TrFrag {
    // rows (k slots) r0..r0+3 per 16-lane group via two transposing reads; row addresses are arbitrary
    static __device__ __forceinline__ typename Mma<T>::frag load(const T* base, const int* rowoff, int col0, int lane) {
        const int t = lane & 15;
        const s16x4 lo = lds_read_tr16(base + rowoff[0] + col0 + (t & 3) * 4);
        const s16x4 hi = lds_read_tr16(base + rowoff[1] + col0 + (t & 3) * 4);
        vec<short, 8> v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo[j]; v[4 + j] = hi[j]; }
        return __builtin_bit_cast(typename Mma<T>::frag, v);
    }
};
template <int LD> struct TrFrag<float, LD> {
    static __device__ __forceinline__ Mma<float>::frag load(const float* base, const int* rowoff, int col0, int lane) {
        Mma<float>::frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = base[rowoff[j] + col0 + (lane & 15)];
        return f;
    }
};

// two waves per SIMD as the register budget whatever the LDS footprint says (above 80 KB of LDS the compiler would otherwise plan for ONE
// resident workgroup and take 290+ registers: nothing of the main stream's halo convs - 256 registers per wave - fits beside that on a SIMD)
#ifndef SEG_EMU
#define SEG_W3_WAVES __attribute__((amdgpu_waves_per_eu(2)))
#else
#define SEG_W3_WAVES
#endif
template <class T, int TD, int TH, int TW, int KD, int CP, int CQ>
__global__ __launch_bounds__(256, SEG_W3_OCC) SEG_W3_WAVES void wgrad3_kernel(Wgrad3Args a) {
    typedef Box<TD, TH, TW, KD> B;
    constexpr int DLD = WLd<T, CP>::v, XLD = WLd<T, CQ>::v;
    constexpr int PT = CP / 16, QT = CQ / 16;
    constexpr int NTW = (B::NTAP + 3) / 4;               // taps per wave
    constexpr int NR = sizeof(T) == 4 ? 8 : 2;           // row slots a lane addresses per K step
    __shared__ T Ds[B::V * DLD];
    __shared__ T Xs[B::HV * XLD];
    static_assert(B::V % 32 == 0, "box must hold a multiple of 32 voxels");

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int nqt = a.Q / CQ;
    const int combo = blockIdx.y, p0 = (combo / nqt) * CP, q0 = (combo % nqt) * CQ;
    const T* dr = (const T*)a.dr;
    const T* x = (const T*)a.x;
    const long long nbox = (long long)a.N * ((a.D + TD - 1) / TD) * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);

    f32x4 acc[NTW][PT][QT];
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
        for (int i = 0; i < PT; ++i)
#pragma unroll
            for (int j = 0; j < QT; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- 16-bit tensors: software-pipelined sweep.  Reading the ISA of the first version (round 4, /tmp listing of wgrad3_kernel<f16,...,16,16>):
    // the per-tap `if (tap < NTAP)` was a divergent branch (the wave index lived in a VGPR), and inside it the compiler emitted
    // ds_read x2 -> s_waitcnt lgkmcnt(0) -> v_mfma: one exposed LDS round trip (~100+ clk) per tap with ONE wave per SIMD and nothing to cover
    // it - 7 taps x 6 K steps x ~130 clk = 2.3 us per 192-voxel box, which is what the kernel measured (3 us per box, 220 us at 4 x 96^3; a
    // 2.7x larger box changed nothing: 219 vs 221 us standalone, profiles/r04_session_a.log).  Now: the wave index is a scalar
    // (readfirstlane), every wave runs NTW taps branch-free (the 28th "tap" of the fourth wave re-reads tap 26 into an accumulator that is
    // never stored), and the B fragments travel W items ahead of the MFMAs that use them (W = all taps of a K step for the narrow tiles,
    // 2 for 32 x 32 where one item is 4 MFMAs = 64 clk), the A fragments one K step ahead.
    const int wvs = __builtin_amdgcn_readfirstlane(wv);
    int toff[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tap = wvs + 4 * t;
        toff[t] = B::tap_off(tap < B::NTAP ? tap : B::NTAP - 1) * XLD;
    }
    // `hook(m)` runs in front of item m = (K step, tap): the prefetch path issues the global loads of the NEXT box there, one piece every few
    // items.  Round 4 phase trace (tools/trace_wgrad3.py, profiles/r04_wgrad3_phase_trace.log): with all loads of a box issued in one burst in
    // front of the sweep, a workgroup spent 111 - 157 us of a 220 us launch (16 channels, 4 x 96^3) blocked in that burst - the CU's address /
    // miss queues fill after a few KB and the wave stalls in order until data returns - and 70 us in the sweep, the two strictly one after the
    // other: the memory pipe idled during every sweep and the matrix cores during every burst.
    auto sweep16 = [&](auto&& hook) {
        if constexpr (sizeof(T) == 2) {
            typedef typename Mma<T>::frag Frag;
            constexpr int KS = B::V / 32;
            constexpr int W = (PT * QT >= 4) ? 2 : NTW;            // B-fragment look-ahead, in (K step, tap) items
            static_assert(W <= NTW, "pipeline shape");
            Frag afk[2][PT], bfw[W][QT];
            int dc[2], xc[2];
            auto krows = [&](int ks, int (&d)[2], int (&x)[2]) {
                int lr = 4 * q + (l15 >> 2);
                settle(lr);                                        // opaque: KS x 4 row offsets hoisted out of the box loop cost more registers than ~10 VALU per K step
#pragma unroll
                for (int j = 0; j < 2; ++j) {                      // slot j covers rows 16j + 4q + (l15 >> 2) of the 32-voxel K step
                    const int v = ks * 32 + 16 * j + lr;
                    d[j] = v * DLD;
                    x[j] = B::halo_base(v) * XLD;
                }
            };
            auto ldA = [&](const int (&d)[2], Frag (&o)[PT]) {
#pragma unroll
                for (int i = 0; i < PT; ++i) o[i] = TrFrag<T, DLD>::load(Ds, d, i * 16, lane);
            };
            auto ldB = [&](int t, const int (&x)[2], Frag (&o)[QT]) {
#pragma unroll
                for (int j = 0; j < QT; ++j) o[j] = TrFrag<T, XLD>::load(Xs + toff[t], x, j * 16, lane);
            };
            krows(0, dc, xc);
            ldA(dc, afk[0]);
#pragma unroll
            for (int w = 0; w < W; ++w) ldB(w, xc, bfw[w]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {                      // fully unrolled: every register index below is a constant
                int dn[2], xn[2];
                krows(ks + 1 < KS ? ks + 1 : 0, dn, xn);
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const int m = ks * NTW + t;
                    if (t == 0 && ks + 1 < KS) ldA(dn, afk[(ks + 1) & 1]);      // the A fragments of the following K step start travelling
                    hook(m);
#pragma unroll
                    for (int j = 0; j < QT; ++j)
#pragma unroll
                        for (int i = 0; i < PT; ++i) acc[t][i][j] = Mma<T>::run(afk[ks & 1][i], bfw[m % W][j], acc[t][i][j]);
                    const int mn = m + W;                          // the item that takes over the slot
                    if (mn < KS * NTW) {
                        if (mn / NTW == ks) ldB(mn % NTW, xc, bfw[m % W]);
                        else ldB(mn % NTW, xn, bfw[m % W]);
                    }
                }
                xc[0] = xn[0]; xc[1] = xn[1];
            }
        }
    };
    // one K sweep over the box whose tiles sit in LDS (f32 tensors)
    auto sweep = [&]() {
        if constexpr (sizeof(T) == 2) { sweep16([](int) {}); return; }
#pragma unroll 1
        for (int ks = 0; ks < B::V / 32; ++ks) {
            int drow[NR], xrow[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                // 16-bit: slot j covers rows 16j + 4q + (l15>>2); f32: MFMA j consumes row 4j + q
                const int r = sizeof(T) == 4 ? (4 * j + q) : (16 * j + 4 * q + (l15 >> 2));
                const int v = ks * 32 + r;
                drow[j] = v * DLD;
                xrow[j] = B::halo_base(v) * XLD;
            }
            typename Mma<T>::frag af[PT];
#pragma unroll
            for (int i = 0; i < PT; ++i) af[i] = TrFrag<T, DLD>::load(Ds, drow, i * 16, lane);
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int tap = wv + 4 * t;
                if (tap < B::NTAP) {
                    const int toff = B::tap_off(tap) * XLD;
#pragma unroll
                    for (int j = 0; j < QT; ++j) {
                        const typename Mma<T>::frag bf = TrFrag<T, XLD>::load(Xs + toff, xrow, j * 16, lane);
#pragma unroll
                        for (int i = 0; i < PT; ++i) acc[t][i][j] = Mma<T>::run(af[i], bf, acc[t][i][j]);
                    }
                }
            }
        }
    };
    constexpr int CPV = CP / 8, DN = (B::V * CPV + 255) / 256;
    constexpr int CQV = CQ / 8, XN = (B::HV * CQV + 255) / 256;
    // 16-bit tensors whose x is a stored (activated) tensor: the NEXT box's dR tile and x halo are loaded into registers right after the
    // barrier that publishes the current box and land while its MFMAs run (one HBM round trip per box, hidden); the plain loop below
    // spent two exposed round trips per box (dR tile -> wait -> LDS, then the halo -> wait -> LDS) - the kernel is latency-bound, not
    // bandwidth-bound, so that was most of its time.  Costs (DN + XN) x 4 VGPRs (48 at 32 x 32 channels).
    const bool prefetch = sizeof(T) == 2;
    if (prefetch) {
        const T* x1 = (const T*)a.x1;
        const int C0 = x1 ? a.C0 : a.Q;
        vec<T, 8> dv[DN], xv[XN];
        unsigned dok = 0u, xok = 0u;                      // bit u: piece u holds data (else zeros go to LDS)
        // Branch-free: every piece is loaded from a valid address (voxel 0 where the piece lies outside the volume or past the tile) and zeroed on
        // its way into LDS.  The first version wrapped each load in `if (inside)`: hipcc put an s_waitcnt vmcnt(0) at the end of every such block
        // (the destination is live across the join), so the 7 - 12 "prefetch" loads of a box were 7 - 12 serial HBM round trips.
        // The box-independent part of a piece's address is ONE packed register (z, y, x inside the box / halo; -1 past the tile), made opaque at
        // its use: left alone the compiler hoists the unpacked coordinates out of the box loop and pays for them in registers.
        constexpr int NP = DN + XN, NI = (B::V / 32) * NTW;
        static_assert(NP <= NI, "one piece per item at most");
        const int c8d = tid % CPV, c8x = tid % CQV;        // 256 % CPV == 0: the channel piece of a thread is the same for every u
        int pkd[NP];
#pragma unroll
        for (int u = 0; u < DN; ++u) {
            const int v = (u * 256 + tid) / CPV;
            pkd[u] = v < B::V ? ((v / (TW * TH)) << 16) | (((v / TW) % TH) << 8) | (v % TW) : -1;
        }
#pragma unroll
        for (int u = 0; u < XN; ++u) {
            const int hv = (u * 256 + tid) / CQV;
            pkd[DN + u] = hv < B::HV ? ((hv / (B::HW * B::HH)) << 16) | (((hv / B::HW) % B::HH) << 8) | (hv % B::HW) : -1;
        }
        const int chx = q0 + c8x * 8;
        const T* xsrc0 = chx < C0 ? x + chx : x1 + (chx - C0);       // concat: a 16-byte piece never straddles the two sources
        const int xC = chx < C0 ? C0 : a.Q - C0;
        const T* dsrc0 = dr + p0 + c8d * 8;
        BoxPos bpn;
        auto issue_piece = [&](int pc) {                  // pc is a constant wherever this is called (unrolled)
            int s_ = pkd[pc];
            settle(s_);
            if (pc < DN) {
                const int zz = bpn.z0 + (s_ >> 16), yy = bpn.y0 + ((s_ >> 8) & 255), xx = bpn.x0 + (s_ & 255);
                const bool ok = s_ >= 0 && xx < a.W && yy < a.H && zz < a.D;
                const long long vox = ok ? (long long)((bpn.n * a.D + zz) * a.H + yy) * a.W + xx : 0;       // (n, z, y) rows fit an int
                dok = ok ? (dok | (1u << pc)) : (dok & ~(1u << pc));
                dv[pc < DN ? pc : 0] = load8(dsrc0 + vox * a.P);
            } else {
                const int u = pc - DN;
                const int z = bpn.z0 + (s_ >> 16) - B::PD, y = bpn.y0 + ((s_ >> 8) & 255) - 1, xx = bpn.x0 + (s_ & 255) - 1;
                const bool ok = s_ >= 0 && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
                const long long vox = ok ? (long long)((bpn.n * a.D + z) * a.H + y) * a.W + xx : 0;
                xok = ok ? (xok | (1u << u)) : (xok & ~(1u << u));
                xv[u >= 0 ? u : 0] = load8(xsrc0 + vox * xC);
            }
        };
        long long b = blockIdx.x;
#ifdef SEG_W3_TRACE
        unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = wall_clock64();
#endif
        if (b < nbox) {
            bpn = box_pos<B, TD, TH, TW>(b, a.D, a.H, a.W);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) issue_piece(pc);
        }
        SEG_W3T(5);
        for (; b < nbox; b += gridDim.x) {
            __syncthreads();                              // previous box fully consumed
            SEG_W3T(0);
#pragma unroll
            for (int u = 0; u < DN; ++u) {
                const int i = u * 256 + tid;
                if (i < B::V * CPV) store8(&Ds[(i / CPV) * DLD + (i % CPV) * 8], ((dok >> u) & 1u) ? dv[u] : zero8<T>());
            }
#pragma unroll
            for (int u = 0; u < XN; ++u) {
                const int i = u * 256 + tid;
                if (i < B::HV * CQV) store8(&Xs[(i / CQV) * XLD + (i % CQV) * 8], ((xok >> u) & 1u) ? xv[u] : zero8<T>());
            }
            SEG_W3T(1);
            __syncthreads();
            SEG_W3T(2);
            // the next box (the last trip re-reads its own box: harmless, and the sweep stays branch-free)
            bpn = box_pos<B, TD, TH, TW>(b + gridDim.x < nbox ? b + gridDim.x : b, a.D, a.H, a.W);
            sweep16([&](int m) {
                const int p_lo = (m * NP + NI - 1) / NI, p_hi = ((m + 1) * NP + NI - 1) / NI;       // pieces whose slot floor(p * NI / NP) is this item
                if (p_hi > p_lo) issue_piece(p_lo);
            });
            SEG_W3T(4);
        }
#ifdef SEG_W3_TRACE
        if (a.trace && tid == 0)
            for (int k = 0; k < 8; ++k) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = ph[k];
#endif
    } else {
        for (long long b = blockIdx.x; b < nbox; b += gridDim.x) {
            const BoxPos bp = box_pos<B, TD, TH, TW>(b, a.D, a.H, a.W);
            __syncthreads();                                  // previous box fully consumed
            // dR tile [voxel][CP] (zero rows outside the volume)
            vec<T, 8> dv[DN];
#pragma unroll
            for (int u = 0; u < DN; ++u) {
                const int i = u * 256 + tid;
                const int v = i / CPV, c8 = i % CPV;
                const int xx = bp.x0 + v % TW, yy = bp.y0 + (v / TW) % TH, zz = bp.z0 + v / (TW * TH);
                dv[u] = zero8<T>();
                if (i < B::V * CPV && xx < a.W && yy < a.H && zz < a.D)
                    dv[u] = load8(dr + ((((long long)bp.n * a.D + zz) * a.H + yy) * a.W + xx) * a.P + p0 + c8 * 8);
            }
#pragma unroll
            for (int u = 0; u < DN; ++u) {
                const int i = u * 256 + tid;
                if (i < B::V * CPV) store8(&Ds[(i / CPV) * DLD + (i % CPV) * 8], dv[u]);
            }
            stage_halo<T, B, CQ, XLD>(Xs, x, a.Q, q0, bp, a.D, a.H, a.W, (const T*)a.x1, a.C0);
            __syncthreads();
            sweep();
        }
    }
    // partial tile [p][tap][q] of this workgroup
    float* dst = a.partial + ((long long)combo * a.nb + blockIdx.x) * (CP * B::NTAP * CQ);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tap = wv + 4 * t;
        if (tap < B::NTAP) {
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j < QT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[((i * 16 + 4 * q + r) * B::NTAP + tap) * CQ + j * 16 + l15] = acc[t][i][j][r];
        }
    }
}

// dw[p][q][tap] += sum_b partial[combo][b][p'][tap][q']   (one thread per dw element: coalesced read-modify-write of
// the master gradient, partial tiles gathered through L1/L2).  One launch reduces up to W3_BATCH layers (blockIdx.z): the engine runs the reduce of
// every layer of a level visit in ONE launch behind the last weight-gradient kernel of the visit (round 6: 20 latency-bound reduce launches per VNet3d
// step - 7-8 us each for a few MB, 256 us in all - became 8).  The order of the sum per element is unchanged: it does not depend on the batching.
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(Wgrad3ReduceBatch bt) {
    const Wgrad3Reduce& r = bt.r[blockIdx.z];
    const float* partial = r.partial;
    float* dw = r.dw;
    const int P = r.P, Q = r.Q, CP = r.CP, CQ = r.CQ, ntap = r.ntap, nb = r.nb, qreal = r.qreal;
    const long long sP = r.sP, sQ = r.sQ;
    // grid.y slices the partial list (32 partial tiles per slice); slices meet in dw through one atomic each.  The grid is sized for the largest
    // layer of the batch: the slices / blocks a smaller layer does not need leave at once
    const int nslice = (nb + 31) / 32;
    if ((int)blockIdx.y >= nslice) return;
    const long long total = (long long)P * Q * ntap;
    const int tile = CP * ntap * CQ, nqt = Q / CQ;
    const int b0 = blockIdx.y * 32, b1 = (b0 + 32 < nb) ? b0 + 32 : nb;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        // i enumerates the partial layout [combo][p'][tap][q']: consecutive threads read consecutive floats of every
        // partial tile (the 32..512x larger side of the traffic); the single write per element is the strided one
        const int combo = (int)(i / tile), e = (int)(i % tile);
        const int qq = e % CQ, tap = (e / CQ) % ntap, pp = e / (CQ * ntap);
        const float* src = partial + (long long)combo * nb * tile + e;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = b0;
        for (; b + 16 <= b1; b += 16) {          // 16 independent loads per round trip (the loop is pure L2 latency)
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = src[(long long)(b + u) * tile];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
        }
        for (; b + 4 <= b1; b += 4) {
            s0 += src[(long long)b * tile]; s1 += src[(long long)(b + 1) * tile];
            s2 += src[(long long)(b + 2) * tile]; s3 += src[(long long)(b + 3) * tile];
        }
        for (; b < b1; ++b) s0 += src[(long long)b * tile];
        const float tot = (s0 + s1) + (s2 + s3);
        const int p = (combo / nqt) * CP + pp, qc = (combo % nqt) * CQ + qq;
        if (qc >= qreal) continue;                      // zero-padded input channels (multi-channel image tensor): no such weight
        if (nslice == 1) dw[p * sP + qc * sQ + tap] += tot;
        else atomicAdd(&dw[p * sP + qc * sQ + tap], tot);
    }
}
inline Wgrad3Reduce wgrad3_reduce_desc(const float* partial, float* dw, int P, int Q, int CP, int CQ, int ntap, int nb, long long sP, long long sQ, int qreal) {
    Wgrad3Reduce r;
    r.partial = partial; r.dw = dw; r.P = P; r.Q = Q; r.CP = CP; r.CQ = CQ; r.ntap = ntap; r.nb = nb; r.sP = sP; r.sQ = sQ; r.qreal = qreal;
    return r;
}

template <class T, int TD, int TH, int TW, int KD>
Wgrad3Reduce wgrad3_launch_shape(const Wgrad3Args& a0, float* dw, long long sP, long long sQ, hipStream_t s, int qreal) {
    Wgrad3Args a = a0;
    int CP, CQ;
    wgrad3_tile(a.P, a.Q, (int)sizeof(T), KD == 3 ? 3 : 2, &CP, &CQ);
    const int combos = (a.P / CP) * (a.Q / CQ);
    dim3 grid(a.nb, combos);
#define SEG_W3(CPv, CQv) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3_kernel<T, TD, TH, TW, KD, CPv, CQv>), grid, dim3(256), 0, s, a)
    if (CP == 32 && CQ == 32) SEG_W3(32, 32);
    else if (CP == 32) SEG_W3(32, 16);
    else if (CQ == 32) SEG_W3(16, 32);
    else SEG_W3(16, 16);
#undef SEG_W3
    return wgrad3_reduce_desc(a.partial, dw, a.P, a.Q, CP, CQ, KD * 9, a.nb, sP, sQ, qreal);
}

// 16 -> 16 channels on a large volume (the finest level of the 3-D nets: the longest single launch of the step, 313 us for 226 MB at 4 x 96^3,
// profiles/r03_rocprofv3_kernel_stats.txt): a 4 x 8 x 16 box.  The kernel is latency-bound (one box in flight per workgroup), so what counts is
// bytes per round trip and barriers per voxel: 50 KB instead of 23 KB per trip, 2.7x fewer barrier pairs per voxel, and the halo re-read of x
// drops from 2.8x to 2.1x (PMC round 3: 349 MB fetched for 226 MB algorithmic).  50.5 KB of LDS; 16-bit tensors only (an f32 box would need 127 KB).
template <class T> struct Wgrad3Big16 {
    static bool launch(const Wgrad3Args& a0, float* dw, long long sP, long long sQ, hipStream_t s, int qreal, Wgrad3Reduce* rd) {
        const char* env = knob_s("SEG_W3_BOX16");          // 0: off; 1 (default): where the volume holds enough boxes; 2: wherever the shape fits (tests)
        const int on = env ? atoi(env) : 1;
        int CP, CQ;
        wgrad3_tile(a0.P, a0.Q, 2, 3, &CP, &CQ);
        // P = 16 with 16-channel q-tiles: the 16 -> 16 LUConv of the VNet top level and the finest-level convs of the UNets (16 -> 16, and
        // 32 -> 16 over the decoder's concat: two q-tiles, grid.y = 2)
        if (!on || a0.P != 16 || CQ != 16 || a0.Q % 16 || !wide_box(a0.W)) return false;
        const int combos = a0.Q / 16;
        const long long nbox = num_boxes<4, 8, 16>(a0.N, a0.D, a0.H, a0.W);
        if (on < 2 && nbox < 6ll * a0.nb) return false;    // small volumes keep the 3 x 4 x 16 box (enough boxes per workgroup to amortise its partial tile)
        Wgrad3Args a = a0;
        if (a.nb > nbox) a.nb = (int)nbox;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad3_kernel<T, 4, 8, 16, 3, 16, 16>), dim3(a.nb, combos), dim3(256), 0, s, a);
        *rd = wgrad3_reduce_desc(a.partial, dw, a.P, a.Q, 16, 16, 27, a.nb, sP, sQ, qreal);
        return true;
    }
};
template <> struct Wgrad3Big16<float> {
    static bool launch(const Wgrad3Args&, float*, long long, long long, hipStream_t, int, Wgrad3Reduce*) { return false; }
};

template <class T>
Wgrad3Reduce wgrad3_dispatch(const Wgrad3Args& a, int ndim, float* dw, long long sP, long long sQ, hipStream_t s, int qreal) {
    Wgrad3Reduce rd;
    if (ndim == 3 && Wgrad3Big16<T>::launch(a, dw, sP, sQ, s, qreal, &rd)) return rd;
    if (ndim == 3) return wide_box(a.W) ? wgrad3_launch_shape<T, 3, 4, 16, 3>(a, dw, sP, sQ, s, qreal) : wgrad3_launch_shape<T, 3, 8, 8, 3>(a, dw, sP, sQ, s, qreal);
    return wide_box(a.W) ? wgrad3_launch_shape<T, 1, 8, 16, 1>(a, dw, sP, sQ, s, qreal) : wgrad3_launch_shape<T, 1, 8, 8, 1>(a, dw, sP, sQ, s, qreal);
}

// ------------------------------------------------------------------------------------------------
// image stem: Cimg in 1..3, K = taps*Cimg <= 32 -> ONE MFMA K-step per 16 voxels.  The im2col fragment
// is gathered from the scalar halo in LDS; `center` = 1 evaluates a 1^d conv (K = Cimg) on the same tiles.
// ------------------------------------------------------------------------------------------------
struct StemMArgs {
    const void* in; const void* w; const float* bias; void* out; double* stats;
    const void* dr; float* partial;
    int N, D, H, W, Cimg, Cout, center, nb;
};

// scalar (1..3-channel) image halo -> LDS, loads issued in one batch
template <class T, class B>
__device__ __forceinline__ void stage_scalar_halo(T* Xs, const T* in, int Cimg, const BoxPos& bp, int D, int H, int W) {
    constexpr int NIT = (B::HV * 3 + 255) / 256;
    const int total = B::HV * Cimg;
    T v[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = u * 256 + threadIdx.x;
        const int hv = i / Cimg, ci = i % Cimg;
        const int hx = hv % B::HW, hy = (hv / B::HW) % B::HH, hz = hv / (B::HW * B::HH);
        const int z = bp.z0 + hz - B::PD, y = bp.y0 + hy - 1, x = bp.x0 + hx - 1;
        v[u] = from_f<T>(0.f);
        if (i < total && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
            v[u] = in[((((long long)bp.n * D + z) * H + y) * W + x) * Cimg + ci];
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = u * 256 + threadIdx.x;
        if (i < total) Xs[i] = v[u];
    }
}

template <class B>
__device__ __forceinline__ int stem_k_to_halo(int k, int Cimg, int center, int& ci) {   // reduction index -> halo offset
    const int tap = center ? (B::NTAP / 2) : k / Cimg;
    ci = center ? k : k % Cimg;
    return B::tap_off(tap);
}

// A workgroup handles STEM_NBX consecutive boxes: their (tiny) scalar halos are fetched in ONE round trip, then every box is
// one MFMA per 16 voxels plus the epilogue.  With one box per workgroup the kernel was a 5 us latency chain per 6 KB of output
// (18432 workgroups at 96^3: 1 TB/s).
constexpr int STEM_NBX = 4;
template <class T, int TD, int TH, int TW, int KD>
__global__ __launch_bounds__(256, SEG_STEM_OCC) void stem_fwd_kernel(StemMArgs a) {
    typedef Box<TD, TH, TW, KD> B;
    constexpr int MT = B::V / 64, OLD = 16 + 8;
    constexpr int XS = B::HV * 3, OS = B::V * OLD;
    __shared__ T Xs[STEM_NBX][XS];
    __shared__ __attribute__((aligned(16))) T Os[OS];
    __shared__ float red[512];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const long long nbox = (long long)a.N * ((a.D + TD - 1) / TD) * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    const long long b0 = (long long)blockIdx.x * STEM_NBX;
    const int co0 = blockIdx.y * 16, Cimg = a.Cimg;
    const T* in = (const T*)a.in;
    const int K = a.center ? Cimg : B::NTAP * Cimg;
#pragma unroll
    for (int i = 0; i < STEM_NBX; ++i)
        if (b0 + i < nbox) stage_scalar_halo<T, B>(Xs[i], in, Cimg, box_pos<B, TD, TH, TW>(b0 + i, a.D, a.H, a.W), a.D, a.H, a.W);
    const typename Mma<T>::frag bf = load8((const T*)a.w + (long long)(co0 + l15) * 32 + q * 8);
    int koff[8];                                     // element offset of this lane's 8 reduction slots (-1: padding)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = q * 8 + j;
        int ci;
        const int toff = stem_k_to_halo<B>(k < K ? k : 0, Cimg, a.center, ci);
        koff[j] = k < K ? toff * Cimg + ci : -1;
    }
    __syncthreads();
    for (int i = 0; i < STEM_NBX && b0 + i < nbox; ++i) {
        const BoxPos bp = box_pos<B, TD, TH, TW>(b0 + i, a.D, a.H, a.W);
        f32x4 acc[MT][1];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int hb = B::halo_base((wv * MT + m) * 16 + l15) * Cimg;
            typename Mma<T>::frag af;
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = koff[j] >= 0 ? Xs[i][hb + koff[j]] : from_f<T>(0.f);
            acc[m][0] = Mma<T>::run(af, bf, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        if (i) __syncthreads();                      // the previous box's epilogue is done with Os / red
        box_epilogue<T, B, TW, TH, MT, 1>(acc, Os, red, a.bias, (T*)a.out, a.stats, bp, co0, a.N, a.D, a.H, a.W, a.Cout);
    }
}

// dW[co][k] = sum_v dR[v][co] * xcol[v][k]; partial tile [16][32] per workgroup (Cout == 16 per grid.y slice)
template <class T, int TD, int TH, int TW, int KD>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemMArgs a) {
    typedef Box<TD, TH, TW, KD> B;
    constexpr int DLD = WLd<T, 16>::v, CLD = WLd<T, 32>::v;
    constexpr int NR = sizeof(T) == 4 ? 8 : 2, KS = B::V / 32;
    __shared__ T Ds[B::V * DLD];
    __shared__ T Xc[B::V * CLD];
    __shared__ T Xs[B::HV * 3];
    __shared__ float wred[4 * 2 * 256];
    __shared__ int ktab[32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int co0 = blockIdx.y * 16, Cimg = a.Cimg;
    const int K = a.center ? Cimg : B::NTAP * Cimg;
    if (tid < 32) {
        int ci;
        const int toff = stem_k_to_halo<B>(tid < K ? tid : 0, Cimg, a.center, ci);
        ktab[tid] = tid < K ? toff * Cimg + ci : -1;
    }
    const T* dr = (const T*)a.dr;
    const T* in = (const T*)a.in;
    const long long nbox = (long long)a.N * ((a.D + TD - 1) / TD) * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (long long b = blockIdx.x; b < nbox; b += gridDim.x) {
        const BoxPos bp = box_pos<B, TD, TH, TW>(b, a.D, a.H, a.W);
        __syncthreads();
        for (int i = tid; i < B::V * 2; i += 256) {                  // dR tile [voxel][16]
            const int v = i >> 1, c8 = i & 1;
            const int xx = bp.x0 + v % TW, yy = bp.y0 + (v / TW) % TH, zz = bp.z0 + v / (TW * TH);
            vec<T, 8> val = zero8<T>();
            if (xx < a.W && yy < a.H && zz < a.D)
                val = load8(dr + ((((long long)bp.n * a.D + zz) * a.H + yy) * a.W + xx) * a.Cout + co0 + c8 * 8);
            store8(&Ds[v * DLD + c8 * 8], val);
        }
        stage_scalar_halo<T, B>(Xs, in, Cimg, bp, a.D, a.H, a.W);
        __syncthreads();
        for (int i = tid; i < B::V * 32; i += 256) {                 // im2col tile [voxel][32]
            const int v = i >> 5, k = i & 31;
            const int ko = ktab[k];
            Xc[v * CLD + k] = ko >= 0 ? Xs[B::halo_base(v) * Cimg + ko] : from_f<T>(0.f);
        }
        __syncthreads();
        for (int ks = wv; ks < KS; ks += 4) {                        // K steps split over the 4 waves
            int drow[NR], xrow[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = sizeof(T) == 4 ? (4 * j + q) : (16 * j + 4 * q + (l15 >> 2));
                drow[j] = (ks * 32 + r) * DLD;
                xrow[j] = (ks * 32 + r) * CLD;
            }
            const typename Mma<T>::frag af = TrFrag<T, DLD>::load(Ds, drow, 0, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = Mma<T>::run(af, TrFrag<T, CLD>::load(Xc, xrow, j * 16, lane), acc[j]);
        }
    }
    // cross-wave sum of the [16][32] tile, then ONE partial tile per workgroup
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) wred[(wv * 2 + j) * 256 + (4 * q + r) * 16 + l15] = acc[j][r];
    __syncthreads();
    for (int i = tid; i < 512; i += 256) {
        const int j = i >> 8, e = i & 255;
        const float s = wred[(0 * 2 + j) * 256 + e] + wred[(1 * 2 + j) * 256 + e] + wred[(2 * 2 + j) * 256 + e] + wred[(3 * 2 + j) * 256 + e];
        const int p = e >> 4, kk = j * 16 + (e & 15);
        if (kk < K) a.partial[((long long)blockIdx.x * a.Cout + co0 + p) * K + kk] = s;
    }
}

__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* partial, float* dw, int P, int K, int Cimg, int ntap, int center,
                                                                int nb) {
    // dw layout [co][ci][tap]; partial [nb][co][k], k = tap*Cimg + ci (center: k = ci, tap 0 of a 1^d kernel)
    const int total = P * K;
    const int b0 = blockIdx.y * 32, b1 = (b0 + 32 < nb) ? b0 + 32 : nb;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        float s = 0.f;
        for (int b = b0; b < b1; ++b) s += partial[(long long)b * total + i];
        const int p = i / K, k = i % K;
        const int tap = center ? 0 : k / Cimg, ci = center ? k : k % Cimg;
        atomicAdd(&dw[((long long)p * Cimg + ci) * (center ? 1 : ntap) + tap], s);
    }
}

template <class T, int TD, int TH, int TW, int KD>
void stem_launch_shape(const StemMArgs& a, bool wgrad, float* dw, hipStream_t s) {
    const long long nbox = num_boxes<TD, TH, TW>(a.N, a.D, a.H, a.W);
    if (!wgrad) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_fwd_kernel<T, TD, TH, TW, KD>), dim3((unsigned)((nbox + STEM_NBX - 1) / STEM_NBX), a.Cout / 16), dim3(256), 0,
                           s, a);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(stem_wgrad_kernel<T, TD, TH, TW, KD>), dim3(a.nb, a.Cout / 16), dim3(256), 0, s, a);
        const int K = a.center ? a.Cimg : KD * 9 * a.Cimg;
        hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((a.Cout * K + 255) / 256, (a.nb + 31) / 32), dim3(256), 0, s, (const float*)a.partial, dw,
                           a.Cout, K, a.Cimg, KD * 9, a.center, a.nb);
    }
}

template <class T>
void stem_dispatch(const StemMArgs& a, int ndim, bool wgrad, float* dw, hipStream_t s) {
    if (ndim == 3) {
        if (wide_box(a.W)) stem_launch_shape<T, 3, 4, 16, 3>(a, wgrad, dw, s);
        else stem_launch_shape<T, 3, 8, 8, 3>(a, wgrad, dw, s);
    } else {
        if (wide_box(a.W)) stem_launch_shape<T, 1, 8, 16, 1>(a, wgrad, dw, s);
        else stem_launch_shape<T, 1, 8, 8, 1>(a, wgrad, dw, s);
    }
}

inline long long boxes_for(int ndim, int N, int D, int H, int W) {
    if (ndim == 3) return wide_box(W) ? num_boxes<3, 4, 16>(N, D, H, W) : num_boxes<3, 8, 8>(N, D, H, W);
    return wide_box(W) ? num_boxes<1, 8, 16>(N, 1, H, W) : num_boxes<1, 8, 8>(N, 1, H, W);
}

}  // namespace

// ---- host entry points (declared in kernels.h) -----------------------------------------------------
void launch_conv3(const void* in, const void* w, const float* bias, void* out, double* stats, int N, int D, int H, int W, int Cin,
                  int Cout, int ndim, int dtype, hipStream_t s, const void* in1, int C0) {
    Conv3Args a;
    a.in1 = in1; a.C0 = C0;
    a.in = in; a.w = w; a.bias = bias; a.out = out; a.stats = stats;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.Kpad = ((ndim == 3 ? 27 : 9) * Cin + 31) / 32 * 32;
    static const int dbg = 0;
    a.dbg = dbg;
    a.trace = nullptr;
    // diagnostics (tools/bench_conv3.py): per-workgroup phase timeline of one launch, printed to stderr
    static const int trace_on = 0;
    static unsigned long long* tbuf = nullptr;
    const size_t tmax = 1 << 18;
    if (trace_on) {
        if (!tbuf) (void)hipMalloc(&tbuf, tmax * 6 * sizeof(unsigned long long));
        (void)hipMemsetAsync(tbuf, 0, tmax * 6 * sizeof(unsigned long long), s);
        a.trace = tbuf;
    }
    if (dtype == DT_F32) conv3_dispatch<float>(a, ndim, s);
    else if (dtype == DT_F16) conv3_dispatch<f16>(a, ndim, s);
    else conv3_dispatch<bf16>(a, ndim, s);
    if (trace_on) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(tmax * 6);
        (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
        size_t n = 0;
        unsigned long long t0 = ~0ull, t1 = 0;
        double ph[5] = {0, 0, 0, 0, 0};
        for (size_t i = 0; i < tmax; ++i) {
            const unsigned long long* r = &h[i * 6];
            if (!r[0] || !r[5]) continue;
            ++n;
            if (r[0] < t0) t0 = r[0];
            if (r[5] > t1) t1 = r[5];
            for (int k = 0; k < 5; ++k) ph[k] += (double)(r[k + 1] - r[k]);
        }
        if (n) {
            // start-time histogram in 10 buckets over the kernel span -> how many "rounds" of workgroups there are
            int hist[10] = {0};
            for (size_t i = 0; i < tmax; ++i) if (h[i * 6] && h[i * 6 + 5]) hist[(int)((h[i * 6] - t0) * 10 / (t1 - t0 + 1))]++;
            fprintf(stderr, "[conv3 trace] C%d->%d %dx%dx%d wgs=%zu span=%.1fus  mean/wg(us): stage=%.2f sync=%.2f loop=%.2f sync=%.2f epilogue=%.2f  starts:",
                    Cin, Cout, D, H, W, n, (t1 - t0) * 0.01, ph[0] / n * 0.01, ph[1] / n * 0.01, ph[2] / n * 0.01, ph[3] / n * 0.01, ph[4] / n * 0.01);
            for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", hist[k]);
            fprintf(stderr, "\n");
        }
    }
}

int wgrad3_blocks_per_combo(int ndim, int N, int D, int H, int W, int P, int Q, int esz) {
    int CP, CQ;
    wgrad3_tile(P, Q, esz, ndim, &CP, &CQ);
    const int combos = (P / CP) * (Q / CQ);
    // tuning knobs.  512 while the kernel exposed its staging latency (r01: 256 / 384 / 768 -> 691 / 694 / 680 vs 700 volumes/s); with the next
    // box prefetched into registers one workgroup per CU is enough and a third fewer partial tiles are written and re-read:
    // 256 vs 512 = 946 vs 941 and 959 vs 953 volumes/s in two sessions (profiles/r03_policy_resweep_ab.log, r03_stemx_coefs_ab.log)
    // One box, every BASELINE config (profiles/r03_wgrad_policy_configs_ab.log), total / total16 = 256/256, 512/1024, 256/1024, 512/256:
    // C3 4.15 / 4.20 / 4.18 / 4.15 ms, C4 4.58 / 4.55 / 4.43 / 4.68, C5 4.55 / 4.61 / 4.57 / 4.58, C2 (2-D) 6.65 / 6.40 / 6.68 / 6.37 -> the
    // 2-D boxes keep 512 workgroups
#ifdef SEG_DIAG
    static const int total_env = knob_i("SEG_W3_TOTAL", 0), minbox = knob_i("SEG_W3_MINBOX", 6);      // diagnostic variant builds only (tools/build_variant.py)
#else
    static const int total_env = 0, minbox = 6;
#endif
    const int total = total_env > 0 ? total_env : (ndim == 3 ? 256 : 512);
    // 16 -> 16 channels (the finest level): four workgroups fit a CU (23 KB LDS, 113 VGPRs) and the partial tile is 27 KB, so the
    // staging latency of one workgroup can hide behind the others
    // round 2 (no prefetch), standalone 4x96^3: 512 -> 168 us, 1024 -> 123 us, 2048 -> 137 us (r02_wgrad16_ab.log), step unchanged.  With the
    // prefetching kernel inside the step (profiles/r03_wgrad_policy_ab3.log): 128 -> 947, 256 -> 967 / 967, 384 -> 961, 1024 -> 956-958
    // volumes/s - one workgroup per CU leaves the bandwidth-bound 96^3 GroupNorm passes of the main stream the rest of the machine
    static const int total16 = 256;
    long long nb = (P == 16 && Q == 16 ? total16 : total) / combos;
    if (nb < 1) nb = 1;
    const long long nbox = boxes_for(ndim, N, D, H, W);
    // small levels: every workgroup writes (and the reduce re-reads) a full partial tile, so do not split the
    // box list finer than ~6 boxes per workgroup (measured, profiles/r01_wgrad3_policy_ab_step31.log: 3 -> 694, 4 -> 698,
    // 5 -> 698, 6 -> 700, 8 -> 690 volumes/s; 256 / 384 / 768 total workgroups instead of 512: 691 / 694 / 680)
    if (nb > (nbox + minbox - 1) / minbox) nb = (nbox + minbox - 1) / minbox;
    if (nb < 1) nb = 1;
    return (int)nb;
}

size_t wgrad3_partial_bytes(int ndim, int N, int D, int H, int W, int P, int Q) {
    // sized for the wide (32 x 32) tiling: the 16-channel q-tiles of the 16-bit tensors never need more (twice the combos, at most the same
    // workgroups per combo, half the tile)
    const int CP = P >= 32 ? 32 : 16, CQ = Q >= 32 ? 32 : 16;
    const int combos = (P / CP) * (Q / CQ);
    return (size_t)combos * wgrad3_blocks_per_combo(ndim, N, D, H, W, P, Q, 4) * CP * (ndim == 3 ? 27 : 9) * CQ * sizeof(float);
}

void launch_wgrad3_reduce(const Wgrad3Reduce* list, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += W3_BATCH) {
        Wgrad3ReduceBatch bt;
        const int m = n - i0 < W3_BATCH ? n - i0 : W3_BATCH;
        unsigned gx = 1, gy = 1;
        for (int i = 0; i < m; ++i) {
            bt.r[i] = list[i0 + i];
            const long long total = (long long)list[i0 + i].P * list[i0 + i].Q * list[i0 + i].ntap;
            long long blocks = (total + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            gx = blocks > gx ? (unsigned)blocks : gx;
            const unsigned sl = (unsigned)((list[i0 + i].nb + 31) / 32);
            gy = sl > gy ? sl : gy;
        }
        for (int i = m; i < W3_BATCH; ++i) bt.r[i] = bt.r[0];
        hipLaunchKernelGGL(wgrad3_reduce_kernel, dim3(gx, gy, (unsigned)m), dim3(256), 0, s, bt);
    }
}

void launch_wgrad3(const void* dr, const void* x, float* partial, float* dw, int N, int D, int H, int W, int P, int Q, int ndim,
                   int dtype, hipStream_t s, const void* x1, int C0, int qreal, Wgrad3Reduce* defer) {
    const int T = ndim == 3 ? 27 : 9;
    if (qreal <= 0 || qreal > Q) qreal = Q;
    Wgrad3Args a;
    a.x1 = x1; a.C0 = C0;
    a.dr = dr; a.x = x; a.partial = partial;
    a.N = N; a.D = D; a.H = H; a.W = W; a.P = P; a.Q = Q;
    a.nb = wgrad3_blocks_per_combo(ndim, N, D, H, W, P, Q, dtype == DT_F32 ? 4 : 2);
#ifdef SEG_W3_TRACE
    static unsigned long long* tbuf = nullptr;
    const size_t tmax = 8192;
    if (!tbuf) (void)hipMalloc(&tbuf, tmax * 8 * sizeof(unsigned long long));
    (void)hipMemsetAsync(tbuf, 0, tmax * 8 * sizeof(unsigned long long), s);
    a.trace = tbuf;
#endif
    const Wgrad3Reduce rd = dtype == DT_F32 ? wgrad3_dispatch<float>(a, ndim, dw, (long long)qreal * T, T, s, qreal)
                            : dtype == DT_F16 ? wgrad3_dispatch<f16>(a, ndim, dw, (long long)qreal * T, T, s, qreal)
                                              : wgrad3_dispatch<bf16>(a, ndim, dw, (long long)qreal * T, T, s, qreal);
    if (defer) *defer = rd;                      // the caller batches the reduce (launch_wgrad3_reduce) behind further weight-gradient kernels
    else launch_wgrad3_reduce(&rd, 1, s);
#ifdef SEG_W3_TRACE
    {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h(tmax * 8);
        (void)hipMemcpy(h.data(), tbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        size_t n = 0;
        for (size_t i = 0; i < tmax; ++i) {
            unsigned long long tot = 0;
            for (int k = 0; k < 8; ++k) tot += h[i * 8 + k];
            if (!tot) continue;
            ++n;
            for (int k = 0; k < 8; ++k) ph[k] += (double)h[i * 8 + k];
        }
        if (n) fprintf(stderr, "[wgrad3 trace] P%d Q%d %dx%dx%dx%d wgs=%zu  mean per workgroup (us): first-issue %.2f | barrier1 %.2f  wait+store %.2f  barrier2 %.2f  issue %.2f  sweep %.2f\n",
                       P, Q, N, D, H, W, n, ph[5] / n * 0.01, ph[0] / n * 0.01, ph[1] / n * 0.01, ph[2] / n * 0.01, ph[3] / n * 0.01, ph[4] / n * 0.01);
    }
#endif
}

}  // namespace seg

namespace seg {
int stem_wgrad_blocks(int ndim, int N, int D, int H, int W) {
    long long nb = boxes_for(ndim, N, D, H, W);
    return (int)(nb < 1024 ? nb : 1024);
}
size_t stem_wgrad_partial_bytes(int ndim, int N, int D, int H, int W, int Cout) {
    return (size_t)stem_wgrad_blocks(ndim, N, D, H, W) * Cout * 32 * sizeof(float);
}
// forward: out = conv(in, w) (+bias, statistics); `center` = 1: 1^d conv.  w = packed [Cout][32] in dtype.
void launch_stem_fwd(const void* in, const void* w, const float* bias, void* out, double* stats, int N, int D, int H, int W, int Cimg,
                     int Cout, int center, int ndim, int dtype, hipStream_t s) {
    StemMArgs a{};
    a.in = in; a.w = w; a.bias = bias; a.out = out; a.stats = stats;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cimg = Cimg; a.Cout = Cout; a.center = center;
    if (dtype == DT_F32) stem_dispatch<float>(a, ndim, false, nullptr, s);
    else if (dtype == DT_F16) stem_dispatch<f16>(a, ndim, false, nullptr, s);
    else stem_dispatch<bf16>(a, ndim, false, nullptr, s);
}
void launch_stem_wgrad(const void* dr, const void* in, float* partial, float* dw, int N, int D, int H, int W, int Cimg, int Cout,
                       int center, int ndim, int dtype, hipStream_t s) {
    StemMArgs a{};
    a.dr = dr; a.in = in; a.partial = partial;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cimg = Cimg; a.Cout = Cout; a.center = center;
    a.nb = stem_wgrad_blocks(ndim, N, D, H, W);
    if (dtype == DT_F32) stem_dispatch<float>(a, ndim, true, dw, s);
    else if (dtype == DT_F16) stem_dispatch<f16>(a, ndim, true, dw, s);
    else stem_dispatch<bf16>(a, ndim, true, dw, s);
}
}  // namespace seg
