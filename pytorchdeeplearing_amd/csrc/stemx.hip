// Fused input block of VNet / first block of UNet at full resolution (networks/VNet3d.py:25-43, networks/Unet3d.py:64-86):
//     y = relu(drop(GN(conv3(x)))) [+ relu(drop(GN(conv1(x))))]            x: 1..3 image channels, 16 output channels
// The image stems are K <= 27 contractions - one MFMA per 16 voxels - so their raw outputs are RECOMPUTED wherever they are
// needed instead of being written and re-read as 16-channel tensors (the round-1 pipeline moved ~1.4 GB per step through
// stem_fwd x2 -> gn_act -> gn_bwd_reduce/apply -> stem_wgrad x2 for a 7 MB image):
//   forward   SX_FWD_STATS   GroupNorm sums of both branches straight from the image              (reads the image only)
//             SX_FWD_APPLY   recompute, normalise, ReLU, add, write y                               (image -> y)
//   backward  SX_BWD_REDUCE  recompute r, Q = {sum dz, sum dz*r} per (n, c) and branch             (gradient sources + image)
//             SX_BWD_APPLY   recompute r, d(raw) = A*dz + B*r + C in registers, and feed it as the B operand of a K = 16 MFMA
//                            against the transposed im2col tile: the stem weight gradients without any d(raw) tensor
// (gn_finalize / gn_bwd_finalize of norm.hip run between the passes, unchanged).
// im2col without scalar gathers: the 1-channel halo is kept in LDS as three copies shifted by 0 / 1 / 2 elements, so the 16
// voxels of an x row under any tap (kd, kh, kw) start 8-byte aligned in copy kw; ONE transposing read (ds_read_b64_tr_b16)
// then delivers 4 taps x 16 voxels in MFMA operand layout - two reads per 16x16x32 step instead of eight ds_read_u16 + selects.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace seg {
namespace {

enum { SX_FWD_STATS = 0, SX_FWD_APPLY = 1, SX_BWD_REDUCE = 2, SX_BWD_APPLY = 3 };
constexpr int SX_C = 16;          // output channels of the stems (init_features, fixed by seg_create)
constexpr int SX_MAXCI = 3;       // image channels

typedef seg_stemx_args StemxArgs;

// box of TD x TH rows of 16 voxels; tile m = row (vz, vy) = (m / TH, m % TH)
template <int TD_, int TH_, int KD_> struct SBox {
    static constexpr int TD = TD_, TH = TH_, KD = KD_, V = TD * TH * 16, NTILE = TD * TH;
    static constexpr int HD = TD + KD - 1, HH = TH + 2, HW = 18, HWP = 20, PD = (KD - 1) / 2, NTAP = KD * 9;
    static constexpr int HV = HD * HH * HWP, FRONT = 4, HVP = FRONT + HV + 16;   // one halo plane: FRONT elements of slack in front (copy c holds halo[L + c] at FRONT + L: no range test when
                                                                                  // it is written, 8-byte alignment kept) and slack behind (a shifted read never leaves the plane)
    static __device__ __forceinline__ int tap_row(int tap) { return ((tap / 9) * HH + (tap / 3) % 3) * HWP; }   // (kd, kh) part of the halo offset
};

template <class T, class B, int MODE, int CIMG, int NDY>
__global__ __launch_bounds__(256) void stemx_kernel(StemxArgs a) {
    constexpr bool H16 = sizeof(T) == 2;
    constexpr int NCOPY = H16 ? 3 : 1;                            // shifted copies of every image-channel plane
    constexpr int TM = B::NTILE / 4;                              // tiles per wave
    constexpr bool BWD = MODE == SX_BWD_REDUCE || MODE == SX_BWD_APPLY;
    static_assert(B::NTILE % 4 == 0, "tiles must split over the four waves");
    constexpr int XELEMS = (CIMG * NCOPY + 1) * B::HVP;       // + one all-zero plane: the padding k slots (k >= K) read it at the tile's offset
    constexpr int DELEMS = BWD ? 2 * NDY * B::V * SX_C : 8;         // two buffers of NDY gradient-source tiles
    __shared__ __attribute__((aligned(16))) T Xc[XELEMS];
    __shared__ __attribute__((aligned(16))) T Dy[DELEMS];
    __shared__ float red[4 * 2 * 2 * SX_C * 2];                   // [wave][branch][which][co] partial sums / weight-gradient tiles (2 KB)
    __shared__ float wred[MODE == SX_BWD_APPLY ? 4 * 3 * 256 : 4];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    constexpr int Cimg = CIMG, K = B::NTAP * Cimg, kc0 = (B::NTAP / 2) * Cimg;   // kc0: first k slot of the centre tap
    const int nbr = a.w1 ? 2 : 1;
    const T* img = (const T*)a.img;
    constexpr int ZERO = CIMG * NCOPY * B::HVP;
    const int nbx = a.W / 16 + (a.W % 16 != 0), nby = (a.H + B::TH - 1) / B::TH, nbz = (a.D + B::TD - 1) / B::TD;
    const int nbox = a.N * nbz * nby * nbx;
    const long long vol = (long long)a.D * a.H * a.W;

    // ---- weight fragments (lane: co = l15, k = 8q .. 8q+7); the 1^d branch sits on the centre tap's k slots of the same im2col
    typename Mma<T>::frag b3 = load8((const T*)a.w3 + l15 * 32 + q * 8), b1 = zero8<T>();
    if (nbr == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = q * 8 + j - kc0;
            b1[j] = (kk >= 0 && kk < Cimg) ? ((const T*)a.w1)[l15 * 32 + kk] : from_f<T>(0.f);
        }
    }
    // ---- im2col addressing.  16-bit: two transposing reads per fragment, this lane's address serves k = 8q + 4h + (l15 >> 2), h = 0, 1
    //      (element offset inside the copies, tile base added per tile).  f32: eight scalar reads, k = 8q + j.
    int offA[H16 ? 2 : 8];
    auto koff16 = [&](int k, int piece) -> int {          // copy (ci, kw) + (kd, kh) row + 4 * piece
        if (k >= K) return ZERO + 4 * piece;
        const int tap = k / Cimg, ci = k % Cimg;
        return (ci * NCOPY + tap % 3) * B::HVP + B::FRONT + B::tap_row(tap) + 4 * piece;
    };
    if constexpr (H16) {
#pragma unroll
        for (int h = 0; h < 2; ++h) offA[h] = koff16(8 * q + 4 * h + (l15 >> 2), l15 & 3);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * q + j;
            offA[j] = k < K ? (k % Cimg) * B::HVP + B::FRONT + B::tap_row(k / Cimg) + (k / Cimg) % 3 : -1;
        }
    }
    // transposed im2col rows for the weight gradient: M tile mt holds k = 16 mt + l15, this lane's 4 voxels are 4q .. 4q+3
    int offW[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int k = 16 * mt + l15;
        if constexpr (H16) offW[mt] = koff16(k, q);
        else offW[mt] = k < K ? (k % Cimg) * B::HVP + B::FRONT + B::tap_row(k / Cimg) + (k / Cimg) % 3 + 4 * q : -1;
    }
    for (int i = tid; i < B::HVP; i += 256) Xc[ZERO + i] = from_f<T>(0.f);

    // running sums of this workgroup for the current sample (flushed when the sample changes): [branch][which]
    float acc_s[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    f32x4 accW3[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, accW1 = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur_n = -1;
    auto flush = [&](int n) {                          // non-swapped layout: lane co = l15; sums over the 4 q groups, then over the waves
        if (MODE != SX_FWD_STATS && MODE != SX_BWD_REDUCE) return;
        __syncthreads();
#pragma unroll
        for (int br = 0; br < 2; ++br)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                float v = acc_s[br][w];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (q == 0) red[((wv * 2 + br) * 2 + w) * SX_C + l15] = v;
                acc_s[br][w] = 0.f;
            }
        __syncthreads();
        if (tid < 2 * 2 * SX_C && n >= 0) {
            const int br = tid / (2 * SX_C), w = (tid / SX_C) % 2, co = tid % SX_C;
            if (br < nbr) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s += red[((k * 2 + br) * 2 + w) * SX_C + co];
                double* dst = MODE == SX_FWD_STATS ? (br ? a.stats1 : a.stats3) : (br ? a.Q1 : a.Q3);
                atomicAdd(dst + (((long long)(blockIdx.x % STAT_REP) * a.N + n) * SX_C + co) * 2 + w, s);
            }
        }
    };

    // ---- box pipeline: while box b is multiplied, the image halo of the next box travels to registers and its gradient-source
    //      tiles to the other LDS buffer (asynchronous direct copies); the only exposed latency is the first box's
    // (all index arithmetic of the staging is box-independent and done ONCE: per box a halo element costs three adds, three
    // compares and a load - with it inside the box loop the kernels were bound by integer division, not by memory)
    constexpr int NIT = (B::HD * B::HH * B::HW * CIMG + 255) / 256;
    constexpr int total = B::HD * B::HH * B::HW * CIMG;
    T himg[NIT];
    int hsrc[NIT], hdst[NIT];                 // packed (hz, hy, hx, ci) of this thread's halo elements; LDS index in copy 0 (-1: none)
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
        const int i = u * 256 + tid;
        const int hv = i / CIMG, ci = i % CIMG;
        const int hx = hv % B::HW, hy = (hv / B::HW) % B::HH, hz = hv / (B::HW * B::HH);
        hsrc[u] = ((hz << 8 | hy) << 8 | hx) << 2 | ci;
        hdst[u] = i < total ? ci * NCOPY * B::HVP + B::FRONT + (hv / B::HW) * B::HWP + hx : -1;
    }
    struct BoxAt { int x0, y0, z0, n; };
    auto box_at = [&](int b_) {                  // 32-bit unsigned: the box count stays far below 2^31 (signed division costs twice the scalar instructions)
        unsigned b = (unsigned)b_;
        BoxAt p;
        p.x0 = (int)(b % (unsigned)nbx) * 16; b /= (unsigned)nbx;
        p.y0 = (int)(b % (unsigned)nby) * B::TH; b /= (unsigned)nby;
        p.z0 = (int)(b % (unsigned)nbz) * B::TD;
        p.n = (int)(b / (unsigned)nbz);
        return p;
    };
    auto load_img = [&](const BoxAt& p) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int hs = hsrc[u];
            const int z = p.z0 + (hs >> 18) - B::PD, y = p.y0 + ((hs >> 10) & 255) - 1, x = p.x0 + ((hs >> 2) & 255) - 1;
            himg[u] = from_f<T>(0.f);
            if (hdst[u] >= 0 && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W)
                himg[u] = img[((((long long)p.n * a.D + z) * a.H + y) * a.W + x) * CIMG + (hs & 3)];
        }
    };
    auto issue_dy = [&](const BoxAt& p, int buf) {
        constexpr int GPV = SX_C * (int)sizeof(T) / 16;                          // 16-B granules per voxel row: 2 (16-bit) or 4 (f32)
        constexpr int NG = B::V * GPV, NI = (NG + 255) / 256;
#pragma unroll
        for (int sidx = 0; sidx < NDY; ++sidx) {
            const i32x4 rs = make_rsrc((const T*)a.dy[sidx] + (long long)p.n * vol * SX_C, (unsigned)(vol * SX_C * sizeof(T)));
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int g = (u * 4 + wv) * 64 + lane;
                const int v = g / GPV, piece = g % GPV;
                const int vx = v % 16, vy = (v / 16) % B::TH, vz = v / (16 * B::TH);
                const int z = p.z0 + vz, y = p.y0 + vy, x = p.x0 + vx;
                const bool ok = v < B::V && z < a.D && y < a.H && x < a.W;
                const unsigned off = ok ? (unsigned)((((long long)z * a.H + y) * a.W + x) * SX_C * sizeof(T)) + piece * 16u : DMA_OOB;
                if ((u * 4 + wv) * 64 < B::V * GPV)
                    dma16_async(rs, (T*)((char*)(Dy + (buf * NDY + sidx) * B::V * SX_C) + (size_t)(u * 4 + wv) * 1024), off);
            }
        }
    };
    // ---- per-sample coefficients.  They are loaded for the NEXT box right behind its prefetch and picked up at the top of the next trip,
    // where everything in flight is waited for anyway.  Loaded in the middle of a trip (as the first version did, behind `a.bias3 ? .. : ..`
    // tests) every one of them carried an s_waitcnt vmcnt(0) - which, vmcnt retiring in order, also drained the prefetch just issued: the
    // box pipeline waited for the next box before it multiplied the current one.  All loads branch-free (a missing branch / bias reads a
    // valid dummy address and is ignored).
    struct Coefs { float sc3, sh3, sc1, sh1, cf3[3], cf1[3], sw3[4], tw3[4], sw1[4], tw1[4]; };
    const float* const scale1p = nbr == 2 ? a.scale1 : a.scale3;
    const float* const shift1p = nbr == 2 ? a.shift1 : a.shift3;
    auto load_coefs = [&](int n, Coefs& c) {
        const long long nc = (long long)n * SX_C;
        if (BWD) {
            c.sc3 = a.scale3[nc + l15]; c.sh3 = a.shift3[nc + l15];
            c.sc1 = scale1p[nc + l15]; c.sh1 = shift1p[nc + l15];
            if (MODE == SX_BWD_APPLY) {
                const float* const c1p = nbr == 2 ? a.coef1 : a.coef3;
#pragma unroll
                for (int j = 0; j < 3; ++j) { c.cf3[j] = a.coef3[(nc + l15) * 3 + j]; c.cf1[j] = c1p[(nc + l15) * 3 + j]; }
            }
        }
        if (MODE == SX_FWD_APPLY) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * q + r;
                c.sw3[r] = a.scale3[nc + co]; c.tw3[r] = a.shift3[nc + co];
                c.sw1[r] = scale1p[nc + co]; c.tw1[r] = shift1p[nc + co];
            }
        }
    };
    // biases do not depend on the sample: once, landed before the loop
    float bs3, bs1, bw3[4], bw1[4];
    {
        const float* const dummy = (const float*)a.w3;            // 16 x 32 weights: always there, always >= 64 bytes
        const float* const b3p = a.bias3 ? a.bias3 : dummy;
        const float* const b1p = (nbr == 2 && a.bias1) ? a.bias1 : dummy;
        bs3 = b3p[l15]; bs1 = b1p[l15];
#pragma unroll
        for (int r = 0; r < 4; ++r) { bw3[r] = b3p[4 * q + r]; bw1[r] = b1p[4 * q + r]; }
        settle(bs3); settle(bs1);
#pragma unroll
        for (int r = 0; r < 4; ++r) { settle(bw3[r]); settle(bw1[r]); }
        if (!a.bias3) { bs3 = 0.f; for (int r = 0; r < 4; ++r) bw3[r] = 0.f; }
        if (!(nbr == 2 && a.bias1)) { bs1 = 0.f; for (int r = 0; r < 4; ++r) bw1[r] = 0.f; }
    }
    Coefs cnext{};
    int cur = 0;
    BoxAt pnext{0, 0, 0, 0};
    if ((int)blockIdx.x < nbox) {
        pnext = box_at((int)blockIdx.x);
        load_img(pnext);
        if (BWD) issue_dy(pnext, 0);
        load_coefs(pnext.n, cnext);
    }
    for (int b = blockIdx.x; b < nbox; b += gridDim.x) {
        const BoxAt bp = pnext;                          // (computed one trip ago for the prefetch)
        const int x0 = bp.x0, y0 = bp.y0, z0 = bp.z0, n = bp.n;
        if (n != cur_n) { if (cur_n >= 0) flush(cur_n); cur_n = n; }
        __syncthreads();                                 // the previous box is done with the image copies and the other gradient buffer
        // ---- image halo (loaded during the previous box) -> every shifted copy
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            if (hdst[u] >= 0) {
#pragma unroll
                for (int c = 0; c < NCOPY; ++c) Xc[hdst[u] + c * B::HVP - c] = himg[u];       // copy c holds halo[L + c] at FRONT + L
            }
        }
        wait_vmem();                                     // the gradient tiles and the coefficients of box b have landed
        const Coefs cc = cnext;                          // this box's coefficients (loaded behind its prefetch, one trip ago)
        __syncthreads();
        if (b + (int)gridDim.x < nbox) {
            pnext = box_at(b + (int)gridDim.x);
            load_img(pnext);
            if (BWD) issue_dy(pnext, cur ^ 1);
            load_coefs(pnext.n, cnext);
        }
        const T* DyB = Dy + cur * NDY * B::V * SX_C;
        cur ^= 1;

        const float sc3 = cc.sc3, sh3 = cc.sh3, sc1 = cc.sc1, sh1 = cc.sh1;
        const float* const cf3 = cc.cf3;
        const float* const cf1 = cc.cf1;
        const float* const sw3 = cc.sw3;
        const float* const tw3 = cc.tw3;
        const float* const sw1 = cc.sw1;
        const float* const tw1 = cc.tw1;

        // FULL: the box lies inside the volume (every box of the benchmark shapes) - no row / voxel range tests, no exec-masked blocks in the tile loop
        auto tiles = [&](auto full_t) {
        constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
        for (int mm = 0; mm < TM; ++mm) {
            const int m = wv * TM + mm, vz = m / B::TH, vy = m % B::TH;
            const int tb = (vz * B::HH + vy) * B::HWP;                  // halo index of (tile row, x = 0) under tap (0, 0, 0)
            const int z = z0 + vz, y = y0 + vy;
            const bool row_ok = FULL || (z < a.D && y < a.H);
            // im2col fragment: element j of this lane = voxel l15 under k slot 8q + j
            typename Mma<T>::frag af;
            if constexpr (H16) {
                const s16x4 lo = lds_read_tr16(Xc + tb + offA[0]), hi = lds_read_tr16(Xc + tb + offA[1]);
                vec<short, 8> t;
#pragma unroll
                for (int j = 0; j < 4; ++j) { t[j] = lo[j]; t[4 + j] = hi[j]; }
                af = __builtin_bit_cast(typename Mma<T>::frag, t);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) af[j] = offA[j] >= 0 ? Xc[tb + l15 + offA[j]] : from_f<T>(0.f);
            }
            const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (MODE == SX_FWD_APPLY) {
                // swapped operands: rows = channels 4q + r, column = voxel l15 -> 4 consecutive channels of one voxel per lane
                const f32x4 pa = Mma<T>::run(b3, af, z4);
                f32x4 pb = z4;
                if (nbr == 2) pb = Mma<T>::run(b1, af, z4);
                vec<T, 4> o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ra = to_f(from_f<T>(pa[r] + bw3[r]));
                    float yv = fmaxf(fmaf(sw3[r], ra, tw3[r]), 0.f);
                    if (nbr == 2) {
                        const float rb = to_f(from_f<T>(pb[r] + bw1[r]));
                        yv += fmaxf(fmaf(sw1[r], rb, tw1[r]), 0.f);
                    }
                    o[r] = from_f<T>(yv);
                }
                const int x = x0 + l15;
                if (FULL || (row_ok && x < a.W))
                    *(vec<T, 4>*)((T*)a.out + ((((long long)n * a.D + z) * a.H + y) * a.W + x) * SX_C + 4 * q) = o;
                continue;
            }
            // plain operands: rows = voxels 4q + r, column = channel l15
            const f32x4 pa = Mma<T>::run(af, b3, z4);
            f32x4 pb = z4;
            if (nbr == 2) pb = Mma<T>::run(af, b1, z4);
            float ra[4], rb[4];
            bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ra[r] = to_f(from_f<T>(pa[r] + bs3));
                rb[r] = to_f(from_f<T>(pb[r] + bs1));
                ok[r] = FULL || (row_ok && (x0 + 4 * q + r) < a.W);
            }
            if (MODE == SX_FWD_STATS) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ok[r]) {
                        acc_s[0][0] += ra[r]; acc_s[0][1] = fmaf(ra[r], ra[r], acc_s[0][1]);
                        acc_s[1][0] += rb[r]; acc_s[1][1] = fmaf(rb[r], rb[r], acc_s[1][1]);
                    }
                continue;
            }
            // sum of the gradient sources in the same layout: one transposing read per source
            float dy[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sidx = 0; sidx < NDY; ++sidx) {
                const T* tile = DyB + sidx * B::V * SX_C + m * 16 * SX_C;
                if constexpr (H16) {
                    const s16x4 t = lds_read_tr16(tile + (4 * q + (l15 >> 2)) * SX_C + 4 * (l15 & 3));
                    const vec<T, 4> tv = __builtin_bit_cast(vec<T, 4>, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dy[r] += to_f(tv[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dy[r] += to_f(tile[(4 * q + r) * SX_C + l15]);
                }
            }
            float da[4], db[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                da[r] = (fmaf(sc3, ra[r], sh3) > 0.f) ? dy[r] : 0.f;
                db[r] = (nbr == 2 && fmaf(sc1, rb[r], sh1) > 0.f) ? dy[r] : 0.f;
            }
            if (MODE == SX_BWD_REDUCE) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ok[r]) {
                        acc_s[0][0] += da[r]; acc_s[0][1] = fmaf(da[r], ra[r], acc_s[0][1]);
                        acc_s[1][0] += db[r]; acc_s[1][1] = fmaf(db[r], rb[r], acc_s[1][1]);
                    }
                continue;
            }
            // SX_BWD_APPLY: d(raw) of this tile (rounded to the run dtype like the tensor it replaces) is the B operand of a K = 16
            // step (k = voxel 4q + r, n = channel l15); A = transposed im2col rows (m = k slot, k = voxel)
            typename Mma16<T>::frag d3, d1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d3[r] = from_f<T>(ok[r] ? fmaf(cf3[0], da[r], fmaf(cf3[1], ra[r], cf3[2])) : 0.f);
                d1[r] = from_f<T>(ok[r] ? fmaf(cf1[0], db[r], fmaf(cf1[1], rb[r], cf1[2])) : 0.f);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                typename Mma16<T>::frag xa;
                if constexpr (H16) {
                    xa = *(const typename Mma16<T>::frag*)(Xc + tb + offW[mt]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xa[j] = offW[mt] >= 0 ? Xc[tb + offW[mt] + j] : from_f<T>(0.f);
                }
                accW3[mt] = Mma16<T>::run(xa, d3, accW3[mt]);
                if (mt == 0 && nbr == 2) accW1 = Mma16<T>::run(xa, d1, accW1);
            }
        }
        };
        if (z0 + B::TD <= a.D && y0 + B::TH <= a.H && x0 + 16 <= a.W) tiles(std::true_type{});
        else tiles(std::false_type{});
    }
    if (cur_n >= 0) flush(cur_n);
    if (MODE == SX_BWD_APPLY) {
        // weight-gradient tiles of the four waves -> one partial per workgroup: [wg][branch 3: 16 x K | branch 1: 16 x Cimg]
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wred[(wv * 3 + 0) * 256 + (4 * q + r) * 16 + l15] = accW3[0][r];
            wred[(wv * 3 + 1) * 256 + (4 * q + r) * 16 + l15] = accW3[1][r];
            wred[(wv * 3 + 2) * 256 + (4 * q + r) * 16 + l15] = accW1[r];
        }
        __syncthreads();
        const int K1 = nbr == 2 ? Cimg : 0, per = SX_C * (K + K1);
        float* dst = a.partial + (long long)blockIdx.x * per;
        for (int i = tid; i < 3 * 256; i += 256) {
            const int t = i / 256, e = i % 256, kk = (t == 1 ? 16 : 0) + e / 16, co = e % 16;      // tile t row e/16 = k slot
            const float s = wred[(0 * 3 + t) * 256 + e] + wred[(1 * 3 + t) * 256 + e] + wred[(2 * 3 + t) * 256 + e] + wred[(3 * 3 + t) * 256 + e];
            if (t < 2) { if (kk < K) dst[co * K + kk] = s; }
            else if (kk >= kc0 && kk < kc0 + K1) dst[SX_C * K + co * K1 + (kk - kc0)] = s;
        }
    }
}

// dw3[co][ci][tap] += sum_wg partial[wg][co][tap * Cimg + ci]; dw1[co][ci] likewise (PyTorch weight layouts)
__global__ __launch_bounds__(256) void stemx_wgrad_reduce_kernel(const float* partial, float* dw3, float* dw1, int K, int K1, int Cimg, int ntap, int nwg) {
    const int per = SX_C * (K + K1);
    const int b0 = blockIdx.y * 64, b1 = b0 + 64 < nwg ? b0 + 64 : nwg;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < per; i += gridDim.x * 256) {
        float s0 = 0.f, s1 = 0.f;
        int b = b0;
        for (; b + 2 <= b1; b += 2) { s0 += partial[(long long)b * per + i]; s1 += partial[(long long)(b + 1) * per + i]; }
        if (b < b1) s0 += partial[(long long)b * per + i];
        const float tot = s0 + s1;
        if (i < SX_C * K) {
            const int co = i / K, k = i % K, tap = k / Cimg, ci = k % Cimg;
            atomicAdd(&dw3[((long long)co * Cimg + ci) * ntap + tap], tot);
        } else if (dw1) {
            const int j = i - SX_C * K;
            atomicAdd(&dw1[j], tot);                                   // [co][ci], 1^d kernel
        }
    }
}

template <class T, class B, int CIMG>
void launch_mode(const StemxArgs& a, int mode, int nwg, hipStream_t s) {
    dim3 grid(nwg), block(256);
#define SEG_SXK(M, ND) hipLaunchKernelGGL(HIP_KERNEL_NAME(stemx_kernel<T, B, M, CIMG, ND>), grid, block, 0, s, a)
    switch (mode) {
        case SX_FWD_STATS: SEG_SXK(SX_FWD_STATS, 0); break;
        case SX_FWD_APPLY: SEG_SXK(SX_FWD_APPLY, 0); break;
        case SX_BWD_REDUCE: if (a.ndy == 1) SEG_SXK(SX_BWD_REDUCE, 1); else if (a.ndy == 2) SEG_SXK(SX_BWD_REDUCE, 2); else SEG_SXK(SX_BWD_REDUCE, 3); break;
        default: if (a.ndy == 1) SEG_SXK(SX_BWD_APPLY, 1); else if (a.ndy == 2) SEG_SXK(SX_BWD_APPLY, 2); else SEG_SXK(SX_BWD_APPLY, 3); break;
    }
#undef SEG_SXK
}

// the forward apply pass (no gradient-source tiles in LDS, no accumulators) on a box of its own choice
template <class T, class B, int CIMG>
void launch_fwd_mode(const StemxArgs& a, int mode, hipStream_t s) {
    const long long nb = (long long)a.N * ((a.D + B::TD - 1) / B::TD) * ((a.H + B::TH - 1) / B::TH) * ((a.W + 15) / 16);
    dim3 grid((unsigned)(nb < 2048 ? nb : 2048)), block(256);
    (void)mode;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(stemx_kernel<T, B, SX_FWD_APPLY, CIMG, 0>), grid, block, 0, s, a);
}

inline long long sx_boxes(int ndim, int N, int D, int H, int W) {
    const int td = ndim == 3 ? 2 : 1, th = ndim == 3 ? 8 : 16;
    return (long long)N * ((D + td - 1) / td) * ((H + th - 1) / th) * ((W + 15) / 16);
}

}  // namespace

int stemx_workgroups(int ndim, int N, int D, int H, int W) {
    const long long nb = sx_boxes(ndim, N, ndim == 3 ? D : 1, H, W);
    static const int cap = 2048;
    return (int)(nb < cap ? nb : cap);
}
size_t stemx_partial_bytes(int ndim, int N, int D, int H, int W, int Cimg) {
    const int K = (ndim == 3 ? 27 : 9) * Cimg;
    return (size_t)stemx_workgroups(ndim, N, D, H, W) * SX_C * (K + Cimg) * sizeof(float);
}

void launch_stemx(const seg_stemx_args& a0, int mode, int ndim, int dtype, float* dw3, float* dw1, hipStream_t s) {
    StemxArgs a = a0;
    if (ndim != 3) a.D = 1;
    const int nwg = stemx_workgroups(ndim, a.N, a.D, a.H, a.W);
    if (ndim == 3 && mode == SX_FWD_APPLY) {
        // 4 x 8 x 16 boxes for the forward apply pass: 2.1 halo elements staged per voxel instead of 2.8, half the per-box bookkeeping (MI355X, 4 x 96^3: 46 -> 36 us;
        // the statistics pass keeps the small box - its accumulators take 131 + 36 registers on the large one, three workgroups per SIMD instead of six, 45 us either way)
        if (dtype == DT_F32) launch_fwd_mode<float, SBox<4, 8, 3>, 1>(a, mode, s);
        else if (dtype == DT_F16) launch_fwd_mode<f16, SBox<4, 8, 3>, 1>(a, mode, s);
        else launch_fwd_mode<bf16, SBox<4, 8, 3>, 1>(a, mode, s);
        return;
    }
#define SEG_SX(T) do { if (ndim == 3) launch_mode<T, SBox<2, 8, 3>, 1>(a, mode, nwg, s);                    \
                       else if (a.Cimg == 1) launch_mode<T, SBox<1, 16, 1>, 1>(a, mode, nwg, s);             \
                       else if (a.Cimg == 2) launch_mode<T, SBox<1, 16, 1>, 2>(a, mode, nwg, s);             \
                       else launch_mode<T, SBox<1, 16, 1>, 3>(a, mode, nwg, s); } while (0)
    if (dtype == DT_F32) SEG_SX(float);
    else if (dtype == DT_F16) SEG_SX(f16);
    else SEG_SX(bf16);
#undef SEG_SX
    if (mode == SX_BWD_APPLY) {
        const int ntap = ndim == 3 ? 27 : 9, K = ntap * a.Cimg, K1 = a.w1 ? a.Cimg : 0;
        const int per = SX_C * (K + K1);
        hipLaunchKernelGGL(stemx_wgrad_reduce_kernel, dim3((per + 255) / 256, (nwg + 63) / 64), dim3(256), 0, s, (const float*)a.partial, dw3, dw1, K, K1,
                           a.Cimg, ntap, nwg);
    }
}

}  // namespace seg
