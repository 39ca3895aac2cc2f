// Register-blocked LDS halo-tile kernel for the 3^d / 3^2 stride-1 pad-1 convolutions on 16-bit tensors with
// Cin % 32 == 0 (the LUConv / _block layers from the second level down, networks/VNet3d.py:8, networks/Unet3d.py:66-80)
// — forward and data-gradient (same kernel, flipped fragment-major weights).
//
// What it changes against conv3_kernel (conv3.hip), following the round-1 profile (VERDICT r01 "What's weak"):
//   * ONE workgroup computes up to 64 / 128 output channels of a box from one staged halo (conv3_kernel re-staged the
//     halo once per 16 output channels: 2.8x HBM/L2 traffic, 4 - 16x the staging latency chains per box);
//   * every wave owns a TM x TN grid of 16x16 output tiles, so an A fragment read from LDS feeds TN MFMAs and a B
//     fragment TM MFMAs: (TM + TN) / (TM * TN) KB of operand traffic per MFMA instead of 1.33 KB;
//   * the halo is staged by direct global -> LDS copies (buffer_load_dwordx4 ... lds, common.h dma16): no VGPR round trip,
//     no ds_write pass, zero padding from the buffer's out-of-range rule; the conflict-free XOR swizzle of conv3.hip is
//     kept by choosing which 16-B channel piece each lane fetches;
//   * weights are packed FRAGMENT-MAJOR ([chunk][tap][16-channel tile][lane][8]): a wave's B fragment is one contiguous
//     1 KB line, loaded straight from L2 into a register ring PF steps ahead — no LDS slab, no barrier in the tap loop.
// All 32-channel chunks of a group (NRES resident chunks) are staged before the tap loops start; staging latency is
// hidden by the other workgroup(s) resident on the CU.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace seg {
namespace c3x {

__device__ __forceinline__ int halo_swz_x(int row, int x) { return ((row & 1) << 1) ^ (((x >> 2) & 1) * 3); }   // = conv3.hip halo_swz

// Box of TD x TH x TW output voxels cut into 16-voxel M tiles of TY x TX (1x16, 2x8 or 4x4) voxels; voxel order is tile-major.
template <int TD_, int TH_, int TW_, int KD_, int TX_> struct XBox {
    static constexpr int TD = TD_, TH = TH_, TW = TW_, KD = KD_, TX = TX_, TY = 16 / TX_;
    static_assert(TX_ == 16 || TX_ == 8 || TX_ == 4, "tile width");
    static_assert(TW_ % TX_ == 0 && TH_ % (16 / TX_) == 0, "box must be cut into whole tiles");
    static constexpr int V = TD * TH * TW, NTILE = V / 16;
    static constexpr int HD = TD + KD - 1, HH = TH + 2, HW = TW + 2, HWP = (HW + 3) / 4 * 4;   // row pitch: multiple of 4 voxels (swizzle)
    static constexpr int PD = (KD - 1) / 2, NTAP = KD * 9;
    static constexpr int ROWS = HD * HH;
    static constexpr int GRAN = ROWS * HWP * 4;              // 16-B granules of one 32-channel chunk image
    static constexpr int NINSTR = (GRAN + 63) / 64;          // wave-instructions (1 KB each) per chunk
    static constexpr int CHUNK_ELEMS = NINSTR * 64 * 8;      // 16-bit elements of one resident chunk
    static __device__ __forceinline__ void vox(int v, int& vz, int& vy, int& vx) {
        constexpr int ntx = TW / TX, nty = TH / TY;
        const int t = v >> 4, l = v & 15;
        vx = (t % ntx) * TX + (l % TX);
        vy = ((t / ntx) % nty) * TY + l / TX;
        vz = t / (ntx * nty);
    }
};

struct Conv3xArgs {
    const void* in0; const void* in1; int C0;     // in1: second source of a virtual channel concat (channels C0..Cin-1), or null
    const void* w;                                // fragment-major weights [Cin/32][taps][Cout/16][64 lanes][8]
    const float* bias; void* out; double* stats; int stat_rep;   // stat_rep: replicas of `stats` the workgroups spread over (<= STAT_REP)
    int N, D, H, W, Cin, Cout;
    int remap;                                    // 1: XCD-aware box order (grid.x rounded up to a multiple of 8)
#ifdef SEG_C3X_TRACE
    unsigned long long* trace;                    // diagnostic build (tools/trace_conv3x.py): 8 wall_clock64 stamps per workgroup - 0 start, 4 copies issued, 5 copies landed,
                                                  // 1 barrier passed, 2 tap loops done, 6 tile stored, 7 statistics folded per wave, 3 end
#endif
};
#ifdef SEG_C3X_TRACE
#define SEG_C3XT(k) do { if (a.trace && threadIdx.x == 0) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define SEG_C3XT(k)
#endif

// Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8, MI355X_MICROARCH.md), each with its own L2.  With the
// natural order the x / y / z neighbours of a box - which share most of its halo - sit on other XCDs and every halo is fetched
// from HBM / MALL once per XCD that touches it.  Remapped, XCD k walks the contiguous box range [k*per, (k+1)*per): neighbours
// share an L2.  Returns -1 for the padding blocks of the rounded-up grid.
__device__ __forceinline__ int c3x_box_of_block(int b, int nbox, int remap) {
    if (!remap) return b;
    const int per = (nbox + 7) >> 3;
    const int box = (b & 7) * per + (b >> 3);
    return ((b >> 3) < per && box < nbox) ? box : -1;
}

// Shared epilogue of the conv3x kernels.  The tap loops multiply with SWAPPED operands (D^T = W x X^T: the weight fragment is the MFMA's A
// operand, the halo fragment its B operand - the two have the same per-lane layout), so
//     acc[m][j][r] = out[voxel (wm*TM + m)*16 + l15][channel (wn*TN + j)*16 + 4q + r]:
// a lane holds FOUR CONSECUTIVE CHANNELS of one voxel.  Lanes q / q^1 exchange one 4-channel piece (tiles in pairs) and every lane stores
// 16 B: a wave instruction covers whole 64-B voxel rows of 16 voxels.  No LDS transpose, no barrier in front of the stores (rounds 1-2 wrote the
// tile to LDS with 2-byte stores, synchronised and read it back; every barrier in this epilogue is worth ~1-2 us per workgroup round,
// profiles/r03_graph_stats_ab.log).  GroupNorm statistics: per-lane sums over the TM voxels of its 4 x TN channels (values as stored), a
// butterfly over the 16 voxel lanes, one LDS slot per wave (`red`: WM * BN * 2 floats, its own array), one barrier, fp64 atomics.
// bias_lds: the BN bias values of this workgroup's output channels, staged in LDS by the caller during its prologue (or null: read from global memory here - eight
// dependent L2 round trips in front of the first store)
template <class T, class B, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void c3x_epilogue(f32x4 (&acc)[TM][TN], float* red, const Conv3xArgs& a, int n, int x0, int y0, int z0, int co0,
                                             const float* bias_lds = nullptr, bool stamp = true) {
    (void)stamp;                                           // (trace builds: whether this call records its phase stamps)
    constexpr int BN = WN * TN * 16;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int wm = wv % WM, wn = wv / WM;
    T* out = (T*)a.out;
    float bs[TN][4], cs[TN][4], css[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bs[j][r] = bias_lds ? bias_lds[(wn * TN + j) * 16 + 4 * q + r] : (a.bias ? a.bias[co0 + (wn * TN + j) * 16 + 4 * q + r] : 0.f);
            cs[j][r] = 0.f; css[j][r] = 0.f;
        }
    const bool odd = q & 1;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        int vz, vy, vx;
        B::vox((wm * TM + m) * 16 + l15, vz, vy, vx);
        const int x = x0 + vx, y = y0 + vy, z = z0 + vz;
        const bool ok = x < a.W && y < a.H && z < a.D;
        T* row = out + ((((long long)n * a.D + z) * a.H + y) * a.W + x) * a.Cout + co0 + wn * TN * 16;
        vec<T, 4> o4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T tv = from_f<T>(acc[m][j][r] + bs[j][r]);
                o4[j][r] = tv;
                if (a.stats && ok) { const float f = to_f(tv); cs[j][r] += f; css[j][r] = fmaf(f, f, css[j][r]); }
            }
        if (TN % 2 == 0) {
#pragma unroll
            for (int jp = 0; jp < TN / 2; ++jp) {
                // even q keeps tile A = 2jp (its channels 4q .. 4q+3) and receives A's next four from q + 1; odd q keeps tile B likewise
                const vec<T, 4> mine = odd ? o4[2 * jp + 1] : o4[2 * jp], send = odd ? o4[2 * jp] : o4[2 * jp + 1];
                int sw[2], rw[2];
                __builtin_memcpy(sw, &send, 8);
                rw[0] = __shfl_xor(sw[0], 16); rw[1] = __shfl_xor(sw[1], 16);
                vec<T, 4> recv;
                __builtin_memcpy(&recv, rw, 8);
                vec<T, 8> w8;
#pragma unroll
                for (int r = 0; r < 4; ++r) { w8[r] = odd ? recv[r] : mine[r]; w8[4 + r] = odd ? mine[r] : recv[r]; }
                if (ok) store8(row + (odd ? 2 * jp + 1 : 2 * jp) * 16 + (q >> 1) * 8, w8);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (ok) *(vec<T, 4>*)(row + j * 16 + 4 * q) = o4[j];
        }
    }
    if (stamp) SEG_C3XT(6);
    if (a.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u = row_sum16(cs[j][r]), v = row_sum16(css[j][r]);      // over the 16 voxel lanes of this (q, r) channel
                if (l15 == 0) { const int col = (wn * TN + j) * 16 + 4 * q + r; red[(wm * BN + col) * 2] = u; red[(wm * BN + col) * 2 + 1] = v; }
            }
        if (stamp) SEG_C3XT(7);
        __syncthreads();
        if (tid < BN) {
            double ts = 0.0, tss = 0.0;
#pragma unroll
            for (int k = 0; k < WM; ++k) { ts += (double)red[(k * BN + tid) * 2]; tss += (double)red[(k * BN + tid) * 2 + 1]; }
            double* dst = a.stats + ((long long)(blockIdx.x % a.stat_rep) * a.N * a.Cout + (long long)n * a.Cout + co0 + tid) * 2;
            atomicAdd(dst, ts);
            atomicAdd(dst + 1, tss);
        }
    }
}

template <class T, class B, int TM, int TN, int WM, int WN, int NRES, int PF, int OCC>
__global__ __launch_bounds__(256, OCC) void conv3x_kernel(Conv3xArgs a) {
    SEG_C3XT(0);
    static_assert(sizeof(T) == 2, "16-bit run dtypes only");
    static_assert(WM * WN == 4 && WM * TM == B::NTILE, "wave grid must cover the box");
    static_assert(B::NTAP % (PF + 1) == 0, "register ring must divide the tap count");
    constexpr int BN = WN * TN * 16;
    constexpr int XS_ELEMS = NRES * B::CHUNK_ELEMS;
    // resident chunk images; the epilogue stores straight from the accumulators and only needs the small statistics exchange array
    __shared__ __attribute__((aligned(16))) T Xs[XS_ELEMS];
    __shared__ float red_s[WM * BN * 2];
    __shared__ __attribute__((aligned(16))) float bias_s[BN];
    constexpr int NI = (B::NINSTR + 3) / 4;               // copy instructions per wave and chunk

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int wm = wv % WM, wn = wv / WM;
    // box position
    const int nbx = (a.W + B::TW - 1) / B::TW, nby = (a.H + B::TH - 1) / B::TH, nbz = (a.D + B::TD - 1) / B::TD;
    int bb = c3x_box_of_block(blockIdx.x, a.N * nbz * nby * nbx, a.remap);
    if (bb < 0) return;
    const int x0 = (bb % nbx) * B::TW; bb /= nbx;
    const int y0 = (bb % nby) * B::TH; bb /= nby;
    const int z0 = (bb % nbz) * B::TD;
    const int n = bb / nbz;
    const int co0 = blockIdx.y * BN;
    const int NT_total = a.Cout >> 4;

    // ---- per-lane source of every granule this lane copies (the same for every chunk): voxel index inside the sample * 4
    //      + channel piece, or -1 for padding
    int src[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = u * 4 + wv;                          // instruction index inside the chunk image (wave-uniform)
        const int g = i * 64 + lane;
        const int row = g / (B::HWP * 4), rem = g % (B::HWP * 4);
        const int hx = rem >> 2, slot = rem & 3;
        const int hz = row / B::HH, hy = row % B::HH;
        const int z = z0 + hz - B::PD, y = y0 + hy - 1, x = x0 + hx - 1;
        const bool ok = i < B::NINSTR && row < B::ROWS && hx < B::HW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H &&
                        (unsigned)x < (unsigned)a.W;
        src[u] = ok ? ((((z * a.H) + y) * a.W + x) << 2) + (slot ^ halo_swz_x(row, hx)) : -1;
    }
    const long long vol = (long long)a.D * a.H * a.W;
    const int C1 = a.Cin - a.C0;
    const i32x4 r0 = make_rsrc((const T*)a.in0 + (long long)n * vol * a.C0, (unsigned)(vol * a.C0 * 2));
    const i32x4 r1 = a.in1 ? make_rsrc((const T*)a.in1 + (long long)n * vol * C1, (unsigned)(vol * C1 * 2)) : r0;
    const bool straddle = (a.C0 & 31) != 0;                // a 32-channel chunk may take pieces from both concat sources

    auto issue_chunk = [&](int cc, int buf) {
        T* dst = Xs + buf * B::CHUNK_ELEMS;
        const int ch0 = cc * 32;
        if (!straddle) {
            const bool second = ch0 >= a.C0;
            const i32x4 rs = second ? r1 : r0;
            const unsigned rowb = (unsigned)(second ? C1 : a.C0) * 2u, cho = (unsigned)(second ? ch0 - a.C0 : ch0) * 2u;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = u * 4 + wv;
                if (i < B::NINSTR) {
                    const unsigned off = src[u] >= 0 ? (unsigned)(src[u] >> 2) * rowb + cho + (unsigned)(src[u] & 3) * 16u : DMA_OOB;
                    dma16(rs, dst + i * 512, off);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                const int i = u * 4 + wv;
                if (i < B::NINSTR) {
                    const int ch = ch0 + (src[u] & 3) * 8;
                    const bool second = src[u] >= 0 && ch >= a.C0;
                    // lanes of one instruction may read different sources: two exec-masked copies into the same 1 KB line
                    if (!second) {
                        const unsigned off = src[u] >= 0 ? (unsigned)(src[u] >> 2) * (unsigned)(a.C0 * 2) + (unsigned)ch * 2u : DMA_OOB;
                        dma16(r0, dst + i * 512, off);
                    } else {
                        const unsigned off = (unsigned)(src[u] >> 2) * (unsigned)(C1 * 2) + (unsigned)(ch - a.C0) * 2u;
                        dma16(r1, dst + i * 512, off);
                    }
                }
            }
        }
    };

    // ---- A-fragment addressing: element offset of this lane's 16-B piece per M tile and kw shift (tap offsets are
    //      compile-time immediates; an odd tap row flips the swizzle bit)
    int ab[TM][3];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        int vz, vy, vx;
        B::vox((wm * TM + m) * 16 + l15, vz, vy, vx);
        const int row = vz * B::HH + vy;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) ab[m][kw] = (row * B::HWP + vx) * 32 + ((q ^ halo_swz_x(row, vx + kw)) << 3);
    }

    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // B fragments: buffer loads with a per-lane base offset and a wave-uniform step offset (no address VALU in the tap loop;
    // the PF steps fetched past the last tap read zeros)
    const unsigned wstep = (unsigned)NT_total * 1024u;     // bytes between consecutive (chunk, tap) steps
    const i32x4 wr = make_rsrc(a.w, (unsigned)(a.Cin >> 5) * B::NTAP * wstep);
    const unsigned wl = ((unsigned)(blockIdx.y * (BN / 16) + wn * TN) * 64u + lane) * 16u;

    const int nchunk = a.Cin >> 5;
    for (int g0 = 0; g0 < nchunk; g0 += NRES) {
        const int nres = nchunk - g0 < NRES ? nchunk - g0 : NRES;
        if (g0) __syncthreads();                           // every wave is done reading the previous group
#pragma unroll
        for (int b = 0; b < NRES; ++b)
            if (b < nres) issue_chunk(g0 + b, b);
        // B fragments of the first PF steps travel together with the halo
        typename Mma<T>::frag bq[PF + 1][TN];
        const unsigned wg = (unsigned)g0 * B::NTAP * wstep;
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[s][j] = buffer_load8<T>(wr, wl + j * 1024, wg + s * wstep);
        // the bias of this workgroup's channels travels with the first copies and is parked in LDS for the epilogue
        float bias_v = 0.f;
        if (g0 == 0 && a.bias && threadIdx.x < BN) bias_v = a.bias[co0 + threadIdx.x];
        if (g0 == 0) SEG_C3XT(4);
        wait_vmem();
        if (g0 == 0 && threadIdx.x < BN) bias_s[threadIdx.x] = bias_v;
        if (g0 == 0) SEG_C3XT(5);
        __syncthreads();
        if (g0 == 0) SEG_C3XT(1);
        unsigned wo = wg + PF * wstep;                     // byte offset of the step being prefetched
        for (int b = 0; b < nres; ++b) {
            // Software pipeline, written out: while the MFMAs of tap t run, the A fragments of tap t + 1 travel from LDS and
            // the B fragments of step t + PF from L2.  Every tap offset is an immediate of the ds_read; the swizzle flip of an
            // odd tap row is folded into the per-lane base (ab ^ 16).  sched_group_barrier pins that order in the emitted code.
            const T* Xc = Xs + b * B::CHUNK_ELEMS;
            typename Mma<T>::frag af[2][TM];
#pragma unroll
            for (int m = 0; m < TM; ++m) af[0][m] = load8(&Xc[ab[m][0]]);
            __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);               // tap 0's fragments first (they would otherwise fill tap 1's slot)
#pragma unroll
            for (int t = 0; t < B::NTAP; ++t) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bq[(t + PF) % (PF + 1)][j] = buffer_load8<T>(wr, wl + j * 1024, wo);
                wo += wstep;
                if (t + 1 < B::NTAP) {
                    const int t1 = t + 1, kw = t1 % 3, srow = (t1 / 9) * B::HH + (t1 / 3) % 3;
                    const int toff = (srow * B::HWP + kw) * 32, flip = (srow & 1) << 4;
#pragma unroll
                    for (int m = 0; m < TM; ++m) af[t1 & 1][m] = load8(&Xc[toff + (ab[m][kw] ^ flip)]);
                }
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[m][j] = Mma<T>::run(bq[t % (PF + 1)][j], af[t & 1][m], acc[m][j]);
                __builtin_amdgcn_sched_group_barrier(0x020, TN, 0);           // VMEM reads: B fragments of step t + PF
                if (t + 1 < B::NTAP) __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);   // DS reads: A fragments of tap t + 1
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);      // the MFMAs of tap t
            }
        }
    }
    SEG_C3XT(2);
    c3x_epilogue<T, B, TM, TN, WM, WN>(acc, red_s, a, n, x0, y0, z0, co0, bias_s);      // no barrier: nothing of the halo buffer is reused
    SEG_C3XT(3);
}


template <class T, class B, int TM, int TN, int WM, int WN, int NRES, int PF, int OCC>
void launch_cfg(const Conv3xArgs& a, hipStream_t s) {
    constexpr int BN = WN * TN * 16;
    const long long nbox = (long long)a.N * ((a.D + B::TD - 1) / B::TD) * ((a.H + B::TH - 1) / B::TH) * ((a.W + B::TW - 1) / B::TW);
    dim3 grid(a.remap ? (unsigned)((nbox + 7) / 8 * 8) : (unsigned)nbox, a.Cout / BN);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x_kernel<T, B, TM, TN, WM, WN, NRES, PF, OCC>), grid, dim3(256), 0, s, a);
}



// ------------------------------------------------------------------------------------------------
// Cin == 16 (the full-resolution LUConv / _block layers: networks/VNet3d.py:117-125 up_tr32.ops, networks/Unet3d.py enc1 / dec1).
// One 16x16x32 MFMA step multiplies TWO taps: lanes q = 0, 1 hold the 16 channels of tap 2s, lanes q = 2, 3 those of tap 2s + 1
// (K order of the weights is (tap, ci) flat; seg_pack_desc.frag = 2).  Halo rows are 32 B; 16 consecutive voxels of an x row
// are conflict-free for ds_read_b128 without a swizzle.  The per-lane difference between the two taps of a step takes three
// values only (next kw; next kh row; next kd plane), so three per-lane bases keep every tap offset an immediate.  These layers
// are HBM-bound (Cin = Cout = 16: 226 MB per launch at 4 x 96^3), MFMA time at 40 % of peak is about the same as the stream time.
// ------------------------------------------------------------------------------------------------
template <class T, class B, int TM, int TN, int PF, int OCC>
__global__ __launch_bounds__(256, OCC) void conv3x16_kernel(Conv3xArgs a) {
    static_assert(sizeof(T) == 2, "16-bit run dtypes only");
    static_assert(B::TX == 16, "x rows of 16 voxels");
    static_assert(4 * TM == B::NTILE, "four waves cover the box");
    constexpr int WM = 4, WN = 1, BN = TN * 16;
    constexpr int NSTEP = (B::NTAP + 1) / 2;
    static_assert(NSTEP >= PF + 1, "ring deeper than the loop");
    constexpr int GRAN = B::ROWS * B::HWP * 2, NINSTR = (GRAN + 63) / 64, XS_ELEMS = NINSTR * 64 * 8;
    __shared__ __attribute__((aligned(16))) T Xs[XS_ELEMS];
    __shared__ float red_s[WM * BN * 2];
    __shared__ __attribute__((aligned(16))) float bias_s[BN];
    constexpr int NI = (NINSTR + 3) / 4;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int nbx = (a.W + B::TW - 1) / B::TW, nby = (a.H + B::TH - 1) / B::TH, nbz = (a.D + B::TD - 1) / B::TD;
    int bb = c3x_box_of_block(blockIdx.x, a.N * nbz * nby * nbx, a.remap);
    if (bb < 0) return;
    const int x0 = (bb % nbx) * B::TW; bb /= nbx;
    const int y0 = (bb % nby) * B::TH; bb /= nby;
    const int z0 = (bb % nbz) * B::TD;
    const int n = bb / nbz;
    const int co0 = blockIdx.y * BN;
    const long long vol = (long long)a.D * a.H * a.W;

    // ---- halo: [row][HWP voxels][16 channels], two 16-B granules per voxel, copied directly (zeros outside the volume)
    const i32x4 r0 = make_rsrc((const T*)a.in0 + (long long)n * vol * 16, (unsigned)(vol * 32));
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = u * 4 + wv;
        if (i < NINSTR) {
            const int g = i * 64 + lane;
            const int row = g / (B::HWP * 2), rem = g % (B::HWP * 2);
            const int hx = rem >> 1, piece = rem & 1;
            const int hz = row / B::HH, hy = row % B::HH;
            const int z = z0 + hz - B::PD, y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = row < B::ROWS && hx < B::HW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            dma16(r0, Xs + i * 512, ok ? (unsigned)(((z * a.H + y) * a.W + x) * 32 + piece * 16) : DMA_OOB);
        }
    }
    // ---- B fragments through the register ring (weights: [step][Cout/16][64 lanes][8])
    const unsigned wstep = (unsigned)(a.Cout >> 4) * 1024u;
    const i32x4 wr = make_rsrc(a.w, (unsigned)NSTEP * wstep);
    const unsigned wl = ((unsigned)(blockIdx.y * TN) * 64u + lane) * 16u;
    typename Mma<T>::frag bq[PF + 1][TN];
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[s][j] = buffer_load8<T>(wr, wl + j * 1024, s * wstep);

    // ---- A addressing: lanes q < 2 read tap 2s, lanes q >= 2 tap 2s + 1; the second tap's extra offset is one of three values
    constexpr int D_KW = 16, D_KH = (B::HWP - 2) * 16, D_KD = ((B::HH - 2) * B::HWP - 2) * 16;    // elements
    const int hi = q >> 1, piece = q & 1;
    int ab[TM][3];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        int vz, vy, vx;
        B::vox((wv * TM + m) * 16 + l15, vz, vy, vx);
        const int base = ((vz * B::HH + vy) * B::HWP + vx) * 16 + piece * 8;
        ab[m][0] = base + (hi ? D_KW : 0);
        ab[m][1] = base + (hi ? D_KH : 0);
        ab[m][2] = base + (hi ? D_KD : 0);
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float bias_v = (a.bias && tid < BN) ? a.bias[co0 + tid] : 0.f;      // parked in LDS for the epilogue (see conv3x_kernel)
    wait_vmem();
    if (tid < BN) bias_s[tid] = bias_v;
    __syncthreads();

    unsigned wo = PF * wstep;
    auto tap_off = [](int t) { return ((t / 9) * B::HH + (t / 3) % 3) * B::HWP + t % 3; };       // halo voxels
    typename Mma<T>::frag af[2][TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) af[0][m] = load8(&Xs[ab[m][0] + tap_off(0) * 16]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[(s + PF) % (PF + 1)][j] = buffer_load8<T>(wr, wl + j * 1024, wo);
        wo += wstep;
        if (s + 1 < NSTEP) {
            const int t0 = 2 * (s + 1), t1 = t0 + 1;
            // the second tap of the step: +1 in kw, or the start of the next kh row / kd plane; past the last tap (zero weights)
            // the lanes re-read the first tap (finite data)
            const int d = t1 >= B::NTAP ? -1 : (tap_off(t1) - tap_off(t0) == 1 ? 0 : (t1 % 9 == 0 ? 2 : 1));
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int basev = d < 0 ? ab[m][0] - (hi ? D_KW : 0) : ab[m][d];
                af[(s + 1) & 1][m] = load8(&Xs[basev + tap_off(t0) * 16]);
            }
        }
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[m][j] = Mma<T>::run(bq[s % (PF + 1)][j], af[s & 1][m], acc[m][j]);
        __builtin_amdgcn_sched_group_barrier(0x020, TN, 0);
        if (s + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, TM, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
    c3x_epilogue<T, B, TM, TN, WM, WN>(acc, red_s, a, n, x0, y0, z0, co0, bias_s);
}



// ------------------------------------------------------------------------------------------------
// Cin == 16 with the halo fragments REUSED ACROSS THE kh TAPS (3-D; weights packed with seg_pack_desc.frag = 3).
// conv3x16_kernel reads one 1 KB fragment from LDS per MFMA (one 16-channel output tile: nothing to share a fragment with) and the LDS pipe is its
// busiest unit (SQ_ACTIVE_INST_LDS = 0.73 of the busy cycles at 4 x 96^3, MFMA 0.26: profiles/r06_mfma_util_per_kernel.json).  A wave's TM tiles are
// TM consecutive y rows of ONE z plane, and the fragment of halo row r (16 x voxels, two taps that differ in kd / kw only) is what tile y = r - kh
// needs for kh = 0, 1, 2 - the same lanes, no shift.  So the taps are paired WITHIN a kh row into five steps
//     0..2: (kd, kw = 0 | kw = 1)     3: (kd = 0 | kd = 1, kw = 2)     4: (kd = 2, kw = 2 | zero weights)
// = 15 MFMA steps per tile instead of 14, and every fragment read feeds up to three MFMAs on three accumulators:
// 5 x (TM + 2) reads per wave instead of 14 x TM (TM = 8: 50 against 112).  The 15 weight fragments stay in registers for the whole box.
// ------------------------------------------------------------------------------------------------
template <class T, class B, int TM, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void conv3x16r_kernel(Conv3xArgs a) {
    static_assert(sizeof(T) == 2, "16-bit run dtypes only");
    static_assert(B::TX == 16 && B::TW == 16 && B::KD == 3, "x rows of 16 voxels, 3-D taps");
    static_assert(4 * TM == B::NTILE && B::TH % TM == 0, "a wave covers TM consecutive rows of one plane");
    constexpr int WM = 4, BN = TN * 16, NS = 15;
    constexpr int GRAN = B::ROWS * B::HWP * 2, NINSTR = (GRAN + 63) / 64, XS_ELEMS = NINSTR * 64 * 8;
    __shared__ __attribute__((aligned(16))) T Xs[XS_ELEMS];
    __shared__ float red_s[WM * BN * 2];
    __shared__ __attribute__((aligned(16))) float bias_s[BN];
    constexpr int NI = (NINSTR + 3) / 4;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int nbx = (a.W + B::TW - 1) / B::TW, nby = (a.H + B::TH - 1) / B::TH, nbz = (a.D + B::TD - 1) / B::TD;
    int bb = c3x_box_of_block(blockIdx.x, a.N * nbz * nby * nbx, a.remap);
    if (bb < 0) return;
    const int x0 = (bb % nbx) * B::TW; bb /= nbx;
    const int y0 = (bb % nby) * B::TH; bb /= nby;
    const int z0 = (bb % nbz) * B::TD;
    const int n = bb / nbz;
    const int co0 = blockIdx.y * BN;
    const long long vol = (long long)a.D * a.H * a.W;

    // ---- halo: [row][HWP voxels][16 channels] (as conv3x16_kernel)
    const i32x4 r0 = make_rsrc((const T*)a.in0 + (long long)n * vol * 16, (unsigned)(vol * 32));
#pragma unroll
    for (int u = 0; u < NI; ++u) {
        const int i = u * 4 + wv;
        if (i < NINSTR) {
            const int g = i * 64 + lane;
            const int row = g / (B::HWP * 2), rem = g % (B::HWP * 2);
            const int hx = rem >> 1, piece = rem & 1;
            const int hz = row / B::HH, hy = row % B::HH;
            const int z = z0 + hz - B::PD, y = y0 + hy - 1, x = x0 + hx - 1;
            const bool ok = row < B::ROWS && hx < B::HW && (unsigned)z < (unsigned)a.D && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            dma16(r0, Xs + i * 512, ok ? (unsigned)(((z * a.H + y) * a.W + x) * 32 + piece * 16) : DMA_OOB);
        }
    }
    // ---- all 15 weight fragments per output tile: [step = kh * 5 + pair][Cout/16][64 lanes][8]
    const unsigned wstep = (unsigned)(a.Cout >> 4) * 1024u;
    const i32x4 wr = make_rsrc(a.w, (unsigned)NS * wstep);
    const unsigned wl = ((unsigned)(blockIdx.y * TN) * 64u + lane) * 16u;
    typename Mma<T>::frag wq[NS][TN];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j) wq[s][j] = buffer_load8<T>(wr, wl + j * 1024, s * wstep);

    // ---- fragment addressing: lanes q < 2 read the first tap of a pair, lanes q >= 2 the second (+1 in kw, +1 in kd, or the same voxel again)
    constexpr int ROWP = B::HWP * 16, PLANE = B::HH * ROWP;                 // elements per halo row / plane
    const int hi = q >> 1, piece = q & 1;
    const int t0 = wv * TM, vz = t0 / B::TH, vy0 = t0 % B::TH;
    const int base = (vz * B::HH + vy0) * ROWP + l15 * 16 + piece * 8;
    const int b_kw = base + (hi ? 16 : 0), b_kd = base + (hi ? PLANE : 0);
    auto frag_of = [&](int r, int p) {                                    // halo row r of the wave's rows (0 .. TM + 1), pair p
        const int off = r * ROWP + (p < 3 ? p * PLANE : (p == 3 ? 2 * 16 : 2 * PLANE + 2 * 16));
        return load8(&Xs[(p < 3 ? b_kw : (p == 3 ? b_kd : base)) + off]);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float bias_v = (a.bias && tid < BN) ? a.bias[co0 + tid] : 0.f;
    wait_vmem();
    if (tid < BN) bias_s[tid] = bias_v;
    __syncthreads();

    constexpr int NF = (TM + 2) * 5;
    typename Mma<T>::frag af[2];
    af[0] = frag_of(0, 0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int r = f / 5, p = f % 5;
        if (f + 1 < NF) af[(f + 1) & 1] = frag_of((f + 1) / 5, (f + 1) % 5);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int m = r - kh;
            if (m >= 0 && m < TM) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[m][j] = Mma<T>::run(wq[kh * 5 + p][j], af[f & 1], acc[m][j]);
            }
        }
    }
    c3x_epilogue<T, B, TM, TN, WM, 1>(acc, red_s, a, n, x0, y0, z0, co0, bias_s);
}

template <class T, class B, int TM, int TN, int OCC>
void launch_cfg16r(const Conv3xArgs& a, hipStream_t s) {
    const long long nbox = (long long)a.N * ((a.D + B::TD - 1) / B::TD) * ((a.H + B::TH - 1) / B::TH) * ((a.W + B::TW - 1) / B::TW);
    dim3 grid(a.remap ? (unsigned)((nbox + 7) / 8 * 8) : (unsigned)nbox, a.Cout / (TN * 16));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x16r_kernel<T, B, TM, TN, OCC>), grid, dim3(256), 0, s, a);
}

template <class T, class B, int TM, int TN, int PF, int OCC>
void launch_cfg16(const Conv3xArgs& a, hipStream_t s) {
    const long long nbox = (long long)a.N * ((a.D + B::TD - 1) / B::TD) * ((a.H + B::TH - 1) / B::TH) * ((a.W + B::TW - 1) / B::TW);
    dim3 grid(a.remap ? (unsigned)((nbox + 7) / 8 * 8) : (unsigned)nbox, a.Cout / (TN * 16));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(conv3x16_kernel<T, B, TM, TN, PF, OCC>), grid, dim3(256), 0, s, a);
}

// one translation unit per (dtype, ndim): the tap loops are fully unrolled and each instantiation takes ~10 s to compile
template <class T> bool launch_3d(int id, const Conv3xArgs& a, hipStream_t s);
template <class T> bool launch_2d(int id, const Conv3xArgs& a, hipStream_t s);

/* halo-conv tilings:                  box              TM TN WM WN NRES PF OCC */
#define SEG_C3X_3D_CONV_CASES                                                                                         \
        case 0: launch_cfg<T, XBox<2, 8, 16, 3, 16>, 4, 2, 4, 1, 1, 2, 2>(a, s); return true;                         \
        case 1: launch_cfg<T, XBox<4, 8, 16, 3, 16>, 8, 2, 4, 1, 1, 2, 1>(a, s); return true;                         \
        case 2: launch_cfg<T, XBox<2, 8, 16, 3, 16>, 8, 2, 2, 2, 2, 8, 1>(a, s); return true;                         \
        case 3: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 4, 2, 2, 2, 2, 8, 2>(a, s); return true;                           \
        case 4: launch_cfg<T, XBox<4, 8, 8, 3, 8>, 8, 2, 2, 2, 2, 8, 1>(a, s); return true;                           \
        case 5: launch_cfg<T, XBox<4, 8, 8, 3, 8>, 4, 4, 4, 1, 2, 2, 1>(a, s); return true;                           \
        case 6: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 2, 4, 4, 1, 2, 2, 2>(a, s); return true;                           \
        case 7: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 3, 2, 2, 2, 4, 8, 1>(a, s); return true;                          \
        case 8: launch_cfg<T, XBox<4, 4, 12, 3, 4>, 6, 2, 2, 2, 4, 8, 1>(a, s); return true;                          \
        case 9: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 3, 4, 2, 2, 4, 2, 1>(a, s); return true;                          \
        case 10: launch_cfg<T, XBox<2, 8, 16, 3, 16>, 4, 1, 4, 1, 1, 8, 2>(a, s); return true;                        \
        case 11: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 4, 2, 2, 2, 4, 8, 1>(a, s); return true;                          \
        case 12: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 4, 4, 2, 2, 4, 2, 1>(a, s); return true;                          \
        case 13: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 2, 2, 4, 1, 2, 8, 2>(a, s); return true;                          \
        case 14: launch_cfg<T, XBox<4, 8, 8, 3, 8>, 4, 2, 4, 1, 1, 2, 2>(a, s); return true;                          \
        case 15: launch_cfg<T, XBox<2, 8, 16, 3, 16>, 4, 2, 4, 1, 2, 8, 1>(a, s); return true;                        \
        case 16: launch_cfg<T, XBox<2, 8, 16, 3, 16>, 4, 2, 4, 1, 1, 8, 2>(a, s); return true;                        \
        case 17: launch_cfg<T, XBox<4, 8, 8, 3, 8>, 4, 2, 4, 1, 1, 8, 2>(a, s); return true;                          \
        case 20: launch_cfg<T, XBox<4, 8, 8, 3, 8>, 4, 2, 4, 1, 1, 2, 3>(a, s); return true;                          \
        case 21: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 2, 2, 4, 1, 1, 2, 4>(a, s); return true;                          \
        case 22: launch_cfg<T, XBox<2, 8, 8, 3, 8>, 2, 2, 4, 1, 1, 8, 3>(a, s); return true;                          \
        case 23: launch_cfg<T, XBox<4, 8, 8, 3, 8>, 4, 2, 4, 1, 1, 8, 3>(a, s); return true;                          \
        case 44: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 3, 2, 2, 2, 4, 26, 1>(a, s); return true;                         \
        case 45: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 3, 1, 2, 2, 4, 26, 1>(a, s); return true;                         \
        case 46: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 3, 1, 2, 2, 4, 8, 1>(a, s); return true;                          \
        case 47: launch_cfg<T, XBox<4, 4, 12, 3, 4>, 6, 1, 2, 2, 4, 8, 1>(a, s); return true;                          \
        case 48: launch_cfg<T, XBox<4, 4, 12, 3, 4>, 6, 1, 2, 2, 4, 26, 1>(a, s); return true;                         \
        case 49: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 6, 1, 1, 4, 4, 8, 1>(a, s); return true;                          \
        case 50: launch_cfg<T, XBox<2, 4, 12, 3, 4>, 6, 1, 1, 4, 4, 26, 1>(a, s); return true;      
/* Cin == 16 tilings (conv3x16_kernel): box             TM TN PF OCC */
#define SEG_C3X_3D_C16_CASES                                                                                          \
        case 24: launch_cfg16<T, XBox<2, 8, 16, 3, 16>, 4, 1, 2, 4>(a, s); return true;                               \
        case 25: launch_cfg16<T, XBox<4, 8, 16, 3, 16>, 8, 1, 2, 3>(a, s); return true;                               \
        case 26: launch_cfg16<T, XBox<2, 8, 16, 3, 16>, 4, 2, 2, 3>(a, s); return true;                               \
        case 27: launch_cfg16<T, XBox<4, 8, 16, 3, 16>, 8, 2, 2, 2>(a, s); return true;                               \
        case 28: launch_cfg16r<T, XBox<2, 8, 16, 3, 16>, 4, 1, 4>(a, s); return true;                                 \
        case 29: launch_cfg16r<T, XBox<4, 8, 16, 3, 16>, 8, 1, 3>(a, s); return true;
#define SEG_C3X_3D_BODY switch (id) { SEG_C3X_3D_CONV_CASES SEG_C3X_3D_C16_CASES default: return false; }

#define SEG_C3X_2D_CONV_CASES                                                                                         \
        case 32: launch_cfg<T, XBox<1, 16, 16, 1, 16>, 4, 2, 4, 1, 1, 8, 2>(a, s); return true;                       \
        case 33: launch_cfg<T, XBox<1, 16, 16, 1, 16>, 8, 2, 2, 2, 2, 2, 2>(a, s); return true;                       \
        case 34: launch_cfg<T, XBox<1, 8, 16, 1, 16>, 4, 2, 2, 2, 2, 8, 2>(a, s); return true;                        \
        case 35: launch_cfg<T, XBox<1, 8, 16, 1, 16>, 4, 2, 2, 2, 4, 8, 2>(a, s); return true;                        \
        case 36: launch_cfg<T, XBox<1, 8, 16, 1, 16>, 4, 4, 2, 2, 4, 2, 2>(a, s); return true;                        \
        case 37: launch_cfg<T, XBox<1, 16, 16, 1, 16>, 4, 1, 4, 1, 1, 8, 3>(a, s); return true;                       \
        case 38: launch_cfg<T, XBox<1, 8, 8, 1, 8>, 2, 2, 2, 2, 4, 8, 2>(a, s); return true;                          \
        case 39: launch_cfg<T, XBox<1, 8, 16, 1, 16>, 2, 2, 4, 1, 2, 8, 2>(a, s); return true;      
#define SEG_C3X_2D_C16_CASES                                                                                          \
        case 56: launch_cfg16<T, XBox<1, 16, 16, 1, 16>, 4, 1, 2, 4>(a, s); return true;                              \
        case 57: launch_cfg16<T, XBox<1, 16, 16, 1, 16>, 4, 2, 2, 4>(a, s); return true;
#define SEG_C3X_2D_BODY switch (id) { SEG_C3X_2D_CONV_CASES SEG_C3X_2D_C16_CASES default: return false; }


}  // namespace c3x
}  // namespace seg
