// Implicit-GEMM convolution family for gfx950 (MFMA 16x16, LDS double buffering, NDHWC).
//   conv_igemm_kernel<T,NT,SCATTER> : conv 3^d/2^d s2/1^d (+virtual concat), their data-gradients and
//                                      ConvTranspose 2^d s2 (scatter epilogue); fused bias + GroupNorm
//                                      sum / sum-of-squares partials.
//   conv_stem_kernel<T>             : direct conv for the 1..4-channel image (K=27 is not MFMA shaped)
//   head_fwd/bwd                    : 1^d conv to `numclass` + sigmoid/softmax, and its backward.
// Semantics pinned by torch.nn.Conv3d/ConvTranspose3d call sites: networks/VNet3d.py:8,28,29,49,65,70,88,
// networks/Unet3d.py:26-34,66-80 (reference file:line).
#include <cstdlib>

#include "kernels.h"

namespace seg {

namespace {

constexpr int BM = 128;   // GEMM rows (voxels) per workgroup
constexpr int BK = 32;    // reduction elements per step
constexpr int LDT = BK + 8;

struct RowCoord { int n, d, h, w; };
__device__ __forceinline__ RowCoord decode_row(long long m64, int D, int H, int W) {
    int m = (int)m64;                                   // row counts stay below 2^31: 32-bit division
    RowCoord r;
    r.w = m % W; m /= W;
    r.h = m % H; m /= H;
    r.d = m % D;
    r.n = m / D;
    return r;
}

// Per-column sum / sum-of-squares of an LDS tile [rows][BN] into stats[n][Cout][2] (double atomics).
// rows of the tile may straddle samples when the per-sample volume is not a multiple of BM.
template <class T, int BN>
__device__ __forceinline__ void tile_stats(const T* Os, int ldo, int rows_valid, long long m0, long long Vrow,
                                           double* stats, int Cout, int co_base, float* red, int N, int srep = STAT_REP) {
    constexpr int G = 256 / BN;
    stats += (long long)(blockIdx.x % srep) * N * Cout * 2;           // replica of this workgroup
    const int col = threadIdx.x % BN, g = threadIdx.x / BN;
    const int n_first = (int)(m0 / Vrow), n_last = (int)((m0 + rows_valid - 1) / Vrow);
    if (n_first == n_last) {
        float s = 0.f, ss = 0.f;
        for (int r = g; r < rows_valid; r += G) {
            const float v = to_f(Os[r * ldo + col]);
            s += v; ss += v * v;
        }
        red[(g * BN + col) * 2 + 0] = s;
        red[(g * BN + col) * 2 + 1] = ss;
        __syncthreads();
        if (threadIdx.x < BN) {
            double ts = 0.0, tss = 0.0;
            for (int k = 0; k < G; ++k) { ts += red[(k * BN + col) * 2]; tss += red[(k * BN + col) * 2 + 1]; }
            double* dst = stats + ((long long)n_first * Cout + co_base + col) * 2;
            atomicAdd(dst, ts);
            atomicAdd(dst + 1, tss);
        }
    } else {
        // rare (tiny volumes / sample boundary): run-length flush per thread
        double s = 0.0, ss = 0.0;
        int ncur = -1;
        for (int r = g; r < rows_valid; r += G) {
            const int n = (int)((m0 + r) / Vrow);
            if (n != ncur) {
                if (ncur >= 0) {
                    double* dst = stats + ((long long)ncur * Cout + co_base + col) * 2;
                    atomicAdd(dst, s); atomicAdd(dst + 1, ss);
                }
                ncur = n; s = 0.0; ss = 0.0;
            }
            const float v = to_f(Os[r * ldo + col]);
            s += v; ss += (double)v * v;
        }
        if (ncur >= 0) {
            double* dst = stats + ((long long)ncur * Cout + co_base + col) * 2;
            atomicAdd(dst, s); atomicAdd(dst + 1, ss);
        }
    }
}

// KS = 32-wide reduction slices per pipeline stage.  The deep levels launch 28..432 workgroups whose time is the serial
// chain of stages (one global-load latency each): KS = 2 halves the chain with twice the loads in flight.
template <class T, int NT, bool SCATTER, int KS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a, int srep) {
    constexpr int BN = 16 * NT;
    __shared__ T As[2 * KS * BM * LDT];
    __shared__ T Bs[2 * KS * BN * LDT];
    __shared__ float red[512];
    __shared__ int tapoff[32];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l15 = lane & 15, q = lane >> 4;
    const long long Vrow = (long long)a.OD * a.OH * a.OW;
    const long long M = (long long)a.N * Vrow;
    const long long m0 = (long long)blockIdx.x * BM;
    const int cog0 = blockIdx.y * BN;           // first GEMM column of this block
    const int Cin = a.C0 + a.C1;
    const int lg = 31 - __builtin_clz(Cin);     // Cin is a power of two
    const T* in0 = (const T*)a.in0;
    const T* in1 = (const T*)a.in1;
    const T* wp = (const T*)a.w;

    // ---- A-gather bookkeeping: this thread stages chunk `ac` (8 k) of rows ar0 and ar0+64
    const int ac = tid & 3, ar0 = tid >> 2;
    RowCoord rc[2];
    bool rvalid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + ar0 + 64 * i;
        rvalid[i] = m < M;
        rc[i] = decode_row(rvalid[i] ? m : 0, a.OD, a.OH, a.OW);
        if (!SCATTER) { rc[i].d *= a.sd; rc[i].h *= a.sh; rc[i].w *= a.sw; }
    }
    const bool bload = tid < BN * 4;
    const int brow = tid >> 2;
    // tap offsets through LDS: read from the argument segment they were vector-memory loads whose `s_waitcnt vmcnt(0)` also drained the
    // operand loads in flight, one round trip after the other
    if (tid < 32) {
        const int t = tid < a.taps.n ? tid : 0;
        tapoff[tid] = (a.taps.d[t] & 255) | ((a.taps.h[t] & 255) << 8) | ((a.taps.w[t] & 255) << 16);
    }
    // the epilogue's bias values, loaded HERE in one batch (in the epilogue they were NT serial load -> s_waitcnt vmcnt(0) pairs, one L2
    // round trip each, at the end of a ~5 us workgroup)
    float bj[NT];
    {
        const int co_b = SCATTER ? (cog0 % a.Cout) : cog0;
#pragma unroll
        for (int j = 0; j < NT; ++j) bj[j] = 0.f;
        if (a.bias) {
#pragma unroll
            for (int j = 0; j < NT; ++j) bj[j] = a.bias[co_b + j * 16 + l15];
        }
    }
    int od_ = 0, oh_ = 0, ow_ = 0;                       // scatter form: the output tap of this block's columns
    if (SCATTER) { const int tap = cog0 / a.Cout; od_ = a.taps.d[tap]; oh_ = a.taps.h[tap]; ow_ = a.taps.w[tap]; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NT; ++j) settle(bj[j]);
    if (SCATTER) { settle(od_); settle(oh_); settle(ow_); }

    vec<T, 8> areg[KS][2], breg[KS];
    bool aok[KS][2];
    auto gload = [&](int st) {
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int k0 = (st * KS + u) * BK + ac * 8;
            const int tap = SCATTER ? 0 : (k0 >> lg);
            const int ci = SCATTER ? k0 : (k0 & (Cin - 1));
            const bool kin = k0 < a.K;
            int td = 0, th = 0, tw = 0;
            if (!SCATTER) { const int tp = tapoff[kin ? tap : 0]; td = (signed char)tp; th = (signed char)(tp >> 8); tw = (signed char)(tp >> 16); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int id = rc[i].d + td, ih = rc[i].h + th, iw = rc[i].w + tw;
                const bool ok = kin && rvalid[i] && (unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH &&
                                (unsigned)iw < (unsigned)a.IW;
                // branch-free: a load under a divergent `if` has to land before the paths re-join, which serialised the loads of a
                // stage at one L2 round trip each (r02 trace: 2 us per stage, 11 GB/s per CU).  Out-of-range rows read a safe address
                // and are zeroed on the way into LDS (sstore), after the MFMAs of the current stage.
                const long long vox = (((long long)rc[i].n * a.ID + id) * a.IH + ih) * a.IW + iw;
                const T* src = (ci < a.C0) ? in0 + vox * a.C0 + ci : in1 + vox * a.C1 + (ci - a.C0);
                areg[u][i] = load8(ok ? src : in0);
                aok[u][i] = ok;
            }
            breg[u] = load8(wp + (long long)(cog0 + (bload ? brow : 0)) * a.Kpad + (st * KS + u) * BK + ac * 8);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < KS; ++u) {
#pragma unroll
            for (int i = 0; i < 2; ++i) store8(&As[((buf * KS + u) * BM + ar0 + 64 * i) * LDT + ac * 8], aok[u][i] ? areg[u][i] : zero8<T>());
            if (bload) store8(&Bs[((buf * KS + u) * BN + brow) * LDT + ac * 8], breg[u]);
        }
    };

    f32x4 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.Kpad / (BK * KS);
    gload(0);
    sstore(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < nk) gload(ks + 1);
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            typename Mma<T>::frag af[2], bf[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = load8(&As[((buf * KS + u) * BM + wv * 32 + i * 16 + l15) * LDT + q * 8]);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = load8(&Bs[((buf * KS + u) * BN + j * 16 + l15) * LDT + q * 8]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = Mma<T>::run(af[i], bf[j], acc[i][j]);
        }
        if (ks + 1 < nk) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, stage the tile in LDS (aliases As), coalesced stores, GN partial sums
    constexpr int LDO = BN + 8;
    T* Os = As;
    const int co_real0 = SCATTER ? (cog0 % a.Cout) : cog0;   // first real output channel of the block
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = j * 16 + l15;
        const float b = bj[j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) Os[(wv * 32 + i * 16 + q * 4 + r) * LDO + col] = from_f<T>(acc[i][j][r] + b);
    }
    __syncthreads();
    const int rows_valid = (int)((M - m0) < BM ? (M - m0) : BM);
    T* out = (T*)a.out;
    constexpr int CPR = BN / 8;
    for (int ch = tid; ch < BM * CPR; ch += 256) {
        const int row = ch / CPR, cc = ch % CPR;
        if (row >= rows_valid) continue;
        const vec<T, 8> v = load8(&Os[row * LDO + cc * 8]);
        long long orow;
        if (SCATTER) {
            const RowCoord r = decode_row(m0 + row, a.OD, a.OH, a.OW);
            orow = (((long long)r.n * a.FD + r.d * a.sd + od_) * a.FH + r.h * a.sh + oh_) * a.FW + r.w * a.sw + ow_;
        } else {
            orow = m0 + row;
        }
        store8(out + orow * a.Cout + co_real0 + cc * 8, v);
    }
    if (a.stats) tile_stats<T, BN>(Os, LDO, rows_valid, m0, Vrow, a.stats, a.Cout, co_real0, red, a.N, srep);
}

template <class T>
void conv_dispatch(const ConvArgs& a, hipStream_t s, int srep) {
    const long long M = (long long)a.N * a.OD * a.OH * a.OW;
    const int nt = (a.Cout % 64 == 0) ? 4 : (a.Cout % 32 == 0) ? 2 : 1;
    dim3 grid(cdiv(M, BM), a.Ngemm / (16 * nt));
    // two reduction slices per stage where the launch is a latency chain (fewer than two workgroups per CU) and K allows it; FOUR where the launch does not
    // even fill half the CUs (the stride-2 / transposed convs of the 12^3 and 6^3 levels: 28 ... 112 workgroups whose life is K / 64 serial load -> LDS ->
    // barrier stages, 26 us for K = 1024): 16-bit types only (123 KB of LDS, one workgroup per CU - there is one per CU at most anyway)
    static const int ks_env = 0;
    const bool ks2 = a.Kpad % (2 * BK) == 0 && a.Kpad >= 4 * BK && (ks_env ? ks_env == 2 : (long long)grid.x * grid.y <= 512);
    const bool ks4 = sizeof(T) == 2 && a.Kpad % (4 * BK) == 0 && a.Kpad >= 8 * BK && (ks_env ? ks_env == 4 : (long long)grid.x * grid.y <= 128);
#define SEG_LAUNCH_CONV(NT)                                                                          \
    if (ks4) {                                                                                        \
        if constexpr (sizeof(T) == 2) {                                                               \
            if (a.scatter) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, NT, true, 4>), grid, dim3(256), 0, s, a, srep);  \
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, NT, false, 4>), grid, dim3(256), 0, s, a, srep);     \
        }                                                                                             \
    } else if (ks2) {                                                                                 \
        if (a.scatter) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, NT, true, 2>), grid, dim3(256), 0, s, a, srep);  \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, NT, false, 2>), grid, dim3(256), 0, s, a, srep);     \
    } else if (a.scatter) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, NT, true, 1>), grid, dim3(256), 0, s, a, srep);  \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_igemm_kernel<T, NT, false, 1>), grid, dim3(256), 0, s, a, srep);
    if (nt == 4) { SEG_LAUNCH_CONV(4) } else if (nt == 2) { SEG_LAUNCH_CONV(2) } else { SEG_LAUNCH_CONV(1) }
#undef SEG_LAUNCH_CONV
}

// ------------------------------------------------------------------------------------------------
// Streaming variant for the short-reduction convolutions of the two finest levels (1^d convs, 2^d stride-2 convs,
// ConvTranspose 2^d stride 2 and their data-gradients with K <= 128): these are pure HBM streams (64 B in, 32 B out per
// voxel), and the LDS-staged kernel above spends its time in per-workgroup latency (load -> LDS -> barrier -> 1..4
// MFMAs -> LDS -> store: ~5 us for 4 KB; 27648 workgroups at 96^3).  Here nothing goes through LDS: the weights
// (<= 8 MFMA fragments) live in registers for the whole kernel, every lane loads its own B fragment (8 consecutive k
// of one voxel row = one coalesced 16-B piece) straight from HBM, and the operands are swapped (D = W * X^T) so a lane
// ends up with 4 consecutive output channels of one voxel -> 8-B coalesced stores, no transpose.  Each wave walks a
// strided list of 16-voxel tiles of ONE sample, keeping GroupNorm partial sums in registers.
// ------------------------------------------------------------------------------------------------
// SC: 0 = gather form, 1 = scatter form (ConvTranspose / data-gradient of a strided conv: tile j = tap (j * 16) / Cout), 2 = scatter form
// with Cout == 16: every tile holds the SAME 16 channels, so one set of bias values and one set of GroupNorm partial sums serves all
// NTL tiles (84 VGPRs less at NTL = 8: three waves per SIMD instead of two)
#ifndef SEG_EMU
#define SEG_STREAM_WAVES(x) __attribute__((amdgpu_waves_per_eu(x)))     // lower bound on resident waves per SIMD = upper bound on VGPRs
#else
#define SEG_STREAM_WAVES(x)
#endif
// ACT (gather form): in0 holds the RAW output of a conv + GroupNorm unit; every piece loaded from it becomes relu(scale * r + shift) rounded to T - the
// producer's GroupNorm + channel dropout + ReLU - in the registers of its reader, with the same fmaf / fmaxf / rounding as gn_act_kernel: the activated tensor
// of the VNet up-conv (226 MB written and read again per pass at 4 x 96^3) is never materialised.  A lane's 8 channels are the same for every tile: the
// coefficients are loop-invariant registers.
// RQ (two output tensors): the first one is the gradient dz of an activation whose raw tensor r the lane reads for its own (voxel, 4 channels); the GroupNorm-backward
// sums of that unit - sum dz * gate and sum dz * gate * r, gate = [scale * r + shift > 0], dz as stored (rounded) - ride on this launch through the statistics
// epilogue (same per-lane accumulators, 16-lane sums, fp64 atomics), and gn_bwd_reduce_kernel's pass over (dz, r) is not launched: one tensor read instead of two.
template <class T, int KS, int NTL, int SC, bool ACT = false, bool RQ = false>
__global__ __launch_bounds__(256) SEG_STREAM_WAVES(SC == 2 && NTL == 8 ? 3 : 1) void conv_stream_kernel(ConvArgs a) {
    static_assert(!ACT || SC == 0, "activation on load: gather form only");
    static_assert(!RQ || (SC == 0 && !ACT), "GroupNorm-backward sums: gather form with two outputs");
    constexpr bool SCATTER = SC != 0;
    constexpr bool C16 = SC == 2;
    constexpr int NJ = C16 ? 1 : NTL;
    __shared__ float red[4][NTL * 16][2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const int n = blockIdx.y;
    const long long Vrow = (long long)a.OD * a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    const int lg = 31 - __builtin_clz(Cin);
    const T* in0 = (const T*)a.in0;
    const T* in1 = (const T*)a.in1;
    const T* wp = (const T*)a.w;
    T* out = (T*)a.out;

    typename Mma<T>::frag wf[KS][NTL];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NTL; ++j) wf[ks][j] = load8(wp + (long long)(j * 16 + l15) * a.Kpad + ks * 32 + q * 8);
    float bs[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bs[j][r] = a.bias ? a.bias[(SCATTER ? (j * 16) % a.Cout : j * 16) + 4 * q + r] : 0.f;
    // reduction coordinates of this lane's 8-element piece in every K step
    int td[KS], th[KS], tw[KS], ci[KS];
    bool kin[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k0 = ks * 32 + q * 8;
        kin[ks] = k0 < a.K;
        const int tap = (SCATTER || !kin[ks]) ? 0 : (k0 >> lg);
        ci[ks] = SCATTER ? k0 : (k0 & (Cin - 1));
        td[ks] = SCATTER ? 0 : a.taps.d[tap]; th[ks] = SCATTER ? 0 : a.taps.h[tap]; tw[ks] = SCATTER ? 0 : a.taps.w[tap];
    }
    float s1[NJ][4], s2[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
    // scatter form: the output tap of every 16-channel tile, read ONCE (inside the loop each store sat behind three serial
    // tap-table loads, every one with its own s_waitcnt vmcnt(0))
    int toff[NTL];               // output-row offset of the tile's tap: (kd * FH + kh) * FW + kw
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
        const int tap = SCATTER ? (j * 16) / a.Cout : 0;
        toff[j] = SCATTER ? (a.taps.d[tap] * a.FH + a.taps.h[tap]) * a.FW + a.taps.w[tap] : 0;
        if (SCATTER) settle(toff[j]);
    }

    const int ntile = (int)(Vrow / 16);
    // U tiles per iteration: all their loads are issued before the first MFMA (>= 4 x 16 B in flight per lane)
    constexpr int U = KS >= 4 ? 1 : (NTL >= 8 ? 2 : 4 / KS);     // 8 output tiles per input tile: the stores dominate, two tiles keep a third wave per SIMD
    const int step = gridDim.x * 4;
    // the tap-table entries above are vector-memory loads (dynamic index into the argument segment): make them land HERE.  Left
    // pending at the loop entry they made hipcc put s_waitcnt vmcnt(2) / (1) / (0) in front of every tile's address arithmetic, which
    // (vmcnt retires in order) also waited for the previous tile's load: the U loads "in flight" ran one after the other
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { settle(td[ks]); settle(th[ks]); settle(tw[ks]); settle(ci[ks]); }
    constexpr int NT0 = (RQ && NTL >= 2) ? NTL / 2 : 1;                      // RQ: the first output holds half of the columns (both sources of the concat have the same width)
    float rsc[NT0][4], rsh[NT0][4];
    if (RQ) {
#pragma unroll
        for (int j = 0; j < NT0; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rsc[j][r] = a.rq_scale[(long long)n * a.Cout0 + j * 16 + 4 * q + r];
                rsh[j][r] = a.rq_shift[(long long)n * a.Cout0 + j * 16 + 4 * q + r];
            }
    }
    vec<float, 8> asc[ACT ? KS : 1], ash[ACT ? KS : 1];
    bool afrom0[ACT ? KS : 1];
    if (ACT) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            afrom0[ks] = kin[ks] && ci[ks] < a.C0;
            const int c = afrom0[ks] ? ci[ks] : 0;                       // (the other lanes load valid coefficients and ignore them)
            asc[ks] = *(const vec<float, 8>*)(a.act_scale + (long long)n * a.C0 + c);
            ash[ks] = *(const vec<float, 8>*)(a.act_shift + (long long)n * a.C0 + c);
        }
    }
    for (int t0 = blockIdx.x * 4 + wv; t0 < ntile; t0 += step * U) {
        typename Mma<T>::frag xf[U][KS];
        int d_[U], h_[U], w_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u * step;
            int m = (t < ntile ? t : t0) * 16 + l15;
            w_[u] = m % a.OW; m /= a.OW;
            h_[u] = m % a.OH;
            d_[u] = m / a.OH;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xf[u][ks] = zero8<T>();
                if (kin[ks] && t < ntile) {
                    const int id = SCATTER ? d_[u] : d_[u] * a.sd + td[ks], ih = SCATTER ? h_[u] : h_[u] * a.sh + th[ks],
                              iw = SCATTER ? w_[u] : w_[u] * a.sw + tw[ks];
                    const long long vox = (((long long)n * a.ID + id) * a.IH + ih) * a.IW + iw;
                    xf[u][ks] = (ci[ks] < a.C0) ? load8(in0 + vox * a.C0 + ci[ks]) : load8(in1 + vox * a.C1 + (ci[ks] - a.C0));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u * step;
            if (t >= ntile) break;
            f32x4 acc[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            vec<T, 4> ry[NT0];
            if (RQ) {                                   // this lane's (voxel, 4 channels) of the raw tensor, issued in front of the MFMAs
                const long long orow = (long long)n * Vrow + t * 16 + l15;
#pragma unroll
                for (int j = 0; j < NT0; ++j) ry[j] = *(const vec<T, 4>*)((const T*)a.rq_r + orow * a.Cout0 + j * 16 + 4 * q);
            }
            if (ACT) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    typename Mma<T>::frag v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = from_f<T>(fmaxf(fmaf(asc[ks][e], to_f(xf[u][ks][e]), ash[ks][e]), 0.f));
                    if (afrom0[ks]) xf[u][ks] = v;
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < NTL; ++j) acc[j] = Mma<T>::run(wf[ks][j], xf[u][ks], acc[j]);
            // rounded outputs + GroupNorm partial sums of every tile
            vec<T, 4> o[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o[j][r] = from_f<T>(acc[j][r] + bs[C16 ? 0 : j][r]);
                    const float f = to_f(o[j][r]);
                    if (RQ) {
                        if (j < NT0) {
                            const float yv = to_f(ry[j < NT0 ? j : 0][r]);
                            const float d = (fmaf(rsc[j < NT0 ? j : 0][r], yv, rsh[j < NT0 ? j : 0][r]) > 0.f) ? f : 0.f;
                            s1[j][r] += d;
                            s2[j][r] = fmaf(d, yv, s2[j][r]);
                        }
                    } else {
                        s1[C16 ? 0 : j][r] += f;
                        s2[C16 ? 0 : j][r] = fmaf(f, f, s2[C16 ? 0 : j][r]);
                    }
                }
            // address of the 16-channel group of tile j for this lane's voxel
            auto tile_ptr = [&](int j) -> T* {          // j: a compile-time constant at every call site
                if (SCATTER) {
                    const long long orow = (((long long)n * a.FD + d_[u] * a.sd) * a.FH + h_[u] * a.sh) * a.FW + w_[u] * a.sw + toff[j];
                    return out + orow * a.Cout + (j * 16) % a.Cout;
                }
                const long long orow = (long long)n * Vrow + t * 16 + l15;
                if (a.out1)                             // two output tensors (both data-gradients of a virtual concat from one pass over d(raw)): a wave-uniform choice
                    return j * 16 < a.Cout0 ? out + orow * a.Cout0 + j * 16 : (T*)a.out1 + orow * (a.Cout - a.Cout0) + (j * 16 - a.Cout0);
                return out + orow * a.Cout + j * 16;
            };
            if (NTL % 2 == 0) {
                // tiles in pairs (A, B): lanes q and q^1 swap one 4-channel piece so that even q stores 8 consecutive
                // channels of A and odd q 8 of B: 16-B stores, and for the 2^d transposed convs (taps w-fastest) the
                // two tiles are neighbouring voxels of one fine row -> every store instruction covers whole 64-B runs
                const bool odd = q & 1;
#pragma unroll
                for (int jp = 0; jp < NTL / 2; ++jp) {
                    const vec<T, 4> mine = odd ? o[2 * jp + 1] : o[2 * jp], send = odd ? o[2 * jp] : o[2 * jp + 1];
                    vec<T, 4> recv;
                    constexpr int NW = sizeof(T) * 4 / 4;
                    int sw[NW], rw[NW];
                    __builtin_memcpy(sw, &send, sizeof(send));
#pragma unroll
                    for (int k = 0; k < NW; ++k) rw[k] = __shfl_xor(sw[k], 16);
                    __builtin_memcpy(&recv, rw, sizeof(recv));
                    vec<T, 8> w8;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { w8[r] = odd ? recv[r] : mine[r]; w8[4 + r] = odd ? mine[r] : recv[r]; }
                    T* const pa = tile_ptr(2 * jp);
                    T* const pb = tile_ptr(2 * jp + 1);
                    store8((odd ? pb : pa) + (q >> 1) * 8, w8);
                }
            } else {
#pragma unroll
                for (int j = 0; j < NTL; ++j) *(vec<T, 4>*)(tile_ptr(j) + 4 * q) = o[j];
            }
        }
    }
    if (RQ) {
        // the sums of the first output's channels (tiles below Cout0): 16 voxel lanes, then the four waves, then fp64 atomics into the unit's Q replicas
#pragma unroll
        for (int j = 0; j < NT0; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[j][r], v = s2[j][r];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) { u += __shfl_xor(u, msk); v += __shfl_xor(v, msk); }
                if (l15 == 0) { red[wv][j * 16 + 4 * q + r][0] = u; red[wv][j * 16 + 4 * q + r][1] = v; }
            }
        __syncthreads();
        if (tid < a.Cout0) {
            double ts = 0.0, tss = 0.0;
            for (int k = 0; k < 4; ++k) { ts += red[k][tid][0]; tss += red[k][tid][1]; }
            double* dst = a.rq_Q + (((long long)(blockIdx.x % STAT_REP) * a.N + n) * a.Cout0 + tid) * 2;
            atomicAdd(dst, ts);
            atomicAdd(dst + 1, tss);
        }
    } else
    if (a.stats) {
        // per-channel sums: over the 16 voxel lanes of the wave, then over waves (and taps for the scatter form)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float u = s1[j][r], v = s2[j][r];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) { u += __shfl_xor(u, msk); v += __shfl_xor(v, msk); }
                if (l15 == 0) { red[wv][j * 16 + 4 * q + r][0] = u; red[wv][j * 16 + 4 * q + r][1] = v; }
            }
        __syncthreads();
        if (tid < a.Cout) {
            double ts = 0.0, tss = 0.0;
            for (int col = tid; col < NJ * 16; col += a.Cout)       // gather form: exactly one column
                for (int k = 0; k < 4; ++k) { ts += red[k][col][0]; tss += red[k][col][1]; }
            double* dst = a.stats + (((long long)(blockIdx.x % STAT_REP) * a.N + n) * a.Cout + tid) * 2;
            atomicAdd(dst, ts);
            atomicAdd(dst + 1, tss);
        }
    }
}

bool stream_eligible(const ConvArgs& a) {
    if (a.act_scale && (a.scatter || !a.act_shift || a.C0 % 8)) return false;
    if (a.out1 && (a.scatter || a.stats || a.bias || a.Cout0 <= 0 || a.Cout0 % 16 || a.Cout0 >= a.Ngemm || (a.Ngemm / 16) % 2)) return false;
    if (a.rq_Q && (!a.out1 || !a.rq_r || !a.rq_scale || !a.rq_shift || a.act_scale || a.Cout0 * 2 != a.Ngemm)) return false;
    static const bool off = knob_i("SEG_CONV_STREAM", 1) == 0;
    if (off) return false;
    const long long Vrow = (long long)a.OD * a.OH * a.OW;
    if (Vrow % 16 || a.Kpad > 128 || a.Kpad % 32 || a.Ngemm % 16 || a.Ngemm > 128) return false;
    const int ks = a.Kpad / 32, ntl = a.Ngemm / 16;
    if (ks * ntl > 8 || ks == 3 || (ks == 4 && ntl > 2) || (ks == 2 && ntl > 4)) return false;
    if (ntl != 1 && ntl != 2 && ntl != 4 && ntl != 8) return false;
    if (a.scatter) {
        if (a.Cout % 16 || a.K > a.Kpad) return false;
    } else {
        if (a.Ngemm != a.Cout) return false;
        for (int t = 0; t < a.taps.n; ++t) {       // every tap must stay inside the source for every output voxel
            if (a.taps.d[t] < 0 || a.taps.h[t] < 0 || a.taps.w[t] < 0) return false;
            if ((a.OD - 1) * a.sd + a.taps.d[t] >= a.ID || (a.OH - 1) * a.sh + a.taps.h[t] >= a.IH || (a.OW - 1) * a.sw + a.taps.w[t] >= a.IW)
                return false;
        }
    }
    return true;
}

template <class T>
bool launch_conv_stream(const ConvArgs& a, hipStream_t s) {
    if (!stream_eligible(a)) return false;
    const long long Vrow = (long long)a.OD * a.OH * a.OW;
    const int ks = a.Kpad / 32, ntl = a.Ngemm / 16;
    const int ntile = (int)(Vrow / 16);
    int gx = ntile / 32;                              // >= 8 tiles per wave
    if (gx > 1024) gx = 1024;
    if (gx < 1) gx = 1;
    dim3 grid(gx, a.N);
#define SEG_STREAM(KS, NTL)                                                                                                   \
    if (ks == KS && ntl == NTL) {                                                                                            \
        if (a.scatter && a.Cout == 16 && NTL > 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stream_kernel<T, KS, NTL, 2>), grid, dim3(256), 0, s, a); \
        else if (a.scatter) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stream_kernel<T, KS, NTL, 1>), grid, dim3(256), 0, s, a); \
        else if (a.act_scale) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stream_kernel<T, KS, NTL, 0, true>), grid, dim3(256), 0, s, a); \
        else if (a.rq_Q) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stream_kernel<T, KS, NTL, 0, false, true>), grid, dim3(256), 0, s, a); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stream_kernel<T, KS, NTL, 0>), grid, dim3(256), 0, s, a);               \
        return true;                                                                                                         \
    }
    SEG_STREAM(1, 1) SEG_STREAM(1, 2) SEG_STREAM(1, 4) SEG_STREAM(1, 8) SEG_STREAM(2, 1) SEG_STREAM(2, 2) SEG_STREAM(2, 4)
    SEG_STREAM(4, 1) SEG_STREAM(4, 2)
#undef SEG_STREAM
    return false;
}

// ------------------------------------------------------------------------------------------------
// stem: direct conv, Cimg in 1..4, Cout multiple of 8 (<= 64).  256 voxels per workgroup.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void conv_stem_kernel(StemArgs a) {
    constexpr int MAXW = 27 * 4 * 32;   // taps x Cimg(<=4) x Cout(<=32)
    __shared__ float ws[MAXW];          // [tap][ci][co]
    __shared__ T Os[256 * (32 + 8)];
    __shared__ float red[512];
    const int tid = threadIdx.x;
    const int nt = a.taps.n, Cimg = a.Cimg, Cout = a.Cout;
    for (int i = tid; i < nt * Cimg * Cout; i += 256) {
        const int co = i % Cout, ci = (i / Cout) % Cimg, t = i / (Cout * Cimg);
        ws[i] = a.w[((long long)co * Cimg + ci) * nt + t];
    }
    __syncthreads();
    const long long V = (long long)a.D * a.H * a.W, M = (long long)a.N * V;
    const long long m0 = (long long)blockIdx.x * 256;
    const long long m = m0 + tid;
    const int LDO = Cout + 8;
    const T* in = (const T*)a.in;
    if (m < M) {
        const RowCoord r = decode_row(m, a.D, a.H, a.W);
        for (int c0 = 0; c0 < Cout; c0 += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = a.bias ? a.bias[c0 + j] : 0.f;
            for (int t = 0; t < nt; ++t) {
                const int id = r.d + a.taps.d[t], ih = r.h + a.taps.h[t], iw = r.w + a.taps.w[t];
                if ((unsigned)id >= (unsigned)a.D || (unsigned)ih >= (unsigned)a.H || (unsigned)iw >= (unsigned)a.W) continue;
                const long long vox = (((long long)r.n * a.D + id) * a.H + ih) * a.W + iw;
                for (int ci = 0; ci < Cimg; ++ci) {
                    const float xv = to_f(in[vox * Cimg + ci]);
                    const float* wr = &ws[(t * Cimg + ci) * Cout + c0];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) Os[tid * LDO + c0 + j] = from_f<T>(acc[j]);
        }
    }
    __syncthreads();
    const int rows_valid = (int)((M - m0) < 256 ? (M - m0) : 256);
    T* out = (T*)a.out;
    const int CPR = Cout / 8;
    for (int ch = tid; ch < 256 * CPR; ch += 256) {
        const int row = ch / CPR, cc = ch % CPR;
        if (row < rows_valid) store8(out + (m0 + row) * Cout + cc * 8, load8(&Os[row * LDO + cc * 8]));
    }
    if (a.stats) {
        // column sums, 16 columns at a time with the shared helper
        for (int c0 = 0; c0 < Cout; c0 += 16) {
            tile_stats<T, 16>(Os + c0, LDO, rows_valid, m0, V, a.stats, Cout, c0, red, a.N);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// head: per-voxel dot products to numclass logits, planar fp32 outputs (the reference API returns
// NC[D]HW float32 logits and probabilities).  One thread per voxel.
// ------------------------------------------------------------------------------------------------
constexpr int MAXCLS = 16;      // class cap (misc.hip)

template <class T>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
    __shared__ float ws[MAXCLS * 64 + MAXCLS];
    const int tid = threadIdx.x;
    if (a.zero_ptr && blockIdx.x == 0)
        for (long long i = tid; i < a.zero_n; i += 256) a.zero_ptr[i] = 0.0;
    for (int i = tid; i < a.C * a.Cin; i += 256) ws[i] = a.w[i];
    if (tid < a.C) ws[MAXCLS * 64 + tid] = a.bias ? a.bias[tid] : 0.f;
    __syncthreads();
    const long long M = (long long)a.N * a.V;
    const T* in = (const T*)a.in;
    for (long long m = (long long)blockIdx.x * 256 + tid; m < M; m += (long long)gridDim.x * 256) {
        float z[MAXCLS];
        for (int c = 0; c < a.C; ++c) z[c] = ws[MAXCLS * 64 + c];
        for (int k0 = 0; k0 < a.Cin; k0 += 8) {
            const vec<T, 8> x = load8(in + m * a.Cin + k0);
            for (int c = 0; c < a.C; ++c) {
                float s = z[c];
#pragma unroll
                for (int j = 0; j < 8; ++j) s = fmaf(to_f(x[j]), ws[c * a.Cin + k0 + j], s);
                z[c] = s;
            }
        }
        const long long n = m / a.V, v = m % a.V;
        if (a.C == 1) {
            a.logits[m] = z[0];
            a.probs[m] = 1.f / (1.f + expf(-z[0]));
        } else {
            float mx = z[0];
            for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, z[c]);
            float e[MAXCLS], se = 0.f;
            for (int c = 0; c < a.C; ++c) { e[c] = expf(z[c] - mx); se += e[c]; }
            const float inv = 1.f / se;
            for (int c = 0; c < a.C; ++c) {
                const long long o = (n * a.C + c) * a.V + v;
                a.logits[o] = z[c];
                a.probs[o] = e[c] * inv;
            }
        }
    }
}

// backward: din[m][ci] = sum_c dlogit[n][c][v] w[c][ci]; dw[c][ci] += sum_m dlogit*in; db[c] += sum dlogit.
// grid = (slabs, N); thread = (voxel, 8-channel chunk), two voxels in flight; dw partials are reduced over the block in LDS.
// NC = compile-time class count for 1..4 classes; NC = 8 / 16: runtime a.C <= NC (guarded full unroll so dw/db stay in registers); DIN = the data
// gradient is materialised (the engine keeps it virtual: the consumer recomputes it from dlogits and the head weights).
template <class T, int NC, bool DIN>
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
    constexpr int CM = NC;
    __shared__ float ws[MAXCLS * 64];
    __shared__ float red[256 * 9];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int C = NC <= 4 ? NC : a.C;
    for (int i = tid; i < C * a.Cin; i += 256) ws[i] = a.w[i];
    __syncthreads();
    const int CPR = a.Cin / 8;                 // chunks per voxel (power of two, <= 8)
    const int cc = tid % CPR, vslot = tid / CPR, VPB = 256 / CPR;
    const int V = (int)a.V;
    const T* in = (const T*)a.in + (long long)n * V * a.Cin + cc * 8;
    T* din = DIN ? (T*)a.din + (long long)n * V * a.Cin + cc * 8 : nullptr;
    const float* dlog = a.dlogits + (long long)n * C * V;
    float dw[CM][8], db[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) { db[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dw[c][j] = 0.f; }
    const int stride = gridDim.x * VPB;
    for (int v = blockIdx.x * VPB + vslot; v < V; v += 2 * stride) {
        const int vb = v + stride;
        const bool two = vb < V;
        const vec<T, 8> x0 = load8(in + (long long)v * a.Cin);
        const vec<T, 8> x1 = two ? load8(in + (long long)vb * a.Cin) : zero8<T>();
        float d0[CM], d1[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            d0[c] = (c < C) ? dlog[(long long)c * V + v] : 0.f;
            d1[c] = (c < C && two) ? dlog[(long long)c * V + vb] : 0.f;
        }
        float g0[8], g1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { g0[j] = 0.f; g1[j] = 0.f; }
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                db[c] += d0[c] + d1[c];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dw[c][j] = fmaf(d0[c], to_f(x0[j]), fmaf(d1[c], to_f(x1[j]), dw[c][j]));
                    if (DIN) {
                        const float wv = ws[c * a.Cin + cc * 8 + j];
                        g0[j] = fmaf(d0[c], wv, g0[j]);
                        g1[j] = fmaf(d1[c], wv, g1[j]);
                    }
                }
            }
        }
        if (DIN) {
            vec<T, 8> o0, o1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { o0[j] = from_f<T>(g0[j]); o1[j] = from_f<T>(g1[j]); }
            store8(din + (long long)v * a.Cin, o0);
            if (two) store8(din + (long long)vb * a.Cin, o1);
        }
    }
    // block reduction: for each class, 8 dw columns (+ db) per thread -> sum over threads with equal cc
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (c < C) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid * 9 + j] = dw[c][j];
            red[tid * 9 + 8] = db[c];
            __syncthreads();
            if (tid < CPR * 8) {
                const int ccx = tid / 8, j = tid % 8;
                float s = 0.f;
                for (int t = ccx; t < 256; t += CPR) s += red[t * 9 + j];
                atomicAdd(&a.dw[c * a.Cin + ccx * 8 + j], s);
            }
            if (tid == 255) {
                float s = 0.f;
                for (int t = 0; t < 256; t += CPR) s += red[t * 9 + 8];   // db counted once per voxel (cc == 0 lanes)
                atomicAdd(&a.db[c], s);
            }
        }
    }
}

}  // namespace

bool conv_uses_stream_kernel(const ConvArgs& a) { return stream_eligible(a); }

void launch_conv_igemm(const ConvArgs& a, int dtype, hipStream_t s, int stat_rep) {
    const int srep = (stat_rep > 0 && stat_rep <= STAT_REP) ? stat_rep : STAT_REP;
    if (a.rq_Q && !stream_eligible(a)) { fprintf(stderr, "segengine: GroupNorm-backward sums on a conv launch need the streaming kernel with two outputs (internal error)\n"); abort(); }
    if (a.out1 && !stream_eligible(a)) { fprintf(stderr, "segengine: two output tensors need the streaming conv kernel (internal error)\n"); abort(); }
    if (a.act_scale && !stream_eligible(a)) { fprintf(stderr, "segengine: activation on load needs the streaming conv kernel (internal error)\n"); abort(); }
    if (dtype == DT_F32) { if (!launch_conv_stream<float>(a, s)) conv_dispatch<float>(a, s, srep); }
    else if (dtype == DT_F16) { if (!launch_conv_stream<f16>(a, s)) conv_dispatch<f16>(a, s, srep); }
    else { if (!launch_conv_stream<bf16>(a, s)) conv_dispatch<bf16>(a, s, srep); }
}

void launch_conv_stem(const StemArgs& a, int dtype, hipStream_t s) {
    const long long M = (long long)a.N * a.D * a.H * a.W;
    dim3 grid(cdiv(M, 256));
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stem_kernel<float>), grid, dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stem_kernel<f16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(conv_stem_kernel<bf16>), grid, dim3(256), 0, s, a);
}

void launch_head_fwd(const HeadArgs& a, int dtype, hipStream_t s) {
    const long long M = (long long)a.N * a.V;
    int blocks = cdiv(M, 256);
    if (blocks > 8192) blocks = 8192;
    dim3 grid(blocks);
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(head_fwd_kernel<float>), grid, dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(head_fwd_kernel<f16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(head_fwd_kernel<bf16>), grid, dim3(256), 0, s, a);
}

template <class T>
static void head_bwd_dispatch(const HeadBwdArgs& a, dim3 grid, hipStream_t s) {
#define SEG_HB(NC)                                                                                                       \
    if (a.din) hipLaunchKernelGGL(HIP_KERNEL_NAME(head_bwd_kernel<T, NC, true>), grid, dim3(256), 0, s, a);               \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(head_bwd_kernel<T, NC, false>), grid, dim3(256), 0, s, a);
    if (a.C == 1) { SEG_HB(1) } else if (a.C == 2) { SEG_HB(2) } else if (a.C == 3) { SEG_HB(3) } else if (a.C == 4) { SEG_HB(4) }
    else if (a.C <= 8) { SEG_HB(8) } else { SEG_HB(16) }
#undef SEG_HB
}

void launch_head_bwd(const HeadBwdArgs& a, int dtype, hipStream_t s) {
    const int VPB = 256 / (a.Cin / 8);
    int blocks = cdiv(a.V, 2 * VPB);           // per sample; two voxels per thread and trip
    static const int cap = 1024;      // tuning knob: workgroups per launch
    const int per_n = cap / a.N > 0 ? cap / a.N : 1;
    if (blocks > per_n) blocks = per_n;
    dim3 grid(blocks, a.N);
    if (dtype == DT_F32) head_bwd_dispatch<float>(a, grid, s);
    else if (dtype == DT_F16) head_bwd_dispatch<f16>(a, grid, s);
    else head_bwd_dispatch<bf16>(a, grid, s);
}

}  // namespace seg
