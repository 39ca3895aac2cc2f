// Weight-gradient GEMM for gfx950:  dW[p][tap][q] += sum_m dR[m][p] * X[vox(m,tap)][q]
// Both operands are reduced along the voxel axis m, which is the SLOW axis of the channels-last
// tensors, so both MFMA fragments need a transpose: tiles are staged row-major [m][c] in LDS and
// read with ds_read_b64_tr_b16 (f16/bf16) or element-wise (f32, v_mfma_f32_16x16x4_f32 wants one
// k per lane).  Partial tiles are accumulated into the fp32 master-gradient with atomics; the voxel
// axis is split over gridDim.y.  Pinned by autograd of the conv call sites in networks/VNet3d.py /
// Unet3d.py (reference), i.e. torch `convolution_backward` weight gradients.
#include "kernels.h"

namespace seg {
namespace {

constexpr int WM = 32;        // voxel rows per step (one MFMA K-step)
constexpr int WT = 64;        // max tile extent along p and q
constexpr int LDW = WT + 8;

struct RowCoord { int n, d, h, w; };
__device__ __forceinline__ RowCoord decode_row(long long m, int D, int H, int W) {
    RowCoord r;
    r.w = (int)(m % W); m /= W;
    r.h = (int)(m % H); m /= H;
    r.d = (int)(m % D);
    r.n = (int)(m / D);
    return r;
}

template <class T> struct TFrag {
    // 16-bit types: two transposing reads -> 8 k-values (rows 4q..4q+3 and 16+4q..16+4q+3) of column col0+l15
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

template <class T, bool STEM>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a, int TP, int TQ, int ntq, int ntile, long long Mc) {
    __shared__ T Ds[WM * LDW];
    __shared__ T Xs[WM * LDW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tap = STEM ? 0 : (int)(blockIdx.x / ntile);
    const int tile = (int)(blockIdx.x % ntile);
    const int p0 = (tile / ntq) * TP, q0 = (tile % ntq) * TQ;
    const long long M = (long long)a.N * a.OD * a.OH * a.OW;
    const long long mbeg = (long long)blockIdx.y * Mc;
    const long long mend = (mbeg + Mc < M) ? mbeg + Mc : M;
    const int Qc = a.C0 + a.C1;
    const T* dr = (const T*)a.dr;
    const T* x0 = (const T*)a.x0;
    const T* x1 = (const T*)a.x1;
    int td = 0, th = 0, tw = 0;
    if (!STEM) { td = a.taps.d[tap]; th = a.taps.h[tap]; tw = a.taps.w[tap]; }

    const int nt_p = TP / 16, nt_q = (TQ + 15) / 16, n16 = nt_p * nt_q;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int cpr_p = TP / 8;
    const int cpr_q = STEM ? 8 : TQ / 8;
    for (long long ms = mbeg; ms < mend; ms += WM) {
        // ---- stage dR rows
        if (tid < WM * cpr_p) {
            const int row = tid / cpr_p, cc = tid % cpr_p;
            const long long m = ms + row;
            vec<T, 8> v = zero8<T>();
            if (m < mend) v = load8(dr + m * a.P + p0 + cc * 8);
            store8(&Ds[row * LDW + cc * 8], v);
        }
        // ---- stage gathered X rows
        if (STEM) {
            const int row = tid >> 3, c4 = (tid & 7) * 4;
            const long long m = ms + row;
            const bool mv = m < mend;
            const RowCoord r = decode_row(mv ? m : 0, a.OD, a.OH, a.OW);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = c4 + j;
                float xv = 0.f;
                if (mv && col < a.Q) {
                    const int tp = col / a.C0, ci = col % a.C0;
                    const int id = r.d + a.taps.d[tp], ih = r.h + a.taps.h[tp], iw = r.w + a.taps.w[tp];
                    if ((unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW)
                        xv = to_f(x0[((((long long)r.n * a.ID + id) * a.IH + ih) * a.IW + iw) * a.C0 + ci]);
                }
                Xs[row * LDW + col] = from_f<T>(xv);
            }
        } else if (tid < WM * cpr_q) {
            const int row = tid / cpr_q, cc = tid % cpr_q;
            const long long m = ms + row;
            vec<T, 8> v = zero8<T>();
            if (m < mend) {
                const RowCoord r = decode_row(m, a.OD, a.OH, a.OW);
                const int id = r.d * a.sd + td, ih = r.h * a.sh + th, iw = r.w * a.sw + tw;
                if ((unsigned)id < (unsigned)a.ID && (unsigned)ih < (unsigned)a.IH && (unsigned)iw < (unsigned)a.IW) {
                    const long long vox = (((long long)r.n * a.ID + id) * a.IH + ih) * a.IW + iw;
                    const int qc = q0 + cc * 8;
                    v = (qc < a.C0) ? load8(x0 + vox * a.C0 + qc) : load8(x1 + vox * a.C1 + (qc - a.C0));
                }
            }
            store8(&Xs[row * LDW + cc * 8], v);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tt = wv + 4 * i;
            if (tt < n16) {
                const int pi = tt / nt_q, qi = tt % nt_q;
                const typename Mma<T>::frag af = TFrag<T>::load(Ds, pi * 16, lane);
                const typename Mma<T>::frag bf = TFrag<T>::load(Xs, qi * 16, lane);
                acc[i] = Mma<T>::run(af, bf, acc[i]);
            }
        }
        __syncthreads();
    }
    (void)Qc;
    // ---- scatter-add the partial tile into the fp32 master gradient
    const int l15 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tt = wv + 4 * i;
        if (tt < n16) {
            const int pi = tt / nt_q, qi = tt % nt_q;
            const int qq = q0 + qi * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + pi * 16 + 4 * q + r;
                if (STEM) {
                    if (qq < a.Q) {
                        const int tp = qq / a.C0, ci = qq % a.C0;
                        atomicAdd(a.dw + p * a.sP + ci * a.sQ + tp * a.sT, acc[i][r]);
                    }
                } else {
                    atomicAdd(a.dw + p * a.sP + qq * a.sQ + tap * a.sT, acc[i][r]);
                }
            }
        }
    }
}

template <class T>
void wgrad_dispatch(const WgradArgs& a, hipStream_t s) {
    const long long M = (long long)a.N * a.OD * a.OH * a.OW;
    const int TP = a.P < WT ? a.P : WT;
    int TQ, ntq, bx;
    if (a.stem) { TQ = 32; ntq = 1; }
    else { TQ = a.Q < WT ? a.Q : WT; ntq = a.Q / TQ; }
    const int ntile = (a.P / TP) * ntq;
    bx = a.stem ? ntile : a.taps.n * ntile;
    long long gy = 4096 / bx;
    if (gy < 1) gy = 1;
    const long long maxgy = (M + WM - 1) / WM;
    if (gy > maxgy) gy = maxgy;
    long long Mc = (M + gy - 1) / gy;
    Mc = (Mc + WM - 1) / WM * WM;
    gy = (M + Mc - 1) / Mc;
    dim3 grid(bx, (unsigned)gy);
    if (a.stem) hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<T, true>), grid, dim3(256), 0, s, a, TP, TQ, ntq, ntile, Mc);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(wgrad_kernel<T, false>), grid, dim3(256), 0, s, a, TP, TQ, ntq, ntile, Mc);
}

}  // namespace

void launch_wgrad(const WgradArgs& a, int dtype, hipStream_t s) {
    if (dtype == DT_F32) wgrad_dispatch<float>(a, s);
    else if (dtype == DT_F16) wgrad_dispatch<f16>(a, s);
    else wgrad_dispatch<bf16>(a, s);
}

}  // namespace seg
