// Memory-bound helpers of the segmentation engine: layout ingest, weight packing, MaxPool 2^d,
// the seven losses + Dice/IoU metrics (model/losses.py, model/metric.py of the reference), fused
// Adam/AdamW (torch.optim semantics, model/modelVNet.py:548, model/modelUnet.py:849) and the
// channel-dropout multiplier generator.
#include "kernels.h"

namespace seg {
namespace {

inline int ew_blocks(long long total_threads, int cap = 16384) {
    long long b = (total_threads + 255) / 256;
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------- ingest: fp32 NC[D]HW -> T N[D]HWC
template <class T>
__global__ __launch_bounds__(256) void ingest_kernel(const float* x, T* out, int N, int C, long long V, int Csrc, StepRider rd) {
    if (blockIdx.x == 0 && threadIdx.x == 0) step_rider_run(rd);
    // Csrc < C: the image tensor is zero-padded to C channels (multi-channel 3-D inputs run through the 16-channel halo convs)
    const long long total = (long long)N * V * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long long nv = i / C;
        const long long n = nv / V, v = nv % V;
        out[i] = from_f<T>(c < Csrc ? x[(n * Csrc + c) * V + v] : 0.f);
    }
}

// one image channel: NC[D]HW and N[D]HWC are the same memory - a plain conversion, 8 elements per thread (two 16-B loads, one wide store)
template <class T>
__global__ __launch_bounds__(256) void ingest1_kernel(const float* x, T* out, long long total8, StepRider rd) {
    if (blockIdx.x == 0 && threadIdx.x == 0) step_rider_run(rd);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long long)gridDim.x * 256) {
        const vec<float, 4> a = *(const vec<float, 4>*)(x + i * 8), b = *(const vec<float, 4>*)(x + i * 8 + 4);
        vec<T, 8> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = from_f<T>(a[j]); o[4 + j] = from_f<T>(b[j]); }
        store8(out + i * 8, o);
    }
}

// ---------------------------------------------------------------- weight packing
// One workgroup per destination row (R1 x R2 rows of Kpad elements).  The row's K = T*Cc source elements are read in
// SOURCE order (the unit-stride index fastest: runs of T consecutive floats for PyTorch's [..][k^d] weight layout) into
// LDS and written out as one contiguous run.  The first version walked the destination order and read the fp32
// master weights with a 108-B stride: 670 MB of HBM fetches for a 38 MB re-layout (PMC, profiles/r01_pmc_*).
constexpr int PACK_MAXK = 27 * 256 + 32;
template <class T>
__global__ __launch_bounds__(256) void pack_kernel(const PackDesc* descs, StepRider rd) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) step_rider_run(rd);
    __shared__ float row_s[PACK_MAXK];
    const PackDesc d = descs[blockIdx.y];
    const long long rows = (long long)d.R1 * d.R2;
    const int K = d.T * d.Cc;
    const int Cs = (d.csrc > 0 && d.csrc < d.Cc) ? d.csrc : d.Cc;          // channels the source holds; the rest of the layout is zero
    T* dst = (T*)d.dst;
    const bool t_fast = d.sT == 1 || d.sC != 1;         // which source index is contiguous
    if (K > PACK_MAXK - 32) {                            // rows too long for the LDS row buffer: destination-order walk
        const long long total = rows * d.Kpad;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int k = (int)(i % d.Kpad);
            const long long row = i / d.Kpad;
            float v = 0.f;
            if (k < K) {
                const int t = k / d.Cc, c = k % d.Cc;
                const int tt = d.flipT ? (d.T - 1 - t) : t;
                if (c < Cs) v = d.src[(row / d.R2) * d.s1 + (row % d.R2) * d.s2 + tt * d.sT + c * d.sC];
            }
            dst[i] = from_f<T>(v);
        }
        return;
    }
    for (long long row = blockIdx.x; row < rows; row += gridDim.x) {
        const long long r1 = row / d.R2, r2 = row % d.R2;
        const float* src = d.src + r1 * d.s1 + r2 * d.s2;
        if (Cs < d.Cc) {
            for (int j = threadIdx.x; j < K; j += 256) row_s[j] = 0.f;
            __syncthreads();
        }
        for (int j = threadIdx.x; j < d.T * Cs; j += 256) {
            const int t = t_fast ? j % d.T : j / Cs, c = t_fast ? j / d.T : j % Cs;
            const int tt = d.flipT ? (d.T - 1 - t) : t;
            row_s[t * d.Cc + c] = src[tt * d.sT + c * d.sC];
        }
        __syncthreads();
        if (d.frag == 2) {
            // fragment-major over the FLAT k = t*Cc + c axis (Cc = 16: two taps per 32-deep MFMA step), zero-padded to Kpad
            for (int k8 = threadIdx.x; k8 < d.Kpad / 8; k8 += 256) {
                vec<T, 8> v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = from_f<T>(k8 * 8 + j < K ? row_s[k8 * 8 + j] : 0.f);
                const long long f = (long long)(k8 >> 2) * (rows >> 4) + (row >> 4);
                store8(dst + (f * 64 + (k8 & 3) * 16 + (row & 15)) * 8, v);
            }
        } else if (d.frag == 3) {
            // Cc == 16, T == 27 (conv3x16r_kernel): 15 steps = kh * 5 + pair, the two taps of a step differ in kd / kw only -
            // pairs 0..2: (kd, kw = 0 | 1), 3: (kd = 0 | 1, kw = 2), 4: (kd = 2, kw = 2 | zeros); lanes as in frag 2
            for (int k8 = threadIdx.x; k8 < 60; k8 += 256) {
                const int step = k8 >> 2, qq = k8 & 3, second = qq >> 1, pc = qq & 1, kh = step / 5, p = step % 5;
                const int kd = p < 3 ? p : (p == 3 ? second : 2), kw = p < 3 ? second : 2;
                const int t = (p == 4 && second) ? -1 : kd * 9 + kh * 3 + kw;
                vec<T, 8> v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = from_f<T>(t >= 0 ? row_s[t * 16 + pc * 8 + j] : 0.f);
                const long long f = (long long)step * (rows >> 4) + (row >> 4);
                store8(dst + (f * 64 + qq * 16 + (row & 15)) * 8, v);
            }
        } else if (d.frag) {
            // fragment-major: the 8 consecutive channels (t, c8*8 ..) of this row are lane 16*(c8%4) + row%16 of the
            // (chunk c8/4, tap t, tile row/16) fragment
            for (int k8 = threadIdx.x; k8 < K / 8; k8 += 256) {
                const int t = k8 / (d.Cc / 8), c8 = k8 % (d.Cc / 8);
                vec<T, 8> v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = from_f<T>(row_s[t * d.Cc + c8 * 8 + j]);
                const long long f = ((long long)(c8 >> 2) * d.T + t) * (rows >> 4) + (row >> 4);
                store8(dst + (f * 64 + (c8 & 3) * 16 + (row & 15)) * 8, v);
            }
        } else {
            for (int k = threadIdx.x; k < d.Kpad; k += 256) dst[row * d.Kpad + k] = from_f<T>(k < K ? row_s[k] : 0.f);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- MaxPool (kernel = stride = p)
template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolArgs a) {
    const int OD = a.D / a.pd, OH = a.H / a.ph, OW = a.W / a.pw, CPR = a.C / 8;
    const long long total = (long long)a.N * OD * OH * OW * CPR;
    const T* in = (const T*)a.in;
    T* out = (T*)a.out;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cc = (int)(i % CPR);
        long long o = i / CPR;
        const int ow = (int)(o % OW); o /= OW;
        const int oh = (int)(o % OH); o /= OH;
        const int od = (int)(o % OD);
        const long long n = o / OD;
        float best[8];
        bool first = true;
        for (int dz = 0; dz < a.pd; ++dz)
            for (int dy = 0; dy < a.ph; ++dy)
                for (int dx = 0; dx < a.pw; ++dx) {
                    const long long vox = ((n * a.D + od * a.pd + dz) * a.H + oh * a.ph + dy) * a.W + ow * a.pw + dx;
                    const vec<T, 8> x = load8(in + vox * a.C + cc * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xv = to_f(x[j]);
                        if (first || xv > best[j]) best[j] = xv;
                    }
                    first = false;
                }
        vec<T, 8> r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = from_f<T>(best[j]);
        store8(out + i * 8, r);
    }
}

// gradient goes to the FIRST maximum of each window in (d,h,w) scan order (ATen max_pool backward)
template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolArgs a) {
    const int OD = a.D / a.pd, OH = a.H / a.ph, OW = a.W / a.pw, CPR = a.C / 8;
    const long long total = (long long)a.N * OD * OH * OW * CPR;
    const T* in = (const T*)a.in;
    const T* dout = (const T*)a.dout;
    T* din = (T*)a.din;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cc = (int)(i % CPR);
        long long o = i / CPR;
        const int ow = (int)(o % OW); o /= OW;
        const int oh = (int)(o % OH); o /= OH;
        const int od = (int)(o % OD);
        const long long n = o / OD;
        float best[8];
        int arg[8];
        int pos = 0;
        for (int dz = 0; dz < a.pd; ++dz)
            for (int dy = 0; dy < a.ph; ++dy)
                for (int dx = 0; dx < a.pw; ++dx, ++pos) {
                    const long long vox = ((n * a.D + od * a.pd + dz) * a.H + oh * a.ph + dy) * a.W + ow * a.pw + dx;
                    const vec<T, 8> x = load8(in + vox * a.C + cc * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float xv = to_f(x[j]);
                        if (pos == 0 || xv > best[j]) { best[j] = xv; arg[j] = pos; }
                    }
                }
        const vec<T, 8> g = load8(dout + i * 8);
        pos = 0;
        for (int dz = 0; dz < a.pd; ++dz)
            for (int dy = 0; dy < a.ph; ++dy)
                for (int dx = 0; dx < a.pw; ++dx, ++pos) {
                    const long long vox = ((n * a.D + od * a.pd + dz) * a.H + oh * a.ph + dy) * a.W + ow * a.pw + dx;
                    vec<T, 8> r;
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] = (arg[j] == pos) ? g[j] : from_f<T>(0.f);
                    store8(din + vox * a.C + cc * 8, r);
                }
    }
}

// ---------------------------------------------------------------- losses + metrics
constexpr int MAXCLS = 16;                  // class cap (reference example.py:118,134,191 builds the multi-class wrappers with numclass=16);
                                            // the kernels are instantiated with a register bound MC = 8 (C <= 8) or 16
constexpr int S_GLOBAL = 0;                 // [0]=sum p*y [1]=sum p [2]=sum y [3]=sum bce|nll [4]=sum focal [5]=samples behind these sums
constexpr int S_COUNT = 5;                  // written by the fold (phase 0/1), SUM-all-reduced with the rest: the global sample count on the device
constexpr int S_CLASS = 8;                  // + 4*c : I_c, P_c (MutilSSLoss: sum p_c^2), Y_c, sum y_c p_c^2 (MutilSSLoss only)
constexpr int CLS_NV = 4;
constexpr int S_METRIC = S_CLASS + CLS_NV * MAXCLS;   // + ((n*C + c)*3) : inter, msum, ysum of thresholded masks
inline __host__ __device__ int s_coef(int N, int C) { return S_METRIC + 3 * N * C; }   // + 4 + 2*MAXCLS coefficients
}  // namespace
__host__ __device__ size_t loss_sums_count(int N, int C) { return (size_t)s_coef(N, C) + 4 + 2 * MAXCLS; }
namespace {

__device__ __forceinline__ float bce_with_logits(float z, float y) {
    return fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z)));
}

// grid = (slabs, N); one block reduces LOSS_VPB voxels of one sample, four voxels in flight per thread.  The transcendental
// terms are only evaluated for the loss kinds that use their sums (Dice-type losses need p alone); partial sums go through a
// wave reduction, one LDS fold over the four waves and ONE set of fp64 atomics per block.
constexpr int LOSS_VPB = 256 * 16;
constexpr int LOSS_NVAL = 5 + (CLS_NV + 3) * MAXCLS;
template <int MC>
__global__ __launch_bounds__(256) void loss_reduce_kernel(LossArgs a) {
    __shared__ float part[4][LOSS_NVAL];
    const int n = blockIdx.y, tid = threadIdx.x, C = a.C;
    const long long v0 = (long long)blockIdx.x * LOSS_VPB;
    const long long v1 = (v0 + LOSS_VPB < a.V) ? v0 + LOSS_VPB : a.V;
    float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float cls[MC][CLS_NV], met[MC][3];
#pragma unroll
    for (int c = 0; c < MC; ++c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) met[c][j] = 0.f;
#pragma unroll
        for (int j = 0; j < CLS_NV; ++j) cls[c][j] = 0.f;
    }
    if (C == 1) {
        const bool need_bce = a.kind == L_BIN_CE || a.kind == L_BIN_FOCAL || a.kind == L_BIN_CE_DICE;
        const bool need_focal = a.kind == L_BIN_FOCAL;
        const bool need_ss = a.kind == L_BIN_SS;             // sensitivity-specificity: slots 3 / 4 carry sum p^2 and sum y*p^2 instead
        const long long base = (long long)n * a.V;
        for (long long v = v0 + tid; v < v1; v += 1024) {
            float z[4], y[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long vv = v + u * 256;
                ok[u] = vv < v1;
                z[u] = ok[u] ? a.logits[base + vv] : 0.f;
                y[u] = ok[u] ? (float)load_label(a.target, a.label_type, base + vv) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!ok[u]) continue;
                const float p = a.kind == L_BIN_MCC ? z[u] : 1.f / (1.f + expf(-z[u]));      // MCC_Loss takes probabilities
                g[0] += p * y[u]; g[1] += p; g[2] += y[u];
                if (need_ss) { g[3] += p * p; g[4] += y[u] * p * p; }
                if (need_bce) {
                    const float b = bce_with_logits(z[u], y[u]);
                    g[3] += b;
                    if (need_focal) {
                        const float pt = expf(-b);
                        g[4] += a.focal_alpha * powf(1.f - pt, a.focal_gamma) * b;
                    }
                }
                const float mk = p > 0.5f ? 1.f : 0.f;
                met[0][0] += mk * y[u]; met[0][1] += mk; met[0][2] += y[u];
            }
        }
    } else {
        const bool need_focal = a.kind == L_MC_FOCAL;
        const bool mc_ss = a.kind == L_MC_SS;
        for (long long v = v0 + tid; v < v1; v += 256) {
            const int t = load_label(a.target, a.label_type, (long long)n * a.V + v);
            float z[MC], mx = -3.0e38f;
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                z[c] = (c < C) ? a.logits[((long long)n * C + c) * a.V + v] : -3.0e38f;
                mx = fmaxf(mx, z[c]);
            }
            float se = 0.f, e[MC];
#pragma unroll
            for (int c = 0; c < MC; ++c) { e[c] = (c < C) ? expf(z[c] - mx) : 0.f; se += e[c]; }
            const float inv = 1.f / se;
            const float lse = mx + logf(se);
            float zt = 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) {
                if (c < C) {
                    const float p = e[c] * inv;
                    const float y = (c == t) ? 1.f : 0.f;
                    if (c == t) zt = z[c];
                    cls[c][0] += y * p; cls[c][1] += mc_ss ? p * p : p; cls[c][2] += y;
                    if (mc_ss) cls[c][3] += y * p * p;
                    const float mk = p > 0.5f ? 1.f : 0.f;
                    met[c][0] += mk * y; met[c][1] += mk; met[c][2] += y;
                }
            }
            const float nll = lse - zt;
            g[3] += nll;
            if (need_focal) {
                const float pt = expf(-nll);
                g[4] += powf(1.f - pt, a.focal_gamma) * nll;
            }
        }
    }
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const float s = wave_sum(g[j]);
        if (lane == 0) part[wv][j] = s;
    }
#pragma unroll
    for (int c = 0; c < MC; ++c) {
        if (c < C) {
#pragma unroll
            for (int j = 0; j < CLS_NV; ++j) {
                const float s = (C > 1 && (j < 3 || a.kind == L_MC_SS)) ? wave_sum(cls[c][j]) : 0.f;
                if (lane == 0) part[wv][5 + CLS_NV * c + j] = s;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float m = wave_sum(met[c][j]);
                if (lane == 0) part[wv][5 + CLS_NV * MAXCLS + 3 * c + j] = m;
            }
        }
    }
    __syncthreads();
    double* sums = a.sums + (long long)(blockIdx.x % STAT_REP) * loss_sums_count(a.N, C);
    if (tid < LOSS_NVAL) {
        const int k = tid < 5 ? 0 : (tid < 5 + CLS_NV * MAXCLS ? 1 : 2);
        const int r = k == 0 ? tid : (k == 1 ? tid - 5 : tid - 5 - CLS_NV * MAXCLS);      // CLS_NV*c + j / 3*c + j for the per-class blocks
        if (k == 0 || r < (k == 1 ? CLS_NV : 3) * C) {
            const float s = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
            if (s != 0.f) {
                if (k == 0) atomicAdd(sums + S_GLOBAL + r, (double)s);
                else if (k == 1) atomicAdd(sums + S_CLASS + r, (double)s);
                else atomicAdd(sums + S_METRIC + (long long)n * C * 3 + r, (double)s);
            }
        }
    }
}

// single block: scalar loss, metrics and the coefficients consumed by the backward pass
__global__ __launch_bounds__(64) void loss_finalize_kernel(LossArgs a) {
    const int C = a.C, N = a.N;
    double* S = a.sums;
    const int cnt_all = s_coef(N, C);
    if (a.phase != 2) {
        for (int i = threadIdx.x; i < cnt_all; i += 64) {          // fold the replicas into copy 0
            double t = S[i];
            for (int rep = 1; rep < STAT_REP; ++rep) t += S[(long long)rep * loss_sums_count(N, C) + i];
            S[i] = i == S_COUNT ? (double)N : t;
        }
        __syncthreads();
    }
    // phase 1 stops here: the caller SUM-all-reduces the first S_METRIC doubles of copy 0 across ranks (exact
    // global-batch loss, SURVEY section 8e mode ii), then runs phase 2 with the global sample count
    if (threadIdx.x != 0 || a.phase == 1) return;
    double* K = S + s_coef(N, C);
    const double smooth = 1e-5, eps = 1e-7;
    // sample count behind the batch-global sums: the caller's n_global, or (phase 2 with n_global == 0) the count that was exchanged
    // together with the sums - no host read on unequal shards
    const double Ntot = (a.n_global > 0 ? (double)a.n_global : (a.phase == 2 ? S[S_COUNT] : (double)N)) * (double)a.V;
    double loss = 0.0;
    for (int i = 0; i < 4 + 2 * MAXCLS; ++i) K[i] = 0.0;
    if (C == 1) {
        const double I = S[0], P = S[1], Y = S[2];
        double D = P + Y + smooth, c0 = 0.0;
        if (D < eps) D = eps; else c0 = (2.0 * I + smooth) / (D * D);
        const double dice = 1.0 - (2.0 * I + smooth) / D;
        const double bce = S[3] / Ntot, focal = S[4] / Ntot;
        if (a.kind == L_BIN_DICE) loss = dice;
        else if (a.kind == L_BIN_CE) loss = bce;
        else if (a.kind == L_BIN_FOCAL) loss = focal;
        else loss = bce + dice;
        K[0] = -2.0 / D;      // d dice / d p_i = K0*y_i + K1
        K[1] = c0;
        K[2] = 1.0 / Ntot;
        // losses that are other ratios of the same three sums: d L / d p_i = K0*y_i + K1 with K0 = dL/dI, K1 = dL/dP
        if (a.kind == L_BIN_JACCARD) {              // model/losses.py:9-30
            double U = P + Y - I + smooth;
            const bool cl = U < eps;
            if (cl) U = eps;
            loss = 1.0 - (I + smooth) / U;
            K[0] = cl ? -1.0 / U : -(1.0 / U + (I + smooth) / (U * U));
            K[1] = cl ? 0.0 : (I + smooth) / (U * U);
        } else if (a.kind == L_BIN_ELDICE) {        // model/losses.py:56-74: clamp((-log(dsc + smooth))^0.3, 0, 2)
            const double dsc = (2.0 * I + smooth) / D;
            const double t = -log(dsc + smooth), v = pow(t, 0.3);
            loss = v < 0.0 ? 0.0 : (v > 2.0 ? 2.0 : v);               // NaN (dsc + smooth > 1) propagates as in the reference
            const double g = (v >= 0.0 && v <= 2.0) ? 0.3 * pow(t, -0.7) * (-1.0 / (dsc + smooth)) : 0.0;
            K[0] = g * 2.0 / D;
            K[1] = (P + Y + smooth < eps) ? 0.0 : -g * (2.0 * I + smooth) / (D * D);
        } else if (a.kind == L_BIN_SS) {            // model/losses.py:77-99: r*sum((p-y)^2 y)/(smooth+Y) + (1-r)*sum((p-y)^2 (1-y))/(smooth+N-Y), r = 0.1
            // y in {0,1}: sum (p-y)^2 y = sum y p^2 - 2 I + Y;  sum (p-y)^2 (1-y) = sum p^2 - sum y p^2
            const double r = 0.1, P2 = S[3], YP2 = S[4];
            const double ds = smooth + Y, db = smooth + (Ntot - Y);
            loss = r * (YP2 - 2.0 * I + Y) / ds + (1.0 - r) * (P2 - YP2) / db;
            K[0] = 2.0 * r / ds;                     // d loss / d p_i = K0 * y_i * (p_i - 1) + K1 * p_i * (1 - y_i)
            K[1] = 2.0 * (1.0 - r) / db;
        } else if (a.kind == L_BIN_MCC) {           // model/losses.py:200-232 with torch.add(a, 1, b) read as a + 1*b (the torch 1.x signature)
            // tp = I, fp = P - I, fn = Y - I, tn = Nt - P - Y + I  =>  tp*tn - fp*fn = I*Nt - P*Y; (tp+fp)(tp+fn)(tn+fp)(tn+fn) = P*Y*(Nt-Y)*(Nt-P)
            const double num = I * Ntot - P * Y;
            const double rad = P * Y * (Ntot - Y) * (Ntot - P);
            const double den = sqrt(rad > 0.0 ? rad : 0.0);
            loss = 1.0 - num / (den + 1.0);
            const double dden_dP = den > 0.0 ? Y * (Ntot - Y) * (Ntot - 2.0 * P) / (2.0 * den) : 0.0;
            K[0] = -Ntot / (den + 1.0);              // d loss / d p_i = K0 * y_i + K1 (the input IS p: no sigmoid factor in the backward)
            K[1] = -(-Y * (den + 1.0) - num * dden_dP) / ((den + 1.0) * (den + 1.0));
        } else if (a.kind == L_BIN_TVERSKY) {       // model/losses.py:102-126: alpha = 0.3 (false positives), beta = 0.7 (false negatives)
            const double al = 0.3, be = 0.7;
            const double den = I + al * (P - I) + be * (Y - I) + smooth;
            const double v = 1.0 - (I + smooth) / den;
            loss = v < 0.0 ? 0.0 : (v > 2.0 ? 2.0 : v);
            const bool in = v >= 0.0 && v <= 2.0;
            K[0] = in ? -(den - (I + smooth) * (1.0 - al - be)) / (den * den) : 0.0;
            K[1] = in ? (I + smooth) * al / (den * den) : 0.0;
        }
    } else {
        int cnt = 0;
        for (int c = 0; c < C; ++c) cnt += S[S_CLASS + CLS_NV * c + 2] > 0.0;
        double dl = 0.0;
        for (int c = 0; c < C; ++c) {
            const double I = S[S_CLASS + CLS_NV * c], P = S[S_CLASS + CLS_NV * c + 1], Y = S[S_CLASS + CLS_NV * c + 2];
            const double al = a.class_alpha ? (double)a.class_alpha[c] : 1.0;
            const double D = Y + P + smooth;
            double dice = (2.0 * I + smooth) / D;
            const bool present = Y > 0.0;
            double ac = 0.0, bc = 0.0;
            if (present && dice >= eps) {
                ac = -al / cnt * 2.0 / D;
                bc = al / cnt * (2.0 * I + smooth) / (D * D);
            }
            if (dice < eps) dice = eps;
            if (present) dl += -dice * al / cnt;
            K[4 + c] = ac;                 // d L / d p_c(v) = ac*y_c(v) + bc
            K[4 + MAXCLS + c] = bc;
        }
        if (a.kind == L_MC_DICE) loss = dl;
        else if (a.kind == L_MC_CE) loss = S[3] / Ntot;
        else if (a.kind == L_MC_CE_DICE) loss = dl + S[3] / Ntot;         // model/losses.py:328-342
        else if (a.kind == L_MC_ELDICE) {
            // model/losses.py:345-382: dice_c (0 for absent classes) * alpha_c -> clamp(sum_c (-log(. + smooth))^0.3 / present, 0, 2)
            double tot = 0.0;
            for (int c = 0; c < C; ++c) {
                const double I = S[S_CLASS + CLS_NV * c], P = S[S_CLASS + CLS_NV * c + 1], Y = S[S_CLASS + CLS_NV * c + 2];
                const double al = a.class_alpha ? (double)a.class_alpha[c] : 1.0;
                const double D = Y + P + smooth;
                double dice = (2.0 * I + smooth) / D;
                const bool live = Y > 0.0 && dice >= eps;
                if (dice < eps) dice = eps;
                const double d = (Y > 0.0 ? dice : 0.0) * al;
                const double t = -log(d + smooth);
                tot += pow(t, 0.3);
                const double g = live ? 0.3 * pow(t, -0.7) * (-1.0 / (d + smooth)) * al / cnt : 0.0;     // d(sum/cnt) / d dice_c
                K[4 + c] = g * 2.0 / D;
                K[4 + MAXCLS + c] = -g * (2.0 * I + smooth) / (D * D);
            }
            const double v = tot / cnt;
            loss = v < 0.0 ? 0.0 : (v > 2.0 ? 2.0 : v);
            if (!(v >= 0.0 && v <= 2.0)) for (int c = 0; c < C; ++c) { K[4 + c] = 0.0; K[4 + MAXCLS + c] = 0.0; }
        } else if (a.kind == L_MC_TVERSKY) {
            // model/losses.py:421-459 with self.beta set by the caller (the class never defines it; oracle/make_golden.py:LOSS_REPAIRS):
            // -(tp + s) / (tp + alpha_c fp + beta fn + s) per present class, times alpha_c, over the present-class count
            const double be = (double)a.focal_gamma;                  // beta rides in the focal_gamma slot of the call
            double tot = 0.0;
            for (int c = 0; c < C; ++c) {
                const double I = S[S_CLASS + CLS_NV * c], P = S[S_CLASS + CLS_NV * c + 1], Y = S[S_CLASS + CLS_NV * c + 2];
                const double al = a.class_alpha ? (double)a.class_alpha[c] : 1.0;
                const double den = I + al * (P - I) + be * (Y - I) + smooth;
                const bool present = Y > 0.0;
                if (present) tot += -(I + smooth) / den * al;
                const double w = present ? al / cnt : 0.0;
                K[4 + c] = -w * (den - (I + smooth) * (1.0 - al - be)) / (den * den);     // d/dI
                K[4 + MAXCLS + c] = w * (I + smooth) * al / (den * den);                 // d/dP
            }
            loss = tot / cnt;
        } else if (a.kind == L_MC_SS) {
            // model/losses.py:385-418 with self.r set by the caller (never defined in the class): per class
            // r * sum((y-p)^2 y) / (Y + s) + (1-r) * sum((y-p)^2 (1-y)) / (Y + s)   [both denominators are sum(y_true) in the reference]
            const double r = (double)a.focal_gamma;                   // r rides in the focal_gamma slot of the call
            double tot = 0.0;
            for (int c = 0; c < C; ++c) {
                const double I = S[S_CLASS + CLS_NV * c], P2 = S[S_CLASS + CLS_NV * c + 1], Y = S[S_CLASS + CLS_NV * c + 2], YP2 = S[S_CLASS + CLS_NV * c + 3];
                const double al = a.class_alpha ? (double)a.class_alpha[c] : 1.0;
                const bool present = Y > 0.0;
                const double ssv = (r * (YP2 - 2.0 * I + Y) + (1.0 - r) * (P2 - YP2)) / (Y + smooth);
                if (present) tot += ssv * al;
                const double w = present ? al / cnt / (Y + smooth) : 0.0;
                K[4 + c] = 2.0 * r * w;                    // d L / d p_c(v) = K_a * y (p - 1) + K_b * p (1 - y)
                K[4 + MAXCLS + c] = 2.0 * (1.0 - r) * w;
            }
            loss = tot / cnt;
        } else loss = S[4] / Ntot;
        K[2] = 1.0 / Ntot;
    }
    // metrics: model/metric.py:146-181 (binary: class 0; multi-class: classes 1..C-1)
    double dsum = 0.0, isum = 0.0;
    const int c_lo = (C == 1) ? 0 : 1;
    for (int c = c_lo; c < C; ++c) {
        double dc = 0.0, ic = 0.0;
        for (int n = 0; n < N; ++n) {
            const double* m = S + S_METRIC + ((long long)n * C + c) * 3;
            dc += (2.0 * m[0] + 1e-5) / (m[1] + m[2] + 1e-5);
            ic += (m[0] + 1e-5) / (m[1] + m[2] - m[0] + 1e-5);
        }
        dsum += dc / N; isum += ic / N;
    }
    const int nc = (C == 1) ? 1 : (C - 1);
    a.out[0] = (float)loss;
    a.out[1] = (float)(dsum / nc);
    a.out[2] = (float)(isum / nc);
}

template <int MC>
__global__ __launch_bounds__(256) void loss_backward_kernel(LossArgs a) {
    const int C = a.C;
    const double* K = a.sums + s_coef(a.N, C);
    const long long total = (long long)a.N * a.V;
    const float gs = a.grad_scale;
    if (C == 1) {
        const float k0 = (float)K[0], k1 = (float)K[1], kn = (float)K[2];
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const float z = a.logits[i];
            const float y = (float)load_label(a.target, a.label_type, i);
            const float p = 1.f / (1.f + expf(-z));
            float dz = 0.f;
            if (a.kind == L_BIN_MCC) dz = k0 * y + k1;           // gradient with respect to the probability input itself
            else if (a.kind == L_BIN_SS) dz += (k0 * y * (p - 1.f) + k1 * p * (1.f - y)) * p * (1.f - p);
            else if (a.kind == L_BIN_DICE || a.kind == L_BIN_CE_DICE || a.kind >= L_BIN_JACCARD) dz += (k0 * y + k1) * p * (1.f - p);
            if (a.kind == L_BIN_CE || a.kind == L_BIN_CE_DICE) dz += (p - y) * kn;
            if (a.kind == L_BIN_FOCAL) {
                const float b = bce_with_logits(z, y);
                const float pt = expf(-b), om = 1.f - pt;
                const float w = powf(om, a.focal_gamma) + a.focal_gamma * powf(om, a.focal_gamma - 1.f) * pt * b;
                dz += a.focal_alpha * w * (p - y) * kn;
            }
            a.dlogits[i] = dz * gs;
        }
    } else {
        const float kn = (float)K[2];
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long n = i / a.V, v = i % a.V;
            const int t = load_label(a.target, a.label_type, i);
            // every per-class loop is unrolled over the MC slots of the instantiation with the class count as a predicate: the arrays stay in registers
            // (a loop over the run-time class count indexes them dynamically and sends them to scratch: 78 us at 2 x 128^3 x 4 classes, round 5)
            float z[MC], p[MC], mx = -3.0e38f, se = 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) { z[c] = (c < C) ? a.logits[(n * C + c) * a.V + v] : -3.0e38f; if (c < C) mx = fmaxf(mx, z[c]); }
#pragma unroll
            for (int c = 0; c < MC; ++c) { p[c] = (c < C) ? expf(z[c] - mx) : 0.f; if (c < C) se += p[c]; }
            const float inv = 1.f / se;
#pragma unroll
            for (int c = 0; c < MC; ++c) p[c] *= inv;
            const bool dice_like = a.kind == L_MC_DICE || a.kind == L_MC_CE_DICE || a.kind == L_MC_ELDICE || a.kind == L_MC_TVERSKY ||
                                   a.kind == L_MC_SS;
            const bool ce_like = a.kind == L_MC_CE || a.kind == L_MC_FOCAL || a.kind == L_MC_CE_DICE;
            float dz[MC];
#pragma unroll
            for (int c = 0; c < MC; ++c) dz[c] = 0.f;
            if (dice_like) {
                float gsum = 0.f, gc[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    gc[c] = 0.f;
                    if (c < C) {
                        if (a.kind == L_MC_SS) gc[c] = (c == t) ? (float)K[4 + c] * (p[c] - 1.f) : (float)K[4 + MAXCLS + c] * p[c];
                        else gc[c] = (float)K[4 + c] * ((c == t) ? 1.f : 0.f) + (float)K[4 + MAXCLS + c];
                        gsum += p[c] * gc[c];
                    }
                }
#pragma unroll
                for (int c = 0; c < MC; ++c) dz[c] += p[c] * (gc[c] - gsum);
            }
            if (ce_like) {
                float w = kn;
                if (a.kind == L_MC_FOCAL) {
                    float zt = 0.f;
#pragma unroll
                    for (int c = 0; c < MC; ++c) if (c == t) zt = z[c];
                    const float nll = (mx + logf(se)) - zt;
                    const float pt = expf(-nll), om = 1.f - pt;
                    w *= powf(om, a.focal_gamma) + a.focal_gamma * powf(om, a.focal_gamma - 1.f) * pt * nll;
                }
#pragma unroll
                for (int c = 0; c < MC; ++c) dz[c] += w * (p[c] - ((c == t) ? 1.f : 0.f));
            }
#pragma unroll
            for (int c = 0; c < MC; ++c) if (c < C) a.dlogits[(n * C + c) * a.V + v] = dz[c] * gs;
        }
    }
}

// metric on probabilities: thresholded masks per (sample, class)
__global__ __launch_bounds__(256) void metric_reduce_kernel(const float* probs, const void* target, int lt, int N, int C, long long V, double* sums) {
    const int n = blockIdx.y, tid = threadIdx.x;
    const long long v0 = (long long)blockIdx.x * LOSS_VPB;
    const long long v1 = (v0 + LOSS_VPB < V) ? v0 + LOSS_VPB : V;
    const int c_lo = (C == 1) ? 0 : 1;
    for (int c = c_lo; c < C; ++c) {
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        for (long long v = v0 + tid; v < v1; v += 256) {
            const int t = load_label(target, lt, (long long)n * V + v);
            const float y = (C == 1) ? (float)t : ((t == c) ? 1.f : 0.f);
            const float mk = probs[((long long)n * C + c) * V + v] > 0.5f ? 1.f : 0.f;
            m0 += mk * y; m1 += mk; m2 += y;
        }
        m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2);
        if ((tid & 63) == 0) {
            double* d = sums + ((long long)n * C + c) * 3;
            atomicAdd(d, (double)m0); atomicAdd(d + 1, (double)m1); atomicAdd(d + 2, (double)m2);
        }
    }
}
__global__ __launch_bounds__(64) void metric_finalize_kernel(int N, int C, const double* sums, float* out2) {
    if (threadIdx.x != 0) return;
    double dsum = 0.0, isum = 0.0;
    const int c_lo = (C == 1) ? 0 : 1;
    for (int c = c_lo; c < C; ++c) {
        double dc = 0.0, ic = 0.0;
        for (int n = 0; n < N; ++n) {
            const double* m = sums + ((long long)n * C + c) * 3;
            dc += (2.0 * m[0] + 1e-5) / (m[1] + m[2] + 1e-5);
            ic += (m[0] + 1e-5) / (m[1] + m[2] - m[0] + 1e-5);
        }
        dsum += dc / N; isum += ic / N;
    }
    const int nc = (C == 1) ? 1 : (C - 1);
    out2[0] = (float)(dsum / nc);
    out2[1] = (float)(isum / nc);
}

// column sums of a [M][C] tensor (bias gradient of convs that are not followed by GroupNorm)
template <class T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, float* out, long long M, int C, long long rows_per_block) {
    __shared__ float red[256 * 8];
    const int tid = threadIdx.x, CPR = C / 8, G = 256 / CPR;
    const int cc = tid % CPR, g = tid / CPR;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    const long long m0 = (long long)blockIdx.x * rows_per_block;
    const long long m1 = (m0 + rows_per_block < M) ? m0 + rows_per_block : M;
    if (tid < G * CPR) {
        long long m = m0 + g;
        for (; m + 3ll * G < m1; m += 4ll * G) {             // four rows in flight per thread
            vec<T, 8> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = load8(x + (m + (long long)u * G) * C + cc * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += to_f(v[u][j]);
        }
        for (; m < m1; m += G) {
            const vec<T, 8> v = load8(x + m * C + cc * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += to_f(v[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid * 8 + j] = s[j];
    __syncthreads();
    for (int col = tid; col < C; col += 256) {
        float t = 0.f;
        for (int k = 0; k < G; ++k) t += red[(k * CPR + col / 8) * 8 + (col & 7)];
        atomicAdd(out + col, t);
    }
}

// ---------------------------------------------------------------- optimiser
__global__ __launch_bounds__(256) void grad_check_kernel(const float* g, long long n, int* found_inf) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.0e38f);
    }
    if (bad) atomicOr(found_inf, 1);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    if (a.found_inf && *a.found_inf) return;
    const int t = *a.step + 1;
    const float bc1 = 1.f - powf(a.beta1, (float)t), bc2 = 1.f - powf(a.beta2, (float)t);
    const float step_size = a.lr / bc1, rs_bc2 = 1.f / sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
        float p = a.p[i], g = a.g[i] * a.inv_scale;
        if (a.decoupled) p *= (1.f - a.lr * a.weight_decay);
        else if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
        const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
        a.m[i] = m; a.v[i] = v;
        a.p[i] = p - step_size * m / (sqrtf(v) * rs_bc2 + a.eps);
    }
}
// step counter += 1 unless the step was skipped; a skipped step is tallied in found_inf[1] (= state[2] of seg_adam_step: the host
// reads it every few dozen steps to back the loss scale off, engine.py)
__global__ __launch_bounds__(64) void adam_bump_kernel(int* step, int* found_inf, int tally) {
    if (threadIdx.x == 0) {
        if (!(found_inf && *found_inf)) *step += 1;
        else if (tally) found_inf[1] += 1;
    }
}

// ---------------------------------------------------------------- channel dropout multipliers
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* masks, long long total, float p, unsigned long long seed, const int* step) {
    const unsigned long long st = step ? (unsigned long long)*step : 0ull;
    const float keep = 1.f - p, inv = 1.f / keep;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const unsigned long long h = splitmix64(splitmix64(seed ^ (st * 0xD1B54A32D192ED03ull)) + (unsigned long long)i);
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
        masks[i] = (u < keep) ? inv : 0.f;
    }
}

}  // namespace

void launch_ingest(const float* x, void* out, int N, int C, long long V, int dtype, hipStream_t s, int Csrc, StepRider rd) {
    if (Csrc <= 0 || Csrc > C) Csrc = C;
    if (C == 1 && Csrc == 1 && dtype != DT_F32 && ((long long)N * V) % 8 == 0 && (((unsigned long long)x | (unsigned long long)out) & 15) == 0) {
        const long long total8 = (long long)N * V / 8;
        dim3 g1(ew_blocks(total8));
        if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(ingest1_kernel<f16>), g1, dim3(256), 0, s, x, (f16*)out, total8, rd);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(ingest1_kernel<bf16>), g1, dim3(256), 0, s, x, (bf16*)out, total8, rd);
        return;
    }
    dim3 grid(ew_blocks((long long)N * V * C));
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(ingest_kernel<float>), grid, dim3(256), 0, s, x, (float*)out, N, C, V, Csrc, rd);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(ingest_kernel<f16>), grid, dim3(256), 0, s, x, (f16*)out, N, C, V, Csrc, rd);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(ingest_kernel<bf16>), grid, dim3(256), 0, s, x, (bf16*)out, N, C, V, Csrc, rd);
}

// predict() post-processing on the device (modelVNet.py:670-676, modelUnet.py:672-680): probs planar fp32 [N][C][V] ->
// uint8 mask [N][V]: C == 1: (p > threshold) * scale; C > 1: index of the FIRST maximum over the class axis (np.argmax).
__global__ __launch_bounds__(256) void mask_kernel(const float* probs, unsigned char* out, int N, int C, long long V, float threshold, int scale) {
    const long long total = (long long)N * V;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i % V;
        const float* p = probs + n * C * V + v;
        int m;
        if (C == 1) {
            m = p[0] > threshold ? scale : 0;
        } else {
            float best = p[0];
            m = 0;
            for (int c = 1; c < C; ++c) {
                const float x = p[(long long)c * V];
                if (x > best) { best = x; m = c; }
            }
        }
        out[i] = (unsigned char)m;
    }
}
void launch_mask(const float* probs, unsigned char* out, int N, int C, long long V, float threshold, int scale, hipStream_t s) {
    hipLaunchKernelGGL(mask_kernel, dim3(ew_blocks((long long)N * V)), dim3(256), 0, s, probs, out, N, C, V, threshold, scale);
}

void launch_pack(const PackDesc* descs_dev, int ndesc, int max_elems, int dtype, hipStream_t s, StepRider rd) {
    (void)max_elems;
    // rows are strided over `wgs` workgroups per descriptor.  The 256-channel levels hold most of the bytes in descriptors of 256
    // rows: at 64 workgroups each one walked four 27 KB rows back to back (67 us per step, latency-bound)
    static const int wgs = 256;
    dim3 grid(wgs, ndesc);
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_kernel<float>), grid, dim3(256), 0, s, descs_dev, rd);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_kernel<f16>), grid, dim3(256), 0, s, descs_dev, rd);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(pack_kernel<bf16>), grid, dim3(256), 0, s, descs_dev, rd);
}

void launch_maxpool_fwd(const PoolArgs& a, int dtype, hipStream_t s) {
    dim3 grid(ew_blocks((long long)a.N * (a.D / a.pd) * (a.H / a.ph) * (a.W / a.pw) * (a.C / 8)));
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(maxpool_fwd_kernel<float>), grid, dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(maxpool_fwd_kernel<f16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(maxpool_fwd_kernel<bf16>), grid, dim3(256), 0, s, a);
}
void launch_maxpool_bwd(const PoolArgs& a, int dtype, hipStream_t s) {
    dim3 grid(ew_blocks((long long)a.N * (a.D / a.pd) * (a.H / a.ph) * (a.W / a.pw) * (a.C / 8)));
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(maxpool_bwd_kernel<float>), grid, dim3(256), 0, s, a);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(maxpool_bwd_kernel<f16>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(maxpool_bwd_kernel<bf16>), grid, dim3(256), 0, s, a);
}



int loss_shared_count() { return S_METRIC; }
void launch_loss_forward(const LossArgs& a, hipStream_t s) {
    if (a.phase == 2) {                       // sums already reduced, folded (and exchanged) by a phase-1 call
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, a);
        return;
    }
    if (!a.prezeroed) (void)hipMemsetAsync(a.sums, 0, loss_sums_count(a.N, a.C) * sizeof(double) * STAT_REP, s);
    dim3 grid(cdiv(a.V, LOSS_VPB), a.N);
    if (a.C <= 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(loss_reduce_kernel<8>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(loss_reduce_kernel<16>), grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, s, a);
}
void launch_loss_backward(const LossArgs& a, hipStream_t s) {
    const dim3 grid(ew_blocks((long long)a.N * a.V));
    if (a.C <= 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(loss_backward_kernel<8>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(loss_backward_kernel<16>), grid, dim3(256), 0, s, a);
}

void launch_metric(const float* probs, const void* target, int lt, int N, int C, long long V, double* sums, float* out2, hipStream_t s) {
    (void)hipMemsetAsync(sums, 0, sizeof(double) * 3 * N * C, s);
    hipLaunchKernelGGL(metric_reduce_kernel, dim3(cdiv(V, LOSS_VPB), N), dim3(256), 0, s, probs, target, lt, N, C, V, sums);
    hipLaunchKernelGGL(metric_finalize_kernel, dim3(1), dim3(64), 0, s, N, C, (const double*)sums, out2);
}

void launch_colsum(const void* x, float* out, long long M, int C, int dtype, hipStream_t s) {
    // every block ends in one atomic per column onto the SAME C addresses (~60 ns each, serialised): at most 512 blocks
    long long rpb = 4096;
    while (cdiv(M, rpb) > 512) rpb *= 2;
    dim3 grid(cdiv(M, rpb));
    if (dtype == DT_F32) hipLaunchKernelGGL(HIP_KERNEL_NAME(colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)x, out, M, C, rpb);
    else if (dtype == DT_F16) hipLaunchKernelGGL(HIP_KERNEL_NAME(colsum_kernel<f16>), grid, dim3(256), 0, s, (const f16*)x, out, M, C, rpb);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(colsum_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)x, out, M, C, rpb);
}

void launch_grad_check(const float* g, long long n, int* found_inf, hipStream_t s) {
    hipLaunchKernelGGL(grad_check_kernel, dim3(ew_blocks(n, 2048)), dim3(256), 0, s, g, n, found_inf);
}
void launch_adam(const AdamArgs& a, hipStream_t s, bool bump) {
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(a.n, 4096)), dim3(256), 0, s, a);
    if (bump) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(64), 0, s, a.step, a.found_inf, 1);
}


void launch_dropout_masks(float* masks, int L, int N, int ld, float p, unsigned long long seed, const int* step, hipStream_t s, bool bump) {
    const long long total = (long long)L * N * ld;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_blocks(total, 1024)), dim3(256), 0, s, masks, total, p, seed, step);
    if (step && bump) hipLaunchKernelGGL(adam_bump_kernel, dim3(1), dim3(64), 0, s, (int*)step, (int*)nullptr, 0);
}

}  // namespace seg
