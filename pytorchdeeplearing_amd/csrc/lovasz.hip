// Lovasz hinge / Lovasz "softmax" losses (reference: model/lovasz.py:20-141 through model/losses.py:235-242 BinaryLovaszLoss and
// :462-473 LovaszLoss, both with per_image = False and no ignore index).
//
//   binary (C == 1): errors e_i = 1 - z_i * (2 y_i - 1) over ALL N*V voxels, sorted descending; with the labels in that order,
//                    g_k = J_k - J_{k-1}, J_k = 1 - (G - c_k) / (G + (k + 1 - c_k)), c = inclusive cumsum of the sorted labels, G = c_last;
//                    loss = sum_k relu(e_(k)) * g_k;   d loss / d z_i = -(2 y_i - 1) * g_rank(i) * [e_i > 0]
//   multi (C > 1):   per PRESENT class c: e_i = |fg_i - x_{i,c}| (the reference passes the logits as `probas`: no soft-max), same sort,
//                    loss_c = sum_k e_(k) g_k; loss = mean over present classes;   d / d x_{i,c} = -sign(fg_i - x_{i,c}) * g_rank(i) / #present
//
// The sort is a library radix sort (rocPRIM, key = error, value = voxel index), the label cumsum a library scan; the error, the
// Jaccard-gradient, the dot product and the gradient scatter are the kernels below.  The host-checker build (SEG_EMU) has no rocPRIM:
// it sorts and scans on the host (test infrastructure only).
#include "kernels.h"

#ifndef SEG_EMU
#include <cstring>
#include <rocprim/rocprim.hpp>
#else
#include <algorithm>
#include <numeric>
#include <vector>
#endif

namespace seg {
namespace {

__global__ __launch_bounds__(256) void lov_keys_kernel(const float* x, const void* target, int lt, int C, int cls, long long V, long long P,
                                                       float* keys, unsigned* idx) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < P; i += (long long)gridDim.x * 256) {
        const long long n = i / V, v = i % V;
        const float xv = x[(n * C + cls) * V + v];
        const int t = load_label(target, lt, i);
        float e;
        if (C == 1) e = 1.f - xv * (2.f * (float)t - 1.f);
        else e = fabsf(((t == cls) ? 1.f : 0.f) - xv);
        keys[i] = e;
        idx[i] = (unsigned)i;
    }
}

__global__ __launch_bounds__(256) void lov_gt_kernel(const unsigned* idx_sorted, const void* target, int lt, int C, int cls, long long P,
                                                     unsigned* gt) {
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < P; k += (long long)gridDim.x * 256) {
        const int t = load_label(target, lt, idx_sorted[k]);
        gt[k] = (C == 1) ? (unsigned)(t != 0) : (unsigned)(t == cls);
    }
}

__device__ __forceinline__ float lov_jaccard(float gts, long long k, unsigned ck) {
    // model/lovasz.py:25-28 in float32: intersection = gts - cumsum(gt), union = gts + cumsum(1 - gt)
    const float inter = gts - (float)ck;
    const float uni = gts + (float)(k + 1 - (long long)ck);
    return 1.f - inter / uni;
}

// acc[0] += loss contribution (fp64); dx[...] = d loss_c / d x (before the 1/#present of the multi-class mean); present[cls] = G > 0
__global__ __launch_bounds__(256) void lov_grad_kernel(const float* keys_sorted, const unsigned* idx_sorted, const unsigned* cum, const float* x,
                                                       int C, int cls, long long V, long long P, float* dx, double* acc, int* present) {
    __shared__ double red[4];
    const unsigned G = cum[P - 1];
    const float gts = (float)G;
    const bool skip = (C > 1 && G == 0);                   // classes='present': an absent class contributes nothing (lovasz.py:127-128)
    double part = 0.0;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < P; k += (long long)gridDim.x * 256) {
        const unsigned ck = cum[k], ck1 = k ? cum[k - 1] : 0u;
        const float jk = lov_jaccard(gts, k, ck);
        const float g = k ? jk - lov_jaccard(gts, k - 1, ck1) : jk;
        const float e = keys_sorted[k];
        const unsigned i = idx_sorted[k];
        const long long n = i / V, v = i % V;
        const long long o = (n * C + cls) * V + v;
        if (C == 1) {
            const float sgn = (ck != ck1) ? 1.f : -1.f;     // 2 y - 1 of the voxel at rank k
            part += (double)(fmaxf(e, 0.f) * g);
            dx[o] = e > 0.f ? -sgn * g : 0.f;
        } else if (skip) {
            dx[o] = 0.f;
        } else {
            const float u = ((ck != ck1) ? 1.f : 0.f) - x[o];
            part += (double)(e * g);
            dx[o] = u > 0.f ? -g : (u < 0.f ? g : 0.f);
        }
    }
    part = wave_sum_d(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = red[0] + red[1] + red[2] + red[3];
        if (s != 0.0) atomicAdd(acc + cls, s);
        if (blockIdx.x == 0) present[cls] = (C == 1 || G > 0) ? 1 : 0;
    }
}

__global__ __launch_bounds__(64) void lov_finalize_kernel(const double* acc, const int* present, int C, float* out1, float* inv_cnt) {
    if (threadIdx.x) return;
    double tot = 0.0;
    int cnt = 0;
    for (int c = 0; c < C; ++c) if (present[c]) { tot += acc[c]; ++cnt; }
    const double inv = cnt ? 1.0 / cnt : 0.0;             // no class present: mean of an empty list; the reference raises there
    out1[0] = (float)(tot * inv);
    *inv_cnt = (float)inv;
}

__global__ __launch_bounds__(256) void lov_scale_kernel(float* dx, long long total, const float* inv_cnt) {
    const float s = *inv_cnt;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) dx[i] *= s;
}

struct LovWs {
    float* keys; float* keys_s; unsigned* idx; unsigned* idx_s; unsigned* gt; unsigned* cum;
    double* acc; int* present; float* inv_cnt; void* tmp; size_t tmp_bytes;
};
constexpr int LOV_MAXCLS = 16;
inline size_t a256(size_t v) { return (v + 255) / 256 * 256; }

size_t lov_tmp_bytes(long long P) {
#ifndef SEG_EMU
    size_t a = 0, b = 0;
    float* kf = nullptr; unsigned* ku = nullptr;
    if (rocprim::radix_sort_pairs_desc(nullptr, a, kf, kf, ku, ku, (size_t)P, 0, 32, (hipStream_t) nullptr) != hipSuccess) return (size_t)-1;
    if (rocprim::inclusive_scan(nullptr, b, ku, ku, (size_t)P, rocprim::plus<unsigned>(), (hipStream_t) nullptr) != hipSuccess) return (size_t)-1;
    return a256(a > b ? a : b);
#else
    (void)P;
    return 256;
#endif
}

LovWs lov_carve(void* ws, long long P) {
    char* p = (char*)ws;
    LovWs w;
    auto take = [&](size_t bytes) { void* r = p; p += a256(bytes); return r; };
    w.keys = (float*)take((size_t)P * 4); w.keys_s = (float*)take((size_t)P * 4);
    w.idx = (unsigned*)take((size_t)P * 4); w.idx_s = (unsigned*)take((size_t)P * 4);
    w.gt = (unsigned*)take((size_t)P * 4); w.cum = (unsigned*)take((size_t)P * 4);
    w.acc = (double*)take(LOV_MAXCLS * 8); w.present = (int*)take(LOV_MAXCLS * 4); w.inv_cnt = (float*)take(4);
    w.tmp = p;
    w.tmp_bytes = lov_tmp_bytes(P);
    return w;
}

}  // namespace

long long lovasz_ws_bytes(long long P) {
    if (P < 1 || P >= (1ll << 32)) return -1;
    const size_t t = lov_tmp_bytes(P);
    if (t == (size_t)-1) return -1;
    return (long long)(6 * a256((size_t)P * 4) + a256(LOV_MAXCLS * 8) + a256(LOV_MAXCLS * 4) + a256(4) + t);
}

int launch_lovasz(const float* x, const void* target, int lt, int N, int C, long long V, void* ws, float* out1, float* dx, hipStream_t s) {
    const long long P = (long long)N * V;
    LovWs w = lov_carve(ws, P);
    if (w.tmp_bytes == (size_t)-1) return -1;
    const dim3 grid((unsigned)(P / 256 + 1 < 2048 ? P / 256 + 1 : 2048));
    (void)hipMemsetAsync(w.acc, 0, a256(LOV_MAXCLS * 8) + a256(LOV_MAXCLS * 4), s);      // acc and present are adjacent
    for (int cls = 0; cls < C; ++cls) {
        hipLaunchKernelGGL(lov_keys_kernel, grid, dim3(256), 0, s, x, target, lt, C, cls, V, P, w.keys, w.idx);
#ifndef SEG_EMU
        size_t tb = w.tmp_bytes;
        if (rocprim::radix_sort_pairs_desc(w.tmp, tb, w.keys, w.keys_s, w.idx, w.idx_s, (size_t)P, 0, 32, s) != hipSuccess) return -1;
#else
        {
            std::vector<unsigned> order((size_t)P);
            std::iota(order.begin(), order.end(), 0u);
            std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return w.keys[a] > w.keys[b]; });
            for (long long k = 0; k < P; ++k) { w.keys_s[k] = w.keys[order[k]]; w.idx_s[k] = w.idx[order[k]]; }
        }
#endif
        hipLaunchKernelGGL(lov_gt_kernel, grid, dim3(256), 0, s, (const unsigned*)w.idx_s, target, lt, C, cls, P, w.gt);
#ifndef SEG_EMU
        tb = w.tmp_bytes;
        if (rocprim::inclusive_scan(w.tmp, tb, w.gt, w.cum, (size_t)P, rocprim::plus<unsigned>(), s) != hipSuccess) return -1;
#else
        { unsigned run = 0; for (long long k = 0; k < P; ++k) { run += w.gt[k]; w.cum[k] = run; } }
#endif
        hipLaunchKernelGGL(lov_grad_kernel, grid, dim3(256), 0, s, (const float*)w.keys_s, (const unsigned*)w.idx_s, (const unsigned*)w.cum, x, C,
                           cls, V, P, dx, w.acc, w.present);
    }
    hipLaunchKernelGGL(lov_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)w.acc, (const int*)w.present, C, out1, w.inv_cnt);
    if (C > 1) hipLaunchKernelGGL(lov_scale_kernel, grid, dim3(256), 0, s, dx, (long long)N * C * V, (const float*)w.inv_cnt);
    return 0;
}

}  // namespace seg
