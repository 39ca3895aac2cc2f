// SSIM / SSIM3D (reference: model/lossesSSIM.py:30-99 — Gaussian window 11, sigma 1.5, zero padding, one depth-wise window per channel).
//
//   mu_a = G * x_a,  e_ab = G * (x_a x_b),  s_ab = e_ab - mu_a mu_b
//   map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s11 + s22 + C2)),  C1 = 0.01^2, C2 = 0.03^2;  ssim = mean(map)
//
// The reference convolves with the full 11^d window (the outer product of the 1-D Gaussian); here the window is applied separably, one
// axis per pass (the same fp32 weights, different rounding order: ~1e-7 relative).  Planar fp32 tensors [planes = N*C][D][H][W].
// Forward: products -> 5 x nd blur passes -> one map pass that also leaves d map / d {mu1, mu2, e11, e22, e12} for the backward pass.
// Backward: the five derivative maps are blurred again (the zero-padded Gaussian is self-adjoint) and combined:
//   d ssim / d x1 = G*(g_mu1) + 2 x1 G*(g_e11) + x2 G*(g_e12)      (x2: symmetric)
#include "kernels.h"

namespace seg {
namespace {

constexpr int SSIM_MAXW = 15;
struct SsimWin { int n; float g[SSIM_MAXW]; };

__global__ __launch_bounds__(256) void ssim_prod_kernel(const float* x1, const float* x2, float* p11, float* p22, float* p12, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float a = x1[i], b = x2[i];
        p11[i] = a * a; p22[i] = b * b; p12[i] = a * b;
    }
}

// out[p][z][y][x] = sum_k g[k] * in[.. coordinate along `axis` + k - n/2 ..], zero outside the volume (F.conv padding = n // 2)
__global__ __launch_bounds__(256) void ssim_blur_kernel(const float* in, float* out, int planes, int D, int H, int W, int axis, SsimWin w) {
    const long long total = (long long)planes * D * H * W;
    const int len = axis == 0 ? D : (axis == 1 ? H : W);
    const long long str = axis == 0 ? (long long)H * W : (axis == 1 ? W : 1);
    const int half = w.n / 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)((i / str) % len);
        float s = 0.f;
        for (int k = 0; k < w.n; ++k) {
            const int cc = c + k - half;
            if (cc >= 0 && cc < len) s = fmaf(w.g[k], in[i + (long long)(k - half) * str], s);
        }
        out[i] = s;
    }
}

// map + derivative maps (in place over the five statistics) + per-sample sums of the map (fp64, one atomic per block and sample)
__global__ __launch_bounds__(256) void ssim_map_kernel(float* mu1, float* mu2, float* e11, float* e22, float* e12, long long per_sample,
                                                       double* sums, double* colsums, int W) {
    __shared__ double red[4];
    // column sums of the (N, W) result: gathered per workgroup in LDS and flushed with ONE global atomic per column and workgroup (one
    // global fp64 atomic per voxel on N * W addresses serialised); rows longer than the LDS table keep the per-voxel atomics
    constexpr int COLS_MAX = 1024;
    __shared__ double cols_s[COLS_MAX];
    const bool cols_lds = colsums && W <= COLS_MAX;
    if (cols_lds) {
        for (int c = threadIdx.x; c < W; c += 256) cols_s[c] = 0.0;
        __syncthreads();
    }
    const int n = blockIdx.y;
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    double part = 0.0;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per_sample; j += (long long)gridDim.x * 256) {
        const long long i = (long long)n * per_sample + j;
        const float m1 = mu1[i], m2 = mu2[i];
        const float m11 = m1 * m1, m22 = m2 * m2, m12 = m1 * m2;
        const float s11 = e11[i] - m11, s22 = e22[i] - m22, s12 = e12[i] - m12;
        const float A1 = 2.f * m12 + C1, A2 = 2.f * s12 + C2, B1 = m11 + m22 + C1, B2 = s11 + s22 + C2;
        const float inv = 1.f / (B1 * B2);
        const float map = A1 * A2 * inv;
        part += (double)map;
        if (cols_lds) atomicAdd(&cols_s[(int)(j % W)], (double)map);                          // ssim3D(size_average=False): means over (C, D, H) per (n, w)
        else if (colsums) atomicAdd(colsums + (long long)n * W + (int)(j % W), (double)map);
        // derivatives with mu and e as independent variables (s_ab = e_ab - mu_a mu_b)
        const float d_e12 = 2.f * A1 * inv;
        const float d_e = -map / B2;                                   // d / d e11 = d / d e22
        const float k1 = 2.f * (A2 - A1) * inv, k2 = 2.f * map * (B2 - B1) * inv;
        mu1[i] = m2 * k1 - m1 * k2;
        mu2[i] = m1 * k1 - m2 * k2;
        e11[i] = d_e; e22[i] = d_e; e12[i] = d_e12;
    }
    part = wave_sum_d(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s = red[0] + red[1] + red[2] + red[3];
        if (s != 0.0) atomicAdd(sums + n, s);
    }
    if (cols_lds)
        for (int c = threadIdx.x; c < W; c += 256)
            if (cols_s[c] != 0.0) atomicAdd(colsums + (long long)n * W + c, cols_s[c]);
}

__global__ __launch_bounds__(256) void ssim_cols_finalize_kernel(const double* colsums, int NW, double inv_count, float* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < NW) out[i] = (float)(colsums[i] * inv_count);
}

__global__ __launch_bounds__(64) void ssim_finalize_kernel(const double* sums, int N, long long per_sample, float* out) {
    if (threadIdx.x) return;
    double tot = 0.0;
    for (int n = 0; n < N; ++n) { tot += sums[n]; out[1 + n] = (float)(sums[n] / (double)per_sample); }
    out[0] = (float)(tot / ((double)per_sample * N));
}

// dx1 = scale_n * (b_mu1 + 2 x1 b_e11 + x2 b_e12), dx2 = scale_n * (b_mu2 + 2 x2 b_e22 + x1 b_e12); scale = gscale[per-sample or 0]
__global__ __launch_bounds__(256) void ssim_combine_kernel(const float* x1, const float* x2, const float* bmu1, const float* bmu2, const float* be11,
                                                           const float* be22, const float* be12, const float* gscale, int per_sample_scale,
                                                           long long per_sample, float* dx1, float* dx2) {
    const int n = blockIdx.y;
    const float sc0 = gscale[per_sample_scale == 1 ? n : 0];
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < per_sample; j += (long long)gridDim.x * 256) {
        const long long i = (long long)n * per_sample + j;
        const float sc = per_sample_scale >= 2 ? 1.f : sc0;      // >= 2 (row length W, gscale[n][w]): the maps were scaled per voxel before the blur
        const float a = x1[i], b = x2[i], g12 = be12[i];
        if (dx1) dx1[i] = sc * (bmu1[i] + 2.f * a * be11[i] + b * g12);
        if (dx2) dx2[i] = sc * (bmu2[i] + 2.f * b * be22[i] + a * g12);
    }
}

// position-dependent output weights (gscale[n][x]): d loss / d map varies inside a sample, so it multiplies the derivative maps BEFORE the adjoint blur
__global__ __launch_bounds__(256) void ssim_scale_cols_kernel(float* m0, float* m1, float* m2, float* m3, float* m4, const float* gscale, int W,
                                                              long long per_sample, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float sc = gscale[(i / per_sample) * W + (int)(i % W)];
        m0[i] *= sc; m1[i] *= sc; m2[i] *= sc; m3[i] *= sc; m4[i] *= sc;
    }
}

inline size_t a256s(size_t v) { return (v + 255) / 256 * 256; }
SsimWin make_window(int n) {
    // model/lossesSSIM.py:30-32 gaussian(window_size, 1.5): float32 exp values divided by their float32 sum
    SsimWin w; w.n = n;
    float tot = 0.f;
    for (int k = 0; k < n; ++k) { w.g[k] = (float)exp(-(double)((k - n / 2) * (k - n / 2)) / (2.0 * 1.5 * 1.5)); tot += w.g[k]; }
    for (int k = 0; k < n; ++k) w.g[k] /= tot;
    for (int k = n; k < SSIM_MAXW; ++k) w.g[k] = 0.f;
    return w;
}

}  // namespace

// ws layout: 5 statistic / derivative maps, 1 scratch map, N doubles of per-sample sums
long long ssim_ws_bytes(int planes, long long v) { return (long long)(6 * a256s((size_t)planes * v * 4) + a256s(64 * 8)); }

static void blur_all_axes(float* buf, float* tmp, int planes, int D, int H, int W, int nd, const SsimWin& w, hipStream_t s) {
    const long long total = (long long)planes * D * H * W;
    const dim3 grid((unsigned)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192));
    // nd passes ping-pong between buf and tmp; the result must end in buf
    float* a = buf; float* b = tmp;
    for (int ax = 3 - nd; ax < 3; ++ax) {
        hipLaunchKernelGGL(ssim_blur_kernel, grid, dim3(256), 0, s, (const float*)a, b, planes, D, H, W, ax, w);
        float* t = a; a = b; b = t;
    }
    if (a != buf) (void)hipMemcpyAsync(buf, a, (size_t)total * 4, hipMemcpyDeviceToDevice, s);
}

int launch_ssim_forward(const float* x1, const float* x2, int N, int C, int D, int H, int W, int nd, int window, void* ws_, float* out, hipStream_t s,
                        float* out_cols) {
    if (N > 64 || window < 1 || window > SSIM_MAXW || !(window & 1)) return -1;
    const int planes = N * C;
    const long long v = (long long)D * H * W, total = planes * v;
    const size_t m = a256s((size_t)total * 4);
    char* ws = (char*)ws_;
    float* mp[6];
    for (int i = 0; i < 6; ++i) mp[i] = (float*)(ws + i * m);
    double* sums = (double*)(ws + 6 * m);
    const SsimWin w = make_window(window);
    const dim3 grid((unsigned)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192));
    (void)hipMemcpyAsync(mp[0], x1, (size_t)total * 4, hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(mp[1], x2, (size_t)total * 4, hipMemcpyDeviceToDevice, s);
    hipLaunchKernelGGL(ssim_prod_kernel, grid, dim3(256), 0, s, x1, x2, mp[2], mp[3], mp[4], total);
    for (int i = 0; i < 5; ++i) blur_all_axes(mp[i], mp[5], planes, D, H, W, nd, w, s);
    (void)hipMemsetAsync(sums, 0, 64 * 8, s);
    const long long per_sample = (long long)C * v;
    const unsigned gx = (unsigned)(per_sample / 256 + 1 < 2048 ? per_sample / 256 + 1 : 2048);
    // out_cols ([N][W], optional): the means over (C, D, H) - what the reference's `.mean(1).mean(1).mean(1)` of a 5-D map leaves (model/lossesSSIM.py:
    // 92-97 with size_average=False).  The column sums live in the scratch map, free once the blur passes are done
    double* colsums = out_cols ? (double*)mp[5] : nullptr;
    if (colsums) {
        if ((size_t)N * W * 8 > m) return -1;
        (void)hipMemsetAsync(colsums, 0, (size_t)N * W * 8, s);
    }
    hipLaunchKernelGGL(ssim_map_kernel, dim3(gx, N), dim3(256), 0, s, mp[0], mp[1], mp[2], mp[3], mp[4], per_sample, sums, colsums, W);
    hipLaunchKernelGGL(ssim_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)sums, N, per_sample, out);
    if (colsums)
        hipLaunchKernelGGL(ssim_cols_finalize_kernel, dim3((N * W + 255) / 256), dim3(256), 0, s, (const double*)colsums, N * W, 1.0 / ((double)C * D * H), out_cols);
    return 0;
}

int launch_ssim_backward(const float* x1, const float* x2, int N, int C, int D, int H, int W, int nd, int window, void* ws_, const float* gscale,
                         int per_sample_scale, float* dx1, float* dx2, hipStream_t s) {
    if (N > 64 || window < 1 || window > SSIM_MAXW || !(window & 1)) return -1;
    const int planes = N * C;
    const long long v = (long long)D * H * W, total = planes * v;
    const size_t m = a256s((size_t)total * 4);
    char* ws = (char*)ws_;
    float* mp[6];
    for (int i = 0; i < 6; ++i) mp[i] = (float*)(ws + i * m);
    const SsimWin w = make_window(window);
    const long long per_sample = (long long)C * v;
    if (per_sample_scale >= 2) {
        if (per_sample_scale != W) return -1;
        hipLaunchKernelGGL(ssim_scale_cols_kernel, dim3((unsigned)(total / 256 + 1 < 8192 ? total / 256 + 1 : 8192)), dim3(256), 0, s, mp[0], mp[1], mp[2], mp[3],
                           mp[4], gscale, W, per_sample, total);
    }
    for (int i = 0; i < 5; ++i) blur_all_axes(mp[i], mp[5], planes, D, H, W, nd, w, s);      // the derivative maps of the forward pass, in place
    const unsigned gx = (unsigned)(per_sample / 256 + 1 < 2048 ? per_sample / 256 + 1 : 2048);
    hipLaunchKernelGGL(ssim_combine_kernel, dim3(gx, N), dim3(256), 0, s, x1, x2, (const float*)mp[0], (const float*)mp[1], (const float*)mp[2],
                       (const float*)mp[3], (const float*)mp[4], gscale, per_sample_scale, per_sample, dx1, dx2);
    return 0;
}

}  // namespace seg
