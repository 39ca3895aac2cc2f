// Soft-clDice building blocks (model/lossescldice.py:5-59 of the reference — which is non-functional as shipped,
// SURVEY.md §8a L8; these kernels implement the *intended* algorithm: rank test on x.dim(), __init__ spelled right).
//   soft_skeletonize: 10 x { e = minpool3(x); x = relu(x - relu(maxpool3(e) - e)) }   (3^3 for 5-D, 3^2 for 4-D inputs,
//   stride 1, pad 1 with the padding ignored like torch's max_pool).
// Tensors are planar fp32 [planes][D][H][W] (probabilities / one-hot targets as the reference API passes them).
// Backward routes gradients to the FIRST extremum of every window in (d,h,w) scan order (ATen max_pool backward).
#include "kernels.h"

namespace seg {
namespace {

struct Vol { int planes, D, H, W, nd; };   // nd = 3: pool over (D,H,W); nd = 2: pool over (H,W) only

__device__ __forceinline__ void vox(long long i, const Vol& v, int& p, int& z, int& y, int& x) {
    x = (int)(i % v.W); i /= v.W;
    y = (int)(i % v.H); i /= v.H;
    z = (int)(i % v.D);
    p = (int)(i / v.D);
}

// extremum of the window centred at (z,y,x) and the linear index of its first occurrence
template <bool IS_MIN>
__device__ __forceinline__ float window_ext(const float* base, const Vol& v, int z, int y, int x, long long& arg) {
    float best = 0.f;
    bool first = true;
    arg = 0;
    const int z0 = v.nd == 3 ? z - 1 : z, z1 = v.nd == 3 ? z + 1 : z;
    for (int zz = z0; zz <= z1; ++zz) {
        if ((unsigned)zz >= (unsigned)v.D) continue;
        for (int yy = y - 1; yy <= y + 1; ++yy) {
            if ((unsigned)yy >= (unsigned)v.H) continue;
            for (int xx = x - 1; xx <= x + 1; ++xx) {
                if ((unsigned)xx >= (unsigned)v.W) continue;
                const long long o = ((long long)zz * v.H + yy) * v.W + xx;
                const float val = base[o];
                if (first || (IS_MIN ? val < best : val > best)) { best = val; arg = o; first = false; }
            }
        }
    }
    return best;
}

template <bool IS_MIN>
__global__ __launch_bounds__(256) void pool3_kernel(const float* x, float* out, Vol v) {
    const long long V = (long long)v.D * v.H * v.W, total = V * v.planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int p, z, y, xx; vox(i, v, p, z, y, xx);
        long long arg;
        out[i] = window_ext<IS_MIN>(x + (long long)p * V, v, z, y, xx, arg);
    }
}

// out = relu(x - relu(maxpool3(e) - e))
__global__ __launch_bounds__(256) void skel_update_kernel(const float* x, const float* e, float* out, Vol v) {
    const long long V = (long long)v.D * v.H * v.W, total = V * v.planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int p, z, y, xx; vox(i, v, p, z, y, xx);
        long long arg;
        const float mx = window_ext<false>(e + (long long)p * V, v, z, y, xx, arg);
        const float c = fmaxf(mx - e[i], 0.f);
        out[i] = fmaxf(x[i] - c, 0.f);
    }
}

// backward of the update: dx = g*[x-c>0]; du = -dx*[mx-e>0]; de[i] -= du; de[argmax window(e)] += du   (de zeroed by caller)
__global__ __launch_bounds__(256) void skel_update_bwd_kernel(const float* g, const float* x, const float* e, float* dx, float* de, Vol v) {
    const long long V = (long long)v.D * v.H * v.W, total = V * v.planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int p, z, y, xx; vox(i, v, p, z, y, xx);
        long long arg;
        const float mx = window_ext<false>(e + (long long)p * V, v, z, y, xx, arg);
        const float u = mx - e[i];
        const float c = fmaxf(u, 0.f);
        const float gt = (x[i] - c > 0.f) ? g[i] : 0.f;
        dx[i] = gt;
        if (u > 0.f && gt != 0.f) {
            atomicAdd(&de[i], gt);                                   // x' = x - (mx - e): direct path d/de = +g
            atomicAdd(&de[(long long)p * V + arg], -gt);             // through maxpool3(e)
        }
    }
}

// din[first extremum of window i of src] += dout[i]
template <bool IS_MIN>
__global__ __launch_bounds__(256) void pool3_bwd_kernel(const float* src, const float* dout, float* din, Vol v) {
    const long long V = (long long)v.D * v.H * v.W, total = V * v.planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float d = dout[i];
        if (d == 0.f) continue;
        int p, z, y, xx; vox(i, v, p, z, y, xx);
        long long arg;
        window_ext<IS_MIN>(src + (long long)p * V, v, z, y, xx, arg);
        atomicAdd(&din[(long long)p * V + arg], d);
    }
}

// out[p] = {sum a*b, sum a} over the plane (fp64)
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* a, const float* b, double* out, long long V) {
    const int p = blockIdx.y;
    const long long v0 = (long long)blockIdx.x * 4096, v1 = (v0 + 4096 < V) ? v0 + 4096 : V;
    float s0 = 0.f, s1 = 0.f;
    for (long long i = v0 + threadIdx.x; i < v1; i += 256) {
        const float av = a[(long long)p * V + i];
        s0 += av * b[(long long)p * V + i];
        s1 += av;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[p * 2], (double)s0); atomicAdd(&out[p * 2 + 1], (double)s1); }
}

// out = a[p]*in + b[p]   (accumulate: out += ...)
__global__ __launch_bounds__(256) void plane_axpb_kernel(const float* in, const float* a, const float* b, float* out, long long V, int planes,
                                                         int accumulate) {
    const long long total = V * planes;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int p = (int)(i / V);
        const float r = fmaf(a[p], in[i], b[p]);
        out[i] = accumulate ? out[i] + r : r;
    }
}

inline int blocks_for(long long n) { long long b = (n + 255) / 256; return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

}  // namespace

void launch_pool3(const float* x, float* out, int planes, int D, int H, int W, int nd, int is_min, hipStream_t s) {
    Vol v{planes, D, H, W, nd};
    const long long n = (long long)planes * D * H * W;
    if (is_min) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool3_kernel<true>), dim3(blocks_for(n)), dim3(256), 0, s, x, out, v);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(pool3_kernel<false>), dim3(blocks_for(n)), dim3(256), 0, s, x, out, v);
}
void launch_skel_update(const float* x, const float* e, float* out, int planes, int D, int H, int W, int nd, hipStream_t s) {
    Vol v{planes, D, H, W, nd};
    hipLaunchKernelGGL(skel_update_kernel, dim3(blocks_for((long long)planes * D * H * W)), dim3(256), 0, s, x, e, out, v);
}
void launch_skel_update_bwd(const float* g, const float* x, const float* e, float* dx, float* de, int planes, int D, int H, int W, int nd,
                            hipStream_t s) {
    Vol v{planes, D, H, W, nd};
    hipLaunchKernelGGL(skel_update_bwd_kernel, dim3(blocks_for((long long)planes * D * H * W)), dim3(256), 0, s, g, x, e, dx, de, v);
}
void launch_pool3_bwd(const float* src, const float* dout, float* din, int planes, int D, int H, int W, int nd, int is_min, hipStream_t s) {
    Vol v{planes, D, H, W, nd};
    const long long n = (long long)planes * D * H * W;
    if (is_min) hipLaunchKernelGGL(HIP_KERNEL_NAME(pool3_bwd_kernel<true>), dim3(blocks_for(n)), dim3(256), 0, s, src, dout, din, v);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(pool3_bwd_kernel<false>), dim3(blocks_for(n)), dim3(256), 0, s, src, dout, din, v);
}
void launch_plane_dot(const float* a, const float* b, double* out, int planes, long long V, hipStream_t s) {
    (void)hipMemsetAsync(out, 0, sizeof(double) * 2 * planes, s);
    hipLaunchKernelGGL(plane_dot_kernel, dim3(cdiv(V, 4096), planes), dim3(256), 0, s, a, b, out, V);
}
void launch_plane_axpb(const float* in, const float* a, const float* b, float* out, int planes, long long V, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(plane_axpb_kernel, dim3(blocks_for(V * planes)), dim3(256), 0, s, in, a, b, out, V, planes, accumulate);
}

}  // namespace seg
