// Pre/post-processing either side of `predict`, on the device so a volume crosses PCIe
// once in and one byte per voxel out:
//   * resample3d      - SimpleITK ResampleImageFilter with the identity transform as the reference drives it
//                       (dataprocess/utils.py:99-145 resize_image_itkwithsize / resize_image_itk): linear / nearest
//   * normalize_meanstd    - ConvertitkTrunctedValue(..., 'meanstd') (dataprocess/utils.py:148-179): clip + itk NormalizeImageFilter
//   * normalize_percentile - normalize() (dataprocess/utils.py:182-204): np.percentile(5/95) clip + z-score over the non-zero voxels;
//                            the two percentiles are exact order statistics found by a 3-pass radix select (no sort)
//   * gather_patches / stitch_mask - the crop and the `out_mask[...] += patch; out_mask[out_mask != 0] = 1` of
//                            inference_patch (model/modelUnet.py:707-763), batched
// All kernels are HBM-bound streaming passes over planar [D][H][W] volumes (one channel, the reference's 3-D models).
#include "kernels.h"

namespace seg {
namespace {

inline int pp_blocks(long long total, int cap = 4096) {
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------------- resample
// Output voxel (z,y,x) sits at continuous input index c = idx * step per axis (same origin and direction, output spacing =
// input spacing * step).  itk::ImageFunction::IsInsideBuffer: -0.5 <= c < size - 0.5, otherwise the default pixel value 0.
// Linear = itk::LinearInterpolateImageFunction (double arithmetic, neighbours beyond the last index clamp to it);
// nearest = itk::NearestNeighborInterpolateImageFunction (Math::RoundHalfIntegerUp).
template <class T>
__device__ __forceinline__ T resample_voxel(const ResampleArgs& a, const T* src, int z, int y, int x) {
    const double cz = z * a.fz, cy = y * a.fy, cx = x * a.fx;
    const bool inside = cz >= -0.5 && cz < a.sD - 0.5 && cy >= -0.5 && cy < a.sH - 0.5 && cx >= -0.5 && cx < a.sW - 0.5;
    if (!inside) return (T)0;
    if (a.mode == RS_NEAREST) {
        const int iz = (int)floor(cz + 0.5), iy = (int)floor(cy + 0.5), ix = (int)floor(cx + 0.5);
        return src[((long long)iz * a.sH + iy) * a.sW + ix];
    }
    int z0 = (int)floor(cz), y0 = (int)floor(cy), x0 = (int)floor(cx);
    z0 = z0 < 0 ? 0 : z0; y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
    double dz = cz - z0, dy = cy - y0, dx = cx - x0;
    dz = dz < 0.0 ? 0.0 : dz; dy = dy < 0.0 ? 0.0 : dy; dx = dx < 0.0 ? 0.0 : dx;
    const int z1 = z0 + 1 < a.sD ? z0 + 1 : a.sD - 1, y1 = y0 + 1 < a.sH ? y0 + 1 : a.sH - 1, x1 = x0 + 1 < a.sW ? x0 + 1 : a.sW - 1;
    const T* p00 = src + ((long long)z0 * a.sH + y0) * a.sW;
    const T* p01 = src + ((long long)z0 * a.sH + y1) * a.sW;
    const T* p10 = src + ((long long)z1 * a.sH + y0) * a.sW;
    const T* p11 = src + ((long long)z1 * a.sH + y1) * a.sW;
    const double v000 = (double)p00[x0], v001 = (double)p00[x1], v010 = (double)p01[x0], v011 = (double)p01[x1];
    const double v100 = (double)p10[x0], v101 = (double)p10[x1], v110 = (double)p11[x0], v111 = (double)p11[x1];
    const double a00 = v000 + dx * (v001 - v000), a01 = v010 + dx * (v011 - v010);
    const double a10 = v100 + dx * (v101 - v100), a11 = v110 + dx * (v111 - v110);
    const double b0 = a00 + dy * (a01 - a00), b1 = a10 + dy * (a11 - a10);
    return (T)(b0 + dz * (b1 - b0));
}
// a thread produces VEC consecutive voxels of an output row and stores them as one 16-byte (f32 x4, u8 x16) or 4-byte vector:
// the nearest-neighbour up-sampling of a uint8 mask is store-bound, one byte per lane wastes 15/16 of every write
template <class T, int VEC>
__global__ __launch_bounds__(256) void resample3d_kernel(ResampleArgs a) {
    const T* src = (const T*)a.src;
    T* dst = (T*)a.dst;
    const int wv = a.dW / VEC;                       // the launcher picks VEC so that dW % VEC == 0
    const long long total = (long long)a.dD * a.dH * wv;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xv = (int)(i % wv), y = (int)((i / wv) % a.dH), z = (int)(i / ((long long)wv * a.dH));
        vec<T, VEC> out;
#pragma unroll
        for (int j = 0; j < VEC; ++j) out[j] = resample_voxel<T>(a, src, z, y, xv * VEC + j);
        *(vec<T, VEC>*)(dst + ((long long)z * a.dH + y) * a.dW + (long long)xv * VEC) = out;
    }
}
template <class T>
__global__ __launch_bounds__(256) void resample3d_scalar_kernel(ResampleArgs a) {
    const T* src = (const T*)a.src;
    T* dst = (T*)a.dst;
    const long long total = (long long)a.dD * a.dH * a.dW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256)
        dst[i] = resample_voxel<T>(a, src, (int)(i / ((long long)a.dW * a.dH)), (int)((i / a.dW) % a.dH), (int)(i % a.dW));
}

// ------------------------------------------------------------------------------------------- clip + mean/std
constexpr int NS_SUM = 0, NS_SQ = 1, NS_CNT = 2, NS_STRIDE = 4;      // per replica: sum, sum of squares, count (fp64)

__device__ __forceinline__ void ns_flush(double* sums, double s, double q, double c) {
    s = wave_sum_d(s); q = wave_sum_d(q); c = wave_sum_d(c);
    if ((threadIdx.x & 63) == 0) {
        double* d = sums + (long long)((blockIdx.x * 4 + (threadIdx.x >> 6)) % STAT_REP) * NS_STRIDE;
        atomicAdd(d + NS_SUM, s); atomicAdd(d + NS_SQ, q); atomicAdd(d + NS_CNT, c);
    }
}
__device__ __forceinline__ void ns_fold(const double* sums, double& s, double& q, double& c) {
    s = q = c = 0.0;
    for (int r = 0; r < STAT_REP; ++r) { s += sums[r * NS_STRIDE + NS_SUM]; q += sums[r * NS_STRIDE + NS_SQ]; c += sums[r * NS_STRIDE + NS_CNT]; }
}

__global__ __launch_bounds__(256) void meanstd_stats_kernel(const float* x, long long n, int clip, float lo, float hi, double* sums) {
    double s = 0.0, q = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i];
        if (clip) v = v > hi ? hi : (v < lo ? lo : v);
        s += (double)v; q += (double)v * (double)v;
    }
    ns_flush(sums, s, q, 0.0);
}
// itk::NormalizeImageFilter = StatisticsImageFilter (mean, sigma with the N-1 denominator, double) + ShiftScale(-mean, 1/sigma)
__global__ __launch_bounds__(256) void meanstd_apply_kernel(const float* x, float* out, long long n, int clip, float lo, float hi, const double* sums) {
    double s, q, c;
    ns_fold(sums, s, q, c);
    const double mean = s / (double)n;
    double var = n > 1 ? (q - s * s / (double)n) / (double)(n - 1) : 0.0;
    var = var < 0.0 ? 0.0 : var;
    const double sigma = sqrt(var);
    const double scale = sigma > 0.0 ? 1.0 / sigma : 1.0;      // a constant volume maps to zeros instead of ITK's division by zero
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i];
        if (clip) v = v > hi ? hi : (v < lo ? lo : v);
        out[i] = (float)(((double)v - mean) * scale);
    }
}

// ------------------------------------------------------------------------------------------- percentile clip + z-score
// Workspace layout (bytes): [0, 32 KiB) four 2048-bin histograms (u32); then SelState.
constexpr int SEL_BINS = 2048;
struct SelState {
    unsigned prefix[4];             // key bits resolved so far for the four order statistics (lo/hi index of both percentiles)
    unsigned long long rank[4];     // rank still to resolve inside the current prefix bucket
    unsigned key_min_nz, key_max_nz;   // min / max key over the non-zero clipped values (std == 0 test, exact)
    unsigned pad[2];
    double sums[STAT_REP * NS_STRIDE];  // non-zero clipped values: sum, sum of squares, count
};
constexpr size_t SEL_WS_BYTES = 4 * SEL_BINS * sizeof(unsigned) + sizeof(SelState);

__device__ __forceinline__ unsigned f2key(float f) {            // monotone: a < b  <=>  key(a) < key(b)
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ int sel_shift(int pass) { return pass == 0 ? 21 : (pass == 1 ? 10 : 0); }
__device__ __forceinline__ int sel_bits(int pass) { return pass == 2 ? 10 : 11; }

__global__ __launch_bounds__(64) void sel_init_kernel(unsigned* hist, SelState* st, unsigned long long k0, unsigned long long k1,
                                                      unsigned long long k2, unsigned long long k3) {
    for (int i = threadIdx.x; i < 4 * SEL_BINS; i += 64) hist[i] = 0u;
    for (int i = threadIdx.x; i < STAT_REP * NS_STRIDE; i += 64) st->sums[i] = 0.0;
    if (threadIdx.x == 0) {
        st->rank[0] = k0; st->rank[1] = k1; st->rank[2] = k2; st->rank[3] = k3;
        for (int r = 0; r < 4; ++r) st->prefix[r] = 0u;
        st->key_min_nz = 0xFFFFFFFFu; st->key_max_nz = 0u;
    }
}

// one radix digit: histogram of the elements that match each statistic's resolved prefix (pass 0: a single shared histogram)
__global__ __launch_bounds__(256) void sel_hist_kernel(const float* x, long long n, unsigned* hist, const SelState* st, int pass) {
    __shared__ unsigned lh[4 * SEL_BINS];
    const int nr = pass == 0 ? 1 : 4;
    for (int i = threadIdx.x; i < nr * SEL_BINS; i += 256) lh[i] = 0u;
    __syncthreads();
    const int shift = sel_shift(pass), bits = sel_bits(pass);
    const unsigned dmask = (1u << bits) - 1u;
    const int hshift = shift + bits;                     // bits above the current digit (32 on pass 0: nothing to match)
    unsigned pre[4];
    for (int r = 0; r < 4; ++r) pre[r] = pass == 0 ? 0u : (st->prefix[r] >> hshift);
    // Plain LDS atomics.  A per-digit wave vote (one atomic per distinct digit and wave) was measured and is slower on
    // noise-like volumes: ~25 distinct leading digits per wave cost more vote rounds than the serialised atomics they replace
    // (256^3: 0.63 -> 0.88 ms, profiles/r01_prepost_step22.jsonl vs r01_prepost_step23.jsonl).
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned k = f2key(x[i]);
        const unsigned d = (k >> shift) & dmask;
        if (pass == 0) {
            atomicAdd(&lh[d], 1u);
        } else {
            const unsigned hi = k >> hshift;
            for (int r = 0; r < 4; ++r)
                if (hi == pre[r] && (r == 0 || pre[r] != pre[r - 1])) atomicAdd(&lh[r * SEL_BINS + d], 1u);   // equal prefixes share a histogram
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * SEL_BINS; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// single workgroup: per statistic, find the digit whose cumulative count passes the rank; extend the prefix; clear the histograms
__global__ __launch_bounds__(256) void sel_scan_kernel(unsigned* hist, SelState* st, int pass) {
    __shared__ unsigned long long part[256];
    __shared__ int src_of[4];
    const int tid = threadIdx.x, shift = sel_shift(pass), bits = sel_bits(pass), hshift = shift + bits;
    const int per = SEL_BINS / 256;
    if (tid == 0) {
        for (int r = 0; r < 4; ++r) {           // statistics with equal prefixes were counted once, in the first one's histogram
            int s = r;
            if (pass == 0) s = 0;
            else while (s > 0 && (st->prefix[s - 1] >> hshift) == (st->prefix[r] >> hshift)) --s;
            src_of[r] = s;
        }
    }
    __syncthreads();
    for (int r = 0; r < 4; ++r) {
        const unsigned* h = hist + src_of[r] * SEL_BINS;
        unsigned long long loc = 0;
        for (int j = 0; j < per; ++j) loc += h[tid * per + j];
        part[tid] = loc;
        __syncthreads();
        if (tid == 0) {
            unsigned long long cum = 0;
            for (int t = 0; t < 256; ++t) { const unsigned long long v = part[t]; part[t] = cum; cum += v; }
        }
        __syncthreads();
        const unsigned long long rank = st->rank[r], before = part[tid];
        __syncthreads();
        if (rank >= before && rank < before + loc) {
            unsigned long long cum = before;
            for (int j = 0; j < per; ++j) {
                const unsigned c = h[tid * per + j];
                if (rank < cum + c) {
                    st->prefix[r] |= (unsigned)(tid * per + j) << shift;
                    st->rank[r] = rank - cum;
                    break;
                }
                cum += c;
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < 4 * SEL_BINS; i += 256) hist[i] = 0u;
}

// np.percentile's float32 linear interpolation (numpy lib/_function_base_impl.py _lerp): separate multiply and add, all fp32
__device__ __forceinline__ float np_lerp_f32(float a, float b, float t) {
#pragma clang fp contract(off)
    const float d = b - a;
    if (t >= 0.5f) { const float m = d * (1.0f - t); return b - m; }
    const float m = d * t;
    return a + m;
}
__device__ __forceinline__ void pn_bounds(const SelState* st, float g_lo, float g_hi, float& t, float& b) {
    t = np_lerp_f32(key2f(st->prefix[0]), key2f(st->prefix[1]), g_lo);
    b = np_lerp_f32(key2f(st->prefix[2]), key2f(st->prefix[3]), g_hi);
}

__global__ __launch_bounds__(256) void pnorm_stats_kernel(const float* x, long long n, SelState* st, float g_lo, float g_hi) {
    float t, b;
    pn_bounds(st, g_lo, g_hi, t, b);
    double s = 0.0, q = 0.0, c = 0.0;
    unsigned kmin = 0xFFFFFFFFu, kmax = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i];
        v = v < t ? t : (v > b ? b : v);            // np.clip(slice, t, b) = minimum(maximum(x, t), b)
        if (v != 0.f) {
            s += (double)v; q += (double)v * (double)v; c += 1.0;
            const unsigned k = f2key(v);
            kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
        }
    }
    ns_flush(st->sums, s, q, c);
    for (int m = 32; m >= 1; m >>= 1) {
        const unsigned a0 = __shfl_xor(kmin, m), a1 = __shfl_xor(kmax, m);
        kmin = a0 < kmin ? a0 : kmin; kmax = a1 > kmax ? a1 : kmax;
    }
    if ((threadIdx.x & 63) == 0 && kmin <= kmax) { atomicMin(&st->key_min_nz, kmin); atomicMax(&st->key_max_nz, kmax); }
}
__global__ __launch_bounds__(256) void pnorm_apply_kernel(const float* x, float* out, long long n, const SelState* st, float g_lo, float g_hi) {
    float t, b;
    pn_bounds(st, g_lo, g_hi, t, b);
    double s, q, c;
    ns_fold(st->sums, s, q, c);
    // `if np.std(slice) == 0 or np.std(image_nonzero) == 0: return slice` (utils.py:196-197): the clipped volume is constant
    // iff t == b; the non-zero subset is constant iff its min == max (it is never empty when t != b)
    const bool flat = !(t < b) || st->key_min_nz >= st->key_max_nz || c < 1.0;
    const double mean_d = c > 0.0 ? s / c : 0.0;
    double var = c > 0.0 ? q / c - mean_d * mean_d : 0.0;
    var = var < 0.0 ? 0.0 : var;
    const float mean = (float)mean_d, sd = (float)sqrt(var);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = x[i];
        v = v < t ? t : (v > b ? b : v);
        out[i] = flat ? v : (v - mean) / sd;
    }
}

// ------------------------------------------------------------------------------------------- patches
__global__ __launch_bounds__(256) void gather_patches_kernel(const float* vol, int D, int H, int W, const int* origins, int nb, int pd, int ph,
                                                             int pw, float* out) {
    const long long pv = (long long)pd * ph * pw, total = pv * nb;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / pv);
        const long long r = i % pv;
        const int x = (int)(r % pw), y = (int)((r / pw) % ph), z = (int)(r / ((long long)pw * ph));
        const int* o = origins + 3 * b;
        out[i] = vol[((long long)(o[0] + z) * H + (o[1] + y)) * W + (o[2] + x)];
    }
}
// out[window] |= (mask != 0): every writer stores the same value 1, so overlapping windows need no atomics
__global__ __launch_bounds__(256) void stitch_mask_kernel(const unsigned char* masks, const int* origins, int nb, int pd, int ph, int pw,
                                                          unsigned char* out, int D, int H, int W) {
    const long long pv = (long long)pd * ph * pw, total = pv * nb;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (!masks[i]) continue;
        const int b = (int)(i / pv);
        const long long r = i % pv;
        const int x = (int)(r % pw), y = (int)((r / pw) % ph), z = (int)(r / ((long long)pw * ph));
        const int* o = origins + 3 * b;
        out[((long long)(o[0] + z) * H + (o[1] + y)) * W + (o[2] + x)] = 1;
    }
}

}  // namespace

void launch_resample3d(const ResampleArgs& a, int elem_type, hipStream_t s) {
    const long long n = (long long)a.dD * a.dH * a.dW;
    const bool al16 = ((uintptr_t)a.dst & 15) == 0;
    if (elem_type == 0) {
        if (al16 && a.dW % 4 == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(resample3d_kernel<float, 4>), dim3(pp_blocks(n / 4, 16384)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(resample3d_scalar_kernel<float>), dim3(pp_blocks(n, 16384)), dim3(256), 0, s, a);
    } else {
        if (al16 && a.dW % 16 == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(resample3d_kernel<unsigned char, 16>), dim3(pp_blocks(n / 16, 16384)), dim3(256), 0, s, a);
        else if (al16 && a.dW % 4 == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(resample3d_kernel<unsigned char, 4>), dim3(pp_blocks(n / 4, 16384)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(resample3d_scalar_kernel<unsigned char>), dim3(pp_blocks(n, 16384)), dim3(256), 0, s, a);
    }
}

size_t normalize_ws_bytes() { return SEL_WS_BYTES; }

void launch_normalize_meanstd(const float* x, float* out, long long n, int clip, float lo, float hi, void* ws, hipStream_t s) {
    double* sums = (double*)ws;
    (void)hipMemsetAsync(sums, 0, sizeof(double) * STAT_REP * NS_STRIDE, s);
    hipLaunchKernelGGL(meanstd_stats_kernel, dim3(pp_blocks(n)), dim3(256), 0, s, x, n, clip, lo, hi, sums);
    hipLaunchKernelGGL(meanstd_apply_kernel, dim3(pp_blocks(n, 16384)), dim3(256), 0, s, x, out, n, clip, lo, hi, (const double*)sums);
}

void launch_normalize_percentile(const float* x, float* out, long long n, float q_lo, float q_hi, void* ws, hipStream_t s) {
    unsigned* hist = (unsigned*)ws;
    SelState* st = (SelState*)((char*)ws + 4 * SEL_BINS * sizeof(unsigned));
    // np.percentile on a float32 array keeps float32 throughout: q/100, the virtual index (n-1)*q and its fraction
    unsigned long long k[4];
    float g[2];
    const float qs[2] = {q_lo, q_hi};
    for (int j = 0; j < 2; ++j) {
        volatile float q32 = qs[j] / 100.0f;
        volatile float vi = (float)(n - 1) * q32;
        long long lo = (long long)floorf(vi);
        if (lo < 0) lo = 0;
        if (lo > n - 1) lo = n - 1;
        const long long hi = lo + 1 < n ? lo + 1 : n - 1;
        g[j] = vi - (float)lo;
        k[2 * j] = (unsigned long long)lo; k[2 * j + 1] = (unsigned long long)hi;
    }
    hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(64), 0, s, hist, st, k[0], k[1], k[2], k[3]);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(sel_hist_kernel, dim3(pp_blocks(n, 1024)), dim3(256), 0, s, x, n, hist, (const SelState*)st, pass);
        hipLaunchKernelGGL(sel_scan_kernel, dim3(1), dim3(256), 0, s, hist, st, pass);
    }
    hipLaunchKernelGGL(pnorm_stats_kernel, dim3(pp_blocks(n)), dim3(256), 0, s, x, n, st, g[0], g[1]);
    hipLaunchKernelGGL(pnorm_apply_kernel, dim3(pp_blocks(n, 16384)), dim3(256), 0, s, x, out, n, (const SelState*)st, g[0], g[1]);
}

void launch_gather_patches(const float* vol, int D, int H, int W, const int* origins, int nb, int pd, int ph, int pw, float* out, hipStream_t s) {
    hipLaunchKernelGGL(gather_patches_kernel, dim3(pp_blocks((long long)nb * pd * ph * pw, 16384)), dim3(256), 0, s, vol, D, H, W, origins, nb,
                       pd, ph, pw, out);
}
void launch_stitch_mask(const unsigned char* masks, const int* origins, int nb, int pd, int ph, int pw, unsigned char* out, int D, int H, int W,
                        hipStream_t s) {
    hipLaunchKernelGGL(stitch_mask_kernel, dim3(pp_blocks((long long)nb * pd * ph * pw, 16384)), dim3(256), 0, s, masks, origins, nb, pd, ph, pw,
                       out, D, H, W);
}

}  // namespace seg
