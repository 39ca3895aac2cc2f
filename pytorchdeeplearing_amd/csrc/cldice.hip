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

// out[p] = {sum a*b, sum a} over the plane (fp64): per-workgroup partials (plain stores), then one small fold - atomics of
// a thousand workgroups on two addresses per plane serialised (105 us for a 16 MB plane)
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* a, const float* b, double* partial, long long V) {
    __shared__ double wsum[4][2];
    const int p = blockIdx.y;
    const long long v0 = (long long)blockIdx.x * 4096, v1 = (v0 + 4096 < V) ? v0 + 4096 : V;
    float s0 = 0.f, s1 = 0.f;
    for (long long i = v0 + threadIdx.x; i < v1; i += 256) {
        const float av = a[(long long)p * V + i];
        s0 += av * b[(long long)p * V + i];
        s1 += av;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6][0] = s0; wsum[threadIdx.x >> 6][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2)
        partial[((long long)p * gridDim.x + blockIdx.x) * 2 + threadIdx.x] =
            (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void plane_dot_fold_kernel(const double* partial, double* out, int nblk) {
    const int p = blockIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) { s0 += partial[((long long)p * nblk + i) * 2]; s1 += partial[((long long)p * nblk + i) * 2 + 1]; }
    __shared__ double red[4][2];
    s0 = wave_sum_d(s0); s1 = wave_sum_d(s1);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s0; red[threadIdx.x >> 6][1] = s1; }
    __syncthreads();
    if (threadIdx.x < 2) out[p * 2 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
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

// One skeleton iteration in ONE pass over HBM: e = minpool3(x), out = relu(x - relu(maxpool3(e) - e)); both are written (the
// backward pass needs e).  A workgroup owns a TZ x TY x TX output tile: x with a 2-voxel halo sits in LDS (+inf outside the
// volume), e is formed on the 1-voxel halo (-inf outside), then the update reads only LDS.  Pure min / max / sub / relu in
// fp32: bit-identical to the two-kernel form (pool3_kernel + skel_update_kernel), which read every input 27 times from L2.
// (Round 3 tried SEPARABLE pools - min over x, then y, then z; likewise max: 38 k instead of 83 k LDS reads per tile, same bits - and
// lost: six more barriers and 46 KB instead of 22 KB of LDS per workgroup; C5 + clDice 7.67 vs 7.38 ms per step.  Not kept.)
// FOUR x-consecutive voxels per thread (round 5).  The first form (one voxel per thread, rounds 1-4) issued one ds_read_b32 per window element: 81 LDS
// instructions per output voxel, and the rocprofv3 stats of the C5 + clDice step (profiles/r05_kernel_stats_c5_cldice.txt) put its 20 launches at
// 44 us each - 0.88 ms of a 6.3 ms step - for 48 MB of traffic per launch (1.1 TB/s: not the memory system, the LDS instruction stream).  Here a
// thread reads a window ROW of its four voxels as one ds_read_b128 + one ds_read_b64 (six values; rows are 36 floats = 9 x 16 B, groups start at
// multiples of four, so both reads are aligned), forms the four row minima / maxima with v_min3 / v_max3 and folds the nine rows: 18 LDS
// instructions per FOUR voxels in either pass, results stored as one 16-B vector.  Same min / max / sub / relu in fp32, same operand values: the
// outputs are bit-identical to the two-kernel form (tests/test_cldice.py).
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
template <int TZ, int TY, int TX, bool ND3>
__global__ __launch_bounds__(256) void skel_iter4_kernel(const float* x, float* e_out, float* x_out, Vol v) {
    static_assert(TX % 4 == 0, "four voxels per thread along x");
    constexpr int HZ = ND3 ? 2 : 0, H1 = ND3 ? 1 : 0;
    constexpr int XZ = TZ + 2 * HZ, XY = TY + 4, XX = TX + 4;               // x with a 2-voxel halo; XX is a multiple of 4 (16-B rows)
    constexpr int EZ = TZ + 2 * H1, EY = TY + 2, EX = TX + 2, EXP = TX + 4; // e on the 1-voxel halo; row pitch padded to a multiple of 4
    constexpr int NKZ = ND3 ? 3 : 1;
    __shared__ __attribute__((aligned(16))) float xs[XZ * XY * XX + 4];     // (+4: the last group of the last row reads two floats past it)
    __shared__ __attribute__((aligned(16))) float es[EZ * EY * EXP + 4];
    const float INF = __builtin_huge_valf();
    const int ntx = (v.W + TX - 1) / TX, nty = (v.H + TY - 1) / TY, ntz = (v.D + TZ - 1) / TZ;
    int b = blockIdx.x;
    const int x0 = (b % ntx) * TX; b /= ntx;
    const int y0 = (b % nty) * TY; b /= nty;
    const int z0 = (b % ntz) * TZ;
    const int p = b / ntz;
    const long long V = (long long)v.D * v.H * v.W;
    const float* xp = x + (long long)p * V;
    {
        constexpr int NE = XZ * XY * XX, NIT = (NE + 255) / 256;
        float hv[NIT];
        bool hin[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int lx = i % XX, ly = (i / XX) % XY, lz = i / (XX * XY);
            const int gx = x0 + lx - 2, gy = y0 + ly - 2, gz = z0 + lz - HZ;
            hin[k] = i < NE && (unsigned)gx < (unsigned)v.W && (unsigned)gy < (unsigned)v.H && (unsigned)gz < (unsigned)v.D;
            hv[k] = xp[hin[k] ? ((long long)gz * v.H + gy) * v.W + gx : 0];
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < NE) xs[i] = hin[k] ? hv[k] : INF;
        }
        if (threadIdx.x < 4) xs[NE + threadIdx.x] = INF;
    }
    __syncthreads();
    // ---- e = minpool3(x) on the 1-voxel halo: groups of four e values (e index a .. a+3 <-> x indices a .. a+5)
    constexpr int EG = (EX + 3) / 4, NEG = EZ * EY * EG;
    for (int i = threadIdx.x; i < NEG; i += 256) {
        const int a = (i % EG) * 4, ly = (i / EG) % EY, lz = i / (EG * EY);
        float m[4] = {INF, INF, INF, INF};
#pragma unroll
        for (int dz = 0; dz < NKZ; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* row = &xs[((lz + dz) * XY + ly + dy) * XX + a];
                const vec<float, 4> q4 = *(const vec<float, 4>*)row;
                const vec<float, 2> q2 = *(const vec<float, 2>*)(row + 4);
                m[0] = fminf(m[0], min3f(q4[0], q4[1], q4[2]));
                m[1] = fminf(m[1], min3f(q4[1], q4[2], q4[3]));
                m[2] = fminf(m[2], min3f(q4[2], q4[3], q2[0]));
                m[3] = fminf(m[3], min3f(q4[3], q2[0], q2[1]));
            }
        const int gy = y0 + ly - 1, gz = z0 + lz - H1;
        const bool rin = (unsigned)gy < (unsigned)v.H && (unsigned)gz < (unsigned)v.D;
        vec<float, 4> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (rin && (unsigned)(x0 + a + j - 1) < (unsigned)v.W) ? m[j] : -INF;   // outside the volume: ignored by the max-pool
        *(vec<float, 4>*)&es[(lz * EY + ly) * EXP + a] = o;
    }
    if (threadIdx.x < 4) es[EZ * EY * EXP + threadIdx.x] = -INF;
    __syncthreads();
    // ---- update: four outputs at x index o4 .. o4+3 <-> e indices o4 .. o4+5
    constexpr int OG = TX / 4, NOG = TZ * TY * OG;
    for (int i = threadIdx.x; i < NOG; i += 256) {
        const int o4 = (i % OG) * 4, ly = (i / OG) % TY, lz = i / (OG * TY);
        const int gx = x0 + o4, gy = y0 + ly, gz = z0 + lz;
        if (gx >= v.W || gy >= v.H || gz >= v.D) continue;
        float mx[4] = {-INF, -INF, -INF, -INF}, ev[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dz = 0; dz < NKZ; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const float* row = &es[((lz + dz) * EY + ly + dy) * EXP + o4];
                const vec<float, 4> q4 = *(const vec<float, 4>*)row;
                const vec<float, 2> q2 = *(const vec<float, 2>*)(row + 4);
                mx[0] = fmaxf(mx[0], max3f(q4[0], q4[1], q4[2]));
                mx[1] = fmaxf(mx[1], max3f(q4[1], q4[2], q4[3]));
                mx[2] = fmaxf(mx[2], max3f(q4[2], q4[3], q2[0]));
                mx[3] = fmaxf(mx[3], max3f(q4[3], q2[0], q2[1]));
                if (dz == H1 && dy == 1) { ev[0] = q4[1]; ev[1] = q4[2]; ev[2] = q4[3]; ev[3] = q2[0]; }      // the centres of the four windows
            }
        const float* xr = &xs[((lz + HZ) * XY + ly + 2) * XX + o4 + 2];
        const vec<float, 2> xa = *(const vec<float, 2>*)xr, xb = *(const vec<float, 2>*)(xr + 2);
        const float xv[4] = {xa[0], xa[1], xb[0], xb[1]};
        vec<float, 4> eo, xo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { eo[j] = ev[j]; xo[j] = fmaxf(xv[j] - fmaxf(mx[j] - ev[j], 0.f), 0.f); }
        const long long o = (long long)p * V + ((long long)gz * v.H + gy) * v.W + gx;
        if (gx + 3 < v.W && (v.W & 3) == 0) {
            *(vec<float, 4>*)&e_out[o] = eo;
            *(vec<float, 4>*)&x_out[o] = xo;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (gx + j < v.W) { e_out[o + j] = eo[j]; x_out[o + j] = xo[j]; }
        }
    }
}

// ---- the skeleton of a BINARY image in bits (round 5).  On a {0, 1} image every step of the iteration stays in {0, 1}: e = minpool3(x) is an erosion,
// maxpool3(e) a dilation, c = relu(maxpool3(e) - e) = dil(e) & ~e and x' = relu(x - c) = x & ~c.  The target of the binary clDice term is such an
// image (the train loop binarises the labels before any loss sees them, model/modelVNet.py:576), so its skeleton is computed on a bit image - 32
// voxels per word along x, 512 KB for a 160^3 volume instead of 16 MB, one word per thread, neighbours from L2 - and expanded to fp32 once at the
// end: EXACTLY the values the fp32 kernels produce (tests/test_cldice.py: bit-for-bit against skel_iter on the float labels).  Padding follows the
// pools: outside the volume counts as 1 for the erosion (min-pool ignores it) and as 0 for the dilation.  The bits past W in a row's last word are
// kept 0.  Two launches per iteration (the dilation needs the eroded neighbours), ~4 us each instead of one 44 us tile pass.
struct BitVol { int rows_z, D, H, W, WX, nd3; };     // rows_z = planes * D (2-D pooling: every depth slice is its own image, D = 1)
__device__ __forceinline__ unsigned bit_tail(const BitVol& b, int wx) {     // bits of word wx that lie at x >= W
    const int n = b.W - wx * 32;
    return n >= 32 ? 0u : ~((1u << n) - 1u);
}
// labels -> {float y = (label != 0), bits}: a wave takes 64 x-consecutive voxels of one row per trip
__global__ __launch_bounds__(256) void cld_labels_bits_kernel(const void* target, int lt, float* y, unsigned* bits, BitVol b, long long nrows) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6, nwave = ((long long)gridDim.x * 256) >> 6;
    for (long long r = wave; r < nrows; r += nwave) {
        for (int xb = 0; xb < b.WX * 32; xb += 64) {
            const int xx = xb + lane;
            const bool in = xx < b.W;
            const int lab = in ? (load_label(target, lt, r * b.W + xx) != 0) : 0;
            if (in) y[r * b.W + xx] = (float)lab;
            const unsigned long long m = __ballot(lab);
            if (lane == 0) {
                bits[r * b.WX + (xb >> 5)] = (unsigned)m;
                if ((xb >> 5) + 1 < b.WX) bits[r * b.WX + (xb >> 5) + 1] = (unsigned)(m >> 32);
            }
        }
    }
}
// MODE 0: out = erode(in).  MODE 1: out = x & ~(dilate(in) & ~in) with in = the eroded image, x = the image it was eroded from.
template <int MODE>
__global__ __launch_bounds__(256) void cld_bits_kernel(const unsigned* in, const unsigned* xsrc, unsigned* out, BitVol b) {
    const long long total = (long long)b.rows_z * b.H * b.WX;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int wx = (int)(i % b.WX);
        const long long r = i / b.WX;
        const int y = (int)(r % b.H);
        const long long pz = r / b.H;
        const int z = (int)(pz % b.D);
        const unsigned tail = bit_tail(b, wx), tl = wx > 0 ? bit_tail(b, wx - 1) : 0u, tr = wx + 1 < b.WX ? bit_tail(b, wx + 1) : 0u;
        unsigned acc = MODE == 0 ? 0xffffffffu : 0u;
        for (int dz = (b.nd3 ? -1 : 0); dz <= (b.nd3 ? 1 : 0); ++dz) {
            if ((unsigned)(z + dz) >= (unsigned)b.D) continue;               // rows outside the volume: neutral for both pools
            for (int dy = -1; dy <= 1; ++dy) {
                if ((unsigned)(y + dy) >= (unsigned)b.H) continue;
                const unsigned* row = in + ((pz + dz) * b.H + (y + dy)) * b.WX;
                unsigned c = row[wx], l = wx > 0 ? row[wx - 1] : 0u, rr = wx + 1 < b.WX ? row[wx + 1] : 0u;
                if (MODE == 0) {           // erosion: everything outside the row's W voxels reads as 1
                    c |= tail;
                    l = wx > 0 ? (l | tl) : 0xffffffffu;
                    rr = wx + 1 < b.WX ? (rr | tr) : 0xffffffffu;
                    acc &= c & ((c >> 1) | (rr << 31)) & ((c << 1) | (l >> 31));
                } else {                   // dilation: outside reads as 0 (the stored tails are 0)
                    acc |= c | ((c >> 1) | (rr << 31)) | ((c << 1) | (l >> 31));
                }
            }
        }
        if (MODE == 0) out[i] = acc & ~tail;
        else out[i] = xsrc[i] & ~(acc & ~in[i]) & ~tail;
    }
}
__global__ __launch_bounds__(256) void cld_bits_expand_kernel(const unsigned* bits, float* out, BitVol b, long long nrows) {
    const long long total = nrows * b.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % b.W);
        const long long r = i / b.W;
        out[i] = (float)((bits[r * b.WX + (xx >> 5)] >> (xx & 31)) & 1u);
    }
}

// Backward of one skeleton iteration as two GATHER passes over LDS tiles (no atomics, no zero-fill, deterministic):
//   A: for every window w on the tile + 1 halo: u = maxpool(e)(w) - e(w), gt = g * [x - relu(u) > 0], gw = gt * [u > 0] and the
//      position am(w) of the FIRST maximum of e in w;  dx_direct(q) = gt(q),  de(q) = gw(q) - sum_{w in N(q), am(w) = q} gw(w)
//   B: am(w) = first minimum of x in w;  dx(q) = dx_direct(q) + sum_{w in N(q), am(w) = q} de(w)
// (the scatter form above routes the same terms with fp32 atomics).
// FOUR x-consecutive windows / voxels per thread (round 5; see skel_iter4_kernel): a window row of the four is one
// ds_read_b128 + one ds_read_b64 of `src` (18 LDS instructions for four windows instead of 108), the gather reads (first-extremum index, weight)
// rows the same way (36 instead of 216).  Comparisons and additions happen in the same (dz, dy, dx) order per window / voxel as in
// the one-voxel-per-thread form of rounds 1-4, so the results are bit-identical to it.  The 20 launches of that form were the largest item of the clDice term:
// 1.33 ms of the 6.3 ms C5 + clDice step (profiles/r05_kernel_stats_c5_cldice.txt).
template <int TZ, int TY, int TX, bool ND3, bool PASS_B>
__global__ __launch_bounds__(256) void skel_bwd_tile4_kernel(const float* g, const float* x, const float* e, float* dx, float* de, Vol v) {
    static_assert(TX % 4 == 0, "four voxels per thread along x");
    constexpr int HZ = ND3 ? 2 : 0, H1 = ND3 ? 1 : 0, NKZ = ND3 ? 3 : 1;
    constexpr int XZ = TZ + 2 * HZ, XY = TY + 4, XX = TX + 4;                  // source (e in pass A, x in pass B) with a 2-voxel halo
    constexpr int EZ = TZ + 2 * H1, EY = TY + 2, EX = TX + 2, EXP = TX + 4;    // windows: tile + 1 halo; row pitch padded to a multiple of 4
    __shared__ __attribute__((aligned(16))) float src[XZ * XY * XX + 4];
    __shared__ __attribute__((aligned(16))) float gws[EZ * EY * EXP + 4];
    __shared__ __attribute__((aligned(16))) int ams[EZ * EY * EXP + 4];
    const float INF = __builtin_huge_valf();
    const int ntx = (v.W + TX - 1) / TX, nty = (v.H + TY - 1) / TY, ntz = (v.D + TZ - 1) / TZ;
    int b = blockIdx.x;
    const int x0 = (b % ntx) * TX; b /= ntx;
    const int y0 = (b % nty) * TY; b /= nty;
    const int z0 = (b % ntz) * TZ;
    const int p = b / ntz;
    const long long V = (long long)v.D * v.H * v.W, base = (long long)p * V;
    const float* sp = (PASS_B ? x : e) + base;
    constexpr int NE = XZ * XY * XX, NIT = (NE + 255) / 256;
    constexpr int EG = (EX + 3) / 4, NEG = EZ * EY * EG, WIT = (NEG + 255) / 256;
    constexpr int OG = TX / 4, NOG = TZ * TY * OG, TIT = (NOG + 255) / 256;
    float w0[WIT][4], w1[WIT][4], t0[TIT][4];
    {
        float hv[NIT];
        bool hin[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int lx = i % XX, ly = (i / XX) % XY, lz = i / (XX * XY);
            const int gx = x0 + lx - 2, gy = y0 + ly - 2, gz = z0 + lz - HZ;
            hin[k] = i < NE && (unsigned)gx < (unsigned)v.W && (unsigned)gy < (unsigned)v.H && (unsigned)gz < (unsigned)v.D;
            hv[k] = sp[hin[k] ? ((long long)gz * v.H + gy) * v.W + gx : 0];
        }
#pragma unroll
        for (int k = 0; k < WIT; ++k) {
            const int i = threadIdx.x + k * 256;
            const int a = (i % EG) * 4, ly = (i / EG) % EY, lz = i / (EG * EY);
            const int gy = y0 + ly - 1, gz = z0 + lz - H1;
            const bool rin = i < NEG && (unsigned)gy < (unsigned)v.H && (unsigned)gz < (unsigned)v.D;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gx = x0 + a + j - 1;
                const bool in = rin && a + j < EX && (unsigned)gx < (unsigned)v.W;
                const long long o = in ? base + ((long long)gz * v.H + gy) * v.W + gx : base;
                w0[k][j] = PASS_B ? de[o] : x[o];
                w1[k][j] = PASS_B ? 0.f : g[o];
            }
        }
        if (PASS_B) {
#pragma unroll
            for (int k = 0; k < TIT; ++k) {
                const int i = threadIdx.x + k * 256;
                const int o4 = (i % OG) * 4, ly = (i / OG) % TY, lz = i / (OG * TY);
                const int gy = y0 + ly, gz = z0 + lz;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = x0 + o4 + j;
                    const bool in = i < NOG && gx < v.W && gy < v.H && gz < v.D;
                    t0[k][j] = dx[in ? base + ((long long)gz * v.H + gy) * v.W + gx : base];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = threadIdx.x + k * 256;
            if (i < NE) src[i] = hin[k] ? hv[k] : (PASS_B ? INF : -INF);      // never the extremum
        }
        if (threadIdx.x < 4) src[NE + threadIdx.x] = PASS_B ? INF : -INF;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < WIT; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i >= NEG) continue;
        const int a = (i % EG) * 4, ly = (i / EG) % EY, lz = i / (EG * EY);
        const int gy = y0 + ly - 1, gz = z0 + lz - H1;
        const bool rin = (unsigned)gy < (unsigned)v.H && (unsigned)gz < (unsigned)v.D;
        float best[4], ec[4] = {0.f, 0.f, 0.f, 0.f};
        int am[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { best[j] = PASS_B ? INF : -INF; am[j] = -1; }
        // first extremum of each window in (z, y, x) scan order; the padding holds -/+inf and can never win the strict comparison
#pragma unroll
        for (int dz = 0; dz < NKZ; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int rb = ((lz + dz) * XY + ly + dy) * XX + a;
                const vec<float, 4> q4 = *(const vec<float, 4>*)&src[rb];
                const vec<float, 2> q2 = *(const vec<float, 2>*)&src[rb + 4];
                const float q[6] = {q4[0], q4[1], q4[2], q4[3], q2[0], q2[1]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dx_ = 0; dx_ < 3; ++dx_) {
                        const float val = q[j + dx_];
                        const bool better = PASS_B ? val < best[j] : val > best[j];
                        best[j] = better ? val : best[j];
                        am[j] = better ? rb + j + dx_ : am[j];
                    }
                if (dz == HZ - H1 && dy == 1) { ec[0] = q[1]; ec[1] = q[2]; ec[2] = q[3]; ec[3] = q[4]; }     // the centres of the four windows
            }
        vec<float, 4> wo;
        vec<int, 4> ao;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gx = x0 + a + j - 1;
            float wgt = 0.f;
            int amj = -1;
            if (rin && a + j < EX && (unsigned)gx < (unsigned)v.W) {
                amj = am[j];
                if (PASS_B) {
                    wgt = w0[k][j];
                } else {
                    const float u = best[j] - ec[j];
                    const float gt = (w0[k][j] - fmaxf(u, 0.f) > 0.f) ? w1[k][j] : 0.f;
                    wgt = u > 0.f ? gt : 0.f;
                    // tile-interior windows also publish the direct term of dx
                    const int tx = a + j - 1, ty = ly - 1, tz = lz - H1;
                    if ((unsigned)tx < (unsigned)TX && (unsigned)ty < (unsigned)TY && (unsigned)tz < (unsigned)TZ)
                        dx[base + ((long long)gz * v.H + gy) * v.W + gx] = gt;
                }
            }
            wo[j] = wgt; ao[j] = amj;
        }
        *(vec<float, 4>*)&gws[(lz * EY + ly) * EXP + a] = wo;
        *(vec<int, 4>*)&ams[(lz * EY + ly) * EXP + a] = ao;
    }
    if (threadIdx.x < 4) { gws[EZ * EY * EXP + threadIdx.x] = 0.f; ams[EZ * EY * EXP + threadIdx.x] = -1; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TIT; ++k) {
        const int i = threadIdx.x + k * 256;
        if (i >= NOG) continue;
        const int o4 = (i % OG) * 4, ly = (i / OG) % TY, lz = i / (OG * TY);
        const int gx = x0 + o4, gy = y0 + ly, gz = z0 + lz;
        if (gx >= v.W || gy >= v.H || gz >= v.D) continue;
        const int me = ((lz + HZ) * XY + ly + 2) * XX + o4 + 2;          // the first voxel's index in `src`; voxel j sits at me + j
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, gc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dz = 0; dz < NKZ; ++dz)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int rb = ((lz + dz) * EY + ly + dy) * EXP + o4;
                const vec<int, 4> a4 = *(const vec<int, 4>*)&ams[rb];
                const vec<int, 2> a2 = *(const vec<int, 2>*)&ams[rb + 4];
                const vec<float, 4> g4 = *(const vec<float, 4>*)&gws[rb];
                const vec<float, 2> g2 = *(const vec<float, 2>*)&gws[rb + 4];
                const int aa[6] = {a4[0], a4[1], a4[2], a4[3], a2[0], a2[1]};
                const float gg[6] = {g4[0], g4[1], g4[2], g4[3], g2[0], g2[1]};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int dx_ = 0; dx_ < 3; ++dx_)
                        if (aa[j + dx_] == me + j) acc[j] += gg[j + dx_];
                if (dz == H1 && dy == 1) { gc[0] = gg[1]; gc[1] = gg[2]; gc[2] = gg[3]; gc[3] = gg[4]; }       // this voxel's own window
            }
        const long long o = base + ((long long)gz * v.H + gy) * v.W + gx;
        vec<float, 4> r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = PASS_B ? t0[k][j] + acc[j] : gc[j] - acc[j];
        float* dst = PASS_B ? dx : de;
        if (gx + 3 < v.W && (v.W & 3) == 0) {
            *(vec<float, 4>*)&dst[o] = r;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (gx + j < v.W) dst[o + j] = r[j];
        }
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
void launch_skel_iter(const float* x, float* e_out, float* x_out, int planes, int D, int H, int W, int nd, hipStream_t s) {
    // four x-consecutive voxels per thread (round 5; the one-voxel-per-thread kernels of rounds 1-4 gave the same bits and are gone)
    if (nd == 3) {
        Vol v{planes, D, H, W, 3};
        // 8 x 8 x 32 tiles: 1.66 instead of 1.99 halo windows per output voxel (C5 + clDice 5.72 vs 5.81 ms per step, profiles/r05_cldice_tile_ab.log);
        // shallow volumes take 4 x 8 x 32 tiles
        if (D >= 8) {
            const long long nb8 = (long long)planes * ((D + 7) / 8) * ((H + 7) / 8) * ((W + 31) / 32);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_iter4_kernel<8, 8, 32, true>), dim3((unsigned)nb8), dim3(256), 0, s, x, e_out, x_out, v);
        } else {
            const long long nb = (long long)planes * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 31) / 32);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_iter4_kernel<4, 8, 32, true>), dim3((unsigned)nb), dim3(256), 0, s, x, e_out, x_out, v);
        }
    } else {
        // 2-D pooling: every depth slice of every plane is an independent image
        Vol v{planes * D, 1, H, W, 2};
        const long long nb = (long long)planes * D * ((H + 15) / 16) * ((W + 63) / 64);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_iter4_kernel<1, 16, 64, false>), dim3((unsigned)nb), dim3(256), 0, s, x, e_out, x_out, v);
    }
}
void launch_skel_iter_bwd(const float* g, const float* x, const float* e, float* dx, float* de, int planes, int D, int H, int W, int nd,
                          hipStream_t s) {
    if (nd == 3) {
        Vol v{planes, D, H, W, 3};
        if (D >= 8) {
            const long long nb8 = (long long)planes * ((D + 7) / 8) * ((H + 7) / 8) * ((W + 31) / 32);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_bwd_tile4_kernel<8, 8, 32, true, false>), dim3((unsigned)nb8), dim3(256), 0, s, g, x, e, dx, de, v);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_bwd_tile4_kernel<8, 8, 32, true, true>), dim3((unsigned)nb8), dim3(256), 0, s, g, x, e, dx, de, v);
        } else {
            const long long nb = (long long)planes * ((D + 3) / 4) * ((H + 7) / 8) * ((W + 31) / 32);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_bwd_tile4_kernel<4, 8, 32, true, false>), dim3((unsigned)nb), dim3(256), 0, s, g, x, e, dx, de, v);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_bwd_tile4_kernel<4, 8, 32, true, true>), dim3((unsigned)nb), dim3(256), 0, s, g, x, e, dx, de, v);
        }
    } else {
        Vol v{planes * D, 1, H, W, 2};
        const long long nb = (long long)planes * D * ((H + 15) / 16) * ((W + 63) / 64);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_bwd_tile4_kernel<1, 16, 64, false, false>), dim3((unsigned)nb), dim3(256), 0, s, g, x, e, dx, de, v);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(skel_bwd_tile4_kernel<1, 16, 64, false, true>), dim3((unsigned)nb), dim3(256), 0, s, g, x, e, dx, de, v);
    }
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
size_t plane_dot_scratch_bytes(int planes, long long V) { return (size_t)planes * cdiv(V, 4096) * 2 * sizeof(double); }
void launch_plane_dot(const float* a, const float* b, double* out, double* scratch, int planes, long long V, hipStream_t s) {
    const int nblk = cdiv(V, 4096);
    hipLaunchKernelGGL(plane_dot_kernel, dim3(nblk, planes), dim3(256), 0, s, a, b, scratch, V);
    hipLaunchKernelGGL(plane_dot_fold_kernel, dim3(planes), dim3(256), 0, s, (const double*)scratch, out, nblk);
}
void launch_plane_axpb(const float* in, const float* a, const float* b, float* out, int planes, long long V, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(plane_axpb_kernel, dim3(blocks_for(V * planes)), dim3(256), 0, s, in, a, b, out, V, planes, accumulate);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Binary soft-clDice as ONE engine call (model/lossescldice.py:37-59, repaired as in oracle/make_golden.py:CLDICE_REPAIRS): both skeletons,
// the two normalised intersections, the ratio, and the whole backward down to d loss / d logit — no host round trip, no allocation
// (caller-planned workspace), every volume-sized pass in the kernels above.  Same arithmetic as the autograd composition in
// pytorchdeeplearing_amd/lossescldice.py (tests/test_cldice.py compares the two).
// ------------------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void cld_labels_kernel(const void* target, int lt, float* y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) y[i] = (float)load_label(target, lt, i);
}
// sums1[p] = {sum cl_pred*y, sum cl_pred}, sums2[p] = {sum cl_tgt*pred, sum cl_tgt}.  coef = [a1 | b1 | a2 | zero] (P floats each):
// d loss / d cl_pred = a1*y + b1, direct d loss / d pred = a2*cl_tgt, all times `gscale` (loss weight x loss scale).
__global__ void cld_finalize_kernel(const double* sums1, const double* sums2, int P, float gscale, float* out1, float* coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double smooth = 1e-5, eps = 1e-7;
    double I = 0.0, A = 0.0, B = 0.0;
    for (int p = 0; p < P; ++p) {
        const double fi = (double)(float)((sums1[2 * p] + 1.0) / (sums1[2 * p + 1] + 1.0));     // the autograd path keeps the ratios in fp32
        const double ft = (double)(float)((sums2[2 * p] + 1.0) / (sums2[2 * p + 1] + 1.0));
        I += fi * ft; A += fi; B += ft;
    }
    double D = A + B + smooth;
    const double live = D >= eps ? 1.0 : 0.0;            // clamp_min passes the gradient only above the floor
    if (D < eps) D = eps;
    const double num = 2.0 * I + smooth;
    out1[0] = (float)(1.0 - num / D);
    for (int p = 0; p < P; ++p) {
        const double in1 = sums1[2 * p] + 1.0, s1 = sums1[2 * p + 1] + 1.0, s2 = sums2[2 * p + 1] + 1.0;
        const double fi = (double)(float)(in1 / s1), ft = (double)(float)((sums2[2 * p] + 1.0) / s2);
        const double gi = -(2.0 * ft / D - live * num / (D * D)) * gscale;      // d loss / d iflat_p
        const double gt = -(2.0 * fi / D - live * num / (D * D)) * gscale;      // d loss / d tflat_p
        coef[p] = (float)(gi / s1);
        coef[P + p] = (float)(-gi * in1 / (s1 * s1));
        coef[2 * P + p] = (float)(gt / s2);
        coef[3 * P + p] = 0.f;
    }
}
__global__ __launch_bounds__(256) void cld_sigmoid_accum_kernel(const float* dprobs, const float* probs, float* dlogits, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float p = probs[i];
        dlogits[i] += dprobs[i] * p * (1.f - p);
    }
}
struct CldLayout { size_t y, work, tgt, grad, sums, scratch, coef, total; };
CldLayout cld_layout(int P, long long V, int width) {
    CldLayout L;
    const size_t vol = ((size_t)P * V * sizeof(float) + 255) / 256 * 256;
    size_t cur = 0;
    L.y = cur; cur += vol;
    L.work = cur; cur += (size_t)2 * (width > 0 ? width : 1) * vol;
    L.tgt = cur; cur += 3 * vol;
    L.grad = cur; cur += 3 * vol;
    L.sums = cur; cur += ((size_t)4 * P * sizeof(double) + 255) / 256 * 256;
    L.scratch = cur; cur += (plane_dot_scratch_bytes(P, V) + 255) / 256 * 256;
    L.coef = cur; cur += ((size_t)4 * P * sizeof(float) + 255) / 256 * 256;
    L.total = cur;
    return L;
}
}  // namespace

size_t cldice_binary_ws_bytes(int planes, long long V, int width) { return cld_layout(planes, V, width).total; }

// the part that depends on the labels alone (float labels + their skeleton): may run on another stream next to the forward pass
void launch_cldice_target(const void* target, int label_type, int planes, int D, int H, int W, int nd, int width, void* ws_, hipStream_t s) {
    const long long V = (long long)D * H * W, n = (long long)planes * V;
    const CldLayout L = cld_layout(planes, V, width);
    char* ws = (char*)ws_;
    const size_t vol = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    float* y = (float*)(ws + L.y);
    float* t[3] = {(float*)(ws + L.tgt), (float*)(ws + L.tgt + vol), (float*)(ws + L.tgt + 2 * vol)};
    // The target is read as a BINARY mask, y = (label != 0) - what the reference's train loop hands to every loss (model/modelVNet.py:576: labels are
    // binarised before the loss; include/segengine.h says so for seg_cldice_target / seg_cldice_binary) - and its skeleton is computed on bits
    // (cld_bits_kernel): exactly the values of the fp32 iteration on that mask.  The three bit images (current x, eroded e, next x; rows padded to whole
    // 32-voxel words, images to 64 words) live in the fp32 scratch volume t[0]; where they do not fit into it - a few hundred voxels, or rows of one or
    // two voxels (they used to spill into t[1] / t[2], which the expand pass writes) - the fp32 tile kernels compute the same skeleton.
    const long long nwords_all = (long long)planes * D * H * ((W + 31) / 32);
    const bool bits_fit = (size_t)3 * (size_t)((nwords_all + 63) / 64 * 64) * sizeof(unsigned) <= vol;
    if (width <= 0 || !bits_fit) {
        hipLaunchKernelGGL(cld_labels_kernel, dim3(blocks_for(n)), dim3(256), 0, s, target, label_type | LT_BINARIZE, y, n);
        const float* tc = y;
        for (int it = 0; it < width; ++it) { float* nx = t[1 + (it & 1)]; launch_skel_iter(tc, t[0], nx, planes, D, H, W, nd, s); tc = nx; }
        return;
    }
    BitVol b{nd == 3 ? planes * D : planes * D, nd == 3 ? D : 1, H, W, (W + 31) / 32, nd == 3 ? 1 : 0};
    const long long nrows = (long long)planes * D * H, nwords = nrows * b.WX;
    // three bit images in the fp32 scratch volume t[0] (which the bit path does not need): current x, eroded e, next x
    unsigned* bx = (unsigned*)t[0];
    unsigned* be = bx + ((nwords + 63) / 64) * 64;
    unsigned* bn = be + ((nwords + 63) / 64) * 64;
    hipLaunchKernelGGL(cld_labels_bits_kernel, dim3(blocks_for(nrows * 64)), dim3(256), 0, s, target, label_type, y, bx, b, nrows);
    for (int it = 0; it < width; ++it) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cld_bits_kernel<0>), dim3(blocks_for(nwords)), dim3(256), 0, s, (const unsigned*)bx, (const unsigned*)nullptr, be, b);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cld_bits_kernel<1>), dim3(blocks_for(nwords)), dim3(256), 0, s, (const unsigned*)be, (const unsigned*)bx, bn, b);
        unsigned* sw = bx; bx = bn; bn = sw;
    }
    hipLaunchKernelGGL(cld_bits_expand_kernel, dim3(blocks_for(n)), dim3(256), 0, s, (const unsigned*)bx, t[1 + ((width - 1) & 1)], b, nrows);
}

void launch_cldice_binary(const float* probs, const void* target, int label_type, int planes, int D, int H, int W, int nd, int width,
                          float gscale, void* ws_, float* out1, float* dlogits, int target_ready, hipStream_t s) {
    const long long V = (long long)D * H * W, n = (long long)planes * V;
    const CldLayout L = cld_layout(planes, V, width);
    char* ws = (char*)ws_;
    const size_t vol = ((size_t)n * sizeof(float) + 255) / 256 * 256;
    float* y = (float*)(ws + L.y);
    auto work = [&](int it, int which) { return (float*)(ws + L.work + ((size_t)it * 2 + which) * vol); };     // 0: e_it, 1: x_{it+1}
    float* t[3] = {(float*)(ws + L.tgt), (float*)(ws + L.tgt + vol), (float*)(ws + L.tgt + 2 * vol)};
    float* g[3] = {(float*)(ws + L.grad), (float*)(ws + L.grad + vol), (float*)(ws + L.grad + 2 * vol)};
    double* sums1 = (double*)(ws + L.sums);
    double* sums2 = sums1 + 2 * planes;
    double* scratch = (double*)(ws + L.scratch);
    float* coef = (float*)(ws + L.coef);
    if (!target_ready) launch_cldice_target(target, label_type, planes, D, H, W, nd, width, ws_, s);
    // skeleton of the prediction: every iteration's input and eroded image stay for the backward pass
    const float* cur = probs;
    for (int it = 0; it < width; ++it) { launch_skel_iter(cur, work(it, 0), work(it, 1), planes, D, H, W, nd, s); cur = work(it, 1); }
    const float* cl_pred = cur;
    // skeleton of the target (launch_cldice_target): the last iterate sits in the ping-pong buffer of iteration width-1
    const float* cl_tgt = width > 0 ? t[1 + ((width - 1) & 1)] : y;
    launch_plane_dot(cl_pred, y, sums1, scratch, planes, V, s);
    launch_plane_dot(cl_tgt, probs, sums2, scratch, planes, V, s);
    hipLaunchKernelGGL(cld_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)sums1, (const double*)sums2, planes, gscale, out1, coef);
    if (!dlogits) return;
    // backward: d cl_pred (affine map of y per plane) through the skeleton iterations, plus the direct term through tflat
    float* gc = g[0];
    launch_plane_axpb(y, coef, coef + planes, gc, planes, V, 0, s);
    for (int it = width - 1; it >= 0; --it) {
        float* dx = (gc == g[0]) ? g[1] : g[0];
        launch_skel_iter_bwd(gc, it == 0 ? probs : work(it - 1, 1), work(it, 0), dx, g[2], planes, D, H, W, nd, s);
        gc = dx;
    }
    launch_plane_axpb(cl_tgt, coef + 2 * planes, coef + 3 * planes, gc, planes, V, 1, s);
    hipLaunchKernelGGL(cld_sigmoid_accum_kernel, dim3(blocks_for(n)), dim3(256), 0, s, (const float*)gc, probs, dlogits, n);
}

}  // namespace seg
