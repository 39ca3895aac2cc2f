// GroupNorm statistics finalize as a per-workgroup prologue (shared by norm.hip's elementwise passes and by the halo convs that apply the
// producer's GroupNorm + dropout + ReLU while staging their input, conv3x_impl.h).
#pragma once
#include "kernels.h"

namespace seg {

// The finalize of one sample inside a consumer workgroup (256 threads, C = 16..256): fold the `rep` replica partial sums,
// reduce the channels of a group with lane butterflies (the cpg = C/8 channels of a group are neighbouring lanes) and leave
// scale / shift of every channel in LDS.  Same formulas as gn_finalize_kernel.  `publish`: this workgroup also writes the
// per-(n, c) coefficients and per-(n, g) moments the backward pass reads.  Ends with a barrier.
__device__ __forceinline__ void gn_fold_block(const GnFinArgs& f, int n, bool publish, double (*part)[2], float* sc_s, float* sh_s) {
    const int tid = threadIdx.x, C = f.C, cpg = C / GN_GROUPS;
    const int S = 256 / C;                              // replica slices folded side by side (C <= 256)
    const int c = tid % C, sl = tid / C;
    const int nrep = f.rep > 0 ? f.rep : STAT_REP;
    // the per-channel parameters do not depend on the statistics: their loads travel together with the first replica loads instead of
    // forming a second L2 round trip behind the barrier (every workgroup of every elementwise launch runs this prologue)
    float mk = 1.f, ga = 0.f, be = 0.f;
    if (tid < C) {
        mk = f.mask ? f.mask[(long long)n * f.mask_ld + tid] : 1.f;
        ga = f.gamma[tid]; be = f.beta[tid];
    }
    double s = 0.0, ss = 0.0;
    for (int r0 = sl; r0 < nrep; r0 += 4 * S) {         // four independent 16-B loads per trip
        double v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rep = r0 + u * S;
            const double* st = f.stats + (((long long)(rep < nrep ? rep : sl) * f.N + n) * C + c) * 2;
            v0[u] = st[0]; v1[u] = st[1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r0 + u * S < nrep) { s += v0[u]; ss += v1[u]; }
    }
    part[tid][0] = s; part[tid][1] = ss;
    __syncthreads();
    s = 0.0; ss = 0.0;
    if (tid < C)
        for (int k = 0; k < S; ++k) { s += part[tid + k * C][0]; ss += part[tid + k * C][1]; }
    for (int o = cpg >> 1; o >= 1; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if (tid < C) {
        const double cnt = (double)cpg * (double)f.V;
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = mk * ga * rstd, sh = mk * (be - ga * (float)mean * rstd);
        sc_s[c] = sc; sh_s[c] = sh;
        if (publish) {
            f.scale[(long long)n * C + c] = sc;
            f.shift[(long long)n * C + c] = sh;
            if (c % cpg == 0) { f.mean[n * GN_GROUPS + c / cpg] = (float)mean; f.rstd[n * GN_GROUPS + c / cpg] = rstd; }
        }
    }
    __syncthreads();
}

}  // namespace seg
