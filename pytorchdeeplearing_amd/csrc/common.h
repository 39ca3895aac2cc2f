// Shared device helpers for the gfx950 segmentation engine (wave64, MFMA 16x16, NDHWC tensors).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace seg {

typedef _Float16 f16;
typedef __bf16 bf16;
template <class T, int N>
using vec = T __attribute__((ext_vector_type(N)));
typedef vec<float, 4> f32x4;
typedef vec<short, 4> s16x4;

enum DType { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };
template <class T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int v = DT_F32; };
template <> struct dtype_of<f16> { static constexpr int v = DT_F16; };
template <> struct dtype_of<bf16> { static constexpr int v = DT_BF16; };

constexpr int kWave = 64;
constexpr int GN_GROUPS = 8;
// Reduction targets that many workgroups hit (GroupNorm sums, loss sums) are replicated STAT_REP times and
// summed by the tiny finalize kernels: same-address atomics serialize at ~60 ns each on gfx950.
constexpr int STAT_REP = 32;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// One K-step of the implicit GEMMs is 32 reduction elements; every lane owns 8 consecutive k
// (k0 = 8*(lane>>4)) of row/col (lane&15).  f16/bf16: one v_mfma_f32_16x16x32.  f32: eight
// v_mfma_f32_16x16x4_f32, instruction j consuming k = 8*(lane>>4)+j of every lane group (A and B
// use the same k permutation, so the product is unchanged); exact f32 (fmaf chain).
template <class T> struct Mma;
template <> struct Mma<float> {
    typedef vec<float, 8> frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};
template <> struct Mma<f16> {
    typedef vec<f16, 8> frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    typedef vec<bf16, 8> frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

// K = 16 step (a lane owns k = 4*(lane>>4) .. +3 of row/col lane&15): the accumulator layout of a 16x16 tile (col = lane&15,
// rows 4*(lane>>4) + r) IS the B-operand layout of this step, so a tile computed by one MFMA feeds the next without leaving
// the registers (fused stem weight gradient, stemx.hip).  f32: four 16x16x4 steps, step j taking element j of every lane.
template <class T> struct Mma16;
template <> struct Mma16<float> {
    typedef vec<float, 4> frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
        return c;
    }
};
template <> struct Mma16<f16> {
    typedef vec<f16, 4> frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma16<bf16> {
    typedef vec<bf16, 4> frag;
    static __device__ __forceinline__ f32x4 run(const frag& a, const frag& b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};

// LDS transposing read (gfx950 ds_read_b64_tr_b16): within each 16-lane group the lanes' 8-byte
// reads form a [4][16] block of 16-bit values; lane t receives column t (4 consecutive rows).
__device__ __forceinline__ s16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

template <class T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <class T> __device__ __forceinline__ T from_f(float v) { return (T)v; }

template <class T> __device__ __forceinline__ vec<T, 8> zero8() {
    vec<T, 8> z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (T)0.0f;
    return z;
}
template <class T> __device__ __forceinline__ vec<T, 8> load8(const T* p) { return *(const vec<T, 8>*)p; }
template <class T> __device__ __forceinline__ void store8(T* p, const vec<T, 8>& v) { *(vec<T, 8>*)p = v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15); every lane receives its row's total.  Four v_add_f32 with a DPP operand (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror) instead of four ds_bpermute round trips through the LDS crossbar (~120 clk each, and hipcc puts an s_waitcnt behind every one:
// the 64 bpermutes of the conv3x statistics epilogue were 2.05 us of a 10.9 us workgroup, profiles/r05_conv3x_phase_trace_before.log).  The pairs that are added
// are the ones a butterfly over masks 1, 2, 4, 8 adds (a + b == b + a), so the result is bit-identical to it.
#ifndef SEG_EMU
template <int CTRL> __device__ __forceinline__ float dpp_take(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_take<0xB1>(v);       // quad_perm [1, 0, 3, 2]
    v += dpp_take<0x4E>(v);       // quad_perm [2, 3, 0, 1]
    v += dpp_take<0x141>(v);      // row_half_mirror
    v += dpp_take<0x140>(v);      // row_mirror
    return v;
}
// Sum over the lanes of a DPP row that are congruent modulo CG (1, 2 or 4): quad_perm xor 1 (CG = 1), xor 2 (CG <= 2), then row_ror 4 and 8
template <int CG> __device__ __forceinline__ float row_sum_mod(float v) {
    if (CG <= 1) v += dpp_take<0xB1>(v);
    if (CG <= 2) v += dpp_take<0x4E>(v);
    v += dpp_take<0x124>(v);      // row_ror:4
    v += dpp_take<0x128>(v);      // row_ror:8
    return v;
}
#else
__device__ __forceinline__ float row_sum16(float v) {
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
template <int CG> __device__ __forceinline__ float row_sum_mod(float v) {
#pragma unroll
    for (int m = CG; m < 16; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
#endif
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}


// ---- direct global -> LDS copies (gfx950 buffer_load_dwordx4 ... lds) -----------------------------------------------
// One wave-instruction moves 64 x 16 B: lane l copies the 16 bytes at (resource base + voffset_l) to
// (lds_wave_base + 16*l) without touching a VGPR; a voffset at or beyond the resource's byte range stores ZEROS
// (raw-buffer out-of-range semantics), which is how the zero padding of a halo is produced.  The LDS image of a
// wave-instruction is therefore lane-linear; layouts are swizzled by choosing which global piece a lane fetches.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned DMA_OOB = 0x80000000u;     // any offset >= the resource range (ranges are kept below 2 GB)
#ifndef SEG_EMU
__device__ void seg_raw_buffer_load_lds(i32x4 rsrc, __attribute__((address_space(3))) unsigned* lds, int size, int voffset, int soffset,
                                        int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
#endif
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(unsigned)(p & 0xffffffffull);
    r[1] = (int)(unsigned)((p >> 32) & 0xffffull);    // base[47:32], stride 0
    r[2] = (int)bytes;                                // num_records (bytes)
    r[3] = 0x00020000;                                // gfx9 raw buffer: 32-bit data format
    return r;
}
// 16-B load through a buffer resource: address = base + voffset (per lane) + soffset (wave-uniform, an SGPR); out-of-range
// reads return zeros, so prefetching past the end of a weight array needs neither a branch nor a clamp
#ifndef SEG_EMU
__device__ i32x4 seg_raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
#endif
template <class T> __device__ __forceinline__ vec<T, 8> buffer_load8(i32x4 rsrc, unsigned voffset, unsigned soffset) {
    return __builtin_bit_cast(vec<T, 8>, seg_raw_buffer_load_b128(rsrc, (int)voffset, (int)soffset, 0));
}
template <class T> __device__ __forceinline__ void dma16(i32x4 rsrc, T* lds_wave_base, unsigned voffset) {
    seg_raw_buffer_load_lds(rsrc, (__attribute__((address_space(3))) unsigned*)(lds_wave_base), 16, (int)voffset, 0, 0, 0);
}
// The same copy, invisible to the compiler's s_waitcnt bookkeeping: hipcc drains a dma16 (it may alias any ds_read of the same
// LDS array) before the next LDS read, which serialises "copy box i + 1 while box i is multiplied".  Written as one asm statement
// the copy stays in flight until the caller's own wait_vmem() + barrier.  M0 (the LDS destination base) is saved and restored
// inside the statement; the leading s_nop covers an SGPR operand freshly written by a VALU readfirstlane
// (cdna_hip_programming.md section 5.7).  The host checker has no asynchrony to model and runs the plain copy.
#ifndef SEG_EMU
template <class T> __device__ __forceinline__ void dma16_async(i32x4 rsrc, T* lds_wave_base, unsigned voffset) {
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)lds_wave_base);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voffset), "s"(lds_addr), "s"(rsrc)
                 : "memory");
}
#else
template <class T> __device__ __forceinline__ void dma16_async(i32x4 rsrc, T* lds_wave_base, unsigned voffset) { dma16(rsrc, lds_wave_base, voffset); }
#endif
// all of this wave's outstanding global loads / LDS copies have landed (s_waitcnt vmcnt(0); expcnt / lgkmcnt untouched)
__device__ __forceinline__ void wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// "this loaded value has arrived": an empty asm statement that reads v makes hipcc place the value's s_waitcnt HERE (in front of a
// loop) instead of in front of its first use inside the loop, where - vmcnt retiring in order - the wait also drains every younger
// load of the loop body and the "loads in flight" run one after the other
#ifndef SEG_EMU
__device__ __forceinline__ void settle(int& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void settle(float& v) { asm volatile("" : "+v"(v)); }
#else
__device__ __forceinline__ void settle(int&) {}
__device__ __forceinline__ void settle(float&) {}
#endif

// ---- one 8-byte word handed from one workgroup to others INSIDE a kernel (gn_bwd_coop_kernel) ----------------------------------------------
// Agent-scope relaxed atomics: the store goes to the memory side (sc1), the polling load misses every L2, so workgroups on different XCDs (whose L2s are
// not coherent with each other) see the word without a fence - a release fence would write back whatever the co-running weight-gradient kernels have
// dirty in this XCD's L2 (round 3 measured that: the step went from 6 to 13 ms).  The host checker runs workgroups one after another: plain accesses.
#ifndef SEG_EMU
__device__ __forceinline__ void xwg_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long xwg_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void xwg_pause() { __builtin_amdgcn_s_sleep(1); }
#else
__device__ __forceinline__ void xwg_store(unsigned long long* p, unsigned long long v) { *p = v; }
__device__ __forceinline__ unsigned long long xwg_load(const unsigned long long* p) { return *p; }
__device__ __forceinline__ void xwg_pause() {}
#endif

// labels arrive as u8 / i32 / i64 / f32 class ids (the reference's datasets hand out int64)
// LT_BINARIZE (flag, or-ed into the type): the label is read as (value != 0) - the `y[y != 0] = 1` of the reference's binary
// training loops (model/modelVNet.py:576, modelUnet.py:576) done by the consumer kernels instead of a host pass over the labels
enum LabelType { LT_U8 = 0, LT_I32 = 1, LT_I64 = 2, LT_F32 = 3, LT_BINARIZE = 16 };
__device__ __forceinline__ int load_label(const void* p, int lt, long long i) {
    int v;
    switch (lt & 15) {
        case LT_U8: v = (int)((const uint8_t*)p)[i]; break;
        case LT_I32: v = ((const int*)p)[i]; break;
        case LT_I64: v = (int)((const long long*)p)[i]; break;
        default: v = (int)((const float*)p)[i]; break;
    }
    return (lt & LT_BINARIZE) ? (v != 0) : v;
}

// ---- environment switches (host side): the library reads the switches documented in include/segengine.h and nothing else; each selects a complete,
// tested path.  The tuning knobs and measured-slower paths of rounds 1-5 are gone from the sources (round 6); what each one measured is in
// profiles/HISTORY.md.
inline const char* knob_s(const char* name) { return getenv(name); }
inline int knob_i(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace seg
