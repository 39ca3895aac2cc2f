// TEST INFRASTRUCTURE ONLY — host-side checker for the wave64 execution model.
//
// This header is NOT a compatibility layer and never ships in the product library.  It exists so
// that the *unmodified* gfx950 kernel sources under pytorchdeeplearing_amd/csrc can be compiled
// with the host clang (`-x c++ -I tests/emu`) and their indexing / tiling / reduction logic checked
// against the oracle on the GPU-less build container before GPU minutes are spent.  It models:
//   * a workgroup as N fibers (one per lane) in one OS thread, grouped into 64-lane waves;
//   * __syncthreads() as a block-wide rendezvous; LDS (`__shared__`) as block-static storage;
//   * wave collectives (MFMA 16x16x32 f16/bf16, 16x16x4 f32, 32x32 forms, ds_read_b64_tr_b16,
//     __shfl*) with the gfx950 lane<->element maps documented in
//     /opt/skills/guides/cdna_hip_programming.md §3 and §2;
//   * atomics as plain read-modify-write (blocks run one after another).
// Nothing here is timed, and nothing in the product package can load the resulting library:
// tests inject it explicitly (tests/emu/build_emu.py).
#pragma once

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define SEG_EMU 1

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
typedef void* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };

inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
template <class P> inline hipError_t hipMalloc(P** p, size_t n) { *p = (P*)malloc(n); return *p ? hipSuccess : 1; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
// stream memory operations (flag forks of the engine): launches run immediately and in program order here, so a wait has nothing to wait for;
// the flag word itself is ordinary memory the kernels store to (seg_plan_count 9 checks that every number handed out was stored)
enum hipDeviceAttribute_t { hipDeviceAttributeCanUseStreamWaitValue = 1 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 1; return hipSuccess; }
constexpr unsigned hipMallocSignalMemory = 2, hipStreamWaitValueGte = 0;
inline hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { *p = calloc(1, n); return *p ? hipSuccess : 1; }
inline hipError_t hipStreamWaitValue32(hipStream_t, void*, uint32_t, unsigned, uint32_t = 0xffffffffu) { return hipSuccess; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
constexpr unsigned hipStreamNonBlocking = 1;
constexpr unsigned hipEventDisableTiming = 2;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
// stream capture / graphs: not modelled - the checker executes every launch immediately, so a capture is refused and the callers fall back
typedef void* hipGraph_t;
typedef void* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
constexpr hipError_t hipErrorNotSupported = 801;
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, unsigned long long) { *e = nullptr; return hipErrorNotSupported; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ alignas(16) static
#define __restrict__ __restrict

namespace emu {

enum YieldKind { Y_NONE = 0, Y_WAVE = 1, Y_BLOCK = 2, Y_DONE = 3 };

// Minimal x86-64 SysV context switch (callee-saved registers + stack pointer); glibc's swapcontext
// costs a sigprocmask syscall per switch, which dominated the checker's run time.
__attribute__((naked, noinline)) static void ctx_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
        "movq %rsp, (%rdi)\n\t"
        "movq %rsi, %rsp\n\t"
        "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
        "ret\n\t");
}

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    dim3 tid;
    int lane = 0, wave = 0;
    YieldKind yk = Y_NONE;
};

struct WaveScratch {
    alignas(16) unsigned char a[64][64];   // per-lane operand A (up to 64 B)
    alignas(16) unsigned char b[64][64];   // per-lane operand B
};

struct State {
    void* sched_sp = nullptr;
    std::vector<Fiber> fibers;
    std::vector<WaveScratch> ws;
    Fiber* cur = nullptr;
    dim3 blockIdx_, blockDim_, gridDim_;
    const std::function<void()>* body = nullptr;
};

inline State& S() {
    static State s;
    return s;
}

inline void yield(YieldKind k) {
    State& s = S();
    Fiber* f = s.cur;
    f->yk = k;
    ctx_switch(&f->sp, s.sched_sp);
}

inline void fiber_entry() {
    State& s = S();
    (*s.body)();
    s.cur->yk = Y_DONE;
    ctx_switch(&s.cur->sp, s.sched_sp);
    abort();   // a finished fiber is never resumed
}

constexpr size_t kStack = 256 * 1024;

inline void run_block(const std::function<void()>& body, unsigned nthreads) {
    State& s = S();
    assert(nthreads % 64 == 0 && "emu: block size must be a multiple of the 64-lane wave");
    s.body = &body;
    if (s.fibers.size() < nthreads) {
        size_t old = s.fibers.size();
        s.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; ++i) s.fibers[i].stack = (char*)aligned_alloc(64, kStack);
    }
    const unsigned nw = nthreads / 64;
    if (s.ws.size() < nw) s.ws.resize(nw);
    for (unsigned i = 0; i < nthreads; ++i) {
        Fiber& f = s.fibers[i];
        // initial frame: six zeroed callee-saved slots + the entry address consumed by `ret`
        void** top = (void**)(f.stack + kStack);          // 64-byte aligned
        top[-1] = nullptr;                                 // keeps rsp % 16 == 8 at fiber_entry
        top[-2] = (void*)&fiber_entry;
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
        f.sp = (void*)(top - 8);
        f.tid = dim3(i % s.blockDim_.x, (i / s.blockDim_.x) % s.blockDim_.y, i / (s.blockDim_.x * s.blockDim_.y));
        f.lane = i % 64;
        f.wave = i / 64;
        f.yk = Y_NONE;
    }
    // Scheduler: run each wave until it reaches a block barrier (or finishes); wave-level
    // collectives only rendezvous the 64 lanes of that wave.
    std::vector<int> wstate(nw, Y_NONE);
    for (;;) {
        bool all_done = true;
        for (unsigned w = 0; w < nw; ++w) {
            if (wstate[w] == Y_DONE) continue;
            all_done = false;
            for (;;) {
                int kind = -1;
                for (unsigned l = 0; l < 64; ++l) {
                    Fiber& f = s.fibers[w * 64 + l];
                    if (f.yk == Y_DONE) { if (kind == -1) kind = Y_DONE; else if (kind != Y_DONE) { fprintf(stderr, "emu: lane %u of wave %u exited while siblings wait in a collective\n", l, w); abort(); } continue; }
                    s.cur = &f;
                    ctx_switch(&s.sched_sp, f.sp);
                    if (kind == -1) kind = f.yk;
                    else if (kind != (int)f.yk) {
                        fprintf(stderr, "emu: divergent collective in wave %u (lane %u: %d vs %d) block (%u,%u,%u)\n", w, l, (int)f.yk, kind,
                                s.blockIdx_.x, s.blockIdx_.y, s.blockIdx_.z);
                        abort();
                    }
                }
                wstate[w] = kind;
                if (kind != Y_WAVE) break;   // block barrier or done: switch to the next wave
            }
        }
        if (all_done) break;
        // every live wave must now be parked at the block barrier
        for (unsigned w = 0; w < nw; ++w)
            if (wstate[w] == Y_BLOCK) wstate[w] = Y_NONE;
            else if (wstate[w] != Y_DONE) { fprintf(stderr, "emu: barrier mismatch\n"); abort(); }
    }
}

template <class F>
inline void launch(dim3 grid, dim3 block, F&& f) {
    State& s = S();
    s.gridDim_ = grid;
    s.blockDim_ = block;
    std::function<void()> body = f;
    const unsigned nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                s.blockIdx_ = dim3(bx, by, bz);
                run_block(body, nthreads);
            }
}

inline Fiber& me() { return *S().cur; }
inline WaveScratch& wsc() { return S().ws[me().wave]; }

}  // namespace emu

#define threadIdx (emu::S().cur->tid)
#define blockIdx (emu::S().blockIdx_)
#define blockDim (emu::S().blockDim_)
#define gridDim (emu::S().gridDim_)
#define warpSize 64

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    emu::launch(dim3(grid), dim3(block), [=]() { kern(__VA_ARGS__); })

inline void __syncthreads() { emu::yield(emu::Y_BLOCK); }
inline unsigned long long wall_clock64() { return 0ull; }   // diagnostics only (SEG_CONV3_TRACE is a GPU tool)
inline void __threadfence() {}

// ---- atomics (blocks/fibers are serialized) ----------------------------------------------------
template <class T>
inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float)v; return o; }
template <class T>
inline T atomicMax(T* p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <class T>
inline T atomicMin(T* p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <class T>
inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

// ---- wave shuffles ------------------------------------------------------------------------------
template <class T>
inline T emu_shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 64, "");
    auto& w = emu::wsc();
    const int l = emu::me().lane;
    memcpy(w.a[l], &v, sizeof(T));
    emu::yield(emu::Y_WAVE);
    T r;
    memcpy(&r, w.a[src & 63], sizeof(T));
    emu::yield(emu::Y_WAVE);
    return r;
}
template <class T>
inline T __shfl_xor(T v, int mask, int = 64) { return emu_shfl_idx(v, emu::me().lane ^ mask); }
template <class T>
inline T __shfl_down(T v, int d, int = 64) { int l = emu::me().lane; return emu_shfl_idx(v, l + d < 64 ? l + d : l); }
template <class T>
inline T __shfl(T v, int src, int = 64) { return emu_shfl_idx(v, src); }

// wave-wide vote: every lane of the wave must reach it (as on the hardware when EXEC is full)
inline unsigned long long __ballot(int pred) {
    auto& w = emu::wsc();
    const int l = emu::me().lane;
    const int p = pred != 0;
    memcpy(w.a[l], &p, sizeof(int));
    emu::yield(emu::Y_WAVE);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) { int q; memcpy(&q, w.a[i], sizeof(int)); if (q) m |= 1ull << i; }
    emu::yield(emu::Y_WAVE);
    return m;
}

// ---- MFMA ---------------------------------------------------------------------------------------
namespace emu {
template <class T, int N>
using vec = T __attribute__((ext_vector_type(N)));

// D(MxN) += A(MxK) B(KxN); A/B fragments hold KPL consecutive k per lane at k0 = KPL*(lane / M).
// C/D: 16x16 -> 4 regs, row = 4*(lane>>4)+r, col = lane&15
//      32x32 -> 16 regs, row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31
template <int M, int KPL, class TA, int NACC>
inline vec<float, NACC> mfma(const vec<TA, KPL>& a, const vec<TA, KPL>& b, vec<float, NACC> c) {
    auto& w = wsc();
    const int l = me().lane;
    memcpy(w.a[l], &a, sizeof(a));
    memcpy(w.b[l], &b, sizeof(b));
    yield(Y_WAVE);
    constexpr int KG = 64 / M;   // lane groups along k
    const int col = l % M;
    for (int r = 0; r < NACC; ++r) {
        const int row = (M == 16) ? 4 * (l >> 4) + r : (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int g = 0; g < KG; ++g) {
            vec<TA, KPL> fa, fb;
            memcpy(&fa, w.a[g * M + row], sizeof(fa));
            memcpy(&fb, w.b[g * M + col], sizeof(fb));
            for (int j = 0; j < KPL; ++j) acc = fmaf((float)fa[j], (float)fb[j], acc);
        }
        c[r] = acc;
    }
    yield(Y_WAVE);
    return c;
}
inline vec<float, 4> mfma_16x16x4f32(float a, float b, vec<float, 4> c) {
    vec<float, 1> va, vb;
    va[0] = a; vb[0] = b;
    return mfma<16, 1, float, 4>(va, vb, c);
}
inline vec<float, 16> mfma_32x32x2f32(float a, float b, vec<float, 16> c) {
    vec<float, 1> va, vb;
    va[0] = a; vb[0] = b;
    return mfma<32, 1, float, 16>(va, vb, c);
}

// ds_read_b64_tr_b16: per 16-lane group the lanes' 8-byte reads form a 4x16 block of b16
// (lane 4*r+s supplies row r, cols 4s..4s+3); lane t of the group receives column t (4 values).
inline vec<short, 4> ds_read_tr16_b64(const void* p) {
    auto& w = wsc();
    const int l = me().lane;
    assert(((uintptr_t)p & 7) == 0 && "ds_read_b64_tr_b16 needs an 8-byte aligned LDS address");
    memcpy(w.a[l], &p, sizeof(p));
    yield(Y_WAVE);
    vec<short, 4> r;
    const int g = l & ~15, t = l & 15;
    for (int j = 0; j < 4; ++j) {
        const short* src;
        memcpy(&src, w.a[g + 4 * j + (t >> 2)], sizeof(src));
        r[j] = src[t & 3];
    }
    yield(Y_WAVE);
    return r;
}
}  // namespace emu

#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu::mfma<16, 8, _Float16, 4>(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) emu::mfma<16, 8, __bf16, 4>(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma<32, 8, _Float16, 16>(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma<32, 8, __bf16, 16>(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu::mfma_16x16x4f32(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, x, y, z) emu::mfma<16, 4, _Float16, 4>(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, x, y, z) \
    emu::mfma<16, 4, __bf16, 4>(__builtin_bit_cast(emu::vec<__bf16, 4>, a), __builtin_bit_cast(emu::vec<__bf16, 4>, b), c)
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2f32(a, b, c)
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16(p) emu::ds_read_tr16_b64((const void*)(p))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
// direct global -> LDS copy (buffer_load_dwordx4 ... lds): lane l copies `size` bytes from base + voffset to lds + size*l;
// offsets at or beyond the resource's byte range store zeros.  Executed synchronously (the checker has no memory latency).
typedef int emu_i32x4 __attribute__((ext_vector_type(4)));
inline void seg_raw_buffer_load_lds(emu_i32x4 rsrc, __attribute__((address_space(3))) unsigned* lds, int size, int voffset, int soffset, int offset, int aux) {
    (void)aux;
    const unsigned long long base = (unsigned long long)(unsigned)rsrc[0] | ((unsigned long long)((unsigned)rsrc[1] & 0xffffu) << 32);
    const unsigned range = (unsigned)rsrc[2];
    const unsigned long long off = (unsigned long long)(unsigned)voffset + (unsigned)soffset + (unsigned)offset;
    unsigned char* dst = (unsigned char*)(uintptr_t)lds + (size_t)size * emu::me().lane;
    assert(((uintptr_t)dst & (size >= 16 ? 15 : 3)) == 0 && "LDS destination of a direct load must be aligned");
    if (off + size > range) memset(dst, 0, size);
    else memcpy(dst, (const unsigned char*)base + off, size);
}
inline emu_i32x4 seg_raw_buffer_load_b128(emu_i32x4 rsrc, int voffset, int soffset, int aux) {
    (void)aux;
    const unsigned long long base = (unsigned long long)(unsigned)rsrc[0] | ((unsigned long long)((unsigned)rsrc[1] & 0xffffu) << 32);
    const unsigned range = (unsigned)rsrc[2];
    const unsigned long long off = (unsigned long long)(unsigned)voffset + (unsigned)soffset;
    emu_i32x4 r = {0, 0, 0, 0};
    if (off + 16 <= range) memcpy(&r, (const unsigned char*)base + off, 16);
    return r;
}
#define HIP_KERNEL_NAME(...) __VA_ARGS__
