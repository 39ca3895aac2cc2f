// What does handing work to a second stream cost the FIRST stream?  (the train step forks its weight gradients to a second
// queue ~19 times per step; rocprofv3 shows the main queue idle ~6.4 us at every hipEventRecord - 3 % of the step.)
//
// main stream, R iterations:   produce(A_i, stamp i)  ->  [fork i]  ->  next(B)          (two ~10 us kernels)
// side stream (low priority):  [wait i]  ->  check(A_i == stamp i)                        (counts stale words)
// fork / wait mechanisms:
//   0  none, no side work                                  the main stream's own time
//   1  hipEventRecord(main) + hipStreamWaitEvent(side)     what the engine does today
//   2  a one-wave kernel on main stores i to a flag in signal memory (hipMallocSignalMemory), hipStreamWaitValue32(side, flag >= i): the command
//      processor waits, no CU spins
//   3  `next` itself stores the flag from its first thread (in-order queue: it starts after `produce` and its end-of-kernel release have
//      completed), hipStreamWaitValue32 on the side stream: nothing extra on the main queue at all
//   4  hipStreamWriteValue32(main) + hipStreamWaitValue32(side)
// Reported: main-stream time per iteration (events around the main loop), wall per iteration with the side work drained, stale words seen by `check`.
// Build: hipcc --offload-arch=gfx950 -O2 fork_cost.hip -o fork_cost        Run under `timeout`: a wait that never releases would hang the queue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ __launch_bounds__(256) void produce(int* A, int n, int stamp, unsigned* flag, unsigned seq) {
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) A[i] = stamp;
}
__global__ __launch_bounds__(256) void check(const int* A, int n, int stamp, int* errors) {
    int bad = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) bad += A[i] != stamp;
    if (bad) atomicAdd(errors, bad);
}
__global__ void sig(unsigned* flag, unsigned seq) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 200;
    const int n = 1 << 21;                                       // 8 MB per buffer: ~10 us to write
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    std::vector<int*> A(R);
    for (int i = 0; i < R; ++i) CK(hipMalloc(&A[i], (size_t)n * 4));
    int *B, *errors;
    CK(hipMalloc(&B, (size_t)n * 4)); CK(hipMalloc(&errors, 4));
    unsigned* flag = nullptr;
    const hipError_t se = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(8 bytes, hipMallocSignalMemory): %s\n", hipGetErrorString(se));
    if (se != hipSuccess) { (void)hipGetLastError(); CK(hipMalloc((void**)&flag, 8)); }
    CK(hipMemset(flag, 0, 8));
    hipStream_t mainq, side;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&mainq, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo));
    std::vector<hipEvent_t> ev(R);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1, t2;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
    unsigned base = 0;
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 5; ++mode) {
            if (mode >= 2 && !can) continue;
            CK(hipMemsetAsync(errors, 0, 4, mainq));
            for (int i = 0; i < R; ++i) CK(hipMemsetAsync(A[i], 0xff, (size_t)n * 4, mainq));
            CK(hipStreamSynchronize(mainq));
            CK(hipEventRecord(t0, mainq));
            for (int i = 0; i < R; ++i) {
                const unsigned seq = base + i + 1;
                hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, mainq, A[i], n, i + 1, (unsigned*)nullptr, 0u);
                if (mode == 1) { CK(hipEventRecord(ev[i], mainq)); CK(hipStreamWaitEvent(side, ev[i], 0)); }
                if (mode == 2) hipLaunchKernelGGL(sig, dim3(1), dim3(64), 0, mainq, flag, seq);
                if (mode == 4) CK(hipStreamWriteValue32(mainq, flag, seq, 0));
                if (mode >= 2) CK(hipStreamWaitValue32(side, flag, seq, hipStreamWaitValueGte, 0xffffffffu));
                if (mode >= 1) hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, side, (const int*)A[i], n, i + 1, errors);
                hipLaunchKernelGGL(produce, dim3(1024), dim3(256), 0, mainq, B, n, i, mode == 3 ? flag : (unsigned*)nullptr, seq);
            }
            CK(hipEventRecord(t1, mainq));
            if (mode >= 2) hipLaunchKernelGGL(sig, dim3(1), dim3(64), 0, mainq, flag, base + R + 1);      // releases every wait, whatever happened
            CK(hipStreamSynchronize(mainq));
            CK(hipStreamSynchronize(side));
            CK(hipEventRecord(t2, mainq));
            CK(hipStreamSynchronize(mainq));
            base += R + 1;
            float m = 0, w = 0;
            CK(hipEventElapsedTime(&m, t0, t1)); CK(hipEventElapsedTime(&w, t0, t2));
            int herr = 0;
            CK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
            printf("rep %d mode %d (%s): main stream %.2f us / iteration, wall with the side work drained %.2f us / iteration, stale words %d\n", rep, mode,
                   mode == 0 ? "no fork, no side work" : mode == 1 ? "hipEventRecord + hipStreamWaitEvent"
                   : mode == 2 ? "one-wave flag kernel + hipStreamWaitValue32" : mode == 3 ? "next kernel stores the flag + hipStreamWaitValue32"
                   : "hipStreamWriteValue32 + hipStreamWaitValue32", m * 1000.f / R, w * 1000.f / R, herr);
        }
    return 0;
}
