// What does a fork point cost on the main queue?  Chain of K short kernels on one stream with, between consecutive kernels:
//   (a) nothing   (b) hipEventRecord + a second stream waiting on it   (c) hipStreamWriteValue32 + hipStreamWaitValue32 on the second stream
// Reports the chain time per kernel (HIP events around the chain).  Build: hipcc --offload-arch=gfx950 -O2 fork_cost.hip -o fork_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin_kernel(float* p, int iters) {
    float v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 123.456f) p[0] = v;
}
__global__ void tiny_kernel(float* p) { if (p[0] == 123.456f) p[1] = 1.f; }
int main() {
    hipStream_t main_s, side;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&main_s, hipStreamNonBlocking, hi);
    hipStreamCreateWithPriority(&side, hipStreamNonBlocking, lo);
    float* buf; hipMalloc(&buf, 1 << 20); hipMemset(buf, 0, 1 << 20);
    unsigned* flag = nullptr;
    bool have_sig = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) { hipMalloc((void**)&flag, 64); }
    hipMemset(flag, 0, 64);
    const int K = 40;
    std::vector<hipEvent_t> ev(K);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    unsigned seq = 0;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(t0, main_s);
            for (int k = 0; k < K; ++k) {
                hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(256), 0, main_s, buf, 2000);
                if (mode == 1) { hipEventRecord(ev[k], main_s); hipStreamWaitEvent(side, ev[k], 0); hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, buf); }
                if (mode == 2) { ++seq; hipStreamWriteValue32(main_s, flag, seq, 0); hipStreamWaitValue32(side, flag, seq, hipStreamWaitValueGte, 0xffffffffu);
                                 hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, side, buf); }
                if (mode == 3) { hipEventRecord(ev[k], main_s); }      // record only, nobody waits
            }
            hipEventRecord(t1, main_s);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, t0, t1);
            if (rep == 2) printf("mode %d (%s): %.2f us per kernel, signal memory %d, last error %s\n", mode,
                                 mode == 0 ? "plain chain" : mode == 1 ? "event fork" : mode == 2 ? "write/wait value fork" : "event record only",
                                 ms * 1000.f / K, (int)have_sig, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
