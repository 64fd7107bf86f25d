// How many kernels of different streams does the GPU run at once?  N streams, each gets one small long-running kernel
// (32 blocks of 64 threads spinning ~200 us): wall time ~ 200 us x ceil(N / concurrency).
//   hipcc --offload-arch=gfx950 -O3 -o streams streams.hip && ./streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1;
}
int main() {
    int* sink; hipMalloc(&sink, 64);
    for (int flags = 0; flags < 2; ++flags)
        for (int n = 1; n <= 8; ++n) {
            hipStream_t st[8];
            for (int i = 0; i < n; ++i) {
                if (flags) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking); else hipStreamCreate(&st[i]);
            }
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st[i], 1000, sink);   // warm
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(32), dim3(64), 0, st[i], 20000, sink);   // 200 us at 100 MHz
            hipDeviceSynchronize();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%s streams %d: %.0f us  (concurrency ~ %.1f)\n", flags ? "non-blocking" : "default     ", n, us, n * 200.0 / us);
            for (int i = 0; i < n; ++i) hipStreamDestroy(st[i]);
        }
    return 0;
}
