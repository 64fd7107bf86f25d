// Gap between DEPENDENT kernel launches on one stream: plain launches vs the same chain captured into a hipGraph.
// A kernel of 64 blocks spins ~K microseconds; chains of 25 launches, 128 chains (the shape of config 4's propagator).
//   hipcc --offload-arch=gfx950 -O3 -o launch_gap launch_gap.hip && ./launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(float* p, int ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) p[blockIdx.x] += 1.0f;      // a real dependency between consecutive launches
}
int main() {
    float* p; hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int chain = 25, reps = 128;
    for (int ticks : {0, 500, 1500}) {                // 0 / 5 / 15 us of kernel
        auto run_plain = [&]() { for (int r = 0; r < reps; ++r) for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, p, ticks); };
        run_plain(); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        run_plain(); hipStreamSynchronize(s);
        const double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, p, ticks);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        const double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        const int n = chain * reps;
        printf("kernel %4.1f us: plain %.2f us per launch (gap %.2f), graph %.2f us per launch (gap %.2f)\n", ticks * 0.01,
               plain / n, plain / n - ticks * 0.01, graph / n, graph / n - ticks * 0.01);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
