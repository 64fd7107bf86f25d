// Sustained rate of v_mfma_f32_32x32x16_f16 on this box: what the split-operand kernels can reach at best.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
// Modes: accumulators per wave (4 = the conv kernel's hi0 hi1 lo0 lo1 lo0 lo1 pattern, 8 = all independent), waves per
// SIMD (blocks of 256 threads, LDS padding limits blocks per CU), VALU instructions interleaved per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PATTERN, int NVALU>
__global__ __launch_bounds__(256, 1) void mfma_loop(float* out, int iters, long long* clk) {
    extern __shared__ char pad[];
    f16x8 a0, a1, b0, b1;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(threadIdx.x * 0.001f + i); a1[i] = (_Float16)(i * 0.5f); b0[i] = (_Float16)(0.25f * i); b1[i] = (_Float16)(threadIdx.x * 0.002f); }
    f32x16 acc[8];
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.37f + i;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#define MF(K, A, B) acc[K] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[K], 0, 0, 0);
#define VV for (int q = 0; q < NVALU; ++q) { v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f); }
        if (PATTERN == 4) {          // the conv kernel's k-step: hi0 hi1 lo0 lo1 lo0 lo1
            MF(0, a0, b0) VV MF(1, a1, b0) VV MF(2, a0, b1) VV MF(3, a1, b1) VV MF(2, a1, b0) VV MF(3, a0, b1) VV
        } else {                     // eight independent accumulators (6 of them per iteration, same count)
            MF(0, a0, b0) VV MF(1, a1, b0) VV MF(2, a0, b1) VV MF(3, a1, b1) VV MF(4, a1, b0) VV MF(5, a0, b1) VV
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.0f;
    for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int PATTERN, int NVALU>
static void run(const char* name, int waves_per_simd, int iters) {
    float* out; long long* clk;
    const int blocks = 256 * waves_per_simd * 4;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, 16);
    const size_t lds = waves_per_simd >= 4 ? 0 : (160 * 1024) / waves_per_simd - 1024;    // blocks per CU = waves per SIMD
    hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_loop<PATTERN, NVALU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<PATTERN, NVALU>), dim3(blocks), dim3(256), lds, 0, out, iters, clk);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flop = (double)blocks * 4 * iters * 6 * 32.0 * 32 * 16 * 2;
    printf("%-34s waves/SIMD %d  valu/mfma %d : %8.1f TFLOP/s  (%.3f ms; shader clk %.2f GHz; %.1f cycles per MFMA per SIMD)\n", name,
           waves_per_simd, NVALU, flop / best / 1e9, best, (double)h[0] / ((double)h[1] / 100e6) / 1e9,
           (double)h[0] / ((double)iters * 6 * waves_per_simd));
    hipFree(out); hipFree(clk);
}

int main() {
    const int it = 20000;
    run<4, 0>("conv pattern (4 acc)", 1, it);
    run<4, 0>("conv pattern (4 acc)", 2, it);
    run<8, 0>("independent (6 acc)", 1, it);
    run<8, 0>("independent (6 acc)", 2, it);
    run<4, 4>("conv pattern + 4 VALU", 1, it);
    run<4, 4>("conv pattern + 4 VALU", 2, it);
    run<4, 6>("conv pattern + 6 VALU", 2, it);
    run<4, 8>("conv pattern + 8 VALU", 1, it);
    run<4, 8>("conv pattern + 8 VALU", 2, it);
    run<4, 12>("conv pattern + 12 VALU", 2, it);
    run<4, 16>("conv pattern + 16 VALU", 2, it);
    return 0;
}
