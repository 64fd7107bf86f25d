// Which instruction class, interleaved with v_mfma_f32_32x32x16_f16, costs matrix-pipe time?  One wave per SIMD
// (256-thread blocks, one per CU) and two; per k-step of SIX MFMAs: NV plain VALU, NT transcendental, NR ds_read_b128
// (consumed one k-step later), NW ds_write_b128, NG global_load_dword (consumed one k-step later), spread over the six gaps.
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops -o mfma_mix mfma_mix.hip && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NT, int NR, int NW, int NG, int ORDER = 0>
__global__ __launch_bounds__(256, 1) void mix(float* out, const float* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
    __syncthreads();
    f16x8 a0, a1, b0, b1;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(tid * 0.001f + i); a1[i] = (_Float16)(i * 0.5f); b0[i] = (_Float16)(0.25f * i); b1[i] = (_Float16)(tid * 0.002f); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = tid * 0.37f + i;
    uint4 fr[2][6];                       // fragments read one iteration ahead
    for (int q = 0; q < 6; ++q) { fr[0][q] = make_uint4(q, tid, 1, 2); fr[1][q] = make_uint4(3, q, tid, 4); }
    float g[2][6];
    for (int q = 0; q < 6; ++q) { g[0][q] = 0.f; g[1][q] = 0.f; }
    const char* rbase = lds + (tid & 63) * 16;
    char* wbase = lds + 8192 + tid * 16;
    const float* gp = src + tid;
    auto body = [&](auto cur_tag, int it) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        auto mfma = [&](int m) __attribute__((always_inline)) {
            f16x8 A = m & 1 ? a1 : a0, B = m & 2 ? b1 : b0;
            if (NR > 0) { A = __builtin_bit_cast(f16x8, fr[cur][m]); }
            acc[m < 2 ? m : 2 + (m & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[m < 2 ? m : 2 + (m & 1)], 0, 0, 0);
        };
        auto others = [&](int m) __attribute__((always_inline)) {
            // instruction q of a class goes to gap (q mod 6)
#pragma unroll
            for (int q = m; q < NR; q += 6) fr[nxt][q % 6] = *reinterpret_cast<const uint4*>(rbase + ((q + it) & 7) * 1024);
#pragma unroll
            for (int q = m; q < NW; q += 6) *reinterpret_cast<uint4*>(wbase + (q & 1) * 4096) = make_uint4(__float_as_uint(v[0]), m, q, it);
#pragma unroll
            for (int q = m; q < NG; q += 6) g[nxt][q % 6] = gp[((it * 6 + q) & 1023) * 256];
#pragma unroll
            for (int q = m; q < NV; q += 6) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = m; q < NT; q += 6) v[(q + 3) & 7] = __builtin_amdgcn_exp2f(v[(q + 3) & 7]);
            if (NG > m) v[m & 7] += g[cur][m];
        };
        if (ORDER == 0) {
#pragma unroll
            for (int m = 0; m < 6; ++m) { mfma(m); others(m); __builtin_amdgcn_sched_barrier(0); }
        } else {
#pragma unroll
            for (int m = 0; m < 6; ++m) mfma(m);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 6; ++m) others(m);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it < iters; it += 2) {
        body(std::integral_constant<int, 0>{}, it);
        body(std::integral_constant<int, 1>{}, it + 1);
    }
    float s = 0.0f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NV, int NT, int NR, int NW, int NG, int ORDER = 0>
static void run(const char* name, int wps, int iters, float* out, const float* src) {
    const int blocks = 256 * wps * 2;
    const size_t lds = wps == 1 ? 100 * 1024 : 64 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(mix<NV, NT, NR, NW, NG, ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mix<NV, NT, NR, NW, NG, ORDER>), dim3(blocks), dim3(256), lds, 0, out, src, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double flop = (double)blocks * 4 * iters * 6 * 32.0 * 32 * 16 * 2;
    printf("%-44s waves/SIMD %d : %7.1f TFLOP/s\n", name, wps, flop / best / 1e9);
}

int main() {
    float *out, *src;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&src, 1024 * 256 * 4 + 4096); hipMemset(src, 0, 1024 * 256 * 4 + 4096);
    const int it = 8000;
    for (int wps = 1; wps <= 2; ++wps) {
        run<0, 0, 0, 0, 0>("MFMA only", wps, it, out, src);
        run<19, 3, 7, 2, 3>("conv3 k-step mix: 19 VALU 3 exp 7 rd 2 wr 3 gl", wps, it, out, src);
        run<19, 3, 7, 0, 3>("  without the ds_writes", wps, it, out, src);
        run<19, 3, 7, 2, 0>("  without the global loads", wps, it, out, src);
        run<19, 3, 0, 2, 3>("  without the ds_reads", wps, it, out, src);
        run<0, 0, 7, 2, 3>("  without VALU / exp", wps, it, out, src);
        run<19, 3, 7, 0, 0>("  VALU + exp + reads only", wps, it, out, src);
        run<10, 2, 7, 1, 1>("  halved staging: 10 VALU 2 exp 7 rd 1 wr 1 gl", wps, it, out, src);
        run<19, 3, 7, 2, 3, 1>("conv3 mix, MFMAs back to back then the rest", wps, it, out, src);
        run<19, 3, 4, 2, 3>("  4 reads instead of 7", wps, it, out, src);
    }
    return 0;
}
