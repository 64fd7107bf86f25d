// Does wave SPECIALISATION recover the matrix-pipe time the 3x3 kernel's staging instructions cost?
// One 512-thread block per CU (2 waves per SIMD).  Two structures doing the same work per stage:
//   uniform    : all 8 waves run the shipped kernel's per-k-step mix (6 MFMAs + 19 VALU + 3 exp + 7 ds_read_b128 +
//                2 ds_write_b128 + 3 global loads), 5 k-steps per stage, one barrier per stage  (tools/mfma_peak/mfma_mix.hip)
//   specialised: waves 0-3 ("consumers", one per SIMD) issue ONLY fragment reads + MFMAs (12 MFMAs + 8 ds_read_b128 per
//                k-step: a 64-cout x 64-pixel wave tile), waves 4-7 ("producers", one per SIMD) issue ONLY the staging
//                of the whole block: per stage PV VALU + PT transcendental + PW ds_write_b128 + PG global loads (+ PD
//                16-byte LDS-DMA loads for the weight slab), spread evenly; one barrier per stage.
// Reported: TFLOP/s of executed fp16 MFMA work over the whole chip (256 blocks x 2 rounds).
//   hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops -o spec_mix spec_mix.hip && ./spec_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KSTEPS 5

// ---- uniform: every wave does MFMAs and staging --------------------------------------------------------------------
template <int NV, int NT, int NR, int NW, int NG>
__global__ __launch_bounds__(512, 1) void uniform_k(float* out, const float* src, int stages) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 32768 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
    __syncthreads();
    f16x8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = (_Float16)(0.25f * i); b1[i] = (_Float16)(tid * 0.002f); }
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = tid * 0.37f + i;
    uint4 fr[2][6];
    for (int q = 0; q < 6; ++q) { fr[0][q] = make_uint4(q, tid, 1, 2); fr[1][q] = make_uint4(3, q, tid, 4); }
    float g[2][6];
    for (int q = 0; q < 6; ++q) { g[0][q] = 0.f; g[1][q] = 0.f; }
    const char* rbase = lds + (tid & 63) * 16;
    char* wbase = lds + 16384 + tid * 16;
    const float* gp = src + tid;
    auto body = [&](auto cur_tag, int it) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            f16x8 A = __builtin_bit_cast(f16x8, fr[cur][m]);
            acc[m < 2 ? m : 2 + (m & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, m & 2 ? b1 : b0, acc[m < 2 ? m : 2 + (m & 1)], 0, 0, 0);
#pragma unroll
            for (int q = m; q < NR; q += 6) fr[nxt][q % 6] = *reinterpret_cast<const uint4*>(rbase + ((q + it) & 7) * 1024);
#pragma unroll
            for (int q = m; q < NW; q += 6) *reinterpret_cast<uint4*>(wbase + (q & 1) * 8192) = make_uint4(__float_as_uint(v[0]), m, q, it);
#pragma unroll
            for (int q = m; q < NG; q += 6) g[nxt][q % 6] = gp[((it * 6 + q) & 1023) * 512];
#pragma unroll
            for (int q = m; q < NV; q += 6) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = m; q < NT; q += 6) v[(q + 3) & 7] = __builtin_amdgcn_exp2f(v[(q + 3) & 7]);
            if (NG > m) v[m & 7] += g[cur][m];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s = 0; s < stages; ++s) {
        body(std::integral_constant<int, 0>{}, 0);
        body(std::integral_constant<int, 1>{}, 1);
        body(std::integral_constant<int, 0>{}, 2);
        body(std::integral_constant<int, 1>{}, 3);
        body(std::integral_constant<int, 0>{}, 4);
        __syncthreads();
    }
    float s = 0.0f;
    for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + tid] = s;
}

// ---- specialised ------------------------------------------------------------------------------------------------------
// per STAGE: producers PV VALU, PT exp, PW ds_write_b128, PG global dword loads, PD LDS-DMA 16-byte loads;
// consumers KSTEPS x (CM MFMAs + CR ds_read_b128).  PRIO: s_setprio of the consumer waves.
template <int PV, int PT, int PW, int PG, int PD, int CM, int CR, int PRIO>
__global__ __launch_bounds__(512, 1) void spec_k(float* out, const float* src, int stages) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
    __syncthreads();
    if (wave < 4) {
        // -------- consumer
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        constexpr int NACC = CM >= 12 ? 8 : 4;
        f32x16 acc[NACC];
        for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;
        uint4 fr[2][CR];
        for (int q = 0; q < CR; ++q) { fr[0][q] = make_uint4(q, tid, 1, 2); fr[1][q] = make_uint4(3, q, tid, 4); }
        const char* rbase = lds + lane * 16 + wave * 4096;
        auto kstep = [&](auto cur_tag, int it) __attribute__((always_inline)) {
            constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
#pragma unroll
            for (int q = 0; q < CR; ++q) fr[nxt][q] = *reinterpret_cast<const uint4*>(rbase + ((q + it) & 3) * 1024 + (q >> 2) * 16384);
#pragma unroll
            for (int m = 0; m < CM; ++m) {
                const f16x8 A = __builtin_bit_cast(f16x8, fr[cur][m % (CR / 2)]);
                const f16x8 B = __builtin_bit_cast(f16x8, fr[cur][CR / 2 + (m % (CR / 2))]);
                acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[m % NACC], 0, 0, 0);
            }
            // reads first (they belong to the NEXT k-step), then the MFMAs back to back
            __builtin_amdgcn_sched_group_barrier(0x100, CR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CM, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int s = 0; s < stages; ++s) {
            kstep(std::integral_constant<int, 0>{}, 0);
            kstep(std::integral_constant<int, 1>{}, 1);
            kstep(std::integral_constant<int, 0>{}, 2);
            kstep(std::integral_constant<int, 1>{}, 3);
            kstep(std::integral_constant<int, 0>{}, 4);
            __builtin_amdgcn_s_barrier();
        }
        float s = 0.0f;
        for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {
        // -------- producer
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = tid * 0.37f + i;
        float g[PG > 0 ? PG : 1];
        for (int q = 0; q < (PG > 0 ? PG : 1); ++q) g[q] = 0.f;
        char* wbase = lds + 32768 + (tid - 256) * 16;
        const float* gp = src + tid;
        const char* dsrc = reinterpret_cast<const char*>(src) + (tid - 256) * 16;
        for (int s = 0; s < stages; ++s) {
            // consume last stage's loads, request this stage's
#pragma unroll
            for (int q = 0; q < PG; ++q) v[q & 7] += g[q];
#pragma unroll
            for (int q = 0; q < PG; ++q) g[q] = gp[((s * PG + q) & 1023) * 512];
#pragma unroll
            for (int q = 0; q < PD; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dsrc + (((s * PD + q) & 63)) * 4096),
                                                 (__attribute__((address_space(3))) void*)(lds + 49152 + (wave - 4) * 1024 + (q & 3) * 4096), 16, 0, 0);
#pragma unroll
            for (int q = 0; q < PV; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], 1.0001f, 0.5f);
#pragma unroll
            for (int q = 0; q < PT; ++q) v[(q + 3) & 7] = __builtin_amdgcn_exp2f(v[(q + 3) & 7]);
#pragma unroll
            for (int q = 0; q < PW; ++q) *reinterpret_cast<uint4*>(wbase + (q & 3) * 4096) = make_uint4(__float_as_uint(v[q & 7]), q, s, 1);
            __builtin_amdgcn_s_barrier();
        }
        float s = 0.0f;
        for (int i = 0; i < 8; ++i) s += v[i];
        out[blockIdx.x * 512 + tid] = s;
    }
}

template <typename K>
static float time_kernel(K kern, int blocks, size_t lds, float* out, const float* src, int stages) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, out, src, stages);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return best;
}

int main() {
    float *out, *src;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&src, 1024 * 512 * 4 + 65536 * 8); hipMemset(src, 0, 1024 * 512 * 4 + 65536 * 8);
    const int stages = 2000, blocks = 512;
    const size_t lds = 100 * 1024;
    const double mf = 32.0 * 32 * 16 * 2;
#define UNI(NV, NT, NR, NW, NG, name) { float ms = time_kernel(uniform_k<NV, NT, NR, NW, NG>, blocks, lds, out, src, stages); \
        printf("uniform  %-58s %7.1f TFLOP/s\n", name, (double)blocks * 8 * stages * KSTEPS * 6 * mf / ms / 1e9); }
#define SPEC(PV, PT, PW, PG, PD, CM, CR, PRIO, name) { float ms = time_kernel(spec_k<PV, PT, PW, PG, PD, CM, CR, PRIO>, blocks, lds, out, src, stages); \
        printf("special. %-58s %7.1f TFLOP/s\n", name, (double)blocks * 4 * stages * KSTEPS * CM * mf / ms / 1e9); }
    UNI(0, 0, 0, 0, 0, "MFMA only (8 waves)")
    UNI(0, 0, 7, 0, 0, "MFMA + 7 ds_read per k-step")
    UNI(19, 3, 7, 2, 3, "shipped mix: 19 VALU 3 exp 7 rd 2 wr 3 gl per k-step")
    UNI(19, 3, 7, 1, 2, "  weights by DMA-like saving: 1 wr 2 gl")
    SPEC(0, 0, 0, 0, 0, 12, 8, 0, "consumers alone: 12 MFMA + 8 rd per k-step, idle producers")
    SPEC(0, 0, 0, 0, 0, 6, 6, 0, "consumers alone:  6 MFMA + 6 rd per k-step")
    // the staging of a 64-cout x 256-pixel block per 8-channel stage, on 4 producer waves (per wave):
    //   patch 324 units / 256 lanes -> 2 units = 16 values: 16 loads, ~11 VALU + 2 transcendental per value, 4 ds_write_b128
    //   weights 1152 16-byte units / 256 lanes -> 5 (register path: +5 loads +5 writes; DMA path: 5 LDS-DMA)
    SPEC(176, 32, 9, 21, 0, 12, 8, 0, "full staging, weights through registers")
    SPEC(176, 32, 4, 16, 5, 12, 8, 0, "full staging, weights by LDS-DMA")
    SPEC(176, 32, 4, 16, 5, 12, 8, 1, "  same, consumers at s_setprio 1")
    SPEC(176, 32, 4, 16, 5, 12, 8, 3, "  same, consumers at s_setprio 3")
    SPEC(176, 0, 4, 16, 5, 12, 8, 0, "  without the transcendentals")
    SPEC(0, 0, 4, 16, 5, 12, 8, 0, "  without VALU / exp")
    SPEC(176, 32, 0, 0, 0, 12, 8, 0, "  VALU / exp only")
    SPEC(352, 64, 8, 32, 5, 12, 8, 0, "double patch staging (dilated / halo-heavy tiles)")
    SPEC(88, 16, 2, 8, 5, 12, 8, 0, "half patch staging (128-cout tile shares the patch)")
    return 0;
}
