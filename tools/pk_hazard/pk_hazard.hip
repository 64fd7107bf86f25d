// Standalone reproducer of the "co-residency" observation of round 1: a packed-fp32 VALU instruction with
// op_sel operand selection (v_pk_fma_f32: the GroupNorm x*scale+shift of the fp32 1x1 kernel, as the SLP vectoriser
// formed it) returned a wrong operand for a quarter of a wave while waves of an MFMA-dense kernel shared its SIMD.
// TEST INFRASTRUCTURE (tests/test_gpu_parity.py::test_packed_fp32_corun); compiled WITHOUT -packed-fp32-ops, on purpose.
//
//   victim   : every thread loads 4 consecutive floats, reads a (scale, shift) pair from LDS (ds_read_b64) and forms
//              x*scale+shift twice: with two v_pk_fma_f32 ... op_sel:[0,0,1] op_sel_hi:[1,0,1] and with four v_fma_f32;
//              any bitwise difference between the two is counted (same wave, same operands, same moment).
//   aggressor: ds_read_b128 + v_mfma_f32_32x32x16_bf16 loop, 4 waves per block, sized to co-reside with the victim.
//   pk_run() : `launches` victim launches on one stream while the aggressor loops on another; returns the mismatch
//              counts of the packed and of the scalar-only control variant (which re-computes with v_fma twice).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sfma(float a, float b, float c) {
    float d;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ x, const float2* __restrict__ ss, int C, int HW,
                                                     float* __restrict__ y, unsigned long long* bad) {
    extern __shared__ float2 ssl[];
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) ssl[c] = ss[(long)blockIdx.y * C + c];
    __syncthreads();
    const long p4 = (long)blockIdx.x * 256 + tid;
    if (p4 * 4 >= HW) return;
    unsigned long long nbad = 0;
    for (int c = 0; c < C; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(x + ((long)blockIdx.y * C + c) * HW + p4 * 4);
        const float2 st = ssl[c];
        float r0, r1, r2, r3;
        if (PACKED) {
            // exactly the pair hipcc emitted in the fp32 1x1 kernel: the SECOND instruction's destination is the
            // (scale, shift) register pair it also reads through op_sel
            //     v_pk_fma_f32 v[6:7], v[82:83], v[2:3], v[2:3] op_sel:[0,0,1] op_sel_hi:[1,0,1]
            //     v_pk_fma_f32 v[2:3], v[84:85], v[2:3], v[2:3] op_sel:[0,0,1] op_sel_hi:[1,0,1]
            f32x2 a = {v.x, v.y}, b = {v.z, v.w}, s = {st.x, st.y}, o0;
            asm volatile("v_pk_fma_f32 %0, %2, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                         "v_pk_fma_f32 %1, %3, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                         : "=&v"(o0), "+v"(s) : "v"(a), "v"(b));
            r0 = o0[0]; r1 = o0[1]; r2 = s[0]; r3 = s[1];
        } else {
            r0 = sfma(v.x, st.x, st.y); r1 = sfma(v.y, st.x, st.y); r2 = sfma(v.z, st.x, st.y); r3 = sfma(v.w, st.x, st.y);
        }
        // reference: four scalar v_fma_f32 (inline asm, so the SLP vectoriser cannot pack them)
        const float q0 = sfma(v.x, st.x, st.y), q1 = sfma(v.y, st.x, st.y), q2 = sfma(v.z, st.x, st.y), q3 = sfma(v.w, st.x, st.y);
        nbad += (__float_as_uint(r0) != __float_as_uint(q0)) + (__float_as_uint(r1) != __float_as_uint(q1)) +
                (__float_as_uint(r2) != __float_as_uint(q2)) + (__float_as_uint(r3) != __float_as_uint(q3));
        *reinterpret_cast<float4*>(y + ((long)blockIdx.y * C + c) * HW + p4 * 4) = make_float4(r0, r1, r2, r3);
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256, 1) void aggressor_kernel(float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16 * 1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * 1024; i += 256) lds[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 22));
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
    const char* base = reinterpret_cast<const char*>(lds) + (tid & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(base + ((it + j) & 15) * 1024);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(base + ((it + j + 5) & 15) * 1024 + 16384);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 123.456f) sink[tid] = s;          // keeps the loop alive, never true in practice
}

extern "C" int pk_run(int B, int C, int HW, int rounds, int launches, int with_aggressor, unsigned long long* bad_packed,
                      unsigned long long* bad_scalar) {
    float *x = nullptr, *y = nullptr, *sink = nullptr;
    float2* ss = nullptr;
    unsigned long long* dbad = nullptr;
    const size_t n = (size_t)B * C * HW;
    if (hipMalloc(&x, n * 4) || hipMalloc(&y, n * 4) || hipMalloc(&ss, (size_t)B * C * 8) || hipMalloc(&sink, 1024) ||
        hipMalloc(&dbad, 16)) return -1;
    float* hx = (float*)malloc(n * 4);
    float2* hs = (float2*)malloc((size_t)B * C * 8);
    uint64_t seed = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17; return (float)((seed >> 40) & 0xFFFF) / 32768.0f - 1.0f; };
    for (size_t i = 0; i < n; ++i) hx[i] = rnd();
    for (size_t i = 0; i < (size_t)B * C; ++i) hs[i] = make_float2(1.0f + 0.1f * rnd(), 0.1f * rnd());
    hipMemcpy(x, hx, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(ss, hs, (size_t)B * C * 8, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 16);
    free(hx); free(hs);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    const dim3 grid((HW / 4 + 255) / 256, B);
    for (int r = 0; r < rounds; ++r) {
        for (int l = 0; l < launches; ++l) {
            if (with_aggressor) hipLaunchKernelGGL(aggressor_kernel, dim3(512), dim3(256), 0, sb, sink, 600);
            hipLaunchKernelGGL((victim_kernel<true>), grid, dim3(256), C * 8, sa, x, ss, C, HW, y, dbad);
            hipLaunchKernelGGL((victim_kernel<false>), grid, dim3(256), C * 8, sa, x, ss, C, HW, y, dbad + 1);
        }
        hipStreamSynchronize(sa);
        hipStreamSynchronize(sb);
    }
    unsigned long long h[2] = {0, 0};
    hipMemcpy(h, dbad, 16, hipMemcpyDeviceToHost);
    if (bad_packed) *bad_packed = h[0];
    if (bad_scalar) *bad_scalar = h[1];
    hipStreamDestroy(sa); hipStreamDestroy(sb);
    hipFree(x); hipFree(y); hipFree(ss); hipFree(sink); hipFree(dbad);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
