// Backward pass of the latent training rollout (reference: train_stage2_ns2d.py:126-141 `forward` feeding
// loss.backward() at :215): the kernels that the inference path does not have.  All fp32.
//   * weight-gradient of a stride-1 "same" convolution (3x3 with dilation and per-axis zero / circular padding, 1x1)
//     on the fp32 matrix instruction, K = batch x pixels, deterministic (one block owns an output tile and walks K in a
//     fixed order; gradients of the BPTT steps are accumulated in launch order);
//   * data-gradient = the forward convolution kernels with flipped / transposed weight packs (pack kernel below);
//   * GroupNorm forward that keeps (mean, rstd) and GroupNorm backward; exact-erf GELU forward / backward;
//   * bias gradient and the reduction of per-sample partials.
// Hand-written HIP for gfx950; no CPU fallback.
#include "lns_train_kernels.h"

namespace lns {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CTRL>
__device__ __forceinline__ float tdpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float t_wave_sum(float v) {           // fixed order; the same value in every lane
    v += tdpp<0xB1>(v);
    v += tdpp<0x4E>(v);
    v += tdpp<0x141>(v);
    v += tdpp<0x140>(v);
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float t_block_sum(float v, float* red) {   // blockDim.x == 256
    v = t_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------------------------------------
// OIHW weight -> the fp32 kernels' pack [tap][Cin_pad][Cout_pad].  transpose_flip: the pack of the DATA-GRADIENT
// convolution (input channels = Cout, output channels = Cin, taps mirrored): dX = conv(dY, W^T flipped).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_conv_w_kernel(const float* w, float* dst, int Cout, int Cin, int k, int Cin_pad,
                                                          int Cout_pad, int transpose_flip) {
    const int taps = k * k;
    const long n = (long)taps * Cin_pad * Cout_pad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int o = (int)(i % Cout_pad);
        const int c = (int)((i / Cout_pad) % Cin_pad);
        const int t = (int)(i / ((long)Cout_pad * Cin_pad));
        float v = 0.0f;
        if (!transpose_flip) {
            if (o < Cout && c < Cin) v = w[((long)o * Cin + c) * taps + t];
        } else {                                   // pack input channel c = a forward OUTPUT channel, pack output o = a forward INPUT channel
            if (o < Cin && c < Cout) v = w[((long)c * Cin + o) * taps + (taps - 1 - t)];
        }
        dst[i] = v;
    }
}
hipError_t launch_pack_conv_w(const float* w, float* dst, int Cout, int Cin, int k, int Cin_pad, int Cout_pad, int transpose_flip,
                              hipStream_t s) {
    const long n = (long)k * k * Cin_pad * Cout_pad;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(pack_conv_w_kernel, dim3(blocks), dim3(256), 0, s, w, dst, Cout, Cin, k, Cin_pad, Cout_pad, transpose_flip);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm forward, training form: y = (x - mean) rstd gamma + beta and (mean, rstd) per (sample, group).
// One block per (group, sample); two-pass statistics (biased variance), reference: nn.GroupNorm.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_train_fwd_kernel(GnTrainArgs a) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y, cg = a.C / a.groups;
    const long n = (long)cg * a.HW;
    const long base = ((long)b * a.C + (long)g * cg) * a.HW;
    const float* x = a.x + base;
    float s = 0.0f;
    for (long i = threadIdx.x; i < n; i += 256) s += x[i];
    const float mean = t_block_sum(s, red) / (float)n;
    float q = 0.0f;
    for (long i = threadIdx.x; i < n; i += 256) { const float d = x[i] - mean; q += d * d; }
    const float var = t_block_sum(q, red) / (float)n;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (threadIdx.x == 0) { a.stats[((long)b * a.groups + g) * 2] = mean; a.stats[((long)b * a.groups + g) * 2 + 1] = rstd; }
    float* y = a.y + base;
    for (long i = threadIdx.x; i < n; i += 256) {
        const int c = g * cg + (int)(i / a.HW);
        y[i] = (x[i] - mean) * rstd * a.gamma[c] + a.beta[c];
    }
}
hipError_t launch_gn_train_fwd(const GnTrainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(gn_train_fwd_kernel, dim3(a.groups, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}

// GroupNorm backward.  dx = rstd (g dy - mean_g(g dy) - xhat mean_g(g dy xhat)) [+ add]; per-sample partials of
// dgamma[c] = sum dy xhat, dbeta[c] = sum dy go to part [B][C][2] (reduced over the batch by colsum2_kernel).
__global__ __launch_bounds__(256) void gn_train_bwd_kernel(GnTrainArgs a) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y, cg = a.C / a.groups;
    const long n = (long)cg * a.HW;
    const long base = ((long)b * a.C + (long)g * cg) * a.HW;
    const float* x = a.x + base;
    const float* dy = a.dy + base;
    const float mean = a.stats[((long)b * a.groups + g) * 2], rstd = a.stats[((long)b * a.groups + g) * 2 + 1];
    float s1 = 0.0f, s2 = 0.0f;
    for (long i = threadIdx.x; i < n; i += 256) {
        const int c = g * cg + (int)(i / a.HW);
        const float xh = (x[i] - mean) * rstd, gd = dy[i] * a.gamma[c];
        s1 += gd; s2 += gd * xh;
    }
    const float m1 = t_block_sum(s1, red) / (float)n;
    const float m2 = t_block_sum(s2, red) / (float)n;
    float* dx = a.dx + base;
    const float* add = a.add ? a.add + base : nullptr;
    for (long i = threadIdx.x; i < n; i += 256) {
        const int c = g * cg + (int)(i / a.HW);
        const float xh = (x[i] - mean) * rstd, gd = dy[i] * a.gamma[c];
        const float v = rstd * (gd - m1 - xh * m2);
        dx[i] = add ? add[i] + v : v;
    }
    // per-channel partials of this sample: one wave per channel at a time (fixed order inside the wave reduction)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int cl = wave; cl < cg; cl += 4) {
        const float* xc = x + (long)cl * a.HW;
        const float* dc = dy + (long)cl * a.HW;
        float sg = 0.0f, sb = 0.0f;
        for (int p = lane; p < a.HW; p += 64) { const float d = dc[p]; sg += d * (xc[p] - mean) * rstd; sb += d; }
        sg = t_wave_sum(sg); sb = t_wave_sum(sb);
        if (lane == 0) {
            float* pp = a.part + ((long)b * a.C + g * cg + cl) * 2;
            pp[0] = sg; pp[1] = sb;
        }
    }
}
hipError_t launch_gn_train_bwd(const GnTrainArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(gn_train_bwd_kernel, dim3(a.groups, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
// dst_a[c] (+)= sum_b part[b][c][0], dst_b[c] (+)= sum_b part[b][c][1], b in ascending order
__global__ __launch_bounds__(256) void colsum2_kernel(const float* part, int B, int C, float* dst_a, float* dst_b, int accumulate) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float sa = 0.0f, sb = 0.0f;
    for (int b = 0; b < B; ++b) { sa += part[((long)b * C + c) * 2]; sb += part[((long)b * C + c) * 2 + 1]; }
    dst_a[c] = accumulate ? dst_a[c] + sa : sa;
    dst_b[c] = accumulate ? dst_b[c] + sb : sb;
}
hipError_t launch_colsum2(const float* part, int B, int C, float* dst_a, float* dst_b, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(colsum2_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, B, C, dst_a, dst_b, accumulate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// exact-erf GELU (nn.GELU default) forward and backward
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* u, float* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = u[i];
        y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* dy, const float* u, float* du, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = u[i];
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
        du[i] = dy[i] * (cdf + v * pdf);
    }
}
static int ew_blocks(long n) { const long b = (n + 255) / 256; return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048); }
hipError_t launch_gelu_fwd(const float* u, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, u, y, n);
    return hipGetLastError();
}
hipError_t launch_gelu_bwd(const float* dy, const float* u, float* du, long n, hipStream_t s) {
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dy, u, du, n);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* y, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a[i] + b[i];
}
hipError_t launch_add(const float* a, const float* b, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, a, b, y, n);
    return hipGetLastError();
}

// y[r][i] = a[r][i] (+ b[r][i]); rows `sa` / `sb` / `sy` floats apart (strided views of [B][T][...] tensors)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* a, long sa, const float* b, long sb, float* y, long sy, long n) {
    const int r = blockIdx.y;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        y[r * sy + i] = b ? a[r * sa + i] + b[r * sb + i] : a[r * sa + i];
}
hipError_t launch_add_rows(const float* a, long sa, const float* b, long sb, float* y, long sy, int rows, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_rows_kernel, dim3(ew_blocks(n), rows), dim3(256), 0, s, a, sa, b, sb, y, sy, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// bias gradient: db[c] (+)= sum_{b, p} dy[b][c][p]; one block per channel, fixed order
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* dy, int B, int C, int HW, float* db, int accumulate) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float* p = dy + ((long)b * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = t_block_sum(s, red);
    if (threadIdx.x == 0) db[c] = accumulate ? db[c] + s : s;
}
hipError_t launch_bias_grad(const float* dy, int B, int C, int HW, float* db, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, s, dy, B, C, HW, db, accumulate);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a stride-1 "same" convolution:
//   dW[co][ci][ty][tx] (+)= sum_b sum_{y,x} dY[b][co][y][x] * Xpad[b][ci][y + ty*dil][x + tx*dil]
// where Xpad is the padded view described by rowmap / colmap (padded coordinate -> source row / column or -1), the same
// maps the forward kernels gather through.  Block = 256 threads = 4 waves = a 64 (co) x 64 (ci) tile of ONE tap; K =
// batch x pixels walked in chunks of 64 pixels staged through LDS as [pixel][channel] (so that the fp32 MFMA's A / B
// fragments -- lane = (channel, k) -- are conflict-free reads); v_mfma_f32_32x32x2_f32, fp32 accumulate, one fixed order.
// ---------------------------------------------------------------------------------------------------------------------
#define WG_PX 64
#define WG_LD 65
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
    __shared__ float sdy[WG_PX * WG_LD];      // [pixel][co]
    __shared__ float sx[WG_PX * WG_LD];       // [pixel][ci]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64, tap = blockIdx.z;
    const int ty = tap / a.k, tx = tap - ty * a.k;
    const int HW = a.H * a.W;
    const int wm = wave >> 1, wn = wave & 1;                    // wave tile: co 32*wm.., ci 32*wn..
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int l31 = lane & 31, kh = lane >> 5;
    for (int b = 0; b < a.B; ++b) {
        const float* dyb = a.dy + (long)b * a.Cout * HW;
        const float* xb = a.x + (long)b * a.Cin * HW;
        for (int p0 = 0; p0 < HW; p0 += WG_PX) {
            __syncthreads();
            // stage: thread -> (channel = tid / 4 .. , 16 pixels); coalesced along pixels
            for (int e = tid; e < 64 * WG_PX; e += 256) {
                const int ch = e / WG_PX, px = e - ch * WG_PX;
                const int p = p0 + px;
                float vd = 0.0f, vx = 0.0f;
                if (p < HW) {
                    if (co0 + ch < a.Cout) vd = dyb[(long)(co0 + ch) * HW + p];
                    if (ci0 + ch < a.Cin) {
                        const int y = p / a.W, x = p - y * a.W;
                        const int sy = a.rowmap[y + ty * a.dil], sxx = a.colmap[x + tx * a.dil];
                        if (sy >= 0 && sxx >= 0) vx = xb[(long)(ci0 + ch) * HW + (long)sy * a.W + sxx];
                    }
                }
                sdy[px * WG_LD + ch] = vd;
                sx[px * WG_LD + ch] = vx;
            }
            __syncthreads();
#pragma unroll 8
            for (int k2 = 0; k2 < WG_PX; k2 += 2) {
                const float av = sdy[(k2 + kh) * WG_LD + wm * 32 + l31];
                const float bv = sx[(k2 + kh) * WG_LD + wn * 32 + l31];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
    }
    // D[row = co][col = ci]: lane holds column l31, rows (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, ci = ci0 + wn * 32 + l31;
        if (co < a.Cout && ci < a.Cin) {
            float* d = a.dw + ((long)co * a.Cin + ci) * (a.k * a.k) + tap;
            *d = a.accumulate ? *d + acc[r] : acc[r];
        }
    }
}
hipError_t launch_conv_wgrad(const WgradArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((a.Cout + 63) / 64, (a.Cin + 63) / 64, a.k * a.k), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Conditional propagator (train_stage2_twophase_conditional.py:25-121): per-channel reductions / modulation of the
// fields, and the small per-sample vector network (fourier embedding -> MLP -> per-block Linear / GroupNorm / 1x1 convs on
// [B, D, 1, 1]).  The vector kernels are one 256-thread block each: B <= 1024 samples x <= 512 features.
// ---------------------------------------------------------------------------------------------------------------------
// out[b][c] (+)= sum_p a[b][c][p] * (b2 ? b2[b][c][p] : 1): one wave per (b, c), fixed order
__global__ __launch_bounds__(256) void chan_dot_kernel(const float* a, const float* b2, float* out, int BC, int HW, int accumulate) {
    const int bc = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (bc >= BC) return;
    const float* pa = a + (long)bc * HW;
    const float* pb = b2 ? b2 + (long)bc * HW : nullptr;
    float s = 0.0f;
    for (int p = lane; p < HW; p += 64) s += pb ? pa[p] * pb[p] : pa[p];
    s = t_wave_sum(s);
    if (lane == 0) out[bc] = accumulate ? out[bc] + s : s;
}
hipError_t launch_chan_dot(const float* a, const float* b2, float* out, int BC, int HW, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(chan_dot_kernel, dim3((BC + 3) / 4), dim3(256), 0, s, a, b2, out, BC, HW, accumulate);
    return hipGetLastError();
}
// y[b][c][p] = (add ? add[..] : 0) + x[b][c][p] * (1 + m[b][c])
__global__ __launch_bounds__(256) void chan_scale_kernel(const float* x, const float* m, const float* add, float* y, int BC, int HW) {
    const long n = (long)BC * HW;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i] * (1.0f + m[i / HW]);
        y[i] = add ? add[i] + v : v;
    }
}
hipError_t launch_chan_scale(const float* x, const float* m, const float* add, float* y, int BC, int HW, hipStream_t s) {
    hipLaunchKernelGGL(chan_scale_kernel, dim3(ew_blocks((long)BC * HW)), dim3(256), 0, s, x, m, add, y, BC, HW);
    return hipGetLastError();
}

// fourier_embedding(param, E) (modules/cond_utils.py:19-38): [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) i / half)
__global__ __launch_bounds__(256) void vec_fourier_kernel(const float* param, float* out, int B, int E) {
    const int half = E / 2;
    for (int i = threadIdx.x; i < B * E; i += 256) {
        const int b = i / E, j = i - b * E;
        float v = 0.0f;
        if (j < 2 * half) {
            const int k = j < half ? j : j - half;
            const float f = expf(-9.210340371976184f * (float)k / (float)half);
            const float arg = param[b] * f;
            v = j < half ? cosf(arg) : sinf(arg);
        }
        out[i] = v;
    }
}
// y[b][o] = sum_i x[b][i] W[o][i] + bias[o]   (nn.Linear / a 1x1 conv on [B, I, 1, 1]; W row-major [O][I])
__global__ __launch_bounds__(256) void vec_linear_fwd_kernel(const float* x, const float* W, const float* bias, float* y, int B, int I, int O) {
    for (int e = threadIdx.x; e < B * O; e += 256) {
        const int b = e / O, o = e - b * O;
        float s = bias ? bias[o] : 0.0f;
        for (int i = 0; i < I; ++i) s = fmaf(x[b * I + i], W[o * I + i], s);
        y[e] = s;
    }
}
// dx[b][i] (+)= sum_o dy[b][o] W[o][i] ; dW[o][i] (+)= sum_b dy[b][o] x[b][i] ; db[o] (+)= sum_b dy[b][o]
__global__ __launch_bounds__(256) void vec_linear_bwd_kernel(const float* dy, const float* x, const float* W, float* dx, int dx_acc,
                                                             float* dW, float* db, int p_acc, int B, int I, int O) {
    if (dx)
        for (int e = threadIdx.x; e < B * I; e += 256) {
            const int b = e / I, i = e - b * I;
            float s = 0.0f;
            for (int o = 0; o < O; ++o) s = fmaf(dy[b * O + o], W[o * I + i], s);
            dx[e] = dx_acc ? dx[e] + s : s;
        }
    for (int e = threadIdx.x; e < O * I; e += 256) {
        const int o = e / I, i = e - o * I;
        float s = 0.0f;
        for (int b = 0; b < B; ++b) s = fmaf(dy[b * O + o], x[b * I + i], s);
        dW[e] = p_acc ? dW[e] + s : s;
    }
    if (db)
        for (int o = threadIdx.x; o < O; o += 256) {
            float s = 0.0f;
            for (int b = 0; b < B; ++b) s += dy[b * O + o];
            db[o] = p_acc ? db[o] + s : s;
        }
}
__global__ __launch_bounds__(256) void vec_gelu_kernel(const float* u, const float* dy, float* out, int n) {   // dy null: forward
    for (int i = threadIdx.x; i < n; i += 256) {
        const float v = u[i];
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
        out[i] = dy ? dy[i] * (cdf + v * 0.39894228040143267794f * expf(-0.5f * v * v)) : v * cdf;
    }
}
// GroupNorm(1, D) of a [B, D, 1, 1] tensor = normalisation over the D features of a sample.  fwd: y, stats [B][2];
// bwd: dx, dgamma (+)=, dbeta (+)=
__global__ __launch_bounds__(256) void vec_gn_kernel(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                                                     const float* dy, float* dx, float* dgamma, float* dbeta, int p_acc, int B, int D, float eps) {
    __shared__ float sm[1024 * 2];
    for (int b = threadIdx.x; b < B; b += 256) {
        const float* xb = x + (long)b * D;
        float s = 0.0f;
        for (int i = 0; i < D; ++i) s += xb[i];
        const float mean = s / (float)D;
        float q = 0.0f;
        for (int i = 0; i < D; ++i) { const float d = xb[i] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(q / (float)D + eps);
        sm[2 * b] = mean; sm[2 * b + 1] = rstd;
        if (stats) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
        if (!dy) {
            for (int i = 0; i < D; ++i) y[(long)b * D + i] = (xb[i] - mean) * rstd * gamma[i] + beta[i];
        } else {
            float s1 = 0.0f, s2 = 0.0f;
            for (int i = 0; i < D; ++i) { const float xh = (xb[i] - mean) * rstd, gd = dy[(long)b * D + i] * gamma[i]; s1 += gd; s2 += gd * xh; }
            s1 /= (float)D; s2 /= (float)D;
            for (int i = 0; i < D; ++i) {
                const float xh = (xb[i] - mean) * rstd, gd = dy[(long)b * D + i] * gamma[i];
                dx[(long)b * D + i] = rstd * (gd - s1 - xh * s2);
            }
        }
    }
    if (dy) {
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += 256) {
            float sg = 0.0f, sb = 0.0f;
            for (int b = 0; b < B; ++b) { const float d = dy[(long)b * D + i]; sg += d * (x[(long)b * D + i] - sm[2 * b]) * sm[2 * b + 1]; sb += d; }
            dgamma[i] = p_acc ? dgamma[i] + sg : sg;
            dbeta[i] = p_acc ? dbeta[i] + sb : sb;
        }
    }
}
hipError_t launch_vec_fourier(const float* param, float* out, int B, int E, hipStream_t s) {
    hipLaunchKernelGGL(vec_fourier_kernel, dim3(1), dim3(256), 0, s, param, out, B, E);
    return hipGetLastError();
}
hipError_t launch_vec_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int I, int O, hipStream_t s) {
    hipLaunchKernelGGL(vec_linear_fwd_kernel, dim3(1), dim3(256), 0, s, x, W, bias, y, B, I, O);
    return hipGetLastError();
}
hipError_t launch_vec_linear_bwd(const float* dy, const float* x, const float* W, float* dx, int dx_acc, float* dW, float* db, int p_acc,
                                 int B, int I, int O, hipStream_t s) {
    hipLaunchKernelGGL(vec_linear_bwd_kernel, dim3(1), dim3(256), 0, s, dy, x, W, dx, dx_acc, dW, db, p_acc, B, I, O);
    return hipGetLastError();
}
hipError_t launch_vec_gelu(const float* u, const float* dy, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(vec_gelu_kernel, dim3(1), dim3(256), 0, s, u, dy, out, n);
    return hipGetLastError();
}
hipError_t launch_vec_gn(const float* x, const float* gamma, const float* beta, float* y, float* stats, const float* dy, float* dx,
                         float* dgamma, float* dbeta, int p_acc, int B, int D, float eps, hipStream_t s) {
    if (B > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vec_gn_kernel, dim3(1), dim3(256), 0, s, x, gamma, beta, y, stats, dy, dx, dgamma, dbeta, p_acc, B, D, eps);
    return hipGetLastError();
}

}  // namespace lns
