// Kernels of the latent training rollout's backward pass (lns_train_kernels.hip).  All tensors fp32 NCHW.
#pragma once
#include <hip/hip_runtime.h>

namespace lns {

// OIHW weight -> [tap][Cin_pad][Cout_pad] pack of the fp32 convolution kernels; transpose_flip = 1: the pack of the
// data-gradient convolution (channels swapped, taps mirrored; pads given for THAT convolution: Cin_pad >= Cout, ...)
hipError_t launch_pack_conv_w(const float* w, float* dst, int Cout, int Cin, int k, int Cin_pad, int Cout_pad, int transpose_flip,
                              hipStream_t s);

struct GnTrainArgs {
    const float* x; float* y;          // forward: y = GN(x)
    float* stats;                      // [B][groups][2] (mean, rstd): written by forward, read by backward
    const float* gamma; const float* beta;
    const float* dy; float* dx;        // backward
    const float* add;                  // backward: dx = add + ...  (gradient arriving over the skip connection) or null
    float* part;                       // backward: [B][C][2] per-sample (dgamma, dbeta) partials
    int B, C, HW, groups; float eps;
};
hipError_t launch_gn_train_fwd(const GnTrainArgs& a, hipStream_t s);
hipError_t launch_gn_train_bwd(const GnTrainArgs& a, hipStream_t s);
hipError_t launch_colsum2(const float* part, int B, int C, float* dst_a, float* dst_b, int accumulate, hipStream_t s);

hipError_t launch_gelu_fwd(const float* u, float* y, long n, hipStream_t s);
hipError_t launch_gelu_bwd(const float* dy, const float* u, float* du, long n, hipStream_t s);
hipError_t launch_add(const float* a, const float* b, float* y, long n, hipStream_t s);
hipError_t launch_add_rows(const float* a, long sa, const float* b, long sb, float* y, long sy, int rows, long n, hipStream_t s);
hipError_t launch_bias_grad(const float* dy, int B, int C, int HW, float* db, int accumulate, hipStream_t s);

struct WgradArgs {
    const float* dy;                   // [B][Cout][H][W]
    const float* x;                    // [B][Cin][H][W]
    const int* rowmap; const int* colmap;   // padded coordinate -> source row / column or -1 (H + (k-1) dil entries)
    float* dw;                         // [Cout][Cin][k][k] (OIHW, the parameter's own layout)
    int B, Cin, Cout, H, W, k, dil, accumulate;
};
hipError_t launch_conv_wgrad(const WgradArgs& a, hipStream_t s);

// conditional propagator: per-channel reductions / modulation of fields, per-sample vector network (one block each)
hipError_t launch_chan_dot(const float* a, const float* b2, float* out, int BC, int HW, int accumulate, hipStream_t s);
hipError_t launch_chan_scale(const float* x, const float* m, const float* add, float* y, int BC, int HW, hipStream_t s);
hipError_t launch_vec_fourier(const float* param, float* out, int B, int E, hipStream_t s);
hipError_t launch_vec_linear_fwd(const float* x, const float* W, const float* bias, float* y, int B, int I, int O, hipStream_t s);
hipError_t launch_vec_linear_bwd(const float* dy, const float* x, const float* W, float* dx, int dx_acc, float* dW, float* db, int p_acc,
                                 int B, int I, int O, hipStream_t s);
hipError_t launch_vec_gelu(const float* u, const float* dy, float* out, int n, hipStream_t s);
hipError_t launch_vec_gn(const float* x, const float* gamma, const float* beta, float* y, float* stats, const float* dy, float* dx,
                         float* dgamma, float* dbeta, int p_acc, int B, int D, float eps, hipStream_t s);

}  // namespace lns
