/*
 * lns.h -- C ABI of the MI355X-native LNS rollout engine (liblns_hip.so).
 *
 * Drop-in boundary for ONE hot path of BaratiLab/LNS-Latent-Neural-PDE-Solver:
 *     LatentDynamics.predict(x, steps[, param], to_x)      train_stage2_ns2d.py:143-158
 *         = SimpleAutoencoder.encode                          modules/autoencoder2d.py:174-177
 *         -> steps x ( SimpleCNN.forward                      train_stage2_ns2d.py:82-87
 *                      ; SimpleAutoencoder.decode )           modules/autoencoder2d.py:179-182
 * (and the SW / two-phase / conditional variants of the same three functions).
 *
 * The reference has NO FFI/plugin layer of its own (it is 100% Python/PyTorch,
 * SURVEY.md F1), so there is no foreign interface to mirror symbol-for-symbol:
 * each entry point below names the reference Python method it replaces.  The
 * ABI is plain C: opaque handle, plain pointers and sizes, int status codes,
 * no C++/torch types.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all tensors are fp32, NCHW, contiguous unless a batch stride is given;
 *   - `x`, `z`, `y`, `out`, `param`, `workspace` are DEVICE pointers owned by
 *     the caller (e.g. torch `data_ptr()`); weights passed to lns_set_weight
 *     are HOST pointers;
 *   - `stream` is a hipStream_t (void*), e.g. torch.cuda.current_stream().cuda_stream;
 *     calls are asynchronous with respect to the host;
 *   - every function returns 0 on success, a negative LNS_E* code otherwise and
 *     never throws; lns_last_error() gives the message;
 *   - an engine handle is not thread-safe; use one handle per GPU.
 *   - device memory is allocated only by lns_finalize_weights() (packed weights)
 *     and lns_prepare() (per-shape launch plans: index maps / rotary tables);
 *     the run calls use caller-provided workspace only.
 */
#ifndef LNS_H_
#define LNS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNS_ABI_VERSION 2
#define LNS_MAX_STAGES 8
#define LNS_MAX_KEY 160

/* status codes */
#define LNS_OK 0
#define LNS_EINVAL (-1)     /* bad argument / unsupported configuration */
#define LNS_ENOKEY (-2)     /* unknown state_dict key / shape mismatch   */
#define LNS_ESTATE (-3)     /* call order (weights not finalized, ...)   */
#define LNS_ENOMEM (-4)     /* workspace too small / device alloc failed */
#define LNS_EHIP (-5)       /* HIP runtime error                         */
#define LNS_ENONFINITE (-6) /* lns_check_finite: a tensor of the last run holds inf / NaN */

/* autoencoder flavour: which reference file the AE follows */
#define LNS_AE_NONE 0
#define LNS_AE_SQUARE 1         /* modules/autoencoder2d.py                 */
#define LNS_AE_NONSQUARED 2     /* modules/autoencoder2d_nonsquared.py      */
#define LNS_AE_HALF_PERIODIC 3  /* modules/autoencoder2d_half_periodic.py   */

/* propagator flavour */
#define LNS_PROP_NONE 0
#define LNS_PROP_PLAIN 1        /* SimpleCNN, train_stage2_{ns2d,SW,twophase}.py:56-87        */
#define LNS_PROP_CONDITIONAL 2  /* SimpleCNN, train_stage2_twophase_conditional.py:78-121      */

/* per-axis boundary handling of the 'same' convolutions */
#define LNS_PAD_ZEROS 0
#define LNS_PAD_CIRCULAR 1

/* Flat mirror of the YAML keys the reference reads from `args`
 * (modules/autoencoder2d.py:19-27,78-92; train_stage2_ns2d.py:94-104). */
typedef struct lns_config {
    int32_t abi_version;            /* = LNS_ABI_VERSION */
    int32_t ae_kind;                /* LNS_AE_*   */
    int32_t prop_kind;              /* LNS_PROP_* */
    int32_t in_channels;
    int32_t latent_dim;
    int32_t Ly, Lx;                 /* output field size            */
    int32_t res_h, res_w;           /* `resolution` / `resolutions` */
    int32_t latent_resolution;
    int32_t ae_pad_y, ae_pad_x;     /* LNS_PAD_* of AE convs (is_periodic / periodic_direction) */
    int32_t n_encoder_channels;
    int32_t encoder_channels[LNS_MAX_STAGES];
    int32_t encoder_res_blocks;
    int32_t use_attn_enc;
    int32_t n_decoder_channels;
    int32_t decoder_channels[LNS_MAX_STAGES];
    int32_t decoder_res_blocks;
    int32_t n_attn_resolutions;
    int32_t attn_resolutions[LNS_MAX_STAGES];
    int32_t n_fourier_resolutions;
    int32_t fourier_resolutions[LNS_MAX_STAGES];
    int32_t use_fa;
    int32_t final_smoothing;
    int32_t disable_coarse_attn;
    int32_t attn_heads;
    int32_t attn_dim;
    float hw_ratio;                 /* nonsquared / half-periodic AEs */
    int32_t prop_n_block;
    int32_t prop_n_embd;
    int32_t prop_dilation;
    int32_t prop_pad_y, prop_pad_x; /* LNS_PAD_* of the propagator's 3x3 convs */
    int32_t cond_emb_dim;           /* conditional propagator only */
    char ae_prefix[32];             /* state_dict prefix of the AE: "vq_ae." / "ae." / "" */
    char prop_prefix[32];           /* "propagator." / ""                                 */
    /* ABI 2: ConditionalSimpleAutoencoder (modules/autoencoder2d_nonsquared.py:279-305): the encoder is CondEncoder
     * (:71-145, CondResidualBlock modules/cond_utils.py:58-128) and encode takes `param`; LNS_AE_NONSQUARED only */
    int32_t cond_encoder;
    int32_t cond_emb_channels;
} lns_config;

typedef struct lns_engine lns_engine;

/* message of the last failed lns_create() on this thread */
const char* lns_create_error(void);

/* Build the layer program for `cfg` (replaces LatentDynamics.__init__ /
 * SimpleAutoencoder.__init__ / SimpleCNN.__init__: train_stage2_ns2d.py:91-104,
 * modules/autoencoder2d.py:161-167, train_stage2_ns2d.py:57-80).  No GPU needed. */
int lns_create(const lns_config* cfg, lns_engine** out);
void lns_destroy(lns_engine* e);
const char* lns_last_error(const lns_engine* e);

/* Parameter table = the reference's state_dict() keys and shapes. */
int lns_num_params(const lns_engine* e);
int lns_param_info(const lns_engine* e, int index, char* key, int key_capacity,
                   int64_t* shape /* [8] */, int* ndim, int* is_buffer);

/* load_state_dict(): copy one tensor (HOST pointer, fp32, contiguous). */
int lns_set_weight(lns_engine* e, const char* key, const float* host_data,
                   const int64_t* shape, int ndim);
/* Repack all weights into kernel-native layout and upload to HIP device `device`. */
int lns_finalize_weights(lns_engine* e, int device);

/* Scheduling options of the rollout (they change HOW the launch sets are issued, never a bit of the result):
 *   "decode_group"    steps decoded by one launch set at batch B * k (default 1; 0 = automatic: about 256 samples per
 *                     launch set.  Measured on NS2d-128 B=64: k = 4 cuts the serial kernel time 11 % and the launches
 *                     2x but the overlapped rollout is 2 % slower than k = 1, tools/sched_sweep.py)
 *   "decode_streams"  decode streams of the overlapped rollout (1..4)
 *   "overlap"         1: latent chain on a side stream, decodes round-robin on the decode streams; 0: one stream
 *   "prop_priority"   1: the side stream of the latent chain is created with the highest stream priority
 *   "fa_chunk_mb"     FABlock2D: in_proj -> sandwich -> to_out are issued per group of samples whose 512-plane tensor is at
 *                     most this many MB (it then stays in the 256 MB Infinity Cache between the three kernels instead of going
 *                     to HBM three times); 0 = whole batch per launch.  Cached plans are rebuilt.  Default: LNS_FA_CHUNK_MB
 *   "fa_fused_gpb"    plane groups (of 16) one block of the fused FABlock kernel walks; 0 = automatic.  Cached plans are rebuilt.
 *   "track_nonfinite" 1: lns_check_finite also remembers the plan runs whose amax record has been reused since (the
 *                     earlier steps / decode groups of a rollout): one extra one-block launch per plan run (default 0)
 * One option selects an ARITHMETIC FORM (results differ at rounding level, ~2e-7 relative on the decoded field):
 *   "fa_fused"        2 (default; LNS_FA_FUSED): FABlock2D on 64 x 64 planes with 64 channels and on 32 x 32 planes with 128
 *                     channels computes in_proj inside the sandwich kernel (csrc/fa_fused.inc) -- the heads * dim_head plane tensor
 *                     is never stored; 1 / 3: other forms of the 64 x 64 kernel (single-buffered band image / the generic kernel:
 *                     same bits as 2, slower); 0: in_proj as its own 1x1 convolution, then the sandwich.  Cached plans are rebuilt.
 * Defaults come from LNS_DECODE_GROUP / LNS_DECODE_STREAMS / LNS_NO_OVERLAP / LNS_PROP_PRIORITY at lns_create().
 * Changing an option changes the workspace size: call lns_prepare() again. */
int lns_set_option(lns_engine* e, const char* name, long value);

/* Latent shape for the configured field size: z is [B, C, H, W]. */
int lns_latent_shape(const lns_engine* e, int* C, int* H, int* W);

/* Build (and cache) launch plans for batch B; returns the workspace bytes the run
 * calls need for that batch through *workspace_bytes (may be NULL). */
int lns_prepare(lns_engine* e, int B, size_t* workspace_bytes);

/* SimpleAutoencoder.encode: x [B,Cin,Ly,Lx] -> z [B,latent_dim,h,w]. */
int lns_encode(lns_engine* e, const float* x, int B, float* z,
               void* workspace, size_t workspace_bytes, void* stream);
/* ConditionalSimpleAutoencoder.encode(x, param) (cfg.cond_encoder = 1): param [B] device, one value per sample. */
int lns_encode_cond(lns_engine* e, const float* x, const float* param, int B, float* z,
                    void* workspace, size_t workspace_bytes, void* stream);
/* encode(x * scale + shift) with the affine map applied in the first convolution's prologue (device table
 * scale_shift [B][in_channels][2]): the dataset normalisation (u - mean) / (std + eps) of the reference's encode_dataset
 * (dataset/ns2d_fno_stage2_simpleae.py:78-93, dataset/Stage2_SW.py:74-105, dataset/twophase_flow_stage2.py:304-337)
 * without a normalised copy of the frames.  param: NULL, or [B] for a conditional encoder. */
int lns_encode_affine(lns_engine* e, const float* x, const float* scale_shift, const float* param, int B, float* z,
                      void* workspace, size_t workspace_bytes, void* stream);
/* SimpleAutoencoder.decode: z [B,latent_dim,h,w] -> y [B,Cin,Ly,Lx]. */
int lns_decode(lns_engine* e, const float* z, int B, float* y,
               void* workspace, size_t workspace_bytes, void* stream);
/* SimpleCNN.forward: z_in [B,latent_dim,H,W] (+ param [B] or NULL) -> z_out (same shape). */
int lns_propagate(lns_engine* e, const float* z_in, const float* param, int B, int H, int W,
                  float* z_out, void* workspace, size_t workspace_bytes, void* stream);
/* LatentDynamics.predict(x, T[, param], to_x): out is [B,T,Cin,Ly,Lx] if to_x else
 * [B,T,latent_dim,h,w]; latents_out (nullable) additionally receives [B,T,latent_dim,h,w]. */
int lns_rollout(lns_engine* e, const float* x, const float* param, int B, int T, int to_x,
                float* out, float* latents_out, void* workspace, size_t workspace_bytes,
                void* stream);

/* The same loop started from a latent state (chunked rollouts, e.g. to overlap the
 * gather of finished step blocks with the remaining steps): z_in [B,latent_dim,h,w] ->
 * out [B,T,...]; z_last (nullable) receives the latent after step T. */
int lns_rollout_latent(lns_engine* e, const float* z_in, const float* param, int B, int T, int to_x,
                       float* out, float* z_last, void* workspace, size_t workspace_bytes, void* stream);

/* Post-run health check.  Every kernel of a plan records, per sample, the running maximum of |y| of the tensor it
 * produces (the side channel from which the split-operand convolutions derive their activation scale); a NaN or inf
 * anywhere in a tensor survives in it.  This call synchronises `stream`, reads those few KB back from `workspace`
 * (the one the last lns_encode / lns_decode / lns_propagate / lns_rollout* call for batch B used) and returns
 * LNS_ENONFINITE with lns_last_error() naming the first layer (in execution order) and sample whose output was not
 * finite -- or LNS_OK.  `workspace` / B must be those of the last call (LNS_ESTATE otherwise, also after
 * lns_finalize_weights / lns_set_option, which drop the plans the records belong to; LNS_ENOMEM if workspace_bytes
 * does not cover the records).  Coverage: every tensor a layer wrote in the LAST RUN of each plan of that call,
 * the plan outputs (z, y) included; earlier runs of the same plan in that call (steps < T of a rollout) only with the
 * "track_nonfinite" option.  The reference has no equivalent (its fields would silently carry NaN); nothing on
 * the hot path depends on it. */
int lns_check_finite(lns_engine* e, int B, void* workspace, size_t workspace_bytes, void* stream);

/* ---- training rollout of the latent propagator ------------------------------------------------
 * Reference: LatentDynamics.forward, train_stage2_ns2d.py:126-141 (SW / two-phase: same text) -- z_pred[:, t] =
 * propagator(z_pred[:, t-1]) started at z_in -- feeding loss.backward() at train_stage2_ns2d.py:215.  The loss is the
 * caller's (a Python callable in the reference): lns_train_forward returns z_pred [B,T,c,h,w] and keeps a tape of the
 * step's intermediates in `workspace`; lns_train_backward takes dL/dz_pred and writes the gradient of every propagator
 * parameter (BPTT over the T steps, accumulated in a fixed order: deterministic) and, if asked, of z_in.
 * `params` / `grads`: arrays of lns_num_params() DEVICE pointers indexed like lns_param_info (entries of tensors that
 * do not belong to the propagator are ignored and may be null); parameters are read from the device at every call
 * (an optimiser updates them in place between calls) -- lns_set_weight / lns_finalize_weights are not involved.
 * All contractions run on the exact-fp32 matrix instruction.  Plain propagators (NS2d, SW, two-phase) and the
 * conditional one (train_stage2_twophase_conditional.py:25-121; `param` [B], no gradient w.r.t. it).  The same
 * workspace and parameter values must be used for the backward call. */
int lns_train_workspace_bytes(lns_engine* e, int B, int h, int w, int T, size_t* bytes);
int lns_train_forward(lns_engine* e, const float* const* params, const float* z_in, const float* param_or_null,
                      int B, int h, int w, int T, float* z_pred, void* workspace, size_t workspace_bytes, void* stream);
int lns_train_backward(lns_engine* e, const float* const* params, const float* z_in, const float* z_pred,
                       const float* grad_z_pred, int B, int h, int w, int T, float* const* grads, float* grad_z_in,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- diagnostics -------------------------------------------------------- */
/* Layer trace: when enabled the run calls synchronise after every reference
 * module boundary and keep a host copy of its output (tests compare them with
 * the oracle layer by layer).  Slow; never enable in production. */
int lns_trace_enable(lns_engine* e, int on);
int lns_trace_count(const lns_engine* e);
int lns_trace_info(const lns_engine* e, int index, char* name, int name_capacity,
                   int64_t* shape /* [4] */);
int lns_trace_copy(const lns_engine* e, int index, float* host_out);

/* Kernel-level timing: after a run call with timing enabled, per-kernel-class
 * elapsed milliseconds measured with HIP events on the call's stream. */
int lns_timing_enable(lns_engine* e, int on);
int lns_timing_count(const lns_engine* e);
int lns_timing_info(const lns_engine* e, int index, char* name, int name_capacity,
                    double* total_ms, int64_t* launches, double* flops, double* bytes);
/* Records named "class/form" follow the per-class ones: the same launches split by kernel form (nine-tap / four-tap /
 * fp32-MFMA 3x3, streaming / input-stationary / fused 1x1, ...).  mfma_flops: FLOP the record's launches EXECUTED on the
 * matrix pipe -- the three products of the split-operand scheme and every padded tap slot, channel and tile included --
 * in FLOP of the form's instruction ("f16x2 ...": fp16 MFMA, "fp32 MFMA ...": fp32 MFMA); `flops` above stays algorithmic. */
int lns_timing_mfma_flops(const lns_engine* e, int index, double* mfma_flops);
/* (index -1: instead of a record's FLOP, the per-launch overhead in MICROSECONDS that the engine measured for its
 *  (event, launch, event) timing around empty launches and subtracted from every timed launch; -1000 before the first timed run) */

/* Build-time features of this library: "experimental" = compiled with -DLNS_EXPERIMENTAL (the measured-slower kernel
 * forms behind op-level variants 15 / 16 / 18 / 19 exist; the shipped library does not carry them).  1 / 0; -1: unknown name.
 * (No reference counterpart: the reference is pure Python.) */
int lns_build_has(const char* feature);

/* ---- kernel-level entry points (unit tests of the HIP kernels) ---------- */
/* General fused convolution (the implicit-GEMM MFMA kernel):
 *   y = act_out( conv(act_in(x * scale + shift)) + bias + badd ) + residual
 * x [B,Cin,Hin,Win] optionally nearest-resized to (Hv,Wv) before padding;
 * w [Cout,Cin,k,k] HOST pointer; ss [B,Cin,2] device (scale,shift) or NULL;
 * act: 0 none, 1 swish, 2 gelu.  tile_variant <0 = automatic; >= 0: a kernel / tile form (lns_kernels.h ConvVariant) in
 * the low byte, plus tensor layouts: | 0x100 = x is channel-octet-interleaved ("OCT8": [Cin/8][Hin*Win][8] per sample,
 * Cin % 8 == 0), | 0x200 = y and residual are OCT8 ([Cout/8][Hout*Wout][8]) -- the engine's layout of conv <-> conv
 * intermediates.
 * amax_out (device, [B][16] unsigned, zero-initialised by the caller, or NULL): the maximum over the 16 words of
 * sample b is the IEEE bit pattern of max |y[b]| -- the side channel from which the split-operand (f16x2) kernels of a plan derive their per-sample
 * power-of-two activation scale.  The entry point computes the same quantity for x itself before the launch. */
int lns_op_conv2d(const float* x, int B, int Cin, int Hin, int Win, int Hv, int Wv,
                  const float* w_host, const float* bias_host, int Cout, int ksize, int stride,
                  int dilation, int pad_t, int pad_b, int pad_l, int pad_r, int mode_y, int mode_x,
                  const float* ss, int act_in, int act_out, const float* residual,
                  const float* badd, float* y, int tile_variant, void* stream, unsigned* amax_out);
/* Diagnostic: runs two convolutions (GroupNorm+Swish prologue, circular padding, stride 1) repeatedly on two HIP
 * streams so that their workgroups share compute units, and counts output words that differ from what each
 * convolution produces alone.  variant_*: tile variant as in lns_op_conv2d (-1 automatic, 6 = bf16x3 3x3 kernel). */
int lns_op_conv_pair_stress(int B, int H, int W, int cin_a, int cout_a, int ksize_a, int variant_a, int cin_b, int cout_b,
                            int ksize_b, int variant_b, int rounds, int launches, long long* mismatches_a,
                            long long* mismatches_b);

/* GroupNorm statistics -> per-(b,c) (scale,shift) such that norm(x) = x*scale+shift. */
int lns_op_groupnorm_stats(const float* x, int B, int C, int HW, int groups, float eps,
                           const float* gamma_host, const float* beta_host, const float* premul,
                           float* ss, void* stream);
/* softmax attention: qkv [B,3*heads*dim_head,n] channel-major -> o [B,heads*dim_head,n] */
int lns_op_attention(const float* qkv, int B, int heads, int dim_head, int n, float scale,
                     float* o, void* stream);
/* FABlock2D core: for every channel plane P of u [B,heads*C,H,W]:
 *   P <- instance_norm( Kx[b,h] . P . Ky[b,h]^T ), kx [B,heads,H,H], ky [B,heads,W,W] */
int lns_op_fa_sandwich(const float* u, const float* kx, const float* ky, int B, int heads, int C,
                       int H, int W, float eps, int apply_instance_norm, float* out, void* stream);

/* FourierBasicBlock (modules/basics.py:531-583) and CondFourierBasicBlock
 * (modules/fourier_cond.py:84-117) as standalone ops (not reached by any shipped config):
 *   y = [x +] act( irfft2(modes(rfft2(x)) . W{1,2} [* FreqLinear(cond)]) + conv1x1(x) [+ Linear(cond)] )
 * x [B,Cin,H,W], y [B,Cout,H,W] device; all weights HOST pointers in the reference's state_dict layout:
 *   w1,w2 [Cin,Cout,m1,m2,2]; conv_w [Cout,Cin,1,1], conv_b [Cout];
 *   conditional only (cond != NULL, device [B,Cin]): freq_w [Cin,4*m1*m2], freq_b [1,4*m1*m2], lin_w [Cout,Cin], lin_b [Cout].
 * activation: 1 silu, 2 gelu, 3 relu, 4 tanh, 5 sigmoid (ACTIVATION_REGISTRY, basics.py:10-16; the conditional block
 * hard-codes GELU, fourier_cond.py:114); residual != 0 adds x (needs Cin == Cout, basics.py:581-582). */
int lns_op_fourier_block(const float* x, int B, int Cin, int Cout, int H, int W, int m1, int m2, const float* w1_host,
                         const float* w2_host, const float* conv_w_host, const float* conv_b_host,
                         const float* cond, const float* freq_w_host, const float* freq_b_host,
                         const float* lin_w_host, const float* lin_b_host, int activation, int residual,
                         float* y, void* stream);

/* The same block as an object with DEVICE-RESIDENT weights: what a module instance (FourierBasicBlock.forward,
 * modules/basics.py:574-583; CondFourierBasicBlock.forward, modules/fourier_cond.py:106-117) calls on every forward.
 * create uploads the weights once (host pointers as above; freq_w_host == NULL: the unconditional block) onto HIP device
 * `device`; forward only launches (asynchronous on `stream`; the scratch buffer grows when a larger shape arrives);
 * cond must be given exactly when the block is conditional. */
typedef struct lns_fourier_block lns_fourier_block;
int lns_fourier_block_create(int Cin, int Cout, int m1, int m2, const float* w1_host, const float* w2_host,
                             const float* conv_w_host, const float* conv_b_host, const float* freq_w_host,
                             const float* freq_b_host, const float* lin_w_host, const float* lin_b_host, int activation,
                             int residual, int device, lns_fourier_block** out);
int lns_fourier_block_forward(lns_fourier_block* h, const float* x, const float* cond, int B, int H, int W, float* y,
                              void* stream);
void lns_fourier_block_destroy(lns_fourier_block* h);

/* ---- "next row" (SURVEY 8f-2): the step right after the path -------------------------------------------------
 * Fused denormalise + relative-L2 metric of a decoded rollout against the ground truth, one pass over both tensors:
 *   yd = y*std + mean (dataset/ns2d_fno_stage2_simpleae.py:140-149), err = sqrt(sum (yhat_d - y_d)^2 / max(sum y_d^2, eps))
 *   frame-wise over (H,W)   -> frame_out [B,T,C]   (relative_lp_loss(reduce_dim=(3,4)), training_utils.py:9-23,
 *   sequence-wise over (T,H,W) -> seq_out [B,C]     (reduce_dim=(1,3,4)),                train_stage2_ns2d.py:254-257)
 * yhat, y [B,T,C,H,W] device fp32 (normalised); scratch: device, >= B*T*C*2 floats; either output may be NULL. */
int lns_metric_rel_l2(const float* yhat, const float* y, int B, int T, int C, int HW, float mean, float std, float eps,
                      float* frame_out, float* seq_out, float* scratch, void* stream);

/* Same metric with per-channel statistics and the boundary handling of the other datasets' denormalize():
 *   v = x*std[c] + mean[c]                     (dataset/Stage2_SW.py:60-72, per-channel u / v / pres stats)
 *   flags[c] & LNS_METRIC_ZERO_WALLS: first/last row and column set to 0 after the affine map
 *                                              (closed-tank velocities, dataset/twophase_flow_stage2.py:370-383)
 *   flags[c] & LNS_METRIC_CLAMP:      v clamped to [clamp_lo, clamp_hi]   (VOF channel, :388, [0, 1+1e-8])
 * applied to both yhat and y as train_stage2_twophase*.py:251-254 / :287-290 do.  mean/std/flags: HOST arrays of C
 * entries (NULL = 0 / 1 / 0); C <= 8. */
#define LNS_METRIC_ZERO_WALLS 1
#define LNS_METRIC_CLAMP 2
int lns_metric_rel_l2_ch(const float* yhat, const float* y, int B, int T, int C, int H, int W, const float* mean_host,
                         const float* std_host, const int* flags_host, float clamp_lo, float clamp_hi, float eps,
                         float* frame_out, float* seq_out, float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LNS_H_ */
