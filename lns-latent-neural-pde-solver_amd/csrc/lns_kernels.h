// Kernel argument blocks and host-side launchers of the gfx950 HIP kernels.
// All tensors fp32 NCHW.  See DESIGN.md for the roofline of each kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lns {

enum Act { ACT_NONE = 0, ACT_SWISH = 1, ACT_GELU = 2, ACT_RELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5 };

// ---------------------------------------------------------------------------
// fused implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32)
//   y = act_out(conv(act_in(x*scale+shift)) + bias + badd) + residual
// ---------------------------------------------------------------------------
struct ConvArgs {
    const float* x;        // [B, Cin, Hin, Win], batch stride x_bs floats
    long x_bs;
    int Cin, Hin, Win;
    const float* w;        // packed [taps][Cin_pad][Cout_pad]
    const float* bias;     // [Cout] or null
    const float* ss;       // [B][Cin][2] (scale, shift) or null
    int act_in, act_out;
    const int* rowmap;     // padded/virtual row  -> source row or -1
    const int* colmap;     // padded/virtual col  -> source col or -1
    float* y;              // [B, Cout, Hout, Wout], batch stride y_bs
    long y_bs;
    // two-level batch addressing of y (step-batched decode: launch sample s = step j * y_bdiv + trajectory b writes
    // out[b][t0 + j]): y + (s % y_bdiv) * y_bs + (s / y_bdiv) * y_bs2 when y_bdiv > 0
    int y_bdiv; long y_bs2;
    int Cout, Hout, Wout;
    const float* res;      // residual, same shape as y (batch stride res_bs) or null
    long res_bs;
    const float* badd;     // [B][Cout] broadcast add or null
    int ks, stride, dil;
    int Cin_pad, Cout_pad;
    int kc_log2;           // channels per LDS stage = 1<<kc_log2
    int tiles_x, tiles_y, cout_tiles;
    int bw_log2;           // pixel tile width = 1<<bw_log2, height = TN>>bw_log2
    int PH, PW;            // staged patch extent
    unsigned ph_magic;     // ceil(2^32 / PH), 0 when PH == 1
    int vec4;              // 1x1 only: 16-byte patch loads (HW % 4 == 0, 16-byte aligned bases)
    // optional fused second 1x1 conv (Cout -> Cout, Cout == tile height 64): applied after act_out,
    // before the residual add.  w2: packed [Cout][Cout2_pad] slab of the 1x1 weight, bias2 [Cout].
    const float* w2; const float* bias2; int Cout2_pad;
    float w2scale;         // split-operand kernels: power-of-two scale applied to w2 before its fp16 split
    // bf16x3 path (3x3, stride 1): weights pre-split into 3 bf16 terms,
    // layout [cout tile][stage of 8 ch][split 3][tap 9][64 cout][8 ch]
    const void* wb;
    float unscale;         // accumulator scale of the bf16x3 / f16x2 epilogue: 1, or 1 / (16 * weight scale) for the fp16 form
    // 1: `wb` holds the PHASE slabs of a 3x3 convolution applied to the 2x nearest-upsampled input (convu_pack_weight):
    // x is the SOURCE tensor, tiles / patch / maps are those of a plain pad-1 3x3 convolution of the source, and
    // Hout = 2 Hin, Wout = 2 Win; every block computes 128 source pixels of one of the four output phases (f16x2 kernel)
    int up2;               // (2: the resident-weights form, conv3_up2r.inc, experimental; 3: the quad-phase form, conv3_up2q.inc --
                           //  one block per source tile stages the patch once and walks the four phases, two blocks per CU)
    // GroupNorm folded into this convolution's prologue (split-operand 3x3 / 1x1 kernels): instead of the finished
    // (scale, shift) table `ss`, the per-(sample, 128-pixel tile, channel) (mean, M2) partials the PRODUCER's epilogue left
    // ([B][gn_tiles][Cin][2], ConvArgs::stat_part of that launch) plus gamma / beta; every block of a sample merges them
    // itself (same code, same order: same bits), which removes the statistics / finalize launch from the dependency chain
    const float* gn_part; const float* gn_gamma; const float* gn_beta; int gn_tiles, gn_groups; float gn_eps;
    // ragged single tile (planes below 128 pixels, e.g. the 7 x 15 latents of the two-phase models): the producer's
    // partial covers stat_count valid pixels (0 = the full 128; -1 = ragged tiles of a larger plane: every block counts its
    // own valid pixels, merged by gn_tile_finalize_kernel with unequal counts), the consumer divides by gn_count (0 = 128);
    // gn_premul [B][Cin] or null: statistics of x * premul, the GroupNorm's per-sample channel multiplier
    int stat_count, gn_count; const float* gn_premul;
    // split-operand 3x3 kernel: source maps computed in the kernel instead of read from rowmap / colmap (one memory latency
    // less before a block's first patch loads).  Set by the planner when the layer has no resize and pads <= the plane:
    // padded position p -> p - map_pad; outside [0, size): wrapped once (map_circ bit) or none; p >= map_ext: none.
    // Index 0 = rows, 1 = columns.  The tables stay valid either way (every other kernel reads them).
    int map_arith, map_circ[2], map_pad[2], map_ext[2];
    int w8;                // CV_F64 launches: 0 = the 8-wave form (conv3_w8.inc) when the launch is small, 1 = always, -1 = never
    int ct_per_block;      // 1x1 bf16x3, input-stationary form: cout tiles walked by one block (0 = streaming form)
    // 3x3 split-operand kernel, 128-pixel tiles that cover the plane exactly: per (sample, pixel tile, channel)
    // (mean, centred second moment) of the stored output, [B][tiles][Cout][2]; the following GroupNorm merges them
    // (launch_gn_tile_finalize) instead of reading the tensor again.  Null: not requested.
    float* stat_part;
    int B;
    // Dynamic activation scale of the split-operand (f16x2) kernels.  The producer of x recorded, per sample, the bit
    // pattern of max |x| (amax_in, [B] unsigned; the maximum of IEEE bit patterns of non-negative floats is the maximum
    // of the values, and a NaN anywhere survives as the largest pattern); when x is produced by a kernel with an
    // analytic bound (LayerNorm, InstanceNorm) amax_in is null and amax_in_const holds the bound.  The kernel derives
    // bound = max_c(|scale_c| amax + |shift_c|) (or amax without a prologue) and stages activations multiplied by the
    // power of two S that puts `bound` in [2^14, 2^15): no fixed input range, fp16 never overflows, and the low term
    // stays normal for every element within 2^-17 of the sample's maximum.
    // Layout of every amax vector: [B][LNS_AMAX_SUB] -- same-address atomics cost ~0.3 us each at the memory side, so a
    // block publishes ONE value into sub-slot (blockIdx.x % LNS_AMAX_SUB) of its sample and the consumer takes the
    // maximum of the 16 words (one 64-byte scalar load).
    const unsigned* amax_in;
    float amax_in_const;
    // 1: amax_in_const already bounds the TRANSFORMED input (a GroupNorm output: |gamma| sqrt(n_group) + |beta|, a
    // function of the layer only), the kernel uses it as is -- no per-sample maximum, no reduction in the prologue
    int bound_final;
    unsigned* amax_out;    // [B] or null: atomic max of the bit patterns of |y| over everything this launch stores
    // Channel-octet-interleaved activations ("OCT8"): a sample is [C/8][H*W][8] fp32 instead of planar [C][H*W] -- the 8
    // channels of one K stage of a pixel are 32 contiguous bytes (two 16-byte loads instead of eight dword gathers), and the
    // four consecutive couts a lane holds of a 32x32 MFMA D tile are one 16-byte store.  A per-tensor planner property of
    // conv <-> conv intermediates (C % 8 == 0; x, z, y and everything the plane-wise kernels read stay NCHW).
    // x_oct: the input is OCT8 (f16x2 3x3 kernels, split-operand 1x1 kernels); y_oct: the output AND the residual are.
    int x_oct, y_oct;
    // batch-chunked launches (FABlock in_proj -> sandwich -> to_out per group of samples, so that the 512-plane tensor of a
    // chunk is still in the 256 MB Infinity Cache when the next kernel reads it): the launch covers samples b0 .. b0 + B - 1
    // of every per-sample array (blockIdx.y counts from b0)
    int b0;
#ifdef LNS_TS
    long long* dbg_ts;     // diagnostic build only: [blocks][8] phase timestamps (100 MHz wall clock) + hardware ids
#endif
};
// power-of-two activation scale from a bound on |x| (host mirror of the device rule; tests)
float convf_scale_for_bound(float bound);

// tile variants: (TM couts x TN pixels) per 256-thread block
enum ConvVariant { CV_L128 = 0, CV_L64 = 1, CV_M128 = 2, CV_M64 = 3, CV_S64 = 4, CV_S32 = 5, CV_COUNT = 6,
                   CV_B64 = 6 /* bf16x3 3x3 kernel */, CV_B1 = 7 /* bf16x3 1x1 kernel */,          // both 64 couts x 128 pixels
                   CV_B32 = 9 /* bf16x3 3x3 kernel, 32 couts x 128 pixels (same bits as CV_B64) */,
                   CV_F64 = 11 /* 3x3 kernel with the two-term fp16 split (f16x2), 64 couts x 128 pixels */,
                   CV_THIN = 12 /* streaming 1x1 projection to <= 4 output channels (VALU, HBM-bound) */,
                   CV_F32 = 13 /* f16x2 3x3 kernel, 32 couts x 128 pixels (same bits as CV_F64) */,
                   CV_F256 = 14 /* f16x2 3x3 kernel, 64 couts x 256 pixels: a wave owns 64 couts x 64 pixels (same bits as CV_F64) */,
                   // f16x2 3x3 kernel with producer / consumer wave specialisation (conv3_pc.inc; same bits as CV_F64):
                   CV_P128 = 15 /* 64 couts x 128 pixels per 512-thread block */, CV_P256 = 16 /* 64 couts x 256 pixels */ };
inline bool cv_is_pc(int v) { return v == CV_P128 || v == CV_P256; }
inline bool cv_is_f16x2_3x3(int v) { return v == CV_F64 || v == CV_F32 || v == CV_F256 || cv_is_pc(v); }
inline bool cv_is_split_3x3(int v) { return v == CV_B64 || v == CV_B32 || cv_is_f16x2_3x3(v); }
struct ConvVariantInfo { int TM, TN; };
ConvVariantInfo conv_variant_info(int v);
size_t conv_lds_bytes(int variant, const ConvArgs& a);
int conv_pick_kc_log2(int ks, int stride, int kc_log2_max);
bool conv_fits(int variant, const ConvArgs& a);
hipError_t launch_conv(int variant, const ConvArgs& a, hipStream_t s);
// 3x3 convolution on bf16 MFMA with every fp32 operand split into three bf16 terms (6 products,
// two accumulators): fp32-level accuracy at 2.7x the fp32-MFMA rate.  Tile 64 couts x 128 pixels.
// f16x2 scheme: activations are multiplied by a per-sample power of two before the fp16 split (ConvArgs::amax_in):
// bound * S lies in [2^14, 2^15), the fp16 range (65504) is never reached and the absolute error of an element's
// two-term representation is <= max(2^-22 |x|, 2^-39 bound).  S is clamped to [2^-100, 2^60].
#define CONVF_TARGET_EXP 14
#define LNS_AMAX_SUB 16
#define CONVB_SLAB_BYTES 27648           // one (cout tile, stage) weight slab: 3 splits x 9 taps x 64 couts x 8 ch bf16
size_t convb_lds_bytes(const ConvArgs& a, int tile_couts, int splits, int ring);
bool convb_fits(const ConvArgs& a);
hipError_t launch_conv_bf16x3(int variant, const ConvArgs& a, hipStream_t s);
// producer / consumer form (nt = 1: 128-pixel tiles, 2: 256-pixel tiles); one 512-thread block per CU walks several tiles
size_t convpc_lds_bytes(const ConvArgs& a, int nt);
bool convpc_geom_fits(const ConvArgs& a, int nt);     // patch / LDS only (planning time: pointers and strides unknown)
bool convpc_fits(const ConvArgs& a, int nt);
hipError_t launch_conv_pc(int nt, const ConvArgs& a, hipStream_t s);
// host-side packing of one [Cout][Cin][3][3] weight (cout offset co0 inside the pack) into the bf16x3 slab layout
size_t convb_weight_bytes(int Cout, int Cin_pad);
bool convuq_fits(const ConvArgs& a);        // quad-phase form of the upsampling conv (ConvArgs::up2 == 3; a.up2 != 0 on entry)
size_t convuq_lds_bytes(const ConvArgs& a);
bool convur_fits(const ConvArgs& a);        // resident-patch form of the upsampling conv (ConvArgs::up2 == 2)
size_t convur_lds_bytes(const ConvArgs& a);
void convb_pack_weight(void* dst, const float* w, int co0, int cout, int cin, int Cin_pad);
void convf_pack_weight(void* dst, const float* w, int co0, int cout, int cin, int Cin_pad, float wscale);   // f16x2 slabs
size_t convu_weight_bytes(int Cout, int Cin_pad);                                                          // phase slabs (ConvArgs::up2)
void convu_pack_weight(void* dst, const float* w, int co0, int cout, int cin, int Cin_pad, float wscale);
// 1x1 bf16x3 kernel: slabs [cout tile][stage of 32 ch][split 3][octet 4][64 cout][8 ch], Cin_pad % 32 == 0
size_t convb1_lds_bytes(const ConvArgs& a);
bool convb1_fits(const ConvArgs& a);
size_t convb1_weight_bytes(int Cout, int Cin_pad);
void convb1_pack_weight(void* dst, const float* w, int co0, int cout, int cin, int Cin_pad, float wscale);
bool convb1_is_f16();    // the 1x1 kernels use the two-term fp16 split (dynamic activation scale) rather than bf16x3
hipError_t launch_conv1_bf16x3(const ConvArgs& a, hipStream_t s);
bool conv1_thin_fits(const ConvArgs& a);
hipError_t launch_conv1_thin(const ConvArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------
// GroupNorm statistics -> per-(b,c) scale/shift (fused into the consumer conv)
// ---------------------------------------------------------------------------
struct GnStatsArgs {
    const float* x; long x_bs; int C, HW, groups; float eps;
    const float* gamma; const float* beta;   // device, [C] (null => 1 / 0)
    const float* premul;                     // [B][C] or null: stats of x*premul
    float* ss;                               // [B][C][2]
    int B;
};
// part: [B][C][2] scratch enabling the two-stage path (may be null)
bool gn_stats_two_stage(const GnStatsArgs& a);
hipError_t launch_gn_stats(const GnStatsArgs& a, float* part, hipStream_t s);
// GroupNorm scale/shift from the per-tile partials a convolution epilogue left (ConvArgs::stat_part); a.x unused
#define GN_TILE_PIXELS 128
// count > 0: every partial covers that many pixels.  count == 0: ragged tiling (GnTileGeom): tile t covers
// min(BH, H - ty BH) x min(BW, W - tx BW) pixels (3x3 producers) or min(128, HW - 128 t) (1x1 producers); unequal-count merge.
struct GnTileGeom { int tiles_x, bw_log2, H, W, flat; };
hipError_t launch_gn_tile_finalize(const GnStatsArgs& a, const float* tile_part, int tiles, int count, GnTileGeom geom, hipStream_t s);

// LayerNorm over channels of a channel-major token tensor + positional embedding
struct LnPeArgs {
    const float* x; long x_bs; int C, n; float eps;
    const float* gamma; const float* beta;   // [C]
    const float* pe_t;                       // [C][pe_stride] transposed positional table or null
    int pe_stride;
    float* h;                                // [B][C][n]
    int B;
};
hipError_t launch_ln_pe(const LnPeArgs& a, hipStream_t s);

// softmax attention, channel-major qkv [B, 3*heads*D, n] -> o [B, heads*D, n]
struct AttnArgs { const float* qkv; int B, heads, D, n; float scale; float* o;
                  const unsigned* amax_in; };   // [B][LNS_AMAX_SUB] max |qkv| per sample or null: enables the f16x2 form
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);

// FABlock2D pieces ----------------------------------------------------------
struct FaPoolArgs { const float* v; int B, C, H, W; float* mx; float* my; };   // mx [B,H,C], my [B,W,C]
hipError_t launch_fa_pool(const FaPoolArgs& a, hipStream_t s);

struct FaReducerArgs {                 // PoolingReducer on pooled rows [rows, C]
    const float* m; long rows; int n;  // rows = B*n ; output u [B, Out, n] channel-major
    int C, Hid, Out;
    const float* win_t;                // [C][C]   (in-major)
    const float* ln_g; const float* ln_b;
    const float* w1_t;                 // [C][Hid]
    const float* w2_t;                 // [Hid][Out]
    const float* b2;                   // [Out]
    float* u;                          // may be null when the to_qk projection below is fused
    unsigned* amax_out;                // [B] or null: running max |u| per sample (consumer: the to_qk split-operand conv)
    // optional fused LowRankKernel.to_qk (1x1 conv Out -> Mqk on the reducer output): qk [B, Mqk, n]
    const float* wqk_t; int ldqk;      // in-major [Out][ldqk] (the conv's fp32 pack)
    const float* bqk;                  // [Mqk] or null
    int Mqk; float* qk;
};
hipError_t launch_fa_reducer(const FaReducerArgs& a, hipStream_t s);
// both axes (x: rows = B*H, y: rows = B*W) of one FABlock in ONE launch
hipError_t launch_fa_reducer2(const FaReducerArgs& ax, const FaReducerArgs& ay, hipStream_t s);

struct FaLrkArgs {                     // rotary + q k^T
    const float* qk;                   // [B, 2*heads*DK, n]
    int B, heads, DK, n;
    const float* cs;                   // [DK/2][n][2] (cos, sin), frequency-major
    float* kmat;                       // [B, heads, n, n]
};
hipError_t launch_fa_lrk(const FaLrkArgs& a, hipStream_t s);
hipError_t launch_fa_lrk2(const FaLrkArgs& ax, const FaLrkArgs& ay, hipStream_t s);   // both axes in one launch

struct FaSandwichArgs {
    const float* u; const float* kx; const float* ky;
    int B, heads, C, H, W; float eps; int instnorm; float* out;
    const unsigned* amax_u;            // [B][LNS_AMAX_SUB] max |u| per sample (bit patterns) or null: enables the f16x2 form
    int b_rev;                         // f16x2 form: walk the samples in reverse launch order (scheduling only)
    int b0;                            // batch-chunked launch: samples b0 .. b0 + B - 1 (ConvArgs::b0)
};
hipError_t launch_fa_sandwich(const FaSandwichArgs& a, hipStream_t s);
size_t fa_sandwich_lds_bytes(int H, int W);

// FABlock2D with in_proj inside the sandwich (fa_fused.inc): the heads * dim_head plane tensor between in_proj and the
// sandwich is never stored.  64 x 64 planes, 64 input channels, dim_head a multiple of 16.
//   launch_fa_gsplit : G = x * scale + shift (the in_norm GroupNorm) once per sample as two fp16 terms of G * S (S: the power
//                      of two that puts `bound` at 2^14 .. 2^15), [B][H*W/16][Cin/32][hi | lo][4 quarters][16 pixels][8]; records max |G| per sample
//   launch_fa_fused  : per (sample, head, 16 planes): P = W G band by band on v_mfma_f32_16x16x32_f16, Y = Kx P Ky^T,
//                      InstanceNorm, store
struct FaGsplitArgs { const float* x; long x_bs; int Cin, HW; const float* ss; float bound; void* gs; unsigned* amax_out; int B; };
struct FaFusedArgs {
    const void* gs; const unsigned* amax_g; float bound;      // the pre-pass's output, its max |G| record, the bound it scaled by
    const void* wp; float w_inv, wrow_max;                    // fa_fused_pack_weight image, 1 / its scale, max_c sum_k |W[c][k]|
    const float* kx; const float* ky;
    int B, heads, C, Cin, H, W; float eps; int instnorm; float* out; int b_rev;
    int single_buffer;     // 1: fa_fused_kernel (one band image, in_proj and sandwich phases alternate); 0: fa_fused2_kernel (two
                           // swizzled band images: the next band's in_proj runs inside the sandwich, one barrier per band).  Same bits.
    int gpb;               // plane groups (of 16) one block walks: Kx / Ky of the (sample, head) are staged once for them
    long long* dbg_ts;     // -DFAF_TS builds: [blocks][24] phase timestamps of wave 0 (null otherwise)
};
bool fa_fused_fits(int H, int W, int Cin, int dim_head);
size_t fa_fused_gs_bytes(int B, int H, int W, int Cin);
size_t fa_fused_weight_bytes(int planes, int Cin);
void fa_fused_pack_weight(void* dst, const float* w, int planes, int Cin, float wscale);
hipError_t launch_fa_gsplit(const FaGsplitArgs& a, hipStream_t s);
hipError_t launch_fa_fused(const FaFusedArgs& a, hipStream_t s);

// conditional propagator: per-sample embedding MLPs (tiny; step-invariant) ------
struct CondBaseArgs {            // ce = W2 act(W0 fourier_embedding(param) + b0) + b2
    const float* param; int B, E;
    const float* freqs;          // [E/2]
    const float* w0_t; const float* b0; const float* w2_t; const float* b2;   // in-major [E][Hd], [Hd][E]
    float* ce;                   // [B][E]
    int Hd;                      // hidden width (0: E)
    int act;                     // ACT_GELU (conditional propagator) / ACT_SWISH (CondEncoder.embed)
};
hipError_t launch_cond_base(const CondBaseArgs& a, hipStream_t s);
struct CondBlockArgs {           // emb = Wce ce + bce ; mul = 1 + conv1(gelu(conv1(GN1(emb))))
    const float* ce; int B, E, D;
    const float* wce_t; const float* bce;          // [E][D], [D]
    const float* gn_g; const float* gn_b;          // [D]
    const float* c1_t; const float* c1_b;          // [D][D] in-major, [D]
    const float* c3_t; const float* c3_b;
    float* emb; float* mul;                        // [B][D]
};
hipError_t launch_cond_block(const CondBlockArgs& a, hipStream_t s);

// y = act(x * scale[b,c] + shift[b,c])  (materialises a pending GroupNorm + activation)
struct ApplyArgs { const float* x; long x_bs; const float* ss; int act; float* y; int B, C, HW; unsigned* amax_out; };
hipError_t launch_apply(const ApplyArgs& a, hipStream_t s);

// Fourier blocks (opt-in): truncated DFT as dense contractions --------------------
//   xf = DFT_H(DFT_W(x)) restricted to rows {0..m1-1, H-m1..H-1} x cols {0..m2-1}
//   of[b,o] = sum_i xf[b,i] * w{1,2}[i,o] (* emb[b])   ;   y = irfft2(of)
struct SpectralArgs {
    const float* x; long x_bs; int B, Cin, Cout, H, W, m1, m2;
    const float* w1; const float* w2;     // [Cin][Cout][m1][m2][2]
    const float* emb;                     // [B][m1][m2][2(lo,hi)][2(re,im)] conditional scaling or null
    float* t1;                            // [B][max(Cin,Cout)][H][m2][2]
    float* xf;                            // [B][Cin][2*m1][m2][2]
    float* of;                            // [B][Cout][2*m1][m2][2]
    float* y;                             // [B][Cout][H][W]
};
hipError_t launch_spectral(const SpectralArgs& a, hipStream_t s);

// y = [skip +] act(a + b + e[b,c])     (skip null: residual=False; act 0: GELU)
struct FourierCombineArgs { const float* a; const float* b; const float* e; const float* skip; long skip_bs; float* y; long y_bs; int B, C, HW; unsigned* amax_out; int act; };
hipError_t launch_fourier_combine(const FourierCombineArgs& a, hipStream_t s);

// out[b,o] = bias[o] + sum_i in[b,i] * w[i*ldo + o*ldi]   (tiny dense layer on per-sample vectors)
struct VecLinearArgs { const float* in; const float* w; const float* bias; float* out; int B, In, Out, ldi, ldo; };
hipError_t launch_vec_linear(const VecLinearArgs& a, hipStream_t s);

// fused denormalise + relative-L2 metric of a rollout (scratch: B*T*C*2 floats)
hipError_t launch_metric_rel_l2(const float* yhat, const float* y, int B, int T, int C, int HW, float mean, float sd, float eps,
                                float* frame_out, float* seq_out, float* scratch, hipStream_t s);
#define LNS_METRIC_MAX_CH 8
struct MetricChannelSpec {
    float mean[LNS_METRIC_MAX_CH], std[LNS_METRIC_MAX_CH];
    int flags[LNS_METRIC_MAX_CH];      // 1: zero the four wall rows/columns, 2: clamp to [lo, hi]
    float lo, hi;
};
hipError_t launch_metric_rel_l2_ch(const float* yhat, const float* y, int B, int T, int C, int H, int W,
                                   const MetricChannelSpec& spec, float eps, float* frame_out, float* seq_out,
                                   float* scratch, hipStream_t s);

// per-sample max |x| (bit patterns) of a [B, n] tensor with batch stride x_bs into amax [B][LNS_AMAX_SUB] (atomic max)
hipError_t launch_amax(const float* x, long x_bs, long n, int B, unsigned* amax, hipStream_t s);
hipError_t launch_amax_sticky(const unsigned* amax, int n, unsigned* flag, hipStream_t s);

bool build_has_experimental();   // compiled with -DLNS_EXPERIMENTAL: the measured-slower kernel forms (variants 15, 16, 18, 19) exist
hipError_t launch_empty(hipStream_t s);   // one empty block (calibration of the HIP-event timing overhead)
hipError_t init_kernels();   // sets dynamic-LDS attributes; needs a GPU

}  // namespace lns
