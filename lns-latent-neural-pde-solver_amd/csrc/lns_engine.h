// Internal structures of the LNS rollout engine (host side, C++).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/lns.h"
#include "lns_kernels.h"

namespace lns {

// ---- parameter table (= reference state_dict) -------------------------------
struct Param {
    std::string key;
    std::vector<int64_t> shape;
    std::vector<float> host;
    bool is_buffer = false;
    bool is_set = false;
    size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

// conv / linear weight in kernel-native layout [taps][Cin_pad][Cout_pad]
struct ConvPack {
    std::vector<std::string> wkeys;   // concatenated along Cout
    std::vector<std::string> bkeys;   // "" entry => zero bias for that slice
    std::vector<int> couts;
    int cin = 0, cout = 0, k = 1;
    bool has_bias = false;
    int Cin_pad = 0, Cout_pad = 0, kc_log2 = 3;
    size_t w_off = 0, b_off = 0;      // float offsets in the device weight blob
    size_t wb_off = 0; bool has_wb = false;   // bf16x3 / f16x2 slabs
    bool f16 = false; float wscale = 1.0f;    // 3x3: slabs hold the two-term fp16 split of w * wscale
    // the conv follows a 2x nearest upsample (UpSampleBlock, final nn.Upsample): phase slabs of the four-tap form too
    bool up2 = false; size_t wu_off = 0; bool has_wu = false; float wscale_up = 1.0f;
};

enum VecXform { VX_NONE = 0, VX_TRANSPOSE2D = 1, VX_PE_T = 2 };
struct VecPack { std::string key; int xform = VX_NONE; size_t off = 0; size_t count = 0; };

// ---- layer IR (mirrors the reference nn.Sequential entries) ------------------
enum LType { LT_CONV, LT_SWISH, LT_GN, LT_RES, LT_UP2, LT_RESIZE, LT_SA, LT_FA, LT_FOURIER, LT_PROPBLOCK, LT_CONDBLOCK, LT_CONDRES };

struct Layer {
    LType type = LT_CONV;
    std::string name;
    // conv
    int pack = -1, k = 1, stride = 1, dil = 1;
    int pad[4] = {0, 0, 0, 0};   // top, bottom, left, right
    int mode_y = 0, mode_x = 0;
    // group norm
    int groups = 0; float eps = 0; int vg = -1, vb = -1; int C = 0;
    // residual block
    int g1 = -1, b1 = -1, g2 = -1, b2 = -1, conv1 = -1, conv2 = -1, chup = -1, cin = 0, cout = 0;
    // resize
    int outH = 0, outW = 0;
    // self attention
    int heads = 0, dim_head = 0, ln_g = -1, ln_b = -1, pe = -1, pe_len = 0, qkv = -1, proj = -1;
    // factorized attention
    int fa_g = -1, fa_b = -1, inproj = -1, toin = -1, qkx = -1, qky = -1, out1 = -1, out3 = -1;
    int rx[6] = {-1, -1, -1, -1, -1, -1}, ry[6] = {-1, -1, -1, -1, -1, -1};
    std::string invf_x, invf_y;
    int fa_lat = 0, fa_dk = 0;
    // propagator block
    int p_g1 = -1, p_b1 = -1, p_c1 = -1, p_c3 = -1, p_c5 = -1, p_g2 = -1, p_b2 = -1, p_f1 = -1, p_f3 = -1;
    // conditional block extras
    int c_g = -1, c_b = -1, c_conv = -1, blk_index = 0;
    // CondResidualBlock (cond_utils.py:58-128): cr_lin_w / cr_lin_b = cond_emb Linear (vec ids), E = its input width
    int cr_lin_w = -1, cr_lin_b = -1, cr_E = 0;
    // fourier block
    int f_conv = -1, f_w1 = -1, f_w2 = -1, m1 = 0, m2 = 0, f_cond_w = -1, f_cond_b = -1, f_lin = -1;
};

// ---- launch plan ---------------------------------------------------------------
enum Space { SP_NULL = 0, SP_WS = 1, SP_WT = 2, SP_CT = 3, SP_EXT0 = 4 };   // ext slots: 4..11
enum ExtSlot { EX_IN = 0, EX_OUT = 1, EX_PARAM = 2, EX_SS = 3 /* [B][C][2] affine prologue of the input (lns_encode_affine) */, EX_COUNT = 4 };

enum OpType { OP_CONV, OP_GNSTATS, OP_LNPE, OP_ATTN, OP_FAPOOL, OP_FARED, OP_FARED2, OP_FALRK, OP_FALRK2, OP_FASAND, OP_FAGSPLIT, OP_FAFUSED, OP_CONDBASE, OP_CONDBLK,
              OP_APPLY, OP_SPECTRAL, OP_FCOMBINE, OP_VECLIN, OP_TRACE };

struct Op {
    OpType type;
    std::string name;
    int cls = 0;          // timing class
    double flops = 0, bytes = 0;
    // what the launch EXECUTES on the matrix pipe (padding of taps / channels / tiles and the three products of the
    // split-operand scheme included), in FLOP of the instruction's own dtype; `form` names the kernel form and the pipe
    // ("f16x2 ...": v_mfma_f32_32x32x16_f16, "fp32 ...": v_mfma_f32_32x32x2_f32, others: no matrix work booked)
    double mfma_flops = 0;
    const char* form = "";
    int chunk_group = 0;      // > 0: one of several batch-chunked launches of the same layer (same id: same layer)
    int variant = 0;
    // (every argument block is zero-initialised: a field added to one of them -- e.g. the batch-chunk offset b0 -- must never
    //  reach a launch as stack garbage from a site that fills the block member by member)
    ConvArgs conv = {};
    GnStatsArgs gn = {};
    const float* gn_tile_part = nullptr;   // GroupNorm fed by a convolution's per-tile partials (tagged pointer)
    int gn_tiles = 0;
    int gn_count = 0;           // pixels behind every tile partial (GN_TILE_PIXELS, or the plane of a ragged single tile);
    GnTileGeom gn_geom = {0, 0, 0, 0, 0};   // -1: ragged tiles of a larger plane, counts from gn_geom (unequal-count merge)
    LnPeArgs ln = {};
    AttnArgs at = {};
    FaPoolArgs fp = {};
    FaReducerArgs fr = {}, fr2 = {};     // fr2 / fl2: second axis of the merged two-axis launches
    FaLrkArgs fl = {}, fl2 = {};
    FaSandwichArgs fs = {};
    FaGsplitArgs fg = {};      // FABlock with in_proj inside the sandwich (fa_fused.inc): the input split pre-pass
    FaFusedArgs ff = {};       // ... and the fused kernel
    CondBaseArgs cb = {};
    CondBlockArgs ck = {};
    ApplyArgs ap = {};
    SpectralArgs sp = {};
    FourierCombineArgs fc = {};
    VecLinearArgs vl = {};
    // trace
    uint64_t t_ptr = 0; long t_bs = 0; int tC = 0, tH = 0, tW = 0;
};

struct Plan {
    std::vector<Op> ops;
    std::vector<int> consts_i;        // host copy of int constants
    std::vector<float> consts_f;      // host copy of float constants
    void* d_consts = nullptr;         // device: ints then floats
    size_t arena_bytes = 0;
    int B = 0, H = 0, W = 0;
    int kind = 0; long key = 0;       // which cache holds this plan, under which key
    // amax side channel (dynamic activation scale of the split-operand convs): [slots][B] unsigned inside the arena,
    // zeroed by one memset at the start of every run of the plan; slot i belongs to the tensor amax_names[i]
    size_t amax_off = 0, amax_bytes = 0;
    std::vector<std::string> amax_names;
};

// tensor handed in by the caller / the rollout loop: base, batch stride (floats) and, for step-batched launches, the
// two-level form (sample s -> (s % bdiv) * bs + (s / bdiv) * bs2; bdiv 0: plain)
struct ExtT { const void* ptr = nullptr; long bs = 0; long bs2 = 0; int bdiv = 0; };

struct TraceRec { std::string name; int B, C, H, W; std::vector<float> data; };
struct TimeRec { std::string name; double ms = 0; int64_t launches = 0; double flops = 0, bytes = 0, mfma_flops = 0; };

}  // namespace lns

struct lns_engine {
    lns_config cfg;
    std::string err;
    std::vector<lns::Param> params;
    std::map<std::string, int> pindex;
    std::vector<lns::ConvPack> packs;
    std::vector<lns::VecPack> vecs;
    std::vector<lns::Layer> enc, dec, prop;
    int lat_C = 0, lat_H = 0, lat_W = 0;
    // device weights
    float* d_weights = nullptr;
    size_t weights_floats = 0;
    bool finalized = false;
    int device = -1;
    // plans keyed by batch (encode/decode) or (B,H,W) for the propagator
    std::map<long, lns::Plan> enc_plans, dec_plans, prop_plans;
    // propagate / decode overlap
    void* side_stream = nullptr;
    std::vector<hipStream_t> dec_streams;
    std::vector<hipEvent_t> events;
    // scheduling options (lns_set_option; defaults from the LNS_* environment variables of the same meaning)
    int opt_decode_group = 1;      // steps decoded per launch set; 0 = automatic (about 256 samples per launch set)
    int opt_decode_streams = 3;
    int opt_overlap = 1;           // propagator / decode streams; 0 = everything on the caller's stream
    int opt_prop_priority = 0;     // 1: the propagator's side stream is created with the highest priority
    // FABlock2D: in_proj -> sandwich -> to_out run per group of samples whose 512-plane tensor is at most this many MB, so
    // that it is still in the Infinity Cache (256 MB, shared by the streams of the overlapped rollout) when the next kernel
    // reads it; 0 = whole batch per launch.  Scheduling only: the bits do not change.
    int opt_fa_chunk_mb = 0;
    // FABlock2D at 64 x 64 planes, 64 channels: in_proj computed inside the sandwich kernel (fa_fused.inc) instead of written to
    // and read from HBM as a heads * dim_head plane tensor.  A planning rule ("fa_fused" option / LNS_FA_FUSED).
    int opt_fa_fused = 2;          // 0: off, 1: single-buffered kernel, 2: double-buffered kernel (default), 3: the generic kernel
                                   // for the 64 x 64 block as well (tests); the 32 x 32 block (128 channels) uses the generic one for 1..3
    int opt_fa_fused_gpb = 0;      // plane groups (of 16) per block of the fused kernel; 0 = automatic (scheduling only)
    // what the last top-level call ran, for lns_check_finite: (plan kind, plan key, arena offset inside the caller's
    // workspace) -- no pointers into the plan caches or the workspace, which the caller may drop at any time -- plus
    // the workspace and batch of that call.  Cleared whenever plans are dropped.
    struct RanRec { int kind; long key; size_t arena_off; };
    std::vector<RanRec> ran;
    const void* ran_ws = nullptr;
    int ran_B = 0;
    // "track_nonfinite" option: a device word per plan kind that remembers a non-finite amax record of ANY plan run of
    // the last call (one extra 1-block launch per plan run; off by default)
    int opt_track_nonfinite = 0;
    unsigned* d_sticky = nullptr;
    bool sticky_armed = false;
    // diagnostics
    bool trace_on = false;
    std::vector<lns::TraceRec> trace;
    bool timing_on = false;
    std::vector<lns::TimeRec> timing;
    // what an (event, launch, event) triple reports beyond the kernel's own duration: measured once per engine around empty
    // launches (median of 64 pairs, minus ~1 us for the empty kernel itself) and subtracted from every timed launch, so that
    // the per-form times agree with a rocprofv3 kernel trace (without it they read ~4.6 us per launch high)
    double timing_overhead_ms = -1.0;
};
