// Model construction: turns an lns_config into the layer program and the
// parameter table (= the reference's state_dict keys/shapes).  Host only.
//
// Every builder follows the constructor of the reference module it replaces
// (file:line cited per function; paths relative to the upstream tree).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "lns_engine.h"

namespace lns {

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Builder {
    lns_engine* e;
    explicit Builder(lns_engine* e_) : e(e_) {}

    int add_param(const std::string& key, std::vector<int64_t> shape, bool is_buffer = false) {
        if (e->pindex.count(key)) throw std::runtime_error("duplicate parameter key " + key);
        Param p;
        p.key = key;
        p.shape = std::move(shape);
        p.is_buffer = is_buffer;
        e->params.push_back(p);
        e->pindex[key] = (int)e->params.size() - 1;
        return (int)e->params.size() - 1;
    }

    static void size_pack(ConvPack& p) {
        p.kc_log2 = (p.k == 3) ? 3 : 5;
        p.Cin_pad = round_up(p.cin, 1 << p.kc_log2);
        p.Cout_pad = p.cout <= 32 ? 32 : (p.cout <= 64 ? 64 : round_up(p.cout, 128));
    }

    // nn.Conv2d(cin, cout, k) parameters `${pfx}.weight` / `${pfx}.bias`
    int conv(const std::string& pfx, int cin, int cout, int k, bool bias = true) {
        ConvPack p;
        p.cin = cin; p.cout = cout; p.k = k; p.has_bias = bias;
        add_param(pfx + ".weight", {cout, cin, k, k});
        p.wkeys.push_back(pfx + ".weight");
        p.couts.push_back(cout);
        if (bias) { add_param(pfx + ".bias", {cout}); p.bkeys.push_back(pfx + ".bias"); }
        else p.bkeys.push_back("");
        size_pack(p);
        e->packs.push_back(p);
        return (int)e->packs.size() - 1;
    }
    // nn.Linear(in, out): 2-D weight, used through the same 1x1 GEMM path
    void linear_params(const std::string& pfx, int in, int out, bool bias) {
        add_param(pfx + ".weight", {out, in});
        if (bias) add_param(pfx + ".bias", {out});
    }
    int linear_pack(const std::vector<std::string>& pfxs, const std::vector<bool>& biases, int in, int out_each) {
        ConvPack p;
        p.cin = in; p.k = 1; p.cout = 0; p.has_bias = false;
        for (size_t i = 0; i < pfxs.size(); ++i) {
            p.wkeys.push_back(pfxs[i] + ".weight");
            p.bkeys.push_back(biases[i] ? pfxs[i] + ".bias" : "");
            p.couts.push_back(out_each);
            p.cout += out_each;
            p.has_bias = p.has_bias || biases[i];
        }
        size_pack(p);
        e->packs.push_back(p);
        return (int)e->packs.size() - 1;
    }
    int vec(const std::string& key, int xform = VX_NONE) {
        VecPack v;
        v.key = key; v.xform = xform;
        v.count = e->params[e->pindex.at(key)].numel();
        e->vecs.push_back(v);
        return (int)e->vecs.size() - 1;
    }
    int vec_param(const std::string& key, std::vector<int64_t> shape, int xform = VX_NONE) {
        add_param(key, std::move(shape));
        return vec(key, xform);
    }
};

static bool in_list(const int32_t* lst, int n, int v) {
    for (int i = 0; i < n; ++i) if (lst[i] == v) return true;
    return false;
}

// ---------------------------------------------------------------------------
static Layer conv_layer(const std::string& name, int pack, int k, int stride, int dil, int pt, int pb, int pl,
                        int pr, int my, int mx) {
    Layer l;
    l.type = LT_CONV; l.name = name; l.pack = pack; l.k = k; l.stride = stride; l.dil = dil;
    l.pad[0] = pt; l.pad[1] = pb; l.pad[2] = pl; l.pad[3] = pr; l.mode_y = my; l.mode_x = mx;
    return l;
}
static Layer same_conv(Builder& b, const std::string& name, int cin, int cout, int k, int my, int mx, int dil = 1,
                       bool bias = true) {
    const int p = dil * (k - 1) / 2;
    return conv_layer(name, b.conv(name, cin, cout, k, bias), k, 1, dil, p, p, p, p, my, mx);
}
static Layer swish_layer(const std::string& name) { Layer l; l.type = LT_SWISH; l.name = name; return l; }

// basics.GroupNorm wrapper (32 groups, eps 1e-6): modules/basics.py:18-24  -> keys `${pfx}.gn.*`
static Layer gn32_layer(Builder& b, const std::string& pfx, int C) {
    Layer l;
    l.type = LT_GN; l.name = pfx; l.groups = 32; l.eps = 1e-6f; l.C = C;
    l.vg = b.vec_param(pfx + ".gn.weight", {C});
    l.vb = b.vec_param(pfx + ".gn.bias", {C});
    return l;
}
// raw nn.GroupNorm(groups, C) (eps 1e-5)
static Layer gn_raw_layer(Builder& b, const std::string& pfx, int groups, int C) {
    Layer l;
    l.type = LT_GN; l.name = pfx; l.groups = groups; l.eps = 1e-5f; l.C = C;
    l.vg = b.vec_param(pfx + ".weight", {C});
    l.vb = b.vec_param(pfx + ".bias", {C});
    return l;
}

// ResidualBlock (num_dimensions=2): modules/basics.py:245-256,272-276
static Layer res_layer(Builder& b, const std::string& p, int cin, int cout, int my, int mx) {
    Layer l;
    l.type = LT_RES; l.name = p; l.cin = cin; l.cout = cout; l.mode_y = my; l.mode_x = mx;
    l.g1 = b.vec_param(p + ".block.0.gn.weight", {cin});
    l.b1 = b.vec_param(p + ".block.0.gn.bias", {cin});
    l.conv1 = b.conv(p + ".block.2", cin, cout, 3);
    l.g2 = b.vec_param(p + ".block.3.gn.weight", {cout});
    l.b2 = b.vec_param(p + ".block.3.gn.bias", {cout});
    l.conv2 = b.conv(p + ".block.5", cout, cout, 3);
    if (cin != cout) l.chup = b.conv(p + ".channel_up", cin, cout, 1);
    return l;
}
// HalfPeriodicResBlock2d: modules/autoencoder2d_half_periodic.py:77-103
static Layer hp_res_layer(Builder& b, const std::string& p, int cin, int cout, int my, int mx) {
    Layer l;
    l.type = LT_RES; l.name = p; l.cin = cin; l.cout = cout; l.mode_y = my; l.mode_x = mx;
    l.g1 = b.vec_param(p + ".norm_act1.norm_act.0.gn.weight", {cin});
    l.b1 = b.vec_param(p + ".norm_act1.norm_act.0.gn.bias", {cin});
    l.g2 = b.vec_param(p + ".norm_act2.norm_act.0.gn.weight", {cout});
    l.b2 = b.vec_param(p + ".norm_act2.norm_act.0.gn.bias", {cout});
    l.conv1 = b.conv(p + ".conv1", cin, cout, 3);
    l.conv2 = b.conv(p + ".conv2", cout, cout, 3);
    if (cin != cout) l.chup = b.conv(p + ".channel_up", cin, cout, 1);
    return l;
}

// SABlock: modules/basics.py:331-404
static Layer sa_layer(Builder& b, const std::string& p, int dim, int heads, int dim_head, bool use_pe, int block_size) {
    Layer l;
    l.type = LT_SA; l.name = p; l.C = dim; l.heads = heads; l.dim_head = dim_head;
    const int inner = heads * dim_head;
    if (use_pe) { l.pe = b.vec_param(p + ".pe", {1, block_size, dim}, VX_PE_T); l.pe_len = block_size; }
    l.ln_g = b.vec_param(p + ".ln.weight", {dim});
    l.ln_b = b.vec_param(p + ".ln.bias", {dim});
    b.linear_params(p + ".to_q", dim, inner, false);
    b.linear_params(p + ".to_k", dim, inner, false);
    b.linear_params(p + ".to_v", dim, inner, true);
    l.qkv = b.linear_pack({p + ".to_q", p + ".to_k", p + ".to_v"}, {false, false, true}, dim, inner);
    b.linear_params(p + ".proj_out", inner, dim, true);
    l.proj = b.linear_pack({p + ".proj_out"}, {true}, inner, dim);
    return l;
}

// PoolingReducer(in=dim, hidden=dim, out=latent): modules/factorized_attention.py:72-94
static void reducer_params(Builder& b, const std::string& p, int dim, int lat, int* ids) {
    ids[0] = b.vec_param(p + ".to_in.weight", {dim, dim}, VX_TRANSPOSE2D);
    ids[1] = b.vec_param(p + ".out_ffn.0.weight", {dim});
    ids[2] = b.vec_param(p + ".out_ffn.0.bias", {dim});
    ids[3] = b.vec_param(p + ".out_ffn.1.weight", {2 * dim, dim}, VX_TRANSPOSE2D);
    ids[4] = b.vec_param(p + ".out_ffn.3.weight", {lat, 2 * dim}, VX_TRANSPOSE2D);
    ids[5] = b.vec_param(p + ".out_ffn.3.bias", {lat});
}

// FABlock2D(dim, dim_head, latent_dim, heads, dim_out): modules/factorized_attention.py:97-159
static Layer fa_layer(Builder& b, const std::string& p, int dim, int dim_head, int lat, int heads) {
    Layer l;
    l.type = LT_FA; l.name = p; l.C = dim; l.heads = heads; l.dim_head = dim_head; l.fa_lat = lat;
    l.fa_dk = dim_head * 2;   // kernel_multiplier = 2
    l.fa_g = b.vec_param(p + ".in_norm.weight", {dim});
    l.fa_b = b.vec_param(p + ".in_norm.bias", {dim});
    l.inproj = b.conv(p + ".in_proj", dim, heads * dim_head, 1, false);
    l.toin = b.conv(p + ".to_in.0", dim, dim, 1, false);
    reducer_params(b, p + ".to_x.0", dim, lat, l.rx);
    reducer_params(b, p + ".to_y.1", dim, lat, l.ry);
    b.linear_params(p + ".low_rank_kernel_x.to_qk", lat, l.fa_dk * heads * 2, false);
    l.qkx = b.linear_pack({p + ".low_rank_kernel_x.to_qk"}, {false}, lat, l.fa_dk * heads * 2);
    l.invf_x = p + ".low_rank_kernel_x.pos_emb.inv_freq";
    b.add_param(l.invf_x, {l.fa_dk / 2}, true);
    b.linear_params(p + ".low_rank_kernel_y.to_qk", lat, l.fa_dk * heads * 2, false);
    l.qky = b.linear_pack({p + ".low_rank_kernel_y.to_qk"}, {false}, lat, l.fa_dk * heads * 2);
    l.invf_y = p + ".low_rank_kernel_y.pos_emb.inv_freq";
    b.add_param(l.invf_y, {l.fa_dk / 2}, true);
    l.out1 = b.conv(p + ".to_out.1", heads * dim_head, dim, 1, false);
    l.out3 = b.conv(p + ".to_out.3", dim, dim, 1, false);
    return l;
}

// FourierBasicBlock(in, out, modes): modules/basics.py:531-583 (+ SpectralConv2d :99-149)
static Layer fourier_layer(Builder& b, const std::string& p, int cin, int cout, int m1, int m2) {
    Layer l;
    l.type = LT_FOURIER; l.name = p; l.cin = cin; l.cout = cout; l.m1 = m1; l.m2 = m2;
    l.f_w1 = b.vec_param(p + ".fourier.weights1", {cin, cout, m1, m2, 2});
    l.f_w2 = b.vec_param(p + ".fourier.weights2", {cin, cout, m1, m2, 2});
    l.f_conv = b.conv(p + ".conv", cin, cout, 1);
    return l;
}

static Layer attn_layer(Builder& b, const lns_config& c, const std::string& p, int dim, bool use_pe, int block_size) {
    if (c.use_fa) return fa_layer(b, p, dim, c.attn_dim, c.attn_dim, c.attn_heads);
    return sa_layer(b, p, dim, c.attn_heads, c.attn_dim, use_pe, block_size);
}

// ---------------------------------------------------------------------------
// Encoders: modules/autoencoder2d.py:16-72, autoencoder2d_nonsquared.py:17-68,
//           autoencoder2d_half_periodic.py:106-144
// ---------------------------------------------------------------------------
static void build_encoder(Builder& b, const lns_config& c, std::vector<Layer>& L) {
    const std::string p = std::string(c.ae_prefix) + "encoder.model.";
    const int my = c.ae_pad_y, mx = c.ae_pad_x;
    const int n = c.n_encoder_channels;
    const int32_t* ch = c.encoder_channels;
    auto name = [&](int i) { return p + std::to_string(i); };
    const int expect = (int)std::log2((double)(c.res_h / c.latent_resolution));
    if (n - 2 != expect) throw std::runtime_error("len(encoder_channels)-2 must equal log2(resolution//latent_resolution)");
    int idx = 0;
    L.push_back(conv_layer(name(0), b.conv(name(0), c.in_channels, ch[0], 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
    L.push_back(swish_layer(name(1)));
    idx = 2;
    const bool hp = c.ae_kind == LNS_AE_HALF_PERIODIC;
    const bool sq = c.ae_kind == LNS_AE_SQUARE;
    if (hp) L.push_back(hp_res_layer(b, name(idx++), ch[0], ch[0], my, mx));
    else L.push_back(same_conv(b, name(idx++), ch[0], ch[0], 3, my, mx)), (void)0;
    int res = c.res_h;
    for (int i = 0; i < n - 1; ++i) {
        int cin = ch[i];
        const int cout = ch[i + 1];
        for (int j = 0; j < c.encoder_res_blocks; ++j) {
            if (hp) L.push_back(hp_res_layer(b, name(idx++), cin, cout, my, mx));
            else L.push_back(res_layer(b, name(idx++), cin, cout, my, mx));
            cin = cout;
            if (c.ae_kind == LNS_AE_NONSQUARED && in_list(c.fourier_resolutions, c.n_fourier_resolutions, res)) {
                const int m = res <= 32 ? 6 : 10;
                L.push_back(fourier_layer(b, name(idx++), cin, cout, m, (int)(m * c.hw_ratio)));
            }
        }
        if (sq) {
            if (in_list(c.attn_resolutions, c.n_attn_resolutions, res) && c.use_attn_enc)
                L.push_back(attn_layer(b, c, name(idx++), cin, true, res * res));
            if (in_list(c.fourier_resolutions, c.n_fourier_resolutions, res)) {
                const int m = res <= 32 ? 6 : 10;
                L.push_back(fourier_layer(b, name(idx++), cin, cout, m, m));
            }
        }
        if (i != n - 2) {
            const std::string dn = name(idx++) + ".conv_layer";
            const int pk = b.conv(dn, ch[i + 1], ch[i + 1], 3);
            if (hp) {
                // DownSampleBlock2d: HalfPeriodicConv2d(ch, ch, 3, 2, 1)  (autoencoder2d_half_periodic.py:68-74)
                L.push_back(conv_layer(dn, pk, 3, 2, 1, 1, 1, 1, 1, my, mx));
            } else if (my == LNS_PAD_CIRCULAR) {
                // DownSampleBlock: F.pad circular (1,1,1,1) then Conv2d(3, stride 2, pad 0)  (basics.py:302-328)
                L.push_back(conv_layer(dn, pk, 3, 2, 1, 1, 1, 1, 1, LNS_PAD_CIRCULAR, LNS_PAD_CIRCULAR));
            } else {
                L.push_back(conv_layer(dn, pk, 3, 2, 1, 0, 1, 0, 1, LNS_PAD_ZEROS, LNS_PAD_ZEROS));
            }
            res /= 2;
        }
    }
    const int cl = ch[n - 1];
    if (hp) L.push_back(hp_res_layer(b, name(idx++), cl, cl, my, mx));
    else if (sq) L.push_back(same_conv(b, name(idx++), cl, cl, 3, my, mx));
    else L.push_back(res_layer(b, name(idx++), cl, cl, my, mx));
    L.push_back(gn32_layer(b, name(idx++), cl));
    L.push_back(swish_layer(name(idx++)));
    L.push_back(conv_layer(name(idx), b.conv(name(idx), cl, c.latent_dim, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
}

// CondResidualBlock(in, out, cond_channels, norm=True, n_groups=1, GELU): modules/cond_utils.py:58-128
static Layer cond_res_layer(Builder& b, const std::string& p, int cin, int cout, int E, int my, int mx) {
    Layer l;
    l.type = LT_CONDRES; l.name = p; l.cin = cin; l.cout = cout; l.mode_y = my; l.mode_x = mx; l.cr_E = E;
    l.conv1 = b.conv(p + ".conv1", cin, cout, 3);
    l.conv2 = b.conv(p + ".conv2", cout, cout, 3);
    if (cin != cout) l.chup = b.conv(p + ".shortcut", cin, cout, 1);
    l.g1 = b.vec_param(p + ".norm1.weight", {cin});
    l.b1 = b.vec_param(p + ".norm1.bias", {cin});
    l.g2 = b.vec_param(p + ".norm2.weight", {cout});
    l.b2 = b.vec_param(p + ".norm2.bias", {cout});
    l.cr_lin_w = b.vec_param(p + ".cond_emb.weight", {cout, E});
    l.cr_lin_b = b.vec_param(p + ".cond_emb.bias", {cout});
    return l;
}

// CondEncoder: modules/autoencoder2d_nonsquared.py:71-145
static void build_cond_encoder(Builder& b, const lns_config& c, std::vector<Layer>& L) {
    const std::string p = std::string(c.ae_prefix) + "encoder.";
    const int my = c.ae_pad_y, mx = c.ae_pad_x;
    const int n = c.n_encoder_channels, E = c.cond_emb_channels;
    const int32_t* ch = c.encoder_channels;
    if (c.ae_kind != LNS_AE_NONSQUARED) throw std::runtime_error("cond_encoder needs the non-squared autoencoder");
    if (E <= 0 || E > 256 || ch[0] > 256) throw std::runtime_error("cond_emb_channels / encoder_channels[0] out of range");
    if (n - 2 != (int)std::log2((double)(c.res_h / c.latent_resolution)))
        throw std::runtime_error("len(encoder_channels)-2 must equal log2(resolution//latent_resolution)");
    // to_in: 1x1 -> Swish -> 3x3
    L.push_back(conv_layer(p + "to_in.0", b.conv(p + "to_in.0", c.in_channels, ch[0], 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
    L.push_back(swish_layer(p + "to_in.1"));
    L.push_back(same_conv(b, p + "to_in.2", ch[0], ch[0], 3, my, mx));
    // embed: Linear(E, ch0) -> Swish -> Linear(ch0, E) on fourier_embedding(param, E)   (:104-106, :128)
    b.vec_param(p + "embed.0.weight", {ch[0], E}, VX_TRANSPOSE2D);
    b.vec_param(p + "embed.0.bias", {ch[0]});
    b.vec_param(p + "embed.2.weight", {E, ch[0]}, VX_TRANSPOSE2D);
    b.vec_param(p + "embed.2.bias", {E});
    for (int i = 0; i < n - 1; ++i) {
        int cin = ch[i];
        const int cout = ch[i + 1];
        const std::string q = p + "layers." + std::to_string(i);
        for (int j = 0; j < c.encoder_res_blocks; ++j) {
            L.push_back(cond_res_layer(b, q + ".0." + std::to_string(j), cin, cout, E, my, mx));
            cin = cout;
        }
        if (i != n - 2) {
            const std::string dn = q + ".1.conv_layer";
            const int pk = b.conv(dn, cout, cout, 3);
            if (my == LNS_PAD_CIRCULAR) L.push_back(conv_layer(dn, pk, 3, 2, 1, 1, 1, 1, 1, LNS_PAD_CIRCULAR, LNS_PAD_CIRCULAR));
            else L.push_back(conv_layer(dn, pk, 3, 2, 1, 0, 1, 0, 1, LNS_PAD_ZEROS, LNS_PAD_ZEROS));
        }
    }
    const int cl = ch[n - 1];
    L.push_back(cond_res_layer(b, p + "to_out_conv", cl, cl, E, my, mx));
    L.push_back(gn32_layer(b, p + "to_out.0", cl));
    L.push_back(swish_layer(p + "to_out.1"));
    L.push_back(conv_layer(p + "to_out.2", b.conv(p + "to_out.2", cl, c.latent_dim, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
}

// ---------------------------------------------------------------------------
// Decoders: modules/autoencoder2d.py:75-156, autoencoder2d_nonsquared.py:148-247,
//           autoencoder2d_half_periodic.py:147-230
// ---------------------------------------------------------------------------
static void build_decoder(Builder& b, const lns_config& c, std::vector<Layer>& L) {
    const std::string p = std::string(c.ae_prefix) + "decoder.model.";
    const int my = c.ae_pad_y, mx = c.ae_pad_x;
    const int n = c.n_decoder_channels;
    const int32_t* ch = c.decoder_channels;
    auto name = [&](int i) { return p + std::to_string(i); };
    const bool hp = c.ae_kind == LNS_AE_HALF_PERIODIC;
    const bool sq = c.ae_kind == LNS_AE_SQUARE;
    auto rb = [&](const std::string& nm, int cin, int cout) {
        return hp ? hp_res_layer(b, nm, cin, cout, my, mx) : res_layer(b, nm, cin, cout, my, mx);
    };
    int res = c.latent_resolution;
    const double hw = (double)c.res_w / (double)c.res_h;   // `resolutions[1] / resolutions[0]`
    auto blk = [&](int r) { return sq ? r * r : r * (int)(r * (hw + 0.5)); };
    int cin = ch[0];
    int idx = 0;
    if (sq) L.push_back(conv_layer(name(0), b.conv(name(0), c.latent_dim, cin, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
    else L.push_back(same_conv(b, name(0), c.latent_dim, cin, 3, my, mx));
    idx = 1;
    if (hp) {
        if (!c.disable_coarse_attn) {
            L.push_back(sa_layer(b, name(idx++), cin, c.attn_heads, c.attn_dim, false, 0));
            L.push_back(rb(name(idx++), cin, cin));
        } else {
            L.push_back(rb(name(idx++), cin, cin));
            L.push_back(rb(name(idx++), cin, cin));
        }
    } else {
        L.push_back(rb(name(idx++), cin, cin));
        if (!c.disable_coarse_attn) L.push_back(sa_layer(b, name(idx++), cin, c.attn_heads, c.attn_dim, true, blk(res)));
        L.push_back(rb(name(idx++), cin, cin));
    }
    for (int i = 0; i < n; ++i) {
        const int cout = ch[i];
        for (int j = 0; j < c.decoder_res_blocks; ++j) {
            L.push_back(rb(name(idx++), cin, cout));
            cin = cout;
            if (!sq && in_list(c.attn_resolutions, c.n_attn_resolutions, res))
                L.push_back(attn_layer(b, c, name(idx++), cin, !hp, blk(res)));
        }
        if (sq && in_list(c.attn_resolutions, c.n_attn_resolutions, res))
            L.push_back(attn_layer(b, c, name(idx++), cin, true, blk(res)));
        if (i != 0 && i != n - 1) {
            // UpSampleBlock: F.interpolate(scale_factor=2.0) then 3x3 conv  (basics.py:279-299)
            Layer u; u.type = LT_UP2; u.name = name(idx);
            L.push_back(u);
            L.push_back(same_conv(b, name(idx) + ".conv_layer", cin, cin, 3, my, mx));
            b.e->packs[L.back().pack].up2 = true;       // (the planner takes the four-tap phase form when the resize is exactly 2x)
            ++idx;
            res *= 2;
        }
    }
    {   // nn.Upsample(size=(Ly, Lx), mode='nearest')
        Layer u; u.type = LT_RESIZE; u.name = name(idx++); u.outH = c.Ly; u.outW = c.Lx;
        L.push_back(u);
    }
    res = c.Ly;
    L.push_back(same_conv(b, name(idx), cin, cin, 3, my, mx)); ++idx;
    b.e->packs[L.back().pack].up2 = true;
    if (c.final_smoothing) {
        L.push_back(fourier_layer(b, name(idx++), cin, cin, 16, sq ? 16 : (int)(16 * hw)));
    } else {
        if (in_list(c.attn_resolutions, c.n_attn_resolutions, res))
            L.push_back(attn_layer(b, c, name(idx++), cin, !hp, blk(res)));
        if (sq) L.push_back(conv_layer(name(idx), b.conv(name(idx), cin, cin, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
        else L.push_back(same_conv(b, name(idx), cin, cin, 3, my, mx));
        ++idx;
    }
    if (sq) L.push_back(gn_raw_layer(b, name(idx++), 8, cin));      // nn.GroupNorm(8, C): autoencoder2d.py:149
    else L.push_back(gn32_layer(b, name(idx++), cin));
    L.push_back(swish_layer(name(idx++)));
    L.push_back(conv_layer(name(idx), b.conv(name(idx), cin, c.in_channels, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
}

// ---------------------------------------------------------------------------
// Propagators: train_stage2_ns2d.py:25-87 (SW :25-87, twophase :25-87),
//              train_stage2_twophase_conditional.py:25-121
// ---------------------------------------------------------------------------
static void build_propagator(Builder& b, const lns_config& c, std::vector<Layer>& L) {
    const std::string p = c.prop_prefix;
    const int D = c.prop_n_embd;
    const int my = c.prop_pad_y, mx = c.prop_pad_x;
    L.push_back(conv_layer(p + "in_proj", b.conv(p + "in_proj", c.latent_dim, D, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
    if (c.prop_kind == LNS_PROP_CONDITIONAL) {
        const int E = c.cond_emb_dim;
        b.vec_param(p + "cond_emb_proj.0.weight", {E, E}, VX_TRANSPOSE2D);
        b.vec_param(p + "cond_emb_proj.0.bias", {E});
        b.vec_param(p + "cond_emb_proj.2.weight", {E, E}, VX_TRANSPOSE2D);
        b.vec_param(p + "cond_emb_proj.2.bias", {E});
    }
    for (int i = 0; i < c.prop_n_block; ++i) {
        const std::string q = p + "net." + std::to_string(i);
        Layer l;
        l.name = q; l.C = D; l.dil = c.prop_dilation; l.mode_y = my; l.mode_x = mx; l.blk_index = i;
        if (c.prop_kind == LNS_PROP_PLAIN) {
            l.type = LT_PROPBLOCK;
            l.p_g1 = b.vec_param(q + ".conv.0.weight", {D});
            l.p_b1 = b.vec_param(q + ".conv.0.bias", {D});
            l.p_c1 = b.conv(q + ".conv.1", D, D, 3);
            l.p_c3 = b.conv(q + ".conv.3", D, D, 3);
            l.p_c5 = b.conv(q + ".conv.5", D, D, 3);
        } else {
            l.type = LT_CONDBLOCK;
            const int E = c.cond_emb_dim;
            // per-block embedding MLP inputs, packed contiguously for the cond kernel
            b.vec_param(q + ".cond_emb.weight", {D, E}, VX_TRANSPOSE2D);
            b.vec_param(q + ".cond_emb.bias", {D});
            l.p_g1 = b.vec_param(q + ".conv1.0.weight", {D});
            l.p_b1 = b.vec_param(q + ".conv1.0.bias", {D});
            l.p_c1 = b.conv(q + ".conv1.1", D, D, 3);
            l.p_c3 = b.conv(q + ".conv1.3", D, D, 3);
            l.c_g = b.vec_param(q + ".cond_conv1.0.weight", {D});
            l.c_b = b.vec_param(q + ".cond_conv1.0.bias", {D});
            l.c_conv = b.conv(q + ".cond_conv1.2", D, D, 3);
            b.vec_param(q + ".cond_conv2.0.weight", {D});
            b.vec_param(q + ".cond_conv2.0.bias", {D});
            b.vec_param(q + ".cond_conv2.1.weight", {D, D, 1, 1}, VX_TRANSPOSE2D);
            b.vec_param(q + ".cond_conv2.1.bias", {D});
            b.vec_param(q + ".cond_conv2.3.weight", {D, D, 1, 1}, VX_TRANSPOSE2D);
            b.vec_param(q + ".cond_conv2.3.bias", {D});
        }
        l.p_g2 = b.vec_param(q + ".ffn.0.weight", {D});
        l.p_b2 = b.vec_param(q + ".ffn.0.bias", {D});
        l.p_f1 = b.conv(q + ".ffn.1", D, D, 1, false);
        l.p_f3 = b.conv(q + ".ffn.3", D, D, 1, false);
        L.push_back(l);
    }
    L.push_back(gn32_layer(b, p + "out_proj.0", D));
    L.push_back(conv_layer(p + "out_proj.1", b.conv(p + "out_proj.1", D, c.latent_dim, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
}

static int conv_out(int in, int pad_lo, int pad_hi, int k, int stride, int dil) {
    return (in + pad_lo + pad_hi - dil * (k - 1) - 1) / stride + 1;
}

void build_model(lns_engine* e) {
    const lns_config& c = e->cfg;
    Builder b(e);
    if (c.ae_kind != LNS_AE_NONE) {
        if (c.n_encoder_channels < 2 || c.n_encoder_channels > LNS_MAX_STAGES || c.n_decoder_channels < 1 ||
            c.n_decoder_channels > LNS_MAX_STAGES)
            throw std::runtime_error("bad channel list length");
        if (c.cond_encoder) build_cond_encoder(b, c, e->enc);
        else build_encoder(b, c, e->enc);
        build_decoder(b, c, e->dec);
        const std::string q = std::string(c.ae_prefix) + "quant_conv";
        e->enc.push_back(conv_layer(q, b.conv(q, c.latent_dim, c.latent_dim, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
        const std::string pq = std::string(c.ae_prefix) + "post_quant_conv";
        e->dec.insert(e->dec.begin(),
                      conv_layer(pq, b.conv(pq, c.latent_dim, c.latent_dim, 1), 1, 1, 1, 0, 0, 0, 0, 0, 0));
        // latent spatial size: follow the stride-2 convs of the encoder
        int H = c.Ly, W = c.Lx;
        for (const Layer& l : e->enc)
            if (l.type == LT_CONV && l.stride == 2) {
                H = conv_out(H, l.pad[0], l.pad[1], 3, 2, 1);
                W = conv_out(W, l.pad[2], l.pad[3], 3, 2, 1);
            }
        e->lat_C = c.latent_dim; e->lat_H = H; e->lat_W = W;
    }
    if (c.prop_kind != LNS_PROP_NONE) build_propagator(b, c, e->prop);
}

}  // namespace lns
