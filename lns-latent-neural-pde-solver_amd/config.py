"""Hyper-parameter namespaces for the hot path.

The reference reads flat YAML keys as attributes of an `args` namespace
(e.g. modules/autoencoder2d.py:19-27,78-92; train_stage2_ns2d.py:94-104).  The
drop-in classes in this package read the SAME attribute names, so a reference
YAML can be passed through `namespace_from_dict(yaml.safe_load(...))` unchanged.

The named presets are the configurations of BASELINE.json / SURVEY.md section 8d.
"""
import copy
import types

_COMMON = dict(
    encoder_channels=[64, 64, 64, 128, 128],
    fourier_resolutions=[],
    encoder_res_blocks=1,
    use_attn_enc=False,
    use_fa=True,
    decoder_channels=[128, 128, 64, 64],
    decoder_res_blocks=1,
    final_smoothing=False,
    disable_coarse_attn=False,
    noise_level=0.0,
    prop_n_embd=128,
)

PRESETS = {
    # BASELINE.json configs 1, 2, 5 (scaled from configs/ns2d_stage2_prop.yml:7-35)
    "ns2d_128": dict(_COMMON, family="ns2d", latent_dim=16, Ly=128, Lx=128, resolution=128,
                     in_channels=3, latent_resolution=16, is_periodic=True,
                     attn_resolutions=[32, 64], attn_heads=8, attn_dim=64,
                     prop_n_block=3, dilation=2),
    # the repository's own NS2d config (configs/ns2d_stage2_prop.yml)
    "ns2d_64": dict(_COMMON, family="ns2d", latent_dim=16, Ly=64, Lx=64, resolution=64,
                    in_channels=1, latent_resolution=8, is_periodic=True,
                    attn_resolutions=[16, 32], attn_heads=8, attn_dim=64,
                    prop_n_block=3, dilation=2),
    # small, fast NS2d-shaped case for CPU tests / per-op goldens
    "ns2d_mini": dict(family="ns2d", latent_dim=8, Ly=32, Lx=32, resolution=32,
                      in_channels=2, latent_resolution=4, is_periodic=True,
                      encoder_channels=[32, 32, 32, 64, 64], fourier_resolutions=[],
                      encoder_res_blocks=1, use_attn_enc=False, use_fa=True,
                      decoder_channels=[64, 64, 32, 32], attn_resolutions=[8, 16],
                      decoder_res_blocks=1, final_smoothing=False, attn_heads=2, attn_dim=32,
                      disable_coarse_attn=False, noise_level=0.0,
                      prop_n_block=2, prop_n_embd=64, dilation=2),
    # BASELINE.json config 3: shallow water 96x192x5 through autoencoder2d_nonsquared
    "sw_96x192x5": dict(_COMMON, family="sw_nonsquared", latent_dim=64, Ly=96, Lx=192,
                        resolutions=[96, 192], hw_ratio=2, in_channels=5, latent_resolution=12,
                        is_periodic=True, attn_resolutions=[24, 48],
                        decoder_attn_heads=8, decoder_attn_dim=64,
                        prop_n_block=4, dilation=3),
    # the repository's own SW config (configs/SW_stage2_prop.yml): half-periodic AE
    "sw_half_periodic": dict(_COMMON, family="sw_half_periodic", latent_dim=64, Ly=96, Lx=192,
                             resolutions=[96, 192], hw_ratio=2, in_channels=3,
                             latent_resolution=12, periodic_direction="x",
                             attn_resolutions=[24, 48], decoder_attn_heads=8,
                             decoder_attn_dim=64, prop_n_block=4, dilation=3),
    # BASELINE.json config 4 (configs/twophase_stage2_cond_prop.yml:7-34 verbatim)
    "twophase_cond": dict(_COMMON, family="twophase_cond", latent_dim=64, Ly=61, Lx=121,
                          resolutions=[61, 121], hw_ratio=2, in_channels=4, latent_resolution=7,
                          is_periodic=False, cond_channels=1, cond_emb_channels=64,
                          attn_resolutions=[15, 30], decoder_attn_heads=8, decoder_attn_dim=64,
                          prop_n_block=4, dilation=2),
    # ConditionalSimpleAutoencoder on the two-phase shape (modules/autoencoder2d_nonsquared.py:279-305; the keys
    # cond_channels / cond_emb_channels are those of configs/twophase_stage2_cond_prop.yml:13-14)
    "twophase_cond_ae": dict(_COMMON, family="twophase", latent_dim=64, Ly=61, Lx=121,
                             resolutions=[61, 121], hw_ratio=2, in_channels=4, latent_resolution=7,
                             is_periodic=False, cond_channels=1, cond_emb_channels=64, cond_encoder=True,
                             attn_resolutions=[15, 30], decoder_attn_heads=8, decoder_attn_dim=64,
                             prop_n_block=4, dilation=2),
    # the same at a size the CPU suite runs in seconds
    "cond_ae_mini": dict(family="twophase", latent_dim=8, Ly=29, Lx=57, resolutions=[29, 57], hw_ratio=2,
                         in_channels=3, latent_resolution=7, is_periodic=False, cond_channels=1, cond_emb_channels=16,
                         cond_encoder=True, encoder_channels=[32, 32, 64, 64], fourier_resolutions=[],
                         encoder_res_blocks=1, use_attn_enc=False, use_fa=True, decoder_channels=[64, 32, 32],
                         attn_resolutions=[14], decoder_res_blocks=1, final_smoothing=False,
                         decoder_attn_heads=2, decoder_attn_dim=32, disable_coarse_attn=False, noise_level=0.0,
                         prop_n_block=2, prop_n_embd=64, dilation=2),
    # configs/twophase_stage2_prop.yml (unconditional two-phase)
    "twophase": dict(_COMMON, family="twophase", latent_dim=64, Ly=61, Lx=121,
                     resolutions=[61, 121], hw_ratio=2, in_channels=4, latent_resolution=7,
                     is_periodic=False, attn_resolutions=[15, 30], decoder_attn_heads=8,
                     decoder_attn_dim=64, prop_n_block=4, dilation=2),
}


def namespace_from_dict(d) -> types.SimpleNamespace:
    return types.SimpleNamespace(**copy.deepcopy(dict(d)))


def preset(name: str, **overrides) -> types.SimpleNamespace:
    d = copy.deepcopy(PRESETS[name])
    d.update(overrides)
    return types.SimpleNamespace(**d)
