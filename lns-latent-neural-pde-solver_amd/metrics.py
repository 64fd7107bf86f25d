"""Validation-loop helpers next to the hot path: fused denormalise + relative-L2 metric, bulk
dataset encode.  Mirrors the reference call sites:

    y_hat = val_dataset.denormalize(model.predict(x, T, to_x=True)); y = val_dataset.denormalize(y)
    frame_wise = relative_lp_loss(y_hat, y, reduce_dim=(3, 4), p=2)        # train_stage2_ns2d.py:253-257
    seq_wise   = relative_lp_loss(y_hat, y, reduce_dim=(1, 3, 4), p=2)
    dataset.encode_dataset(vq_ae, device)                                   # ns2d_fno_stage2_simpleae.py:81-93

The other datasets' denormalize() (per-channel statistics, dataset/Stage2_SW.py:60-72; closed-tank wall velocities
zeroed and VOF clamped, dataset/twophase_flow_stage2.py:370-390) are the same kernel with a per-channel spec:
`relative_l2(..., mean=[...], std=[...], zero_wall_channels=(0, 1), clamp_channels=(3,))` or `twophase_spec(...)`.
"""
import ctypes

import torch

from . import _lib


def twophase_spec(vel_mean, vel_std, prs_mean, prs_std):
    """Keyword arguments for relative_l2 that restate TwoPhaseFlow dataset denormalisation
    (dataset/twophase_flow_stage2.py:370-390): channels (u, v, p, vof)."""
    return dict(mean=[vel_mean, vel_mean, prs_mean, 0.0], std=[vel_std, vel_std, prs_std, 1.0],
                zero_wall_channels=(0, 1), clamp_channels=(3,), clamp=(0.0, 1.0 + 1e-8))


def relative_l2(y_hat, y, mean=0.0, std=1.0, eps=1e-8, zero_wall_channels=(), clamp_channels=(), clamp=(0.0, 1.0 + 1e-8)):
    """(frame_wise [B,T,C], seq_wise [B,C]) relative L2 errors of a normalised rollout y_hat against the normalised
    ground truth y, both [B,T,C,H,W], after the dataset's denormalisation: x*std + mean with scalar or per-channel
    statistics, optionally the wall rows/columns of `zero_wall_channels` zeroed and `clamp_channels` clamped.
    One HIP pass over both tensors; CPU tensors raise -- there is no CPU fallback."""
    if not (y_hat.is_cuda and y.is_cuda):
        raise RuntimeError("lns_amd.metrics.relative_l2 needs CUDA/HIP tensors (no CPU fallback)")
    if y_hat.shape != y.shape or y_hat.dim() != 5:
        raise ValueError("expected two [B,T,C,H,W] tensors of the same shape")
    y_hat = y_hat.contiguous().float()
    y = y.contiguous().float()
    B, T, C, H, W = y.shape
    frame = torch.empty((B, T, C), dtype=torch.float32, device=y.device)
    seq = torch.empty((B, C), dtype=torch.float32, device=y.device)
    scratch = torch.empty((B * T * C * 2,), dtype=torch.float32, device=y.device)
    if y_hat.device != y.device:
        raise ValueError("y_hat is on %s but y is on %s" % (y_hat.device, y.device))
    L = _lib.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
    per_channel = (not isinstance(mean, (int, float))) or (not isinstance(std, (int, float))) or \
        len(zero_wall_channels) > 0 or len(clamp_channels) > 0
    # the metric entry points launch on the CURRENT device (include/lns.h): make that the tensors' device
    with torch.cuda.device(y.device):
        rc = _launch_metric(L, per_channel, y_hat, y, B, T, C, H, W, mean, std, eps, zero_wall_channels, clamp_channels,
                            clamp, frame, seq, scratch, stream)
    if rc != 0:
        raise RuntimeError("lns_metric_rel_l2 failed (rc=%d)" % rc)
    return frame, seq


def _launch_metric(L, per_channel, y_hat, y, B, T, C, H, W, mean, std, eps, zero_wall_channels, clamp_channels, clamp,
                   frame, seq, scratch, stream):
    if not per_channel:
        rc = L.lns_metric_rel_l2(y_hat.data_ptr(), y.data_ptr(), B, T, C, H * W, float(mean), float(std), float(eps),
                                 frame.data_ptr(), seq.data_ptr(), scratch.data_ptr(), stream)
    else:
        def per_c(v):
            v = [float(v)] * C if isinstance(v, (int, float)) else [float(e) for e in v]
            if len(v) != C:
                raise ValueError("per-channel statistics need %d entries" % C)
            return (ctypes.c_float * C)(*v)
        flags = [0] * C
        for c in zero_wall_channels:
            flags[c] |= 1
        for c in clamp_channels:
            flags[c] |= 2
        m, sd, fl = per_c(mean), per_c(std), (ctypes.c_int * C)(*flags)
        rc = L.lns_metric_rel_l2_ch(y_hat.data_ptr(), y.data_ptr(), B, T, C, H, W, m, sd, fl, float(clamp[0]),
                                    float(clamp[1]), float(eps), frame.data_ptr(), seq.data_ptr(),
                                    scratch.data_ptr(), stream)
    return rc


def sw_norm(u_mean, u_std, v_mean, v_std, pres_mean, pres_std):
    """Keyword arguments for encode_dataset restating Stage2_SW.normalize (dataset/Stage2_SW.py:74-78): channels
    (u, v, pres), each by its own statistics, no epsilon."""
    return dict(mean=[u_mean, v_mean, pres_mean], std=[u_std, v_std, pres_std], eps=0.0)


def twophase_norm(vel_mean, vel_std, prs_mean, prs_std):
    """Keyword arguments for encode_dataset restating TwoPhaseFlow.normalize_data + the channel layout of its
    encode_dataset (dataset/twophase_flow_stage2.py:304-313, :325-326): (u, v) by the velocity statistics, pressure by its
    own, the VOF channel as is; chunks of 32 frames (:329-334)."""
    return dict(mean=[vel_mean, vel_mean, prs_mean, 0.0], std=[vel_std, vel_std, prs_std, 1.0], eps=0.0, chunk=32)


@torch.no_grad()
def encode_dataset(autoencoder, frames, chunk=32, mean=0.0, std=1.0, eps=1e-8, out_device="cpu", param=None):
    """Pre-encodes a whole trajectory set for stage-2 training: frames [N,C,H,W] (raw, un-normalised; torch tensor or
    numpy array) -> latents [N, latent_dim, h, w], `chunk` frames per encoder call, with the dataset normalisation
    (u - mean) / (std + eps) applied first on the device.  mean / std: scalars (NS2d,
    dataset/ns2d_fno_stage2_simpleae.py:78-93) or per-channel sequences (`sw_norm(...)`: dataset/Stage2_SW.py:74-105;
    `twophase_norm(...)`: dataset/twophase_flow_stage2.py:304-337).  `param` [N] (one value per frame) is passed to a
    ConditionalSimpleAutoencoder's encode."""
    frames = torch.as_tensor(frames)
    dev = next(iter(autoencoder.parameters())).device
    C = frames.shape[1]

    def stat(v):
        v = [float(v)] * C if isinstance(v, (int, float)) else [float(e) for e in v]
        if len(v) != C:
            raise ValueError("per-channel statistics need %d entries" % C)
        return v
    # (u - mean) / (std + eps) = u * scale + shift, in fp64 on the host: 2 C numbers.  The map is applied inside the
    # encoder's first convolution (its per-(sample, channel) scale / shift prologue, lns_encode_affine): no normalised copy
    # of the frames is ever written, and no tensor arithmetic runs outside the HIP kernels.
    m, sd = stat(mean), stat(std)
    table = torch.tensor([[1.0 / (s_ + eps), -m_ / (s_ + eps)] for m_, s_ in zip(m, sd)], dtype=torch.float64)
    table = table.to(torch.float32).to(dev)
    if param is not None:
        param = torch.as_tensor(param)
    outs = []
    for s in range(0, frames.shape[0], chunk):
        u = frames[s:s + chunk].to(dev, dtype=torch.float32)
        ss = table.unsqueeze(0).expand(u.shape[0], C, 2).contiguous()
        eng = autoencoder._engine(u)
        z = eng.encode(u, None if param is None else param[s:s + chunk].to(dev), scale_shift=ss)
        outs.append(z.to(out_device))
    return torch.cat(outs, 0)
