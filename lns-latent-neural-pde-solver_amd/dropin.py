"""Drop-in `nn.Module` surface of the hot path.

The classes here keep the reference's constructor arguments, method names,
attribute names and state_dict keys/shapes
    SimpleAutoencoder(args).encode/decode/forward/load_checkpoint   modules/autoencoder2d.py:160-186
    SimpleCNN(latent_dim, [cond_emb_dim,] prop_n_block, prop_n_embd, dilation)(z[, param])
                                                                   train_stage2_ns2d.py:56-87
    LatentDynamics(args).x_to_z/z_to_x/predict/load_autoencoder     train_stage2_ns2d.py:90-158
but hold NO arithmetic: the parameter tree is generated from the engine's
parameter table (the single definition of the architecture lives in
csrc/lns_model.cpp) and every call runs the HIP engine through the C ABI.
Gradients exist through ONE entry point, `LatentDynamics.forward(z_in, z_out[, param], loss_fn)` (the stage-2 training
rollout), and only for the propagator's parameters.  Everything else is inference (the reference wraps this
path in torch.no_grad()): autoencoder parameters are created with requires_grad=False (stage-1 training is out of scope),
and `SimpleCNN.forward` / `SimpleAutoencoder.{encode,decode,forward}` raise LnsError when they are called with autograd
enabled on something that requires grad -- instead of returning a tensor without grad_fn that only fails at backward().
"""
import math
import types

import numpy as np
import torch
import torch.nn as nn

from . import engine as _engine
from ._lib import (LNS_AE_HALF_PERIODIC, LNS_AE_NONE, LNS_AE_NONSQUARED, LNS_AE_SQUARE,
                   LNS_PAD_CIRCULAR, LNS_PAD_ZEROS, LNS_PROP_CONDITIONAL, LNS_PROP_NONE,
                   LNS_PROP_PLAIN, LnsError)


# Bumped whenever a drop-in module's parameter TREE may have changed identity: a Parameter / buffer object replaced
# (`mod.x.weight = nn.Parameter(...)`, `load_state_dict(assign=True)`, `.to()` / `.cuda()` / `.float()`, which always
# rebind buffers).  `_Hosted._engine` caches the flat (key, tensor) list and re-walks the tree when the epoch moved;
# in-place changes (optimizer steps, `load_state_dict`, `copy_`) are caught by (storage pointer, version) per tensor.
_TREE_EPOCH = [0]


class _TreeWatch(nn.Module):
    """Mix-in of every drop-in module: any rebinding of a tensor / sub-module invalidates the cached flat list."""

    def __setattr__(self, name, value):
        if isinstance(value, (torch.Tensor, nn.Module)) or name in self.__dict__.get("_parameters", ()) or \
                name in self.__dict__.get("_buffers", ()):
            _TREE_EPOCH[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        _TREE_EPOCH[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        _TREE_EPOCH[0] += 1
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        _TREE_EPOCH[0] += 1
        super().register_buffer(name, tensor, persistent=persistent)

    def _apply(self, fn, *a, **k):
        _TREE_EPOCH[0] += 1
        r = super()._apply(fn, *a, **k)
        _TREE_EPOCH[0] += 1
        return r

    def load_state_dict(self, *a, **k):
        _TREE_EPOCH[0] += 1
        r = super().load_state_dict(*a, **k)
        _TREE_EPOCH[0] += 1
        return r

    def _load_from_state_dict(self, *a, **k):      # the per-module step of a PARENT's load_state_dict(assign=True)
        super()._load_from_state_dict(*a, **k)
        _TREE_EPOCH[0] += 1


class _Node(_TreeWatch):
    """Parameter container mirroring one reference sub-module."""

    def forward(self, *a, **k):
        raise LnsError("sub-modules of the drop-in hold parameters only; call the owning "
                       "SimpleAutoencoder / SimpleCNN / LatentDynamics")


def _init_value(key, shape, sibling_weight_shape):
    """PyTorch-default-like initial values (reference constructors rely on
    nn.Conv2d/nn.Linear defaults, SABlock._init_weights basics.py:358-369,
    zero_module cond_utils.py:12-16, spectral weights basics.py:119-124)."""
    leaf = key.rsplit(".", 1)[-1]
    t = torch.empty(shape, dtype=torch.float32)
    sa = any(s in key for s in (".to_q.", ".to_k.", ".to_v.", ".proj_out."))
    zero = (".cond_conv1.2." in key) or (".cond_conv2.3." in key) or \
        (key.startswith("encoder.") and ".conv2." in key and (".layers." in key or ".to_out_conv." in key))   # zero_module (cond_utils.py:96-97)
    if leaf == "inv_freq":
        dim = 2 * shape[0]
        return 1.0 / (10000 ** (torch.arange(0, dim, 2).float() / dim))
    if zero:
        return t.zero_()
    if leaf == "pe":
        return t.normal_(0.0, 0.02)
    if leaf in ("weights1", "weights2") and len(shape) == 5:
        return t.uniform_(0, 1).mul_(1.0 / (shape[0] * shape[1]))
    if len(shape) >= 2:
        if sa:
            return t.normal_(0.0, 0.02)
        fan_in = int(np.prod(shape[1:]))
        b = 1.0 / math.sqrt(fan_in)
        return t.uniform_(-b, b)
    if leaf == "weight":
        return t.fill_(1.0)
    if sibling_weight_shape is not None and len(sibling_weight_shape) >= 2 and not sa:
        b = 1.0 / math.sqrt(int(np.prod(sibling_weight_shape[1:])))
        return t.uniform_(-b, b)
    return t.zero_()


def _no_backward(what, *tensors, params=()):
    """Called at the top of an inference-only forward.  With autograd enabled and an input (or a parameter) that requires
    grad the caller expects a differentiable result; there is no backward for this path, so say so here."""
    if not torch.is_grad_enabled():
        return
    if any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors) or any(p.requires_grad for p in params):
        raise LnsError("%s has no backward: gradients exist only through LatentDynamics.forward(z_in, z_out[, param], "
                       "loss_fn); call it under torch.no_grad() (or on tensors / parameters with requires_grad=False)" % what)


def _grow_tree(root, table, strip="", requires_grad=True):
    shapes = {k: s for k, s, _ in table}
    for key, shape, is_buffer in table:
        assert key.startswith(strip), (key, strip)
        parts = key[len(strip):].split(".")
        node = root
        for p in parts[:-1]:
            if not hasattr(node, p):
                node.add_module(p, _Node())
            node = getattr(node, p)
        sib = shapes.get(key.rsplit(".", 1)[0] + ".weight")
        val = _init_value(key, shape, sib)
        if is_buffer:
            node.register_buffer(parts[-1], val)
        else:
            node.register_parameter(parts[-1], nn.Parameter(val, requires_grad=requires_grad))


class _Hosted(_TreeWatch):
    """A module whose forward runs on an lns engine.  The engine belongs to the
    outermost constructed object (`_owner`); nested views share it."""

    def _init_host(self, cfg=None, owner=None, prefix=""):
        object.__setattr__(self, "_owner_ref", owner if owner is not None else self)
        object.__setattr__(self, "_prefix", prefix)
        if owner is None:
            object.__setattr__(self, "_eng", _engine.Engine(cfg))
            object.__setattr__(self, "_sig", None)
            object.__setattr__(self, "_flat", None)
            object.__setattr__(self, "_flat_epoch", -1)

    @property
    def _owner(self):
        return self._owner_ref

    def _weights_signature(self):
        """(flat [(key, tensor)] list of the owner's CURRENT parameter tree, hashable signature of its contents)."""
        own = self._owner
        flat = own._flat
        if flat is None or own._flat_epoch != _TREE_EPOCH[0]:
            flat = [(k, t) for k, t in list(own.named_parameters()) + list(own.named_buffers())]
            object.__setattr__(own, "_flat", flat)
            object.__setattr__(own, "_flat_epoch", _TREE_EPOCH[0])
        return flat, tuple((id(t), t.data_ptr(), t._version) for _, t in flat)

    def _engine(self, like):
        """Engine with the current weights resident on `like`'s device."""
        own = self._owner
        if not isinstance(like, torch.Tensor) or not like.is_cuda:
            raise LnsError("the LNS drop-in runs on HIP device tensors only (input is on %s); there is "
                           "no CPU fallback -- use the reference implementation on CPU"
                           % (getattr(like, "device", "host")))
        # The tensors of the parameter tree are looked up once per TREE EPOCH (the walk costs 0.5 ms; anything that can
        # replace a Parameter / buffer object bumps _TREE_EPOCH, see _TreeWatch); a call compares (object identity,
        # storage pointer, version) of each tensor and re-packs the weights when something changed.
        flat, wsig = own._weights_signature()
        dev = like.device.index if like.device.index is not None else torch.cuda.current_device()
        sig = (dev, wsig)
        if own._sig != sig:
            own._eng.load_weights({k: t.detach().to("cpu", torch.float32).numpy() for k, t in flat}, dev)
            object.__setattr__(own, "_sig", sig)
        return own._eng


# ---------------------------------------------------------------------------
class SimpleAutoencoder(_Hosted):
    """modules/autoencoder2d{,_nonsquared,_half_periodic}.py `SimpleAutoencoder`."""
    _ae_kind = LNS_AE_SQUARE

    def __init__(self, args, _owner=None, _prefix=""):
        super().__init__()
        self.args = args
        if _owner is None:
            cfg = _engine.make_config(args, ae_kind=self._ae_kind, prop_kind=LNS_PROP_NONE)
            self._init_host(cfg=cfg)
            _grow_tree(self, self._eng.params, requires_grad=False)      # no backward through the autoencoder (module docstring)
        else:
            self._init_host(owner=_owner, prefix=_prefix)
            _grow_tree(self, [p for p in _owner._eng.params if p[0].startswith(_prefix)], strip=_prefix, requires_grad=False)

    def encode(self, x):
        _no_backward("SimpleAutoencoder.encode", x, params=self.parameters())
        with torch.no_grad():
            return self._engine(x).encode(x)

    def decode(self, z):
        _no_backward("SimpleAutoencoder.decode", z, params=self.parameters())
        with torch.no_grad():
            return self._engine(z).decode(z)

    def forward(self, x):
        return self.decode(self.encode(x))

    def load_checkpoint(self, path, device=None):
        ckpt = torch.load(path, map_location=device)
        self.load_state_dict(ckpt, strict=True)


class SimpleAutoencoderNonSquared(SimpleAutoencoder):
    _ae_kind = LNS_AE_NONSQUARED


class SimpleAutoencoderHalfPeriodic(SimpleAutoencoder):
    _ae_kind = LNS_AE_HALF_PERIODIC


class ConditionalSimpleAutoencoder(SimpleAutoencoder):
    """modules/autoencoder2d_nonsquared.py:279-305: `CondEncoder` (:71-145, `CondResidualBlock`s conditioned on
    embed(fourier_embedding(param)), modules/cond_utils.py:58-128) + the plain non-squared Decoder.
    encode(x, param) / decode(z) / forward(x, param); state_dict keys as the reference's
    (`encoder.to_in.*`, `encoder.embed.*`, `encoder.layers.i.0.j.{conv1,conv2,shortcut,norm1,norm2,cond_emb}.*`,
    `encoder.layers.i.1.conv_layer.*`, `encoder.to_out_conv.*`, `encoder.to_out.*`, `decoder.model.*`, ...)."""
    _ae_kind = LNS_AE_NONSQUARED

    def __init__(self, args):
        if not getattr(args, "cond_encoder", False):
            args = types.SimpleNamespace(**vars(args))
            args.cond_encoder = True
        super().__init__(args)

    def encode(self, x, param):
        _no_backward("ConditionalSimpleAutoencoder.encode", x, params=self.parameters())
        with torch.no_grad():
            return self._engine(x).encode(x, param)

    def forward(self, x, param):
        return self.decode(self.encode(x, param))

    def load_checkpoint(self, path):
        self.load_state_dict(torch.load(path), strict=True)


# ---------------------------------------------------------------------------
class SimpleCNN(_Hosted):
    """Latent propagator `SimpleCNN` of the stage-2 scripts.  `pad` = (mode_y, mode_x)."""
    _prop_kind = LNS_PROP_PLAIN
    _pad = (LNS_PAD_CIRCULAR, LNS_PAD_CIRCULAR)

    def __init__(self, latent_dim, prop_n_block, prop_n_embd, dilation=2, _owner=None, _prefix="",
                 _cond_emb_dim=None):
        super().__init__()
        self.latent_dim = latent_dim
        self.prop_n_block = prop_n_block
        self.prop_n_embd = prop_n_embd
        if _owner is None:
            a = types.SimpleNamespace(latent_dim=latent_dim, prop_n_block=prop_n_block, prop_n_embd=prop_n_embd,
                                      dilation=dilation, cond_emb_dim=_cond_emb_dim or latent_dim)
            cfg = _engine.make_config(a, ae_kind=LNS_AE_NONE, prop_kind=self._prop_kind, prop_pad=self._pad)
            self._init_host(cfg=cfg)
            _grow_tree(self, self._eng.params)
        else:
            self._init_host(owner=_owner, prefix=_prefix)
            _grow_tree(self, [p for p in _owner._eng.params if p[0].startswith(_prefix)], strip=_prefix)

    def forward(self, z, param=None):
        _no_backward("SimpleCNN.forward", z, params=self.parameters())
        with torch.no_grad():
            return self._engine(z).propagate(z, param)


class SimpleCNNHalfPeriodic(SimpleCNN):       # train_stage2_SW.py:56-87 (periodic_direction='x')
    _pad = (LNS_PAD_ZEROS, LNS_PAD_CIRCULAR)


class SimpleCNNZeros(SimpleCNN):              # train_stage2_twophase.py:56-87 (padding_mode='zeros')
    _pad = (LNS_PAD_ZEROS, LNS_PAD_ZEROS)


class SimpleCNNConditional(SimpleCNN):        # train_stage2_twophase_conditional.py:78-121
    _prop_kind = LNS_PROP_CONDITIONAL
    _pad = (LNS_PAD_ZEROS, LNS_PAD_ZEROS)

    def __init__(self, latent_dim, cond_emb_dim, prop_n_block, prop_n_embd, dilation=2, _owner=None, _prefix=""):
        super().__init__(latent_dim, prop_n_block, prop_n_embd, dilation, _owner=_owner, _prefix=_prefix,
                         _cond_emb_dim=cond_emb_dim)
        self.cond_emb_dim = cond_emb_dim

    def forward(self, z, param):
        _no_backward("SimpleCNN.forward", z, params=self.parameters())
        with torch.no_grad():
            return self._engine(z).propagate(z, param)


# ---------------------------------------------------------------------------
class _LatentRolloutFn(torch.autograd.Function):
    """z_pred = rollout(z0; propagator parameters) on the HIP engine, differentiable w.r.t. the parameters and z0
    (include/lns.h "training rollout"; csrc/lns_train.inc).  The parameters are read from their device tensors at every
    call, so an optimiser's in-place updates need no re-upload."""

    @staticmethod
    def forward(ctx, own, names, T, z0, param, *tensors):
        if not z0.is_cuda:
            raise LnsError("the LNS drop-in runs on HIP device tensors only (input is on %s); there is no CPU fallback" % z0.device)
        eng = own._eng
        params = {k: t.detach() for k, t in zip(names, tensors)}
        for k, t in params.items():
            if not t.is_cuda:
                raise LnsError("training rollout: parameter %s is not on the HIP device (call model.cuda())" % k)
        with torch.cuda.device(z0.device):
            z_pred, ws = eng.train_forward(params, z0.detach(), T, param=param)
        ctx.own, ctx.names, ctx.ws = own, names, ws
        ctx.save_for_backward(z0, z_pred, *tensors)
        return z_pred

    @staticmethod
    def backward(ctx, grad_out):
        z0, z_pred, *tensors = ctx.saved_tensors
        params = {k: t.detach() for k, t in zip(ctx.names, tensors)}
        with torch.cuda.device(z0.device):
            grads, gz = ctx.own._eng.train_backward(params, z0, z_pred, grad_out.contiguous().float(), ctx.ws,
                                                    need_z_grad=ctx.needs_input_grad[3])
        out = [None, None, None, gz, None]
        for i, k in enumerate(ctx.names):
            out.append(grads[k] if ctx.needs_input_grad[5 + i] else None)
        return tuple(out)


# ---------------------------------------------------------------------------
class LatentDynamics(_Hosted):
    """`LatentDynamics` of train_stage2_ns2d.py:90-158 (and the SW / two-phase variants)."""
    _family = "ns2d"
    _ae_cls = SimpleAutoencoder
    _prop_cls = SimpleCNN
    _ae_attr = "vq_ae"
    _conditional = False

    def __init__(self, args):
        super().__init__()
        if getattr(args, "family", None) is None:
            args = types.SimpleNamespace(**vars(args))
            args.family = self._family
        self.args = args
        self.latent_resolution = args.latent_resolution
        self.latent_dim = args.latent_dim
        cfg = _engine.make_config(args, ae_kind=self._ae_cls._ae_kind, prop_kind=self._prop_cls._prop_kind,
                                  ae_prefix=self._ae_attr + ".", prop_prefix="propagator.",
                                  prop_pad=self._prop_cls._pad)
        self._init_host(cfg=cfg)
        ae = self._ae_cls(args, _owner=self, _prefix=self._ae_attr + ".")
        self.add_module(self._ae_attr, ae)
        if self._conditional:
            prop = self._prop_cls(args.latent_dim, args.latent_dim, args.prop_n_block, args.prop_n_embd,
                                  args.dilation, _owner=self, _prefix="propagator.")
        else:
            prop = self._prop_cls(args.latent_dim, args.prop_n_block, args.prop_n_embd, args.dilation,
                                  _owner=self, _prefix="propagator.")
        self.propagator = prop

    @property
    def _ae(self):
        return getattr(self, self._ae_attr)

    def load_autoencoder(self, args):
        print("Loading pretrained autoencoder from {}".format(args.pretrained_checkpoint_path))
        self._ae.load_checkpoint(args.pretrained_checkpoint_path, device=getattr(args, "device", None))
        print("Pretrained autoencoder loaded successfully")
        for p in self._ae.parameters():
            p.requires_grad = False
        self._ae.eval()

    @torch.no_grad()
    def x_to_z(self, x):
        return self._ae.encode(x)

    @torch.no_grad()
    def z_to_x(self, z):
        return self._ae.decode(z)

    def forward(self, z_in, z_out, *rest):
        """forward(z_in, z_out, loss_fn) -- conditional: forward(z_in, z_out, param, loss_fn): the loss of the latent
        rollout started at z_in[:, 0] against the pre-encoded targets z_out [B,t_out,c,h,w]
        (train_stage2_ns2d.py:126-141; conditional train_stage2_twophase_conditional.py:160-175).
        With autograd enabled (training, train_stage2_ns2d.py:213-215) the rollout runs through `_LatentRolloutFn`:
        the HIP training forward keeps a tape and `loss.backward()` runs the HIP backward through time, filling `.grad`
        of the propagator's parameters (no gradient w.r.t. `param`).  Under torch.no_grad() (validation) it is the
        inference rollout."""
        if torch.is_grad_enabled():
            if len(rest) != (2 if self._conditional else 1):
                raise TypeError("forward() takes (z_in, z_out, param, loss_fn)" if self._conditional else "forward() takes (z_in, z_out, loss_fn)")
            cparam, loss_fn = (rest[0], rest[1]) if self._conditional else (None, rest[0])
            if z_in.dim() != 5 or z_in.shape[1] != 1:
                raise AssertionError("z_in must be [B,1,c,h,w] (t_in == 1)")
            z0 = z_in[:, 0].contiguous().float()
            own = self._owner
            named = dict(own.named_parameters())
            names = [k for k in named if k.startswith("propagator.")]
            z_pred = _LatentRolloutFn.apply(own, names, int(z_out.shape[1]), z0, cparam, *[named[k] for k in names])
            return loss_fn(z_pred, z_out)
        param = None
        if self._conditional:
            if len(rest) != 2:
                raise TypeError("forward() takes (z_in, z_out, param, loss_fn)")
            param, loss_fn = rest
        else:
            if len(rest) != 1:
                raise TypeError("forward() takes (z_in, z_out, loss_fn)")
            loss_fn, = rest
        if z_in.dim() != 5 or z_in.shape[1] != 1:
            raise AssertionError("z_in must be [B,1,c,h,w] (t_in == 1)")
        z0 = z_in[:, 0].contiguous()
        z_pred, _ = self._engine(z0).rollout_latent(z0, z_out.shape[1], param=param, to_x=False)
        return loss_fn(z_pred, z_out)

    @staticmethod
    def _fields(x):
        # SW / two-phase loaders hand [B,1,C,H,W]; the reference squeezes (B=1 safe here)
        if x.dim() == 5 and x.shape[1] == 1:
            x = x[:, 0]
        return x

    @torch.no_grad()
    def predict(self, x, steps, *rest, to_x=False, return_latents=False):
        """predict(x, steps, to_x=False) -- conditional: predict(x, steps, param, to_x=False)."""
        param = None
        if self._conditional:
            if not rest:
                raise TypeError("predict() missing required argument: 'param'")
            param, rest = rest[0], rest[1:]
        if rest:
            to_x = rest[0]
        x = self._fields(x)
        return self._engine(x).rollout(x, steps, param=param, to_x=to_x, return_latents=return_latents)


class LatentDynamicsSW(LatentDynamics):                 # train_stage2_SW.py:90-159
    _family = "sw_half_periodic"
    _ae_cls = SimpleAutoencoderHalfPeriodic
    _prop_cls = SimpleCNNHalfPeriodic


class LatentDynamicsSWNonSquared(LatentDynamics):       # BASELINE config 3 (SW through autoencoder2d_nonsquared)
    _family = "sw_nonsquared"
    _ae_cls = SimpleAutoencoderNonSquared
    _prop_cls = SimpleCNNHalfPeriodic


class LatentDynamicsTwoPhase(LatentDynamics):           # train_stage2_twophase.py:90-159
    _family = "twophase"
    _ae_cls = SimpleAutoencoderNonSquared
    _prop_cls = SimpleCNNZeros


class LatentDynamicsTwoPhaseConditional(LatentDynamics):  # train_stage2_twophase_conditional.py:124-193
    _family = "twophase_cond"
    _ae_cls = SimpleAutoencoderNonSquared
    _prop_cls = SimpleCNNConditional
    _ae_attr = "ae"
    _conditional = True


FAMILY_CLASSES = {
    "ns2d": LatentDynamics,
    "sw_half_periodic": LatentDynamicsSW,
    "sw_nonsquared": LatentDynamicsSWNonSquared,
    "twophase": LatentDynamicsTwoPhase,
    "twophase_cond": LatentDynamicsTwoPhaseConditional,
}


def build_dynamics(args):
    return FAMILY_CLASSES[args.family](args)
