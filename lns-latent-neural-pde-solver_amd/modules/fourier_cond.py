"""Drop-ins for the Fourier blocks as standalone modules (no shipped config
reaches them, so they are exposed as ops with unit-level parity):
  FourierBasicBlock      reference modules/basics.py:531-583 (+ SpectralConv2d :99-149)
  CondFourierBasicBlock  reference modules/fourier_cond.py:84-117 (+ SpectralConv2d :32-81, FreqLinear :16-29)
Same constructor arguments and state_dict keys; forward runs the HIP kernels through the C ABI."""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from ..dropin import _Node
from .._lib import LnsError


def _host(t):
    a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
    return a, a.ctypes.data_as(ctypes.c_void_p)


_ACTS = {"silu": 1, "gelu": 2, "relu": 3, "tanh": 4, "sigmoid": 5}     # ACTIVATION_REGISTRY, modules/basics.py:10-16


class _FourierBase(nn.Module):
    def __init__(self, in_planes, planes, modes, residual=True, conditional=False, activation="gelu"):
        super().__init__()
        if len(modes) != 2:
            raise NotImplementedError("only the 2-D block is on the accelerated path (the hot path is 2-D, SURVEY 2.1)")
        if residual and in_planes != planes:
            raise ValueError("residual=True adds the input to the output: in_planes must equal planes")
        self.modes = list(modes)
        self.in_planes, self.planes, self.residual = in_planes, planes, bool(residual)
        self._act = _ACTS[activation]
        m1, m2 = modes
        scale = 1.0 / (in_planes * planes)
        self.fourier = _Node()
        self.fourier.weights1 = nn.Parameter(scale * torch.rand(in_planes, planes, m1, m2, 2), requires_grad=False)
        self.fourier.weights2 = nn.Parameter(scale * torch.rand(in_planes, planes, m1, m2, 2), requires_grad=False)
        self.conv = _Node()
        b = 1.0 / np.sqrt(in_planes)
        self.conv.weight = nn.Parameter(torch.empty(planes, in_planes, 1, 1).uniform_(-b, b), requires_grad=False)
        self.conv.bias = nn.Parameter(torch.empty(planes).uniform_(-b, b), requires_grad=False)
        if conditional:
            fs = 1.0 / (in_planes + 4 * m1 * m2)
            self.fourier.cond_emb = _Node()
            self.fourier.cond_emb.weights = nn.Parameter(fs * torch.randn(in_planes, 4 * m1 * m2), requires_grad=False)
            self.fourier.cond_emb.bias = nn.Parameter(torch.zeros(1, 4 * m1 * m2), requires_grad=False)
            self.cond_emb = _Node()
            self.cond_emb.weight = nn.Parameter(torch.empty(planes, in_planes).uniform_(-b, b), requires_grad=False)
            self.cond_emb.bias = nn.Parameter(torch.empty(planes).uniform_(-b, b), requires_grad=False)

    def _weights(self):
        ws = [self.fourier.weights1, self.fourier.weights2, self.conv.weight, self.conv.bias]
        if hasattr(self.fourier, "cond_emb"):
            ws += [self.fourier.cond_emb.weights, self.fourier.cond_emb.bias, self.cond_emb.weight, self.cond_emb.bias]
        return ws

    def _handle_for(self, device):
        """The engine-side block object (weights resident on `device`).  Rebuilt when a parameter tensor was replaced or
        written in place (load_state_dict, an optimiser step, .to()) or the device changed; otherwise a forward uploads
        nothing."""
        ws = self._weights()
        sig = (device.index,) + tuple((id(w), w.data_ptr(), w._version) for w in ws)
        if getattr(self, "_h_sig", None) == sig:
            return self._h
        self._drop_handle()
        keep = [_host(w) for w in ws]
        ptr = [k[1] for k in keep] + [None] * (8 - len(keep))
        h = ctypes.c_void_p()
        rc = _lib.lib().lns_fourier_block_create(self.in_planes, self.planes, self.modes[0], self.modes[1], ptr[0], ptr[1],
                                                 ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], ptr[7], self._act,
                                                 int(self.residual), device.index, ctypes.byref(h))
        if rc != 0:
            raise LnsError("lns_fourier_block_create failed (%d)" % rc)
        self._h, self._h_sig = h, sig
        return h

    def _drop_handle(self):
        h = getattr(self, "_h", None)
        if h is not None:
            _lib.lib().lns_fourier_block_destroy(h)
        self._h, self._h_sig = None, None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:  # noqa: BLE001 (interpreter shutdown)
            pass

    @torch.no_grad()
    def _run(self, x, cond):
        if not x.is_cuda or x.dtype != torch.float32:
            raise LnsError("HIP device fp32 tensors only (no CPU fallback)")
        x = x.contiguous()
        B, C, H, W = x.shape
        if C != self.in_planes:
            raise LnsError("expected %d input channels, got %d" % (self.in_planes, C))
        if (cond is not None) != hasattr(self.fourier, "cond_emb"):
            raise LnsError("cond_emb must be given exactly for the conditional block")
        y = torch.empty((B, self.planes, H, W), dtype=torch.float32, device=x.device)
        cptr = None
        if cond is not None:
            cond = cond.to(device=x.device, dtype=torch.float32).contiguous()
            cptr = ctypes.c_void_p(cond.data_ptr())
        with torch.cuda.device(x.device):
            h = self._handle_for(x.device)
            rc = _lib.lib().lns_fourier_block_forward(h, ctypes.c_void_p(x.data_ptr()), cptr, B, H, W,
                                                      ctypes.c_void_p(y.data_ptr()),
                                                      ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc != 0:
            raise LnsError("lns_fourier_block_forward failed (%d)" % rc)
        return y


class FourierBasicBlock(_FourierBase):
    def __init__(self, in_planes, planes, modes, activation="gelu", residual=True):
        if activation not in _ACTS:
            raise NotImplementedError(f"Activation {activation} not implemented")
        super().__init__(in_planes, planes, modes, residual, conditional=False, activation=activation)

    def forward(self, x):
        return self._run(x, None)


class CondFourierBasicBlock(_FourierBase):
    def __init__(self, in_planes, planes, modes, residual=True):
        super().__init__(in_planes, planes, modes, residual, conditional=True)

    def forward(self, x, cond_emb):
        return self._run(x, cond_emb)
