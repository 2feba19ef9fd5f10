"""Base class of the HIP-backed backbones and the autograd bridge to libopendpd_hip.so.

The reference backbones are nn.Modules composed of ATen ops with the duck type
`forward(x:(B,T,2), h_0) -> (B,T,2)` (models.py:150-160).  A NativeBackbone keeps the same
parameters (same names, shapes and registration order, so `state_dict()` is interchangeable with
the reference's) but stores them as views into ONE contiguous fp32 buffer — the `params` pointer
of the C ABI — and runs forward/backward as hand-written HIP kernels.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib


class RnnParams(nn.Module):
    """Parameter holder with torch.nn.GRU / nn.LSTM's names and construction-time RNG consumption
    (uniform(-1/sqrt(H), 1/sqrt(H)) over weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0 in that
    order), so that `reset_parameters()` afterwards sees the same generator state as the reference."""

    def __init__(self, input_size, hidden_size, gates, num_layers=1):
        super().__init__()
        self.input_size, self.hidden_size, self.gates, self.num_layers = input_size, hidden_size, gates, num_layers
        G = gates * hidden_size
        self.weight_ih_l0 = nn.Parameter(torch.empty(G, input_size))
        self.weight_hh_l0 = nn.Parameter(torch.empty(G, hidden_size))
        self.bias_ih_l0 = nn.Parameter(torch.empty(G))
        self.bias_hh_l0 = nn.Parameter(torch.empty(G))
        for layer in range(1, num_layers):      # torch.nn.RNNBase registers (and initialises) layer after layer; a layer's input is the one below's state
            setattr(self, f"weight_ih_l{layer}", nn.Parameter(torch.empty(G, hidden_size)))
            setattr(self, f"weight_hh_l{layer}", nn.Parameter(torch.empty(G, hidden_size)))
            setattr(self, f"bias_ih_l{layer}", nn.Parameter(torch.empty(G)))
            setattr(self, f"bias_hh_l{layer}", nn.Parameter(torch.empty(G)))
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for w in self.parameters():
            nn.init.uniform_(w, -stdv, stdv)


def init_gatewise(rnn, hidden_size, xavier_suffix="weight_ih_l0"):
    """Initialisation rule shared by the reference's recurrent backbones (e.g. gru.py:27-37):
    biases 0; every H-row gate block of a weight orthogonal; gate blocks of the input weight
    re-drawn xavier-uniform.  Same call order => same RNG consumption."""
    for name, p in rnn.named_parameters():
        n_gates = p.shape[0] // hidden_size
        if "bias" in name:
            nn.init.constant_(p, 0)
        if "weight" in name:
            for g in range(n_gates):
                nn.init.orthogonal_(p[g * hidden_size:(g + 1) * hidden_size, :])
        if xavier_suffix in name:
            for g in range(n_gates):
                nn.init.xavier_uniform_(p[g * hidden_size:(g + 1) * hidden_size, :])


def init_linear(lin, weight="xavier"):
    for name, p in lin.named_parameters():
        if "weight" in name:
            if weight == "xavier":
                nn.init.xavier_uniform_(p)
            elif weight == "kaiming":
                nn.init.kaiming_uniform_(p)
            elif weight == "orthogonal":
                nn.init.orthogonal_(p)
        if "bias" in name:
            nn.init.constant_(p, 0)


class _BackboneFn(torch.autograd.Function):
    """y = backbone(x) through odpd_backbone_fwd / odpd_backbone_bwd."""

    @staticmethod
    def forward(ctx, x, mod, grad_mode, *params):
        lib = _lib.load()
        if not x.is_cuda:
            raise RuntimeError("opendpd_amd backbones run on a HIP device only (no CPU fallback)")
        x = x.contiguous().float()
        B, T = x.shape[0], x.shape[1]
        flat = mod.flat_params()
        # (needs_input_grad looks at requires_grad only: under torch.no_grad() — net_eval, run_dpd — nothing will call backward, so no
        # checkpoints are asked for and the kernels may take their inference path)
        need_grad = grad_mode and any(ctx.needs_input_grad)      # (autograd switches grad mode off inside forward: the caller passes it)
        if mod.dx_needs_flag:
            # delta backbones: dL/dx lives in the 16-sequences-per-wave kernels only; the flag routes forward, checkpoint sizing
            # and backward of THIS call to them (include/opendpd_hip.h: ODPD_FLAG_NEED_DX)
            mod.desc.flags = (mod.desc.flags & ~_lib.FLAG_NEED_DX) | (_lib.FLAG_NEED_DX if ctx.needs_input_grad[0] else 0)
        ctx.flags = mod.desc.flags
        y = torch.empty_like(x)
        ckpt = None
        if need_grad:
            n = lib.odpd_ckpt_floats(C.byref(mod.desc), B, T)
            _lib.check(0 if n >= 0 else int(n), "odpd_ckpt_floats")
            ckpt = torch.empty(max(int(n), 1), dtype=torch.float32, device=x.device)
        stats = mod._stats_buffer(x.device)
        rc = lib.odpd_backbone_fwd(_lib.stream_ptr(), C.byref(mod.desc), B, T, _lib.ptr(flat), _lib.ptr(x),
                                   _lib.ptr(y), _lib.ptr(ckpt), _lib.ptr(stats))
        _lib.check(rc, f"odpd_backbone_fwd[{mod.backbone_name}]")
        ctx.mod = mod
        ctx.save_for_backward(x, ckpt if ckpt is not None else x.new_empty(0), flat)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        mod = ctx.mod
        x, ckpt, flat = ctx.saved_tensors
        B, T = x.shape[0], x.shape[1]
        dy = dy.contiguous().float()
        need_dx = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[3:])
        P = mod.n_flat
        partials = grad = dx = None
        mod.desc.flags = ctx.flags
        if need_w or (need_dx and mod.dx_needs_flag):      # the delta kernels compute the weight gradients in every backward launch
            rows = int(lib.odpd_partial_rows(C.byref(mod.desc), B, T, 0))
            _lib.check(0 if rows > 0 else rows, "odpd_partial_rows")
            partials = torch.empty(rows, P + _lib.LOSS_COLS, dtype=torch.float32, device=x.device)
        if need_dx:
            dx = torch.empty_like(x)
        if not (need_w or need_dx):
            return (None, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
        rc = lib.odpd_backbone_bwd(_lib.stream_ptr(), C.byref(mod.desc), B, T, _lib.ptr(flat), _lib.ptr(x),
                                   _lib.ptr(dy), _lib.ptr(ckpt) if ckpt.numel() else None, _lib.ptr(partials),
                                   _lib.ptr(dx))
        _lib.check(rc, f"odpd_backbone_bwd[{mod.backbone_name}]")
        if mod.dx_needs_flag:
            mod.desc.flags &= ~_lib.FLAG_NEED_DX       # per-call selection: nothing stale for the next user of the descriptor
        gparams = (None,) * (len(ctx.needs_input_grad) - 3)
        if need_w:
            grad = torch.empty(P + _lib.LOSS_COLS, dtype=torch.float32, device=x.device)
            rc = lib.odpd_reduce_partials(_lib.stream_ptr(), partials.shape[0], P, _lib.ptr(partials), _lib.ptr(grad), 0)
            _lib.check(rc, "odpd_reduce_partials")
            gparams = tuple(grad[o:o + n].view(shape) if need else None
                            for (o, n, shape), need in zip(mod._slices, ctx.needs_input_grad[3:]))
        return (dx, None, None) + gparams


class NativeBackbone(nn.Module):
    """nn.Module whose parameters are views into one flat fp32 buffer consumed by the HIP kernels."""

    backbone_name = None
    native = True
    dx_needs_flag = False      # True for the delta backbones (ODPD_FLAG_NEED_DX)

    def _finalize(self, hidden_size, thx=0.0, thh=0.0, bits_w=0, bits_a=0):
        """Call at the end of __init__ once every parameter holder is registered."""
        self.desc = _lib.ModelDesc(_lib.BACKBONE_IDS[self.backbone_name], int(hidden_size), float(thx), float(thh), int(bits_w),
                                   int(bits_a), 0)
        self._slices = []
        off = 0
        for p in self.parameters():
            self._slices.append((off, p.numel(), tuple(p.shape)))
            off += p.numel()
        self.n_flat = off
        self._flat = None
        self._stats = None
        self._sentinels = None

    # -- flat parameter buffer ---------------------------------------------------------------
    def _reflatten(self):
        ps = list(self.parameters())
        dev = ps[0].device
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous().to(dev)
            for p, (o, n, shape) in zip(ps, self._slices):
                p.data = flat[o:o + n].view(shape)
        self._flat = flat
        # first and last parameter stand guard for the whole buffer on the per-step fast path
        self._sentinels = ((ps[0], 0), (ps[-1], 4 * self._slices[-1][0]))

    def flat_params(self, full_check=False):
        """The contiguous parameter buffer (re-built if a parameter was re-pointed, e.g. by .to() — `_apply` drops it;
        the per-call check looks at the first and last parameter only, `full_check` at every one)."""
        f = self._flat
        ok = f is not None
        if ok:
            base = f.data_ptr()
            if full_check:
                for p, (o, n, _) in zip(self.parameters(), self._slices):
                    if p.data_ptr() != base + 4 * o or p.device != f.device:
                        ok = False
                        break
            else:
                for p, off in self._sentinels:
                    if p.data_ptr() != base + off:
                        ok = False
                        break
        if not ok:
            self._reflatten()
        return self._flat

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flat = None
        return out

    def _stats_buffer(self, device):
        return None

    def forward(self, x, h_0=None):
        return _BackboneFn.apply(x, self, torch.is_grad_enabled(), *self.parameters())
