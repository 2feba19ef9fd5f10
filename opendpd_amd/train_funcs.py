"""Train / eval loops: drop-in for the reference's modules/train_funcs.py.

`net_train` / `net_eval` keep the reference signatures and semantics (train_funcs.py:16-90):
per batch zero_grad -> forward -> loss -> backward -> clip_grad_norm_ -> optimizer.step, and the
returned log['loss'] is the unweighted mean of the per-batch losses (train_funcs.py:50).

Fast path: when `optimizer` is a `FusedAdamW` built over a HIP-backed `CoreModel`, the whole step runs
as three launches on the current stream and never synchronises with the host:
    odpd_train_fwd_bwd  (forward + loss + BPTT, BPTT state in LDS)
 -> odpd_reduce_partials (deterministic reduction of per-wavefront gradient rows)
 -> [one RCCL all-reduce of P+4 floats when the batch is sharded over ranks]
 -> odpd_clip_adamw_step (global-norm clip + AdamW).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .backbones.native import NativeBackbone
from .dist import shard_range
from .models import CascadedModel, CoreModel


def _sample_format(x_stream, y_stream):
    """enum odpd_sample_format of a pair of resident (N,2) streams: fp32, or bf16 pairs (one 32-bit word per I/Q sample)"""
    if x_stream.dtype != y_stream.dtype or x_stream.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"resident streams must both be float32 or both bfloat16 (got {x_stream.dtype}, {y_stream.dtype})")
    return _lib.SAMPLES_BF16 if x_stream.dtype == torch.bfloat16 else _lib.SAMPLES_F32


def _loss_kind(criterion):
    if isinstance(criterion, nn.MSELoss) and criterion.reduction == "mean":
        return "l2"
    if isinstance(criterion, nn.L1Loss) and criterion.reduction == "mean":
        return "l1"
    if criterion in ("l2", "l1"):
        return criterion
    return None


class FusedAdamW:
    """torch.optim.AdamW (defaults of project.py:283) over the flat parameter buffer of ONE HIP-backed
    CoreModel, fused with clip_grad_norm_.  Exposes `param_groups[0]['lr']` like a torch optimizer so
    ReduceLROnPlateau-style schedulers (project.py:289-296) can drive it."""

    kind = "adamw"

    def __init__(self, net, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, process_group=None):
        # a CascadedModel trains its DPD only: the PA is frozen (models.py:169-171, train_dpd.py:60-63)
        self.pa = None
        trained = net
        if isinstance(net, CascadedModel):
            if any(p.requires_grad for p in net.pa_model.parameters()):
                raise ValueError("FusedAdamW on a CascadedModel expects freeze_pa_model() to have been called")
            trained, self.pa = net.dpd_model, net.pa_model.backbone
            if not isinstance(self.pa, NativeBackbone):
                raise TypeError("FusedAdamW needs a HIP-backed PA model (this one runs through ATen)")
            if self.pa.dx_needs_flag:       # a frozen delta PA is asked for dL/du in every step
                self.pa.desc.flags |= _lib.FLAG_NEED_DX
        if not (isinstance(trained, CoreModel) and isinstance(trained.backbone, NativeBackbone)):
            raise TypeError("FusedAdamW needs a HIP-backed CoreModel (or a CascadedModel of two)")
        self.net = net
        self.trained = trained
        self.backbone = trained.backbone
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.process_group = process_group
        self.step_count = 0
        self._state_dev = None
        self._tuning_gen = -1
        self._fused_ok, self._partials, self._cascade_bufs = {}, {}, {}

    # -- state -------------------------------------------------------------------------------
    def _ensure(self, device):
        # buffers sized by odpd_partial_rows / odpd_train_workspace_floats / odpd_ckpt_floats depend on the kernel-selection knobs
        # as well as on (B, T): a knob change invalidates every cached one (a larger grid would write past a stale buffer)
        self._check_tuning()
        if self._state_dev == device:
            return
        P = self.backbone.n_flat
        self.grad = torch.zeros(P + _lib.LOSS_COLS, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(P, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(P, dtype=torch.float32, device=device)
        self.norm = torch.zeros(1, dtype=torch.float32, device=device)
        self._partials = {}
        self._cascade_bufs = {}
        self._state_dev = device

    def _check_tuning(self):
        gen = int(_lib.load().odpd_tuning_generation())
        if gen != self._tuning_gen:
            self._partials, self._cascade_bufs, self._fused_ok = {}, {}, {}
            self._tuning_gen = gen

    def _mode(self):
        """Descriptor flags that take part in the kernel selection (quantised models: ODPD_FLAG_EVAL follows the module's mode)."""
        if hasattr(self.backbone, "sync_mode"):
            self.backbone.sync_mode()
        return int(self.backbone.desc.flags) & _lib.FLAG_EVAL

    def partials(self, B, T, device):
        self._ensure(device)
        key = (B, T, "fused", self._mode())
        if key not in self._partials:
            lib = _lib.load()
            rows = int(lib.odpd_partial_rows(C.byref(self.backbone.desc), B, T, 1))
            _lib.check(0 if rows > 0 else rows, "odpd_partial_rows")
            self._partials[key] = torch.empty(rows, self.backbone.n_flat + _lib.LOSS_COLS, dtype=torch.float32, device=device)
        return self._partials[key]

    def train_workspace(self, B, T, device):
        """Scratch of the fused kernel (HBM BPTT checkpoints of the large-batch path; None when not needed)."""
        self._ensure(device)
        key = (B, T, "ws", self._mode())
        if key not in self._partials:
            lib = _lib.load()
            n = int(lib.odpd_train_workspace_floats(C.byref(self.backbone.desc), B, T))
            _lib.check(0 if n >= 0 else n, "odpd_train_workspace_floats")
            self._partials[key] = torch.empty(n, dtype=torch.float32, device=device) if n > 0 else None
        if self._partials[key] is None and (self.backbone.dx_needs_flag or getattr(self.backbone, "fused_stats", False)):
            # delta backbones: the fused step's `workspace` argument carries the four sparsity counters of its forward pass (double[4];
            # None while the module's statistics are switched off) — include/opendpd_hip.h, odpd_train_fwd_bwd
            return self.backbone._stats_buffer(device)
        return self._partials[key]

    def has_fused(self, B, T):
        """True when the backbone has a single-launch fwd+loss+bwd kernel for this batch shape."""
        key = (B, T, "has_fused")
        self._check_tuning()
        if self.backbone.dx_needs_flag:      # delta backbones: the trained model is never asked for dL/dx by this optimiser — a flag left behind by an
            self.backbone.desc.flags &= ~_lib.FLAG_NEED_DX      # autograd call on the same module would route it to the dL/dx kernels
        key = key + (self._mode(),)      # quantised models: the module's train / eval mode selects the kernel too (ODPD_FLAG_EVAL)
        if key not in self._fused_ok:
            lib = _lib.load()
            self._fused_ok[key] = int(lib.odpd_partial_rows(C.byref(self.backbone.desc), B, T, 1)) > 0
        return self._fused_ok[key]

    def bwd_partials(self, B, T, device):
        """Partial-gradient rows of the split backward kernel (cascade path)."""
        self._ensure(device)
        key = (B, T, "bwd")
        if key not in self._partials:
            lib = _lib.load()
            rows = int(lib.odpd_partial_rows(C.byref(self.backbone.desc), B, T, 0))
            _lib.check(0 if rows > 0 else rows, "odpd_partial_rows")
            self._partials[key] = torch.empty(rows, self.backbone.n_flat + _lib.LOSS_COLS, dtype=torch.float32,
                                              device=device)
        return self._partials[key]

    def cascade_one_launch(self, B, T, device):
        """Partial-gradient rows of the one-launch cascade step (odpd_cascade_fwd_bwd: DPD wave + frozen-PA wave per frame), or None
        where the pair of models / the batch shape is not served by it."""
        self._ensure(device)
        key = (B, T, "casc", int(self.backbone.desc.flags))     # (a quantised DPD in eval() is not served by the one-launch step)
        if key not in self._partials:
            rows = -1
            if self.pa is not None and self.pa.native and self.backbone.native:
                rows = int(_lib.load().odpd_cascade_rows(C.byref(self.backbone.desc), C.byref(self.pa.desc), B, T))
            self._partials[key] = (torch.empty(rows, self.backbone.n_flat + _lib.LOSS_COLS, dtype=torch.float32, device=device)
                                   if rows > 0 else None)
        return self._partials[key]

    def cascade_buffers(self, B, T, device):
        """u = DPD(x), y = PA(u), dy, du, checkpoints, loss scratch — allocated once per batch shape."""
        self._ensure(device)
        if (B, T) not in self._cascade_bufs:
            lib = _lib.load()
            def ck(bb):
                n = int(lib.odpd_ckpt_floats(C.byref(bb.desc), B, T))
                _lib.check(0 if n >= 0 else n, "odpd_ckpt_floats")
                return torch.empty(max(n, 1), dtype=torch.float32, device=device)
            mk = lambda: torch.empty(B, T, 2, dtype=torch.float32, device=device)
            d = dict(u=mk(), dy=mk(), ck_d=ck(self.backbone),
                     loss=torch.zeros(_lib.LOSS_WS, dtype=torch.float32, device=device))
            if self.pa is not None:
                d.update(y=mk(), du=mk(), ck_p=ck(self.pa), pa_part=None, loss_rows=None)
                rows = int(lib.odpd_frozen_loss_rows(C.byref(self.pa.desc), B, T))     # > 0: forward + loss + dL/du in one launch
                if rows > 0:
                    d["loss_rows"] = torch.empty(rows, _lib.LOSS_COLS, dtype=torch.float32, device=device)
                    d["loss4"] = torch.zeros(_lib.LOSS_COLS, dtype=torch.float32, device=device)
                if self.pa.dx_needs_flag:   # the delta backward kernels write their weight gradients in every launch: scratch
                    rows = int(lib.odpd_partial_rows(C.byref(self.pa.desc), B, T, 0))
                    _lib.check(0 if rows > 0 else rows, "odpd_partial_rows")
                    d["pa_part"] = torch.empty(rows, self.pa.n_flat + _lib.LOSS_COLS, dtype=torch.float32, device=device)
            self._cascade_bufs[(B, T)] = d
        return self._cascade_bufs[(B, T)]

    def zero_grad(self, set_to_none=True):
        for p in self.net.parameters():
            p.grad = None

    def world_size(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    def native_comm(self):
        """The library-owned RCCL communicator (dist.native_comm) when the default group runs on RCCL, else None."""
        if self.process_group is not None:
            return None
        from .dist import native_comm
        return native_comm(self.backbone.flat_params().device)

    def allreduce_grad(self):
        """The ONE collective of the data-parallel step: sum of P+4 floats (gradient + loss partial) — enqueued by the library on
        the step's stream when it owns an RCCL communicator, through torch.distributed otherwise (gloo groups of the CPU / one-GPU tests)."""
        comm = self.native_comm()
        if comm is not None:
            comm.allreduce_sum_(self.grad)
            return
        from .dist import allreduce_sum_
        allreduce_sum_(self.grad, self.process_group)

    def _shard_sizes(self, loader):
        """this rank's share of the epoch's global batch sizes (full batches and tail) under the native data-parallel loop"""
        comm = self.native_comm()
        sizes = set()
        lo, hi = C.c_int64(), C.c_int64()
        for gb in self._epoch_batches(loader):
            _lib.load().odpd_shard_range(gb, comm.rank, comm.world, C.byref(lo), C.byref(hi))
            if hi.value > lo.value:
                sizes.add(hi.value - lo.value)
        return sizes

    def empty_step(self, max_norm, count):
        """A rank whose shard of the global batch is empty: zero gradient into the all-reduce, then the common update."""
        dev = self.backbone.flat_params().device
        self._ensure(dev)
        self.grad.zero_()
        self.reduce_and_apply(max_norm)
        return self.grad[self.backbone.n_flat] / count

    def reduce_and_apply(self, max_norm, stream=None):
        """The tail of a data-parallel step: all-reduce of self.grad (this rank's gradient + loss partial sum) over the ranks, then
        clip + optimiser.  With a library-owned communicator both go to the library in ONE call (odpd_clip_optim_step_dp: the one-shot
        exchange is the optimiser kernel's prologue; RCCL's all-reduce is enqueued in front of it); else torch.distributed, then apply()."""
        comm = self.native_comm()
        if comm is None:
            self.allreduce_grad()
        self.apply(max_norm, stream, comm=comm)

    def apply(self, max_norm, stream=None, comm=None):
        """clip (max_norm, 0 = off) + AdamW on the flat buffers; self.grad must hold the global gradient — or, with `comm` (a
        dist.NativeComm), this rank's share of it: the library then sums over the ranks first."""
        lib = _lib.load()
        g = self.param_groups[0]
        self.step_count += 1
        flat = self.backbone.flat_params()
        if stream is None:
            stream = _lib.stream_ptr()
        if comm is not None:
            frozen = getattr(self.backbone, "frozen_mask", None)
            if frozen is not None and (frozen.device != flat.device or frozen.dtype != torch.uint8):
                self.backbone.frozen_mask = frozen = frozen.to(device=flat.device, dtype=torch.uint8).contiguous()
            adamw = self.kind == "adamw"
            rc = lib.odpd_clip_optim_step_dp(stream, comm.handle, -1 if adamw else _lib.OPTIMIZER_IDS[self.kind], self.backbone.n_flat,
                                             _lib.ptr(flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                             self.step_count, float(g["lr"]), float(g["betas"][0]) if adamw else 0.0,
                                             float(g["betas"][1]) if adamw else 0.0, float(g["eps"]) if adamw else 0.0,
                                             float(g["weight_decay"]) if adamw else 0.0, float(max_norm or 0.0), _lib.ptr(self.norm),
                                             _lib.ptr(frozen) if frozen is not None else None)
            if rc:
                _lib.check(rc, "odpd_clip_optim_step_dp")
            return
        frozen = getattr(self.backbone, "frozen_mask", None)     # parameters torch.optim.AdamW would skip (grad is None)
        if frozen is not None and (frozen.device != flat.device or frozen.dtype != torch.uint8):
            # one byte per parameter, resident next to the parameters: the kernel skips those columns (no host sync, no extra launch)
            self.backbone.frozen_mask = frozen = frozen.to(device=flat.device, dtype=torch.uint8).contiguous()
        if self.kind != "adamw":            # project.py:274-297's other optimisers, with the hyper-parameters the reference builds them with
            # (those hyper-parameters are constants of the kernel: an edited param_group other than `lr` would be ignored — say so)
            fixed = {"adam": dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0), "sgd": dict(momentum=0.9, weight_decay=0.0),
                     "rmsprop": dict(alpha=0.99, eps=1e-8, weight_decay=0.0)}[self.kind]
            for key, val in fixed.items():
                if key in g and g[key] != val:
                    raise NotImplementedError(f"Fused{self.kind.upper()}: param_groups[0]['{key}'] = {g[key]!r}, but the fused step is built with the "
                                              f"reference's {key} = {val!r} (project.py:274-297); only 'lr' can be changed")
            rc = lib.odpd_clip_optim_step(stream, _lib.OPTIMIZER_IDS[self.kind], self.backbone.n_flat, _lib.ptr(flat), _lib.ptr(self.grad),
                                          _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.step_count, float(g["lr"]),
                                          float(max_norm or 0.0), _lib.ptr(self.norm), _lib.ptr(frozen) if frozen is not None else None)
            if rc:
                _lib.check(rc, "odpd_clip_optim_step")
            return
        rc = lib.odpd_clip_adamw_step_masked(stream, self.backbone.n_flat, _lib.ptr(flat), _lib.ptr(self.grad),
                                             _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.step_count,
                                             float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                             float(g["weight_decay"]), float(max_norm or 0.0), _lib.ptr(self.norm),
                                             _lib.ptr(frozen) if frozen is not None else None)
        if rc:
            _lib.check(rc, "odpd_clip_adamw_step")

    def can_run_epoch(self, loader):
        """True when odpd_train_epoch can drive a whole epoch: single fused backbone, one process, resident streams."""
        if not (self.pa is None and getattr(self.backbone, "frozen_mask", None) is None
                and all(hasattr(loader, k) for k in ("epoch_order", "x", "y", "frame_length", "stride", "batch_size")) and loader.x.is_cuda):
            return False
        if loader.x.dtype == torch.bfloat16 and not self.reads_bf16_frames():
            return False
        if self.world_size() > 1 or self.native_comm() is not None:
            # sharded epoch from C++ (odpd_train_epoch_dp): needs the library-owned RCCL communicator and fused kernels for this rank's shards
            if self.native_comm() is None:
                return False
            sizes = self._shard_sizes(loader)
            return self._one_workspace_meaning(sizes, loader.frame_length) and all(
                self.has_fused(b, loader.frame_length) and self.reads_frames(b, loader.frame_length) for b in sizes)
        sizes = self._epoch_batches(loader)
        return self._one_workspace_meaning(sizes, loader.frame_length) and all(
            self.has_fused(b, loader.frame_length) and self.reads_frames(b, loader.frame_length) for b in sizes)

    def _one_workspace_meaning(self, sizes, T):
        """The `workspace` argument of the epoch entry points is ONE pointer for every batch of the epoch.  For backbones with sparsity
        counters it carries the counters when a batch runs on a one-frame-per-wave kernel (workspace floats = 0) and checkpoint scratch when
        it runs on a 16-sequences-per-wave kernel (> 0): an epoch whose full and tail batches fall on different sides cannot be served by one
        pointer — the tail's counters would land in the scratch — and goes through the per-step path instead."""
        if not (self.backbone.dx_needs_flag or getattr(self.backbone, "fused_stats", False)):
            return True
        ws = [int(_lib.load().odpd_train_workspace_floats(C.byref(self.backbone.desc), b, T)) for b in sizes]
        return not (any(w > 0 for w in ws) and any(w == 0 for w in ws))

    @staticmethod
    def _epoch_batches(loader):
        """the batch sizes of one epoch: the full batches and the tail"""
        B = min(loader.batch_size, loader.n)
        return {B, loader.n - (loader.n - 1) // B * B}

    def reads_frames(self, B, T):
        """True when the backbone's fused kernel for this batch shape addresses frames inside resident streams
        (odpd_framed_train_supported_shape)."""
        return bool(_lib.load().odpd_framed_train_supported_shape(C.byref(self.backbone.desc), B, T))

    def reads_bf16_frames(self):
        """bf16 sample storage is read in place by the float GRU family's fused train kernels only"""
        return self.pa is None and self.backbone.backbone_name in ("gru", "dgru", "qgru", "qgru_amp1") and self.backbone.desc.bits_w == 0

    def can_run_split_epoch(self, loader):
        """True when odpd_train_epoch_split can drive a whole epoch: a single backbone WITHOUT a frame-reading fused kernel for the
        epoch's batch shapes, one process, resident streams (the forward / loss / backward / reduce / optimiser chain per step then
        runs from C++ instead of from Python)."""
        return (self.pa is None and self.world_size() == 1 and not self.can_run_epoch(loader)
                and all(hasattr(loader, k) for k in ("epoch_order", "x", "y", "frame_length", "stride", "batch_size"))
                and loader.x.is_cuda and loader.x.dtype == torch.float32 and hasattr(self.backbone, "desc"))

    def train_epoch_split(self, loader, loss_kind, max_norm):
        """One epoch through odpd_train_epoch_split: returns the per-batch mean losses (device tensor)."""
        lib = _lib.load()
        bb = self.backbone
        dev, T, n = loader.x.device, loader.frame_length, loader.n
        B = min(loader.batch_size, n)
        self._ensure(dev)
        n_steps = (n + B - 1) // B
        last = n - (n_steps - 1) * B
        if bb.dx_needs_flag:               # as _cascade_train_step: the trained model is never asked for dL/dx here
            bb.desc.flags &= ~_lib.FLAG_NEED_DX
        if hasattr(bb, "sync_mode"):
            bb.sync_mode()
        key = (B, T, last, "split-epoch")
        if key not in self._partials:
            rows = [int(lib.odpd_partial_rows(C.byref(bb.desc), b, T, 0)) for b in {B, last}]
            _lib.check(0 if min(rows) > 0 else min(rows), "odpd_partial_rows")
            mk = lambda: torch.empty(B, T, 2, dtype=torch.float32, device=dev)
            self._partials[key] = (torch.empty(max(rows), bb.n_flat + _lib.LOSS_COLS, dtype=torch.float32, device=dev), mk(), mk())
        part, xbuf, tbuf = self._partials[key]
        buf = self.cascade_buffers(B, T, dev)
        order = loader.epoch_order()
        losses = torch.empty(n_steps, dtype=torch.float32, device=dev)
        fr = _lib.Frames(loader.x.data_ptr(), loader.y.data_ptr(), order.data_ptr(), n, T, loader.stride, _sample_format(loader.x, loader.y), 0)
        g = self.param_groups[0]
        flat = bb.flat_params(full_check=True)
        frozen = getattr(bb, "frozen_mask", None)
        if frozen is not None and (frozen.device != flat.device or frozen.dtype != torch.uint8):
            bb.frozen_mask = frozen = frozen.to(device=flat.device, dtype=torch.uint8).contiguous()
        adamw = self.kind == "adamw"
        rc = lib.odpd_train_epoch_split(_lib.stream_ptr(), C.byref(bb.desc), _lib.LOSS_IDS[loss_kind], C.byref(fr), B,
                                        -1 if adamw else _lib.OPTIMIZER_IDS[self.kind], _lib.ptr(flat), _lib.ptr(self.grad),
                                        _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.step_count + 1, float(g["lr"]),
                                        float(g["betas"][0]) if adamw else 0.0, float(g["betas"][1]) if adamw else 0.0,
                                        float(g["eps"]) if adamw else 0.0, float(g["weight_decay"]) if adamw else 0.0,
                                        float(max_norm or 0.0), _lib.ptr(frozen) if frozen is not None else None,
                                        _lib.ptr(xbuf), _lib.ptr(tbuf), _lib.ptr(buf["u"]), _lib.ptr(buf["dy"]), _lib.ptr(buf["ck_d"]),
                                        _lib.ptr(part), _lib.ptr(buf["loss"]), _lib.ptr(bb._stats_buffer(dev)), _lib.ptr(losses))
        _lib.check(rc, "odpd_train_epoch_split")
        self.step_count += n_steps
        self._keepalive = order     # the launches read `order` asynchronously
        return losses

    def can_run_cascade_epoch(self, loader):
        """True when odpd_train_epoch_cascade can drive a whole train_dpd epoch: frozen PA behind the trained DPD, every batch of the epoch
        served by the one-launch cascade step, one process, resident streams."""
        if not (self.pa is not None
                and all(hasattr(loader, k) for k in ("epoch_order", "x", "y", "frame_length", "stride", "batch_size")) and loader.x.is_cuda
                and loader.x.dtype == torch.float32):
            return False
        for bb in (self.backbone, self.pa):       # quantised models: the descriptor's ODPD_FLAG_EVAL must say what the module says NOW
            if hasattr(bb, "sync_mode"):          # (an evaluation pass leaves it set; the one-launch step serves train mode only)
                bb.sync_mode()
        dev = loader.x.device
        if self.world_size() > 1 or self.native_comm() is not None:
            # sharded epoch from C++: needs the library-owned RCCL communicator and the one-launch step for this rank's shards
            return self.native_comm() is not None and all(self.cascade_one_launch(b, loader.frame_length, dev) is not None
                                                          for b in self._shard_sizes(loader))
        return all(self.cascade_one_launch(b, loader.frame_length, dev) is not None for b in self._epoch_batches(loader))

    def train_epoch_cascade(self, loader, loss_kind, max_norm):
        """One train_dpd epoch through odpd_train_epoch_cascade: returns the per-batch mean losses (device tensor)."""
        lib = _lib.load()
        dpd, pa = self.backbone, self.pa
        dev, T, n = loader.x.device, loader.frame_length, loader.n
        B = min(loader.batch_size, n)
        self._ensure(dev)
        n_steps = (n + B - 1) // B
        comm = self.native_comm()
        sizes = self._shard_sizes(loader) if comm is not None else self._epoch_batches(loader)
        if sizes:
            part = max((self.cascade_one_launch(b, T, dev) for b in sizes), key=lambda p: p.shape[0])
        else:       # a rank all of whose shards are empty (global batch smaller than the world): it still joins every all-reduce
            part = torch.empty(1, dpd.n_flat + _lib.LOSS_COLS, dtype=torch.float32, device=dev)
        order = loader.epoch_order()
        losses = torch.empty(n_steps, dtype=torch.float32, device=dev)
        fr = _lib.Frames(loader.x.data_ptr(), loader.y.data_ptr(), order.data_ptr(), n, T, loader.stride, _sample_format(loader.x, loader.y), 0)
        g = self.param_groups[0]
        adamw = self.kind == "adamw"
        flat = dpd.flat_params(full_check=True)
        frozen = getattr(dpd, "frozen_mask", None)     # parameters torch.optim.AdamW would skip (grad is None): the QAT models' 16-bit output scales
        if frozen is not None and (frozen.device != flat.device or frozen.dtype != torch.uint8):
            dpd.frozen_mask = frozen = frozen.to(device=flat.device, dtype=torch.uint8).contiguous()
        for bb in (dpd, pa):       # quantised models: train / eval mode of the module -> ODPD_FLAG_EVAL
            if hasattr(bb, "sync_mode"):
                bb.sync_mode()
        rc = lib.odpd_train_epoch_cascade(_lib.stream_ptr(), comm.handle if comm is not None else None, C.byref(dpd.desc), C.byref(pa.desc), _lib.LOSS_IDS[loss_kind], C.byref(fr), B,
                                          -1 if adamw else _lib.OPTIMIZER_IDS[self.kind], _lib.ptr(dpd.flat_params(full_check=True)),
                                          _lib.ptr(pa.flat_params(full_check=True)), _lib.ptr(self.grad), _lib.ptr(self.exp_avg),
                                          _lib.ptr(self.exp_avg_sq), self.step_count + 1, float(g["lr"]), float(g["betas"][0]) if adamw else 0.0,
                                          float(g["betas"][1]) if adamw else 0.0, float(g["eps"]) if adamw else 0.0,
                                          float(g["weight_decay"]) if adamw else 0.0, float(max_norm or 0.0),
                                          _lib.ptr(frozen) if frozen is not None else None, _lib.ptr(part),
                                          _lib.ptr(dpd._stats_buffer(dev)), _lib.ptr(losses))
        _lib.check(rc, "odpd_train_epoch_cascade")
        self.step_count += n_steps
        self._keepalive = order     # the launches read `order` asynchronously
        return losses

    def train_epoch(self, loader, loss_kind, max_norm):
        """One epoch through the native loop (odpd_train_epoch): returns the per-batch mean losses (device tensor)."""
        lib = _lib.load()
        dev, T, n = loader.x.device, loader.frame_length, loader.n
        B = min(loader.batch_size, n)
        self._ensure(dev)
        n_steps = (n + B - 1) // B
        last = n - (n_steps - 1) * B
        comm = self.native_comm()
        key = (B, T, last, "epoch", comm is not None, self._mode())
        if key not in self._partials:
            sizes = self._shard_sizes(loader) if comm is not None else {B, last}
            rows = [int(lib.odpd_partial_rows(C.byref(self.backbone.desc), b, T, 1)) for b in sizes] or [1]
            _lib.check(0 if min(rows) > 0 else min(rows), "odpd_partial_rows")
            ws = max([int(lib.odpd_train_workspace_floats(C.byref(self.backbone.desc), b, T)) for b in sizes] or [0])
            self._partials[key] = (torch.empty(max(rows), self.backbone.n_flat + _lib.LOSS_COLS, dtype=torch.float32, device=dev),
                                   torch.empty(ws, dtype=torch.float32, device=dev) if ws > 0 else None)
        part, ws = self._partials[key]
        if ws is None and (self.backbone.dx_needs_flag or getattr(self.backbone, "fused_stats", False)):      # delta backbones: `workspace` = the sparsity counters (see train_workspace)
            ws = self.backbone._stats_buffer(dev)
        order = loader.epoch_order()
        losses = torch.empty(n_steps, dtype=torch.float32, device=dev)
        fr = _lib.Frames(loader.x.data_ptr(), loader.y.data_ptr(), order.data_ptr(), n, T, loader.stride, _sample_format(loader.x, loader.y), 0)
        g = self.param_groups[0]
        flat = self.backbone.flat_params(full_check=True)
        if comm is not None:
            adamw = self.kind == "adamw"
            rc = lib.odpd_train_epoch_dp(_lib.stream_ptr(), comm.handle, C.byref(self.backbone.desc), _lib.LOSS_IDS[loss_kind], C.byref(fr), B,
                                         -1 if adamw else _lib.OPTIMIZER_IDS[self.kind], _lib.ptr(flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg),
                                         _lib.ptr(self.exp_avg_sq), self.step_count + 1, float(g["lr"]), float(g["betas"][0]) if adamw else 0.0,
                                         float(g["betas"][1]) if adamw else 0.0, float(g["eps"]) if adamw else 0.0,
                                         float(g["weight_decay"]) if adamw else 0.0, float(max_norm or 0.0), _lib.ptr(part), _lib.ptr(ws),
                                         _lib.ptr(losses))
        elif self.kind == "adamw":
            rc = lib.odpd_train_epoch(_lib.stream_ptr(), C.byref(self.backbone.desc), _lib.LOSS_IDS[loss_kind], C.byref(fr), B,
                                      _lib.ptr(flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                      self.step_count + 1, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                      float(g["eps"]), float(g["weight_decay"]), float(max_norm or 0.0), _lib.ptr(part),
                                      _lib.ptr(ws), _lib.ptr(losses))
        else:
            rc = lib.odpd_train_epoch_opt(_lib.stream_ptr(), C.byref(self.backbone.desc), _lib.LOSS_IDS[loss_kind], C.byref(fr), B,
                                          _lib.OPTIMIZER_IDS[self.kind], _lib.ptr(flat), _lib.ptr(self.grad), _lib.ptr(self.exp_avg),
                                          _lib.ptr(self.exp_avg_sq), self.step_count + 1, float(g["lr"]), float(max_norm or 0.0),
                                          _lib.ptr(part), _lib.ptr(ws), _lib.ptr(losses))
        _lib.check(rc, "odpd_train_epoch")
        self.step_count += n_steps
        self._keepalive = order     # the launches read `order` asynchronously
        return losses

    def step(self, max_norm=0.0):
        """Generic-path step: gathers p.grad (set by autograd) into the flat gradient, then apply()."""
        ps = list(self.trained.parameters())
        self._ensure(ps[0].device)
        with torch.no_grad():
            torch.cat([p.grad.reshape(-1) for p in ps], out=self.grad[:self.backbone.n_flat])
        self.reduce_and_apply(max_norm)


class FrameBatch:
    """A batch given as frames of resident (N,2) streams: frame f = samples [f*stride, f*stride + T) (IQFrameDataset,
    data_collector.py:239-247) — nothing is materialised, the fused kernels address the streams directly."""

    def __init__(self, x_stream, y_stream, order, frame_length, stride=1):
        self.x, self.y, self.order, self.T, self.stride = x_stream.contiguous(), y_stream.contiguous(), order.contiguous(), frame_length, stride
        assert self.x.is_cuda and self.order.dtype == torch.int64
        self.desc = _lib.Frames(self.x.data_ptr(), self.y.data_ptr(), self.order.data_ptr(), self.order.numel(), frame_length, stride,
                                _sample_format(self.x, self.y), 0)
        self.shape = (self.order.numel(), frame_length, 2)
        self.device = self.x.device



class FusedAdam(FusedAdamW):
    """torch.optim.Adam(lr) as project.py:278-279 builds it (no weight decay) on the same fused step"""
    kind = "adam"

    def __init__(self, net, lr=5e-4, process_group=None):
        super().__init__(net, lr=lr, weight_decay=0.0, process_group=process_group)


class FusedSGD(FusedAdamW):
    """torch.optim.SGD(lr, momentum=0.9) as project.py:276-277 builds it; `exp_avg` holds the momentum buffer"""
    kind = "sgd"

    def __init__(self, net, lr=5e-4, process_group=None):
        super().__init__(net, lr=lr, weight_decay=0.0, process_group=process_group)
        self.param_groups[0].update(momentum=0.9)


class FusedRMSprop(FusedAdamW):
    """torch.optim.RMSprop(lr) as project.py:287-288 builds it (alpha 0.99, eps 1e-8); `exp_avg_sq` holds the square average"""
    kind = "rmsprop"

    def __init__(self, net, lr=5e-4, process_group=None):
        super().__init__(net, lr=lr, weight_decay=0.0, process_group=process_group)
        self.param_groups[0].update(alpha=0.99)


def fused_train_step(opt, x, target, loss_kind="l2", grad_clip_val=0.0, global_count=None, timing=None):
    """One optimiser step (train_funcs.py:33-44) on device tensors x, target of shape (B,T,2) — or on a `FrameBatch`
    passed as `x` (target ignored).  Returns the loss as a 0-dim device tensor (no host sync).  `global_count` =
    number of target elements of the GLOBAL batch when x is this rank's shard (default: this batch)."""
    lib = _lib.load()
    bb = opt.backbone
    B, T = x.shape[0], x.shape[1]
    n = B * T * 2
    count = int(global_count or n)
    framed = isinstance(x, FrameBatch)
    if framed and (opt.pa is not None or not opt.has_fused(B, T) or not opt.reads_frames(B, T)):
        raise RuntimeError("FrameBatch input needs a single backbone whose fused kernel for this batch shape reads frames from streams "
                           "(GRU family, gmp, rvtdcnn; lstm / vdlstm / pgjanet at the reference's batch sizes)")
    if opt.pa is not None or not opt.has_fused(B, T):
        return _cascade_train_step(opt, x, target, loss_kind, grad_clip_val, count, timing if isinstance(timing, list) else None)
    part = opt.partials(B, T, x.device)
    ws = opt.train_workspace(B, T, x.device)
    flat = bb.flat_params()
    st = _lib.stream_ptr()
    if hasattr(bb, "sync_mode"):      # quantised models: train / eval mode of the module -> ODPD_FLAG_EVAL
        bb.sync_mode()
    if timing is not None and not isinstance(timing, list):
        timing[0].record()
    if framed:
        rc = lib.odpd_train_fwd_bwd_framed(st, C.byref(bb.desc), _lib.LOSS_IDS[loss_kind], C.byref(x.desc), 0, B, count,
                                           _lib.ptr(flat), _lib.ptr(part), _lib.ptr(ws))
    else:
        rc = lib.odpd_train_fwd_bwd(st, C.byref(bb.desc), _lib.LOSS_IDS[loss_kind], B, T, count,
                                    _lib.ptr(flat), _lib.ptr(x), _lib.ptr(target), _lib.ptr(part), _lib.ptr(ws))
    if timing is not None and not isinstance(timing, list):
        timing[1].record()
    if rc:
        _lib.check(rc, f"odpd_train_fwd_bwd[{bb.backbone_name}]")
    rc = lib.odpd_reduce_partials(st, part.shape[0], bb.n_flat, _lib.ptr(part), _lib.ptr(opt.grad), 0)
    if rc:
        _lib.check(rc, "odpd_reduce_partials")
    opt.reduce_and_apply(grad_clip_val, st)
    return opt.grad[bb.n_flat] / count      # column P = sum of squared / absolute errors (summed over the ranks with the gradient)


def _cascade_train_step(opt, x, target, loss_kind, grad_clip_val, count, timing=None):
    """Split-kernel step.  `timing` (a list): (name, start event, end event) of every launch group is appended (bench.py prices the
    cascade's kernels one by one with it).  With a frozen PA (train_dpd, steps/train_dpd.py:60-63, models.py:173-176):
    y = PA(DPD(x)) chained on the stream — DPD fwd, then the frozen PA's forward + loss + dL/du as ONE launch where
    odpd_frozen_loss_dx serves the PA (GRU family), else PA fwd, loss, PA bwd (dL/du only); then DPD bwd.  Without a PA (backbones that have no fused kernel yet): fwd, loss, bwd."""
    lib = _lib.load()
    dpd, pa = opt.backbone, opt.pa
    B, T = x.shape[0], x.shape[1]
    # delta backbones: buffers and kernels are selected through the descriptor — pin the selection against a flag left behind
    # by an autograd call on the same module (the trained model is never asked for dL/dx here, a frozen delta PA always is)
    if dpd.dx_needs_flag:
        dpd.desc.flags &= ~_lib.FLAG_NEED_DX
    if pa is not None and pa.dx_needs_flag:
        pa.desc.flags |= _lib.FLAG_NEED_DX
    for bb in (dpd, pa):       # quantised models: train / eval mode of the module -> ODPD_FLAG_EVAL (an evaluation pass may have left it set)
        if hasattr(bb, "sync_mode"):
            bb.sync_mode()
    st = _lib.stream_ptr()
    fd, fp = dpd.flat_params(), (pa.flat_params() if pa is not None else None)

    def mark(name=None, _open=[]):
        """open / close a timed span (no-op without `timing`)"""
        if timing is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if _open:
            timing.append((_open.pop(), _open.pop(), ev))
        if name is not None:
            _open.extend([ev, name])

    one = opt.cascade_one_launch(B, T, x.device)
    if one is not None:
        # the reference's own batch sizes, GRU-family DPD and PA: the whole step body as one launch (csrc/gru_cascade.hip)
        mark("cascade_fwd_loss_bwd")
        _lib.check(lib.odpd_cascade_fwd_bwd(st, C.byref(dpd.desc), C.byref(pa.desc), _lib.LOSS_IDS[loss_kind], B, T, count, _lib.ptr(fd),
                                            _lib.ptr(fp), _lib.ptr(x), _lib.ptr(target), None, 0, _lib.ptr(one), _lib.ptr(dpd._stats_buffer(x.device))),
                   "cascade fwd + loss + bwd")
        mark("reduce_clip_optimiser")
        # (column P of the rows = loss partial sums: the reduced loss travels with the gradient through the one all-reduce)
        _lib.check(lib.odpd_reduce_partials(st, one.shape[0], dpd.n_flat, _lib.ptr(one), _lib.ptr(opt.grad), 0), "reduce")
        opt.reduce_and_apply(grad_clip_val)
        loss = opt.grad[dpd.n_flat] / count
        mark()
        return loss
    buf = opt.cascade_buffers(B, T, x.device)
    part = opt.bwd_partials(B, T, x.device)
    mark("dpd_fwd")
    _lib.check(lib.odpd_backbone_fwd(st, C.byref(dpd.desc), B, T, _lib.ptr(fd), _lib.ptr(x), _lib.ptr(buf["u"]),
                                     _lib.ptr(buf["ck_d"]), _lib.ptr(dpd._stats_buffer(x.device))), "dpd fwd")
    mark("pa_fwd_loss_dx" if pa is not None else "loss")
    loss_sum = None
    if pa is not None and buf["loss_rows"] is not None:
        # frozen PA in front of the loss: forward, loss and dL/du in one launch (16-sequences-per-wave GRU-family kernels)
        _lib.check(lib.odpd_frozen_loss_dx(st, C.byref(pa.desc), _lib.LOSS_IDS[loss_kind], B, T, count, _lib.ptr(fp),
                                           _lib.ptr(buf["u"]), _lib.ptr(target), _lib.ptr(buf["du"]), _lib.ptr(buf["loss_rows"]),
                                           _lib.ptr(buf["ck_p"])), "pa fwd + loss + dL/du")
        _lib.check(lib.odpd_reduce_partials(st, buf["loss_rows"].shape[0], 0, _lib.ptr(buf["loss_rows"]), _lib.ptr(buf["loss4"]), 0),
                   "reduce loss rows")
        loss_sum, du = buf["loss4"][0], buf["du"]
    else:
        if pa is not None:
            _lib.check(lib.odpd_backbone_fwd(st, C.byref(pa.desc), B, T, _lib.ptr(fp), _lib.ptr(buf["u"]), _lib.ptr(buf["y"]),
                                             _lib.ptr(buf["ck_p"]), _lib.ptr(pa._stats_buffer(x.device))), "pa fwd")
        y = buf["y"] if pa is not None else buf["u"]
        _lib.check(lib.odpd_loss_fwd_bwd(st, _lib.LOSS_IDS[loss_kind], B * T * 2, count, _lib.ptr(y), _lib.ptr(target),
                                         _lib.ptr(buf["dy"]), _lib.ptr(buf["loss"])), "loss")
        du = buf["dy"]
        if pa is not None:
            _lib.check(lib.odpd_backbone_bwd(st, C.byref(pa.desc), B, T, _lib.ptr(fp), _lib.ptr(buf["u"]), _lib.ptr(buf["dy"]),
                                             _lib.ptr(buf["ck_p"]), _lib.ptr(buf["pa_part"]), _lib.ptr(buf["du"])), "pa bwd")
            du = buf["du"]
    mark("dpd_bwd")
    _lib.check(lib.odpd_backbone_bwd(st, C.byref(dpd.desc), B, T, _lib.ptr(fd), _lib.ptr(x), _lib.ptr(du),
                                     _lib.ptr(buf["ck_d"]), _lib.ptr(part), None), "dpd bwd")
    mark("reduce_clip_optimiser")
    _lib.check(lib.odpd_reduce_partials(st, part.shape[0], dpd.n_flat, _lib.ptr(part), _lib.ptr(opt.grad), 0), "reduce")
    # loss scalar travels with the gradient (column P) so that one all-reduce covers both:
    # grad[P] = sum of errors of this rank = mean * count
    opt.grad[dpd.n_flat] = loss_sum if loss_sum is not None else buf["loss"][0] * count
    opt.reduce_and_apply(grad_clip_val)
    loss = opt.grad[dpd.n_flat] / count
    mark()
    return loss


def _check_exchange_health(optimizer):
    """Data-parallel runs: a one-shot exchange that timed out on SOME rank poisoned that rank's gradient sum with NaN while its peers
    carried on with a valid one — the replicas have diverged and nothing downstream would notice.  Once per epoch (the epoch's
    loss read-back has synchronised the device already) every rank reads its time-out counter, the ranks agree on the verdict, and ALL of
    them raise together when any of them saw a time-out."""
    comm = optimizer.native_comm() if isinstance(optimizer, FusedAdamW) else None
    if comm is None:
        return
    from .dist import _agree
    mine = comm.errors()
    if not _agree(mine == 0, optimizer.backbone.flat_params().device):
        raise RuntimeError(f"data-parallel gradient exchange timed out during this epoch (this rank: {mine} exchange(s)); a peer was lost or "
                           "stalled — the replicas are no longer identical, the run cannot be continued")


def net_train(log, net, dataloader, optimizer, criterion, grad_clip_val, device):
    """Reference signature (train_funcs.py:16-22)."""
    net = _net_train(log, net, dataloader, optimizer, criterion, grad_clip_val, device)
    _check_exchange_health(optimizer)
    return net


def _net_train(log, net, dataloader, optimizer, criterion, grad_clip_val, device):
    net = net.train()
    losses = []
    kind = _loss_kind(criterion)
    fast = isinstance(optimizer, FusedAdamW) and optimizer.net is net and kind is not None
    if fast and optimizer.can_run_epoch(dataloader):
        # whole epoch in the native loop: frames read in place from the resident streams, 3 launches per step, no Python
        losses = optimizer.train_epoch(dataloader, kind, grad_clip_val)
        optimizer.last_epoch_losses = losses          # per-step mean losses of the epoch (device tensor): the reference only logs their mean
        log["loss"] = float(losses.double().mean().item()) if losses.numel() else float("nan")
        return net
    if fast and optimizer.can_run_cascade_epoch(dataloader):
        # train_dpd at the reference's batch sizes, GRU-family DPD and PA: one launch per step body, the epoch issued from the native loop
        losses = optimizer.train_epoch_cascade(dataloader, kind, grad_clip_val)
        optimizer.last_epoch_losses = losses          # per-step mean losses of the epoch (device tensor): the reference only logs their mean
        log["loss"] = float(losses.double().mean().item()) if losses.numel() else float("nan")
        return net
    if fast and optimizer.can_run_split_epoch(dataloader):
        # no fused kernel for these batch shapes: the split chain of every step, issued from the native loop as well
        losses = optimizer.train_epoch_split(dataloader, kind, grad_clip_val)
        optimizer.last_epoch_losses = losses          # per-step mean losses of the epoch (device tensor): the reference only logs their mean
        log["loss"] = float(losses.double().mean().item()) if losses.numel() else float("nan")
        return net
    world = optimizer.world_size() if isinstance(optimizer, FusedAdamW) else 1
    if world > 1 and not fast:
        raise RuntimeError("data-parallel training needs the fused step (FusedAdamW on this net, mean l1 / l2 loss)")
    rank = 0
    if world > 1:
        import torch.distributed as dist
        rank = dist.get_rank(optimizer.process_group)
    for features, targets in dataloader:
        features = features.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        if fast:
            count = None
            if world > 1:
                # every rank draws the same global batch (same seed, same loader) and keeps its contiguous shard; the loss
                # gradient is normalised by the GLOBAL element count, so the all-reduced sum is the global-batch gradient
                count = features.shape[0] * features.shape[1] * features.shape[2]
                lo, hi = shard_range(features.shape[0], rank, world)
                features, targets = features[lo:hi], targets[lo:hi]
            if features.shape[0] == 0:      # fewer frames than ranks in the last batch: contribute zeros, stay in step
                loss = optimizer.empty_step(grad_clip_val, count)
            else:
                loss = fused_train_step(optimizer, features.contiguous().float(), targets.contiguous().float(), kind,
                                        grad_clip_val, global_count=count)
            losses.append(loss)
            continue
        optimizer.zero_grad()
        out = net(features)
        loss = criterion(out, targets)
        loss.backward()
        if isinstance(optimizer, FusedAdamW):
            optimizer.step(grad_clip_val)
        else:
            if grad_clip_val != 0:
                nn.utils.clip_grad_norm_(net.parameters(), grad_clip_val)
            optimizer.step()
        losses.append(loss.detach())
    # one host sync per epoch instead of one .item() per step (train_funcs.py:48)
    stacked = torch.stack([l.float() for l in losses]) if losses else None
    if fast:
        optimizer.last_epoch_losses = stacked
    log["loss"] = float(np.mean(stacked.cpu().numpy())) if losses else float("nan")
    return net


def net_eval_pair(log_a, log_b, net, loader_a, loader_b, criterion, device):
    """Validation and test evaluation in ONE forward launch per batch pair.  The recurrent kernels are latency-bound on these
    tiny batches (1-3 long segments): running the two splits side by side costs the time of one.  Same results as two
    net_eval calls (the sequences of a batch are independent); falls back to them when the splits do not pair up."""
    batches_a, batches_b = list(loader_a), list(loader_b)
    pairable = len(batches_a) == len(batches_b) and all(fa.shape[1:] == fb.shape[1:] for (fa, _), (fb, _) in zip(batches_a, batches_b))
    if not pairable:
        _, pa, ga = net_eval(log_a, net, batches_a, criterion, device)
        _, pb, gb = net_eval(log_b, net, batches_b, criterion, device)
        return net, (pa, ga), (pb, gb)
    net = net.eval()
    out = ([], [], []), ([], [], [])
    with torch.no_grad():
        for (fa, ta), (fb, tb) in zip(batches_a, batches_b):
            fa, ta, fb, tb = fa.to(device), ta.to(device), fb.to(device), tb.to(device)
            y = net(torch.cat((fa, fb), dim=0))
            for (losses, pred, truth), o, t in ((out[0], y[:fa.shape[0]], ta), (out[1], y[fa.shape[0]:], tb)):
                losses.append(criterion(o, t)); pred.append(o); truth.append(t)
    res = []
    for log, (losses, pred, truth) in ((log_a, out[0]), (log_b, out[1])):
        log["loss"] = float(np.mean(torch.stack([l.float() for l in losses]).cpu().numpy())) if losses else float("nan")
        res.append((torch.cat(pred, dim=0).cpu().numpy(), torch.cat(truth, dim=0).cpu().numpy()))
    return net, res[0], res[1]


def net_eval(log, net, dataloader, criterion, device):
    """Reference signature (train_funcs.py:57-61); returns (net, prediction, ground_truth) as numpy."""
    net = net.eval()
    losses, prediction, ground_truth = [], [], []
    with torch.no_grad():
        for features, targets in dataloader:
            features = features.to(device)
            targets = targets.to(device)
            outputs = net(features)
            losses.append(criterion(outputs, targets))
            prediction.append(outputs)
            ground_truth.append(targets)
    log["loss"] = float(np.mean(torch.stack([l.float() for l in losses]).cpu().numpy())) if losses else float("nan")
    prediction = torch.cat(prediction, dim=0).cpu().numpy()
    ground_truth = torch.cat(ground_truth, dim=0).cpu().numpy()
    return net, prediction, ground_truth
