"""ctypes front-end of the CPU oracle (oracle/odpd_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package (opendpd_amd) never does.  Arrays are numpy; `precision` selects the fp32 or fp64
build of the same C source.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

BACKBONES = {"gru": 0, "dgru": 1, "qgru": 2, "qgru_amp1": 3, "lstm": 4, "vdlstm": 5, "deltagru": 6,
             "deltagru_tcnskip": 7, "tcnn": 8, "pgjanet": 9, "gmp": 10, "rvtdcnn": 11, "neuraltx": 12, "deltajanet": 13, "dvrjanet": 14, "bojanet": 15, "apnrru": 16, "mcldnn": 17}


_HEAD_ONLY = (BACKBONES["lstm"], BACKBONES["vdlstm"], BACKBONES["deltajanet"], BACKBONES["neuraltx"], BACKBONES["rvtdcnn"], BACKBONES["pgjanet"])      # --quant swaps only their nn.Linear heads (quant_envs.py:40-60)


class Model(C.Structure):
    """Mirror of odpd_model_t (include/opendpd_hip.h)."""
    _fields_ = [("backbone", C.c_int32), ("hidden", C.c_int32), ("thx", C.c_float), ("thh", C.c_float),
                ("bits_w", C.c_int32), ("bits_a", C.c_int32), ("flags", C.c_int32)]


def make_model(backbone, hidden, thx=0.0, thh=0.0, bits_w=0, bits_a=0):
    return Model(BACKBONES[backbone], int(hidden), float(thx), float(thh), int(bits_w), int(bits_a), 0)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not all(os.path.exists(os.path.join(_BUILD, f"liboracle_{p}.so")) for p in ("f32", "f64")):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))


class Oracle:
    def __init__(self, precision="f32"):
        build()
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.lib = C.CDLL(os.path.join(_BUILD, f"liboracle_{precision}.so"))
        L = self.lib
        L.oracle_param_count.restype = C.c_int64
        L.oracle_loss_fwd_bwd.restype = C.c_double
        L.oracle_clip_adamw_step.restype = C.c_double
        L.oracle_train_step.restype = C.c_double
        assert L.oracle_real_bytes() == np.dtype(self.dtype).itemsize

    def _a(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p) if a is not None else None

    def param_count(self, m):
        return int(self.lib.oracle_param_count(C.byref(m)))

    def max_threads(self):
        return int(self.lib.oracle_max_threads())

    def set_threads(self, n):
        self.lib.oracle_set_threads(int(n))

    def forward(self, m, params, x, stats=None):
        x = self._a(x)
        params = self._a(params)
        B, T = x.shape[0], x.shape[1]
        y = np.empty_like(x)
        st = np.zeros(4, dtype=np.float64) if stats is None else stats
        rc = self.lib.oracle_backbone_fwd(C.byref(m), B, T, self._p(params), self._p(x), self._p(y), self._p(st))
        if rc:
            raise RuntimeError(f"oracle_backbone_fwd failed rc={rc}")
        return (y, st) if stats is None else y

    def backward(self, m, params, x, dy, need_dx=True):
        x = self._a(x)
        params = self._a(params)
        dy = self._a(dy)
        B, T = x.shape[0], x.shape[1]
        dp = np.zeros(self.param_count(m), dtype=self.dtype)
        dx = np.zeros_like(x) if need_dx else None
        rc = self.lib.oracle_backbone_bwd(C.byref(m), B, T, self._p(params), self._p(x), self._p(dy), self._p(dp),
                                          self._p(dx))
        if rc:
            raise RuntimeError(f"oracle_backbone_bwd failed rc={rc}")
        return dp, dx

    def qat_forward(self, m, params, x, eval_mode=False, stats=None):
        """Quantised model (m.bits_w > 0; gru, dgru, qgru, qgru_amp1, deltagru_tcnskip): train-mode (float output) or eval-mode
        (16-bit output grid).  `stats` (4 doubles) accumulates the delta cell's sparsity counters."""
        if m.backbone in _HEAD_ONLY:      # lstm / vdlstm / deltajanet: only the nn.Linear heads are quantised; ODPD_FLAG_EVAL = eval mode
            m2 = Model(m.backbone, m.hidden, m.thx, m.thh, m.bits_w, m.bits_a, 1 if eval_mode else 0)
            return self.forward(m2, params, x, stats=np.zeros(4))
        x = self._a(x)
        params = self._a(params)
        y = np.empty_like(x)
        rc = self.lib.oracle_qat_fwd(C.byref(m), x.shape[0], x.shape[1], self._p(params), self._p(x), self._p(y),
                                     1 if eval_mode else 0, self._p(stats))
        if rc:
            raise RuntimeError(f"oracle_qat_fwd failed rc={rc}")
        return y

    def qat_backward(self, m, params, x, dy, need_dx=True):
        if m.backbone in _HEAD_ONLY:
            return self.backward(m, params, x, dy, need_dx)
        x = self._a(x)
        params = self._a(params)
        dy = self._a(dy)
        dp = np.zeros(self.param_count(m), dtype=self.dtype)
        dx = np.zeros_like(x) if need_dx else None
        rc = self.lib.oracle_qat_bwd(C.byref(m), x.shape[0], x.shape[1], self._p(params), self._p(x), self._p(dy),
                                     self._p(dp), self._p(dx))
        if rc:
            raise RuntimeError(f"oracle_qat_bwd failed rc={rc}")
        return dp, dx

    def loss(self, kind, y, target, count=None):
        y = self._a(y)
        target = self._a(target)
        dy = np.empty_like(y)
        n = y.size
        val = self.lib.oracle_loss_fwd_bwd(0 if kind == "l2" else 1, C.c_int64(n), C.c_int64(count or n),
                                           self._p(y), self._p(target), self._p(dy))
        return float(val), dy

    def clip_adamw(self, params, grad, m, v, step, lr, max_norm, tensor_sizes=None, betas=(0.9, 0.999), eps=1e-8,
                   wd=0.01):
        """In place on params/grad/m/v (contiguous arrays of self.dtype). Returns the pre-clip norm."""
        for a in (params, grad, m, v):
            assert a.dtype == self.dtype and a.flags.c_contiguous
        ts = np.ascontiguousarray(tensor_sizes, dtype=np.int64) if tensor_sizes is not None else None
        return float(self.lib.oracle_clip_adamw_step(
            C.c_int64(params.size), self._p(params), self._p(grad), self._p(m), self._p(v), C.c_int64(step),
            C.c_double(lr), C.c_double(betas[0]), C.c_double(betas[1]), C.c_double(eps), C.c_double(wd),
            C.c_double(max_norm), self._p(ts), 0 if ts is None else len(ts)))

    def train_step(self, m, params, x, target, exp_avg, exp_avg_sq, step, lr, max_norm, loss_kind="l2", scratch=None):
        """Whole fwd+loss+bwd+clip+AdamW step in place on params/exp_avg/exp_avg_sq; returns loss."""
        B, T = x.shape[0], x.shape[1]
        if scratch is None:
            scratch = (np.empty_like(x), np.empty_like(x), np.empty(params.size, dtype=self.dtype))
        y, dy, g = scratch
        return float(self.lib.oracle_train_step(
            C.byref(m), 0 if loss_kind == "l2" else 1, B, T, self._p(params), self._p(x), self._p(target),
            self._p(exp_avg), self._p(exp_avg_sq), C.c_int64(step), C.c_double(lr), C.c_double(max_norm),
            self._p(y), self._p(dy), self._p(g)))
