"""ctypes binding of libopendpd_hip.so (include/opendpd_hip.h).  There is NO CPU fallback: if the
library is missing or the tensors are not on a HIP device, calls raise."""
import ctypes as C
import os

import torch

from . import build as _build

BACKBONE_IDS = {"gru": 0, "dgru": 1, "qgru": 2, "qgru_amp1": 3, "lstm": 4, "vdlstm": 5, "deltagru": 6,
                "deltagru_tcnskip": 7, "tcnn": 8, "pgjanet": 9, "gmp": 10, "rvtdcnn": 11, "neuraltx": 12, "deltajanet": 13, "dvrjanet": 14, "bojanet": 15, "apnrru": 16, "mcldnn": 17}
LOSS_IDS = {"l2": 0, "l1": 1}
FLAG_EVAL, FLAG_NEED_DX, FLAG_TWO_LAYERS = 1, 2, 4      # odpd_model_t.flags (include/opendpd_hip.h)
LOSS_COLS = 4        # extra columns of a partials row (column P = loss partial sum)
LOSS_WS = 1 + 256    # floats behind `loss_out` (result + per-block scratch)
ABI_VERSION = 13     # odpd_abi_version() of the library these argument lists belong to


class ModelDesc(C.Structure):
    """odpd_model_t"""
    _fields_ = [("backbone", C.c_int32), ("hidden", C.c_int32), ("thx", C.c_float), ("thh", C.c_float),
                ("bits_w", C.c_int32), ("bits_a", C.c_int32), ("flags", C.c_int32)]


class Frames(C.Structure):
    """odpd_frames_t"""
    _fields_ = [("x_stream", C.c_void_p), ("y_stream", C.c_void_p), ("order", C.c_void_p), ("n_frames", C.c_int64),
                ("frame_length", C.c_int32), ("stride", C.c_int32), ("sample_format", C.c_int32), ("reserved", C.c_int32)]


SAMPLES_F32, SAMPLES_BF16 = 0, 1     # enum odpd_sample_format
SWEEP_S16 = 1                        # flags of odpd_train_epoch_sweep


class SweepRun(C.Structure):
    """odpd_sweep_run_t"""
    _fields_ = [("params", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("partials", C.c_void_p),
                ("losses_out", C.c_void_p), ("y", C.c_void_p), ("workspace", C.c_void_p), ("order", C.c_void_p), ("lr", C.c_double)]


_EXPORTS = {
    # name: (restype, argtypes)
    "odpd_abi_version": (C.c_int, []),
    "odpd_built_arch": (C.c_char_p, []),
    "odpd_probe_issue_ns": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "odpd_set_tuning": (C.c_int, [C.c_char_p, C.c_int64]),
    "odpd_tuning_generation": (C.c_int64, []),
    "odpd_param_count": (C.c_int64, [C.POINTER(ModelDesc)]),
    "odpd_ckpt_floats": (C.c_int64, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_partial_rows": (C.c_int64, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int]),
    "odpd_backbone_fwd": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_backbone_bwd": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_reduce_partials": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]),
    "odpd_loss_fwd_bwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "odpd_train_workspace_floats": (C.c_int64, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_train_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_frozen_loss_rows": (C.c_int64, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_frozen_loss_dx": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_cascade_rows": (C.c_int64, [C.POINTER(ModelDesc), C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_cascade_fwd_bwd": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "odpd_train_epoch_cascade": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ModelDesc), C.POINTER(ModelDesc), C.c_int, C.POINTER(Frames), C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_framed_train_supported": (C.c_int, [C.POINTER(ModelDesc)]),
    "odpd_framed_train_supported_shape": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_train_fwd_bwd_framed": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(Frames), C.c_int64, C.c_int,
                                            C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_train_epoch": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(Frames), C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_clip_adamw_step": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                       C.c_void_p]),
    "odpd_clip_adamw_step_masked": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.c_void_p, C.c_void_p]),
    "odpd_train_epoch_opt": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(Frames), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_train_epoch_split": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(Frames), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "odpd_comm_unique_id": (C.c_int, [C.c_void_p]),
    "odpd_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "odpd_comm_destroy": (C.c_int, [C.c_void_p]),
    "odpd_comm_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "odpd_xchg_create": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p), C.c_void_p]),
    "odpd_xchg_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "odpd_xchg_unlink": (C.c_int, [C.c_void_p]),
    "odpd_comm_set_timeout_ms": (C.c_int, [C.c_void_p, C.c_int64]),
    "odpd_comm_kind": (C.c_int, [C.c_void_p]),
    "odpd_comm_errors": (C.c_int, [C.c_void_p]),
    "odpd_clip_optim_step_dp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
    "odpd_shard_range": (None, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "odpd_train_epoch_dp": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(Frames), C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "odpd_sweep_scratch_bytes": (C.c_int64, [C.c_int, C.c_int64]),
    "odpd_sweep_train_supported": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_sweep_fwd_supported": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.c_int]),
    "odpd_train_epoch_sweep": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(SweepRun), C.c_int, C.POINTER(Frames), C.c_int, C.c_int64,
                                         C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]),
    "odpd_sweep_s16_supported": (C.c_int, [C.POINTER(ModelDesc)]),
    "odpd_sweep_partial_rows": (C.c_int64, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int]),
    "odpd_sweep_workspace_floats": (C.c_int64, [C.POINTER(ModelDesc), C.c_int, C.c_int, C.c_int]),
    "odpd_backbone_fwd_sweep": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.c_int, C.POINTER(SweepRun), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "odpd_clip_optim_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                       C.c_double, C.c_double, C.c_void_p, C.c_void_p]),
}
OPTIMIZER_IDS = {"adamw": 0, "adam": 1, "sgd": 2, "rmsprop": 3}      # enum odpd_optimizer

_lib = None


def lib_path():
    """The in-tree library; $OPENDPD_HIP_LIB points at another build of the same ABI (kernel experiments, tools/exp_time.py)."""
    return os.environ.get("OPENDPD_HIP_LIB") or _build.LIB


def exported_symbols():
    """Names include/opendpd_hip.h declares (used by the CPU test that checks the .so exports them)."""
    return sorted(_EXPORTS)


def load():
    """Load the HIP library; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). opendpd_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in _EXPORTS.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the argument lists above are those of ONE ABI: a stale build (e.g. an experiment library behind $OPENDPD_HIP_LIB) would be
    # called with mismatched arguments through ctypes without any complaint
    if lib.odpd_abi_version() != ABI_VERSION or lib.odpd_built_arch() != b"gfx950":
        raise RuntimeError(f"{path}: ABI version {lib.odpd_abi_version()} / arch {lib.odpd_built_arch()!r}, this package binds ABI "
                           f"{ABI_VERSION} for gfx950 — rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported backbone/hidden size", -3: "collective failed (RCCL missing / refused, or the one-shot exchange could not map its peers)"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"{what} failed: {kind}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (raw-handle query: ~1 us, no Stream object)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous fp32 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("opendpd_amd kernels need tensors on a HIP device (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("opendpd_amd kernels need contiguous tensors")
    return C.c_void_p(t.data_ptr())
