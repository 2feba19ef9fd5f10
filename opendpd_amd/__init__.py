"""opendpd_amd — MI355X-native OpenDPD training hot path (HIP kernels behind the reference's
CoreModel / CascadedModel registry).  See DESIGN.md."""
import os as _os

# dmabuf IPC: the hipIpc gradient exchange (csrc/odpd_xchg.h) and RCCL share device memory across the one-process-per-GPU ranks; the host
# driver of this image has no legacy IPC mode (`hipIpcGetMemHandle: invalid argument` without this).  Must be in the environment before
# the first HIP call of the process, hence at package import; a value the caller exported wins.
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from .models import CoreModel, CascadedModel  # noqa: F401
from .api import train_pa, train_dpd, run_dpd, load_dataset, create_dataset, OpenDPDTrainer  # noqa: F401
from .sweep import train_pa_sweep  # noqa: F401

__all__ = ["train_pa", "train_dpd", "run_dpd", "load_dataset", "create_dataset", "OpenDPDTrainer",      # opendpd/__init__.py
           "CoreModel", "CascadedModel",                                                                 # models.py
           "train_pa_sweep"]                                                                             # K runs in lockstep (sweep.py)
__version__ = "0.1.0"
