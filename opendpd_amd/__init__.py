"""opendpd_amd — MI355X-native OpenDPD training hot path (HIP kernels behind the reference's
CoreModel / CascadedModel registry).  See DESIGN.md."""
from .models import CoreModel, CascadedModel  # noqa: F401
from .api import train_pa, train_dpd, run_dpd, load_dataset, create_dataset, OpenDPDTrainer  # noqa: F401
from .sweep import train_pa_sweep  # noqa: F401

__all__ = ["train_pa", "train_dpd", "run_dpd", "load_dataset", "create_dataset", "OpenDPDTrainer",      # opendpd/__init__.py
           "CoreModel", "CascadedModel",                                                                 # models.py
           "train_pa_sweep"]                                                                             # K runs in lockstep (sweep.py)
__version__ = "0.1.0"
