"""HIP-backed backbones behind the reference's registry names (models.py:26-141)."""
from .gru import GRU, DGRU, QGRU, QGRUAmp1  # noqa: F401
from .lstm import LSTM, VDLSTM  # noqa: F401
from .deltagru import DeltaGRU, DeltaJANET, TResDeltaGRU  # noqa: F401
from .pgjanet import PGJANET  # noqa: F401
from .dvrjanet import DVRJANET  # noqa: F401
from .bojanet import BOJANET  # noqa: F401
from .apnrru import APNRRU  # noqa: F401
from .mcldnn import MCLDNN  # noqa: F401
from .tcnn import TCNN, NeuralTX  # noqa: F401
from .gmp import GMP  # noqa: F401
from .rvtdcnn import RVTDCNN  # noqa: F401
