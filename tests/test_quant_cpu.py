"""CPU checks of the QAT model surgery (opendpd_amd/quant.py): same names, same RNG consumption, identical initial state."""
import numpy as np
import torch

from tests.golden_util import Fixture


class _Proj:
    quant = True
    pretrained_model = ""


def _build(bb, H, bits):
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    torch.manual_seed(0)
    fnet = CoreModel(2, H, 1, bb)
    _Proj.n_bits_w = _Proj.n_bits_a = bits
    torch.manual_seed(123)       # the fixture generator seeds here, before the reference's get_quant_model
    return get_quant_model(_Proj, fnet)


def test_quant_state_dict_matches_reference_bitwise():
    for name, bb, bits in [("quant_qgru_h10_w8a8", "qgru", 8), ("quant_qgru_amp1_h10_w16a16", "qgru_amp1", 16)]:
        fx = Fixture(name)
        q = _build(bb, fx.meta["hidden"], bits)
        sd = q.state_dict()
        ref_keys = fx.keys("sd")
        assert list(sd.keys()) == ref_keys
        for k in ref_keys:      # side-effect buffers included: no forward has run yet, they hold their construction-time values
            assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
        assert sum(p.numel() for p in q.parameters()) == fx.meta["n_param"]


def test_identity_when_quant_off():
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model

    class P:
        quant = False
    net = CoreModel(2, 10, 1, "qgru")
    assert get_quant_model(P, net) is net
