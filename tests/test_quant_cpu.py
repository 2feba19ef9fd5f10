"""CPU checks of the QAT model surgery (opendpd_amd/quant.py): same names, same RNG consumption, identical initial state."""
import numpy as np
import pytest
import torch

from tests.golden_util import Fixture


class _Proj:
    quant = True
    pretrained_model = ""


def _build(bb, H, bits, thx=0.0, thh=0.0, pretrained=""):
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    torch.manual_seed(0)
    fnet = CoreModel(2, H, 1, bb, thx=thx, thh=thh)
    _Proj.n_bits_w = _Proj.n_bits_a = bits
    _Proj.pretrained_model = pretrained
    torch.manual_seed(123)       # the fixture generator seeds here, before the reference's get_quant_model
    try:
        return get_quant_model(_Proj, fnet)
    finally:
        _Proj.pretrained_model = ""


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


def test_general_surgery_state_dicts_and_rng_match_the_reference(tmp_path):
    """gru / dgru (GRU swap + INT_Linear heads), qgru beyond 16 units, deltagru_tcnskip (its op modules and Linears; also from a float
    `--pretrained_model` checkpoint = the OpenDPDv2 flow): identical keys, order, values (parameters AND buffers) and the same global
    RNG state afterwards as the reference's get_quant_model (oracle/gen_golden_quant_more.py)."""
    from tests.test_oracle_golden import QAT_MORE
    for name, bb, bits in QAT_MORE:
        fx = Fixture(name)
        pre = ""
        if fx.meta["pretrained"]:
            pre = str(tmp_path / (name + ".pt"))
            torch.save({k: torch.from_numpy(fx["pre/" + k]) for k in fx.keys("pre")}, pre)
        q = _build(bb, fx.meta["hidden"], bits, fx.meta["thx"], fx.meta["thh"], pre)
        rng_after = torch.rand(4).numpy()
        sd = q.state_dict()
        assert list(sd.keys()) == fx.keys("sd"), name
        for k in fx.keys("sd"):
            assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), (name, k)
        assert sum(p.numel() for p in q.parameters()) == fx.meta["n_param"]
        assert np.array_equal(rng_after, fx["rng_after"]), name
        if fx.meta["pretrained"]:
            assert np.array_equal(sd["backbone.rnn.x2h.weight"].numpy(), fx["pre/backbone.rnn.x2h.weight"])


def warnings_ctx():
    import warnings
    ctx = warnings.catch_warnings()
    ctx.__enter__()
    warnings.simplefilter("ignore")

    class _C:
        def __enter__(self_):
            return self_

        def __exit__(self_, *a):
            ctx.__exit__(*a)
            return False
    return _C()


def test_surgery_on_the_other_backbones():
    """gmp / tcnn hold nothing the surgery swaps (the reference returns an identical copy); deltagru's quantised model cannot run in
    the reference either; the partially swapped backbones and configurations beyond the kernels say so."""
    import pytest
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    _Proj.n_bits_w = _Proj.n_bits_a = 8
    for bb in ("gmp", "tcnn"):
        net = CoreModel(2, 11, 1, bb)
        assert get_quant_model(_Proj, net) is net
    with pytest.raises(RuntimeError):
        get_quant_model(_Proj, CoreModel(2, 8, 1, "deltagru"))
    # apnrru / bojanet / dvrjanet / mcldnn: quantised through the announced ATen route since r05 (tests/test_quant_partial_cpu.py)
    with warnings_ctx():
        q = get_quant_model(_Proj, CoreModel(2, 8, 1, "apnrru"))
    assert not q.backbone.native and any("weight_quantizer" in k for k in q.state_dict())
    # num_layers means nothing to pgjanet / rvtdcnn / neuraltx (models.py never hands it to them): `--quant --DPD_num_layers 2` runs in the
    # reference and here; for a recurrent core it is outside the kernels
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for bb in ("pgjanet", "rvtdcnn", "neuraltx"):
            q = get_quant_model(_Proj, CoreModel(2, 8, 2, bb))
            assert any("quantizer" in k for k in q.state_dict()), bb
        with pytest.raises(NotImplementedError):
            get_quant_model(_Proj, CoreModel(2, 8, 2, "lstm"))


def test_head_only_surgery_state_dict_and_rng_match_the_reference():
    """lstm / vdlstm / deltajanet / neuraltx: the surgery finds only the nn.Linear heads to swap (the recurrent core stays float) — identical keys, order, values (parameters and buffers) and the
    same global RNG state afterwards as the reference's get_quant_model (oracle/gen_golden_quant_more.py)."""
    from tests.test_oracle_golden import QAT_HEADS
    for name, bb, bits in QAT_HEADS:
        fx = Fixture(name)
        q = _build(bb, fx.meta["hidden"], bits)
        rng_after = torch.rand(4).numpy()
        sd = q.state_dict()
        assert list(sd.keys()) == fx.keys("sd"), name
        for k in fx.keys("sd"):
            # (the float core is the seeded construction, not the surgery: a 40-column orthogonal init goes through a blocked QR whose
            # last bit depends on the LAPACK threading of the process that drew it)
            same = np.array_equal(sd[k].numpy(), fx["sd/" + k]) or (".rnn.weight" in k and np.abs(sd[k].numpy() - fx["sd/" + k]).max() < 5e-7)
            assert same, (name, k)
        assert sum(p.numel() for p in q.parameters()) == fx.meta["n_param"] == q.backbone.n_flat
        assert np.array_equal(rng_after, fx["rng_after"]), name
        last = "IQ_match" if bb == "neuraltx" else "W_o" if bb == "pgjanet" else "fc_out"      # (neuraltx: bias-free, and the last named_children entry)
        tail = [f"backbone.{last}.weight"] + ([] if bb == "neuraltx" else [f"backbone.{last}.bias"]) + [
            f"backbone.{last}.weight_quantizer.scale", f"backbone.{last}.act_quantizer.scale", f"backbone.{last}.out_quantizer.scale"]
        assert [n for n, _ in q.named_parameters()][-len(tail):] == tail


@pytest.mark.parametrize("bb,H", [("lstm", 14), ("vdlstm", 9), ("deltajanet", 12), ("neuraltx", 10), ("rvtdcnn", 7), ("pgjanet", 6)])
def test_pretrained_float_checkpoint_reaches_the_layer_surgeries(tmp_path, capsys, bb, H):
    """--pretrained_model on the backbones whose surgery swaps Linear / Conv2d layers (quant_envs.py:173-182: the checkpoint is strict-loaded
    into the float model BEFORE the swap): every weight of the result is the checkpoint's — also the float core's —, the swapped layers'
    biases are re-drawn by the INT layers' constructors, rvtdcnn's INT_Conv2D takes its weight scale from the LOADED weights (init_step_size),
    and a checkpoint with other keys makes the call fall back to the float model with the reference's warning."""
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import _HEAD_LAYERS, get_quant_model

    class P:
        quant = True
        n_bits_w = n_bits_a = 8
    torch.manual_seed(5)
    src = CoreModel(2, H, 1, bb)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(0.01)                                  # biases start at 0 in several backbones: make "re-drawn" visible
    ckpt = tmp_path / "float.pt"
    torch.save(src.state_dict(), ckpt)
    P.pretrained_model = str(ckpt)
    torch.manual_seed(0)
    fnet = CoreModel(2, H, 1, bb)
    torch.manual_seed(123)
    q = get_quant_model(P, fnet)
    assert q is not fnet and q.backbone.desc.bits_w == 8
    sd, ref = q.state_dict(), src.state_dict()
    heads = _HEAD_LAYERS[bb]
    for k, v in ref.items():
        layer = k.split(".")[1]
        if k.endswith("bias") and layer in heads:
            assert not torch.equal(sd[k], v), k           # INT_Linear / INT_Conv2D draw a fresh default-init bias
        else:
            assert torch.equal(sd[k], v), k
    if bb == "rvtdcnn":
        want = ref["backbone.Conv2d.weight"].abs().mean() * 2 / 127 ** 0.5
        assert torch.equal(sd["backbone.Conv2d.weight_quantizer.scale"], want) and sd["backbone.Conv2d.weight_quantizer.scale"].dim() == 0
    # a quantised checkpoint is not a float checkpoint: other keys -> the float model comes back, with the warning
    bad = tmp_path / "quant.pt"
    torch.save(sd, bad)
    P.pretrained_model = str(bad)
    torch.manual_seed(0)
    fnet = CoreModel(2, H, 1, bb)
    capsys.readouterr()
    assert get_quant_model(P, fnet) is fnet and "[WARN] Quantization setup failed" in capsys.readouterr().out


def test_identity_when_quant_off():
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model

    class P:
        quant = False
    net = CoreModel(2, 10, 1, "qgru")
    assert get_quant_model(P, net) is net


def test_pretrained_model_follows_the_reference_load_path(tmp_path, capsys):
    """quant_envs.py:173-182: a PyGRU-keyed float checkpoint is loaded BEFORE quantisation (weights kept, biases re-drawn, default
    scales); nn.GRU-keyed float checkpoints and quantised checkpoints make the reference fall back to the float model with a warning."""
    import json
    from opendpd_amd import CoreModel
    from opendpd_amd.quant import get_quant_model
    fx = Fixture("quant_pretrained_qgru_h10")
    H, bits = fx.meta["hidden"], fx.meta["bits"]
    assert fx.meta["outcomes"] == {"pygru": "quantised", "nngru": "float model returned unchanged",
                                   "quant": "float model returned unchanged", "missing": "float model returned unchanged"}

    class P:
        quant = True
        n_bits_w = n_bits_a = bits
    pre = tmp_path / "pygru.pt"
    torch.save({k: torch.from_numpy(fx["pre/" + k]) for k in fx.keys("pre")}, pre)
    P.pretrained_model = str(pre)
    torch.manual_seed(0)
    fnet = CoreModel(2, H, 1, "qgru")
    torch.manual_seed(123)
    q = get_quant_model(P, fnet)
    assert q is not fnet
    sd = q.state_dict()
    assert list(sd.keys()) == fx.keys("sd")
    for k in fx.keys("sd"):
        assert np.array_equal(sd[k].numpy(), fx["sd/" + k]), k
    # the weights are the checkpoint's, the biases are not (INT_Linear re-draws them)
    assert np.array_equal(sd["backbone.fc_out.weight"].numpy(), fx["pre/backbone.fc_out.weight"])
    assert not np.array_equal(sd["backbone.fc_out.bias"].numpy(), fx["pre/backbone.fc_out.bias"])
    # the two refused kinds: a float nn.GRU checkpoint and a quantised checkpoint
    torch.manual_seed(1)
    nn_sd = CoreModel(2, H, 1, "qgru").state_dict()
    assert list(nn_sd.keys()) == json.loads(str(fx["nngru_keys"]))
    for name, ckpt in (("nngru", nn_sd), ("quant", sd)):
        path = tmp_path / f"{name}.pt"
        torch.save(ckpt, path)
        P.pretrained_model = str(path)
        torch.manual_seed(0)
        fnet = CoreModel(2, H, 1, "qgru")
        before = {k: v.clone() for k, v in fnet.state_dict().items()}
        capsys.readouterr()
        torch.manual_seed(321)
        out = get_quant_model(P, fnet)
        assert out is fnet and "[WARN] Quantization setup failed" in capsys.readouterr().out
        assert all(torch.equal(before[k], v) for k, v in out.state_dict().items())
        # the reference had already built (and re-initialised) the GRU of GRUCells when its strict load failed: same RNG state afterwards,
        # i.e. the same DataLoader shuffle order in the train_dpd that follows
        assert np.array_equal(torch.rand(4).numpy(), fx[f"rng_after_refused_{name}"]), name
    # an unreadable file: warned about and refused the same way (torch.load raises inside the reference's try block)
    P.pretrained_model = str(tmp_path / "missing.pt")
    torch.manual_seed(0)
    fnet = CoreModel(2, H, 1, "qgru")
    torch.manual_seed(321)
    assert get_quant_model(P, fnet) is fnet and "[WARN] Quantization setup failed" in capsys.readouterr().out
    assert np.array_equal(torch.rand(4).numpy(), fx["rng_after_refused_missing"])
