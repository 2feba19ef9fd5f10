"""Pins the CPU oracle (oracle/odpd_oracle.c) to the reference: every function of the oracle is
checked against golden vectors that oracle/gen_golden.py produced by RUNNING the reference
(/root/reference, CPU path) — forward outputs, loss, parameter/input gradients and three
clip+AdamW steps (modules/train_funcs.py:33-44)."""
import numpy as np
import pytest

from oracle.oracle import Oracle, make_model
from tests.golden_util import Fixture, apnrru_noise_mask, rel_err

# fp32 accumulate-order noise between ATen/oneDNN and straight C loops
FWD_TOL = 2e-5     # relative to max|y|
GRAD_TOL = 2e-4    # relative to max|g| per tensor group
STEP_TOL = 2e-5    # parameters after AdamW steps, relative

SINGLE = [
    ("gru_h11", "gru"), ("gru_h23", "gru"), ("dgru_h13", "dgru"), ("dgru_h8", "dgru"), ("dgru_h23", "dgru"),
    ("qgru_h10", "qgru"), ("qgru_h16", "qgru"), ("qgru_amp1_h10", "qgru_amp1"),
    ("lstm_h14", "lstm"), ("vdlstm_h13", "vdlstm"),
    ("deltagru_h15_dense", "deltagru"), ("deltagru_h15_th", "deltagru"),
    ("tres_h15_dense", "deltagru_tcnskip"), ("tres_h15_th", "deltagru_tcnskip"),
    ("deltagru_h24_th", "deltagru"), ("tres_h30_th", "deltagru_tcnskip"),
    ("tcnn_c35", "tcnn"), ("pgjanet_h11", "pgjanet"), ("gmp_m11", "gmp"),
    ("rvtdcnn_h25", "rvtdcnn"), ("rvtdcnn_h6", "rvtdcnn"), ("neuraltx_c36", "neuraltx"), ("neuraltx_c12", "neuraltx"),
    ("deltajanet_h15", "deltajanet"), ("deltajanet_h22", "deltajanet"), ("dvrjanet_h12_k3", "dvrjanet"), ("dvrjanet_h8_k4", "dvrjanet"), ("bojanet_h12", "bojanet"), ("bojanet_h16", "bojanet"), ("bojanet_h5", "bojanet"), ("apnrru_h8", "apnrru"), ("apnrru_h14", "apnrru"), ("apnrru_h5", "apnrru"), ("mcldnn_c8", "mcldnn"), ("mcldnn_c3", "mcldnn"),
]


@pytest.fixture(scope="module")
def orc():
    return Oracle("f32")


@pytest.fixture(scope="module")
def orc64():
    return Oracle("f64")


@pytest.mark.parametrize("name,bb", SINGLE)
def test_forward_loss_grads(orc, orc64, name, bb):
    fx = Fixture(name)
    m = make_model(bb, fx.meta["hidden"], fx.meta.get("thx", 0), fx.meta.get("thh", 0), bits_w=fx.meta.get("num_dvr_units", 0))
    names = fx.keys("sd")
    p = fx.flat("sd", names)
    assert orc.param_count(m) == p.size == fx.meta["n_param"]
    y, st = orc.forward(m, p, fx["x"])
    assert rel_err(y, fx["y"]) < FWD_TOL
    if "stats" in fx:   # delta sparsity counters (deltagru.py:241-247): exact integer counts
        assert np.array_equal(st, fx["stats"]), (st, fx["stats"])
        _, sta = orc.forward(m, p, fx["xa"])
        assert np.array_equal(sta, fx["stats_a"]), (sta, fx["stats_a"])
    # config-shaped frames (T=200, real APA_200MHz slices)
    ya, _ = orc.forward(m, p, fx["xa"])
    assert rel_err(ya, fx["ya"]) < FWD_TOL
    la, _ = orc.loss("l2", ya, fx["ta"])
    assert abs(la - float(fx["loss_a"])) < 1e-5 * max(1.0, abs(float(fx["loss_a"])))
    loss, dy = orc.loss("l2", y, fx["tgt"])
    assert abs(loss - fx["losses"][0]) < 1e-5 * max(1.0, fx["losses"][0])
    dp, dx = orc.backward(m, p, fx["x"], dy)
    g_ref = fx.flat("g", names)
    assert rel_err(dp, g_ref) < GRAD_TOL
    assert rel_err(dx, fx["gx"]) < GRAD_TOL
    # fp64 build agrees with both (sanity on the restatement itself)
    y64, _ = orc64.forward(m, p, fx["x"])
    assert rel_err(y64, fx["y"]) < FWD_TOL


@pytest.mark.parametrize("name,bb", SINGLE)
def test_three_adamw_steps(orc, name, bb):
    fx = Fixture(name)
    m = make_model(bb, fx.meta["hidden"], fx.meta.get("thx", 0), fx.meta.get("thh", 0), bits_w=fx.meta.get("num_dvr_units", 0))
    names = fx.keys("sd")
    sizes = fx.sizes(names)
    p = fx.flat("sd", names).copy()
    mom = np.zeros_like(p)
    var = np.zeros_like(p)
    x, tgt = fx["x"], fx["tgt"]
    # apnrru: one input feature is 0 up to rounding, AdamW makes full steps out of its noise gradient (golden_util.apnrru_noise_mask)
    keep = apnrru_noise_mask(fx.meta["hidden"]) if bb == "apnrru" else np.ones(p.size, dtype=bool)
    for s in range(1, 4):
        y, _ = orc.forward(m, p, x)
        loss, dy = orc.loss("l2", y, tgt)
        assert abs(loss - fx["losses"][s - 1]) < 2e-5 * max(1.0, fx["losses"][s - 1])
        g, _ = orc.backward(m, p, x, dy, need_dx=False)
        orc.clip_adamw(p, g, mom, var, s, fx.meta["lr"], fx.meta["clip"], tensor_sizes=sizes)
        assert rel_err(p[keep], fx.flat(f"p{s}", names)[keep]) < STEP_TOL
    assert rel_err(mom[keep], fx.flat("m3", names)[keep]) < 1e-3
    assert rel_err(var[keep], fx.flat("v3", names)[keep]) < 1e-3


def test_l1_loss(orc):
    rng = np.random.RandomState(0)
    y = rng.randn(3, 5, 2).astype(np.float32)
    t = rng.randn(3, 5, 2).astype(np.float32)
    l, dy = orc.loss("l1", y, t)
    assert abs(l - np.mean(np.abs(y - t))) < 1e-6
    assert np.allclose(dy, np.sign(y - t) / y.size)


def test_bad_args(orc):
    m = make_model("gru", 200)  # hidden too large for the oracle
    with pytest.raises(RuntimeError):
        orc.forward(m, np.zeros(10, np.float32), np.zeros((1, 4, 2), np.float32))


# ---- quantisation-aware QGRU (quant/__init__.py:20-37 -> quant_envs.py:138-306) -----------------------------
QAT = [("quant_qgru_h10_w8a8", "qgru", 8), ("quant_qgru_amp1_h10_w8a8", "qgru_amp1", 8),
       ("quant_qgru_h10_w16a16", "qgru", 16), ("quant_qgru_amp1_h10_w16a16", "qgru_amp1", 16)]
# the generic surgery (quant_envs.py:114-130, 290-306) on the other GRU-cell backbones, wider qgru, and deltagru_tcnskip
# (oracle/gen_golden_quant_more.py)
QAT_MORE = [("quant_gru_h11_w8a8", "gru", 8), ("quant_gru_h23_w8a8", "gru", 8), ("quant_gru_h11_w16a16", "gru", 16),
            ("quant_dgru_h13_w8a8", "dgru", 8), ("quant_dgru_h23_w8a8", "dgru", 8), ("quant_dgru_h13_w16a16", "dgru", 16),
            ("quant_qgru_h20_w8a8", "qgru", 8), ("quant_qgru_h30_w8a8", "qgru", 8), ("quant_qgru_amp1_h20_w16a16", "qgru_amp1", 16),
            ("quant_tres_h15_w8a8_th", "deltagru_tcnskip", 8), ("quant_tres_h15_w8a8_dense", "deltagru_tcnskip", 8),
            ("quant_tres_h15_w16a16_th", "deltagru_tcnskip", 16), ("quant_tres_h30_w8a8_th", "deltagru_tcnskip", 8),
            ("quant_tres_h15_w16a16_pre", "deltagru_tcnskip", 16), ("quant_tres_h15_w8a8_pre", "deltagru_tcnskip", 8)]
# float recurrent core (nn.LSTM; deltajanet's nn.Parameter cell), INT_Linear heads (the surgery finds only nn.Linear layers to swap)
QAT_HEADS = [("quant_lstm_h14_w8a8", "lstm", 8), ("quant_lstm_h14_w16a16", "lstm", 16), ("quant_lstm_h24_w8a8", "lstm", 8), ("quant_lstm_h40_w8a8", "lstm", 8),
             ("quant_vdlstm_h13_w8a8", "vdlstm", 8), ("quant_vdlstm_h13_w16a16", "vdlstm", 16),
             ("quant_deltajanet_h12_w8a8", "deltajanet", 8), ("quant_deltajanet_h40_w16a16", "deltajanet", 16),
             ("quant_neuraltx_h12_w8a8", "neuraltx", 8), ("quant_neuraltx_h20_w16a16", "neuraltx", 16),
             ("quant_rvtdcnn_h12_w8a8", "rvtdcnn", 8), ("quant_rvtdcnn_h6_w16a16", "rvtdcnn", 16), ("quant_rvtdcnn_h32_w8a8", "rvtdcnn", 8),
             ("quant_pgjanet_h11_w8a8", "pgjanet", 8), ("quant_pgjanet_h9_w16a16", "pgjanet", 16), ("quant_pgjanet_h24_w8a8", "pgjanet", 8)]
_BUFFERS = ("n_bits", "pow2_scale", "decimal_num", "integer_num")


def qat_param_names(fx, prefix="sd"):
    return [k for k in fx.keys(prefix) if not any(t in k for t in _BUFFERS)]


@pytest.mark.parametrize("name,bb,bits", QAT + QAT_MORE)
def test_qat_forward_and_grads(orc, name, bb, bits):
    """INT8: the integer-grid restatement reproduces the reference BIT FOR BIT (train-mode float outputs, eval-mode
    16-bit grid outputs, before and after three training steps).  INT16: fp32 accumulation order matters
    (32-bit products), agreement to one output LSB (2^-14)."""
    fx = Fixture(name)
    m = make_model(bb, fx.meta["hidden"], fx.meta.get("thx", 0), fx.meta.get("thh", 0), bits_w=bits, bits_a=bits)
    names = qat_param_names(fx)
    p = fx.flat("sd", names)
    assert orc.param_count(m) == p.size == fx.meta["n_param"]
    p3 = fx.flat("sd3", names)
    if "stats" in fx:      # delta cell: exact sparsity counters of the train-mode forward and of the config-shaped eval forward
        # (16-bit grids: 32-bit products make the fp32 summation order visible — one LSB of a state may flip a threshold decision)
        flips = 0 if bits == 8 else 2
        st = np.zeros(4)
        orc.qat_forward(m, p, fx["x"], stats=st)
        assert np.abs(st - fx["stats"]).max() <= flips, (st, fx["stats"])
        st = np.zeros(4)
        ya = orc.qat_forward(m, p, fx["xa"], eval_mode=True, stats=st)
        assert np.abs(st - fx["stats_a"]).max() <= flips, (st, fx["stats_a"])
        if bits == 8:
            assert np.abs(ya - fx["ya_eval"]).max() <= 2.5e-7
        else:   # a flipped threshold decision sends THAT sequence onto another trajectory from the flip on: at most `flips` of them
            bad = np.abs(ya - fx["ya_eval"]).reshape(ya.shape[0], -1).max(1) > 2.0 ** -12
            assert bad.sum() <= np.abs(st - fx["stats_a"]).max(), (bad, st, fx["stats_a"])
    outs = [(orc.qat_forward(m, p, fx["x"]), fx["y"]), (orc.qat_forward(m, p, fx["x"], eval_mode=True), fx["y_eval"]),
            (orc.qat_forward(m, p3, fx["x"]), fx["y_p3_train"]), (orc.qat_forward(m, p3, fx["x"], eval_mode=True), fx["y_p3_eval"])]
    # deltagru_tcnskip adds its FLOAT skip path (Conv1d / Hardswish are not swapped) to the grid-valued fc_out result: the sum carries
    # the skip's rounding (1 ulp ~ 6e-8), three orders below one step of the coarsest grid involved (2^-14) — a flip in the integer
    # part could not hide under it
    exact_tol = 2.5e-7 if bb == "deltagru_tcnskip" else 0.0
    for got, ref in outs:
        if bits == 8:
            assert np.abs(got - ref).max() <= exact_tol
        else:
            assert np.abs(got - ref).max() <= 2.0 ** -13
    y = outs[0][0]
    loss, dy = orc.loss("l2", y, fx["tgt"])
    assert abs(loss - fx["losses"][0]) < 1e-5
    dp, dx = orc.qat_backward(m, p, fx["x"], dy)
    gref = np.concatenate([(fx["g/" + k].reshape(-1) if ("g/" + k) in fx else np.zeros(fx["sd/" + k].size, np.float32))
                           for k in names])
    assert rel_err(dp, gref) < (1e-5 if bits == 8 else 1e-4)
    assert rel_err(dx, fx["gx"]) < (1e-5 if bits == 8 else 1e-4)
    # the scale parameters receive an exact 0.0 gradient (round() kills it) — quantizers.py:56-65
    off = 0
    for k in names:
        n = fx["sd/" + k].size
        if "scale" in k:
            assert dp[off] == 0.0
        off += n


def grid_close(got, ref, step, flips=2, tol=2e-6):
    """Outputs of a quantised head behind a FLOAT recurrent core: equal to fp32 rounding, except where a state within ~1e-7 of a rounding
    boundary of the activation grid lands on the other side (then that sample moves by a few weight x grid-step products)."""
    d = np.abs(got - ref)
    return (d > tol).sum() <= flips and d.max() <= step


@pytest.mark.parametrize("name,bb,bits", QAT_HEADS)
def test_quantised_heads_forward_and_grads(orc, name, bb, bits):
    """lstm / vdlstm / deltajanet under --quant: the surgery (quant_envs.py:40-60, 290-306) swaps only fc_out (vdlstm: fc_lambda_1,
    fc_lambda_2, fc_out) for INT_Linear; the recurrent core stays float.  Fixtures from the reference: train- and eval-mode outputs before and after three steps, loss,
    gradients (scale parameters: exactly 0; out_quantizer scales: no gradient), input gradient."""
    fx = Fixture(name)
    m = make_model(bb, fx.meta["hidden"], bits_w=bits, bits_a=bits)
    names = qat_param_names(fx)
    p = fx.flat("sd", names)
    assert orc.param_count(m) == p.size == fx.meta["n_param"]
    p3 = fx.flat("sd3", names)
    step = 2.0 ** (2 - bits) * 4
    for got, ref in [(orc.qat_forward(m, p, fx["x"]), fx["y"]), (orc.qat_forward(m, p, fx["x"], eval_mode=True), fx["y_eval"]),
                     (orc.qat_forward(m, p3, fx["x"]), fx["y_p3_train"]), (orc.qat_forward(m, p3, fx["x"], eval_mode=True), fx["y_p3_eval"]),
                     (orc.qat_forward(m, p, fx["xa"], eval_mode=True), fx["ya_eval"])]:
        # (16-bit grids are 256 x finer: a 1e-7 difference of the float core crosses a rounding boundary that much more often —
        # seen: <= 1 % of the outputs, each by one activation step x a head weight)
        # (pgjanet: the quantised layers sit INSIDE the recurrence — a 16-bit flip travels on through the state)
        assert grid_close(got, ref, step, flips=2 if bits == 8 else got.size // (5 if bb == "pgjanet" else 25)), np.abs(got - ref).max()
    y = orc.qat_forward(m, p, fx["x"])
    loss, dy = orc.loss("l2", y, fx["tgt"])
    assert abs(loss - fx["losses"][0]) < 1e-5
    dp, dx = orc.qat_backward(m, p, fx["x"], dy)
    gref = np.concatenate([(fx["g/" + k].reshape(-1) if ("g/" + k) in fx else np.zeros(fx["sd/" + k].size, np.float32)) for k in names])
    gtol = 1e-4 if (bb == "pgjanet" and bits == 16) else 2e-5
    assert rel_err(dp, gref) < gtol
    assert rel_err(dx, fx["gx"]) < gtol
    off = 0
    for k in names:
        if "scale" in k:
            assert dp[off] == 0.0
        off += fx["sd/" + k].size


@pytest.mark.parametrize("name,bb,bits", QAT[:2] + [c for c in QAT_MORE if c[2] == 8])
def test_qat_three_adamw_steps(orc, name, bb, bits):
    """AdamW decays the zero-gradient scales (weight decay) but skips the out_quantizer scales whose grad is None."""
    fx = Fixture(name)
    m = make_model(bb, fx.meta["hidden"], fx.meta.get("thx", 0), fx.meta.get("thh", 0), bits_w=bits, bits_a=bits)
    names = qat_param_names(fx)
    p = fx.flat("sd", names).copy()
    skip = np.concatenate([np.full(fx["sd/" + k].size, "out_quantizer" in k) for k in names])
    mom, var = np.zeros_like(p), np.zeros_like(p)
    for s in range(1, 4):
        y = orc.qat_forward(m, p, fx["x"])
        loss, dy = orc.loss("l2", y, fx["tgt"])
        assert abs(loss - fx["losses"][s - 1]) < 1e-5
        g, _ = orc.qat_backward(m, p, fx["x"], dy, need_dx=False)
        keep = p[skip].copy()
        orc.clip_adamw(p, g, mom, var, s, fx.meta["lr"], fx.meta["clip"])
        p[skip] = keep
        mom[skip] = 0
        var[skip] = 0
        assert rel_err(p, fx.flat(f"p{s}", names)) < 3e-6
