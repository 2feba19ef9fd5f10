"""The index algebra of csrc/gru_s16x.hip's train step, restated in numpy (no GPU): the three-way bf16 split (csrc/odpd_x3.h), the operand layout of
v_mfma_f32_16x16x32_bf16, the selection matrices that TRANSPOSE an operand on the matrix pipe, the contraction over (sequence, time step) whose
result is the gradient of the forward table, and the map from accumulator registers to parameters (one writer per parameter).  The GPU tests
(tests/test_gru_s16x_train_gpu.py) check the kernel against the oracle; this file pins WHY its index arithmetic is right, for whoever edits it next.

Operand layout of the instruction (lane l, element i; cdna_hip_programming.md): A[m = l & 15][k = 8 (l >> 4) + i], B[k = 8 (l >> 4) + i][n = l & 15],
D[m = 4 (l >> 4) + r][n = l & 15]."""
import numpy as np

U, NFS = 6, 2          # units and feature slots per lane (S16X<., 6>)


# ---- bf16 arithmetic --------------------------------------------------------------------------------------------------------------------
def bf16_rne(x):
    """round-to-nearest-even to bf16, returned as float32 (what v_cvt_pk_bf16_f32 followed by a 16-bit left shift gives)"""
    b = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    t1 = bf16_rne(x)
    r = x - t1
    t2 = bf16_rne(r)
    t3 = bf16_rne(r - t2)
    return t1, t2, t3


def test_three_bf16_terms_carry_an_fp32_value_exactly():
    rng = np.random.RandomState(0)
    x = (rng.randn(200000) * 10.0 ** rng.uniform(-6, 3, 200000)).astype(np.float32)
    t1, t2, t3 = split3(x)
    assert np.array_equal((t1.astype(np.float64) + t2 + t3).astype(np.float32), x)          # exact (no rounding in the residual subtractions)
    assert np.array_equal(t1.astype(np.float64) + t2 + t3, x.astype(np.float64))
    # each term is a bf16 value: its low 16 bits are zero
    for t in (t1, t2, t3):
        assert not (t.view(np.uint32) & 0xFFFF).any()
    # what the six kept products drop is below 2^-22 |a b| (three products of weight 2^-24 and below)
    a, b = x[:50000], x[50000:100000]
    A, B = split3(a), split3(b)
    kept = sum(A[i].astype(np.float64) * B[j] for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)))
    err = np.abs(kept - a.astype(np.float64) * b)
    assert (err <= 2.0 ** -22 * np.abs(a.astype(np.float64) * b) + 1e-300).all()


# ---- the instruction, lane by lane --------------------------------------------------------------------------------------------------------
def mfma(A, B, C=None):
    """A, B: (64, 8) per-lane operands; returns D as (64, 4) per-lane registers"""
    Am, Bm = np.zeros((16, 32)), np.zeros((32, 16))
    for l in range(64):
        for i in range(8):
            Am[l & 15, 8 * (l >> 4) + i] = A[l, i]
            Bm[8 * (l >> 4) + i, l & 15] = B[l, i]
    Dm = Am @ Bm
    D = np.zeros((64, 4))
    for l in range(64):
        for r in range(4):
            D[l, r] = Dm[4 * (l >> 4) + r, l & 15]
    return D if C is None else D + C


def sel_g(h):
    """s16x_sel: rows 4 q + r of an M tile <- elements 4 h + r of quad q"""
    E = np.zeros((64, 8))
    for l in range(64):
        nn, kq = l & 15, l >> 4
        if kq == (nn >> 2):
            E[l, 4 * h + (nn & 3)] = 1.0
    return E


def sel_v(j):
    """columns 8 (kq - 2 j) + i of N tile j <- element i of quad kq"""
    E = np.zeros((64, 8))
    for l in range(64):
        nn, kq = l & 15, l >> 4
        if kq == 2 * j + (nn >> 3):
            E[l, nn & 7] = 1.0
    return E


def test_selection_matrices_transpose_an_operand_on_the_matrix_pipe():
    rng = np.random.RandomState(1)
    X = rng.randn(64, 8)          # lane (n = sequence, q): its eight values (one term of a split)
    for h in range(2):
        D = mfma(X, sel_g(h))     # data as the A operand, selection as B
        for l in range(64):
            row, Q = l & 15, l >> 4          # output lane = (tile row, sequence quad)
            for R in range(4):
                src = (4 * Q + R) + 16 * (row >> 2)          # lane of sequence 4 Q + R in quad row >> 2
                assert D[l, R] == X[src, 4 * h + (row & 3)]
    for j in range(2):
        D = mfma(X, sel_v(j))
        for l in range(64):
            col, Q = l & 15, l >> 4
            for R in range(4):
                src = (4 * Q + R) + 16 * (2 * j + (col >> 3))
                assert D[l, R] == X[src, col & 7]


def test_contraction_over_sequences_and_steps_is_the_gradient_of_the_forward_table():
    """d tab[tile T][row 4 q + r][k = 8 kq + i] = sum_{sequence, step} G_{4T + r}(n, q, step) V_i(n, kq, step), formed as in s16x_train_block: transposed
    halves of two steps concatenated along K (element i >> 2 = step)"""
    rng = np.random.RandomState(2)
    G = rng.randn(2, 64, 24)          # [step][lane][e]: the lane's 24 gate derivatives [d r_pre | d z_pre | d(W_hn h) | d n_pre] x 6 units
    V = rng.randn(2, 64, 8)           # [step][lane][i]: the lane's cell operand [h_0 .. h_5, f_0, f_1]
    want = np.zeros((6, 16, 32))
    for T in range(6):
        for m in range(16):
            for k in range(32):
                q, r, kq, i = m >> 2, m & 3, k >> 3, k & 7
                want[T, m, k] = sum(G[s, n + 16 * q, 4 * T + r] * V[s, n + 16 * kq, i] for s in range(2) for n in range(16))
    vt = [[mfma(V[s], sel_v(j)) for j in range(2)] for s in range(2)]          # [step][N tile] -> (64, 4)
    for c in range(3):                # chunk c = values 8 c .. 8 c + 7 = M tiles 2 c, 2 c + 1
        for hh in range(2):
            T = 2 * c + hh
            gt = [mfma(G[s][:, 8 * c:8 * c + 8], sel_g(hh)) for s in range(2)]
            A = np.concatenate([gt[0], gt[1]], axis=1)          # K element i = 4 step + R
            for j in range(2):
                Bop = np.concatenate([vt[0][j], vt[1][j]], axis=1)
                D = mfma(A, Bop)
                for l in range(64):
                    for R in range(4):
                        assert abs(D[l, R] - want[T, 4 * (l >> 4) + R, 16 * j + (l & 15)]) < 1e-9


# ---- accumulator registers -> parameters (s16x_write_row) ------------------------------------------------------------------------------------
def gru_layout(H, F, dgru):
    o = {}
    off = 0
    for name, size in (("w_ih", 3 * H * F), ("w_hh", 3 * H * H), ("b_ih", 3 * H), ("b_hh", 3 * H), ("w_out", 2 * (H + 6 if dgru else H)), ("b_out", 2),
                       ("w_hid", H * H if dgru else 0), ("b_hid", H if dgru else 0)):
        o[name] = off
        off += size
    o["P"] = off
    return o


def write_row(H, F, dgru):
    """who writes which parameter: {parameter index: [(what, gate, unit, column)]}, following s16x_write_row statement by statement"""
    L, OW = gru_layout(H, F, dgru), (H + 6 if dgru else H)
    writers = {}

    def put(idx, what):
        writers.setdefault(idx, []).append(what)

    for lane in range(64):
        n, q = lane & 15, lane >> 4
        for j in range(2):
            kq, i = 2 * j + (n >> 3), n & 7
            ku, fsl = U * kq + i, NFS * kq + (i - U)
            for t in range(6):
                for R in range(4):
                    s = 4 * t + R
                    gate, u = s // U, U * q + s % U
                    if u >= H:
                        continue
                    if i < U:
                        if gate != 3 and ku < H:
                            put(L["w_hh"] + (gate * H + u) * H + ku, ("w_hh", gate, u, ku))
                    elif gate == 2:
                        if fsl == F:
                            put(L["b_hh"] + 2 * H + u, ("b_hn", 2, u, None))
                    else:
                        g = 2 if gate == 3 else gate
                        if fsl < F:
                            put(L["w_ih"] + (g * H + u) * F + fsl, ("w_ih", g, u, fsl))
                        elif fsl == F:
                            put(L["b_ih"] + g * H + u, ("b_ih", g, u, None))
                            if gate < 2:
                                put(L["b_hh"] + g * H + u, ("b_hh", g, u, None))
            if dgru:
                for t in range(2):
                    for R in range(4):
                        jj, u = 4 * t + R, U * q + 4 * t + R
                        if jj < U and u < H:
                            if i < U:
                                if ku < H:
                                    put(L["w_hid"] + u * H + ku, ("w_hid", 0, u, ku))
                            elif fsl == F:
                                put(L["b_hid"] + u, ("b_hid", 0, u, None))
        if n == 0:
            for cc in range(2):
                for j in range(U):
                    if U * q + j < H:
                        put(L["w_out"] + cc * OW + U * q + j, ("w_out", cc, U * q + j, None))
                for e in range(NFS):
                    slot = NFS * q + e
                    if dgru and slot < F:
                        put(L["w_out"] + cc * OW + H + slot, ("w_out_feat", cc, slot, None))
                    elif slot == F:
                        put(L["b_out"] + cc, ("b_out", cc, None, None))
    return L, writers


def test_every_parameter_has_exactly_one_writer_and_the_right_one():
    for H in range(17, 25):
        for F, dgru in ((2, False), (4, False), (6, True)):
            L, w = write_row(H, F, dgru)
            assert sorted(w) == list(range(L["P"])), (H, F, dgru, sorted(set(range(L["P"])) - set(w))[:5])
            assert all(len(v) == 1 for v in w.values()), (H, F, dgru)
            # spot checks against the reference's named_parameters() layout (backbones/gru.py, dgru.py): weight_ih_l0[g H + u][f], weight_hh_l0[g H + u][k]
            assert w[L["w_ih"] + (2 * H + 3) * F + 1][0] == ("w_ih", 2, 3, 1)
            assert w[L["w_hh"] + (1 * H + (H - 1)) * H + 5][0] == ("w_hh", 1, H - 1, 5)
            assert w[L["b_hh"] + 2 * H + 7][0] == ("b_hn", 2, 7, None)
            if dgru:
                assert w[L["w_hid"] + 4 * H + (H - 1)][0] == ("w_hid", 0, 4, H - 1)
                assert w[L["w_out"] + (H + 6) + H + 5][0] == ("w_out_feat", 1, 5, None)
