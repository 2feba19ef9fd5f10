"""The unit-slot numbering of csrc/qat_s16.hip (r06: three / two unit slots per lane at hidden <= 12 / <= 8), restated in numpy (no GPU): which
unit a lane's slot holds, that the operand tiles built from it compute the same mat-vec as the plain weights, and that the gradient write-out has
exactly one writer per parameter.  The GPU tests (tests/test_quant_more_gpu.py::test_three_unit_slots_*) check the kernels against the oracle and
against the four-slot kernels; this file pins WHY the renumbering is only a renumbering, for whoever edits `q16_unit`, `q16_entry` or `q16_write_row`.

Mapping (csrc/qat_s16.hip): lane l = (n = l & 15, q = l >> 4); tile index j = 4 q + e; f32 MFMA 16x16x4 chunk c takes element c of every quad as the
four K entries (k = 4 c' ... — the chunk's K index IS the quad), so a mat-vec over a unit tile is  D[row m] = sum_{q, e} A_e[m][q] v_e[q],  the A
operand of chunk e holding the weight of (row m, tile index 4 q + e)."""
import numpy as np


def q16_unit(U, kt, j, H):
    """unit of tile index j of tile kt; H = "no unit" (csrc/qat_s16.hip q16_unit)"""
    if U == 4:
        return 16 * kt + j
    return U * (j >> 2) + (j & 3) if (j & 3) < U else H


def slots_for(H):
    return 4 if H > 12 else (2 if H <= 8 else 3)      # qat_s16.hip unit_slots (GRUCell kinds, knob qat_u3 = 1)


def test_every_unit_has_exactly_one_slot_and_dead_slots_are_the_high_elements():
    for H in range(1, 17):
        for U in {4, slots_for(H)}:
            owner = {}
            for j in range(16):
                u = q16_unit(U, 0, j, H)
                if u < H:
                    assert u not in owner
                    owner[u] = j
                if (j & 3) >= U:
                    assert u >= H          # elements U .. 3 of every quad are dead
            assert sorted(owner) == list(range(H)), (H, U)


def tile_matvec(U, H, W, v):
    """(W v) through the operand tiles of q16_entry's HH group and the lanes' slot vectors: rows m <-> q16_unit(m), K entry (chunk e, quad q) <-> q16_unit(4 q + e);
    chunks e >= U are not issued (s16n_matvec<NT, U>)."""
    A = np.zeros((4, 16, 4))          # [chunk e][row m][quad q]
    vec = np.zeros((4, 4))            # [quad q][element e]: the lane's own values
    for m in range(16):
        o = q16_unit(U, 0, m, H)
        for q in range(4):
            for e in range(4):
                k = q16_unit(U, 0, 4 * q + e, H)
                if o < H and k < H:
                    A[e, m, q] = W[o, k]
    for q in range(4):
        for e in range(4):
            k = q16_unit(U, 0, 4 * q + e, H)
            vec[q, e] = v[k] if k < H else 123.0      # garbage in a dead slot must not matter where the chunk is issued at all
    D = np.zeros(16)
    for e in range(U):                # the dead chunks are skipped
        D += A[e] @ vec[:, e]
    out = np.zeros(H)
    for m in range(16):               # D row 4 q + r sits in lane quad q, register r: the lane's slot r
        o = q16_unit(U, 0, m, H)
        if o < H:
            out[o] = D[m]
        else:
            assert D[m] == 0.0        # dead rows: zero weights
    return out


def test_the_renumbered_tiles_compute_the_same_matvec():
    rng = np.random.RandomState(0)
    for H in range(1, 17):
        W, v = rng.randint(-128, 128, (H, H)).astype(np.float64), rng.randint(-128, 128, H).astype(np.float64)
        for U in {4, slots_for(H)}:
            assert np.array_equal(tile_matvec(U, H, W, v), W @ v), (H, U)


def test_int8_operand_bytes_follow_the_same_numbering():
    """I8W groups (q16_entry): byte j of the lane's h word <-> K index 8 q + j <-> unit q16_unit(4 q + j); the B operand's bytes are the lane's own
    packed slots, so the integer dot product is the same sum over real units."""
    rng = np.random.RandomState(1)
    for H in (3, 6, 8, 10, 12, 13, 16):
        U = slots_for(H)
        W, h = rng.randint(-128, 128, (H, H)), rng.randint(-128, 128, H)
        for m in range(16):
            o = q16_unit(U, 0, m, H)
            acc = 0
            for q in range(4):
                for j in range(4):
                    u = q16_unit(U, 0, 4 * q + j, H)
                    wbyte = W[o, u] if (o < H and u < H) else 0
                    hbyte = h[u] if (u < H and j < U) else 0          # dead slots are packed as 0 (std_cell: hb[i] = 0 for i >= U)
                    acc += int(wbyte) * int(hbyte)
            if o < H:
                assert acc == int(W[o] @ h)
            else:
                assert acc == 0


def layout(kind, H):
    F = {"gru": 2, "q4": 4, "dgru": 6}[kind]
    OW = H + 6 if kind == "dgru" else H
    o, L = 0, {}
    for name, size in (("wx", 3 * H * F), ("bx", 3 * H), ("sx", 3), ("wh", 3 * H * H), ("bh", 3 * H), ("sh", 3), ("sgate", 4), ("wo", 2 * OW), ("bo", 2), ("so", 3)):
        L[name] = o
        o += size
    if kind == "dgru":
        for name, size in (("whid", H * H), ("bhid", H), ("shid", 3)):
            L[name] = o
            o += size
    L["P"], L["F"], L["OW"] = o, F, OW
    return L


def write_row_writers(kind, H, U):
    """who writes which parameter, following q16_write_row statement by statement (non-merged tiles, one unit tile)"""
    L = layout(kind, H)
    F, OW = L["F"], L["OW"]
    w = {}

    def put(idx, what):
        w.setdefault(idx, []).append(what)

    for lane in range(64):
        n, q = lane & 15, lane >> 4
        for rr in range(4):
            u = q16_unit(U, 0, 4 * q + rr, H)
            if u < H:
                for g in range(3):
                    fs = n
                    if fs < F:
                        put(L["wx"] + (g * H + u) * F + fs, ("wx", g, u, fs))
                    if fs == F:
                        put(L["bx"] + g * H + u, ("bx", g, u))
                        if g < 2:
                            put(L["bh"] + g * H + u, ("bh", g, u))
                    col = q16_unit(U, 0, n, H)
                    if col < H:
                        put(L["wh"] + (g * H + u) * H + col, ("wh", g, u, col))
                if kind == "dgru":
                    col = q16_unit(U, 0, n, H)
                    if col < H:
                        put(L["whid"] + u * H + col, ("whid", u, col))
            if n == 0 and u < H:
                put(L["wo"] + u, ("wo", 0, u))
                put(L["wo"] + OW + u, ("wo", 1, u))
                put(L["bh"] + 2 * H + u, ("bhn", u))
                if kind == "dgru":
                    put(L["bhid"] + u, ("bhid", u))
        if kind == "dgru":
            for cc in range(2):
                for c in range(2):
                    slot = 4 * c + q
                    if n == 0 and slot < 6:
                        put(L["wo"] + cc * OW + H + slot, ("wof", cc, slot))
        if n == 0 and q == 0:
            put(L["bo"], ("bo", 0))
            put(L["bo"] + 1, ("bo", 1))
    return L, w


def test_gradient_write_out_has_one_writer_per_parameter_in_every_slot_layout():
    for kind in ("gru", "q4", "dgru"):
        for H in range(1, 17):
            for U in {4, slots_for(H)}:
                L, w = write_row_writers(kind, H, U)
                scales = set()
                for name, size in (("sx", 3), ("sh", 3), ("sgate", 4), ("so", 3)) + ((("shid", 3),) if kind == "dgru" else ()):
                    scales.update(range(L[name], L[name] + size))
                assert sorted(w) == sorted(set(range(L["P"])) - scales), (kind, H, U)      # every parameter but the scales (exact zero gradient)
                assert all(len(v) == 1 for v in w.values()), (kind, H, U)
                if H >= 3:
                    assert w[L["wh"] + (1 * H + (H - 1)) * H + 2][0] == ("wh", 1, H - 1, 2)
                    assert w[L["wx"] + (2 * H + 1) * L["F"] + 1][0] == ("wx", 2, 1, 1)
