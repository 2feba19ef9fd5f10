// qat_s16.hip — quantisation-aware models on the 16-sequences-per-wave ("S16") mapping.
//
// The reference's surgery is generic (quant/quant_envs.py:114-130 swaps every nn.GRU for the Python GRU of GRUCells,
// :290-306 every Sigmoid / Tanh / Add / Mul module and every nn.Linear), so `--quant` applies to
//   gru               backbones/gru.py        GRUCell on [I,Q]                      + INT_Linear fc_out
//   dgru              backbones/dgru.py       GRUCell on [I,Q,a,a^3,sin,cos]        + INT_Linear fc_hid, relu, cat, INT_Linear fc_out
//   qgru / qgru_amp1  backbones/qgru*.py      GRUCell on 4 features, ANY hidden size (bash_scripts/quant_qgru_dpd_regr.sh:74)
//   deltagru_tcnskip  backbones/deltagru_tcnskip.py:156-162, 266-291: its delta layer routes the gate arithmetic through
//                     such modules — the OpenDPDv2 recipe (bash_scripts/OpenDPDv2.sh:84-117, W16A16 from a float checkpoint)
// GRUCell (quant/modules/gru.py:43-59):  x_t = x2h(x), h_t = h2h(h)  (INT_Linear: F.linear(q_a(in), q_w(W), b), quant_layers.py:70-76)
//   r = Qsig(Qadd(x_r + h_r)), z likewise, n = Qtanh(Qadd(x_n + Qmul(r h_n))), h' = Qadd(Qmul(z h) + Qmul((1 - z) n))
// delta layer:  dm += x2h(q_a(dx)) + h2h(q_a(dh)) on the thresholded deltas (fp32 accumulators, bias-free),
//   r = Qsig(dm_r), z = Qsig(dm_z), n = Qtanh(Qadd(dm_n + Qmul(r dm_nh))), h' = Qadd(Qmul(Qadd(1 - z) n) + Qmul(z h)),
//   y = fc_out(h') [+ 16-bit output quantiser in eval mode] + float TCN skip.
// Quantiser (quantizers.py:15-97): s = 2^round(log2|scale|), q(x) = rint(clamp(x / s, Qn, Qp)) s, straight-through gradient
// inside the clamp range, exactly 0 for the scale parameters.
//
// Mapping as in gru_s16n.hip / delta_s16.hip: lane (n = sequence, q = unit quad) owns units 16 kt + 4 q + i of NT tiles
// (hidden <= 16 NT, NT <= 2); every mat-vec is an exact-fp32 v_mfma_f32_16x16x4_f32 whose A operand is the QUANTISED weight
// (streamed from an LDS table) and whose B operand is the lane's own quantised activation.  On 8-bit grids every product and
// every partial sum is an integer multiple of 2^-(e_a + e_w) below 2^24: the accumulations are exact in any order, the fp32 bias
// is added once afterwards like F.linear does, and the results are BIT-IDENTICAL with the reference's.  (16-bit grids: 32-bit
// products, the summation order becomes visible at the level of one LSB — in the reference as well.)
// Gates: for <= 8 activation bits sigmoid / tanh inputs on the add-quantiser grid are looked up in 2^bits-entry LDS tables
// (evaluated in double at kernel start); the delta cell's sigmoids take the raw accumulators: their QUANTISED value is found
// exactly from a table of the rounding boundaries logit((k - 1/2) s) around an fp32 first guess.  Wider grids: ~2-ulp fp32 evaluations.
// This file is compiled with FP contraction off: a fused multiply-add would change roundings the reference does not have.
#include <type_traits>

#include "odpd_qat.h"

#pragma clang fp contract(off)

namespace odpd {
namespace q16 {

// table groups ([group][lane] float4) and sizes for NT tiles of 16 hidden units
template <int MK, int NT> struct QT {
    using K = Kind<MK>;
    static constexpr int NB = K::TRES ? 0 : 3 * NT;
    static constexpr int IH = 0;                          // g*NT + mt           : (chunk 0, chunk 1) q_w(W_x)[g][16mt+m][4e+q]
    static constexpr int HH = IH + 3 * NT;                // (g*NT + mt)*NT + kt : q_w(W_h)[g][16mt+m][16kt+4q+e]
    static constexpr int BX = HH + 3 * NT * NT;           // g*NT + mt           : b_x[g][16mt+4q+e]   (GRUCell)
    static constexpr int BH = BX + NB;                    //                       b_h
    static constexpr int WOUT = BH + NB;                  // cc*NT + mt          : q_w(fc_out)[cc][16mt+4q+e]
    static constexpr int HID = WOUT + 2 * NT;             // mt*NT + kt          : q_w(fc_hid)[16mt+m][16kt+4q+e]   (dgru)
    static constexpr int BHID = HID + (K::DGRU ? NT * NT : 0);
    static constexpr int WOF = BHID + (K::DGRU ? NT : 0); // one group: q_w(fc_out)[0][H+q], [0][H+4+q], [1][H+q], [1][H+4+q]
    static constexpr int I8W = WOF + (K::DGRU ? 1 : 0);   // NT == 1: two groups of int8-packed weights for the integer matrix pipe (see I8Ops)
    static constexpr int NG_FWD = I8W + (NT == 1 ? 2 : 0);
    static constexpr int HHT = NG_FWD;                    // (g*NT + mt)*NT + kt : q_w(W_h)[g][16kt+4q+e][16mt+m]
    static constexpr int HIDT = HHT + 3 * NT * NT;        // mt*NT + kt          : q_w(fc_hid)[16kt+4q+e][16mt+m]
    static constexpr int NG_BWD = HIDT + (K::DGRU ? NT * NT : 0);
    static constexpr int IHT = NG_BWD;                    // g*NT + kt : q_w(W_x)[g][16kt+4q+e][slot(m)], slot(m) = 4 (m & 3) + (m >> 2)
    static constexpr int NG_DX = IHT + 3 * NT;
    static constexpr int S = NT == 1 ? 2 : 1;             // BPTT checkpoint stride of this family (what the 512-register file holds)
    static constexpr int kCk = K::TRES ? 6 * NT + 1 : NT; // float4 per lane per checkpoint
    static constexpr int kTiles = 5 * NT + 1 + (K::DGRU ? 2 * NT : 0);
};

// UNIT SLOTS PER LANE.  U = 4: tile index j = 4 q + e of tile kt (row j of an A operand, element e of lane quad q) is unit 16 kt + j.
// U = 3 (one tile, hidden <= 12) and U = 2 (hidden <= 8): unit U q + e for e < U; elements U .. 3 of every per-unit vector, rows / columns
// 4 q + U .. 4 q + 3 of every operand tile and those bytes of the packed int8 operands are DEAD (zero weights, never written out), and the
// kernels do not compute them: a quarter (half) of the per-unit arithmetic of the cell, the head and the backward goes away, and so do the
// dead K chunks of every mat-vec over units (hidden 10, the reference's QGRU: 12 unit slots per 16 sequences instead of 16).
// q16_unit answers `H` for "no unit".
template <int U>
__host__ __device__ __forceinline__ int q16_unit(int kt, int j, int H) {
    if (U == 4) return 16 * kt + j;
    return (j & 3) < U ? U * (j >> 2) + (j & 3) : H;
}
#define Q16_EACHU _Pragma("unroll") for (int i = 0; i < U; ++i)

template <int MK, int NT, int U>
__device__ __forceinline__ float4 q16_entry(const float* pl, const QatLayout& L, const WQ& wq, int grp, int m, int q) {
    using T = QT<MK, NT>;
    using K = Kind<MK>;
    constexpr int F = K::F;
    const int H = L.H;
    static_assert(U == 4 || NT == 1, "fewer than four units per lane: one unit tile only");
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (NT == 1 && (grp == T::I8W || grp == T::I8W + 1)) {
        // v_mfma_i32_16x16x32_i8 A operands of lane (m, q): byte j <-> K index 8 q + j.  h part: bytes 0..3 = k_w(W_h)[g][m][4q+j];
        // x part: bytes 4, 5 = k_w(W_x)[g][m][slot 4c+q], c = 0, 1.  Entry = {wh_r, wx_r, wh_z, wx_z} | {wh_n, wx_n, 0, 0} as bit patterns.
        auto byte_of = [](float kf) { return (unsigned)((int)kf) & 0xffu; };
        auto wh = [&](int g) {
            unsigned w = 0;
            const int o = q16_unit<U>(0, m, H);
            for (int j = 0; j < 4; ++j) { const int u = q16_unit<U>(0, 4 * q + j, H); if (o < H && u < H) w |= byte_of(kq(pl[L.o_wh + (g * H + o) * H + u], wq.h)) << (8 * j); }
            return w;
        };
        auto wx = [&](int g) {
            unsigned w = 0;
            const int o = q16_unit<U>(0, m, H);
            for (int c = 0; c < K::NCH; ++c) { const int slot = 4 * c + q; if (o < H && slot < F) w |= byte_of(kq(pl[L.o_wx + (g * H + o) * F + slot], wq.x)) << (8 * c); }
            return w;
        };
        if (grp == T::I8W) return make_float4(__uint_as_float(wh(0)), __uint_as_float(wx(0)), __uint_as_float(wh(1)), __uint_as_float(wx(1)));
        return make_float4(__uint_as_float(wh(2)), __uint_as_float(wx(2)), 0.0f, 0.0f);
    }
    if (K::DGRU && grp == T::WOF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int cc = e >> 1, slot = 4 * (e & 1) + q;
            v[e] = slot < 6 ? kq(pl[L.o_wo + cc * L.OW + H + slot], wq.o) : 0.0f;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (grp < T::HH) {
            const int g = grp / NT, o = q16_unit<U>(grp % NT, m, H), slot = 4 * e + q;
            v[e] = (e < K::NCH && slot < F && o < H) ? kq(pl[L.o_wx + (g * H + o) * F + slot], wq.x) : 0.0f;
        } else if (grp < T::BX) {
            const int r = grp - T::HH, g = r / (NT * NT), o = q16_unit<U>((r / NT) % NT, m, H), k = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = (o < H && k < H) ? kq(pl[L.o_wh + (g * H + o) * H + k], wq.h) : 0.0f;
        } else if (grp < T::BH) {
            const int r = grp - T::BX, u = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = u < H ? pl[L.o_bx + (r / NT) * H + u] : 0.0f;
        } else if (grp < T::WOUT) {
            const int r = grp - T::BH, u = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = u < H ? pl[L.o_bh + (r / NT) * H + u] : 0.0f;
        } else if (grp < T::HID) {
            const int r = grp - T::WOUT, u = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = u < H ? kq(pl[L.o_wo + (r / NT) * L.OW + u], wq.o) : 0.0f;
        } else if (grp < T::BHID) {
            const int r = grp - T::HID, o = q16_unit<U>(r / NT, m, H), k = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = (o < H && k < H) ? kq(pl[L.o_whid + o * H + k], wq.hid) : 0.0f;
        } else if (grp < T::WOF) {
            const int u = q16_unit<U>(grp - T::BHID, 4 * q + e, H);
            v[e] = u < H ? pl[L.o_bhid + u] : 0.0f;
        } else if (grp < T::HIDT) {
            const int r = grp - T::HHT, g = r / (NT * NT), i = q16_unit<U>((r / NT) % NT, m, H), k = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = (i < H && k < H) ? kq(pl[L.o_wh + (g * H + k) * H + i], wq.h) : 0.0f;
        } else if (grp < T::IHT) {
            const int r = grp - T::HIDT, i = q16_unit<U>(r / NT, m, H), k = q16_unit<U>(r % NT, 4 * q + e, H);
            v[e] = (i < H && k < H) ? kq(pl[L.o_whid + k * H + i], wq.hid) : 0.0f;
        } else {
            // transposed input weights with the output rows permuted so that D row 4 q' + i = slot 4 i + q': the MFMA result of
            // lane (n, q) element c IS the gradient of the lane's own feature slot 4 c + q
            const int r = grp - T::IHT, g = r / NT, k = q16_unit<U>(r % NT, 4 * q + e, H), slot = 4 * (m & 3) + (m >> 2);
            v[e] = (slot < F && k < H) ? kq(pl[L.o_wx + (g * H + k) * F + slot], wq.x) : 0.0f;
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// ---- features on the lane's slots ------------------------------------------------------------------------------------
// fs[c] = feature 4c+q of the lane's sequence (0 beyond F), computed with the reference's operation order (torch.pow(i,2) +
// torch.pow(q,2), sqrt, pow(amp,3) = a*a*a, true divisions) — the values are about to be rounded onto a grid
template <int MK>
__device__ __forceinline__ void q16_slots(float2 xv, float2 xn, const float (&oh)[4], float (&fs)[Kind<MK>::NCH]) {
    const float I = xv.x, Q = xv.y;
    if constexpr (MK == K_GRU) {
        fs[0] = oh[0] * I + oh[1] * Q;
    } else if constexpr (MK == K_Q4) {
        const float a2 = I * I + Q * Q;
        fs[0] = (oh[0] * I + oh[1] * Q) + (oh[2] * a2 + oh[3] * (a2 * a2));
    } else {
        const float a2 = I * I + Q * Q, a = sqrtf(a2), a3 = a * a * a;
        fs[0] = (oh[0] * I + oh[1] * Q) + (oh[2] * a + oh[3] * a3);
        if constexpr (MK == K_DGRU) fs[1] = oh[0] * (Q / a) + oh[1] * (I / a);        // sin, cos (dgru.py:66-68)
        if constexpr (MK == K_TRES) fs[1] = oh[0] * xn.x + oh[1] * xn.y;              // torch.roll(x, -1) (deltagru_tcnskip.py:91-100)
    }
}

// ---- integer matrix pipe (W8A8, one unit tile) -------------------------------------------------------------------------
// The mat-vecs of the forward / recompute run on v_mfma_i32_16x16x32_i8 when weights AND activations are 8-bit grids: K = 32 covers the
// lane quad's 4 own units (bytes 0..3 of the quad's 8-byte K group) and its feature slots (bytes 4, 5); lane (n, q) feeds its own packed
// k_a as the B operand — no cross-lane movement, as in the fp32 mapping — and the A operand holds k_w of EITHER the h part OR the x part
// (the two sums are needed separately: each gets its own scale and fp32 bias).  6 instructions of 16 cycles on the matrix pipe instead of
// 15 f32 MFMAs of 32 cycles on the FMA lanes; int32 accumulation is exact.  (tools/probes/mfma_i32_16x16x32_i8_layout.hip: layout, cost.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
// (kRneMagic = 1.5 x 2^23, odpd_qat.h: (v + magic) rounds v to the nearest-even integer, which sits in the low mantissa bits)
__device__ __forceinline__ int i8_bits(float clamped) { return __builtin_bit_cast(int, clamped + kRneMagic); }       // low byte = int8 of rint(v)
__device__ __forceinline__ int i8_pack4(int b0, int b1, int b2, int b3) {
    const unsigned lo = __builtin_amdgcn_perm((unsigned)b1, (unsigned)b0, 0x0c0c0400u), hi = __builtin_amdgcn_perm((unsigned)b3, (unsigned)b2, 0x04000c0cu);
    return (int)(lo | hi);
}
__device__ __forceinline__ long i8_operand(int lo, int hi) { return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo); }
// xs[g], hs[g] (g = r, z, n) of the lane's four units from the packed activations: bh = its 4 own units, bx = its (<= 2) feature slots
__device__ __forceinline__ void i8_matvecs(TabPtr tl, int base, int bh, int bx, f32x4 (&xs)[3], f32x4 (&hs)[3]) {
    const float4 w0 = tab_ld(tl, base * 64), w1 = tab_ld(tl, (base + 1) * 64);
    const int wh[3] = {__builtin_bit_cast(int, w0.x), __builtin_bit_cast(int, w0.z), __builtin_bit_cast(int, w1.x)};
    const int wx[3] = {__builtin_bit_cast(int, w0.y), __builtin_bit_cast(int, w0.w), __builtin_bit_cast(int, w1.y)};
    const long b = i8_operand(bh, bx);
    // (r06, tried: accumulators seeded with the bit pattern of 1.5 x 2^23, so that the result IS a float and a fast-class v_sub_f32 replaces the
    // v_cvt_f32_i32.  Every W8A8 result came out wrong; cause not established — the ISA looked as intended.  Zero accumulators, an inline
    // constant, stay.)
    const i32x4 z4 = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const i32x4 dh = __builtin_amdgcn_mfma_i32_16x16x32_i8(i8_operand(wh[g], 0), b, z4, 0, 0, 0);
        const i32x4 dx = __builtin_amdgcn_mfma_i32_16x16x32_i8(i8_operand(0, wx[g]), b, z4, 0, 0, 0);
        ODPD_EACH4 { hs[g][i] = (float)dh[i]; xs[g][i] = (float)dx[i]; }
    }
}

// ---- GRUCell step ----------------------------------------------------------------------------------------------------
// what the backward of one step needs, with the straight-through masks already multiplied into the factors they gate:
//   c2 = p_ah p_m2, c3 = p_ah p_m3, An = p_n tanh' p_an, Az = p_z sig' p_az, B1 = p_m1 r, B2A = p_m1 h_n (p_r sig' p_ar),
//   pph = p(q_a(h)) s_hw  (the weight scale of the transposed mat-vec rides on the mask)
template <int NT> struct SaveS { f32x4 hp[NT], hqk[NT], n[NT], z[NT], c2[NT], c3[NT], An[NT], Az[NT], B1[NT], B2A[NT], pph[NT], hnew[NT]; };

template <int MK, int NT, int U, bool LUT, bool SAVE>
__device__ __forceinline__ void std_cell(TabPtr tl, const QSc& qs, const QK& k, const float4* lutq, const float (&fqk)[Kind<MK>::NCH],
                                         f32x4 (&h)[NT], SaveS<NT>& sv) {
    using T = QT<MK, NT>;
    constexpr int NCH = Kind<MK>::NCH;
    constexpr bool I8 = LUT && NT == 1;       // 8-bit weights and activations, one unit tile: the integer matrix pipe
    const unsigned lbias = LUT ? lut_bias(lutq) : 0u;
    f32x4 hqk[NT];
    int hb[4] = {0, 0, 0, 0};
    if constexpr (U < 4) {                    // dead elements: constants (whatever reads a whole vector — tile stores, checkpoints — sees zeros)
        hqk[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SAVE) sv = SaveS<NT>();
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
        Q16_EACHU {
            const float v = h[kt][i] * k.inv_ha, m = gm(v, k);
            if constexpr (I8) {
                hb[i] = i8_bits(m);
                if constexpr (SAVE) hqk[kt][i] = __builtin_bit_cast(float, hb[i]) - kRneMagic;
            } else {
                hqk[kt][i] = rintf(m);
            }
            if constexpr (SAVE) { sv.hqk[kt][i] = hqk[kt][i]; sv.pph[kt][i] = m == v ? k.s_hw : 0.0f; }
        }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 xs[3][NT], hs[3][NT];
    if constexpr (I8) {
        f32x4 x1[3], h1[3];
        const int bx = NCH > 1 ? (int)__builtin_amdgcn_perm((unsigned)i8_bits(fqk[NCH - 1]), (unsigned)i8_bits(fqk[0]), 0x0c0c0400u)
                               : (i8_bits(fqk[0]) & 0xff);
        i8_matvecs(tl, T::I8W, i8_pack4(hb[0], hb[1], hb[2], hb[3]), bx, x1, h1);
#pragma unroll
        for (int g = 0; g < 3; ++g) { xs[g][0] = x1[g]; hs[g][0] = h1[g]; }
    } else {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const float4 w = tab_ld(tl, (T::IH + g * NT + mt) * 64);
            xs[g][mt] = mfma4(w.x, fqk[0], z4);
            if constexpr (NCH > 1) xs[g][mt] = mfma4(w.y, fqk[1], xs[g][mt]);
            hs[g][mt] = z4;
        }
        s16n_matvec<NT, U>(tl, T::HH + g * NT * NT, hqk, hs[g]);
    }
    }
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const f32x4 Bxr = as_f32x4(tab_ld(tl, (T::BX + 0 * NT + mt) * 64)), Bxz = as_f32x4(tab_ld(tl, (T::BX + 1 * NT + mt) * 64)),
                    Bxn = as_f32x4(tab_ld(tl, (T::BX + 2 * NT + mt) * 64)), Bhr = as_f32x4(tab_ld(tl, (T::BH + 0 * NT + mt) * 64)),
                    Bhz = as_f32x4(tab_ld(tl, (T::BH + 1 * NT + mt) * 64)), Bhn = as_f32x4(tab_ld(tl, (T::BH + 2 * NT + mt) * 64));
        Q16_EACHU {
            const float hv = h[mt][i];
            // x_t = x2h(x), h_t = h2h(h): exact integer sums, scale and fp32 bias in one FMA (== F.linear's result)
            const float xr = __builtin_fmaf(xs[0][mt][i], k.Sx, Bxr[i]), hr = __builtin_fmaf(hs[0][mt][i], k.Sh, Bhr[i]);
            const float xz = __builtin_fmaf(xs[1][mt][i], k.Sx, Bxz[i]), hz = __builtin_fmaf(hs[1][mt][i], k.Sh, Bhz[i]);
            const float xn = __builtin_fmaf(xs[2][mt][i], k.Sx, Bxn[i]), hn = __builtin_fmaf(hs[2][mt][i], k.Sh, Bhn[i]);
            const float vr = (xr + hr) * k.inv_add, vz = (xz + hz) * k.inv_add;
            const float mr = gm(vr, k), mz = gm(vz, k);
            const Gate Gr = sig_grid_m<LUT>(mr, qs, k, lutq, lbias), Gz = sig_grid_m<LUT>(mz, qs, k, lutq, lbias);
            const float pm1 = Gr.c * hn, mm1 = gm(pm1, k);                                     // Qmul(r h_n)
            const float vn = __builtin_fmaf(rintf(mm1), k.s_mul, xn) * k.inv_add, mn = gm(vn, k);      // Qadd(x_n + .)
            const Gate Gn = tanh_grid_m<LUT>(mn, qs, k, lutq, lbias);
            const float omz = __builtin_fmaf(Gz.c, -k.s_mul, 1.0f);                             // 1 - z, plain
            const float pm2 = Gz.c * hv, pm3 = omz * Gn.c;
            const float mm2 = gm(pm2, k), mm3 = gm(pm3, k);
            const float vh = (rintf(mm2) + rintf(mm3)) * k.c_ma, mh = gm(vh, k);
            const float hnew = rintf(mh) * k.s_add;
            if constexpr (SAVE) {
                const bool pah = mh == vh;
                sv.c2[mt][i] = (pah && mm2 == pm2) ? 1.0f : 0.0f;
                sv.c3[mt][i] = (pah && mm3 == pm3) ? 1.0f : 0.0f;
                sv.An[mt][i] = mn == vn ? Gn.d : 0.0f;
                sv.Az[mt][i] = mz == vz ? Gz.d : 0.0f;
                const float Ar = mr == vr ? Gr.d : 0.0f;
                const bool p1 = mm1 == pm1;
                sv.B1[mt][i] = p1 ? Gr.c * k.s_mul : 0.0f;
                sv.B2A[mt][i] = p1 ? hn * Ar : 0.0f;
                sv.hp[mt][i] = hv; sv.n[mt][i] = Gn.c * k.s_mul; sv.z[mt][i] = Gz.c * k.s_mul; sv.hnew[mt][i] = hnew;
            }
            h[mt][i] = hnew;
        }
    }
}

// ---- delta cell step -------------------------------------------------------------------------------------------------
template <int NT> struct StateD { f32x4 h[NT], hp[NT], dmr[NT], dmz[NT], dmn[NT], dmnh[NT]; float xp[2]; };
//   npo = n p_omz, mh = threshold keep mask of dh, pph = p(q_a(dh)) s_hw; mx / px likewise for the two feature slots (px carries s_xw)
template <int NT> struct SaveD {
    f32x4 hp[NT], qdhk[NT], n[NT], npo[NT], omz[NT], z[NT], c2[NT], c3[NT], An[NT], Az[NT], B1[NT], B2A[NT], mh[NT], pph[NT], hnew[NT];
    float fqk[2], mx[2], px[2];
};

template <int NT, bool LUT, bool SAVE>
__device__ __forceinline__ void delta_cell(TabPtr tl, const QSc& qs, const QK& k, const float4* lutq, const float* thr, int K,
                                           const float (&fs)[2], float thx, float thh, const bool (&slot_ok)[2], const f32x4 (&unit_ok)[NT],
                                           StateD<NT>& st, SaveD<NT>& sv, float& zx, float& zh) {
    using T = QT<K_TRES, NT>;
    float fqk[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float d = fs[c] - st.xp[c];
        const bool keep = !(__builtin_fabsf(d) < thx);                 // masked_fill(|d| < th, 0)  (deltagru_tcnskip.py:218-228)
        const float dxm = keep ? d : 0.0f;
        st.xp[c] = (__builtin_fabsf(d) >= thx) ? fs[c] : st.xp[c];
        zx += (slot_ok[c] && dxm == 0.0f) ? 1.0f : 0.0f;
        const float v = dxm * k.inv_xa, m = gm(v, k);
        fqk[c] = rintf(m);
        if constexpr (SAVE) { sv.fqk[c] = fqk[c]; sv.mx[c] = keep ? 1.0f : 0.0f; sv.px[c] = m == v ? k.s_xw : 0.0f; }
    }
    constexpr bool I8 = LUT && NT == 1;       // 8-bit weights and activations, one unit tile: the integer matrix pipe
    f32x4 qdhk[NT];
    int hb[4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
        ODPD_EACH4 {
            const float d = st.h[kt][i] - st.hp[kt][i];
            const bool keep = !(__builtin_fabsf(d) < thh);
            const float dhm = keep ? d : 0.0f;
            st.hp[kt][i] = (__builtin_fabsf(d) >= thh) ? st.h[kt][i] : st.hp[kt][i];
            zh += (unit_ok[kt][i] != 0.0f && dhm == 0.0f) ? 1.0f : 0.0f;
            const float v = dhm * k.inv_ha, m = gm(v, k);
            if constexpr (I8) {
                hb[i] = i8_bits(m);
                if constexpr (SAVE) qdhk[kt][i] = __builtin_bit_cast(float, hb[i]) - kRneMagic;
            } else {
                qdhk[kt][i] = rintf(m);
            }
            if constexpr (SAVE) { sv.qdhk[kt][i] = qdhk[kt][i]; sv.mh[kt][i] = keep ? 1.0f : 0.0f; sv.pph[kt][i] = m == v ? k.s_hw : 0.0f; }
        }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 xs[3][NT], hs[3][NT];
    if constexpr (I8) {
        f32x4 x1[3], h1[3];
        const int bx = (int)__builtin_amdgcn_perm((unsigned)i8_bits(fqk[1]), (unsigned)i8_bits(fqk[0]), 0x0c0c0400u);
        i8_matvecs(tl, T::I8W, i8_pack4(hb[0], hb[1], hb[2], hb[3]), bx, x1, h1);
#pragma unroll
        for (int g = 0; g < 3; ++g) { xs[g][0] = x1[g]; hs[g][0] = h1[g]; }
    } else {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const float4 w = tab_ld(tl, (T::IH + g * NT + mt) * 64);
            xs[g][mt] = mfma4(w.x, fqk[0], z4);
            xs[g][mt] = mfma4(w.y, fqk[1], xs[g][mt]);
            hs[g][mt] = z4;
        }
        s16n_matvec<NT>(tl, T::HH + g * NT * NT, qdhk, hs[g]);
    }
    }
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
        ODPD_EACH4 {
            const float hv = st.h[mt][i];
            // mac_x = x2h(dx) + dm; dm_r = mac_x_r + mac_h_r, dm_n = mac_x_n, dm_nh = mac_h_n + dm_nh  (deltagru_tcnskip.py:236-246)
            const float dmr = __builtin_fmaf(hs[0][mt][i], k.Sh, __builtin_fmaf(xs[0][mt][i], k.Sx, st.dmr[mt][i]));
            const float dmz = __builtin_fmaf(hs[1][mt][i], k.Sh, __builtin_fmaf(xs[1][mt][i], k.Sx, st.dmz[mt][i]));
            const float dmn = __builtin_fmaf(xs[2][mt][i], k.Sx, st.dmn[mt][i]);
            const float dmnh = __builtin_fmaf(hs[2][mt][i], k.Sh, st.dmnh[mt][i]);
            st.dmr[mt][i] = dmr; st.dmz[mt][i] = dmz; st.dmn[mt][i] = dmn; st.dmnh[mt][i] = dmnh;
            const Gate Gr = sig_any<LUT>(dmr, qs, k, thr, K), Gz = sig_any<LUT>(dmz, qs, k, thr, K);
            const float pm1 = Gr.c * dmnh, mm1 = gm(pm1, k);
            const float vn = __builtin_fmaf(rintf(mm1), k.s_mul, dmn) * k.inv_add, mn = gm(vn, k);
            const Gate Gn = tanh_grid<LUT>(rintf(mn), qs, k, lutq);
            const float vo = __builtin_fmaf(Gz.c, -k.s_mul, 1.0f) * k.inv_add, mo = gm(vo, k);          // self.add(1, -gate_z)  (deltagru_tcnskip.py:290)
            const float omz = rintf(mo) * k.s_add;
            const float pm3 = omz * Gn.c, pm2 = Gz.c * hv;
            const float mm3 = gm(pm3, k), mm2 = gm(pm2, k);
            const float vh = (rintf(mm3) + rintf(mm2)) * k.c_ma, mhv = gm(vh, k);
            const float hnew = rintf(mhv) * k.s_add;
            if constexpr (SAVE) {
                const bool pah = mhv == vh;
                sv.c2[mt][i] = (pah && mm2 == pm2) ? 1.0f : 0.0f;
                sv.c3[mt][i] = (pah && mm3 == pm3) ? 1.0f : 0.0f;
                sv.An[mt][i] = mn == vn ? Gn.d : 0.0f;
                sv.Az[mt][i] = Gz.d;
                const bool p1 = mm1 == pm1;
                sv.B1[mt][i] = p1 ? Gr.c * k.s_mul : 0.0f;
                sv.B2A[mt][i] = p1 ? dmnh * Gr.d : 0.0f;
                const float nt = Gn.c * k.s_mul;
                sv.hp[mt][i] = hv; sv.n[mt][i] = nt; sv.npo[mt][i] = mo == vo ? nt : 0.0f; sv.omz[mt][i] = omz; sv.z[mt][i] = Gz.c * k.s_mul;
                sv.hnew[mt][i] = hnew;
            }
            st.h[mt][i] = hnew;
        }
}

// ---- heads -----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float q16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float q16_hsg(float v) { return v < -3.0f ? 0.0f : (v <= 3.0f ? v * (1.0f / 3.0f) + 0.5f : 1.0f); }
template <int MK>
struct Scalars {                 // wave-uniform parameters
    float bout[2], w1[18], w2[6];
    __device__ __forceinline__ void load(const float* pl, const QatLayout& L) {
        constexpr bool TRES = Kind<MK>::TRES;
        bout[0] = TRES ? 0.0f : q16_uni(pl[L.o_bo]);
        bout[1] = TRES ? 0.0f : q16_uni(pl[L.o_bo + 1]);
#pragma unroll
        for (int i = 0; i < 18; ++i) w1[i] = TRES ? q16_uni(pl[L.o_tcn0 + i]) : 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) w2[i] = TRES ? q16_uni(pl[L.o_tcn2 + i]) : 0.0f;
    }
};
// TCN skip of one sample: s1[3] pre-activations of the first conv, s2[2] of the second (float path, Conv1d / Hardswish are not swapped)
template <int MK>
__device__ __forceinline__ void q16_tcn(const Scalars<MK>& sc, float2 xm, float2 xc, float2 xq, float (&s1)[3], float (&s2)[2]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = sc.w1[c * 6] * xm.x;
        v = __builtin_fmaf(sc.w1[c * 6 + 1], xc.x, v); v = __builtin_fmaf(sc.w1[c * 6 + 2], xq.x, v);
        v = __builtin_fmaf(sc.w1[c * 6 + 3], xm.y, v); v = __builtin_fmaf(sc.w1[c * 6 + 4], xc.y, v);
        s1[c] = __builtin_fmaf(sc.w1[c * 6 + 5], xq.y, v);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        float v = sc.w2[o * 3] * hardswishf_(s1[0]);
        v = __builtin_fmaf(sc.w2[o * 3 + 1], hardswishf_(s1[1]), v);
        s2[o] = __builtin_fmaf(sc.w2[o * 3 + 2], hardswishf_(s1[2]), v);
    }
}

// everything the head's backward needs from its forward (recomputed there); masks carry the weight scale of the way back
template <int MK, int NT> struct HeadOut {
    f32x4 hok[NT];           // q_a(fc_out input) on the units (dgru: of relu(fc_hid)), grid units
    f32x4 pho[NT];           // its pass mask * s_ow
    f32x4 h2k[NT], ph2[NT], hidpre[NT];     // dgru: q_a(h') of fc_hid (grid units), mask * s_hidw, pre-activation
    float cofk[2], pcof[2];  // dgru: q_a(feature slot) of fc_out, mask * s_ow
};
// y (scaled, before bias / output quantiser / skip) of one step; `fs` = the lane's FLOAT feature slots (dgru's cat)
template <int MK, int NT, int U, bool SAVE>
__device__ __forceinline__ void head_fwd(TabPtr tl, const QK& k, const f32x4 (&h)[NT], const float (&fs)[Kind<MK>::NCH], HeadOut<MK, NT>& ho,
                                         float& y0, float& y1) {
    using T = QT<MK, NT>;
    float p0 = 0.0f, p1 = 0.0f;
    if constexpr (U < 4) ho = HeadOut<MK, NT>();
    if constexpr (Kind<MK>::DGRU) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        f32x4 acc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            acc[kt] = z4;
            Q16_EACHU {
                const float v = h[kt][i] * k.inv_hida, m = gm(v, k);
                ho.h2k[kt][i] = rintf(m);
                if constexpr (SAVE) ho.ph2[kt][i] = m == v ? k.s_hidw : 0.0f;
            }
        }
        s16n_matvec<NT, U>(tl, T::HID, ho.h2k, acc);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const f32x4 b = as_f32x4(tab_ld(tl, (T::BHID + mt) * 64));
            const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
            Q16_EACHU {
                const float pre = __builtin_fmaf(acc[mt][i], k.Shid, b[i]), hid = pre > 0.0f ? pre : 0.0f;      // torch.relu
                const float v = hid * k.inv_oa, m = gm(v, k);
                ho.hok[mt][i] = rintf(m);
                if constexpr (SAVE) { ho.hidpre[mt][i] = pre; ho.pho[mt][i] = m == v ? k.s_ow : 0.0f; }
                p0 = __builtin_fmaf(w0[i], ho.hok[mt][i], p0); p1 = __builtin_fmaf(w1[i], ho.hok[mt][i], p1);
            }
        }
        const float4 wf = tab_ld(tl, T::WOF * 64);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float v = fs[c] * k.inv_oa, m = gm(v, k);
            ho.cofk[c] = rintf(m);
            if constexpr (SAVE) ho.pcof[c] = m == v ? k.s_ow : 0.0f;
        }
        p0 = __builtin_fmaf(wf.x, ho.cofk[0], p0); p0 = __builtin_fmaf(wf.y, ho.cofk[1], p0);
        p1 = __builtin_fmaf(wf.z, ho.cofk[0], p1); p1 = __builtin_fmaf(wf.w, ho.cofk[1], p1);
    } else {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
            Q16_EACHU {
                const float v = h[mt][i] * k.inv_oa, m = gm(v, k);
                ho.hok[mt][i] = rintf(m);
                if constexpr (SAVE) ho.pho[mt][i] = m == v ? k.s_ow : 0.0f;
                p0 = __builtin_fmaf(w0[i], ho.hok[mt][i], p0); p1 = __builtin_fmaf(w1[i], ho.hok[mt][i], p1);
            }
        }
    }
    y0 = quad_sum(p0); y1 = quad_sum(p1);      // exact integer sums on 8-bit grids: any order
}

// staging geometry of the FORWARD kernels.  GRUCell kinds (r06): 16-step chunks — a wave's LDS share falls 8.4 -> 4.4 KB, so that TWO eight-wave
// workgroups fit a CU next to the operand table and the gate LUT (90 KB per workgroup before: the kernels compiled for four waves per SIMD ran at
// two).  The delta cell keeps the 32-step chunks of its TCN halo.
template <int MK> struct FwdGeo {
    static constexpr int CH = Kind<MK>::TRES ? kChunk : 16, XROW = CH + 2 * Kind<MK>::HALO + 1, YROW = CH + 1;
    static constexpr int kWaveF = 2 * 16 * XROW + 2 * 16 * YROW;
};
template <int MK, int CH = kChunk>
__device__ __forceinline__ void q16_stage_x(float2* lds, const float* g, int b0, int B, int T, int t0, int lane, const long long* fidx = nullptr,
                                            int fstride = 0) {
    using K = Kind<MK>;
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int PER = CH + 2 * K::HALO, TOT = 16 * PER, N = (TOT + 63) / 64, XROW = CH + 2 * K::HALO + 1;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        if (e < TOT) {
            const int m = e / PER, pos = e % PER, tg = t0 - K::HALO + pos;
            float2 v = make_float2(0.0f, 0.0f);          // outside the frame: the conv's zero padding
            if (tg >= 0 && tg < T) v = (b0 + m < B) ? g2[(fidx ? (size_t)fidx[b0 + m] * fstride : (size_t)(b0 + m) * T) + tg] : make_float2(0.5f, 0.5f);
            else if (!K::TRES) v = make_float2(0.5f, 0.5f);
            lds[m * XROW + pos] = v;
        }
    }
}
template <int CH>
__device__ __forceinline__ void q16_stage_out(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
    constexpr int N = 16 * CH / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH;
        if (tt < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + tt] = lds[m * (CH + 1) + tt];
    }
}
__device__ __forceinline__ float4 q16_f4(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }

template <int NT>
__device__ __forceinline__ void init_state(StateD<NT>& st) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) { st.h[mt] = z4; st.hp[mt] = z4; st.dmr[mt] = z4; st.dmz[mt] = z4; st.dmn[mt] = z4; st.dmnh[mt] = z4; }
    st.xp[0] = st.xp[1] = 0.0f;
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <int MK, int NT, bool LUT, int U>
__device__ __forceinline__ void q16_fwd_body(const SeqArgs& a, int bits_w, int bits_a, int eval_mode) {
    using T = QT<MK, NT>;
    using K = Kind<MK>;
    constexpr int S = T::S, NCH = K::NCH;
    using Geo = FwdGeo<MK>;
    constexpr int kWaveF = Geo::kWaveF, CH = Geo::CH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const QatLayout L = qat_layout(MK, a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    const QSc qs = load_qsc<MK>(pl, L, bits_a);
    const WQ wq = make_wq(pl, L, bits_w);
    const QK k = make_qk(qs, wq);
    const int nlut = LUT ? (1 << bits_a) : 0;
    float* lut = tab + s16_tab_floats(T::NG_FWD);
    {
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int grp = wave; grp < T::NG_FWD; grp += nwb) t4[grp * 64 + lane] = q16_entry<MK, NT, U>(pl, L, wq, grp, n, q);
        if constexpr (LUT) fill_luts(lut, qs, k, bits_a, K::TRES);
        __syncthreads();
    }
    const int Ksig = sig_levels(qs.sig);
    const float4* lutq = reinterpret_cast<const float4*>(lut) - (int)qs.add.qn;
    const float* thr = lut + 4 * nlut;
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    Scalars<MK> sc;
    sc.load(pl, L);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    const bool slot_ok[2] = {true, q < 2};
    f32x4 unit_ok[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) ODPD_EACH4 unit_ok[kt][i] = (q16_unit<U>(kt, 4 * q + i, a.H) < a.H) ? 1.0f : 0.0f;
    float* wbase = lut + (LUT ? 4 * nlut + kMaxThr : 0) + (size_t)wave * kWaveF;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * Geo::XROW;
    const float2* xr = xs + n * Geo::XROW + K::HALO;
    float zx = 0.0f, zh = 0.0f;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * T::kCk * 64 + lane : nullptr;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[a.frame_idx ? (size_t)a.frame_idx[b0 + n] * a.frame_stride : (size_t)(b0 + n) * a.T] : make_float2(0.5f, 0.5f);
        StateD<NT> st;
        init_state<NT>(st);
        float zxs = 0.0f, zhs = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += CH) {
            const int len = min(CH, a.T - t0);
            wave_lds_fence();
            q16_stage_x<MK, CH>(xs, a.x, b0, a.B, a.T, t0, lane, a.frame_idx, a.frame_stride);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xr[tt];
                float2 xn = make_float2(0.f, 0.f);
                if constexpr (K::TRES) xn = (t0 + tt + 1 < a.T) ? xr[tt + 1] : x0;      // torch.roll(x, -1): the last step sees sample 0
                float fs[NCH];
                q16_slots<MK>(xv, xn, oh, fs);
                const TabPtr tlo = opaque(tl);
                if constexpr (K::TRES) {
                    SaveD<NT> sv;
                    delta_cell<NT, LUT, false>(tlo, qs, k, lutq, thr, Ksig, fs, a.thx, a.thh, slot_ok, unit_ok, st, sv, zxs, zhs);
                } else {
                    float fqk[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) fqk[c] = gk(fs[c] * k.inv_xa, k);
                    SaveS<NT> sv;
                    std_cell<MK, NT, U, LUT, false>(tlo, qs, k, lutq, fqk, st.h, sv);
                }
                if (a.y != nullptr) {      // (the fused train step's forward launch wants the checkpoints only: no head)
                    HeadOut<MK, NT> ho;
                    float y0, y1;
                    head_fwd<MK, NT, U, false>(tlo, k, st.h, fs, ho, y0, y1);
                    y0 = __builtin_fmaf(y0, k.So, sc.bout[0]); y1 = __builtin_fmaf(y1, k.So, sc.bout[1]);
                    if (eval_mode) { y0 = qapply(y0, qs.out); y1 = qapply(y1, qs.out); }   // fc_out's 16-bit out_quantizer (quant_layers.py:77-80)
                    if constexpr (K::TRES) {
                        float s1[3], s2[2];
                        q16_tcn<MK>(sc, xr[tt - kHalo], xv, xr[tt + kHalo], s1, s2);
                        y0 += hardswishf_(s2[0]); y1 += hardswishf_(s2[1]);
                    }
                    if (q == 0) ys[n * Geo::YROW + tt] = make_float2(y0, y1);
                }
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    float4* c = ck + (size_t)(t1 / S) * T::kCk * 64;
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        c[kt * 64] = q16_f4(st.h[kt]);
                        if constexpr (K::TRES) {
                            c[(1 * NT + kt) * 64] = q16_f4(st.hp[kt]);
                            c[(2 * NT + kt) * 64] = q16_f4(st.dmr[kt]); c[(3 * NT + kt) * 64] = q16_f4(st.dmz[kt]);
                            c[(4 * NT + kt) * 64] = q16_f4(st.dmn[kt]); c[(5 * NT + kt) * 64] = q16_f4(st.dmnh[kt]);
                        }
                    }
                    if constexpr (K::TRES) c[6 * NT * 64] = make_float4(st.xp[0], st.xp[1], 0.0f, 0.0f);
                }
            }
            wave_lds_fence();
            if (a.y != nullptr) q16_stage_out<CH>(ys, a.y, b0, a.B, a.T, t0, len, lane);      // (the fused train step needs the checkpoints only)
        }
        if (valid) { zx += zxs; zh += zhs; }
    }
    if constexpr (K::TRES) {
        if (a.stats != nullptr) {
            float tx = zx, th = zh;
            for (int o = 32; o > 0; o >>= 1) { tx += __shfl_down(tx, o); th += __shfl_down(th, o); }
            if (lane == 0) {
                atomicAdd(&a.stats[0], (double)tx);
                atomicAdd(&a.stats[2], (double)th);
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                atomicAdd(&a.stats[1], 6.0 * (double)a.B * (double)a.T);
                atomicAdd(&a.stats[3], (double)a.H * (double)a.B * (double)a.T);
            }
        }
    }
}
template <int MK, int NT, bool LUT>
__global__ __launch_bounds__(512) void qat16_fwd_kernel(SeqArgs a, int bits_w, int bits_a, int eval_mode) {
    q16_fwd_body<MK, NT, LUT, 4>(a, bits_w, bits_a, eval_mode);
}
// U = 3 / 2 units per lane: one unit tile, hidden <= 12 / <= 8 (see q16_unit)
template <int MK, bool LUT, int U>
__global__ __launch_bounds__(512) void qat16u_fwd_kernel(SeqArgs a, int bits_w, int bits_a, int eval_mode) {
    q16_fwd_body<MK, 1, LUT, U>(a, bits_w, bits_a, eval_mode);
}

// -------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------
// weight-gradient accumulators hold sum(d (x) k_a) with the activation in grid units: the activation scale is applied once, at write-out
template <int MK, int NT>
struct Grad {
    f32x4 tih[3][NT], thh[3][NT][NT];
    f32x4 dwout[2][NT], dbhn[NT];
    f32x4 thid[NT][NT], dbhid[NT];
    float dbout[2], dwof[2][2], dw1[18], dw2[6];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            dwout[0][a] = dwout[1][a] = dbhn[a] = dbhid[a] = z4;
#pragma unroll
            for (int b = 0; b < NT; ++b) thid[a][b] = z4;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                tih[g][a] = z4;
#pragma unroll
                for (int b = 0; b < NT; ++b) thh[g][a][b] = z4;
            }
        }
        dbout[0] = dbout[1] = dwof[0][0] = dwof[0][1] = dwof[1][0] = dwof[1][1] = 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i) dw1[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) dw2[i] = 0.f;
    }
};
template <int NT> struct Carry { f32x4 gh[NT], ghp[NT], gr[NT], gz[NT], gn[NT], gnh[NT]; float gxp[2], wrap[2]; };

// head backward of one step: accumulates the head's parameter gradients, adds dL/dh' to gh and, for dgru, returns the head's
// share of dL/d(feature slot)
template <int MK, int NT, int U>
__device__ __forceinline__ void head_bwd(TabPtr tl, Grad<MK, NT>& G, const HeadOut<MK, NT>& ho, float2 dyv, int q, f32x4 (&gh)[NT],
                                         float (&dfs)[2], float* tiles) {
    using T = QT<MK, NT>;
    G.dbout[0] += q == 0 ? dyv.x : 0.0f;
    G.dbout[1] += q == 0 ? dyv.y : 0.0f;
    if constexpr (Kind<MK>::DGRU) {
        const float4 wf = tab_ld(tl, T::WOF * 64);
        dfs[0] = (dyv.x * wf.x + dyv.y * wf.z) * ho.pcof[0];
        dfs[1] = (dyv.x * wf.y + dyv.y * wf.w) * ho.pcof[1];
#pragma unroll
        for (int c = 0; c < 2; ++c) { G.dwof[0][c] = __builtin_fmaf(dyv.x, ho.cofk[c], G.dwof[0][c]); G.dwof[1][c] = __builtin_fmaf(dyv.y, ho.cofk[c], G.dwof[1][c]); }
        f32x4 dpre[NT], back[NT];
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
            back[mt] = z4; dpre[mt] = z4;
            Q16_EACHU {
                G.dwout[0][mt][i] = __builtin_fmaf(dyv.x, ho.hok[mt][i], G.dwout[0][mt][i]);
                G.dwout[1][mt][i] = __builtin_fmaf(dyv.y, ho.hok[mt][i], G.dwout[1][mt][i]);
                const float dcat = (dyv.x * w0[i] + dyv.y * w1[i]) * ho.pho[mt][i];
                dpre[mt][i] = ho.hidpre[mt][i] > 0.0f ? dcat : 0.0f;
                G.dbhid[mt][i] += dpre[mt][i];
            }
        }
        s16n_matvec<NT, U>(tl, T::HIDT, dpre, back);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) Q16_EACHU gh[mt][i] += back[mt][i] * ho.ph2[mt][i];
        // dW_hid += dpre^T (x) h2 through the transpose tiles (slots 5 NT + 1 ..)
        float* tp = tiles + (5 * NT + 1) * kTileFloats;
        const int n = threadIdx.x & 15;
        wave_lds_fence();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) { tile_put(tp + kt * kTileFloats, n, q, dpre[kt]); tile_put(tp + (NT + kt) * kTileFloats, n, q, ho.h2k[kt]); }
        wave_lds_fence();
        float hT[NT][4];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) tile_get(tp + (NT + kt) * kTileFloats, n, q, hT[kt]);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            float dT[4];
            tile_get(tp + mt * kTileFloats, n, q, dT);
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) G.thid[mt][nt] = mfma4(dT[c], hT[nt][c], G.thid[mt][nt]);
        }
    } else {
        dfs[0] = dfs[1] = 0.0f;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
            Q16_EACHU {
                G.dwout[0][mt][i] = __builtin_fmaf(dyv.x, ho.hok[mt][i], G.dwout[0][mt][i]);
                G.dwout[1][mt][i] = __builtin_fmaf(dyv.y, ho.hok[mt][i], G.dwout[1][mt][i]);
                gh[mt][i] += (dyv.x * w0[i] + dyv.y * w1[i]) * ho.pho[mt][i];
            }
        }
    }
}

// weight-gradient MFMAs of one step: dW_x[g] += d_x[g]^T (x) fslot, dW_h[g] += d_h[g]^T (x) hq, through the transpose tiles
// (d_x = d_h for r, z; for the n gate d_x[2] = gate gradient, d_h[2] = gradient of the state-side term).
// `fcol` >= 0 (one unit tile, hidden + F + 1 <= 16): the feature slots ride in the padded unit columns fcol .. fcol + F of the hq tile,
// so ONE product per gradient row covers both weight blocks — r and z need no input-side MFMAs at all (their tih stays 0 and
// write-out reads those columns of thh), the n gate multiplies its two different gradients with the same tile: 16 MFMAs instead of 24.
template <int MK, int NT, bool MERGE>
__device__ __forceinline__ void wgrad_tiles(Grad<MK, NT>& G, float* tiles, int n, int q, const f32x4 (&dr)[NT], const f32x4 (&dz)[NT],
                                            const f32x4 (&dn)[NT], const f32x4 (&dnh)[NT], const f32x4 (&hq)[NT], float f0, float f1) {
    constexpr int fcol = 16 - (Kind<MK>::F + 1);
    auto tile = [tiles](int qty, int kt) { return tiles + (qty * NT + kt) * kTileFloats; };   // 0 dr 1 dz 2 dn 3 dnh 4 hq
    float* t_f = tiles + 5 * NT * kTileFloats;
    wave_lds_fence();
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        tile_put(tile(0, kt), n, q, dr[kt]); tile_put(tile(1, kt), n, q, dz[kt]); tile_put(tile(2, kt), n, q, dn[kt]);
        tile_put(tile(3, kt), n, q, dnh[kt]); tile_put(tile(4, kt), n, q, hq[kt]);
    }
    if constexpr (MERGE) {          // (after the float4 stores of the same wave: LDS operations of a wave complete in order)
        float* th = tile(4, 0);
        if (fcol + q < 16) th[n * kTilePitch + fcol + q] = f0;
        if (fcol + 4 + q < 16) th[n * kTilePitch + fcol + 4 + q] = f1;
    } else {
        t_f[n * kTilePitch + q] = f0;
        t_f[n * kTilePitch + 4 + q] = f1;
    }
    wave_lds_fence();
    float hT[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) tile_get(tile(4, kt), n, q, hT[kt]);
    if constexpr (MERGE) {
        float rT[4], zT[4], nT[4], gT[4];
        tile_get(tile(0, 0), n, q, rT); tile_get(tile(1, 0), n, q, zT);
        tile_get(tile(2, 0), n, q, nT); tile_get(tile(3, 0), n, q, gT);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            G.thh[0][0][0] = mfma4(rT[c], hT[0][c], G.thh[0][0][0]);
            G.thh[1][0][0] = mfma4(zT[c], hT[0][c], G.thh[1][0][0]);
            G.thh[2][0][0] = mfma4(gT[c], hT[0][c], G.thh[2][0][0]);
            G.tih[2][0] = mfma4(nT[c], hT[0][c], G.tih[2][0]);
        }
    } else {
    float fT[4];
    tile_get(t_f, n, q, fT);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        float rT[4], zT[4], nT[4], gT[4];
        tile_get(tile(0, mt), n, q, rT); tile_get(tile(1, mt), n, q, zT);
        tile_get(tile(2, mt), n, q, nT); tile_get(tile(3, mt), n, q, gT);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            G.tih[0][mt] = mfma4(rT[c], fT[c], G.tih[0][mt]);
            G.tih[1][mt] = mfma4(zT[c], fT[c], G.tih[1][mt]);
            G.tih[2][mt] = mfma4(nT[c], fT[c], G.tih[2][mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                G.thh[0][mt][nt] = mfma4(rT[c], hT[nt][c], G.thh[0][mt][nt]);
                G.thh[1][mt][nt] = mfma4(zT[c], hT[nt][c], G.thh[1][mt][nt]);
                G.thh[2][mt][nt] = mfma4(gT[c], hT[nt][c], G.thh[2][mt][nt]);
            }
        }
    }
    }
}
// the feature slots can share the hq tile when one unit tile leaves F + 1 padded columns: hidden + F + 1 <= 16 (a launch-time choice
// of the kernel instantiation)
template <int MK>
__host__ __device__ constexpr int merge_max_hidden() { return 16 - (Kind<MK>::F + 1); }

// dL/d(lane's feature slots) -> dL/dI, dL/dQ of the sample (summed over the sequence's four lanes); TRES: (nI, nQ) = the share of
// sample t + 1 (features I_next, Q_next)
template <int MK>
__device__ __forceinline__ void slots_bwd(float2 xv, const float (&oh)[4], const float (&dfs)[2], float& dI, float& dQ, float& nI, float& nQ) {
    nI = nQ = 0.0f;
    if constexpr (MK == K_GRU) {
        dI = oh[0] * dfs[0]; dQ = oh[1] * dfs[0];
    } else if constexpr (MK == K_Q4 || MK == K_A4) {
        const float df[4] = {oh[0] * dfs[0], oh[1] * dfs[0], oh[2] * dfs[0], oh[3] * dfs[0]};
        feat_bwd<MK == K_Q4 ? FEAT_Q4 : FEAT_A4>(xv.x, xv.y, df, dI, dQ);
    } else if constexpr (MK == K_DGRU) {
        const float df[6] = {oh[0] * dfs[0], oh[1] * dfs[0], oh[2] * dfs[0], oh[3] * dfs[0], oh[0] * dfs[1], oh[1] * dfs[1]};
        feat_bwd<FEAT_DGRU6>(xv.x, xv.y, df, dI, dQ);
    } else {
        const float df[4] = {oh[0] * dfs[0], oh[1] * dfs[0], oh[2] * dfs[0], oh[3] * dfs[0]};
        feat_bwd<FEAT_A4>(xv.x, xv.y, df, dI, dQ);
        nI = quad_sum(oh[0] * dfs[1]); nQ = quad_sum(oh[1] * dfs[1]);
    }
    dI = quad_sum(dI); dQ = quad_sum(dQ);
}

template <int MK, int NT, int U, bool LUT, bool FULL, bool DX, bool LOSS, bool MERGE>
__device__ __forceinline__ void q16_bwd_block(const SeqArgs& a, TabPtr tl0, const QSc& qs, const QK& k, const float4* lutq, const float* thr,
                                              int Ksig, const Scalars<MK>& sc, const float (&oh)[4], Grad<MK, NT>& G, const float2* xr,
                                              const float2* dys, float2* dxs, float* tiles, float2 x0, int n, int q, int tglob, int tloc,
                                              int nstep, int chunk_len, float* dxrow, StateD<NT> st, Carry<NT>& C, const S16Loss& lossc,
                                              float& loss_acc) {
    using T = QT<MK, NT>;
    using K = Kind<MK>;
    constexpr int S = T::S, NCH = K::NCH;
    const bool slot_ok[2] = {true, q < 2};
    f32x4 all_units[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) all_units[kt] = f32x4{1.f, 1.f, 1.f, 1.f};
    typedef typename std::conditional<K::TRES, SaveD<NT>, SaveS<NT>>::type Save;
    Save sv[S];
    float fq_s[S][2], px_s[S][2];
    TabPtr tl = opaque(tl0);
    {
        float zx = 0.f, zh = 0.f;
#pragma unroll
        for (int si = 0; si < S; ++si) {
            if (FULL || si < nstep) {
                const float2 xv = xr[tloc + si];
                float2 xn = make_float2(0.f, 0.f);
                if constexpr (K::TRES) xn = (tglob + si + 1 < a.T) ? xr[tloc + si + 1] : x0;
                float fs[NCH];
                q16_slots<MK>(xv, xn, oh, fs);
                if constexpr (K::TRES) {
                    delta_cell<NT, LUT, true>(tl, qs, k, lutq, thr, Ksig, fs, a.thx, a.thh, slot_ok, all_units, st, sv[si], zx, zh);
                } else {
                    float fqk[NCH];
                    fq_s[si][1] = 0.0f; px_s[si][1] = 0.0f;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float v = fs[c] * k.inv_xa, m = gm(v, k);
                        fqk[c] = rintf(m);
                        fq_s[si][c] = fqk[c]; px_s[si][c] = m == v ? k.s_xw : 0.0f;
                    }
                    std_cell<MK, NT, U, LUT, true>(tl, qs, k, lutq, fqk, st.h, sv[si]);
                }
            }
        }
    }
    tl = opaque(tl0);
    const int F = K::F;
    // the constant slot of the feature tile (column F of tih = the x-side bias gradient of the GRUCell; the tile is in units of s_xa)
    const float one_c0 = (!K::TRES && F < 4 && q == F) ? k.inv_xa : 0.0f, one_c1 = (!K::TRES && F >= 4 && 4 + q == F) ? k.inv_xa : 0.0f;
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            const int tt = tloc + si;
            float2 dyv = dys[n * kChunkPad + tt];            // LOSS: the staged TARGET, turned into dL/dy below
            const float2 xv = xr[tt];
            float2 xn = make_float2(0.f, 0.f);
            if constexpr (K::TRES) xn = (tglob + si + 1 < a.T) ? xr[tt + 1] : x0;
            float fs[NCH];
            q16_slots<MK>(xv, xn, oh, fs);
            // head: recomputed from the step's new state
            HeadOut<MK, NT> ho;
            {
                float y0, y1;
                head_fwd<MK, NT, U, true>(tl, k, sv[si].hnew, fs, ho, y0, y1);
                if constexpr (LOSS) {      // train-mode output of the step, the loss and dL/dy on the fly (no y / dy round trip through HBM)
                    y0 = __builtin_fmaf(y0, k.So, sc.bout[0]); y1 = __builtin_fmaf(y1, k.So, sc.bout[1]);
                    if constexpr (K::TRES) {
                        float s1[3], s2[2];
                        q16_tcn<MK>(sc, xr[tt - kHalo], xv, xr[tt + kHalo], s1, s2);
                        y0 += hardswishf_(s2[0]); y1 += hardswishf_(s2[1]);
                    }
                    float d0, d1;
                    s16_loss(lossc, y0 - dyv.x, y1 - dyv.y, d0, d1, loss_acc);
                    dyv = make_float2(d0, d1);
                }
            }
            if constexpr (K::TRES) {
                if (q == 0) {    // TCN skip gradients: per-sequence work, one lane of the four
                    float s1[3], s2[2];
                    const float2 xm = xr[tt - kHalo], xc = xr[tt], xq = xr[tt + kHalo];
                    q16_tcn<MK>(sc, xm, xc, xq, s1, s2);
                    const float d2[2] = {dyv.x * q16_hsg(s2[0]), dyv.y * q16_hsg(s2[1])};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float hs = hardswishf_(s1[c]);
                        G.dw2[c] = __builtin_fmaf(d2[0], hs, G.dw2[c]);
                        G.dw2[3 + c] = __builtin_fmaf(d2[1], hs, G.dw2[3 + c]);
                        const float d1 = __builtin_fmaf(d2[0], sc.w2[c], d2[1] * sc.w2[3 + c]) * q16_hsg(s1[c]);
                        G.dw1[c * 6 + 0] = __builtin_fmaf(d1, xm.x, G.dw1[c * 6 + 0]); G.dw1[c * 6 + 1] = __builtin_fmaf(d1, xc.x, G.dw1[c * 6 + 1]);
                        G.dw1[c * 6 + 2] = __builtin_fmaf(d1, xq.x, G.dw1[c * 6 + 2]); G.dw1[c * 6 + 3] = __builtin_fmaf(d1, xm.y, G.dw1[c * 6 + 3]);
                        G.dw1[c * 6 + 4] = __builtin_fmaf(d1, xc.y, G.dw1[c * 6 + 4]); G.dw1[c * 6 + 5] = __builtin_fmaf(d1, xq.y, G.dw1[c * 6 + 5]);
                    }
                }
            }
            float dfs_head[2];
            head_bwd<MK, NT, U>(tl, G, ho, dyv, q, C.gh, dfs_head, tiles);
            float dfs[2] = {0.0f, 0.0f};
            if constexpr (K::TRES) {
                const SaveD<NT>& v = sv[si];
                f32x4 ghprev[NT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
                    ODPD_EACH4 {
                        const float g2 = C.gh[mt][i] * v.c2[mt][i], g3 = C.gh[mt][i] * v.c3[mt][i];
                        const float dz = g2 * v.hp[mt][i] - g3 * v.npo[mt][i];
                        const float dan = g3 * v.omz[mt][i] * v.An[mt][i];
                        ghprev[mt][i] = g2 * v.z[mt][i];
                        C.gn[mt][i] += dan;
                        C.gnh[mt][i] += dan * v.B1[mt][i];
                        C.gr[mt][i] += dan * v.B2A[mt][i];
                        C.gz[mt][i] += dz * v.Az[mt][i];
                    }
                f32x4 ddh[NT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) ddh[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                s16n_matvec<NT>(tl, T::HHT + 0 * NT * NT, C.gr, ddh);
                s16n_matvec<NT>(tl, T::HHT + 1 * NT * NT, C.gz, ddh);
                s16n_matvec<NT>(tl, T::HHT + 2 * NT * NT, C.gnh, ddh);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt)
                    ODPD_EACH4 {
                        const float m = v.mh[mt][i], g2 = ddh[mt][i] * v.pph[mt][i];
                        C.gh[mt][i] = ghprev[mt][i] + m * (g2 + C.ghp[mt][i]);
                        C.ghp[mt][i] = (1.0f - m) * C.ghp[mt][i] - m * g2;
                    }
                if constexpr (DX) {
                    f32x4 ds = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        const f32x4 wr = as_f32x4(tab_ld(tl, (T::IHT + 0 * NT + kt) * 64)), wz = as_f32x4(tab_ld(tl, (T::IHT + 1 * NT + kt) * 64)),
                                    wn = as_f32x4(tab_ld(tl, (T::IHT + 2 * NT + kt) * 64));
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            ds = mfma4(wr[c], C.gr[kt][c], ds); ds = mfma4(wz[c], C.gz[kt][c], ds); ds = mfma4(wn[c], C.gn[kt][c], ds);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float m = v.mx[c], g = ds[c] * v.px[c];
                        dfs[c] = m * (g + C.gxp[c]);
                        C.gxp[c] = (1.0f - m) * C.gxp[c] - m * g;
                    }
                }
                wgrad_tiles<MK, NT, MERGE>(G, tiles, n, q, C.gr, C.gz, C.gn, C.gnh, v.qdhk, v.fqk[0], v.fqk[1]);
            } else {
                const SaveS<NT>& v = sv[si];
                f32x4 dar[NT], daz[NT], dan[NT], dhtn[NT], dhdir[NT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    if constexpr (U < 4) dar[mt] = daz[mt] = dan[mt] = dhtn[mt] = dhdir[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    Q16_EACHU {
                        const float g2 = C.gh[mt][i] * v.c2[mt][i], g3 = C.gh[mt][i] * v.c3[mt][i];
                        const float dz = g2 * v.hp[mt][i] - g3 * v.n[mt][i];
                        const float da = g3 * (1.0f - v.z[mt][i]) * v.An[mt][i];
                        dan[mt][i] = da;
                        dhtn[mt][i] = da * v.B1[mt][i];
                        dar[mt][i] = da * v.B2A[mt][i];
                        daz[mt][i] = dz * v.Az[mt][i];
                        dhdir[mt][i] = g2 * v.z[mt][i];
                        G.dbhn[mt][i] += dhtn[mt][i];
                    }
                }
                f32x4 ddh[NT];
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) ddh[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                s16n_matvec<NT, U>(tl, T::HHT + 0 * NT * NT, dar, ddh);
                s16n_matvec<NT, U>(tl, T::HHT + 1 * NT * NT, daz, ddh);
                s16n_matvec<NT, U>(tl, T::HHT + 2 * NT * NT, dhtn, ddh);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) Q16_EACHU C.gh[mt][i] = dhdir[mt][i] + ddh[mt][i] * v.pph[mt][i];
                if constexpr (DX) {
                    f32x4 ds = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        const f32x4 wr = as_f32x4(tab_ld(tl, (T::IHT + 0 * NT + kt) * 64)), wz = as_f32x4(tab_ld(tl, (T::IHT + 1 * NT + kt) * 64)),
                                    wn = as_f32x4(tab_ld(tl, (T::IHT + 2 * NT + kt) * 64));
#pragma unroll
                        for (int c = 0; c < U; ++c) {
                            ds = mfma4(wr[c], dar[kt][c], ds); ds = mfma4(wz[c], daz[kt][c], ds); ds = mfma4(wn[c], dan[kt][c], ds);
                        }
                    }
                    dfs[0] = ds[0] * px_s[si][0] + dfs_head[0];
                    dfs[1] = (NCH > 1 ? ds[1] * px_s[si][1] : 0.0f) + dfs_head[1];
                }
                wgrad_tiles<MK, NT, MERGE>(G, tiles, n, q, dar, daz, dan, dhtn, v.hqk, fq_s[si][0] + one_c0, fq_s[si][1] + one_c1);
            }
            if constexpr (DX) {
                float dI, dQ, nI, nQ;
                slots_bwd<MK>(xv, oh, dfs, dI, dQ, nI, nQ);
                if (q == 0) {
                    dxs[n * kChunkPad + tt] = make_float2(dI, dQ);
                    if constexpr (K::TRES) {
                        const int t1 = tglob + si + 1;
                        if (t1 >= a.T) { C.wrap[0] = nI; C.wrap[1] = nQ; }                 // torch.roll: the last step's "next" is sample 0
                        else if (tt + 1 < chunk_len) { dxs[n * kChunkPad + tt + 1].x += nI; dxs[n * kChunkPad + tt + 1].y += nQ; }
                        else if (dxrow != nullptr) {   // sample t + 1 lives in the chunk this wave flushed before: add at L2
                            __threadfence();
                            atomicAdd(dxrow + 2 * t1, nI);
                            atomicAdd(dxrow + 2 * t1 + 1, nQ);
                        }
                    }
                }
            }
        }
    }
}

template <int MK, int NT, int U, bool MERGE>
__device__ __forceinline__ void q16_write_row(float* prow, const float* pl, const QatLayout& L, const WQ& wq, const QK& k, Grad<MK, NT>& G,
                                              int lane, int n, int q) {
    using K = Kind<MK>;
    static_assert(U == 4 || !MERGE, "merged feature columns: four units per lane only");
    constexpr int F = K::F;
    const int H = L.H;
    constexpr int fcol = MERGE ? 16 - (F + 1) : -1;
    for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.f;       // incl. the scale parameters: exact zero gradient
    wave_lds_fence();
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int u = q16_unit<U>(mt, 4 * q + rr, H);
            if (u < H) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    // merged tiles (fcol >= 0): slot j of gate g sits in column fcol + j of thh[g] (r, z) / of tih[2] (n)
                    const int fs = fcol >= 0 ? n - fcol : n;
                    const float tx = ((fcol >= 0 && g < 2) ? G.thh[g][mt][0][rr] : G.tih[g][mt][rr]) * k.s_xa;
                    if (fs >= 0 && fs < F) { const int j = L.o_wx + (g * H + u) * F + fs; prow[j] = tx * qpass(pl[j], wq.x); }
                    if (!K::TRES && fs == F) { prow[L.o_bx + g * H + u] = tx; if (g < 2) prow[L.o_bh + g * H + u] = tx; }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (q16_unit<U>(nt, n, H) < H) { const int j = L.o_wh + (g * H + u) * H + q16_unit<U>(nt, n, H); prow[j] = G.thh[g][mt][nt][rr] * k.s_ha * qpass(pl[j], wq.h); }
                }
                if constexpr (K::DGRU) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (q16_unit<U>(nt, n, H) < H) { const int j = L.o_whid + u * H + q16_unit<U>(nt, n, H); prow[j] = G.thid[mt][nt][rr] * k.s_hida * qpass(pl[j], wq.hid); }
                }
            }
            const float w0 = row_sum16(G.dwout[0][mt][rr]) * k.s_oa, w1 = row_sum16(G.dwout[1][mt][rr]) * k.s_oa;
            const float bn = row_sum16(G.dbhn[mt][rr]), bhid = row_sum16(G.dbhid[mt][rr]);
            if (n == 0 && u < H) {
                prow[L.o_wo + u] = w0 * qpass(pl[L.o_wo + u], wq.o);
                prow[L.o_wo + L.OW + u] = w1 * qpass(pl[L.o_wo + L.OW + u], wq.o);
                if constexpr (!K::TRES) prow[L.o_bh + 2 * H + u] = bn;
                if constexpr (K::DGRU) prow[L.o_bhid + u] = bhid;
            }
        }
    if constexpr (K::DGRU) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float v = row_sum16(G.dwof[cc][c]) * k.s_oa;
                const int slot = 4 * c + q;
                if (n == 0 && slot < 6) { const int j = L.o_wo + cc * L.OW + H + slot; prow[j] = v * qpass(pl[j], wq.o); }
            }
    }
    if constexpr (K::TRES) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const float v = row_sum16(G.dw1[i]);
            if (n == 0 && q == 0) prow[L.o_tcn0 + i] = v;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float v = row_sum16(G.dw2[i]);
            if (n == 0 && q == 0) prow[L.o_tcn2 + i] = v;
        }
    } else {
        const float b0 = row_sum16(G.dbout[0]), b1 = row_sum16(G.dbout[1]);
        if (n == 0 && q == 0) { prow[L.o_bo] = b0; prow[L.o_bo + 1] = b1; }
    }
}

// two waves per SIMD (eight-wave workgroups, 256 registers) where the block state fits without spilling: the GRUCell kinds at one unit
// tile without dL/dx; the others keep the whole 512-register file (one wave per SIMD)
// (the dgru kind at two waves per SIMD spills 110 .. 170 B per lane and measured 1.46 -> 1.82 ms: it stays at one)
template <int MK, int NT, bool DX> struct BwdOcc { static constexpr bool W2 = NT == 1 && !Kind<MK>::TRES && !Kind<MK>::DGRU && !DX; };
// LOSS: `a.target` instead of `a.dy` — the step's output, the loss and dL/dy are formed inside (the fused train step's second launch)
template <int MK, int NT, bool LUT, bool DX, bool LOSS, bool MERGE, int U>
__device__ __forceinline__ void q16_bwd_body(const SeqArgs& a, int bits_w, int bits_a) {
    using T = QT<MK, NT>;
    using K = Kind<MK>;
    constexpr int S = T::S;
    constexpr int kWaveF = 2 * 16 * K::XSTRIDE + (DX ? 2 : 1) * 2 * 16 * kChunkPad + T::kTiles * kTileFloats;
    constexpr int kGroups = DX ? T::NG_DX : T::NG_BWD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const QatLayout L = qat_layout(MK, a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    const QSc qs = load_qsc<MK>(pl, L, bits_a);
    const WQ wq = make_wq(pl, L, bits_w);
    const QK k = make_qk(qs, wq);
    const int nlut = LUT ? (1 << bits_a) : 0;
    float* lut = tab + s16_tab_floats(kGroups);
    {
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int grp = wave; grp < kGroups; grp += nwb) t4[grp * 64 + lane] = q16_entry<MK, NT, U>(pl, L, wq, grp, n, q);
        if constexpr (LUT) fill_luts(lut, qs, k, bits_a, K::TRES);
        __syncthreads();
    }
    const int Ksig = sig_levels(qs.sig);
    const float4* lutq = reinterpret_cast<const float4*>(lut) - (int)qs.add.qn;
    const float* thr = lut + 4 * nlut;
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    Scalars<MK> sc;
    sc.load(pl, L);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = lut + (LUT ? 4 * nlut + kMaxThr : 0) + (size_t)wave * kWaveF;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * K::XSTRIDE;
    float2* dxs = dys + 16 * kChunkPad;                   // DX only
    float* tiles = reinterpret_cast<float*>(dys + (DX ? 2 : 1) * 16 * kChunkPad);
    for (int i = lane; i < kTileFloats; i += 64) tiles[5 * NT * kTileFloats + i] = 0.0f;       // feature tile: columns 8..15 stay 0
    const float2* xr = xs + n * K::XSTRIDE + K::HALO;
    Grad<MK, NT> G;
    G.zero();
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, valid ? a.inv_count : 0.0f, valid && q == 0);
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * T::kCk * 64 + lane;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[a.frame_idx ? (size_t)a.frame_idx[b0 + n] * a.frame_stride : (size_t)(b0 + n) * a.T] : make_float2(0.5f, 0.5f);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        Carry<NT> C;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) { C.gh[kt] = z4; C.ghp[kt] = z4; C.gr[kt] = z4; C.gz[kt] = z4; C.gn[kt] = z4; C.gnh[kt] = z4; }
        C.gxp[0] = C.gxp[1] = C.wrap[0] = C.wrap[1] = 0.0f;
        float* dxrow = (DX && valid) ? a.dx + (size_t)(b0 + n) * a.T * 2 : nullptr;
        int cur_chunk = -1, cur_len = 0;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        wave_lds_fence();
                        stage_out<16>(dxs, a.dx, b0, a.B, a.T, cur_chunk * kChunk, cur_len, lane);
                    }
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                cur_len = len;
                q16_stage_x<MK>(xs, a.x, b0, a.B, a.T, t0, lane, a.frame_idx, a.frame_stride);
                stage_in<16>(dys, LOSS ? a.target : a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f), a.frame_idx, a.frame_stride);
                wave_lds_fence();
                cur_chunk = chunk;
            }
            StateD<NT> st;
            init_state<NT>(st);
            if (blk) {
                const float4* c = ck + (size_t)blk * T::kCk * 64;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    st.h[kt] = as_f32x4(c[kt * 64]);
#pragma unroll
                    for (int i = U; i < 4; ++i) st.h[kt][i] = 0.0f;   // (stored as 0: told to the compiler)
                    if constexpr (K::TRES) {
                        st.hp[kt] = as_f32x4(c[(1 * NT + kt) * 64]);
                        st.dmr[kt] = as_f32x4(c[(2 * NT + kt) * 64]); st.dmz[kt] = as_f32x4(c[(3 * NT + kt) * 64]);
                        st.dmn[kt] = as_f32x4(c[(4 * NT + kt) * 64]); st.dmnh[kt] = as_f32x4(c[(5 * NT + kt) * 64]);
                    }
                }
                if constexpr (K::TRES) {
                    const float4 xp = c[6 * NT * 64];
                    st.xp[0] = xp.x; st.xp[1] = xp.y;
                }
            }
            if (nstep == S) q16_bwd_block<MK, NT, U, LUT, true, DX, LOSS, MERGE>(a, tl, qs, k, lutq, thr, Ksig, sc, oh, G, xr, dys, dxs, tiles, x0, n, q, tb, tb - t0, nstep, cur_len, dxrow, st, C, lossc, loss_acc);
            else q16_bwd_block<MK, NT, U, LUT, false, DX, LOSS, MERGE>(a, tl, qs, k, lutq, thr, Ksig, sc, oh, G, xr, dys, dxs, tiles, x0, n, q, tb, tb - t0, nstep, cur_len, dxrow, st, C, lossc, loss_acc);
        }
        if constexpr (DX) {
            wave_lds_fence();
            if constexpr (K::TRES) {
                if (q == 0) { dxs[n * kChunkPad].x += C.wrap[0]; dxs[n * kChunkPad].y += C.wrap[1]; }   // roll(x, -1): step T-1 saw sample 0
                wave_lds_fence();
            }
            stage_out<16>(dxs, a.dx, b0, a.B, a.T, 0, cur_len, lane);
            wave_lds_fence();
        }
    }
    if (a.partials == nullptr) return;
    // the partial rows are built in LDS over the tables: the staged parameters (smem[0 .. P)) stay for the pass masks
    const int P4 = L.P + kLossCols;
    __syncthreads();
    float* rows = smem + pad4(L.P);
    q16_write_row<MK, NT, U, MERGE>(rows + wave * P4, pl, L, wq, k, G, lane, n, q);
    if constexpr (LOSS) {      // column P of the row: the un-normalised loss sum of the wave's sequences
        float ls = loss_acc;
        for (int o = 32; o > 0; o >>= 1) ls += __shfl_down(ls, o);
        wave_lds_fence();
        if (lane == 0) rows[wave * P4 + L.P] = ls;
    }
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = rows[i];
        for (int wv = 1; wv < nwb; ++wv) v += rows[wv * P4 + i];
        prow[i] = v;
    }
}
template <int MK, int NT, bool LUT, bool DX, bool LOSS = false, bool MERGE = false>
__global__ __launch_bounds__((BwdOcc<MK, NT, DX>::W2 ? 512 : 256)) void qat16_bwd_kernel(SeqArgs a, int bits_w, int bits_a) {
    q16_bwd_body<MK, NT, LUT, DX, LOSS, MERGE, 4>(a, bits_w, bits_a);
}
// U = 3 / 2 units per lane: one unit tile, hidden <= 12 / <= 8 (see q16_unit)
template <int MK, bool LUT, bool DX, bool LOSS, int U>
__global__ __launch_bounds__((BwdOcc<MK, 1, DX>::W2 ? 512 : 256)) void qat16u_bwd_kernel(SeqArgs a, int bits_w, int bits_a) {
    q16_bwd_body<MK, 1, LUT, DX, LOSS, false, U>(a, bits_w, bits_a);
}

// dL/dx through the TRes skip path  skip = HS(conv2(HS(conv1(x)))), conv1: k3, dilation 16, zero padding (time-parallel)
__global__ __launch_bounds__(256) void qtres_skip_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, float* __restrict__ dx, int B, int T) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * T) return;
    const int b = (int)(idx / T), t = (int)(idx % T);
    const float2* x2 = reinterpret_cast<const float2*>(x) + (size_t)b * T;
    const float2* d2y = reinterpret_cast<const float2*>(dy) + (size_t)b * T;
    auto at = [&](int p) { return (p >= 0 && p < T) ? x2[p] : make_float2(0.0f, 0.0f); };
    float gI = 0.0f, gQ = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = t - 16 * (k - 1);
        if (p < 0 || p >= T) continue;
        const float2 xm = at(p - 16), xc = at(p), xq = at(p + 16), dyp = d2y[p];
        float s1[3], s2[2] = {0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s1[c] = w1[c * 6] * xm.x + w1[c * 6 + 1] * xc.x + w1[c * 6 + 2] * xq.x + w1[c * 6 + 3] * xm.y + w1[c * 6 + 4] * xc.y + w1[c * 6 + 5] * xq.y;
            const float hs = hardswishf_(s1[c]);
            s2[0] = __builtin_fmaf(w2[c], hs, s2[0]); s2[1] = __builtin_fmaf(w2[3 + c], hs, s2[1]);
        }
        const float e0 = dyp.x * q16_hsg(s2[0]), e1 = dyp.y * q16_hsg(s2[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d1 = __builtin_fmaf(e0, w2[c], e1 * w2[3 + c]) * q16_hsg(s1[c]);
            gI = __builtin_fmaf(w1[c * 6 + k], d1, gI);
            gQ = __builtin_fmaf(w1[c * 6 + 3 + k], d1, gQ);
        }
    }
    float2* o = reinterpret_cast<float2*>(dx) + (size_t)b * T + t;
    const float2 cur = *o;
    *o = make_float2(cur.x + gI, cur.y + gQ);
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
static int kind_of(const odpd_model_t* m) {
    switch (m->backbone) {
    case ODPD_GRU: return K_GRU;
    case ODPD_DGRU: return K_DGRU;
    case ODPD_QGRU: return K_Q4;
    case ODPD_QGRU_AMP1: return K_A4;
    case ODPD_TRES_DELTAGRU: return K_TRES;
    default: return -1;
    }
}
static int tiles_of(int H) { return (H + 15) / 16; }
static bool model_ok(const odpd_model_t* m) {
    return kind_of(m) >= 0 && m->hidden >= 1 && m->hidden <= 32 && m->bits_w >= 2 && m->bits_w <= 16 && m->bits_a >= 2 && m->bits_a <= 16;
}
static int stride_of(int nt) { return nt == 1 ? 2 : 1; }
template <int MK, int NT> static int groups(bool bwd, bool dx) { return !bwd ? QT<MK, NT>::NG_FWD : (dx ? QT<MK, NT>::NG_DX : QT<MK, NT>::NG_BWD); }
template <int MK, int NT>
static size_t lds_bytes(int P, int waves, int bits_a, bool lut, bool bwd, bool dx) {
    using K = Kind<MK>;
    const size_t per_wave = bwd ? 2 * 16 * K::XSTRIDE + (dx ? 2 : 1) * 2 * 16 * kChunkPad + QT<MK, NT>::kTiles * kTileFloats
                                : FwdGeo<MK>::kWaveF;
    size_t n = ((size_t)pad4(P) + s16_tab_floats(groups<MK, NT>(bwd, dx)) + (lut ? 4 * (1 << bits_a) + kMaxThr : 0) + (size_t)waves * per_wave) * sizeof(float);
    const size_t need = ((size_t)pad4(P) + (size_t)waves * (P + kLossCols)) * sizeof(float);
    if (bwd && n < need) n = need;
    return n;
}
// waves per workgroup: one per CU while the batch leaves CUs idle (latency regime), then up to `max_waves` sharing one operand table,
// as many as the LDS holds; the grid never exceeds the CU count (persistent waves)
template <int MK, int NT>
static LaunchShape shape(const odpd_model_t* m, int ngroups, bool bwd, bool dx, int max_waves) {
    const int P = qat_layout(MK, m->hidden).P, cus = device_cus();
    const bool lut = m->bits_a <= 8 && m->bits_w <= 8;
    LaunchShape ls;
    ls.waves = 1;
    while (ls.waves < max_waves && ngroups > ls.waves * cus) ls.waves *= 2;
    while (ls.waves > 1 && lds_bytes<MK, NT>(P, ls.waves, m->bits_a, lut, bwd, dx) > kMaxLds) --ls.waves;
    const int need = (ngroups + ls.waves - 1) / ls.waves;
    ls.grid = need < cus ? need : cus;
    if (ls.grid < 1) ls.grid = 1;
    return ls;
}
// the backward's grid (= rows of partials) must not depend on whether dL/dx is asked for: it is the grid of the launch WITHOUT dL/dx
template <int MK, int NT>
static int bwd_grid(const odpd_model_t* m, int ngroups) { return shape<MK, NT>(m, ngroups, true, false, BwdOcc<MK, NT, false>::W2 ? 8 : 4).grid; }
// unit slots per lane: 3 at hidden <= 12, 2 at hidden <= 8 (one unit tile, GRUCell kinds; the delta cell's only recipe is hidden 15), else 4
template <int MK> static int unit_slots(const odpd_model_t* m) {
    if (Kind<MK>::TRES || m->hidden > 12 || tuning().qat_u3 == 0) return 4;
    return m->hidden <= 8 ? 2 : 3;
}
typedef void (*FwdKernel)(SeqArgs, int, int, int);
typedef void (*BwdKernel)(SeqArgs, int, int);
template <int MK, int NT, bool LUT, int U> static FwdKernel fwd_kernel_of() {
    if constexpr (U < 4) return qat16u_fwd_kernel<MK, LUT, U>;
    else return qat16_fwd_kernel<MK, NT, LUT>;
}
template <int MK, int NT, bool LUT, bool DX, bool LOSS, int U> static BwdKernel bwd_kernel_of() {
    if constexpr (U < 4) return qat16u_bwd_kernel<MK, LUT, DX, LOSS, U>;
    else return qat16_bwd_kernel<MK, NT, LUT, DX, LOSS>;
}
template <int MK, int NT, bool LUT, int U = 4>
static int launch(hipStream_t st, const odpd_model_t* m, SeqArgs a, int mode) {
    using T = QT<MK, NT>;
    static_assert(U == 4 || (NT == 1 && !Kind<MK>::TRES), "fewer than four unit slots per lane: one tile of a GRUCell kind");
    const int P = qat_layout(MK, m->hidden).P;
    a.nck = (a.T + T::S - 1) / T::S;
    // (MERGE = true instantiations of the backward — input-side weight gradients inside the hq tile's padded columns, hidden <=
    // merge_max_hidden: 16 instead of 24 weight-gradient MFMAs per step — are not launched: at two waves per SIMD they spill ~200 B per
    // lane and measured 1.011 -> 0.995 ms only (profiles/r03/README.md); the code path stays for a register-leaner backward)
    if (mode == 1) {
        LaunchShape ls = shape<MK, NT>(m, a.ngroups, false, false, 8);
        const size_t lds = lds_bytes<MK, NT>(P, ls.waves, m->bits_a, LUT, false, false);
        if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
        if (ls.waves == 8 && 2 * lds <= kMaxLds) {      // two workgroups per CU (four waves per SIMD) where the LDS holds both
            const int need = (a.ngroups + 7) / 8, cap = 2 * device_cus();
            ls.grid = need < cap ? need : cap;
        }
        auto k = fwd_kernel_of<MK, NT, LUT, U>();
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a, m->bits_w, m->bits_a, (m->flags & ODPD_FLAG_EVAL) ? 1 : 0);
        return (int)hipGetLastError();
    }
    if (mode == 0) {       // fused train step: forward (checkpoints only) + backward with the loss formed inside — two launches, no y / dy
        if (!a.partials || !a.target || (!a.ckpt && a.nck > 1)) return ODPD_EINVAL;
        SeqArgs f = a;
        f.y = nullptr; f.stats = nullptr;
        if (int e = launch<MK, NT, LUT, U>(st, m, f, 1)) return e;
        LaunchShape ls = shape<MK, NT>(m, a.ngroups, true, false, BwdOcc<MK, NT, false>::W2 ? 8 : 4);
        ls.grid = bwd_grid<MK, NT>(m, a.ngroups);
        const size_t lds = lds_bytes<MK, NT>(P, ls.waves, m->bits_a, LUT, true, false);
        if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
        SeqArgs b = a;
        b.dx = nullptr;
        auto go = [&](auto k) {
            if (int e = allow_big_lds(k, lds)) return e;
            hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, b, m->bits_w, m->bits_a);
            return (int)hipGetLastError();
        };
        return go(bwd_kernel_of<MK, NT, LUT, false, true, U>());
    }
    if (a.partials == nullptr && a.dx == nullptr) return ODPD_EINVAL;
    if (!a.ckpt && a.nck > 1) return ODPD_EINVAL;
    const bool dx = a.dx != nullptr;
    LaunchShape ls = shape<MK, NT>(m, a.ngroups, true, dx, (dx ? BwdOcc<MK, NT, true>::W2 : BwdOcc<MK, NT, false>::W2) ? 8 : 4);
    ls.grid = bwd_grid<MK, NT>(m, a.ngroups);
    const size_t lds = lds_bytes<MK, NT>(P, ls.waves, m->bits_a, LUT, true, dx);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto go = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a, m->bits_w, m->bits_a);
        return (int)hipGetLastError();
    };
    if (!dx) return go(bwd_kernel_of<MK, NT, LUT, false, false, U>());
    if (int e = go(bwd_kernel_of<MK, NT, LUT, true, false, U>())) return e;
    if (MK == K_TRES) {
        const QatLayout L = qat_layout(MK, m->hidden);
        const long n = (long)a.B * a.T;
        hipLaunchKernelGGL(qtres_skip_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.x, a.dy, a.params + L.o_tcn0,
                           a.params + L.o_tcn2, a.dx, a.B, a.T);
    }
    return (int)hipGetLastError();
}
template <int MK>
static int launch_kind(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int mode) {
    const int nt = tiles_of(m->hidden);
    // the table / integer-pipe build needs 8-bit activations AND (for the int8 operands of one unit tile) 8-bit weights; any other
    // combination runs the table-free build (fp32 gates, fp32 MFMAs), which is exact for every grid width
    const bool lut = m->bits_a <= 8 && m->bits_w <= 8;
    if constexpr (!Kind<MK>::TRES) {
        const int u = nt == 1 ? unit_slots<MK>(m) : 4;
        if (u == 3) return lut ? launch<MK, 1, true, 3>(st, m, a, mode) : launch<MK, 1, false, 3>(st, m, a, mode);
        if (u == 2) return lut ? launch<MK, 1, true, 2>(st, m, a, mode) : launch<MK, 1, false, 2>(st, m, a, mode);
    }
    if (nt == 1) return lut ? launch<MK, 1, true>(st, m, a, mode) : launch<MK, 1, false>(st, m, a, mode);
    if (nt == 2) return lut ? launch<MK, 2, true>(st, m, a, mode) : launch<MK, 2, false>(st, m, a, mode);
    return ODPD_EUNSUPPORTED;
}
template <int MK>
static int rows_kind(const odpd_model_t* m, int ngroups) {
    return tiles_of(m->hidden) == 1 ? bwd_grid<MK, 1>(m, ngroups) : bwd_grid<MK, 2>(m, ngroups);
}
}  // namespace q16

bool qat_s16_supported(const odpd_model_t* m) { return q16::model_ok(m); }
int64_t qat_s16_param_count(const odpd_model_t* m) {
    return q16::model_ok(m) ? (int64_t)q16::qat_layout(q16::kind_of(m), m->hidden).P : (int64_t)ODPD_EUNSUPPORTED;
}
int64_t qat_s16_ckpt_floats(const odpd_model_t* m, int B, int T) {
    if (!q16::model_ok(m)) return ODPD_EUNSUPPORTED;
    const int nt = q16::tiles_of(m->hidden), S = q16::stride_of(nt);
    const int per = m->backbone == ODPD_TRES_DELTAGRU ? 6 * nt + 1 : nt;
    return (int64_t)((B + 15) / 16) * ((T + S - 1) / S) * per * 256;
}
int qat_s16_rows(const odpd_model_t* m, int B) {
    if (!q16::model_ok(m)) return ODPD_EUNSUPPORTED;
    const int ng = (B + 15) / 16;
    switch (q16::kind_of(m)) {
    case q16::K_GRU: return q16::rows_kind<q16::K_GRU>(m, ng);
    case q16::K_DGRU: return q16::rows_kind<q16::K_DGRU>(m, ng);
    case q16::K_Q4: return q16::rows_kind<q16::K_Q4>(m, ng);
    case q16::K_A4: return q16::rows_kind<q16::K_A4>(m, ng);
    default: return q16::rows_kind<q16::K_TRES>(m, ng);
    }
}
// mode 0 fused train step (a.ckpt = workspace of qat_s16_ckpt_floats, a.target, a.partials), 1 forward, 2 backward
int qat_s16_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    if (!q16::model_ok(m)) return ODPD_EUNSUPPORTED;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    switch (q16::kind_of(m)) {
    case q16::K_GRU: return q16::launch_kind<q16::K_GRU>(st, m, a, mode);
    case q16::K_DGRU: return q16::launch_kind<q16::K_DGRU>(st, m, a, mode);
    case q16::K_Q4: return q16::launch_kind<q16::K_Q4>(st, m, a, mode);
    case q16::K_A4: return q16::launch_kind<q16::K_A4>(st, m, a, mode);
    default: return q16::launch_kind<q16::K_TRES>(st, m, a, mode);
    }
}

}  // namespace odpd
