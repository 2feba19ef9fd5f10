// odpd_s16.h — building blocks shared by the 16-sequences-per-wave ("S16") kernels: gru_s16.hip (hidden <= 16, operands
// pinned in registers) and gru_s16n.hip (hidden <= 16 * NT, operands streamed from LDS).  Lane l = (n = l & 15 sequence,
// q = l >> 4 unit quad); see gru_s16.hip for the mapping and the MFMA operand layout.
#pragma once
#include "odpd_seq.h"

namespace odpd {

constexpr int kTilePitch = 20;                  // floats per sequence row of a transpose tile (16 + pad, 16 B aligned)
constexpr int kTileFloats = 16 * kTilePitch;
constexpr int kS16Tiles = 7;                    // drp dzp dnp dgh hp dhid feat
constexpr int kS16WaveFloats = 2 * 2 * 16 * kChunkPad + kS16Tiles * kTileFloats;

template <int FM> struct S16Cfg {
    static constexpr int F = FeatDim<FM>::F;
    static constexpr int NCH = (F + 4) / 4;     // K-chunks of the input projection (F features + constant-1 slot)
};

__host__ __device__ constexpr int s16_tab_floats(int groups) { return groups * 64 * 4; }

template <int FM, bool DG>
__device__ __forceinline__ float s16_wih_slot(const float* pl, const GruLayout& L, int g, int c, int m, int q) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH;
    const int H = L.H, k = 4 * c + q;
    if (c >= NCH || m >= H) return 0.0f;
    if (k < F) return pl[L.o_w_ih + (g * H + m) * F + k];
    if (k == F) return pl[L.o_b_ih + g * H + m] + (g < 2 ? pl[L.o_b_hh + g * H + m] : 0.0f);
    return 0.0f;
}
template <int FM, bool DG>
__device__ __forceinline__ float s16_woutf_slot(const float* pl, const GruLayout& L, int cc, int c, int q) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH;
    const int H = L.H, OW = DG ? H + 6 : H, k = 4 * c + q;
    if (c >= NCH) return 0.0f;
    if (DG && k < F) return pl[L.o_w_out + cc * OW + H + k];
    if (k == F) return pl[L.o_b_out + cc];
    return 0.0f;
}
// The r and z gate operands (W_h{r,z}, W_i{r,z}, their biases) are stored pre-multiplied by -log2(e): the MFMA
// accumulators then hold -log2(e) * pre-activation and sigmoid is exp2 / add / rcp with no scaling multiply.
constexpr float kNegLog2e = -1.4426950408889634f;
__device__ __forceinline__ f32x4 as_f32x4(const float4& v) { f32x4 r = {v.x, v.y, v.z, v.w}; return r; }

// feature slots of the lane: fs[c] = slot 4c+q of [feat_0 .. feat_{F-1}, 1, 0, ..], selected branch-free with the
// lane's one-hot quad indicator oh[e] = (q == e)  (exact: one non-zero term)
template <int FM>
__device__ __forceinline__ void s16_slots(float I, float Q, const float (&oh)[4], float (&fs)[S16Cfg<FM>::NCH]) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH;
    float f[F];
    feat_fwd<FM>(I, Q, f);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float acc = (F >= 4 * c && F < 4 * c + 4) ? oh[(F - 4 * c) & 3] : 0.0f;   // the constant-1 slot
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * c + e < F) acc = __builtin_fmaf(oh[e], f[(4 * c + e) < F ? (4 * c + e) : 0], acc);
        fs[c] = acc;
    }
}

// K-packing (hidden <= 13, two feature chunks): the chunk-1 slots [feat_4.., 1] ride in the three padded hidden K positions
// 13, 14, 15 of the recurrent mat-vec — B operand elements 1..3 of the quad-3 lanes, whose own h is identically 0 — so the cell
// needs one input-projection MFMA chunk instead of two.  pk[j] = value of slot 4 + j on the quad-3 lanes, 0 elsewhere.
template <int FM>
__device__ __forceinline__ void s16_slots_pk(float I, float Q, const float (&oh)[4], float (&fs)[S16Cfg<FM>::NCH], float (&pk)[3]) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH;
    float f[F];
    feat_fwd<FM>(I, Q, f);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float acc = (F >= 4 * c && F < 4 * c + 4) ? oh[(F - 4 * c) & 3] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * c + e < F) acc = __builtin_fmaf(oh[e], f[(4 * c + e) < F ? (4 * c + e) : 0], acc);
        fs[c] = acc;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) pk[j] = (4 + j < F) ? oh[3] * f[(4 + j) < F ? (4 + j) : 0] : ((4 + j == F) ? oh[3] : 0.0f);
}

// ---- stage-major 4-wide element-wise helpers (same arithmetic as sigmoidf_ / tanhf_ of odpd_device.h) ----
#define ODPD_EACH4 _Pragma("unroll") for (int i = 0; i < 4; ++i)
// relu of an MFMA result in ONE VALU op: v_med3_f32(v, 0, +inf) (fmaxf() would first quiet a possible sNaN with a second v_max).
// NOT inline asm (r01..r03: `asm("v_max_f32 %0, 0, %1")`): the operand is an MFMA result, and the wait states a VALU read of an
// in-flight XDL result needs are inserted by the compiler's hazard recognizer only for instructions it can see — an asm statement is
// opaque to it, so whenever the post-RA scheduler moved the asm directly behind the last v_mfma of the fc_hid chain it read the
// accumulator before the write-back.  That was the r02 "wrong-result build" (spilling eight-wave gru16n_kernel<DGRU, frozen-PA,
// four K-chunks>: DGRU only — the only kernels with a relu —, schedule dependent, gone with -enable-post-misched=false; r04:
// tools/exp_s16n8w.py).  #ifdef ODPD_RELU_ASM keeps the old form for that reproduction.
__device__ __forceinline__ float relu_(float v) {
#ifdef ODPD_RELU_ASM
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
#else
    return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff());
#endif
}
// per-element loops, NOT whole-vector arithmetic: <4 x float> IR ops become v_pk_*_f32, which issue at half
// rate on gfx950 and measured 6 % slower here (profiles/r01/ubench_issue_costs.md)
__device__ __forceinline__ f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }
__device__ __forceinline__ f32x4 fma4(const f32x4& a, const f32x4& b, const f32x4& c) {
    f32x4 r;
    ODPD_EACH4 r[i] = __builtin_fmaf(a[i], b[i], c[i]);
    return r;
}
__device__ __forceinline__ f32x4 mul4(const f32x4& a, const f32x4& b) {
    f32x4 r;
    ODPD_EACH4 r[i] = a[i] * b[i];
    return r;
}
__device__ __forceinline__ f32x4 add4(const f32x4& a, const f32x4& b) {
    f32x4 r;
    ODPD_EACH4 r[i] = a[i] + b[i];
    return r;
}
__device__ __forceinline__ f32x4 sub4(const f32x4& a, const f32x4& b) {
    f32x4 r;
    ODPD_EACH4 r[i] = a[i] - b[i];
    return r;
}
#ifdef ODPD_X_NOTRANS      // removal experiment (timing only): transcendentals as one FMA each
#define ODPD_XEXP2(v) __builtin_fmaf((v), 0.25f, 1.0f)
#define ODPD_XRCP(v) __builtin_fmaf((v), -0.25f, 1.0f)
#else
#define ODPD_XEXP2(v) __builtin_amdgcn_exp2f(v)
#define ODPD_XRCP(v) fast_rcp(v)
#endif
__device__ __forceinline__ f32x4 exp2_4(const f32x4& v) {
    f32x4 r;
    ODPD_EACH4 r[i] = ODPD_XEXP2(v[i]);
    return r;
}
__device__ __forceinline__ f32x4 rcp4(const f32x4& v) {
    f32x4 r;
    ODPD_EACH4 r[i] = ODPD_XRCP(v[i]);
    return r;
}
// sigmoid of a pre-activation that arrives already multiplied by -log2(e)
__device__ __forceinline__ f32x4 sigmoid4_prescaled(const f32x4& v) {
    return rcp4(add4(exp2_4(v), splat4(1.0f)));
}
// tanh(v) = 1 - 2 / (exp2(2 log2(e) v) + 1): 5 VALU ops, |error| <= 1.9e-7 over the whole range (the row-rotated
// kernels' tanhf_ adds a polynomial branch for |v| < 0.3 to reach 1.1e-7; both are far inside the 2e-5 parity
// tolerance and of the size of one rounding of a value near 1)
__device__ __forceinline__ f32x4 tanh4(const f32x4& v) {
    const f32x4 e = add4(exp2_4(mul4(v, splat4(2.8853900817779268f))), splat4(1.0f));
    return fma4(rcp4(e), splat4(-2.0f), splat4(1.0f));
}

// tanh with RELATIVE accuracy near 0 (error < 1.2e-7 of the result): the single formula above has an ABSOLUTE error of
// 1.9e-7, which is 1e-4 of tanh(1e-3) — visible (2e-5 in y) in the backbones whose features all scale with the signal
// amplitude (gru, qgru*, lstm, vdlstm) when the whole signal is small; dgru's sin/cos features are O(1) and keep tanh4.
// Below 0.3 the odd polynomial of tanhf_ takes over through an arithmetic blend (weight 0 or 1, no compare / select).
__device__ __forceinline__ f32x4 tanh4_rel(const f32x4& v) {
    f32x4 r;
    ODPD_EACH4 {
        const float x = v[i], x2 = x * x;
        float p = __builtin_fmaf(x2, 0.021869488536155203f, -0.053968253968253971f);
        p = __builtin_fmaf(x2, p, 0.13333333333333333f);
        p = __builtin_fmaf(x2, p, -0.33333333333333333f);
        p = __builtin_fmaf(x2 * x, p, x);
        const float e = ODPD_XEXP2(x * 2.8853900817779268f) + 1.0f;
        const float t = __builtin_fmaf(ODPD_XRCP(e), -2.0f, 1.0f);
        const float w = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fabsf(x), -0x1p100f, 0.3f * 0x1p100f), 0.0f, 1.0f);   // |x| < 0.3
        r[i] = __builtin_fmaf(w, p - t, t);
    }
    return r;
}
// tanh of the amplitude-scaled backbones vs the O(1)-feature ones (FM = feature map of the GRU family)
template <int FM>
__device__ __forceinline__ f32x4 tanh4_for(const f32x4& v) {
    if constexpr (FM == FEAT_DGRU6) return tanh4(v);
    else return tanh4_rel(v);
}

// tanh with the row-rotated kernels' accuracy (polynomial below 0.3): used where a rounding-level difference could
// flip a threshold decision (delta backbones)
__device__ __forceinline__ f32x4 tanh4_precise(const f32x4& v) {
    f32x4 r;
    ODPD_EACH4 r[i] = tanhf_(v[i]);
    return r;
}
// ---- loss and dL/dy of one I/Q sample without compares or selects --------------------------------------------------
// (a VOP2 v_cndmask through vcc measured ~21 cycles next to the MFMA stream; these are plain multiplies / FMAs / one bfi)
// L2: dy = 2 sc d, loss += d0^2 + d1^2;  L1: dy = sc sign(d) (0 at d = 0), loss += |d0| + |d1|.  `counts`: this lane
// adds the sample to the loss sum (one quad per sequence).  For L2 the results are bit-identical to the select form.
struct S16Loss { float c2, c1, w2, w1; };
__device__ __forceinline__ S16Loss s16_loss_setup(bool l2, float sc, bool counts) {
    S16Loss L;
    L.c2 = l2 ? 2.0f * sc : 0.0f; L.c1 = l2 ? 0.0f : sc;
    L.w2 = (l2 && counts) ? 1.0f : 0.0f; L.w1 = (!l2 && counts) ? 1.0f : 0.0f;
    return L;
}
__device__ __forceinline__ float s16_sign(float d) {      // -1, 0, +1 (|d| < 2^-100 would give a fraction: never a loss residual)
    return __builtin_copysignf(__builtin_amdgcn_fmed3f(__builtin_fabsf(d) * 0x1p100f, 0.0f, 1.0f), d);
}
__device__ __forceinline__ void s16_loss(const S16Loss& L, float d0, float d1, float& dy0, float& dy1, float& acc) {
    dy0 = __builtin_fmaf(L.c1, s16_sign(d0), L.c2 * d0);
    dy1 = __builtin_fmaf(L.c1, s16_sign(d1), L.c2 * d1);
    acc = __builtin_fmaf(L.w2, __builtin_fmaf(d0, d0, d1 * d1), acc);
    acc = __builtin_fmaf(L.w1, __builtin_fabsf(d0) + __builtin_fabsf(d1), acc);
}
// relu'(v) as a multiplier: 0 for v <= 0, 1 for v >= 2^-100 (v_mul with the clamp modifier instead of v_cmp + v_cndmask)
__device__ __forceinline__ float relu_gate(float v) { return __builtin_amdgcn_fmed3f(v * 0x1p100f, 0.0f, 1.0f); }

// sum over the four quads of a sequence (lanes n, n+16, n+32, n+48); every lane gets the total.
// Two VALU cross-row swaps (gfx950 v_permlane16_swap / v_permlane32_swap: both results of a swap of v with itself are the two partner
// rows' values on EVERY lane, so no select is needed) instead of ds_swizzle + ds_bpermute: no LDS-pipe round trips inside the step loops.
// Same additions in the same order as the swizzle form ((own + row partner) + (the other half's pair)): bit-identical sums.
__device__ __forceinline__ float quad_sum(float v) {
#ifdef ODPD_QUAD_SUM_LDS
    v += swap16(v);
    v += __shfl_xor(v, 32);
    return v;
#else
    const int iv = __builtin_bit_cast(int, v);
    const auto a = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);      // a[0] = rows (0, 0, 2, 2), a[1] = rows (1, 1, 3, 3)
    const int a0 = a[0], a1 = a[1];
    const float s = __builtin_bit_cast(float, a0) + __builtin_bit_cast(float, a1);
    const int is = __builtin_bit_cast(int, s);
    const auto b = __builtin_amdgcn_permlane32_swap(is, is, false, false);      // b[0] = halves (lo, lo), b[1] = (hi, hi)
    const int b0 = b[0], b1 = b[1];
    return __builtin_bit_cast(float, b0) + __builtin_bit_cast(float, b1);
#endif
}

// transpose tile: lane (n,q) stores X[n][4q..4q+3]; lane (u,k) loads X[4k+c][u], c = 0..3
__device__ __forceinline__ void tile_put(float* tile, int n, int q, const f32x4& v) {
    *reinterpret_cast<float4*>(tile + n * kTilePitch + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void tile_get(const float* tile, int u, int k, float (&out)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = tile[(4 * k + c) * kTilePitch + u];
}


// ---- operands streamed from an LDS table ([group][lane] float4), NT x NT tiles of 16 units ----------------------
// acc[mt] += sum_{kt, c} T[base + mt*NT + kt][c] (x) v[kt][c]    — one streamed ds_read_b128 per (mt, kt).
// NCK: K-chunks of the LAST unit tile that hold real units (gru_s16n.hip packs that tile element-major); the rest multiply zeros
template <int NT, int NCK = 4>
__device__ __forceinline__ void s16n_matvec(TabPtr tl, int base, const f32x4 (&v)[NT], f32x4 (&acc)[NT]) {
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            const float4 w = tab_ld(tl, (base + mt * NT + kt) * 64);
            constexpr int kLast = NT - 1;
            acc[mt] = mfma4(w.x, v[kt][0], acc[mt]);
            if (kt < kLast || NCK > 1) acc[mt] = mfma4(w.y, v[kt][1], acc[mt]);
            if (kt < kLast || NCK > 2) acc[mt] = mfma4(w.z, v[kt][2], acc[mt]);
            if (kt < kLast || NCK > 3) acc[mt] = mfma4(w.w, v[kt][3], acc[mt]);
        }
}


}  // namespace odpd
