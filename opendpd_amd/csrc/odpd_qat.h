// odpd_qat.h — pieces shared by the quantisation-aware translation units (qat_s16.hip, gru_cascade.hip): parameter layout, quantisers in
// grid units, gate functions.  Everything behind this header is compiled with FP contraction off: a fused multiply-add would change
// roundings the reference does not have (an including file that wants contraction back says `#pragma clang fp contract(fast)` after it).
#pragma once
#include "odpd_s16.h"
#include "odpd_quant.h"

#pragma clang fp contract(off)

namespace odpd {
namespace q16 {

enum { K_GRU = 0, K_DGRU = 1, K_Q4 = 2, K_A4 = 3, K_TRES = 4 };
constexpr int kHalo = 16;                                   // TCN taps at t-16, t, t+16

template <int MK> struct Kind {
    static constexpr bool TRES = MK == K_TRES, DGRU = MK == K_DGRU;
    static constexpr int F = MK == K_GRU ? 2 : ((MK == K_Q4 || MK == K_A4) ? 4 : 6);
    static constexpr int NCH = (F + 3) / 4;                 // feature-slot chunks (slot 4c+q on quad q)
    static constexpr int HALO = TRES ? kHalo : 0;
    static constexpr int XSTRIDE = kChunk + 2 * HALO + 1;   // float2 per sequence row of the staged x
};

// parameter layout = named_parameters() of the quantised model (see oracle/odpd_oracle.c::qgru_layout)
struct QatLayout {
    int kind, H, F, OW;
    int o_wx, o_bx, o_sxw, o_sxa, o_sxo, o_wh, o_bh, o_shw, o_sha, o_sho, o_ssig, o_stanh, o_sadd, o_smul, o_wo, o_bo, o_sow, o_soa,
        o_soo, o_whid, o_bhid, o_shidw, o_shida, o_shido, o_tcn0, o_tcn2, P;
};
__host__ __device__ inline QatLayout qat_layout(int kind, int H) {
    QatLayout L;
    L.kind = kind; L.H = H;
    L.F = kind == K_GRU ? 2 : ((kind == K_Q4 || kind == K_A4) ? 4 : 6);
    L.OW = kind == K_DGRU ? H + 6 : H;
    const int F = L.F;
    int o = 0;
    L.o_bx = L.o_bh = L.o_bo = L.o_whid = L.o_bhid = L.o_shidw = L.o_shida = L.o_shido = L.o_tcn0 = L.o_tcn2 = 0;
    if (kind == K_TRES) {
        L.o_wx = o; o += 3 * H * F; L.o_sxw = o++; L.o_sxa = o++; L.o_sxo = o++;
        L.o_wh = o; o += 3 * H * H; L.o_shw = o++; L.o_sha = o++; L.o_sho = o++;
        L.o_sadd = o++; L.o_smul = o++; L.o_ssig = o++; L.o_stanh = o++;
        L.o_wo = o; o += 2 * H; L.o_sow = o++; L.o_soa = o++; L.o_soo = o++;
        L.o_tcn0 = o; o += 18; L.o_tcn2 = o; o += 6;
    } else {
        L.o_wx = o; o += 3 * H * F; L.o_bx = o; o += 3 * H; L.o_sxw = o++; L.o_sxa = o++; L.o_sxo = o++;
        L.o_wh = o; o += 3 * H * H; L.o_bh = o; o += 3 * H; L.o_shw = o++; L.o_sha = o++; L.o_sho = o++;
        L.o_ssig = o++; L.o_stanh = o++; L.o_sadd = o++; L.o_smul = o++;
        L.o_wo = o; o += 2 * L.OW; L.o_bo = o; o += 2; L.o_sow = o++; L.o_soa = o++; L.o_soo = o++;
        if (kind == K_DGRU) { L.o_whid = o; o += H * H; L.o_bhid = o; o += H; L.o_shidw = o++; L.o_shida = o++; L.o_shido = o++; }
    }
    L.P = o;
    return L;
}

// activation-side quantisers (wave-uniform)
struct QSc { Quant xa, ha, oa, sig, tnh, add, mul, out, hida; };
// GRID UNITS.  Inside the kernels a quantised value travels as the integer k of q(x) = k s (an integer-valued float): the trailing
// "* s" of every quantiser and the leading "/ s" of the next one fold into power-of-two constants (exact), the LUTs return gate values
// already quantised and pre-scaled, and the mat-vecs run on integers (weights k_w, activations k_a), their scale s_a s_w being applied
// by the one FMA that adds the fp32 bias:  fl(k S + b) == fl(fl(k S) + b)  because k S is exact.  Every rounding the reference makes
// is made here too, on the same real number.
struct QK {
    float qn, qp;                               // activation clamp (bits_a)
    float inv_xa, inv_ha, inv_oa, inv_hida, inv_add;
    float Sx, Sh, So, Shid;                     // s_act * s_weight of x2h, h2h, fc_out, fc_hid
    float s_add, s_mul, c_ma;                   // c_ma = s_mul / s_add
    float c_sm, c_tm;                           // s_sig / s_mul, s_tanh / s_mul
    float s_xa, s_ha, s_oa, s_hida;             // weight-gradient scales
    float s_xw, s_hw, s_ow, s_hidw;             // data-gradient scales (ride on the pass masks)
    float inv_sig, inv_tnh;
};
__device__ __forceinline__ float gk(float v, const QK& k) { return rintf(__builtin_amdgcn_fmed3f(v, k.qn, k.qp)); }      // v = x / s
__device__ __forceinline__ float gm(float v, const QK& k) { return __builtin_amdgcn_fmed3f(v, k.qn, k.qp); }

struct WQ { Quant x, h, o, hid; };       // weight quantisers
__device__ __forceinline__ WQ make_wq(const float* pl, const QatLayout& L, int bits_w) {
    WQ w;
    w.x = make_quant(pl[L.o_sxw], bits_w); w.h = make_quant(pl[L.o_shw], bits_w); w.o = make_quant(pl[L.o_sow], bits_w);
    w.hid = L.kind == K_DGRU ? make_quant(pl[L.o_shidw], bits_w) : w.o;
    return w;
}

__device__ __forceinline__ float kq(float w, const Quant& q) { return rintf(__builtin_amdgcn_fmed3f(w * q.inv, q.qn, q.qp)); }   // weight in grid units

// ---- gate functions ----------------------------------------------------------------------------------------------------
// LDS (LUT builds, <= 8 activation bits): one float4 per add-quantiser grid point x = k s_add (index = k - Qn):
//   .x = q_sig(sigmoid(x)) / s_mul      .y = pass_sig  sigmoid'(x)      (pass = the value lies inside the quantiser's clamp range)
//   .z = q_tanh(tanh(x)) / s_mul        .w = pass_tanh tanh'(x)
// evaluated in double at kernel start (sigmoid / tanh rounded to fp32 like the reference's activations, then quantised).  Behind
// it (delta cell) thr[0 .. K+1]: thr[0] = -inf, thr[k] = smallest float x with rint(sigmoid(x) / s_sig) >= k, thr[K+1] = +inf.
constexpr int kMaxThr = 132;
struct Gate { float c, d; };            // quantised value / s_mul, masked derivative
__device__ __forceinline__ int sig_levels(const Quant& qsig) {      // K: quantised sigmoid values above 0 that can occur
    if (qsig.s > 1.0f) return 0;
    const float n = qsig.inv;                                      // 1 / s, an integer
    return (int)(n < qsig.qp ? n : qsig.qp);
}
// Gates of the wider grids (no table): fp32 evaluations good to ~2 ulp — the argument product x log2(e) in two pieces (its rounding
// error would otherwise be amplified by |x|), v_exp_f32, one Newton step on v_rcp_f32.  On a 16-bit grid that leaves the rounded
// result open only within ~4e-3 LSB of a rounding boundary — the same order as the reference's own fp32 sigmoid / tanh (torch's
// vectorised kernels are 1-2 ulp off the real value too), so an evaluation in double (r01 - r03 first build: 2.4 x the step time)
// bought no parity: agreement for W16A16 is to one LSB either way.
__device__ __forceinline__ float sigmoid_acc(float x) {
    const float c_hi = -1.4426950216293335f, c_lo = -1.925963033500259e-8f;
    const float t_hi = x * c_hi, t_lo = __builtin_fmaf(x, c_hi, -t_hi) + x * c_lo;
    float e = __builtin_amdgcn_exp2f(t_hi);
    e = __builtin_fmaf(e, t_lo * 0.6931471805599453f, e);
    const float d = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return e < 3.0e38f ? r : 0.0f;            // (d = inf: the Newton step would form inf * 0)
}
__device__ __forceinline__ float tanh_acc(float x) { return __builtin_fmaf(2.0f, sigmoid_acc(2.0f * x), -1.0f); }
__device__ __forceinline__ Gate sig_gate(float rf, const QSc& qs, const QK& k) {
    const float v = rf * qs.sig.inv, m = __builtin_amdgcn_fmed3f(v, qs.sig.qn, qs.sig.qp);
    Gate g;
    g.c = rintf(m) * k.c_sm;
    g.d = m == v ? rf * (1.0f - rf) : 0.0f;
    return g;
}
__device__ __forceinline__ Gate tanh_gate(float nf, const QSc& qs, const QK& k) {
    const float v = nf * qs.tnh.inv, m = __builtin_amdgcn_fmed3f(v, qs.tnh.qn, qs.tnh.qp);
    Gate g;
    g.c = rintf(m) * k.c_tm;
    g.d = m == v ? 1.0f - nf * nf : 0.0f;
    return g;
}
__device__ __forceinline__ void fill_luts(float* lut, const QSc& qs, const QK& k, int bits, bool with_thr) {
    const int n = 1 << bits;
    float4* l4 = reinterpret_cast<float4*>(lut);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)((float)(i + (int)qs.add.qn) * qs.add.s);
        const Gate gs = sig_gate((float)(1.0 / (1.0 + exp(-x))), qs, k), gt = tanh_gate((float)tanh(x), qs, k);
        l4[i] = make_float4(gs.c, gs.d, gt.c, gt.d);
    }
    if (with_thr) {
        float* thr = lut + 4 * n;
        const int K = sig_levels(qs.sig);
        for (int j = threadIdx.x; j <= K + 1; j += blockDim.x) {
            float t;
            if (j == 0) t = -__builtin_inff();
            else if (j == K + 1) t = __builtin_inff();
            else {
                const double p = ((double)j - 0.5) * (double)qs.sig.s;
                if (p >= 1.0) t = __builtin_inff();
                else {
                    const double b = log(p / (1.0 - p));
                    t = (float)b;
                    if ((double)t < b) {                                                     // smallest float >= b
                        const int bits32 = __builtin_bit_cast(int, t);
                        t = __builtin_bit_cast(float, t > 0.0f ? bits32 + 1 : (t < 0.0f ? bits32 - 1 : 1));
                    }
                }
            }
            thr[j] = t;
        }
    }
}
// MAGIC-NUMBER ROUNDING (r06).  (v + 1.5 x 2^23) rounds v to the nearest-even integer k and leaves it in the low mantissa bits:
// bits = 0x4B400000 + k for |k| < 2^22.  One v_add_f32 (the fast issue class, profiles/r06/ubench_issue_costs.txt) instead of v_rndne_f32 +
// v_cvt_i32_f32 (two of the four-cycle class), and the table address follows with one v_lshl_add_u32:
//   &lutq[k] = base + 16 k = (bits << 4) + (base - (0x4B400000 << 4))   (mod 2^32; LDS addresses are 32 bits)
constexpr float kRneMagic = 12582912.0f;
typedef float lut_f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) lut_f2* LutLds;
__device__ __forceinline__ unsigned lut_bias(const float4* lutq) {
    return (unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) float4*)lutq + 0x4C000000u;      // 0x4C000000 = -(0x4B400000 << 4) mod 2^32
}
// gates of a CLAMPED, not yet rounded value m on the add grid (the rounding happens in the index)
template <bool LUT>
__device__ __forceinline__ Gate sig_grid_m(float m, const QSc& qs, const QK& k, const float4* lutq, unsigned bias) {
    if constexpr (LUT) {
        const lut_f2 e = *(LutLds)(__SIZE_TYPE__)((__builtin_bit_cast(unsigned, m + kRneMagic) << 4) + bias);
        Gate g; g.c = e[0]; g.d = e[1]; return g;
    } else return sig_gate(sigmoid_acc(rintf(m) * k.s_add), qs, k);
}
template <bool LUT>
__device__ __forceinline__ Gate tanh_grid_m(float m, const QSc& qs, const QK& k, const float4* lutq, unsigned bias) {
    if constexpr (LUT) {
        const lut_f2 e = *((LutLds)(__SIZE_TYPE__)((__builtin_bit_cast(unsigned, m + kRneMagic) << 4) + bias) + 1);
        Gate g; g.c = e[0]; g.d = e[1]; return g;
    } else return tanh_gate(tanh_acc(rintf(m) * k.s_add), qs, k);
}
// gates of a value on the add grid, given as its integer ak (lutq = table base shifted by -Qn entries)
template <bool LUT>
__device__ __forceinline__ Gate sig_grid(float ak, const QSc& qs, const QK& k, const float4* lutq) {
    if constexpr (LUT) { const float2 e = *reinterpret_cast<const float2*>(&lutq[(int)ak]); Gate g; g.c = e.x; g.d = e.y; return g; }
    else return sig_gate(sigmoid_acc(ak * k.s_add), qs, k);
}
template <bool LUT>
__device__ __forceinline__ Gate tanh_grid(float ak, const QSc& qs, const QK& k, const float4* lutq) {
    if constexpr (LUT) { const float2 e = *(reinterpret_cast<const float2*>(&lutq[(int)ak]) + 1); Gate g; g.c = e.x; g.d = e.y; return g; }
    else return tanh_gate(tanh_acc(ak * k.s_add), qs, k);
}
// quantised sigmoid of an arbitrary float (the delta cell's accumulators): exact through the boundary table around an fp32 guess
template <bool LUT>
__device__ __forceinline__ Gate sig_any(float x, const QSc& qs, const QK& k, const float* thr, int K) {
    if constexpr (LUT) {
        const float rf = sigmoidf_(x), v = rf * qs.sig.inv;
        int j = (int)rintf(__builtin_amdgcn_fmed3f(v, 0.0f, (float)K));
        j += (x >= thr[j + 1]) ? 1 : 0;
        j -= (x < thr[j]) ? 1 : 0;
        Gate g;
        g.c = (float)j * k.c_sm;
        g.d = v <= qs.sig.qp ? rf * (1.0f - rf) : 0.0f;
        return g;
    } else {
        return sig_gate(sigmoid_acc(x), qs, k);
    }
}

template <int MK>
__device__ __forceinline__ QSc load_qsc(const float* pl, const QatLayout& L, int bits_a) {
    QSc q;
    q.xa = make_quant(pl[L.o_sxa], bits_a); q.ha = make_quant(pl[L.o_sha], bits_a); q.oa = make_quant(pl[L.o_soa], bits_a);
    q.sig = make_quant(pl[L.o_ssig], bits_a); q.tnh = make_quant(pl[L.o_stanh], bits_a);
    q.add = make_quant(pl[L.o_sadd], bits_a); q.mul = make_quant(pl[L.o_smul], bits_a);
    q.out = make_quant(pl[L.o_soo], 16);
    q.hida = Kind<MK>::DGRU ? make_quant(pl[L.o_shida], bits_a) : q.oa;
    return q;
}
__device__ __forceinline__ QK make_qk(const QSc& q, const WQ& w) {
    QK k;
    k.qn = q.add.qn; k.qp = q.add.qp;
    k.inv_xa = q.xa.inv; k.inv_ha = q.ha.inv; k.inv_oa = q.oa.inv; k.inv_hida = q.hida.inv; k.inv_add = q.add.inv;
    k.Sx = q.xa.s * w.x.s; k.Sh = q.ha.s * w.h.s; k.So = q.oa.s * w.o.s; k.Shid = q.hida.s * w.hid.s;
    k.s_add = q.add.s; k.s_mul = q.mul.s; k.c_ma = q.mul.s * q.add.inv;
    k.c_sm = q.sig.s * q.mul.inv; k.c_tm = q.tnh.s * q.mul.inv;
    k.s_xa = q.xa.s; k.s_ha = q.ha.s; k.s_oa = q.oa.s; k.s_hida = q.hida.s;
    k.s_xw = w.x.s; k.s_hw = w.h.s; k.s_ow = w.o.s; k.s_hidw = w.hid.s;
    k.inv_sig = q.sig.inv; k.inv_tnh = q.tnh.inv;
    return k;
}
}  // namespace q16
}  // namespace odpd
