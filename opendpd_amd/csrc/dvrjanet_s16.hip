// dvrjanet_s16.hip — DVRJANET (backbones/dvrjanet.py:5-112) in the S16 mapping (see gru_s16.hip / odpd_s16.h; structure as
// janet_s16.hip): a wave = 16 sequences, lane (n = sequence, q = unit quad) owns units 4q + i (hidden <= 16); the seven HxH blocks
// (W_ph, W_ah, W_f acting on hs = h_I + h_Q; the two halves of W_ccos on h_I and a cos(th); the two halves of W_csin on h_Q and
// a sin(th)) are exact-fp32 MFMA mat-vecs with operands streamed from an LDS table, their transposes carry the data gradients, the
// weight gradients are 16x16 MFMA outer-product tiles fed through per-wave LDS transposes.  Per step (dvrjanet.py:54-94):
//   mag = |x|, theta = atan2(Q, I);  th = w_pt theta + W_ph hs;  ap = w_ax mag + W_ah hs;  a = sum_k c_k |ap - k/K|  (DVR, :32-42);
//   f = s(W_f hs + b);  g_c = tanh(W_cc [h_I, a cos th] + b),  g_s = tanh(W_cs [h_Q, a sin th] + b);
//   h_I' = f h_I + (1-f) g_c,  h_Q' = f h_Q + (1-f) g_s;  y = (w_o1 . h_I' + b, w_o2 . h_Q' + b).
// K = num_dvr_units (odpd_model_t::bits_w, <= 8 here).  sin / cos / atan2 are the device library's accurate functions (the
// reference's are torch's).  The f rows are stored pre-multiplied by -log2(e).  BPTT: checkpoint of (h_I, h_Q) every kCkptStride
// steps + block recompute; dL/dx through theta and |x|.  One mapping for every batch size (there is no row-rotated DVRJANET kernel).
#include "odpd_s16.h"

namespace odpd {
namespace {

constexpr int kDvrMaxK = 8;
struct DvrLayout { int H, K, o_cs, o_wph, o_wpt, o_wah, o_wax, o_wf, o_bf, o_wcc, o_bcc, o_wcs, o_bcs, o_wo1, o_bo1, o_wo2, o_bo2, P; };
__host__ __device__ inline DvrLayout dvr_layout(int H, int K) {
    DvrLayout L; L.H = H; L.K = K; int o = 0;
    L.o_cs = o; o += K;
    L.o_wph = o; o += H * H; L.o_wpt = o; o += H; L.o_wah = o; o += H * H; L.o_wax = o; o += H;
    L.o_wf = o; o += H * H; L.o_bf = o; o += H;
    L.o_wcc = o; o += 2 * H * H; L.o_bcc = o; o += H;
    L.o_wcs = o; o += 2 * H * H; L.o_bcs = o; o += H;
    L.o_wo1 = o; o += H; L.o_bo1 = o; o += 1; L.o_wo2 = o; o += H; L.o_bo2 = o; o += 1;
    L.P = o;
    return L;
}

// table groups (hidden <= 16: one unit tile).  Block r: 0 W_ph, 1 W_ah, 2 W_f (on hs), 3 W_cc|h, 4 W_cc|a, 5 W_cs|h, 6 W_cs|a
struct V16 {
    static constexpr int FW = 0;           // r : M_r[m][4q+e]           (f rows pre-scaled)
    static constexpr int TR = FW + 7;      // r : M_r[4q+e][m]
    static constexpr int SC = TR + 7;      // j : per-unit scalars w_pt, w_ax, b_f (pre-scaled), b_cc, b_cs at unit 4q+e
    static constexpr int WOUT = SC + 5;    // cc : w_o{1,2}[4q+e]
    static constexpr int NG = WOUT + 2;
    static constexpr int kTiles = 10;      // dth dap dfp dgc dgs | hs hIp hQp vc vs
};
__device__ __forceinline__ float v16_block(const float* pl, const DvrLayout& L, int r, int o, int k) {
    const int H = L.H;
    if (o >= H || k >= H) return 0.0f;
    switch (r) {
    case 0: return pl[L.o_wph + o * H + k];
    case 1: return pl[L.o_wah + o * H + k];
    case 2: return pl[L.o_wf + o * H + k];
    case 3: return pl[L.o_wcc + o * 2 * H + k];
    case 4: return pl[L.o_wcc + o * 2 * H + H + k];
    case 5: return pl[L.o_wcs + o * 2 * H + k];
    default: return pl[L.o_wcs + o * 2 * H + H + k];
    }
}
__device__ __forceinline__ float4 v16_entry(const float* pl, const DvrLayout& L, int grp, int m, int q) {
    const int H = L.H;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        if (grp < V16::TR) v[e] = v16_block(pl, L, grp, m, k) * (grp == 2 ? kNegLog2e : 1.0f);
        else if (grp < V16::SC) v[e] = v16_block(pl, L, grp - V16::TR, k, m);
        else if (grp < V16::WOUT) {
            const int j = grp - V16::SC;
            float s = 0.0f;
            if (k < H) s = j == 0 ? pl[L.o_wpt + k] : j == 1 ? pl[L.o_wax + k] : j == 2 ? pl[L.o_bf + k] * kNegLog2e : j == 3 ? pl[L.o_bcc + k] : pl[L.o_bcs + k];
            v[e] = s;
        } else {
            v[e] = k < H ? pl[(grp == V16::WOUT ? L.o_wo1 : L.o_wo2) + k] : 0.0f;
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void v16_build_table(float* tab, const float* pl, const DvrLayout& L, int lane, int wave, int nwb) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < V16::NG; grp += nwb) t4[grp * 64 + lane] = v16_entry(pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}
__device__ __forceinline__ float v16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// the DVR knots k/K, k = 1..K (dvrjanet.py:38-40: |x - k/num_k| c_k), formed on the host and passed by value (scalar registers)
struct DvrKnots { float v[kDvrMaxK]; };
inline DvrKnots dvr_knots(int K) {
    DvrKnots kn;
    for (int k = 0; k < kDvrMaxK; ++k) kn.v[k] = (float)((double)(k + 1) / (double)K);
    return kn;
}
// wave-uniform scalars: the DVR coefficients, their knots k/K and the two output biases
struct V16Uni {
    float cs[kDvrMaxK], knot[kDvrMaxK], bo1, bo2;
    int K;
    __device__ __forceinline__ void load(const float* pl, const DvrLayout& L, const DvrKnots& kn) {
        K = L.K;
#pragma unroll
        for (int k = 0; k < kDvrMaxK; ++k) {
            cs[k] = v16_uni(pl[L.o_cs + (k < L.K ? k : 0)]) * (k < L.K ? 1.0f : 0.0f);      // unused units: coefficient 0
            knot[k] = kn.v[k];
        }
        bo1 = v16_uni(pl[L.o_bo1]); bo2 = v16_uni(pl[L.o_bo2]);
    }
};
__device__ __forceinline__ void v16_inputs(float2 xv, float& mag, float& theta) {
    mag = __builtin_amdgcn_sqrtf(__builtin_fmaf(xv.x, xv.x, xv.y * xv.y));
    theta = atan2f(xv.y, xv.x);
}
// sin and cos to ~1 ulp for |x| < 2^15 (the phases here are a few radians): three-term Cody-Waite reduction to [-pi/4, pi/4] and
// the single-precision minimax polynomials; straight-line code (the device library's sincosf carries a Payne-Hanek slow path that
// costs the backward kernel its registers)
__device__ __forceinline__ void v16_sincos(float x, float& s, float& c) {
    const float k = __builtin_rintf(x * 0.6366197723675814f);
    float r = __builtin_fmaf(k, -1.5703125f, x);
    r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(k, -7.54978995489188e-8f, r);
    const float z = r * r;
    const float ps = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float pc = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                                    z * z, __builtin_fmaf(z, -0.5f, 1.0f));
    const int j = (int)k;
    const float sv = (j & 1) ? pc : ps, cv = (j & 1) ? ps : pc;
    s = (j & 2) ? -sv : sv;
    c = ((j + 1) & 2) ? -cv : cv;
}
// a table pointer the compiler may only use once `dep` exists: keeps the LDS operand loads of a later step from being hoisted above
// the recurrence (all four steps' operands issued up front is what costs the registers)
__device__ __forceinline__ TabPtr v16_after(TabPtr p, float dep) {
    asm volatile("" : "+v"(p) : "v"(dep));
    return p;
}
__device__ __forceinline__ f32x4 v16_tied(f32x4 v, float dep) {      // the same for a saved activation
    float a = v[0], b = v[1], c = v[2], d = v[3];
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(dep));
    return f32x4{a, b, c, d};
}
__device__ __forceinline__ f32x4 v16_mv(TabPtr tl, int grp, const f32x4& v, f32x4 acc) {
    f32x4 a1[1] = {acc};
    const f32x4 v1[1] = {v};
    s16n_matvec<1>(tl, grp, v1, a1);
    return a1[0];
}

// forward step.  Padded units (>= H) have all-zero operands: th = ap = 0, f = 1/2, g = 0 -> both states stay 0; their a cos(th)
// is not 0 but only meets zero table columns.
__device__ __forceinline__ void v16_cell_fwd(TabPtr tl, const V16Uni& U, float mag, float theta, f32x4& hI, f32x4& hQ, f32x4& th,
                                             f32x4& ap, f32x4& at, f32x4& co, f32x4& si, f32x4& f, f32x4& gc, f32x4& gs) {
    const f32x4 hs = add4(hI, hQ);
    th = v16_mv(tl, V16::FW + 0, hs, mul4(as_f32x4(tab_ld(tl, (V16::SC + 0) * 64)), splat4(theta)));
    ap = v16_mv(tl, V16::FW + 1, hs, mul4(as_f32x4(tab_ld(tl, (V16::SC + 1) * 64)), splat4(mag)));
    const f32x4 pf = v16_mv(tl, V16::FW + 2, hs, as_f32x4(tab_ld(tl, (V16::SC + 2) * 64)));
    at = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kDvrMaxK; ++k)
        ODPD_EACH4 at[i] = __builtin_fmaf(__builtin_fabsf(ap[i] - U.knot[k]), U.cs[k], at[i]);
    ODPD_EACH4 { float sv, cv; v16_sincos(th[i], sv, cv); si[i] = sv; co[i] = cv; }
    f = sigmoid4_prescaled(pf);
    f32x4 pc = as_f32x4(tab_ld(tl, (V16::SC + 3) * 64)), ps = as_f32x4(tab_ld(tl, (V16::SC + 4) * 64));
    pc = v16_mv(tl, V16::FW + 3, hI, pc);
    pc = v16_mv(tl, V16::FW + 4, mul4(at, co), pc);
    ps = v16_mv(tl, V16::FW + 5, hQ, ps);
    ps = v16_mv(tl, V16::FW + 6, mul4(at, si), ps);
    gc = tanh4_precise(pc); gs = tanh4_precise(ps);
    hI = fma4(f, sub4(hI, gc), gc);              // f h + (1 - f) g
    hQ = fma4(f, sub4(hQ, gs), gs);
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void dvr16_fwd_kernel(SeqArgs a, int K, DvrKnots kn) {
    constexpr int S = kCkptStride, kWave = 2 * 2 * 16 * kChunkPad;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const DvrLayout L = dvr_layout(a.H, K);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    v16_build_table(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    V16Uni U;
    U.load(pl, L, kn);
    float* wbase = tab + s16_tab_floats(V16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * kChunkPad;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * 2 * 64 + lane : nullptr;   // [ckpt][hI | hQ][lane]
        f32x4 hI = {0.f, 0.f, 0.f, 0.f}, hQ = hI;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                float mag, theta;
                v16_inputs(xs[n * kChunkPad + tt], mag, theta);
                f32x4 th, ap, at, co, si, f, gc, gs;
                v16_cell_fwd(opaque(tl), U, mag, theta, hI, hQ, th, ap, at, co, si, f, gc, gs);
                const f32x4 w1 = as_f32x4(tab_ld(tl, V16::WOUT * 64)), w2 = as_f32x4(tab_ld(tl, (V16::WOUT + 1) * 64));
                float s0 = 0.0f, s1 = 0.0f;
                ODPD_EACH4 { s0 = __builtin_fmaf(w1[i], hI[i], s0); s1 = __builtin_fmaf(w2[i], hQ[i], s1); }
                const float y0 = quad_sum(s0) + U.bo1, y1 = quad_sum(s1) + U.bo2;
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    ck[((size_t)(t1 / S) * 2) * 64] = make_float4(hI[0], hI[1], hI[2], hI[3]);
                    ck[((size_t)(t1 / S) * 2 + 1) * 64] = make_float4(hQ[0], hQ[1], hQ[2], hQ[3]);
                }
            }
            wave_lds_fence();
            stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------
struct V16Grad {
    f32x4 t[7];                            // dW blocks in table order
    f32x4 ds[2], db[3], dwo[2];           // d w_pt, d w_ax | d b_f, d b_cc, d b_cs | d w_o1, d w_o2
    float dcs[kDvrMaxK], dbo[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 7; ++r) t[r] = z4;
        ds[0] = ds[1] = db[0] = db[1] = db[2] = dwo[0] = dwo[1] = z4;
#pragma unroll
        for (int k = 0; k < kDvrMaxK; ++k) dcs[k] = 0.f;
        dbo[0] = dbo[1] = 0.f;
    }
};

template <bool NW, bool DX, bool FULL>
__device__ __forceinline__ void v16_bwd_block(TabPtr tl0, const V16Uni& U, V16Grad& G, const float2* xs, const float2* dys, float2* dxs,
                                              float* tiles, int n, int q, int tloc, int nstep, f32x4 hI, f32x4 hQ, f32x4& dhI, f32x4& dhQ) {
    constexpr int S = kCkptStride;
    f32x4 hIp_s[S], hQp_s[S], th_s[S], ap_s[S], f_s[S], gc_s[S], gs_s[S];      // a, cos, sin are recomputed from ap / th
    float mag_s[S], theta_s[S];
    TabPtr tl = opaque(tl0);
#pragma unroll
    for (int si = 0; si < S; ++si) {
        if (FULL || si < nstep) {
            v16_inputs(xs[n * kChunkPad + tloc + si], mag_s[si], theta_s[si]);
            hIp_s[si] = hI; hQp_s[si] = hQ;
            f32x4 at, co, sn;
            v16_cell_fwd(v16_after(tl0, hI[0]), U, mag_s[si], theta_s[si], hI, hQ, th_s[si], ap_s[si], at, co, sn, f_s[si], gc_s[si], gs_s[si]);
        }
    }
    tl = opaque(tl0);
    auto tile = [tiles](int qty) { return tiles + qty * kTileFloats; };     // 0 dth 1 dap 2 dfp 3 dgc 4 dgs | 5 hs 6 hIp 7 hQp 8 vc 9 vs
    const f32x4 one = splat4(1.0f), z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            tl = v16_after(tl0, dhI[0]);
            const int tt = tloc + si;
            const float2 dyv = dys[n * kChunkPad + tt];
            if constexpr (NW) { G.dbo[0] += q == 0 ? dyv.x : 0.0f; G.dbo[1] += q == 0 ? dyv.y : 0.0f; }
            const f32x4 w1 = as_f32x4(tab_ld(tl, V16::WOUT * 64)), w2 = as_f32x4(tab_ld(tl, (V16::WOUT + 1) * 64));
            const float dep = dhI[0];
            const f32x4 hIp = v16_tied(hIp_s[si], dep), hQp = v16_tied(hQp_s[si], dep), f = v16_tied(f_s[si], dep), gc = v16_tied(gc_s[si], dep),
                        gs = v16_tied(gs_s[si], dep);
            const f32x4 gI = fma4(splat4(dyv.x), w1, dhI), gQ = fma4(splat4(dyv.y), w2, dhQ);
            const f32x4 dfp = mul4(add4(mul4(gI, sub4(hIp, gc)), mul4(gQ, sub4(hQp, gs))), mul4(f, sub4(one, f)));
            f32x4 dgc, dgs;
            ODPD_EACH4 {
                dgc[i] = gI[i] * (1.0f - f[i]) * __builtin_fmaf(-gc[i], gc[i], 1.0f);
                dgs[i] = gQ[i] * (1.0f - f[i]) * __builtin_fmaf(-gs[i], gs[i], 1.0f);
            }
            f32x4 nI = mul4(gI, f), nQ = mul4(gQ, f);
            if constexpr (NW) {
                G.dwo[0] = fma4(splat4(dyv.x), fma4(f, sub4(hIp, gc), gc), G.dwo[0]);
                G.dwo[1] = fma4(splat4(dyv.y), fma4(f, sub4(hQp, gs), gs), G.dwo[1]);
                G.db[0] = add4(G.db[0], dfp); G.db[1] = add4(G.db[1], dgc); G.db[2] = add4(G.db[2], dgs);
            }
            nI = v16_mv(tl, V16::TR + 3, dgc, nI);
            nQ = v16_mv(tl, V16::TR + 5, dgs, nQ);
            const f32x4 dvc = v16_mv(tl, V16::TR + 4, dgc, z4), dvs = v16_mv(tl, V16::TR + 6, dgs, z4);
            // through a cos(th), a sin(th) and the DVR
            // (cos, sin, a and the |ap - k/K| are re-formed here, tied to this step's gradients: left free, the compiler forms all four
            // steps' worth right after the recompute and runs out of registers)
            f32x4 dth, dap, at, co, sn;
            ODPD_EACH4 {
                float thv = th_s[si][i], apv = ap_s[si][i];
                asm volatile("" : "+v"(thv), "+v"(apv) : "v"(dvc[i]));
                float sv, cv;
                v16_sincos(thv, sv, cv);
                const float dat = __builtin_fmaf(dvc[i], cv, dvs[i] * sv);
                float slope = 0.0f, av = 0.0f;
#pragma unroll
                for (int k = 0; k < kDvrMaxK; ++k) {
                    const float u = apv - U.knot[k], au = __builtin_fabsf(u);
                    av = __builtin_fmaf(au, U.cs[k], av);
                    if constexpr (NW) G.dcs[k] = __builtin_fmaf(dat, au, G.dcs[k]);
                    slope = __builtin_fmaf(U.cs[k], u > 0.0f ? 1.0f : (u < 0.0f ? -1.0f : 0.0f), slope);     // torch.abs: gradient 0 at 0
                }
                dth[i] = av * __builtin_fmaf(dvs[i], cv, -dvc[i] * sv);
                dap[i] = dat * slope;
                at[i] = av; co[i] = cv; sn[i] = sv;
            }
            const f32x4 wpt = as_f32x4(tab_ld(tl, (V16::SC + 0) * 64)), wax = as_f32x4(tab_ld(tl, (V16::SC + 1) * 64));
            if constexpr (NW) {
                G.ds[0] = fma4(dth, splat4(theta_s[si]), G.ds[0]);
                G.ds[1] = fma4(dap, splat4(mag_s[si]), G.ds[1]);
            }
            if constexpr (DX) {
                float gth = 0.0f, gmag = 0.0f;
                ODPD_EACH4 { gth = __builtin_fmaf(wpt[i], dth[i], gth); gmag = __builtin_fmaf(wax[i], dap[i], gmag); }
                gth = quad_sum(gth); gmag = quad_sum(gmag);
                // mag = |x|: d/dI = I/mag;  theta = atan2(Q, I): d/dI = -Q/mag^2, d/dQ = I/mag^2
                const float2 xv = xs[n * kChunkPad + tt];
                const float im = fast_rcp(mag_s[si]), gm = gmag * im, gt = gth * im * im;
                if (q == 0) dxs[n * kChunkPad + tt] = make_float2(__builtin_fmaf(gm, xv.x, -gt * xv.y), __builtin_fmaf(gm, xv.y, gt * xv.x));
            }
            f32x4 dhs = v16_mv(tl, V16::TR + 0, dth, z4);
            dhs = v16_mv(tl, V16::TR + 1, dap, dhs);
            dhs = v16_mv(tl, V16::TR + 2, dfp, dhs);          // (only the forward copy of W_f carries -log2(e))
            dhI = add4(nI, dhs); dhQ = add4(nQ, dhs);
            if constexpr (NW) {
                // weight gradients: dM_r += d_r^T (x) src_r
                wave_lds_fence();
                tile_put(tile(0), n, q, dth); tile_put(tile(1), n, q, dap); tile_put(tile(2), n, q, dfp);
                tile_put(tile(3), n, q, dgc); tile_put(tile(4), n, q, dgs);
                tile_put(tile(5), n, q, add4(hIp, hQp)); tile_put(tile(6), n, q, hIp); tile_put(tile(7), n, q, hQp);
                tile_put(tile(8), n, q, mul4(at, co)); tile_put(tile(9), n, q, mul4(at, sn));
                wave_lds_fence();
                float dT[5][4], sT[5][4];
#pragma unroll
                for (int j = 0; j < 5; ++j) { tile_get(tile(j), n, q, dT[j]); tile_get(tile(5 + j), n, q, sT[j]); }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    G.t[0] = mfma4(dT[0][c], sT[0][c], G.t[0]);      // W_ph  : dth (x) hs
                    G.t[1] = mfma4(dT[1][c], sT[0][c], G.t[1]);      // W_ah  : dap (x) hs
                    G.t[2] = mfma4(dT[2][c], sT[0][c], G.t[2]);      // W_f   : dfp (x) hs
                    G.t[3] = mfma4(dT[3][c], sT[1][c], G.t[3]);      // W_cc|h: dgc (x) hI
                    G.t[4] = mfma4(dT[3][c], sT[3][c], G.t[4]);      // W_cc|a: dgc (x) a cos
                    G.t[5] = mfma4(dT[4][c], sT[2][c], G.t[5]);      // W_cs|h: dgs (x) hQ
                    G.t[6] = mfma4(dT[4][c], sT[4][c], G.t[6]);      // W_cs|a: dgs (x) a sin
                }
            }
        }
    }
}

__device__ __forceinline__ void v16_write_row(float* prow, const DvrLayout& L, V16Grad& G, int lane, int n, int q) {
    const int H = L.H;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int o = 4 * q + rr;
        if (o < H && n < H) {
            prow[L.o_wph + o * H + n] = G.t[0][rr];
            prow[L.o_wah + o * H + n] = G.t[1][rr];
            prow[L.o_wf + o * H + n] = G.t[2][rr];
            prow[L.o_wcc + o * 2 * H + n] = G.t[3][rr];
            prow[L.o_wcc + o * 2 * H + H + n] = G.t[4][rr];
            prow[L.o_wcs + o * 2 * H + n] = G.t[5][rr];
            prow[L.o_wcs + o * 2 * H + H + n] = G.t[6][rr];
        }
        const float s0 = row_sum16(G.ds[0][rr]), s1 = row_sum16(G.ds[1][rr]);
        const float b0 = row_sum16(G.db[0][rr]), b1 = row_sum16(G.db[1][rr]), b2 = row_sum16(G.db[2][rr]);
        const float w0 = row_sum16(G.dwo[0][rr]), w1 = row_sum16(G.dwo[1][rr]);
        if (n == 0 && o < H) {
            prow[L.o_wpt + o] = s0; prow[L.o_wax + o] = s1;
            prow[L.o_bf + o] = b0; prow[L.o_bcc + o] = b1; prow[L.o_bcs + o] = b2;
            prow[L.o_wo1 + o] = w0; prow[L.o_wo2 + o] = w1;
        }
    }
#pragma unroll
    for (int k = 0; k < kDvrMaxK; ++k) {        // sum over the wave's units and sequences
        float v = G.dcs[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0 && k < L.K) prow[L.o_cs + k] = v;
    }
    const float d0 = row_sum16(G.dbo[0]), d1 = row_sum16(G.dbo[1]);      // accumulated on the q == 0 lanes only
    if (lane == 0) { prow[L.o_bo1] = d0; prow[L.o_bo2] = d1; }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(256, 1) void dvr16_bwd_kernel(SeqArgs a, int K, DvrKnots kn) {
    constexpr int S = kCkptStride;
    constexpr int kWave = (DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? V16::kTiles * kTileFloats : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const DvrLayout L = dvr_layout(a.H, K);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    v16_build_table(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    V16Uni U;
    U.load(pl, L, kn);
    float* wbase = tab + s16_tab_floats(V16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * kChunkPad;
    float2* dxs = dys + 16 * kChunkPad;
    float* tiles = reinterpret_cast<float*>(dys + (DX ? 2 : 1) * 16 * kChunkPad);
    V16Grad G;
    G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * 2 * 64 + lane;
        f32x4 dhI = {0.f, 0.f, 0.f, 0.f}, dhQ = dhI;
        int cur_chunk = -1;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        const int pt0 = cur_chunk * kChunk;
                        wave_lds_fence();
                        stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    }
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
                stage_in<16>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 hI0 = blk ? as_f32x4(ck[((size_t)blk * 2) * 64]) : z4, hQ0 = blk ? as_f32x4(ck[((size_t)blk * 2 + 1) * 64]) : z4;
            if (nstep == S) v16_bwd_block<NW, DX, true>(tl, U, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, hI0, hQ0, dhI, dhQ);
            else v16_bwd_block<NW, DX, false>(tl, U, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, hI0, hQ0, dhI, dhQ);
        }
        if constexpr (DX) {
            if (cur_chunk >= 0) {
                const int pt0 = cur_chunk * kChunk;
                wave_lds_fence();
                stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                wave_lds_fence();
            }
        }
    }
    if constexpr (NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        v16_write_row(smem + wave * P4, L, G, lane, n, q);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

LaunchShape v16_shape(int ngroups, int waves) {
    LaunchShape ls;
    ls.waves = waves;
    const int need = (ngroups + waves - 1) / waves, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
template <bool NW, bool DX>
int v16_launch_bwd(hipStream_t st, const SeqArgs& a, int P, int K) {
    const LaunchShape ls = v16_shape(a.ngroups, 4);
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(V16::NG) +
                  (size_t)ls.waves * ((DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? V16::kTiles * kTileFloats : 0))) * sizeof(float);
    if (NW && lds < reduce_scratch_bytes(P, ls.waves)) lds = reduce_scratch_bytes(P, ls.waves);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = dvr16_bwd_kernel<NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a, K, dvr_knots(K));
    return (int)hipGetLastError();
}


// -------------------------------------------------------------------------------------------------
// Gate-parallel fused train kernel for the reference's own batch sizes (train_funcs.py:28-48; every frame gets a SIMD of its own): ONE
// sequence per single-wave workgroup, every row of the wave carries h_I and h_Q of the 16 units.
//   front     |x| and atan2 of the frame, lane = time step;
//   forward   round A on hs = h_I + h_Q: rows W_ph | W_ah | W_f | -, one rotated dot product per row, the three results handed to every row;
//             the DVR, sin / cos; round B: rows W_cc|h on h_I, W_cc|a on a cos, W_cs|h on h_Q, W_cs|a on a sin, summed across the row
//             pairs, the other pair's tanh by one swap; a step record (cos, sin, a, ap, f, g_c, g_s, h_I, h_Q) parked in LDS;
//   head      both read-outs, loss, dL/dy with lane = time step;
//   backward  the two rounds with the transposed blocks; the step's seven H x H weight gradients as TWO 4-block MFMAs
//             (v_mfma_f32_16x16x1_4b_f32): (d_gc | d_gc | d_gs | d_gs) x (h_I | a cos | h_Q | a sin) and (d_th | d_ap | d_f | 0) x hs.
// Weight gradients only (the frozen-PA role stays on the S16 kernels).  Taken while the frame's records fit the CU's LDS.
// -------------------------------------------------------------------------------------------------
constexpr int kVgpRec = 9 * 16 + 1;          // floats per step record (odd: conflict-free with lane = unit and with lane = time step)
__host__ __device__ inline int vgp_buffer_floats(int T) { return kVgpRec * (T + 1) + 4 * ((T + 3) & ~3) + 256; }
__global__ __launch_bounds__(64) void dvr_gp_train_kernel(SeqArgs a, int K, DvrKnots kn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;
    const DvrLayout L = dvr_layout(a.H, K);
    const int H = L.H, T = a.T, Tp = (T + 3) & ~3;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* rec = smem + pad4(L.P);                                   // record t + 1 = step t; record 0: h_I = h_Q = 0
    float2* mt = reinterpret_cast<float2*>(rec + kVgpRec * (T + 1)); // (|x|, theta) of step t
    float2* dyb = mt + Tp;
    float* dump = reinterpret_cast<float*>(dyb + Tp);
    V16Uni U;
    U.load(pl, L, kn);
    float wA[16], wB[16], wAT[16], wBT[16];
    {
        const int dir = rot_dir(col);
        const int oa = role == 0 ? L.o_wph : role == 1 ? L.o_wah : L.o_wf;
        const int ob = (role < 2 ? L.o_wcc : L.o_wcs) + (role & 1) * H;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15;
            const bool ok = col < H && m < H;
            wA[k] = (ok && role < 3) ? pl[oa + col * H + m] : 0.0f;
            wAT[k] = (ok && role < 3) ? pl[oa + m * H + col] : 0.0f;
            wB[k] = ok ? pl[ob + col * 2 * H + m] : 0.0f;
            wBT[k] = ok ? pl[ob + m * 2 * H + col] : 0.0f;
        }
    }
    const bool vo = col < H;
    const float sA = vo ? (role == 0 ? pl[L.o_wpt + col] : role == 1 ? pl[L.o_wax + col] : 0.0f) : 0.0f;      // the row's scalar-input column
    const float bA = (vo && role == 2) ? pl[L.o_bf + col] : 0.0f;
    const float bB = vo ? pl[(role < 2 ? L.o_bcc : L.o_bcs) + col] : 0.0f;
    const float wo1 = vo ? pl[L.o_wo1 + col] : 0.0f, wo2 = vo ? pl[L.o_wo2 + col] : 0.0f;
    const RowMasks rm = row_masks();
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    // the record fields a row parks: row 0 cos, sin, a | row 1 ap, f | row 2 g_c, g_s | row 3 h_I, h_Q
    const int f0 = role == 0 ? 0 : role == 1 ? 3 : role == 2 ? 5 : 7;
    const int pk0 = (int)(rec - smem) + kVgpRec + f0 * 16 + col;
    const int pk2 = role == 0 ? pk0 + 32 : (int)(dump - smem) + lane;
    if (lane < 32) rec[7 * 16 + lane] = 0.0f;

    f32x16 acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
    float dcs[kDvrMaxK], dw1[16], dw2[16];
#pragma unroll
    for (int k = 0; k < kDvrMaxK; ++k) dcs[k] = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { dw1[j] = 0.0f; dw2[j] = 0.0f; }
    float dsA = 0.0f, dbf = 0.0f, dbcc = 0.0f, dbcs = 0.0f, dbo1 = 0.0f, dbo2 = 0.0f, loss_acc = 0.0f;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        wave_lds_fence();
        for (int t = lane; t < T; t += 64) {
            float mag, theta;
            v16_inputs(xg[t], mag, theta);
            mt[t] = make_float2(mag, theta);
        }
        wave_lds_fence();
        // ---- forward recurrence ----
        {
            float hI = 0.0f, hQ = 0.0f;
            int pk = pk0, pk_c = pk2;
            for (int t = 0; t < T; ++t) {
                const float2 in = mt[t];
                const float pa = rotdot(0.0f, wA, hI + hQ);
                const float own = __builtin_fmaf(sA, vsel(rm.m[0], in.y, in.x), pa) + bA;
                float g4[4];
                gather_rows(vsel(rm.m[2], sigmoidf_(own), own), g4);
                const float th = g4[0], ap = g4[1], f = g4[2];
                float at = 0.0f;
#pragma unroll
                for (int k = 0; k < kDvrMaxK; ++k) at = __builtin_fmaf(__builtin_fabsf(ap - U.knot[k]), U.cs[k], at);
                float si, co;
                v16_sincos(th, si, co);
                const float vc = at * co, vs = at * si;
                const float opnd = vsel(rm.m[0], hI, vsel(rm.m[1], vc, vsel(rm.m[2], hQ, vs)));
                float pb = rotdot(0.0f, wB, opnd);
                pb += xor16(pb);
                const float gv = tanhf_(pb + bB), ogv = xor32(gv);
                const float gc = vsel(rm.m[0] | rm.m[1], gv, ogv), gs = vsel(rm.m[0] | rm.m[1], ogv, gv);
                hI = __builtin_fmaf(f, hI - gc, gc);
                hQ = __builtin_fmaf(f, hQ - gs, gs);
                smem[pk] = vsel(rm.m[0], co, vsel(rm.m[1], ap, vsel(rm.m[2], gc, hI)));
                smem[pk + 16] = vsel(rm.m[0], si, vsel(rm.m[1], f, vsel(rm.m[2], gs, hQ)));
                smem[pk_c] = at;
                pk += kVgpRec;
                pk_c += role == 0 ? kVgpRec : 0;
            }
        }
        wave_lds_fence();
        // ---- read-outs, loss and dL/dy of every step; lane = time step ----
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
                const float* hv = rec + (t + 1) * kVgpRec + 7 * 16;
                float y0 = U.bo1, y1 = U.bo2;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < H) {
                        y0 = __builtin_fmaf(pl[L.o_wo1 + j], hv[j], y0);
                        y1 = __builtin_fmaf(pl[L.o_wo2 + j], hv[16 + j], y1);
                    }
                const float2 tv = tg[t];
                float dy0, dy1;
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
                dbo1 += dy0; dbo2 += dy1;
                dyb[t] = make_float2(dy0, dy1);
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < H) {
                        dw1[j] = __builtin_fmaf(dy0, hv[j], dw1[j]);
                        dw2[j] = __builtin_fmaf(dy1, hv[16 + j], dw2[j]);
                    }
            }
        }
        wave_lds_fence();
        // ---- backward recurrence ----
        {
            float dhI = 0.0f, dhQ = 0.0f;
            for (int t = T - 1; t >= 0; --t) {
                const float* r1 = rec + (t + 1) * kVgpRec + col;
                const float co = r1[0], si = r1[16], at = r1[32], ap = r1[48], f = r1[64], gc = r1[80], gs = r1[96];
                const float hIp = r1[7 * 16 - kVgpRec], hQp = r1[8 * 16 - kVgpRec];
                const float2 in = mt[t], dyv = dyb[t];
                const float gI = __builtin_fmaf(dyv.x, wo1, dhI), gQ = __builtin_fmaf(dyv.y, wo2, dhQ);
                const float df = __builtin_fmaf(gI, hIp - gc, gQ * (hQp - gs));
                const float dfp = df * (f * (1.0f - f));
                const float dgc = (gI * (1.0f - f)) * __builtin_fmaf(-gc, gc, 1.0f);
                const float dgs = (gQ * (1.0f - f)) * __builtin_fmaf(-gs, gs, 1.0f);
                const float dB_row = vsel(rm.m[0] | rm.m[1], dgc, dgs);
                float g4[4];
                gather_rows(rotdot(0.0f, wBT, dB_row), g4);                       // h_I's share | dL/d(a cos) | h_Q's share | dL/d(a sin)
                const float dvc = g4[1], dvs = g4[3];
                const float dat = __builtin_fmaf(dvc, co, dvs * si);
                const float dth = at * __builtin_fmaf(dvs, co, -(dvc * si));
                float slope = 0.0f;
#pragma unroll
                for (int k = 0; k < kDvrMaxK; ++k) {
                    const float u = ap - U.knot[k];
                    dcs[k] = __builtin_fmaf(dat, __builtin_fabsf(u), dcs[k]);
                    slope = __builtin_fmaf(U.cs[k], s16_sign(u), slope);
                }
                const float dap = dat * slope;
                const float dA_row = vsel(rm.m[0], dth, vsel(rm.m[1], dap, vsel(rm.m[2], dfp, 0.0f)));
                float pat = rotdot(0.0f, wAT, dA_row);
                pat = sum_rows4(pat);
                dhI = __builtin_fmaf(gI, f, g4[0]) + pat;
                dhQ = __builtin_fmaf(gQ, f, g4[2]) + pat;
                const float opnd = vsel(rm.m[0], hIp, vsel(rm.m[1], at * co, vsel(rm.m[2], hQp, at * si)));
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(dB_row, opnd, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(dA_row, hIp + hQp, acc2, 0, 0, 0);
                dsA = __builtin_fmaf(dA_row, vsel(rm.m[0], in.y, in.x), dsA);      // row 0: W_ptheta, row 1: W_ax
                dbf += dfp; dbcc += dgc; dbcs += dgs;
            }
        }
    }
    // ---- the workgroup's row of partial gradients (every entry written) ----
    float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
    float lp = loss_acc, s0 = dbo1, s1 = dbo2;
    for (int o = 32; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float v1 = dw1[j], v2 = dw2[j];
        for (int o = 32; o > 0; o >>= 1) { v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
        if (lane == 0 && j < H) { prow[L.o_wo1 + j] = v1; prow[L.o_wo2 + j] = v2; }
    }
#pragma unroll
    for (int k = 0; k < kDvrMaxK; ++k) {
        float v = (role == 0 && vo) ? dcs[k] : 0.0f;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0 && k < K) prow[L.o_cs + k] = v;
    }
    if (vo) {
        if (role == 0) { prow[L.o_wpt + col] = dsA; prow[L.o_bf + col] = dbf; prow[L.o_bcc + col] = dbcc; prow[L.o_bcs + col] = dbcs; }
        if (role == 1) prow[L.o_wax + col] = dsA;
    }
    if (lane == 0) {
        prow[L.o_bo1] = s0; prow[L.o_bo2] = s1;
        prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
    }
    // 4-block MFMAs: register 4 blk + rr of lane l = entry (4 (l / 16) + rr, l % 16) of block blk
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * role + rr;
            if (i < H && col < H) {
                prow[(blk < 2 ? L.o_wcc : L.o_wcs) + i * 2 * H + (blk & 1) * H + col] = acc1[4 * blk + rr];
                if (blk < 3) prow[(blk == 0 ? L.o_wph : blk == 1 ? L.o_wah : L.o_wf) + i * H + col] = acc2[4 * blk + rr];
            }
        }
}

// Evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): ONE sequence per wave, the forward half of
// dvr_gp_train_kernel in chunks of kVevChunk steps; no checkpoints.
constexpr int kVevChunk = 256, kVevPitch = 33;
constexpr int kVevFloats = 2 * kVevChunk + kVevPitch * kVevChunk + 64;
__global__ __launch_bounds__(64) void dvr_gp_eval_kernel(SeqArgs a, int K, DvrKnots kn) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int Tc = kVevChunk;
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;
    const DvrLayout L = dvr_layout(a.H, K);
    const int H = L.H, T = a.T;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float2* mt = reinterpret_cast<float2*>(smem + pad4(L.P));       // (|x|, theta) of step tt
    float* hh = reinterpret_cast<float*>(mt + Tc);                   // [Tc][33]: h_I | h_Q after step tt
    float* dump = hh + kVevPitch * Tc;
    V16Uni U;
    U.load(pl, L, kn);
    float wA[16], wB[16];
    {
        const int dir = rot_dir(col);
        const int oa = role == 0 ? L.o_wph : role == 1 ? L.o_wah : L.o_wf;
        const int ob = (role < 2 ? L.o_wcc : L.o_wcs) + (role & 1) * H;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15;
            const bool ok = col < H && m < H;
            wA[k] = (ok && role < 3) ? pl[oa + col * H + m] : 0.0f;
            wB[k] = ok ? pl[ob + col * 2 * H + m] : 0.0f;
        }
    }
    const bool vo = col < H;
    const float sA = vo ? (role == 0 ? pl[L.o_wpt + col] : role == 1 ? pl[L.o_wax + col] : 0.0f) : 0.0f;
    const float bA = (vo && role == 2) ? pl[L.o_bf + col] : 0.0f;
    const float bB = vo ? pl[(role < 2 ? L.o_bcc : L.o_bcs) + col] : 0.0f;
    const RowMasks rm = row_masks();
    const int pk0 = role < 2 ? (int)(hh - smem) + role * 16 + col : (int)(dump - smem) + lane, pk_step = role < 2 ? kVevPitch : 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float hI = 0.0f, hQ = 0.0f;
        for (int t0 = 0; t0 < T; t0 += Tc) {
            const int len = min(Tc, T - t0);
            wave_lds_fence();
            for (int tt = lane; tt < len; tt += 64) {
                float mag, theta;
                v16_inputs(xg[t0 + tt], mag, theta);
                mt[tt] = make_float2(mag, theta);
            }
            wave_lds_fence();
            {
                int pk = pk0;
                for (int tt = 0; tt < len; ++tt) {
                    const float2 in = mt[tt];
                    const float pa = rotdot(0.0f, wA, hI + hQ);
                    const float own = __builtin_fmaf(sA, vsel(rm.m[0], in.y, in.x), pa) + bA;
                    float g4[4];
                    gather_rows(vsel(rm.m[2], sigmoidf_(own), own), g4);
                    const float th = g4[0], ap = g4[1], f = g4[2];
                    float at = 0.0f;
#pragma unroll
                    for (int k = 0; k < kDvrMaxK; ++k) at = __builtin_fmaf(__builtin_fabsf(ap - U.knot[k]), U.cs[k], at);
                    float si, co;
                    v16_sincos(th, si, co);
                    const float opnd = vsel(rm.m[0], hI, vsel(rm.m[1], at * co, vsel(rm.m[2], hQ, at * si)));
                    float pb = rotdot(0.0f, wB, opnd);
                    pb += xor16(pb);
                    const float gv = tanhf_(pb + bB), ogv = xor32(gv);
                    const float gc = vsel(rm.m[0] | rm.m[1], gv, ogv), gs = vsel(rm.m[0] | rm.m[1], ogv, gv);
                    hI = __builtin_fmaf(f, hI - gc, gc);
                    hQ = __builtin_fmaf(f, hQ - gs, gs);
                    smem[pk] = vsel(rm.m[0], hI, hQ);
                    pk += pk_step;
                }
            }
            wave_lds_fence();
            for (int tt = lane; tt < len; tt += 64) {
                const float* hv = hh + tt * kVevPitch;
                float y0 = U.bo1, y1 = U.bo2;
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < H) {
                        y0 = __builtin_fmaf(pl[L.o_wo1 + j], hv[j], y0);
                        y1 = __builtin_fmaf(pl[L.o_wo2 + j], hv[16 + j], y1);
                    }
                yg[t0 + tt] = make_float2(y0, y1);
            }
        }
    }
}

static size_t dvr_gp_lds_bytes(int P, int T) { return ((size_t)pad4(P) + vgp_buffer_floats(T)) * sizeof(float); }
static int dvr_gp_blocks_per_cu(int P, int T) {
    const size_t lds = dvr_gp_lds_bytes(P, T);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}

}  // namespace

bool dvrjanet_ok(const odpd_model_t* m) { return m->hidden >= 1 && m->hidden <= 16 && m->bits_w >= 1 && m->bits_w <= kDvrMaxK; }
int64_t dvrjanet_param_count(const odpd_model_t* m) {
    return dvrjanet_ok(m) ? (int64_t)dvr_layout(m->hidden, m->bits_w).P : (int64_t)ODPD_EUNSUPPORTED;
}
int dvrjanet_rows(const odpd_model_t* m, int B) {
    if (!dvrjanet_ok(m)) return ODPD_EUNSUPPORTED;
    return v16_shape((B + 15) / 16, 4).grid;
}
int64_t dvrjanet_ckpt_floats(const odpd_model_t* m, int B, int T) {
    if (!dvrjanet_ok(m)) return ODPD_EUNSUPPORTED;
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * 2 * 256;
}
// the gate-parallel fused train kernel: one sequence per single-wave workgroup, the frame's step records in LDS
bool dvrjanet_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (!dvrjanet_ok(m)) return false;
    const int per_cu = dvr_gp_blocks_per_cu(dvr_layout(m->hidden, m->bits_w).P, T);
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && per_cu > 0;
    // up to five rounds of workgroups (measured: profiles/r03/gp_train_bench_f4.txt): the alternative is the forward / loss / backward chain of the S16 kernels
    return (long)B <= 5L * device_cus() * per_cu;
}
int dvrjanet_gp_rows(const odpd_model_t* m, int B, int T) {
    const long cap = (long)device_cus() * dvr_gp_blocks_per_cu(dvr_layout(m->hidden, m->bits_w).P, T);
    return B < cap ? B : (int)cap;
}
int dvrjanet_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const int K = m->bits_w;
    const size_t lds = dvr_gp_lds_bytes(dvr_layout(m->hidden, K).P, a.T);
    if (int e = allow_big_lds(dvr_gp_train_kernel, lds)) return e;
    hipLaunchKernelGGL(dvr_gp_train_kernel, dim3(dvrjanet_gp_rows(m, a.B, a.T)), dim3(64), lds, st, a, K, dvr_knots(K));
    return (int)hipGetLastError();
}
// mode 1 forward, 2 backward
int dvrjanet_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    if (!dvrjanet_ok(m)) return ODPD_EUNSUPPORTED;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int K = m->bits_w, P = dvr_layout(m->hidden, K).P;
    if (mode == 1 && !a.ckpt && a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0) {
        // sequences that each get a SIMD of their own (inference: no checkpoints)
        const size_t lds = ((size_t)pad4(P) + kVevFloats) * sizeof(float);
        if (int e = allow_big_lds(dvr_gp_eval_kernel, lds)) return e;
        hipLaunchKernelGGL(dvr_gp_eval_kernel, dim3(a.B), dim3(64), lds, st, a, K, dvr_knots(K));
        return (int)hipGetLastError();
    }
    if (mode == 1) {
        const LaunchShape ls = v16_shape(a.ngroups, a.ngroups <= 4 * device_cus() ? 4 : 8);
        const size_t lds = ((size_t)pad4(P) + s16_tab_floats(V16::NG) + (size_t)ls.waves * (2 * 2 * 16 * kChunkPad)) * sizeof(float);
        auto k = dvr16_fwd_kernel;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a, K, dvr_knots(K));
        return (int)hipGetLastError();
    }
    if (!a.ckpt && a.nck > 1) return ODPD_EINVAL;
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (!nw && !dx) return ODPD_EINVAL;
    if (nw && dx) return v16_launch_bwd<true, true>(st, a, P, K);
    if (nw) return v16_launch_bwd<true, false>(st, a, P, K);
    return v16_launch_bwd<false, true>(st, a, P, K);
}

}  // namespace odpd
