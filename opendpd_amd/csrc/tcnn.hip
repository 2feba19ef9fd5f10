// tcnn.hip — kernels for the TCNN backbone (backbones/tcnn.py:5-97):
//   feat = [I,Q,a,a^3,sin,cos] -> Conv1d(6->C,k1,bias) -> Hardswish -> 4 x [depthwise Conv1d(C,C,k5,dil d,pad 2d,no bias)
//   -> Hardswish], d = 1,2,4,8 -> Conv1d(C->2,k1,no bias);  y = net + [I,Q]                       (tcnn.py:82-96)
// The network is separable per channel except for the two 1x1 convolutions, and non-recurrent.
//
// Register-resident time tiles: a 16-lane DPP row owns one sequence (a wave = 4 sequences), lane r of the row owns the
// R consecutive steps [R r, R r + R) of a 16 R-step tile in registers.  A dilated tap at offset o of element i is
// element (i + o) mod R of lane r + floor((i + o) / R): in-lane taps are plain FMAs, cross-lane taps are ONE
// v_fmac_f32_dpp (row_shr / row_shl, bound_ctrl:0) — no LDS exchange, no barriers, and lanes outside the row read 0,
// which IS the zero padding of the convolutions when the tile holds the whole frame (T <= 256: R = ceil(T/16) from
// {4, 8, 13, 16}; T = 200 -> R = 13, 96 % of the slots carry a sample).  Longer frames use R = 16 tiles with a 30-step
// halo each side (receptive-field radius 2 (1+2+4+8) = 30; 60 for dL/dx).  Every stage's activation is 0 outside [0,T)
// (PyTorch pads each conv input): out-of-frame pre-activations are stored as -100, for which hardswish and its
// derivative are both exactly 0.
//   forward : wave = (4 sequences, channel subset); channels looped inside, y accumulates in registers; for small
//             batches the channels of a tile are split over up to 4 waves of the workgroup (LDS sum of y).
//   backward: wave = (4 sequences, channel subset) x loop over tiles; per channel the five stages are recomputed
//             (pre-activations stay in registers), back-propagated, the 29 per-channel gradients are reduced over the
//             wave and added to the wave's accumulator row in LDS; block sum -> partial row blockIdx.x.
//   dL/dx   : (frozen PA of a cascade) as backward without the weight gradients; dL/dfeat accumulates in registers.
//
// NTX = true: the NeuralTX backbone (backbones/neuraltx.py:5-137) on the same tiles — a complex 5-tap FIR in front (conv_I / conv_Q,
// zero padding 2: the taps are the same DPP row shifts as the depthwise taps), the stack on the 4 features [f_I, f_Q, |f|, |f|^3]
// of the FILTERED signal, and y = net + IQ_match f + f.  Its backward kernels additionally accumulate dL/dfeat over the wave's
// channels and turn it into the FIR tap gradients (weights kernel) or, through the transposed FIR, into dL/dx (dx kernel).
#include <type_traits>
#include <utility>

#include "odpd_seq.h"
#include "odpd_quant.h"

namespace odpd {

struct TcnnLayout { int C, F, o_ci, o_cq, o_w0, o_b0, o_dw[4], o_w5, o_m, P; };
// qh: `--quant` on neuraltx — IQ_match is an INT_Linear (the one layer the surgery's Conv2d / Linear map finds, quant_envs.py:145-148):
// its weight, activation and (never used: no module is named fc_out) output scales behind IQ_match.weight
__host__ __device__ inline TcnnLayout tcnn_layout(int C, bool ntx = false, bool qh = false) {
    TcnnLayout L; L.C = C; L.F = ntx ? 4 : 6; int o = 0;
    L.o_ci = o; L.o_cq = o + 5;
    if (ntx) o += 10;                                     // conv_I.weight, conv_Q.weight (neuraltx.py:18-19)
    L.o_w0 = o; o += L.F * C; L.o_b0 = o; o += C;
    for (int l = 0; l < 4; ++l) { L.o_dw[l] = o; o += 5 * C; }
    L.o_w5 = o; o += 2 * C;
    L.o_m = o;
    if (ntx) o += 4;                                      // IQ_match.weight (2,2) (neuraltx.py:38)
    if (ntx && qh) o += 3;
    L.P = o;
    return L;
}
constexpr int kTHalo = 30;     // receptive-field radius 2 * (1 + 2 + 4 + 8)
constexpr int kTHaloDx = 60;   // dx[t] needs dL/dpre at t +- 30, whose activations need x at +- 30 around them
constexpr int kNtxFir = 2;     // NeuralTX: the FIR in front adds 2 steps each side (and 2 more for the transposed FIR of dL/dx)

// tile geometry: B * ntiles row-sized work items (sequence-major), four of them per wave
struct TcnnGeom { int R, halo, tile, ntiles, ngroups; };
inline TcnnGeom tcnn_geom(int B, int T, int halo) {
    TcnnGeom g;
    if (T <= 256) { g.R = T <= 64 ? 4 : (T <= 128 ? 8 : (T <= 208 ? 13 : 16)); g.halo = 0; g.tile = 16 * g.R; g.ntiles = 1; }
    else { g.R = 16; g.halo = halo; g.tile = 256 - 2 * halo; g.ntiles = (T + g.tile - 1) / g.tile; }
    g.ngroups = (B * g.ntiles + 3) / 4;
    return g;
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
// hardswish(v) = v relu6(v + 3) / 6 and its derivative (0 | v/3 + 1/2 on [-3, 3] | 1)
__device__ __forceinline__ float hs_(float v) { return v * clamp01(__builtin_fmaf(v, 1.0f / 6.0f, 0.5f)); }
__device__ __forceinline__ float hsg_(float v) {
    const float u = __builtin_fmaf(v, 1.0f / 3.0f, 0.5f);
    return __builtin_fabsf(v) <= 3.0f ? u : clamp01(u);
}
constexpr float kDead = -100.0f;   // pre-activation of an out-of-frame step: hs_ = hsg_ = 0

constexpr int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }
// acc += w * (element I + OFF of the row's tile, 0 outside the row).  The compiler does not model hazards inside asm:
// a DPP read of a VGPR written by one of the two previous VALU instructions needs 2 wait states -> s_nop 1.
template <int R, int I, int OFF>
__device__ __forceinline__ void tap(float& acc, const float (&v)[R], float w) {
    constexpr int idx = I + OFF, S = fdiv(idx, R), E = idx - S * R;
    if constexpr (S == 0) acc = __builtin_fmaf(w, v[E], acc);
    else if constexpr (S > 0 && S < 16)
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(acc) : "v"(v[E]), "v"(w), "n"(S));
    else if constexpr (S < 0 && S > -16)
        asm("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_shr:%3 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(acc) : "v"(v[E]), "v"(w), "n"(-S));
}
// out[i] = sum_k w[k] in[i + SIGN d (k - 2)]
template <int R, int D, int SIGN>
__device__ __forceinline__ void conv5(float (&out)[R], const float (&in)[R], const float (&w)[5]) {
    static_for<R>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        float s = w[2] * in[i];
        tap<R, i, -2 * D * SIGN>(s, in, w[0]);
        tap<R, i, -D * SIGN>(s, in, w[1]);
        tap<R, i, D * SIGN>(s, in, w[3]);
        tap<R, i, 2 * D * SIGN>(s, in, w[4]);
        out[i] = s;
    });
}
// gw[k] += sum_i gp[i] in[i + d (k - 2)]
template <int R, int D>
__device__ __forceinline__ void conv5_wgrad(float (&gw)[5], const float (&gp)[R], const float (&in)[R]) {
    static_for<R>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        tap<R, i, -2 * D>(gw[0], in, gp[i]);
        tap<R, i, -D>(gw[1], in, gp[i]);
        gw[2] = __builtin_fmaf(gp[i], in[i], gw[2]);
        tap<R, i, D>(gw[3], in, gp[i]);
        tap<R, i, 2 * D>(gw[4], in, gp[i]);
    });
}

// one channel's parameters (wave-uniform)
struct TcnnChan { float w0[6], b0, dw[4][5], w5[2]; };
__device__ __forceinline__ TcnnChan tcnn_chan(const float* __restrict__ p, const TcnnLayout& L, int c) {
    TcnnChan k;
#pragma unroll
    for (int i = 0; i < 6; ++i) k.w0[i] = i < L.F ? p[L.o_w0 + c * L.F + i] : 0.0f;
    k.b0 = p[L.o_b0 + c];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int j = 0; j < 5; ++j) k.dw[l][j] = p[L.o_dw[l] + c * 5 + j];
    k.w5[0] = p[L.o_w5 + c]; k.w5[1] = p[L.o_w5 + L.C + c];
    return k;
}

// NeuralTX parameters outside the channel stack (wave-uniform)
// q (bits_w > 0): m = q_w(IQ_match.weight), mm = the weight quantiser's pass mask, qa = IQ_match's activation quantiser
struct NtxFir { float ci[5], cq[5], m[4], mm[4]; q16::Quant qa; bool q; };
__device__ __forceinline__ NtxFir ntx_fir(const float* __restrict__ p, const TcnnLayout& L, int bits_w, int bits_a) {
    NtxFir f;
#pragma unroll
    for (int k = 0; k < 5; ++k) { f.ci[k] = p[L.o_ci + k]; f.cq[k] = p[L.o_cq + k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { f.m[k] = p[L.o_m + k]; f.mm[k] = 1.0f; }
    f.q = bits_w > 0;
    f.qa = q16::Quant{1.0f, 1.0f, 0.0f, 0.0f};
    if (f.q) {      // (wave-uniform values formed on the VALU: moved to scalar registers, the hot loops are register-bound)
        auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
        const q16::Quant qw = q16::make_quant(p[L.o_m + 4], bits_w);
        const q16::Quant qa = q16::make_quant(p[L.o_m + 5], bits_a);
        f.qa = q16::Quant{uni(qa.s), uni(qa.inv), uni(qa.qn), uni(qa.qp)};
#pragma unroll
        for (int k = 0; k < 4; ++k) { f.mm[k] = uni(q16::qpass(f.m[k], qw)); f.m[k] = uni(q16::qapply(f.m[k], qw)); }
    }
    return f;
}

// the row's tile: sequence b, element i of lane r at time t0 + i.  xi / xq = what the channel stack sees (NTX: the filtered
// signal f); NTX keeps the raw samples in rxi / rxq (0 outside the frame = the FIR's zero padding)
template <int R, bool NTX = false>
struct TcnnTile {
    int b, t0, own0, own1;   // [own0, own1): steps this tile writes y / dx for and reads dy of (weight gradients)
    bool bok;
    float xi[R], xq[R], am[R], ia[R];
    float rxi[NTX ? R : 1], rxq[NTX ? R : 1];
    bool valid[R];
    __device__ __forceinline__ void locate(const SeqArgs& a, const TcnnGeom& g, int grp, bool active, int lane) {
        const int r = lane & 15, item = grp * 4 + (lane >> 4), ti = item % g.ntiles;
        b = item / g.ntiles; bok = active && b < a.B;
        own0 = ti * g.tile; own1 = min(own0 + g.tile, a.T);
        t0 = own0 - g.halo + R * r;
    }
    __device__ __forceinline__ void load_x(const SeqArgs& a, const NtxFir* fir = nullptr) {
        const float2* x2 = reinterpret_cast<const float2*>(a.x) + (size_t)(bok ? b : 0) * a.T;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int t = t0 + i;
            valid[i] = bok && t >= 0 && t < a.T;
            const float2 v = valid[i] ? x2[t] : (NTX ? make_float2(0.0f, 0.0f) : make_float2(1.0f, 0.0f));
            xi[i] = v.x; xq[i] = v.y;
        }
        if constexpr (NTX) {
            // f_I = cI * xI - cQ * xQ,  f_Q = cQ * xI + cI * xQ   (neuraltx.py:122-123; Conv1d = cross-correlation, padding 2)
            float nq[5], s0[R], s1[R];
#pragma unroll
            for (int k = 0; k < 5; ++k) nq[k] = -fir->cq[k];
#pragma unroll
            for (int i = 0; i < R; ++i) { rxi[i] = xi[i]; rxq[i] = xq[i]; }
            conv5<R, 1, 1>(s0, rxi, fir->ci);
            conv5<R, 1, 1>(s1, rxq, nq);
#pragma unroll
            for (int i = 0; i < R; ++i) xi[i] = s0[i] + s1[i];
            conv5<R, 1, 1>(s0, rxi, fir->cq);
            conv5<R, 1, 1>(s1, rxq, fir->ci);
#pragma unroll
            for (int i = 0; i < R; ++i) xq[i] = s0[i] + s1[i];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float a2 = __builtin_fmaf(xi[i], xi[i], xq[i] * xq[i]);
            am[i] = __builtin_amdgcn_sqrtf(a2); ia[i] = fast_rcp(am[i]);
        }
    }
    // pre-activation of the 1x1 input conv; out-of-frame steps are dead
    __device__ __forceinline__ float pre0(const TcnnChan& k, int i) const {
        float v = __builtin_fmaf(k.w0[0], xi[i], k.b0);
        v = __builtin_fmaf(k.w0[1], xq[i], v);
        v = __builtin_fmaf(k.w0[2], am[i], v);
        v = __builtin_fmaf(k.w0[3], am[i] * am[i] * am[i], v);
        if constexpr (!NTX) {
            v = __builtin_fmaf(k.w0[4], xq[i] * ia[i], v);
            v = __builtin_fmaf(k.w0[5], xi[i] * ia[i], v);
        }
        return valid[i] ? v : kDead;
    }
    __device__ __forceinline__ bool owns(int i) const { return valid[i] && t0 + i >= own0 && t0 + i < own1; }
};

// forward of one channel; pre[l] = pre-activation of stage l (0: input conv, 1..4: depthwise), returns act of stage 4
template <int R, bool KEEP, bool NTX>
__device__ __forceinline__ void tcnn_chan_fwd(const TcnnTile<R, NTX>& tl, const TcnnChan& k, float (&pre)[5][R], float (&act)[R]) {
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const float v = tl.pre0(k, i);
        if constexpr (KEEP) pre[0][i] = v;
        act[i] = hs_(v);
    }
    static_for<4>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        float s[R];
        conv5<R, 1 << l, 1>(s, act, k.dw[l]);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float v = tl.valid[i] ? s[i] : kDead;
            if constexpr (KEEP) pre[l + 1][i] = v;
            act[i] = hs_(v);
        }
    });
}

// grid = ceil(ngroups * ncw / 4) blocks of 4 waves; ncw in {1, 2, 4} waves share a tile's channels (wave j takes
// channels j, j + ncw, ...); LDS = 4 * 2R * 64 floats when ncw > 1
template <int R, bool NTX>
__global__ __launch_bounds__(256) void tcnn_fwd_kernel(SeqArgs a, TcnnGeom g, int ncw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TcnnLayout L = tcnn_layout(a.H, NTX, NTX && a.bits_w > 0);
    const int gw = blockIdx.x * 4 + wave, grp = gw / ncw, cs = gw % ncw;
    TcnnTile<R, NTX> tl;
    tl.locate(a, g, grp, grp < g.ngroups, lane);
    NtxFir fir;
    if constexpr (NTX) fir = ntx_fir(a.params, L, a.bits_w, a.bits_a);
    tl.load_x(a, &fir);
    float y0[R], y1[R];
#pragma unroll
    for (int i = 0; i < R; ++i) y0[i] = y1[i] = 0.0f;
    for (int c = cs; c < L.C; c += ncw) {
        const TcnnChan k = tcnn_chan(a.params, L, c);
        float pre[5][R], act[R];
        tcnn_chan_fwd<R, false>(tl, k, pre, act);
#pragma unroll
        for (int i = 0; i < R; ++i) {
            y0[i] = __builtin_fmaf(k.w5[0], act[i], y0[i]);
            y1[i] = __builtin_fmaf(k.w5[1], act[i], y1[i]);
        }
    }
    if (ncw > 1) {     // sum the channel subsets of the tile in wave order (deterministic)
        float* mine = smem + wave * (2 * R * 64);
#pragma unroll
        for (int i = 0; i < R; ++i) { mine[(2 * i) * 64 + lane] = y0[i]; mine[(2 * i + 1) * 64 + lane] = y1[i]; }
        __syncthreads();
        if (cs != 0) return;
        for (int j = 1; j < ncw; ++j) {
            const float* o = smem + (wave + j) * (2 * R * 64);
#pragma unroll
            for (int i = 0; i < R; ++i) { y0[i] += o[(2 * i) * 64 + lane]; y1[i] += o[(2 * i + 1) * 64 + lane]; }
        }
    }
    float2* y2 = reinterpret_cast<float2*>(a.y) + (size_t)(tl.bok ? tl.b : 0) * a.T;
#pragma unroll
    for (int i = 0; i < R; ++i)
        if (tl.owns(i)) {
            if constexpr (NTX) {      // + IQ_match f + f (neuraltx.py:135)
                const float fi = fir.q ? q16::qapply(tl.xi[i], fir.qa) : tl.xi[i], fq = fir.q ? q16::qapply(tl.xq[i], fir.qa) : tl.xq[i];
                const float r0 = __builtin_fmaf(fir.m[0], fi, __builtin_fmaf(fir.m[1], fq, tl.xi[i]));
                const float r1 = __builtin_fmaf(fir.m[2], fi, __builtin_fmaf(fir.m[3], fq, tl.xq[i]));
                y2[tl.t0 + i] = make_float2(y0[i] + r0, y1[i] + r1);
            } else {
                y2[tl.t0 + i] = make_float2(y0[i] + tl.xi[i], y1[i] + tl.xq[i]);     // + residual [I, Q]
            }
        }
}

// back-propagation of one channel from g = dL/d act_4 down to gp0 = dL/d pre_0; GW: accumulate the weight gradients
template <int R, bool GW>
__device__ __forceinline__ void tcnn_chan_bwd(const TcnnChan& k, const float (&pre)[5][R], float (&g)[R], float (&gdw)[4][5]) {
    static_for<4>([&](auto lc) {
        constexpr int l = 3 - decltype(lc)::value;
        float gp[R];
#pragma unroll
        for (int i = 0; i < R; ++i) gp[i] = g[i] * hsg_(pre[l + 1][i]);
        if constexpr (GW) {
            float in[R];
#pragma unroll
            for (int i = 0; i < R; ++i) in[i] = hs_(pre[l][i]);
            conv5_wgrad<R, 1 << l>(gdw[l], gp, in);
        }
        conv5<R, 1 << l, -1>(g, gp, k.dw[l]);
    });
#pragma unroll
    for (int i = 0; i < R; ++i) g[i] *= hsg_(pre[0][i]);
}

// gradient j of the channel: its sum over the row's 16 lanes is deposited on lane j & 15 of the row (A: j < 16, B: j >= 16)
template <int J>
__device__ __forceinline__ void tcnn_deposit(float v, int r, float& depA, float& depB) {
    const float s = row_sum16(v);
    if constexpr (J < 16) depA = r == J ? s : depA;
    else depB = r == J - 16 ? s : depB;
}
// column of gradient j (0..5 w0, 6 b0, 7..26 depthwise taps, 27..28 w5) of channel c in the parameter row
__device__ __forceinline__ int tcnn_col(const TcnnLayout& L, int c, int j) {
    if (j < 6) return L.o_w0 + c * L.F + (j < L.F ? j : 0);        // NeuralTX has 4 input features: columns 4, 5 carry zeros
    if (j == 6) return L.o_b0 + c;
    if (j < 27) return L.o_dw[0] + ((j - 7) / 5) * 5 * L.C + c * 5 + (j - 7) % 5;
    return L.o_w5 + (j - 27) * L.C + c;
}

// grid = rows blocks of 4 waves; wave gw works on tiles gw / ncw, + nwaves / ncw, ... and channels gw % ncw, + ncw, ...
// LDS per wave: accumulator row of P floats + the tile's dy (R float2 per lane, [i][lane]: conflict-free ds_read_b64)
template <int R, bool NTX>
__global__ __launch_bounds__(256, (R <= 8 || (R <= 13 && !NTX)) ? 2 : 1) void tcnn_bwd_kernel(SeqArgs a, TcnnGeom g, int ncw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15;
    const TcnnLayout L = tcnn_layout(a.H, NTX, NTX && a.bits_w > 0);
    NtxFir fir;
    if constexpr (NTX) fir = ntx_fir(a.params, L, a.bits_w, a.bits_a);
    const int Pp = pad4(L.P);
    float* row = smem + wave * Pp;
    float2* dyl = reinterpret_cast<float2*>(smem + 4 * Pp) + wave * (R * 64) + lane;
    for (int i = lane; i < L.P; i += 64) row[i] = 0.0f;
    const int gw = blockIdx.x * 4 + wave, cs = gw % ncw, gstep = gridDim.x * 4 / ncw;
    for (int grp = gw / ncw; grp < g.ngroups; grp += gstep) {
        TcnnTile<R, NTX> tl;
        tl.locate(a, g, grp, true, lane);
        tl.load_x(a, &fir);
        float dfa[NTX ? 4 : 1][R];          // NTX: dL/d[f_I, f_Q, |f|, |f|^3] summed over this wave's channels
        if constexpr (NTX) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < R; ++i) dfa[j][i] = 0.0f;
        }
        {
            const float2* d2 = reinterpret_cast<const float2*>(a.dy) + (size_t)(tl.bok ? tl.b : 0) * a.T;
            wave_lds_fence();
#pragma unroll
            for (int i = 0; i < R; ++i) dyl[i * 64] = tl.owns(i) ? d2[tl.t0 + i] : make_float2(0.0f, 0.0f);
            wave_lds_fence();
        }
        for (int c = cs; c < L.C; c += ncw) {
            const TcnnChan k = tcnn_chan(a.params, L, c);
            float pre[5][R], act[R], gg[R], gdw[4][5], gw5[2] = {0.0f, 0.0f};
            tcnn_chan_fwd<R, true>(tl, k, pre, act);
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int j = 0; j < 5; ++j) gdw[l][j] = 0.0f;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float2 dy = dyl[i * 64];
                gw5[0] = __builtin_fmaf(dy.x, act[i], gw5[0]);
                gw5[1] = __builtin_fmaf(dy.y, act[i], gw5[1]);
                gg[i] = __builtin_fmaf(dy.x, k.w5[0], dy.y * k.w5[1]);
            }
            tcnn_chan_bwd<R, true>(k, pre, gg, gdw);
            float gw0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gb0 = 0.0f;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float gp0 = gg[i];
                gb0 += gp0;
                gw0[0] = __builtin_fmaf(gp0, tl.xi[i], gw0[0]);
                gw0[1] = __builtin_fmaf(gp0, tl.xq[i], gw0[1]);
                gw0[2] = __builtin_fmaf(gp0, tl.am[i], gw0[2]);
                gw0[3] = __builtin_fmaf(gp0, tl.am[i] * tl.am[i] * tl.am[i], gw0[3]);
                if constexpr (!NTX) {
                    gw0[4] = __builtin_fmaf(gp0, tl.xq[i] * tl.ia[i], gw0[4]);
                    gw0[5] = __builtin_fmaf(gp0, tl.xi[i] * tl.ia[i], gw0[5]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) dfa[j][i] = __builtin_fmaf(gp0, k.w0[j], dfa[j][i]);
                }
            }
            // row sums deposited on lanes 0..15 (A) / 0..12 (B) of every row, then the four rows add to the wave's
            // accumulator row one after the other (LDS operations of a wave execute in order: fixed summation order)
            float depA = 0.0f, depB = 0.0f;
            static_for<6>([&](auto jc) { tcnn_deposit<decltype(jc)::value>(gw0[decltype(jc)::value], r, depA, depB); });
            tcnn_deposit<6>(gb0, r, depA, depB);
            static_for<20>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                tcnn_deposit<7 + j>(gdw[j / 5][j % 5], r, depA, depB);
            });
            tcnn_deposit<27>(gw5[0], r, depA, depB);
            tcnn_deposit<28>(gw5[1], r, depA, depB);
            const int colA = tcnn_col(L, c, r), colB = tcnn_col(L, c, 16 + (r < 13 ? r : 12));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((lane >> 4) == q) {
                    atomicAdd(&row[colA], depA);
                    if (r < 13) atomicAdd(&row[colB], depB);
                }
        }
        if constexpr (NTX) {
            // skip paths y = net + IQ_match f + f (one of the channel-split waves adds them), then dL/df through |f| and the FIR
            // tap gradients dL/dcI[k] = sum_t gI[t] xI[t+k-2] + gQ[t] xQ[t+k-2], dL/dcQ[k] = sum_t gQ[t] xI[t+k-2] - gI[t] xQ[t+k-2]
            float gm[4] = {0.f, 0.f, 0.f, 0.f}, gI[R], gQ[R], nI[R];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const float2 dy = cs == 0 ? dyl[i * 64] : make_float2(0.0f, 0.0f);
                const float fi = fir.q ? q16::qapply(tl.xi[i], fir.qa) : tl.xi[i], fq = fir.q ? q16::qapply(tl.xq[i], fir.qa) : tl.xq[i];
                gm[0] = __builtin_fmaf(dy.x, fi, gm[0]); gm[1] = __builtin_fmaf(dy.x, fq, gm[1]);
                gm[2] = __builtin_fmaf(dy.y, fi, gm[2]); gm[3] = __builtin_fmaf(dy.y, fq, gm[3]);
                float d4[4] = {dfa[0][i] + __builtin_fmaf(dy.x, fir.m[0], __builtin_fmaf(dy.y, fir.m[2], dy.x)),
                               dfa[1][i] + __builtin_fmaf(dy.x, fir.m[1], __builtin_fmaf(dy.y, fir.m[3], dy.y)), dfa[2][i], dfa[3][i]};
                if (fir.q) {      // dL/df through IQ_match's activation quantiser: its pass mask on the INT_Linear share only
                    d4[0] = dfa[0][i] + __builtin_fmaf(q16::qpass(tl.xi[i], fir.qa), __builtin_fmaf(dy.x, fir.m[0], dy.y * fir.m[2]), dy.x);
                    d4[1] = dfa[1][i] + __builtin_fmaf(q16::qpass(tl.xq[i], fir.qa), __builtin_fmaf(dy.x, fir.m[1], dy.y * fir.m[3]), dy.y);
                }
                float dI, dQ;
                feat_bwd<FEAT_A4>(tl.xi[i], tl.xq[i], d4, dI, dQ);
                gI[i] = tl.valid[i] ? dI : 0.0f; gQ[i] = tl.valid[i] ? dQ : 0.0f; nI[i] = -gI[i];
            }
            float gci[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, gcq[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            conv5_wgrad<R, 1>(gci, gI, tl.rxi);
            conv5_wgrad<R, 1>(gci, gQ, tl.rxq);
            conv5_wgrad<R, 1>(gcq, gQ, tl.rxi);
            conv5_wgrad<R, 1>(gcq, nI, tl.rxq);
            float depA = 0.0f, depB = 0.0f;
            static_for<5>([&](auto jc) { tcnn_deposit<decltype(jc)::value>(gci[decltype(jc)::value], r, depA, depB); });
            static_for<5>([&](auto jc) { tcnn_deposit<5 + decltype(jc)::value>(gcq[decltype(jc)::value], r, depA, depB); });
            static_for<4>([&](auto jc) { tcnn_deposit<10 + decltype(jc)::value>(gm[decltype(jc)::value] * fir.mm[decltype(jc)::value], r, depA, depB); });
            const int col = r < 10 ? L.o_ci + r : L.o_m + (r < 14 ? r - 10 : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((lane >> 4) == q && r < 14) atomicAdd(&row[col], depA);
        }
    }
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
    for (int i = threadIdx.x; i < L.P + kLossCols; i += blockDim.x)
        prow[i] = i < L.P ? (smem[i] + smem[Pp + i]) + (smem[2 * Pp + i] + smem[3 * Pp + i]) : 0.0f;
}

// grid as the forward; dy is read over the whole tile (halo included); LDS = 4 * 6R * 64 floats when ncw > 1
template <int R, bool NTX>
__global__ __launch_bounds__(256, 1) void tcnn_dx_kernel(SeqArgs a, TcnnGeom g, int ncw) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TcnnLayout L = tcnn_layout(a.H, NTX, NTX && a.bits_w > 0);
    constexpr int F = NTX ? 4 : 6;
    const int gw = blockIdx.x * 4 + wave, grp = gw / ncw, cs = gw % ncw;
    TcnnTile<R, NTX> tl;
    tl.locate(a, g, grp, grp < g.ngroups, lane);
    NtxFir fir;
    if constexpr (NTX) fir = ntx_fir(a.params, L, a.bits_w, a.bits_a);
    tl.load_x(a, &fir);
    float dy0[R], dy1[R], df[F][R];
    {
        const float2* d2 = reinterpret_cast<const float2*>(a.dy) + (size_t)(tl.bok ? tl.b : 0) * a.T;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const float2 v = tl.valid[i] ? d2[tl.t0 + i] : make_float2(0.0f, 0.0f);
            dy0[i] = v.x; dy1[i] = v.y;
#pragma unroll
            for (int j = 0; j < F; ++j) df[j][i] = 0.0f;
        }
    }
    for (int c = cs; c < L.C; c += ncw) {
        const TcnnChan k = tcnn_chan(a.params, L, c);
        float pre[5][R], act[R], gg[R], gdw[4][5];
        tcnn_chan_fwd<R, true>(tl, k, pre, act);
#pragma unroll
        for (int i = 0; i < R; ++i) gg[i] = __builtin_fmaf(dy0[i], k.w5[0], dy1[i] * k.w5[1]);
        tcnn_chan_bwd<R, false>(k, pre, gg, gdw);
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < F; ++j) df[j][i] = __builtin_fmaf(gg[i], k.w0[j], df[j][i]);
    }
    if (ncw > 1) {
        float* mine = smem + wave * (6 * R * 64);
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int j = 0; j < F; ++j) mine[(6 * i + j) * 64 + lane] = df[j][i];
        __syncthreads();
        if (cs != 0) return;
        for (int w = 1; w < ncw; ++w) {
            const float* o = smem + (wave + w) * (6 * R * 64);
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < F; ++j) df[j][i] += o[(6 * i + j) * 64 + lane];
        }
    }
    float2* dx2 = reinterpret_cast<float2*>(a.dx) + (size_t)(tl.bok ? tl.b : 0) * a.T;
    if constexpr (NTX) {
        // dL/df (skip paths + the channel stack through |f|), then the transposed FIR:
        // dxI[s] = sum_k cI[k] gI[s-k+2] + cQ[k] gQ[s-k+2],  dxQ[s] = sum_k cI[k] gQ[s-k+2] - cQ[k] gI[s-k+2]
        float gI[R], gQ[R], nq[5], s0[R], s1[R], s2[R], s3[R];
#pragma unroll
        for (int k = 0; k < 5; ++k) nq[k] = -fir.cq[k];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float d4[4] = {df[0][i] + __builtin_fmaf(dy0[i], fir.m[0], __builtin_fmaf(dy1[i], fir.m[2], dy0[i])),
                           df[1][i] + __builtin_fmaf(dy0[i], fir.m[1], __builtin_fmaf(dy1[i], fir.m[3], dy1[i])), df[2][i], df[3][i]};
            if (fir.q) {
                d4[0] = df[0][i] + __builtin_fmaf(q16::qpass(tl.xi[i], fir.qa), __builtin_fmaf(dy0[i], fir.m[0], dy1[i] * fir.m[2]), dy0[i]);
                d4[1] = df[1][i] + __builtin_fmaf(q16::qpass(tl.xq[i], fir.qa), __builtin_fmaf(dy0[i], fir.m[1], dy1[i] * fir.m[3]), dy1[i]);
            }
            float dI, dQ;
            feat_bwd<FEAT_A4>(tl.xi[i], tl.xq[i], d4, dI, dQ);
            gI[i] = tl.valid[i] ? dI : 0.0f; gQ[i] = tl.valid[i] ? dQ : 0.0f;
        }
        conv5<R, 1, -1>(s0, gI, fir.ci);
        conv5<R, 1, -1>(s1, gQ, fir.cq);
        conv5<R, 1, -1>(s2, gQ, fir.ci);
        conv5<R, 1, -1>(s3, gI, nq);
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (tl.owns(i)) dx2[tl.t0 + i] = make_float2(s0[i] + s1[i], s2[i] + s3[i]);
    } else {
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (tl.owns(i)) {
                float d6[6], dI, dQ;
#pragma unroll
                for (int j = 0; j < 6; ++j) d6[j] = df[j][i];
                feat_bwd<FEAT_DGRU6>(tl.xi[i], tl.xq[i], d6, dI, dQ);
                dx2[tl.t0 + i] = make_float2(dI + dy0[i], dQ + dy1[i]);   // + residual path
            }
    }
}

// ---- host side -------------------------------------------------------------------------------------------------
// channel split: the smallest power of two that puts >= 8 waves on every CU, capped at `cap`
static int tcnn_split(int ngroups, int cap) {
    const int want = 8 * device_cus();
    int ncw = 1;
    while (ncw < cap && ngroups * ncw < want) ncw *= 2;
    return ncw;
}
struct TcnnBwdShape { int ncw, grid; };
static TcnnBwdShape tcnn_bwd_shape(const TcnnGeom& g) {
    TcnnBwdShape s;
    s.ncw = tcnn_split(g.ngroups, 32);
    const long waves = (long)g.ngroups * s.ncw;
    long grid = (waves + 3) / 4;
    const long cap = 2L * device_cus();
    if (grid > cap) grid = cap;
    s.grid = (int)((grid + 7) / 8 * 8);          // 4 * grid is a multiple of every ncw
    return s;
}
template <int R, bool NTX>
static int tcnn_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int mode) {
    const int P = tcnn_layout(m->hidden, NTX, NTX && m->bits_w > 0).P;
    const int fir = NTX ? kNtxFir : 0;
    if (mode == 0) {
        const TcnnGeom g = tcnn_geom(a.B, a.T, kTHalo + fir);
        const int ncw = tcnn_split(g.ngroups, 4);
        hipLaunchKernelGGL((tcnn_fwd_kernel<R, NTX>), dim3((g.ngroups * ncw + 3) / 4), dim3(256), ncw > 1 ? 4 * 2 * R * 64 * sizeof(float) : 0, st, a, g, ncw);
    } else if (mode == 1) {
        const TcnnGeom g = tcnn_geom(a.B, a.T, kTHalo + fir);
        const TcnnBwdShape s = tcnn_bwd_shape(g);
        const size_t lds = ((size_t)4 * pad4(P) + (size_t)4 * 2 * R * 64) * sizeof(float);
        auto k = tcnn_bwd_kernel<R, NTX>;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(s.grid), dim3(256), lds, st, a, g, s.ncw);
    } else {
        const TcnnGeom g = tcnn_geom(a.B, a.T, kTHaloDx + 2 * fir);
        const int ncw = tcnn_split(g.ngroups, 4);
        const size_t lds = ncw > 1 ? (size_t)4 * 6 * R * 64 * sizeof(float) : 0;
        auto k = tcnn_dx_kernel<R, NTX>;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((g.ngroups * ncw + 3) / 4), dim3(256), lds, st, a, g, ncw);
    }
    return (int)hipGetLastError();
}
template <bool NTX>
static int tcnn_dispatch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int mode) {
    switch (tcnn_geom(a.B, a.T, kTHalo).R) {
    case 4: return tcnn_launch<4, NTX>(st, m, a, mode);
    case 8: return tcnn_launch<8, NTX>(st, m, a, mode);
    case 13: return tcnn_launch<13, NTX>(st, m, a, mode);
    default: return tcnn_launch<16, NTX>(st, m, a, mode);
    }
}
static int tcnn_any(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int mode) {
    return m->backbone == ODPD_NEURALTX ? tcnn_dispatch<true>(st, m, a, mode) : tcnn_dispatch<false>(st, m, a, mode);
}

int tcnn_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->hidden > 64) return ODPD_EUNSUPPORTED;
    return tcnn_any(st, m, a, 0);
}
int tcnn_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->hidden > 64) return ODPD_EUNSUPPORTED;
    if (a.partials == nullptr && a.dx == nullptr) return ODPD_EINVAL;
    if (a.partials != nullptr)
        if (int e = tcnn_any(st, m, a, 1)) return e;
    if (a.dx != nullptr) return tcnn_any(st, m, a, 2);
    return 0;
}
int tcnn_rows(const odpd_model_t* m, int B, int T) {
    if (m->hidden > 64) return ODPD_EUNSUPPORTED;
    return tcnn_bwd_shape(tcnn_geom(B, T, kTHalo + (m->backbone == ODPD_NEURALTX ? kNtxFir : 0))).grid;
}

}  // namespace odpd
