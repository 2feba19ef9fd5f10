// lstm_family.hip — persistent-RNN kernels for the nn.LSTM based backbones of the reference:
//   lstm    backbones/lstm.py:4-48     y = fc_out(LSTM(x)), initial (h,c) = (0,0)  (lstm.py:46 passes (h_0,h_0))
//   vdlstm  backbones/vdlstm.py:5-111  input = |x| over a 4-sample window with CIRCULAR left padding
//                                      (vdlstm.py:66-74), y = fc_out(cat(l1*cos, l2*sin)) with
//                                      l1 = fc_lambda_1(h), l2 = fc_lambda_2(h)          (vdlstm.py:77-80)
// nn.LSTM cell, gate order i,f,g,o:  c' = f*c + i*g,  h' = o*tanh(c').
// Same structure as gru_family.hip: LDS rotated-quad tables for W_hh / W_hh^T pulled into registers
// per phase, BPTT by checkpoint (h,c every kCkptStride steps) + block recompute, exact-fp32 MFMA
// weight gradients, one partial-gradient row per workgroup.
#include "odpd_seq.h"
#include "odpd_s16.h"
#include "odpd_lstm.h"
#include "odpd_quant.h"

namespace odpd {

template <int R>
__device__ __forceinline__ void load_rot4(float (&w)[4][R][16], TabPtr tlane, int first_row) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int rb = 0; rb < R; ++rb) load_rot(w[g][rb], tlane + (first_row + g * R + rb) * 4 * 64);
}

// VDLSTM under `--quant`: fc_lambda_1, fc_lambda_2 and fc_out are INT_Linear layers with their own weight / activation / output scales
struct VdQuant { q16::Quant w1, a1, w2, a2, wo, ao, oo; };
__device__ __forceinline__ VdQuant vd_quantisers(const float* pl, const LstmLayout& L, int bits_w, int bits_a) {
    VdQuant q;
    q.w1 = q16::make_quant(pl[L.o_q_l1], bits_w); q.a1 = q16::make_quant(pl[L.o_q_l1 + 1], bits_a);
    q.w2 = q16::make_quant(pl[L.o_q_l2], bits_w); q.a2 = q16::make_quant(pl[L.o_q_l2 + 1], bits_a);
    q.wo = q16::make_quant(pl[L.o_q_out], bits_w); q.ao = q16::make_quant(pl[L.o_q_out + 1], bits_a); q.oo = q16::make_quant(pl[L.o_q_out + 2], 16);
    return q;
}
// the staged copies of the three heads' weights become q_w(W) in place: every later read of them — tables, per-lane columns, the uniform fc_out
// reads — sees the quantised value; the weight quantisers' pass masks are formed from the global copy when the gradients are written.
// (call between stage_params and the barrier that ends fill_lstm_tabs)
__device__ __forceinline__ void vd_quantise_staged(float* pl, const LstmLayout& L, const VdQuant& q) {
    const int H = L.H;      // (three loops: a select between the quantisers inside one loop would put the struct in scratch)
    for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) pl[L.o_w_l1 + i] = q16::qapply(pl[L.o_w_l1 + i], q.w1);
    for (int i = threadIdx.x; i < 4 * H; i += blockDim.x) pl[L.o_w_l2 + i] = q16::qapply(pl[L.o_w_l2 + i], q.w2);
    for (int i = threadIdx.x; i < 16; i += blockDim.x) pl[L.o_w_out + i] = q16::qapply(pl[L.o_w_out + i], q.wo);
}
__device__ __forceinline__ float4 qapply4(float4 v, const q16::Quant& q) {
    return make_float4(q16::qapply(v.x, q), q16::qapply(v.y, q), q16::qapply(v.z, q), q16::qapply(v.w, q));
}
template <int R, bool VD>
struct LstmW {
    float wih[4][VD ? 4 : 2];
    float b[4];               // b_ih + b_hh per gate
    float wout[2], bout[2];   // plain LSTM head
    float wl1[4], wl2[4];     // VD: fc_lambda_{1,2}.weight[k][o]
    q16::Quant qa;            // quantised head (lstm_quantise_head): fc_out's activation quantiser; wout = q_w(weight), woutm = its pass mask
    float woutm[2];
    VdQuant vq;               // VD quantised heads (weights quantised in the staged copy: vd_quantise_staged)
};
// fc_out as INT_Linear (`--quant` on lstm: quant_layers.py:48-85): the lane's head columns become their quantised values
template <int R, bool VD>
__device__ __forceinline__ void lstm_quantise_head(LstmW<R, VD>& w, const float* pl, const LstmLayout& L, int bits_w, int bits_a) {
    const q16::Quant qw = q16::make_quant(pl[L.o_q_out], bits_w);
    w.qa = q16::make_quant(pl[L.o_q_out + 1], bits_a);
#pragma unroll
    for (int c = 0; c < 2; ++c) { w.woutm[c] = q16::qpass(w.wout[c], qw); w.wout[c] = q16::qapply(w.wout[c], qw); }
}
template <int R, bool VD>
__device__ __forceinline__ void load_lstm_w(LstmW<R, VD>& w, const float* pl, const LstmLayout& L, int row, int col) {
    constexpr int F = VD ? 4 : 2;
    const int H = L.H, o = 16 * row + col;
    const bool vo = o < H;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < F; ++i) w.wih[g][i] = vo ? pl[L.o_w_ih + (g * H + o) * F + i] : 0.0f;
        w.b[g] = vo ? pl[L.o_b_ih + g * H + o] + pl[L.o_b_hh + g * H + o] : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        w.wout[c] = (!VD && vo) ? pl[L.o_w_out + c * H + o] : 0.0f;
        w.bout[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pl[L.o_b_out + c])));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w.wl1[k] = (VD && vo) ? pl[L.o_w_l1 + k * H + o] : 0.0f;
        w.wl2[k] = (VD && vo) ? pl[L.o_w_l2 + k * H + o] : 0.0f;
    }
}
__device__ __forceinline__ float uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// window of the VDLSTM input at one step
struct VdWin { float a[4], cw[4], sw[4]; };
__device__ __forceinline__ void vd_elem(float2 xv, float& a, float& cw, float& sw) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y);
    a = __builtin_amdgcn_sqrtf(a2);
    const float ia = fast_rcp(a);
    cw = xv.x * ia; sw = xv.y * ia;
}

template <int R, bool VD>
__device__ __forceinline__ void lstm_cell_fwd(const LstmW<R, VD>& w, const float (&whh)[4][R][16],
                                              const float (&xin)[VD ? 4 : 2], float& h, float& c, float& gi, float& gf,
                                              float& gg, float& go, float& tc) {
    constexpr int F = VD ? 4 : 2;
    float a0 = w.b[0], a1 = w.b[1], a2 = w.b[2], a3 = w.b[3];
#pragma unroll
    for (int i = 0; i < F; ++i) {
        a0 = __builtin_fmaf(w.wih[0][i], xin[i], a0);
        a1 = __builtin_fmaf(w.wih[1][i], xin[i], a1);
        a2 = __builtin_fmaf(w.wih[2][i], xin[i], a2);
        a3 = __builtin_fmaf(w.wih[3][i], xin[i], a3);
    }
    float z = 0.0f;
    rotdot3(a0, a1, a2, whh[0][0], whh[1][0], whh[2][0], h);
    a3 = rotdot(a3, whh[3][0], h);
    if constexpr (R == 2) {
        const float hx = swap16(h);
        rotdot3(a0, a1, a2, whh[0][1], whh[1][1], whh[2][1], hx);
        a3 = rotdot(a3, whh[3][1], hx);
    }
    (void)z;
    gi = sigmoidf_(a0); gf = sigmoidf_(a1); gg = tanhf_(a2); go = sigmoidf_(a3);
    c = __builtin_fmaf(gf, c, gi * gg);
    tc = tanhf_(c);
    h = go * tc;
}

// output head.  VD: y_c = sum_k Wo[c][k] cw[k] l1[k] + Wo[c][4+k] sw[k] l2[k] + bo[c], l = fc_lambda(h)
template <int R, bool VD>
__device__ __forceinline__ void lstm_head(const LstmW<R, VD>& w, const float* pl, const LstmLayout& L, float h,
                                          const VdWin& win, float& y0, float& y1) {
    if constexpr (VD) {
        float q0 = 0.f, q1 = 0.f, c0 = w.bout[0], c1 = w.bout[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float u0 = pl[L.o_w_out + k] * win.cw[k], v0 = pl[L.o_w_out + 4 + k] * win.sw[k];
            const float u1 = pl[L.o_w_out + 8 + k] * win.cw[k], v1 = pl[L.o_w_out + 12 + k] * win.sw[k];
            q0 = __builtin_fmaf(u0, w.wl1[k], __builtin_fmaf(v0, w.wl2[k], q0));
            q1 = __builtin_fmaf(u1, w.wl1[k], __builtin_fmaf(v1, w.wl2[k], q1));
            c0 = __builtin_fmaf(u0, pl[L.o_b_l1 + k], __builtin_fmaf(v0, pl[L.o_b_l2 + k], c0));
            c1 = __builtin_fmaf(u1, pl[L.o_b_l1 + k], __builtin_fmaf(v1, pl[L.o_b_l2 + k], c1));
        }
        y0 = seq_sum<R>(q0 * h) + c0;
        y1 = seq_sum<R>(q1 * h) + c1;
    } else {
        y0 = seq_sum<R>(w.wout[0] * h) + w.bout[0];
        y1 = seq_sum<R>(w.wout[1] * h) + w.bout[1];
    }
}

// ---- staging with a 3-sample circular left halo (VDLSTM windows) --------------------------------
constexpr int kHalo = 3;
constexpr int kHaloStride = kChunk + kHalo + 2;   // float2 per sequence row (odd/2 -> conflict-free pairs)
template <int SPW>
__device__ __forceinline__ void stage_in_halo(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int PER = kChunk + kHalo, TOT = SPW * PER, N = (TOT + 63) / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        if (e < TOT) {
            const int m = e / PER, pos = e % PER;
            int tg = t0 - kHalo + pos;
            if (tg < 0) tg += T;     // circular padding: the frame's own last samples (vdlstm.py:66-74)
            float2 v = make_float2(0.5f, 0.5f);
            // tg can stay negative for frames shorter than the halo (plain LSTM with T < 3: the halo is not used there)
            if (pos < len + kHalo && b0 + m < B && tg >= 0 && tg < T) v = g2[(size_t)(b0 + m) * T + tg];
            lds[m * kHaloStride + pos] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// forward kernel
// -------------------------------------------------------------------------------------------------
template <int R, bool VD>
__global__ __launch_bounds__(kMaxThreads) void lstm_fwd_kernel(SeqArgs a) {
    constexpr int F = VD ? 4 : 2, SPW = 4 / R, LPS = 16 * R, S = kCkptStride;
    using T = LstmTabs<R>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<R>();
    const int lane = id.lane, s = id.s;
    const LstmLayout L = lstm_layout(a.H, VD);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_lstm_tabs<R, false>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + T::kFloats) + id.wave * (SPW * kHaloStride + SPW * kChunkPad);
    float2* ys = xs + SPW * kHaloStride;
    LstmW<R, VD> w;
    load_lstm_w<R, VD>(w, pl, L, id.row, id.col);
    float whh[4][R][16];
    load_rot4<R>(whh, tlane, T::kHH);
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float h = 0.0f, c = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            stage_in_halo<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
            const float2* xr = xs + s * kHaloStride + kHalo;   // xr[tt] = x[t0 + tt], xr[-1..-3] = halo
            VdWin win;
            if constexpr (VD) {
#pragma unroll
                for (int k = 0; k < 3; ++k) vd_elem(xr[k - 3], win.a[k + 1], win.cw[k + 1], win.sw[k + 1]);
            }
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xr[tt];
                float xin[F];
                if constexpr (VD) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { win.a[k] = win.a[k + 1]; win.cw[k] = win.cw[k + 1]; win.sw[k] = win.sw[k + 1]; }
                    vd_elem(xv, win.a[3], win.cw[3], win.sw[3]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) xin[k] = win.a[k];
                } else { xin[0] = xv.x; xin[1] = xv.y; }
                float gi, gf, gg, go, tc, y0, y1;
                lstm_cell_fwd<R, VD>(w, whh, xin, h, c, gi, gf, gg, go, tc);
                lstm_head<R, VD>(w, pl, L, h, win, y0, y1);
                if ((lane & (LPS - 1)) == 0) ys[s * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (a.ckpt != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    float* ck = a.ckpt + ((size_t)grp * a.nck + t1 / S) * 128;
                    ck[lane] = h; ck[64 + lane] = c;
                }
            }
            wave_lds_fence();
            stage_out<SPW>(ys, a.y, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
        }
    }
}

// -------------------------------------------------------------------------------------------------
// evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): gate-parallel, ONE sequence per wave.
// The four 16-lane rows hold the same (h, c); row k computes gate k (i, f, g, o) of every unit with its own rotated W_hh rows — one
// rotated dot product per step instead of four (hidden 17..32: four instead of sixteen) — applies the gate's activation, and three
// cross-row swaps hand all four gates to every row, which then updates (c, h) redundantly.  What does not depend on the state
// leaves the step loop: the inputs of a 64-step chunk (I, Q, or |x| and its unit phasor for the VDLSTM window) are computed with
// lane = time step and parked in LDS, a step parks h, and the chunk's 64 outputs (fc_out; VD: fc_lambda_1/2, the phasor products,
// fc_out) are formed afterwards, one time step per lane.  The next chunk's samples are in flight while the current one is stepped.
// -------------------------------------------------------------------------------------------------
template <int NB> struct LstmEvalLds {
    static constexpr int kHistStride = 64 * NB + 4;
    static constexpr int kHeadFloats = 8 * 16 * NB;          // plain: fc_out rows 0..1; VD: fc_lambda_1 rows 0..3, fc_lambda_2 rows 4..7 (zero padded)
    static constexpr int kFloats = (kEvalChunk + kHalo) * 4 + kEvalChunk * kHistStride + kHeadFloats;
};
// QH (plain lstm): fc_out is an INT_Linear (`--quant`, quant_layers.py:48-85) — the head table holds the weights' grid indices, the chunk's h
// values are quantised with lane = time step, the integer sum takes its scale s_a s_w and the fp32 bias in one FMA; eval mode adds the
// 16-bit output quantiser.
template <int NB, bool VD, bool CK, bool QH = false>      // CK: also writes the BPTT checkpoints (the forward of the split train path)
__global__ __launch_bounds__(64) void lstm_eval_kernel(SeqArgs a) {
    constexpr int F = VD ? 4 : 2, EC = kEvalChunk, HS = LstmEvalLds<NB>::kHistStride;
    using T = LstmTabs<NB>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;      // gate i | f | g | o
    const LstmLayout L = lstm_layout(a.H, VD, QH);
    const int H = L.H;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    VdQuant vq{};
    if constexpr (QH && VD) { vq = vd_quantisers(pl, L, a.bits_w, a.bits_a); vd_quantise_staged(pl, L, vq); }
    float* tab = smem + pad4(L.P);
    fill_lstm_tabs<NB, false>(tab, pl, L, lane, 0, 1);
    float* ftab = tab + T::kFloats;                    // [kHalo + EC][4]: entry kHalo + i = inputs of time t0 + i; entries 0..2 = the three before
    float* hist = ftab + (kHalo + EC) * 4;             // [EC][HS]: entry i = h of time t0 + i, every lane's copy
    float* hw = hist + EC * HS;
    q16::Quant qw{}, qa{}, qo{};
    if constexpr (QH && !VD) {
        qw = q16::make_quant(pl[L.o_q_out], a.bits_w); qa = q16::make_quant(pl[L.o_q_out + 1], a.bits_a); qo = q16::make_quant(pl[L.o_q_out + 2], 16);
    }
    for (int i = lane; i < LstmEvalLds<NB>::kHeadFloats; i += 64) {
        const int r = i / (16 * NB), u = i % (16 * NB);
        float v = 0.0f;
        if (u < H) {
            if (VD) v = r < 4 ? pl[L.o_w_l1 + r * H + u] : pl[L.o_w_l2 + (r - 4) * H + u];
            else if (r < 2) v = QH ? q16::qgrid(pl[L.o_w_out + r * H + u], qw) : pl[L.o_w_out + r * H + u];
        }
        hw[i] = v;
    }
    wave_lds_fence();
    float win[NB][F], bg[NB], wrec[NB][NB][16];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
        const int o = 16 * ob + col;
        const bool vo = o < H;
#pragma unroll
        for (int i = 0; i < F; ++i) win[ob][i] = vo ? pl[L.o_w_ih + (role * H + o) * F + i] : 0.0f;
        bg[ob] = vo ? pl[L.o_b_ih + role * H + o] + pl[L.o_b_hh + role * H + o] : 0.0f;
        TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + 16 * ob + col);
#pragma unroll
        for (int rb = 0; rb < NB; ++rb) load_rot(wrec[ob][(ob + rb) % NB], tl + (T::kHH + role * NB + rb) * 4 * 64);
    }
    const bool is_g = role == 2;
    const float4* hw4 = reinterpret_cast<const float4*>(hw);

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        float h[NB], cs[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) { h[kb] = 0.0f; cs[kb] = 0.0f; }
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * a.T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * a.T;
        auto park = [&](int entry, float2 xv) {           // inputs of one time step
            float4 e = make_float4(xv.x, xv.y, 0.0f, 0.0f);
            if constexpr (VD) vd_elem(xv, e.x, e.y, e.z);
            reinterpret_cast<float4*>(ftab)[entry] = e;
        };
        float2 raw = lane < a.T ? xg[lane] : make_float2(0.5f, 0.5f);
        wave_lds_fence();
        if (VD && lane < kHalo) park(lane, xg[a.T - kHalo + lane]);      // circular left padding: the frame's own last samples (vdlstm.py:66-74)
        for (int t0 = 0; t0 < a.T; t0 += EC) {
            const int len = min(EC, a.T - t0);
            wave_lds_fence();
            park(kHalo + lane, raw);
            wave_lds_fence();
            raw = t0 + EC + lane < a.T ? xg[t0 + EC + lane] : make_float2(0.5f, 0.5f);
            for (int tt = 0; tt < len; ++tt) {
                float xin[F];
                if constexpr (VD) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xin[k] = ftab[(tt + k) * 4];          // |x| of times t - 3 .. t
                } else {
                    const float2 xv = *reinterpret_cast<const float2*>(ftab + (kHalo + tt) * 4);
                    xin[0] = xv.x; xin[1] = xv.y;
                }
                float pre[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    float acc = bg[ob];
#pragma unroll
                    for (int i = 0; i < F; ++i) acc = __builtin_fmaf(win[ob][i], xin[i], acc);
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) acc = rotdot(acc, wrec[ob][kb], h[kb]);
                    pre[ob] = acc;
                }
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    const float sg = sigmoidf_(pre[ob]), th = tanhf_(pre[ob]);
                    float g[4];
                    gather_rows(is_g ? th : sg, g);
                    cs[ob] = __builtin_fmaf(g[1], cs[ob], g[0] * g[2]);
                    h[ob] = g[3] * tanhf_(cs[ob]);
                    hist[tt * HS + 64 * ob + lane] = h[ob];
                }
                // BPTT checkpoints in the layout of the row-rotated backward (4 / NB sequences per wave-task, lane = 16 NB s + 16 ob + col)
                if constexpr (CK) {
                    const int t1 = t0 + tt + 1;
                    if ((t1 % kCkptStride) == 0 && t1 < a.T && role == 0) {
                        float* ck = a.ckpt + ((size_t)(b / (4 / NB)) * a.nck + t1 / kCkptStride) * 128 + 16 * NB * (b % (4 / NB)) + col;
#pragma unroll
                        for (int ob = 0; ob < NB; ++ob) { ck[16 * ob] = h[ob]; ck[64 + 16 * ob] = cs[ob]; }
                    }
                }
            }
            wave_lds_fence();
            // the chunk's outputs, lane = time step
            if (lane < len) {
                const float4* hv4 = reinterpret_cast<const float4*>(hist + lane * HS);             // row 0's copy
                float acc[VD ? 8 : 2];
#pragma unroll
                for (int r = 0; r < (VD ? 8 : 2); ++r) acc[r] = QH ? 0.0f : VD ? (r < 4 ? pl[L.o_b_l1 + r] : pl[L.o_b_l2 + r - 4]) : pl[L.o_b_out + r];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 hv = hv4[16 * ob + q], hv2 = hv;
                        if constexpr (QH && !VD) hv = make_float4(q16::qgrid(hv.x, qa), q16::qgrid(hv.y, qa), q16::qgrid(hv.z, qa), q16::qgrid(hv.w, qa));
                        if constexpr (QH && VD) { hv = qapply4(hv2, vq.a1); hv2 = qapply4(hv2, vq.a2); }      // fc_lambda_1 / _2: their own activation grids
#pragma unroll
                        for (int r = 0; r < (VD ? 8 : 2); ++r) {
                            const float4 w = hw4[4 * NB * r + 4 * ob + q];
                            const float4 hx = (QH && VD && r >= 4) ? hv2 : hv;
                            acc[r] = __builtin_fmaf(w.x, hx.x, acc[r]); acc[r] = __builtin_fmaf(w.y, hx.y, acc[r]);
                            acc[r] = __builtin_fmaf(w.z, hx.z, acc[r]); acc[r] = __builtin_fmaf(w.w, hx.w, acc[r]);
                        }
                    }
                float y0, y1;
                if constexpr (VD) {
                    // y = fc_out(cat(l1 * cos, l2 * sin)) over the four-sample window (vdlstm.py:77-80)
                    // QH: every sum is an exact grid sum (started at 0), its fp32 bias added once at the end
                    if constexpr (QH) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) acc[r] += r < 4 ? pl[L.o_b_l1 + r] : pl[L.o_b_l2 + r - 4];
                    }
                    y0 = QH ? 0.0f : pl[L.o_b_out]; y1 = QH ? 0.0f : pl[L.o_b_out + 1];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 e = reinterpret_cast<const float4*>(ftab)[lane + k];
                        float lc = acc[k] * e.y, ls = acc[4 + k] * e.z;
                        if constexpr (QH) { lc = q16::qapply(lc, vq.ao); ls = q16::qapply(ls, vq.ao); }
                        y0 = __builtin_fmaf(pl[L.o_w_out + k], lc, __builtin_fmaf(pl[L.o_w_out + 4 + k], ls, y0));
                        y1 = __builtin_fmaf(pl[L.o_w_out + 8 + k], lc, __builtin_fmaf(pl[L.o_w_out + 12 + k], ls, y1));
                    }
                    if constexpr (QH) {
                        y0 += pl[L.o_b_out]; y1 += pl[L.o_b_out + 1];
                        if (a.eval_out) { y0 = q16::qapply(y0, vq.oo); y1 = q16::qapply(y1, vq.oo); }
                    }
                } else if constexpr (QH) {
                    const float S = qa.s * qw.s;
                    y0 = __builtin_fmaf(acc[0], S, pl[L.o_b_out]); y1 = __builtin_fmaf(acc[1], S, pl[L.o_b_out + 1]);
                    if (a.eval_out) { y0 = q16::qapply(y0, qo); y1 = q16::qapply(y1, qo); }
                } else { y0 = acc[0]; y1 = acc[1]; }
                yg[t0 + lane] = make_float2(y0, y1);
            }
            // the chunk's last three inputs become the next one's halo
            float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < kHalo) carry = reinterpret_cast<const float4*>(ftab)[len + lane];
            wave_lds_fence();
            if (lane < kHalo) reinterpret_cast<float4*>(ftab)[lane] = carry;
        }
        wave_lds_fence();
    }
}

// -------------------------------------------------------------------------------------------------
// Gate-parallel fused train kernel for the reference's own batch sizes (train_funcs.py:28-48; a wave is alone on its SIMD there and the
// T-serial chain is the whole cost): ONE sequence per wave (one wave per workgroup), hidden <= 16, only the recurrence in the step loops.
//   forward   as lstm_eval_kernel (row k = gate k); h(t), c(t), tanh c(t) — and, PG, the four gates — of every step are parked in LDS;
//   head      outputs, loss, dL/dy (VD: dL/d lambda_1/2, the fc_out gradient) of all T steps with lane = time step;
//   backward  per step ONE rotated dot product with the transposed weights: every row multiplies its own gate's pre-activation gradient,
//             the cross-row sum is dL/dh(t-1); the weight gradients of a step are TWO 4-block MFMAs (v_mfma_f32_16x16x1_4b_f32, block k =
//             gate k): (d_i | d_f | d_g | d_o) x h(t-1) and x (inputs | 1).  !PG recomputes the gates from the parked h(t-1).
// One partial-gradient row per workgroup.
// -------------------------------------------------------------------------------------------------
template <bool VD>
__host__ __device__ inline int lstm_gp_buffer_floats(int T, bool pg) {
    const int Tp = (T + 63) & ~63;
    const int buf = (Tp + 4) * 4 + 3 * (Tp + 2) * 16 + Tp * 2 + (VD ? Tp * 8 : 0) + 256 + 8 * 16 + (pg ? Tp * 64 : 0);
    const int tabf = LstmTabs<1>::kFloats;
    return buf > tabf ? buf : tabf;
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <bool VD, bool PG, bool QH = false>      // QH: fc_out as INT_Linear (see lstm_eval_kernel); train mode
__global__ __launch_bounds__(64) void lstm_gp_train_kernel(SeqArgs a) {
    constexpr int F = VD ? 4 : 2, NH = VD ? 8 : 2;
    using TB = LstmTabs<1>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;      // gate i | f | g | o
    const LstmLayout L = lstm_layout(a.H, VD, QH);
    const int H = L.H, T = a.T, Tp = (T + 63) & ~63;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    VdQuant vq{};
    if constexpr (QH && VD) { vq = vd_quantisers(pl, L, a.bits_w, a.bits_a); vd_quantise_staged(pl, L, vq); }
    float* tab = smem + pad4(L.P);
    fill_lstm_tabs<1, true>(tab, pl, L, lane, 0, 1);
    const bool vo = col < H, is_g = role == 2;
    float wF[16], wT[16];
    {
        TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + col);
        load_rot(wF, tl + (TB::kHH + role) * 4 * 64);
        load_rot(wT, tl + (TB::kHHT + role) * 4 * 64);
    }
    float win[F];
#pragma unroll
    for (int i = 0; i < F; ++i) win[i] = vo ? pl[L.o_w_ih + (role * H + col) * F + i] : 0.0f;
    const float bg = vo ? pl[L.o_b_ih + role * H + col] + pl[L.o_b_hh + role * H + col] : 0.0f;
    // head operands of a unit: plain fc_out columns, or the fc_lambda_1/2 columns of the VDLSTM
    float wh[NH];
#pragma unroll
    for (int r = 0; r < NH; ++r)
        wh[r] = !vo ? 0.0f : VD ? (r < 4 ? pl[L.o_w_l1 + r * H + col] : pl[L.o_w_l2 + (r - 4) * H + col]) : pl[L.o_w_out + r * H + col];
    q16::Quant qw{}, qa{};
    [[maybe_unused]] float whm[NH];                  // QH: the weight quantiser's pass mask of the unit's head columns
    if constexpr (QH && !VD) {
        qw = q16::make_quant(pl[L.o_q_out], a.bits_w); qa = q16::make_quant(pl[L.o_q_out + 1], a.bits_a);
#pragma unroll
        for (int r = 0; r < NH; ++r) { whm[r] = q16::qpass(wh[r], qw); wh[r] = q16::qapply(wh[r], qw); }
    }
    if constexpr (QH && VD) {      // (wh: already the quantised values, vd_quantise_staged) the masks from the global copy
#pragma unroll
        for (int r = 0; r < NH; ++r)
            whm[r] = vo ? q16::qpass(a.params[(r < 4 ? L.o_w_l1 + r * H : L.o_w_l2 + (r - 4) * H) + col], r < 4 ? vq.w1 : vq.w2) : 0.0f;
    }
    wave_lds_fence();
    // per-time buffers over the tables
    float* ftab = tab;                                  // [Tp + 4][4]   entry 3 + t = inputs of step t (I, Q | |x|, cos, sin); entries 0..2 = the circular halo
    float* hist = ftab + (Tp + 4) * 4;                  // [Tp + 2][16]  entry t + 1 = h(t), entry 0 = 0
    float* cpk = hist + (Tp + 2) * 16;                  // [Tp + 2][16]  entry t + 1 = c(t), entry 0 = 0
    float* tpk = cpk + (Tp + 2) * 16;                   // [Tp + 2][16]  entry t + 1 = tanh c(t)
    float* dyb = tpk + (Tp + 2) * 16;                   // [Tp][2]       dL/dy(t)
    float* dlb = dyb + Tp * 2;                          // VD: [Tp][8]   dL/d lambda_1[0..3], lambda_2[0..3] of step t
    float* dump = dlb + (VD ? Tp * 8 : 0);              // [256]
    float* hw = dump + 256;                             // head rows [8][16], zero padded (as lstm_eval_kernel)
    float* gpk = hw + 8 * 16;                           // PG: [Tp][16][4]  i, f, g, o of step t
    for (int i = lane; i < 8 * 16; i += 64) {
        const int r = i >> 4, u = i & 15;
        float v = 0.0f;
        if (u < H) {
            if (VD) v = r < 4 ? pl[L.o_w_l1 + r * H + u] : pl[L.o_w_l2 + (r - 4) * H + u];
            else if (r < 2) v = QH ? q16::qgrid(pl[L.o_w_out + r * H + u], qw) : pl[L.o_w_out + r * H + u];
        }
        hw[i] = v;
    }
    if (lane < 16) { hist[lane] = 0.0f; cpk[lane] = 0.0f; tpk[lane] = 0.0f; }
    const float4* hw4 = reinterpret_cast<const float4*>(hw);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    // the per-step stores of the forward pass: row 1 parks h, row 2 c, row 3 tanh c (row 0 hits the dump); PG: row 0 parks the gates
    const int park0 = role == 0 ? (int)(dump - smem) + lane : (int)((role == 1 ? hist : role == 2 ? cpk : tpk) - smem) + 16 + col;
    const int park_step = role == 0 ? 0 : 16;
    const int gpark0 = role == 0 ? (int)(gpk - smem) + 4 * col : (int)(dump - smem) + 4 * lane, gpark_step = role == 0 ? 64 : 0;
    const int xcol = VD ? 4 * col : 4 * 3 + col;                               // input `col` of step t: ftab[4 t + xcol]

    f32x16 acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
    float dwh[NH], tacc[VD ? 26 : 2], loss_acc = 0.0f;     // per unit: head-row gradients; per time lane: fc_out (VD) and head bias gradients
#pragma unroll
    for (int r = 0; r < NH; ++r) dwh[r] = 0.0f;
#pragma unroll
    for (int r = 0; r < (VD ? 26 : 2); ++r) tacc[r] = 0.0f;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        auto park_in = [&](int entry, float2 xv) {
            float4 e = make_float4(xv.x, xv.y, 0.0f, 0.0f);
            if constexpr (VD) vd_elem(xv, e.x, e.y, e.z);
            reinterpret_cast<float4*>(ftab)[entry] = e;
        };
        // ---- forward ----
        {
            float h = 0.0f, c = 0.0f;
            int park = park0, gpark = gpark0;
            float2 raw = lane < T ? xg[lane] : make_float2(0.5f, 0.5f);
            wave_lds_fence();
            if (VD && lane < kHalo) park_in(lane, xg[T - kHalo + lane]);         // circular left padding (vdlstm.py:66-74)
            for (int t0 = 0; t0 < T; t0 += kEvalChunk) {
                const int len = min(kEvalChunk, T - t0);
                wave_lds_fence();
                park_in(kHalo + t0 + lane, raw);
                wave_lds_fence();
                raw = t0 + kEvalChunk + lane < T ? xg[t0 + kEvalChunk + lane] : make_float2(0.5f, 0.5f);
                for (int tt = 0; tt < len; ++tt) {
                    const int t = t0 + tt;
                    float xin[F];
                    if constexpr (VD) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) xin[k] = ftab[(t + k) * 4];
                    } else {
                        const float2 xv = *reinterpret_cast<const float2*>(ftab + (kHalo + t) * 4);
                        xin[0] = xv.x; xin[1] = xv.y;
                    }
                    float acc = bg;
#pragma unroll
                    for (int i = 0; i < F; ++i) acc = __builtin_fmaf(win[i], xin[i], acc);
                    acc = rotdot(acc, wF, h);
                    const float sg = sigmoidf_(acc), th = tanhf_(acc);
                    float g[4];
                    gather_rows(is_g ? th : sg, g);
                    c = __builtin_fmaf(g[1], c, g[0] * g[2]);
                    const float tc = tanhf_(c);
                    h = g[3] * tc;
                    smem[park] = role == 2 ? c : role == 3 ? tc : h;
                    park += park_step;
                    if constexpr (PG) {
                        *reinterpret_cast<float4*>(smem + gpark) = make_float4(g[0], g[1], g[2], g[3]);
                        gpark += gpark_step;
                    }
                }
            }
            wave_lds_fence();
        }
        // ---- outputs, loss and the head's gradients of every step, lane = time step ----
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
                const float4* hv4 = reinterpret_cast<const float4*>(hist + (t + 1) * 16);
                float acc[NH];
#pragma unroll
                for (int r = 0; r < NH; ++r) acc[r] = QH ? 0.0f : VD ? (r < 4 ? pl[L.o_b_l1 + r] : pl[L.o_b_l2 + r - 4]) : pl[L.o_b_out + r];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 hv = hv4[q], hv2 = hv;
                    if constexpr (QH && !VD) hv = make_float4(q16::qgrid(hv.x, qa), q16::qgrid(hv.y, qa), q16::qgrid(hv.z, qa), q16::qgrid(hv.w, qa));
                    if constexpr (QH && VD) { hv = qapply4(hv2, vq.a1); hv2 = qapply4(hv2, vq.a2); }
#pragma unroll
                    for (int r = 0; r < NH; ++r) {
                        const float4 w = hw4[4 * r + q];
                        const float4 hx = (QH && VD && r >= 4) ? hv2 : hv;
                        acc[r] = __builtin_fmaf(w.x, hx.x, acc[r]); acc[r] = __builtin_fmaf(w.y, hx.y, acc[r]);
                        acc[r] = __builtin_fmaf(w.z, hx.z, acc[r]); acc[r] = __builtin_fmaf(w.w, hx.w, acc[r]);
                    }
                }
                const float2 tv = tg[t];
                float dy0, dy1;
                if constexpr (VD) {
                    // y = fc_out(cat(l1 * cos, l2 * sin)) over the four-sample window (vdlstm.py:77-80)
                    // QH: the three heads as INT_Linear — exact grid sums started at 0, the fp32 bias added once; fc_out's inputs (l cos, l sin)
                    // quantised on its activation grid, whose pass mask gates their gradients
                    float y0 = QH ? 0.0f : pl[L.o_b_out], y1 = QH ? 0.0f : pl[L.o_b_out + 1], lc[4], ls[4], cw[4], sw[4];
                    [[maybe_unused]] float pzc[4], pzs[4];
                    if constexpr (QH) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) acc[r] += r < 4 ? pl[L.o_b_l1 + r] : pl[L.o_b_l2 + r - 4];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float4 e = reinterpret_cast<const float4*>(ftab)[t + k];
                        cw[k] = e.y; sw[k] = e.z;
                        lc[k] = acc[k] * e.y; ls[k] = acc[4 + k] * e.z;
                        if constexpr (QH) {
                            pzc[k] = q16::qpass(lc[k], vq.ao); pzs[k] = q16::qpass(ls[k], vq.ao);
                            lc[k] = q16::qapply(lc[k], vq.ao); ls[k] = q16::qapply(ls[k], vq.ao);
                        }
                        y0 = __builtin_fmaf(pl[L.o_w_out + k], lc[k], __builtin_fmaf(pl[L.o_w_out + 4 + k], ls[k], y0));
                        y1 = __builtin_fmaf(pl[L.o_w_out + 8 + k], lc[k], __builtin_fmaf(pl[L.o_w_out + 12 + k], ls[k], y1));
                    }
                    if constexpr (QH) { y0 += pl[L.o_b_out]; y1 += pl[L.o_b_out + 1]; }
                    s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
                    float dl[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        dl[k] = __builtin_fmaf(dy0, pl[L.o_w_out + k], dy1 * pl[L.o_w_out + 8 + k]) * cw[k];
                        dl[4 + k] = __builtin_fmaf(dy0, pl[L.o_w_out + 4 + k], dy1 * pl[L.o_w_out + 12 + k]) * sw[k];
                        if constexpr (QH) { dl[k] *= pzc[k]; dl[4 + k] *= pzs[k]; }
                        tacc[k] = __builtin_fmaf(dy0, lc[k], tacc[k]); tacc[4 + k] = __builtin_fmaf(dy0, ls[k], tacc[4 + k]);      // fc_out row 0
                        tacc[8 + k] = __builtin_fmaf(dy1, lc[k], tacc[8 + k]); tacc[12 + k] = __builtin_fmaf(dy1, ls[k], tacc[12 + k]);
                        tacc[18 + k] += dl[k]; tacc[22 + k] += dl[4 + k];                                                           // fc_lambda biases
                    }
                    tacc[16] += dy0; tacc[17] += dy1;
                    reinterpret_cast<float4*>(dlb)[2 * t] = make_float4(dl[0], dl[1], dl[2], dl[3]);
                    reinterpret_cast<float4*>(dlb)[2 * t + 1] = make_float4(dl[4], dl[5], dl[6], dl[7]);
                } else {
                    if constexpr (QH) {      // integer sums: scale s_a s_w and the fp32 bias in one FMA
                        const float S = qa.s * qw.s;
                        acc[0] = __builtin_fmaf(acc[0], S, pl[L.o_b_out]); acc[1] = __builtin_fmaf(acc[1], S, pl[L.o_b_out + 1]);
                    }
                    s16_loss(lossc, acc[0] - tv.x, acc[1] - tv.y, dy0, dy1, loss_acc);
                    tacc[0] += dy0; tacc[1] += dy1;
                }
                *reinterpret_cast<float2*>(dyb + 2 * t) = make_float2(dy0, dy1);
            }
        }
        wave_lds_fence();
        // ---- backward ----
        {
            float dh = 0.0f, dc = 0.0f;
            for (int t = T - 1; t >= 0; --t) {
                const float hp = hist[t * 16 + col], ht = hist[(t + 1) * 16 + col];
                const float cp = cpk[t * 16 + col], tc = tpk[(t + 1) * 16 + col];
                float hg[NH];                                                       // dL/d(head rows) of step t: dy, or d lambda_1/2
                if constexpr (VD) {
                    const float4 d0 = reinterpret_cast<const float4*>(dlb)[2 * t], d1 = reinterpret_cast<const float4*>(dlb)[2 * t + 1];
                    hg[0] = d0.x; hg[1] = d0.y; hg[2] = d0.z; hg[3] = d0.w; hg[4] = d1.x; hg[5] = d1.y; hg[6] = d1.z; hg[7] = d1.w;
                } else {
                    const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
                    hg[0] = dyv.x; hg[1] = dyv.y;
                }
                float g[4];
                if constexpr (PG) {
                    const float4 gv = reinterpret_cast<const float4*>(gpk)[t * 16 + col];
                    g[0] = gv.x; g[1] = gv.y; g[2] = gv.z; g[3] = gv.w;
                } else {
                    float xin[F];
                    if constexpr (VD) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) xin[k] = ftab[(t + k) * 4];
                    } else {
                        const float2 xv = *reinterpret_cast<const float2*>(ftab + (kHalo + t) * 4);
                        xin[0] = xv.x; xin[1] = xv.y;
                    }
                    float acc = bg;
#pragma unroll
                    for (int i = 0; i < F; ++i) acc = __builtin_fmaf(win[i], xin[i], acc);
                    acc = rotdot(acc, wF, hp);
                    const float sg = sigmoidf_(acc), th = tanhf_(acc);
                    gather_rows(is_g ? th : sg, g);
                }
                float dht = dh;
                if constexpr (QH && VD) {      // fc_lambda_1 / _2: each on its own activation grid
                    const float hq1 = q16::qapply(ht, vq.a1), hq2 = q16::qapply(ht, vq.a2);
                    float hd1 = 0.0f, hd2 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        hd1 = __builtin_fmaf(hg[r], wh[r], hd1); dwh[r] = __builtin_fmaf(hg[r], hq1, dwh[r]);
                        hd2 = __builtin_fmaf(hg[4 + r], wh[4 + r], hd2); dwh[4 + r] = __builtin_fmaf(hg[4 + r], hq2, dwh[4 + r]);
                    }
                    dht += (q16::qpassb(ht, vq.a1) ? hd1 : 0.0f) + (q16::qpassb(ht, vq.a2) ? hd2 : 0.0f);
                } else if constexpr (QH) {      // dL/dW on q_a(h), dL/dh through the activation quantiser's pass mask
                    const float hq = q16::qapply(ht, qa);
                    float hd = 0.0f;
#pragma unroll
                    for (int r = 0; r < NH; ++r) { hd = __builtin_fmaf(hg[r], wh[r], hd); dwh[r] = __builtin_fmaf(hg[r], hq, dwh[r]); }
                    dht += q16::qpassb(ht, qa) ? hd : 0.0f;
                } else {
#pragma unroll
                    for (int r = 0; r < NH; ++r) { dht = __builtin_fmaf(hg[r], wh[r], dht); dwh[r] = __builtin_fmaf(hg[r], ht, dwh[r]); }
                }
                const float dct = __builtin_fmaf(dht * g[3], __builtin_fmaf(-tc, tc, 1.0f), dc);       // dL/dc(t)
                dc = dct * g[1];
                // the row's own pre-activation gradient: d_i = dc g i (1 - i), d_f = dc c(t-1) f (1 - f), d_g = dc i (1 - g^2), d_o = dh tanh c o (1 - o)
                const float own = role == 0 ? g[0] : role == 1 ? g[1] : role == 2 ? g[2] : g[3];
                const float mul = role == 0 ? g[2] : role == 1 ? cp : role == 2 ? g[0] : tc;
                const float up = (role == 3 ? dht : dct) * mul;
                const float d_row = up * (is_g ? __builtin_fmaf(-own, own, 1.0f) : own * (1.0f - own));
                float part = rotdot(0.0f, wT, d_row);
                part = sum_rows4(part);
                dh = part;
                const float xsx = col < F ? ftab[4 * t + xcol] : (col == F ? 1.0f : 0.0f);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_row, hp, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_row, xsx, acc2, 0, 0, 0);
            }
        }
        wave_lds_fence();
    }
    // ---- the workgroup's row of partial gradients (every entry written) ----
    float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
    const float lp = wave_sum(loss_acc);
#pragma unroll
    for (int r = 0; r < (VD ? 26 : 2); ++r) tacc[r] = wave_sum(tacc[r]);
    if (vo && role == 0) {
#pragma unroll
        for (int r = 0; r < NH; ++r) {
            if (VD) prow[(r < 4 ? L.o_w_l1 + r * H : L.o_w_l2 + (r - 4) * H) + col] = QH ? dwh[r] * whm[r] : dwh[r];
            else prow[L.o_w_out + r * H + col] = QH ? dwh[r] * whm[r] : dwh[r];
        }
    }
    if (lane == 0) {
        if constexpr (QH) {      // the scales: zero gradients (quantizers.py:56-65)
            prow[L.o_q_out] = 0.0f; prow[L.o_q_out + 1] = 0.0f; prow[L.o_q_out + 2] = 0.0f;
            if constexpr (VD) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { prow[L.o_q_l1 + k] = 0.0f; prow[L.o_q_l2 + k] = 0.0f; }
            }
        }
        if constexpr (VD) {
#pragma unroll
            for (int k = 0; k < 16; ++k) prow[L.o_w_out + k] = QH ? tacc[k] * q16::qpass(a.params[L.o_w_out + k], vq.wo) : tacc[k];
            prow[L.o_b_out] = tacc[16]; prow[L.o_b_out + 1] = tacc[17];
#pragma unroll
            for (int k = 0; k < 4; ++k) { prow[L.o_b_l1 + k] = tacc[18 + k]; prow[L.o_b_l2 + k] = tacc[22 + k]; }
        } else { prow[L.o_b_out] = tacc[0]; prow[L.o_b_out + 1] = tacc[1]; }
        prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
    }
    // MFMA block k = gate k; register 4 k + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * role + rr;
            if (i < H) {
                if (col < H) prow[L.o_w_hh + (k * H + i) * H + col] = acc1[4 * k + rr];
                const float v = acc2[4 * k + rr];
                if (col < F) prow[L.o_w_ih + (k * H + i) * F + col] = v;
                else if (col == F) { prow[L.o_b_ih + k * H + i] = v; prow[L.o_b_hh + k * H + i] = v; }
            }
        }
}

// -------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------
template <int R, bool VD>
struct LstmGrad {
    f32x4 thh[4][R][R];   // dW_hh tiles
    f32x4 tih[4][R];      // dW_ih | bias column F
    float dwout[2], dbout[2];          // plain head
    float dwl1[4], dwl2[4];            // VD: d fc_lambda weights (per lane column o)
    float dbl1[4], dbl2[4], dwo[16];   // VD: uniform-per-sequence accumulators
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int x = 0; x < R; ++x) {
                tih[g][x] = z4;
#pragma unroll
                for (int y = 0; y < R; ++y) thh[g][x][y] = z4;
            }
        dwout[0] = dwout[1] = dbout[0] = dbout[1] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dwl1[k] = dwl2[k] = dbl1[k] = dbl2[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) dwo[k] = 0.f;
    }
};

// VDLSTM dL/dx: sample t feeds the windows of steps t..t+3 (slot 3..0), so its gradient w.r.t. (a, cos, sin)
// is complete once the backward sweep (descending t) has processed step t.  `acc` is the sliding window of
// partial sums for samples t-3..t (per sequence, replicated on the sequence's lanes); slot 3 is finalised
// and the window shifts every step.  What is left in slots 1..3 after step 0 belongs to the circular halo
// (samples T-3..T-1, vdlstm.py:66-74) and is added by vd_dx_wrap().
struct VdAcc {
    float a[4], c[4], s[4];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = c[k] = s[k] = 0.0f;
    }
};

template <int R, bool VD, bool NW, bool DX, bool FULL, bool QH = false>
__device__ __forceinline__ void lstm_bwd_block(const SeqArgs& a, const LstmW<R, VD>& w, const float* pl, const LstmLayout& L,
                                               TabPtr tlane, LstmGrad<R, VD>& G, const LaneId& id, const float2* xr,
                                               const float2* dys, float2* dxs, int tloc, int nstep, float h, float c,
                                               float& dh, float& dc, VdAcc& acc) {
    constexpr int F = VD ? 4 : 2, LPS = 16 * R, S = kCkptStride;
    using T = LstmTabs<R>;
    const int lane = id.lane, col = id.col, row = id.row, s = id.s;
    float hp_s[S], cp_s[S], i_s[S], f_s[S], g_s[S], o_s[S], tc_s[S];
    tlane = opaque(tlane);
    {
        float whh[4][R][16];
        load_rot4<R>(whh, tlane, T::kHH);
        VdWin win;
        if constexpr (VD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) vd_elem(xr[tloc + k - 3], win.a[k + 1], win.cw[k + 1], win.sw[k + 1]);
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            if (FULL || i < nstep) {
                const float2 xv = xr[tloc + i];
                float xin[F];
                if constexpr (VD) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) win.a[k] = win.a[k + 1];
                    float cw_, sw_;
                    vd_elem(xv, win.a[3], cw_, sw_);
#pragma unroll
                    for (int k = 0; k < 4; ++k) xin[k] = win.a[k];
                } else { xin[0] = xv.x; xin[1] = xv.y; }
                hp_s[i] = h; cp_s[i] = c;
                lstm_cell_fwd<R, VD>(w, whh, xin, h, c, i_s[i], f_s[i], g_s[i], o_s[i], tc_s[i]);
            }
        }
    }
    tlane = opaque(tlane);
    float whhT[4][R][16];
    load_rot4<R>(whhT, tlane, T::kHHT);
#pragma unroll
    for (int i = S - 1; i >= 0; --i) {
        if (FULL || i < nstep) {
            const int tt = tloc + i;
            const float2 dyv = dys[s * kChunkPad + tt];
            const float ht = o_s[i] * tc_s[i];
            float xin[F];
            VdWin win;
            if constexpr (VD) {
#pragma unroll
                for (int k = 0; k < 4; ++k) vd_elem(xr[tt + k - 3], win.a[k], win.cw[k], win.sw[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k) xin[k] = win.a[k];
            } else { const float2 xv = xr[tt]; xin[0] = xv.x; xin[1] = xv.y; }
            float dht = dh;
            if constexpr (VD && QH) {
                // the three heads as INT_Linear (weights already quantised in the staged copy): fc_lambda_1 / _2 on q_a(h) of their own grids,
                // fc_out on q_a(l cos), q_a(l sin); every activation quantiser's pass mask gates the gradient that flows back through it
                const float hq1 = q16::qapply(ht, w.vq.a1), hq2 = q16::qapply(ht, w.vq.a2);
                float hd1 = 0.0f, hd2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float l1 = seq_sum<R>(w.wl1[k] * hq1) + pl[L.o_b_l1 + k], l2 = seq_sum<R>(w.wl2[k] * hq2) + pl[L.o_b_l2 + k];
                    const float z1 = l1 * win.cw[k], z2 = l2 * win.sw[k];
                    const float dz1 = q16::qpass(z1, w.vq.ao) * __builtin_fmaf(dyv.x, pl[L.o_w_out + k], dyv.y * pl[L.o_w_out + 8 + k]);
                    const float dz2 = q16::qpass(z2, w.vq.ao) * __builtin_fmaf(dyv.x, pl[L.o_w_out + 4 + k], dyv.y * pl[L.o_w_out + 12 + k]);
                    const float d1 = dz1 * win.cw[k], d2 = dz2 * win.sw[k];
                    hd1 = __builtin_fmaf(d1, w.wl1[k], hd1); hd2 = __builtin_fmaf(d2, w.wl2[k], hd2);
                    if constexpr (DX) {
                        acc.c[k] = __builtin_fmaf(dz1, l1, acc.c[k]);
                        acc.s[k] = __builtin_fmaf(dz2, l2, acc.s[k]);
                    }
                    if constexpr (NW) {
                        G.dwl1[k] = __builtin_fmaf(d1, hq1, G.dwl1[k]);
                        G.dwl2[k] = __builtin_fmaf(d2, hq2, G.dwl2[k]);
                        G.dbl1[k] += d1; G.dbl2[k] += d2;
                        const float zq1 = q16::qapply(z1, w.vq.ao), zq2 = q16::qapply(z2, w.vq.ao);
                        G.dwo[k] = __builtin_fmaf(dyv.x, zq1, G.dwo[k]);
                        G.dwo[4 + k] = __builtin_fmaf(dyv.x, zq2, G.dwo[4 + k]);
                        G.dwo[8 + k] = __builtin_fmaf(dyv.y, zq1, G.dwo[8 + k]);
                        G.dwo[12 + k] = __builtin_fmaf(dyv.y, zq2, G.dwo[12 + k]);
                    }
                }
                dht += (q16::qpassb(ht, w.vq.a1) ? hd1 : 0.0f) + (q16::qpassb(ht, w.vq.a2) ? hd2 : 0.0f);
            } else if constexpr (VD) {
                // dz[k] = sum_c dy_c Wo[c][k];  dl1[k] = dz[k] cw[k];  dl2[k] = dz[4+k] sw[k]
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dz1 = __builtin_fmaf(dyv.x, pl[L.o_w_out + k], dyv.y * pl[L.o_w_out + 8 + k]);
                    const float dz2 = __builtin_fmaf(dyv.x, pl[L.o_w_out + 4 + k], dyv.y * pl[L.o_w_out + 12 + k]);
                    const float d1 = dz1 * win.cw[k], d2 = dz2 * win.sw[k];
                    dht = __builtin_fmaf(d1, w.wl1[k], __builtin_fmaf(d2, w.wl2[k], dht));
                    float l1 = 0.0f, l2 = 0.0f;
                    if constexpr (NW || DX) {
                        // fc_out weight gradient and d/d(cos, sin) need l1[k], l2[k] (reductions over the hidden units)
                        l1 = seq_sum<R>(w.wl1[k] * ht) + pl[L.o_b_l1 + k];
                        l2 = seq_sum<R>(w.wl2[k] * ht) + pl[L.o_b_l2 + k];
                    }
                    if constexpr (DX) {
                        acc.c[k] = __builtin_fmaf(dz1, l1, acc.c[k]);
                        acc.s[k] = __builtin_fmaf(dz2, l2, acc.s[k]);
                    }
                    if constexpr (NW) {
                        G.dwl1[k] = __builtin_fmaf(d1, ht, G.dwl1[k]);
                        G.dwl2[k] = __builtin_fmaf(d2, ht, G.dwl2[k]);
                        G.dbl1[k] += d1; G.dbl2[k] += d2;
                        const float z1 = l1 * win.cw[k], z2 = l2 * win.sw[k];
                        G.dwo[k] = __builtin_fmaf(dyv.x, z1, G.dwo[k]);
                        G.dwo[4 + k] = __builtin_fmaf(dyv.x, z2, G.dwo[4 + k]);
                        G.dwo[8 + k] = __builtin_fmaf(dyv.y, z1, G.dwo[8 + k]);
                        G.dwo[12 + k] = __builtin_fmaf(dyv.y, z2, G.dwo[12 + k]);
                    }
                }
            } else if constexpr (QH) {      // dL/dW on q_a(h), dL/dh through the activation quantiser's pass mask
                const float hd = __builtin_fmaf(dyv.x, w.wout[0], dyv.y * w.wout[1]);
                dht += q16::qpassb(ht, w.qa) ? hd : 0.0f;
                if constexpr (NW) {
                    const float hq = q16::qapply(ht, w.qa);
                    G.dwout[0] = __builtin_fmaf(dyv.x, hq, G.dwout[0]);
                    G.dwout[1] = __builtin_fmaf(dyv.y, hq, G.dwout[1]);
                }
            } else {
                dht = __builtin_fmaf(dyv.x, w.wout[0], __builtin_fmaf(dyv.y, w.wout[1], dht));
                if constexpr (NW) {
                    G.dwout[0] = __builtin_fmaf(dyv.x, ht, G.dwout[0]);
                    G.dwout[1] = __builtin_fmaf(dyv.y, ht, G.dwout[1]);
                }
            }
            if constexpr (NW) { G.dbout[0] += dyv.x; G.dbout[1] += dyv.y; }
            // cell backward
            const float dO = dht * tc_s[i];
            const float dct = __builtin_fmaf(dht * o_s[i], __builtin_fmaf(-tc_s[i], tc_s[i], 1.0f), dc);
            const float dpi = (dct * g_s[i]) * (i_s[i] * (1.0f - i_s[i]));
            const float dpf = (dct * cp_s[i]) * (f_s[i] * (1.0f - f_s[i]));
            const float dpg = (dct * i_s[i]) * __builtin_fmaf(-g_s[i], g_s[i], 1.0f);
            const float dpo = dO * (o_s[i] * (1.0f - o_s[i]));
            dc = dct * f_s[i];
            if constexpr (NW) {
                float fsx = (col == F) ? 1.0f : 0.0f;
#pragma unroll
                for (int q = 0; q < F; ++q) fsx = (col == q) ? xin[q] : fsx;
                const float hp = hp_s[i];
                if constexpr (R == 1) {
                    G.tih[0][0] = mfma4(dpi, fsx, G.tih[0][0]); G.tih[1][0] = mfma4(dpf, fsx, G.tih[1][0]);
                    G.tih[2][0] = mfma4(dpg, fsx, G.tih[2][0]); G.tih[3][0] = mfma4(dpo, fsx, G.tih[3][0]);
                    G.thh[0][0][0] = mfma4(dpi, hp, G.thh[0][0][0]); G.thh[1][0][0] = mfma4(dpf, hp, G.thh[1][0][0]);
                    G.thh[2][0][0] = mfma4(dpg, hp, G.thh[2][0][0]); G.thh[3][0][0] = mfma4(dpo, hp, G.thh[3][0][0]);
                } else {
                    const float hpx = swap16(hp);
#pragma unroll
                    for (int qo = 0; qo < R; ++qo) {
                        const bool mine = row == qo;
                        const float d[4] = {mine ? dpi : 0.f, mine ? dpf : 0.f, mine ? dpg : 0.f, mine ? dpo : 0.f};
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            G.tih[g][qo] = mfma4(d[g], fsx, G.tih[g][qo]);
#pragma unroll
                            for (int qm = 0; qm < R; ++qm) G.thh[g][qo][qm] = mfma4(d[g], qm == qo ? hp : hpx, G.thh[g][qo][qm]);
                        }
                    }
                }
            }
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            rotdot3x(d0, d1, d2, whhT[0][0], whhT[1][0], whhT[2][0], dpi, dpf, dpg);
            d0 = rotdot(d0, whhT[3][0], dpo);
            if constexpr (R == 2) {
                rotdot3x(d0, d1, d2, whhT[0][1], whhT[1][1], whhT[2][1], swap16(dpi), swap16(dpf), swap16(dpg));
                d0 = rotdot(d0, whhT[3][1], swap16(dpo));
            }
            dh = d0 + d1 + d2;
            if constexpr (DX && VD) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float p = __builtin_fmaf(w.wih[0][k], dpi, __builtin_fmaf(w.wih[1][k], dpf,
                                    __builtin_fmaf(w.wih[2][k], dpg, w.wih[3][k] * dpo)));
                    acc.a[k] += seq_sum<R>(p);
                }
                const float2 g = polar_sample_bwd(win.a[3], win.cw[3], win.sw[3], acc.a[3], acc.c[3], acc.s[3]);
                if ((lane & (LPS - 1)) == 0) dxs[s * kChunkPad + tt] = g;
#pragma unroll
                for (int k = 3; k > 0; --k) { acc.a[k] = acc.a[k - 1]; acc.c[k] = acc.c[k - 1]; acc.s[k] = acc.s[k - 1]; }
                acc.a[0] = acc.c[0] = acc.s[0] = 0.0f;
            }
            if constexpr (DX && !VD) {
                float dI = __builtin_fmaf(w.wih[0][0], dpi, __builtin_fmaf(w.wih[1][0], dpf, __builtin_fmaf(w.wih[2][0], dpg, w.wih[3][0] * dpo)));
                float dQ = __builtin_fmaf(w.wih[0][1], dpi, __builtin_fmaf(w.wih[1][1], dpf, __builtin_fmaf(w.wih[2][1], dpg, w.wih[3][1] * dpo)));
                dI = seq_sum<R>(dI); dQ = seq_sum<R>(dQ);
                if ((lane & (LPS - 1)) == 0) dxs[s * kChunkPad + tt] = make_float2(dI, dQ);
            }
        }
    }
}

template <int R, bool VD, bool QH = false>
__device__ __forceinline__ void lstm_write_partials(float* prow, const LstmLayout& L, LstmGrad<R, VD>& G, int lane, int row,
                                                    int col, const LstmW<R, VD>* w = nullptr, const float* gparams = nullptr) {
    constexpr int F = VD ? 4 : 2;
    const int H = L.H, o = 16 * row + col, seq = lane / (16 * R);
    const int g4 = lane >> 4, c = lane & 15;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int qo = 0; qo < R; ++qo)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * qo + 4 * g4 + rr;
                if (i < H) {
                    const float v = G.tih[g][qo][rr];
                    if (c < F) prow[L.o_w_ih + (g * H + i) * F + c] = v;
                    else if (c == F) { prow[L.o_b_ih + g * H + i] = v; prow[L.o_b_hh + g * H + i] = v; }
#pragma unroll
                    for (int qm = 0; qm < R; ++qm) {
                        const int j = 16 * qm + c;
                        if (j < H) prow[L.o_w_hh + (g * H + i) * H + j] = G.thh[g][qo][qm][rr];
                    }
                }
            }
    const float db0 = across_seqs<R>(G.dbout[0]), db1 = across_seqs<R>(G.dbout[1]);
    if (lane == 0) { prow[L.o_b_out] = db0; prow[L.o_b_out + 1] = db1; }
    if constexpr (VD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a1 = across_seqs<R>(G.dwl1[k]), a2 = across_seqs<R>(G.dwl2[k]);
            const float b1 = across_seqs<R>(G.dbl1[k]), b2 = across_seqs<R>(G.dbl2[k]);
            if constexpr (QH) {      // the weight quantisers' pass masks, from the global (unquantised) copy
                if (o < H) { a1 *= q16::qpass(gparams[L.o_w_l1 + k * H + o], w->vq.w1); a2 *= q16::qpass(gparams[L.o_w_l2 + k * H + o], w->vq.w2); }
            }
            if (seq == 0 && o < H) { prow[L.o_w_l1 + k * H + o] = a1; prow[L.o_w_l2 + k * H + o] = a2; }
            if (lane == 0) { prow[L.o_b_l1 + k] = b1; prow[L.o_b_l2 + k] = b2; }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float v = across_seqs<R>(G.dwo[k]);
            if constexpr (QH) v *= q16::qpass(gparams[L.o_w_out + k], w->vq.wo);
            if (lane == 0) prow[L.o_w_out + k] = v;
        }
        if constexpr (QH) {      // the nine scales: zero gradients (quantizers.py:56-65)
            if (lane < 3) { prow[L.o_q_l1 + lane] = 0.0f; prow[L.o_q_l2 + lane] = 0.0f; prow[L.o_q_out + lane] = 0.0f; }
        }
    } else {
        float w0 = across_seqs<R>(G.dwout[0]), w1 = across_seqs<R>(G.dwout[1]);
        if constexpr (QH) {      // the weight quantiser's pass mask; the three scales get zero gradients (quantizers.py:56-65)
            w0 *= w->woutm[0]; w1 *= w->woutm[1];
            if (lane < 3) prow[L.o_q_out + lane] = 0.0f;
        }
        if (seq == 0 && o < H) { prow[L.o_w_out + o] = w0; prow[L.o_w_out + H + o] = w1; }
    }
}

template <int R, bool VD, bool NW, bool DX, bool QH = false>
__global__ __launch_bounds__((R == 1 && !VD) ? kMaxThreads : kMaxThreads / 2, (R == 1 && !VD) ? 2 : 1) void lstm_bwd_kernel(SeqArgs a) {
    constexpr int SPW = 4 / R, S = kCkptStride;
    using T = LstmTabs<R>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<R>();
    const int lane = id.lane;
    const LstmLayout L = lstm_layout(a.H, VD, QH);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    LstmW<R, VD> w;
    if constexpr (QH && VD) { w.vq = vd_quantisers(pl, L, a.bits_w, a.bits_a); vd_quantise_staged(pl, L, w.vq); }
    float* tab = smem + pad4(L.P);
    fill_lstm_tabs<R, true>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + T::kFloats) + id.wave * (SPW * kHaloStride + 2 * SPW * kChunkPad);
    float2* dys = xs + SPW * kHaloStride;
    float2* dxs = dys + SPW * kChunkPad;
    load_lstm_w<R, VD>(w, pl, L, id.row, id.col);
    if constexpr (QH && !VD) lstm_quantise_head<R, VD>(w, pl, L, a.bits_w, a.bits_a);
    LstmGrad<R, VD> G;
    G.zero();
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float dh = 0.0f, dc = 0.0f;
        VdAcc acc;
        acc.zero();
        int cur_chunk = -1;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        const int pt0 = cur_chunk * kChunk;
                        wave_lds_fence();
                        stage_out<SPW>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    }
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                stage_in_halo<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane);
                stage_in<SPW>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            const float* ck = a.ckpt + ((size_t)grp * a.nck + blk) * 128;
            const float h0 = blk ? ck[lane] : 0.0f, c0 = blk ? ck[64 + lane] : 0.0f;
            const float2* xr = xs + id.s * kHaloStride + kHalo;
            if (nstep == S)
                lstm_bwd_block<R, VD, NW, DX, true, QH>(a, w, pl, L, tlane, G, id, xr, dys, dxs, tb - t0, nstep, h0, c0, dh, dc, acc);
            else
                lstm_bwd_block<R, VD, NW, DX, false, QH>(a, w, pl, L, tlane, G, id, xr, dys, dxs, tb - t0, nstep, h0, c0, dh, dc, acc);
        }
        if constexpr (DX) {
            if (cur_chunk >= 0) {
                const int pt0 = cur_chunk * kChunk;
                wave_lds_fence();
                stage_out<SPW>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                wave_lds_fence();
            }
            if constexpr (VD) {
                // circular halo: steps 0..2 also read samples T-3..T-1; their share is in window slots 1..3.
                // One add per element, after this wave's own dx stores are acknowledged: deterministic.
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                const int seq = b0 + id.s;
                if ((lane & (16 * R - 1)) == 0 && seq < a.B) {
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const size_t e = (size_t)seq * a.T + (a.T - 4 + k);
                        const float2 xv = reinterpret_cast<const float2*>(a.x)[e];
                        float a_, cw, sw;
                        vd_elem(xv, a_, cw, sw);
                        const float2 g = polar_sample_bwd(a_, cw, sw, acc.a[k], acc.c[k], acc.s[k]);
                        atomicAdd(a.dx + 2 * e, g.x);
                        atomicAdd(a.dx + 2 * e + 1, g.y);
                    }
                }
            }
        }
    }
    if constexpr (NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        lstm_write_partials<R, VD, QH>(smem + id.wave * P4, L, G, lane, id.row, id.col, &w, a.params);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < id.nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
static size_t lstm_lds_bytes(int P, int R, int waves, size_t wave_floats, bool reduce) {
    size_t n = ((size_t)pad4(P) + 8 * R * 4 * 64 * 4 + (size_t)waves * wave_floats) * sizeof(float);
    if (reduce && n < reduce_scratch_bytes(P, waves)) n = reduce_scratch_bytes(P, waves);
    return n;
}
static LaunchShape lstm_bwd_shape(int R, bool vd, int ngroups) {
    const int w = (R == 1 && !vd) ? 8 : 4;   // VDLSTM / two-row models need > 256 registers: one wave per SIMD
    return persistent_shape(ngroups, w, w);
}

template <int R, bool VD>
static int lstm_launch_fwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = persistent_shape(a.ngroups, R == 1 ? 16 : 8);
    const size_t lds = lstm_lds_bytes(P, R, ls.waves, 2 * ((4 / R) * kHaloStride + (4 / R) * kChunkPad), false);
    auto k = lstm_fwd_kernel<R, VD>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
template <int NB, bool VD>
static int lstm_launch_eval(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = ((size_t)pad4(P) + LstmTabs<NB>::kFloats + LstmEvalLds<NB>::kFloats) * sizeof(float);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(a.B), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    if (a.bits_w > 0) return a.ckpt ? launch(lstm_eval_kernel<NB, VD, true, true>) : launch(lstm_eval_kernel<NB, VD, false, true>);
    return a.ckpt ? launch(lstm_eval_kernel<NB, VD, true>) : launch(lstm_eval_kernel<NB, VD, false>);
}
template <int R, bool VD, bool NW, bool DX>
static int lstm_launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = lstm_bwd_shape(R, VD, a.ngroups);
    const size_t lds = lstm_lds_bytes(P, R, ls.waves, 2 * ((4 / R) * kHaloStride + 2 * (4 / R) * kChunkPad), NW);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    };
    if (a.bits_w > 0) return launch(lstm_bwd_kernel<R, VD, NW, DX, true>);
    return launch(lstm_bwd_kernel<R, VD, NW, DX>);
}
template <int R, bool VD>
static int lstm_launch_bwd_mode(hipStream_t st, const SeqArgs& a, int P) {
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && !dx) return lstm_launch_bwd<R, VD, true, false>(st, a, P);
    if (!nw && dx) return lstm_launch_bwd<R, VD, false, true>(st, a, P);
    if (nw && dx) return lstm_launch_bwd<R, VD, true, true>(st, a, P);
    return ODPD_EINVAL;
}

#define ODPD_LSTM_DISPATCH(FN, ...)                                   \
    if (R == 1 && !vd) return FN<1, false>(__VA_ARGS__);               \
    if (R == 2 && !vd) return FN<2, false>(__VA_ARGS__);               \
    if (R == 1 && vd) return FN<1, true>(__VA_ARGS__);                 \
    if (R == 2 && vd) return FN<2, true>(__VA_ARGS__);

// the gate-parallel fused train kernel: one sequence per single-wave workgroup, BPTT state of the whole frame in LDS
static size_t lstm_gp_lds_bytes(int P, bool vd, int T, bool pg) {
    return ((size_t)pad4(P) + (vd ? lstm_gp_buffer_floats<true>(T, pg) : lstm_gp_buffer_floats<false>(T, pg))) * sizeof(float);
}
static int lstm_gp_blocks_per_cu(int P, bool vd, int T, bool pg) {
    const size_t lds = lstm_gp_lds_bytes(P, vd, T, pg);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}
static bool lstm_gp_parks_gates(int P, bool vd, int B, int T) { return (long)B <= (long)device_cus() * lstm_gp_blocks_per_cu(P, vd, T, true); }
bool lstm_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if ((m->backbone != ODPD_LSTM && m->backbone != ODPD_VDLSTM) || m->hidden > 16 || lstm_train_uses_s16(m, B)) return false;
    const bool vd = m->backbone == ODPD_VDLSTM;
    if (vd && T < kHalo) return false;
    const int P = lstm_layout(m->hidden, vd, m->bits_w > 0).P;
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && lstm_gp_blocks_per_cu(P, vd, T, false) > 0;
    // up to two rounds of workgroups: the alternative here is the forward / loss / backward chain of the row-rotated kernels
    return (long)B <= 2L * device_cus() * lstm_gp_blocks_per_cu(P, vd, T, false);
}
int lstm_gp_rows(const odpd_model_t* m, int B, int T) {
    const bool vd = m->backbone == ODPD_VDLSTM;
    const int P = lstm_layout(m->hidden, vd, m->bits_w > 0).P;
    const bool pg = lstm_gp_parks_gates(P, vd, B, T);
    const long cap = (long)device_cus() * (kMaxLds / lstm_gp_lds_bytes(P, vd, T, pg));
    return B < cap ? B : (int)cap;
}
int lstm_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const bool vd = m->backbone == ODPD_VDLSTM;
    const int P = lstm_layout(m->hidden, vd, m->bits_w > 0).P;
    const bool pg = lstm_gp_parks_gates(P, vd, a.B, a.T);
    const size_t lds = lstm_gp_lds_bytes(P, vd, a.T, pg);
    const int grid = lstm_gp_rows(m, a.B, a.T);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    if (vd && m->bits_w > 0) return pg ? launch(lstm_gp_train_kernel<true, true, true>) : launch(lstm_gp_train_kernel<true, false, true>);
    if (vd) return pg ? launch(lstm_gp_train_kernel<true, true>) : launch(lstm_gp_train_kernel<true, false>);
    if (m->bits_w > 0) return pg ? launch(lstm_gp_train_kernel<false, true, true>) : launch(lstm_gp_train_kernel<false, false, true>);
    return pg ? launch(lstm_gp_train_kernel<false, true>) : launch(lstm_gp_train_kernel<false, false>);
}
int lstm_family_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const int R = rows_per_seq(m->hidden);
    const bool vd = m->backbone == ODPD_VDLSTM;
    if (!R) return ODPD_EUNSUPPORTED;
    if (vd && a.T < kHalo) return ODPD_EINVAL;
    const int P = lstm_layout(m->hidden, vd, m->bits_w > 0).P;
    // sequences that each get a SIMD of their own (inference, and the checkpoint-writing forward of the split train path): the gate-parallel kernel
    if (m->bits_w > 0 || (a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0)) { ODPD_LSTM_DISPATCH(lstm_launch_eval, st, a, P) }
    ODPD_LSTM_DISPATCH(lstm_launch_fwd, st, a, P)
    return ODPD_EUNSUPPORTED;
}
int lstm_family_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const int R = rows_per_seq(m->hidden);
    const bool vd = m->backbone == ODPD_VDLSTM;
    if (!R) return ODPD_EUNSUPPORTED;
    if (vd && a.T < kHalo) return ODPD_EINVAL;
    const int P = lstm_layout(m->hidden, vd, m->bits_w > 0).P;
    ODPD_LSTM_DISPATCH(lstm_launch_bwd_mode, st, a, P)
    return ODPD_EUNSUPPORTED;
}
int lstm_family_rows(const odpd_model_t* m, int B) {
    const int R = rows_per_seq(m->hidden);
    if (!R) return ODPD_EUNSUPPORTED;
    return lstm_bwd_shape(R, m->backbone == ODPD_VDLSTM, num_groups(B, R)).grid;
}

}  // namespace odpd
