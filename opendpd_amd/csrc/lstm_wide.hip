// lstm_wide.hip — the plain LSTM backbone (backbones/lstm.py:4-48: nn.LSTM(2 -> H) with zero (h, c), fc_out H -> 2) with 33 .. 64 hidden units:
// the mapping of gru_wide.hip — ONE sequence per single-wave workgroup, LANE = HIDDEN UNIT — for four gates (nn.LSTM order i, f, g, o).
//   forward   the W_hh rows of gates i, f, g stay in the lane's registers (3 x 64), gate o's row is read from a padded LDS copy (row stride
//             65: lane = unit reads are conflict-free); the state is broadcast through LDS once per step; fc_out with lane = time step on
//             64-step chunks; i, f, g, o, c and h of every step go to the per-sequence record in HBM (`ckpt`: B x T x 6 x 64 floats) when a
//             backward pass follows;
//   backward  chunks in reverse; the lane of unit k forms dL/dh(t-1)[k] from the four gate gradients (broadcast through LDS) and column k
//             of W_hh (staged parameters: consecutive lanes, consecutive addresses); dW_hh as outer products on the 4-block MFMA with
//             the state rotated by 0 / 16 / 32 / 48 lanes (4 gates x 4 rotations x 16 accumulators = the whole AGPR file), dW_ih, the
//             biases and fc_out on the VALU.  One row of partial gradients per workgroup (every entry written).
#include "odpd_seq.h"
#include "odpd_quant.h"

namespace odpd {
namespace {
constexpr int kLC = 64;          // time steps per chunk
constexpr int kLS = 65;          // row stride of the per-chunk [time][unit] arrays and of the padded gate-o rows
constexpr int kLHs = ((kLC + 1) * kLS + 3) & ~3;
constexpr int kLNS = 6;          // record of a step: i, f, g, o, c, h

__host__ __device__ inline int lstmw_fwd_floats(int P) { return pad4(P) + kLC * 2 + 64 + kLC * kLS + 64 * kLS; }
__host__ __device__ inline int lstmw_bwd_floats(int P) { return pad4(P) + kLC * 2 + kLC * 2 + kLC * 2 + 4 * 64 + kLHs; }

template <bool SAVE>
__global__ __launch_bounds__(64) void wide_lstm_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const LstmLayout L = lstm_layout(a.H, 0, a.bits_w > 0);
    const int H = L.H, T = a.T;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* xb = smem + pad4(L.P);              // [64][2]: I, Q of the chunk's steps
    float* hb = xb + kLC * 2;                  // [64]: the state, for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: h of the chunk's steps
    float* wop = hist + kLC * kLS;             // [64][65]: gate o's W_hh rows, zero padded
    const bool vo = lane < H;
    for (int i = lane; i < 64 * kLS; i += 64) {
        const int j = i / kLS, k = i % kLS;
        wop[i] = (j < H && k < H) ? pl[L.o_w_hh + (3 * H + j) * H + k] : 0.0f;
    }
    // `--quant` (bits_w > 0; run-time, wave-uniform): fc_out is an INT_Linear (quant_layers.py:48-85) — its weights become their quantised
    // values in the staged copy, the chunk's states are quantised where the head reads them, ODPD_FLAG_EVAL adds the 16-bit output grid
    const bool qh = a.bits_w > 0;
    q16::Quant qa{1.0f, 1.0f, 0.0f, 0.0f}, qo{1.0f, 1.0f, 0.0f, 0.0f};
    if (qh) {
        const q16::Quant qw = q16::make_quant(pl[L.o_q_out], a.bits_w);
        qa = q16::make_quant(pl[L.o_q_out + 1], a.bits_a);
        qo = q16::make_quant(pl[L.o_q_out + 2], 16);
        wave_lds_fence();
        for (int i = lane; i < 2 * H; i += 64) pl[L.o_w_out + i] = q16::qapply(pl[L.o_w_out + i], qw);
        wave_lds_fence();
    }
    float whh[3][64], wih[4][2], bg[4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 64; ++k) whh[g][k] = (vo && k < H) ? pl[L.o_w_hh + (g * H + lane) * H + k] : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        wih[g][0] = vo ? pl[L.o_w_ih + (g * H + lane) * 2] : 0.0f;
        wih[g][1] = vo ? pl[L.o_w_ih + (g * H + lane) * 2 + 1] : 0.0f;
        bg[g] = vo ? pl[L.o_b_ih + g * H + lane] + pl[L.o_b_hh + g * H + lane] : 0.0f;
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kLNS * 64 : nullptr;
        float h = 0.0f, c = 0.0f;
        for (int t0 = 0; t0 < T; t0 += kLC) {
            const int len = min(kLC, T - t0);
            wave_lds_fence();
            reinterpret_cast<float2*>(xb)[lane] = t0 + lane < T ? xg[t0 + lane] : make_float2(0.0f, 0.0f);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                hb[lane] = h;
                wave_lds_fence();
                const float2 xv = reinterpret_cast<const float2*>(xb)[tt];
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = __builtin_fmaf(wih[g][1], xv.y, __builtin_fmaf(wih[g][0], xv.x, bg[g]));
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
                const float* wo = wop + lane * kLS;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 hv = hb4[q];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        pre[g] = __builtin_fmaf(whh[g][4 * q], hv.x, pre[g]); pre[g] = __builtin_fmaf(whh[g][4 * q + 1], hv.y, pre[g]);
                        pre[g] = __builtin_fmaf(whh[g][4 * q + 2], hv.z, pre[g]); pre[g] = __builtin_fmaf(whh[g][4 * q + 3], hv.w, pre[g]);
                    }
                    pre[3] = __builtin_fmaf(wo[4 * q], hv.x, pre[3]); pre[3] = __builtin_fmaf(wo[4 * q + 1], hv.y, pre[3]);
                    pre[3] = __builtin_fmaf(wo[4 * q + 2], hv.z, pre[3]); pre[3] = __builtin_fmaf(wo[4 * q + 3], hv.w, pre[3]);
                }
                const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), go = sigmoidf_(pre[3]);
                const float cn = vo ? __builtin_fmaf(gf, c, gi * gg) : 0.0f;
                const float hn = vo ? go * tanhf_(cn) : 0.0f;
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * kLNS * 64 + lane;
                    s[0] = gi; s[64] = gf; s[128] = gg; s[192] = go; s[256] = cn; s[320] = hn;
                }
                c = cn; h = hn;
                hist[tt * kLS + lane] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step
                const float* hr = hist + lane * kLS;
                float y0 = qh ? 0.0f : pl[L.o_b_out], y1 = qh ? 0.0f : pl[L.o_b_out + 1];
                for (int j = 0; j < H; ++j) {
                    const float hv = qh ? q16::qapply(hr[j], qa) : hr[j];
                    y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + H + j], hv, y1);
                }
                if (qh) {      // grid sums first, then the float bias (F.linear(q_a(h), q_w(W), b))
                    y0 += pl[L.o_b_out]; y1 += pl[L.o_b_out + 1];
                    if (a.eval_out) { y0 = q16::qapply(y0, qo); y1 = q16::qapply(y1, qo); }
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(64) void wide_lstm_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
    const LstmLayout L = lstm_layout(a.H, 0, a.bits_w > 0);
    const int H = L.H, T = a.T, NC = (T + kLC - 1) / kLC;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* xb = smem + pad4(L.P);              // [64][2]  I, Q of the chunk's steps
    float* dxb = xb + kLC * 2;                 // [64][2]  dL/dx of the chunk's steps
    float* dyb = dxb + kLC * 2;                // [64][2]  dL/dy of the chunk's steps
    float* dgb = dyb + kLC * 2;                // [4][64]  the step's gate gradients, for the broadcast reads
    float* hs = dgb + 4 * 64;                  // [65][65] row i = h(t0 - 1 + i)
    const bool vo = lane < H;
    float wih[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        wih[g][0] = vo ? pl[L.o_w_ih + (g * H + lane) * 2] : 0.0f;
        wih[g][1] = vo ? pl[L.o_w_ih + (g * H + lane) * 2 + 1] : 0.0f;
    }
    float wo0 = vo ? pl[L.o_w_out + lane] : 0.0f, wo1 = vo ? pl[L.o_w_out + H + lane] : 0.0f, wm0 = 1.0f, wm1 = 1.0f;
    const bool qh = a.bits_w > 0;      // quantised head (see the forward kernel): q_w(W) columns, their pass masks for dW_out
    float qa_inv = 1.0f, qa_s = 1.0f, qa_qn = 0.0f, qa_qp = 0.0f;
    if (qh) {
        const q16::Quant qw = q16::make_quant(pl[L.o_q_out], a.bits_w), qa = q16::make_quant(pl[L.o_q_out + 1], a.bits_a);
        wm0 = q16::qpass(wo0, qw); wm1 = q16::qpass(wo1, qw);
        wo0 = q16::qapply(wo0, qw); wo1 = q16::qapply(wo1, qw);
        auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
        qa_inv = uni(qa.inv); qa_s = uni(qa.s); qa_qn = uni(qa.qn); qa_qp = uni(qa.qp);
    }
    f32x16 acc[4][4];                          // dW_hh: gate g, the state rotated by 16 r lanes
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    float dwih[4][2], dbs[4] = {0.f, 0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { dwih[g][0] = 0.0f; dwih[g][1] = 0.0f; }
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kLNS * 64;
        float dh = 0.0f, dc = 0.0f;
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kLC, len = min(kLC, T - t0);
            wave_lds_fence();
            reinterpret_cast<float2*>(xb)[lane] = t0 + lane < T ? xg[t0 + lane] : make_float2(0.0f, 0.0f);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            hs[lane] = t0 > 0 ? sv[(size_t)(t0 - 1) * kLNS * 64 + 320 + lane] : 0.0f;
            for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * kLS + lane] = sv[(size_t)(t0 + tt) * kLNS * 64 + 320 + lane];
            wave_lds_fence();
            // the chunk's steps in reverse, lane = unit (the next step's record is in flight while this one is worked on)
            float in_, fn_, gn_, on_, cn_, cpn_;
            {
                const float* s = sv + (size_t)(t0 + len - 1) * kLNS * 64 + lane;
                in_ = s[0]; fn_ = s[64]; gn_ = s[128]; on_ = s[192]; cn_ = s[256];
                cpn_ = t0 + len - 1 > 0 ? s[256 - kLNS * 64] : 0.0f;
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float gi = in_, gf = fn_, gg = gn_, go = on_, ct = cn_, cp = cpn_;
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * kLNS * 64 + lane;
                    in_ = s[0]; fn_ = s[64]; gn_ = s[128]; on_ = s[192]; cn_ = s[256];
                    cpn_ = t0 + tt - 1 > 0 ? s[256 - kLNS * 64] : 0.0f;
                }
                const float hp = hs[tt * kLS + lane], ht = hs[(tt + 1) * kLS + lane];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                float dht, hhead = ht;
                if (qh) {      // the head saw q_a(h); dL/dh passes where h lies inside the activation grid
                    const float v = ht * qa_inv, m = __builtin_amdgcn_fmed3f(v, qa_qn, qa_qp);
                    hhead = rintf(m) * qa_s;
                    dht = __builtin_fmaf(m == v ? 1.0f : 0.0f, __builtin_fmaf(d.x, wo0, d.y * wo1), dh);
                } else dht = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, dh));
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, hhead, dwo0); dwo1 = __builtin_fmaf(d.y, hhead, dwo1); }
                const float tc = tanhf_(ct);
                const float dct = __builtin_fmaf(dht * go, __builtin_fmaf(-tc, tc, 1.0f), dc);      // dL/dc(t)
                const float dpi = vo ? (dct * gg) * (gi * (1.0f - gi)) : 0.0f;
                const float dpf = vo ? (dct * cp) * (gf * (1.0f - gf)) : 0.0f;
                const float dpg = vo ? (dct * gi) * __builtin_fmaf(-gg, gg, 1.0f) : 0.0f;
                const float dpo = vo ? (dht * tc) * (go * (1.0f - go)) : 0.0f;
                dc = vo ? dct * gf : 0.0f;
                dgb[lane] = dpi; dgb[64 + lane] = dpf; dgb[128 + lane] = dpg; dgb[192 + lane] = dpo;
                wave_lds_fence();
                float dhn = 0.0f;
                {
                    const float* w0 = pl + L.o_w_hh + (vo ? lane : 0);      // (lanes beyond H read column 0: finite values, result discarded)
                    const int HH = H * H;
                    for (int j4 = 0; j4 < H; j4 += 4) {
                        const float4 a0 = *reinterpret_cast<const float4*>(dgb + j4), a1 = *reinterpret_cast<const float4*>(dgb + 64 + j4),
                                     a2 = *reinterpret_cast<const float4*>(dgb + 128 + j4), a3 = *reinterpret_cast<const float4*>(dgb + 192 + j4);
                        const float v0[4] = {a0.x, a0.y, a0.z, a0.w}, v1[4] = {a1.x, a1.y, a1.z, a1.w}, v2[4] = {a2.x, a2.y, a2.z, a2.w},
                                    v3[4] = {a3.x, a3.y, a3.z, a3.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* wr = w0 + min(j4 + e, H - 1) * H;      // (rows beyond H: gate gradients are zero there)
                            dhn = __builtin_fmaf(v0[e], wr[0], dhn); dhn = __builtin_fmaf(v1[e], wr[HH], dhn);
                            dhn = __builtin_fmaf(v2[e], wr[2 * HH], dhn); dhn = __builtin_fmaf(v3[e], wr[3 * HH], dhn);
                        }
                    }
                }
                dh = vo ? dhn : 0.0f;
                const float2 xv = reinterpret_cast<const float2*>(xb)[tt];
                if constexpr (NW) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float hpr = rr == 0 ? hp : __shfl(hp, (lane + 16 * rr) & 63);
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpi, hpr, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpf, hpr, acc[1][rr], 0, 0, 0);
                        acc[2][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpg, hpr, acc[2][rr], 0, 0, 0);
                        acc[3][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpo, hpr, acc[3][rr], 0, 0, 0);
                    }
                    const float dp[4] = {dpi, dpf, dpg, dpo};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        dwih[g][0] = __builtin_fmaf(dp[g], xv.x, dwih[g][0]); dwih[g][1] = __builtin_fmaf(dp[g], xv.y, dwih[g][1]);
                        dbs[g] += dp[g];
                    }
                }
                if constexpr (DX) {
                    float d0 = __builtin_fmaf(dpi, wih[0][0], __builtin_fmaf(dpf, wih[1][0], __builtin_fmaf(dpg, wih[2][0], dpo * wih[3][0])));
                    float d1 = __builtin_fmaf(dpi, wih[0][1], __builtin_fmaf(dpf, wih[1][1], __builtin_fmaf(dpg, wih[2][1], dpo * wih[3][1])));
                    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); }
                    if (lane == 0) reinterpret_cast<float2*>(dxb)[tt] = make_float2(d0, d1);
                }
                wave_lds_fence();
            }
            if constexpr (DX) {
                wave_lds_fence();
                if (lane < len) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t0 + lane] = reinterpret_cast<const float2*>(dxb)[lane];
            }
        }
        wave_lds_fence();
    }
    if constexpr (NW) {
        float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
        for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.0f;
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_fence();
        for (int o = 32; o > 0; o >>= 1) { tb0 += __shfl_xor(tb0, o); tb1 += __shfl_xor(tb1, o); }
        if (lane == 0) { prow[L.o_b_out] = tb0; prow[L.o_b_out + 1] = tb1; }
        if (vo) {
            prow[L.o_w_out + lane] = dwo0 * wm0; prow[L.o_w_out + H + lane] = dwo1 * wm1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                prow[L.o_w_ih + (g * H + lane) * 2] = dwih[g][0]; prow[L.o_w_ih + (g * H + lane) * 2 + 1] = dwih[g][1];
                prow[L.o_b_ih + g * H + lane] = dbs[g]; prow[L.o_b_hh + g * H + lane] = dbs[g];
            }
        }
        // MFMA block bb of (gate g, rotation rr): register 4 bb + i of lane l = entry (row 4 (l / 16) + i, column l % 16) of the block
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ju = 16 * bb + 4 * quad + i, ku = 16 * ((bb + rr) & 3) + col;
                        if (ju < H && ku < H) prow[L.o_w_hh + (g * H + ju) * H + ku] = acc[g][rr][4 * bb + i];
                    }
    }
}

template <typename K>
int lstmw_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// lstm of 33 .. 64 hidden units, float or with a quantised head (bits_w > 0: fc_out as INT_Linear; vdlstm's quantised heads keep the 32-unit envelope)
bool lstm_wide_ok(const odpd_model_t* m) {
    return m->backbone == ODPD_LSTM && m->hidden > 32 && m->hidden <= 64 && !(m->flags & ODPD_FLAG_TWO_LAYERS) &&
           (m->bits_w == 0 || (m->bits_w <= 16 && m->bits_a > 0 && m->bits_a <= 16));
}
int64_t lstm_wide_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kLNS * 64; }
int lstm_wide_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int lstm_wide_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!lstm_wide_ok(m)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)lstmw_fwd_floats(lstm_layout(m->hidden, 0, m->bits_w > 0).P) * sizeof(float);
    const int grid = lstm_wide_rows(m, a.B);
    return a.ckpt ? lstmw_launch(st, wide_lstm_fwd_kernel<true>, grid, lds, a) : lstmw_launch(st, wide_lstm_fwd_kernel<false>, grid, lds, a);
}
int lstm_wide_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!lstm_wide_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)lstmw_bwd_floats(lstm_layout(m->hidden, 0, m->bits_w > 0).P) * sizeof(float);
    const int grid = lstm_wide_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return lstmw_launch(st, wide_lstm_bwd_kernel<true, true>, grid, lds, a);
    if (nw) return lstmw_launch(st, wide_lstm_bwd_kernel<true, false>, grid, lds, a);
    return lstmw_launch(st, wide_lstm_bwd_kernel<false, true>, grid, lds, a);
}

}  // namespace odpd
