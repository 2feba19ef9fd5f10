// vdlstm_wide.hip — the VDLSTM backbone (backbones/vdlstm.py:5-111) with 33 .. 64 hidden units: lstm_wide.hip's mapping (one sequence per
// single-wave workgroup, LANE = HIDDEN UNIT) with VDLSTM's input and head:
//   input     the window (|x(t-3)|, .., |x(t)|) over the frame with CIRCULAR left padding (the frame's own last three samples, vdlstm.py:66-74);
//             amplitude, cos = I / a, sin = Q / a of a 64-step chunk (+ its three-sample halo) are formed with lane = time step;
//   head      lambda_1 = fc_lambda_1(h), lambda_2 = fc_lambda_2(h) (4 rows each), y = fc_out(cat(lambda_1 cos_w, lambda_2 sin_w)) over the same window
//             (vdlstm.py:77-80), with lane = time step;
//   backward  the head's gradients of a chunk with lane = time step (dL/d lambda -> fc_lambda^T -> dL/dh(t) rows in LDS; fc_out and the biases per
//             time lane; the fc_lambda rows per unit lane over the chunk), then the recurrence in reverse as in lstm_wide.hip.  dL/dx: every sample
//             feeds four windows (as amplitude into the cell, as cos / sin into the head), also across the circular wrap — the three gradient
//             streams dL/d(a, cos, sin) are accumulated per SAMPLE in LDS over the whole frame and turned into dL/d(I, Q) at the end of the
//             sequence (so dL/dx needs 3 T floats of LDS: frames up to ~5 000 samples; longer ones are refused for that mode).
// Per-step records (i, f, g, o, c, h) in HBM: B x T x 6 x 64 floats.
#include "odpd_seq.h"

namespace odpd {
namespace {
constexpr int kVC = 64, kVS = 65, kVNS = 6, kVW = 68;      // chunk, row stride, record slots, window array length (3 halo + 64 + pad)
constexpr int kVHs = ((kVC + 1) * kVS + 3) & ~3;

__device__ __forceinline__ void vdw_elem(float2 xv, float& a, float& cw, float& sw) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y);
    a = __builtin_amdgcn_sqrtf(a2);
    const float ia = fast_rcp(a);
    cw = xv.x * ia; sw = xv.y * ia;
}
// entry i of the window arrays = time t0 - 3 + i (negative times wrap to the frame's end: circular padding)
__device__ __forceinline__ void vdw_stage(float* av, float* cv, float* sn, const float2* xg, int t0, int T, int lane) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int i = pass * 64 + lane;
        if (i < kVW) {
            int t = t0 - 3 + i;
            if (t < 0) t += T;
            float a = 1.0f, c = 1.0f, s = 0.0f;
            if (t < T) vdw_elem(xg[t], a, c, s);
            av[i] = a; cv[i] = c; sn[i] = s;
        }
    }
}
__host__ __device__ inline int vdw_fwd_floats(int P) { return pad4(P) + 3 * kVW + 64 + kVC * kVS + 64 * kVS; }
__host__ __device__ inline int vdw_bwd_floats(int P, int T, bool dx) {
    return pad4(P) + 3 * kVW + kVC * 2 + kVC * 8 + 4 * 64 + kVHs + kVC * kVS + (dx ? 3 * ((T + 3) & ~3) : 0);
}

template <bool SAVE>
__global__ __launch_bounds__(64) void wide_vdlstm_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const LstmLayout L = lstm_layout(a.H, 1);
    const int H = L.H, T = a.T;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* av = smem + pad4(L.P);              // [68] amplitudes, [68] cos, [68] sin of the chunk's window range
    float* cv = av + kVW;
    float* sn = cv + kVW;
    float* hb = sn + kVW;                      // [64]: the state, for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: h of the chunk's steps
    float* wop = hist + kVC * kVS;             // [64][65]: gate o's W_hh rows, zero padded
    const bool vo = lane < H;
    for (int i = lane; i < 64 * kVS; i += 64) {
        const int j = i / kVS, k = i % kVS;
        wop[i] = (j < H && k < H) ? pl[L.o_w_hh + (3 * H + j) * H + k] : 0.0f;
    }
    float whh[3][64], wih[4][4], bg[4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 64; ++k) whh[g][k] = (vo && k < H) ? pl[L.o_w_hh + (g * H + lane) * H + k] : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wih[g][k] = vo ? pl[L.o_w_ih + (g * H + lane) * 4 + k] : 0.0f;
        bg[g] = vo ? pl[L.o_b_ih + g * H + lane] + pl[L.o_b_hh + g * H + lane] : 0.0f;
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kVNS * 64 : nullptr;
        float h = 0.0f, c = 0.0f;
        for (int t0 = 0; t0 < T; t0 += kVC) {
            const int len = min(kVC, T - t0);
            wave_lds_fence();
            vdw_stage(av, cv, sn, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                hb[lane] = h;
                wave_lds_fence();
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pre[g] = bg[g];
#pragma unroll
                    for (int k = 0; k < 4; ++k) pre[g] = __builtin_fmaf(wih[g][k], av[tt + k], pre[g]);      // |x| of times t - 3 .. t
                }
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
                const float* wo = wop + lane * kVS;
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
                    float* s = sv + (size_t)(t0 + tt) * kVNS * 64 + lane;
                    s[0] = gi; s[64] = gf; s[128] = gg; s[192] = go; s[256] = cn; s[320] = hn;
                }
                c = cn; h = hn;
                hist[tt * kVS + lane] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step
                const float* hr = hist + lane * kVS;
                float l1[4], l2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { l1[k] = pl[L.o_b_l1 + k]; l2[k] = pl[L.o_b_l2 + k]; }
                for (int j = 0; j < H; ++j) {
                    const float hv = hr[j];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        l1[k] = __builtin_fmaf(pl[L.o_w_l1 + k * H + j], hv, l1[k]); l2[k] = __builtin_fmaf(pl[L.o_w_l2 + k * H + j], hv, l2[k]);
                    }
                }
                float y0 = pl[L.o_b_out], y1 = pl[L.o_b_out + 1];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lc = l1[k] * cv[lane + k], ls = l2[k] * sn[lane + k];
                    y0 = __builtin_fmaf(pl[L.o_w_out + k], lc, __builtin_fmaf(pl[L.o_w_out + 4 + k], ls, y0));
                    y1 = __builtin_fmaf(pl[L.o_w_out + 8 + k], lc, __builtin_fmaf(pl[L.o_w_out + 12 + k], ls, y1));
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(64) void wide_vdlstm_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
    const LstmLayout L = lstm_layout(a.H, 1);
    const int H = L.H, T = a.T, NC = (T + kVC - 1) / kVC, Tp = (T + 3) & ~3;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* av = smem + pad4(L.P);
    float* cv = av + kVW;
    float* sn = cv + kVW;
    float* dyb = sn + kVW;                     // [64][2]  dL/dy of the chunk's steps
    float* dlb = dyb + kVC * 2;                // [64][8]  dL/d lambda_1[0..3], lambda_2[0..3] of the chunk's steps
    float* dgb = dlb + kVC * 8;                // [4][64]  the step's gate gradients, for the broadcast reads
    float* hs = dgb + 4 * 64;                  // [65][65] row i = h(t0 - 1 + i)
    float* x1 = hs + kVHs;                     // [64][65] the head's dL/dh of the chunk's steps
    float* gacc = x1 + kVC * kVS;              // DX: [3][Tp] dL/d(a, cos, sin) per sample of the frame
    const bool vo = lane < H;
    float wih[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) wih[g][k] = vo ? pl[L.o_w_ih + (g * H + lane) * 4 + k] : 0.0f;
    f32x16 acc[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    float dwih[4][4], dbs[4] = {0.f, 0.f, 0.f, 0.f}, dwl[8], tacc[26];      // per unit: W_ih rows, gate biases, fc_lambda columns; per time lane: fc_out (16 + 2), fc_lambda biases (8)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) dwih[g][k] = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) dwl[r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 26; ++i) tacc[i] = 0.0f;
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kVNS * 64;
        float dh = 0.0f, dc = 0.0f;
        if constexpr (DX) {
            for (int i = lane; i < 3 * Tp; i += 64) gacc[i] = 0.0f;
        }
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kVC, len = min(kVC, T - t0);
            wave_lds_fence();
            vdw_stage(av, cv, sn, xg, t0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            hs[lane] = t0 > 0 ? sv[(size_t)(t0 - 1) * kVNS * 64 + 320 + lane] : 0.0f;
            for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * kVS + lane] = sv[(size_t)(t0 + tt) * kVNS * 64 + 320 + lane];
            wave_lds_fence();
            // ---- the head's gradients of the chunk, lane = time step ----
            {
                float dl[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (lane < len) {
                    const float* hr = hs + (lane + 1) * kVS;
                    float l1[4], l2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { l1[k] = pl[L.o_b_l1 + k]; l2[k] = pl[L.o_b_l2 + k]; }
                    for (int j = 0; j < H; ++j) {
                        const float hv = hr[j];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            l1[k] = __builtin_fmaf(pl[L.o_w_l1 + k * H + j], hv, l1[k]); l2[k] = __builtin_fmaf(pl[L.o_w_l2 + k * H + j], hv, l2[k]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float cw = cv[lane + k], sw = sn[lane + k];
                        const float dz1 = __builtin_fmaf(dyv.x, pl[L.o_w_out + k], dyv.y * pl[L.o_w_out + 8 + k]);
                        const float dz2 = __builtin_fmaf(dyv.x, pl[L.o_w_out + 4 + k], dyv.y * pl[L.o_w_out + 12 + k]);
                        dl[k] = dz1 * cw; dl[4 + k] = dz2 * sw;
                        if constexpr (NW) {
                            const float lc = l1[k] * cw, ls = l2[k] * sw;
                            tacc[k] = __builtin_fmaf(dyv.x, lc, tacc[k]); tacc[4 + k] = __builtin_fmaf(dyv.x, ls, tacc[4 + k]);
                            tacc[8 + k] = __builtin_fmaf(dyv.y, lc, tacc[8 + k]); tacc[12 + k] = __builtin_fmaf(dyv.y, ls, tacc[12 + k]);
                            tacc[18 + k] += dl[k]; tacc[22 + k] += dl[4 + k];
                        }
                        if constexpr (DX) {      // cos / sin of sample (t - 3 + k) mod T, as the head of step t sees them
                            int ts = t0 + lane - 3 + k;
                            if (ts < 0) ts += T;
                            atomicAdd(gacc + Tp + ts, dz1 * l1[k]);
                            atomicAdd(gacc + 2 * Tp + ts, dz2 * l2[k]);
                        }
                    }
                    if constexpr (NW) { tacc[16] += dyv.x; tacc[17] += dyv.y; }
                }
                reinterpret_cast<float4*>(dlb)[2 * lane] = make_float4(dl[0], dl[1], dl[2], dl[3]);
                reinterpret_cast<float4*>(dlb)[2 * lane + 1] = make_float4(dl[4], dl[5], dl[6], dl[7]);
                // dL/dh(t) from the head: fc_lambda_1^T dl1 + fc_lambda_2^T dl2
                for (int j = 0; j < 64; ++j) {
                    float v = 0.0f;
                    if (j < H) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v = __builtin_fmaf(dl[k], pl[L.o_w_l1 + k * H + j], __builtin_fmaf(dl[4 + k], pl[L.o_w_l2 + k * H + j], v));
                    }
                    x1[lane * kVS + j] = v;
                }
            }
            wave_lds_fence();
            if constexpr (NW) {      // the fc_lambda rows of this lane's unit over the chunk
                for (int tt = 0; tt < len; ++tt) {
                    const float ht = hs[(tt + 1) * kVS + lane];
                    const float4 d0 = reinterpret_cast<const float4*>(dlb)[2 * tt], d1 = reinterpret_cast<const float4*>(dlb)[2 * tt + 1];
                    dwl[0] = __builtin_fmaf(d0.x, ht, dwl[0]); dwl[1] = __builtin_fmaf(d0.y, ht, dwl[1]); dwl[2] = __builtin_fmaf(d0.z, ht, dwl[2]);
                    dwl[3] = __builtin_fmaf(d0.w, ht, dwl[3]); dwl[4] = __builtin_fmaf(d1.x, ht, dwl[4]); dwl[5] = __builtin_fmaf(d1.y, ht, dwl[5]);
                    dwl[6] = __builtin_fmaf(d1.z, ht, dwl[6]); dwl[7] = __builtin_fmaf(d1.w, ht, dwl[7]);
                }
            }
            // ---- the chunk's steps in reverse, lane = unit (the next step's record is in flight while this one is worked on) ----
            float in_, fn_, gn_, on_, cn_, cpn_;
            {
                const float* s = sv + (size_t)(t0 + len - 1) * kVNS * 64 + lane;
                in_ = s[0]; fn_ = s[64]; gn_ = s[128]; on_ = s[192]; cn_ = s[256];
                cpn_ = t0 + len - 1 > 0 ? s[256 - kVNS * 64] : 0.0f;
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float gi = in_, gf = fn_, gg = gn_, go = on_, ct = cn_, cp = cpn_;
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * kVNS * 64 + lane;
                    in_ = s[0]; fn_ = s[64]; gn_ = s[128]; on_ = s[192]; cn_ = s[256];
                    cpn_ = t0 + tt - 1 > 0 ? s[256 - kVNS * 64] : 0.0f;
                }
                const float hp = hs[tt * kVS + lane];
                const float dht = dh + x1[tt * kVS + lane];
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
                const float dp[4] = {dpi, dpf, dpg, dpo};
                if constexpr (NW) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float hpr = rr == 0 ? hp : __shfl(hp, (lane + 16 * rr) & 63);
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpi, hpr, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpf, hpr, acc[1][rr], 0, 0, 0);
                        acc[2][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpg, hpr, acc[2][rr], 0, 0, 0);
                        acc[3][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpo, hpr, acc[3][rr], 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) dwih[g][k] = __builtin_fmaf(dp[g], av[tt + k], dwih[g][k]);
                        dbs[g] += dp[g];
                    }
                }
                if constexpr (DX) {      // amplitude of sample (t - 3 + k) mod T, as the cell of step t sees it
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float v = __builtin_fmaf(dpi, wih[0][k], __builtin_fmaf(dpf, wih[1][k], __builtin_fmaf(dpg, wih[2][k], dpo * wih[3][k])));
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        if (lane == 0) {
                            int ts = t0 + tt - 3 + k;
                            if (ts < 0) ts += T;
                            gacc[ts] += v;
                        }
                    }
                }
                wave_lds_fence();
            }
        }
        if constexpr (DX) {      // dL/d(a, cos, sin) of every sample -> dL/d(I, Q)
            wave_lds_fence();
            for (int t = lane; t < T; t += 64) {
                float a_, cw, sw;
                vdw_elem(xg[t], a_, cw, sw);
                reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t] = polar_sample_bwd(a_, cw, sw, gacc[t], gacc[Tp + t], gacc[2 * Tp + t]);
            }
        }
        wave_lds_fence();
    }
    if constexpr (NW) {
        float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
        for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.0f;
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 26; ++i)
            for (int o = 32; o > 0; o >>= 1) tacc[i] += __shfl_xor(tacc[i], o);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) prow[L.o_w_out + k] = tacc[k];
            prow[L.o_b_out] = tacc[16]; prow[L.o_b_out + 1] = tacc[17];
#pragma unroll
            for (int k = 0; k < 4; ++k) { prow[L.o_b_l1 + k] = tacc[18 + k]; prow[L.o_b_l2 + k] = tacc[22 + k]; }
        }
        if (vo) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { prow[L.o_w_l1 + k * H + lane] = dwl[k]; prow[L.o_w_l2 + k * H + lane] = dwl[4 + k]; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k < 4; ++k) prow[L.o_w_ih + (g * H + lane) * 4 + k] = dwih[g][k];
                prow[L.o_b_ih + g * H + lane] = dbs[g]; prow[L.o_b_hh + g * H + lane] = dbs[g];
            }
        }
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
int vdw_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// float vdlstm of 33 .. 64 hidden units
bool vdlstm_wide_ok(const odpd_model_t* m) { return m->backbone == ODPD_VDLSTM && m->bits_w == 0 && m->hidden > 32 && m->hidden <= 64; }
int64_t vdlstm_wide_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kVNS * 64; }
int vdlstm_wide_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int vdlstm_wide_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!vdlstm_wide_ok(m)) return ODPD_EUNSUPPORTED;
    if (a.T < 3) return ODPD_EINVAL;      // the circular pad takes the frame's own last three samples (vdlstm.py:66-74)
    const size_t lds = (size_t)vdw_fwd_floats(lstm_layout(m->hidden, 1).P) * sizeof(float);
    const int grid = vdlstm_wide_rows(m, a.B);
    return a.ckpt ? vdw_launch(st, wide_vdlstm_fwd_kernel<true>, grid, lds, a) : vdw_launch(st, wide_vdlstm_fwd_kernel<false>, grid, lds, a);
}
int vdlstm_wide_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!vdlstm_wide_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt || a.T < 3) return ODPD_EINVAL;
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    const size_t lds = (size_t)vdw_bwd_floats(lstm_layout(m->hidden, 1).P, a.T, dx) * sizeof(float);
    const int grid = vdlstm_wide_rows(m, a.B);
    if (nw && dx) return vdw_launch(st, wide_vdlstm_bwd_kernel<true, true>, grid, lds, a);
    if (nw) return vdw_launch(st, wide_vdlstm_bwd_kernel<true, false>, grid, lds, a);
    return vdw_launch(st, wide_vdlstm_bwd_kernel<false, true>, grid, lds, a);
}

}  // namespace odpd
