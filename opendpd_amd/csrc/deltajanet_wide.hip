// deltajanet_wide.hip — DeltaJANET (backbones/deltajanet.py:11-274) with 33 .. 64 hidden units on delta_wide.hip's mapping (one sequence per single-wave
// workgroup, LANE = HIDDEN UNIT): a two-gate delta cell — f and g both sigmoids, h = (1 - f) g + f h, accumulators dm_f, dm_g started at b_ih + b_hh and
// updated as dm = (W_ih dx + dm) + W_hh dh (deltajanet.py:198-206, 229-251).  The wrapper builds its layer with thx = thh = 0 whatever it is given
// (:23-27): no delta is ever masked, the reference values x_p / h_p move every step, and the counters record exact repeats only.
//   records   f, g, h and the state delta of every step, and (lanes 0..5 of a fifth slot) the feature deltas: B x T x 5 x 64 floats in `ckpt`.
//   backward  carried accumulator gradients G_f, G_g; with every delta kept, the reference-value gradients reduce to G_hp(t) = -W_hh^T G(t+1) and
//             G_xp(t) = -W_ih^T G(t+1) (the oracle's dj_seq_bwd); dW_hh as rotated 4-block MFMA outer products against the state delta.
//   --quant   (QH; bits_w > 0, EVERY hidden size up to 64 — the tile kernels of delta_family.hip / delta_s16.hip serve float models only): the
//             surgery finds one nn.Linear in this backbone, fc_out (the gates are nn.Parameter tensors, deltajanet.py:100-113), and makes it an
//             INT_Linear (quant_layers.py:48-85): y = q_w(W) q_a(h) + b, three scale parameters behind fc_out.bias (gradients exactly 0),
//             ODPD_FLAG_EVAL: the 16-bit output grid.  The recurrent cell stays float.
#include "odpd_seq.h"
#include "odpd_delta.h"
#include "odpd_quant.h"

namespace odpd {
namespace {
constexpr int kQC = 64, kQS = 65, kQNS = 5;
constexpr int kQHs = ((kQC + 1) * kQS + 3) & ~3;

__host__ __device__ inline int dj_fwd_floats(int P) { return pad4(P) + kQC * 8 + 64 + kQC * kQS; }
__host__ __device__ inline int dj_bwd_floats(int P) { return pad4(P) + kQC * 8 + kQC * 2 + kQC * 2 + 2 * 64 + 64 + kQHs; }
__device__ __forceinline__ void dj_stage_features(float* ftab, const float2* xg, int t0, int T, int lane) {
    const int t = t0 + lane;
    float f[6] = {0.5f, 0.5f, 0.7f, 0.35f, 0.7f, 0.7f};
    if (t < T) delta_feat<false>(xg[t], xg[t], f);
    reinterpret_cast<float4*>(ftab)[2 * lane] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4*>(ftab)[2 * lane + 1] = make_float4(f[4], f[5], 0.0f, 0.0f);
}

template <bool SAVE, bool QH>
__global__ __launch_bounds__(64) void wide_deltajanet_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const DeltaLayout L = delta_layout(a.H, 0, 2);
    const int H = L.H, T = a.T, P = L.P + (QH ? 3 : 0);
    float* pl = smem;
    stage_params(pl, a.params, P);
    float* ftab = smem + pad4(P);              // [64][8]: features of the chunk's steps
    float* hb = ftab + kQC * 8;                // [64]: the state deltas, for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: h of the chunk's steps
    const bool vo = lane < H;
    float whh[2][64], wih[2][6], dm0[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int k = 0; k < 64; ++k) whh[g][k] = (vo && k < H) ? pl[L.o_w_hh + (g * H + lane) * H + k] : 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) wih[g][i] = vo ? pl[L.o_w_ih + (g * H + lane) * 6 + i] : 0.0f;
        dm0[g] = vo ? pl[L.o_b_ih + g * H + lane] + pl[L.o_b_hh + g * H + lane] : 0.0f;
    }
    q16::Quant qa{}, qo{};
    if constexpr (QH) {      // fc_out's weights become their quantised values in the staged copy
        const q16::Quant qw = q16::make_quant(pl[L.P], a.bits_w);
        qa = q16::make_quant(pl[L.P + 1], a.bits_a);
        qo = q16::make_quant(pl[L.P + 2], 16);
        wave_lds_fence();
        for (int i = lane; i < 2 * H; i += 64) pl[L.o_w_out + i] = q16::qapply(pl[L.o_w_out + i], qw);
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kQNS * 64 : nullptr;
        float h = 0.0f, hp = 0.0f, xp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dmf = dm0[0], dmg = dm0[1], zx = 0.0f, zh = 0.0f;
        for (int t0 = 0; t0 < T; t0 += kQC) {
            const int len = min(kQC, T - t0);
            wave_lds_fence();
            dj_stage_features(ftab, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 1];
                const float fe[6] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y};
                float dxm[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) { dxm[i] = fe[i] - xp[i]; zx += dxm[i] == 0.0f ? 1.0f : 0.0f; xp[i] = fe[i]; }
                const float dhm = vo ? h - hp : 0.0f;
                zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
                hp = h;
                hb[lane] = dhm;
                wave_lds_fence();
                float ax[2] = {0.f, 0.f}, bh[2] = {0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int i = 0; i < 6; ++i) ax[g] = __builtin_fmaf(wih[g][i], dxm[i], ax[g]);
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 hv = hb4[q];
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        bh[g] = __builtin_fmaf(whh[g][4 * q], hv.x, bh[g]); bh[g] = __builtin_fmaf(whh[g][4 * q + 1], hv.y, bh[g]);
                        bh[g] = __builtin_fmaf(whh[g][4 * q + 2], hv.z, bh[g]); bh[g] = __builtin_fmaf(whh[g][4 * q + 3], hv.w, bh[g]);
                    }
                }
                dmf = (ax[0] + dmf) + bh[0]; dmg = (ax[1] + dmg) + bh[1];
                const float fg = sigmoidf_(dmf), gg = sigmoidf_(dmg);
                const float hn = vo ? __builtin_fmaf(fg, h - gg, gg) : 0.0f;       // (1 - f) g + f h
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * kQNS * 64 + lane;
                    s[0] = fg; s[64] = gg; s[128] = hn; s[192] = dhm;
                    float aux = 0.0f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) aux = lane == i ? dxm[i] : aux;
                    s[256] = aux;
                }
                h = hn;
                hist[tt * kQS + lane] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step
                const float* hr = hist + lane * kQS;
                float y0 = QH ? 0.0f : pl[L.o_b_out], y1 = QH ? 0.0f : pl[L.o_b_out + 1];
                for (int j = 0; j < H; ++j) {
                    const float hv = QH ? q16::qapply(hr[j], qa) : hr[j];
                    y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + H + j], hv, y1);
                }
                if constexpr (QH) {      // grid sums first, then the float bias (INT_Linear: F.linear(q_a(x), q_w(W), b))
                    y0 += pl[L.o_b_out]; y1 += pl[L.o_b_out + 1];
                    if (a.eval_out) { y0 = q16::qapply(y0, qo); y1 = q16::qapply(y1, qo); }
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        if (a.stats != nullptr) {
            for (int o = 32; o > 0; o >>= 1) zh += __shfl_xor(zh, o);
            if (lane == 0) {
                atomicAdd(&a.stats[0], (double)zx); atomicAdd(&a.stats[1], 6.0 * (double)T);
                atomicAdd(&a.stats[2], (double)zh); atomicAdd(&a.stats[3], (double)H * (double)T);
            }
        }
        wave_lds_fence();
    }
}

template <bool NW, bool DX, bool QH>
__global__ __launch_bounds__(64) void wide_deltajanet_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
    const DeltaLayout L = delta_layout(a.H, 0, 2);
    const int H = L.H, T = a.T, NC = (T + kQC - 1) / kQC, P = L.P + (QH ? 3 : 0);
    float* pl = smem;
    stage_params(pl, a.params, P);
    float* ftab = smem + pad4(P);              // [64][8]  features of the chunk's steps (for dL/dx)
    float* dxb = ftab + kQC * 8;               // [64][2]  dL/dx of the chunk's steps
    float* dyb = dxb + kQC * 2;                // [64][2]  dL/dy of the chunk's steps
    float* dgb = dyb + kQC * 2;                // [2][64]  the step's G_f, G_g, for the broadcast reads
    float* auxb = dgb + 2 * 64;                // [64]     the step's feature deltas (0..5)
    float* hs = auxb + 64;                     // [65][65] row i = h(t0 - 1 + i)
    const bool vo = lane < H;
    float wih[2][6];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 6; ++i) wih[g][i] = vo ? pl[L.o_w_ih + (g * H + lane) * 6 + i] : 0.0f;
    float wo0 = vo ? pl[L.o_w_out + lane] : 0.0f, wo1 = vo ? pl[L.o_w_out + H + lane] : 0.0f, wm0 = 1.0f, wm1 = 1.0f;
    q16::Quant qa{};
    if constexpr (QH) {      // the head columns as quantised values, their straight-through masks for dW_out
        const q16::Quant qw = q16::make_quant(pl[L.P], a.bits_w);
        qa = q16::make_quant(pl[L.P + 1], a.bits_a);
        wm0 = q16::qpass(wo0, qw); wm1 = q16::qpass(wo1, qw);
        wo0 = q16::qapply(wo0, qw); wo1 = q16::qapply(wo1, qw);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    float dwih[2][6], dbs[2] = {0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < 6; ++i) dwih[g][i] = 0.0f;
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kQNS * 64;
        float Gh = 0.0f, Ghp = 0.0f, Gf = 0.0f, Gg = 0.0f, Gxp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kQC, len = min(kQC, T - t0);
            wave_lds_fence();
            dj_stage_features(ftab, xg, t0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            hs[lane] = t0 > 0 ? sv[(size_t)(t0 - 1) * kQNS * 64 + 128 + lane] : 0.0f;
            for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * kQS + lane] = sv[(size_t)(t0 + tt) * kQNS * 64 + 128 + lane];
            wave_lds_fence();
            float fn_, gn_, dn_, an_;
            {
                const float* s = sv + (size_t)(t0 + len - 1) * kQNS * 64 + lane;
                fn_ = s[0]; gn_ = s[64]; dn_ = s[192]; an_ = s[256];
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float fg = fn_, gg = gn_, dhm = dn_, aux = an_;
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * kQNS * 64 + lane;
                    fn_ = s[0]; gn_ = s[64]; dn_ = s[192]; an_ = s[256];
                }
                const float hprev = hs[tt * kQS + lane], ht = hs[(tt + 1) * kQS + lane];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                if constexpr (QH) {      // q_a(h) feeds the head; dL/dh passes where h lies inside the activation grid
                    const float hq = q16::qapply(ht, qa);
                    Gh = __builtin_fmaf(q16::qpass(ht, qa), __builtin_fmaf(d.x, wo0, d.y * wo1), Gh);
                    if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, hq, dwo0); dwo1 = __builtin_fmaf(d.y, hq, dwo1); }
                } else {
                    Gh = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, Gh));
                    if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, ht, dwo0); dwo1 = __builtin_fmaf(d.y, ht, dwo1); }
                }
                const float dg = Gh * (1.0f - fg), df = Gh * (hprev - gg);
                Gf += vo ? df * (fg * (1.0f - fg)) : 0.0f;
                Gg += vo ? dg * (gg * (1.0f - gg)) : 0.0f;
                auxb[lane] = aux;
                dgb[lane] = Gf; dgb[64 + lane] = Gg;
                wave_lds_fence();
                float ddh = 0.0f;
                {
                    const float* w0 = pl + L.o_w_hh + (vo ? lane : 0);
                    const int HH = H * H;
                    for (int j4 = 0; j4 < H; j4 += 4) {
                        const float4 g0 = *reinterpret_cast<const float4*>(dgb + j4), g1 = *reinterpret_cast<const float4*>(dgb + 64 + j4);
                        const float v0[4] = {g0.x, g0.y, g0.z, g0.w}, v1[4] = {g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* wr = w0 + min(j4 + e, H - 1) * H;
                            ddh = __builtin_fmaf(v0[e], wr[0], ddh); ddh = __builtin_fmaf(v1[e], wr[HH], ddh);
                        }
                    }
                }
                if constexpr (NW) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float dr_ = rr == 0 ? dhm : __shfl(dhm, (lane + 16 * rr) & 63);
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(Gf, dr_, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(Gg, dr_, acc[1][rr], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const float dxm = auxb[i];
                        dwih[0][i] = __builtin_fmaf(Gf, dxm, dwih[0][i]); dwih[1][i] = __builtin_fmaf(Gg, dxm, dwih[1][i]);
                    }
                }
                if constexpr (DX) {      // dx_t = f_t - f_{t-1}: dL/df_t = ddx_t - ddx_{t+1} (G_xp carries -ddx_{t+1})
                    float df6[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        float v = __builtin_fmaf(Gf, wih[0][i], Gg * wih[1][i]);
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        df6[i] = v + Gxp[i];
                        Gxp[i] = -v;
                    }
                    if (lane == 0) {
                        const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt];
                        const float I = f0.x, Q = f0.y, a2 = __builtin_fmaf(I, I, Q * Q), am = f0.z, ia = fast_rcp(am), ia2 = fast_rcp(a2);
                        const float da = df6[2] + 3.0f * a2 * df6[3] - (Q * ia2) * df6[4] - (I * ia2) * df6[5];
                        reinterpret_cast<float2*>(dxb)[tt] = make_float2(df6[0] + df6[5] * ia + da * I * ia, df6[1] + df6[4] * ia + da * Q * ia);
                    }
                }
                const float Gnext = Gh * fg + ddh + Ghp;
                Ghp = vo ? -ddh : 0.0f;
                Gh = vo ? Gnext : 0.0f;
                wave_lds_fence();
            }
            if constexpr (DX) {
                wave_lds_fence();
                if (lane < len) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t0 + lane] = reinterpret_cast<const float2*>(dxb)[lane];
            }
        }
        if constexpr (NW) { dbs[0] += Gf; dbs[1] += Gg; }      // the accumulators started at the biases
        wave_lds_fence();
    }
    if constexpr (NW) {
        float* prow = a.partials + (size_t)blockIdx.x * (P + kLossCols);      // (the scale parameters' columns stay 0: round() inside the quantiser)
        for (int i = lane; i < P + kLossCols; i += 64) prow[i] = 0.0f;
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_fence();
        for (int o = 32; o > 0; o >>= 1) { tb0 += __shfl_xor(tb0, o); tb1 += __shfl_xor(tb1, o); }
        if (lane == 0) { prow[L.o_b_out] = tb0; prow[L.o_b_out + 1] = tb1; }
        if (vo) {
            prow[L.o_w_out + lane] = dwo0 * wm0; prow[L.o_w_out + H + lane] = dwo1 * wm1;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int i = 0; i < 6; ++i) prow[L.o_w_ih + (g * H + lane) * 6 + i] = dwih[g][i];
                prow[L.o_b_ih + g * H + lane] = dbs[g]; prow[L.o_b_hh + g * H + lane] = dbs[g];
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
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
int dj_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// deltajanet of 33 .. 64 hidden units; with a quantised head (bits_w > 0): every hidden size up to 64
bool deltajanet_wide_ok(const odpd_model_t* m) {
    if (m->backbone != ODPD_DELTAJANET || m->hidden > 64 || (m->flags & ODPD_FLAG_TWO_LAYERS)) return false;
    if (m->bits_w > 0) return m->bits_w <= 16 && m->bits_a > 0 && m->bits_a <= 16;
    return m->hidden > 32;
}
static int dj_params(const odpd_model_t* m) { return delta_layout(m->hidden, 0, 2).P + (m->bits_w > 0 ? 3 : 0); }
int64_t deltajanet_wide_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kQNS * 64; }
int deltajanet_wide_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int deltajanet_wide_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!deltajanet_wide_ok(m)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)dj_fwd_floats(dj_params(m)) * sizeof(float);
    const int grid = deltajanet_wide_rows(m, a.B);
    if (m->bits_w > 0)
        return a.ckpt ? dj_launch(st, wide_deltajanet_fwd_kernel<true, true>, grid, lds, a) : dj_launch(st, wide_deltajanet_fwd_kernel<false, true>, grid, lds, a);
    return a.ckpt ? dj_launch(st, wide_deltajanet_fwd_kernel<true, false>, grid, lds, a) : dj_launch(st, wide_deltajanet_fwd_kernel<false, false>, grid, lds, a);
}
int deltajanet_wide_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!deltajanet_wide_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)dj_bwd_floats(dj_params(m)) * sizeof(float);
    const int grid = deltajanet_wide_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (m->bits_w > 0) {
        if (nw && dx) return dj_launch(st, wide_deltajanet_bwd_kernel<true, true, true>, grid, lds, a);
        if (nw) return dj_launch(st, wide_deltajanet_bwd_kernel<true, false, true>, grid, lds, a);
        return dj_launch(st, wide_deltajanet_bwd_kernel<false, true, true>, grid, lds, a);
    }
    if (nw && dx) return dj_launch(st, wide_deltajanet_bwd_kernel<true, true, false>, grid, lds, a);
    if (nw) return dj_launch(st, wide_deltajanet_bwd_kernel<true, false, false>, grid, lds, a);
    return dj_launch(st, wide_deltajanet_bwd_kernel<false, true, false>, grid, lds, a);
}

}  // namespace odpd
