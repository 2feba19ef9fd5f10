// pgjanet_q.hip — `--quant` on pgjanet (reference quant/quant_envs.py:145-148, 285-306 on backbones/pgjanet.py:5-84): the cell's six nn.Linear —
// W_a, W_p1, W_p2 on [h, |x| / cos / sin], W_f, W_g on [h, u], W_o on h — become INT_Linear (quant/qmodules/quant_layers.py:48-85): each
// quantises ITS input on an activation grid of its own and its weight on a weight grid (three scale parameters behind each layer's bias); the
// tanh / sigmoid calls are functional and stay float; no module is named fc_out, so the 16-bit output quantiser never runs.
//
// A plain kernel pair next to the float ones (janet_family.hip, janet_s16.hip, janet_wide.hip): ONE sequence per single-wave workgroup, LANE =
// HIDDEN UNIT (hidden <= 32).  The parameters are staged in LDS with the six weight matrices quantised IN PLACE; per step the lane of unit k
// writes q_l(h_k) for the five gate layers (and q_l(u_k) for W_f, W_g) to LDS, the lane of unit j runs its rows against those broadcasts (row
// reads: lanes stride H + 1 / 2 H floats apart; transposed reads in the backward pass: consecutive lanes, consecutive addresses).  The
// forward pass records (a_n, p1, p2, u, f, g, h') per step in HBM (`ckpt`: B x T x 8 x 64 floats) when a backward pass follows.  Backward:
// reverse steps, the gate gradients broadcast through LDS, the transposed mat-vecs with every layer's activation pass mask, the weight
// gradients accumulated in the registers of the lane that owns the rows (deposited in a second LDS copy of the parameter layout at the end), the weight
// quantisers' pass masks applied at write-out from the unquantised weights, the 18 scale columns exact zeros.
#include "odpd_seq.h"
#include "odpd_quant.h"

#pragma clang fp contract(off)

namespace odpd {
namespace {
constexpr int kPC = 64, kPS = 33, kPNS = 8;
struct PgqLayout { int H, ow[6], ob[6], oq[6], P; };      // 0 W_a  1 W_p1  2 W_p2  3 W_f  4 W_g  5 W_o
__host__ __device__ inline int pgq_nin(int l, int H) { return l < 3 ? H + 1 : (l < 5 ? 2 * H : H); }
__host__ __device__ inline PgqLayout pgq_layout(int H) {
    PgqLayout L; L.H = H; int o = 0;
    for (int l = 0; l < 6; ++l) {
        const int nout = l < 5 ? H : 2;
        L.ow[l] = o; o += nout * pgq_nin(l, H); L.ob[l] = o; o += nout; L.oq[l] = o; o += 3;
    }
    L.P = o;
    return L;
}
__host__ __device__ inline int pgq_fwd_floats(int P) { return pad4(P) + kPC * 4 + 8 * 32 + kPC * kPS; }
__host__ __device__ inline int pgq_bwd_floats(int P) { return 2 * pad4(P) + kPC * 4 + kPC * 2 + kPC * 2 + 16 * 32 + 4; }

// |x|, cos(theta), sin(theta) of the chunk's steps, lane = time step (theta = atan2(Q, I): cos = I / |x|, sin = Q / |x|)
__device__ __forceinline__ void pgq_stage_inputs(float* ftab, const float2* xg, int t0, int T, int lane) {
    const float2 xv = t0 + lane < T ? xg[t0 + lane] : make_float2(0.6f, 0.8f);
    const float am = sqrtf(xv.x * xv.x + xv.y * xv.y);
    reinterpret_cast<float4*>(ftab)[lane] = make_float4(am, xv.x / am, xv.y / am, 0.0f);
}
struct PgqQ { q16::Quant a[6]; };
// stage the parameters, form the activation quantisers, quantise the six weight matrices in the staged copy
__device__ __forceinline__ void pgq_setup(float* pl, const SeqArgs& a, const PgqLayout& L, PgqQ& Q, int lane) {
    stage_params(pl, a.params, L.P);
    wave_lds_fence();
#pragma unroll
    for (int l = 0; l < 6; ++l) Q.a[l] = q16::make_quant(pl[L.oq[l] + 1], a.bits_a);
    q16::Quant qw[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) qw[l] = q16::make_quant(pl[L.oq[l]], a.bits_w);
    wave_lds_fence();
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        const int n = (l < 5 ? L.H : 2) * pgq_nin(l, L.H);
        for (int i = lane; i < n; i += 64) pl[L.ow[l] + i] = q16::qapply(pl[L.ow[l] + i], qw[l]);
    }
    wave_lds_fence();
}

template <bool SAVE>
__global__ __launch_bounds__(64) void pgq_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const PgqLayout L = pgq_layout(a.H);
    const int H = L.H, T = a.T, H1 = H + 1, H2 = 2 * H;
    float* pl = smem;
    float* ftab = smem + pad4(L.P);            // [64][4]: |x|, cos, sin of the chunk's steps
    float* vq = ftab + kPC * 4;                // [8][32]: q_l(h) for l = 0 .. 4, q_3(u), q_4(u)
    float* hist = vq + 8 * 32;                 // [64][33]: h of the chunk's steps
    PgqQ Q;
    pgq_setup(pl, a, L, Q, lane);
    const bool vo = lane < H;
    const int j = vo ? lane : 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kPNS * 64 : nullptr;
        float h = 0.0f;
        for (int t0 = 0; t0 < T; t0 += kPC) {
            const int len = min(kPC, T - t0);
            wave_lds_fence();
            pgq_stage_inputs(ftab, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                if (lane < 32) {
#pragma unroll
                    for (int l = 0; l < 5; ++l) vq[l * 32 + lane] = vo ? q16::qapply(h, Q.a[l]) : 0.0f;
                }
                wave_lds_fence();
                const float4 in = reinterpret_cast<const float4*>(ftab)[tt];
                const float sc[3] = {in.x, in.y, in.z};
                float gate[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const float* wr = pl + L.ow[l] + j * H1;
                    float acc = 0.0f;
#pragma unroll 8
                    for (int k = 0; k < H; ++k) acc = __builtin_fmaf(wr[k], vq[l * 32 + k], acc);
                    acc = __builtin_fmaf(wr[H], q16::qapply(sc[l], Q.a[l]), acc);
                    gate[l] = tanhf_(acc + pl[L.ob[l] + j]);
                }
                const float an = gate[0], p1 = gate[1], p2 = gate[2];
                const float u = (an * p1 * p2) * ((1.0f - an) * (1.0f - p1) * (1.0f - p2));
                if (lane < 32) {
                    vq[5 * 32 + lane] = vo ? q16::qapply(u, Q.a[3]) : 0.0f;
                    vq[6 * 32 + lane] = vo ? q16::qapply(u, Q.a[4]) : 0.0f;
                }
                wave_lds_fence();
                float pre[2];
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const float* wr = pl + L.ow[3 + l] + j * H2;
                    float acc = 0.0f;
#pragma unroll 8
                    for (int k = 0; k < H; ++k) acc = __builtin_fmaf(wr[k], vq[(3 + l) * 32 + k], acc);
#pragma unroll 8
                    for (int k = 0; k < H; ++k) acc = __builtin_fmaf(wr[H + k], vq[(5 + l) * 32 + k], acc);
                    pre[l] = acc + pl[L.ob[3 + l] + j];
                }
                const float f = sigmoidf_(pre[0]), g = tanhf_(pre[1]);
                const float hn = vo ? f * h + (1.0f - f) * g : 0.0f;
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * kPNS * 64 + lane;
                    s[0] = an; s[64] = p1; s[128] = p2; s[192] = u; s[256] = f; s[320] = g; s[384] = hn;
                }
                h = hn;
                if (lane < 32) hist[tt * kPS + lane] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step: W_o on q_5(h)
                const float* hr = hist + lane * kPS;
                float y0 = 0.0f, y1 = 0.0f;
#pragma unroll 8
                for (int k = 0; k < H; ++k) {
                    const float hv = q16::qapply(hr[k], Q.a[5]);
                    y0 = __builtin_fmaf(pl[L.ow[5] + k], hv, y0); y1 = __builtin_fmaf(pl[L.ow[5] + H + k], hv, y1);
                }
                yg[t0 + lane] = make_float2(y0 + pl[L.ob[5]], y1 + pl[L.ob[5] + 1]);
            }
        }
        wave_lds_fence();
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(64) void pgq_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const PgqLayout L = pgq_layout(a.H);
    const int H = L.H, T = a.T, NC = (T + kPC - 1) / kPC, H1 = H + 1, H2 = 2 * H;
    float* pl = smem;
    float* gw = smem + pad4(L.P);              // weight-gradient accumulators in the parameter layout (lane j owns the rows of unit j)
    float* ftab = gw + pad4(L.P);              // [64][4]  |x|, cos, sin of the chunk's steps
    float* dxb = ftab + kPC * 4;               // [64][2]  dL/dx of the chunk's steps
    float* dyb = dxb + kPC * 2;                // [64][2]  dL/dy of the chunk's steps
    float* vb = dyb + kPC * 2;                 // [16][32] 0..4 q_l(h(t-1)), 5 q_3(u), 6 q_4(u), 7 d_f, 8 d_g, 9 d_a, 10 d_p1, 11 d_p2
    PgqQ Q;
    pgq_setup(pl, a, L, Q, lane);
    for (int i = lane; i < pad4(L.P); i += 64) gw[i] = 0.0f;
    const bool vo = lane < H;
    const int j = vo ? lane : 0;
    const float wo0 = vo ? pl[L.ow[5] + j] : 0.0f, wo1 = vo ? pl[L.ow[5] + H + j] : 0.0f;
    float db[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
    // the weight-gradient rows of unit j in registers (padded to 32 columns: the broadcast vectors are zero beyond H); written to `gw` at the end
    float ga[3][33], gfu[2][64];
#pragma unroll
    for (int l = 0; l < 3; ++l)
#pragma unroll
        for (int k = 0; k < 33; ++k) ga[l][k] = 0.0f;
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
        for (int k = 0; k < 64; ++k) gfu[l][k] = 0.0f;
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kPNS * 64;
        float dh = 0.0f;
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kPC, len = min(kPC, T - t0);
            wave_lds_fence();
            pgq_stage_inputs(ftab, xg, t0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            wave_lds_fence();
            for (int tt = len - 1; tt >= 0; --tt) {
                const int t = t0 + tt;
                const float* s = sv + (size_t)t * kPNS * 64 + lane;
                const float an = s[0], p1 = s[64], p2 = s[128], u = s[192], f = s[256], g = s[320], ht = s[384];
                const float hp = t > 0 ? s[384 - kPNS * 64] : 0.0f;
                const float4 in = reinterpret_cast<const float4*>(ftab)[tt];
                const float sc[3] = {in.x, in.y, in.z};
                // the step's quantised layer inputs (what the forward pass multiplied with) for the weight gradients
                if (lane < 32) {
#pragma unroll
                    for (int l = 0; l < 5; ++l) vb[l * 32 + lane] = vo ? q16::qapply(hp, Q.a[l]) : 0.0f;
                    vb[5 * 32 + lane] = vo ? q16::qapply(u, Q.a[3]) : 0.0f;
                    vb[6 * 32 + lane] = vo ? q16::qapply(u, Q.a[4]) : 0.0f;
                }
                // read-out: dL/dh through W_o's activation mask
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                const float hoq = q16::qapply(ht, Q.a[5]);
                dh = __builtin_fmaf(q16::qpass(ht, Q.a[5]), d.x * wo0 + d.y * wo1, dh);
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, hoq, dwo0); dwo1 = __builtin_fmaf(d.y, hoq, dwo1); }
                const float dfp = vo ? (dh * (hp - g)) * (f * (1.0f - f)) : 0.0f;
                const float dgp = vo ? (dh * (1.0f - f)) * (1.0f - g * g) : 0.0f;
                float dhp = dh * f;
                if (lane < 32) { vb[7 * 32 + lane] = dfp; vb[8 * 32 + lane] = dgp; }
                wave_lds_fence();
                // W_f^T d_f, W_g^T d_g: the h part (-> dL/dh(t-1)) and the u part (-> dL/du), each through its layer's activation mask
                float du = 0.0f;
                {
                    float ah[2] = {0.f, 0.f}, au[2] = {0.f, 0.f};
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        const float* wc = pl + L.ow[3 + l] + j;
#pragma unroll 8
                        for (int r = 0; r < H; ++r) {
                            const float dv = vb[(7 + l) * 32 + r];
                            ah[l] = __builtin_fmaf(wc[r * H2], dv, ah[l]); au[l] = __builtin_fmaf(wc[r * H2 + H], dv, au[l]);
                        }
                        dhp = __builtin_fmaf(q16::qpass(hp, Q.a[3 + l]), ah[l], dhp);
                        du = __builtin_fmaf(q16::qpass(u, Q.a[3 + l]), au[l], du);
                    }
                }
                // u = A(a) A(p1) A(p2), A(v) = v (1 - v), A'(v) = 1 - 2 v; through the tanh of the three input gates
                const float Aa = an * (1.0f - an), Ab = p1 * (1.0f - p1), Ac = p2 * (1.0f - p2);
                float dpre[3];
                dpre[0] = vo ? ((du * (1.0f - 2.0f * an)) * (Ab * Ac)) * (1.0f - an * an) : 0.0f;
                dpre[1] = vo ? ((du * (1.0f - 2.0f * p1)) * (Aa * Ac)) * (1.0f - p1 * p1) : 0.0f;
                dpre[2] = vo ? ((du * (1.0f - 2.0f * p2)) * (Aa * Ab)) * (1.0f - p2 * p2) : 0.0f;
                if (lane < 32) {
#pragma unroll
                    for (int l = 0; l < 3; ++l) vb[(9 + l) * 32 + lane] = dpre[l];
                }
                wave_lds_fence();
                float dsc[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    const float* wc = pl + L.ow[l] + j;
                    float ah = 0.0f;
#pragma unroll 8
                    for (int r = 0; r < H; ++r) ah = __builtin_fmaf(wc[r * H1], vb[(9 + l) * 32 + r], ah);
                    dhp = __builtin_fmaf(q16::qpass(hp, Q.a[l]), ah, dhp);
                    if constexpr (DX) {      // the scalar input's column: sum over the units
                        float v = vo ? pl[L.ow[l] + j * H1 + H] * dpre[l] : 0.0f;
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        dsc[l] = v * q16::qpass(sc[l], Q.a[l]);
                    }
                }
                if constexpr (NW) {      // the rows of unit j: d (x) q_l(input) (dpre / dfp / dgp are 0 on lanes without a unit)
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const float4* v4 = reinterpret_cast<const float4*>(vb + l * 32);
#pragma unroll
                        for (int q4 = 0; q4 < 8; ++q4) {
                            const float4 v = v4[q4];
                            ga[l][4 * q4] = __builtin_fmaf(dpre[l], v.x, ga[l][4 * q4]); ga[l][4 * q4 + 1] = __builtin_fmaf(dpre[l], v.y, ga[l][4 * q4 + 1]);
                            ga[l][4 * q4 + 2] = __builtin_fmaf(dpre[l], v.z, ga[l][4 * q4 + 2]); ga[l][4 * q4 + 3] = __builtin_fmaf(dpre[l], v.w, ga[l][4 * q4 + 3]);
                        }
                        ga[l][32] = __builtin_fmaf(dpre[l], q16::qapply(sc[l], Q.a[l]), ga[l][32]);
                        db[l] += dpre[l];
                    }
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        const float dv = l == 0 ? dfp : dgp;
                        const float4* h4 = reinterpret_cast<const float4*>(vb + (3 + l) * 32);
                        const float4* u4 = reinterpret_cast<const float4*>(vb + (5 + l) * 32);
#pragma unroll
                        for (int q4 = 0; q4 < 8; ++q4) {
                            const float4 v = h4[q4], w = u4[q4];
                            gfu[l][4 * q4] = __builtin_fmaf(dv, v.x, gfu[l][4 * q4]); gfu[l][4 * q4 + 1] = __builtin_fmaf(dv, v.y, gfu[l][4 * q4 + 1]);
                            gfu[l][4 * q4 + 2] = __builtin_fmaf(dv, v.z, gfu[l][4 * q4 + 2]); gfu[l][4 * q4 + 3] = __builtin_fmaf(dv, v.w, gfu[l][4 * q4 + 3]);
                            gfu[l][32 + 4 * q4] = __builtin_fmaf(dv, w.x, gfu[l][32 + 4 * q4]); gfu[l][32 + 4 * q4 + 1] = __builtin_fmaf(dv, w.y, gfu[l][32 + 4 * q4 + 1]);
                            gfu[l][32 + 4 * q4 + 2] = __builtin_fmaf(dv, w.z, gfu[l][32 + 4 * q4 + 2]); gfu[l][32 + 4 * q4 + 3] = __builtin_fmaf(dv, w.w, gfu[l][32 + 4 * q4 + 3]);
                        }
                        db[3 + l] += dv;
                    }
                }
                if constexpr (DX) {
                    if (lane == 0) {      // theta = atan2(Q, I): dtheta = -sin dcos + cos dsin; dtheta/dI = -Q / a^2, dtheta/dQ = I / a^2
                        const float am = in.x, ct = in.y, st = in.z, I = ct * am, Qv = st * am, a2 = am * am;
                        const float dth = -st * dsc[1] + ct * dsc[2];
                        reinterpret_cast<float2*>(dxb)[tt] = make_float2(dsc[0] * I / am - dth * Qv / a2, dsc[0] * Qv / am + dth * I / a2);
                    }
                }
                dh = vo ? dhp : 0.0f;
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
        for (int o = 32; o > 0; o >>= 1) { tb0 += __shfl_xor(tb0, o); tb1 += __shfl_xor(tb1, o); }
        if (vo) {
#pragma unroll
            for (int l = 0; l < 3; ++l) {
#pragma unroll
                for (int k = 0; k < 32; ++k) if (k < H) gw[L.ow[l] + j * H1 + k] = ga[l][k];
                gw[L.ow[l] + j * H1 + H] = ga[l][32];
            }
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int k = 0; k < 32; ++k) if (k < H) { gw[L.ow[3 + l] + j * H2 + k] = gfu[l][k]; gw[L.ow[3 + l] + j * H2 + H + k] = gfu[l][32 + k]; }
#pragma unroll
            for (int l = 0; l < 5; ++l) gw[L.ob[l] + j] = db[l];
            gw[L.ow[5] + j] = dwo0; gw[L.ow[5] + H + j] = dwo1;
        }
        if (lane == 0) { gw[L.ob[5]] = tb0; gw[L.ob[5] + 1] = tb1; }
        wave_lds_fence();
        // weight quantisers' pass masks from the unquantised weights; scale columns exact zeros
        q16::Quant qw[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) qw[l] = q16::make_quant(a.params[L.oq[l]], a.bits_w);
        for (int i = lane; i < L.P + kLossCols; i += 64) {
            float v = i < L.P ? gw[i] : 0.0f;
#pragma unroll
            for (int l = 0; l < 6; ++l) {
                if (i >= L.ow[l] && i < L.ob[l]) v *= q16::qpass(a.params[i], qw[l]);
                if (i >= L.oq[l] && i < L.oq[l] + 3) v = 0.0f;
            }
            prow[i] = v;
        }
    }
}

template <typename K> int pgq_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

bool pgjanet_q_ok(const odpd_model_t* m) {
    return m->backbone == ODPD_PGJANET && m->bits_w > 0 && m->bits_w <= 16 && m->bits_a > 0 && m->bits_a <= 16 && m->hidden >= 1 && m->hidden <= 32 &&
           !(m->flags & ODPD_FLAG_TWO_LAYERS);
}
int64_t pgjanet_q_param_count(const odpd_model_t* m) { return pgq_layout(m->hidden).P; }
int64_t pgjanet_q_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kPNS * 64; }
int pgjanet_q_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int pgjanet_q_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!pgjanet_q_ok(m)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)pgq_fwd_floats(pgq_layout(m->hidden).P) * sizeof(float);
    const int grid = pgjanet_q_rows(m, a.B);
    return a.ckpt ? pgq_launch(st, pgq_fwd_kernel<true>, grid, lds, a) : pgq_launch(st, pgq_fwd_kernel<false>, grid, lds, a);
}
int pgjanet_q_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!pgjanet_q_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)pgq_bwd_floats(pgq_layout(m->hidden).P) * sizeof(float);
    const int grid = pgjanet_q_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return pgq_launch(st, pgq_bwd_kernel<true, true>, grid, lds, a);
    if (nw) return pgq_launch(st, pgq_bwd_kernel<true, false>, grid, lds, a);
    return pgq_launch(st, pgq_bwd_kernel<false, true>, grid, lds, a);
}

}  // namespace odpd
