// delta_wide.hip — deltagru / deltagru_tcnskip (TRes-DeltaGRU) with 33 .. 64 hidden units (backbones/deltagru.py:10-276, deltagru_tcnskip.py:11-304):
// gru_wide.hip's mapping — ONE sequence per single-wave workgroup, LANE = HIDDEN UNIT — around the delta cell:
//   forward   per step the six feature deltas (wave-uniform: every lane keeps x_p) and the lane's state delta are thresholded (|d| < th -> 0, the
//             reference value x_p / h_p moves only when the delta passes), the masked state deltas are broadcast through LDS and multiplied
//             with the lane's W_hh rows (registers), and the four accumulators of the unit (dm_r, dm_z, dm_n, dm_nh) take the sums in the
//             reference's order: dm = (W_ih dx + dm) + W_hh dh (deltagru.py:198-206).  fc_out — and TRes's non-causal TCN skip (taps t - 16, t,
//             t + 16) — with lane = time step on 64-step chunks.  The sparsity counters are summed per sequence and added to `stats` once.
//   records   r, z, n, dm_nh, h, the masked state delta, its mask, and (lanes 0..5 / 8..13 of an eighth slot) the masked feature deltas and their
//             masks of every step: B x T x 8 x 64 floats in `ckpt`.
//   backward  reverse steps with the carried accumulator gradients (G_r, G_z, G_n, G_nh) and the carried reference-value gradients (G_hp per
//             lane, G_xp wave-uniform): one transposed mat-vec per step for W_hh^T (G_r, G_z, G_nh), dW_hh as rotated 4-block MFMA outer
//             products against the masked state delta, dW_ih on the VALU.  dL/dx: dL/d(features) per step and the TCN's tap gradients are
//             collected per SAMPLE in LDS over the frame (8 T floats) and turned into dL/d(I, Q) at the end of the sequence.
#include "odpd_seq.h"
#include "odpd_delta.h"

namespace odpd {
namespace {
constexpr int kDC = 64, kDS = 65, kDNS = 8;
constexpr int kDHs = ((kDC + 1) * kDS + 3) & ~3;

__host__ __device__ inline int dw_fwd_floats(int P) { return pad4(P) + kDC * 8 + 64 + kDC * kDS; }
__host__ __device__ inline int dw_bwd_floats(int P, int T, bool dx) {
    return pad4(P) + kDC * 2 + 4 * 64 + 64 + kDHs + (dx ? 8 * ((T + 3) & ~3) : 0);
}
template <bool TRES>
__device__ __forceinline__ void dw_stage_features(float* ftab, const float2* xg, int t0, int T, int lane) {
    const int t = t0 + lane;
    float f[6] = {0.5f, 0.5f, 0.7f, 0.35f, 0.7f, 0.7f};
    if (t < T) delta_feat<TRES>(xg[t], xg[t + 1 < T ? t + 1 : 0], f);            // torch.roll(x, -1): the last step sees sample 0
    reinterpret_cast<float4*>(ftab)[2 * lane] = make_float4(f[0], f[1], f[2], f[3]);
    reinterpret_cast<float4*>(ftab)[2 * lane + 1] = make_float4(f[4], f[5], 0.0f, 0.0f);
}
// TCN skip of one time step (deltagru_tcnskip.py:32-49): conv(2 -> 3, k 3, dilation 16, zero padded), Hardswish, conv(3 -> 2, k 1), Hardswish
__device__ __forceinline__ void dw_tcn(const float* pl, const DeltaLayout& L, const float2* xg, int t, int T, float (&s1)[3], float (&s2)[2], float2 (&tap)[3]) {
    const float2 zero = make_float2(0.0f, 0.0f);
    tap[0] = zero; tap[1] = xg[t]; tap[2] = zero;      // (`if`, not `?:` on the loads: see stage_in_ch, odpd_seq.h)
    if (t - kDHalo >= 0) tap[0] = xg[t - kDHalo];
    if (t + kDHalo < T) tap[2] = xg[t + kDHalo];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) a = __builtin_fmaf(pl[L.o_tcn0 + (c * 2 + 0) * 3 + k], tap[k].x, a);
#pragma unroll
        for (int k = 0; k < 3; ++k) a = __builtin_fmaf(pl[L.o_tcn0 + (c * 2 + 1) * 3 + k], tap[k].y, a);
        s1[c] = a;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) a = __builtin_fmaf(pl[L.o_tcn2 + o * 3 + c], hardswishf_(s1[c]), a);
        s2[o] = a;
    }
}

template <bool TRES, bool SAVE>
__global__ __launch_bounds__(64) void wide_delta_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const DeltaLayout L = delta_layout(a.H, TRES);
    const int H = L.H, T = a.T;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [64][8]: features of the chunk's steps
    float* hb = ftab + kDC * 8;                // [64]: the masked state deltas, for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: h of the chunk's steps
    const bool vo = lane < H;
    float whh[3][64], wih[3][6], dm0[4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int k = 0; k < 64; ++k) whh[g][k] = (vo && k < H) ? pl[L.o_w_hh + (g * H + lane) * H + k] : 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) wih[g][i] = vo ? pl[L.o_w_ih + (g * H + lane) * 6 + i] : 0.0f;
    }
    // accumulators start at the biases (deltagru.py:165-170), at zero for the bias-free TRes layer
    dm0[0] = (!TRES && vo) ? pl[L.o_b_ih + lane] + pl[L.o_b_hh + lane] : 0.0f;
    dm0[1] = (!TRES && vo) ? pl[L.o_b_ih + H + lane] + pl[L.o_b_hh + H + lane] : 0.0f;
    dm0[2] = (!TRES && vo) ? pl[L.o_b_ih + 2 * H + lane] : 0.0f;
    dm0[3] = (!TRES && vo) ? pl[L.o_b_hh + 2 * H + lane] : 0.0f;
    const float thx = a.thx, thh = a.thh;
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kDNS * 64 : nullptr;
        float h = 0.0f, hp = 0.0f, xp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float dmr = dm0[0], dmz = dm0[1], dmn = dm0[2], dmnh = dm0[3], zx = 0.0f, zh = 0.0f;
        for (int t0 = 0; t0 < T; t0 += kDC) {
            const int len = min(kDC, T - t0);
            wave_lds_fence();
            dw_stage_features<TRES>(ftab, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 1];
                const float fe[6] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y};
                float dxm[6], mxv[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) {      // wave-uniform: every lane runs the same compare on the same values
                    const float d = fe[i] - xp[i], ad = __builtin_fabsf(d);
                    mxv[i] = ad < thx ? 0.0f : 1.0f;
                    dxm[i] = mxv[i] != 0.0f ? d : 0.0f;
                    zx += dxm[i] == 0.0f ? 1.0f : 0.0f;
                    xp[i] = ad >= thx ? fe[i] : xp[i];
                }
                const float dlt = h - hp, adh = __builtin_fabsf(dlt);
                const float mh = adh < thh ? 0.0f : 1.0f;
                const float dhm = (vo && mh != 0.0f) ? dlt : 0.0f;
                zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
                hp = adh >= thh ? h : hp;
                hb[lane] = dhm;
                wave_lds_fence();
                float ax[3] = {0.f, 0.f, 0.f}, bh[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < 6; ++i) ax[g] = __builtin_fmaf(wih[g][i], dxm[i], ax[g]);
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 hv = hb4[q];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        bh[g] = __builtin_fmaf(whh[g][4 * q], hv.x, bh[g]); bh[g] = __builtin_fmaf(whh[g][4 * q + 1], hv.y, bh[g]);
                        bh[g] = __builtin_fmaf(whh[g][4 * q + 2], hv.z, bh[g]); bh[g] = __builtin_fmaf(whh[g][4 * q + 3], hv.w, bh[g]);
                    }
                }
                // dm = (W_ih dx + dm) + W_hh dh; the n gate keeps its hidden part in dm_nh (deltagru.py:198-206, 246-251)
                dmr = (ax[0] + dmr) + bh[0]; dmz = (ax[1] + dmz) + bh[1]; dmn = ax[2] + dmn; dmnh = bh[2] + dmnh;
                const float r = sigmoidf_(dmr), z = sigmoidf_(dmz);
                const float n = tanhf_(__builtin_fmaf(r, dmnh, dmn));
                const float hn = vo ? __builtin_fmaf(z, h - n, n) : 0.0f;
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * kDNS * 64 + lane;
                    s[0] = r; s[64] = z; s[128] = n; s[192] = dmnh; s[256] = hn; s[320] = dhm; s[384] = mh;
                    float aux = 0.0f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) aux = lane == i ? dxm[i] : (lane == 8 + i ? mxv[i] : aux);
                    s[448] = aux;
                }
                h = hn;
                hist[tt * kDS + lane] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step
                const float* hr = hist + lane * kDS;
                float y0 = TRES ? 0.0f : pl[L.o_b_out], y1 = TRES ? 0.0f : pl[L.o_b_out + 1];
                for (int j = 0; j < H; ++j) {
                    const float hv = hr[j];
                    y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + H + j], hv, y1);
                }
                if constexpr (TRES) {
                    float s1[3], s2[2];
                    float2 tap[3];
                    dw_tcn(pl, L, xg, t0 + lane, T, s1, s2, tap);
                    y0 += hardswishf_(s2[0]); y1 += hardswishf_(s2[1]);
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        if (a.stats != nullptr) {      // num_dx_zeros, num_dx_numel, num_dh_zeros, num_dh_numel of this sequence (deltagru.py:179-192)
            for (int o = 32; o > 0; o >>= 1) zh += __shfl_xor(zh, o);
            if (lane == 0) {
                atomicAdd(&a.stats[0], (double)zx); atomicAdd(&a.stats[1], 6.0 * (double)T);
                atomicAdd(&a.stats[2], (double)zh); atomicAdd(&a.stats[3], (double)H * (double)T);
            }
        }
        wave_lds_fence();
    }
}

template <bool TRES, bool NW, bool DX>
__global__ __launch_bounds__(64) void wide_delta_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
    const DeltaLayout L = delta_layout(a.H, TRES);
    const int H = L.H, T = a.T, NC = (T + kDC - 1) / kDC, Tp = (T + 3) & ~3;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* dyb = smem + pad4(L.P);             // [64][2]  dL/dy of the chunk's steps
    float* dgb = dyb + kDC * 2;                // [4][64]  the step's G_r, G_z, G_nh, for the broadcast reads
    float* auxb = dgb + 4 * 64;                // [64]     the step's masked feature deltas (0..5) and their masks (8..13)
    float* hs = auxb + 64;                     // [65][65] row i = h(t0 - 1 + i)
    float* dff = hs + kDHs;                    // DX: [6][Tp] dL/d(features) per step; [2][Tp] the TCN's dL/dx per sample
    const bool vo = lane < H;
    float wih[3][6];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 6; ++i) wih[g][i] = vo ? pl[L.o_w_ih + (g * H + lane) * 6 + i] : 0.0f;
    const float wo0 = vo ? pl[L.o_w_out + lane] : 0.0f, wo1 = vo ? pl[L.o_w_out + H + lane] : 0.0f;
    f32x16 acc[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    float dwih[3][6], dbs[4] = {0.f, 0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tacc[26];      // tacc: fc_out bias (2) | TCN conv1 (18), conv2 (6), per time lane
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 6; ++i) dwih[g][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 26; ++i) tacc[i] = 0.0f;
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kDNS * 64;
        float Gh = 0.0f, Ghp = 0.0f, Gr = 0.0f, Gz = 0.0f, Gn = 0.0f, Gnh = 0.0f, Gxp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (DX) {
            for (int i = lane; i < 8 * Tp; i += 64) dff[i] = 0.0f;
        }
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kDC, len = min(kDC, T - t0);
            wave_lds_fence();
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW && !TRES) { tacc[0] += dyv.x; tacc[1] += dyv.y; }
            hs[lane] = t0 > 0 ? sv[(size_t)(t0 - 1) * kDNS * 64 + 256 + lane] : 0.0f;
            for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * kDS + lane] = sv[(size_t)(t0 + tt) * kDNS * 64 + 256 + lane];
            if constexpr (TRES) {      // the TCN skip's gradients, lane = time step (state-free)
                if (lane < len) {
                    const int t = t0 + lane;
                    float s1[3], s2[2];
                    float2 tap[3];
                    dw_tcn(pl, L, xg, t, T, s1, s2, tap);
                    const float d2[2] = {dyv.x * hswish_grad_(s2[0]), dyv.y * hswish_grad_(s2[1])};
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) {
                        const float hv = hardswishf_(s1[cc]);
                        const float d1 = __builtin_fmaf(d2[0], pl[L.o_tcn2 + cc], d2[1] * pl[L.o_tcn2 + 3 + cc]) * hswish_grad_(s1[cc]);
                        if constexpr (NW) {
                            tacc[20 + cc] = __builtin_fmaf(d2[0], hv, tacc[20 + cc]); tacc[23 + cc] = __builtin_fmaf(d2[1], hv, tacc[23 + cc]);
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                tacc[2 + (cc * 2 + 0) * 3 + k] = __builtin_fmaf(d1, tap[k].x, tacc[2 + (cc * 2 + 0) * 3 + k]);
                                tacc[2 + (cc * 2 + 1) * 3 + k] = __builtin_fmaf(d1, tap[k].y, tacc[2 + (cc * 2 + 1) * 3 + k]);
                            }
                        }
                        if constexpr (DX) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const int ts = t + kDHalo * (k - 1);
                                if (ts >= 0 && ts < T) {
                                    atomicAdd(dff + 6 * Tp + ts, d1 * pl[L.o_tcn0 + (cc * 2 + 0) * 3 + k]);
                                    atomicAdd(dff + 7 * Tp + ts, d1 * pl[L.o_tcn0 + (cc * 2 + 1) * 3 + k]);
                                }
                            }
                        }
                    }
                }
            }
            wave_lds_fence();
            // ---- the chunk's steps in reverse, lane = unit (the next step's record is in flight while this one is worked on) ----
            float rn, zn, nn, qn, dn_, mn_, an_;
            {
                const float* s = sv + (size_t)(t0 + len - 1) * kDNS * 64 + lane;
                rn = s[0]; zn = s[64]; nn = s[128]; qn = s[192]; dn_ = s[320]; mn_ = s[384]; an_ = s[448];
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float r = rn, z = zn, n = nn, dmnh = qn, dhm = dn_, mh = mn_, aux = an_;
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * kDNS * 64 + lane;
                    rn = s[0]; zn = s[64]; nn = s[128]; qn = s[192]; dn_ = s[320]; mn_ = s[384]; an_ = s[448];
                }
                const float hprev = hs[tt * kDS + lane], ht = hs[(tt + 1) * kDS + lane];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                Gh = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, Gh));
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, ht, dwo0); dwo1 = __builtin_fmaf(d.y, ht, dwo1); }
                // cell backward onto the CARRIED accumulator gradients (the accumulators are running sums)
                const float dnv = Gh * (1.0f - z), dzv = Gh * (hprev - n);
                const float dpre = vo ? dnv * __builtin_fmaf(-n, n, 1.0f) : 0.0f;
                Gn += dpre;
                Gnh = __builtin_fmaf(dpre, r, Gnh);
                Gr = __builtin_fmaf(dpre * dmnh, r * (1.0f - r), Gr);
                Gz += vo ? dzv * (z * (1.0f - z)) : 0.0f;
                auxb[lane] = aux;
                dgb[lane] = Gr; dgb[64 + lane] = Gz; dgb[128 + lane] = Gnh;
                wave_lds_fence();
                float ddh = 0.0f;
                {
                    const float* w0 = pl + L.o_w_hh + (vo ? lane : 0);      // (lanes beyond H read column 0: finite values, result discarded)
                    const int HH = H * H;
                    for (int j4 = 0; j4 < H; j4 += 4) {
                        const float4 g0 = *reinterpret_cast<const float4*>(dgb + j4), g1 = *reinterpret_cast<const float4*>(dgb + 64 + j4),
                                     g2 = *reinterpret_cast<const float4*>(dgb + 128 + j4);
                        const float v0[4] = {g0.x, g0.y, g0.z, g0.w}, v1[4] = {g1.x, g1.y, g1.z, g1.w}, v2[4] = {g2.x, g2.y, g2.z, g2.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* wr = w0 + min(j4 + e, H - 1) * H;      // (rows beyond H: their gradients are zero)
                            ddh = __builtin_fmaf(v0[e], wr[0], ddh); ddh = __builtin_fmaf(v1[e], wr[HH], ddh); ddh = __builtin_fmaf(v2[e], wr[2 * HH], ddh);
                        }
                    }
                }
                float dxm[6], mxv[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) { dxm[i] = auxb[i]; mxv[i] = auxb[8 + i]; }
                if constexpr (NW) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float dr_ = rr == 0 ? dhm : __shfl(dhm, (lane + 16 * rr) & 63);
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(Gr, dr_, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(Gz, dr_, acc[1][rr], 0, 0, 0);
                        acc[2][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(Gnh, dr_, acc[2][rr], 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        dwih[0][i] = __builtin_fmaf(Gr, dxm[i], dwih[0][i]); dwih[1][i] = __builtin_fmaf(Gz, dxm[i], dwih[1][i]);
                        dwih[2][i] = __builtin_fmaf(Gn, dxm[i], dwih[2][i]);
                    }
                }
                if constexpr (DX) {      // dL/d(feature i) of this step through the thresholded delta and the carried reference-value gradient
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        float v = __builtin_fmaf(Gr, wih[0][i], __builtin_fmaf(Gz, wih[1][i], Gn * wih[2][i]));
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        const float mk = mxv[i];
                        if (lane == 0) dff[i * Tp + t0 + tt] = mk * v + mk * Gxp[i];
                        Gxp[i] = (1.0f - mk) * Gxp[i] - mk * v;
                    }
                }
                // dL/dh(t-1): the direct path, the delta path and the carried gradient of the reference value h_p
                const float Gnext = Gh * z + mh * ddh + mh * Ghp;
                Ghp = vo ? (1.0f - mh) * Ghp - mh * ddh : 0.0f;
                Gh = vo ? Gnext : 0.0f;
                wave_lds_fence();
            }
        }
        if constexpr (NW && !TRES) { dbs[0] += Gr; dbs[1] += Gz; dbs[2] += Gn; dbs[3] += Gnh; }      // the accumulators started at the biases
        if constexpr (DX) {      // dL/d(features) and the TCN's tap gradients of every sample -> dL/d(I, Q) (the oracle's closing loop)
            wave_lds_fence();
            for (int t = lane; t < T; t += 64) {
                const float2 xv = xg[t];
                const float I = xv.x, Q = xv.y, a2 = __builtin_fmaf(I, I, Q * Q), am = __builtin_amdgcn_sqrtf(a2), ia = fast_rcp(am);
                const float df0 = dff[t], df1 = dff[Tp + t], df2 = dff[2 * Tp + t], df3 = dff[3 * Tp + t], df4 = dff[4 * Tp + t], df5 = dff[5 * Tp + t];
                float dI, dQ;
                if constexpr (TRES) {
                    const float da = __builtin_fmaf(3.0f * a2, df3, df2);
                    const int tp = t > 0 ? t - 1 : T - 1;      // this sample is the "next" sample of step t - 1 (torch.roll)
                    dI = df0 + da * I * ia + dff[4 * Tp + tp] + dff[6 * Tp + t];
                    dQ = df1 + da * Q * ia + dff[5 * Tp + tp] + dff[7 * Tp + t];
                } else {
                    const float ia2 = fast_rcp(a2);
                    const float da = df2 + 3.0f * a2 * df3 - (Q * ia2) * df4 - (I * ia2) * df5;
                    dI = df0 + df5 * ia + da * I * ia;
                    dQ = df1 + df4 * ia + da * Q * ia;
                }
                reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t] = make_float2(dI, dQ);
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
            if constexpr (TRES) {
#pragma unroll
                for (int i = 0; i < 18; ++i) prow[L.o_tcn0 + i] = tacc[2 + i];
#pragma unroll
                for (int i = 0; i < 6; ++i) prow[L.o_tcn2 + i] = tacc[20 + i];
            } else { prow[L.o_b_out] = tacc[0]; prow[L.o_b_out + 1] = tacc[1]; }
        }
        if (vo) {
            prow[L.o_w_out + lane] = dwo0; prow[L.o_w_out + H + lane] = dwo1;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int i = 0; i < 6; ++i) prow[L.o_w_ih + (g * H + lane) * 6 + i] = dwih[g][i];
                if constexpr (!TRES) {
                    prow[L.o_b_ih + g * H + lane] = dbs[g];
                    prow[L.o_b_hh + g * H + lane] = g < 2 ? dbs[g] : dbs[3];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g)
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
int dw_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// float deltagru / deltagru_tcnskip of 33 .. 64 hidden units
bool delta_wide_ok(const odpd_model_t* m) {
    return (m->backbone == ODPD_DELTAGRU || m->backbone == ODPD_TRES_DELTAGRU) && m->bits_w == 0 && m->hidden > 32 && m->hidden <= 64;
}
int64_t delta_wide_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kDNS * 64; }
int delta_wide_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int delta_wide_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!delta_wide_ok(m)) return ODPD_EUNSUPPORTED;
    const bool tres = m->backbone == ODPD_TRES_DELTAGRU;
    const size_t lds = (size_t)dw_fwd_floats(delta_layout(m->hidden, tres).P) * sizeof(float);
    const int grid = delta_wide_rows(m, a.B);
    if (tres) return a.ckpt ? dw_launch(st, wide_delta_fwd_kernel<true, true>, grid, lds, a) : dw_launch(st, wide_delta_fwd_kernel<true, false>, grid, lds, a);
    return a.ckpt ? dw_launch(st, wide_delta_fwd_kernel<false, true>, grid, lds, a) : dw_launch(st, wide_delta_fwd_kernel<false, false>, grid, lds, a);
}
int delta_wide_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!delta_wide_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const bool tres = m->backbone == ODPD_TRES_DELTAGRU, nw = a.partials != nullptr, dx = a.dx != nullptr;
    const size_t lds = (size_t)dw_bwd_floats(delta_layout(m->hidden, tres).P, a.T, dx) * sizeof(float);
    const int grid = delta_wide_rows(m, a.B);
#define ODPD_DW_BWD(TR_)                                                                               \
    if (tres == TR_) {                                                                                 \
        if (nw && dx) return dw_launch(st, wide_delta_bwd_kernel<TR_, true, true>, grid, lds, a);      \
        if (nw) return dw_launch(st, wide_delta_bwd_kernel<TR_, true, false>, grid, lds, a);           \
        return dw_launch(st, wide_delta_bwd_kernel<TR_, false, true>, grid, lds, a);                   \
    }
    ODPD_DW_BWD(true) ODPD_DW_BWD(false)
#undef ODPD_DW_BWD
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
