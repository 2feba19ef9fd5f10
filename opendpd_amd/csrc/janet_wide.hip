// janet_wide.hip — PGJANET (backbones/pgjanet.py:5-84) with 17 .. 32 hidden units: one sequence per single-wave workgroup, LANE = HIDDEN UNIT, the
// wave's two halves sharing a unit's work (the tile kernels of janet_family.hip / janet_s16.hip stop at one 16-unit tile).
//   cell      a_n = tanh(W_a [h, |x|]), p1 = tanh(W_p1 [h, cos]), p2 = tanh(W_p2 [h, sin]), u = A(a_n) A(p1) A(p2) with A(v) = v (1 - v),
//             f = sigmoid(W_f [h, u]), g = tanh(W_g [h, u]), h' = f h + (1 - f) g, y = W_o h' (pgjanet.py:33-72).
//   forward   lanes 0 .. 31 (half A) hold the h-rows of W_a, W_p1, W_p2 of unit j, lanes 32 .. 63 (half B) the h-rows AND the u-rows of W_f, W_g of the
//             same unit.  Step: broadcast h through LDS (8 x ds_read_b128) -> A forms a_n, p1, p2, u and writes u to LDS; B has the h-parts of f, g
//             -> broadcast u -> B finishes f, g, h'.  |x|, cos, sin of a 64-step chunk and fc_out with lane = time step.  Records (a_n, p1, p2, u |
//             f, g, h') of every step in HBM: B x T x 4 x 64 floats.
//   backward  reverse steps: B forms d_f, d_g (pre-activation gradients) -> LDS; B multiplies them with the h-columns of W_f, W_g (dL/dh(t-1)), A with
//             their u-columns (dL/du) — the two transposed mat-vecs of that stage run side by side in the two halves —; A turns dL/du into d_a, d_p1,
//             d_p2 -> LDS; B adds W_a^T d_a + W_p1^T d_p1 + W_p2^T d_p2.  The seven H x H gradient blocks accumulate as 4-block MFMA outer products
//             (block = (row half-tile, column half-tile) of the 32 x 32 block): 7 x 16 accumulators.
#include "odpd_seq.h"

namespace odpd {
namespace {
constexpr int kJC = 64, kJS = 33, kJNS = 4;

__host__ __device__ inline int jw_fwd_floats(int P) { return pad4(P) + kJC * 4 + 64 + kJC * kJS; }
__host__ __device__ inline int jw_bwd_floats(int P) { return pad4(P) + kJC * 4 + kJC * 2 + kJC * 2 + 8 * 32 + (kJC + 1) * kJS + 3; }

// |x|, cos(theta), sin(theta) of the chunk's steps (theta = atan2(Q, I): cos = I / |x|, sin = Q / |x|), lane = time step
__device__ __forceinline__ void jw_stage_inputs(float* ftab, const float2* xg, int t0, int T, int lane) {
    const float2 xv = t0 + lane < T ? xg[t0 + lane] : make_float2(0.6f, 0.8f);
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), a = __builtin_amdgcn_sqrtf(a2), ia = fast_rcp(a);
    reinterpret_cast<float4*>(ftab)[lane] = make_float4(a, xv.x * ia, xv.y * ia, 0.0f);
}

template <bool SAVE>
__global__ __launch_bounds__(64) void wide_pgjanet_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31;
    const bool hb_ = lane >= 32;                     // half B: the f / g gates and the state
    const JanetLayout L = janet_layout(a.H);
    const int H = L.H, T = a.T, H1 = H + 1, H2 = 2 * H;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [64][4]: |x|, cos, sin of the chunk's steps
    float* hb = ftab + kJC * 4;                // [32] h | [32] u, for the broadcast reads
    float* hist = hb + 64;                     // [64][33]: h of the chunk's steps
    const bool vo = ju < H;
    // half A: rows of W_a, W_p1, W_p2 over h, their scalar-input column and bias; half B: rows of W_f, W_g over h (wh) and over u (wu)
    float wh[3][32], wu[2][32], ws[3], bs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ow = r == 0 ? L.o_wa : r == 1 ? L.o_wp1 : L.o_wp2, ob = r == 0 ? L.o_ba : r == 1 ? L.o_bp1 : L.o_bp2;
        const int owB = r == 0 ? L.o_wf : L.o_wg, obB = r == 0 ? L.o_bf : L.o_bg;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float v = 0.0f;
            if (vo && k < H) v = hb_ ? (r < 2 ? pl[owB + ju * H2 + k] : 0.0f) : pl[ow + ju * H1 + k];
            wh[r][k] = v;
            if (r < 2) wu[r][k] = (hb_ && vo && k < H) ? pl[owB + ju * H2 + H + k] : 0.0f;
        }
        ws[r] = (!hb_ && vo) ? pl[ow + ju * H1 + H] : 0.0f;
        bs[r] = !vo ? 0.0f : hb_ ? (r < 2 ? pl[obB + ju] : 0.0f) : pl[ob + ju];
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kJNS * 64 : nullptr;
        float h = 0.0f;                                  // (half B)
        for (int t0 = 0; t0 < T; t0 += kJC) {
            const int len = min(kJC, T - t0);
            wave_lds_fence();
            jw_stage_inputs(ftab, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                if (hb_) hb[ju] = h;
                wave_lds_fence();
                const float4 in = reinterpret_cast<const float4*>(ftab)[tt];
                float acc[3] = {bs[0], bs[1], bs[2]};
                const float4* h4 = reinterpret_cast<const float4*>(hb);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 hv = h4[q];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        acc[r] = __builtin_fmaf(wh[r][4 * q], hv.x, acc[r]); acc[r] = __builtin_fmaf(wh[r][4 * q + 1], hv.y, acc[r]);
                        acc[r] = __builtin_fmaf(wh[r][4 * q + 2], hv.z, acc[r]); acc[r] = __builtin_fmaf(wh[r][4 * q + 3], hv.w, acc[r]);
                    }
                }
                // half A: the three tanh gates and u
                const float an = tanhf_(__builtin_fmaf(ws[0], in.x, acc[0])), p1 = tanhf_(__builtin_fmaf(ws[1], in.y, acc[1])),
                            p2 = tanhf_(__builtin_fmaf(ws[2], in.z, acc[2]));
                const float u = (an * p1 * p2) * ((1.0f - an) * (1.0f - p1) * (1.0f - p2));
                if (!hb_) hb[32 + ju] = vo ? u : 0.0f;
                wave_lds_fence();
                // half B: f, g, h'
                float pf = acc[0], pg = acc[1];
                const float4* u4 = reinterpret_cast<const float4*>(hb + 32);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 uv = u4[q];
                    pf = __builtin_fmaf(wu[0][4 * q], uv.x, pf); pf = __builtin_fmaf(wu[0][4 * q + 1], uv.y, pf);
                    pf = __builtin_fmaf(wu[0][4 * q + 2], uv.z, pf); pf = __builtin_fmaf(wu[0][4 * q + 3], uv.w, pf);
                    pg = __builtin_fmaf(wu[1][4 * q], uv.x, pg); pg = __builtin_fmaf(wu[1][4 * q + 1], uv.y, pg);
                    pg = __builtin_fmaf(wu[1][4 * q + 2], uv.z, pg); pg = __builtin_fmaf(wu[1][4 * q + 3], uv.w, pg);
                }
                const float f = sigmoidf_(pf), g = tanhf_(pg);
                const float hn = (hb_ && vo) ? __builtin_fmaf(f, h - g, g) : 0.0f;          // f h + (1 - f) g
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * kJNS * 64 + lane;
                    s[0] = hb_ ? f : an; s[64] = hb_ ? g : p1; s[128] = hb_ ? hn : p2; s[192] = hb_ ? 0.0f : u;
                }
                h = hn;
                if (hb_) hist[tt * kJS + ju] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step
                const float* hr = hist + lane * kJS;
                float y0 = pl[L.o_bo], y1 = pl[L.o_bo + 1];
                for (int j = 0; j < H; ++j) {
                    const float hv = hr[j];
                    y0 = __builtin_fmaf(pl[L.o_wo + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_wo + H + j], hv, y1);
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(64) void wide_pgjanet_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31, col = lane & 15, quad = lane >> 4;
    const bool hb_ = lane >= 32;
    const JanetLayout L = janet_layout(a.H);
    const int H = L.H, T = a.T, NC = (T + kJC - 1) / kJC, H1 = H + 1, H2 = 2 * H;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [64][4]  |x|, cos, sin of the chunk's steps
    float* dxb = ftab + kJC * 4;               // [64][2]  dL/dx of the chunk's steps
    float* dyb = dxb + kJC * 2;                // [64][2]  dL/dy of the chunk's steps
    float* vb = dyb + kJC * 2;                 // [8][32]  the step's vectors: 0 d_f, 1 d_g, 2 d_a, 3 d_p1, 4 d_p2, 5 h(t-1), 6 u(t)
    float* hs = vb + 8 * 32;                   // [65][33] row i = h(t0 - 1 + i)
    const bool vo = ju < H;
    const float wo0 = (hb_ && vo) ? pl[L.o_wo + ju] : 0.0f, wo1 = (hb_ && vo) ? pl[L.o_wo + H + ju] : 0.0f;
    const float ws0 = (!hb_ && vo) ? pl[L.o_wa + ju * H1 + H] : 0.0f, ws1 = (!hb_ && vo) ? pl[L.o_wp1 + ju * H1 + H] : 0.0f,
                ws2 = (!hb_ && vo) ? pl[L.o_wp2 + ju * H1 + H] : 0.0f;
    // dW blocks: 0 W_f over h, 1 W_f over u, 2 W_g over h, 3 W_g over u, 4 W_a over h, 5 W_p1 over h, 6 W_p2 over h
    f32x16 acc[7];
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
    // per lane: half A: scalar columns and biases of W_a, W_p1, W_p2; half B: biases of W_f, W_g and the fc_out columns of the unit
    float dsc[3] = {0.f, 0.f, 0.f}, dbi[3] = {0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
    const int rb = 16 * (quad >> 1) + col, cb = 16 * (quad & 1) + col;      // this lane's entries of a row-operand / column-operand of the 4-block MFMA
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kJNS * 64;
        float dh = 0.0f;                                 // (half B)
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kJC, len = min(kJC, T - t0);
            wave_lds_fence();
            jw_stage_inputs(ftab, xg, t0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            if (hb_) {
                hs[ju] = t0 > 0 ? sv[(size_t)(t0 - 1) * kJNS * 64 + 128 + lane] : 0.0f;
                for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * kJS + ju] = sv[(size_t)(t0 + tt) * kJNS * 64 + 128 + lane];
            }
            wave_lds_fence();
            float r0n, r1n, r2n, r3n;                    // the next step's record (in flight while this one is worked on)
            {
                const float* s = sv + (size_t)(t0 + len - 1) * kJNS * 64 + lane;
                r0n = s[0]; r1n = s[64]; r2n = s[128]; r3n = s[192];
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float r0 = r0n, r1 = r1n, r2 = r2n, r3 = r3n;      // A: a_n, p1, p2, u | B: f, g, h(t), -
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * kJNS * 64 + lane;
                    r0n = s[0]; r1n = s[64]; r2n = s[128]; r3n = s[192];
                }
                const float4 in = reinterpret_cast<const float4*>(ftab)[tt];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                const float hp = hs[tt * kJS + ju];
                // ---- half B: h' = f h + (1 - f) g ----
                const float dht = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, dh));
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, r2, dwo0); dwo1 = __builtin_fmaf(d.y, r2, dwo1); }
                const float dfp = (hb_ && vo) ? (dht * (hp - r1)) * (r0 * (1.0f - r0)) : 0.0f;
                const float dgp = (hb_ && vo) ? (dht * (1.0f - r0)) * __builtin_fmaf(-r1, r1, 1.0f) : 0.0f;
                if (hb_) { vb[ju] = dfp; vb[32 + ju] = dgp; vb[5 * 32 + ju] = hp; }
                else vb[6 * 32 + ju] = r3;
                wave_lds_fence();
                // B: dL/dh(t-1) through the h-columns of W_f, W_g; A: dL/du through their u-columns
                float tv = 0.0f;
                {
                    const int cofs = (vo ? ju : 0) + (hb_ ? 0 : H);
                    const float* wf = pl + L.o_wf + cofs;
                    const float* wg = pl + L.o_wg + cofs;
                    for (int j4 = 0; j4 < H; j4 += 4) {
                        const float4 g0 = *reinterpret_cast<const float4*>(vb + j4), g1 = *reinterpret_cast<const float4*>(vb + 32 + j4);
                        const float v0[4] = {g0.x, g0.y, g0.z, g0.w}, v1[4] = {g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = min(j4 + e, H - 1);      // (rows beyond H: their gradients are zero)
                            tv = __builtin_fmaf(v0[e], wf[j * H2], tv); tv = __builtin_fmaf(v1[e], wg[j * H2], tv);
                        }
                    }
                }
                // ---- half A: u = A(a_n) A(p1) A(p2), A(v) = v (1 - v) ----
                const float Aa = r0 * (1.0f - r0), Ab = r1 * (1.0f - r1), Ac = r2 * (1.0f - r2);
                const float du = (!hb_ && vo) ? tv : 0.0f;
                const float dap = (du * (1.0f - 2.0f * r0) * Ab * Ac) * __builtin_fmaf(-r0, r0, 1.0f);
                const float dbp = (du * Aa * (1.0f - 2.0f * r1) * Ac) * __builtin_fmaf(-r1, r1, 1.0f);
                const float dcp = (du * Aa * Ab * (1.0f - 2.0f * r2)) * __builtin_fmaf(-r2, r2, 1.0f);
                if (!hb_) { vb[2 * 32 + ju] = dap; vb[3 * 32 + ju] = dbp; vb[4 * 32 + ju] = dcp; }
                wave_lds_fence();
                // B: + W_a^T d_a + W_p1^T d_p1 + W_p2^T d_p2 (h-columns)
                float dhn = hb_ ? __builtin_fmaf(dht, r0, tv) : 0.0f;
                {
                    const int cofs = vo ? ju : 0;
                    const float* wa = pl + L.o_wa + cofs;
                    const float* w1 = pl + L.o_wp1 + cofs;
                    const float* w2 = pl + L.o_wp2 + cofs;
                    for (int j4 = 0; j4 < H; j4 += 4) {
                        const float4 g0 = *reinterpret_cast<const float4*>(vb + 64 + j4), g1 = *reinterpret_cast<const float4*>(vb + 96 + j4),
                                     g2 = *reinterpret_cast<const float4*>(vb + 128 + j4);
                        const float v0[4] = {g0.x, g0.y, g0.z, g0.w}, v1[4] = {g1.x, g1.y, g1.z, g1.w}, v2[4] = {g2.x, g2.y, g2.z, g2.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = min(j4 + e, H - 1);
                            dhn = __builtin_fmaf(v0[e], wa[j * H1], dhn); dhn = __builtin_fmaf(v1[e], w1[j * H1], dhn); dhn = __builtin_fmaf(v2[e], w2[j * H1], dhn);
                        }
                    }
                }
                dh = (hb_ && vo) ? dhn : 0.0f;
                if constexpr (NW) {
                    // outer products: row operand = a gradient vector, column operand = h(t-1) or u(t); block (quad >> 1, quad & 1) of the 32 x 32 block
                    const float gfr = vb[rb], ggr = vb[32 + rb], gar = vb[64 + rb], g1r = vb[96 + rb], g2r = vb[128 + rb];
                    const float hc = vb[5 * 32 + cb], uc = vb[6 * 32 + cb];
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x1f32(gfr, hc, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x1f32(gfr, uc, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_16x16x1f32(ggr, hc, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_16x16x1f32(ggr, uc, acc[3], 0, 0, 0);
                    acc[4] = __builtin_amdgcn_mfma_f32_16x16x1f32(gar, hc, acc[4], 0, 0, 0);
                    acc[5] = __builtin_amdgcn_mfma_f32_16x16x1f32(g1r, hc, acc[5], 0, 0, 0);
                    acc[6] = __builtin_amdgcn_mfma_f32_16x16x1f32(g2r, hc, acc[6], 0, 0, 0);
                    if (hb_) { dbi[0] += dfp; dbi[1] += dgp; }
                    else {
                        dbi[0] += dap; dbi[1] += dbp; dbi[2] += dcp;
                        dsc[0] = __builtin_fmaf(dap, in.x, dsc[0]); dsc[1] = __builtin_fmaf(dbp, in.y, dsc[1]); dsc[2] = __builtin_fmaf(dcp, in.z, dsc[2]);
                    }
                }
                if constexpr (DX) {      // theta = atan2(Q, I): d theta = -sin d cos + cos d sin; d theta / dI = -Q / a^2, d theta / dQ = I / a^2
                    float damp = dap * ws0, dct = dbp * ws1, dst = dcp * ws2;
                    for (int o = 32; o > 0; o >>= 1) { damp += __shfl_xor(damp, o); dct += __shfl_xor(dct, o); dst += __shfl_xor(dst, o); }
                    if (lane == 0) {
                        const float dth = __builtin_fmaf(-in.z, dct, in.y * dst), ia = fast_rcp(in.x);
                        // I / a = cos, Q / a = sin; Q / a^2 = sin / a, I / a^2 = cos / a
                        reinterpret_cast<float2*>(dxb)[tt] = make_float2(__builtin_fmaf(damp, in.y, -dth * in.z * ia), __builtin_fmaf(damp, in.z, dth * in.y * ia));
                    }
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
        if (lane == 0) { prow[L.o_bo] = tb0; prow[L.o_bo + 1] = tb1; }
        if (vo) {
            if (hb_) {
                prow[L.o_wo + ju] = dwo0; prow[L.o_wo + H + ju] = dwo1;
                prow[L.o_bf + ju] = dbi[0]; prow[L.o_bg + ju] = dbi[1];
            } else {
                prow[L.o_ba + ju] = dbi[0]; prow[L.o_bp1 + ju] = dbi[1]; prow[L.o_bp2 + ju] = dbi[2];
                prow[L.o_wa + ju * H1 + H] = dsc[0]; prow[L.o_wp1 + ju * H1 + H] = dsc[1]; prow[L.o_wp2 + ju * H1 + H] = dsc[2];
            }
        }
        // MFMA block bb = (row half-tile bb >> 1, column half-tile bb & 1): register 4 bb + i of lane l = entry (row 4 (l / 16) + i, column l % 16)
#pragma unroll
        for (int m = 0; m < 7; ++m)
#pragma unroll
            for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int jr = 16 * (bb >> 1) + 4 * quad + i, kc = 16 * (bb & 1) + col;
                    if (jr < H && kc < H) {
                        const int idx = m == 0 ? L.o_wf + jr * H2 + kc : m == 1 ? L.o_wf + jr * H2 + H + kc : m == 2 ? L.o_wg + jr * H2 + kc
                                      : m == 3 ? L.o_wg + jr * H2 + H + kc : m == 4 ? L.o_wa + jr * H1 + kc : m == 5 ? L.o_wp1 + jr * H1 + kc
                                                                                                                     : L.o_wp2 + jr * H1 + kc;
                        prow[idx] = acc[m][4 * bb + i];
                    }
                }
    }
}

template <typename K>
int jw_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// pgjanet of 17 .. 32 hidden units
bool pgjanet_wide_ok(const odpd_model_t* m) { return m->backbone == ODPD_PGJANET && m->bits_w == 0 && m->hidden > 16 && m->hidden <= 32; }
int64_t pgjanet_wide_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kJNS * 64; }
int pgjanet_wide_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int pgjanet_wide_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!pgjanet_wide_ok(m)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)jw_fwd_floats(janet_layout(m->hidden).P) * sizeof(float);
    const int grid = pgjanet_wide_rows(m, a.B);
    return a.ckpt ? jw_launch(st, wide_pgjanet_fwd_kernel<true>, grid, lds, a) : jw_launch(st, wide_pgjanet_fwd_kernel<false>, grid, lds, a);
}
int pgjanet_wide_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!pgjanet_wide_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)jw_bwd_floats(janet_layout(m->hidden).P) * sizeof(float);
    const int grid = pgjanet_wide_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return jw_launch(st, wide_pgjanet_bwd_kernel<true, true>, grid, lds, a);
    if (nw) return jw_launch(st, wide_pgjanet_bwd_kernel<true, false>, grid, lds, a);
    return jw_launch(st, wide_pgjanet_bwd_kernel<false, true>, grid, lds, a);
}

}  // namespace odpd
