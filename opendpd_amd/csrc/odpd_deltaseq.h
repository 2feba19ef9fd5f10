// odpd_deltaseq.h — a delta backbone (deltagru / TRes-DeltaGRU, backbones/deltagru.py, deltagru_tcnskip.py) as the trained DPD of a cascade
// on ONE wave: the step arithmetic of delta_eval_kernel / delta_gp_bwd_kernel (delta_family.hip: rows r | z | n | -, one rotated dot product
// per step and orientation, x-side delta memory one feature per lane, weight gradients as two 4-block MFMAs) restated chunk-wise for the
// cascade kernel (gru_cascade.hip).  A delta cell's accumulators are running sums over the whole frame, so its backward cannot re-form
// the gates from the parked h(t-1) alone; the whole-frame parked state of delta_gp_bwd_kernel (130 floats per step) would not fit beside the
// PA's.  Instead the forward pass keeps the cell state at every chunk start (5 floats per lane) and a backward chunk first runs its forward
// steps again from there, parking the chunk's state only: 32 x 130 floats.
#pragma once
#include "odpd_delta.h"

namespace odpd {

template <bool TRES>
struct DeltaSeq {
    static constexpr int C = 32;      // = kCascChunk (odpd_gpseq.h)
    // ---- LDS region: parameters | max(weight tables, buffers) ----
    __host__ __device__ static int tp(int T) { return (T + 63) & ~63; }
    __host__ __device__ static int nchunks(int T) { return (T + C - 1) / C; }
    __host__ __device__ static int off_ck(int T) { return tp(T) * 8; }                              // feat [Tp][8]: f0..f5, skip0, skip1
    __host__ __device__ static int off_hist(int T) { return off_ck(T) + nchunks(T) * 5 * 64; }      // ck [chunks][5][64]: h, h_p, x_p, acc_x, acc_h
    __host__ __device__ static int off_gpk(int T) { return off_hist(T) + (C + 1) * 16; }            // hist [C + 1][16]: entry i + 1 = h(t0 + i)
    __host__ __device__ static int off_dm2(int T) { return off_gpk(T) + C * 64; }                   // gpk [C][16][4]: r, z, n, dm_nh
    __host__ __device__ static int off_dmx(int T) { return off_dm2(T) + C * 32; }                   // dm2 [C][16][2]: masked dh, its mask
    __host__ __device__ static int off_dyb(int T) { return off_dmx(T) + C * 8; }                    // dmx [C][8]: the six masked dx
    __host__ __device__ static int off_dump(int T) { return off_dyb(T) + tp(T) * 2; }               // dyb [Tp][2]: dL/du(t), written by the PA wave
    __host__ __device__ static int off_hw(int T) { return off_dump(T) + 256; }
    __host__ __device__ static int buf_floats(int T) { return off_hw(T) + 32; }
    __host__ __device__ static int region_floats(int T, int P) {
        const int buf = buf_floats(T);
        return pad4(P) + (buf > kDTabFloats ? buf : kDTabFloats);
    }

    // ---- registers ----
    float wrec[16], wT[16], wih[6], accx0, acch0, wo0, wo1, bo0, bo1, w1[3][6], w2[2][3], thx, thh;
    float h, hp, xp, accx, acch;                          // cell state (h replicated on every row; x_p: feature `fc` of the lane)
    float gh_c, ghp, accg, gn;                            // backward carries: dL/dh, dL/dh_p, the row's accumulator gradient, G_n
    f32x16 acc1, acc2;
    float dwo0, dwo1, dbo0, dbo1, dbg, dbn, tw1[3][6], tw2[2][3], zx, zh;
    float *smem, *pl, *feat, *ck, *hist, *gpk, *dm2, *dmx, *dyb, *dump, *hw;
    DeltaLayout L;
    RowMasks rm;
    int H, T, lane, col, role, fc, p4_0, p4_step, p2_0, p2_step, p1_0, p1_step;
    bool vo;

    // (one workgroup barrier inside: fill_delta_tabs)
    __device__ __forceinline__ void setup(float* base, float* region, const float* params, int Hm, int T_, float thx_, float thh_) {
        smem = base;
        lane = threadIdx.x & 63; col = lane & 15; role = lane >> 4;      // r | z | n | -
        L = delta_layout(Hm, TRES);
        H = L.H; T = T_; thx = thx_; thh = thh_;
        pl = region;
        for (int i = lane; i < L.P; i += 64) pl[i] = params[i];
        wave_lds_fence();
        float* tab = region + pad4(L.P);
        fill_delta_tabs<true>(tab, pl, L, lane, 0, 1);
        vo = col < H;
        const bool gate_row = role < 3;
        {
            TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
            load_rot(wrec, tl + (gate_row ? role : 0) * 4 * 64);
            load_rot(wT, tl + (3 + (gate_row ? role : 0)) * 4 * 64);
#pragma unroll
            for (int k = 0; k < 16; ++k) { wrec[k] = gate_row ? wrec[k] : 0.0f; wT[k] = gate_row ? wT[k] : 0.0f; }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) wih[i] = (vo && gate_row) ? pl[L.o_w_ih + (role * H + col) * 6 + i] : 0.0f;
        accx0 = 0.0f; acch0 = 0.0f;
        if (!TRES && vo) {
            if (role < 2) accx0 = pl[L.o_b_ih + role * H + col] + pl[L.o_b_hh + role * H + col];
            if (role == 2) { accx0 = pl[L.o_b_ih + 2 * H + col]; acch0 = pl[L.o_b_hh + 2 * H + col]; }
        }
        fc = col < 6 ? col : 5;
        wo0 = vo ? pl[L.o_w_out + col] : 0.0f; wo1 = vo ? pl[L.o_w_out + H + col] : 0.0f;
        bo0 = TRES ? 0.0f : pl[L.o_b_out]; bo1 = TRES ? 0.0f : pl[L.o_b_out + 1];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { w1[ch][k] = TRES ? pl[L.o_tcn0 + ch * 6 + k] : 0.0f; tw1[ch][k] = 0.0f; }
#pragma unroll
            for (int oo = 0; oo < 2; ++oo) { w2[oo][ch] = TRES ? pl[L.o_tcn2 + oo * 3 + ch] : 0.0f; tw2[oo][ch] = 0.0f; }
        }
        wave_lds_fence();
        // buffers over the tables
        feat = tab; ck = tab + off_ck(T); hist = tab + off_hist(T); gpk = tab + off_gpk(T); dm2 = tab + off_dm2(T); dmx = tab + off_dmx(T);
        dyb = tab + off_dyb(T); dump = tab + off_dump(T); hw = tab + off_hw(T);
        if (lane < 32) hw[lane] = (lane & 15) < H ? pl[L.o_w_out + (lane >> 4) * H + (lane & 15)] : 0.0f;
        rm = row_masks();
        const int dmp = (int)(dump - smem);
        // per-step stores of the recomputed forward steps: row 2 parks (r, z, n, dm_nh), row 1 (masked dh, mask), row 3 h(t), row 0 the masked dx of its lanes 0..7
        p4_0 = role == 2 ? (int)(gpk - smem) + 4 * col : dmp + 4 * lane; p4_step = role == 2 ? 64 : 0;
        p2_0 = role == 1 ? (int)(dm2 - smem) + 2 * col : dmp + 2 * lane; p2_step = role == 1 ? 32 : 0;
        p1_0 = role == 3 ? (int)(hist - smem) + 16 + col : (role == 0 && col < 8) ? (int)(dmx - smem) + col : dmp + lane;
        p1_step = role == 3 ? 16 : (role == 0 && col < 8) ? 8 : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
        dwo0 = 0.0f; dwo1 = 0.0f; dbo0 = 0.0f; dbo1 = 0.0f; dbg = 0.0f; dbn = 0.0f; zx = 0.0f; zh = 0.0f;
        wave_lds_fence();
    }

    __device__ __forceinline__ void fwd_begin() { h = 0.0f; hp = 0.0f; xp = 0.0f; accx = accx0; acch = acch0; }

    // one step of the cell at local index tt (features of time t in feat).  PARK: the recomputed steps of a backward chunk
    template <bool PARK>
    __device__ __forceinline__ void step(int t, int& q4, int& q2, int& q1) {
        const float fv = feat[t * 8 + fc];
        const float d = fv - xp, ad = __builtin_fabsf(d);
        const float dm = !(ad < thx) ? d : 0.0f;                          // masked_fill(|d| < th, 0)  (deltagru.py:179-183)
        xp = (ad >= thx) ? fv : xp;
        if constexpr (!PARK) zx += (dm == 0.0f) ? 1.0f : 0.0f;
        float ax = accx;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            ax = __builtin_fmaf(wih[i], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dm), i)), ax);
        const float dhv = h - hp, adh = __builtin_fabsf(dhv);
        const bool keeph = !(adh < thh);
        const float dhm = keeph ? dhv : 0.0f;
        hp = (adh >= thh) ? h : hp;
        if constexpr (!PARK) zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
        const bool nrow = role == 2;
        const float res = rotdot(nrow ? acch : ax, wrec, dhm);            // rows r, z: dm += W_ih dx + W_hh dh; row n: dm_nh += W_hn dh
        accx = nrow ? ax : res; acch = nrow ? res : acch;
        const float sg = sigmoidf_(res);
        const float r = dup32(sg).lo;                                       // row 2 <- r of row 0
        const float n = tanhf_(__builtin_fmaf(r, res, ax));               // row 2
        float g4[4];
        gather_rows(nrow ? n : sg, g4);
        h = __builtin_fmaf(g4[1], h - g4[2], g4[2]);
        if constexpr (PARK) {
            *reinterpret_cast<float4*>(smem + q4) = make_float4(r, g4[1], g4[2], res);
            *reinterpret_cast<float2*>(smem + q2) = make_float2(dhm, keeph ? 1.0f : 0.0f);
            smem[q1] = role == 3 ? h : dm;
            q4 += p4_step; q2 += p2_step;
        } else {
            smem[q1] = role == 3 ? h : dm;                                // (row 3: h(t) for the chunk's fc_out; the others: dump / unused dmx)
        }
        q1 += p1_step;
    }

    // forward chunk c = steps t0 .. t0 + len - 1: features and the TRes skip with lane = time step, the cell state at the chunk start kept,
    // the recurrence, then fc_out (+ skip) of the chunk with lane = time step: sink(t, u0, u1)
    template <typename Sink>
    __device__ __forceinline__ void fwd_chunk(int c, int t0, int len, const float2* xg, Sink sink) {
        {
            const int t = t0 + lane;
            const float2 zero = make_float2(0.0f, 0.0f);
            const float2 rc = t < T ? xg[t] : make_float2(0.5f, 0.5f);
            float2 rn = zero, rm_ = zero, rp = zero;
            if constexpr (TRES) {
                rn = t + 1 < T ? xg[t + 1] : xg[0];                       // torch.roll: the last step sees sample 0
                rm_ = (t - kDHalo >= 0 && t - kDHalo < T) ? xg[t - kDHalo] : zero;
                rp = t + kDHalo < T ? xg[t + kDHalo] : zero;
            }
            float f[6];
            delta_feat<TRES>(rc, rn, f);
            float sk[2] = {0.0f, 0.0f};
            if constexpr (TRES) {
                float s2[2] = {0.0f, 0.0f};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float s1 = w1[ch][0] * rm_.x;
                    s1 = __builtin_fmaf(w1[ch][1], rc.x, s1); s1 = __builtin_fmaf(w1[ch][2], rp.x, s1);
                    s1 = __builtin_fmaf(w1[ch][3], rm_.y, s1); s1 = __builtin_fmaf(w1[ch][4], rc.y, s1); s1 = __builtin_fmaf(w1[ch][5], rp.y, s1);
                    const float hs = hardswishf_(s1);
                    s2[0] = __builtin_fmaf(w2[0][ch], hs, s2[0]); s2[1] = __builtin_fmaf(w2[1][ch], hs, s2[1]);
                }
                sk[0] = hardswishf_(s2[0]); sk[1] = hardswishf_(s2[1]);
            }
            if (lane < len) {
                reinterpret_cast<float4*>(feat)[2 * t] = make_float4(f[0], f[1], f[2], f[3]);
                reinterpret_cast<float4*>(feat)[2 * t + 1] = make_float4(f[4], f[5], sk[0], sk[1]);
            }
            float* k = ck + c * 5 * 64 + lane;
            k[0] = h; k[64] = hp; k[128] = xp; k[192] = accx; k[256] = acch;
        }
        wave_lds_fence();
        int q4 = 0, q2 = 0, q1 = p1_0;
        for (int tt = 0; tt < len; ++tt) step<false>(t0 + tt, q4, q2, q1);
        wave_lds_fence();
        if (lane < len) {
            const int t = t0 + lane;
            const float4* hv4 = reinterpret_cast<const float4*>(hist + (lane + 1) * 16);
            const float4* hw4 = reinterpret_cast<const float4*>(hw);
            float y0 = bo0, y1 = bo1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 hv = hv4[q], a0 = hw4[q], a1 = hw4[4 + q];
                y0 = __builtin_fmaf(a0.x, hv.x, y0); y0 = __builtin_fmaf(a0.y, hv.y, y0); y0 = __builtin_fmaf(a0.z, hv.z, y0); y0 = __builtin_fmaf(a0.w, hv.w, y0);
                y1 = __builtin_fmaf(a1.x, hv.x, y1); y1 = __builtin_fmaf(a1.y, hv.y, y1); y1 = __builtin_fmaf(a1.z, hv.z, y1); y1 = __builtin_fmaf(a1.w, hv.w, y1);
            }
            if constexpr (TRES) { y0 += feat[t * 8 + 6]; y1 += feat[t * 8 + 7]; }
            sink(t, y0, y1);
        }
        wave_lds_fence();
    }

    __device__ __forceinline__ void bwd_begin() { gh_c = 0.0f; ghp = 0.0f; accg = 0.0f; gn = 0.0f; }

    // backward chunk c: the forward steps again from the kept state, parking what the backward steps need; dL/du of the chunk (dyb, written
    // by the PA wave) and the TCN gradient with lane = time step; then the steps t0 + len - 1 .. t0
    __device__ __forceinline__ void bwd_chunk(int c, int t0, int len, const float2* xg) {
        {
            const float* k = ck + c * 5 * 64 + lane;
            h = k[0]; hp = k[64]; xp = k[128]; accx = k[192]; acch = k[256];
            if (role == 3) hist[col] = h;                                 // entry 0 = h(t0 - 1)
        }
        wave_lds_fence();
        int q4 = p4_0, q2 = p2_0, q1 = p1_0;
        for (int tt = 0; tt < len; ++tt) step<true>(t0 + tt, q4, q2, q1);
        wave_lds_fence();
        if (lane < len) {
            const int t = t0 + lane;
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
            dbo0 += dyv.x; dbo1 += dyv.y;
            if constexpr (TRES) {
                const float2 zero = make_float2(0.0f, 0.0f);
                const float2 xc = xg[t], xm = t - kDHalo >= 0 ? xg[t - kDHalo] : zero, xq = t + kDHalo < T ? xg[t + kDHalo] : zero;
                float s1[3], hs[3], s2a = 0.0f, s2b = 0.0f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    s1[ch] = w1[ch][0] * xm.x;
                    s1[ch] = __builtin_fmaf(w1[ch][1], xc.x, s1[ch]); s1[ch] = __builtin_fmaf(w1[ch][2], xq.x, s1[ch]);
                    s1[ch] = __builtin_fmaf(w1[ch][3], xm.y, s1[ch]); s1[ch] = __builtin_fmaf(w1[ch][4], xc.y, s1[ch]);
                    s1[ch] = __builtin_fmaf(w1[ch][5], xq.y, s1[ch]);
                    hs[ch] = hardswishf_(s1[ch]);
                    s2a = __builtin_fmaf(w2[0][ch], hs[ch], s2a); s2b = __builtin_fmaf(w2[1][ch], hs[ch], s2b);
                }
                const float d2a = dyv.x * hswish_grad_(s2a), d2b = dyv.y * hswish_grad_(s2b);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    tw2[0][ch] = __builtin_fmaf(d2a, hs[ch], tw2[0][ch]); tw2[1][ch] = __builtin_fmaf(d2b, hs[ch], tw2[1][ch]);
                    const float d1 = __builtin_fmaf(d2a, w2[0][ch], d2b * w2[1][ch]) * hswish_grad_(s1[ch]);
                    tw1[ch][0] = __builtin_fmaf(d1, xm.x, tw1[ch][0]); tw1[ch][1] = __builtin_fmaf(d1, xc.x, tw1[ch][1]);
                    tw1[ch][2] = __builtin_fmaf(d1, xq.x, tw1[ch][2]); tw1[ch][3] = __builtin_fmaf(d1, xm.y, tw1[ch][3]);
                    tw1[ch][4] = __builtin_fmaf(d1, xc.y, tw1[ch][4]); tw1[ch][5] = __builtin_fmaf(d1, xq.y, tw1[ch][5]);
                }
            }
        }
        for (int tt = len - 1; tt >= 0; --tt) {
            const float hprev = hist[tt * 16 + col], ht = hist[(tt + 1) * 16 + col];
            const float4 g = reinterpret_cast<const float4*>(gpk)[tt * 16 + col];
            const float2 dd = reinterpret_cast<const float2*>(dm2)[tt * 16 + col];
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * (t0 + tt));
            const float fsx = col < 6 ? dmx[tt * 8 + col] : 0.0f;
            const float r = g.x, z = g.y, n = g.z, nh = g.w, dhm = dd.x, mk = dd.y;
            const float gh = gh_c + __builtin_fmaf(dyv.x, wo0, dyv.y * wo1);
            dwo0 = __builtin_fmaf(dyv.x, ht, dwo0); dwo1 = __builtin_fmaf(dyv.y, ht, dwo1);
            const float dn = gh * (1.0f - z), dz = gh * (hprev - n);
            float ghprev = gh * z;
            const float dpre = dn * __builtin_fmaf(-n, n, 1.0f);
            gn += dpre;
            const float c_nh = __builtin_fmaf(dpre, r, accg), c_r = __builtin_fmaf(dpre * nh, r * (1.0f - r), accg),
                        c_z = __builtin_fmaf(dz, z * (1.0f - z), accg);
            accg = vsel(rm.m[0], c_r, vsel(rm.m[1], c_z, c_nh));
            acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(accg, dhm, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(vsel(rm.m[2], gn, accg), fsx, acc2, 0, 0, 0);
            float ddh = rotdot(0.0f, wT, accg);
            ddh = sum_rows4(ddh);
            ghprev = __builtin_fmaf(mk, ddh + ghp, ghprev);
            ghp = __builtin_fmaf(-mk, ddh, (1.0f - mk) * ghp);
            gh_c = ghprev;
        }
        wave_lds_fence();
    }
    // end of a frame's backward: the gradient w.r.t. the initial accumulators = bias gradients (deltagru.py:165-170)
    __device__ __forceinline__ void bwd_end() { dbg += accg; dbn += gn; }

    // the workgroup's row of partial gradients (every entry written)
    __device__ __forceinline__ void write_partials(float* prow, float loss) {
        if (lane < kLossCols) prow[L.P + lane] = lane == 0 ? loss : 0.0f;
        if (vo && role == 0) { prow[L.o_w_out + col] = dwo0; prow[L.o_w_out + H + col] = dwo1; }
        if constexpr (TRES) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
                for (int k = 0; k < 6; ++k) { const float v = wave_sum_(tw1[ch][k]); if (lane == 0) prow[L.o_tcn0 + ch * 6 + k] = v; }
#pragma unroll
                for (int oo = 0; oo < 2; ++oo) { const float v = wave_sum_(tw2[oo][ch]); if (lane == 0) prow[L.o_tcn2 + oo * 3 + ch] = v; }
            }
        } else {
            const float b0 = wave_sum_(dbo0), b1 = wave_sum_(dbo1);
            if (lane == 0) { prow[L.o_b_out] = b0; prow[L.o_b_out + 1] = b1; }
            if (vo) {
                if (role < 2) { prow[L.o_b_ih + role * H + col] = dbg; prow[L.o_b_hh + role * H + col] = dbg; }
                if (role == 2) { prow[L.o_b_ih + 2 * H + col] = dbn; prow[L.o_b_hh + 2 * H + col] = dbg; }
            }
        }
        // MFMA block k = gate k (r, z, n); register 4 k + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 4 * role + rr;
                if (i < H) {
                    if (col < H) prow[L.o_w_hh + (k * H + i) * H + col] = acc1[4 * k + rr];
                    if (col < 6) prow[L.o_w_ih + (k * H + i) * 6 + col] = acc2[4 * k + rr];
                }
            }
    }
    // sparsity counters of the forward passes (deltagru.py:214-226): dx zeros = the six feature lanes of row 0, dh zeros = the hidden units of row 0
    __device__ __forceinline__ void add_stats(double* stats, int B) {
        if (stats == nullptr) return;
        float tx = (role == 0 && col < 6) ? zx : 0.0f, th = role == 0 ? zh : 0.0f;
        for (int o = 32; o > 0; o >>= 1) { tx += __shfl_down(tx, o); th += __shfl_down(th, o); }
        if (lane == 0) {
            atomicAdd(&stats[0], (double)tx);
            atomicAdd(&stats[2], (double)th);
        }
        if (blockIdx.x == 0 && lane == 0) {
            atomicAdd(&stats[1], 6.0 * (double)B * (double)T);
            atomicAdd(&stats[3], (double)H * (double)B * (double)T);
        }
    }
};

}  // namespace odpd
