// janet_s16.hip — PGJANET (backbones/pgjanet.py:5-84) in the S16 mapping (see gru_s16.hip / odpd_s16.h): a wave = 16
// sequences, lane (n = sequence, q = unit quad) owns units 16kt + 4q + i of NT tiles; the seven HxH blocks (W_a, W_p1,
// W_p2, W_f, W_g acting on h; W_f, W_g acting on u) are exact-fp32 MFMA mat-vecs with operands streamed from an LDS
// table, their transposes carry the data gradients, and the weight gradients are 16x16 MFMA outer-product tiles fed
// through per-wave LDS transposes.  Per step (pgjanet.py:33-72):
//   a = tanh(W_a [h,|x|] + b), p1 = tanh(W_p1 [h,cos] + b), p2 = tanh(W_p2 [h,sin] + b),
//   u = a p1 p2 (1-a)(1-p1)(1-p2),  f = s(W_f [h,u] + b), g = tanh(W_g [h,u] + b),  h' = f h + (1-f) g,  y = W_o h' + b.
// The f rows (weights and bias) are stored pre-multiplied by -log2(e).  BPTT: checkpoint of h every kCkptStride steps
// (one float4 per lane and unit tile) + block recompute; dL/dx (frozen PA of a cascade) through the three scalar input
// columns and the polar features.  Used from the batch size that fills the chip with 16-sequence waves.
#include "odpd_s16.h"

namespace odpd {

// table groups for NT tiles of 16 hidden units.  Block r: 0 a_h, 1 p1_h, 2 p2_h, 3 f_h, 4 g_h, 5 f_u, 6 g_u
template <int NT>
struct J16 {
    static constexpr int FW = 0;                        // (r*NT + mt)*NT + kt : M_r[16mt+m][16kt+4q+e]   (f rows pre-scaled)
    static constexpr int TR = FW + 7 * NT * NT;         // (r*NT + mt)*NT + kt : M_r[16kt+4q+e][16mt+m]
    static constexpr int SC = TR + 7 * NT * NT;         // j*NT + mt : per-unit scalars sa sp1 sp2 ba bp1 bp2 bf bg [16mt+4q+e]
    static constexpr int WOUT = SC + 8 * NT;            // cc*NT + mt : fc_out[cc][16mt+4q+e]
    static constexpr int NG = WOUT + 2 * NT;
    static constexpr int kTiles = 7 * NT;               // dfp dgp dap dbp dcp hp u per unit tile
};

__device__ __forceinline__ float j16_block(const float* pl, const JanetLayout& L, int r, int o, int k) {
    const int H = L.H;
    if (o >= H || k >= H) return 0.0f;
    const int base = r == 0 ? L.o_wa : r == 1 ? L.o_wp1 : r == 2 ? L.o_wp2 : (r == 3 || r == 5) ? L.o_wf : L.o_wg;
    const int ld = r < 3 ? H + 1 : 2 * H, coff = r >= 5 ? H : 0;
    return pl[base + o * ld + coff + k];
}
template <int NT>
__device__ __forceinline__ float4 j16_entry(const float* pl, const JanetLayout& L, int grp, int m, int q) {
    using T = J16<NT>;
    const int H = L.H;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (grp < T::TR) {
            const int r = grp / (NT * NT), o = 16 * ((grp / NT) % NT) + m, k = 16 * (grp % NT) + 4 * q + e;
            v[e] = j16_block(pl, L, r, o, k) * ((r == 3 || r == 5) ? kNegLog2e : 1.0f);
        } else if (grp < T::SC) {
            const int g2 = grp - T::TR, r = g2 / (NT * NT), i = 16 * ((g2 / NT) % NT) + m, k = 16 * (g2 % NT) + 4 * q + e;
            v[e] = j16_block(pl, L, r, k, i);
        } else if (grp < T::WOUT) {
            const int g2 = grp - T::SC, j = g2 / NT, k = 16 * (g2 % NT) + 4 * q + e;
            float s = 0.0f;
            if (k < H) {
                if (j == 0) s = pl[L.o_wa + k * (H + 1) + H];
                else if (j == 1) s = pl[L.o_wp1 + k * (H + 1) + H];
                else if (j == 2) s = pl[L.o_wp2 + k * (H + 1) + H];
                else if (j == 3) s = pl[L.o_ba + k];
                else if (j == 4) s = pl[L.o_bp1 + k];
                else if (j == 5) s = pl[L.o_bp2 + k];
                else if (j == 6) s = pl[L.o_bf + k] * kNegLog2e;
                else s = pl[L.o_bg + k];
            }
            v[e] = s;
        } else {
            const int g2 = grp - T::WOUT, k = 16 * (g2 % NT) + 4 * q + e;
            v[e] = k < H ? pl[L.o_wo + (g2 / NT) * H + k] : 0.0f;
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float j16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ void j16_inputs(float2 xv, float& amp, float& ct, float& st) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y);
    amp = __builtin_amdgcn_sqrtf(a2);
    const float ia = fast_rcp(amp);
    ct = xv.x * ia; st = xv.y * ia;
}

// forward step; padded units have all-zero operands: a = p1 = p2 = 0 -> u = 0, f = 1/2, g = 0 -> h stays 0
template <int NT>
__device__ __forceinline__ void j16_cell_fwd(TabPtr tl, float amp, float ct, float st, f32x4 (&h)[NT], f32x4 (&an)[NT],
                                             f32x4 (&p1)[NT], f32x4 (&p2)[NT], f32x4 (&u)[NT], f32x4 (&f)[NT], f32x4 (&g)[NT]) {
    using T = J16<NT>;
    f32x4 pa[NT], pb[NT], pc[NT], pf[NT], pg[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        pa[mt] = fma4(as_f32x4(tab_ld(tl, (T::SC + 0 * NT + mt) * 64)), splat4(amp), as_f32x4(tab_ld(tl, (T::SC + 3 * NT + mt) * 64)));
        pb[mt] = fma4(as_f32x4(tab_ld(tl, (T::SC + 1 * NT + mt) * 64)), splat4(ct), as_f32x4(tab_ld(tl, (T::SC + 4 * NT + mt) * 64)));
        pc[mt] = fma4(as_f32x4(tab_ld(tl, (T::SC + 2 * NT + mt) * 64)), splat4(st), as_f32x4(tab_ld(tl, (T::SC + 5 * NT + mt) * 64)));
        pf[mt] = as_f32x4(tab_ld(tl, (T::SC + 6 * NT + mt) * 64));
        pg[mt] = as_f32x4(tab_ld(tl, (T::SC + 7 * NT + mt) * 64));
    }
    s16n_matvec<NT>(tl, T::FW + 0 * NT * NT, h, pa);
    s16n_matvec<NT>(tl, T::FW + 1 * NT * NT, h, pb);
    s16n_matvec<NT>(tl, T::FW + 2 * NT * NT, h, pc);
    s16n_matvec<NT>(tl, T::FW + 3 * NT * NT, h, pf);
    s16n_matvec<NT>(tl, T::FW + 4 * NT * NT, h, pg);
    const f32x4 one = splat4(1.0f);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        // PGJANET's tanh arguments are small (u <= 1/64 drives g; a, p1, p2 enter u as a product), so the RELATIVE accuracy
        // near 0 matters: 1 - 2 / (e + 1) alone has an absolute error of 1.9e-7 = 1e-4 relative at 1e-3, which showed as
        // 2e-5 in y -> the polynomial-below-0.3 variant of the row-rotated kernels for all four tanh
        an[mt] = tanh4_precise(pa[mt]); p1[mt] = tanh4_precise(pb[mt]); p2[mt] = tanh4_precise(pc[mt]);
        u[mt] = mul4(mul4(mul4(an[mt], p1[mt]), p2[mt]), mul4(mul4(sub4(one, an[mt]), sub4(one, p1[mt])), sub4(one, p2[mt])));
    }
    s16n_matvec<NT>(tl, T::FW + 5 * NT * NT, u, pf);
    s16n_matvec<NT>(tl, T::FW + 6 * NT * NT, u, pg);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        f[mt] = sigmoid4_prescaled(pf[mt]);
        g[mt] = tanh4_precise(pg[mt]);
        h[mt] = fma4(f[mt], sub4(h[mt], g[mt]), g[mt]);      // f h + (1 - f) g
    }
}

template <int NT>
__device__ __forceinline__ void j16_build_table(float* tab, const float* pl, const JanetLayout& L, int lane, int wave, int nwb) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < J16<NT>::NG; grp += nwb) t4[grp * 64 + lane] = j16_entry<NT>(pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
// (r06: 16-step chunks — the kernel's 96 registers allow four waves per SIMD, its 94 KB workgroup let one onto a CU)
constexpr int kJ16FwdChunk = 16;
template <int NT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void janet16_fwd_kernel(SeqArgs a) {
    using T = J16<NT>;
    constexpr int S = kCkptStride, CH = kJ16FwdChunk, kWave = 2 * 2 * 16 * (CH + 1);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const JanetLayout L = janet_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    j16_build_table<NT>(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const float bo0 = j16_uni(pl[L.o_bo]), bo1 = j16_uni(pl[L.o_bo + 1]);
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * (CH + 1);
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * NT * 64 + lane : nullptr;
        f32x4 h[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) h[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < a.T; t0 += CH) {
            const int len = min(CH, a.T - t0);
            wave_lds_fence();
            stage_in_ch<CH>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                float amp, ct, st;
                j16_inputs(xs[n * (CH + 1) + tt], amp, ct, st);
                f32x4 an[NT], p1[NT], p2[NT], u[NT], f[NT], g[NT];
                j16_cell_fwd<NT>(opaque(tl), amp, ct, st, h, an, p1, p2, u, f, g);
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
                    ODPD_EACH4 { s0 = __builtin_fmaf(w0[i], h[mt][i], s0); s1 = __builtin_fmaf(w1[i], h[mt][i], s1); }
                }
                const float y0 = quad_sum(s0) + bo0, y1 = quad_sum(s1) + bo1;
                if (q == 0) ys[n * (CH + 1) + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) ck[((size_t)(t1 / S) * NT + kt) * 64] = make_float4(h[kt][0], h[kt][1], h[kt][2], h[kt][3]);
                }
            }
            wave_lds_fence();
            stage_out_ch<CH>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------
template <int NT>
struct J16Grad {
    f32x4 t[7][NT][NT];                   // dW blocks in table order
    f32x4 ds[3][NT], db[5][NT], dwo[2][NT];
    float dbo[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int j = 0; j < 3; ++j) ds[j][a] = z4;
#pragma unroll
            for (int j = 0; j < 5; ++j) db[j][a] = z4;
            dwo[0][a] = dwo[1][a] = z4;
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int b = 0; b < NT; ++b) t[r][a][b] = z4;
        }
        dbo[0] = dbo[1] = 0.f;
    }
};

template <int NT, bool NW, bool DX, bool FULL>
__device__ __forceinline__ void j16_bwd_block(const SeqArgs& a, TabPtr tl0, J16Grad<NT>& G, const float2* xs, const float2* dys,
                                              float2* dxs, float* tiles, int n, int q, int tloc, int nstep, f32x4 (&h)[NT],
                                              f32x4 (&dh)[NT]) {
    using T = J16<NT>;
    constexpr int S = kCkptStride;
    f32x4 hp_s[S][NT], an_s[S][NT], p1_s[S][NT], p2_s[S][NT], u_s[S][NT], f_s[S][NT], g_s[S][NT];
    TabPtr tl = opaque(tl0);
#pragma unroll
    for (int si = 0; si < S; ++si) {
        if (FULL || si < nstep) {
            float amp, ct, st;
            j16_inputs(xs[n * kChunkPad + tloc + si], amp, ct, st);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) hp_s[si][kt] = h[kt];
            j16_cell_fwd<NT>(tl, amp, ct, st, h, an_s[si], p1_s[si], p2_s[si], u_s[si], f_s[si], g_s[si]);
        }
    }
    tl = opaque(tl0);
    auto tile = [tiles](int qty, int kt) { return tiles + (qty * NT + kt) * kTileFloats; };   // 0 dfp 1 dgp 2 dap 3 dbp 4 dcp 5 hp 6 u
    const f32x4 one = splat4(1.0f);
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            const int tt = tloc + si;
            const float2 dyv = dys[n * kChunkPad + tt];
            float amp, ct, st;
            j16_inputs(xs[n * kChunkPad + tt], amp, ct, st);
            if constexpr (NW) { G.dbo[0] += q == 0 ? dyv.x : 0.0f; G.dbo[1] += q == 0 ? dyv.y : 0.0f; }
            f32x4 dfp[NT], dgp[NT], dhp[NT], du[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
                const f32x4 hp = hp_s[si][mt], f = f_s[si][mt], g = g_s[si][mt];
                const f32x4 dht = add4(dh[mt], fma4(splat4(dyv.x), w0, mul4(w1, splat4(dyv.y))));
                const f32x4 hmg = sub4(hp, g);
                dfp[mt] = mul4(mul4(dht, hmg), mul4(f, sub4(one, f)));
                f32x4 omg2;
                ODPD_EACH4 omg2[i] = __builtin_fmaf(-g[i], g[i], 1.0f);
                dgp[mt] = mul4(mul4(dht, sub4(one, f)), omg2);
                dhp[mt] = mul4(dht, f);
                du[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (NW) {
                    const f32x4 ht = fma4(f, hmg, g);
                    G.dwo[0][mt] = fma4(splat4(dyv.x), ht, G.dwo[0][mt]);
                    G.dwo[1][mt] = fma4(splat4(dyv.y), ht, G.dwo[1][mt]);
                    G.db[3][mt] = add4(G.db[3][mt], dfp[mt]); G.db[4][mt] = add4(G.db[4][mt], dgp[mt]);
                }
            }
            s16n_matvec<NT>(tl, T::TR + 3 * NT * NT, dfp, dhp);
            s16n_matvec<NT>(tl, T::TR + 4 * NT * NT, dgp, dhp);
            s16n_matvec<NT>(tl, T::TR + 5 * NT * NT, dfp, du);
            s16n_matvec<NT>(tl, T::TR + 6 * NT * NT, dgp, du);
            f32x4 dap[NT], dbp[NT], dcp[NT];
            float ga = 0.0f, gc = 0.0f, gs = 0.0f;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const f32x4 an = an_s[si][mt], p1 = p1_s[si][mt], p2 = p2_s[si][mt];
                const f32x4 Aa = mul4(an, sub4(one, an)), Ab = mul4(p1, sub4(one, p1)), Ac = mul4(p2, sub4(one, p2));
                f32x4 ta, tb, tc;
                ODPD_EACH4 {
                    ta[i] = __builtin_fmaf(-2.0f, an[i], 1.0f) * __builtin_fmaf(-an[i], an[i], 1.0f);
                    tb[i] = __builtin_fmaf(-2.0f, p1[i], 1.0f) * __builtin_fmaf(-p1[i], p1[i], 1.0f);
                    tc[i] = __builtin_fmaf(-2.0f, p2[i], 1.0f) * __builtin_fmaf(-p2[i], p2[i], 1.0f);
                }
                dap[mt] = mul4(mul4(du[mt], ta), mul4(Ab, Ac));
                dbp[mt] = mul4(mul4(du[mt], tb), mul4(Aa, Ac));
                dcp[mt] = mul4(mul4(du[mt], tc), mul4(Aa, Ab));
                if constexpr (NW) {
                    G.db[0][mt] = add4(G.db[0][mt], dap[mt]); G.db[1][mt] = add4(G.db[1][mt], dbp[mt]); G.db[2][mt] = add4(G.db[2][mt], dcp[mt]);
                    G.ds[0][mt] = fma4(dap[mt], splat4(amp), G.ds[0][mt]);
                    G.ds[1][mt] = fma4(dbp[mt], splat4(ct), G.ds[1][mt]);
                    G.ds[2][mt] = fma4(dcp[mt], splat4(st), G.ds[2][mt]);
                }
                if constexpr (DX) {
                    const f32x4 sa = as_f32x4(tab_ld(tl, (T::SC + 0 * NT + mt) * 64)), sp1 = as_f32x4(tab_ld(tl, (T::SC + 1 * NT + mt) * 64)),
                                sp2 = as_f32x4(tab_ld(tl, (T::SC + 2 * NT + mt) * 64));
                    ODPD_EACH4 {
                        ga = __builtin_fmaf(sa[i], dap[mt][i], ga);
                        gc = __builtin_fmaf(sp1[i], dbp[mt][i], gc);
                        gs = __builtin_fmaf(sp2[i], dcp[mt][i], gs);
                    }
                }
            }
            if constexpr (DX) {
                const float2 gx = polar_sample_bwd(amp, ct, st, quad_sum(ga), quad_sum(gc), quad_sum(gs));
                if (q == 0) dxs[n * kChunkPad + tt] = gx;
            }
            s16n_matvec<NT>(tl, T::TR + 0 * NT * NT, dap, dhp);
            s16n_matvec<NT>(tl, T::TR + 1 * NT * NT, dbp, dhp);
            s16n_matvec<NT>(tl, T::TR + 2 * NT * NT, dcp, dhp);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) dh[mt] = dhp[mt];
            if constexpr (NW) {
                // weight gradients: dM_r += d_r^T (x) src_r  (src = h_prev for r < 5, u for r = 5, 6)
                wave_lds_fence();
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    tile_put(tile(0, kt), n, q, dfp[kt]); tile_put(tile(1, kt), n, q, dgp[kt]);
                    tile_put(tile(2, kt), n, q, dap[kt]); tile_put(tile(3, kt), n, q, dbp[kt]); tile_put(tile(4, kt), n, q, dcp[kt]);
                    tile_put(tile(5, kt), n, q, hp_s[si][kt]); tile_put(tile(6, kt), n, q, u_s[si][kt]);
                }
                wave_lds_fence();
                float hT[NT][4], uT[NT][4];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) { tile_get(tile(5, kt), n, q, hT[kt]); tile_get(tile(6, kt), n, q, uT[kt]); }
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    float fT[4], gT[4], aT[4], bT[4], cT[4];
                    tile_get(tile(0, mt), n, q, fT); tile_get(tile(1, mt), n, q, gT);
                    tile_get(tile(2, mt), n, q, aT); tile_get(tile(3, mt), n, q, bT); tile_get(tile(4, mt), n, q, cT);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            G.t[0][mt][nt] = mfma4(aT[c], hT[nt][c], G.t[0][mt][nt]);
                            G.t[1][mt][nt] = mfma4(bT[c], hT[nt][c], G.t[1][mt][nt]);
                            G.t[2][mt][nt] = mfma4(cT[c], hT[nt][c], G.t[2][mt][nt]);
                            G.t[3][mt][nt] = mfma4(fT[c], hT[nt][c], G.t[3][mt][nt]);
                            G.t[4][mt][nt] = mfma4(gT[c], hT[nt][c], G.t[4][mt][nt]);
                            G.t[5][mt][nt] = mfma4(fT[c], uT[nt][c], G.t[5][mt][nt]);
                            G.t[6][mt][nt] = mfma4(gT[c], uT[nt][c], G.t[6][mt][nt]);
                        }
                }
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void j16_write_row(float* prow, const JanetLayout& L, J16Grad<NT>& G, int lane, int n, int q) {
    const int H = L.H;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
    const int base[7] = {L.o_wa, L.o_wp1, L.o_wp2, L.o_wf, L.o_wg, L.o_wf, L.o_wg};
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int o = 16 * mt + 4 * q + rr;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int ld = r < 3 ? H + 1 : 2 * H, coff = r >= 5 ? H : 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (o < H && 16 * nt + n < H) prow[base[r] + o * ld + coff + 16 * nt + n] = G.t[r][mt][nt][rr];
            }
            float ds[3], db[5];
#pragma unroll
            for (int j = 0; j < 3; ++j) ds[j] = row_sum16(G.ds[j][mt][rr]);
#pragma unroll
            for (int j = 0; j < 5; ++j) db[j] = row_sum16(G.db[j][mt][rr]);
            const float w0 = row_sum16(G.dwo[0][mt][rr]), w1 = row_sum16(G.dwo[1][mt][rr]);
            if (n == 0 && o < H) {
                prow[L.o_wa + o * (H + 1) + H] = ds[0]; prow[L.o_wp1 + o * (H + 1) + H] = ds[1]; prow[L.o_wp2 + o * (H + 1) + H] = ds[2];
                prow[L.o_ba + o] = db[0]; prow[L.o_bp1 + o] = db[1]; prow[L.o_bp2 + o] = db[2];
                prow[L.o_bf + o] = db[3]; prow[L.o_bg + o] = db[4];
                prow[L.o_wo + o] = w0; prow[L.o_wo + H + o] = w1;
            }
        }
    const float b0 = row_sum16(G.dbo[0]), b1 = row_sum16(G.dbo[1]);      // accumulated on the q == 0 lanes only
    if (lane == 0) { prow[L.o_bo] = b0; prow[L.o_bo + 1] = b1; }
}

template <int NT, bool NW, bool DX>
__global__ __launch_bounds__(256, 1) void janet16_bwd_kernel(SeqArgs a) {
    using T = J16<NT>;
    constexpr int S = kCkptStride;
    constexpr int kWave = (DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? T::kTiles * kTileFloats : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const JanetLayout L = janet_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    j16_build_table<NT>(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * kChunkPad;
    float2* dxs = dys + 16 * kChunkPad;
    float* tiles = reinterpret_cast<float*>(dys + (DX ? 2 : 1) * 16 * kChunkPad);
    J16Grad<NT> G;
    if constexpr (NW) G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * NT * 64 + lane;
        f32x4 dh[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) dh[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
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
            f32x4 h0[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) h0[kt] = blk ? as_f32x4(ck[((size_t)blk * NT + kt) * 64]) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (nstep == S) j16_bwd_block<NT, NW, DX, true>(a, tl, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, h0, dh);
            else j16_bwd_block<NT, NW, DX, false>(a, tl, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, h0, dh);
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
        j16_write_row<NT>(smem + wave * P4, L, G, lane, n, q);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
bool janet_uses_s16(const odpd_model_t* m, int B) {
    if (m->backbone != ODPD_PGJANET || m->hidden > 16) return false;
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 16L * 4 * device_cus();
    return B >= min_batch;
}
static LaunchShape j16_shape(int ngroups, int waves) {
    LaunchShape ls;
    ls.waves = waves;
    const int need = (ngroups + waves - 1) / waves, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
int janet_s16_rows(const odpd_model_t* m, int B) {
    (void)m;
    return j16_shape((B + 15) / 16, 4).grid;
}
int64_t janet_s16_ckpt_floats(const odpd_model_t* m, int B, int T) {
    (void)m;
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * 256;
}
template <bool NW, bool DX>
static int j16_launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    using T = J16<1>;
    const LaunchShape ls = j16_shape(a.ngroups, 4);
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(T::NG) +
                  (size_t)ls.waves * ((DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? T::kTiles * kTileFloats : 0))) * sizeof(float);
    if (NW && lds < reduce_scratch_bytes(P, ls.waves)) lds = reduce_scratch_bytes(P, ls.waves);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = janet16_bwd_kernel<1, NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
int janet_s16_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    using T = J16<1>;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = janet_layout(m->hidden).P;
    if (mode == 1) {
        LaunchShape ls = j16_shape(a.ngroups, a.ngroups <= 4 * device_cus() ? 4 : 8);
        const size_t lds = ((size_t)pad4(P) + s16_tab_floats(T::NG) + (size_t)ls.waves * (2 * 2 * 16 * (kJ16FwdChunk + 1))) * sizeof(float);
        if (ls.waves == 8 && 2 * lds <= kMaxLds) {      // two eight-wave workgroups per CU = four waves per SIMD
            const int need = (a.ngroups + 7) / 8, cap = 2 * device_cus();
            ls.grid = need < cap ? need : cap;
        }
        auto k = janet16_fwd_kernel<1>;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    }
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (!nw && !dx) return ODPD_EINVAL;
    if (nw && dx) return j16_launch_bwd<true, true>(st, a, P);
    if (nw) return j16_launch_bwd<true, false>(st, a, P);
    return j16_launch_bwd<false, true>(st, a, P);
}

}  // namespace odpd
