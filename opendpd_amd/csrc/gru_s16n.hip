// gru_s16n.hip — S16 kernels of the GRU family for hidden sizes up to 16 * NT (NT = 2: hidden 17..32, e.g. the
// reference's default PA size 23; backbones/{gru,dgru,qgru,qgru_amp1}.py + modules/train_funcs.py:33-39).
//
// Same lane mapping as gru_s16.hip — a wave holds 16 sequences, lane (n, q) — but the lane owns NT x 4 hidden units
// (unit 16*kt + 4q + i, kt < NT) and every mat-vec is NT x NT tiles of exact-fp32 v_mfma_f32_16x16x4_f32:
//   pre[16mt + .][n] += sum_{kt,c} W[16mt + m][16kt + 4q + c] h_n[16kt + 4q + c]     (A from LDS, B = own h[kt][c])
// 16 * NT * 16 * NT weights no longer fit registers next to the BPTT block state, so the MFMA A operands are STREAMED:
// one ds_read_b128 per (gate, M tile, K tile) delivers the four K-chunk operands right before their four MFMAs.
// Weight gradients: per step the wave transposes d(gates), h and the feature slots through LDS tiles (one tile per
// quantity and unit tile) and accumulates NT x NT tiles per gate.  BPTT: h checkpoints every kCkptStride steps in an
// HBM workspace ([task][ckpt][kt][lane] float4), block recompute.  One wave per SIMD (the block state of NT = 2 alone
// is 168 registers).  Kernels: fused train step, split forward, split backward (weight partials and / or dL/dx).
#include "odpd_s16.h"

namespace odpd {

// table groups (float4 per lane each); e = element of the float4, m = lane & 15, q = lane >> 4
template <int NT>
struct S16N {
    static constexpr int HH = 0;                        // (g*NT + mt)*NT + kt : W_hg[16mt+m][16kt+4q+e]  (r,z pre-scaled)
    static constexpr int IH = HH + 3 * NT * NT;         // g*NT + mt          : slot 4e+q of [W_ig | b] for e < NCH
    static constexpr int BHN = IH + 3 * NT;             // mt                 : b_hn[16mt+4q+e]
    static constexpr int HHT = BHN + NT;                // (g*NT + mt)*NT + kt : W_hg[16kt+4q+e][16mt+m]
    static constexpr int HID = HHT + 3 * NT * NT;       // mt*NT + kt         : fc_hid[16mt+m][16kt+4q+e]
    static constexpr int HIDT = HID + NT * NT;          // mt*NT + kt         : fc_hid[16kt+4q+e][16mt+m]
    static constexpr int BHID = HIDT + NT * NT;         // mt
    static constexpr int WOUT = BHID + NT;              // cc*NT + mt         : fc_out[cc][16mt+4q+e]
    static constexpr int WOUTF = WOUT + 2 * NT;         // fc_out feature / bias slots (cc = e>>1, chunk = e&1)
    static constexpr int NG = WOUTF + 1;
    static constexpr int IHT = NG;                      // g*NT + kt          : W_ig[16kt+4q+e][m]  (m = feature slot)
    static constexpr int WFD = IHT + 3 * NT;            // cc                 : fc_out[cc][H + 4q+e]
    static constexpr int NG_DX = WFD + 2;
    static constexpr int kTiles = 6 * NT + 1;           // drp dzp dnp dgh hp dhid per unit tile + feature tile
    static constexpr int kWaveFloats = 2 * 2 * 16 * kChunkPad + kTiles * kTileFloats;
};

// Which hidden unit sits in slot s (= 4 q + e on the owning lane, = row / column s of a 16x16 operand tile) of unit tile t.
// Full tiles: unit 16 t + s.  The LAST tile is filled element-major (slot 4 q + e <- rank 4 e + q), so that its H - 16 (NT-1)
// units occupy the elements e < ceil((H - 16 (NT-1)) / 4) of every quad: the K-chunks above that count multiply zeros only
// and are skipped (`nck` of s16n_matvec) — hidden 23 runs 6 of its 8 K-chunks.
template <int NT>
__device__ __forceinline__ int s16n_unit(int t, int s) {
    return t < NT - 1 ? 16 * t + s : 16 * t + 4 * (s & 3) + (s >> 2);
}
template <int NT>
__host__ __device__ inline int s16n_last_chunks(int H) { return (H - 16 * (NT - 1) + 3) / 4; }

template <int FM, bool DG, int NT>
__device__ __forceinline__ float4 s16n_entry(const float* pl, const GruLayout& L, int grp, int m, int q) {
    using T = S16N<NT>;
    constexpr int F = S16Cfg<FM>::F;
    const int H = L.H, OW = DG ? H + 6 : H;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (grp < T::IH) {
            const int g = grp / (NT * NT), mt = (grp / NT) % NT, kt = grp % NT, o = s16n_unit<NT>(mt, m), k = s16n_unit<NT>(kt, 4 * q + e);
            v[e] = (o < H && k < H) ? pl[L.o_w_hh + (g * H + o) * H + k] * (g < 2 ? kNegLog2e : 1.0f) : 0.0f;
        } else if (grp < T::BHN) {
            const int g = (grp - T::IH) / NT, mt = (grp - T::IH) % NT;
            v[e] = (g < 2 ? kNegLog2e : 1.0f) * s16_wih_slot<FM, DG>(pl, L, g, e, s16n_unit<NT>(mt, m), q);
        } else if (grp < T::HHT) {
            const int u = s16n_unit<NT>(grp - T::BHN, 4 * q + e);
            v[e] = u < H ? pl[L.o_b_hh + 2 * H + u] : 0.0f;
        } else if (grp < T::HID) {
            const int r = grp - T::HHT, g = r / (NT * NT), mt = (r / NT) % NT, kt = r % NT, i = s16n_unit<NT>(mt, m), k = s16n_unit<NT>(kt, 4 * q + e);
            v[e] = (i < H && k < H) ? pl[L.o_w_hh + (g * H + k) * H + i] : 0.0f;
        } else if (grp < T::HIDT) {
            const int r = grp - T::HID, o = s16n_unit<NT>(r / NT, m), k = s16n_unit<NT>(r % NT, 4 * q + e);
            v[e] = (DG && o < H && k < H) ? pl[L.o_w_hid + o * H + k] : 0.0f;
        } else if (grp < T::BHID) {
            const int r = grp - T::HIDT, i = s16n_unit<NT>(r / NT, m), k = s16n_unit<NT>(r % NT, 4 * q + e);
            v[e] = (DG && i < H && k < H) ? pl[L.o_w_hid + k * H + i] : 0.0f;
        } else if (grp < T::WOUT) {
            const int u = s16n_unit<NT>(grp - T::BHID, 4 * q + e);
            v[e] = (DG && u < H) ? pl[L.o_b_hid + u] : 0.0f;
        } else if (grp < T::WOUTF) {
            const int r = grp - T::WOUT, u = s16n_unit<NT>(r % NT, 4 * q + e);
            v[e] = u < H ? pl[L.o_w_out + (r / NT) * OW + u] : 0.0f;
        } else if (grp == T::WOUTF) {
            v[e] = s16_woutf_slot<FM, DG>(pl, L, e >> 1, e & 1, q);
        } else if (grp < T::WFD) {
            const int r = grp - T::IHT, g = r / NT, k = s16n_unit<NT>(r % NT, 4 * q + e);
            v[e] = (m < F && k < H) ? pl[L.o_w_ih + (g * H + k) * F + m] : 0.0f;
        } else {
            const int k = 4 * q + e;
            v[e] = (DG && k < F) ? pl[L.o_w_out + (grp - T::WFD) * OW + H + k] : 0.0f;
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <int FM, bool DG, int NT>
__device__ __forceinline__ void s16n_fill_table(float* tab, const float* pl, const GruLayout& L, int lane, int wave, int nwb, int ngroups) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < ngroups; grp += nwb) t4[grp * 64 + lane] = s16n_entry<FM, DG, NT>(pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}

template <int FM, bool DG, int NT, int NCK = 4>
__device__ __forceinline__ void s16n_cell_fwd(TabPtr tl, const float (&fs)[S16Cfg<FM>::NCH], f32x4 (&h)[NT], f32x4 (&r)[NT],
                                              f32x4 (&z)[NT], f32x4 (&n)[NT], f32x4 (&g)[NT]) {
    using T = S16N<NT>;
    constexpr int NCH = S16Cfg<FM>::NCH;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 ar[NT], az[NT], an[NT], ah[NT];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        ar[mt] = zero; az[mt] = zero; an[mt] = zero;
        ah[mt] = as_f32x4(tab_ld(tl, (T::BHN + mt) * 64));
        const float4 wr = tab_ld(tl, (T::IH + 0 * NT + mt) * 64), wz = tab_ld(tl, (T::IH + 1 * NT + mt) * 64),
                     wn = tab_ld(tl, (T::IH + 2 * NT + mt) * 64);
        ar[mt] = mfma4(wr.x, fs[0], ar[mt]); az[mt] = mfma4(wz.x, fs[0], az[mt]); an[mt] = mfma4(wn.x, fs[0], an[mt]);
        if constexpr (NCH > 1) {
            ar[mt] = mfma4(wr.y, fs[NCH - 1], ar[mt]); az[mt] = mfma4(wz.y, fs[NCH - 1], az[mt]); an[mt] = mfma4(wn.y, fs[NCH - 1], an[mt]);
        }
    }
    s16n_matvec<NT, NCK>(tl, T::HH + 0 * NT * NT, h, ar);
    s16n_matvec<NT, NCK>(tl, T::HH + 1 * NT * NT, h, az);
    s16n_matvec<NT, NCK>(tl, T::HH + 2 * NT * NT, h, ah);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        r[mt] = sigmoid4_prescaled(ar[mt]);
        z[mt] = sigmoid4_prescaled(az[mt]);
        g[mt] = ah[mt];
        n[mt] = tanh4_for<FM>(fma4(r[mt], ah[mt], an[mt]));
        h[mt] = fma4(z[mt], sub4(h[mt], n[mt]), n[mt]);
    }
}

// head: act (relu(fc_hid h) for DGRU, h otherwise) and the two fc_out partial sums of the lane
template <int FM, bool DG, int NT, int NCK = 4>
__device__ __forceinline__ void s16n_head(TabPtr tl, const f32x4 (&ht)[NT], const float (&fs)[S16Cfg<FM>::NCH], f32x4 (&hid)[NT],
                                          f32x4 (&act)[NT], float& p0, float& p1) {
    using T = S16N<NT>;
    constexpr int NCH = S16Cfg<FM>::NCH;
    if constexpr (DG) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) hid[mt] = as_f32x4(tab_ld(tl, (T::BHID + mt) * 64));
        s16n_matvec<NT, NCK>(tl, T::HID, ht, hid);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) ODPD_EACH4 act[mt][i] = relu_(hid[mt][i]);
    } else {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) { act[mt] = ht[mt]; hid[mt] = ht[mt]; }
    }
    const float4 wf = tab_ld(tl, T::WOUTF * 64);
    p0 = wf.x * fs[0]; p1 = wf.z * fs[0];
    if constexpr (NCH > 1) { p0 = __builtin_fmaf(wf.y, fs[NCH - 1], p0); p1 = __builtin_fmaf(wf.w, fs[NCH - 1], p1); }
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
        ODPD_EACH4 p0 = __builtin_fmaf(w0[i], act[mt][i], p0);
        ODPD_EACH4 p1 = __builtin_fmaf(w1[i], act[mt][i], p1);
    }
}

template <bool DG, int NT>
struct S16NGrad {
    f32x4 thh[3][NT][NT], tih[3][NT], thid[NT][NT];
    f32x4 db_hn[NT], db_hid[NT], dwout[2][NT];
    float dwf[2][2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            db_hn[a] = z4; db_hid[a] = z4; dwout[0][a] = z4; dwout[1][a] = z4;
#pragma unroll
            for (int g = 0; g < 3; ++g) tih[g][a] = z4;
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                thid[a][b] = z4;
#pragma unroll
                for (int g = 0; g < 3; ++g) thh[g][a][b] = z4;
            }
        }
        dwf[0][0] = dwf[0][1] = dwf[1][0] = dwf[1][1] = 0.f;
    }
};

// one block of <= S steps (see s16_block in gru_s16.hip for the argument conventions)
template <int FM, bool DG, int NT, bool FUSED, bool NW, bool DX, bool FULL, int NCK = 4>
__device__ __forceinline__ void s16n_block(const SeqArgs& a, TabPtr tl0, const float (&oh)[4], S16NGrad<DG, NT>& G,
                                           const float2* xs, const float2* ts, float2* dxs, float* tiles, int n, int q, int tloc,
                                           int nstep, bool valid, bool last_blk, const f32x4 (&h0)[NT], f32x4 (&dh)[NT],
                                           float (&hTn)[NT][4], float& loss_acc) {
    using T = S16N<NT>;
    constexpr int NCH = S16Cfg<FM>::NCH, S = kCkptStride, F = S16Cfg<FM>::F;
    f32x4 h[NT], hp_s[S][NT], r_s[S][NT], z_s[S][NT], n_s[S][NT], g_s[S][NT];
    float fs_s[S][NCH];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) h[kt] = h0[kt];
    TabPtr tl = opaque(tl0);
#pragma unroll
    for (int st = 0; st < S; ++st) {
        if (FULL || st < nstep) {
            const float2 xv = xs[n * kChunkPad + tloc + st];
            s16_slots<FM>(xv.x, xv.y, oh, fs_s[st]);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) hp_s[st][kt] = h[kt];
            s16n_cell_fwd<FM, DG, NT, NCK>(tl, fs_s[st], h, r_s[st], z_s[st], n_s[st], g_s[st]);
        }
    }
    tl = opaque(tl0);
    auto tile = [tiles](int qty, int kt) { return tiles + (qty * NT + kt) * kTileFloats; };   // qty: 0 drp 1 dzp 2 dnp 3 dgh 4 hp 5 dhid
    float* t_f = tiles + 6 * NT * kTileFloats;
    if (NW && DG && last_blk) {
        wave_lds_fence();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) tile_put(tile(4, kt), n, q, h[kt]);
        wave_lds_fence();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) tile_get(tile(4, kt), n, q, hTn[kt]);
    }
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, valid ? a.inv_count : 0.0f, valid && q == 0);
#pragma unroll
    for (int st = S - 1; st >= 0; --st) {
        if (FULL || st < nstep) {
            f32x4 ht[NT], hid[NT], act[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) ht[kt] = fma4(z_s[st][kt], sub4(hp_s[st][kt], n_s[st][kt]), n_s[st][kt]);
            float p0, p1;
            s16n_head<FM, DG, NT, NCK>(tl, ht, fs_s[st], hid, act, p0, p1);
            const float2 tv = ts[n * kChunkPad + tloc + st];
            float dy0 = tv.x, dy1 = tv.y;
            if constexpr (FUSED) {
                const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                const float d0 = y0 - tv.x, d1 = y1 - tv.y;
                s16_loss(lossc, d0, d1, dy0, dy1, loss_acc);
            }
            if constexpr (NW) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    G.dwf[0][c] = __builtin_fmaf(dy0, fs_s[st][c], G.dwf[0][c]);
                    G.dwf[1][c] = __builtin_fmaf(dy1, fs_s[st][c], G.dwf[1][c]);
                }
            }
            f32x4 dht[NT], dhid[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
                if constexpr (NW) {
                    G.dwout[0][mt] = fma4(splat4(dy0), act[mt], G.dwout[0][mt]);
                    G.dwout[1][mt] = fma4(splat4(dy1), act[mt], G.dwout[1][mt]);
                }
                const f32x4 dact = fma4(splat4(dy0), w0, mul4(w1, splat4(dy1)));
                if constexpr (DG) {
                    ODPD_EACH4 dhid[mt][i] = dact[i] * relu_gate(hid[mt][i]);
                    if constexpr (NW) G.db_hid[mt] = add4(G.db_hid[mt], dhid[mt]);
                    dht[mt] = dh[mt];
                } else {
                    dht[mt] = add4(dh[mt], dact);
                    dhid[mt] = dact;
                }
            }
            if constexpr (DG) s16n_matvec<NT, NCK>(tl, T::HIDT, dhid, dht);
            f32x4 drp[NT], dzp[NT], dnp[NT], dgh[NT], acc[NT];
            const f32x4 one = splat4(1.0f);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const f32x4 r = r_s[st][mt], z = z_s[st][mt], nn = n_s[st][mt];
                const f32x4 omz = sub4(one, z), omr = sub4(one, r);
                f32x4 omn2;
                ODPD_EACH4 omn2[i] = __builtin_fmaf(-nn[i], nn[i], 1.0f);
                const f32x4 dn = mul4(dht[mt], omz);
                acc[mt] = mul4(dht[mt], z);
                dnp[mt] = mul4(dn, omn2);
                dgh[mt] = mul4(dnp[mt], r);
                dzp[mt] = mul4(mul4(sub4(hp_s[st][mt], nn), z), dn);
                drp[mt] = mul4(mul4(dgh[mt], g_s[st][mt]), omr);
                if constexpr (NW) G.db_hn[mt] = add4(G.db_hn[mt], dgh[mt]);
            }
            s16n_matvec<NT, NCK>(tl, T::HHT + 0 * NT * NT, drp, acc);
            s16n_matvec<NT, NCK>(tl, T::HHT + 1 * NT * NT, dzp, acc);
            s16n_matvec<NT, NCK>(tl, T::HHT + 2 * NT * NT, dgh, acc);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) dh[mt] = acc[mt];
            if constexpr (DX) {
                f32x4 ds = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const float4 wr = tab_ld(tl, (T::IHT + 0 * NT + kt) * 64), wz = tab_ld(tl, (T::IHT + 1 * NT + kt) * 64),
                                 wn = tab_ld(tl, (T::IHT + 2 * NT + kt) * 64);
                    const f32x4 wrv = as_f32x4(wr), wzv = as_f32x4(wz), wnv = as_f32x4(wn);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (kt < NT - 1 || c < NCK) {
                            ds = mfma4(wrv[c], drp[kt][c], ds); ds = mfma4(wzv[c], dzp[kt][c], ds); ds = mfma4(wnv[c], dnp[kt][c], ds);
                        }
                }
                if constexpr (DG) {
                    const f32x4 f0 = as_f32x4(tab_ld(tl, (T::WFD + 0) * 64)), f1 = as_f32x4(tab_ld(tl, (T::WFD + 1) * 64));
                    ds = fma4(splat4(dy0), f0, fma4(splat4(dy1), f1, ds));
                }
                float df[F];
#pragma unroll
                for (int j = 0; j < F; ++j) df[j] = j < 4 ? ds[j & 3] : swap16(ds[j & 3]);
                const float2 xv = xs[n * kChunkPad + tloc + st];
                float dI, dQ;
                feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
                if (q == 0) dxs[n * kChunkPad + tloc + st] = make_float2(dI, dQ);
            }
            if constexpr (NW) {
                wave_lds_fence();
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    tile_put(tile(0, kt), n, q, drp[kt]);
                    tile_put(tile(1, kt), n, q, dzp[kt]);
                    tile_put(tile(2, kt), n, q, dnp[kt]);
                    tile_put(tile(3, kt), n, q, dgh[kt]);
                    tile_put(tile(4, kt), n, q, hp_s[st][kt]);
                    if constexpr (DG) tile_put(tile(5, kt), n, q, dhid[kt]);
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c) t_f[n * kTilePitch + 4 * c + q] = fs_s[st][c];
                wave_lds_fence();
                float fT[4], hT[NT][4];
                tile_get(t_f, n, q, fT);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) tile_get(tile(4, kt), n, q, hT[kt]);
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    float rT[4], zT[4], nT[4], gT[4], dT[4];
                    tile_get(tile(0, mt), n, q, rT);
                    tile_get(tile(1, mt), n, q, zT);
                    tile_get(tile(2, mt), n, q, nT);
                    tile_get(tile(3, mt), n, q, gT);
                    if constexpr (DG) tile_get(tile(5, mt), n, q, dT);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        G.tih[0][mt] = mfma4(rT[c], fT[c], G.tih[0][mt]);
                        G.tih[1][mt] = mfma4(zT[c], fT[c], G.tih[1][mt]);
                        G.tih[2][mt] = mfma4(nT[c], fT[c], G.tih[2][mt]);
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            G.thh[0][mt][nt] = mfma4(rT[c], hT[nt][c], G.thh[0][mt][nt]);
                            G.thh[1][mt][nt] = mfma4(zT[c], hT[nt][c], G.thh[1][mt][nt]);
                            G.thh[2][mt][nt] = mfma4(gT[c], hT[nt][c], G.thh[2][mt][nt]);
                            if constexpr (DG) G.thid[mt][nt] = mfma4(dT[c], hTn[nt][c], G.thid[mt][nt]);
                        }
                    }
                }
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) hTn[kt][c] = hT[kt][c];
            }
        }
    }
}

template <int FM, bool DG, int NT>
__device__ __forceinline__ void s16n_write_row(float* prow, const GruLayout& L, S16NGrad<DG, NT>& G, int n, int q, float loss_acc) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH;
    const int H = L.H, OW = DG ? H + 6 : H;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = s16n_unit<NT>(mt, 4 * q + rr);
            if (i < H) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float v = G.tih[g][mt][rr];
                    if (n < F) prow[L.o_w_ih + (g * H + i) * F + n] = v;
                    else if (n == F) {
                        prow[L.o_b_ih + g * H + i] = v;
                        if (g < 2) prow[L.o_b_hh + g * H + i] = v;
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (s16n_unit<NT>(nt, n) < H) prow[L.o_w_hh + (g * H + i) * H + s16n_unit<NT>(nt, n)] = G.thh[g][mt][nt][rr];
                }
                if constexpr (DG) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (s16n_unit<NT>(nt, n) < H) prow[L.o_w_hid + i * H + s16n_unit<NT>(nt, n)] = G.thid[mt][nt][rr];
                }
            }
            const float bhn = row_sum16(G.db_hn[mt][rr]), bhid = row_sum16(G.db_hid[mt][rr]);
            const float w0 = row_sum16(G.dwout[0][mt][rr]), w1 = row_sum16(G.dwout[1][mt][rr]);
            if (n == 0 && i < H) {
                prow[L.o_b_hh + 2 * H + i] = bhn;
                prow[L.o_w_out + i] = w0;
                prow[L.o_w_out + OW + i] = w1;
                if constexpr (DG) prow[L.o_b_hid + i] = bhid;
            }
        }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = 4 * c + q;
            const float v = row_sum16(G.dwf[cc][c]);
            if (n == 0) {
                if (DG && k < F) prow[L.o_w_out + cc * OW + H + k] = v;
                else if (k == F) prow[L.o_b_out + cc] = v;
            }
        }
    const float lp = row_sum16(loss_acc);
    if (n == 0 && q == 0) {
        prow[L.P] = lp;
        prow[L.P + 1] = 0.f; prow[L.P + 2] = 0.f; prow[L.P + 3] = 0.f;
    }
}

// MODE 0: fused train (x, target -> partials; checkpoints in `ckpt` workspace)   1: forward (y, optional ckpt)
// MODE 2: backward from dy (partials if NW, dx if DX)
// the frozen-model variants (no weight-gradient accumulators) run two waves per SIMD like the forward kernel.
// History of the one exception that r02/r03 carried here: the fused frozen-PA step (MODE 0 without NW) with four K-chunks in the last unit
// tile (hidden 25..32) spills 160..220 B per lane under the 256-register cap of an eight-wave workgroup, and its DGRU instantiation
// computed 1e-2 .. 2e-1 wrong losses / gradients in that build (exact at one wave per SIMD, at -O1, with -enable-post-misched=false), so the
// launch shape kept those sizes at one wave per SIMD.  r04 root cause (tools/asm_mfma_hazards.py, tools/exp_s16n8w.py): relu_ was an
// inline-asm v_max_f32 with a write-only output; in exactly that instantiation the allocator gave the output the register that a v_mfma
// issued three slots earlier was still reading as its C operand, and the hazard recognizer does not look inside asm.  With relu_ as a
// builtin (odpd_s16.h) the build is exact (448 sweep cases x 5 flavours) and 14 % faster than the four-wave launch: 2.20 -> 1.89 ms at
// 32 768 x 200, hidden 25 .. 32 (profiles/r04/s16n_eight_wave.txt).  ODPD_EXP_S16N_4W restores the old launch shape for comparisons.
#ifdef ODPD_EXP_S16N_4W
constexpr bool s16n_two_waves_per_simd(int mode, bool nw, int nck) { return mode == 1 || (mode == 2 && !nw) || (mode == 0 && !nw && nck <= 2); }
#else
constexpr bool s16n_two_waves_per_simd(int mode, bool nw, int nck) { return mode == 1 || (mode == 2 && !nw) || (mode == 0 && !nw); }
#endif
template <int FM, bool DG, int NT, int MODE, bool NW, bool DX, int NCK>
__global__ __launch_bounds__(s16n_two_waves_per_simd(MODE, NW, NCK) ? 512 : 256, 1) void gru16n_kernel(SeqArgs a) {
    using T = S16N<NT>;
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH, S = kCkptStride;
    constexpr int kGroups = (MODE != 1 && DX) ? T::NG_DX : T::NG;
    constexpr int kWave = (MODE != 1 && DX ? 3 : 2) * 2 * 16 * kChunkPad + ((MODE != 1 && NW) ? T::kTiles * kTileFloats : 0);
    static_assert(NCH <= 2, "operand tables are sized for two feature K-chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    // LDS: [operand table][staged parameters, re-used by the per-wave regions once the table is built]
    float* tab = smem;
    float* pl = tab + s16_tab_floats(kGroups);
    stage_params(pl, a.params, L.P);
    s16n_fill_table<FM, DG, NT>(tab, pl, L, lane, wave, nwb, kGroups);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(kGroups) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + 16 * kChunkPad;                     // target (fused), dy (backward), y (forward)
    float2* dxs = ts + 16 * kChunkPad;                    // backward with DX only
    float* tiles = reinterpret_cast<float*>(ts + ((MODE != 1 && DX) ? 2 : 1) * 16 * kChunkPad);
    if constexpr (MODE != 1 && NW)
        for (int i = lane; i < kTileFloats; i += 64) tiles[6 * NT * kTileFloats + i] = 0.0f;
    S16NGrad<DG, NT> G;
    if constexpr (MODE != 1) G.zero();
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * NT * 64 + lane : nullptr;   // [ckpt][kt][lane]
        if constexpr (MODE != 2) {
            // ---- forward ----
            f32x4 h[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) h[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; t0 < a.T; t0 += kChunk) {
                const int len = min(kChunk, a.T - t0);
                wave_lds_fence();
                stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                wave_lds_fence();
                for (int tt = 0; tt < len; ++tt) {
                    const float2 xv = xs[n * kChunkPad + tt];
                    float fs[NCH];
                    f32x4 r[NT], z[NT], nn[NT], g[NT];
                    s16_slots<FM>(xv.x, xv.y, oh, fs);
                    s16n_cell_fwd<FM, DG, NT, NCK>(opaque(tl), fs, h, r, z, nn, g);
                    if constexpr (MODE == 1) {
                        f32x4 hid[NT], act[NT];
                        float p0, p1;
                        s16n_head<FM, DG, NT, NCK>(opaque(tl), h, fs, hid, act, p0, p1);
                        const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                        if (q == 0) ts[n * kChunkPad + tt] = make_float2(y0, y1);
                    }
                    const int t1 = t0 + tt + 1;
                    if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
#pragma unroll
                        for (int kt = 0; kt < NT; ++kt) ck[((size_t)(t1 / S) * NT + kt) * 64] = make_float4(h[kt][0], h[kt][1], h[kt][2], h[kt][3]);
                    }
                }
                if constexpr (MODE == 1) {
                    wave_lds_fence();
                    stage_out<16>(ts, a.y, b0, a.B, a.T, t0, len, lane);
                }
            }
        }
        if constexpr (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if constexpr (MODE != 1) {
            // ---- backward ----
            f32x4 dh[NT];
            float hTn[NT][4];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                dh[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) hTn[kt][c] = 0.0f;
            }
            int cur_chunk = -1;
            for (int blk = a.nck - 1; blk >= 0; --blk) {
                const int tb = blk * S, nstep = min(S, a.T - tb);
                const int chunk = tb / kChunk, t0 = chunk * kChunk;
                f32x4 h0[NT];
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const float4 v = blk ? ck[((size_t)blk * NT + kt) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
                    h0[kt] = as_f32x4(v);
                }
                if (chunk != cur_chunk) {
                    if constexpr (MODE != 1 && DX) {
                        if (cur_chunk >= 0) {
                            const int pt0 = cur_chunk * kChunk;
                            wave_lds_fence();
                            stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                        }
                    }
                    wave_lds_fence();
                    const int len = min(kChunk, a.T - t0);
                    stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                    if constexpr (MODE == 0)
                        stage_in<16>(ts, a.target, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                    else
                        stage_in<16>(ts, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                    wave_lds_fence();
                    cur_chunk = chunk;
                }
                constexpr bool FUSED = MODE == 0, NWm = NW, DXm = MODE != 1 && DX;
                if (nstep == S)
                    s16n_block<FM, DG, NT, FUSED, NWm, DXm, true, NCK>(a, tl, oh, G, xs, ts, dxs, tiles, n, q, tb - t0, nstep, valid, blk == a.nck - 1, h0, dh, hTn, loss_acc);
                else
                    s16n_block<FM, DG, NT, FUSED, NWm, DXm, false, NCK>(a, tl, oh, G, xs, ts, dxs, tiles, n, q, tb - t0, nstep, valid, blk == a.nck - 1, h0, dh, hTn, loss_acc);
            }
            if constexpr (MODE != 1 && DX) {
                if (cur_chunk >= 0) {
                    const int pt0 = cur_chunk * kChunk;
                    wave_lds_fence();
                    stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    wave_lds_fence();
                }
            }
        }
    }
    if constexpr (MODE == 0 && !NW) {      // frozen PA of a cascade: the loss partial of the workgroup, nothing else
        const float lp = row_sum16(loss_acc);          // accumulated on the q == 0 lanes
        __syncthreads();
        if (lane == 0) smem[wave] = lp;
        __syncthreads();
        if (threadIdx.x < kLossCols) {
            float v = 0.0f;
            if (threadIdx.x == 0)
                for (int wv = 0; wv < nwb; ++wv) v += smem[wv];
            a.partials[(size_t)blockIdx.x * kLossCols + threadIdx.x] = v;
        }
    }
    if constexpr (MODE != 1 && NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        s16n_write_row<FM, DG, NT>(smem + wave * P4, L, G, n, q, loss_acc);
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
static bool s16n_cfg(const odpd_model_t* m, int& FM, bool& DG) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; DG = false; return true;
    case ODPD_DGRU: FM = FEAT_DGRU6; DG = true; return true;
    case ODPD_QGRU: FM = FEAT_Q4; DG = false; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; DG = false; return true;
    default: return false;
    }
}
static int s16n_tiles(int H) { return (H + 15) / 16; }
static LaunchShape s16n_shape(int ngroups) {      // one 4-wave workgroup per CU (one wave per SIMD)
    LaunchShape ls;
    ls.waves = 4;
    const int need = (ngroups + 3) / 4, cap = device_cus();
    ls.grid = need < cap ? need : cap;
    return ls;
}
int gru_s16n_rows(const odpd_model_t* m, int B) {
    (void)m;
    return s16n_shape((B + 15) / 16).grid;
}
int64_t gru_s16n_ckpt_floats(const odpd_model_t* m, int B, int T) {
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * s16n_tiles(m->hidden) * 256;
}

template <int FM, bool DG, int NT, int MODE, bool NW, bool DX, int NCK>
static int launch_s16n(hipStream_t st, const SeqArgs& a, int P) {
    using T = S16N<NT>;
    LaunchShape ls = s16n_shape(a.ngroups);     // the grid (= rows of partials) never depends on the backward's flavour
    if (s16n_two_waves_per_simd(MODE, NW, NCK)) {      // forward / backward of a frozen model
        ls.waves = 8;
        if (MODE != 0) {                        // the fused frozen-PA step writes one loss row per workgroup: its grid is the host's row count
            const int need = (a.ngroups + 7) / 8, cap = device_cus();
            ls.grid = need < cap ? need : cap;
        }
    }
    const int groups = (MODE != 1 && DX) ? T::NG_DX : T::NG;
    const int wave_floats = (MODE != 1 && DX ? 3 : 2) * 2 * 16 * kChunkPad + ((MODE != 1 && NW) ? T::kTiles * kTileFloats : 0);
    auto bytes = [&](int waves) {
        size_t body = (size_t)waves * wave_floats;
        if (body < (size_t)pad4(P)) body = pad4(P);
        size_t nb = ((size_t)s16_tab_floats(groups) + body) * sizeof(float);
        if (MODE != 1 && NW && nb < reduce_scratch_bytes(P, waves)) nb = reduce_scratch_bytes(P, waves);
        return nb;
    };
    size_t lds = bytes(ls.waves);
    if (lds > kMaxLds && ls.waves == 8) {       // eight waves do not fit next to the table: back to one wave per SIMD
        ls = s16n_shape(a.ngroups);
        lds = bytes(ls.waves);
    }
    if (lds > kMaxLds) { ls.waves = 2; lds = bytes(2); }      // weight gradients + dL/dx in one launch: two waves per CU
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = gru16n_kernel<FM, DG, NT, MODE, NW, DX, NCK>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
template <int FM, bool DG, int NT, int NCK>
static int launch_s16n_mode(hipStream_t st, const SeqArgs& a, int P, int mode) {
    if (mode == 0) return launch_s16n<FM, DG, NT, 0, true, false, NCK>(st, a, P);
    if (mode == 3) return launch_s16n<FM, DG, NT, 0, false, true, NCK>(st, a, P);      // forward + loss + dL/dx (frozen PA)
    if (mode == 1) return launch_s16n<FM, DG, NT, 1, false, false, NCK>(st, a, P);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return launch_s16n<FM, DG, NT, 2, true, true, NCK>(st, a, P);
    if (nw) return launch_s16n<FM, DG, NT, 2, true, false, NCK>(st, a, P);
    if (dx) return launch_s16n<FM, DG, NT, 2, false, true, NCK>(st, a, P);
    return ODPD_EINVAL;
}
// hidden 17..24: the last unit tile holds <= 8 units = two K-chunks (instantiated: 2 and 4 chunks)
template <int FM, bool DG, int NT>
static int launch_s16n_chunks(hipStream_t st, const SeqArgs& a, int P, int mode, int H) {
    if (s16n_last_chunks<NT>(H) <= 2) return launch_s16n_mode<FM, DG, NT, 2>(st, a, P, mode);
    return launch_s16n_mode<FM, DG, NT, 4>(st, a, P, mode);
}

// mode 0 fused train, 1 forward, 2 backward, 3 frozen-PA loss step (forward + loss + dL/dx, a.partials = loss rows)
int gru_s16n_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    int FM; bool DG;
    if (!s16n_cfg(m, FM, DG) || s16n_tiles(m->hidden) != 2) return ODPD_EUNSUPPORTED;
    if ((mode == 0 || mode == 3) && !a0.ckpt) return ODPD_EINVAL;
    if (mode == 3 && (!a0.dx || !a0.partials || !a0.target)) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    if (FM == FEAT_RAW2) return launch_s16n_chunks<FEAT_RAW2, false, 2>(st, a, P, mode, m->hidden);
    if (FM == FEAT_DGRU6) return launch_s16n_chunks<FEAT_DGRU6, true, 2>(st, a, P, mode, m->hidden);
    if (FM == FEAT_Q4) return launch_s16n_chunks<FEAT_Q4, false, 2>(st, a, P, mode, m->hidden);
    return launch_s16n_chunks<FEAT_A4, false, 2>(st, a, P, mode, m->hidden);
}

}  // namespace odpd
