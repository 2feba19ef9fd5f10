// gru_s16.hip — "S16" kernels of the nn.GRU based backbones (gru, dgru, qgru, qgru_amp1 float paths; reference
// backbones/{gru,dgru,qgru,qgru_amp1}.py + modules/train_funcs.py:33-39) for hidden <= 16 and batches large enough
// to fill the chip with 16-sequence wavefronts: the fused train kernel (gru16_train_kernel) and the split forward /
// backward kernels (gru16_fwd_kernel, gru16_bwd_kernel: weight-gradient partials and / or dL/dx).
//
// Lane mapping (differs from gru_family.hip): a wavefront holds SIXTEEN sequences; lane l = (n, q) with
// n = l & 15 the sequence and q = l >> 4 a quad of hidden units; the lane owns units 4q..4q+3 of sequence n
// (state h[0..3] in registers).  With this mapping every mat-vec of the recurrence is an exact-fp32
// v_mfma_f32_16x16x4_f32 with NO cross-lane data movement:
//     D[m][n] += sum_k A[m][k] B[k][n],  lane feeds A[l&15][l>>4], B[l>>4][l&15], holds D[4(l>>4)+i][l&15]
//   pre[u][n] = sum_v W[u][v] h_n[v]:  K-chunk c pairs k = q with hidden index v = 4q + c, so the B operand of
//   chunk c is simply the lane's own h[c] and the A operand is the constant W[l&15][4q+c]; D lands as
//   pre[4q+i][n] — exactly the units the lane owns.  16 sequences x (3 gates x 16 x 16) MACs = 12 MFMAs
//   (360 issue cycles) instead of 4 x 45 half-rate v_fmac_dpp (4 x 190) in the row-rotated mapping.
//   The input projection W_ih [I,Q,|x|,..] + b rides in (F+1+3)/4 more K-chunks (feature slot 4c+q; the
//   slot after the last feature is the constant 1, which carries the bias).
// Weight gradients are contractions over (sequence, time): the operands are needed with the sequence index
// on K, i.e. transposed.  Each step the wave bounces d(gates), h and the feature slots through a private
// LDS tile (float4 store, 4 conflict-free b32 loads; no VALU work) and accumulates
//     dW_hh[g] += dg_g^T h_{t-1},  dW_ih|b[g] += dg_g^T [feat,1],  dW_hid += dhid^T h_t
// with 4 MFMAs per 16x16 tile.
// BPTT state: h checkpoints every kCkptStride steps go to an HBM workspace (one coalesced 1 KB store per
// wave; the next block's checkpoint is prefetched a block ahead), blocks are recomputed into registers.
#include "odpd_s16.h"

namespace odpd {

// ---------------------------------------------------------------------------------------------------
// Per-workgroup LDS weight table: entry [group][lane] is a float4 of MFMA operands of lane (m = lane & 15,
// q = lane >> 4).  The forward groups and the backward groups time-share the same registers: a phase pulls
// only its own set (ds_read_b128), like the rotated-quad tables of gru_family.hip.
//   forward : 0-2 W_hg[m][4q+c]   3-4 W_ig|b slots (k = 4c+q)   5 b_hn[4q+i]
//   backward: 6-8 W_hg[4q+c][m]   9 fc_hid[m][4q+c]   10 fc_hid[4q+c][m]   11 b_hid[4q+i]
//             12-13 fc_out[c][4q+i]   14 fc_out feature/bias slots
// ---------------------------------------------------------------------------------------------------
//   dL/dx   : 15-17 W_ig[4q+c][m] (m = feature slot)   18-19 fc_out[cc][H + 4q+i] (feature part, D layout)
constexpr int kS16Groups = 15;        // fused train / weight-gradient-only kernels
constexpr int kS16GroupsDx = 20;      // kernels that also produce dL/dx
template <int FM, bool DG, bool PACK = false>
__device__ __forceinline__ float4 s16_table_entry(const float* pl, const GruLayout& L, int grp, int m, int q) {
    const int H = L.H, OW = DG ? H + 6 : H;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;              // hidden index of K-slot e / own unit e
        const bool mk = m < H && k < H;
        if (grp < 3) {
            v[e] = mk ? pl[L.o_w_hh + (grp * H + m) * H + k] * (grp < 2 ? kNegLog2e : 1.0f) : 0.0f;
            // K-packing (H <= 13): K positions 13..15 carry the chunk-1 input slots 4..6 (odpd_s16.h: s16_slots_pk)
            if (PACK && grp < 2 && k >= 13) v[e] = kNegLog2e * s16_wih_slot<FM, DG>(pl, L, grp, 1, m, k - 13);   // r, z only
        }
        else if (grp == 3) v[e] = kNegLog2e * s16_wih_slot<FM, DG>(pl, L, e >> 1, e & 1, m, q);
        else if (grp == 4) v[e] = e < 2 ? s16_wih_slot<FM, DG>(pl, L, 2, e, m, q) : 0.0f;
        else if (grp == 5) v[e] = k < H ? pl[L.o_b_hh + 2 * H + k] : 0.0f;
        else if (grp < 9) v[e] = mk ? pl[L.o_w_hh + ((grp - 6) * H + k) * H + m] : 0.0f;
        else if (grp == 9) v[e] = (DG && mk) ? pl[L.o_w_hid + m * H + k] : 0.0f;
        else if (grp == 10) v[e] = (DG && mk) ? pl[L.o_w_hid + k * H + m] : 0.0f;
        else if (grp == 11) v[e] = (DG && k < H) ? pl[L.o_b_hid + k] : 0.0f;
        else if (grp < 14) v[e] = k < H ? pl[L.o_w_out + (grp - 12) * OW + k] : 0.0f;
        else if (grp == 14) v[e] = s16_woutf_slot<FM, DG>(pl, L, e >> 1, e & 1, q);
        else if (grp < 18) v[e] = (m < S16Cfg<FM>::F && k < H) ? pl[L.o_w_ih + ((grp - 15) * H + k) * S16Cfg<FM>::F + m] : 0.0f;
        else v[e] = (DG && k < S16Cfg<FM>::F) ? pl[L.o_w_out + (grp - 18) * OW + H + k] : 0.0f;   // k = 4q+e is a slot here
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
template <int FM, bool DG, bool PACK = false>
__device__ __forceinline__ void s16_fill_table(float* tab, const float* pl, const GruLayout& L, int lane, int wave, int nwb,
                                               int ngroups = kS16Groups) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < ngroups; grp += nwb)
        if (DG || (grp < 9 || grp > 11)) t4[grp * 64 + lane] = s16_table_entry<FM, DG, PACK>(pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}

template <int FM>
struct S16Fw {                     // operands of the forward / recompute phase
    float whh[3][4];               // A: W_hg[m][4q+c]
    float wih[3][2];               // A: slot k = 4c+q -> W_ig[m][k] | b_ig(+b_hg) | 0
    f32x4 bhn;                     // C operand of the W_hn h accumulator
};
template <int FM, bool DG>
struct S16Bw {                     // operands of the backward phase
    float whhT[3][4];              // A: W_hg[4q+c][m]
    float whid[4], whidT[4];       // DG: fc_hid[m][4q+c], fc_hid[4q+c][m]
    f32x4 bhid;
    f32x4 wout[2];                 // fc_out weights of the lane's own units
    float woutf[2][2];             // fc_out weight of feature slot k (DG) | fc_out bias at the constant-1 slot
};
template <int FM>
__device__ __forceinline__ void s16_load_fw(S16Fw<FM>& w, TabPtr tl) {
    tl = opaque(tl);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float4 v = tab_ld(tl, g * 64);
        w.whh[g][0] = v.x; w.whh[g][1] = v.y; w.whh[g][2] = v.z; w.whh[g][3] = v.w;
    }
    const float4 a = tab_ld(tl, 3 * 64), b = tab_ld(tl, 4 * 64);
    w.wih[0][0] = a.x; w.wih[0][1] = a.y; w.wih[1][0] = a.z; w.wih[1][1] = a.w; w.wih[2][0] = b.x; w.wih[2][1] = b.y;
    w.bhn = as_f32x4(tab_ld(tl, 5 * 64));
}
template <int FM, bool DG>
__device__ __forceinline__ void s16_load_bw(S16Bw<FM, DG>& w, TabPtr tl) {
    tl = opaque(tl);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const float4 v = tab_ld(tl, (6 + g) * 64);
        w.whhT[g][0] = v.x; w.whhT[g][1] = v.y; w.whhT[g][2] = v.z; w.whhT[g][3] = v.w;
    }
    if constexpr (DG) {
        const float4 a = tab_ld(tl, 9 * 64), b = tab_ld(tl, 10 * 64);
        w.whid[0] = a.x; w.whid[1] = a.y; w.whid[2] = a.z; w.whid[3] = a.w;
        w.whidT[0] = b.x; w.whidT[1] = b.y; w.whidT[2] = b.z; w.whidT[3] = b.w;
        w.bhid = as_f32x4(tab_ld(tl, 11 * 64));
    }
    w.wout[0] = as_f32x4(tab_ld(tl, 12 * 64));
    w.wout[1] = as_f32x4(tab_ld(tl, 13 * 64));
    const float4 f = tab_ld(tl, 14 * 64);
    w.woutf[0][0] = f.x; w.woutf[0][1] = f.y; w.woutf[1][0] = f.z; w.woutf[1][1] = f.w;
}

template <int FM, bool PACK = false>
__device__ __forceinline__ void s16_cell_fwd(const S16Fw<FM>& w, const float (&fs)[S16Cfg<FM>::NCH], f32x4& h, f32x4& r,
                                             f32x4& z, f32x4& n, f32x4& g, const float* pk = nullptr) {
    constexpr int NCH = S16Cfg<FM>::NCH;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 ar = zero, az = zero, an = zero, ah = w.bhn;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (!(PACK && c == 1)) {
            ar = mfma4(w.wih[0][c], fs[c], ar);
            az = mfma4(w.wih[1][c], fs[c], az);
        }
        an = mfma4(w.wih[2][c], fs[c], an);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // PACK: on the quad-3 lanes (own h identically 0 there) elements 1..3 carry the chunk-1 input slots of the r and z
        // gates; the n gate keeps its own chunk (its input part must stay outside r (.) (W_hn h), and one chunk holds only
        // one padded K position)
        const float hb = (PACK && c >= 1) ? h[c] + pk[c - 1] : h[c];
        ar = mfma4(w.whh[0][c], hb, ar);
        az = mfma4(w.whh[1][c], hb, az);
        ah = mfma4(w.whh[2][c], h[c], ah);
    }
    // stage-major element-wise code: every stage is four independent ops (one per owned unit), so a
    // dependent VALU op never issues right behind its producer (8-cycle dependent-issue latency)
    r = sigmoid4_prescaled(ar);
    z = sigmoid4_prescaled(az);
    g = ah;
    n = tanh4_for<FM>(fma4(r, ah, an));
    h = fma4(z, sub4(h, n), n);
}

template <bool DG>
struct S16Grad {
    f32x4 thh[3], tih[3], thid;
    f32x4 db_hn, db_hid, dwout[2];
    float dwf[2][2];
    // K-packed weight gradient (s16_packgrad): dW_i{r,z} of feature slots 0..3 as per-lane sums over the lane's four sequences
    // (transposed domain: lane (u, k) holds unit u, sequences 4k..4k+3); tih[0], tih[1] are then unused
    float vr[4], vz[4];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) { thh[g] = z4; tih[g] = z4; }
        thid = z4; db_hn = z4; db_hid = z4; dwout[0] = z4; dwout[1] = z4;
        dwf[0][0] = dwf[0][1] = dwf[1][0] = dwf[1][1] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) vr[j] = vz[j] = 0.f;
    }
};
// K-packing of the WEIGHT GRADIENT (r04; the forward's K-packing is s16_slots_pk): hidden <= 13 leaves columns 13..15 of the
// W_hh-gradient tiles empty, and the r / z gates' dW_hh and dW_ih tiles share their A operand (the transposed gate gradient).  So
// feature slots 4, 5, 6 (f4, f5 and the constant 1 that carries the bias gradient) ride in columns 13..15 of the h tile — three b32
// LDS stores per step — and come out of the SAME MFMAs as dW_hh; slots 0..3 are 32 plain FMAs in the transposed domain on a
// broadcast float4 of the feature tile.  The two dW_ih tiles of r and z (8 of the 28 weight-gradient MFMAs of a step) disappear; the n
// gate keeps its tile (its dW_hh tile contracts r (.) dnp, not dnp).
#ifdef ODPD_NO_PACKGRAD
constexpr bool s16_packgrad(bool pack, bool nw) { return false; }
#else
constexpr bool s16_packgrad(bool pack, bool nw) { return pack && nw; }
#endif

// One block of <= S steps: recompute the forward pass into registers, then back-propagate.
//   h    : state at the start of the block          dh  : in/out carry dL/dh
//   hTn  : transposed h of the step after the current one (fc_hid weight gradient operand)
//   FUSED: `ts` holds the target, y / loss / dL/dy are formed here;  else `ts` holds dL/dy
//   NW   : accumulate weight gradients into G        DX : write dL/dx of the block's steps to dxs
template <int FM, bool DG, bool FUSED, bool NW, bool DX, bool FULL, bool PACK = false>
__device__ __forceinline__ void s16_block(const SeqArgs& a, TabPtr tl, const float (&oh)[4], S16Grad<DG>& G,
                                          const float2* xs, const float2* ts, float2* dxs, float* tiles, int n, int q, int tloc,
                                          int nstep, bool valid, bool last_blk, f32x4 h, f32x4& dh, float (&hTn)[4],
                                          float& loss_acc) {
    constexpr int NCH = S16Cfg<FM>::NCH, S = kCkptStride;
    f32x4 hp_s[S], r_s[S], z_s[S], n_s[S], g_s[S];
    float fs_s[S][NCH];
    {
        S16Fw<FM> wf;
        s16_load_fw<FM>(wf, tl);
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (FULL || st < nstep) {
                const float2 xv = xs[n * kChunkPad + tloc + st];
                hp_s[st] = h;
                if constexpr (PACK) {
                    float pk[3];
                    s16_slots_pk<FM>(xv.x, xv.y, oh, fs_s[st], pk);
                    s16_cell_fwd<FM, true>(wf, fs_s[st], h, r_s[st], z_s[st], n_s[st], g_s[st], pk);
                } else {
                    s16_slots<FM>(xv.x, xv.y, oh, fs_s[st]);
                    s16_cell_fwd<FM>(wf, fs_s[st], h, r_s[st], z_s[st], n_s[st], g_s[st]);
                }
            }
        }
    }
    S16Bw<FM, DG> w;
    s16_load_bw<FM, DG>(w, tl);
    float* t_r = tiles, *t_z = tiles + kTileFloats, *t_n = tiles + 2 * kTileFloats, *t_g = tiles + 3 * kTileFloats;
    float* t_h = tiles + 4 * kTileFloats, *t_d = tiles + 5 * kTileFloats, *t_f = tiles + 6 * kTileFloats;
    float wihT[3][4];
    f32x4 wfD[2];
    if constexpr (DX) {
        TabPtr tx = opaque(tl);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float4 v = tab_ld(tx, (15 + g) * 64);
            wihT[g][0] = v.x; wihT[g][1] = v.y; wihT[g][2] = v.z; wihT[g][3] = v.w;
        }
        wfD[0] = as_f32x4(tab_ld(tx, 18 * 64));
        wfD[1] = as_f32x4(tab_ld(tx, 19 * 64));
    }
    if (NW && DG && last_blk) {   // transposed final state of the frame: operand of the last step's dW_hid
        wave_lds_fence();
        tile_put(t_h, n, q, h);
        wave_lds_fence();
        tile_get(t_h, n, q, hTn);
    }
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, valid ? a.inv_count : 0.0f, valid && q == 0);
#pragma unroll
    for (int st = S - 1; st >= 0; --st) {
        if (FULL || st < nstep) {
            __builtin_amdgcn_sched_barrier(0);      // keep every backward step's work together in the unrolled block (r04: -0.8 %, same bits)
            const f32x4 hp = hp_s[st], r = r_s[st], z = z_s[st], nn = n_s[st], gh = g_s[st];
            f32x4 act, hid;
            const f32x4 ht = fma4(z, sub4(hp, nn), nn);
            if constexpr (DG) {
                hid = w.bhid;
#pragma unroll
                for (int c = 0; c < 4; ++c) hid = mfma4(w.whid[c], ht[c], hid);
                ODPD_EACH4 act[i] = relu_(hid[i]);
            } else {
                act = ht;
            }
            const float2 tv = ts[n * kChunkPad + tloc + st];
            float dy0 = tv.x, dy1 = tv.y;
            if constexpr (FUSED) {
                // y = fc_out(cat(act, feat)) (+ bias through the constant-1 slot), loss and dL/dy
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    p0 = __builtin_fmaf(w.woutf[0][c], fs_s[st][c], p0);
                    p1 = __builtin_fmaf(w.woutf[1][c], fs_s[st][c], p1);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0 = __builtin_fmaf(w.wout[0][e], act[e], p0);
                    p1 = __builtin_fmaf(w.wout[1][e], act[e], p1);
                }
                const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                const float d0 = y0 - tv.x, d1 = y1 - tv.y;
                s16_loss(lossc, d0, d1, dy0, dy1, loss_acc);
            }
            if constexpr (NW) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c < 2) {
                        G.dwf[0][c] = __builtin_fmaf(dy0, fs_s[st][c], G.dwf[0][c]);
                        G.dwf[1][c] = __builtin_fmaf(dy1, fs_s[st][c], G.dwf[1][c]);
                    }
                }
                G.dwout[0] = fma4(splat4(dy0), act, G.dwout[0]);
                G.dwout[1] = fma4(splat4(dy1), act, G.dwout[1]);
            }
            f32x4 dht, dhid;
            const f32x4 dact = fma4(splat4(dy0), w.wout[0], mul4(w.wout[1], splat4(dy1)));
            if constexpr (DG) {
                ODPD_EACH4 dhid[i] = dact[i] * relu_gate(hid[i]);
                if constexpr (NW) G.db_hid = add4(G.db_hid, dhid);
                // (r04 timing experiments, profiles/r04/headline_experiments.md: opening the chain at 0 and adding the carried dL/dh
                // afterwards — one add instead of four MFMAs on the step-to-step chain — changes nothing: +0.3 %, noise)
                dht = dh;
#pragma unroll
                for (int c = 0; c < 4; ++c) dht = mfma4(w.whidT[c], dhid[c], dht);
            } else {
                dht = add4(dh, dact);
            }
            // cell backward, stage-major (see s16_cell_fwd)
            const f32x4 one = splat4(1.0f);
            f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
            // dn = dht (1-z); dnp = dn (1-n^2); dgh = dnp r; drp = dgh ghn (1-r); dzp = (hp-n) z dn
            const f32x4 omz = sub4(one, z), omr = sub4(one, r);
            f32x4 omn2;
            ODPD_EACH4 omn2[i] = __builtin_fmaf(-nn[i], nn[i], 1.0f);
            const f32x4 dn = mul4(dht, omz);
            f32x4 acc0 = mul4(dht, z);
            const f32x4 dnp = mul4(dn, omn2);
            const f32x4 dgh = mul4(dnp, r);
            const f32x4 dzp = mul4(mul4(sub4(hp, nn), z), dn);
            const f32x4 drp = mul4(mul4(dgh, gh), omr);
            if constexpr (NW) G.db_hn = add4(G.db_hn, dgh);
            float rT[4], zT[4], nT[4], gT[4], hT[4], dT[4], fT[4];
            constexpr bool PG = s16_packgrad(PACK, NW);
            // the step's transposes through LDS (sequence index onto K): tile stores, then the loads of the transposed operands
            auto transposes = [&]() {
                wave_lds_fence();
                tile_put(t_r, n, q, drp);
                tile_put(t_z, n, q, dzp);
                tile_put(t_n, n, q, dnp);
                tile_put(t_g, n, q, dgh);
                tile_put(t_h, n, q, hp);
                if constexpr (PG) {      // columns 13..15 of the h tile (units 13..15: identically 0) <- feature slots 4, 5, 6 of the sequence
                    t_h[n * kTilePitch + 13 + q] = fs_s[st][1];      // (quad 3 holds slot 7 = 0 and lands in the row's pad column 16: no exec masking)
                }
                if constexpr (DG) tile_put(t_d, n, q, dhid);
#pragma unroll
                for (int c = 0; c < NCH; ++c) t_f[n * kTilePitch + 4 * c + q] = fs_s[st][c];
                wave_lds_fence();
                tile_get(t_r, n, q, rT);
                tile_get(t_z, n, q, zT);
                tile_get(t_n, n, q, nT);
                tile_get(t_g, n, q, gT);
                tile_get(t_h, n, q, hT);
                tile_get(t_f, n, q, fT);
                if constexpr (DG) tile_get(t_d, n, q, dT);
            };
            // (issued HERE, in front of the 12 W_hh^T MFMAs that could cover the LDS round trip, and pinned with a sched_barrier, the step
            // got 1 % SLOWER: 28 more live registers across those MFMAs, scratch 212 -> 252 B per lane — profiles/r04/headline_experiments.md)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc0 = mfma4(w.whhT[0][c], drp[c], acc0);
                acc1 = mfma4(w.whhT[1][c], dzp[c], acc1);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c & 1) acc1 = mfma4(w.whhT[2][c], dgh[c], acc1);
                else acc0 = mfma4(w.whhT[2][c], dgh[c], acc0);
            }
            dh = add4(acc0, acc1);
            if constexpr (DX) {
                // dL/d(feature slot j) of sequence n = sum_g sum_u W_ig[u][j] dpre_g[u] (+ fc_out feature columns): the
                // same in-place MFMA pattern with the feature slots on M; D lands as slots 4q..4q+3 on lane (n,q)
                f32x4 ds0 = {0.f, 0.f, 0.f, 0.f}, ds1 = ds0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ds0 = mfma4(wihT[0][c], drp[c], ds0);
                    ds1 = mfma4(wihT[1][c], dzp[c], ds1);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c & 1) ds1 = mfma4(wihT[2][c], dnp[c], ds1);
                    else ds0 = mfma4(wihT[2][c], dnp[c], ds0);
                }
                f32x4 ds = add4(ds0, ds1);
                if constexpr (DG) ds = fma4(splat4(dy0), wfD[0], fma4(splat4(dy1), wfD[1], ds));
                constexpr int F = S16Cfg<FM>::F;
                float df[F];
#pragma unroll
                for (int j = 0; j < F; ++j) df[j] = j < 4 ? ds[j & 3] : swap16(ds[j & 3]);   // slots 4..7 live on quad 1
                const float2 xv = xs[n * kChunkPad + tloc + st];
                float dI, dQ;
                feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
                if (q == 0) dxs[n * kChunkPad + tloc + st] = make_float2(dI, dQ);
            }
            if constexpr (NW) {
            // weight gradients: transpose through LDS (sequence index onto K), then rank-16 MFMA updates
            transposes();
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                G.thh[0] = mfma4(rT[c], hT[c], G.thh[0]);
                G.thh[1] = mfma4(zT[c], hT[c], G.thh[1]);
                G.thh[2] = mfma4(gT[c], hT[c], G.thh[2]);
                if constexpr (!PG) {
                    G.tih[0] = mfma4(rT[c], fT[c], G.tih[0]);
                    G.tih[1] = mfma4(zT[c], fT[c], G.tih[1]);
                }
                G.tih[2] = mfma4(nT[c], fT[c], G.tih[2]);
                if constexpr (DG) G.thid = mfma4(dT[c], hTn[c], G.thid);
            }
            if constexpr (PG) {      // slots 0..3 of sequence 4q + c: one broadcast float4 per c (all 16 unit lanes read the same address)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 f = *reinterpret_cast<const float4*>(t_f + (4 * q + c) * kTilePitch);
                    G.vr[0] = __builtin_fmaf(rT[c], f.x, G.vr[0]); G.vr[1] = __builtin_fmaf(rT[c], f.y, G.vr[1]);
                    G.vr[2] = __builtin_fmaf(rT[c], f.z, G.vr[2]); G.vr[3] = __builtin_fmaf(rT[c], f.w, G.vr[3]);
                    G.vz[0] = __builtin_fmaf(zT[c], f.x, G.vz[0]); G.vz[1] = __builtin_fmaf(zT[c], f.y, G.vz[1]);
                    G.vz[2] = __builtin_fmaf(zT[c], f.z, G.vz[2]); G.vz[3] = __builtin_fmaf(zT[c], f.w, G.vz[3]);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) hTn[c] = hT[c];
            }
        }
    }
}

// the wave's row of partial gradients (every entry written), layout = flattened parameter order + 4 loss columns
template <int FM, bool DG, bool PG = false>
__device__ __forceinline__ void s16_write_row(float* prow, const GruLayout& L, S16Grad<DG>& G, int n, int q, float loss_acc) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH;
    const int H = L.H, OW = DG ? H + 6 : H;
    // tiles: lane (c = n, g4 = q) holds rows 4q+rr, column c
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * q + rr;
            if (i < H) {
                if (PG && g < 2) {       // K-packed: slots 4, 5, 6 of the r / z gates sit in columns 13, 14, 15 of their dW_hh tiles
                    const float v = G.thh[g][rr];
                    if (n >= 13 && n - 9 < F) prow[L.o_w_ih + (g * H + i) * F + n - 9] = v;
                    else if (n - 9 == F) { prow[L.o_b_ih + g * H + i] = v; prow[L.o_b_hh + g * H + i] = v; }
                } else {
                    const float v = G.tih[g][rr];
                    if (n < F) prow[L.o_w_ih + (g * H + i) * F + n] = v;
                    else if (n == F) {
                        prow[L.o_b_ih + g * H + i] = v;
                        if (g < 2) prow[L.o_b_hh + g * H + i] = v;
                    }
                }
                if (n < H) prow[L.o_w_hh + (g * H + i) * H + n] = G.thh[g][rr];
            }
        }
    if constexpr (PG) {      // slots 0..3: lane (u = n, k = q) summed its sequences 4k..4k+3; the four k of a unit meet here
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = quad_sum(G.vr[j]), z = quad_sum(G.vz[j]);
            if (q == 0 && n < H) { prow[L.o_w_ih + (0 * H + n) * F + j] = r; prow[L.o_w_ih + (1 * H + n) * F + j] = z; }
        }
    }
    if constexpr (DG) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * q + rr;
            if (i < H && n < H) prow[L.o_w_hid + i * H + n] = G.thid[rr];
        }
    }
    // per-unit scalars: sum over the 16 sequences of the DPP row
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int u = 4 * q + e;
        const float bhn = row_sum16(G.db_hn[e]), bhid = row_sum16(G.db_hid[e]);
        const float w0 = row_sum16(G.dwout[0][e]), w1 = row_sum16(G.dwout[1][e]);
        if (n == 0 && u < H) {
            prow[L.o_b_hh + 2 * H + u] = bhn;
            prow[L.o_w_out + u] = w0;
            prow[L.o_w_out + OW + u] = w1;
            if constexpr (DG) prow[L.o_b_hid + u] = bhid;
        }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int k = 4 * c + q;
            const float v = row_sum16(G.dwf[cc][c < 2 ? c : 0]);
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

// NW: weight-gradient partials (train_pa / the trained model).  !NW && DX: the frozen PA of a cascade in one launch — forward,
// loss and dL/dx (written to a.dx), one loss partial per workgroup in a.partials[blockIdx.x * kLossCols].
// (the body takes its workgroup index and count as arguments: the sweep launch below runs it for K models side by side)
template <int FM, bool DG, int OCC, bool PACK, bool NW = true, bool DX = false>
__device__ __forceinline__ void gru16_train_body(const SeqArgs& a, const int bid, const int nbl) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH, S = kCkptStride;
    constexpr int kGroups = DX ? kS16GroupsDx : kS16Groups;
    constexpr int kWave = (DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? kS16Tiles * kTileFloats : 0);
    static_assert(NCH <= 2, "operand tables and dwf accumulators are sized for two K-chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    s16_fill_table<FM, DG, PACK>(tab, pl, L, lane, wave, nwb, kGroups);
    TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(kGroups) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + 16 * kChunkPad;
    float2* dxs = DX ? ts + 16 * kChunkPad : nullptr;
    float* tiles = reinterpret_cast<float*>(ts + (DX ? 2 : 1) * 16 * kChunkPad);
    if constexpr (NW)
        for (int i = lane; i < kTileFloats; i += 64) tiles[6 * kTileFloats + i] = 0.0f;   // unused feature columns stay 0
    S16Grad<DG> G;
    G.zero();
    float loss_acc = 0.0f;
    const int nwaves = nbl * nwb;
    for (int grp = bid * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        float4* ck = reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * 64 + lane;
        // ---- forward: cell only, checkpoint every S steps ----
        {
            S16Fw<FM> w;
            s16_load_fw<FM>(w, tl);
            f32x4 h = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = 0; t0 < a.T; t0 += kChunk) {
                const int len = min(kChunk, a.T - t0);
                wave_lds_fence();
                stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                wave_lds_fence();
                int tt = 0;
                for (; tt + S <= len; tt += S) {
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const float2 xv = xs[n * kChunkPad + tt + i];
                        float fs[NCH];
                        f32x4 r, z, nn, g;
                        if constexpr (PACK) {
                            float pk[3];
                            s16_slots_pk<FM>(xv.x, xv.y, oh, fs, pk);
                            s16_cell_fwd<FM, true>(w, fs, h, r, z, nn, g, pk);
                        } else {
                            s16_slots<FM>(xv.x, xv.y, oh, fs);
                            s16_cell_fwd<FM>(w, fs, h, r, z, nn, g);
                        }
                    }
                    const int t1 = t0 + tt + S;
                    if (t1 < a.T) ck[(size_t)(t1 / S) * 64] = make_float4(h[0], h[1], h[2], h[3]);
                }
                for (; tt < len; ++tt) {
                    const float2 xv = xs[n * kChunkPad + tt];
                    float fs[NCH];
                    f32x4 r, z, nn, g;
                    if constexpr (PACK) {
                        float pk[3];
                        s16_slots_pk<FM>(xv.x, xv.y, oh, fs, pk);
                        s16_cell_fwd<FM, true>(w, fs, h, r, z, nn, g, pk);
                    } else {
                        s16_slots<FM>(xv.x, xv.y, oh, fs);
                        s16_cell_fwd<FM>(w, fs, h, r, z, nn, g);
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // own checkpoint stores precede the loads below
        // ---- backward ----
        f32x4 dh = {0.f, 0.f, 0.f, 0.f};
        float hTn[4] = {0.f, 0.f, 0.f, 0.f};
        int cur_chunk = -1;
        float4 h0n = a.nck > 1 ? ck[(size_t)(a.nck - 1) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            const f32x4 h0 = {h0n.x, h0n.y, h0n.z, h0n.w};
            h0n = blk > 1 ? ck[(size_t)(blk - 1) * 64] : make_float4(0.f, 0.f, 0.f, 0.f);   // prefetch a block ahead
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
                stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                stage_in<16>(ts, a.target, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                wave_lds_fence();
                cur_chunk = chunk;
            }
            if (nstep == S)
                s16_block<FM, DG, true, NW, DX, true, PACK>(a, tl, oh, G, xs, ts, dxs, tiles, n, q, tb - t0, nstep, valid, blk == a.nck - 1, h0, dh, hTn, loss_acc);
            else
                s16_block<FM, DG, true, NW, DX, false, PACK>(a, tl, oh, G, xs, ts, dxs, tiles, n, q, tb - t0, nstep, valid, blk == a.nck - 1, h0, dh, hTn, loss_acc);
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
    if constexpr (!NW) {      // loss partial of the workgroup (summed over its waves in order)
        const float lp = row_sum16(loss_acc);          // accumulated on the q == 0 lanes
        __syncthreads();
        if (lane == 0) smem[wave] = lp;
        __syncthreads();
        if (threadIdx.x < kLossCols) {
            float v = 0.0f;
            if (threadIdx.x == 0)
                for (int wv = 0; wv < nwb; ++wv) v += smem[wv];
            a.partials[(size_t)bid * kLossCols + threadIdx.x] = v;
        }
        return;
    }
    // ---- one row of partial gradients per workgroup (fixed summation order) ----
    const int P4 = L.P + kLossCols;
    __syncthreads();
    s16_write_row<FM, DG, s16_packgrad(PACK, NW)>(smem + wave * P4, L, G, n, q, loss_acc);
    __syncthreads();
    float* prow = a.partials + (size_t)bid * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = smem[i];
        for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}
template <int FM, bool DG, int OCC, bool PACK, bool NW = true, bool DX = false>
__global__ __launch_bounds__(64 * 4 * OCC, OCC) void gru16_train_kernel(SeqArgs a) {
    gru16_train_body<FM, DG, OCC, PACK, NW, DX>(a, blockIdx.x, gridDim.x);
}
// K independent runs of one model shape in lockstep on the 16-sequences-per-wave kernel (odpd_train_epoch_sweep, ODPD_SWEEP_S16): run k owns
// workgroups [k G, (k + 1) G) and sees the launch it would have alone with this kernel forced (odpd_set_tuning("s16_min_batch", 0))
template <int FM, bool DG, int OCC, bool PACK>
__global__ __launch_bounds__(64 * 4 * OCC, OCC) void gru16_train_sweep_kernel(SeqArgs a, const SweepRun* __restrict__ runs, int G, long long first) {
    const SweepRun r = runs[blockIdx.x / G];
    a.params = r.params; a.partials = r.partials; a.frame_idx = r.order + first; a.ckpt = r.workspace;
    gru16_train_body<FM, DG, OCC, PACK>(a, blockIdx.x % G, G);
}

// -------------------------------------------------------------------------------------------------
// split kernels (odpd_backbone_fwd / odpd_backbone_bwd at large batch: autograd path, cascades, inference)
// -------------------------------------------------------------------------------------------------
// BPTT checkpoints of the split kernels, [task][ckpt] records of 64 float4 (r06: only REAL units travel).  Quads whose four unit slots are all real
// store their float4 at the usual place; the quad holding the last H % 4 units stores that many dwords, 16 lanes contiguous per element, inside its
// own float4 region; quads of padding store nothing (their h is 0 for ever: zero weights, zero biases).  Hidden 13: 832 of 1 024 bytes per
// checkpoint and 16 sequences, written once and read once.  (The fused train kernel keeps whole records: its register allocation is not touched.)
__device__ __forceinline__ void s16_ckpt_store(float4* rec, int lane, int n, int q, int H, const f32x4& h) {
    const int nfull = H >> 2, rem = H & 3;
    if (q < nfull) rec[lane] = make_float4(h[0], h[1], h[2], h[3]);
    else if (q == nfull) {
        float* f = reinterpret_cast<float*>(rec + nfull * 16) + n;
        if (rem > 0) f[0] = h[0];
        if (rem > 1) f[16] = h[1];
        if (rem > 2) f[32] = h[2];
    }
}
__device__ __forceinline__ float4 s16_ckpt_load(const float4* rec, int lane, int n, int q, int H) {
    const int nfull = H >> 2, rem = H & 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < nfull) v = rec[lane];
    else if (q == nfull) {
        const float* f = reinterpret_cast<const float*>(rec + nfull * 16) + n;
        if (rem > 0) v.x = f[0];
        if (rem > 1) v.y = f[16];
        if (rem > 2) v.z = f[32];
    }
    return v;
}

// forward: y for every step, optional checkpoints of h ([task][ckpt] records, s16_ckpt_store)
template <int FM, bool DG, bool PACK>
__global__ __launch_bounds__(1024) void gru16_fwd_kernel(SeqArgs a) {
    constexpr int F = S16Cfg<FM>::F, NCH = S16Cfg<FM>::NCH, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    s16_fill_table<FM, DG, PACK>(tab, pl, L, lane, wave, nwb);
    TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float2* xs = reinterpret_cast<float2*>(tab + s16_tab_floats(kS16Groups)) + (size_t)wave * (2 * 16 * kChunkPad);
    float2* ys = xs + 16 * kChunkPad;
    S16Fw<FM> wf;
    s16_load_fw<FM>(wf, tl);
    S16Bw<FM, DG> wh;            // only the head operands (fc_hid, fc_out) are used here
    s16_load_bw<FM, DG>(wh, tl);
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * 64 : nullptr;
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xs[n * kChunkPad + tt];
                float fs[NCH];
                f32x4 r, z, nn, g, act;
                if constexpr (PACK) {
                    float pk[3];
                    s16_slots_pk<FM>(xv.x, xv.y, oh, fs, pk);
                    s16_cell_fwd<FM, true>(wf, fs, h, r, z, nn, g, pk);
                } else {
                    s16_slots<FM>(xv.x, xv.y, oh, fs);
                    s16_cell_fwd<FM>(wf, fs, h, r, z, nn, g);
                }
                if constexpr (DG) {
                    f32x4 hid = wh.bhid;
#pragma unroll
                    for (int c = 0; c < 4; ++c) hid = mfma4(wh.whid[c], h[c], hid);
                    ODPD_EACH4 act[i] = relu_(hid[i]);
                } else {
                    act = h;
                }
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    p0 = __builtin_fmaf(wh.woutf[0][c], fs[c], p0);
                    p1 = __builtin_fmaf(wh.woutf[1][c], fs[c], p1);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0 = __builtin_fmaf(wh.wout[0][e], act[e], p0);
                    p1 = __builtin_fmaf(wh.wout[1][e], act[e], p1);
                }
                const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) s16_ckpt_store(ck + (size_t)(t1 / S) * 64, lane, n, q, a.H, h);
            }
            wave_lds_fence();
            stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
    }
}

// backward from dL/dy: weight-gradient partials (NW) and / or dL/dx (DX)
template <int FM, bool DG, bool NW, bool DX, int WAVES, bool PACK = false>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void gru16_bwd_kernel(SeqArgs a) {
    constexpr int F = S16Cfg<FM>::F, S = kCkptStride;
    constexpr int kGroups = DX ? kS16GroupsDx : kS16Groups;
    constexpr int kWave = (DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? kS16Tiles * kTileFloats : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    s16_fill_table<FM, DG, PACK>(tab, pl, L, lane, wave, nwb, kGroups);
    TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(kGroups) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * kChunkPad;
    float2* dxs = DX ? dys + 16 * kChunkPad : nullptr;
    float* tiles = reinterpret_cast<float*>(dys + (DX ? 2 : 1) * 16 * kChunkPad);
    if constexpr (NW)
        for (int i = lane; i < kTileFloats; i += 64) tiles[6 * kTileFloats + i] = 0.0f;
    S16Grad<DG> G;
    G.zero();
    float unused = 0.0f;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * 64;
        f32x4 dh = {0.f, 0.f, 0.f, 0.f};
        float hTn[4] = {0.f, 0.f, 0.f, 0.f};
        int cur_chunk = -1;
        float4 h0n = a.nck > 1 ? s16_ckpt_load(ck + (size_t)(a.nck - 1) * 64, lane, n, q, a.H) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            const f32x4 h0 = {h0n.x, h0n.y, h0n.z, h0n.w};
            h0n = blk > 1 ? s16_ckpt_load(ck + (size_t)(blk - 1) * 64, lane, n, q, a.H) : make_float4(0.f, 0.f, 0.f, 0.f);
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
            if (nstep == S)
                s16_block<FM, DG, false, NW, DX, true, PACK>(a, tl, oh, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, true, blk == a.nck - 1, h0, dh, hTn, unused);
            else
                s16_block<FM, DG, false, NW, DX, false, PACK>(a, tl, oh, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, true, blk == a.nck - 1, h0, dh, hTn, unused);
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
        s16_write_row<FM, DG, s16_packgrad(PACK, NW)>(smem + wave * P4, L, G, n, q, 0.0f);
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
static bool s16_cfg(const odpd_model_t* m, int& FM, bool& DG) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; DG = false; return true;
    case ODPD_DGRU: FM = FEAT_DGRU6; DG = true; return true;
    case ODPD_QGRU: FM = FEAT_Q4; DG = false; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; DG = false; return true;
    default: return false;
    }
}
static int s16_param_count(int H, int FM, bool DG) {
    return gru_layout(H, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
}
static size_t s16_lds_bytes(int P, int waves) {
    size_t nbytes = ((size_t)pad4(P) + s16_tab_floats(kS16Groups) + (size_t)waves * kS16WaveFloats) * sizeof(float);
    const size_t red = reduce_scratch_bytes(P, waves);
    return nbytes > red ? nbytes : red;
}
// waves per SIMD: forced by the tuning knob, else 1 while one wave per SIMD covers the batch (a lone wave runs
// its task ~1.7x faster than two sharing a SIMD), 2 beyond that (+20 % throughput)
static int s16_occupancy(int ngroups) {
    const int forced = tuning().s16_occupancy;
    if (forced == 1 || forced == 2) return forced;
    return ngroups <= 4 * device_cus() ? 1 : 2;
}
// one workgroup per CU: 4 waves (1 per SIMD) or 8 waves (2 per SIMD) share one operand table
static LaunchShape s16_shape(int ngroups) {
    LaunchShape ls;
    ls.waves = 4 * s16_occupancy(ngroups);
    const int need = (ngroups + ls.waves - 1) / ls.waves, cap = device_cus();
    ls.grid = need < cap ? need : cap;
    return ls;
}

int gru_s16_groups(int B) { return (B + 15) / 16; }
int64_t gru_s16_workspace_floats(const odpd_model_t* m, int B, int T) {
    (void)m;
    return (int64_t)gru_s16_groups(B) * num_ckpt(T) * 256;
}
int gru_s16_rows(const odpd_model_t* m, int B) {
    (void)m;
    return s16_shape(gru_s16_groups(B)).grid;
}

template <int FM, bool DG, int OCC, bool PACK>
static int launch_s16(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = s16_shape(a.ngroups);
    const size_t lds = s16_lds_bytes(P, ls.waves);
    auto k = gru16_train_kernel<FM, DG, OCC, PACK>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
// frozen PA of a cascade: forward + loss + dL/du in one launch (a.x = u, a.target, a.dx = du, a.partials = loss rows)
template <int FM, bool DG, int OCC, bool PACK>
static int launch_s16_lossdx(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = s16_shape(a.ngroups);
    const size_t lds = ((size_t)pad4(P) + s16_tab_floats(kS16GroupsDx) + (size_t)ls.waves * 3 * 2 * 16 * kChunkPad) * sizeof(float);
    auto k = gru16_train_kernel<FM, DG, OCC, PACK, false, true>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
template <int FM, bool DG>
static int launch_s16_lossdx_occ(hipStream_t st, const SeqArgs& a, int P) {
    constexpr bool kCanPack = S16Cfg<FM>::NCH == 2;
    if (s16_occupancy(a.ngroups) == 1) return launch_s16_lossdx<FM, DG, 1, false>(st, a, P);
    if constexpr (kCanPack) { if (a.H <= 13) return launch_s16_lossdx<FM, DG, 2, true>(st, a, P); }
    return launch_s16_lossdx<FM, DG, 2, false>(st, a, P);
}
template <int FM, bool DG>
static int launch_s16_occ(hipStream_t st, const SeqArgs& a, int P) {
    // K-packing: two input chunks and hidden <= 13 (three padded K positions free for the r / z chunk-1 slots)
    constexpr bool kCanPack = S16Cfg<FM>::NCH == 2;
    const bool pack = kCanPack && a.H <= 13;
    if (s16_occupancy(a.ngroups) == 1) return launch_s16<FM, DG, 1, false>(st, a, P);
    if constexpr (kCanPack) { if (pack) return launch_s16<FM, DG, 2, true>(st, a, P); }
    return launch_s16<FM, DG, 2, false>(st, a, P);
}

// ---- split kernels: launch shapes -----------------------------------------------------------------
// one workgroup per CU; as many waves per workgroup (4, 8, 16 = 1, 2, 4 per SIMD) as the batch can feed
static LaunchShape s16_fwd_shape(int ngroups) {
    LaunchShape ls;
    const int cus = device_cus();
    ls.waves = 4;
    while (ls.waves < 16 && ngroups > ls.waves * cus) ls.waves *= 2;
    const int need = (ngroups + ls.waves - 1) / ls.waves;
    ls.grid = need < cus ? need : cus;
    return ls;
}
static LaunchShape s16_bwd_shape(int ngroups, bool nw, bool dx) {
    LaunchShape ls;
    const int cus = device_cus();
    const int maxw = (nw && dx) ? 4 : 8;    // both outputs: the per-wave LDS only fits one wave per SIMD
    ls.waves = ngroups <= 4 * cus ? 4 : maxw;
    const int need = (ngroups + ls.waves - 1) / ls.waves;
    ls.grid = need < cus ? need : cus;
    return ls;
}
// rows of partials of the split backward: the grid of the weight-gradient-only shape, also used when dL/dx is
// produced in the same launch (so the row count does not depend on whether dx was requested)
int gru_s16_bwd_rows(const odpd_model_t* m, int B) {
    (void)m;
    return s16_bwd_shape(gru_s16_groups(B), true, false).grid;
}

template <int FM, bool DG>
static int launch_s16_fwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = s16_fwd_shape(a.ngroups);
    const size_t lds = ((size_t)pad4(P) + s16_tab_floats(kS16Groups) + (size_t)ls.waves * 2 * 2 * 16 * kChunkPad) * sizeof(float);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    };
    if constexpr (S16Cfg<FM>::NCH == 2) { if (a.H <= 13) return launch(gru16_fwd_kernel<FM, DG, true>); }   // K-packing
    return launch(gru16_fwd_kernel<FM, DG, false>);
}
template <int FM, bool DG, bool NW, bool DX, int WAVES>
static int launch_s16_bwd(hipStream_t st, const SeqArgs& a, int P) {
    // the number of partial rows must not depend on DX: a NW+DX launch keeps the grid of the NW-only shape
    LaunchShape ls = s16_bwd_shape(a.ngroups, NW, DX);
    if (NW) ls.grid = s16_bwd_shape(a.ngroups, true, false).grid;
    const int waves = ls.waves < WAVES ? ls.waves : WAVES;
    const int wave_floats = (DX ? 3 : 2) * 2 * 16 * kChunkPad + (NW ? kS16Tiles * kTileFloats : 0);
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(DX ? kS16GroupsDx : kS16Groups) + (size_t)waves * wave_floats) * sizeof(float);
    if (NW && lds < reduce_scratch_bytes(P, waves)) lds = reduce_scratch_bytes(P, waves);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * waves), lds, st, a);
        return (int)hipGetLastError();
    };
    // K-packing (as the fused train kernel): the weight-gradient-only backward of a trained DPD with <= 13 units
    if constexpr (S16Cfg<FM>::NCH == 2 && NW && !DX) { if (a.H <= 13) return launch(gru16_bwd_kernel<FM, DG, NW, DX, WAVES, true>); }
    return launch(gru16_bwd_kernel<FM, DG, NW, DX, WAVES>);
}
template <int FM, bool DG>
static int launch_s16_bwd_mode(hipStream_t st, const SeqArgs& a, int P) {
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return launch_s16_bwd<FM, DG, true, true, 4>(st, a, P);
    if (nw) return launch_s16_bwd<FM, DG, true, false, 8>(st, a, P);
    if (dx) return launch_s16_bwd<FM, DG, false, true, 8>(st, a, P);
    return ODPD_EINVAL;
}
#define ODPD_S16_DISPATCH(FN, ...)                                                   \
    if (FM == FEAT_RAW2) return FN<FEAT_RAW2, false>(__VA_ARGS__);                   \
    if (FM == FEAT_DGRU6) return FN<FEAT_DGRU6, true>(__VA_ARGS__);                  \
    if (FM == FEAT_Q4) return FN<FEAT_Q4, false>(__VA_ARGS__);                       \
    return FN<FEAT_A4, false>(__VA_ARGS__);

int gru_s16_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0) {
    int FM; bool DG;
    if (!s16_cfg(m, FM, DG) || m->hidden > 16) return ODPD_EUNSUPPORTED;
    SeqArgs a = a0;
    a.ngroups = gru_s16_groups(a.B);
    const int P = s16_param_count(m->hidden, FM, DG);
    ODPD_S16_DISPATCH(launch_s16_fwd, st, a, P)
}
int gru_s16_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0) {
    int FM; bool DG;
    if (!s16_cfg(m, FM, DG) || m->hidden > 16) return ODPD_EUNSUPPORTED;
    SeqArgs a = a0;
    a.ngroups = gru_s16_groups(a.B);
    const int P = s16_param_count(m->hidden, FM, DG);
    ODPD_S16_DISPATCH(launch_s16_bwd_mode, st, a, P)
}

int gru_s16_lossdx(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0) {
    int FM; bool DG;
    if (!s16_cfg(m, FM, DG) || m->hidden > 16) return ODPD_EUNSUPPORTED;
    if (!a0.ckpt || !a0.dx || !a0.partials || !a0.target) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = gru_s16_groups(a.B);
    const int P = s16_param_count(m->hidden, FM, DG);
    ODPD_S16_DISPATCH(launch_s16_lossdx_occ, st, a, P)
}

template <int FM, bool DG, int OCC, bool PACK>
static int launch_s16_sweep(hipStream_t st, const SeqArgs& a, int P, const SweepRun* runs, int K, long long first) {
    const LaunchShape ls = s16_shape(a.ngroups);
    const size_t lds = s16_lds_bytes(P, ls.waves);
    auto k = gru16_train_sweep_kernel<FM, DG, OCC, PACK>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)ls.grid * K), dim3(64 * ls.waves), lds, st, a, runs, ls.grid, first);
    return (int)hipGetLastError();
}
template <int FM, bool DG>
static int launch_s16_sweep_occ(hipStream_t st, const SeqArgs& a, int P, const SweepRun* runs, int K, long long first) {
    constexpr bool kCanPack = S16Cfg<FM>::NCH == 2;
    const bool pack = kCanPack && a.H <= 13;
    if (s16_occupancy(a.ngroups) == 1) return launch_s16_sweep<FM, DG, 1, false>(st, a, P, runs, K, first);      // (as launch_s16_occ chooses for this batch)
    if constexpr (kCanPack) { if (pack) return launch_s16_sweep<FM, DG, 2, true>(st, a, P, runs, K, first); }
    return launch_s16_sweep<FM, DG, 2, false>(st, a, P, runs, K, first);
}
bool gru_s16_sweep_ok(const odpd_model_t* m) {
    int FM; bool DG;
    return s16_cfg(m, FM, DG) && m->hidden <= 16 && m->bits_w == 0 && !(m->flags & ODPD_FLAG_TWO_LAYERS);
}
int gru_s16_sweep_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, const SweepRun* runs, int K, long long first) {
    int FM; bool DG;
    if (!s16_cfg(m, FM, DG) || !gru_s16_sweep_ok(m) || K <= 0 || !runs) return ODPD_EUNSUPPORTED;
    SeqArgs a = a0;
    a.ngroups = gru_s16_groups(a.B);
    const int P = s16_param_count(m->hidden, FM, DG);
    if (FM == FEAT_RAW2) return launch_s16_sweep_occ<FEAT_RAW2, false>(st, a, P, runs, K, first);
    if (FM == FEAT_DGRU6) return launch_s16_sweep_occ<FEAT_DGRU6, true>(st, a, P, runs, K, first);
    if (FM == FEAT_Q4) return launch_s16_sweep_occ<FEAT_Q4, false>(st, a, P, runs, K, first);
    return launch_s16_sweep_occ<FEAT_A4, false>(st, a, P, runs, K, first);
}

int gru_s16_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0) {
    int FM; bool DG;
    if (!s16_cfg(m, FM, DG) || m->hidden > 16) return ODPD_EUNSUPPORTED;
    if (!a0.ckpt) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = gru_s16_groups(a.B);
    const int P = s16_param_count(m->hidden, FM, DG);
    if (FM == FEAT_RAW2) return launch_s16_occ<FEAT_RAW2, false>(st, a, P);
    if (FM == FEAT_DGRU6) return launch_s16_occ<FEAT_DGRU6, true>(st, a, P);
    if (FM == FEAT_Q4) return launch_s16_occ<FEAT_Q4, false>(st, a, P);
    return launch_s16_occ<FEAT_A4, false>(st, a, P);
}

}  // namespace odpd
