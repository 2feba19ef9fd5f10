// odpd_qatseq.h — a quantisation-aware GRUCell model (the surgery's result on gru / qgru / qgru_amp1: quant/modules/gru.py:43-59, INT_Linear
// fc_out) as the trained DPD of a cascade on ONE wave (BASELINE config 5: QGRU INT8 train_dpd) — the per-unit arithmetic of qat_s16.hip's
// std_cell / head_fwd / head_bwd / backward block in grid units (odpd_qat.h: every rounding the reference makes is made on the same real
// number; on 8-bit grids the integer mat-vec sums are exact in any order, so the results stay bit-identical with the reference's), on the
// gate-parallel mapping of the one-sequence-per-wave kernels: rows r | z | n | - of the wave hold one gate each — the row's quantised
// input weights as wave-uniform FMAs on the step's quantised features, its quantised recurrent weights as one rotated dot product on the
// quantised state — the three gates' sums meet on every row through cross-row swaps and the quantiser chain of a unit runs redundantly on
// the four rows.  The cell state (h only) is kept at every chunk start; a backward chunk runs its forward steps again from there, parking
// the step's pre-combined straight-through factors (qat_s16.hip, SaveS) for its 32 steps, then back-propagates: one transposed rotated dot
// product per step, weight gradients as two 4-block MFMAs on grid-unit operands, scaled (and masked by the weight quantisers' pass
// ranges) once at write-out; the scale parameters get an exact 0.  hidden <= 16.
// K_DGRU (r04): the quantised dgru (GRUCell on the six features, fc_hid + relu + cat([hid, features]) + fc_out, all INT_Linear: dgru.py:59-74
// under the surgery) for evaluation passes and the train_pa step.  Its head runs with LANE = TIME STEP on the chunk (dg_head): q_a(h) on
// fc_hid's grid, the H x H integer mat-vec against q_w(W_hid) read as LDS broadcasts, scale + bias in one FMA, relu, fc_out's activation
// grid on [hid | features]; backwards the same lanes form dL/d(pre-activation) and W_hid^T of it (handed to the serial loop through LDS),
// dW_hid is a 16 x 16 x 4 MFMA over the chunk's time steps, the head's vectors are summed with lane = unit.
#pragma once
#include "odpd_qat.h"
#include "odpd_seq.h"

namespace odpd {
namespace q16 {

// NB: unit blocks (1: hidden <= 16; 2: hidden 17..32 — two 16-unit blocks per gate row, as GpSeq)
template <int MK, bool LUT, int NB = 1>
struct QatSeq {
    static_assert(MK == K_GRU || MK == K_Q4 || MK == K_A4 || MK == K_DGRU, "GRUCell kinds");
    static constexpr bool DG = MK == K_DGRU;
    static constexpr int F = Kind<MK>::F, C = 32, HB = 16 * NB;      // C = kCascChunk (odpd_gpseq.h)
    static constexpr int FQ = DG ? 16 : 4;              // floats per step in fq: the cell's quantised features (DG: [0..5]; [8..13] = the same six on fc_out's grid)
    static constexpr int kDgFloats = DG ? 16 + HB + HB * HB + 4 * C * HB : 0;      // wof [2][8] | bhid [HB] | q_w(W_hid) [HB][HB] | dpre, hok, headg, h2k [C][HB]
    static constexpr int NSV = 14;                      // parked per unit and step: hp hqk n z c2 c3 An Az B1 B2A pph hnew hok pho
    static constexpr int kTabFloats = 6 * NB * 4 * 64 * 4;   // q_w(W_h) rotated rows (gate, relative input block) + transposed
    __host__ __device__ static int tp(int T) { return (T + 63) & ~63; }
    __host__ __device__ static int nchunks(int T) { return (T + C - 1) / C; }
    __host__ __device__ static int off_buf() { return LUT ? 4 * 256 : 0; }                         // LUT builds (<= 8 activation bits): [256][4] first; tables | buffers start here
    __host__ __device__ static int off_ck(int T) { return tp(T) * FQ; }                            // fq [Tp][FQ]: quantised features (grid units)
    __host__ __device__ static int off_dyb(int T) { return off_ck(T) + nchunks(T) * 64 * NB; }     // ck [chunks][NB][64]: h at the chunk start
    __host__ __device__ static int off_sv(int T) { return off_dyb(T) + tp(T) * 2; }                // dyb [Tp][2]: dL/du(t), written by the PA wave
    __host__ __device__ static int off_hist(int T) { return off_sv(T) + C * HB * NSV; }            // sv [C][NSV][HB]
    __host__ __device__ static int off_dump(int T) { return off_hist(T) + (C + 1) * HB; }          // hist [C + 1][HB]: entry i + 1 = h(t0 + i)
    __host__ __device__ static int off_hw(int T) { return off_dump(T) + 512 * NB; }                // dump: where the rows that park nothing store
    __host__ __device__ static int buf_floats(int T) { return off_hw(T) + 2 * HB + kDgFloats; }
    __host__ __device__ static int region_floats(int T, int P) {
        const int buf = buf_floats(T);
        return pad4(P) + off_buf() + (buf > kTabFloats ? buf : kTabFloats);
    }
    // (dL/du buffer of the region, for the PA wave)
    __host__ __device__ static int off_dyb_region(int T, int P) { return pad4(P) + off_buf() + off_dyb(T); }

    // ---- registers ----
    float *wofp, *bhidp, *whidp, *dpreb, *hokb, *headg, *h2kb;      // DG: the head's LDS tables and chunk buffers
    f32x4 thid[NB][NB];                                             // DG: dW_hid tiles (dpre x q_a(h), grid units on the B side)
    float dg_w0, dg_w1, dg_bh;                                      // DG: lane = unit (lanes 32 .. 37: feature slot): fc_out column / fc_hid bias gradients
    float wrec[NB][NB][16], wT[NB][NB][16], wx[NB][F], bx[NB][3], bh[NB][3], wo0[NB], wo1[NB], bo0, bo1;
    QSc qs;
    QK k;
    WQ wq;
    QatLayout L;
    float h[NB], gh[NB];                      // cell state (replicated on every row); backward carry dL/dh
    f32x16 acc1[NB][NB], acc2[NB];
    float dwo0[NB], dwo1[NB], dbo0, dbo1, dbhn[NB];
    float *smem, *pl, *fq, *ck, *dyb, *sv, *hist, *dump, *hw;
    const float4* lutq;
    RowMasks rm;
    int H, T, lane, col, role, bits_a, svp0, svp_step, hp0, hp_step, fbase;
    bool vo[NB], ring, eval_out;        // ring: forward-only use on frames of any length (buffers of ONE chunk); eval_out: fc_out's 16-bit output quantiser

    // (one workgroup barrier inside)  Tb: the frame length the buffers are laid out for (T, or one chunk for forward-only use)
    __device__ __forceinline__ void setup(float* base, float* region, const float* params, int Hm, int T_, int Tb, int bits_w, int bits_a_) {
        smem = base; ring = false; eval_out = false; fbase = 0;
        lane = threadIdx.x & 63; col = lane & 15; role = lane >> 4;      // r | z | n | -
        L = qat_layout(MK, Hm);
        H = L.H; T = T_; bits_a = bits_a_;
        pl = region;
        for (int i = lane; i < L.P; i += 64) pl[i] = params[i];
        wave_lds_fence();
        qs = load_qsc<MK>(pl, L, bits_a);
        wq = make_wq(pl, L, bits_w);
        k = make_qk(qs, wq);
        float* lut = region + pad4(L.P);
        float* tab = lut + off_buf();
        const bool gate_row = role < 3;
        const int g = gate_row ? role : 0;
        if constexpr (LUT) {      // as fill_luts (odpd_qat.h), one wave
            float4* l4 = reinterpret_cast<float4*>(lut);
            for (int i = lane; i < (1 << bits_a); i += 64) {
                const double x = (double)((float)(i + (int)qs.add.qn) * qs.add.s);
                const Gate gs = sig_gate((float)(1.0 / (1.0 + exp(-x))), qs, k), gt = tanh_gate((float)tanh(x), qs, k);
                l4[i] = make_float4(gs.c, gs.d, gt.c, gt.d);
            }
        }
        lutq = reinterpret_cast<const float4*>(lut) - (int)qs.add.qn;
        {   // rotated-quad tables of the quantised recurrent weights (grid units), the layout of fill_gru_tabs (odpd_gru.h): lane (ob = its
            // 16-lane row & (NB - 1), col) holds output unit 16 ob + col; row gg NB + rb: against input block (ob + rb) % NB; rows 3 NB ..: transposed
            const int dir = rot_dir(col), ob = (lane >> 4) & (NB - 1), o = 16 * ob + col;
            float4* t4 = reinterpret_cast<float4*>(tab);
            for (int idx = 0; idx < 6 * NB * 4; ++idx) {
                const int tr = idx >> 2, q = idx & 3;
                const bool transposed = tr >= 3 * NB;
                const int local = transposed ? tr - 3 * NB : tr, gg = local / NB, rb = local % NB;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = 16 * ((ob + rb) % NB) + ((col + dir * (4 * q + e)) & 15);
                    const bool ok = o < H && m < H;
                    v[e] = ok ? kq(pl[L.o_wh + (gg * H + (transposed ? m : o)) * H + (transposed ? o : m)], wq.h) : 0.0f;
                }
                t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const int o = 16 * ob + col;
            vo[ob] = o < H;
            TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + 16 * ob + col);
#pragma unroll
            for (int rb = 0; rb < NB; ++rb) {
                const int kb = (ob + rb) % NB;
                load_rot(wrec[ob][kb], tl + (g * NB + rb) * 4 * 64);
                load_rot(wT[ob][kb], tl + (3 * NB + g * NB + rb) * 4 * 64);
#pragma unroll
                for (int i = 0; i < 16; ++i) { wrec[ob][kb][i] = gate_row ? wrec[ob][kb][i] : 0.0f; wT[ob][kb][i] = gate_row ? wT[ob][kb][i] : 0.0f; }
            }
#pragma unroll
            for (int i = 0; i < F; ++i) wx[ob][i] = (vo[ob] && gate_row) ? kq(pl[L.o_wx + (g * H + o) * F + i], wq.x) : 0.0f;
#pragma unroll
            for (int gg = 0; gg < 3; ++gg) { bx[ob][gg] = vo[ob] ? pl[L.o_bx + gg * H + o] : 0.0f; bh[ob][gg] = vo[ob] ? pl[L.o_bh + gg * H + o] : 0.0f; }
            wo0[ob] = vo[ob] ? kq(pl[L.o_wo + o], wq.o) : 0.0f; wo1[ob] = vo[ob] ? kq(pl[L.o_wo + L.OW + o], wq.o) : 0.0f;
        }
        bo0 = pl[L.o_bo]; bo1 = pl[L.o_bo + 1];
        wave_lds_fence();
        fq = tab; ck = tab + off_ck(Tb); dyb = tab + off_dyb(Tb); sv = tab + off_sv(Tb); hist = tab + off_hist(Tb); dump = tab + off_dump(Tb);
        hw = tab + off_hw(Tb);
        for (int i = lane; i < 2 * HB; i += 64) hw[i] = (i % HB) < H ? kq(pl[L.o_wo + (i / HB) * L.OW + (i % HB)], wq.o) : 0.0f;
        if constexpr (DG) {
            wofp = hw + 2 * HB; bhidp = wofp + 16; whidp = bhidp + HB; dpreb = whidp + HB * HB; hokb = dpreb + C * HB; headg = hokb + C * HB;
            h2kb = headg + C * HB;
            if (lane < 16) wofp[lane] = (lane & 7) < 6 ? kq(pl[L.o_wo + (lane >> 3) * L.OW + H + (lane & 7)], wq.o) : 0.0f;
            for (int i = lane; i < HB; i += 64) bhidp[i] = i < H ? pl[L.o_bhid + i] : 0.0f;
            for (int i = lane; i < HB * HB; i += 64) whidp[i] = ((i / HB) < H && (i % HB) < H) ? kq(pl[L.o_whid + (i / HB) * H + (i % HB)], wq.hid) : 0.0f;
            for (int i = lane; i < 4 * C * HB; i += 64) dpreb[i] = 0.0f;      // (columns >= H are never written again: zero operands of the dW_hid tiles)
#pragma unroll
            for (int mt = 0; mt < NB; ++mt)
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) thid[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            dg_w0 = 0.0f; dg_w1 = 0.0f; dg_bh = 0.0f;
        }
        rm = row_masks();
        // the recomputed steps' stores: row 0 parks the unit's NSV factors, row 3 h(t) (the others hit the dump)
        svp0 = role == 0 ? (int)(sv - smem) + col : (int)(dump - smem) + lane; svp_step = role == 0 ? HB * NSV : 0;
        hp0 = role == 3 ? (int)(hist - smem) + HB + col : (int)(dump - smem) + 320 * NB + lane; hp_step = role == 3 ? HB : 0;
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc2[ob][i] = 0.0f;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) acc1[ob][kb][i] = 0.0f;
            }
            dwo0[ob] = 0.0f; dwo1[ob] = 0.0f; dbhn[ob] = 0.0f;
        }
        dbo0 = 0.0f; dbo1 = 0.0f;
        wave_lds_fence();
    }

    // the model's quantised input features of one sample, grid units (qat_s16.hip: q16_slots' operation order, then q_a)
    __device__ __forceinline__ void put_features(float* dst, float2 xv) const {
        const float I = xv.x, Q = xv.y;
        float f[8] = {I, Q, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (MK == K_Q4) { const float a2 = I * I + Q * Q; f[2] = a2; f[3] = a2 * a2; }
        if constexpr (MK == K_A4) { const float a2 = I * I + Q * Q, a = sqrtf(a2); f[2] = a; f[3] = a * a * a; }
        if constexpr (DG) { const float a2 = I * I + Q * Q, a = sqrtf(a2); f[2] = a; f[3] = a * a * a; f[4] = Q / a; f[5] = I / a; }      // dgru.py:61-68
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = i < F ? gk(f[i] * k.inv_xa, k) : 0.0f;
        reinterpret_cast<float4*>(dst)[0] = make_float4(r[0], r[1], r[2], r[3]);
        if constexpr (DG) {      // ... and as fc_out's inputs (the cat's feature part on fc_out's activation grid)
            reinterpret_cast<float4*>(dst)[1] = make_float4(r[4], r[5], 0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = i < F ? gk(f[i] * k.inv_oa, k) : 0.0f;
            reinterpret_cast<float4*>(dst)[2] = make_float4(r[0], r[1], r[2], r[3]);
            reinterpret_cast<float4*>(dst)[3] = make_float4(r[4], r[5], 0.0f, 0.0f);
        }
    }

    __device__ __forceinline__ void fwd_begin() {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) h[ob] = 0.0f;
    }

    // one GRUCell step (std_cell, qat_s16.hip) at time t.  SAVE: parks the backward's factors of the step at `sp` / h(t) at `hq`
    template <bool SAVE>
    __device__ __forceinline__ void step(int t, int sp, int hq_) {
        const float4* fp4 = reinterpret_cast<const float4*>(fq + (t - fbase) * FQ);
        const float4 f4 = fp4[0], f5 = DG ? fp4[1] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float ff[8] = {f4.x, f4.y, f4.z, f4.w, f5.x, f5.y, 0.0f, 0.0f};
        float v0[NB], m0[NB], hqk[NB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) { v0[ob] = h[ob] * k.inv_ha; m0[ob] = gm(v0[ob], k); hqk[ob] = rintf(m0[ob]); }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const float hv = h[ob];
            float xsum = 0.0f;
#pragma unroll
            for (int i = 0; i < F; ++i) xsum = __builtin_fmaf(wx[ob][i], ff[i], xsum);
            float hsum = 0.0f;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) hsum = rotdot(hsum, wrec[ob][kb], hqk[kb]);
            float X[4], Hs[4];
            gather_rows(xsum, X);
            gather_rows(hsum, Hs);
            // x_t = x2h(x), h_t = h2h(h): exact integer sums, scale and fp32 bias in one FMA (== F.linear's result)
            const float xr = __builtin_fmaf(X[0], k.Sx, bx[ob][0]), hr = __builtin_fmaf(Hs[0], k.Sh, bh[ob][0]);
            const float xz = __builtin_fmaf(X[1], k.Sx, bx[ob][1]), hz = __builtin_fmaf(Hs[1], k.Sh, bh[ob][1]);
            const float xn = __builtin_fmaf(X[2], k.Sx, bx[ob][2]), hn = __builtin_fmaf(Hs[2], k.Sh, bh[ob][2]);
            const float vr = (xr + hr) * k.inv_add, vz = (xz + hz) * k.inv_add;
            const float mr = gm(vr, k), mz = gm(vz, k);
            const Gate Gr = sig_grid<LUT>(rintf(mr), qs, k, lutq), Gz = sig_grid<LUT>(rintf(mz), qs, k, lutq);
            const float pm1 = Gr.c * hn, mm1 = gm(pm1, k);                                     // Qmul(r h_n)
            const float vn = __builtin_fmaf(rintf(mm1), k.s_mul, xn) * k.inv_add, mn = gm(vn, k);      // Qadd(x_n + .)
            const Gate Gn = tanh_grid<LUT>(rintf(mn), qs, k, lutq);
            const float omz = __builtin_fmaf(Gz.c, -k.s_mul, 1.0f);                             // 1 - z, plain
            const float pm2 = Gz.c * hv, pm3 = omz * Gn.c;
            const float mm2 = gm(pm2, k), mm3 = gm(pm3, k);
            const float vh = (rintf(mm2) + rintf(mm3)) * k.c_ma, mh = gm(vh, k);
            const float hnew = rintf(mh) * k.s_add;
            if constexpr (SAVE) {
                const bool pah = mh == vh;
                const float Ar = mr == vr ? Gr.d : 0.0f;
                const bool p1 = mm1 == pm1;
                const float vo_ = hnew * k.inv_oa, mo = gm(vo_, k);
                float* s = smem + sp + 16 * ob;
                s[0 * HB] = hv; s[1 * HB] = hqk[ob]; s[2 * HB] = Gn.c * k.s_mul; s[3 * HB] = Gz.c * k.s_mul;
                s[4 * HB] = (pah && mm2 == pm2) ? 1.0f : 0.0f; s[5 * HB] = (pah && mm3 == pm3) ? 1.0f : 0.0f;
                s[6 * HB] = mn == vn ? Gn.d : 0.0f; s[7 * HB] = mz == vz ? Gz.d : 0.0f;
                s[8 * HB] = p1 ? Gr.c * k.s_mul : 0.0f; s[9 * HB] = p1 ? hn * Ar : 0.0f;
                s[10 * HB] = m0[ob] == v0[ob] ? k.s_hw : 0.0f; s[11 * HB] = hnew;
                s[12 * HB] = rintf(mo); s[13 * HB] = mo == vo_ ? k.s_ow : 0.0f;
            }
            h[ob] = hnew;
            smem[hq_ + 16 * ob] = hnew;
        }
    }

    // dgru's head on the chunk's steps, LANE = TIME STEP (head_fwd / head_bwd of qat_s16.hip per step): forward -> sink(t, y0, y1);
    // BWD: dL/dpre (fc_hid's pre-activation) and q_a(relu) of every unit -> dpreb / hokb, q_a(h) -> h2kb, W_hid^T dL/dpre through fc_hid's
    // activation mask (carrying s_hidw) -> headg, rows of steps beyond the chunk's length zero
    template <bool BWD, typename Sink>
    __device__ __forceinline__ void dg_head(int t0, int len, Sink sink) {
        const bool on = lane < len;
        const int tl = on ? lane : 0;
        const float* hv = hist + (tl + 1) * HB;
        float h2k[HB], back[HB];
#pragma unroll
        for (int u = 0; u < HB; ++u) { h2k[u] = gk(hv[u] * k.inv_hida, k); back[u] = 0.0f; }
        float2 dyv = make_float2(0.0f, 0.0f);
        if constexpr (BWD) dyv = *reinterpret_cast<const float2*>(dyb + 2 * (t0 + tl));
        float p0 = 0.0f, p1 = 0.0f;
        for (int m = 0; m < H; ++m) {
            const float4* wr = reinterpret_cast<const float4*>(whidp + m * HB);
            float w[HB];
#pragma unroll
            for (int q = 0; q < HB / 4; ++q) { const float4 w4 = wr[q]; w[4 * q] = w4.x; w[4 * q + 1] = w4.y; w[4 * q + 2] = w4.z; w[4 * q + 3] = w4.w; }
            float acc = 0.0f;
#pragma unroll
            for (int u = 0; u < HB; ++u) acc = __builtin_fmaf(w[u], h2k[u], acc);
            const float pre = __builtin_fmaf(acc, k.Shid, bhidp[m]), hid = pre > 0.0f ? pre : 0.0f;      // torch.relu
            const float v = hid * k.inv_oa, mm = gm(v, k), hok = rintf(mm);
            if constexpr (!BWD) { p0 = __builtin_fmaf(hw[m], hok, p0); p1 = __builtin_fmaf(hw[HB + m], hok, p1); }
            else {
                const float pho = mm == v ? k.s_ow : 0.0f;
                const float dcat = (dyv.x * hw[m] + dyv.y * hw[HB + m]) * pho;
                const float dpre = (on && pre > 0.0f) ? dcat : 0.0f;
                if (lane < C) { dpreb[lane * HB + m] = dpre; hokb[lane * HB + m] = on ? hok : 0.0f; }
#pragma unroll
                for (int u = 0; u < HB; ++u) back[u] = __builtin_fmaf(w[u], dpre, back[u]);
            }
        }
        if constexpr (!BWD) {
            const float* fr = fq + (t0 + tl - fbase) * FQ + 8;
#pragma unroll
            for (int c = 0; c < 6; ++c) { p0 = __builtin_fmaf(wofp[c], fr[c], p0); p1 = __builtin_fmaf(wofp[8 + c], fr[c], p1); }
            float y0 = __builtin_fmaf(p0, k.So, bo0), y1 = __builtin_fmaf(p1, k.So, bo1);
            if (eval_out) { y0 = qapply(y0, qs.out); y1 = qapply(y1, qs.out); }
            if (on) sink(t0 + lane, y0, y1);
        } else if (lane < C) {
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const float v = hv[u] * k.inv_hida;
                headg[lane * HB + u] = back[u] * (gm(v, k) == v ? k.s_hidw : 0.0f);
                h2kb[lane * HB + u] = h2k[u];
            }
        }
    }
    // ... and the head's parameter gradients of the chunk: dW_hid as 16 x 16 x 4 MFMAs over the time steps (A = dL/dpre, B = q_a(h)), the
    // vectors with lane = unit (lanes 32 .. 37: fc_out's feature columns)
    __device__ __forceinline__ void dg_reduce(int t0, int len) {
#pragma unroll
        for (int kk = 0; kk < C / 4; ++kk) {
            float av[NB], bv[NB];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) { av[ob] = dpreb[(4 * kk + role) * HB + 16 * ob + col]; bv[ob] = h2kb[(4 * kk + role) * HB + 16 * ob + col]; }
#pragma unroll
            for (int mt = 0; mt < NB; ++mt)
#pragma unroll
                for (int nt = 0; nt < NB; ++nt) thid[mt][nt] = mfma4(av[mt], bv[nt], thid[mt][nt]);
        }
        const int fslot = lane - 32;
        const bool isu = lane < HB, isf = fslot >= 0 && fslot < 6;
        if (isu || isf) {
            for (int tt = 0; tt < len; ++tt) {
                const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * (t0 + tt));
                const float a = isu ? hokb[tt * HB + lane] : fq[(t0 + tt) * FQ + 8 + fslot];
                dg_w0 = __builtin_fmaf(dyv.x, a, dg_w0); dg_w1 = __builtin_fmaf(dyv.y, a, dg_w1);
                if (isu) dg_bh += dpreb[tt * HB + lane];
            }
        }
    }

    // forward chunk c: quantised features with lane = time step, the cell state kept, the recurrence, fc_out with lane = time step
    template <typename Sink>
    __device__ __forceinline__ void fwd_chunk(int c, int t0, int len, const float2* xg, Sink sink) {
        fbase = ring ? t0 : 0;
        if (lane < len) put_features(fq + (t0 + lane - fbase) * FQ, xg[t0 + lane]);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) ck[((ring ? 0 : c) * NB + ob) * 64 + lane] = h[ob];
        wave_lds_fence();
        int hq_ = hp0;
        for (int tt = 0; tt < len; ++tt) { step<false>(t0 + tt, 0, hq_); hq_ += hp_step; }
        wave_lds_fence();
        if constexpr (DG) dg_head<false>(t0, len, sink);
        else if (lane < len) {
            const float* hv = hist + (lane + 1) * HB;
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                const float hok = gk(hv[u] * k.inv_oa, k);
                p0 = __builtin_fmaf(hw[u], hok, p0); p1 = __builtin_fmaf(hw[HB + u], hok, p1);
            }
            float y0 = __builtin_fmaf(p0, k.So, bo0), y1 = __builtin_fmaf(p1, k.So, bo1);
            if (eval_out) { y0 = qapply(y0, qs.out); y1 = qapply(y1, qs.out); }                   // fc_out's 16-bit out_quantizer, eval mode only (quant_layers.py:77-80)
            sink(t0 + lane, y0, y1);
        }
        wave_lds_fence();
    }

    __device__ __forceinline__ void bwd_begin() {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) gh[ob] = 0.0f;
    }

    // backward chunk c: forward steps again from the kept state (parking the factors), then the steps t0 + len - 1 .. t0
    __device__ __forceinline__ void bwd_chunk(int c, int t0, int len) {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) h[ob] = ck[(c * NB + ob) * 64 + lane];
        wave_lds_fence();
        {
            int sp = svp0, hq_ = hp0;
            for (int tt = 0; tt < len; ++tt) { step<true>(t0 + tt, sp, hq_); sp += svp_step; hq_ += hp_step; }
        }
        wave_lds_fence();
        if constexpr (DG) {
            dg_head<true>(t0, len, [](int, float, float) {});
            wave_lds_fence();
            dg_reduce(t0, len);
        }
        for (int tt = len - 1; tt >= 0; --tt) {
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * (t0 + tt));
            dbo0 += dyv.x; dbo1 += dyv.y;
            float d_h[NB], d_x[NB], dhdir[NB], pph[NB], hqk[NB];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                const float* s = sv + tt * HB * NSV + 16 * ob + col;
                const float hp_ = s[0], n = s[2 * HB], z = s[3 * HB], c2 = s[4 * HB], c3 = s[5 * HB], An = s[6 * HB], Az = s[7 * HB],
                            B1 = s[8 * HB], B2A = s[9 * HB], hok = s[12 * HB], pho = s[13 * HB];
                hqk[ob] = s[HB]; pph[ob] = s[10 * HB];
                // head (head_bwd): fc_out's parameter gradients, dL/dh' through the activation quantiser's pass mask (carrying s_ow)
                float g;
                if constexpr (DG) g = gh[ob] + headg[tt * HB + 16 * ob + col];      // (dg_head: W_hid^T dL/dpre through fc_hid's activation mask)
                else {
                    dwo0[ob] = __builtin_fmaf(dyv.x, hok, dwo0[ob]); dwo1[ob] = __builtin_fmaf(dyv.y, hok, dwo1[ob]);
                    g = gh[ob] + (dyv.x * wo0[ob] + dyv.y * wo1[ob]) * pho;
                }
                const float g2 = g * c2, g3 = g * c3;
                const float dz = g2 * hp_ - g3 * n;
                const float da = g3 * (1.0f - z) * An;
                const float dhtn = da * B1, dar = da * B2A, daz = dz * Az;
                dhdir[ob] = g2 * z;
                dbhn[ob] += dhtn;
                d_h[ob] = vsel(rm.m[0], dar, vsel(rm.m[1], daz, vsel(rm.m[2], dhtn, 0.0f)));
                d_x[ob] = vsel(rm.m[0], dar, vsel(rm.m[1], daz, vsel(rm.m[2], da, 0.0f)));
            }
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                float ddh = 0.0f;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) ddh = rotdot(ddh, wT[ob][kb], d_h[kb]);
                ddh = sum_rows4(ddh);
                gh[ob] = dhdir[ob] + ddh * pph[ob];
            }
            // weight gradients on grid-unit operands: (d_r | d_z | d_hn) x q_a(h), (d_r | d_z | d_n) x (q_a(features) | 1 / s_xa)
            float fsx;
            if constexpr (DG) fsx = col < F ? fq[(t0 + tt) * FQ + col] : (col == F ? k.inv_xa : 0.0f);
            else {
                const float4 f4 = reinterpret_cast<const float4*>(fq)[t0 + tt];
                fsx = col == 0 ? f4.x : col == 1 ? f4.y : (col == 2 && F > 2) ? f4.z : (col == 3 && F > 3) ? f4.w : (col == F ? k.inv_xa : 0.0f);
            }
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) acc1[ob][kb] = __builtin_amdgcn_mfma_f32_16x16x1f32(d_h[ob], hqk[kb], acc1[ob][kb], 0, 0, 0);
                acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x1f32(d_x[ob], fsx, acc2[ob], 0, 0, 0);
            }
        }
        wave_lds_fence();
    }

    // the workgroup's row of partial gradients (q16_write_row): activation scales and weight pass masks applied here, scale parameters 0
    __device__ __forceinline__ void write_partials(float* prow, float loss) {
        for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.0f;
        wave_lds_fence();
        __builtin_amdgcn_s_waitcnt(0);
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const int o = 16 * ob + col;
            if (vo[ob] && role == 0) {
                if constexpr (!DG) {
                    prow[L.o_wo + o] = dwo0[ob] * k.s_oa * qpass(pl[L.o_wo + o], wq.o);
                    prow[L.o_wo + L.OW + o] = dwo1[ob] * k.s_oa * qpass(pl[L.o_wo + L.OW + o], wq.o);
                }
                prow[L.o_bh + 2 * H + o] = dbhn[ob];
            }
        }
        if constexpr (DG) {
            const int fslot = lane - 32;
            const int oc = lane < HB ? (lane < H ? lane : -1) : ((fslot >= 0 && fslot < 6) ? H + fslot : -1);      // fc_out column of the lane: unit | feature slot
            if (oc >= 0) {
                prow[L.o_wo + oc] = dg_w0 * k.s_oa * qpass(pl[L.o_wo + oc], wq.o);
                prow[L.o_wo + L.OW + oc] = dg_w1 * k.s_oa * qpass(pl[L.o_wo + L.OW + oc], wq.o);
                if (oc < H) prow[L.o_bhid + oc] = dg_bh;
            }
#pragma unroll
            for (int mt = 0; mt < NB; ++mt)
#pragma unroll
                for (int nt = 0; nt < NB; ++nt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int mrow = 16 * mt + 4 * role + rr, ucol = 16 * nt + col;
                        if (mrow < H && ucol < H) { const int j = L.o_whid + mrow * H + ucol; prow[j] = thid[mt][nt][rr] * k.s_hida * qpass(pl[j], wq.hid); }
                    }
        }
        if (lane == 0) { prow[L.o_bo] = dbo0; prow[L.o_bo + 1] = dbo1; prow[L.P] = loss; }
        // MFMA block g = gate g (r, z, n); register 4 g + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int i = 16 * ob + 4 * role + rr;
                    if (i < H) {
#pragma unroll
                        for (int kb = 0; kb < NB; ++kb) {
                            const int jc = 16 * kb + col;
                            if (jc < H) { const int j = L.o_wh + (g * H + i) * H + jc; prow[j] = acc1[ob][kb][4 * g + rr] * k.s_ha * qpass(pl[j], wq.h); }
                        }
                        const float tx = acc2[ob][4 * g + rr] * k.s_xa;
                        if (col < F) { const int j = L.o_wx + (g * H + i) * F + col; prow[j] = tx * qpass(pl[j], wq.x); }
                        else if (col == F) { prow[L.o_bx + g * H + i] = tx; if (g < 2) prow[L.o_bh + g * H + i] = tx; }
                    }
                }
    }
};

// The quantised TRes-DeltaGRU (the OpenDPDv2 recipe: deltagru_tcnskip.py:156-162, 266-291 under the surgery; qat_s16.hip delta_cell) as the
// DPD wave: thresholded deltas of the six features (one feature per lane, as DeltaSeq) and of the state, quantised (q_a) and multiplied with
// the quantised bias-free x2h / h2h weights — integer sums, scaled by one FMA each into the four fp32 accumulators of a unit —, sigmoids of
// the raw accumulators through the boundary table (LUT builds), Qadd(1, -z), the float TCN skip with lane = time step.  The cell state
// (h, h_p, the four accumulators, x_p: 7 floats per lane) is kept at every chunk start; a backward chunk runs its forward steps again from
// there and parks the step's factors (SaveD), then back-propagates with CARRIED accumulator gradients (G_r, G_z, G_n, G_nh: the
// accumulators are running sums, so their gradients are, backwards).  hidden <= 16.
template <bool LUT>
struct QatDeltaSeq {
    static constexpr int F = 6, C = 32;
    static constexpr int NSV = 15;                      // parked per unit and step: hp qdhk npo omz z c2 c3 An Az B1 B2A mh pph hok pho
    static constexpr int kTabFloats = 6 * 4 * 64 * 4;
    __host__ __device__ static int tp(int T) { return (T + 63) & ~63; }
    __host__ __device__ static int nchunks(int T) { return (T + C - 1) / C; }
    __host__ __device__ static int off_buf() { return LUT ? 4 * 256 + kMaxThr + 4 : 0; }              // LUT | boundary table first; tables | buffers start here
    __host__ __device__ static int off_ck(int T) { return tp(T) * 8; }                                // feat [Tp][8]: f0..f5 (float), skip0, skip1
    __host__ __device__ static int off_dyb(int T) { return off_ck(T) + nchunks(T) * 7 * 64; }         // ck [chunks][7][64]
    __host__ __device__ static int off_sv(int T) { return off_dyb(T) + tp(T) * 2; }
    __host__ __device__ static int off_fqs(int T) { return off_sv(T) + C * 16 * NSV; }                // sv [C][NSV][16]
    __host__ __device__ static int off_hist(int T) { return off_fqs(T) + C * 8; }                     // fqs [C][8]: the step's six quantised masked dx
    __host__ __device__ static int off_dump(int T) { return off_hist(T) + (C + 1) * 16; }
    __host__ __device__ static int off_hw(int T) { return off_dump(T) + 512; }
    __host__ __device__ static int buf_floats(int T) { return off_hw(T) + 32; }
    __host__ __device__ static int region_floats(int T, int P) {
        const int buf = buf_floats(T);
        return pad4(P) + off_buf() + (buf > kTabFloats ? buf : kTabFloats);
    }
    __host__ __device__ static int off_dyb_region(int T, int P) { return pad4(P) + off_buf() + off_dyb(T); }

    float wrec[16], wT[16], wx[F], wo0, wo1, w1[18], w2[6], thx, thh;
    QSc qs;
    QK k;
    WQ wq;
    QatLayout L;
    float h, hp, xp, dmr, dmz, dmn, dmnh;             // cell state (replicated on every row; x_p: feature `fc` of the lane)
    float gh, ghp, gr, gz, gn, gnh;                   // backward carries
    f32x16 acc1, acc2;
    float dwo0, dwo1, tw1[18], tw2[6], zx, zh;
    float *smem, *pl, *feat, *ck, *dyb, *sv, *fqs, *hist, *dump, *hw;
    const float4* lutq;
    const float* thr;
    RowMasks rm;
    int H, T, lane, col, role, fc, Ksig, svp0, svp_step, hp0, hp_step, fq0, fq_step, fbase;
    bool vo, ring, eval_out;

    // (one workgroup barrier inside)  Tb: the frame length the buffers are laid out for (T, or one chunk for forward-only use)
    __device__ __forceinline__ void setup(float* base, float* region, const float* params, int Hm, int T_, int Tb, int bits_w, int bits_a, float thx_,
                                          float thh_) {
        smem = base; ring = false; eval_out = false; fbase = 0;
        lane = threadIdx.x & 63; col = lane & 15; role = lane >> 4;      // r | z | n | -
        L = qat_layout(K_TRES, Hm);
        H = L.H; T = T_; thx = thx_; thh = thh_;
        pl = region;
        for (int i = lane; i < L.P; i += 64) pl[i] = params[i];
        wave_lds_fence();
        qs = load_qsc<K_TRES>(pl, L, bits_a);
        wq = make_wq(pl, L, bits_w);
        k = make_qk(qs, wq);
        Ksig = sig_levels(qs.sig);
        float* lut = region + pad4(L.P);
        float* tab = lut + off_buf();
        vo = col < H;
        const bool gate_row = role < 3;
        const int g = gate_row ? role : 0;
        if constexpr (LUT) {      // as fill_luts (odpd_qat.h) with the sigmoid boundary table, one wave
            float4* l4 = reinterpret_cast<float4*>(lut);
            for (int i = lane; i < (1 << bits_a); i += 64) {
                const double x = (double)((float)(i + (int)qs.add.qn) * qs.add.s);
                const Gate gs = sig_gate((float)(1.0 / (1.0 + exp(-x))), qs, k), gt = tanh_gate((float)tanh(x), qs, k);
                l4[i] = make_float4(gs.c, gs.d, gt.c, gt.d);
            }
            float* th = lut + 4 * 256;
            for (int j = lane; j <= Ksig + 1; j += 64) {
                float t;
                if (j == 0) t = -__builtin_inff();
                else if (j == Ksig + 1) t = __builtin_inff();
                else {
                    const double p = ((double)j - 0.5) * (double)qs.sig.s;
                    if (p >= 1.0) t = __builtin_inff();
                    else {
                        const double b = log(p / (1.0 - p));
                        t = (float)b;
                        if ((double)t < b) {                                                     // smallest float >= b
                            const int bits32 = __builtin_bit_cast(int, t);
                            t = __builtin_bit_cast(float, t > 0.0f ? bits32 + 1 : (t < 0.0f ? bits32 - 1 : 1));
                        }
                    }
                }
                th[j] = t;
            }
        }
        lutq = reinterpret_cast<const float4*>(lut) - (int)qs.add.qn;
        thr = lut + 4 * 256;
        {
            const int dir = rot_dir(col);
            float4* t4 = reinterpret_cast<float4*>(tab);
            for (int idx = 0; idx < 6 * 4; ++idx) {
                const int tr = idx >> 2, q = idx & 3, gg = tr % 3;
                const bool transposed = tr >= 3;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = (col + dir * (4 * q + e)) & 15;
                    const bool ok = col < H && m < H;
                    v[e] = ok ? kq(pl[L.o_wh + (gg * H + (transposed ? m : col)) * H + (transposed ? col : m)], wq.h) : 0.0f;
                }
                t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        {
            TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
            load_rot(wrec, tl + g * 4 * 64);
            load_rot(wT, tl + (3 + g) * 4 * 64);
#pragma unroll
            for (int i = 0; i < 16; ++i) { wrec[i] = gate_row ? wrec[i] : 0.0f; wT[i] = gate_row ? wT[i] : 0.0f; }
        }
#pragma unroll
        for (int i = 0; i < F; ++i) wx[i] = (vo && gate_row) ? kq(pl[L.o_wx + (g * H + col) * F + i], wq.x) : 0.0f;
        wo0 = vo ? kq(pl[L.o_wo + col], wq.o) : 0.0f; wo1 = vo ? kq(pl[L.o_wo + L.OW + col], wq.o) : 0.0f;
#pragma unroll
        for (int i = 0; i < 18; ++i) { w1[i] = pl[L.o_tcn0 + i]; tw1[i] = 0.0f; }
#pragma unroll
        for (int i = 0; i < 6; ++i) { w2[i] = pl[L.o_tcn2 + i]; tw2[i] = 0.0f; }
        fc = col < 6 ? col : 5;
        wave_lds_fence();
        feat = tab; ck = tab + off_ck(Tb); dyb = tab + off_dyb(Tb); sv = tab + off_sv(Tb); fqs = tab + off_fqs(Tb); hist = tab + off_hist(Tb);
        dump = tab + off_dump(Tb); hw = tab + off_hw(Tb);
        if (lane < 32) hw[lane] = (lane & 15) < H ? kq(pl[L.o_wo + (lane >> 4) * L.OW + (lane & 15)], wq.o) : 0.0f;
        rm = row_masks();
        // the recomputed steps' stores: row 0 parks the unit's NSV factors, row 3 h(t), row 1's lanes 0..7 the quantised masked dx
        svp0 = role == 0 ? (int)(sv - smem) + col : (int)(dump - smem) + lane; svp_step = role == 0 ? 16 * NSV : 0;
        hp0 = role == 3 ? (int)(hist - smem) + 16 + col : (int)(dump - smem) + 320 + lane; hp_step = role == 3 ? 16 : 0;
        fq0 = (role == 1 && col < 8) ? (int)(fqs - smem) + col : (int)(dump - smem) + 384 + lane; fq_step = (role == 1 && col < 8) ? 8 : 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
        dwo0 = 0.0f; dwo1 = 0.0f; zx = 0.0f; zh = 0.0f;
        wave_lds_fence();
    }
    // TCN skip pre-activations of one sample (q16_tcn, qat_s16.hip): float path, Conv1d / Hardswish are not swapped
    __device__ __forceinline__ void tcn(float2 xm, float2 xc, float2 xq, float (&s1)[3], float (&s2)[2]) const {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = w1[c * 6] * xm.x;
            v = __builtin_fmaf(w1[c * 6 + 1], xc.x, v); v = __builtin_fmaf(w1[c * 6 + 2], xq.x, v);
            v = __builtin_fmaf(w1[c * 6 + 3], xm.y, v); v = __builtin_fmaf(w1[c * 6 + 4], xc.y, v);
            s1[c] = __builtin_fmaf(w1[c * 6 + 5], xq.y, v);
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float v = w2[o * 3] * hardswishf_(s1[0]);
            v = __builtin_fmaf(w2[o * 3 + 1], hardswishf_(s1[1]), v);
            s2[o] = __builtin_fmaf(w2[o * 3 + 2], hardswishf_(s1[2]), v);
        }
    }

    __device__ __forceinline__ void fwd_begin() { h = 0.0f; hp = 0.0f; xp = 0.0f; dmr = 0.0f; dmz = 0.0f; dmn = 0.0f; dmnh = 0.0f; }

    // one quantised delta-cell step (delta_cell, qat_s16.hip) at time t.  SAVE: parks the backward's factors
    template <bool SAVE>
    __device__ __forceinline__ void step(int t, int sp, int hq_, int fqp) {
        // x side, one feature per lane: thresholded delta, its quantised value in grid units
        const float fv = feat[(t - fbase) * 8 + fc];
        const float d = fv - xp;
        const bool keep = !(__builtin_fabsf(d) < thx);                 // masked_fill(|d| < th, 0)  (deltagru_tcnskip.py:218-228)
        const float dxm = keep ? d : 0.0f;
        xp = (__builtin_fabsf(d) >= thx) ? fv : xp;
        if constexpr (!SAVE) zx += (dxm == 0.0f) ? 1.0f : 0.0f;
        const float vx = dxm * k.inv_xa, fqk = rintf(gm(vx, k));
        float xsum = 0.0f;
#pragma unroll
        for (int i = 0; i < F; ++i)
            xsum = __builtin_fmaf(wx[i], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fqk), i)), xsum);
        // h side
        const float hv = h;
        const float dh_ = hv - hp;
        const bool keeph = !(__builtin_fabsf(dh_) < thh);
        const float dhm = keeph ? dh_ : 0.0f;
        hp = (__builtin_fabsf(dh_) >= thh) ? hv : hp;
        if constexpr (!SAVE) zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
        const float v0 = dhm * k.inv_ha, m0 = gm(v0, k), qdhk = rintf(m0);
        const float hsum = rotdot(0.0f, wrec, qdhk);
        float X[4], Hs[4];
        gather_rows(xsum, X);
        gather_rows(hsum, Hs);
        // mac_x = x2h(dx) + dm; dm_r = mac_x_r + mac_h_r, dm_n = mac_x_n, dm_nh = mac_h_n + dm_nh  (deltagru_tcnskip.py:236-246)
        dmr = __builtin_fmaf(Hs[0], k.Sh, __builtin_fmaf(X[0], k.Sx, dmr));
        dmz = __builtin_fmaf(Hs[1], k.Sh, __builtin_fmaf(X[1], k.Sx, dmz));
        dmn = __builtin_fmaf(X[2], k.Sx, dmn);
        dmnh = __builtin_fmaf(Hs[2], k.Sh, dmnh);
        const Gate Gr = sig_any<LUT>(dmr, qs, k, thr, Ksig), Gz = sig_any<LUT>(dmz, qs, k, thr, Ksig);
        const float pm1 = Gr.c * dmnh, mm1 = gm(pm1, k);
        const float vn = __builtin_fmaf(rintf(mm1), k.s_mul, dmn) * k.inv_add, mn = gm(vn, k);
        const Gate Gn = tanh_grid<LUT>(rintf(mn), qs, k, lutq);
        const float vo_ = __builtin_fmaf(Gz.c, -k.s_mul, 1.0f) * k.inv_add, mo = gm(vo_, k);          // self.add(1, -gate_z)  (deltagru_tcnskip.py:290)
        const float omz = rintf(mo) * k.s_add;
        const float pm3 = omz * Gn.c, pm2 = Gz.c * hv;
        const float mm3 = gm(pm3, k), mm2 = gm(pm2, k);
        const float vh = (rintf(mm3) + rintf(mm2)) * k.c_ma, mhv = gm(vh, k);
        const float hnew = rintf(mhv) * k.s_add;
        if constexpr (SAVE) {
            const bool pah = mhv == vh;
            const bool p1 = mm1 == pm1;
            const float nt = Gn.c * k.s_mul;
            const float vq = hnew * k.inv_oa, mq = gm(vq, k);
            float* s = smem + sp;
            s[0 * 16] = hv; s[1 * 16] = qdhk; s[2 * 16] = mo == vo_ ? nt : 0.0f; s[3 * 16] = omz; s[4 * 16] = Gz.c * k.s_mul;
            s[5 * 16] = (pah && mm2 == pm2) ? 1.0f : 0.0f; s[6 * 16] = (pah && mm3 == pm3) ? 1.0f : 0.0f;
            s[7 * 16] = mn == vn ? Gn.d : 0.0f; s[8 * 16] = Gz.d;
            s[9 * 16] = p1 ? Gr.c * k.s_mul : 0.0f; s[10 * 16] = p1 ? dmnh * Gr.d : 0.0f;
            s[11 * 16] = keeph ? 1.0f : 0.0f; s[12 * 16] = m0 == v0 ? k.s_hw : 0.0f;
            s[13 * 16] = rintf(mq); s[14 * 16] = mq == vq ? k.s_ow : 0.0f;
            smem[fqp] = fqk;
        }
        h = hnew;
        smem[hq_] = hnew;
    }

    // forward chunk c: features and the TCN skip with lane = time step, the cell state kept, the recurrence, fc_out + skip with lane = time step
    template <typename Sink>
    __device__ __forceinline__ void fwd_chunk(int c, int t0, int len, const float2* xg, Sink sink) {
        {
            const int t = t0 + lane;
            const float2 zero = make_float2(0.0f, 0.0f);
            const float2 rc = t < T ? xg[t] : make_float2(0.5f, 0.5f);
            const float2 rn = t + 1 < T ? xg[t + 1] : xg[0];                       // torch.roll(x, -1): the last step sees sample 0
            const float2 rm_ = (t - kHalo >= 0 && t - kHalo < T) ? xg[t - kHalo] : zero;
            const float2 rp = t + kHalo < T ? xg[t + kHalo] : zero;
            const float a2 = rc.x * rc.x + rc.y * rc.y, a = sqrtf(a2), a3 = a * a * a;      // (q16_slots' operation order)
            float s1[3], s2[2];
            tcn(rm_, rc, rp, s1, s2);
            fbase = ring ? t0 : 0;
            if (lane < len) {
                reinterpret_cast<float4*>(feat)[2 * (t - fbase)] = make_float4(rc.x, rc.y, a, a3);
                reinterpret_cast<float4*>(feat)[2 * (t - fbase) + 1] = make_float4(rn.x, rn.y, hardswishf_(s2[0]), hardswishf_(s2[1]));
            }
            float* kk = ck + (ring ? 0 : c) * 7 * 64 + lane;
            kk[0] = h; kk[64] = hp; kk[128] = xp; kk[192] = dmr; kk[256] = dmz; kk[320] = dmn; kk[384] = dmnh;
        }
        wave_lds_fence();
        int hq_ = hp0;
        for (int tt = 0; tt < len; ++tt) { step<false>(t0 + tt, 0, hq_, 0); hq_ += hp_step; }
        wave_lds_fence();
        if (lane < len) {
            const int t = t0 + lane;
            const float* hv = hist + (lane + 1) * 16;
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float hok = gk(hv[u] * k.inv_oa, k);
                p0 = __builtin_fmaf(hw[u], hok, p0); p1 = __builtin_fmaf(hw[16 + u], hok, p1);
            }
            float y0 = __builtin_fmaf(p0, k.So, 0.0f), y1 = __builtin_fmaf(p1, k.So, 0.0f);       // bias-free fc_out
            if (eval_out) { y0 = qapply(y0, qs.out); y1 = qapply(y1, qs.out); }                   // its 16-bit out_quantizer, eval mode only (quant_layers.py:77-80)
            y0 += feat[(t - fbase) * 8 + 6]; y1 += feat[(t - fbase) * 8 + 7];
            sink(t, y0, y1);
        }
        wave_lds_fence();
    }

    __device__ __forceinline__ void bwd_begin() { gh = 0.0f; ghp = 0.0f; gr = 0.0f; gz = 0.0f; gn = 0.0f; gnh = 0.0f; }

    __device__ __forceinline__ void bwd_chunk(int c, int t0, int len, const float2* xg) {
        {
            const float* kk = ck + c * 7 * 64 + lane;
            h = kk[0]; hp = kk[64]; xp = kk[128]; dmr = kk[192]; dmz = kk[256]; dmn = kk[320]; dmnh = kk[384];
        }
        wave_lds_fence();
        {
            int sp = svp0, hq_ = hp0, fqp = fq0;
            for (int tt = 0; tt < len; ++tt) { step<true>(t0 + tt, sp, hq_, fqp); sp += svp_step; hq_ += hp_step; fqp += fq_step; }
        }
        wave_lds_fence();
        if (lane < len) {      // TCN skip gradients (float), lane = time step
            const int t = t0 + lane;
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
            const float2 zero = make_float2(0.0f, 0.0f);
            const float2 xc = xg[t], xm = t - kHalo >= 0 ? xg[t - kHalo] : zero, xq = t + kHalo < T ? xg[t + kHalo] : zero;
            float s1[3], s2[2];
            tcn(xm, xc, xq, s1, s2);
            const float d2[2] = {dyv.x * q16_hsg_(s2[0]), dyv.y * q16_hsg_(s2[1])};
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const float hs = hardswishf_(s1[cc]);
                tw2[cc] = __builtin_fmaf(d2[0], hs, tw2[cc]);
                tw2[3 + cc] = __builtin_fmaf(d2[1], hs, tw2[3 + cc]);
                const float d1 = __builtin_fmaf(d2[0], w2[cc], d2[1] * w2[3 + cc]) * q16_hsg_(s1[cc]);
                tw1[cc * 6 + 0] = __builtin_fmaf(d1, xm.x, tw1[cc * 6 + 0]); tw1[cc * 6 + 1] = __builtin_fmaf(d1, xc.x, tw1[cc * 6 + 1]);
                tw1[cc * 6 + 2] = __builtin_fmaf(d1, xq.x, tw1[cc * 6 + 2]); tw1[cc * 6 + 3] = __builtin_fmaf(d1, xm.y, tw1[cc * 6 + 3]);
                tw1[cc * 6 + 4] = __builtin_fmaf(d1, xc.y, tw1[cc * 6 + 4]); tw1[cc * 6 + 5] = __builtin_fmaf(d1, xq.y, tw1[cc * 6 + 5]);
            }
        }
        for (int tt = len - 1; tt >= 0; --tt) {
            const float* s = sv + tt * 16 * NSV + col;
            const float hp_ = s[0], qdhk = s[16], npo = s[2 * 16], omz = s[3 * 16], z = s[4 * 16], c2 = s[5 * 16], c3 = s[6 * 16], An = s[7 * 16],
                        Az = s[8 * 16], B1 = s[9 * 16], B2A = s[10 * 16], mh = s[11 * 16], pph = s[12 * 16], hok = s[13 * 16], pho = s[14 * 16];
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * (t0 + tt));
            dwo0 = __builtin_fmaf(dyv.x, hok, dwo0); dwo1 = __builtin_fmaf(dyv.y, hok, dwo1);
            const float g = gh + (dyv.x * wo0 + dyv.y * wo1) * pho;
            const float g2 = g * c2, g3 = g * c3;
            const float dz = g2 * hp_ - g3 * npo;
            const float dan = g3 * omz * An;
            const float ghprev = g2 * z;
            gn += dan; gnh += dan * B1; gr += dan * B2A; gz += dz * Az;
            const float d_h = vsel(rm.m[0], gr, vsel(rm.m[1], gz, vsel(rm.m[2], gnh, 0.0f)));
            const float d_x = vsel(rm.m[0], gr, vsel(rm.m[1], gz, vsel(rm.m[2], gn, 0.0f)));
            float ddh = rotdot(0.0f, wT, d_h);
            ddh = sum_rows4(ddh);
            const float g2b = ddh * pph;
            gh = ghprev + mh * (g2b + ghp);
            ghp = (1.0f - mh) * ghp - mh * g2b;
            const float fsx = col < 6 ? fqs[tt * 8 + col] : 0.0f;
            acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_h, qdhk, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_x, fsx, acc2, 0, 0, 0);
        }
        wave_lds_fence();
    }

    __device__ __forceinline__ void write_partials(float* prow, float loss) {
        for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.0f;
        wave_lds_fence();
        __builtin_amdgcn_s_waitcnt(0);
        if (vo && role == 0) {
            prow[L.o_wo + col] = dwo0 * k.s_oa * qpass(pl[L.o_wo + col], wq.o);
            prow[L.o_wo + L.OW + col] = dwo1 * k.s_oa * qpass(pl[L.o_wo + L.OW + col], wq.o);
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) { const float v = wsum(tw1[i]); if (lane == 0) prow[L.o_tcn0 + i] = v; }
#pragma unroll
        for (int i = 0; i < 6; ++i) { const float v = wsum(tw2[i]); if (lane == 0) prow[L.o_tcn2 + i] = v; }
        if (lane == 0) prow[L.P] = loss;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 4 * role + rr;
                if (i < H) {
                    if (col < H) { const int j = L.o_wh + (g * H + i) * H + col; prow[j] = acc1[4 * g + rr] * k.s_ha * qpass(pl[j], wq.h); }
                    if (col < F) { const int j = L.o_wx + (g * H + i) * F + col; prow[j] = acc2[4 * g + rr] * k.s_xa * qpass(pl[j], wq.x); }
                }
            }
    }
    __device__ __forceinline__ static float wsum(float v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    }
    __device__ __forceinline__ static float q16_hsg_(float v) { return v < -3.0f ? 0.0f : (v <= 3.0f ? v * (1.0f / 3.0f) + 0.5f : 1.0f); }
    // sparsity counters of the forward passes: dx zeros = the six feature lanes of row 0, dh zeros = the hidden units of row 0
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

}  // namespace q16
}  // namespace odpd
