// odpd_gpseq.h — one GRU-family model on one wave (gate-parallel, one sequence per wave): the step arithmetic of gru_gp_train_kernel
// (gru_family.hip) as a struct of register-resident state + an LDS region, with chunk-wise methods, for the cascade kernels (gru_cascade.hip)
// whose waves run different models side by side.
#pragma once
#include "odpd_gru.h"

namespace odpd {

constexpr int kCascChunk = 32;
constexpr int kDxPitch = kCascChunk + 1;       // odd: the 16 lanes of a row hit 16 banks

// One model of the cascade on one wave.  TRAIN: weight gradients (the DPD); else frozen, dL/dx out (the PA).
// HALF (frozen, NB = 2, hidden 17..24 — the reference's default PA has 23 units): the second block holds its <= 8 units twice, so the
// rotated dot products over it take 8 rotations instead of 16 (odpd_gru.h, fill_gru_tabs<.., HALF>): 48 of the 192 DPP FMAs of a time step.
template <int NB, int FM, bool DG, bool TRAIN, bool HALF = false>
struct GpSeq {
    static_assert(!HALF || (NB == 2 && !TRAIN), "half-block layout: frozen two-block models");
    static constexpr int F = FeatDim<FM>::F, HB = 16 * NB;
    using TB = GruTabs<NB, DG>;
    // ---- LDS region: parameters | max(weight tables, per-time buffers) ----
    __host__ __device__ static int tp(int T) { return (T + 63) & ~63; }
    __host__ __device__ static int off_hist(int T) { return tp(T) * 8; }
    __host__ __device__ static int off_actb(int T) { return off_hist(T) + (T + 2) * HB; }
    __host__ __device__ static int off_dyb(int T) { return off_actb(T) + (DG ? T * HB : 0); }
    __host__ __device__ static int off_dump(int T) { return off_dyb(T) + tp(T) * 2; }
    __host__ __device__ static int off_hw(int T) { return off_dump(T) + 256 * NB; }
    __host__ __device__ static int off_dpk(int T) { return off_hw(T) + 2 * HB + 16; }
    __host__ __device__ static int off_ubuf(int T) { return off_dpk(T) + (TRAIN ? 0 : 3 * HB * kDxPitch); }
    __host__ __device__ static int buf_floats(int T) { return off_ubuf(T) + (TRAIN ? 0 : tp(T) * 2); }
    __host__ __device__ static int region_floats(int T, int P) {
        const int buf = buf_floats(T), tabf = TB::kFloats;
        return pad4(P) + (buf > tabf ? buf : tabf);
    }

    // ---- registers ----
    float wF[NB][NB][16], wT[NB][NB][16];          // the row's rotated weights, forward and transposed: [output block][input block]
    float win[NB][F], b_in[NB], b_rec[NB], wo0[NB], wo1[NB], bo0, bo1;
    f32x16 acc1[NB][NB], acc2[NB];
    float dmisc[NB], dwo0[NB], dwo1[NB], dwf0, dwf1, dbo0, dbo1;
    float h[NB], carry[NB], dhid_cur[NB];
    float *smem, *pl, *ftab, *hist, *actb, *dyb, *dump, *hw, *dpk;
    GruLayout L;
    RowMasks rm;
    int H, OW, T, lane, col, role, park0, park_step, park, dpark0, dpark_gate, dpark_ob, dpark_t;
    bool head_row, odd;

    // the unit a lane of output block ob carries
    __device__ __forceinline__ int unit(int ob) const { return 16 * ob + ((HALF && ob == 1) ? (col & 7) : col); }
    // rotated dot product over input block kb
    __device__ __forceinline__ static float rd(float acc, const float (&w)[16], float v, int kb) {
        return (HALF && kb == 1) ? rotdot8(acc, w, v) : rotdot(acc, w, v);
    }
    // forward dot products.  (gru_family.hip's solo kernels add a dgru's products as ONE chain since r05; here the two chains stay: the W16A16
    // QAT stage of the OpenDPDv2 recipe — thresholds + 2^-14 grids — turns that rounding-level change of its frozen PA into a first-epoch TRAIN_LOSS of
    // 0.00235 instead of 0.00212 (reference log: 0.00217), and the logged trajectories are the anchors: tests/test_e2e_gpu.py)
    __device__ __forceinline__ static float rd1(float acc, const float (&w)[16], float v, int kb) { return rd(acc, w, v, kb); }
    // (one workgroup barrier inside: fill_gru_tabs)
    __device__ __forceinline__ void setup(float* base, float* region, const float* params, int Hm, int T_) {
        smem = base;
        lane = threadIdx.x & 63; col = lane & 15; role = lane >> 4;      // 0 r | 1 n | 2 head | 3 z
        L = gru_layout(Hm, F, DG);
        H = L.H; OW = DG ? H + 6 : H; T = T_;
        pl = region;
        for (int i = lane; i < L.P; i += 64) pl[i] = params[i];
        wave_lds_fence();
        float* tab = region + pad4(L.P);
        fill_gru_tabs<NB, DG, true, HALF>(tab, pl, L, lane, 0, 1);
        const int gate = role == 0 ? 0 : role == 3 ? 1 : 2;
        head_row = role == 2; odd = role & 1;
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const int o = unit(ob);
            const bool vo = o < H;
            TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + 16 * ob + col);
            int rf = TB::kHH + gate * NB, rt = TB::kHHT + gate * NB;
            if constexpr (DG) { if (head_row) { rf = TB::kHID; rt = TB::kHIDT; } }
#pragma unroll
            for (int rb = 0; rb < NB; ++rb) {
                const int kb = (ob + rb) % NB;
                load_rot(wF[ob][kb], tl + (rf + rb) * 4 * 64);
                load_rot(wT[ob][kb], tl + (rt + rb) * 4 * 64);
                if (!DG && head_row) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { wF[ob][kb][k] = 0.0f; wT[ob][kb][k] = 0.0f; }
                }
            }
#pragma unroll
            for (int i = 0; i < F; ++i) win[ob][i] = (vo && !head_row) ? pl[L.o_w_ih + (gate * H + o) * F + i] : 0.0f;
            b_in[ob] = 0.0f; b_rec[ob] = 0.0f;
            if (vo) {
                if (role == 0 || role == 3) b_in[ob] = pl[L.o_b_ih + gate * H + o] + pl[L.o_b_hh + gate * H + o];
                if (role == 1) { b_in[ob] = pl[L.o_b_ih + 2 * H + o]; b_rec[ob] = pl[L.o_b_hh + 2 * H + o]; }
                if (head_row && DG) b_rec[ob] = pl[L.o_b_hid + o];
            }
            wo0[ob] = vo ? pl[L.o_w_out + o] : 0.0f; wo1[ob] = vo ? pl[L.o_w_out + OW + o] : 0.0f;
        }
        bo0 = pl[L.o_b_out]; bo1 = pl[L.o_b_out + 1];
        wave_lds_fence();
        // per-time buffers over the tables
        ftab = tab;                                  // [Tp][8]   features of step t
        hist = tab + off_hist(T);                    // [T + 2][HB]   entry t + 1 = h(t), entry 0 = h(-1) = 0
        actb = tab + off_actb(T);                    // DGRU: [T][HB]   relu(fc_hid h(t) + b)
        dyb = tab + off_dyb(T);                      // [Tp][2]   dL/dy(t)
        dump = tab + off_dump(T);
        hw = tab + off_hw(T);                        // fc_out: [2][HB] hidden columns (zero padded) | [2][8] feature columns
        dpk = tab + off_dpk(T);                      // frozen: [3][HB][kDxPitch]   d_r, d_z, d_n of unit u at step t0 + i
        for (int i = lane; i < 2 * HB + 16; i += 64) {
            float v = 0.0f;
            if (i < 2 * HB) { const int c = i / HB, u = i % HB; if (u < H) v = pl[L.o_w_out + c * OW + u]; }
            else { const int j = i - 2 * HB, c = j >> 3, k = j & 7; if (DG && k < 6) v = pl[L.o_w_out + c * OW + H + k]; }
            hw[i] = v;
        }
        if (lane < HB) hist[lane] = 0.0f;
        rm = row_masks();
        // the per-step stores of the forward pass: row 1 parks h(t), the head row parks relu(fc_hid h(t-1)), rows 0 / 3 hit the dump
        park0 = role == 1 ? (int)(hist - smem) + HB + col : (head_row && DG) ? (int)(actb - smem) - HB + col : (int)(dump - smem) + lane;
        park_step = (role == 1 || (head_row && DG)) ? HB : 0;
        // frozen, backward: row 1 parks (d_r, d_z, d_n) of its unit, the other rows hit the dump
        dpark0 = role == 1 ? (int)(dpk - smem) + col * kDxPitch : (int)(dump - smem) + lane;
        dpark_gate = role == 1 ? HB * kDxPitch : 0; dpark_ob = role == 1 ? 16 * kDxPitch : 0; dpark_t = role == 1 ? 1 : 0;
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc2[ob][i] = 0.0f;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) acc1[ob][kb][i] = 0.0f;
            }
            dmisc[ob] = 0.0f; dwo0[ob] = 0.0f; dwo1[ob] = 0.0f;
        }
        dwf0 = 0.0f; dwf1 = 0.0f; dbo0 = 0.0f; dbo1 = 0.0f;
        wave_lds_fence();
    }

    // the gates of one step from h(t-1): arec = W h + b of the row's gate, rows 1 and 3 end with (z, n), row 1 also with r
    __device__ __forceinline__ void gates(const float (&f)[F], const float (&hin)[NB], float (&arec)[NB], float (&r1)[NB], float (&zz)[NB],
                                          float (&nn)[NB]) const {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            arec[ob] = b_rec[ob];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) arec[ob] = rd1(arec[ob], wF[ob][kb], hin[kb], kb);
        }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            float ain = b_in[ob];
#pragma unroll
            for (int i = 0; i < F; ++i) ain = __builtin_fmaf(win[ob][i], f[i], ain);
            const float sg = sigmoidf_(ain + arec[ob]);                     // r (row 0), z (row 3)
            const RowDup s2 = dup16(sg);                                    // .even on row 1: r, .odd on row 3: z
            r1[ob] = s2.even;
            const float n = tanhf_(__builtin_fmaf(r1[ob], arec[ob], ain));  // row 1
            const HalfDup nz = dup32(role == 1 ? n : s2.odd);               // rows 1 and 3: lo = n, hi = z
            zz[ob] = nz.hi; nn[ob] = nz.lo;
        }
    }
    __device__ __forceinline__ void load_feat(int t, float (&f)[F]) const {
        const float4* ftab4 = reinterpret_cast<const float4*>(ftab);
        const float4 fa = ftab4[2 * t];
        f[0] = fa.x; f[1] = fa.y;
        if constexpr (F > 2) { f[2] = fa.z; f[3] = fa.w; }
        if constexpr (F > 4) { const float4 fb = ftab4[2 * t + 1]; f[4] = fb.x; f[5] = fb.y; }
    }
    // features of sample t (this model's input) into a feature table, lane = time step
    __device__ __forceinline__ static void write_feat(float* ftab_, int t, float I, float Q) {
        float f[F], f8[8];
        feat_fwd<FM>(I, Q, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) f8[i] = i < F ? f[i] : 0.0f;
        reinterpret_cast<float4*>(ftab_)[2 * t] = make_float4(f8[0], f8[1], f8[2], f8[3]);
        reinterpret_cast<float4*>(ftab_)[2 * t + 1] = make_float4(f8[4], f8[5], f8[6], f8[7]);
    }

    __device__ __forceinline__ void fwd_begin() {
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) h[kb] = 0.0f;
        park = park0;
    }
    // steps t0 .. t0 + len - 1 of the recurrence (features in ftab)
    __device__ __forceinline__ void fwd_steps(int t0, int len) {
        for (int tt = 0; tt < len; ++tt) {
            float f[F], arec[NB], r1[NB], zz[NB], nn[NB];
            load_feat(t0 + tt, f);
            gates(f, h, arec, r1, zz, nn);
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                const float h13 = __builtin_fmaf(zz[ob], h[ob] - nn[ob], nn[ob]);       // rows 1 and 3: (1 - z) n + z h
                h[ob] = dup16(h13).odd;                                                 // every row: its half's row 1 / 3
                // (the head row's first store, act(-1), lands in hist's pad entry)
                smem[park + 16 * ob] = head_row ? __builtin_fmaxf(arec[ob], 0.0f) : h[ob];
            }
            park += park_step;
        }
    }
    // DGRU: relu(fc_hid h(t)) of the step just done (the head row runs one step behind)
    __device__ __forceinline__ void store_act(int t) {
        if constexpr (DG) {
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                float arec = b_rec[ob];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) arec = rd1(arec, wF[ob][kb], h[kb], kb);
                if (head_row) actb[t * HB + 16 * ob + col] = __builtin_fmaxf(arec, 0.0f);
            }
        }
    }
    // fc_out of steps t0 .. t0 + len - 1, lane = time step: sink(t, y0, y1)
    template <typename Sink>
    __device__ __forceinline__ void head_chunk(int t0, int len, Sink sink) const {
        const float4* ftab4 = reinterpret_cast<const float4*>(ftab);
        const float4* hw4 = reinterpret_cast<const float4*>(hw);
        const int t = t0 + lane;
        if (lane < len) {
            const float4* hv4 = reinterpret_cast<const float4*>(DG ? actb + t * HB : hist + (t + 1) * HB);
            float y0 = bo0, y1 = bo1;
#pragma unroll
            for (int q = 0; q < 4 * NB; ++q) {
                const float4 hv = hv4[q], w0 = hw4[q], w1 = hw4[4 * NB + q];
                y0 = __builtin_fmaf(w0.x, hv.x, y0); y0 = __builtin_fmaf(w0.y, hv.y, y0); y0 = __builtin_fmaf(w0.z, hv.z, y0); y0 = __builtin_fmaf(w0.w, hv.w, y0);
                y1 = __builtin_fmaf(w1.x, hv.x, y1); y1 = __builtin_fmaf(w1.y, hv.y, y1); y1 = __builtin_fmaf(w1.z, hv.z, y1); y1 = __builtin_fmaf(w1.w, hv.w, y1);
            }
            if constexpr (DG) {
                const float4 fa = ftab4[2 * t], fb = ftab4[2 * t + 1];
                const float4 u0 = hw4[8 * NB], u1 = hw4[8 * NB + 1], v0 = hw4[8 * NB + 2], v1 = hw4[8 * NB + 3];
                y0 = __builtin_fmaf(u0.x, fa.x, y0); y0 = __builtin_fmaf(u0.y, fa.y, y0); y0 = __builtin_fmaf(u0.z, fa.z, y0); y0 = __builtin_fmaf(u0.w, fa.w, y0);
                y0 = __builtin_fmaf(u1.x, fb.x, y0); y0 = __builtin_fmaf(u1.y, fb.y, y0);
                y1 = __builtin_fmaf(v0.x, fa.x, y1); y1 = __builtin_fmaf(v0.y, fa.y, y1); y1 = __builtin_fmaf(v0.z, fa.z, y1); y1 = __builtin_fmaf(v0.w, fa.w, y1);
                y1 = __builtin_fmaf(v1.x, fb.x, y1); y1 = __builtin_fmaf(v1.y, fb.y, y1);
            }
            sink(t, y0, y1);
        }
    }

    __device__ __forceinline__ void bwd_begin() {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) { carry[ob] = 0.0f; dhid_cur[ob] = 0.0f; }
        if constexpr (DG) {
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * (T - 1));
#pragma unroll
            for (int ob = 0; ob < NB; ++ob)
                dhid_cur[ob] = __builtin_fmaf(dyv.x, wo0[ob], dyv.y * wo1[ob]) * relu_gate(actb[(T - 1) * HB + 16 * ob + col]);
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                float part = 0.0f;
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) part = rd(part, wT[ob][kb], head_row ? dhid_cur[kb] : 0.0f, kb);
                part = sum_rows4(part);
                carry[ob] = part;
            }
        }
    }
    // backward steps hi .. lo (descending); frozen: the pre-activation gradients of step t are parked at column t - tbase
    __device__ __forceinline__ void bwd_steps(int hi, int lo, int tbase) {
        for (int t = hi; t >= lo; --t) {
            float hp[NB], ht[NB], at[NB];
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                hp[ob] = hist[t * HB + 16 * ob + col]; ht[ob] = hist[(t + 1) * HB + 16 * ob + col];
                at[ob] = DG ? actb[t * HB + 16 * ob + col] : ht[ob];
            }
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
            const float fsx = col < F ? ftab[t * 8 + col] : (col == F ? 1.0f : 0.0f);
            // the gates of step t again, from the parked h(t-1) (off the carry chain: it fills the chain's stalls)
            float arec[NB], r1[NB], zz[NB], nn[NB];                              // rows 1 and 3: z, n; row 1: r, W_hn h + b_hn
            {
                float f[F];
                load_feat(t, f);
                gates(f, hp, arec, r1, zz, nn);
            }
            // dL/dh(t) and the pre-activation gradients (rows 1 / 3; row 0 receives d_r from row 1)
            float d_row[NB], dnp[NB], dhid_prev[NB], zterm[NB];
            float2 dyp = make_float2(0.0f, 0.0f);
            const int tm = t > 0 ? t - 1 : 0;
            if constexpr (DG) dyp = *reinterpret_cast<const float2*>(dyb + 2 * tm);
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                const float g01 = __builtin_fmaf(dyv.x, wo0[ob], dyv.y * wo1[ob]);
                const float dht = DG ? carry[ob] : carry[ob] + g01;
                const float dn = dht * (1.0f - zz[ob]), dz = dht * (hp[ob] - nn[ob]);
                dnp[ob] = dn * __builtin_fmaf(-nn[ob], nn[ob], 1.0f);
                const float dgh = dnp[ob] * r1[ob];
                const float drp1 = (dnp[ob] * arec[ob]) * (r1[ob] * (1.0f - r1[ob]));
                const float dzp = dz * (zz[ob] * (1.0f - zz[ob]));
                const float drp0 = dup16(drp1).odd;                              // (row 0 <- row 1)
                dhid_prev[ob] = 0.0f;
                if constexpr (DG) {
                    const float atp = actb[tm * HB + 16 * ob + col];
                    dhid_prev[ob] = t > 0 ? __builtin_fmaf(dyp.x, wo0[ob], dyp.y * wo1[ob]) * relu_gate(atp) : 0.0f;
                }
                d_row[ob] = vsel(rm.m[0], drp0, vsel(rm.m[1], dgh, vsel(rm.m[3], dzp, dhid_prev[ob])));
                zterm[ob] = vsel(rm.m[3], dht * zz[ob], 0.0f);
                if constexpr (TRAIN) {
                    dmisc[ob] += vsel(rm.m[1], dgh, dhid_cur[ob]);               // row 1: db_hn, head row: db_hid
                    dwo0[ob] = __builtin_fmaf(dyv.x, at[ob], dwo0[ob]); dwo1[ob] = __builtin_fmaf(dyv.y, at[ob], dwo1[ob]);
                } else {
                    const int dp = dpark0 + ob * dpark_ob + (t - tbase) * dpark_t;
                    smem[dp] = drp1; smem[dp + dpark_gate] = dzp; smem[dp + 2 * dpark_gate] = dnp[ob];
                }
            }
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                float part = zterm[ob];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) part = rd(part, wT[ob][kb], d_row[kb], kb);
                part = sum_rows4(part);
                carry[ob] = part;                                                // dL/dh(t-1): W_hh^T d + z dL/dh(t) + fc_hid^T dhid(t-1)
            }
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                if constexpr (TRAIN) {
                    // weight gradients: 4-block MFMAs, block k = the outer product of row k's operands
                    const float a1 = vsel(rm.m[2], dhid_cur[ob], d_row[ob]);
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb)
                        acc1[ob][kb] = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, vsel(rm.m[2], ht[kb], hp[kb]), acc1[ob][kb], 0, 0, 0);
                    acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x1f32(vsel(rm.m[1], dnp[ob], vsel(rm.m[2], 0.0f, d_row[ob])), fsx, acc2[ob], 0, 0, 0);
                }
                dhid_cur[ob] = dhid_prev[ob];
            }
            if constexpr (TRAIN) {
                if constexpr (DG) {
                    const float fs = col < 6 ? fsx : 0.0f;
                    dwf0 = __builtin_fmaf(dyv.x, fs, dwf0); dwf1 = __builtin_fmaf(dyv.y, fs, dwf1);
                }
                dbo0 += dyv.x; dbo1 += dyv.y;
            }
        }
    }
    // frozen: dL/dx of steps t0 .. t0 + len - 1, lane = time step: W_ih^T (d_r | d_z | d_n)(t) [+ the fc_out feature columns x dL/dy(t)],
    // then the feature Jacobian at x(t) = usrc[t]
    __device__ __forceinline__ void dx_chunk(int t0, int len, const float2* usrc, float2* dst) const {
        if (lane < len) {
            const int t = t0 + lane;
            float df[F];
#pragma unroll
            for (int i = 0; i < F; ++i) df[i] = 0.0f;
            for (int u = 0; u < H; ++u) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float d = dpk[(g * HB + u) * kDxPitch + lane];
                    const float* wr = pl + L.o_w_ih + (g * H + u) * F;
#pragma unroll
                    for (int i = 0; i < F; ++i) df[i] = __builtin_fmaf(wr[i], d, df[i]);
                }
            }
            if constexpr (DG) {
                const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
#pragma unroll
                for (int i = 0; i < 6; ++i) df[i] += __builtin_fmaf(dyv.x, hw[2 * HB + i], dyv.y * hw[2 * HB + 8 + i]);
            }
            const float2 xv = usrc[t];
            float dI, dQ;
            feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
            dst[t] = make_float2(dI, dQ);
        }
    }
    // the workgroup's row of partial gradients (every entry written)
    __device__ __forceinline__ void write_partials(float* prow, float loss) const {
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const int o = 16 * ob + col;
            if (o < H) {
                if (role == 1) prow[L.o_b_hh + 2 * H + o] = dmisc[ob];
                if (DG && head_row) prow[L.o_b_hid + o] = dmisc[ob];
                if (role == 0) { prow[L.o_w_out + o] = dwo0[ob]; prow[L.o_w_out + OW + o] = dwo1[ob]; }
            }
        }
        if (DG && role == 0 && col < 6) { prow[L.o_w_out + H + col] = dwf0; prow[L.o_w_out + OW + H + col] = dwf1; }
        if (lane == 0) {
            prow[L.o_b_out] = dbo0; prow[L.o_b_out + 1] = dbo1;
            prow[L.P] = loss; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
        }
        // MFMA blocks: 0 = r, 1 = n, 2 = fc_hid, 3 = z; register 4 blk + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const int g = blk == 0 ? 0 : blk == 3 ? 1 : 2;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int i = 16 * ob + 4 * role + rr;
                    if (i < H) {
#pragma unroll
                        for (int kb = 0; kb < NB; ++kb) {
                            const int j = 16 * kb + col;
                            if (j < H) {
                                if (blk == 2) { if (DG) prow[L.o_w_hid + i * H + j] = acc1[ob][kb][4 * blk + rr]; }
                                else prow[L.o_w_hh + (g * H + i) * H + j] = acc1[ob][kb][4 * blk + rr];
                            }
                        }
                        if (blk != 2) {
                            const float v = acc2[ob][4 * blk + rr];
                            if (col < F) prow[L.o_w_ih + (g * H + i) * F + col] = v;
                            else if (col == F) {
                                prow[L.o_b_ih + g * H + i] = v;
                                if (g < 2) prow[L.o_b_hh + g * H + i] = v;
                            }
                        }
                    }
                }
            }
    }
};

// The frozen PA's wave of a cascade workgroup.  One workgroup barrier per hand-off (the DPD wave executes the same number): forward
// chunk k - 1 while the DPD wave produces chunk k; loss; backward chunk c + dL/du of the chunk into the DPD wave's dL/dy buffer.
template <typename P>
__device__ __forceinline__ void casc_pa_wave(const CascArgs& a, float* smem, float* rp, float2* pa_ubuf, float2* dpd_dyb, float* xch) {
    const int lane = threadIdx.x & 63, T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    P e;
    e.setup(smem, rp, a.pa_params, a.Hp, T);
    __syncthreads();
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    float2* pa_dyb = reinterpret_cast<float2*>(e.dyb);
    float loss_acc = 0.0f;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        e.fwd_begin();
        for (int k = 0; k <= NC; ++k) {
            if (k >= 1) {
                const int t0 = (k - 1) * kCascChunk;
                e.fwd_steps(t0, min(kCascChunk, T - t0));
            }
            __syncthreads();
        }
        e.store_act(T - 1);
        wave_lds_fence();
        for (int t0 = 0; t0 < T; t0 += kCascChunk)
            e.head_chunk(t0, min(kCascChunk, T - t0), [&](int t, float y0, float y1) {
                const float2 tv = tg[t];
                float dy0, dy1;
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
                pa_dyb[t] = make_float2(dy0, dy1);
            });
        wave_lds_fence();
        e.bwd_begin();
        for (int k = 0; k <= NC; ++k) {
            if (k < NC) {
                const int c = NC - 1 - k, t0 = c * kCascChunk, len = min(kCascChunk, T - t0);
                e.bwd_steps(t0 + len - 1, t0, t0);
                wave_lds_fence();
                e.dx_chunk(t0, len, pa_ubuf, dpd_dyb);
                wave_lds_fence();
            }
            __syncthreads();
        }
    }
    float lp = loss_acc;
    for (int o = 32; o > 0; o >>= 1) lp += __shfl_down(lp, o);
    if (lane == 0) xch[0] = lp;
    __syncthreads();
}

}  // namespace odpd
