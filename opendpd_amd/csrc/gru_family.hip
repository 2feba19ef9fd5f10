// gru_family.hip — persistent-RNN kernels for the nn.GRU based backbones of the reference:
//   gru        backbones/gru.py:4-48        y = fc_out(GRU(x))
//   dgru       backbones/dgru.py:9-74       feat=[I,Q,a,a^3,sin,cos]; y = fc_out(cat(relu(fc_hid(h)), feat))
//   qgru       backbones/qgru.py:9-71       feat=[I,Q,a^2,a^4]  (float path)
//   qgru_amp1  backbones/qgru_amp1.py:9-76  feat=[I,Q,a,a^3]    (float path)
// GRU cell = torch.nn.GRU semantics, gate order r,z,n:
//   r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r*(W_hn h + b_hn)),
//   h' = (1-z)*n + z*h.
//
// Data layout on chip (one wavefront = 4/R sequences, lane mapping in odpd_device.h):
//   * the ~1-3 k parameters are staged once per workgroup into LDS, together with "rotated quad
//     tables" of the recurrent matrices (W_hh, W_hh^T, fc_hid, fc_hid^T): entry [row][quad][lane]
//     holds the 4 weights lane needs for rotations 4q..4q+3.  Kernels pull the rows they need for
//     the current PHASE into registers with ds_read_b128 (W_hh for forward/recompute, W_hh^T for the
//     backward steps) — the two orientations time-share the same registers;
//   * x / target / dy are staged per chunk of kChunk steps; BPTT state is one checkpoint of h every
//     kCkptStride steps (HBM for the split kernels, LDS for the fused one).
// Kernels:
//   gru_fwd_kernel    forward (inference or training forward with checkpoints)
//   gru_bwd_kernel    BPTT: per block of S steps recompute forward into registers, back-propagate;
//                     weight gradients are rank-4 exact-fp32 MFMA updates (v_mfma_f32_16x16x4_f32);
//                     one row of partial gradients per workgroup, fixed summation order.
//   gru_train_kernel  forward (cell only) + backward with y / loss / dL/dy formed on the fly:
//                     HBM traffic = x + target.
// One-sequence-per-wave ("gate-parallel") kernels for batches whose sequences each get a SIMD of their own (the reference's 64 .. 256
// frames, the evaluation segments): the four 16-lane rows of a wave take one gate each, only the recurrence stays in the step loop:
//   gru_eval_kernel      forward (inference; CK: also the checkpoint-writing forward of the split train path)
//   gru_gp_train_kernel  fused train step with the frame's BPTT state in LDS and the weight gradients as 4-block MFMAs
#include "odpd_gru.h"

namespace odpd {

// -------------------------------------------------------------------------------------------------
// register-resident small weights
// -------------------------------------------------------------------------------------------------
template <int R, int F, bool DG>
struct GruW {
    float wih[3][F];
    float b_r, b_z, b_in, b_hn;
    float wout[2], bout[2];
    float bhid;
    float woutf[2];        // DG: fc_out weight of feature `col` (row-0 lanes, col < 6), else 0
};

template <int R, int F, bool DG>
__device__ __forceinline__ void load_gru_w(GruW<R, F, DG>& w, const float* pl, const GruLayout& L, int row, int col) {
    const int H = L.H, o = 16 * row + col;
    const bool vo = o < H;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < F; ++i) w.wih[g][i] = vo ? pl[L.o_w_ih + (g * H + o) * F + i] : 0.0f;
    w.b_r = vo ? pl[L.o_b_ih + o] + pl[L.o_b_hh + o] : 0.0f;
    w.b_z = vo ? pl[L.o_b_ih + H + o] + pl[L.o_b_hh + H + o] : 0.0f;
    w.b_in = vo ? pl[L.o_b_ih + 2 * H + o] : 0.0f;
    w.b_hn = vo ? pl[L.o_b_hh + 2 * H + o] : 0.0f;
    const int OW = DG ? H + 6 : H;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        w.wout[c] = vo ? pl[L.o_w_out + c * OW + o] : 0.0f;
        // uniform across lanes: keep it in an SGPR
        w.bout[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pl[L.o_b_out + c])));
        w.woutf[c] = (DG && row == 0 && col < 6) ? pl[L.o_w_out + c * OW + H + col] : 0.0f;
    }
    w.bhid = (DG && vo) ? pl[L.o_b_hid + o] : 0.0f;
}

// -------------------------------------------------------------------------------------------------
// per-step device functions
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG>
__device__ __forceinline__ void gru_cell_fwd(const GruW<R, FeatDim<FM>::F, DG>& w, const float (&whh)[3][R][16],
                                             const float (&f)[FeatDim<FM>::F], float& h, float& r, float& z, float& n,
                                             float& ghn) {
    constexpr int F = FeatDim<FM>::F;
    float ar = w.b_r, az = w.b_z, an = w.b_in, ah = w.b_hn;
#pragma unroll
    for (int i = 0; i < F; ++i) {
        ar = __builtin_fmaf(w.wih[0][i], f[i], ar);
        az = __builtin_fmaf(w.wih[1][i], f[i], az);
        an = __builtin_fmaf(w.wih[2][i], f[i], an);
    }
    rotdot3(ar, az, ah, whh[0][0], whh[1][0], whh[2][0], h);
    if constexpr (R == 2) {
        float hx = swap16(h);
        rotdot3(ar, az, ah, whh[0][1], whh[1][1], whh[2][1], hx);
    }
    r = sigmoidf_(ar);
    z = sigmoidf_(az);
    ghn = ah;
    n = tanhf_(__builtin_fmaf(r, ah, an));
    h = __builtin_fmaf(z, h - n, n);  // (1-z)*n + z*h
}

// y from the head inputs: DG: act = relu(fc_hid pre-activation), else act = h
template <int R, int FM, bool DG>
__device__ __forceinline__ void gru_head_out(const GruW<R, FeatDim<FM>::F, DG>& w, float act,
                                             const float (&f)[FeatDim<FM>::F], int col, float& y0, float& y1) {
    float p0, p1;
    if constexpr (DG) {
        const float fs = feat_select<6>(f, col, 0.0f);
        p0 = __builtin_fmaf(w.wout[0], act, w.woutf[0] * fs);
        p1 = __builtin_fmaf(w.wout[1], act, w.woutf[1] * fs);
    } else {
        p0 = w.wout[0] * act;
        p1 = w.wout[1] * act;
    }
    y0 = seq_sum<R>(p0) + w.bout[0];
    y1 = seq_sum<R>(p1) + w.bout[1];
}

// gradient accumulators of one wavefront
template <int R, bool DG>
struct GruGrad {
    f32x4 thh[3][R][R];  // dW_hh tiles   [gate][out rowblk][in rowblk]
    f32x4 tih[3][R];     // dW_ih | db_i  [gate][out rowblk]  (col F carries the bias gradient)
    f32x4 thid[DG ? R : 1][DG ? R : 1];
    float db_hn, db_hid, dwout[2], dwoutf[2], dbout[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int a = 0; a < R; ++a) {
                tih[g][a] = z4;
#pragma unroll
                for (int b = 0; b < R; ++b) thh[g][a][b] = z4;
            }
#pragma unroll
        for (int a = 0; a < (DG ? R : 1); ++a)
#pragma unroll
            for (int b = 0; b < (DG ? R : 1); ++b) thid[a][b] = z4;
        db_hn = db_hid = 0.f;
        dwout[0] = dwout[1] = dwoutf[0] = dwoutf[1] = dbout[0] = dbout[1] = 0.f;
    }
};

// one BPTT step.  In: saved hp,r,z,n,ghn,hid of the step, features f, dy, carry dh (dL/dh_t from
// later steps).  Out: dh <- dL/dh_{t-1}; df (if DX) = dL/dfeat.
template <int R, int FM, bool DG, bool NW, bool DX>
__device__ __forceinline__ void gru_step_bwd(const GruW<R, FeatDim<FM>::F, DG>& w, const float (&whhT)[3][R][16],
                                             TabPtr tlane, GruGrad<R, DG>& G, const float (&f)[FeatDim<FM>::F],
                                             float hp, float r, float z, float n, float ghn, float hid, float dy0,
                                             float dy1, int row, int col, float& dh, float (&df)[FeatDim<FM>::F]) {
    constexpr int F = FeatDim<FM>::F;
    using T = GruTabs<R, DG>;
    const float ht = __builtin_fmaf(z, hp - n, n);
    float dht = dh;
    const float g01 = __builtin_fmaf(dy0, w.wout[0], dy1 * w.wout[1]);
    if constexpr (DG) {
        const float a = __builtin_fmaxf(hid, 0.0f);
        const float dhid = g01 * relu_gate(hid);
        if constexpr (NW) {
            const float fs = feat_select<6>(f, col, 0.0f);
            G.dwout[0] = __builtin_fmaf(dy0, a, G.dwout[0]);
            G.dwout[1] = __builtin_fmaf(dy1, a, G.dwout[1]);
            G.dwoutf[0] = __builtin_fmaf(dy0, fs, G.dwoutf[0]);
            G.dwoutf[1] = __builtin_fmaf(dy1, fs, G.dwoutf[1]);
            G.db_hid += dhid;
            if constexpr (R == 1) {
                G.thid[0][0] = mfma4(dhid, ht, G.thid[0][0]);
            } else {
                const float htx = swap16(ht);
#pragma unroll
                for (int qo = 0; qo < R; ++qo) {
                    const float am = (row == qo) ? dhid : 0.0f;
#pragma unroll
                    for (int qm = 0; qm < R; ++qm) G.thid[qo][qm] = mfma4(am, qm == qo ? ht : htx, G.thid[qo][qm]);
                }
            }
        }
        dht = tab_rotdot<R>(dht, tlane, T::kHIDT, dhid);  // dht += fc_hid^T dhid
    } else {
        if constexpr (NW) {
            G.dwout[0] = __builtin_fmaf(dy0, ht, G.dwout[0]);
            G.dwout[1] = __builtin_fmaf(dy1, ht, G.dwout[1]);
        }
        dht += g01;
    }
    if constexpr (NW) { G.dbout[0] += dy0; G.dbout[1] += dy1; }
    // cell
    const float dn = dht * (1.0f - z);
    const float dz = dht * (hp - n);
    const float dnp = dn * __builtin_fmaf(-n, n, 1.0f);
    const float dgh = dnp * r;
    const float drp = (dnp * ghn) * (r * (1.0f - r));
    const float dzp = dz * (z * (1.0f - z));
    if constexpr (NW) {
        G.db_hn += dgh;
        const float fsx = feat_select<F>(f, col, 1.0f);
        if constexpr (R == 1) {
            G.tih[0][0] = mfma4(drp, fsx, G.tih[0][0]);
            G.tih[1][0] = mfma4(dzp, fsx, G.tih[1][0]);
            G.tih[2][0] = mfma4(dnp, fsx, G.tih[2][0]);
            G.thh[0][0][0] = mfma4(drp, hp, G.thh[0][0][0]);
            G.thh[1][0][0] = mfma4(dzp, hp, G.thh[1][0][0]);
            G.thh[2][0][0] = mfma4(dgh, hp, G.thh[2][0][0]);
        } else {
            const float hpx = swap16(hp);
#pragma unroll
            for (int qo = 0; qo < R; ++qo) {
                const bool mine = row == qo;
                const float ar = mine ? drp : 0.0f, az = mine ? dzp : 0.0f, an = mine ? dnp : 0.0f, ag = mine ? dgh : 0.0f;
                G.tih[0][qo] = mfma4(ar, fsx, G.tih[0][qo]);
                G.tih[1][qo] = mfma4(az, fsx, G.tih[1][qo]);
                G.tih[2][qo] = mfma4(an, fsx, G.tih[2][qo]);
#pragma unroll
                for (int qm = 0; qm < R; ++qm) {
                    const float bh = qm == qo ? hp : hpx;
                    G.thh[0][qo][qm] = mfma4(ar, bh, G.thh[0][qo][qm]);
                    G.thh[1][qo][qm] = mfma4(az, bh, G.thh[1][qo][qm]);
                    G.thh[2][qo][qm] = mfma4(ag, bh, G.thh[2][qo][qm]);
                }
            }
        }
    }
    // data gradient to h_{t-1}
    float d0 = dht * z, d1 = 0.0f, d2 = 0.0f;
    rotdot3x(d0, d1, d2, whhT[0][0], whhT[1][0], whhT[2][0], drp, dzp, dgh);
    if constexpr (R == 2) rotdot3x(d0, d1, d2, whhT[0][1], whhT[1][1], whhT[2][1], swap16(drp), swap16(dzp), swap16(dgh));
    dh = d0 + d1 + d2;
    if constexpr (DX) {
#pragma unroll
        for (int i = 0; i < F; ++i) {
            float p = __builtin_fmaf(w.wih[0][i], drp, __builtin_fmaf(w.wih[1][i], dzp, w.wih[2][i] * dnp));
            df[i] = seq_sum<R>(p);
        }
        if constexpr (DG) {
            // fc_out feature columns: lane (row 0, col i) holds woutf[c] for feature i
            const float q = __builtin_fmaf(dy0, w.woutf[0], dy1 * w.woutf[1]);
#pragma unroll
            for (int i = 0; i < F; ++i) df[i] += seq_sum<R>(col == i ? q : 0.0f);
        }
    }
}

// write one wavefront's row of partial gradients (every entry of the row is written)
template <int R, int F, bool DG>
__device__ __forceinline__ void gru_write_partials(float* prow, const GruLayout& L, GruGrad<R, DG>& G, int lane, int row,
                                                   int col, float loss_part) {
    const int H = L.H, o = 16 * row + col, OW = DG ? H + 6 : H;
    const int seq = lane / (16 * R);
    float db_hn = across_seqs<R>(G.db_hn), db_hid = across_seqs<R>(G.db_hid);
    float dw0 = across_seqs<R>(G.dwout[0]), dw1 = across_seqs<R>(G.dwout[1]);
    float df0 = across_seqs<R>(G.dwoutf[0]), df1 = across_seqs<R>(G.dwoutf[1]);
    float db0 = across_seqs<R>(G.dbout[0]), db1 = across_seqs<R>(G.dbout[1]);
    float lp = across_seqs<R>(loss_part);
    if (seq == 0) {
        if (o < H) {
            prow[L.o_b_hh + 2 * H + o] = db_hn;
            prow[L.o_w_out + o] = dw0;
            prow[L.o_w_out + OW + o] = dw1;
            if constexpr (DG) prow[L.o_b_hid + o] = db_hid;
        }
        if (DG && row == 0 && col < 6) {
            prow[L.o_w_out + H + col] = df0;
            prow[L.o_w_out + OW + H + col] = df1;
        }
        if (lane == 0) {
            prow[L.o_b_out] = db0;
            prow[L.o_b_out + 1] = db1;
            prow[L.P] = lp;
            prow[L.P + 1] = 0.f; prow[L.P + 2] = 0.f; prow[L.P + 3] = 0.f;
        }
    }
    const int g4 = lane >> 4, c = lane & 15;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int qo = 0; qo < R; ++qo)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 16 * qo + 4 * g4 + rr;
                if (i < H) {
                    const float v = G.tih[g][qo][rr];
                    if (c < F) prow[L.o_w_ih + (g * H + i) * F + c] = v;
                    else if (c == F) {
                        prow[L.o_b_ih + g * H + i] = v;
                        if (g < 2) prow[L.o_b_hh + g * H + i] = v;
                    }
#pragma unroll
                    for (int qm = 0; qm < R; ++qm) {
                        const int j = 16 * qm + c;
                        if (j < H) prow[L.o_w_hh + (g * H + i) * H + j] = G.thh[g][qo][qm][rr];
                    }
                }
            }
    if constexpr (DG) {
#pragma unroll
        for (int qo = 0; qo < R; ++qo)
#pragma unroll
            for (int qm = 0; qm < R; ++qm)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int i = 16 * qo + 4 * g4 + rr, j = 16 * qm + c;
                    if (i < H && j < H) prow[L.o_w_hid + i * H + j] = G.thid[qo][qm][rr];
                }
    }
}

// Block-level, fixed-order reduction of the waves' gradient rows: every wave scatters its row into
// LDS (re-using the block's whole dynamic LDS, all waves are past their task loops), then the block
// writes ONE row to HBM.  Deterministic: (((w0 + w1) + w2) + ...).
template <int R, int F, bool DG>
__device__ __forceinline__ void gru_block_partials(float* smem, float* partials, const GruLayout& L, GruGrad<R, DG>& G,
                                                   int lane, int wave, int nwb, int row, int col, float loss_part) {
    const int P4 = L.P + kLossCols;
    __syncthreads();
    gru_write_partials<R, F, DG>(smem + wave * P4, L, G, lane, row, col, loss_part);
    __syncthreads();
    float* prow = partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = smem[i];
        for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// forward kernel
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG>
__global__ __launch_bounds__(kMaxThreads) void gru_fwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R, LPS = 16 * R, S = kCkptStride;
    using T = GruTabs<R, DG>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<R>();
    const int lane = id.lane, col = id.col, s = id.s;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_gru_tabs<R, DG, false>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + T::kFloats) + id.wave * (2 * SPW * kChunkPad);
    float2* ys = xs + SPW * kChunkPad;
    GruW<R, F, DG> w;
    load_gru_w<R, F, DG>(w, pl, L, id.row, col);
    float whh[3][R][16];
    load_rot3<R>(whh, tlane, T::kHH);

    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float h = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            // the sample and its features are formed one step AHEAD of the step that consumes them: the LDS read, sqrt and rcp are then off
            // the recurrence's dependent chain (a lone wave at small batch / evaluation is bound by that chain)
            float fnext[F];
            {
                const float2 x0 = xs[s * kChunkPad];
                feat_fwd<FM>(x0.x, x0.y, fnext);
            }
            for (int tt = 0; tt < len; ++tt) {
                float f[F], r, z, n, ghn, y0, y1;
#pragma unroll
                for (int i = 0; i < F; ++i) f[i] = fnext[i];
                {
                    const float2 xn = xs[s * kChunkPad + min(tt + 1, len - 1)];
                    feat_fwd<FM>(xn.x, xn.y, fnext);
                }
                gru_cell_fwd<R, FM, DG>(w, whh, f, h, r, z, n, ghn);
                float act = h;
                if constexpr (DG) act = __builtin_fmaxf(tab_rotdot<R>(w.bhid, tlane, T::kHID, h), 0.0f);
                gru_head_out<R, FM, DG>(w, act, f, col, y0, y1);
                if ((lane & (LPS - 1)) == 0) ys[s * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (a.ckpt != nullptr && (t1 % S) == 0 && t1 < a.T)
                    a.ckpt[((size_t)grp * a.nck + t1 / S) * 64 + lane] = h;
            }
            wave_lds_fence();
            stage_out<SPW>(ys, a.y, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
        }
    }
}

// -------------------------------------------------------------------------------------------------
// evaluation kernel (net_eval / run_dpd, train_funcs.py:57-90: a few very long sequences — (1, 19 662, 2), (3, 2 560, 2) — where one
// wave per sequence runs alone on its SIMD and the step time is the whole cost; the row-rotated forward spends half of it on its 45
// half-rate v_fmac_dpp).  Gate-parallel mapping, ONE sequence per wave: the four 16-lane rows of the wave hold the same h
// and each does ONE of the step's four mat-vecs with its own rotated weights — row 0: r, row 1: n (W_in x apart from W_hn h), row 3: z,
// row 2: the head of the PREVIOUS step (fc_hid for DGRU, fc_out) — so a step issues one rotated dot product (15 v_fmac_dpp) instead of
// three or four (hidden 17..32: four instead of sixteen).  The gates meet through three cross-row moves on the VALU (gfx950's v_permlane16_swap / v_permlane32_swap): r to the
// n row, n <-> z between rows 1 and 3, h' from rows 1 / 3 to rows 0 / 2.  Same arithmetic per element as gru_cell_fwd (the recurrent
// sums run as two chains instead of one).
// -------------------------------------------------------------------------------------------------
template <int NB> struct GruEvalLds {
    static constexpr int kHistStride = 64 * NB + 4;      // one float per lane and unit block; + 4: the per-chunk reads (lane = time step) spread over the banks
    static constexpr int kHeadFloats = 2 * 16 * NB + 16;
    static constexpr int kFloats = (kEvalChunk + 1) * 8 + kEvalChunk * kHistStride + kHeadFloats;
};
// NB = 1: hidden <= 16; NB = 2: hidden 17..32 — h is then two 16-unit blocks, both replicated on every row, and a row does its gate's
// two output blocks (four rotated dot products instead of the twelve + four of the two-row forward).
// What does not depend on the state leaves the step loop: the features of a 64-step chunk are computed with lane = time step and
// parked in LDS (a step reads them as wave-uniform operands), and so is fc_out — a step only parks the head's input (h, or relu(fc_hid h)
// from row 2) and the chunk's 64 outputs are formed afterwards, one time step per lane.  The next chunk's samples are in flight
// while the current one is stepped.
// CK: also writes the BPTT checkpoints (the forward of the split train path)
// HALF (NB = 2, hidden 17..24): the second block holds its <= 8 units twice, 8-rotation dot products over it (fill_gru_tabs<.., HALF>)
template <int NB, int FM, bool DG, bool CK, bool HALF = false>
__device__ __forceinline__ void gru_eval_body(const SeqArgs& a, const int bid, const int nbl) {
    static_assert(!HALF || NB == 2, "half-block layout: two-block models");
    constexpr int F = FeatDim<FM>::F, EC = kEvalChunk, HS = GruEvalLds<NB>::kHistStride;
    using T = GruTabs<NB, DG>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;      // 0 r | 1 n | 2 head | 3 z
    const GruLayout L = gru_layout(a.H, F, DG);
    const int H = L.H, OW = DG ? H + 6 : H;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_gru_tabs<NB, DG, false, HALF>(tab, pl, L, lane, 0, 1);
    float* ftab = tab + T::kFloats;                    // [EC + 1][8]: entry 1 + i = features of time t0 + i, entry 0 = of time t0 - 1
    float* hist = ftab + (EC + 1) * 8;                 // [EC][HS]: entry i = the head's input of time t0 + i - 1, every lane's copy
    float* hw = hist + EC * HS;                        // fc_out: [2][16 NB] hidden columns (zero padded) | [2][8] feature columns
    for (int i = lane; i < GruEvalLds<NB>::kHeadFloats; i += 64) {
        float v = 0.0f;
        if (i < 32 * NB) { const int c = i / (16 * NB), u = i % (16 * NB); if (u < H) v = pl[L.o_w_out + c * OW + u]; }
        else { const int j = i - 32 * NB, c = j >> 3, k = j & 7; if (DG && k < 6) v = pl[L.o_w_out + c * OW + H + k]; }
        hw[i] = v;
    }
    wave_lds_fence();
    // the row's own operands: its gate's input weights and biases for both output blocks, and the rotated recurrent weights
    // wrec[ob][kb] = rows 16 ob + col of the gate's matrix against K-block kb (the table holds them per (output block, relative K-block))
    const int gate = role == 0 ? 0 : role == 3 ? 1 : 2;
    float win[NB][F], wrec[NB][NB][16], b_in[NB], b_rec[NB], wo0[NB], wo1[NB];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
        const bool replica = HALF && ob == 1 && col >= 8;                  // the lane carries a second copy of unit 16 + col % 8
        const int o = 16 * ob + ((HALF && ob == 1) ? (col & 7) : col);
        const bool vo = o < H;
#pragma unroll
        for (int i = 0; i < F; ++i) win[ob][i] = (vo && role != 2) ? pl[L.o_w_ih + (gate * H + o) * F + i] : 0.0f;
        b_in[ob] = 0.0f; b_rec[ob] = 0.0f;
        if (vo) {
            if (role == 0 || role == 3) b_in[ob] = pl[L.o_b_ih + gate * H + o] + pl[L.o_b_hh + gate * H + o];
            if (role == 1) { b_in[ob] = pl[L.o_b_ih + 2 * H + o]; b_rec[ob] = pl[L.o_b_hh + 2 * H + o]; }
            if (role == 2 && DG) b_rec[ob] = pl[L.o_b_hid + o];
        }
        // table entries are per lane of the two-row layout: lanes 16 ob .. 16 ob + 15 hold output block ob; entry rb covers K-block (ob + rb) % NB
        TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + 16 * ob + col);
#pragma unroll
        for (int rb = 0; rb < NB; ++rb) {
            const int kb = (ob + rb) % NB;
            if (role != 2 || DG) load_rot(wrec[ob][kb], tl + ((role == 2 ? T::kHID : T::kHH + gate * NB) + rb) * 4 * 64);
            else {
#pragma unroll
                for (int k = 0; k < 16; ++k) wrec[ob][kb][k] = 0.0f;
            }
        }
        // (fc_out of the last step is a sum over the row's lanes: a replica must not count twice)
        wo0[ob] = (vo && !replica) ? pl[L.o_w_out + o] : 0.0f; wo1[ob] = (vo && !replica) ? pl[L.o_w_out + OW + o] : 0.0f;
    }
    const float wf0 = (DG && col < 6) ? pl[L.o_w_out + H + col] : 0.0f, wf1 = (DG && col < 6) ? pl[L.o_w_out + OW + H + col] : 0.0f;
    const float bo0 = pl[L.o_b_out], bo1 = pl[L.o_b_out + 1];
    const bool odd = role & 1;
    const float4* ftab4 = reinterpret_cast<const float4*>(ftab);

    for (int b = bid; b < a.B; b += nbl) {
        float h[NB];
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) h[kb] = 0.0f;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * a.T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * a.T;
        auto matvec = [&](float (&arec)[NB]) {
#pragma unroll
            for (int ob = 0; ob < NB; ++ob) {
                arec[ob] = b_rec[ob];
#pragma unroll
                for (int kb = 0; kb < NB; ++kb) arec[ob] = (HALF && kb == 1) ? rotdot8_1(arec[ob], wrec[ob][kb], h[kb]) : rotdot1(arec[ob], wrec[ob][kb], h[kb]);
            }
        };
        float2 raw = lane < a.T ? xg[lane] : make_float2(0.5f, 0.5f);
        for (int t0 = 0; t0 < a.T; t0 += EC) {
            const int len = __builtin_amdgcn_readfirstlane(min(EC, a.T - t0));     // (a scalar trip count: the step loop's counter off the VALU)
            {
                float f[F];
                feat_fwd<FM>(raw.x, raw.y, f);
                float f8[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) f8[i] = i < F ? f[i] : 0.0f;
                wave_lds_fence();
                reinterpret_cast<float4*>(ftab)[2 * (1 + lane)] = make_float4(f8[0], f8[1], f8[2], f8[3]);
                reinterpret_cast<float4*>(ftab)[2 * (1 + lane) + 1] = make_float4(f8[4], f8[5], f8[6], f8[7]);
                wave_lds_fence();
            }
            raw = t0 + EC + lane < a.T ? xg[t0 + EC + lane] : make_float2(0.5f, 0.5f);
            for (int tt = 0; tt < len; ++tt) {
                float f[F];
                {
                    const float4 fa = ftab4[2 * (1 + tt)];
                    f[0] = fa.x; f[1] = fa.y;
                    if constexpr (F > 2) { f[2] = fa.z; f[3] = fa.w; }
                    if constexpr (F > 4) { const float4 fb = ftab4[2 * (1 + tt) + 1]; f[4] = fb.x; f[5] = fb.y; }
                }
                float arec[NB];
                matvec(arec);
                // the head's input of the PREVIOUS step rides on this step's mat-vec (row 2): parked, fc_out follows per chunk
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) hist[tt * HS + 64 * ob + lane] = DG ? __builtin_fmaxf(arec[ob], 0.0f) : h[ob];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    float ain = b_in[ob];
#pragma unroll
                    for (int i = 0; i < F; ++i) ain = __builtin_fmaf(win[ob][i], f[i], ain);
                    const float sg = sigmoidf_(ain + arec[ob]);                 // r (row 0), z (row 3)
                    const RowDup s2 = dup16(sg);                                // .even on row 1: r, .odd on row 3: z
                    const float n = tanhf_(__builtin_fmaf(s2.even, arec[ob], ain));   // row 1
                    const HalfDup nz = dup32(role == 1 ? n : s2.odd);           // rows 1 and 3: lo = n, hi = z
                    const float h13 = __builtin_fmaf(nz.hi, h[ob] - nz.lo, nz.lo);    // rows 1 and 3: (1 - z) n + z h
                    h[ob] = dup16(h13).odd;                                     // every row: its half's row 1 / 3
                }
                // BPTT checkpoints in the layout of the row-rotated backward (4 / NB sequences per wave-task, lane = 16 NB s + 16 ob + col)
                if constexpr (CK) {
                    const int t1 = t0 + tt + 1;
                    if ((t1 % kCkptStride) == 0 && t1 < a.T && role == 0) {
                        float* ck = a.ckpt + ((size_t)(b / (4 / NB)) * a.nck + t1 / kCkptStride) * 64 + 16 * NB * (b % (4 / NB)) + col;
#pragma unroll
                        for (int ob = 0; ob < NB; ++ob) ck[16 * ob] = (HALF && ob == 1 && col >= 8) ? 0.0f : h[ob];      // (replica lanes: padding units there)
                    }
                }
            }
            wave_lds_fence();
            // fc_out of the chunk, lane = time step: entry `lane` of hist / ftab belongs to time t0 + lane - 1
            if (lane < len && t0 + lane >= 1) {
                const float4* hv4 = reinterpret_cast<const float4*>(hist + lane * HS + 32);        // row 2's copy
                const float4* hw4 = reinterpret_cast<const float4*>(hw);
                float y0 = bo0, y1 = bo1;
#pragma unroll
                for (int ob = 0; ob < NB; ++ob)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 hv = hv4[16 * ob + q], w0 = hw4[4 * ob + q], w1 = hw4[4 * NB + 4 * ob + q];
                        y0 = __builtin_fmaf(w0.x, hv.x, y0); y0 = __builtin_fmaf(w0.y, hv.y, y0); y0 = __builtin_fmaf(w0.z, hv.z, y0); y0 = __builtin_fmaf(w0.w, hv.w, y0);
                        y1 = __builtin_fmaf(w1.x, hv.x, y1); y1 = __builtin_fmaf(w1.y, hv.y, y1); y1 = __builtin_fmaf(w1.z, hv.z, y1); y1 = __builtin_fmaf(w1.w, hv.w, y1);
                    }
                if constexpr (DG) {
                    const float4 fa = ftab4[2 * lane], fb = ftab4[2 * lane + 1];
                    const float4 u0 = hw4[8 * NB], u1 = hw4[8 * NB + 1], v0 = hw4[8 * NB + 2], v1 = hw4[8 * NB + 3];
                    y0 = __builtin_fmaf(u0.x, fa.x, y0); y0 = __builtin_fmaf(u0.y, fa.y, y0); y0 = __builtin_fmaf(u0.z, fa.z, y0); y0 = __builtin_fmaf(u0.w, fa.w, y0);
                    y0 = __builtin_fmaf(u1.x, fb.x, y0); y0 = __builtin_fmaf(u1.y, fb.y, y0);
                    y1 = __builtin_fmaf(v0.x, fa.x, y1); y1 = __builtin_fmaf(v0.y, fa.y, y1); y1 = __builtin_fmaf(v0.z, fa.z, y1); y1 = __builtin_fmaf(v0.w, fa.w, y1);
                    y1 = __builtin_fmaf(v1.x, fb.x, y1); y1 = __builtin_fmaf(v1.y, fb.y, y1);
                }
                yg[t0 + lane - 1] = make_float2(y0, y1);
            }
            // the chunk's last features become entry 0 of the next one
            float carry = 0.0f;
            if (lane < 8) carry = ftab[len * 8 + lane];
            wave_lds_fence();
            if (lane < 8) ftab[lane] = carry;
        }
        wave_lds_fence();
        // the head of the last state (time T - 1: its features are entry 0 now)
        float arec[NB];
        matvec(arec);
        float p0 = 0.0f, p1 = 0.0f;
        if constexpr (DG) { const float fsp = col < 6 ? ftab[col] : 0.0f; p0 = wf0 * fsp; p1 = wf1 * fsp; }
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
            const float act = DG ? __builtin_fmaxf(arec[ob], 0.0f) : h[ob];
            p0 = __builtin_fmaf(wo0[ob], act, p0); p1 = __builtin_fmaf(wo1[ob], act, p1);
        }
        const float y0 = row_sum16(p0) + bo0, y1 = row_sum16(p1) + bo1;
        if (lane == 32) yg[a.T - 1] = make_float2(y0, y1);
        wave_lds_fence();
    }
}
template <int NB, int FM, bool DG, bool CK, bool HALF = false>
__global__ __launch_bounds__(64) void gru_eval_kernel(SeqArgs a) {
    gru_eval_body<NB, FM, DG, CK, HALF>(a, blockIdx.x, gridDim.x);
}
// the evaluation pass of K models of one shape on the same sequences (odpd_backbone_fwd_sweep): model k owns workgroups [k G, (k + 1) G)
template <int NB, int FM, bool DG, bool HALF>
__global__ __launch_bounds__(64) void gru_eval_sweep_kernel(SeqArgs a, const SweepRun* __restrict__ runs, int G) {
    const SweepRun r = runs[blockIdx.x / G];
    a.params = r.params; a.y = r.y;
    gru_eval_body<NB, FM, DG, false, HALF>(a, blockIdx.x % G, G);
}

// -------------------------------------------------------------------------------------------------
// One block of S steps: pull W_hh, recompute the forward pass of the block into registers, pull
// W_hh^T into the same registers, back-propagate.  FULL: nstep == S, no per-step guards.
// tloc = first step of the block relative to the staged chunk.
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG, bool NW, bool DX, bool FUSED, bool FULL>
__device__ __forceinline__ void gru_bwd_block(const SeqArgs& a, const GruW<R, FeatDim<FM>::F, DG>& w, TabPtr tlane,
                                              GruGrad<R, DG>& G, const LaneId& id, const float2* xs, const float2* dys,
                                              float2* dxs, int tloc, int nstep, bool valid, float h, float& dh,
                                              float& loss_acc) {
    constexpr int F = FeatDim<FM>::F, LPS = 16 * R, S = kCkptStride;
    using T = GruTabs<R, DG>;
    const int lane = id.lane, col = id.col, row = id.row, s = id.s;
    float hp_s[S], r_s[S], z_s[S], n_s[S], g_s[S], hid_s[S];
    tlane = opaque(tlane);
    {
        float whh[3][R][16];
        load_rot3<R>(whh, tlane, T::kHH);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            if (FULL || i < nstep) {
                const float2 xv = xs[s * kChunkPad + tloc + i];
                float f[F];
                feat_fwd<FM>(xv.x, xv.y, f);
                hp_s[i] = h;
                gru_cell_fwd<R, FM, DG>(w, whh, f, h, r_s[i], z_s[i], n_s[i], g_s[i]);
                hid_s[i] = 0.0f;
                if constexpr (DG) hid_s[i] = tab_rotdot<R>(w.bhid, tlane, T::kHID, h);
            }
        }
    }
    tlane = opaque(tlane);
    float whhT[3][R][16];
    load_rot3<R>(whhT, tlane, T::kHHT);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, valid ? a.inv_count : 0.0f, valid && (lane & (LPS - 1)) == 0);
#pragma unroll
    for (int i = S - 1; i >= 0; --i) {
        if (FULL || i < nstep) {
            const int tt = tloc + i;
            const float2 xv = xs[s * kChunkPad + tt];
            float2 dyv = dys[s * kChunkPad + tt];
            float f[F], df[F];
            feat_fwd<FM>(xv.x, xv.y, f);
            if constexpr (FUSED) {
                // output head from the recomputed state, then loss and dL/dy (train_funcs.py:35-39)
                float y0, y1;
                const float act = DG ? __builtin_fmaxf(hid_s[i], 0.0f) : __builtin_fmaf(z_s[i], hp_s[i] - n_s[i], n_s[i]);
                gru_head_out<R, FM, DG>(w, act, f, col, y0, y1);
                const float d0 = y0 - dyv.x, d1 = y1 - dyv.y;
                s16_loss(lossc, d0, d1, dyv.x, dyv.y, loss_acc);
            }
            gru_step_bwd<R, FM, DG, NW, DX>(w, whhT, tlane, G, f, hp_s[i], r_s[i], z_s[i], n_s[i], g_s[i], hid_s[i],
                                            dyv.x, dyv.y, row, col, dh, df);
            if constexpr (DX) {
                float dI, dQ;
                feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
                if ((lane & (LPS - 1)) == 0) dxs[s * kChunkPad + tt] = make_float2(dI, dQ);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward over one wave-task.  Shared by the stand-alone backward kernel (dy staged per chunk from
// HBM, checkpoints in HBM) and the fused train kernel (FUSED: the target is staged instead of dy;
// checkpoints live in LDS).
//   xs : LDS chunk buffer for x     dys : LDS chunk buffer for dy (target if FUSED)
//   ck : checkpoints of this task, slot c at ck[c*64 + lane]  (HBM, or LDS if FUSED)
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG, bool NW, bool DX, bool FUSED>
__device__ __forceinline__ void gru_bwd_task(const SeqArgs& a, const GruW<R, FeatDim<FM>::F, DG>& w, TabPtr tlane,
                                             GruGrad<R, DG>& G, int b0, const LaneId& id, float2* xs, float2* dys,
                                             float2* dxs, const float* ck, float& loss_acc) {
    constexpr int SPW = 4 / R, S = kCkptStride;
    const int lane = id.lane;
    float dh = 0.0f;
    int cur_chunk = -1;
    const bool valid = b0 + id.s < a.B;
    for (int blk = a.nck - 1; blk >= 0; --blk) {
        const int tb = blk * S, nstep = min(S, a.T - tb);
        const int chunk = tb / kChunk, t0 = chunk * kChunk;
        if (chunk != cur_chunk) {
            if constexpr (DX) {
                if (cur_chunk >= 0) {
                    const int pt0 = cur_chunk * kChunk;
                    wave_lds_fence();
                    stage_out<SPW>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                }
            }
            wave_lds_fence();
            const int len = min(kChunk, a.T - t0);
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
            if constexpr (FUSED)
                stage_in<SPW>(dys, a.target, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
            else
                stage_in<SPW>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
            wave_lds_fence();
            cur_chunk = chunk;
        }
        const float h0 = blk ? ck[blk * 64 + lane] : 0.0f;
        if (nstep == S)  // branch-free body: one basic block, the scheduler overlaps consecutive steps
            gru_bwd_block<R, FM, DG, NW, DX, FUSED, true>(a, w, tlane, G, id, xs, dys, dxs, tb - t0, nstep, valid, h0, dh,
                                                          loss_acc);
        else
            gru_bwd_block<R, FM, DG, NW, DX, FUSED, false>(a, w, tlane, G, id, xs, dys, dxs, tb - t0, nstep, valid, h0, dh,
                                                           loss_acc);
    }
    if constexpr (DX) {
        if (cur_chunk >= 0) {
            const int pt0 = cur_chunk * kChunk;
            wave_lds_fence();
            stage_out<SPW>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
            wave_lds_fence();
        }
    }
}

// one-row models fit two waves per SIMD (8 per CU, one 512-thread workgroup)
template <int R, int FM, bool DG, bool NW, bool DX>
__global__ __launch_bounds__(R == 1 ? kMaxThreads : kMaxThreads / 2, R == 1 ? 2 : 1) void gru_bwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R;
    using T = GruTabs<R, DG>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<R>();
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_gru_tabs<R, DG, true>(tab, pl, L, id.lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + id.lane);
    float2* xs = reinterpret_cast<float2*>(tab + T::kFloats) + id.wave * (3 * SPW * kChunkPad);
    float2* dys = xs + SPW * kChunkPad;
    float2* dxs = dys + SPW * kChunkPad;
    GruW<R, F, DG> w;
    load_gru_w<R, F, DG>(w, pl, L, id.row, id.col);
    GruGrad<R, DG> G;
    G.zero();
    float unused = 0.0f;
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves)
        gru_bwd_task<R, FM, DG, NW, DX, false>(a, w, tlane, G, grp * SPW, id, xs, dys, dxs,
                                               a.ckpt + (size_t)grp * a.nck * 64, unused);
    if constexpr (NW) gru_block_partials<R, F, DG>(smem, a.partials, L, G, id.lane, id.wave, id.nwb, id.row, id.col, 0.0f);
}

// -------------------------------------------------------------------------------------------------
// fused train kernel: per wave-task  (1) forward pass of the cell only, leaving a checkpoint of h
// every S steps in LDS;  (2) backward pass that recomputes each block, forms y, the loss and dL/dy
// on the fly and back-propagates.  HBM traffic = x (twice, the second read is L2-resident) + target.
// LDS per wave: x chunk, target chunk, checkpoints (nck x 64 floats).
// -------------------------------------------------------------------------------------------------
__host__ __device__ inline int train_wave_floats(int T, int R, bool dx = false) {
    return (dx ? 3 : 2) * (2 * (4 / R) * kChunkPad) + num_ckpt_hd(T) * 64;
}

// NW: weight-gradient partials (train_pa / trained model).  !NW && DX: the frozen PA of a cascade in one launch — forward, loss,
// dL/dx into a.dx, one loss partial per workgroup in a.partials[blockIdx.x * kLossCols]
template <int R, int FM, bool DG, bool NW = true, bool DX = false>
__global__ __launch_bounds__(R == 1 ? kMaxThreads : kMaxThreads / 2, R == 1 ? 2 : 1) void gru_train_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R, S = kCkptStride;
    using T = GruTabs<R, DG>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<R>();
    const int lane = id.lane, s = id.s;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_gru_tabs<R, DG, true>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float* wbase = tab + T::kFloats + (size_t)id.wave * train_wave_floats(a.T, R, DX);
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + SPW * kChunkPad;
    float2* dxs = DX ? ts + SPW * kChunkPad : nullptr;
    float* ck = reinterpret_cast<float*>(ts + (DX ? 2 : 1) * SPW * kChunkPad);
    GruW<R, F, DG> w;
    load_gru_w<R, F, DG>(w, pl, L, id.row, id.col);
    GruGrad<R, DG> G;
    G.zero();
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        {
            float whh[3][R][16];
            load_rot3<R>(whh, opaque(tlane), T::kHH);
            float h = 0.0f;
            for (int t0 = 0; t0 < a.T; t0 += kChunk) {
                const int len = min(kChunk, a.T - t0);
                wave_lds_fence();
                stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                wave_lds_fence();
                int tt = 0;
                static_assert(kChunk % S == 0, "chunks start on a checkpoint boundary");
                for (; tt + S <= len; tt += S) {   // S steps per iteration, one checkpoint at the end
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        const float2 xv = xs[s * kChunkPad + tt + i];
                        float f[F], r, z, n, ghn;
                        feat_fwd<FM>(xv.x, xv.y, f);
                        gru_cell_fwd<R, FM, DG>(w, whh, f, h, r, z, n, ghn);
                    }
                    const int t1 = t0 + tt + S;
                    if (t1 < a.T) ck[(t1 / S) * 64 + lane] = h;
                }
                for (; tt < len; ++tt) {            // tail of the last chunk (never ends on a checkpoint < T)
                    const float2 xv = xs[s * kChunkPad + tt];
                    float f[F], r, z, n, ghn;
                    feat_fwd<FM>(xv.x, xv.y, f);
                    gru_cell_fwd<R, FM, DG>(w, whh, f, h, r, z, n, ghn);
                }
            }
        }
        wave_lds_fence();
        gru_bwd_task<R, FM, DG, NW, DX, true>(a, w, tlane, G, b0, id, xs, ts, dxs, ck, loss_acc);
    }
    if constexpr (NW) {
        gru_block_partials<R, F, DG>(smem, a.partials, L, G, lane, id.wave, id.nwb, id.row, id.col, loss_acc);
    } else {       // loss partial of the workgroup: lane sums in lane order, waves in wave order
        float lp = loss_acc;                           // non-zero on the first lane of every sequence only
        for (int o = 32; o > 0; o >>= 1) lp += __shfl_down(lp, o);
        __syncthreads();
        if (lane == 0) smem[id.wave] = lp;
        __syncthreads();
        if (threadIdx.x < kLossCols) {
            float v = 0.0f;
            if (threadIdx.x == 0)
                for (int wv = 0; wv < id.nwb; ++wv) v += smem[wv];
            a.partials[(size_t)blockIdx.x * kLossCols + threadIdx.x] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Gate-parallel fused train kernel for the reference's own batch sizes (64 .. ~1 000 frames of 50 / 200 samples, train_funcs.py:28-48):
// there a wave is alone on its SIMD and the T-serial chain is the whole cost, so ONE sequence per wave (one wave per workgroup) and
// only the recurrence in the step loops.  hidden <= 16.
//   forward   gate-parallel as gru_eval_kernel (rows r | n | head | z, one rotated dot product per step); every h(t) — and, DGRU,
//             every relu(fc_hid h(t)) from the head row — is parked in LDS, features come per 64-step chunk with lane = time step;
//   head      fc_out, the loss and dL/dy of all T steps with lane = time step;
//   backward  per step ONE rotated dot product with the forward weights (gates recomputed from the parked h(t-1): no serial
//             recompute chain) and ONE with the transposed weights — rows r | n | z multiply their own gate's pre-activation gradient,
//             the head row multiplies fc_hid^T dhid of the step below, and the cross-row sum is dL/dh(t-1) complete;
//             the weight gradients of a step are TWO 4-block MFMAs (v_mfma_f32_16x16x1_4b_f32: block k = the outer product of row k's
//             operands): (d_r | d_gh | d_hid | d_z) x (h(t-1) | h(t-1) | h(t) | h(t-1)) and (d_r | d_n | 0 | d_z) x (features | 1).
// LDS per wave: parameters + max(weight tables — read once into registers —, the per-time buffers: 8 + 16 (+ 16) + 2 floats per step).
// One partial-gradient row per workgroup.
// -------------------------------------------------------------------------------------------------
// frozen-model variant: pitch of a unit's row in the parked pre-activation gradients (odd: the 16 lanes of a row hit 16 banks)
__host__ __device__ inline int gp_dx_pitch(int T) { return T | 1; }
template <int NB, bool DG>
__host__ __device__ inline int gp_buffer_floats(int T, bool pg, bool fz = false) {
    const int Tp = (T + 63) & ~63, Th = fz ? T : Tp;
    const int buf = Tp * 8 + (Th + 2) * 16 * NB + (DG ? Th * 16 * NB : 0) + Tp * 2 + 256 * NB + 32 * NB + 16 + (pg ? Tp * 64 * NB : 0) +
                    (fz ? 3 * 16 * NB * gp_dx_pitch(T) : 0);
    const int tabf = GruTabs<NB, DG>::kFloats;
    return buf > tabf ? buf : tabf;
}
// NB = 2: hidden 17..32 — h is two 16-unit blocks replicated on every row, a row does both output blocks of its gate (as gru_eval_kernel);
// the W_hh / fc_hid gradients are one 4-block MFMA per (output block, input block) pair.
// PG: the forward pass also parks (r, W_hn h + b, z, n) of every step (16 B per unit and step) and the backward pass reads them back instead
// of recomputing the gates — taken while the frame's buffers fit the CU's LDS share
// FZ: the frozen model in front of the loss (the PA of train_dpd at the reference's batch sizes): no weight gradients; the backward
// steps park the three pre-activation gradients of every unit instead ([gate][unit][t], 12 B per unit and step) and dL/dx of all T steps
// follows with lane = time step (W_ih^T d, the fc_out feature columns, the feature Jacobian).  partials = loss rows (grid, kLossCols).
// HALF (NB = 2, hidden 17..24 — the reference's default PA has 23 units): the second 16-lane block holds its <= 8 units twice (odpd_gru.h,
// fill_gru_tabs<.., HALF>), the rotated dot products over it take 8 rotations instead of 16: 48 of the 192 DPP FMAs of a time step.  The
// replica lanes carry unit indices >= 24 in every per-unit write-out and in the MFMA blocks' rows / columns, which hidden <= 24 masks off.
// (the body takes its workgroup index and count as arguments: the sweep launch below runs it for K models side by side)
template <int NB, int FM, bool DG, bool PG, bool FZ = false, bool HALF = false>
__device__ __forceinline__ void gru_gp_train_body(const SeqArgs& a, const int bid, const int nbl) {
    static_assert(!(PG && FZ), "the frozen variant recomputes the gates");
    static_assert(!HALF || NB == 2, "half-block layout: two-block models");
    constexpr int F = FeatDim<FM>::F, HB = 16 * NB;
    using TB = GruTabs<NB, DG>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;      // 0 r | 1 n | 2 head | 3 z
    const GruLayout L = gru_layout(a.H, F, DG);
    const int H = L.H, OW = DG ? H + 6 : H, T = a.T, Tp = (T + 63) & ~63, Th = FZ ? T : Tp;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_gru_tabs<NB, DG, true, HALF>(tab, pl, L, lane, 0, 1);
    const int gate = role == 0 ? 0 : role == 3 ? 1 : 2;
    const bool head_row = role == 2;
    // the row's rotated weights, forward and transposed (head row: fc_hid for DGRU, nothing otherwise): [output block][input block]
    float wF[NB][NB][16], wT[NB][NB][16];
    float win[NB][F], b_in[NB], b_rec[NB], wo0[NB], wo1[NB];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
        const int o = 16 * ob + ((HALF && ob == 1) ? (col & 7) : col);      // the unit this lane carries in output block ob
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
    const float bo0 = pl[L.o_b_out], bo1 = pl[L.o_b_out + 1];
    // rotated dot product over input block kb
    auto rd = [](float acc, const float (&w)[16], float v, int kb) { return (HALF && kb == 1) ? rotdot8(acc, w, v) : rotdot(acc, w, v); };
    // (forward: one accumulator chain, the order of the evaluation kernel and of rotdot3 — where it measured faster: the dgru step 0.1244 -> 0.1208 ms, the
    //  plain gru step 0.0940 -> 0.0956: that one keeps two chains)
    auto rd1 = [](float acc, const float (&w)[16], float v, int kb) {
        if constexpr (!DG) return (HALF && kb == 1) ? rotdot8(acc, w, v) : rotdot(acc, w, v);
        else return (HALF && kb == 1) ? rotdot8_1(acc, w, v) : rotdot1(acc, w, v);
    };
    wave_lds_fence();
    // per-time buffers over the tables
    float* ftab = tab;                                  // [Tp][8]   features of step t
    float* hist = ftab + Tp * 8;                        // [Tp + 2][HB]   entry t + 1 = h(t) (unit u at 16 ob + col), entry 0 = h(-1) = 0
    float* actb = hist + (Th + 2) * HB;                 // DGRU: [Tp][HB]   relu(fc_hid h(t) + b)
    float* dyb = actb + (DG ? Th * HB : 0);             // [Tp][2]   dL/dy(t)
    float* dump = dyb + Tp * 2;                         // [256 NB]
    float* hw = dump + 256 * NB;                        // fc_out: [2][HB] hidden columns (zero padded) | [2][8] feature columns
    float* gpk = hw + 2 * HB + 16;                      // PG: [Tp][NB][16][4]   r, W_hn h + b_hn, z, n of step t (written by the n row)
    float* dpk = gpk;                                   // FZ: [3][HB][pitch]   d_r, d_z, d_n of unit u at step t (written by the n row)
    const int pitch = gp_dx_pitch(T);
    for (int i = lane; i < 2 * HB + 16; i += 64) {
        float v = 0.0f;
        if (i < 2 * HB) { const int c = i / HB, u = i % HB; if (u < H) v = pl[L.o_w_out + c * OW + u]; }
        else { const int j = i - 2 * HB, c = j >> 3, k = j & 7; if (DG && k < 6) v = pl[L.o_w_out + c * OW + H + k]; }
        hw[i] = v;
    }
    if (lane < HB) hist[lane] = 0.0f;
    const float4* ftab4 = reinterpret_cast<const float4*>(ftab);
    const float4* hw4 = reinterpret_cast<const float4*>(hw);
    const bool odd = role & 1;
    const RowMasks rm = row_masks();
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    // the per-step stores of the forward pass: row 1 parks h(t), the head row parks relu(fc_hid h(t-1)), rows 0 / 3 hit the dump
    const int park0 = role == 1 ? (int)(hist - smem) + HB + col : (head_row && DG) ? (int)(actb - smem) - HB + col : (int)(dump - smem) + lane;
    const int park_step = (role == 1 || (head_row && DG)) ? HB : 0;
    const int gpark0 = role == 1 ? (int)(gpk - smem) + 4 * col : (int)(dump - smem) + 4 * lane, gpark_step = role == 1 ? 64 * NB : 0;
    // FZ, backward: row 1 parks (d_r, d_z, d_n) of its unit, the other rows hit the dump
    const int dpark0 = role == 1 ? (int)(dpk - smem) + col * pitch : (int)(dump - smem) + lane;
    const int dpark_gate = role == 1 ? HB * pitch : 0, dpark_ob = role == 1 ? 16 * pitch : 0, dpark_t = role == 1 ? 1 : 0;

    f32x16 acc1[NB][NB], acc2[NB];
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc2[ob][i] = 0.0f;
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) acc1[ob][kb][i] = 0.0f;
        }
    }
    float dmisc[NB], dwo0[NB], dwo1[NB], dwf0 = 0.0f, dwf1 = 0.0f, dbo0 = 0.0f, dbo1 = 0.0f, loss_acc = 0.0f;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) { dmisc[ob] = 0.0f; dwo0[ob] = 0.0f; dwo1[ob] = 0.0f; }

    // the gates of one step from h(t-1): arec = W h + b of the row's gate, rows 1 and 3 end with (z, n), row 1 also with r
    auto gates = [&](const float (&f)[F], const float (&hin)[NB], float (&arec)[NB], float (&r1)[NB], float (&zz)[NB], float (&nn)[NB]) {
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
    };
    auto load_feat = [&](int t, float (&f)[F]) {
        const float4 fa = ftab4[2 * t];
        f[0] = fa.x; f[1] = fa.y;
        if constexpr (F > 2) { f[2] = fa.z; f[3] = fa.w; }
        if constexpr (F > 4) { const float4 fb = ftab4[2 * t + 1]; f[4] = fb.x; f[5] = fb.y; }
    };

    for (int b = bid; b < a.B; b += nbl) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const bool bf = a.frame_idx != nullptr && a.frames_bf16 != 0;          // bf16 sample storage of the resident streams
        auto xg = [&](int i) { return ld_iq(a.x, base + i, bf); };
        auto tg = [&](int i) { return ld_iq(a.target, base + i, bf); };
        // ---- forward ----
        {
            float h[NB];
#pragma unroll
            for (int kb = 0; kb < NB; ++kb) h[kb] = 0.0f;
            int park = park0, gpark = gpark0;
            float2 raw = lane < T ? xg(lane) : make_float2(0.5f, 0.5f);
            for (int t0 = 0; t0 < T; t0 += kEvalChunk) {
                const int len = min(kEvalChunk, T - t0);
                {
                    float f[F];
                    feat_fwd<FM>(raw.x, raw.y, f);
                    float f8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) f8[i] = i < F ? f[i] : 0.0f;
                    wave_lds_fence();
                    reinterpret_cast<float4*>(ftab)[2 * (t0 + lane)] = make_float4(f8[0], f8[1], f8[2], f8[3]);
                    reinterpret_cast<float4*>(ftab)[2 * (t0 + lane) + 1] = make_float4(f8[4], f8[5], f8[6], f8[7]);
                    wave_lds_fence();
                }
                raw = t0 + kEvalChunk + lane < T ? xg(t0 + kEvalChunk + lane) : make_float2(0.5f, 0.5f);
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
                        if constexpr (PG) *reinterpret_cast<float4*>(smem + gpark + 64 * ob) = make_float4(r1[ob], arec[ob], zz[ob], nn[ob]);
                    }
                    park += park_step;
                    if constexpr (PG) gpark += gpark_step;
                }
            }
            if constexpr (DG) {
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    float arec = b_rec[ob];
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) arec = rd1(arec, wF[ob][kb], h[kb], kb);
                    if (head_row) actb[(T - 1) * HB + 16 * ob + col] = __builtin_fmaxf(arec, 0.0f);
                }
            }
            wave_lds_fence();
        }
        // ---- fc_out, loss and dL/dy of every step, lane = time step ----
        if (a.dy != nullptr) {            // split train path: dL/dy comes from the caller (the loss kernel, or the PA half of a cascade)
            const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
            for (int t = lane; t < T; t += 64) *reinterpret_cast<float2*>(dyb + 2 * t) = dyg[t];
        } else
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
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
                const float2 tv = tg(t);
                float dy0, dy1;
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
                *reinterpret_cast<float2*>(dyb + 2 * t) = make_float2(dy0, dy1);
            }
        }
        wave_lds_fence();
        // ---- backward ----
        {
            float carry[NB], dhid_cur[NB];       // carry = dL/dh(t) complete (cell path + head path)
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
                    carry[ob] = sum_rows4(part);
                }
            }
            for (int t = T - 1; t >= 0; --t) {
                float hp[NB], ht[NB], at[NB];
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    hp[ob] = hist[t * HB + 16 * ob + col]; ht[ob] = hist[(t + 1) * HB + 16 * ob + col];
                    at[ob] = DG ? actb[t * HB + 16 * ob + col] : ht[ob];
                }
                const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
                const float fsx = col < F ? ftab[t * 8 + col] : (col == F ? 1.0f : 0.0f);
                // the gates of step t: parked by the forward pass, or again from the parked h(t-1)
                float arec[NB], r1[NB], zz[NB], nn[NB];                              // rows 1 and 3: z, n; row 1: r, W_hn h + b_hn
                if constexpr (PG) {
#pragma unroll
                    for (int ob = 0; ob < NB; ++ob) {
                        const float4 g = reinterpret_cast<const float4*>(gpk)[(t * NB + ob) * 16 + col];
                        r1[ob] = g.x; arec[ob] = g.y; zz[ob] = g.z; nn[ob] = g.w;
                    }
                } else {
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
                    // (vsel: a plain ?: on the row index comes out as exec-mask branches here)
                    d_row[ob] = vsel(rm.m[0], drp0, vsel(rm.m[1], dgh, vsel(rm.m[3], dzp, dhid_prev[ob])));
                    zterm[ob] = vsel(rm.m[3], dht * zz[ob], 0.0f);
                    if constexpr (FZ) {
                        const int dp = dpark0 + ob * dpark_ob + t * dpark_t;
                        smem[dp] = drp1; smem[dp + dpark_gate] = dzp; smem[dp + 2 * dpark_gate] = dnp[ob];
                    } else {
                        dmisc[ob] += vsel(rm.m[1], dgh, dhid_cur[ob]);               // row 1: db_hn, head row: db_hid
                        dwo0[ob] = __builtin_fmaf(dyv.x, at[ob], dwo0[ob]); dwo1[ob] = __builtin_fmaf(dyv.y, at[ob], dwo1[ob]);
                    }
                }
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    float part = zterm[ob];
#pragma unroll
                    for (int kb = 0; kb < NB; ++kb) part = rd(part, wT[ob][kb], d_row[kb], kb);
                    carry[ob] = sum_rows4(part);                                                // dL/dh(t-1): W_hh^T d + z dL/dh(t) + fc_hid^T dhid(t-1)
                }
                // weight gradients
#pragma unroll
                for (int ob = 0; ob < NB; ++ob) {
                    if constexpr (!FZ) {
                        const float a1 = vsel(rm.m[2], dhid_cur[ob], d_row[ob]);
#pragma unroll
                        for (int kb = 0; kb < NB; ++kb)
                            acc1[ob][kb] = __builtin_amdgcn_mfma_f32_16x16x1f32(a1, vsel(rm.m[2], ht[kb], hp[kb]), acc1[ob][kb], 0, 0, 0);
                        acc2[ob] = __builtin_amdgcn_mfma_f32_16x16x1f32(vsel(rm.m[1], dnp[ob], vsel(rm.m[2], 0.0f, d_row[ob])), fsx, acc2[ob], 0, 0, 0);
                    }
                    dhid_cur[ob] = dhid_prev[ob];
                }
                if constexpr (!FZ) {
                    if constexpr (DG) {
                        const float fs = col < 6 ? fsx : 0.0f;
                        dwf0 = __builtin_fmaf(dyv.x, fs, dwf0); dwf1 = __builtin_fmaf(dyv.y, fs, dwf1);
                    }
                    dbo0 += dyv.x; dbo1 += dyv.y;
                }
            }
        }
        wave_lds_fence();
        if constexpr (FZ) {
            // ---- dL/dx of every step, lane = time step: W_ih^T (d_r | d_z | d_n)(t) [+ the fc_out feature columns x dL/dy(t)], then the
            //      feature Jacobian at x(t) ----
            float2* dxg = reinterpret_cast<float2*>(a.dx) + (size_t)b * T;
            for (int t0 = 0; t0 < T; t0 += 64) {
                const int t = t0 + lane;
                if (t < T) {
                    float df[F];
#pragma unroll
                    for (int i = 0; i < F; ++i) df[i] = 0.0f;
                    for (int u = 0; u < H; ++u) {
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            const float d = dpk[(g * HB + u) * pitch + t];
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
                    const float2 xv = xg(t);
                    float dI, dQ;
                    feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
                    dxg[t] = make_float2(dI, dQ);
                }
            }
            wave_lds_fence();
        }
    }
    if constexpr (FZ) {      // the workgroup's loss row
        float lp = loss_acc;
        for (int o = 32; o > 0; o >>= 1) lp += __shfl_down(lp, o);
        if (lane < kLossCols) a.partials[(size_t)bid * kLossCols + lane] = lane == 0 ? lp : 0.0f;
        return;
    }
    // ---- the workgroup's row of partial gradients (every entry written) ----
    float* prow = a.partials + (size_t)bid * (L.P + kLossCols);
    float lp = loss_acc;
    for (int o = 32; o > 0; o >>= 1) lp += __shfl_down(lp, o);
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
        prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
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
template <int NB, int FM, bool DG, bool PG, bool FZ = false, bool HALF = false>
__global__ __launch_bounds__(64) void gru_gp_train_kernel(SeqArgs a) {
    gru_gp_train_body<NB, FM, DG, PG, FZ, HALF>(a, blockIdx.x, gridDim.x);
}
// K independent runs of one model shape in lockstep (odpd_train_epoch_sweep: seeds of a sweep, bash_scripts/train_all_pa.sh:26-57): run k owns
// workgroups [k G, (k + 1) G) and sees exactly the launch it would have had alone — its parameters, its epoch order, its partial rows
template <int NB, int FM, bool DG, bool PG, bool HALF>
__global__ __launch_bounds__(64) void gru_gp_train_sweep_kernel(SeqArgs a, const SweepRun* __restrict__ runs, int G, long long first) {
    const SweepRun r = runs[blockIdx.x / G];
    a.params = r.params; a.partials = r.partials; a.frame_idx = r.order + first;
    gru_gp_train_body<NB, FM, DG, PG, false, HALF>(a, blockIdx.x % G, G);
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
constexpr int kFwdWavesPerCU = 16;  // 4 waves per SIMD (forward kernels use <= 128 VGPRs for R = 1)
static int bwd_waves_per_cu(int R) { return R == 1 ? 8 : 4; }

static bool gru_cfg(const odpd_model_t* m, int& FM, bool& DG) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; DG = false; return true;
    case ODPD_DGRU: FM = FEAT_DGRU6; DG = true; return true;
    case ODPD_QGRU: FM = FEAT_Q4; DG = false; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; DG = false; return true;
    default: return false;
    }
}
static int gru_tab_floats(int R, bool DG) { return (DG ? 8 * R : 6 * R) * 4 * 64 * 4; }
// dynamic LDS of a block: params + tables + per-wave region; never smaller than the reduce scratch
static size_t gru_lds_bytes(int P, int R, bool DG, int waves, size_t wave_floats, bool reduce) {
    size_t n = ((size_t)pad4(P) + gru_tab_floats(R, DG) + (size_t)waves * wave_floats) * sizeof(float);
    if (reduce && n < reduce_scratch_bytes(P, waves)) n = reduce_scratch_bytes(P, waves);
    return n;
}
// launch shape of the backward / fused kernels (also fixes the number of partial rows = grid)
static LaunchShape bwd_shape(int R, int ngroups) { return persistent_shape(ngroups, bwd_waves_per_cu(R), R == 1 ? 8 : 4); }
// the fused kernel additionally has to fit its LDS-resident checkpoints: shrink the block if needed
static LaunchShape train_shape(int P, int R, bool DG, int ngroups, int T, size_t* lds_out, bool dx = false) {
    LaunchShape ls = bwd_shape(R, ngroups);
    for (;;) {
        size_t lds = gru_lds_bytes(P, R, DG, ls.waves, train_wave_floats(T, R, dx), true);
        if (lds <= kMaxLds || ls.waves == 1) {
            if (lds_out) *lds_out = lds;
            if (lds > kMaxLds) ls.grid = 0;   // does not fit at all
            else {
                int need = (ngroups + ls.waves - 1) / ls.waves, cap = device_cus() * (int)(kMaxLds / lds);
                ls.grid = need < cap ? need : cap;
            }
            return ls;
        }
        ls.waves /= 2;
    }
}

template <int R, int FM, bool DG>
static int launch_fwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = persistent_shape(a.ngroups, kFwdWavesPerCU);
    const size_t lds = gru_lds_bytes(P, R, DG, ls.waves, 2 * 2 * (4 / R) * kChunkPad, false);
    auto k = gru_fwd_kernel<R, FM, DG>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
template <int R, int FM, bool DG, bool NW, bool DX>
static int launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = bwd_shape(R, a.ngroups);
    const size_t lds = gru_lds_bytes(P, R, DG, ls.waves, 3 * 2 * (4 / R) * kChunkPad, NW);
    auto k = gru_bwd_kernel<R, FM, DG, NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
template <int R, int FM, bool DG>
static int launch_train(hipStream_t st, const SeqArgs& a, int P) {
    size_t lds = 0;
    const LaunchShape ls = train_shape(P, R, DG, a.ngroups, a.T, &lds);
    if (ls.grid <= 0) return ODPD_EUNSUPPORTED;  // frame too long for LDS-resident BPTT state
    auto k = gru_train_kernel<R, FM, DG>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}

// frozen PA of a cascade: forward + loss + dL/du in one launch (a.x = u, a.target, a.dx = du, a.partials = loss rows)
template <int R, int FM, bool DG>
static int launch_lossdx(hipStream_t st, const SeqArgs& a, int P) {
    size_t lds = 0;
    const LaunchShape ls = train_shape(P, R, DG, a.ngroups, a.T, &lds, true);
    if (ls.grid <= 0) return ODPD_EUNSUPPORTED;
    auto k = gru_train_kernel<R, FM, DG, false, true>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}

#define ODPD_GRU_DISPATCH(R_, FM_, DG_, CALL) \
    if (R == R_ && FM == FM_ && DG == DG_) return CALL;
#define ODPD_GRU_DISPATCH_ALL(FN, ...)                                              \
    ODPD_GRU_DISPATCH(1, FEAT_RAW2, false, (FN<1, FEAT_RAW2, false>(__VA_ARGS__)))  \
    ODPD_GRU_DISPATCH(2, FEAT_RAW2, false, (FN<2, FEAT_RAW2, false>(__VA_ARGS__)))  \
    ODPD_GRU_DISPATCH(1, FEAT_DGRU6, true, (FN<1, FEAT_DGRU6, true>(__VA_ARGS__)))  \
    ODPD_GRU_DISPATCH(2, FEAT_DGRU6, true, (FN<2, FEAT_DGRU6, true>(__VA_ARGS__)))  \
    ODPD_GRU_DISPATCH(1, FEAT_Q4, false, (FN<1, FEAT_Q4, false>(__VA_ARGS__)))      \
    ODPD_GRU_DISPATCH(2, FEAT_Q4, false, (FN<2, FEAT_Q4, false>(__VA_ARGS__)))      \
    ODPD_GRU_DISPATCH(1, FEAT_A4, false, (FN<1, FEAT_A4, false>(__VA_ARGS__)))      \
    ODPD_GRU_DISPATCH(2, FEAT_A4, false, (FN<2, FEAT_A4, false>(__VA_ARGS__)))

template <int R, int FM, bool DG>
static int launch_bwd_mode(hipStream_t st, const SeqArgs& a, int P) {
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && !dx) return launch_bwd<R, FM, DG, true, false>(st, a, P);
    if (!nw && dx) return launch_bwd<R, FM, DG, false, true>(st, a, P);
    if (nw && dx) return launch_bwd<R, FM, DG, true, true>(st, a, P);
    return ODPD_EINVAL;
}

static bool gru_setup(const odpd_model_t* m, int& FM, bool& DG, int& R, int& P) {
    if (!gru_cfg(m, FM, DG)) return false;
    R = rows_per_seq(m->hidden);
    if (!R) return false;
    P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    return true;
}

// hidden 17..32: the generic-tile S16 kernels (gru_s16n.hip) take over from the two-row DPP kernels at the same batch
bool gru_uses_s16n(const odpd_model_t* m, int B) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || R != 2) return false;
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 8L * 4 * device_cus();    // measured crossover (DGRU-23, T = 200): 1.54 vs 1.33 ms at 8192
    return B >= min_batch;
}

// The split kernels switch to the 16-sequences-per-wave mapping at the same batch size as the fused one.  Forward
// and backward of one (B,T) batch must agree (checkpoint layout), so the rule only looks at the model and B.
bool gru_split_uses_s16(const odpd_model_t* m, int B) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || R != 1) return false;
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 16L * 4 * device_cus();
    return B >= min_batch;
}

// inference on a few long sequences (no checkpoints asked for): the gate-parallel evaluation kernel, one sequence per wave
template <int NB, int FM, bool DG>
static int launch_eval(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = ((size_t)pad4(P) + GruTabs<NB, DG>::kFloats + GruEvalLds<NB>::kFloats) * sizeof(float);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(a.B), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    if constexpr (NB == 2)
        if (a.H <= 24) return a.ckpt ? launch(gru_eval_kernel<NB, FM, DG, true, true>) : launch(gru_eval_kernel<NB, FM, DG, false, true>);
    return a.ckpt ? launch(gru_eval_kernel<NB, FM, DG, true>) : launch(gru_eval_kernel<NB, FM, DG, false>);
}
bool gru_uses_eval_kernel(const odpd_model_t* m, int B, int T, bool want_ckpt) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || tuning().s16_min_batch == 0 || tuning().gp_max_batch == 0) return false;
    // a few long sequences (net_eval / run_dpd), or any batch whose sequences each get a SIMD of their own before the S16 kernels take over
    // (then also as the checkpoint-writing forward of the split train path)
    return (!want_ckpt && B <= 8 && T >= 256) || (B <= 2 * device_cus() && !gru_split_uses_s16(m, B) && !gru_uses_s16n(m, B));
}
int gru_family_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P)) return ODPD_EUNSUPPORTED;
    if (gru_uses_eval_kernel(m, a.B, a.T, a.ckpt != nullptr)) {
        if (R == 1) {
            if (FM == FEAT_RAW2) return launch_eval<1, FEAT_RAW2, false>(st, a, P);
            if (FM == FEAT_DGRU6) return launch_eval<1, FEAT_DGRU6, true>(st, a, P);
            if (FM == FEAT_Q4) return launch_eval<1, FEAT_Q4, false>(st, a, P);
            return launch_eval<1, FEAT_A4, false>(st, a, P);
        }
        if (FM == FEAT_RAW2) return launch_eval<2, FEAT_RAW2, false>(st, a, P);
        if (FM == FEAT_DGRU6) return launch_eval<2, FEAT_DGRU6, true>(st, a, P);
        if (FM == FEAT_Q4) return launch_eval<2, FEAT_Q4, false>(st, a, P);
        return launch_eval<2, FEAT_A4, false>(st, a, P);
    }
    if (gru_split_uses_s16(m, a.B)) return gru_s16_fwd(st, m, a);
    if (gru_uses_s16n(m, a.B)) return gru_s16x_train_ok(m) ? gru_s16x_fwd(st, m, a) : gru_s16n_launch(st, m, a, 1);
    ODPD_GRU_DISPATCH_ALL(launch_fwd, st, a, P)
    return ODPD_EUNSUPPORTED;
}
static bool gru_bwd_uses_gp(const odpd_model_t* m, int B, int T);
static int gp_grid(int P, int R, bool DG, int B, int T);
template <int R, int FM, bool DG>
static int launch_gp_train(hipStream_t st, const SeqArgs& a, int P);
int gru_family_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P)) return ODPD_EUNSUPPORTED;
    if (gru_split_uses_s16(m, a.B)) return gru_s16_bwd(st, m, a);
    if (gru_uses_s16n(m, a.B)) return gru_s16x_train_ok(m) ? gru_s16x_bwd(st, m, a, gru_s16n_rows(m, a.B)) : gru_s16n_launch(st, m, a, 2);
    if (gru_bwd_uses_gp(m, a.B, a.T)) {
        // weight gradients only: the fused one-sequence-per-wave kernel with dL/dy given (it runs its own forward; the checkpoints stay unread)
        if (a.partials != nullptr && a.dx == nullptr) { ODPD_GRU_DISPATCH_ALL(launch_gp_train, st, a, P) }
        // dL/dx asked for: the row-rotated kernel writes fewer rows than the caller's buffer holds — the rest are zero
        if (a.partials != nullptr) {
            const int have = bwd_shape(R, a.ngroups).grid, rows = gp_grid(P, R, DG, a.B, a.T);
            if (rows > have)
                ODPD_CHECK_HIP(hipMemsetAsync(a.partials + (size_t)have * (P + kLossCols), 0, (size_t)(rows - have) * (P + kLossCols) * sizeof(float), st));
        }
    }
    ODPD_GRU_DISPATCH_ALL(launch_bwd_mode, st, a, P)
    return ODPD_EUNSUPPORTED;
}
// the gate-parallel fused train kernel: one sequence per single-wave workgroup, BPTT state of the whole frame in LDS
static size_t gp_lds_bytes(int P, int R, bool DG, int T, bool pg) {
    const int buf = R == 1 ? (DG ? gp_buffer_floats<1, true>(T, pg) : gp_buffer_floats<1, false>(T, pg))
                           : (DG ? gp_buffer_floats<2, true>(T, pg) : gp_buffer_floats<2, false>(T, pg));
    return ((size_t)pad4(P) + buf) * sizeof(float);
}
// workgroups of a CU that the frame's LDS-resident BPTT state allows (at most one per SIMD)
static int gp_blocks_per_cu(int P, int R, bool DG, int T, bool pg) {
    const size_t lds = gp_lds_bytes(P, R, DG, T, pg);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}
// the variant that parks the gates is taken while every sequence of the batch still gets its own SIMD
static bool gp_parks_gates(int P, int R, bool DG, int B, int T) { return (long)B <= (long)device_cus() * gp_blocks_per_cu(P, R, DG, T, true); }
bool gru_train_uses_gp(const odpd_model_t* m, int B, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || gru_uses_s16n(m, B) || gru_train_uses_s16(m, B, T)) return false;
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && gp_blocks_per_cu(P, R, DG, T, false) > 0;
    return (long)B <= (long)device_cus() * gp_blocks_per_cu(P, R, DG, T, false);        // one sequence per SIMD, all resident at once
}
// the split backward (dL/dy given) on the same kernel: hidden <= 16, every sequence resident at once
static bool gru_bwd_uses_gp(const odpd_model_t* m, int B, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || R != 1 || gru_split_uses_s16(m, B) || tuning().gp_max_batch == 0 || tuning().s16_min_batch == 0) return false;
    return (long)B <= (long)device_cus() * gp_blocks_per_cu(P, R, DG, T, false);
}
static int gp_grid(int P, int R, bool DG, int B, int T) {
    const bool pg = gp_parks_gates(P, R, DG, B, T);
    const long cap = (long)device_cus() * (kMaxLds / gp_lds_bytes(P, R, DG, T, pg));
    return B < cap ? B : (int)cap;
}
template <int R, int FM, bool DG>
static int launch_gp_train(hipStream_t st, const SeqArgs& a, int P) {
    const bool pg = gp_parks_gates(P, R, DG, a.B, a.T);
    const size_t lds = gp_lds_bytes(P, R, DG, a.T, pg);
    const int grid = gp_grid(P, R, DG, a.B, a.T);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    if constexpr (R == 2)
        if (a.H <= 24) return pg ? launch(gru_gp_train_kernel<R, FM, DG, true, false, true>) : launch(gru_gp_train_kernel<R, FM, DG, false, false, true>);
    return pg ? launch(gru_gp_train_kernel<R, FM, DG, true>) : launch(gru_gp_train_kernel<R, FM, DG, false>);
}
// ---- lockstep sweeps: K runs of one model shape, each exactly the solo launch it replaces (same instantiation, same grid per run) ----
bool gru_sweep_train_ok(const odpd_model_t* m, int B, int T) { return m->bits_w == 0 && gru_train_uses_gp(m, B, T); }
int gru_sweep_train_rows(const odpd_model_t* m, int B, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || !gru_sweep_train_ok(m, B, T)) return ODPD_EUNSUPPORTED;
    return gp_grid(P, R, DG, B, T);
}
template <int R, int FM, bool DG>
static int launch_gp_sweep(hipStream_t st, const SeqArgs& a, int P, const SweepRun* runs, int K, long long first) {
    const bool pg = gp_parks_gates(P, R, DG, a.B, a.T);
    const size_t lds = gp_lds_bytes(P, R, DG, a.T, pg);
    const int G = gp_grid(P, R, DG, a.B, a.T);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)G * K), dim3(64), lds, st, a, runs, G, first);
        return (int)hipGetLastError();
    };
    if constexpr (R == 2)
        if (a.H <= 24) return pg ? launch(gru_gp_train_sweep_kernel<R, FM, DG, true, true>) : launch(gru_gp_train_sweep_kernel<R, FM, DG, false, true>);
    return pg ? launch(gru_gp_train_sweep_kernel<R, FM, DG, true, false>) : launch(gru_gp_train_sweep_kernel<R, FM, DG, false, false>);
}
int gru_sweep_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, const SweepRun* runs, int K, long long first) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || !gru_sweep_train_ok(m, a.B, a.T) || K <= 0 || !runs) return ODPD_EUNSUPPORTED;
    if (R == 1) {
        if (FM == FEAT_RAW2) return launch_gp_sweep<1, FEAT_RAW2, false>(st, a, P, runs, K, first);
        if (FM == FEAT_DGRU6) return launch_gp_sweep<1, FEAT_DGRU6, true>(st, a, P, runs, K, first);
        if (FM == FEAT_Q4) return launch_gp_sweep<1, FEAT_Q4, false>(st, a, P, runs, K, first);
        return launch_gp_sweep<1, FEAT_A4, false>(st, a, P, runs, K, first);
    }
    if (FM == FEAT_RAW2) return launch_gp_sweep<2, FEAT_RAW2, false>(st, a, P, runs, K, first);
    if (FM == FEAT_DGRU6) return launch_gp_sweep<2, FEAT_DGRU6, true>(st, a, P, runs, K, first);
    if (FM == FEAT_Q4) return launch_gp_sweep<2, FEAT_Q4, false>(st, a, P, runs, K, first);
    return launch_gp_sweep<2, FEAT_A4, false>(st, a, P, runs, K, first);
}
bool gru_sweep_eval_ok(const odpd_model_t* m, int B, int T) { return m->bits_w == 0 && gru_uses_eval_kernel(m, B, T, false); }
template <int NB, int FM, bool DG>
static int launch_eval_sweep(hipStream_t st, const SeqArgs& a, int P, const SweepRun* runs, int K) {
    const size_t lds = ((size_t)pad4(P) + GruTabs<NB, DG>::kFloats + GruEvalLds<NB>::kFloats) * sizeof(float);
    const int G = a.B;
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3((unsigned)G * K), dim3(64), lds, st, a, runs, G);
        return (int)hipGetLastError();
    };
    if constexpr (NB == 2)
        if (a.H <= 24) return launch(gru_eval_sweep_kernel<NB, FM, DG, true>);
    return launch(gru_eval_sweep_kernel<NB, FM, DG, false>);
}
int gru_sweep_eval(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, const SweepRun* runs, int K) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || !gru_sweep_eval_ok(m, a.B, a.T) || K <= 0 || !runs) return ODPD_EUNSUPPORTED;
    if (R == 1) {
        if (FM == FEAT_RAW2) return launch_eval_sweep<1, FEAT_RAW2, false>(st, a, P, runs, K);
        if (FM == FEAT_DGRU6) return launch_eval_sweep<1, FEAT_DGRU6, true>(st, a, P, runs, K);
        if (FM == FEAT_Q4) return launch_eval_sweep<1, FEAT_Q4, false>(st, a, P, runs, K);
        return launch_eval_sweep<1, FEAT_A4, false>(st, a, P, runs, K);
    }
    if (FM == FEAT_RAW2) return launch_eval_sweep<2, FEAT_RAW2, false>(st, a, P, runs, K);
    if (FM == FEAT_DGRU6) return launch_eval_sweep<2, FEAT_DGRU6, true>(st, a, P, runs, K);
    if (FM == FEAT_Q4) return launch_eval_sweep<2, FEAT_Q4, false>(st, a, P, runs, K);
    return launch_eval_sweep<2, FEAT_A4, false>(st, a, P, runs, K);
}
// the frozen-model variant (forward + loss + dL/dx): taken while every sequence of the batch is resident at once
static size_t gp_fz_lds_bytes(int P, int R, bool DG, int T) {
    const int buf = R == 1 ? (DG ? gp_buffer_floats<1, true>(T, false, true) : gp_buffer_floats<1, false>(T, false, true))
                           : (DG ? gp_buffer_floats<2, true>(T, false, true) : gp_buffer_floats<2, false>(T, false, true));
    return ((size_t)pad4(P) + buf) * sizeof(float);
}
static int gp_fz_grid(int P, int R, bool DG, int B, int T) {
    const size_t lds = gp_fz_lds_bytes(P, R, DG, T);
    if (lds > kMaxLds) return 0;
    const int per_cu = (int)(kMaxLds / lds);
    const long cap = (long)device_cus() * (per_cu < 4 ? per_cu : 4);
    return B <= cap ? B : 0;
}
static bool gru_lossdx_uses_gp(const odpd_model_t* m, int B, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || gru_uses_s16n(m, B) || gru_split_uses_s16(m, B)) return false;
    // hidden 25..32 without the fc_hid head: the two-block steps leave nothing to gain over the row-rotated kernel (0.97x measured,
    // profiles/r03/frozen_pa_bench.md); 17..24 (second block held twice, 8-rotation dot products) 1.07x, with the head 1.2-1.3x, hidden <= 16 1.35-1.85x
    if (R == 2 && !DG && m->hidden > 24 && tuning().gp_max_batch < 0) return false;
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0 && B > max_batch) return false;
    return gp_fz_grid(P, R, DG, B, T) > 0;
}
template <int R, int FM, bool DG>
static int launch_gp_lossdx(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = gp_fz_lds_bytes(P, R, DG, a.T);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(gp_fz_grid(P, R, DG, a.B, a.T)), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    if constexpr (R == 2)
        if (a.H <= 24) return launch(gru_gp_train_kernel<R, FM, DG, false, true, true>);
    return launch(gru_gp_train_kernel<R, FM, DG, false, true>);
}
int gru_family_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P)) return ODPD_EUNSUPPORTED;
    if (gru_train_uses_gp(m, a.B, a.T)) { ODPD_GRU_DISPATCH_ALL(launch_gp_train, st, a, P) }
    ODPD_GRU_DISPATCH_ALL(launch_train, st, a, P)
    return ODPD_EUNSUPPORTED;
}
// frozen PA of a cascade: forward + loss + dL/du in one launch; loss rows = (rows, kLossCols)
int gru_family_lossdx(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P)) return ODPD_EUNSUPPORTED;
    if (gru_uses_s16n(m, a.B)) return gru_s16x_ok(m) ? gru_s16x_lossdx(st, m, a, gru_s16n_rows(m, a.B)) : gru_s16n_launch(st, m, a, 3);
    if (gru_split_uses_s16(m, a.B)) return gru_s16_lossdx(st, m, a);
    if (gru_lossdx_uses_gp(m, a.B, a.T)) { ODPD_GRU_DISPATCH_ALL(launch_gp_lossdx, st, a, P) }
    ODPD_GRU_DISPATCH_ALL(launch_lossdx, st, a, P)
    return ODPD_EUNSUPPORTED;
}
int gru_family_lossdx_rows(const odpd_model_t* m, int B, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P)) return ODPD_EUNSUPPORTED;
    if (gru_uses_s16n(m, B)) return gru_s16n_rows(m, B);
    if (gru_split_uses_s16(m, B)) return gru_s16_rows(m, B);
    if (gru_lossdx_uses_gp(m, B, T)) return gp_fz_grid(P, R, DG, B, T);
    const LaunchShape ls = train_shape(P, R, DG, num_groups(B, R), T, nullptr, true);
    return ls.grid > 0 ? ls.grid : ODPD_EUNSUPPORTED;      // frame too long for LDS-resident BPTT state: use the split calls
}
// rows of partials written by the backward (which = 0) or fused (which = 1, needs T) kernels
int gru_family_rows(const odpd_model_t* m, int B, int which, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P)) return ODPD_EUNSUPPORTED;
    const int ng = num_groups(B, R);
    if (gru_uses_s16n(m, B)) return gru_s16n_rows(m, B);
    if (!which) {
        if (gru_split_uses_s16(m, B)) return gru_s16_bwd_rows(m, B);
        const int have = bwd_shape(R, ng).grid;
        if (gru_bwd_uses_gp(m, B, T)) { const int rows = gp_grid(P, R, DG, B, T); return rows > have ? rows : have; }
        return have;
    }
    if (gru_train_uses_s16(m, B, T)) return gru_s16_rows(m, B);
    if (gru_train_uses_gp(m, B, T)) return gp_grid(P, R, DG, B, T);
    const LaunchShape ls = train_shape(P, R, DG, ng, T, nullptr);
    return ls.grid > 0 ? ls.grid : ODPD_EUNSUPPORTED;
}

// Which fused kernel serves a (B,T) batch: the row-rotated kernel (4 sequences per wave, BPTT state in LDS)
// has the shorter per-step latency and wins while the batch cannot fill the chip with 16-sequence waves; the
// S16 kernel (gru_s16.hip) has ~1.7x the throughput once it can, and no frame-length limit.
// odpd_set_tuning("s16_min_batch") / $ODPD_S16_MIN_BATCH override the crossover (0 = always S16 where
// supported, a huge value = never).
bool gru_train_uses_s16(const odpd_model_t* m, int B, int T) {
    int FM, R, P; bool DG;
    if (!gru_setup(m, FM, DG, R, P) || R != 1) return false;
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 16L * 4 * device_cus();   // one 16-sequence wave per SIMD
    if (B >= min_batch) return true;
    return train_shape(P, R, DG, num_groups(B, R), T, nullptr).grid <= 0;   // frame too long for LDS checkpoints
}

}  // namespace odpd
