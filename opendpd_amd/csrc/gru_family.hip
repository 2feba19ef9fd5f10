// gru_family.hip — persistent-RNN kernels for the nn.GRU based backbones of the reference:
//   gru        backbones/gru.py:4-48        y = fc_out(GRU(x))
//   dgru       backbones/dgru.py:9-74       feat=[I,Q,a,a^3,sin,cos]; y = fc_out(cat(relu(fc_hid(h)), feat))
//   qgru       backbones/qgru.py:9-71       feat=[I,Q,a^2,a^4]  (float path)
//   qgru_amp1  backbones/qgru_amp1.py:9-76  feat=[I,Q,a,a^3]    (float path)
// GRU cell = torch.nn.GRU semantics, gate order r,z,n:
//   r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r*(W_hn h + b_hn)),
//   h' = (1-z)*n + z*h.
//
// Kernels (one wavefront = 4/R sequences, see odpd_device.h for the lane mapping):
//   gru_fwd_kernel    forward over T steps; writes y and, for training, a checkpoint of h every
//                     kCkptStride steps (BPTT state = 64 B per sequence per 4 steps instead of
//                     5 activations per step).
//   gru_bwd_kernel    BPTT: walks the checkpoints backwards, recomputes each block of S steps into
//                     registers, back-propagates it; weight gradients are rank-4 (R=1) exact-fp32
//                     MFMA updates; one row of partial gradients per workgroup (deterministic).
//   gru_train_kernel  forward + loss + backward fused in one launch; dy and the checkpoints stay
//                     in LDS, HBM traffic = x + target only.
#include "odpd_host.h"

namespace odpd {

// -------------------------------------------------------------------------------------------------
// register-resident weights
// -------------------------------------------------------------------------------------------------
template <int R, int F, bool DG>
struct GruW {
    float whh[3][R][16];   // [gate][0 = own row, 1 = other row][k]  W_hg[o][16*rowblk + src_k]
    float wih[3][F];
    float b_r, b_z, b_in, b_hn;
    float wout[2], bout[2];
    float whid[DG ? R : 1][16];
    float bhid;
    float woutf[2];        // DG: fc_out weight of feature `col` (row-0 lanes, col < 6), else 0
};
template <int R, int F, bool DG>
struct GruWT {             // transposed copies for the data-gradient mat-vecs
    float whhT[3][R][16];  // W_hg[16*rowblk + src_k][o]
    const float4* whidT_q; // DG: fc_hid^T streamed from the LDS master copy: quad q of row block rb at
                           //     whidT_q[(rb*4 + q)*64]  (pointer already offset by the lane)
    const float4* whid_q;  // DG: fc_hid (forward orientation) in the same quad layout, used by the
                           //     block recompute so that the backward kernels do not pin it in VGPRs
};
// floats of the per-block LDS region holding the rotated fc_hid^T and fc_hid quads
__host__ __device__ inline int whidT_lds_floats(int R, bool DG) { return DG ? 2 * R * 4 * 64 * 4 : 0; }

template <int R, int F, bool DG>
__device__ __forceinline__ void load_gru_w(GruW<R, F, DG>& w, const float* pl, const GruLayout& L, int row, int col,
                                           const int (&src)[16]) {
    const int H = L.H, o = 16 * row + col;
    const bool vo = o < H;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int rb = 0; rb < R; ++rb)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int m = 16 * ((row + rb) % R) + src[k];
                w.whh[g][rb][k] = (vo && m < H) ? pl[L.o_w_hh + (g * H + o) * H + m] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < F; ++i) w.wih[g][i] = vo ? pl[L.o_w_ih + (g * H + o) * F + i] : 0.0f;
    }
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
    w.bhid = 0.0f;
    if constexpr (DG) {
        w.bhid = vo ? pl[L.o_b_hid + o] : 0.0f;
#pragma unroll
        for (int rb = 0; rb < R; ++rb)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int m = 16 * ((row + rb) % R) + src[k];
                w.whid[rb][k] = (vo && m < H) ? pl[L.o_w_hid + o * H + m] : 0.0f;
            }
    }
}
// wl: per-block LDS region of whidT_lds_floats() floats, filled by wave 0 (every wave holds the same
// per-lane values); contains a __syncthreads() — call from all threads of the block.
template <int R, int F, bool DG>
__device__ __forceinline__ void load_gru_wT(GruWT<R, F, DG>& w, const float* pl, const GruLayout& L, int row, int col,
                                            const int (&src)[16], float* wl) {
    const int H = L.H, o = 16 * row + col;
    const bool vo = o < H;
    const int lane = threadIdx.x & 63;
    float4* wq = reinterpret_cast<float4*>(wl);
#pragma unroll
    for (int rb = 0; rb < R; ++rb)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int m = 16 * ((row + rb) % R) + src[k];
            const bool v = vo && m < H;
#pragma unroll
            for (int g = 0; g < 3; ++g) w.whhT[g][rb][k] = v ? pl[L.o_w_hh + (g * H + m) * H + o] : 0.0f;
        }
    w.whidT_q = nullptr;
    w.whid_q = nullptr;
    if constexpr (DG) {
        if (threadIdx.x < 64) {
#pragma unroll
            for (int rb = 0; rb < R; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int m = 16 * ((row + rb) % R) + src[4 * q + e];
                        t[e] = (vo && m < H) ? pl[L.o_w_hid + m * H + o] : 0.0f;
                    }
                    wq[(rb * 4 + q) * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int m = 16 * ((row + rb) % R) + src[4 * q + e];
                        t[e] = (vo && m < H) ? pl[L.o_w_hid + o * H + m] : 0.0f;
                    }
                    wq[((R + rb) * 4 + q) * 64 + lane] = make_float4(t[0], t[1], t[2], t[3]);
                }
        }
        __syncthreads();
        w.whidT_q = wq + lane;
        w.whid_q = wq + R * 4 * 64 + lane;
    }
}

// -------------------------------------------------------------------------------------------------
// per-step device functions
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG>
__device__ __forceinline__ void gru_cell_fwd(const GruW<R, FeatDim<FM>::F, DG>& w, const float (&f)[FeatDim<FM>::F],
                                             float& h, float& r, float& z, float& n, float& ghn) {
    constexpr int F = FeatDim<FM>::F;
    float ar = w.b_r, az = w.b_z, an = w.b_in, ah = w.b_hn;
#pragma unroll
    for (int i = 0; i < F; ++i) {
        ar = __builtin_fmaf(w.wih[0][i], f[i], ar);
        az = __builtin_fmaf(w.wih[1][i], f[i], az);
        an = __builtin_fmaf(w.wih[2][i], f[i], an);
    }
    rotdot3(ar, az, ah, w.whh[0][0], w.whh[1][0], w.whh[2][0], h);
    if constexpr (R == 2) {
        float hx = swap16(h);
        rotdot3(ar, az, ah, w.whh[0][1], w.whh[1][1], w.whh[2][1], hx);
    }
    r = sigmoidf_(ar);
    z = sigmoidf_(az);
    ghn = ah;
    n = tanhf_(__builtin_fmaf(r, ah, an));
    h = __builtin_fmaf(z, h - n, n);  // (1-z)*n + z*h
}

// output head.  hid = fc_hid pre-activation (DG only)
template <int R, int FM, bool DG>
__device__ __forceinline__ void gru_head_fwd(const GruW<R, FeatDim<FM>::F, DG>& w, float h,
                                             const float (&f)[FeatDim<FM>::F], int col, float& y0, float& y1, float& hid) {
    float p0, p1;
    hid = 0.0f;
    if constexpr (DG) {
        hid = rotdot(w.bhid, w.whid[0], h);
        if constexpr (R == 2) hid = rotdot(hid, w.whid[1], swap16(h));
        float a = __builtin_fmaxf(hid, 0.0f);
        float fs = feat_select<6>(f, col, 0.0f);
        p0 = __builtin_fmaf(w.wout[0], a, w.woutf[0] * fs);
        p1 = __builtin_fmaf(w.wout[1], a, w.woutf[1] * fs);
    } else {
        p0 = w.wout[0] * h;
        p1 = w.wout[1] * h;
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
__device__ __forceinline__ void gru_step_bwd(const GruW<R, FeatDim<FM>::F, DG>& w, const GruWT<R, FeatDim<FM>::F, DG>& wt,
                                             GruGrad<R, DG>& G, const float (&f)[FeatDim<FM>::F], float hp, float r,
                                             float z, float n, float ghn, float hid, float dy0, float dy1, int row,
                                             int col, float& dh, float (&df)[FeatDim<FM>::F]) {
    constexpr int F = FeatDim<FM>::F;
    const float ht = __builtin_fmaf(z, hp - n, n);
    float dht = dh;
    const float g01 = __builtin_fmaf(dy0, w.wout[0], dy1 * w.wout[1]);
    if constexpr (DG) {
        const float a = __builtin_fmaxf(hid, 0.0f);
        const float dhid = hid > 0.0f ? g01 : 0.0f;
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
        {
            const float4* wq = wt.whidT_q;
            dht = rotdot_quads(dht, [wq](int q) { return wq[q * 64]; }, dhid);
            if constexpr (R == 2) dht = rotdot_quads(dht, [wq](int q) { return wq[(4 + q) * 64]; }, swap16(dhid));
        }
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
    rotdot3x(d0, d1, d2, wt.whhT[0][0], wt.whhT[1][0], wt.whhT[2][0], drp, dzp, dgh);
    if constexpr (R == 2)
        rotdot3x(d0, d1, d2, wt.whhT[0][1], wt.whhT[1][1], wt.whhT[2][1], swap16(drp), swap16(dzp), swap16(dgh));
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
// writes ONE row to HBM.  Deterministic: ((w0 + w1) + w2) + w3.
template <int R, int F, bool DG>
__device__ __forceinline__ void gru_block_partials(float* smem, float* partials, const GruLayout& L, GruGrad<R, DG>& G,
                                                   int lane, int wave, int row, int col, float loss_part) {
    const int P4 = L.P + kLossCols;
    __syncthreads();
    gru_write_partials<R, F, DG>(smem + wave * P4, L, G, lane, row, col, loss_part);
    __syncthreads();
    float* prow = partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += kThreads) {
        float v = smem[i];
#pragma unroll
        for (int wv = 1; wv < kWavesPerBlock; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// LDS staging of (B,T,2) streams: one chunk = kChunk steps of the wave's SPW sequences
// -------------------------------------------------------------------------------------------------
static_assert(kChunk == 64, "staging maps lane -> time step inside a chunk");
template <int SPW>
__device__ __forceinline__ void stage_in(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane,
                                         float2 fill) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
#pragma unroll
    for (int m = 0; m < SPW; ++m) {
        float2 v = fill;
        if (lane < len && b0 + m < B) v = g2[(size_t)(b0 + m) * T + t0 + lane];
        lds[m * kChunkPad + lane] = v;
    }
}
template <int SPW>
__device__ __forceinline__ void stage_out(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
#pragma unroll
    for (int m = 0; m < SPW; ++m)
        if (lane < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + lane] = lds[m * kChunkPad + lane];
}

__device__ __forceinline__ void stage_params(float* pl, const float* params, int P) {
    for (int i = threadIdx.x; i < P; i += kThreads) pl[i] = params[i];
    __syncthreads();
}

// -------------------------------------------------------------------------------------------------
// forward kernel
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG>
__global__ __launch_bounds__(kThreads) void gru_fwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R, LPS = 16 * R, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, row = (lane >> 4) & (R - 1), s = lane / LPS;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float2* xs = reinterpret_cast<float2*>(smem + ((L.P + 3) & ~3)) + wave * (2 * SPW * kChunkPad);
    float2* ys = xs + SPW * kChunkPad;
    int src[16];
    rot_sources(src, col);
    GruW<R, F, DG> w;
    load_gru_w<R, F, DG>(w, pl, L, row, col, src);

    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int grp = blockIdx.x * kWavesPerBlock + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float h = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xs[s * kChunkPad + tt];
                float f[F], r, z, n, ghn, y0, y1, hid;
                feat_fwd<FM>(xv.x, xv.y, f);
                gru_cell_fwd<R, FM, DG>(w, f, h, r, z, n, ghn);
                gru_head_fwd<R, FM, DG>(w, h, f, col, y0, y1, hid);
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
// backward over one wave-task (BPTT with block recompute).  Shared by the stand-alone backward
// kernel (dy staged per chunk from HBM, checkpoints in HBM) and the fused train kernel (FUSED: the
// target is staged instead of dy; y, the loss and dy are recomputed here from the block recompute;
// checkpoints live in LDS).
//   xs  : LDS chunk buffer for x        dys : LDS chunk buffer for dy (target if FUSED)
//   ck  : checkpoints of this task, slot c at ck[c*64 + lane]  (HBM, or LDS if FUSED)
// -------------------------------------------------------------------------------------------------
template <int R, int FM, bool DG, bool NW, bool DX, bool FUSED>
__device__ __forceinline__ void gru_bwd_task(const SeqArgs& a, const GruW<R, FeatDim<FM>::F, DG>& w,
                                             const GruWT<R, FeatDim<FM>::F, DG>& wt, GruGrad<R, DG>& G, int b0, int lane,
                                             int row, int col, int s, float2* xs, float2* dys, float2* dxs,
                                             const float* ck, float& loss_acc) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R, LPS = 16 * R, S = kCkptStride;
    float dh = 0.0f;
    int cur_chunk = -1;
    const bool valid = b0 + s < a.B;
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
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            stage_in<SPW>(dys, FUSED ? a.target : a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
            wave_lds_fence();
            cur_chunk = chunk;
        }
        // state at the start of the block, then recompute the block into registers
        float h = blk ? ck[blk * 64 + lane] : 0.0f;
        float hp_s[S], r_s[S], z_s[S], n_s[S], g_s[S], hid_s[S];
#pragma unroll
        for (int i = 0; i < S; ++i) {
            if (i < nstep) {
                const float2 xv = xs[s * kChunkPad + (tb - t0) + i];
                float f[F];
                feat_fwd<FM>(xv.x, xv.y, f);
                hp_s[i] = h;
                gru_cell_fwd<R, FM, DG>(w, f, h, r_s[i], z_s[i], n_s[i], g_s[i]);
                hid_s[i] = 0.0f;
                if constexpr (DG) {
                    const float4* wq = wt.whid_q;
                    hid_s[i] = rotdot_quads(w.bhid, [wq](int q) { return wq[q * 64]; }, h);
                    if constexpr (R == 2)
                        hid_s[i] = rotdot_quads(hid_s[i], [wq](int q) { return wq[(4 + q) * 64]; }, swap16(h));
                }
            }
        }
#pragma unroll
        for (int i = S - 1; i >= 0; --i) {
            if (i < nstep) {
                const int tt = (tb - t0) + i;
                const float2 xv = xs[s * kChunkPad + tt];
                float2 dyv = dys[s * kChunkPad + tt];
                float f[F], df[F];
                feat_fwd<FM>(xv.x, xv.y, f);
                if constexpr (FUSED) {
                    // output head from the recomputed state, then loss and dL/dy (train_funcs.py:35-39)
                    float p0, p1;
                    if constexpr (DG) {
                        const float act = __builtin_fmaxf(hid_s[i], 0.0f), fs = feat_select<6>(f, col, 0.0f);
                        p0 = __builtin_fmaf(w.wout[0], act, w.woutf[0] * fs);
                        p1 = __builtin_fmaf(w.wout[1], act, w.woutf[1] * fs);
                    } else {
                        const float ht = __builtin_fmaf(z_s[i], hp_s[i] - n_s[i], n_s[i]);
                        p0 = w.wout[0] * ht;
                        p1 = w.wout[1] * ht;
                    }
                    const float d0 = seq_sum<R>(p0) + w.bout[0] - dyv.x, d1 = seq_sum<R>(p1) + w.bout[1] - dyv.y;
                    float l;
                    if (a.loss_kind == ODPD_LOSS_L2) {
                        const float sc = 2.0f * a.inv_count;
                        dyv = make_float2(d0 * sc, d1 * sc);
                        l = __builtin_fmaf(d0, d0, d1 * d1);
                    } else {
                        dyv = make_float2(d0 > 0.f ? a.inv_count : (d0 < 0.f ? -a.inv_count : 0.f),
                                          d1 > 0.f ? a.inv_count : (d1 < 0.f ? -a.inv_count : 0.f));
                        l = __builtin_fabsf(d0) + __builtin_fabsf(d1);
                    }
                    if (!valid) dyv = make_float2(0.f, 0.f);
                    loss_acc += (valid && (lane & (LPS - 1)) == 0) ? l : 0.0f;
                }
                gru_step_bwd<R, FM, DG, NW, DX>(w, wt, G, f, hp_s[i], r_s[i], z_s[i], n_s[i], g_s[i], hid_s[i], dyv.x,
                                                dyv.y, row, col, dh, df);
                if constexpr (DX) {
                    float dI, dQ;
                    feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
                    if ((lane & (LPS - 1)) == 0) dxs[s * kChunkPad + tt] = make_float2(dI, dQ);
                }
            }
        }
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

// registers allow two waves per SIMD for one-row models unless both gradient kinds are requested
template <int R, int FM, bool DG, bool NW, bool DX>
__global__ __launch_bounds__(kThreads, (R == 1 && !(NW && DX)) ? 2 : 1) void gru_bwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R, LPS = 16 * R;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, row = (lane >> 4) & (R - 1), s = lane / LPS;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* wl = smem + ((L.P + 3) & ~3);
    float2* xs = reinterpret_cast<float2*>(wl + whidT_lds_floats(R, DG)) + wave * (3 * SPW * kChunkPad);
    float2* dys = xs + SPW * kChunkPad;
    float2* dxs = dys + SPW * kChunkPad;
    int src[16];
    rot_sources(src, col);
    GruW<R, F, DG> w;
    GruWT<R, F, DG> wt;
    load_gru_w<R, F, DG>(w, pl, L, row, col, src);
    load_gru_wT<R, F, DG>(wt, pl, L, row, col, src, wl);
    GruGrad<R, DG> G;
    G.zero();
    float unused = 0.0f;
    const int nwaves = gridDim.x * kWavesPerBlock, wave_global = blockIdx.x * kWavesPerBlock + wave;
    for (int grp = wave_global; grp < a.ngroups; grp += nwaves)
        gru_bwd_task<R, FM, DG, NW, DX, false>(a, w, wt, G, grp * SPW, lane, row, col, s, xs, dys, dxs,
                                               a.ckpt + (size_t)grp * a.nck * 64, unused);
    if constexpr (NW) gru_block_partials<R, F, DG>(smem, a.partials, L, G, lane, wave, row, col, 0.0f);
}

// -------------------------------------------------------------------------------------------------
// fused train kernel: per wave-task  (1) forward pass of the cell only, leaving a checkpoint of h
// every S steps in LDS;  (2) backward pass that recomputes each block, forms y, the loss and dL/dy
// on the fly and back-propagates.  HBM traffic = x (twice, the second read is L2-resident) + target.
// LDS per wave: x chunk, target chunk, checkpoints (nck x 64 floats).
// -------------------------------------------------------------------------------------------------
__host__ __device__ inline int train_wave_floats(int T, int R) {
    const int SPW = 4 / R;
    return 2 * (2 * SPW * kChunkPad) + num_ckpt_hd(T) * 64;
}

template <int R, int FM, bool DG>
__global__ __launch_bounds__(kThreads, R == 1 ? 2 : 1) void gru_train_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, SPW = 4 / R, LPS = 16 * R, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, row = (lane >> 4) & (R - 1), s = lane / LPS;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* wl = smem + ((L.P + 3) & ~3);
    float* wbase = wl + whidT_lds_floats(R, DG) + (size_t)wave * train_wave_floats(a.T, R);
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + SPW * kChunkPad;
    float* ck = reinterpret_cast<float*>(ts + SPW * kChunkPad);
    int src[16];
    rot_sources(src, col);
    GruW<R, F, DG> w;
    GruWT<R, F, DG> wt;
    load_gru_w<R, F, DG>(w, pl, L, row, col, src);
    load_gru_wT<R, F, DG>(wt, pl, L, row, col, src, wl);
    GruGrad<R, DG> G;
    G.zero();
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * kWavesPerBlock, wave_global = blockIdx.x * kWavesPerBlock + wave;
    for (int grp = wave_global; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float h = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xs[s * kChunkPad + tt];
                float f[F], r, z, n, ghn;
                feat_fwd<FM>(xv.x, xv.y, f);
                gru_cell_fwd<R, FM, DG>(w, f, h, r, z, n, ghn);
                const int t1 = t0 + tt + 1;
                if ((t1 % S) == 0 && t1 < a.T) ck[(t1 / S) * 64 + lane] = h;
            }
        }
        wave_lds_fence();
        gru_bwd_task<R, FM, DG, true, false, true>(a, w, wt, G, b0, lane, row, col, s, xs, ts, nullptr, ck, loss_acc);
    }
    gru_block_partials<R, F, DG>(smem, a.partials, L, G, lane, wave, row, col, loss_acc);
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
constexpr int kFwdBlocksPerCU = 4;  // grid caps (blocks are independent: a non-resident block simply queues)
constexpr int kBwdBlocksPerCU = 2;
constexpr int kTrainBlocksPerCU = 2;

static bool gru_cfg(const odpd_model_t* m, int& FM, bool& DG) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; DG = false; return true;
    case ODPD_DGRU: FM = FEAT_DGRU6; DG = true; return true;
    case ODPD_QGRU: FM = FEAT_Q4; DG = false; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; DG = false; return true;
    default: return false;
    }
}
static size_t reduce_scratch_bytes(int P) { return (size_t)kWavesPerBlock * (P + kLossCols) * sizeof(float); }
static size_t gru_lds_bytes(int P, int R, int nbuf, bool dg_bwd = false) {
    size_t n = ((size_t)((P + 3) & ~3) + whidT_lds_floats(R, dg_bwd)) * 4 +
               (size_t)kWavesPerBlock * nbuf * (4 / R) * kChunkPad * sizeof(float2);
    return (dg_bwd || nbuf == 3) && n < reduce_scratch_bytes(P) ? reduce_scratch_bytes(P) : n;
}

template <int R, int FM, bool DG>
static int launch_fwd(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = gru_lds_bytes(P, R, 2);
    hipLaunchKernelGGL((gru_fwd_kernel<R, FM, DG>), dim3(persistent_grid(a.ngroups, kFwdBlocksPerCU)), dim3(kThreads),
                       lds, st, a);
    return (int)hipGetLastError();
}
template <int R, int FM, bool DG, bool NW, bool DX>
static int launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = gru_lds_bytes(P, R, 3, DG);
    hipLaunchKernelGGL((gru_bwd_kernel<R, FM, DG, NW, DX>), dim3(persistent_grid(a.ngroups, kBwdBlocksPerCU)),
                       dim3(kThreads), lds, st, a);
    return (int)hipGetLastError();
}

template <int R, int FM, bool DG>
static int launch_train(hipStream_t st, const SeqArgs& a, int P) {
    size_t lds = ((size_t)((P + 3) & ~3) + whidT_lds_floats(R, DG) +
                  (size_t)kWavesPerBlock * train_wave_floats(a.T, R)) * sizeof(float);
    if (lds < reduce_scratch_bytes(P)) lds = reduce_scratch_bytes(P);
    if (lds > 160 * 1024) return ODPD_EUNSUPPORTED;  // frame too long for LDS-resident BPTT state
    auto k = gru_train_kernel<R, FM, DG>;
    static size_t max_set = 0;
    if (lds > max_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(160 * 1024));
        if (e != hipSuccess) return (int)e;
        max_set = 160 * 1024;
    }
    hipLaunchKernelGGL(k, dim3(persistent_grid(a.ngroups, kTrainBlocksPerCU)), dim3(kThreads), lds, st, a);
    return (int)hipGetLastError();
}

#define ODPD_GRU_DISPATCH(R_, FM_, DG_, CALL)                                                    \
    if (R == R_ && FM == FM_ && DG == DG_) return CALL;

int gru_family_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM; bool DG;
    if (!gru_cfg(m, FM, DG)) return ODPD_EUNSUPPORTED;
    const int R = rows_per_seq(m->hidden);
    if (!R) return ODPD_EUNSUPPORTED;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    ODPD_GRU_DISPATCH(1, FEAT_RAW2, false, (launch_fwd<1, FEAT_RAW2, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_RAW2, false, (launch_fwd<2, FEAT_RAW2, false>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_DGRU6, true, (launch_fwd<1, FEAT_DGRU6, true>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_DGRU6, true, (launch_fwd<2, FEAT_DGRU6, true>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_Q4, false, (launch_fwd<1, FEAT_Q4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_Q4, false, (launch_fwd<2, FEAT_Q4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_A4, false, (launch_fwd<1, FEAT_A4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_A4, false, (launch_fwd<2, FEAT_A4, false>(st, a, P)))
    return ODPD_EUNSUPPORTED;
}

template <int R, int FM, bool DG>
static int launch_bwd_mode(hipStream_t st, const SeqArgs& a, int P) {
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && !dx) return launch_bwd<R, FM, DG, true, false>(st, a, P);
    if (!nw && dx) return launch_bwd<R, FM, DG, false, true>(st, a, P);
    if (nw && dx) return launch_bwd<R, FM, DG, true, true>(st, a, P);
    return ODPD_EINVAL;
}
int gru_family_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM; bool DG;
    if (!gru_cfg(m, FM, DG)) return ODPD_EUNSUPPORTED;
    const int R = rows_per_seq(m->hidden);
    if (!R) return ODPD_EUNSUPPORTED;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    ODPD_GRU_DISPATCH(1, FEAT_RAW2, false, (launch_bwd_mode<1, FEAT_RAW2, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_RAW2, false, (launch_bwd_mode<2, FEAT_RAW2, false>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_DGRU6, true, (launch_bwd_mode<1, FEAT_DGRU6, true>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_DGRU6, true, (launch_bwd_mode<2, FEAT_DGRU6, true>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_Q4, false, (launch_bwd_mode<1, FEAT_Q4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_Q4, false, (launch_bwd_mode<2, FEAT_Q4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_A4, false, (launch_bwd_mode<1, FEAT_A4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_A4, false, (launch_bwd_mode<2, FEAT_A4, false>(st, a, P)))
    return ODPD_EUNSUPPORTED;
}

int gru_family_rows(const odpd_model_t* m, int B, int which) {
    int FM; bool DG;
    if (!gru_cfg(m, FM, DG)) return ODPD_EUNSUPPORTED;
    const int R = rows_per_seq(m->hidden);
    if (!R) return ODPD_EUNSUPPORTED;
    return persistent_grid(num_groups(B, R), which ? kTrainBlocksPerCU : kBwdBlocksPerCU);  // one row per block
}

int gru_family_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM; bool DG;
    if (!gru_cfg(m, FM, DG)) return ODPD_EUNSUPPORTED;
    const int R = rows_per_seq(m->hidden);
    if (!R) return ODPD_EUNSUPPORTED;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    ODPD_GRU_DISPATCH(1, FEAT_RAW2, false, (launch_train<1, FEAT_RAW2, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_RAW2, false, (launch_train<2, FEAT_RAW2, false>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_DGRU6, true, (launch_train<1, FEAT_DGRU6, true>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_DGRU6, true, (launch_train<2, FEAT_DGRU6, true>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_Q4, false, (launch_train<1, FEAT_Q4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_Q4, false, (launch_train<2, FEAT_Q4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(1, FEAT_A4, false, (launch_train<1, FEAT_A4, false>(st, a, P)))
    ODPD_GRU_DISPATCH(2, FEAT_A4, false, (launch_train<2, FEAT_A4, false>(st, a, P)))
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
