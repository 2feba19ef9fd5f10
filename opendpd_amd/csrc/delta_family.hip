// delta_family.hip — persistent-RNN kernels for the delta-network GRU backbones of the reference:
//   deltagru          backbones/deltagru.py:10-276          feat=[I,Q,a,a^3,sin,cos]; nn.GRU weights with biases folded
//                                                            into the accumulators' initial value (deltagru.py:165-170);
//                                                            y = fc_out(h) (+bias)
//   deltagru_tcnskip  backbones/deltagru_tcnskip.py:11-304  ("TRes-DeltaGRU") feat=[I,Q,a,a^3,I_next,Q_next] with
//                                                            next = roll(x,-1) (:88-100); bias-free x2h/h2h, accumulators
//                                                            start at 0; y = fc_out(h) (no bias) + TCN skip
//                                                            (conv 2->3 k3 dil16 pad16, Hardswish, conv 3->2 k1, Hardswish)
// Per step (deltagru.py:235-262):  dx = x - x_p, dh = h - h_p, masked to 0 where |d| < th;  x_p/h_p track the last
// transmitted value;  dm += W_ih dx (+ W_hh dh for r,z),  dm_nh += (W_hh dh)_n;  r = s(dm_r), z = s(dm_z),
// n = tanh(dm_n + r*dm_nh), h = (1-z) n + z h.  Sparsity counters (deltagru.py:241-247) are accumulated on device.
//
// The GPU computes the masked products densely (a 64-lane SIMD cannot skip per-lane zeros); what is kept
// from the delta formulation is the exact arithmetic (thresholded memories and accumulators) and the counters.
// Backward: the gradient through the x_p memory only reaches the INPUT (features carry no parameters), so the
// parameter-gradient path needs the dh/h_p chain only.  dL/dx (delta model as the frozen PA, x.requires_grad) is provided by
// the S16 kernels of delta_s16.hip, which ODPD_FLAG_NEED_DX selects at every batch size.  One 16-lane row per sequence: H <= 16.
#include "odpd_seq.h"
#include "odpd_delta.h"

namespace odpd {

__device__ __forceinline__ void load_rot3d(float (&w)[3][16], TabPtr tlane, int first_row) {
#pragma unroll
    for (int g = 0; g < 3; ++g) load_rot(w[g], tlane + (first_row + g) * 4 * 64);
}

template <bool TRES>
struct DeltaW {
    float wih[3][6];
    float wout[2], bout[2];
    float w1[6];      // TRES: tcn.0.weight[c = min(col,2)][i][k]  (lanes col < 3 are meaningful)
    float w2[2];      // TRES: tcn.2.weight[o][c]
    float dm0[4];     // initial accumulators (biases for deltagru, 0 for TRES): r, z, n, nh
};
template <bool TRES>
__device__ __forceinline__ void load_delta_w(DeltaW<TRES>& w, const float* pl, const DeltaLayout& L, int col) {
    const int H = L.H, o = col;
    const bool vo = o < H;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 6; ++i) w.wih[g][i] = vo ? pl[L.o_w_ih + (g * H + o) * 6 + i] : 0.0f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        w.wout[c] = vo ? pl[L.o_w_out + c * H + o] : 0.0f;
        w.bout[c] = TRES ? 0.0f : __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pl[L.o_b_out + c])));
    }
    const int cc = col < 3 ? col : 2;
#pragma unroll
    for (int k = 0; k < 6; ++k) w.w1[k] = TRES ? pl[L.o_tcn0 + cc * 6 + k] : 0.0f;
#pragma unroll
    for (int oo = 0; oo < 2; ++oo) w.w2[oo] = (TRES && col < 3) ? pl[L.o_tcn2 + oo * 3 + cc] : 0.0f;
    if (TRES) { w.dm0[0] = w.dm0[1] = w.dm0[2] = w.dm0[3] = 0.0f; }
    else {
        w.dm0[0] = vo ? pl[L.o_b_ih + o] + pl[L.o_b_hh + o] : 0.0f;
        w.dm0[1] = vo ? pl[L.o_b_ih + H + o] + pl[L.o_b_hh + H + o] : 0.0f;
        w.dm0[2] = vo ? pl[L.o_b_ih + 2 * H + o] : 0.0f;
        w.dm0[3] = vo ? pl[L.o_b_hh + 2 * H + o] : 0.0f;
    }
}

// recurrent state of one lane (x_p is per sequence, replicated in every lane of the row)
struct DeltaState { float h, hp, dmr, dmz, dmn, dmnh, xp[6]; };

// one forward step.  Outputs what the backward pass needs: hprev, dhm, mh (1/0), r, z, n, dmnh (post-update), and
// fsx = masked dx of feature `col` (MFMA B operand).  zx/zh count exact zeros of the masked deltas.
template <bool TRES>
__device__ __forceinline__ void delta_cell_fwd(const DeltaW<TRES>& w, const float (&whh)[3][16], const float (&f)[6],
                                               float thx, float thh, int col, bool vo, DeltaState& st, float& hprev,
                                               float& dhm, float& mh, float& r, float& z, float& n, float& fsx,
                                               float& zx, float& zh) {
    float ar = st.dmr, az = st.dmz, an = st.dmn;
    fsx = 0.0f;
    float nzx = 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float d = f[i] - st.xp[i];
        const bool keep = !(__builtin_fabsf(d) < thx);       // masked_fill(|d| < th, 0)  (deltagru.py:179-183)
        const float dm = keep ? d : 0.0f;
        st.xp[i] = (__builtin_fabsf(d) >= thx) ? f[i] : st.xp[i];
        nzx += (dm == 0.0f) ? 1.0f : 0.0f;
        fsx = (col == i) ? dm : fsx;
        ar = __builtin_fmaf(w.wih[0][i], dm, ar);
        az = __builtin_fmaf(w.wih[1][i], dm, az);
        an = __builtin_fmaf(w.wih[2][i], dm, an);
    }
    const float dhv = st.h - st.hp;
    const bool keeph = !(__builtin_fabsf(dhv) < thh);
    dhm = keeph ? dhv : 0.0f;
    mh = keeph ? 1.0f : 0.0f;
    st.hp = (__builtin_fabsf(dhv) >= thh) ? st.h : st.hp;
    zx += nzx;
    zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
    float anh = st.dmnh;
    rotdot3(ar, az, anh, whh[0], whh[1], whh[2], dhm);
    st.dmr = ar; st.dmz = az; st.dmn = an; st.dmnh = anh;
    r = sigmoidf_(ar); z = sigmoidf_(az);
    n = tanhf_(__builtin_fmaf(r, anh, an));
    hprev = st.h;
    st.h = __builtin_fmaf(z, st.h - n, n);
}

// TCN skip pre-activations: s1 (lane col<3: channel col), s2[0..1] (all lanes)
template <bool TRES>
__device__ __forceinline__ void tcn_fwd(const DeltaW<TRES>& w, float2 xm, float2 xc, float2 xp16, int col, float& s1,
                                        float& s2a, float& s2b) {
    // weight order tcn.0.weight[c][i][k]: (i=0: k=0,1,2), (i=1: k=0,1,2); taps k=0 -> t-16, k=1 -> t, k=2 -> t+16
    s1 = w.w1[0] * xm.x;
    s1 = __builtin_fmaf(w.w1[1], xc.x, s1); s1 = __builtin_fmaf(w.w1[2], xp16.x, s1);
    s1 = __builtin_fmaf(w.w1[3], xm.y, s1); s1 = __builtin_fmaf(w.w1[4], xc.y, s1); s1 = __builtin_fmaf(w.w1[5], xp16.y, s1);
    const float hs = hardswishf_(s1);
    s2a = row_sum16(w.w2[0] * hs);   // w2 is 0 on lanes col >= 3
    s2b = row_sum16(w.w2[1] * hs);
    (void)col;
}

template <int SPW, int HL>
__device__ __forceinline__ void stage_in_halo2(float2* lds, const float* g, int b0, int B, int T, int t0, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int PER = kChunk + 2 * HL, STR = kChunk + 2 * HL + 1, TOT = SPW * PER, N = (TOT + 63) / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        if (e < TOT) {
            const int m = e / PER, pos = e % PER, tg = t0 - HL + pos;
            // outside the frame: the conv's zero padding.  Inside the frame a sequence beyond the batch
            // (tail group) gets a harmless non-zero dummy so that |x| > 0.
            float2 v = make_float2(0.0f, 0.0f);
            if (tg >= 0 && tg < T) v = (b0 + m < B) ? g2[(size_t)(b0 + m) * T + tg] : make_float2(0.5f, 0.5f);
            lds[m * STR + pos] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// forward kernel
// -------------------------------------------------------------------------------------------------
template <bool TRES>
__global__ __launch_bounds__(kMaxThreads) void delta_fwd_kernel(SeqArgs a) {
    constexpr int SPW = 4, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<1>();
    const int lane = id.lane, col = id.col, s = id.s;
    const DeltaLayout L = delta_layout(a.H, TRES);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_delta_tabs<false>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + kDTabFloats) + id.wave * (SPW * kDStride + SPW * kChunkPad);
    float2* ys = xs + SPW * kDStride;
    DeltaW<TRES> w;
    load_delta_w<TRES>(w, pl, L, col);
    float whh[3][16];
    load_rot3d(whh, tlane, 0);
    const bool vo = col < a.H;
    float zx = 0.0f, zh = 0.0f;
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        const bool valid = b0 + s < a.B;
        DeltaState st;
        st.h = st.hp = 0.0f; st.dmr = w.dm0[0]; st.dmz = w.dm0[1]; st.dmn = w.dm0[2]; st.dmnh = w.dm0[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) st.xp[i] = 0.0f;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[(size_t)(b0 + s) * a.T] : make_float2(0.5f, 0.5f);
        float zxs = 0.0f, zhs = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            stage_in_halo2<SPW, kDHalo>(xs, a.x, b0, a.B, a.T, t0, lane);
            wave_lds_fence();
            const float2* xr = xs + s * kDStride + kDHalo;
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xr[tt];
                const float2 xn = (t0 + tt + 1 < a.T) ? xr[tt + 1] : x0;    // torch.roll(x, -1): last step sees sample 0
                float f[6], hprev, dhm, mh, r, z, n, fsx;
                delta_feat<TRES>(xv, xn, f);
                delta_cell_fwd<TRES>(w, whh, f, a.thx, a.thh, col, vo, st, hprev, dhm, mh, r, z, n, fsx, zxs, zhs);
                float y0 = row_sum16(w.wout[0] * st.h) + w.bout[0], y1 = row_sum16(w.wout[1] * st.h) + w.bout[1];
                if constexpr (TRES) {
                    float s1, s2a, s2b;
                    tcn_fwd<TRES>(w, xr[tt - kDHalo], xv, xr[tt + kDHalo], col, s1, s2a, s2b);
                    y0 += hardswishf_(s2a); y1 += hardswishf_(s2b);
                }
                if (col == 0) ys[s * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (a.ckpt != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    float* ck = a.ckpt + ((size_t)grp * a.nck + t1 / S) * (kDState * 64);
                    ck[lane] = st.h; ck[64 + lane] = st.hp; ck[128 + lane] = st.dmr; ck[192 + lane] = st.dmz;
                    ck[256 + lane] = st.dmn; ck[320 + lane] = st.dmnh;
                    float xpc = 0.0f;
#pragma unroll
                    for (int i = 0; i < 6; ++i) xpc = (col == i) ? st.xp[i] : xpc;
                    ck[384 + lane] = xpc;
                }
            }
            wave_lds_fence();
            stage_out<SPW>(ys, a.y, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
        }
        if (valid) { zx += (col == 0) ? zxs : 0.0f; zh += zhs; }
    }
    if (a.stats != nullptr) {
        // dx zeros are counted once per sequence (lane col 0), dh zeros per hidden unit
        float tx = zx, th = zh;
        for (int o = 32; o > 0; o >>= 1) { tx += __shfl_down(tx, o); th += __shfl_down(th, o); }
        if (lane == 0) {
            atomicAdd(&a.stats[0], (double)tx);
            atomicAdd(&a.stats[2], (double)th);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&a.stats[1], 6.0 * (double)a.B * (double)a.T);
            atomicAdd(&a.stats[3], (double)a.H * (double)a.B * (double)a.T);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): ONE sequence per wave.
//  * gate-parallel: rows 0 / 1 / 2 of the wave accumulate r / z / n with their own rotated W_hh rows — one rotated dot product per
//    step instead of three; r reaches the n row and (z, n) every row through four cross-row swaps, h' is updated redundantly and
//    parked — fc_out of a chunk follows with lane = time step;
//  * everything that does not depend on the state is computed once per 64-step chunk with lane = time step — the six features and
//    the TRes skip (TCN 2->3->2 with its two Hardswish) — and parked in LDS; the raw samples of the next chunk are already in
//    flight while the current one is stepped;
//  * the x-side delta memory is kept one feature per lane (lanes 0..5 of a row: d, mask, x_p in one pass instead of six), the six
//    masked deltas reach the gates as wave-uniform operands (v_readlane).
// Same thresholded arithmetic per element as delta_cell_fwd (accumulation order included); the sparsity counters are kept.
// -------------------------------------------------------------------------------------------------
constexpr int kDEvalHistStride = 64 + 4;
// JAN: deltajanet (backbones/deltajanet.py:229-251) — the same delta formulation with TWO gates, rows f | g | - | -: f = sigmoid(dm_f),
// g = sigmoid(dm_g), h = (1 - f) g + f h; its thresholds are fixed at 0 (deltajanet.py:23-27), fc_out has a bias, no TCN skip.
template <bool TRES, bool CK, bool JAN = false>      // CK: also writes the BPTT checkpoints (the forward of the split train path)
__global__ __launch_bounds__(64) void delta_eval_kernel(SeqArgs a) {
    static_assert(!(JAN && (TRES || CK)), "deltajanet: plain head, no row-rotated backward to write checkpoints for");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;      // r | z | n | -
    const DeltaLayout L = delta_layout(a.H, TRES, JAN ? 2 : 3);
    const int H = L.H, T = a.T;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_delta_tabs<false>(tab, pl, L, lane, 0, 1);
    constexpr int HS = kDEvalHistStride;
    float* feat = tab + kDTabFloats;                                           // [kEvalChunk][8]: f0..f5, skip0, skip1
    float* hist = feat + kEvalChunk * 8;                                       // [kEvalChunk][HS]: h of time t0 + i, every lane's copy
    float* hw = hist + kEvalChunk * HS;                                        // fc_out [2][16], zero padded
    if (lane < 32) hw[lane] = (lane & 15) < L.H ? pl[L.o_w_out + (lane >> 4) * L.H + (lane & 15)] : 0.0f;
    wave_lds_fence();
    const bool vo = col < H, gate_row = role < (JAN ? 2 : 3);
    float wrec[16], wih[6];
    load_rot(wrec, to_tab(reinterpret_cast<const float4*>(tab) + lane) + (gate_row ? role : 0) * 4 * 64);
#pragma unroll
    for (int k = 0; k < 16; ++k) wrec[k] = gate_row ? wrec[k] : 0.0f;
#pragma unroll
    for (int i = 0; i < 6; ++i) wih[i] = (vo && gate_row) ? pl[L.o_w_ih + (role * H + col) * 6 + i] : 0.0f;
    float accx0 = 0.0f, acch0 = 0.0f;
    if (!TRES && vo) {
        if (role < 2) accx0 = pl[L.o_b_ih + role * H + col] + pl[L.o_b_hh + role * H + col];
        if (role == 2 && !JAN) { accx0 = pl[L.o_b_ih + 2 * H + col]; acch0 = pl[L.o_b_hh + 2 * H + col]; }
    }
    const int fc = col < 6 ? col : 5;
    const float bo0 = TRES ? 0.0f : pl[L.o_b_out], bo1 = TRES ? 0.0f : pl[L.o_b_out + 1];
    float w1[3][6], w2[2][3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int k = 0; k < 6; ++k) w1[ch][k] = TRES ? pl[L.o_tcn0 + ch * 6 + k] : 0.0f;
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) w2[oo][ch] = TRES ? pl[L.o_tcn2 + oo * 3 + ch] : 0.0f;
    }
    const float thx = a.thx, thh = a.thh;
    float zx = 0.0f, zh = 0.0f;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float h = 0.0f, hp = 0.0f, xp = 0.0f, accx = accx0, acch = acch0;
        // raw samples of a chunk, lane = time step: x[t], and for TRes x[t + 1] (torch.roll: the last step sees sample 0) and the TCN taps
        float2 rc, rn, rm, rp;
        auto load_raw = [&](int t0) {
            const int t = t0 + lane;
            const float2 zero = make_float2(0.0f, 0.0f);
            rc = t < T ? xg[t] : make_float2(0.5f, 0.5f);
            if constexpr (TRES) {
                rn = t + 1 < T ? xg[t + 1] : xg[0];
                rm = (t - kDHalo >= 0 && t - kDHalo < T) ? xg[t - kDHalo] : zero;
                rp = t + kDHalo < T ? xg[t + kDHalo] : zero;
            }
        };
        load_raw(0);
        for (int t0 = 0; t0 < T; t0 += kEvalChunk) {
            const int len = min(kEvalChunk, T - t0);
            {
                float f[6];
                delta_feat<TRES>(rc, rn, f);
                float sk[2] = {0.0f, 0.0f};
                if constexpr (TRES) {
                    float s2[2] = {0.0f, 0.0f};
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        float s1 = w1[ch][0] * rm.x;
                        s1 = __builtin_fmaf(w1[ch][1], rc.x, s1); s1 = __builtin_fmaf(w1[ch][2], rp.x, s1);
                        s1 = __builtin_fmaf(w1[ch][3], rm.y, s1); s1 = __builtin_fmaf(w1[ch][4], rc.y, s1); s1 = __builtin_fmaf(w1[ch][5], rp.y, s1);
                        const float hs = hardswishf_(s1);
                        s2[0] = __builtin_fmaf(w2[0][ch], hs, s2[0]); s2[1] = __builtin_fmaf(w2[1][ch], hs, s2[1]);
                    }
                    sk[0] = hardswishf_(s2[0]); sk[1] = hardswishf_(s2[1]);
                }
                wave_lds_fence();
                reinterpret_cast<float4*>(feat)[2 * lane] = make_float4(f[0], f[1], f[2], f[3]);
                reinterpret_cast<float4*>(feat)[2 * lane + 1] = make_float4(f[4], f[5], sk[0], sk[1]);
                wave_lds_fence();
            }
            if (t0 + kEvalChunk < T) load_raw(t0 + kEvalChunk);
            for (int tt = 0; tt < len; ++tt) {
                // x side, one feature per lane
                const float fv = feat[tt * 8 + fc];
                const float d = fv - xp, ad = __builtin_fabsf(d);
                const float dm = !(ad < thx) ? d : 0.0f;                          // masked_fill(|d| < th, 0)  (deltagru.py:179-183)
                xp = (ad >= thx) ? fv : xp;
                zx += (dm == 0.0f) ? 1.0f : 0.0f;
                float ax = accx;
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    ax = __builtin_fmaf(wih[i], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dm), i)), ax);
                // h side
                const float dhv = h - hp, adh = __builtin_fabsf(dhv);
                const float dhm = !(adh < thh) ? dhv : 0.0f;
                hp = (adh >= thh) ? h : hp;
                zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
                const bool nrow = !JAN && role == 2;
                const float res = rotdot(nrow ? acch : ax, wrec, dhm);            // rows r, z: dm += W_ih dx + W_hh dh; row n: dm_nh += W_hn dh
                accx = nrow ? ax : res; acch = nrow ? res : acch;
                const float sg = sigmoidf_(res);
                float g4[4];
                if constexpr (JAN) {
                    gather_rows(sg, g4);                                          // f, g on every row
                    h = __builtin_fmaf(g4[0], h - g4[1], g4[1]);                  // (1 - f) g + f h
                } else {
                    const float r = dup32(sg).lo;                                   // row 2 <- r of row 0
                    const float n = tanhf_(__builtin_fmaf(r, res, ax));           // row 2
                    gather_rows(nrow ? n : sg, g4);
                    h = __builtin_fmaf(g4[1], h - g4[2], g4[2]);
                }
                hist[tt * HS + lane] = h;
                if constexpr (CK) {                  // BPTT checkpoints in the layout of the row-rotated backward (lane = 16 s + col, kDState planes)
                    const int t1 = t0 + tt + 1;
                    if ((t1 % kCkptStride) == 0 && t1 < T) {
                        float* ck = a.ckpt + ((size_t)(b >> 2) * a.nck + t1 / kCkptStride) * (kDState * 64) + 16 * (b & 3) + col;
                        if (role == 0) { ck[0] = h; ck[64] = hp; ck[128] = accx; ck[384] = col < 6 ? xp : 0.0f; }
                        if (role == 1) ck[192] = accx;
                        if (role == 2) { ck[256] = accx; ck[320] = acch; }
                    }
                }
            }
            wave_lds_fence();
            // fc_out (+ the skip) of the chunk, lane = time step
            if (lane < len) {
                const float4* hv4 = reinterpret_cast<const float4*>(hist + lane * HS);
                const float4* hw4 = reinterpret_cast<const float4*>(hw);
                float y0 = bo0, y1 = bo1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 hv = hv4[q], w0 = hw4[q], w1 = hw4[4 + q];
                    y0 = __builtin_fmaf(w0.x, hv.x, y0); y0 = __builtin_fmaf(w0.y, hv.y, y0); y0 = __builtin_fmaf(w0.z, hv.z, y0); y0 = __builtin_fmaf(w0.w, hv.w, y0);
                    y1 = __builtin_fmaf(w1.x, hv.x, y1); y1 = __builtin_fmaf(w1.y, hv.y, y1); y1 = __builtin_fmaf(w1.z, hv.z, y1); y1 = __builtin_fmaf(w1.w, hv.w, y1);
                }
                if constexpr (TRES) { y0 += feat[lane * 8 + 6]; y1 += feat[lane * 8 + 7]; }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
    if (a.stats != nullptr) {
        // dx zeros: the six feature lanes of row 0; dh zeros: the hidden units of row 0
        float tx = (role == 0 && col < 6) ? zx : 0.0f, th = role == 0 ? zh : 0.0f;
        for (int o = 32; o > 0; o >>= 1) { tx += __shfl_down(tx, o); th += __shfl_down(th, o); }
        if (lane == 0) {
            atomicAdd(&a.stats[0], (double)tx);
            atomicAdd(&a.stats[2], (double)th);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            atomicAdd(&a.stats[1], 6.0 * (double)a.B * (double)a.T);
            atomicAdd(&a.stats[3], (double)a.H * (double)a.B * (double)a.T);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward kernel (parameter gradients only)
// -------------------------------------------------------------------------------------------------
template <bool TRES>
struct DeltaGrad {
    f32x4 thh[3], tih[3];
    float dwout[2], dbout[2];
    float db[4];          // deltagru: gradient w.r.t. the initial accumulators = bias gradients
    float dw1[6], dw2[2]; // TRES TCN weights (lanes col < 3)
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) { thh[g] = z4; tih[g] = z4; }
        dwout[0] = dwout[1] = dbout[0] = dbout[1] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) db[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) dw1[i] = 0.f;
        dw2[0] = dw2[1] = 0.f;
    }
};
// carried gradients of one lane
struct DeltaCarry { float gh, ghp, gr, gz, gn, gnh; };

template <bool TRES, bool FULL>
__device__ __forceinline__ void delta_bwd_block(const SeqArgs& a, const DeltaW<TRES>& w, TabPtr tlane, DeltaGrad<TRES>& G,
                                                const LaneId& id, const float2* xr, const float2* dys, float2 x0, int tglob,
                                                int tloc, int nstep, DeltaState st, DeltaCarry& C) {
    constexpr int S = kCkptStride;
    const int col = id.col, s = id.s;
    const bool vo = col < a.H;
    float hprev_s[S], dhm_s[S], mh_s[S], r_s[S], z_s[S], n_s[S], nh_s[S], fsx_s[S], ht_s[S];
    tlane = opaque(tlane);
    {
        float whh[3][16];
        load_rot3d(whh, tlane, 0);
        float zx = 0.f, zh = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            if (FULL || i < nstep) {
                const float2 xv = xr[tloc + i];
                const float2 xn = (tglob + i + 1 < a.T) ? xr[tloc + i + 1] : x0;
                float f[6];
                delta_feat<TRES>(xv, xn, f);
                delta_cell_fwd<TRES>(w, whh, f, a.thx, a.thh, col, vo, st, hprev_s[i], dhm_s[i], mh_s[i], r_s[i], z_s[i],
                                     n_s[i], fsx_s[i], zx, zh);
                nh_s[i] = st.dmnh;
                ht_s[i] = st.h;
            }
        }
    }
    tlane = opaque(tlane);
    float whhT[3][16];
    load_rot3d(whhT, tlane, 3);
#pragma unroll
    for (int i = S - 1; i >= 0; --i) {
        if (FULL || i < nstep) {
            const int tt = tloc + i;
            const float2 dyv = dys[s * kChunkPad + tt];
            float gh = C.gh + __builtin_fmaf(dyv.x, w.wout[0], dyv.y * w.wout[1]);
            G.dwout[0] = __builtin_fmaf(dyv.x, ht_s[i], G.dwout[0]);
            G.dwout[1] = __builtin_fmaf(dyv.y, ht_s[i], G.dwout[1]);
            G.dbout[0] += dyv.x; G.dbout[1] += dyv.y;
            if constexpr (TRES) {
                float s1, s2a, s2b;
                const float2 xm = xr[tt - kDHalo], xc = xr[tt], xq = xr[tt + kDHalo];
                tcn_fwd<TRES>(w, xm, xc, xq, col, s1, s2a, s2b);
                const float d2a = dyv.x * hswish_grad_(s2a), d2b = dyv.y * hswish_grad_(s2b);
                const float hs = hardswishf_(s1);
                G.dw2[0] = __builtin_fmaf(d2a, hs, G.dw2[0]);
                G.dw2[1] = __builtin_fmaf(d2b, hs, G.dw2[1]);
                const float d1 = __builtin_fmaf(d2a, w.w2[0], d2b * w.w2[1]) * hswish_grad_(s1);
                G.dw1[0] = __builtin_fmaf(d1, xm.x, G.dw1[0]); G.dw1[1] = __builtin_fmaf(d1, xc.x, G.dw1[1]);
                G.dw1[2] = __builtin_fmaf(d1, xq.x, G.dw1[2]); G.dw1[3] = __builtin_fmaf(d1, xm.y, G.dw1[3]);
                G.dw1[4] = __builtin_fmaf(d1, xc.y, G.dw1[4]); G.dw1[5] = __builtin_fmaf(d1, xq.y, G.dw1[5]);
            }
            const float r = r_s[i], z = z_s[i], n = n_s[i];
            const float dn = gh * (1.0f - z), dz = gh * (hprev_s[i] - n);
            float ghprev = gh * z;
            const float dpre = dn * __builtin_fmaf(-n, n, 1.0f);
            C.gn += dpre;
            C.gnh = __builtin_fmaf(dpre, r, C.gnh);
            C.gr = __builtin_fmaf(dpre * nh_s[i], r * (1.0f - r), C.gr);
            C.gz = __builtin_fmaf(dz, z * (1.0f - z), C.gz);
            // weight gradients: dW_ih += G_dm (x) dx_masked, dW_hh += [G_dm_r, G_dm_z, G_nh] (x) dh_masked
            G.tih[0] = mfma4(C.gr, fsx_s[i], G.tih[0]);
            G.tih[1] = mfma4(C.gz, fsx_s[i], G.tih[1]);
            G.tih[2] = mfma4(C.gn, fsx_s[i], G.tih[2]);
            G.thh[0] = mfma4(C.gr, dhm_s[i], G.thh[0]);
            G.thh[1] = mfma4(C.gz, dhm_s[i], G.thh[1]);
            G.thh[2] = mfma4(C.gnh, dhm_s[i], G.thh[2]);
            // data gradient to the masked dh
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            rotdot3x(d0, d1, d2, whhT[0], whhT[1], whhT[2], C.gr, C.gz, C.gnh);
            const float ddh = d0 + d1 + d2, mk = mh_s[i];
            ghprev = __builtin_fmaf(mk, ddh + C.ghp, ghprev);
            C.ghp = __builtin_fmaf(-mk, ddh, (1.0f - mk) * C.ghp);
            C.gh = ghprev;
        }
    }
}

template <bool TRES>
__device__ __forceinline__ void delta_write_partials(float* prow, const DeltaLayout& L, DeltaGrad<TRES>& G, int lane, int col) {
    const int H = L.H, o = col, seq = lane >> 4, g4 = lane >> 4, c = lane & 15;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * g4 + rr;
            if (i < H) {
                if (c < 6) prow[L.o_w_ih + (g * H + i) * 6 + c] = G.tih[g][rr];
                if (c < H) prow[L.o_w_hh + (g * H + i) * H + c] = G.thh[g][rr];
            }
        }
    const float w0 = across_seqs<1>(G.dwout[0]), w1 = across_seqs<1>(G.dwout[1]);
    if (seq == 0 && o < H) { prow[L.o_w_out + o] = w0; prow[L.o_w_out + H + o] = w1; }
    if constexpr (TRES) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float v = across_seqs<1>(G.dw1[k]);
            if (seq == 0 && col < 3) prow[L.o_tcn0 + col * 6 + k] = v;
        }
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) {
            const float v = across_seqs<1>(G.dw2[oo]);
            if (seq == 0 && col < 3) prow[L.o_tcn2 + oo * 3 + col] = v;
        }
    } else {
        const float b0 = across_seqs<1>(G.dbout[0]), b1 = across_seqs<1>(G.dbout[1]);
        if (lane == 0) { prow[L.o_b_out] = b0; prow[L.o_b_out + 1] = b1; }
        float db[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) db[i] = across_seqs<1>(G.db[i]);
        if (seq == 0 && o < H) {
            prow[L.o_b_ih + o] = db[0]; prow[L.o_b_hh + o] = db[0];
            prow[L.o_b_ih + H + o] = db[1]; prow[L.o_b_hh + H + o] = db[1];
            prow[L.o_b_ih + 2 * H + o] = db[2]; prow[L.o_b_hh + 2 * H + o] = db[3];
        }
    }
}

template <bool TRES>
__global__ __launch_bounds__(kMaxThreads / 2, 1) void delta_bwd_kernel(SeqArgs a) {
    constexpr int SPW = 4, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<1>();
    const int lane = id.lane, col = id.col, s = id.s;
    const DeltaLayout L = delta_layout(a.H, TRES);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_delta_tabs<true>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + kDTabFloats) + id.wave * (SPW * kDStride + SPW * kChunkPad);
    float2* dys = xs + SPW * kDStride;
    DeltaW<TRES> w;
    load_delta_w<TRES>(w, pl, L, col);
    DeltaGrad<TRES> G;
    G.zero();
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        const bool valid = b0 + s < a.B;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[(size_t)(b0 + s) * a.T] : make_float2(0.5f, 0.5f);
        DeltaCarry C = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int cur_chunk = -1;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                stage_in_halo2<SPW, kDHalo>(xs, a.x, b0, a.B, a.T, t0, lane);
                stage_in<SPW>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            DeltaState st;
            if (blk) {
                const float* ck = a.ckpt + ((size_t)grp * a.nck + blk) * (kDState * 64);
                st.h = ck[lane]; st.hp = ck[64 + lane]; st.dmr = ck[128 + lane]; st.dmz = ck[192 + lane];
                st.dmn = ck[256 + lane]; st.dmnh = ck[320 + lane];
#pragma unroll
                for (int i = 0; i < 6; ++i) st.xp[i] = ck[384 + (lane & 48) + i];
            } else {
                st.h = st.hp = 0.0f; st.dmr = w.dm0[0]; st.dmz = w.dm0[1]; st.dmn = w.dm0[2]; st.dmnh = w.dm0[3];
#pragma unroll
                for (int i = 0; i < 6; ++i) st.xp[i] = 0.0f;
            }
            const float2* xr = xs + s * kDStride + kDHalo;
            if (nstep == S) delta_bwd_block<TRES, true>(a, w, tlane, G, id, xr, dys, x0, tb, tb - t0, nstep, st, C);
            else delta_bwd_block<TRES, false>(a, w, tlane, G, id, xr, dys, x0, tb, tb - t0, nstep, st, C);
        }
        // gradient w.r.t. the initial accumulators = bias gradients (deltagru.py:165-170)
        G.db[0] += C.gr; G.db[1] += C.gz; G.db[2] += C.gn; G.db[3] += C.gnh;
    }
    const int P4 = L.P + kLossCols;
    __syncthreads();
    delta_write_partials<TRES>(smem + id.wave * P4, L, G, lane, col);
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = smem[i];
        for (int wv = 1; wv < id.nwb; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// Gate-parallel backward kernel for the reference's own batch sizes (the split train path — train_pa of a delta backbone, the DPD half of a
// train_dpd step — where a wave is alone on its SIMD): ONE sequence per wave (one wave per workgroup), hidden <= 16.  It runs the forward
// pass of delta_eval_kernel again, parking in LDS what the backward step needs of every step — (r, z, n, dm_nh), the masked dh and its mask,
// h(t), the six masked dx — instead of reading checkpoints; dL/dy and the TRes TCN gradient (no state in it) are taken with lane = time
// step; then per step ONE rotated dot product with the transposed weights (rows r | z | n multiply their own accumulator gradient, the
// cross-row sum is the gradient to the masked dh) and TWO 4-block MFMAs: (G_r | G_z | G_nh | -) x masked dh and (G_r | G_z | G_n | -) x masked dx.
// Same arithmetic per element as delta_bwd_block.  One partial-gradient row per workgroup; weight gradients only (dL/dx lives in delta_s16.hip).
// -------------------------------------------------------------------------------------------------
__host__ __device__ inline int delta_gp_buffer_floats(int T) {
    const int Tp = (T + 63) & ~63;
    const int buf = Tp * 8 + (Tp + 1) * 16 + Tp * 64 + Tp * 32 + Tp * 8 + Tp * 2 + 256;
    return buf > kDTabFloats ? buf : kDTabFloats;
}
// JAN: deltajanet, rows f | g | - | - (as delta_eval_kernel<.., JAN>): accumulator gradients G_f += dL/dh (h(t-1) - g) f (1 - f),
// G_g += dL/dh (1 - f) g (1 - g), dL/dh(t-1) = f dL/dh + the delta path.
// FUSED (r04): the whole train_pa step of a delta backbone in this one launch (odpd_train_fwd_bwd) — the forward pass that the kernel
// runs anyway also counts the sparsity statistics (a.stats), and the state-free phase forms y(t) = fc_out h(t) (+ b_out, + the TRes
// skip), the loss and dL/dy itself with lane = time step instead of reading dL/dy: no forward launch, no loss launch, no y / dy round
// trip; frames are addressed in place inside resident streams (a.frame_idx).  Column P of the partial row = the loss partial sum.
template <bool TRES, bool JAN = false, bool FUSED = false>
__global__ __launch_bounds__(64) void delta_gp_bwd_kernel(SeqArgs a) {
    static_assert(!(JAN && TRES), "deltajanet: plain head");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;      // r | z | n | -
    const DeltaLayout L = delta_layout(a.H, TRES, JAN ? 2 : 3);
    const int H = L.H, T = a.T, Tp = (T + 63) & ~63;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_delta_tabs<true>(tab, pl, L, lane, 0, 1);
    const bool vo = col < H, gate_row = role < (JAN ? 2 : 3);
    float wrec[16], wT[16], wih[6];
    {
        TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
        load_rot(wrec, tl + (gate_row ? role : 0) * 4 * 64);
        load_rot(wT, tl + (3 + (gate_row ? role : 0)) * 4 * 64);
#pragma unroll
        for (int k = 0; k < 16; ++k) { wrec[k] = gate_row ? wrec[k] : 0.0f; wT[k] = gate_row ? wT[k] : 0.0f; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) wih[i] = (vo && gate_row) ? pl[L.o_w_ih + (role * H + col) * 6 + i] : 0.0f;
    float accx0 = 0.0f, acch0 = 0.0f;
    if (!TRES && vo) {
        if (role < 2) accx0 = pl[L.o_b_ih + role * H + col] + pl[L.o_b_hh + role * H + col];
        if (role == 2 && !JAN) { accx0 = pl[L.o_b_ih + 2 * H + col]; acch0 = pl[L.o_b_hh + 2 * H + col]; }
    }
    const int fc = col < 6 ? col : 5;
    const float wo0 = vo ? pl[L.o_w_out + col] : 0.0f, wo1 = vo ? pl[L.o_w_out + H + col] : 0.0f;
    float w1[3][6], w2[2][3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int k = 0; k < 6; ++k) w1[ch][k] = TRES ? pl[L.o_tcn0 + ch * 6 + k] : 0.0f;
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) w2[oo][ch] = TRES ? pl[L.o_tcn2 + oo * 3 + ch] : 0.0f;
    }
    const float thx = a.thx, thh = a.thh;
    wave_lds_fence();
    // per-time buffers over the tables
    float* feat = tab;                                  // [Tp][8]   f0..f5 of step t
    float* hist = feat + Tp * 8;                        // [Tp + 1][16]   entry t + 1 = h(t), entry 0 = 0
    float* gpk = hist + (Tp + 1) * 16;                  // [Tp][16][4]   r, z, n, dm_nh of step t
    float* dm2 = gpk + Tp * 64;                         // [Tp][16][2]   masked dh, its mask (1 / 0)
    float* dmx = dm2 + Tp * 32;                         // [Tp][8]   the six masked dx
    float* dyb = dmx + Tp * 8;                          // [Tp][2]   dL/dy(t)
    float* dump = dyb + Tp * 2;                         // [256]
    if (lane < 16) hist[lane] = 0.0f;
    const RowMasks rm = row_masks();
    const int dmp = (int)(dump - smem);
    // per-step stores of the forward pass: row 2 parks (r, z, n, dm_nh), row 1 (masked dh, mask), row 3 h(t), row 0 the masked dx of its lanes 0..7
    const int p4_0 = role == 2 ? (int)(gpk - smem) + 4 * col : dmp + 4 * lane, p4_step = role == 2 ? 64 : 0;
    const int p2_0 = role == 1 ? (int)(dm2 - smem) + 2 * col : dmp + 2 * lane, p2_step = role == 1 ? 32 : 0;
    const int p1_0 = role == 3 ? (int)(hist - smem) + 16 + col : (role == 0 && col < 8) ? (int)(dmx - smem) + col : dmp + lane;
    const int p1_step = role == 3 ? 16 : (role == 0 && col < 8) ? 8 : 0;

    f32x16 acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
    float dwo0 = 0.0f, dwo1 = 0.0f, dbo0 = 0.0f, dbo1 = 0.0f, dbg = 0.0f, dbn = 0.0f, tw1[3][6], tw2[2][3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int k = 0; k < 6; ++k) tw1[ch][k] = 0.0f;
        tw2[0][ch] = 0.0f; tw2[1][ch] = 0.0f;
    }

    float zx = 0.0f, zh = 0.0f, loss_acc = 0.0f;
    const float bo0 = (FUSED && !TRES) ? pl[L.o_b_out] : 0.0f, bo1 = (FUSED && !TRES) ? pl[L.o_b_out + 1] : 0.0f;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = (FUSED && a.frame_idx) ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* dyg = reinterpret_cast<const float2*>(FUSED ? a.target : a.dy) + base;      // FUSED: the target
        // ---- forward, as delta_eval_kernel ----
        {
            float h = 0.0f, hp = 0.0f, xp = 0.0f, accx = accx0, acch = acch0;
            int q4 = p4_0, q2 = p2_0, q1 = p1_0;
            float2 rc, rn = make_float2(0.0f, 0.0f);
            auto load_raw = [&](int t0) {
                const int t = t0 + lane;
                rc = t < T ? xg[t] : make_float2(0.5f, 0.5f);
                if constexpr (TRES) rn = t + 1 < T ? xg[t + 1] : xg[0];
            };
            load_raw(0);
            for (int t0 = 0; t0 < T; t0 += kEvalChunk) {
                const int len = min(kEvalChunk, T - t0);
                {
                    float f[6];
                    delta_feat<TRES>(rc, rn, f);
                    wave_lds_fence();
                    reinterpret_cast<float4*>(feat)[2 * (t0 + lane)] = make_float4(f[0], f[1], f[2], f[3]);
                    reinterpret_cast<float4*>(feat)[2 * (t0 + lane) + 1] = make_float4(f[4], f[5], 0.0f, 0.0f);
                    wave_lds_fence();
                }
                if (t0 + kEvalChunk < T) load_raw(t0 + kEvalChunk);
                for (int tt = 0; tt < len; ++tt) {
                    const float fv = feat[(t0 + tt) * 8 + fc];
                    const float d = fv - xp, ad = __builtin_fabsf(d);
                    const float dm = !(ad < thx) ? d : 0.0f;
                    xp = (ad >= thx) ? fv : xp;
                    if constexpr (FUSED) zx += (dm == 0.0f) ? 1.0f : 0.0f;
                    float ax = accx;
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        ax = __builtin_fmaf(wih[i], __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dm), i)), ax);
                    const float dhv = h - hp, adh = __builtin_fabsf(dhv);
                    const bool keeph = !(adh < thh);
                    const float dhm = keeph ? dhv : 0.0f;
                    hp = (adh >= thh) ? h : hp;
                    if constexpr (FUSED) zh += (vo && dhm == 0.0f) ? 1.0f : 0.0f;
                    const bool nrow = !JAN && role == 2;
                    const float res = rotdot(nrow ? acch : ax, wrec, dhm);
                    accx = nrow ? ax : res; acch = nrow ? res : acch;
                    const float sg = sigmoidf_(res);
                    float g4[4];
                    if constexpr (JAN) {
                        gather_rows(sg, g4);
                        h = __builtin_fmaf(g4[0], h - g4[1], g4[1]);
                        *reinterpret_cast<float4*>(smem + q4) = make_float4(g4[0], g4[1], 0.0f, 0.0f);       // f, g
                    } else {
                        const float r = dup32(sg).lo;                                   // row 2 <- r of row 0
                        const float n = tanhf_(__builtin_fmaf(r, res, ax));           // row 2
                        gather_rows(nrow ? n : sg, g4);
                        h = __builtin_fmaf(g4[1], h - g4[2], g4[2]);
                        *reinterpret_cast<float4*>(smem + q4) = make_float4(r, g4[1], g4[2], res);
                    }
                    *reinterpret_cast<float2*>(smem + q2) = make_float2(dhm, keeph ? 1.0f : 0.0f);
                    smem[q1] = role == 3 ? h : dm;
                    q4 += p4_step; q2 += p2_step; q1 += p1_step;
                }
            }
            wave_lds_fence();
        }
        // ---- dL/dy of every step and the TCN gradient (state-free), lane = time step ----
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
                float2 dyv = dyg[t];                                   // dL/dy(t) — FUSED: the target of step t for now
                float2 xc, xm, xq;
                float s1[3], hs[3], s2a = 0.0f, s2b = 0.0f;
                if constexpr (TRES) {
                    const float2 zero = make_float2(0.0f, 0.0f);
                    xc = xg[t]; xm = t - kDHalo >= 0 ? xg[t - kDHalo] : zero; xq = t + kDHalo < T ? xg[t + kDHalo] : zero;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        s1[ch] = w1[ch][0] * xm.x;
                        s1[ch] = __builtin_fmaf(w1[ch][1], xc.x, s1[ch]); s1[ch] = __builtin_fmaf(w1[ch][2], xq.x, s1[ch]);
                        s1[ch] = __builtin_fmaf(w1[ch][3], xm.y, s1[ch]); s1[ch] = __builtin_fmaf(w1[ch][4], xc.y, s1[ch]);
                        s1[ch] = __builtin_fmaf(w1[ch][5], xq.y, s1[ch]);
                        hs[ch] = hardswishf_(s1[ch]);
                        s2a = __builtin_fmaf(w2[0][ch], hs[ch], s2a); s2b = __builtin_fmaf(w2[1][ch], hs[ch], s2b);
                    }
                }
                if constexpr (FUSED) {
                    // y(t) = fc_out h(t) (+ bias | + the skip), in delta_eval_kernel's order; then the loss term and dL/dy of the step
                    float y0 = bo0, y1 = bo1;
                    const float* hrow = hist + (t + 1) * 16;
                    for (int c = 0; c < H; ++c) {
                        const float hv = hrow[c];
                        y0 = __builtin_fmaf(pl[L.o_w_out + c], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + H + c], hv, y1);
                    }
                    if constexpr (TRES) { y0 += hardswishf_(s2a); y1 += hardswishf_(s2b); }
                    const float d0 = y0 - dyv.x, d1 = y1 - dyv.y;
                    if (a.loss_kind == ODPD_LOSS_L2) {
                        loss_acc += __builtin_fmaf(d0, d0, d1 * d1);
                        dyv = make_float2(2.0f * a.inv_count * d0, 2.0f * a.inv_count * d1);
                    } else {
                        loss_acc += __builtin_fabsf(d0) + __builtin_fabsf(d1);
                        dyv = make_float2(d0 > 0.0f ? a.inv_count : (d0 < 0.0f ? -a.inv_count : 0.0f), d1 > 0.0f ? a.inv_count : (d1 < 0.0f ? -a.inv_count : 0.0f));
                    }
                }
                *reinterpret_cast<float2*>(dyb + 2 * t) = dyv;
                dbo0 += dyv.x; dbo1 += dyv.y;
                if constexpr (TRES) {
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
        }
        wave_lds_fence();
        // ---- backward ----
        {
            float gh_c = 0.0f, ghp = 0.0f, accg = 0.0f, gn = 0.0f;      // carried: dL/dh, dL/dh_p, the row's accumulator gradient (G_r | G_z | G_nh), G_n
            for (int t = T - 1; t >= 0; --t) {
                const float hprev = hist[t * 16 + col], ht = hist[(t + 1) * 16 + col];
                const float4 g = reinterpret_cast<const float4*>(gpk)[t * 16 + col];
                const float2 dd = reinterpret_cast<const float2*>(dm2)[t * 16 + col];
                const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
                const float fsx = col < 6 ? dmx[t * 8 + col] : 0.0f;
                const float dhm = dd.x, mk = dd.y;
                const float gh = gh_c + __builtin_fmaf(dyv.x, wo0, dyv.y * wo1);
                dwo0 = __builtin_fmaf(dyv.x, ht, dwo0); dwo1 = __builtin_fmaf(dyv.y, ht, dwo1);
                float ghprev;
                if constexpr (JAN) {
                    const float f = g.x, gg = g.y;
                    const float dg = gh * (1.0f - f), df = gh * (hprev - gg);
                    ghprev = gh * f;
                    const float c_f = __builtin_fmaf(df, f * (1.0f - f), accg), c_g = __builtin_fmaf(dg, gg * (1.0f - gg), accg);
                    accg = vsel(rm.m[0], c_f, vsel(rm.m[1], c_g, 0.0f));
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(accg, dhm, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(accg, fsx, acc2, 0, 0, 0);
                } else {
                    const float r = g.x, z = g.y, n = g.z, nh = g.w;
                    const float dn = gh * (1.0f - z), dz = gh * (hprev - n);
                    ghprev = gh * z;
                    const float dpre = dn * __builtin_fmaf(-n, n, 1.0f);
                    gn += dpre;
                    const float c_nh = __builtin_fmaf(dpre, r, accg), c_r = __builtin_fmaf(dpre * nh, r * (1.0f - r), accg),
                                c_z = __builtin_fmaf(dz, z * (1.0f - z), accg);
                    accg = vsel(rm.m[0], c_r, vsel(rm.m[1], c_z, c_nh));
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(accg, dhm, acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(vsel(rm.m[2], gn, accg), fsx, acc2, 0, 0, 0);
                }
                float ddh = rotdot(0.0f, wT, accg);
                ddh = sum_rows4(ddh);
                ghprev = __builtin_fmaf(mk, ddh + ghp, ghprev);
                ghp = __builtin_fmaf(-mk, ddh, (1.0f - mk) * ghp);
                gh_c = ghprev;
            }
            // gradient w.r.t. the initial accumulators = bias gradients (deltagru.py:165-170)
            dbg += accg; dbn += gn;
        }
        wave_lds_fence();
    }
    // ---- the workgroup's row of partial gradients (every entry written) ----
    float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
    {
        const float lp = FUSED ? wave_sum_(loss_acc) : 0.0f;
        if (lane < kLossCols) prow[L.P + lane] = lane == 0 ? lp : 0.0f;
    }
    if constexpr (FUSED) {
        if (a.stats != nullptr) {      // as delta_eval_kernel: dx zeros on the six feature lanes of row 0, dh zeros on the hidden units of row 0
            float tx = (role == 0 && col < 6) ? zx : 0.0f, th = role == 0 ? zh : 0.0f;
            for (int o = 32; o > 0; o >>= 1) { tx += __shfl_down(tx, o); th += __shfl_down(th, o); }
            if (lane == 0) {
                atomicAdd(&a.stats[0], (double)tx);
                atomicAdd(&a.stats[2], (double)th);
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) {
                atomicAdd(&a.stats[1], 6.0 * (double)a.B * (double)a.T);
                atomicAdd(&a.stats[3], (double)a.H * (double)a.B * (double)a.T);
            }
        }
    }
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
            if (role == 2 && !JAN) { prow[L.o_b_ih + 2 * H + col] = dbn; prow[L.o_b_hh + 2 * H + col] = dbg; }
        }
    }
    // MFMA block k = gate k (r, z, n); register 4 k + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
    for (int k = 0; k < (JAN ? 2 : 3); ++k)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * role + rr;
            if (i < H) {
                if (col < H) prow[L.o_w_hh + (k * H + i) * H + col] = acc1[4 * k + rr];
                if (col < 6) prow[L.o_w_ih + (k * H + i) * 6 + col] = acc2[4 * k + rr];
            }
        }
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
static size_t delta_lds_bytes(int P, int waves, bool reduce) {
    size_t n = ((size_t)pad4(P) + kDTabFloats + (size_t)waves * 2 * (4 * kDStride + 4 * kChunkPad)) * sizeof(float);
    if (reduce && n < reduce_scratch_bytes(P, waves)) n = reduce_scratch_bytes(P, waves);
    return n;
}
static LaunchShape delta_bwd_shape(int ngroups) { return persistent_shape(ngroups, 4, 4); }

template <bool TRES>
static int delta_launch_fwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = persistent_shape(a.ngroups, 8);
    const size_t lds = delta_lds_bytes(P, ls.waves, false);
    auto k = delta_fwd_kernel<TRES>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
template <bool TRES>
static int delta_launch_eval(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = ((size_t)pad4(P) + kDTabFloats + kEvalChunk * 8 + kEvalChunk * kDEvalHistStride + 32) * sizeof(float);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(a.B), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    return a.ckpt ? launch(delta_eval_kernel<TRES, true>) : launch(delta_eval_kernel<TRES, false>);
}
template <bool TRES>
static int delta_launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    if (a.dx != nullptr) return ODPD_EUNSUPPORTED;   // dL/dx of a delta backbone is not implemented
    if (a.partials == nullptr) return ODPD_EINVAL;
    const LaunchShape ls = delta_bwd_shape(a.ngroups);
    const size_t lds = delta_lds_bytes(P, ls.waves, true);
    auto k = delta_bwd_kernel<TRES>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}

// the gate-parallel backward kernel: one sequence per single-wave workgroup, the frame's parked state in LDS
static size_t delta_gp_lds_bytes(int P, int T) { return ((size_t)pad4(P) + delta_gp_buffer_floats(T)) * sizeof(float); }
static int delta_gp_blocks_per_cu(int P, int T) {
    const size_t lds = delta_gp_lds_bytes(P, T);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}
static bool delta_bwd_uses_gp(const odpd_model_t* m, int B, int T) {
    if (m->hidden > 16 || delta_uses_s16(m, B) || tuning().s16_min_batch == 0) return false;
    const int P = delta_layout(m->hidden, m->backbone == ODPD_TRES_DELTAGRU).P;
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && delta_gp_blocks_per_cu(P, T) > 0;
    return (long)B <= (long)device_cus() * delta_gp_blocks_per_cu(P, T);      // every sequence resident at once
}
static int delta_gp_rows(const odpd_model_t* m, int B, int T) {
    const int P = delta_layout(m->hidden, m->backbone == ODPD_TRES_DELTAGRU).P;
    const long cap = (long)device_cus() * (kMaxLds / delta_gp_lds_bytes(P, T));
    return B < cap ? B : (int)cap;
}
template <bool TRES>
static int delta_launch_gp_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int P) {
    const size_t lds = delta_gp_lds_bytes(P, a.T);
    auto k = delta_gp_bwd_kernel<TRES>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(delta_gp_rows(m, a.B, a.T)), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}

// deltajanet's home is delta_s16.hip (every batch, dL/dx, hidden <= 32); at the reference's own batch sizes — every sequence on a SIMD of its
// own — hidden <= 16 without dL/dx takes the one-sequence-per-wave kernels: delta_eval_kernel<.., JAN> as the forward (it writes no
// checkpoints: the backward below runs the forward again) and delta_gp_bwd_kernel<.., JAN>
static int jan_P(const odpd_model_t* m) { return delta_layout(m->hidden, 0, 2).P; }
static bool jan_gp_ok(const odpd_model_t* m) {
    return m->backbone == ODPD_DELTAJANET && m->hidden <= 16 && !(m->flags & ODPD_FLAG_NEED_DX) && tuning().s16_min_batch != 0 &&
           tuning().gp_max_batch != 0;
}
static bool jan_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (!jan_gp_ok(m)) return false;
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && delta_gp_blocks_per_cu(jan_P(m), T) > 0;
    // up to three frames per workgroup in turn: still ahead of the 16-sequence waves there (768 x 200: 0.31 vs 0.44 ms, profiles/r03/gp_train_bench.txt)
    return (long)B <= 3L * device_cus() * delta_gp_blocks_per_cu(jan_P(m), T);
}
static int jan_gp_rows(const odpd_model_t* m, int B, int T) {
    const long cap = (long)device_cus() * (kMaxLds / delta_gp_lds_bytes(jan_P(m), T));
    return B < cap ? B : (int)cap;
}
static int jan_launch_eval(hipStream_t st, const SeqArgs& a, int P) {
    const size_t lds = ((size_t)pad4(P) + kDTabFloats + kEvalChunk * 8 + kEvalChunk * kDEvalHistStride + 32) * sizeof(float);
    auto k = delta_eval_kernel<false, false, true>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
// ---- fused train step at the reference's batch sizes (one frame per single-wave workgroup): delta_gp_bwd_kernel<.., FUSED> ----
bool delta_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (m->flags & ODPD_FLAG_NEED_DX) return false;
    if (m->backbone == ODPD_DELTAJANET) return jan_train_uses_gp(m, B, T);
    if (m->backbone != ODPD_DELTAGRU && m->backbone != ODPD_TRES_DELTAGRU) return false;
    return delta_bwd_uses_gp(m, B, T);
}
int delta_gp_train_rows(const odpd_model_t* m, int B, int T) {
    return m->backbone == ODPD_DELTAJANET ? jan_gp_rows(m, B, T) : delta_gp_rows(m, B, T);
}
// a.stats: the four sparsity counters of the step's forward pass (nullable); a.x / a.target: (B,T,2) tensors or frames of resident streams
int delta_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!delta_train_uses_gp(m, a.B, a.T)) return ODPD_EUNSUPPORTED;
    const bool jan = m->backbone == ODPD_DELTAJANET, tres = m->backbone == ODPD_TRES_DELTAGRU;
    const int P = jan ? jan_P(m) : delta_layout(m->hidden, tres).P;
    const size_t lds = delta_gp_lds_bytes(P, a.T);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(delta_gp_train_rows(m, a.B, a.T)), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    };
    if (jan) return launch(delta_gp_bwd_kernel<false, true, true>);
    return tres ? launch(delta_gp_bwd_kernel<true, false, true>) : launch(delta_gp_bwd_kernel<false, false, true>);
}
int delta_family_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (jan_gp_ok(m) && ((a.ckpt == nullptr && a.B <= 2 * device_cus()) || (a.ckpt != nullptr && jan_train_uses_gp(m, a.B, a.T))))
        return jan_launch_eval(st, a, jan_P(m));
    if (delta_uses_s16(m, a.B)) return delta_s16_launch(st, m, a, 1);
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    const bool tres = m->backbone == ODPD_TRES_DELTAGRU;
    const int P = delta_layout(m->hidden, tres).P;
    if (a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0)          // sequences that each get a SIMD of their own (inference / checkpoint-writing forward)
        return tres ? delta_launch_eval<true>(st, a, P) : delta_launch_eval<false>(st, a, P);
    return tres ? delta_launch_fwd<true>(st, a, P) : delta_launch_fwd<false>(st, a, P);
}
int delta_family_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (a.dx == nullptr && a.partials != nullptr && jan_train_uses_gp(m, a.B, a.T)) {
        const int P = jan_P(m);
        const size_t lds = delta_gp_lds_bytes(P, a.T);
        auto k = delta_gp_bwd_kernel<false, true>;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(jan_gp_rows(m, a.B, a.T)), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    }
    if (delta_uses_s16(m, a.B)) return delta_s16_launch(st, m, a, 2);
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    const bool tres = m->backbone == ODPD_TRES_DELTAGRU;
    const int P = delta_layout(m->hidden, tres).P;
    if (a.dx == nullptr && a.partials != nullptr && delta_bwd_uses_gp(m, a.B, a.T))
        return tres ? delta_launch_gp_bwd<true>(st, m, a, P) : delta_launch_gp_bwd<false>(st, m, a, P);
    return tres ? delta_launch_bwd<true>(st, a, P) : delta_launch_bwd<false>(st, a, P);
}
int delta_family_rows(const odpd_model_t* m, int B, int T) {
    if (jan_train_uses_gp(m, B, T)) return jan_gp_rows(m, B, T);
    if (delta_uses_s16(m, B)) return delta_s16_rows(m, B);
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    if (delta_bwd_uses_gp(m, B, T)) return delta_gp_rows(m, B, T);
    return delta_bwd_shape(num_groups(B, 1)).grid;
}

}  // namespace odpd
