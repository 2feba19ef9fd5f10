// odpd_lstm.h — pieces shared by the LSTM translation units (lstm_family.hip, gru_cascade.hip): the LDS rotated-quad weight tables, and the
// plain LSTM (backbones/lstm.py) as the trained DPD of a cascade on one wave (LstmSeq).
#pragma once
#include "odpd_s16.h"

namespace odpd {

// table rows: kHH = g*R + rb (g = 0..3), kHHT = 4R + g*R + rb
template <int R> struct LstmTabs {
    static constexpr int kHH = 0, kHHT = 4 * R, kRows = 8 * R, kFloats = kRows * 4 * 64 * 4;
};
template <int R, bool WITH_T>
__device__ __forceinline__ void fill_lstm_tabs(float* tab, const float* pl, const LstmLayout& L, int lane, int wave, int nwb) {
    using T = LstmTabs<R>;
    const int H = L.H, col = lane & 15, row = (lane >> 4) & (R - 1), o = 16 * row + col, dir = rot_dir(col);
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int idx = wave; idx < T::kRows * 4; idx += nwb) {
        const int tr = idx >> 2, q = idx & 3;
        const bool transposed = tr >= T::kHHT;
        if (!WITH_T && transposed) continue;
        const int local = transposed ? tr - T::kHHT : tr, g = local / R, rb = local % R;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = 16 * ((row + rb) % R) + ((col + dir * (4 * q + e)) & 15);
            const bool ok = o < H && m < H;
            v[e] = ok ? pl[L.o_w_hh + g * H * H + (transposed ? m * H + o : o * H + m)] : 0.0f;
        }
        t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
}

// The step arithmetic of lstm_gp_train_kernel<false, false> (lstm_family.hip: row k of the wave = gate k — i | f | g | o —, one rotated dot
// product per step and orientation, h(t), c(t), tanh c(t) of the frame parked in LDS, gates recomputed from the parked h(t-1) in the
// backward steps, weight gradients as two 4-block MFMAs) restated chunk-wise for the cascade kernel (gru_cascade.hip).  hidden <= 16.
struct LstmSeq {
    static constexpr int F = 2;
    using TB = LstmTabs<1>;
    __host__ __device__ static int tp(int T) { return (T + 63) & ~63; }
    __host__ __device__ static int off_hist(int T) { return tp(T) * 2; }                       // xb [Tp][2]: I, Q of step t
    __host__ __device__ static int off_cpk(int T) { return off_hist(T) + (T + 2) * 16; }       // hist [T + 2][16]: entry t + 1 = h(t), entry 0 = 0
    __host__ __device__ static int off_tpk(int T) { return off_cpk(T) + (T + 2) * 16; }        // cpk: c(t)
    __host__ __device__ static int off_dyb(int T) { return off_tpk(T) + (T + 2) * 16; }        // tpk: tanh c(t)
    __host__ __device__ static int off_dump(int T) { return off_dyb(T) + tp(T) * 2; }          // dyb [Tp][2]: dL/du(t), written by the PA wave
    __host__ __device__ static int off_hw(int T) { return off_dump(T) + 256; }
    __host__ __device__ static int buf_floats(int T) { return off_hw(T) + 32; }
    __host__ __device__ static int region_floats(int T, int P) {
        const int buf = buf_floats(T), tabf = TB::kFloats;
        return pad4(P) + (buf > tabf ? buf : tabf);
    }

    float wF[16], wT[16], win[F], bg, wh[2], bo0, bo1;
    float h, c, dh, dc;
    f32x16 acc1, acc2;
    float dwh[2], dbo0, dbo1;
    float *smem, *pl, *xb, *hist, *cpk, *tpk, *dyb, *dump, *hw;
    LstmLayout L;
    int H, T, lane, col, role, park0, park_step, park;
    bool vo, is_g;

    // (one workgroup barrier inside: fill_lstm_tabs)
    __device__ __forceinline__ void setup(float* base, float* region, const float* params, int Hm, int T_) {
        smem = base;
        lane = threadIdx.x & 63; col = lane & 15; role = lane >> 4;      // gate i | f | g | o
        L = lstm_layout(Hm, 0);
        H = L.H; T = T_;
        pl = region;
        for (int i = lane; i < L.P; i += 64) pl[i] = params[i];
        wave_lds_fence();
        float* tab = region + pad4(L.P);
        fill_lstm_tabs<1, true>(tab, pl, L, lane, 0, 1);
        vo = col < H; is_g = role == 2;
        {
            TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + col);
            load_rot(wF, tl + (TB::kHH + role) * 4 * 64);
            load_rot(wT, tl + (TB::kHHT + role) * 4 * 64);
        }
#pragma unroll
        for (int i = 0; i < F; ++i) win[i] = vo ? pl[L.o_w_ih + (role * H + col) * F + i] : 0.0f;
        bg = vo ? pl[L.o_b_ih + role * H + col] + pl[L.o_b_hh + role * H + col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 2; ++r) wh[r] = vo ? pl[L.o_w_out + r * H + col] : 0.0f;
        bo0 = pl[L.o_b_out]; bo1 = pl[L.o_b_out + 1];
        wave_lds_fence();
        xb = tab; hist = tab + off_hist(T); cpk = tab + off_cpk(T); tpk = tab + off_tpk(T); dyb = tab + off_dyb(T); dump = tab + off_dump(T);
        hw = tab + off_hw(T);
        if (lane < 32) hw[lane] = (lane & 15) < H ? pl[L.o_w_out + (lane >> 4) * H + (lane & 15)] : 0.0f;
        if (lane < 16) { hist[lane] = 0.0f; cpk[lane] = 0.0f; tpk[lane] = 0.0f; }
        // the per-step stores of the forward pass: row 1 parks h, row 2 c, row 3 tanh c (row 0 hits the dump)
        park0 = role == 0 ? (int)(dump - smem) + lane : (int)((role == 1 ? hist : role == 2 ? cpk : tpk) - smem) + 16 + col;
        park_step = role == 0 ? 0 : 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
        dwh[0] = 0.0f; dwh[1] = 0.0f; dbo0 = 0.0f; dbo1 = 0.0f;
        wave_lds_fence();
    }
    __device__ __forceinline__ void gates(int t, float hin, float (&g)[4]) const {
        const float2 xv = *reinterpret_cast<const float2*>(xb + 2 * t);
        float acc = __builtin_fmaf(win[1], xv.y, __builtin_fmaf(win[0], xv.x, bg));
        acc = rotdot(acc, wF, hin);
        const float sg = sigmoidf_(acc), th = tanhf_(acc);
        gather_rows(is_g ? th : sg, g);
    }
    __device__ __forceinline__ void fwd_begin() { h = 0.0f; c = 0.0f; park = park0; }
    // forward chunk: inputs with lane = time step, the recurrence, then fc_out of the chunk with lane = time step: sink(t, u0, u1)
    template <typename Sink>
    __device__ __forceinline__ void fwd_chunk(int t0, int len, const float2* xg, Sink sink) {
        if (lane < len) *reinterpret_cast<float2*>(xb + 2 * (t0 + lane)) = xg[t0 + lane];
        wave_lds_fence();
        for (int tt = 0; tt < len; ++tt) {
            float g[4];
            gates(t0 + tt, h, g);
            c = __builtin_fmaf(g[1], c, g[0] * g[2]);
            const float tc = tanhf_(c);
            h = g[3] * tc;
            smem[park] = role == 2 ? c : role == 3 ? tc : h;
            park += park_step;
        }
        wave_lds_fence();
        if (lane < len) {
            const int t = t0 + lane;
            const float4* hv4 = reinterpret_cast<const float4*>(hist + (t + 1) * 16);
            const float4* hw4 = reinterpret_cast<const float4*>(hw);
            float y0 = bo0, y1 = bo1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 hv = hv4[q], a0 = hw4[q], a1 = hw4[4 + q];
                y0 = __builtin_fmaf(a0.x, hv.x, y0); y0 = __builtin_fmaf(a0.y, hv.y, y0); y0 = __builtin_fmaf(a0.z, hv.z, y0); y0 = __builtin_fmaf(a0.w, hv.w, y0);
                y1 = __builtin_fmaf(a1.x, hv.x, y1); y1 = __builtin_fmaf(a1.y, hv.y, y1); y1 = __builtin_fmaf(a1.z, hv.z, y1); y1 = __builtin_fmaf(a1.w, hv.w, y1);
            }
            sink(t, y0, y1);
        }
    }
    __device__ __forceinline__ void bwd_begin() { dh = 0.0f; dc = 0.0f; }
    // backward steps hi .. lo (descending), dL/du of those steps in dyb
    __device__ __forceinline__ void bwd_steps(int hi, int lo) {
        for (int t = hi; t >= lo; --t) {
            const float hp = hist[t * 16 + col], ht = hist[(t + 1) * 16 + col];
            const float cp = cpk[t * 16 + col], tc = tpk[(t + 1) * 16 + col];
            const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
            float g[4];
            gates(t, hp, g);
            float dht = dh;
            dht = __builtin_fmaf(dyv.x, wh[0], dht); dwh[0] = __builtin_fmaf(dyv.x, ht, dwh[0]);
            dht = __builtin_fmaf(dyv.y, wh[1], dht); dwh[1] = __builtin_fmaf(dyv.y, ht, dwh[1]);
            dbo0 += dyv.x; dbo1 += dyv.y;
            const float dct = __builtin_fmaf(dht * g[3], __builtin_fmaf(-tc, tc, 1.0f), dc);       // dL/dc(t)
            dc = dct * g[1];
            // the row's own pre-activation gradient: d_i = dc g i (1 - i), d_f = dc c(t-1) f (1 - f), d_g = dc i (1 - g^2), d_o = dh tanh c o (1 - o)
            const float own = role == 0 ? g[0] : role == 1 ? g[1] : role == 2 ? g[2] : g[3];
            const float mul = role == 0 ? g[2] : role == 1 ? cp : role == 2 ? g[0] : tc;
            const float up = (role == 3 ? dht : dct) * mul;
            const float d_row = up * (is_g ? __builtin_fmaf(-own, own, 1.0f) : own * (1.0f - own));
            float part = rotdot(0.0f, wT, d_row);
            part = sum_rows4(part);
            dh = part;
            const float xsx = col < F ? xb[2 * t + col] : (col == F ? 1.0f : 0.0f);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_row, hp, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_row, xsx, acc2, 0, 0, 0);
        }
    }
    // the workgroup's row of partial gradients (every entry written)
    __device__ __forceinline__ void write_partials(float* prow, float loss) const {
        if (vo && role == 0) { prow[L.o_w_out + col] = dwh[0]; prow[L.o_w_out + H + col] = dwh[1]; }
        if (lane == 0) {
            prow[L.o_b_out] = dbo0; prow[L.o_b_out + 1] = dbo1;
            prow[L.P] = loss; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
        }
        // MFMA block k = gate k; register 4 k + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 4 * role + rr;
                if (i < H) {
                    if (col < H) prow[L.o_w_hh + (k * H + i) * H + col] = acc1[4 * k + rr];
                    const float v = acc2[4 * k + rr];
                    if (col < F) prow[L.o_w_ih + (k * H + i) * F + col] = v;
                    else if (col == F) { prow[L.o_b_ih + k * H + i] = v; prow[L.o_b_hh + k * H + i] = v; }
                }
            }
    }
};

}  // namespace odpd
