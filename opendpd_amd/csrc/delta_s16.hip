// delta_s16.hip — S16 split kernels (forward, backward for parameter gradients) of the delta-network GRU backbones
//   deltagru          backbones/deltagru.py:10-276           feat = [I,Q,a,a^3,sin,cos]; biases = initial accumulators
//   deltagru_tcnskip  backbones/deltagru_tcnskip.py:11-304   feat = [I,Q,a,a^3,I_next,Q_next]; bias-free; + TCN skip
// for batches large enough to fill the chip with 16-sequence wavefronts (the DPD side of the train_dpd cascade at large
// batch, BASELINE config 3).  Arithmetic as in delta_family.hip (thresholded deltas, accumulators `dm`, memories x_p /
// h_p, device sparsity counters, carried accumulator gradients in the backward pass); mapping as in gru_s16.hip: lane
// (n = sequence, q = unit quad), hidden <= 16, mat-vecs on the exact-fp32 MFMA with operands streamed from an LDS
// table — the masked deltas are the B operands, the accumulators are the in-place C/D operands (r, z rows stored
// pre-multiplied by -log2(e)).  Feature deltas live on the feature slots (slot 4c+q on lane q of chunk c).
// Checkpoint per kCkptStride steps: h, h_p, dm_r, dm_z, dm_n, dm_nh (float4 per lane) + the two x_p slots.
#include "odpd_s16.h"

namespace odpd {

namespace d16 {
constexpr int IH = 0;      // g   : (chunk 0, chunk 1) W_ig[m][4e+q]
constexpr int HH = 3;      // g   : W_hg[m][4q+e]
constexpr int HHT = 6;     // g   : W_hg[4q+e][m]
constexpr int WOUT = 9;    // cc  : fc_out[cc][4q+e]
constexpr int DM0 = 11;    // j   : initial accumulators r, z, n, nh at unit 4q+e
constexpr int NG = 15;
constexpr int kHalo = 16;                                   // TCN taps at t-16, t, t+16
constexpr int kStride = kChunk + 2 * kHalo + 1;             // float2 per sequence row of the staged x
constexpr int kCk = 7;                                      // float4 per lane per checkpoint
constexpr int kTiles = 6;                                   // gr gz gn gnh dhm + feature-delta tile
}  // namespace d16

template <bool TRES>
__device__ __forceinline__ float4 d16_entry(const float* pl, const DeltaLayout& L, int grp, int m, int q) {
    const int H = L.H;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        if (grp < d16::HH) {
            const int g = grp, slot = 4 * e + q;
            v[e] = (e < 2 && slot < 6 && m < H) ? pl[L.o_w_ih + (g * H + m) * 6 + slot] * (g < 2 ? kNegLog2e : 1.0f) : 0.0f;
        } else if (grp < d16::HHT) {
            const int g = grp - d16::HH;
            v[e] = (m < H && k < H) ? pl[L.o_w_hh + (g * H + m) * H + k] * (g < 2 ? kNegLog2e : 1.0f) : 0.0f;
        } else if (grp < d16::WOUT) {
            const int g = grp - d16::HHT;
            v[e] = (m < H && k < H) ? pl[L.o_w_hh + (g * H + k) * H + m] : 0.0f;
        } else if (grp < d16::DM0) {
            v[e] = k < H ? pl[L.o_w_out + (grp - d16::WOUT) * H + k] : 0.0f;
        } else {
            const int j = grp - d16::DM0;
            float b = 0.0f;
            if (!TRES && k < H) {
                if (j == 0) b = (pl[L.o_b_ih + k] + pl[L.o_b_hh + k]) * kNegLog2e;
                else if (j == 1) b = (pl[L.o_b_ih + H + k] + pl[L.o_b_hh + H + k]) * kNegLog2e;
                else if (j == 2) b = pl[L.o_b_ih + 2 * H + k];
                else b = pl[L.o_b_hh + 2 * H + k];
            }
            v[e] = b;
        }
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float d16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float d16_hsg(float v) { return v < -3.0f ? 0.0f : (v <= 3.0f ? __builtin_fmaf(v, 1.0f / 3.0f, 0.5f) : 1.0f); }

template <bool TRES>
struct D16Scalars {                 // per-sequence parameters, uniform across lanes
    float bout[2], w1[18], w2[6];
    __device__ __forceinline__ void load(const float* pl, const DeltaLayout& L) {
        bout[0] = TRES ? 0.0f : d16_uni(pl[L.o_b_out]);
        bout[1] = TRES ? 0.0f : d16_uni(pl[L.o_b_out + 1]);
#pragma unroll
        for (int i = 0; i < 18; ++i) w1[i] = TRES ? d16_uni(pl[L.o_tcn0 + i]) : 0.0f;
#pragma unroll
        for (int i = 0; i < 6; ++i) w2[i] = TRES ? d16_uni(pl[L.o_tcn2 + i]) : 0.0f;
    }
};

// recurrent state of one lane
struct D16State { f32x4 h, hp, dmr, dmz, dmn, dmnh; float xp[2]; };

template <bool TRES>
__device__ __forceinline__ void d16_slots(float2 xv, float2 xn, const float (&oh)[4], float (&fs)[2]) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), a = __builtin_amdgcn_sqrtf(a2);
    float f4, f5;
    if constexpr (TRES) { f4 = xn.x; f5 = xn.y; }
    else { const float ia = fast_rcp(a); f4 = xv.y * ia; f5 = xv.x * ia; }
    fs[0] = __builtin_fmaf(oh[0], xv.x, __builtin_fmaf(oh[1], xv.y, __builtin_fmaf(oh[2], a, oh[3] * (a2 * a))));
    fs[1] = __builtin_fmaf(oh[0], f4, oh[1] * f5);
}

// one forward step.  slot_ok[c]: the lane's slot of chunk c is a real feature; unit_ok[i]: a real hidden unit
template <bool TRES>
__device__ __forceinline__ void d16_cell_fwd(TabPtr tl, const float (&fs)[2], float thx, float thh, const bool (&slot_ok)[2],
                                             const f32x4& unit_ok, D16State& st, f32x4& hprev, f32x4& dhm, f32x4& mh, f32x4& r,
                                             f32x4& z, f32x4& n, float (&dxm)[2], float& zx, float& zh) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float d = fs[c] - st.xp[c];
        const bool keep = !(__builtin_fabsf(d) < thx);           // masked_fill(|d| < th, 0)  (deltagru.py:179-183)
        dxm[c] = keep ? d : 0.0f;
        st.xp[c] = (__builtin_fabsf(d) >= thx) ? fs[c] : st.xp[c];
        zx += (slot_ok[c] && dxm[c] == 0.0f) ? 1.0f : 0.0f;
    }
    ODPD_EACH4 {
        const float d = st.h[i] - st.hp[i];
        const bool keep = !(__builtin_fabsf(d) < thh);
        dhm[i] = keep ? d : 0.0f;
        mh[i] = keep ? 1.0f : 0.0f;
        st.hp[i] = (__builtin_fabsf(d) >= thh) ? st.h[i] : st.hp[i];
        zh += (unit_ok[i] != 0.0f && dhm[i] == 0.0f) ? 1.0f : 0.0f;
    }
    const float4 wr = tab_ld(tl, (d16::IH + 0) * 64), wz = tab_ld(tl, (d16::IH + 1) * 64), wn = tab_ld(tl, (d16::IH + 2) * 64);
    st.dmr = mfma4(wr.x, dxm[0], st.dmr); st.dmr = mfma4(wr.y, dxm[1], st.dmr);
    st.dmz = mfma4(wz.x, dxm[0], st.dmz); st.dmz = mfma4(wz.y, dxm[1], st.dmz);
    st.dmn = mfma4(wn.x, dxm[0], st.dmn); st.dmn = mfma4(wn.y, dxm[1], st.dmn);
    const float4 hr = tab_ld(tl, (d16::HH + 0) * 64), hz = tab_ld(tl, (d16::HH + 1) * 64), hn = tab_ld(tl, (d16::HH + 2) * 64);
    st.dmr = mfma4(hr.x, dhm[0], st.dmr); st.dmr = mfma4(hr.y, dhm[1], st.dmr); st.dmr = mfma4(hr.z, dhm[2], st.dmr); st.dmr = mfma4(hr.w, dhm[3], st.dmr);
    st.dmz = mfma4(hz.x, dhm[0], st.dmz); st.dmz = mfma4(hz.y, dhm[1], st.dmz); st.dmz = mfma4(hz.z, dhm[2], st.dmz); st.dmz = mfma4(hz.w, dhm[3], st.dmz);
    st.dmnh = mfma4(hn.x, dhm[0], st.dmnh); st.dmnh = mfma4(hn.y, dhm[1], st.dmnh); st.dmnh = mfma4(hn.z, dhm[2], st.dmnh); st.dmnh = mfma4(hn.w, dhm[3], st.dmnh);
    r = sigmoid4_prescaled(st.dmr);
    z = sigmoid4_prescaled(st.dmz);
    n = tanh4_precise(fma4(r, st.dmnh, st.dmn));
    hprev = st.h;
    st.h = fma4(z, sub4(st.h, n), n);
}

// TCN skip of one sample: s1[3] pre-activations of the first conv, s2[2] of the second
template <bool TRES>
__device__ __forceinline__ void d16_tcn(const D16Scalars<TRES>& sc, float2 xm, float2 xc, float2 xq, float (&s1)[3], float (&s2)[2]) {
    // tcn.0.weight[c][i][k]: taps k = 0,1,2 <-> t-16, t, t+16 of input channel i (I, Q)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = sc.w1[c * 6] * xm.x;
        v = __builtin_fmaf(sc.w1[c * 6 + 1], xc.x, v); v = __builtin_fmaf(sc.w1[c * 6 + 2], xq.x, v);
        v = __builtin_fmaf(sc.w1[c * 6 + 3], xm.y, v); v = __builtin_fmaf(sc.w1[c * 6 + 4], xc.y, v);
        s1[c] = __builtin_fmaf(sc.w1[c * 6 + 5], xq.y, v);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        float v = sc.w2[o * 3] * hardswishf_(s1[0]);
        v = __builtin_fmaf(sc.w2[o * 3 + 1], hardswishf_(s1[1]), v);
        s2[o] = __builtin_fmaf(sc.w2[o * 3 + 2], hardswishf_(s1[2]), v);
    }
}

__device__ __forceinline__ void d16_stage_x(float2* lds, const float* g, int b0, int B, int T, int t0, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int PER = kChunk + 2 * d16::kHalo, TOT = 16 * PER, N = (TOT + 63) / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        if (e < TOT) {
            const int m = e / PER, pos = e % PER, tg = t0 - d16::kHalo + pos;
            float2 v = make_float2(0.0f, 0.0f);          // outside the frame: the conv's zero padding
            if (tg >= 0 && tg < T) v = (b0 + m < B) ? g2[(size_t)(b0 + m) * T + tg] : make_float2(0.5f, 0.5f);
            lds[m * d16::kStride + pos] = v;
        }
    }
}

__device__ __forceinline__ void d16_init_state(TabPtr tl, D16State& st) {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    st.h = z4; st.hp = z4;
    st.dmr = as_f32x4(tab_ld(tl, (d16::DM0 + 0) * 64)); st.dmz = as_f32x4(tab_ld(tl, (d16::DM0 + 1) * 64));
    st.dmn = as_f32x4(tab_ld(tl, (d16::DM0 + 2) * 64)); st.dmnh = as_f32x4(tab_ld(tl, (d16::DM0 + 3) * 64));
    st.xp[0] = st.xp[1] = 0.0f;
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <bool TRES>
__global__ __launch_bounds__(512, 2) void delta16_fwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride;
    constexpr int kWave = 2 * 16 * d16::kStride + 2 * 16 * kChunkPad;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const DeltaLayout L = delta_layout(a.H, TRES);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    {
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int grp = wave; grp < d16::NG; grp += nwb) t4[grp * 64 + lane] = d16_entry<TRES>(pl, L, grp, n, q);
        __syncthreads();
    }
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    D16Scalars<TRES> sc;
    sc.load(pl, L);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    const bool slot_ok[2] = {true, q < 2};
    f32x4 unit_ok;
    ODPD_EACH4 unit_ok[i] = (4 * q + i < a.H) ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(d16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * d16::kStride;
    const float2* xr = xs + n * d16::kStride + d16::kHalo;
    float zx = 0.0f, zh = 0.0f;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * d16::kCk * 64 + lane : nullptr;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[(size_t)(b0 + n) * a.T] : make_float2(0.5f, 0.5f);
        D16State st;
        d16_init_state(tl, st);
        float zxs = 0.0f, zhs = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            d16_stage_x(xs, a.x, b0, a.B, a.T, t0, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xr[tt];
                const float2 xn = (t0 + tt + 1 < a.T) ? xr[tt + 1] : x0;      // torch.roll(x, -1): the last step sees sample 0
                float fs[2], dxm[2];
                f32x4 hprev, dhm, mh, r, z, nn;
                d16_slots<TRES>(xv, xn, oh, fs);
                d16_cell_fwd<TRES>(opaque(tl), fs, a.thx, a.thh, slot_ok, unit_ok, st, hprev, dhm, mh, r, z, nn, dxm, zxs, zhs);
                const f32x4 w0 = as_f32x4(tab_ld(tl, (d16::WOUT + 0) * 64)), w1 = as_f32x4(tab_ld(tl, (d16::WOUT + 1) * 64));
                float p0 = 0.0f, p1 = 0.0f;
                ODPD_EACH4 { p0 = __builtin_fmaf(w0[i], st.h[i], p0); p1 = __builtin_fmaf(w1[i], st.h[i], p1); }
                float y0 = quad_sum(p0) + sc.bout[0], y1 = quad_sum(p1) + sc.bout[1];
                if constexpr (TRES) {
                    float s1[3], s2[2];
                    d16_tcn<TRES>(sc, xr[tt - d16::kHalo], xv, xr[tt + d16::kHalo], s1, s2);
                    y0 += hardswishf_(s2[0]); y1 += hardswishf_(s2[1]);
                }
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    float4* c = ck + (size_t)(t1 / S) * d16::kCk * 64;
                    auto f4 = [](const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); };
                    c[0] = f4(st.h); c[64] = f4(st.hp); c[128] = f4(st.dmr); c[192] = f4(st.dmz); c[256] = f4(st.dmn); c[320] = f4(st.dmnh);
                    c[384] = make_float4(st.xp[0], st.xp[1], 0.0f, 0.0f);
                }
            }
            wave_lds_fence();
            stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
        if (valid) { zx += zxs; zh += zhs; }
    }
    if (a.stats != nullptr) {
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
// backward (parameter gradients)
// -------------------------------------------------------------------------------------------------
template <bool TRES>
struct D16Grad {
    f32x4 thh[3], tih[3];
    f32x4 dwout[2], db[4];
    float dbout[2], dw1[18], dw2[6];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) { thh[g] = z4; tih[g] = z4; }
        dwout[0] = dwout[1] = z4;
#pragma unroll
        for (int j = 0; j < 4; ++j) db[j] = z4;
        dbout[0] = dbout[1] = 0.f;
#pragma unroll
        for (int i = 0; i < 18; ++i) dw1[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) dw2[i] = 0.f;
    }
};
struct D16Carry { f32x4 gh, ghp, gr, gz, gn, gnh; };

template <bool TRES, bool FULL>
__device__ __forceinline__ void d16_bwd_block(const SeqArgs& a, TabPtr tl0, const D16Scalars<TRES>& sc, const float (&oh)[4],
                                              D16Grad<TRES>& G, const float2* xr, const float2* dys, float* tiles, float2 x0,
                                              int n, int q, int tglob, int tloc, int nstep, D16State st, D16Carry& C) {
    constexpr int S = kCkptStride;
    const bool slot_ok[2] = {true, q < 2};
    const f32x4 all_units = {1.f, 1.f, 1.f, 1.f};
    f32x4 hprev_s[S], dhm_s[S], mh_s[S], r_s[S], z_s[S], n_s[S], nh_s[S], ht_s[S];
    float dxm_s[S][2];
    TabPtr tl = opaque(tl0);
    {
        float zx = 0.f, zh = 0.f;
#pragma unroll
        for (int st_i = 0; st_i < S; ++st_i) {
            if (FULL || st_i < nstep) {
                const float2 xv = xr[tloc + st_i];
                const float2 xn = (tglob + st_i + 1 < a.T) ? xr[tloc + st_i + 1] : x0;
                float fs[2];
                d16_slots<TRES>(xv, xn, oh, fs);
                d16_cell_fwd<TRES>(tl, fs, a.thx, a.thh, slot_ok, all_units, st, hprev_s[st_i], dhm_s[st_i], mh_s[st_i], r_s[st_i],
                                   z_s[st_i], n_s[st_i], dxm_s[st_i], zx, zh);
                nh_s[st_i] = st.dmnh;
                ht_s[st_i] = st.h;
            }
        }
    }
    tl = opaque(tl0);
    const f32x4 w0 = as_f32x4(tab_ld(tl, (d16::WOUT + 0) * 64)), w1 = as_f32x4(tab_ld(tl, (d16::WOUT + 1) * 64));
    float* t_r = tiles, *t_z = tiles + kTileFloats, *t_n = tiles + 2 * kTileFloats, *t_g = tiles + 3 * kTileFloats;
    float* t_h = tiles + 4 * kTileFloats, *t_f = tiles + 5 * kTileFloats;
    const f32x4 one = splat4(1.0f);
#pragma unroll
    for (int st_i = S - 1; st_i >= 0; --st_i) {
        if (FULL || st_i < nstep) {
            const int tt = tloc + st_i;
            const float2 dyv = dys[n * kChunkPad + tt];
            const f32x4 gh = add4(C.gh, fma4(splat4(dyv.x), w0, mul4(w1, splat4(dyv.y))));
            G.dwout[0] = fma4(splat4(dyv.x), ht_s[st_i], G.dwout[0]);
            G.dwout[1] = fma4(splat4(dyv.y), ht_s[st_i], G.dwout[1]);
            G.dbout[0] += q == 0 ? dyv.x : 0.0f;
            G.dbout[1] += q == 0 ? dyv.y : 0.0f;
            if constexpr (TRES) {
                if (q == 0) {    // per-sequence work: one lane of the four is enough (wave-uniform branch per quad row)
                    float s1[3], s2[2];
                    const float2 xm = xr[tt - d16::kHalo], xc = xr[tt], xq = xr[tt + d16::kHalo];
                    d16_tcn<TRES>(sc, xm, xc, xq, s1, s2);
                    const float d2[2] = {dyv.x * d16_hsg(s2[0]), dyv.y * d16_hsg(s2[1])};
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float hs = hardswishf_(s1[c]);
                        G.dw2[c] = __builtin_fmaf(d2[0], hs, G.dw2[c]);
                        G.dw2[3 + c] = __builtin_fmaf(d2[1], hs, G.dw2[3 + c]);
                        const float d1 = __builtin_fmaf(d2[0], sc.w2[c], d2[1] * sc.w2[3 + c]) * d16_hsg(s1[c]);
                        G.dw1[c * 6 + 0] = __builtin_fmaf(d1, xm.x, G.dw1[c * 6 + 0]); G.dw1[c * 6 + 1] = __builtin_fmaf(d1, xc.x, G.dw1[c * 6 + 1]);
                        G.dw1[c * 6 + 2] = __builtin_fmaf(d1, xq.x, G.dw1[c * 6 + 2]); G.dw1[c * 6 + 3] = __builtin_fmaf(d1, xm.y, G.dw1[c * 6 + 3]);
                        G.dw1[c * 6 + 4] = __builtin_fmaf(d1, xc.y, G.dw1[c * 6 + 4]); G.dw1[c * 6 + 5] = __builtin_fmaf(d1, xq.y, G.dw1[c * 6 + 5]);
                    }
                }
            }
            const f32x4 r = r_s[st_i], z = z_s[st_i], nn = n_s[st_i];
            const f32x4 dn = mul4(gh, sub4(one, z)), dz = mul4(gh, sub4(hprev_s[st_i], nn));
            f32x4 ghprev = mul4(gh, z), omn2;
            ODPD_EACH4 omn2[i] = __builtin_fmaf(-nn[i], nn[i], 1.0f);
            const f32x4 dpre = mul4(dn, omn2);
            C.gn = add4(C.gn, dpre);
            C.gnh = fma4(dpre, r, C.gnh);
            C.gr = fma4(mul4(dpre, nh_s[st_i]), mul4(r, sub4(one, r)), C.gr);
            C.gz = fma4(dz, mul4(z, sub4(one, z)), C.gz);
            // data gradient to the masked dh: W_hh^T [G_r, G_z, G_nh]
            f32x4 ddh = {0.f, 0.f, 0.f, 0.f};
            {
                const float4 tr = tab_ld(tl, (d16::HHT + 0) * 64), tz = tab_ld(tl, (d16::HHT + 1) * 64), tn = tab_ld(tl, (d16::HHT + 2) * 64);
                ddh = mfma4(tr.x, C.gr[0], ddh); ddh = mfma4(tr.y, C.gr[1], ddh); ddh = mfma4(tr.z, C.gr[2], ddh); ddh = mfma4(tr.w, C.gr[3], ddh);
                ddh = mfma4(tz.x, C.gz[0], ddh); ddh = mfma4(tz.y, C.gz[1], ddh); ddh = mfma4(tz.z, C.gz[2], ddh); ddh = mfma4(tz.w, C.gz[3], ddh);
                ddh = mfma4(tn.x, C.gnh[0], ddh); ddh = mfma4(tn.y, C.gnh[1], ddh); ddh = mfma4(tn.z, C.gnh[2], ddh); ddh = mfma4(tn.w, C.gnh[3], ddh);
            }
            const f32x4 mk = mh_s[st_i];
            ghprev = fma4(mk, add4(ddh, C.ghp), ghprev);
            ODPD_EACH4 C.ghp[i] = __builtin_fmaf(-mk[i], ddh[i], (1.0f - mk[i]) * C.ghp[i]);
            C.gh = ghprev;
            // weight gradients: dW_ih += G_dm^T (x) dx_masked, dW_hh += [G_r, G_z, G_nh]^T (x) dh_masked
            wave_lds_fence();
            tile_put(t_r, n, q, C.gr);
            tile_put(t_z, n, q, C.gz);
            tile_put(t_n, n, q, C.gn);
            tile_put(t_g, n, q, C.gnh);
            tile_put(t_h, n, q, dhm_s[st_i]);
            t_f[n * kTilePitch + q] = dxm_s[st_i][0];
            t_f[n * kTilePitch + 4 + q] = dxm_s[st_i][1];      // slots 6, 7 are zero deltas (the lanes' features are 0 there)
            wave_lds_fence();
            float rT[4], zT[4], nT[4], gT[4], hT[4], fT[4];
            tile_get(t_r, n, q, rT); tile_get(t_z, n, q, zT); tile_get(t_n, n, q, nT); tile_get(t_g, n, q, gT);
            tile_get(t_h, n, q, hT); tile_get(t_f, n, q, fT);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                G.tih[0] = mfma4(rT[c], fT[c], G.tih[0]);
                G.tih[1] = mfma4(zT[c], fT[c], G.tih[1]);
                G.tih[2] = mfma4(nT[c], fT[c], G.tih[2]);
                G.thh[0] = mfma4(rT[c], hT[c], G.thh[0]);
                G.thh[1] = mfma4(zT[c], hT[c], G.thh[1]);
                G.thh[2] = mfma4(gT[c], hT[c], G.thh[2]);
            }
        }
    }
}

template <bool TRES>
__device__ __forceinline__ void d16_write_row(float* prow, const DeltaLayout& L, D16Grad<TRES>& G, int lane, int n, int q) {
    const int H = L.H;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * q + rr;
            if (i < H) {
                if (n < 6) prow[L.o_w_ih + (g * H + i) * 6 + n] = G.tih[g][rr];
                if (n < H) prow[L.o_w_hh + (g * H + i) * H + n] = G.thh[g][rr];
            }
        }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int u = 4 * q + rr;
        const float w0 = row_sum16(G.dwout[0][rr]), w1 = row_sum16(G.dwout[1][rr]);
        float db[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) db[j] = row_sum16(G.db[j][rr]);
        if (n == 0 && u < H) {
            prow[L.o_w_out + u] = w0; prow[L.o_w_out + H + u] = w1;
            if constexpr (!TRES) {
                prow[L.o_b_ih + u] = db[0]; prow[L.o_b_hh + u] = db[0];
                prow[L.o_b_ih + H + u] = db[1]; prow[L.o_b_hh + H + u] = db[1];
                prow[L.o_b_ih + 2 * H + u] = db[2]; prow[L.o_b_hh + 2 * H + u] = db[3];
            }
        }
    }
    if constexpr (TRES) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const float v = row_sum16(G.dw1[i]);
            if (n == 0 && q == 0) prow[L.o_tcn0 + i] = v;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float v = row_sum16(G.dw2[i]);
            if (n == 0 && q == 0) prow[L.o_tcn2 + i] = v;
        }
    } else {
        const float b0 = row_sum16(G.dbout[0]), b1 = row_sum16(G.dbout[1]);
        if (n == 0 && q == 0) { prow[L.o_b_out] = b0; prow[L.o_b_out + 1] = b1; }
    }
}

template <bool TRES>
__global__ __launch_bounds__(256, 1) void delta16_bwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride;
    constexpr int kWave = 2 * 16 * d16::kStride + 2 * 16 * kChunkPad + d16::kTiles * kTileFloats;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const DeltaLayout L = delta_layout(a.H, TRES);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    {
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int grp = wave; grp < d16::NG; grp += nwb) t4[grp * 64 + lane] = d16_entry<TRES>(pl, L, grp, n, q);
        __syncthreads();
    }
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    D16Scalars<TRES> sc;
    sc.load(pl, L);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(d16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * d16::kStride;
    float* tiles = reinterpret_cast<float*>(dys + 16 * kChunkPad);
    for (int i = lane; i < kTileFloats; i += 64) tiles[5 * kTileFloats + i] = 0.0f;
    const float2* xr = xs + n * d16::kStride + d16::kHalo;
    D16Grad<TRES> G;
    G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * d16::kCk * 64 + lane;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[(size_t)(b0 + n) * a.T] : make_float2(0.5f, 0.5f);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        D16Carry C = {z4, z4, z4, z4, z4, z4};
        int cur_chunk = -1;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                d16_stage_x(xs, a.x, b0, a.B, a.T, t0, lane);
                stage_in<16>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            D16State st;
            if (blk) {
                const float4* c = ck + (size_t)blk * d16::kCk * 64;
                st.h = as_f32x4(c[0]); st.hp = as_f32x4(c[64]); st.dmr = as_f32x4(c[128]); st.dmz = as_f32x4(c[192]);
                st.dmn = as_f32x4(c[256]); st.dmnh = as_f32x4(c[320]);
                const float4 xp = c[384];
                st.xp[0] = xp.x; st.xp[1] = xp.y;
            } else {
                d16_init_state(tl, st);
            }
            if (nstep == S) d16_bwd_block<TRES, true>(a, tl, sc, oh, G, xr, dys, tiles, x0, n, q, tb, tb - t0, nstep, st, C);
            else d16_bwd_block<TRES, false>(a, tl, sc, oh, G, xr, dys, tiles, x0, n, q, tb, tb - t0, nstep, st, C);
        }
        // gradient w.r.t. the initial accumulators = bias gradients (deltagru.py:165-170)
        G.db[0] = add4(G.db[0], C.gr); G.db[1] = add4(G.db[1], C.gz); G.db[2] = add4(G.db[2], C.gn); G.db[3] = add4(G.db[3], C.gnh);
    }
    const int P4 = L.P + kLossCols;
    __syncthreads();
    d16_write_row<TRES>(smem + wave * P4, L, G, lane, n, q);
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = smem[i];
        for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
bool delta_uses_s16(const odpd_model_t* m, int B) {
    if ((m->backbone != ODPD_DELTAGRU && m->backbone != ODPD_TRES_DELTAGRU) || m->hidden > 16) return false;
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 16L * 4 * device_cus();
    return B >= min_batch;
}
static LaunchShape d16_fwd_shape(int ngroups) {
    LaunchShape ls;
    const int cus = device_cus();
    ls.waves = ngroups <= 4 * cus ? 4 : 8;
    const int need = (ngroups + ls.waves - 1) / ls.waves;
    ls.grid = need < cus ? need : cus;
    return ls;
}
static LaunchShape d16_bwd_shape(int ngroups) {
    LaunchShape ls;
    ls.waves = 4;
    const int need = (ngroups + 3) / 4, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
int delta_s16_rows(const odpd_model_t* m, int B) {
    (void)m;
    return d16_bwd_shape((B + 15) / 16).grid;
}
int64_t delta_s16_ckpt_floats(const odpd_model_t* m, int B, int T) {
    (void)m;
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * d16::kCk * 256;
}
template <bool TRES>
static int d16_launch(hipStream_t st, const SeqArgs& a, int P, int mode) {
    if (mode == 1) {
        const LaunchShape ls = d16_fwd_shape(a.ngroups);
        const size_t lds = ((size_t)pad4(P) + s16_tab_floats(d16::NG) + (size_t)ls.waves * (2 * 16 * d16::kStride + 2 * 16 * kChunkPad)) * sizeof(float);
        auto k = delta16_fwd_kernel<TRES>;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    }
    if (a.dx != nullptr) return ODPD_EUNSUPPORTED;   // dL/dx of a delta backbone is not implemented
    if (a.partials == nullptr) return ODPD_EINVAL;
    const LaunchShape ls = d16_bwd_shape(a.ngroups);
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(d16::NG) +
                  (size_t)ls.waves * (2 * 16 * d16::kStride + 2 * 16 * kChunkPad + d16::kTiles * kTileFloats)) * sizeof(float);
    if (lds < reduce_scratch_bytes(P, ls.waves)) lds = reduce_scratch_bytes(P, ls.waves);
    auto k = delta16_bwd_kernel<TRES>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
int delta_s16_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const bool tres = m->backbone == ODPD_TRES_DELTAGRU;
    const int P = delta_layout(m->hidden, tres).P;
    return tres ? d16_launch<true>(st, a, P, mode) : d16_launch<false>(st, a, P, mode);
}

}  // namespace odpd
