// bojanet_s16.hip — BOJANET (backbones/bojanet.py:5-138) in the S16 mapping (see gru_s16.hip / odpd_s16.h): a wave = 16 sequences,
// lane (n = sequence, q = quad) owns units 4q + i (hidden <= 16) AND the two demodulator filters 2q, 2q + 1 (q < 3).  Per step
// (bojanet.py:72-104):
//   16-tap complex FIR bank of 6 filters over the zero-left-padded frame: fi_p + j fq_p = sum_m (bI[p][m] + j bQ[p][m]) x[t-15+m]
//     -> ONE 16 x 32 exact-fp32 MFMA product per step (rows = (filter, re/im), K = (tap, I/Q)); the window comes straight from the
//     frame chunk staged in LDS with a 16-sample halo, and the result lands as [fi_a, fq_a, fi_b, fq_b] on the lane of its filters;
//   vector demodulator: mag = sqrt(fi^2 + fq^2) + 1e-8, mag^2, cos = fi / mag, sin = fq / mag;
//   JANET cell on the 12 envelopes: f = s(W_fi e + b + W_fh h), g = tanh(W_gi e + b + W_gh h), h = f h + (1 - f) g (MFMA mat-vecs
//     from the LDS table; the f rows carry -log2(e));
//   phase re-rotation: unit j takes the phase of filter j mod 6 — a 0/1 selection matrix on the MFMA (exact), no cross-lane shuffles;
//   y = (A - Bq, Bq + A) with A = w_I . (h cos) + b_I, Bq = w_Q . (h sin) + b_Q (both outputs mix the two read-outs, :103-104).
// BPTT from checkpoints of h every kCkptStride steps; the FIR's weight gradient is an MFMA outer product of the demodulator gradients
// with the LDS window, dL/dx the transposed bank accumulated into an LDS frame chunk with the same halo (carried across chunks).
// The reference's phase re-rotation cannot be built beyond hidden 18 (its concatenation, :41-53); the kernels take hidden <= 16.
// Deviation: where a filter output is exactly 0 the reference's gradient is NaN (0 * inf through sqrt); here that term is dropped.
#include "odpd_s16.h"

namespace odpd {
namespace {

constexpr int kBojP = 6, kBojM = 16;
constexpr int kBojHalo = 16;                        // staged samples before the chunk (15 used: taps reach back to t - 15)
constexpr int kBojRow = kBojHalo + kChunk + 1;      // float2 row stride of a staged frame chunk: index i <-> time t0 - 16 + i
struct BojLayout { int H, o_bi, o_bq, o_wfi, o_bfi, o_wfh, o_wgi, o_bgi, o_wgh, o_woi, o_boi, o_woq, o_boq, P; };
__host__ __device__ inline BojLayout boj_layout(int H) {
    BojLayout L; L.H = H; int o = 0;
    L.o_bi = o; o += kBojP * kBojM; L.o_bq = o; o += kBojP * kBojM;
    L.o_wfi = o; o += H * 2 * kBojP; L.o_bfi = o; o += H; L.o_wfh = o; o += H * H;
    L.o_wgi = o; o += H * 2 * kBojP; L.o_bgi = o; o += H; L.o_wgh = o; o += H * H;
    L.o_woi = o; o += H; L.o_boi = o; o += 1; L.o_woq = o; o += H; L.o_boq = o; o += 1;
    L.P = o;
    return L;
}

struct B16 {
    static constexpr int FIR = 0;          // jj : A[row m][(tap q + 4 (2 jj + e / 2), e % 2)]
    static constexpr int WFI = FIR + 2;    // W_fi[m][env(q, e)]  (pre-scaled)
    static constexpr int WGI = WFI + 1;
    static constexpr int FH = WGI + 1;     // W_fh[m][4q+e]       (pre-scaled)
    static constexpr int GH = FH + 1;
    static constexpr int SEL = GH + 1;     // 1 where unit m takes the phase of filter 2q + e / 2
    static constexpr int SELTC = SEL + 1;  // transposes, rows = (filter pair, cos | sin) — cos rows only / sin rows only
    static constexpr int SELTS = SELTC + 1;
    static constexpr int TRFH = SELTS + 1; // W_fh[4q+e][m]
    static constexpr int TRGH = TRFH + 1;
    static constexpr int WFIT = TRGH + 1;  // W_fi[4q+e][env(row m)]
    static constexpr int WGIT = WFIT + 1;
    static constexpr int FIRT = WGIT + 1;  // tile : A[row 4q+e][(tap 8 tile + m / 2, m % 2)]
    static constexpr int SC = FIRT + 2;    // b_fi (pre-scaled), b_gi, w_I, w_Q at unit 4q+e
    static constexpr int NG = SC + 4;
    static constexpr int kTiles = 5;       // dfp dgp dF | env hp
};
// the FIR as a real 16 x 32 matrix: row i = 4 qi + r <-> (filter 2 qi + r / 2, r % 2 = 0 re | 1 im), rows 12..15 empty; column (tap, c = 0 I | 1 Q)
__device__ __forceinline__ float b16_fir(const float* pl, const BojLayout& L, int i, int tap, int c) {
    const int qi = i >> 2, r = i & 3, p = 2 * qi + (r >> 1);
    if (qi >= 3) return 0.0f;
    const float bi = pl[L.o_bi + p * kBojM + tap], bq = pl[L.o_bq + p * kBojM + tap];
    return (r & 1) == 0 ? (c == 0 ? bi : -bq) : (c == 0 ? bq : bi);
}
// envelope index of slot (q, e): lane q carries [mag_2q, mag^2_2q, mag_2q+1, mag^2_2q+1]; e = [mag(6), mag^2(6)] (bojanet.py:86-87)
__device__ __forceinline__ int b16_env(int q, int e) { return ((e & 1) ? kBojP : 0) + 2 * q + (e >> 1); }
__device__ __forceinline__ float4 b16_entry(const float* pl, const BojLayout& L, int grp, int m, int q) {
    const int H = L.H;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int u = 4 * q + e;
        float s = 0.0f;
        if (grp < B16::WFI) s = b16_fir(pl, L, m, q + 4 * (2 * (grp - B16::FIR) + (e >> 1)), e & 1);
        else if (grp == B16::WFI) s = (m < H && q < 3) ? pl[L.o_wfi + m * 2 * kBojP + b16_env(q, e)] * kNegLog2e : 0.0f;
        else if (grp == B16::WGI) s = (m < H && q < 3) ? pl[L.o_wgi + m * 2 * kBojP + b16_env(q, e)] : 0.0f;
        else if (grp == B16::FH) s = (m < H && u < H) ? pl[L.o_wfh + m * H + u] * kNegLog2e : 0.0f;
        else if (grp == B16::GH) s = (m < H && u < H) ? pl[L.o_wgh + m * H + u] : 0.0f;
        else if (grp == B16::SEL) s = (m < H && q < 3 && 2 * q + (e >> 1) == m % kBojP) ? 1.0f : 0.0f;
        else if (grp == B16::SELTC || grp == B16::SELTS) {
            const int qi = m >> 2, r = m & 3;
            s = (qi < 3 && u < H && 2 * qi + (r >> 1) == u % kBojP && (r & 1) == (grp == B16::SELTS ? 1 : 0)) ? 1.0f : 0.0f;
        } else if (grp == B16::TRFH) s = (m < H && u < H) ? pl[L.o_wfh + u * H + m] : 0.0f;
        else if (grp == B16::TRGH) s = (m < H && u < H) ? pl[L.o_wgh + u * H + m] : 0.0f;
        else if (grp == B16::WFIT || grp == B16::WGIT) {
            const int qi = m >> 2, r = m & 3;
            s = (qi < 3 && u < H) ? pl[(grp == B16::WFIT ? L.o_wfi : L.o_wgi) + u * 2 * kBojP + b16_env(qi, r)] : 0.0f;
        } else if (grp < B16::SC) s = b16_fir(pl, L, u, 8 * (grp - B16::FIRT) + (m >> 1), m & 1);
        else {
            const int j = grp - B16::SC;
            if (u < H) s = j == 0 ? pl[L.o_bfi + u] * kNegLog2e : j == 1 ? pl[L.o_bgi + u] : j == 2 ? pl[L.o_woi + u] : pl[L.o_woq + u];
        }
        v[e] = s;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void b16_build_table(float* tab, const float* pl, const BojLayout& L, int lane, int wave, int nwb) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < B16::NG; grp += nwb) t4[grp * 64 + lane] = b16_entry(pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}
__device__ __forceinline__ float b16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x4 b16_mv(TabPtr tl, int grp, const f32x4& v, f32x4 acc) {
    f32x4 a1[1] = {acc};
    const f32x4 v1[1] = {v};
    s16n_matvec<1>(tl, grp, v1, a1);
    return a1[0];
}

// frame chunk with its halo: 16 sequences x 48 samples, times t0 - 16 .. t0 + 31 (zeros before the frame: bojanet.py:72-73)
__device__ __forceinline__ void b16_stage_in(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int W = kBojHalo + kChunk;
#pragma unroll
    for (int j = 0; j < 16 * W / 64; ++j) {
        const int e = lane + 64 * j, m = e / W, i = e % W, t = t0 - kBojHalo + i;
        float2 v = make_float2(0.0f, 0.0f);
        if (b0 + m >= B) v = make_float2(0.5f, 0.25f);                  // idle sequence slots: any finite, non-degenerate signal
        else if (t >= 0 && t < t0 + len) v = g2[(size_t)(b0 + m) * T + t];
        lds[m * kBojRow + i] = v;
    }
}
__device__ __forceinline__ void b16_stage_out(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
#pragma unroll
    for (int j = 0; j < 16 * kChunk / 64; ++j) {
        const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
        if (tt < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + tt] = lds[m * kBojRow + kBojHalo + tt];
    }
}
// dL/dx chunk hand-over (backward runs the chunks last to first): what the finished chunk put before its own t0 (indices 1..15)
// belongs to the end of the next (earlier) one (indices 33..47); everything else restarts at 0
__device__ __forceinline__ void b16_dx_carry(float2* lds, int lane, bool first) {
    float2 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j, m = e >> 4, i = e & 15;
        c[j] = first ? make_float2(0.0f, 0.0f) : lds[m * kBojRow + i];
    }
    wave_lds_fence();
    for (int e = lane; e < 16 * kBojRow; e += 64) lds[e] = make_float2(0.0f, 0.0f);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j, m = e >> 4, i = e & 15;
        if (i) lds[m * kBojRow + kChunk + i] = c[j];
    }
    wave_lds_fence();
}

struct B16Front { f32x4 ev, cs; float m0a, m0b, ia, ib; };     // [mag_a, mag_a^2, mag_b, mag_b^2], [cos_a, sin_a, cos_b, sin_b]
// FIR bank (8 MFMAs) at local step tt: window = staged indices tt + 1 .. tt + 16
__device__ __forceinline__ f32x4 b16_fir_fwd(TabPtr tl, const float2* xrow, int tt, int q) {
    f32x4 ff = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const float4 w = tab_ld(tl, (B16::FIR + jj) * 64);
        const float2 x0 = xrow[tt + 1 + q + 8 * jj], x1 = xrow[tt + 1 + q + 8 * jj + 4];
        ff = mfma4(w.x, x0.x, ff); ff = mfma4(w.y, x0.y, ff);
        ff = mfma4(w.z, x1.x, ff); ff = mfma4(w.w, x1.y, ff);
    }
    return ff;
}
__device__ __forceinline__ B16Front b16_demod(const f32x4& ff) {
    B16Front F;
    F.m0a = __builtin_amdgcn_sqrtf(__builtin_fmaf(ff[0], ff[0], ff[1] * ff[1]));
    F.m0b = __builtin_amdgcn_sqrtf(__builtin_fmaf(ff[2], ff[2], ff[3] * ff[3]));
    const float ma = F.m0a + 1e-8f, mb = F.m0b + 1e-8f;
    F.ia = fast_rcp(ma); F.ib = fast_rcp(mb);
    F.ev = f32x4{ma, ma * ma, mb, mb * mb};
    F.cs = f32x4{ff[0] * F.ia, ff[1] * F.ia, ff[2] * F.ib, ff[3] * F.ib};
    return F;
}
__device__ __forceinline__ void b16_cell(TabPtr tl, const f32x4& ev, f32x4& h, f32x4& f, f32x4& g) {
    f32x4 pf = as_f32x4(tab_ld(tl, (B16::SC + 0) * 64)), pg = as_f32x4(tab_ld(tl, (B16::SC + 1) * 64));
    pf = b16_mv(tl, B16::WFI, ev, pf); pg = b16_mv(tl, B16::WGI, ev, pg);
    pf = b16_mv(tl, B16::FH, h, pf); pg = b16_mv(tl, B16::GH, h, pg);
    f = sigmoid4_prescaled(pf); g = tanh4_precise(pg);
    h = fma4(f, sub4(h, g), g);
}
// phases of the units: cosx_j = cos_(j mod 6), sinx_j = sin_(j mod 6)
__device__ __forceinline__ void b16_rotate(TabPtr tl, const f32x4& cs, f32x4& cosx, f32x4& sinx) {
    const float4 s = tab_ld(tl, B16::SEL * 64);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    cosx = mfma4(s.x, cs[0], z4); cosx = mfma4(s.z, cs[2], cosx);
    sinx = mfma4(s.y, cs[1], z4); sinx = mfma4(s.w, cs[3], sinx);
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void boj16_fwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride, kWave = 2 * 16 * kBojRow + 2 * 16 * kChunkPad;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const BojLayout L = boj_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    b16_build_table(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const float boi = b16_uni(pl[L.o_boi]), boq = b16_uni(pl[L.o_boq]);
    float* wbase = tab + s16_tab_floats(B16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * kBojRow;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * 64 + lane : nullptr;
        f32x4 h = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            b16_stage_in(xs, a.x, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const TabPtr tk = opaque(tl);
                const B16Front F = b16_demod(b16_fir_fwd(tk, xs + n * kBojRow, tt, q));
                f32x4 f, g, cosx, sinx;
                b16_cell(tk, F.ev, h, f, g);
                b16_rotate(tk, F.cs, cosx, sinx);
                const f32x4 wi = as_f32x4(tab_ld(tk, (B16::SC + 2) * 64)), wq = as_f32x4(tab_ld(tk, (B16::SC + 3) * 64));
                float s0 = 0.0f, s1 = 0.0f;
                ODPD_EACH4 { s0 = __builtin_fmaf(wi[i], h[i] * cosx[i], s0); s1 = __builtin_fmaf(wq[i], h[i] * sinx[i], s1); }
                const float A = quad_sum(s0) + boi, Bq = quad_sum(s1) + boq;
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(A - Bq, Bq + A);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) ck[(size_t)(t1 / S) * 64] = make_float4(h[0], h[1], h[2], h[3]);
            }
            wave_lds_fence();
            stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------
struct B16Grad {
    f32x4 fir[2], wfi, wgi, wfh, wgh;      // weight-gradient tiles: FIR rows x (16 columns per tile), gates x envelopes, gates x state
    f32x4 dbf, dbg, dwi, dwq;
    float dbo[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        fir[0] = fir[1] = wfi = wgi = wfh = wgh = dbf = dbg = dwi = dwq = z4;
        dbo[0] = dbo[1] = 0.f;
    }
};

template <bool NW, bool DX, bool FULL>
__device__ __forceinline__ void b16_bwd_block(TabPtr tl0, B16Grad& G, const float2* xs, const float2* dys, float2* dxs, float* tiles,
                                              int n, int q, int tloc, int nstep, f32x4 h, f32x4& dh) {
    constexpr int S = kCkptStride;
    f32x4 hp_s[S], f_s[S], g_s[S], ff_s[S];
#pragma unroll
    for (int si = 0; si < S; ++si) {
        if (FULL || si < nstep) {
            const TabPtr tk = opaque(tl0);
            ff_s[si] = b16_fir_fwd(tk, xs + n * kBojRow, tloc + si, q);
            const B16Front F = b16_demod(ff_s[si]);
            hp_s[si] = h;
            b16_cell(tk, F.ev, h, f_s[si], g_s[si]);
        }
    }
    auto tile = [tiles](int qty) { return tiles + qty * kTileFloats; };     // 0 dfp 1 dgp 2 dF | 3 env 4 hp
    const f32x4 one = splat4(1.0f), z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            const TabPtr tl = opaque(tl0);
            const int tt = tloc + si;
            const float2 dyv = dys[n * kChunkPad + tt];
            const float dA = dyv.x + dyv.y, dB = dyv.y - dyv.x;          // y = (A - Bq, Bq + A)
            if constexpr (NW) { G.dbo[0] += q == 0 ? dA : 0.0f; G.dbo[1] += q == 0 ? dB : 0.0f; }
            const f32x4 hp = hp_s[si], f = f_s[si], g = g_s[si], ff = ff_s[si];
            const B16Front F = b16_demod(ff);
            f32x4 cosx, sinx;
            b16_rotate(tl, F.cs, cosx, sinx);
            const f32x4 wi = as_f32x4(tab_ld(tl, (B16::SC + 2) * 64)), wq = as_f32x4(tab_ld(tl, (B16::SC + 3) * 64));
            const f32x4 hn = fma4(f, sub4(hp, g), g);
            f32x4 gh, dcx, dsx, dfp, dgp, nh;
            ODPD_EACH4 {
                const float ai = dA * wi[i], bq = dB * wq[i];
                gh[i] = __builtin_fmaf(ai, cosx[i], __builtin_fmaf(bq, sinx[i], dh[i]));
                dcx[i] = ai * hn[i]; dsx[i] = bq * hn[i];
                dfp[i] = gh[i] * (hp[i] - g[i]) * f[i] * (1.0f - f[i]);
                dgp[i] = gh[i] * (1.0f - f[i]) * __builtin_fmaf(-g[i], g[i], 1.0f);
                nh[i] = gh[i] * f[i];
            }
            if constexpr (NW) {
                ODPD_EACH4 {
                    G.dwi[i] = __builtin_fmaf(dA, hn[i] * cosx[i], G.dwi[i]);
                    G.dwq[i] = __builtin_fmaf(dB, hn[i] * sinx[i], G.dwq[i]);
                }
                G.dbf = add4(G.dbf, dfp); G.dbg = add4(G.dbg, dgp);
            }
            // phases back onto the filters: [dcos_a, dsin_a, dcos_b, dsin_b]
            f32x4 dcs = b16_mv(tl, B16::SELTC, dcx, z4);
            dcs = b16_mv(tl, B16::SELTS, dsx, dcs);
            nh = b16_mv(tl, B16::TRFH, dfp, nh);
            nh = b16_mv(tl, B16::TRGH, dgp, nh);
            dh = nh;
            // envelopes: [dmag_a, dmag^2_a, dmag_b, dmag^2_b]
            f32x4 de = b16_mv(tl, B16::WFIT, dfp, z4);
            de = b16_mv(tl, B16::WGIT, dgp, de);
            // demodulator (bojanet.py:30-39): mag = m0 + eps, mag^2, sin = fq / mag, cos = fi / mag
            f32x4 dF;
            {
                const float ma = F.ev[0], mb = F.ev[2];
                const float dma = __builtin_fmaf(2.0f * ma, de[1], de[0]) - __builtin_fmaf(dcs[1], ff[1], dcs[0] * ff[0]) * (F.ia * F.ia);
                const float dmb = __builtin_fmaf(2.0f * mb, de[3], de[2]) - __builtin_fmaf(dcs[3], ff[3], dcs[2] * ff[2]) * (F.ib * F.ib);
                const float ra = F.m0a > 0.0f ? dma * fast_rcp(F.m0a) : 0.0f, rb = F.m0b > 0.0f ? dmb * fast_rcp(F.m0b) : 0.0f;
                dF[0] = __builtin_fmaf(dcs[0], F.ia, ra * ff[0]); dF[1] = __builtin_fmaf(dcs[1], F.ia, ra * ff[1]);
                dF[2] = __builtin_fmaf(dcs[2], F.ib, rb * ff[2]); dF[3] = __builtin_fmaf(dcs[3], F.ib, rb * ff[3]);
            }
            if constexpr (DX) {
                // transposed bank: the 32 window positions of this step, two per tile and lane: taps 8 tile + 2q, + 1
#pragma unroll
                for (int tile_i = 0; tile_i < 2; ++tile_i) {
                    const f32x4 dw = b16_mv(tl, B16::FIRT + tile_i, dF, z4);
                    float2* d = dxs + n * kBojRow + tt + 1 + 8 * tile_i + 2 * q;
                    float2 v0 = d[0], v1 = d[1];
                    v0.x += dw[0]; v0.y += dw[1]; v1.x += dw[2]; v1.y += dw[3];
                    d[0] = v0; d[1] = v1;
                }
            }
            if constexpr (NW) {
                wave_lds_fence();
                tile_put(tile(0), n, q, dfp); tile_put(tile(1), n, q, dgp); tile_put(tile(2), n, q, dF);
                tile_put(tile(3), n, q, F.ev); tile_put(tile(4), n, q, hp);
                wave_lds_fence();
                float dT[3][4], sT[2][4];
#pragma unroll
                for (int j = 0; j < 3; ++j) tile_get(tile(j), n, q, dT[j]);
                tile_get(tile(3), n, q, sT[0]); tile_get(tile(4), n, q, sT[1]);
                // FIR: rows = demodulator gradients, columns = window positions; this lane is column n of both tiles:
                // position (tap 8 tile + n / 2, n % 2) of sequences 4q + c
                const float* xf = reinterpret_cast<const float*>(xs);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    G.wfi = mfma4(dT[0][c], sT[0][c], G.wfi); G.wgi = mfma4(dT[1][c], sT[0][c], G.wgi);
                    G.wfh = mfma4(dT[0][c], sT[1][c], G.wfh); G.wgh = mfma4(dT[1][c], sT[1][c], G.wgh);
                    const float* xr = xf + 2 * ((4 * q + c) * kBojRow + tt + 1 + (n >> 1)) + (n & 1);
                    G.fir[0] = mfma4(dT[2][c], xr[0], G.fir[0]);
                    G.fir[1] = mfma4(dT[2][c], xr[16], G.fir[1]);
                }
            }
        }
    }
}

// raw: 16 x 32 scratch for the FIR tile (rows (filter, re | im), columns (tap, I | Q))
__device__ __forceinline__ void b16_write_row(float* prow, float* raw, const BojLayout& L, B16Grad& G, int lane, int n, int q) {
    const int H = L.H;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int o = 4 * q + rr;
        raw[o * 32 + n] = G.fir[0][rr]; raw[o * 32 + 16 + n] = G.fir[1][rr];
        if (o < H && n < 12) {
            prow[L.o_wfi + o * 2 * kBojP + b16_env(n >> 2, n & 3)] = G.wfi[rr];
            prow[L.o_wgi + o * 2 * kBojP + b16_env(n >> 2, n & 3)] = G.wgi[rr];
        }
        if (o < H && n < H) { prow[L.o_wfh + o * H + n] = G.wfh[rr]; prow[L.o_wgh + o * H + n] = G.wgh[rr]; }
        const float b0 = row_sum16(G.dbf[rr]), b1 = row_sum16(G.dbg[rr]), w0 = row_sum16(G.dwi[rr]), w1 = row_sum16(G.dwq[rr]);
        if (n == 0 && o < H) { prow[L.o_bfi + o] = b0; prow[L.o_bgi + o] = b1; prow[L.o_woi + o] = w0; prow[L.o_woq + o] = w1; }
    }
    const float d0 = row_sum16(G.dbo[0]), d1 = row_sum16(G.dbo[1]);      // accumulated on the q == 0 lanes only
    if (lane == 0) { prow[L.o_boi] = d0; prow[L.o_boq] = d1; }
    wave_lds_fence();
    // d bI[p][m] = G[(p,re)][(m,I)] + G[(p,im)][(m,Q)],  d bQ[p][m] = G[(p,im)][(m,I)] - G[(p,re)][(m,Q)]
    for (int i = lane; i < kBojP * kBojM; i += 64) {
        const int p = i >> 4, m = i & 15, rre = 4 * (p >> 1) + 2 * (p & 1), cI = 16 * (m >> 3) + 2 * (m & 7);
        prow[L.o_bi + i] = raw[rre * 32 + cI] + raw[(rre + 1) * 32 + cI + 1];
        prow[L.o_bq + i] = raw[(rre + 1) * 32 + cI] - raw[rre * 32 + cI + 1];
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(256, 1) void boj16_bwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride;
    constexpr int kWave = (DX ? 2 : 1) * 2 * 16 * kBojRow + 2 * 16 * kChunkPad + (NW ? B16::kTiles * kTileFloats : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const BojLayout L = boj_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    b16_build_table(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float* wbase = tab + s16_tab_floats(B16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * kBojRow;
    float2* dxs = dys + 16 * kChunkPad;
    float* tiles = reinterpret_cast<float*>(dxs + (DX ? 16 * kBojRow : 0));
    B16Grad G;
    G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * 64 + lane;
        f32x4 dh = {0.f, 0.f, 0.f, 0.f};
        int cur_chunk = -1;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        const int pt0 = cur_chunk * kChunk;
                        wave_lds_fence();
                        b16_stage_out(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    }
                    wave_lds_fence();
                    b16_dx_carry(dxs, lane, cur_chunk < 0);
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                b16_stage_in(xs, a.x, b0, a.B, a.T, t0, len, lane);
                stage_in<16>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 h0 = blk ? as_f32x4(ck[(size_t)blk * 64]) : z4;
            if (nstep == S) b16_bwd_block<NW, DX, true>(tl, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, h0, dh);
            else b16_bwd_block<NW, DX, false>(tl, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, h0, dh);
        }
        if constexpr (DX) {
            if (cur_chunk >= 0) {
                const int pt0 = cur_chunk * kChunk;
                wave_lds_fence();
                b16_stage_out(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                wave_lds_fence();
            }
        }
    }
    if constexpr (NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        b16_write_row(smem + wave * P4, smem + nwb * P4 + wave * 512, L, G, lane, n, q);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

LaunchShape b16_shape(int ngroups, int waves) {
    LaunchShape ls;
    ls.waves = waves;
    const int need = (ngroups + waves - 1) / waves, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
template <bool NW, bool DX>
int b16_launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = b16_shape(a.ngroups, 4);
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(B16::NG) +
                  (size_t)ls.waves * ((DX ? 2 : 1) * 2 * 16 * kBojRow + 2 * 16 * kChunkPad + (NW ? B16::kTiles * kTileFloats : 0))) * sizeof(float);
    const size_t red = (size_t)ls.waves * (P + kLossCols + 512) * sizeof(float);
    if (NW && lds < red) lds = red;
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = boj16_bwd_kernel<NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}


// -------------------------------------------------------------------------------------------------
// Gate-parallel fused train kernel for the reference's own batch sizes (train_funcs.py:28-48: 64 .. 256 frames, every frame gets a SIMD of
// its own): ONE sequence per single-wave workgroup, only the recurrence in the step loops — everything that does not depend on h runs with
// lane = time step over the whole frame:
//   front     the FIR bank, the demodulator and the input halves of both gates (W_fi e + b, W_gi e + b) of every step;
//   forward   rows f | g | f | g of the wave: one rotated dot product with the row's own W_.h, one cross-row swap, h' on every row; f, g, h
//             of the frame parked in LDS;
//   head      phase re-rotation, both read-outs, loss, dL/dy, the read-outs' share of dL/dh(t) and dL/dcos, dL/dsin of every step;
//   backward  rows d_f | d_g | 0 | 0: one transposed rotated dot product, the step's two H x H weight gradients as ONE 4-block MFMA
//             (v_mfma_f32_16x16x1_4b_f32), d_f, d_g parked;
//   back end  dL/de through W_fi, W_gi, the demodulator's gradient (lane = time step); then the sums over time on the matrix pipe: W_fi |
//             b_fi, W_gi | b_gi as (d_f, d_g)^T [e, 1], both FIR banks as (d_fi, d_fq)^T [I window], [Q window].
// Weight gradients only (the frozen-PA role stays on the S16 kernels).  Taken while the frame's state fits the CU's LDS.
// -------------------------------------------------------------------------------------------------
constexpr int kBgpP16 = 17, kBgpP32 = 33;         // row pitches of the [time][unit] buffers: conflict-free for lane = unit AND for lane = time
struct BgpBuf { int xw, fiq, env, xs, fg, hist, dhh, dcs, dump, total; };
__host__ __device__ inline BgpBuf bgp_buf(int T) {
    const int Tp = (T + 3) & ~3;
    BgpBuf b; int o = 0;
    b.xw = o; o += 2 * (Tp + 16);            // float2 [16 + Tp]: index i <-> time i - 16 (zero before the frame)
    b.fiq = o; o += 12 * Tp;                 // [12][Tp]: fi_p, fq_p
    b.env = o; o += 16 * Tp;                 // [16][Tp]: mag_p, mag_p^2, 1, 0, 0, 0 (zero columns past the frame)
    b.xs = o; o += kBgpP32 * Tp;             // [Tp][33]: input halves of f | g; overwritten by d_f | d_g in the backward steps
    b.fg = o; o += kBgpP32 * Tp;             // f | g of step t
    b.hist = o; o += kBgpP16 * (Tp + 1);     // entry t + 1 = h(t), entry 0 = 0
    b.dhh = o; o += kBgpP16 * Tp;            // the read-outs' share of dL/dh(t)
    b.dcs = o; o += 12 * Tp;                 // [12][Tp]: dL/dcos_p, dL/dsin_p, then dL/dfi_p, dL/dfq_p
    b.dump = o; o += 512;
    b.total = o;
    return b;
}
__global__ __launch_bounds__(64) void boj_gp_train_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;
    const BojLayout L = boj_layout(a.H);
    const int H = L.H, T = a.T, Tp = (T + 3) & ~3;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    const BgpBuf O = bgp_buf(T);
    float* buf = smem + pad4(L.P);
    float2* xw = reinterpret_cast<float2*>(buf + O.xw);
    float *fiq = buf + O.fiq, *env = buf + O.env, *xs = buf + O.xs, *fg = buf + O.fg, *hist = buf + O.hist, *dhh = buf + O.dhh;
    float *dcs = buf + O.dcs, *dump = buf + O.dump;
    // the row's recurrent block (rows 0, 2: W_fh, rows 1, 3: W_gh) and its transpose (rows 0, 1 only), rotated for this lane
    float wF[16], wT[16];
    {
        const int dir = rot_dir(col), ow = (role & 1) ? L.o_wgh : L.o_wfh;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15;
            const bool ok = col < H && m < H;
            wF[k] = ok ? pl[ow + col * H + m] : 0.0f;
            wT[k] = (ok && role < 2) ? pl[ow + m * H + col] : 0.0f;
        }
    }
    const bool is_f = (role & 1) == 0;
    const RowMasks rm = row_masks();
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    const int xoff = (role & 1) * 16 + col;
    const int dmp = (int)(dump - smem) + lane;
    const int pk0 = role == 0 ? (int)(fg - smem) + col : role == 1 ? (int)(fg - smem) + 16 + col : role == 2 ? (int)(hist - smem) + kBgpP16 + col : dmp;
    const int pk_step = role < 2 ? kBgpP32 : role == 2 ? kBgpP16 : 0;
    const int dk0 = role < 2 ? (int)(xs - smem) + role * 16 + col : dmp;
    const int dk_step = role < 2 ? kBgpP32 : 0;
    if (lane < 16) { xw[lane] = make_float2(0.0f, 0.0f); hist[lane] = 0.0f; }
    if (lane == 0) hist[16] = 0.0f;

    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.0f;
    f32x4 accF = {0.f, 0.f, 0.f, 0.f}, accG = accF, accI = accF, accQ = accF;
    float dwi[16], dwq[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { dwi[j] = 0.0f; dwq[j] = 0.0f; }
    float dboi = 0.0f, dboq = 0.0f, loss_acc = 0.0f;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        wave_lds_fence();
        for (int t = lane; t < Tp; t += 64) xw[16 + t] = t < T ? xg[t] : make_float2(0.0f, 0.0f);
        wave_lds_fence();
        // ---- front: FIR bank, demodulator, input halves of the gates; lane = time step ----
        for (int t0 = 0; t0 < Tp; t0 += 64) {
            const int t = t0 + lane;
            if (t < Tp) {
                const bool live = t < T;
                float fi[kBojP], fq[kBojP];
#pragma unroll
                for (int p = 0; p < kBojP; ++p) { fi[p] = 0.0f; fq[p] = 0.0f; }
#pragma unroll
                for (int m = 0; m < kBojM; ++m) {
                    const float2 xv = xw[t + 1 + m];                             // time t - 15 + m
#pragma unroll
                    for (int p = 0; p < kBojP; ++p) {
                        const float bi = pl[L.o_bi + p * kBojM + m], bq = pl[L.o_bq + p * kBojM + m];
                        fi[p] += bi * xv.x - bq * xv.y;
                        fq[p] += bq * xv.x + bi * xv.y;
                    }
                }
                float e[2 * kBojP];
#pragma unroll
                for (int p = 0; p < kBojP; ++p) {
                    const float mag = sqrtf(fi[p] * fi[p] + fq[p] * fq[p]) + 1e-8f;
                    e[p] = mag; e[kBojP + p] = mag * mag;
                    fiq[p * Tp + t] = fi[p]; fiq[(kBojP + p) * Tp + t] = fq[p];
                    env[p * Tp + t] = live ? mag : 0.0f; env[(kBojP + p) * Tp + t] = live ? mag * mag : 0.0f;
                }
                env[12 * Tp + t] = live ? 1.0f : 0.0f;
                env[13 * Tp + t] = 0.0f; env[14 * Tp + t] = 0.0f; env[15 * Tp + t] = 0.0f;
                for (int u = 0; u < 16; ++u) {
                    float pf = 0.0f, pg = 0.0f;
                    if (u < H) {
                        pf = pl[L.o_bfi + u]; pg = pl[L.o_bgi + u];
#pragma unroll
                        for (int k = 0; k < 2 * kBojP; ++k) {
                            pf = __builtin_fmaf(pl[L.o_wfi + u * 2 * kBojP + k], e[k], pf);
                            pg = __builtin_fmaf(pl[L.o_wgi + u * 2 * kBojP + k], e[k], pg);
                        }
                    }
                    xs[t * kBgpP32 + u] = live ? pf : 0.0f;
                    xs[t * kBgpP32 + 16 + u] = live ? pg : 0.0f;
                }
            }
        }
        wave_lds_fence();
        // ---- forward recurrence ----
        {
            float h = 0.0f;
            int pk = pk0;
            for (int t = 0; t < T; ++t) {
                const float acc = rotdot(xs[t * kBgpP32 + xoff], wF, h);
                const float sg = sigmoidf_(acc), th = tanhf_(acc);
                const float v = is_f ? sg : th, o = xor16(v);
                const float f = is_f ? v : o, g = is_f ? o : v;
                h = __builtin_fmaf(f, h - g, g);
                smem[pk] = vsel(rm.m[0], f, vsel(rm.m[1], g, h));
                pk += pk_step;
            }
        }
        wave_lds_fence();
        // ---- phase re-rotation, read-outs, loss and dL/dy of every step; lane = time step ----
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
                float co[kBojP], si[kBojP], dco[kBojP], dsi[kBojP];
#pragma unroll
                for (int p = 0; p < kBojP; ++p) {
                    const float fi = fiq[p * Tp + t], fq = fiq[(kBojP + p) * Tp + t];
                    const float mag = sqrtf(fi * fi + fq * fq) + 1e-8f;
                    co[p] = fi / mag; si[p] = fq / mag; dco[p] = 0.0f; dsi[p] = 0.0f;
                }
                const float* hv = hist + (t + 1) * kBgpP16;
                float A = pl[L.o_boi], Bq = pl[L.o_boq];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < H) {
                        A = __builtin_fmaf(pl[L.o_woi + j], hv[j] * co[j % kBojP], A);
                        Bq = __builtin_fmaf(pl[L.o_woq + j], hv[j] * si[j % kBojP], Bq);
                    }
                const float2 tv = tg[t];
                float dy0, dy1;
                s16_loss(lossc, (A - Bq) - tv.x, (Bq + A) - tv.y, dy0, dy1, loss_acc);
                const float dA = dy0 + dy1, dB = dy1 - dy0;
                dboi += dA; dboq += dB;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float dh = 0.0f;
                    if (j < H) {
                        const int q = j % kBojP;
                        const float wi = pl[L.o_woi + j], wq = pl[L.o_woq + j], hj = hv[j];
                        dwi[j] = __builtin_fmaf(dA, hj * co[q], dwi[j]);
                        dwq[j] = __builtin_fmaf(dB, hj * si[q], dwq[j]);
                        dh = __builtin_fmaf(dA * wi, co[q], (dB * wq) * si[q]);
                        dco[q] = __builtin_fmaf(dA * wi, hj, dco[q]);
                        dsi[q] = __builtin_fmaf(dB * wq, hj, dsi[q]);
                    }
                    dhh[t * kBgpP16 + j] = dh;
                }
#pragma unroll
                for (int p = 0; p < kBojP; ++p) { dcs[p * Tp + t] = dco[p]; dcs[(kBojP + p) * Tp + t] = dsi[p]; }
            }
        }
        wave_lds_fence();
        // ---- backward recurrence ----
        {
            float carry = 0.0f;
            int dk = dk0 + (T - 1) * dk_step;
            for (int t = T - 1; t >= 0; --t) {
                const float hp = hist[t * kBgpP16 + col], f = fg[t * kBgpP32 + col], g = fg[t * kBgpP32 + 16 + col];
                const float gh = carry + dhh[t * kBgpP16 + col];
                const float dfp = (gh * (hp - g)) * (f * (1.0f - f));
                const float dgp = (gh * (1.0f - f)) * __builtin_fmaf(-g, g, 1.0f);
                const float d_row = vsel(rm.m[0], dfp, vsel(rm.m[1], dgp, 0.0f));
                float part = rotdot(0.0f, wT, d_row);
                part = sum_rows4(part);
                carry = __builtin_fmaf(gh, f, part);
                smem[dk] = d_row;
                dk -= dk_step;
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_row, hp, acc1, 0, 0, 0);
            }
        }
        wave_lds_fence();
        // ---- dL/de through W_fi, W_gi and the demodulator's gradient; lane = time step ----
        for (int t0 = 0; t0 < Tp; t0 += 64) {
            const int t = t0 + lane;
            if (t < Tp) {
                const bool live = t < T;
                float de[2 * kBojP];
#pragma unroll
                for (int k = 0; k < 2 * kBojP; ++k) de[k] = 0.0f;
                for (int u = 0; u < H; ++u) {
                    const float df = xs[t * kBgpP32 + u], dg = xs[t * kBgpP32 + 16 + u];
#pragma unroll
                    for (int k = 0; k < 2 * kBojP; ++k) {
                        de[k] = __builtin_fmaf(pl[L.o_wfi + u * 2 * kBojP + k], df, de[k]);
                        de[k] = __builtin_fmaf(pl[L.o_wgi + u * 2 * kBojP + k], dg, de[k]);
                    }
                }
#pragma unroll
                for (int p = 0; p < kBojP; ++p) {
                    const float fi = fiq[p * Tp + t], fq = fiq[(kBojP + p) * Tp + t];
                    const float m0 = sqrtf(fi * fi + fq * fq), mag = m0 + 1e-8f;
                    const float dco = dcs[p * Tp + t], dsi = dcs[(kBojP + p) * Tp + t];
                    const float dmag = de[p] + 2.0f * mag * de[kBojP + p] - (dsi * fq + dco * fi) / (mag * mag);
                    const float im0 = m0 > 0.0f ? 1.0f / m0 : 0.0f;            // (a filter output of exactly 0: the term is dropped, as in the S16 kernels)
                    const float dfi = dco / mag + dmag * fi * im0, dfq = dsi / mag + dmag * fq * im0;
                    dcs[p * Tp + t] = live ? dfi : 0.0f; dcs[(kBojP + p) * Tp + t] = live ? dfq : 0.0f;
                }
            }
        }
        wave_lds_fence();
        // ---- the sums over time on the matrix pipe: lane (i = col, k = role) feeds A[i][k], B[k][col] of a 4-step slice ----
        {
            const int arow = (col < 12 ? col : 11) * Tp;
            const float amask = col < 12 ? 1.0f : 0.0f;
            for (int t = role; t < Tp; t += 4) {
                const float af = xs[t * kBgpP32 + col], ag = xs[t * kBgpP32 + 16 + col], be = env[col * Tp + t];
                accF = mfma4(af, be, accF);
                accG = mfma4(ag, be, accG);
                const float ad = dcs[arow + t] * amask;
                const float2 xv = xw[t + 1 + col];                               // tap m = col of step t: time t - 15 + m
                accI = mfma4(ad, xv.x, accI);
                accQ = mfma4(ad, xv.y, accQ);
            }
        }
    }
    // ---- the workgroup's row of partial gradients (every entry written) ----
    float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
    wave_lds_fence();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = 4 * role + rr;
        dump[i * 16 + col] = accI[rr];
        dump[256 + i * 16 + col] = accQ[rr];
        if (i < H) {
            if (col < 2 * kBojP) { prow[L.o_wfi + i * 2 * kBojP + col] = accF[rr]; prow[L.o_wgi + i * 2 * kBojP + col] = accG[rr]; }
            else if (col == 2 * kBojP) { prow[L.o_bfi + i] = accF[rr]; prow[L.o_bgi + i] = accG[rr]; }
            // 4-block MFMA: block 0 = d_f (x) h(t-1), block 1 = d_g (x) h(t-1); register 4 blk + rr of lane l = entry (4 (l / 16) + rr, l % 16)
            if (col < H) { prow[L.o_wfh + i * H + col] = acc1[rr]; prow[L.o_wgh + i * H + col] = acc1[4 + rr]; }
        }
    }
    wave_lds_fence();
    for (int idx = lane; idx < kBojP * kBojM; idx += 64) {
        const int p = idx / kBojM, m = idx % kBojM;
        prow[L.o_bi + idx] = dump[p * 16 + m] + dump[256 + (kBojP + p) * 16 + m];
        prow[L.o_bq + idx] = dump[(kBojP + p) * 16 + m] - dump[256 + p * 16 + m];
    }
    float lp = loss_acc, s0 = dboi, s1 = dboq;
    for (int o = 32; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float vi = dwi[j], vq = dwq[j];
        for (int o = 32; o > 0; o >>= 1) { vi += __shfl_xor(vi, o); vq += __shfl_xor(vq, o); }
        if (lane == 0 && j < H) { prow[L.o_woi + j] = vi; prow[L.o_woq + j] = vq; }
    }
    if (lane == 0) {
        prow[L.o_boi] = s0; prow[L.o_boq] = s1;
        prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
    }
}

// Evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): ONE sequence per wave, the forward half of
// boj_gp_train_kernel in chunks of kBevChunk steps (front with lane = time step, the recurrence, the read-outs with lane = time step); no checkpoints.
constexpr int kBevChunk = 256;
constexpr int kBevFloats = 2 * (kBevChunk + 16) + 12 * kBevChunk + kBgpP32 * kBevChunk + kBgpP16 * (kBevChunk + 1) + 64;
__global__ __launch_bounds__(64) void boj_gp_eval_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int Tc = kBevChunk;
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;
    const BojLayout L = boj_layout(a.H);
    const int H = L.H, T = a.T;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* buf = smem + pad4(L.P);
    float2* xw = reinterpret_cast<float2*>(buf);                    // [16 + Tc]: index i <-> time t0 - 16 + i
    float* fiq = buf + 2 * (Tc + 16);                                // [12][Tc]
    float* xs = fiq + 12 * Tc;                                       // [Tc][33]
    float* hist = xs + kBgpP32 * Tc;                                 // [Tc + 1][17]: entry tt + 1 = h(t0 + tt)
    float* dump = hist + kBgpP16 * (Tc + 1);
    float wF[16];
    {
        const int dir = rot_dir(col), ow = (role & 1) ? L.o_wgh : L.o_wfh;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15;
            wF[k] = (col < H && m < H) ? pl[ow + col * H + m] : 0.0f;
        }
    }
    const bool is_f = (role & 1) == 0;
    const int xoff = (role & 1) * 16 + col;
    const int pk0 = role == 0 ? (int)(hist - smem) + kBgpP16 + col : (int)(dump - smem) + lane, pk_step = role == 0 ? kBgpP16 : 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float h = 0.0f;
        for (int t0 = 0; t0 < T; t0 += Tc) {
            const int len = min(Tc, T - t0);
            wave_lds_fence();
            for (int i = lane; i < len + 16; i += 64) {
                const int t = t0 - 16 + i;
                xw[i] = t >= 0 ? xg[t] : make_float2(0.0f, 0.0f);
            }
            wave_lds_fence();
            for (int tt = lane; tt < len; tt += 64) {
                float fi[kBojP], fq[kBojP];
#pragma unroll
                for (int p = 0; p < kBojP; ++p) { fi[p] = 0.0f; fq[p] = 0.0f; }
#pragma unroll
                for (int m = 0; m < kBojM; ++m) {
                    const float2 xv = xw[tt + 1 + m];
#pragma unroll
                    for (int p = 0; p < kBojP; ++p) {
                        const float bi = pl[L.o_bi + p * kBojM + m], bq = pl[L.o_bq + p * kBojM + m];
                        fi[p] += bi * xv.x - bq * xv.y;
                        fq[p] += bq * xv.x + bi * xv.y;
                    }
                }
                float e[2 * kBojP];
#pragma unroll
                for (int p = 0; p < kBojP; ++p) {
                    const float mag = sqrtf(fi[p] * fi[p] + fq[p] * fq[p]) + 1e-8f;
                    e[p] = mag; e[kBojP + p] = mag * mag;
                    fiq[p * Tc + tt] = fi[p]; fiq[(kBojP + p) * Tc + tt] = fq[p];
                }
                for (int u = 0; u < 16; ++u) {
                    float pf = 0.0f, pg = 0.0f;
                    if (u < H) {
                        pf = pl[L.o_bfi + u]; pg = pl[L.o_bgi + u];
#pragma unroll
                        for (int k = 0; k < 2 * kBojP; ++k) {
                            pf = __builtin_fmaf(pl[L.o_wfi + u * 2 * kBojP + k], e[k], pf);
                            pg = __builtin_fmaf(pl[L.o_wgi + u * 2 * kBojP + k], e[k], pg);
                        }
                    }
                    xs[tt * kBgpP32 + u] = pf;
                    xs[tt * kBgpP32 + 16 + u] = pg;
                }
            }
            wave_lds_fence();
            {
                int pk = pk0;
                for (int tt = 0; tt < len; ++tt) {
                    const float acc = rotdot(xs[tt * kBgpP32 + xoff], wF, h);
                    const float sg = sigmoidf_(acc), th = tanhf_(acc);
                    const float v = is_f ? sg : th, o = xor16(v);
                    const float f = is_f ? v : o, g = is_f ? o : v;
                    h = __builtin_fmaf(f, h - g, g);
                    smem[pk] = h;
                    pk += pk_step;
                }
            }
            wave_lds_fence();
            for (int tt = lane; tt < len; tt += 64) {
                float co[kBojP], si[kBojP];
#pragma unroll
                for (int p = 0; p < kBojP; ++p) {
                    const float fi = fiq[p * Tc + tt], fq = fiq[(kBojP + p) * Tc + tt];
                    const float mag = sqrtf(fi * fi + fq * fq) + 1e-8f;
                    co[p] = fi / mag; si[p] = fq / mag;
                }
                const float* hv = hist + (tt + 1) * kBgpP16;
                float A = pl[L.o_boi], Bq = pl[L.o_boq];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < H) {
                        A = __builtin_fmaf(pl[L.o_woi + j], hv[j] * co[j % kBojP], A);
                        Bq = __builtin_fmaf(pl[L.o_woq + j], hv[j] * si[j % kBojP], Bq);
                    }
                yg[t0 + tt] = make_float2(A - Bq, Bq + A);
            }
        }
    }
}

static size_t boj_gp_lds_bytes(int P, int T) { return ((size_t)pad4(P) + bgp_buf(T).total) * sizeof(float); }
static int boj_gp_blocks_per_cu(int P, int T) {
    const size_t lds = boj_gp_lds_bytes(P, T);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}

}  // namespace

bool bojanet_ok(const odpd_model_t* m) { return m->hidden >= 1 && m->hidden <= 16; }
int64_t bojanet_param_count(const odpd_model_t* m) {
    return bojanet_ok(m) ? (int64_t)boj_layout(m->hidden).P : (int64_t)ODPD_EUNSUPPORTED;
}
int bojanet_rows(const odpd_model_t* m, int B) {
    if (!bojanet_ok(m)) return ODPD_EUNSUPPORTED;
    return b16_shape((B + 15) / 16, 4).grid;
}
int64_t bojanet_ckpt_floats(const odpd_model_t* m, int B, int T) {
    if (!bojanet_ok(m)) return ODPD_EUNSUPPORTED;
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * 256;
}
// the gate-parallel fused train kernel: one sequence per single-wave workgroup, the frame's state in LDS
bool bojanet_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (!bojanet_ok(m) || T < kBojM - 1) return false;
    const int per_cu = boj_gp_blocks_per_cu(boj_layout(m->hidden).P, T);
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && per_cu > 0;
    // up to five rounds of workgroups (measured: profiles/r03/gp_train_bench_f4.txt): the alternative is the forward / loss / backward chain of the S16 kernels
    return (long)B <= 5L * device_cus() * per_cu;
}
int bojanet_gp_rows(const odpd_model_t* m, int B, int T) {
    const long cap = (long)device_cus() * boj_gp_blocks_per_cu(boj_layout(m->hidden).P, T);
    return B < cap ? B : (int)cap;
}
int bojanet_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const size_t lds = boj_gp_lds_bytes(boj_layout(m->hidden).P, a.T);
    if (int e = allow_big_lds(boj_gp_train_kernel, lds)) return e;
    hipLaunchKernelGGL(boj_gp_train_kernel, dim3(bojanet_gp_rows(m, a.B, a.T)), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
// mode 1 forward, 2 backward
int bojanet_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    if (!bojanet_ok(m)) return ODPD_EUNSUPPORTED;
    if (a0.T < kBojM - 1) return ODPD_EINVAL;       // the reference cuts its 15-sample zero pad from the frame itself (bojanet.py:72-73)
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = boj_layout(m->hidden).P;
    if (mode == 1 && !a.ckpt && a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0) {
        // sequences that each get a SIMD of their own (inference: no checkpoints)
        const size_t lds = ((size_t)pad4(P) + kBevFloats) * sizeof(float);
        if (int e = allow_big_lds(boj_gp_eval_kernel, lds)) return e;
        hipLaunchKernelGGL(boj_gp_eval_kernel, dim3(a.B), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    }
    if (mode == 1) {
        const LaunchShape ls = b16_shape(a.ngroups, a.ngroups <= 4 * device_cus() ? 4 : 8);
        const size_t lds = ((size_t)pad4(P) + s16_tab_floats(B16::NG) + (size_t)ls.waves * (2 * 16 * kBojRow + 2 * 16 * kChunkPad)) * sizeof(float);
        auto k = boj16_fwd_kernel;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    }
    if (!a.ckpt && a.nck > 1) return ODPD_EINVAL;
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (!nw && !dx) return ODPD_EINVAL;
    if (nw && dx) return b16_launch_bwd<true, true>(st, a, P);
    if (nw) return b16_launch_bwd<true, false>(st, a, P);
    return b16_launch_bwd<false, true>(st, a, P);
}

}  // namespace odpd
