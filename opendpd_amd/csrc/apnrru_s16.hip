// apnrru_s16.hip — APNRRU (backbones/apnrru.py:5-152) in the S16 mapping (see gru_s16.hip / odpd_s16.h; the FIR / halo / dL/dx machinery
// is bojanet_s16.hip's): a wave = 16 sequences, lane (n = sequence, q = quad) owns slots 4q + e of TWO 16-slot tiles that hold the cell's
// state vector s = [h_I (H), h_Q (H), h_A (3)], n = 2H + 3 <= 31:
//     tile 0: slots 0..H-1 = h_I,  slot 15 = h_A[0],  slot 14 = h_A[2]          tile 1: slots 0..H-1 = h_Q,  slot 15 = h_A[1]
// so that the complex rotations of the state (h <- h r before the cell, h <- conj(r) s' after it) pair tile 0 with tile 1 on the SAME lane
// element.  Per step (apnrru.py:66-128):
//   r = conj(x_t) / |x_t|;  three 16-tap complex FIR filters + the raw sample as a fourth value -> ONE 16 x 32 exact-fp32 MFMA product
//     (the raw sample is a filter whose only tap is 1 at lag 0), rotated by r -> 8 features on the lanes q = 0, 1;
//   v1 = tanh(W_u [feat, s] + b_u) (16 nodes = one tile);  v = tanh(W_h v1 + b_h);  s' = sigmoid(C s) + Z v  — MFMA mat-vecs from the LDS table;
//   y = (A - Bq, Bq + A), A = w_I . h_I, Bq = w_Q . h_Q.
// BPTT from checkpoints of the state every kCkptStride steps; dL/dx through the FIR bank (transposed, accumulated in an LDS frame chunk
// with halo), the raw sample and r.  Frames shorter than 15 samples are refused like the reference's framing (:68-72).
#include "odpd_s16.h"

namespace odpd {
namespace {

constexpr int kApnF = 3, kApnM = 16, kApnNode = 16;
constexpr int kApnHalo = 16;
constexpr int kApnRow = kApnHalo + kChunk + 1;      // float2 row stride of a staged frame chunk: index i <-> time t0 - 16 + i
struct ApnLayout { int H, n, o_bi, o_bq, o_c, o_z, o_wu, o_bu, o_wh, o_bh, o_woi, o_woq, P; };
__host__ __device__ inline ApnLayout apn_layout(int H) {
    ApnLayout L; L.H = H; L.n = 2 * H + 3; int o = 0;
    L.o_bi = o; o += kApnF * kApnM; L.o_bq = o; o += kApnF * kApnM;
    L.o_c = o; o += 1; L.o_z = o; o += L.n;
    L.o_wu = o; o += kApnNode * (8 + L.n); L.o_bu = o; o += kApnNode;
    L.o_wh = o; o += L.n * kApnNode; L.o_bh = o; o += L.n;
    L.o_woi = o; o += H; L.o_woq = o; o += H;
    L.P = o;
    return L;
}
// index into s = [h_I, h_Q, h_A] of slot `slot` of tile `tile` (-1: empty)
__host__ __device__ inline int apn_unit(int H, int tile, int slot) {
    if (slot < H) return tile * H + slot;
    if (slot == 15) return 2 * H + tile;
    if (slot == 14 && tile == 0) return 2 * H + 2;
    return -1;
}

struct A16 {
    static constexpr int FIR = 0;          // jj   : A[row m][(tap q + 4 (2 jj + e / 2), e % 2)]
    static constexpr int WUX = FIR + 2;    //        W_u[m][4q+e]                     (features; q < 2)
    static constexpr int WUS = WUX + 1;    // kt   : W_u[m][8 + unit(kt, 4q+e)]
    static constexpr int WH = WUS + 2;     // mt   : W_h[unit(mt, m)][4q+e]
    static constexpr int WHT = WH + 2;     // kt   : W_h[unit(kt, 4q+e)][m]
    static constexpr int WUST = WHT + 2;   // mt   : W_u[4q+e][8 + unit(mt, m)]
    static constexpr int WUXT = WUST + 2;  //        W_u[4q+e][m]                     (m < 8)
    static constexpr int FIRT = WUXT + 1;  // tile : A[row 4q+e][(tap 8 tile + m / 2, m % 2)]
    static constexpr int BU = FIRT + 2;    //        b_u[4q+e]
    static constexpr int BH = BU + 1;      // mt   : b_h[unit(mt, 4q+e)]
    static constexpr int Z = BH + 2;       // mt   : Z[unit(mt, 4q+e)]
    static constexpr int WO = Z + 2;       // mt   : w_I | w_Q at slot 4q+e (< H)
    static constexpr int NG = WO + 2;
    static constexpr int kTiles = 8;       // dpre2[0] dpre2[1] dpre1 dF | v1 sp[0] sp[1] feat
};
// the FIR bank + raw sample as a real 16 x 32 matrix: row i = 4 qi + r; qi 0: (f0 re, f0 im, f1 re, f1 im), qi 1: (f2 re, f2 im, x_I, x_Q)
__device__ __forceinline__ float a16_fir(const float* pl, const ApnLayout& L, int i, int tap, int c) {
    if (i >= 8) return 0.0f;
    if (i >= 6) return (tap == kApnM - 1 && c == i - 6) ? 1.0f : 0.0f;
    const int p = i >> 1;
    const float bi = pl[L.o_bi + p * kApnM + tap], bq = pl[L.o_bq + p * kApnM + tap];
    return (i & 1) == 0 ? (c == 0 ? bi : -bq) : (c == 0 ? bq : bi);
}
__device__ __forceinline__ float4 a16_entry(const float* pl, const ApnLayout& L, int grp, int m, int q) {
    const int H = L.H, W = 8 + L.n;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 4 * q + e;
        float s = 0.0f;
        if (grp < A16::WUX) s = a16_fir(pl, L, m, q + 4 * (2 * (grp - A16::FIR) + (e >> 1)), e & 1);
        else if (grp == A16::WUX) s = k < 8 ? pl[L.o_wu + m * W + k] : 0.0f;
        else if (grp < A16::WH) { const int u = apn_unit(H, grp - A16::WUS, k); s = u >= 0 ? pl[L.o_wu + m * W + 8 + u] : 0.0f; }
        else if (grp < A16::WHT) { const int u = apn_unit(H, grp - A16::WH, m); s = u >= 0 ? pl[L.o_wh + u * kApnNode + k] : 0.0f; }
        else if (grp < A16::WUST) { const int u = apn_unit(H, grp - A16::WHT, k); s = u >= 0 ? pl[L.o_wh + u * kApnNode + m] : 0.0f; }
        else if (grp < A16::WUXT) { const int u = apn_unit(H, grp - A16::WUST, m); s = u >= 0 ? pl[L.o_wu + k * W + 8 + u] : 0.0f; }
        else if (grp == A16::WUXT) s = m < 8 ? pl[L.o_wu + k * W + m] : 0.0f;
        else if (grp < A16::BU) s = a16_fir(pl, L, k, 8 * (grp - A16::FIRT) + (m >> 1), m & 1);
        else if (grp == A16::BU) s = pl[L.o_bu + k];
        else if (grp < A16::Z) { const int u = apn_unit(H, grp - A16::BH, k); s = u >= 0 ? pl[L.o_bh + u] : 0.0f; }
        else if (grp < A16::WO) { const int u = apn_unit(H, grp - A16::Z, k); s = u >= 0 ? pl[L.o_z + u] : 0.0f; }
        else s = k < H ? pl[(grp == A16::WO ? L.o_woi : L.o_woq) + k] : 0.0f;
        v[e] = s;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void a16_build_table(float* tab, const float* pl, const ApnLayout& L, int lane, int wave, int nwb) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < A16::NG; grp += nwb) t4[grp * 64 + lane] = a16_entry(pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}
__device__ __forceinline__ float a16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x4 a16_mv(TabPtr tl, int grp, const f32x4& v, f32x4 acc) {
    f32x4 a1[1] = {acc};
    const f32x4 v1[1] = {v};
    s16n_matvec<1>(tl, grp, v1, a1);
    return a1[0];
}

// frame chunk with its halo: 16 sequences x 48 samples, times t0 - 16 .. t0 + 31 (zeros before the frame: apnrru.py:68-69)
__device__ __forceinline__ void a16_stage_in(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int W = kApnHalo + kChunk;
#pragma unroll
    for (int j = 0; j < 16 * W / 64; ++j) {
        const int e = lane + 64 * j, m = e / W, i = e % W, t = t0 - kApnHalo + i;
        float2 v = make_float2(0.0f, 0.0f);
        if (b0 + m >= B || t >= t0 + len) v = make_float2(0.5f, 0.25f);     // idle sequence slots / steps: a finite signal with |x| > 0
        else if (t >= 0) v = g2[(size_t)(b0 + m) * T + t];
        lds[m * kApnRow + i] = v;
    }
}
__device__ __forceinline__ void a16_stage_out(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
#pragma unroll
    for (int j = 0; j < 16 * kChunk / 64; ++j) {
        const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
        if (tt < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + tt] = lds[m * kApnRow + kApnHalo + tt];
    }
}
// dL/dx chunk hand-over (backward runs the chunks last to first): what the finished chunk put before its own t0 (indices 1..15)
// belongs to the end of the next (earlier) one (indices 33..47); everything else restarts at 0
__device__ __forceinline__ void a16_dx_carry(float2* lds, int lane, bool first) {
    float2 c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j, m = e >> 4, i = e & 15;
        c[j] = first ? make_float2(0.0f, 0.0f) : lds[m * kApnRow + i];
    }
    wave_lds_fence();
    for (int e = lane; e < 16 * kApnRow; e += 64) lds[e] = make_float2(0.0f, 0.0f);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j, m = e >> 4, i = e & 15;
        if (i) lds[m * kApnRow + kChunk + i] = c[j];
    }
    wave_lds_fence();
}

struct A16Phase { float I, Q, im, rr, ri; };      // r = conj(x) / |x| = (rr, ri)
__device__ __forceinline__ A16Phase a16_phase(float2 xv) {
    A16Phase P;
    P.I = xv.x; P.Q = xv.y;
    P.im = fast_rcp(__builtin_amdgcn_sqrtf(__builtin_fmaf(xv.x, xv.x, xv.y * xv.y)));
    P.rr = xv.x * P.im; P.ri = -xv.y * P.im;
    return P;
}
// FIR bank + raw sample (8 MFMAs) at local step tt: window = staged indices tt + 1 .. tt + 16
__device__ __forceinline__ f32x4 a16_fir_fwd(TabPtr tl, const float2* xrow, int tt, int q) {
    f32x4 ff = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const float4 w = tab_ld(tl, (A16::FIR + jj) * 64);
        const float2 x0 = xrow[tt + 1 + q + 8 * jj], x1 = xrow[tt + 1 + q + 8 * jj + 4];
        ff = mfma4(w.x, x0.x, ff); ff = mfma4(w.y, x0.y, ff);
        ff = mfma4(w.z, x1.x, ff); ff = mfma4(w.w, x1.y, ff);
    }
    return ff;
}
__device__ __forceinline__ f32x4 a16_features(const f32x4& ff, const A16Phase& P) {
    return f32x4{__builtin_fmaf(P.rr, ff[0], -P.ri * ff[1]), __builtin_fmaf(P.ri, ff[0], P.rr * ff[1]),
                 __builtin_fmaf(P.rr, ff[2], -P.ri * ff[3]), __builtin_fmaf(P.ri, ff[2], P.rr * ff[3])};
}
// (a + j b) (rr + j ri) on the slots that hold the complex state (rot[e] = 1), identity elsewhere
__device__ __forceinline__ void a16_rotate(const f32x4& a, const f32x4& b, float rr, float ri, const float (&rot)[4], f32x4& oa, f32x4& ob) {
    ODPD_EACH4 {
        const float ra = __builtin_fmaf(a[i], rr, -b[i] * ri), rb = __builtin_fmaf(a[i], ri, b[i] * rr);
        oa[i] = rot[i] != 0.0f ? ra : a[i];
        ob[i] = rot[i] != 0.0f ? rb : b[i];
    }
}
// the RRU cell on the normalised state sp: v1, v, sigmoid(C sp) and s' = sigmoid(C sp) + Z v
__device__ __forceinline__ void a16_cell(TabPtr tl, float Cn, const f32x4& feat, const f32x4 (&sp)[2], f32x4& v1, f32x4 (&v)[2], f32x4 (&sg)[2], f32x4 (&sn)[2]) {
    f32x4 p1 = as_f32x4(tab_ld(tl, A16::BU * 64));
    p1 = a16_mv(tl, A16::WUX, feat, p1);
    p1 = a16_mv(tl, A16::WUS + 0, sp[0], p1);
    p1 = a16_mv(tl, A16::WUS + 1, sp[1], p1);
    v1 = tanh4_precise(p1);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x4 p2 = a16_mv(tl, A16::WH + mt, v1, as_f32x4(tab_ld(tl, (A16::BH + mt) * 64)));
        v[mt] = tanh4_precise(p2);
        sg[mt] = sigmoid4_prescaled(mul4(splat4(Cn), sp[mt]));
        sn[mt] = fma4(as_f32x4(tab_ld(tl, (A16::Z + mt) * 64)), v[mt], sg[mt]);
    }
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void apn16_fwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride, kWave = 2 * 16 * kApnRow + 2 * 16 * kChunkPad;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const ApnLayout L = apn_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    a16_build_table(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const float Cn = a16_uni(pl[L.o_c]) * kNegLog2e;
    float rot[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rot[e] = 4 * q + e < a.H ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(A16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * kApnRow;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * 2 * 64 + lane : nullptr;   // [ckpt][tile][lane]
        f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            a16_stage_in(xs, a.x, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const TabPtr tk = opaque(tl);
                const float2* xrow = xs + n * kApnRow;
                const A16Phase P = a16_phase(xrow[tt + kApnHalo]);
                const f32x4 feat = a16_features(a16_fir_fwd(tk, xrow, tt, q), P);
                f32x4 sp[2], v1, v[2], sg[2], sn[2];
                a16_rotate(s[0], s[1], P.rr, P.ri, rot, sp[0], sp[1]);
                a16_cell(tk, Cn, feat, sp, v1, v, sg, sn);
                a16_rotate(sn[0], sn[1], P.rr, -P.ri, rot, s[0], s[1]);          // back: times conj(r)
                const f32x4 wi = as_f32x4(tab_ld(tk, (A16::WO + 0) * 64)), wq = as_f32x4(tab_ld(tk, (A16::WO + 1) * 64));
                float s0 = 0.0f, s1 = 0.0f;
                ODPD_EACH4 { s0 = __builtin_fmaf(wi[i], s[0][i], s0); s1 = __builtin_fmaf(wq[i], s[1][i], s1); }
                const float A = quad_sum(s0), Bq = quad_sum(s1);
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(A - Bq, Bq + A);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    ck[((size_t)(t1 / S) * 2) * 64] = make_float4(s[0][0], s[0][1], s[0][2], s[0][3]);
                    ck[((size_t)(t1 / S) * 2 + 1) * 64] = make_float4(s[1][0], s[1][1], s[1][2], s[1][3]);
                }
            }
            wave_lds_fence();
            stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------
struct A16Grad {
    f32x4 fir[2], wux, wus[2], wh[2];      // weight-gradient tiles
    f32x4 dbu, dbh[2], dz[2], dwo[2];
    float dc;
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        fir[0] = fir[1] = wux = wus[0] = wus[1] = wh[0] = wh[1] = dbu = dbh[0] = dbh[1] = dz[0] = dz[1] = dwo[0] = dwo[1] = z4;
        dc = 0.f;
    }
};

template <bool NW, bool DX, bool FULL>
__device__ __forceinline__ void a16_bwd_block(TabPtr tl0, float C, const float (&rot)[4], A16Grad& G, const float2* xs, const float2* dys, float2* dxs,
                                              float* tiles, int n, int q, int tloc, int nstep, f32x4 s0, f32x4 s1, f32x4 (&ds)[2]) {
    constexpr int S = kCkptStride;
    const float Cn = C * kNegLog2e;
    f32x4 sa_s[S], sb_s[S], v1_s[S], va_s[S], vb_s[S], ff_s[S];        // state before the step (both tiles), v1, v (both tiles), FIR outputs
    const float2* xrow = xs + n * kApnRow;
#pragma unroll
    for (int si = 0; si < S; ++si) {
        if (FULL || si < nstep) {
            const TabPtr tk = opaque(tl0);
            const A16Phase P = a16_phase(xrow[tloc + si + kApnHalo]);
            ff_s[si] = a16_fir_fwd(tk, xrow, tloc + si, q);
            const f32x4 feat = a16_features(ff_s[si], P);
            sa_s[si] = s0; sb_s[si] = s1;
            f32x4 sp[2], v[2], sg[2], sn[2];
            a16_rotate(s0, s1, P.rr, P.ri, rot, sp[0], sp[1]);
            a16_cell(tk, Cn, feat, sp, v1_s[si], v, sg, sn);
            va_s[si] = v[0]; vb_s[si] = v[1];
            a16_rotate(sn[0], sn[1], P.rr, -P.ri, rot, s0, s1);
        }
    }
    auto tile = [tiles](int qty) { return tiles + qty * kTileFloats; };     // 0 dpre2[0] 1 dpre2[1] 2 dpre1 3 dF | 4 v1 5 sp[0] 6 sp[1] 7 feat
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            const TabPtr tl = opaque(tl0);
            const int tt = tloc + si;
            const float2 dyv = dys[n * kChunkPad + tt];
            const float dA = dyv.x + dyv.y, dB = dyv.y - dyv.x;          // y = (A - Bq, Bq + A)
            const A16Phase P = a16_phase(xrow[tt + kApnHalo]);
            const f32x4 ff = ff_s[si], v1 = v1_s[si], feat = a16_features(ff, P);
            const f32x4 v[2] = {va_s[si], vb_s[si]};
            f32x4 sp[2], sg[2], sn[2], so[2];
            a16_rotate(sa_s[si], sb_s[si], P.rr, P.ri, rot, sp[0], sp[1]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                sg[mt] = sigmoid4_prescaled(mul4(splat4(Cn), sp[mt]));
                sn[mt] = fma4(as_f32x4(tab_ld(tl, (A16::Z + mt) * 64)), v[mt], sg[mt]);
            }
            a16_rotate(sn[0], sn[1], P.rr, -P.ri, rot, so[0], so[1]);       // the step's output state
            const f32x4 wi = as_f32x4(tab_ld(tl, (A16::WO + 0) * 64)), wq = as_f32x4(tab_ld(tl, (A16::WO + 1) * 64));
            if constexpr (NW) {
                G.dwo[0] = fma4(splat4(dA), so[0], G.dwo[0]);     // (slots >= H: dropped at write-out)
                G.dwo[1] = fma4(splat4(dB), so[1], G.dwo[1]);
            }
            // through the rotation back: so = sn conj(r)
            float drr = 0.0f, dri = 0.0f;
            f32x4 dsn[2];
            ODPD_EACH4 {
                const float gI = __builtin_fmaf(dA, wi[i], ds[0][i]), gQ = __builtin_fmaf(dB, wq[i], ds[1][i]);
                const bool r = rot[i] != 0.0f;
                dsn[0][i] = r ? __builtin_fmaf(gI, P.rr, -gQ * P.ri) : gI;
                dsn[1][i] = r ? __builtin_fmaf(gI, P.ri, gQ * P.rr) : gQ;
                drr += r ? __builtin_fmaf(gI, sn[0][i], gQ * sn[1][i]) : 0.0f;
                dri += r ? __builtin_fmaf(gI, sn[1][i], -gQ * sn[0][i]) : 0.0f;
            }
            f32x4 dsp[2], dpre2[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const f32x4 zt = as_f32x4(tab_ld(tl, (A16::Z + mt) * 64));
                ODPD_EACH4 {
                    const float dsg = sg[mt][i] * (1.0f - sg[mt][i]);
                    if constexpr (NW) {
                        G.dz[mt][i] = __builtin_fmaf(dsn[mt][i], v[mt][i], G.dz[mt][i]);
                        G.dc = __builtin_fmaf(dsn[mt][i] * dsg, sp[mt][i], G.dc);
                    }
                    dsp[mt][i] = dsn[mt][i] * dsg * C;
                    dpre2[mt][i] = dsn[mt][i] * zt[i] * __builtin_fmaf(-v[mt][i], v[mt][i], 1.0f);
                }
                if constexpr (NW) G.dbh[mt] = add4(G.dbh[mt], dpre2[mt]);
            }
            f32x4 dv1 = a16_mv(tl, A16::WHT + 0, dpre2[0], z4);
            dv1 = a16_mv(tl, A16::WHT + 1, dpre2[1], dv1);
            f32x4 dpre1;
            ODPD_EACH4 dpre1[i] = dv1[i] * __builtin_fmaf(-v1[i], v1[i], 1.0f);
            if constexpr (NW) G.dbu = add4(G.dbu, dpre1);
            dsp[0] = a16_mv(tl, A16::WUST + 0, dpre1, dsp[0]);
            dsp[1] = a16_mv(tl, A16::WUST + 1, dpre1, dsp[1]);
            const f32x4 dfe = a16_mv(tl, A16::WUXT, dpre1, z4);          // d feat on the lanes q = 0, 1 (rows 8..15 of the table are empty)
            // the rotation into the normalised frame: sp = s r
            ODPD_EACH4 {
                const bool r = rot[i] != 0.0f;
                const float a0 = dsp[0][i], a1 = dsp[1][i];
                ds[0][i] = r ? __builtin_fmaf(a0, P.rr, a1 * P.ri) : a0;
                ds[1][i] = r ? __builtin_fmaf(a1, P.rr, -a0 * P.ri) : a1;
                drr += r ? __builtin_fmaf(a0, sa_s[si][i], a1 * sb_s[si][i]) : 0.0f;
                dri += r ? __builtin_fmaf(a1, sa_s[si][i], -a0 * sb_s[si][i]) : 0.0f;
            }
            // features: feat = (fi + j fq) r
            f32x4 dF;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float de = dfe[2 * k], dod = dfe[2 * k + 1], fi = ff[2 * k], fq = ff[2 * k + 1];
                dF[2 * k] = __builtin_fmaf(de, P.rr, dod * P.ri);
                dF[2 * k + 1] = __builtin_fmaf(dod, P.rr, -de * P.ri);
                drr += __builtin_fmaf(de, fi, dod * fq);
                dri += __builtin_fmaf(dod, fi, -de * fq);
            }
            if constexpr (DX) {
                // r = (I, -Q) / |x|:  dI = Q w, dQ = -I w with w = (drr Q + dri I) / |x|^3
                const float drt = quad_sum(drr), dit = quad_sum(dri);
                const float w = __builtin_fmaf(drt, P.Q, dit * P.I) * (P.im * P.im * P.im);
#pragma unroll
                for (int tile_i = 0; tile_i < 2; ++tile_i) {
                    const f32x4 dw = a16_mv(tl, A16::FIRT + tile_i, dF, z4);
                    float2* d = dxs + n * kApnRow + tt + 1 + 8 * tile_i + 2 * q;
                    float2 u0 = d[0], u1 = d[1];
                    u0.x += dw[0]; u0.y += dw[1]; u1.x += dw[2]; u1.y += dw[3];
                    if (tile_i == 1 && q == 3) { u1.x = __builtin_fmaf(P.Q, w, u1.x); u1.y = __builtin_fmaf(-P.I, w, u1.y); }     // lag 0 = this sample
                    d[0] = u0; d[1] = u1;
                }
            }
            if constexpr (NW) {
                wave_lds_fence();
                tile_put(tile(0), n, q, dpre2[0]); tile_put(tile(1), n, q, dpre2[1]); tile_put(tile(2), n, q, dpre1); tile_put(tile(3), n, q, dF);
                tile_put(tile(4), n, q, v1); tile_put(tile(5), n, q, sp[0]); tile_put(tile(6), n, q, sp[1]); tile_put(tile(7), n, q, feat);
                wave_lds_fence();
                float dT[4][4], sT[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { tile_get(tile(j), n, q, dT[j]); tile_get(tile(4 + j), n, q, sT[j]); }
                const float* xf = reinterpret_cast<const float*>(xs);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    G.wh[0] = mfma4(dT[0][c], sT[0][c], G.wh[0]); G.wh[1] = mfma4(dT[1][c], sT[0][c], G.wh[1]);       // W_h : dpre2 (x) v1
                    G.wus[0] = mfma4(dT[2][c], sT[1][c], G.wus[0]); G.wus[1] = mfma4(dT[2][c], sT[2][c], G.wus[1]);   // W_u|s: dpre1 (x) sp
                    G.wux = mfma4(dT[2][c], sT[3][c], G.wux);                                                          // W_u|x: dpre1 (x) feat
                    const float* xr = xf + 2 * ((4 * q + c) * kApnRow + tt + 1 + (n >> 1)) + (n & 1);               // FIR: dF (x) window
                    G.fir[0] = mfma4(dT[3][c], xr[0], G.fir[0]);
                    G.fir[1] = mfma4(dT[3][c], xr[16], G.fir[1]);
                }
            }
        }
    }
}

// raw: 16 x 32 scratch for the FIR tile (rows (filter, re | im), columns (tap, I | Q))
__device__ __forceinline__ void a16_write_row(float* prow, float* raw, const ApnLayout& L, A16Grad& G, int lane, int n, int q) {
    const int H = L.H, W = 8 + L.n;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int o = 4 * q + rr;                                   // row of the tile: a node, or a slot
        raw[o * 32 + n] = G.fir[0][rr]; raw[o * 32 + 16 + n] = G.fir[1][rr];
        if (n < 8) prow[L.o_wu + o * W + n] = G.wux[rr];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int uc = apn_unit(H, t, n), ur = apn_unit(H, t, o);
            if (uc >= 0) prow[L.o_wu + o * W + 8 + uc] = G.wus[t][rr];
            if (ur >= 0) prow[L.o_wh + ur * kApnNode + n] = G.wh[t][rr];
            const float bh = row_sum16(G.dbh[t][rr]), dz = row_sum16(G.dz[t][rr]), wo = row_sum16(G.dwo[t][rr]);
            if (n == 0 && ur >= 0) { prow[L.o_bh + ur] = bh; prow[L.o_z + ur] = dz; }
            if (n == 0 && o < H) prow[(t == 0 ? L.o_woi : L.o_woq) + o] = wo;
        }
        const float bu = row_sum16(G.dbu[rr]);
        if (n == 0) prow[L.o_bu + o] = bu;
    }
    float dc = G.dc;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dc += __shfl_xor(dc, m);
    if (lane == 0) prow[L.o_c] = dc;
    wave_lds_fence();
    // d bI[p][m] = G[(p,re)][(m,I)] + G[(p,im)][(m,Q)],  d bQ[p][m] = G[(p,im)][(m,I)] - G[(p,re)][(m,Q)]
    for (int i = lane; i < kApnF * kApnM; i += 64) {
        const int p = i >> 4, m = i & 15, rre = 2 * p, cI = 16 * (m >> 3) + 2 * (m & 7);
        prow[L.o_bi + i] = raw[rre * 32 + cI] + raw[(rre + 1) * 32 + cI + 1];
        prow[L.o_bq + i] = raw[(rre + 1) * 32 + cI] - raw[rre * 32 + cI + 1];
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(256, 1) void apn16_bwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride;
    constexpr int kWave = (DX ? 2 : 1) * 2 * 16 * kApnRow + 2 * 16 * kChunkPad + (NW ? A16::kTiles * kTileFloats : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const ApnLayout L = apn_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    a16_build_table(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const float C = a16_uni(pl[L.o_c]);
    float rot[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) rot[e] = 4 * q + e < a.H ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(A16::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * kApnRow;
    float2* dxs = dys + 16 * kChunkPad;
    float* tiles = reinterpret_cast<float*>(dxs + (DX ? 16 * kApnRow : 0));
    A16Grad G;
    G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * 2 * 64 + lane;
        f32x4 ds[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        int cur_chunk = -1;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        const int pt0 = cur_chunk * kChunk;
                        wave_lds_fence();
                        a16_stage_out(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    }
                    wave_lds_fence();
                    a16_dx_carry(dxs, lane, cur_chunk < 0);
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                a16_stage_in(xs, a.x, b0, a.B, a.T, t0, len, lane);
                stage_in<16>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 sa = blk ? as_f32x4(ck[((size_t)blk * 2) * 64]) : z4, sb = blk ? as_f32x4(ck[((size_t)blk * 2 + 1) * 64]) : z4;
            if (nstep == S) a16_bwd_block<NW, DX, true>(tl, C, rot, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, sa, sb, ds);
            else a16_bwd_block<NW, DX, false>(tl, C, rot, G, xs, dys, dxs, tiles, n, q, tb - t0, nstep, sa, sb, ds);
        }
        if constexpr (DX) {
            if (cur_chunk >= 0) {
                const int pt0 = cur_chunk * kChunk;
                wave_lds_fence();
                a16_stage_out(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                wave_lds_fence();
            }
        }
    }
    if constexpr (NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        a16_write_row(smem + wave * P4, smem + nwb * P4 + wave * 512, L, G, lane, n, q);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

LaunchShape a16_shape(int ngroups, int waves) {
    LaunchShape ls;
    ls.waves = waves;
    const int need = (ngroups + waves - 1) / waves, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
template <bool NW, bool DX>
int a16_launch_bwd(hipStream_t st, const SeqArgs& a, int P) {
    const LaunchShape ls = a16_shape(a.ngroups, 4);
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(A16::NG) +
                  (size_t)ls.waves * ((DX ? 2 : 1) * 2 * 16 * kApnRow + 2 * 16 * kChunkPad + (NW ? A16::kTiles * kTileFloats : 0))) * sizeof(float);
    const size_t red = (size_t)ls.waves * (P + kLossCols + 512) * sizeof(float);
    if (NW && lds < red) lds = red;
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = apn16_bwd_kernel<NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}


// -------------------------------------------------------------------------------------------------
// Gate-parallel fused train kernel for the reference's own batch sizes (train_funcs.py:28-48; every frame gets a SIMD of its own): ONE
// sequence per single-wave workgroup; rows 0, 2 of the wave hold tile 0 of the state vector (h_I | h_A[2] | h_A[0]), rows 1, 3 tile 1 (h_Q |
// h_A[1]) — apn_unit() —, every lane carries h_I AND h_Q of its slot, so both complex rotations are local.  Only the recurrence is in the
// step loops; with lane = time step over the whole frame:
//   front     r = conj(x) / |x|, the FIR bank, the 8 rotated features and their share of W_u's pre-activation;
//   forward   v1: one rotated dot product per tile, summed across the row pair; v: one per slot over the 16 nodes; s' = sigmoid(C s) + Z v; one
//             cross-row swap hands the partner's s'; s, v, v1 and h of the frame parked in LDS;
//   head      both read-outs, loss, dL/dy;
//   backward  the two transposed rotated dot products, the step's weight gradients (W_h, the state columns of W_u) as TWO 4-block MFMAs;
//   back end  dL/dfeat -> dL/dfi, dL/dfq (lane = time step), then the sums over time on the matrix pipe: W_u's feature columns | b_u as
//             d_pre1^T [feat, 1], both FIR banks as (d_fi, d_fq)^T [I window], [Q window].
// Weight gradients only (the frozen-PA role stays on the S16 kernels).  Taken while the frame's state fits the CU's LDS.
// -------------------------------------------------------------------------------------------------
constexpr int kAgpP16 = 17, kAgpP32 = 33;
struct AgpBuf { int xw, rt, fiq, fe, au, sp, v, v1, hh, dab, dump, total; };
__host__ __device__ inline AgpBuf agp_buf(int T) {
    const int Tp = (T + 3) & ~3;
    AgpBuf b; int o = 0;
    b.xw = o; o += 2 * (Tp + 16);            // float2 [16 + Tp]: index i <-> time i - 16 (zero before the frame)
    b.rt = o; o += 2 * Tp;                   // float2 [Tp]: (rr, ri) of step t
    b.fiq = o; o += 6 * Tp;                  // [6][Tp]: fi_q, fq_q of the three filters; then dL/dfi_q, dL/dfq_q
    b.fe = o; o += 16 * Tp;                  // [16][Tp]: the 8 features, 1, 0 ... (zero columns past the frame)
    b.au = o; o += kAgpP16 * Tp;             // [Tp][17]: the features' share of W_u's pre-activation (+ b_u); overwritten by d_pre1 in the backward steps
    b.sp = o; o += kAgpP32 * Tp;             // the rotated state s of step t (tile 0 | tile 1)
    b.v = o; o += kAgpP32 * Tp;              // v of step t
    b.v1 = o; o += kAgpP16 * Tp;             // v1 of step t
    b.hh = o; o += kAgpP32 * Tp;             // h_I | h_Q after step t
    b.dab = o; o += 2 * Tp;                  // float2 [Tp]: dL/dA, dL/dBq
    b.dump = o; o += 512;
    b.total = o;
    return b;
}
__global__ __launch_bounds__(64) void apn_gp_train_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4, tile = role & 1;
    const ApnLayout L = apn_layout(a.H);
    const int H = L.H, n = L.n, T = a.T, Tp = (T + 3) & ~3, WU = 8 + n;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    const AgpBuf O = agp_buf(T);
    float* buf = smem + pad4(L.P);
    float2* xw = reinterpret_cast<float2*>(buf + O.xw);
    float2* rt = reinterpret_cast<float2*>(buf + O.rt);
    float2* dab = reinterpret_cast<float2*>(buf + O.dab);
    float *fiq = buf + O.fiq, *fe = buf + O.fe, *au = buf + O.au, *spk = buf + O.sp, *vpk = buf + O.v, *v1pk = buf + O.v1, *hh = buf + O.hh;
    float* dump = buf + O.dump;
    // rotated weights of this lane: W_u's state columns of the row's tile (node = col), W_h's row of the lane's slot, and both transposed
    float wU[16], wH[16], wUT[16], wHT[16];
    const int unit = apn_unit(H, tile, col);
    {
        const int dir = rot_dir(col);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15, um = apn_unit(H, tile, m);
            wU[k] = um >= 0 ? pl[L.o_wu + col * WU + 8 + um] : 0.0f;                  // node col  <- slot m
            wH[k] = unit >= 0 ? pl[L.o_wh + unit * kApnNode + m] : 0.0f;              // slot col  <- node m
            wUT[k] = unit >= 0 ? pl[L.o_wu + m * WU + 8 + unit] : 0.0f;               // slot col  <- d_pre1 of node m
            wHT[k] = um >= 0 ? pl[L.o_wh + um * kApnNode + col] : 0.0f;               // node col  <- d_pre2 of slot m
        }
    }
    const bool is_h = col < H, live_slot = unit >= 0;
    const float Cc = pl[L.o_c], zz = live_slot ? pl[L.o_z + unit] : 0.0f, bh = live_slot ? pl[L.o_bh + unit] : 0.0f;
    const float woi = is_h ? pl[L.o_woi + col] : 0.0f, woq = is_h ? pl[L.o_woq + col] : 0.0f;
    const float t0m = tile == 0 ? 1.0f : 0.0f;                                        // 1 on the h_I rows
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    const int dmp = (int)(dump - smem) + lane;
    // per-step stores: A rows 0, 1: s -> spk, rows 2, 3: h_I / h_Q -> hh;  B rows 0, 1: v -> vpk, row 2: v1 -> v1pk
    const int pa0 = role < 2 ? (int)(spk - smem) + tile * 16 + col : (int)(hh - smem) + tile * 16 + col, pa_step = kAgpP32;
    const int pb0 = role < 2 ? (int)(vpk - smem) + tile * 16 + col : role == 2 ? (int)(v1pk - smem) + col : dmp;
    const int pb_step = role < 2 ? kAgpP32 : role == 2 ? kAgpP16 : 0;
    const int dk0 = role == 0 ? (int)(au - smem) + col : dmp, dk_step = role == 0 ? kAgpP16 : 0;
    if (lane < 16) xw[lane] = make_float2(0.0f, 0.0f);

    f32x16 acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
    f32x4 accU = {0.f, 0.f, 0.f, 0.f}, accI = accU, accQ = accU;
    float dwi[14], dwq[14];
#pragma unroll
    for (int j = 0; j < 14; ++j) { dwi[j] = 0.0f; dwq[j] = 0.0f; }
    float dZ = 0.0f, dC = 0.0f, dbh = 0.0f, loss_acc = 0.0f;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        wave_lds_fence();
        for (int t = lane; t < Tp; t += 64) xw[16 + t] = t < T ? xg[t] : make_float2(0.0f, 0.0f);
        wave_lds_fence();
        // ---- front: r, FIR bank, rotated features, their share of W_u; lane = time step ----
        for (int t0 = 0; t0 < Tp; t0 += 64) {
            const int t = t0 + lane;
            if (t < Tp) {
                const bool live = t < T;
                const float2 x0 = live ? xw[16 + t] : make_float2(1.0f, 0.0f);
                const float mag = sqrtf(x0.x * x0.x + x0.y * x0.y), rr = x0.x / mag, ri = -x0.y / mag;
                float fi[4], fq[4];
#pragma unroll
                for (int q = 0; q < kApnF; ++q) { fi[q] = 0.0f; fq[q] = 0.0f; }
#pragma unroll
                for (int m = 0; m < kApnM; ++m) {
                    const float2 xv = xw[t + 1 + m];                             // time t - 15 + m
#pragma unroll
                    for (int q = 0; q < kApnF; ++q) {
                        const float bi = pl[L.o_bi + q * kApnM + m], bq = pl[L.o_bq + q * kApnM + m];
                        fi[q] += bi * xv.x - bq * xv.y;
                        fq[q] += bq * xv.x + bi * xv.y;
                    }
                }
                fi[3] = x0.x; fq[3] = x0.y;
                float feat[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    feat[2 * k] = rr * fi[k] - ri * fq[k];
                    feat[2 * k + 1] = ri * fi[k] + rr * fq[k];
                }
                rt[t] = make_float2(rr, ri);
#pragma unroll
                for (int q = 0; q < kApnF; ++q) { fiq[q * Tp + t] = fi[q]; fiq[(kApnF + q) * Tp + t] = fq[q]; }
#pragma unroll
                for (int k = 0; k < 8; ++k) fe[k * Tp + t] = live ? feat[k] : 0.0f;
                fe[8 * Tp + t] = live ? 1.0f : 0.0f;
#pragma unroll
                for (int k = 9; k < 16; ++k) fe[k * Tp + t] = 0.0f;
                for (int o = 0; o < kApnNode; ++o) {
                    float acc = pl[L.o_bu + o];
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc = __builtin_fmaf(pl[L.o_wu + o * WU + k], feat[k], acc);
                    au[t * kAgpP16 + o] = live ? acc : 0.0f;
                }
            }
        }
        wave_lds_fence();
        // ---- forward recurrence ----
        {
            float hI = 0.0f, hQ = 0.0f, hA = 0.0f;
            int pa = pa0, pb = pb0;
            for (int t = 0; t < T; ++t) {
                const float2 r = rt[t];
                // into the normalised frame: s_I = h_I rr - h_Q ri, s_Q = h_I ri + h_Q rr
                const float rot = tile == 0 ? __builtin_fmaf(hI, r.x, -(hQ * r.y)) : __builtin_fmaf(hI, r.y, hQ * r.x);
                const float sp = is_h ? rot : hA;
                float p1 = rotdot(0.0f, wU, sp);
                p1 += xor16(p1);
                const float v1 = tanhf_(p1 + au[t * kAgpP16 + col]);
                const float v = tanhf_(rotdot(bh, wH, v1));
                const float sn = live_slot ? sigmoidf_(Cc * sp) + zz * v : 0.0f;
                const float osn = xor16(sn);
                const float snI = tile == 0 ? sn : osn, snQ = tile == 0 ? osn : sn;
                // back: h_I = s'_I rr + s'_Q ri, h_Q = s'_Q rr - s'_I ri
                hI = __builtin_fmaf(snI, r.x, snQ * r.y);
                hQ = __builtin_fmaf(snQ, r.x, -(snI * r.y));
                hA = sn;
                smem[pa] = role < 2 ? sp : (tile == 0 ? hI : hQ);
                smem[pb] = role < 2 ? v : v1;
                pa += pa_step; pb += pb_step;
            }
        }
        wave_lds_fence();
        // ---- read-outs, loss and dL/dy of every step; lane = time step ----
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
                const float* hv = hh + t * kAgpP32;
                float y0 = 0.0f, y1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 14; ++j)
                    if (j < H) {
                        y0 = __builtin_fmaf(pl[L.o_woi + j], hv[j], y0);
                        y1 = __builtin_fmaf(pl[L.o_woq + j], hv[16 + j], y1);
                    }
                const float2 tv = tg[t];
                float dy0, dy1;
                s16_loss(lossc, (y0 - y1) - tv.x, (y1 + y0) - tv.y, dy0, dy1, loss_acc);
                const float dA = dy0 + dy1, dB = dy1 - dy0;
                dab[t] = make_float2(dA, dB);
#pragma unroll
                for (int j = 0; j < 14; ++j)
                    if (j < H) {
                        dwi[j] = __builtin_fmaf(dA, hv[j], dwi[j]);
                        dwq[j] = __builtin_fmaf(dB, hv[16 + j], dwq[j]);
                    }
            }
        }
        wave_lds_fence();
        // ---- backward recurrence ----
        {
            float dhI = 0.0f, dhQ = 0.0f, dhA = 0.0f;
            int dk = dk0 + (T - 1) * dk_step;
            for (int t = T - 1; t >= 0; --t) {
                const float2 r = rt[t], dd = dab[t];
                const float sp = spk[t * kAgpP32 + tile * 16 + col], v = vpk[t * kAgpP32 + tile * 16 + col], v1 = v1pk[t * kAgpP16 + col];
                const float gI = __builtin_fmaf(dd.x, woi, dhI), gQ = __builtin_fmaf(dd.y, woq, dhQ);
                const float drot = tile == 0 ? __builtin_fmaf(gI, r.x, -(gQ * r.y)) : __builtin_fmaf(gI, r.y, gQ * r.x);
                const float dsn = is_h ? drot : (live_slot ? dhA : 0.0f);
                const float sg = sigmoidf_(Cc * sp), dsg = sg * (1.0f - sg);
                dZ = __builtin_fmaf(dsn, v, dZ);
                dC = __builtin_fmaf(dsn * dsg, sp, dC);
                const float dpre2 = (dsn * zz) * __builtin_fmaf(-v, v, 1.0f);
                dbh += dpre2;
                float dv1 = rotdot(0.0f, wHT, dpre2);
                dv1 += xor16(dv1);
                const float dpre1 = dv1 * __builtin_fmaf(-v1, v1, 1.0f);
                const float dsp = rotdot((dsn * dsg) * Cc, wUT, dpre1);
                const float odsp = xor16(dsp);
                const float dspI = tile == 0 ? dsp : odsp, dspQ = tile == 0 ? odsp : dsp;
                dhI = __builtin_fmaf(dspI, r.x, dspQ * r.y);
                dhQ = __builtin_fmaf(dspQ, r.x, -(dspI * r.y));
                dhA = dsp;
                smem[dk] = dpre1;
                dk -= dk_step;
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(dpre2, v1, acc1, 0, 0, 0);       // blocks 0, 1: d_pre2 of tile 0 | 1  (x) v1
                acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(dpre1, sp, acc2, 0, 0, 0);       // blocks 0, 1: d_pre1 (x) s of tile 0 | 1
            }
        }
        wave_lds_fence();
        // ---- dL/dfeat -> dL/dfi, dL/dfq of the three filters; lane = time step ----
        for (int t0 = 0; t0 < Tp; t0 += 64) {
            const int t = t0 + lane;
            if (t < Tp) {
                const bool live = t < T;
                float df[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) df[k] = 0.0f;
                for (int o = 0; o < kApnNode; ++o) {
                    const float d1 = au[t * kAgpP16 + o];
#pragma unroll
                    for (int k = 0; k < 6; ++k) df[k] = __builtin_fmaf(pl[L.o_wu + o * WU + k], d1, df[k]);
                }
                const float2 r = rt[t];
#pragma unroll
                for (int q = 0; q < kApnF; ++q) {
                    const float dfi = df[2 * q] * r.x + df[2 * q + 1] * r.y, dfq = df[2 * q + 1] * r.x - df[2 * q] * r.y;
                    fiq[q * Tp + t] = live ? dfi : 0.0f; fiq[(kApnF + q) * Tp + t] = live ? dfq : 0.0f;
                }
            }
        }
        wave_lds_fence();
        // ---- the sums over time on the matrix pipe: lane (i = col, k = role) feeds A[i][k], B[k][col] of a 4-step slice ----
        {
            const int arow = (col < 6 ? col : 5) * Tp;
            const float amask = col < 6 ? 1.0f : 0.0f;
            for (int t = role; t < Tp; t += 4) {
                accU = mfma4(au[t * kAgpP16 + col], fe[col * Tp + t], accU);
                const float ad = fiq[arow + t] * amask;
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
        if (col < 8) prow[L.o_wu + i * WU + col] = accU[rr];
        else if (col == 8) prow[L.o_bu + i] = accU[rr];
        // 4-block MFMAs: register 4 blk + rr of lane l = entry (4 (l / 16) + rr, l % 16) of block blk
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int ui = apn_unit(H, blk, i), uc = apn_unit(H, blk, col);
            if (ui >= 0) prow[L.o_wh + ui * kApnNode + col] = acc1[4 * blk + rr];
            if (uc >= 0) prow[L.o_wu + i * WU + 8 + uc] = acc2[4 * blk + rr];
        }
    }
    wave_lds_fence();
    for (int idx = lane; idx < kApnF * kApnM; idx += 64) {
        const int q = idx / kApnM, m = idx % kApnM;
        prow[L.o_bi + idx] = dump[q * 16 + m] + dump[256 + (kApnF + q) * 16 + m];
        prow[L.o_bq + idx] = dump[(kApnF + q) * 16 + m] - dump[256 + q * 16 + m];
    }
    if (role < 2 && live_slot) { prow[L.o_z + unit] = dZ; prow[L.o_bh + unit] = dbh; }
    float lp = loss_acc, sc = role < 2 ? dC : 0.0f;
    for (int o = 32; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); sc += __shfl_xor(sc, o); }
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        float vi = dwi[j], vq = dwq[j];
        for (int o = 32; o > 0; o >>= 1) { vi += __shfl_xor(vi, o); vq += __shfl_xor(vq, o); }
        if (lane == 0 && j < H) { prow[L.o_woi + j] = vi; prow[L.o_woq + j] = vq; }
    }
    if (lane == 0) {
        prow[L.o_c] = sc;
        prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
    }
}

// Evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): ONE sequence per wave, the forward half of
// apn_gp_train_kernel in chunks of kAevChunk steps; no checkpoints.
constexpr int kAevChunk = 256;
constexpr int kAevFloats = 2 * (kAevChunk + 16) + 2 * kAevChunk + kAgpP16 * kAevChunk + kAgpP32 * kAevChunk + 64;
__global__ __launch_bounds__(64) void apn_gp_eval_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int Tc = kAevChunk;
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4, tile = role & 1;
    const ApnLayout L = apn_layout(a.H);
    const int H = L.H, n = L.n, T = a.T, WU = 8 + n;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* buf = smem + pad4(L.P);
    float2* xw = reinterpret_cast<float2*>(buf);                    // [16 + Tc]: index i <-> time t0 - 16 + i
    float2* rt = xw + Tc + 16;                                       // (rr, ri) of step tt
    float* au = reinterpret_cast<float*>(rt + Tc);                   // [Tc][17]
    float* hh = au + kAgpP16 * Tc;                                   // [Tc][33]: h_I | h_Q after step tt
    float* dump = hh + kAgpP32 * Tc;
    float wU[16], wH[16];
    const int unit = apn_unit(H, tile, col);
    {
        const int dir = rot_dir(col);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15, um = apn_unit(H, tile, m);
            wU[k] = um >= 0 ? pl[L.o_wu + col * WU + 8 + um] : 0.0f;
            wH[k] = unit >= 0 ? pl[L.o_wh + unit * kApnNode + m] : 0.0f;
        }
    }
    const bool is_h = col < H, live_slot = unit >= 0;
    const float Cc = pl[L.o_c], zz = live_slot ? pl[L.o_z + unit] : 0.0f, bh = live_slot ? pl[L.o_bh + unit] : 0.0f;
    const int pk0 = role >= 2 ? (int)(hh - smem) + tile * 16 + col : (int)(dump - smem) + lane, pk_step = role >= 2 ? kAgpP32 : 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float hI = 0.0f, hQ = 0.0f, hA = 0.0f;
        for (int t0 = 0; t0 < T; t0 += Tc) {
            const int len = min(Tc, T - t0);
            wave_lds_fence();
            for (int i = lane; i < len + 16; i += 64) {
                const int t = t0 - 16 + i;
                xw[i] = t >= 0 ? xg[t] : make_float2(0.0f, 0.0f);
            }
            wave_lds_fence();
            for (int tt = lane; tt < len; tt += 64) {
                const float2 x0 = xw[16 + tt];
                const float mag = sqrtf(x0.x * x0.x + x0.y * x0.y), rr = x0.x / mag, ri = -x0.y / mag;
                float fi[4], fq[4];
#pragma unroll
                for (int q = 0; q < kApnF; ++q) { fi[q] = 0.0f; fq[q] = 0.0f; }
#pragma unroll
                for (int m = 0; m < kApnM; ++m) {
                    const float2 xv = xw[tt + 1 + m];
#pragma unroll
                    for (int q = 0; q < kApnF; ++q) {
                        const float bi = pl[L.o_bi + q * kApnM + m], bq = pl[L.o_bq + q * kApnM + m];
                        fi[q] += bi * xv.x - bq * xv.y;
                        fq[q] += bq * xv.x + bi * xv.y;
                    }
                }
                fi[3] = x0.x; fq[3] = x0.y;
                float feat[8];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    feat[2 * k4] = rr * fi[k4] - ri * fq[k4];
                    feat[2 * k4 + 1] = ri * fi[k4] + rr * fq[k4];
                }
                rt[tt] = make_float2(rr, ri);
                for (int o = 0; o < kApnNode; ++o) {
                    float acc = pl[L.o_bu + o];
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) acc = __builtin_fmaf(pl[L.o_wu + o * WU + k8], feat[k8], acc);
                    au[tt * kAgpP16 + o] = acc;
                }
            }
            wave_lds_fence();
            {
                int pk = pk0;
                for (int tt = 0; tt < len; ++tt) {
                    const float2 r = rt[tt];
                    const float rot = tile == 0 ? __builtin_fmaf(hI, r.x, -(hQ * r.y)) : __builtin_fmaf(hI, r.y, hQ * r.x);
                    const float sp = is_h ? rot : hA;
                    float p1 = rotdot(0.0f, wU, sp);
                    p1 += xor16(p1);
                    const float v1 = tanhf_(p1 + au[tt * kAgpP16 + col]);
                    const float v = tanhf_(rotdot(bh, wH, v1));
                    const float sn = live_slot ? sigmoidf_(Cc * sp) + zz * v : 0.0f;
                    const float osn = xor16(sn);
                    const float snI = tile == 0 ? sn : osn, snQ = tile == 0 ? osn : sn;
                    hI = __builtin_fmaf(snI, r.x, snQ * r.y);
                    hQ = __builtin_fmaf(snQ, r.x, -(snI * r.y));
                    hA = sn;
                    smem[pk] = tile == 0 ? hI : hQ;
                    pk += pk_step;
                }
            }
            wave_lds_fence();
            for (int tt = lane; tt < len; tt += 64) {
                const float* hv = hh + tt * kAgpP32;
                float y0 = 0.0f, y1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 14; ++j)
                    if (j < H) {
                        y0 = __builtin_fmaf(pl[L.o_woi + j], hv[j], y0);
                        y1 = __builtin_fmaf(pl[L.o_woq + j], hv[16 + j], y1);
                    }
                yg[t0 + tt] = make_float2(y0 - y1, y1 + y0);
            }
        }
    }
}

static size_t apn_gp_lds_bytes(int P, int T) { return ((size_t)pad4(P) + agp_buf(T).total) * sizeof(float); }
static int apn_gp_blocks_per_cu(int P, int T) {
    const size_t lds = apn_gp_lds_bytes(P, T);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}

}  // namespace

bool apnrru_ok(const odpd_model_t* m) { return m->hidden >= 1 && m->hidden <= 14; }
int64_t apnrru_param_count(const odpd_model_t* m) {
    return apnrru_ok(m) ? (int64_t)apn_layout(m->hidden).P : (int64_t)ODPD_EUNSUPPORTED;
}
int apnrru_rows(const odpd_model_t* m, int B) {
    if (!apnrru_ok(m)) return ODPD_EUNSUPPORTED;
    return a16_shape((B + 15) / 16, 4).grid;
}
int64_t apnrru_ckpt_floats(const odpd_model_t* m, int B, int T) {
    if (!apnrru_ok(m)) return ODPD_EUNSUPPORTED;
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * 2 * 256;
}
// the gate-parallel fused train kernel: one sequence per single-wave workgroup, the frame's state in LDS
bool apnrru_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (!apnrru_ok(m) || T < kApnM - 1) return false;
    const int per_cu = apn_gp_blocks_per_cu(apn_layout(m->hidden).P, T);
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && per_cu > 0;
    // up to five rounds of workgroups (measured: profiles/r03/gp_train_bench_f4.txt): the alternative is the forward / loss / backward chain of the S16 kernels
    return (long)B <= 5L * device_cus() * per_cu;
}
int apnrru_gp_rows(const odpd_model_t* m, int B, int T) {
    const long cap = (long)device_cus() * apn_gp_blocks_per_cu(apn_layout(m->hidden).P, T);
    return B < cap ? B : (int)cap;
}
int apnrru_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const size_t lds = apn_gp_lds_bytes(apn_layout(m->hidden).P, a.T);
    if (int e = allow_big_lds(apn_gp_train_kernel, lds)) return e;
    hipLaunchKernelGGL(apn_gp_train_kernel, dim3(apnrru_gp_rows(m, a.B, a.T)), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
// mode 1 forward, 2 backward
int apnrru_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    if (!apnrru_ok(m)) return ODPD_EUNSUPPORTED;
    if (a0.T < kApnM - 1) return ODPD_EINVAL;       // the reference cuts its 15-sample zero pad from the frame itself (apnrru.py:68-69)
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = apn_layout(m->hidden).P;
    if (mode == 1 && !a.ckpt && a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0) {
        // sequences that each get a SIMD of their own (inference: no checkpoints)
        const size_t lds = ((size_t)pad4(P) + kAevFloats) * sizeof(float);
        if (int e = allow_big_lds(apn_gp_eval_kernel, lds)) return e;
        hipLaunchKernelGGL(apn_gp_eval_kernel, dim3(a.B), dim3(64), lds, st, a);
        return (int)hipGetLastError();
    }
    if (mode == 1) {
        const LaunchShape ls = a16_shape(a.ngroups, a.ngroups <= 4 * device_cus() ? 4 : 8);
        const size_t lds = ((size_t)pad4(P) + s16_tab_floats(A16::NG) + (size_t)ls.waves * (2 * 16 * kApnRow + 2 * 16 * kChunkPad)) * sizeof(float);
        auto k = apn16_fwd_kernel;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    }
    if (!a.ckpt && a.nck > 1) return ODPD_EINVAL;
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (!nw && !dx) return ODPD_EINVAL;
    if (nw && dx) return a16_launch_bwd<true, true>(st, a, P);
    if (nw) return a16_launch_bwd<true, false>(st, a, P);
    return a16_launch_bwd<false, true>(st, a, P);
}

}  // namespace odpd
