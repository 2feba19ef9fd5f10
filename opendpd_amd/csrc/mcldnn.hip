// mcldnn.hip — MCLDNN (backbones/mcldnn.py:9-134): per step a 5 x 5 patch [(I, Q, a, a^2, a^3) x the last five samples, the window wrapping
// around the frame] -> Conv2d(1 -> C, 3x3) branch + grouped Conv1d(5 -> 5C, k3) branch (re-read as (C, 5, 5)) -> merged by Conv2d(10 -> 1, 3x3)
// -> nn.LSTM(5C -> 8) -> Linear(8 -> 16) -> Linear(16 -> 2).
//
// There is no activation between the three convolutions, nor between them and the LSTM's input projection, nor between the two linear
// layers: the whole front end is ONE affine map of the 25 patch values,  gates_in = A vec(P) + b  with  A = W_ih M (32 x 25),
// M = d z / d P (5C x 25) the composition of the convolutions (zero padding and the re-read included),  and  y = W_fc h + b_fc  with
// W_fc = W_2 W_1 (2 x 8).  Every workgroup composes A, b, W_fc from the parameters in its prologue (~100 k multiply-adds), the per-sample
// work then does not depend on C at all:
//   S16 mapping, a wave = 16 sequences; the LSTM(8) is pair-packed on the lanes: quads q = 0, 2 hold the gates (i, g) of units 0..3 /
//   4..7, quads q = 1, 3 hold (f, o), the cell and the hidden state of the same units — two 16-row gate tiles instead of four, one
//   ds_swizzle exchange of i g per step;  gates_in as 14 exact-fp32 MFMAs on patch values read straight from a feature table staged in LDS
//   (5 floats per sample, 4-sample circular halo), the recurrent part 8.
// Backward: BPTT from (h, c) checkpoints every kCkptStride steps; dA (32 x 25) as MFMA outer products with the LDS feature table, dL/dx
// through A^T into an LDS feature-gradient table (carried across chunks, the circular wrap added to the frame's last four samples at the
// end), then through the features.  The workgroup DEcomposes its dA, db, dW_fc into the gradients of the three convolutions, W_ih, the
// biases and the two linear layers (the chain rule through A = W_ih M(w_1, w_1d, w_2)) before it writes its row of partials, so the rows
// are ordinary parameter-gradient rows.  Frames shorter than 4 samples are refused like the reference's framing (mcldnn.py:115-118).
#include "odpd_s16.h"

namespace odpd {
namespace {

constexpr int kMclH = 8, kMclG = 32, kMclP = 25, kMclPP = 26;      // PP: the 25 patch columns + the constant column
constexpr int kMclHalo = 4, kMclTimes = kMclHalo + kChunk;
constexpr int kMclRowF = 5 * kMclTimes + 1;                          // floats per sequence row of a feature table: index 5 i + f <-> time t0 - 4 + i
constexpr int kMclMaxC = 16;
constexpr int kMclRaw = kMclG * kMclP + kMclG + kMclG * kMclH + 2 * kMclH + 2;      // dA, db, dW_hh, dW_fc, db_fc of one wave
struct MclLayout { int C, o_w1, o_b1, o_w1d, o_b1d, o_w2, o_b2, o_wih, o_whh, o_bih, o_bhh, o_wf1, o_bf1, o_wf2, o_bf2, P; };
__host__ __device__ inline MclLayout mcl_layout(int C) {
    MclLayout L; L.C = C; int o = 0;
    L.o_w1 = o; o += 9 * C; L.o_b1 = o; o += C;
    L.o_w1d = o; o += 15 * C; L.o_b1d = o; o += 5 * C;
    L.o_w2 = o; o += 90; L.o_b2 = o; o += 1;
    L.o_wih = o; o += kMclG * 5 * C; L.o_whh = o; o += kMclG * kMclH; L.o_bih = o; o += kMclG; L.o_bhh = o; o += kMclG;
    L.o_wf1 = o; o += 16 * kMclH; L.o_bf1 = o; o += 16; L.o_wf2 = o; o += 32; L.o_bf2 = o; o += 2;
    L.P = o;
    return L;
}
// composed operands in LDS: M (5C x 26: 25 patch columns p = 5 mm + f, then the constant z0), A (32 x 26: ... then the bias), W_fc (2 x 8), b_fc (2)
struct MclComp { float *M, *A, *wfc, *bfc; };
__host__ __device__ inline int mcl_comp_floats(int C) { return 5 * C * kMclPP + kMclG * kMclPP + 2 * kMclH + 2; }
__device__ __forceinline__ MclComp mcl_comp(float* base, int C) {
    MclComp c; c.M = base; c.A = base + 5 * C * kMclPP; c.wfc = c.A + kMclG * kMclPP; c.bfc = c.wfc + 2 * kMclH;
    return c;
}
// d z[c][m] / d P[f][mm] and the constant part (column 25): conv2d_2 over (conv2d_1 rows | the re-read grouped conv1d)
__device__ __forceinline__ float mcl_m_entry(const float* pl, const MclLayout& L, int zi, int pp) {
    const int C = L.C, c = zi / 5, m = zi % 5, f = pp % 5, mm = pp / 5;
    const bool cst = pp == kMclP;
    float acc = cst ? pl[L.o_b2] : 0.0f;
    for (int dc = 0; dc < 3; ++dc)
        for (int dm = 0; dm < 3; ++dm) {
            const int cc = c + dc - 1, m2 = m + dm - 1;
            if (cc < 0 || cc >= C || m2 < 0 || m2 >= 5) continue;
            const int dmm = mm - m2 + 1;
            for (int ch = 0; ch < 5; ++ch) {
                const float w2 = pl[L.o_w2 + ch * 9 + dc * 3 + dm];
                const int df = f - ch + 1;
                if (cst) acc += w2 * pl[L.o_b1 + cc];
                else if (df >= 0 && df < 3 && dmm >= 0 && dmm < 3) acc += w2 * pl[L.o_w1 + cc * 9 + df * 3 + dmm];
            }
            for (int fp = 0; fp < 5; ++fp) {
                const float w2 = pl[L.o_w2 + (5 + fp) * 9 + dc * 3 + dm];
                const int oc = 5 * cc + fp;
                if (cst) acc += w2 * pl[L.o_b1d + oc];
                else if (oc / C == f && dmm >= 0 && dmm < 3) acc += w2 * pl[L.o_w1d + oc * 3 + dmm];
            }
        }
    return acc;
}
__device__ __forceinline__ void mcl_compose(const MclComp& K, const float* pl, const MclLayout& L) {
    const int Z = 5 * L.C;
    for (int i = threadIdx.x; i < Z * kMclPP; i += blockDim.x) K.M[i] = mcl_m_entry(pl, L, i / kMclPP, i % kMclPP);
    __syncthreads();
    for (int i = threadIdx.x; i < kMclG * kMclPP; i += blockDim.x) {
        const int g = i / kMclPP, pp = i % kMclPP;
        float acc = pp == kMclP ? pl[L.o_bih + g] + pl[L.o_bhh + g] : 0.0f;
        for (int zi = 0; zi < Z; ++zi) acc = __builtin_fmaf(pl[L.o_wih + g * Z + zi], K.M[zi * kMclPP + pp], acc);
        K.A[i] = acc;
    }
    for (int i = threadIdx.x; i < 2 * kMclH + 2; i += blockDim.x) {
        const int o = i < 2 * kMclH ? i / kMclH : i - 2 * kMclH, j = i % kMclH;
        float acc = i < 2 * kMclH ? 0.0f : pl[L.o_bf2 + o];
        for (int k = 0; k < 16; ++k) acc = __builtin_fmaf(pl[L.o_wf2 + o * 16 + k], i < 2 * kMclH ? pl[L.o_wf1 + k * kMclH + j] : pl[L.o_bf1 + k], acc);
        (i < 2 * kMclH ? K.wfc : K.bfc)[i < 2 * kMclH ? i : o] = acc;
    }
    __syncthreads();
}

// gate tiles: tile 0 rows (q even: i, q odd: f), tile 1 rows (q even: g, q odd: o), unit 4 (q >> 1) + e;  reference row = gate * 8 + unit
__host__ __device__ inline int mcl_gate_row(int tile, int m) {
    const int q = m >> 2, e = m & 3;
    return (2 * tile + (q & 1)) * kMclH + 4 * (q >> 1) + e;
}
// sigmoid rows carry -log2(e); the g rows (tanh) stay as they are
__host__ __device__ inline float mcl_row_scale(int tile, int m) { return (tile == 1 && ((m >> 2) & 1) == 0) ? 1.0f : kNegLog2e; }
// the unit a K-lane (q, e) carries in the state vectors (h, c live on the odd quads)
__host__ __device__ inline int mcl_k_unit(int q, int e) { return (q & 1) ? 4 * (q >> 1) + e : -1; }

struct M16 {
    static constexpr int AT = 0;           // 2 tile + g2 : A[row(tile, m)][p = 4 (4 g2 + e) + q]      (pre-scaled)
    static constexpr int WHH = AT + 4;     // tile        : W_hh[row(tile, m)][unit(q, e)]              (pre-scaled)
    static constexpr int WHHT = WHH + 2;   // kt          : W_hh[row(kt, 4q+e)][unit(m)]
    static constexpr int ATT = WHHT + 2;   // 2 ot + kt   : A[row(kt, 4q+e)][p = 16 ot + m]
    static constexpr int BIAS = ATT + 4;   // tile        : b[row(tile, 4q+e)]                          (pre-scaled)
    static constexpr int WFC = BIAS + 2;   // o           : W_fc[o][unit(q, e)]
    static constexpr int NG = WFC + 2;
    static constexpr int kTiles = 3;       // D0 D1 | h_prev
};
__device__ __forceinline__ float4 m16_entry(const MclComp& K, const float* pl, const MclLayout& L, int grp, int m, int q) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float s = 0.0f;
        if (grp < M16::WHH) {
            const int tile = (grp - M16::AT) >> 1, p = 4 * (4 * ((grp - M16::AT) & 1) + e) + q;
            s = p < kMclP ? K.A[mcl_gate_row(tile, m) * kMclPP + p] * mcl_row_scale(tile, m) : 0.0f;
        } else if (grp < M16::WHHT) {
            const int tile = grp - M16::WHH, u = mcl_k_unit(q, e);
            s = u >= 0 ? pl[L.o_whh + mcl_gate_row(tile, m) * kMclH + u] * mcl_row_scale(tile, m) : 0.0f;
        } else if (grp < M16::ATT) {
            const int u = mcl_k_unit(m >> 2, m & 3);
            s = u >= 0 ? pl[L.o_whh + mcl_gate_row(grp - M16::WHHT, 4 * q + e) * kMclH + u] : 0.0f;
        } else if (grp < M16::BIAS) {
            const int ot = (grp - M16::ATT) >> 1, kt = (grp - M16::ATT) & 1, p = 16 * ot + m;
            s = p < kMclP ? K.A[mcl_gate_row(kt, 4 * q + e) * kMclPP + p] : 0.0f;
        } else if (grp < M16::WFC) {
            const int tile = grp - M16::BIAS;
            s = K.A[mcl_gate_row(tile, 4 * q + e) * kMclPP + kMclP] * mcl_row_scale(tile, 4 * q + e);
        } else {
            const int u = mcl_k_unit(q, e);
            s = u >= 0 ? K.wfc[(grp - M16::WFC) * kMclH + u] : 0.0f;
        }
        v[e] = s;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void m16_build_table(float* tab, const MclComp& K, const float* pl, const MclLayout& L, int lane, int wave, int nwb) {
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int grp = wave; grp < M16::NG; grp += nwb) t4[grp * 64 + lane] = m16_entry(K, pl, L, grp, lane & 15, lane >> 4);
    __syncthreads();
}
__device__ __forceinline__ float m16_uni(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ f32x4 m16_mv(TabPtr tl, int grp, const f32x4& v, f32x4 acc) {
    f32x4 a1[1] = {acc};
    const f32x4 v1[1] = {v};
    s16n_matvec<1>(tl, grp, v1, a1);
    return a1[0];
}
__device__ __forceinline__ f32x4 m16_swap(const f32x4& v) { return f32x4{swap16(v[0]), swap16(v[1]), swap16(v[2]), swap16(v[3])}; }

// feature table of a chunk: 16 sequences x 36 samples (times t0 - 4 .. t0 + 31, the frame wrapping around: mcldnn.py:115-116) x (I, Q, a, a^2, a^3)
__device__ __forceinline__ void m16_stage_feat(float* fl, const float* g, int b0, int B, int T, int t0, int len, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
#pragma unroll
    for (int j = 0; j < 16 * kMclTimes / 64; ++j) {
        const int e = lane + 64 * j, m = e / kMclTimes, i = e % kMclTimes, t = t0 - kMclHalo + i;
        float2 xv = make_float2(0.5f, 0.25f);                           // idle sequence slots / steps past the frame: a finite signal
        if (b0 + m < B && t < t0 + len) xv = g2[(size_t)(b0 + m) * T + (t < 0 ? t + T : t)];
        const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), a = __builtin_amdgcn_sqrtf(a2);
        float* d = fl + m * kMclRowF + 5 * i;
        d[0] = xv.x; d[1] = xv.y; d[2] = a; d[3] = a2; d[4] = a2 * a;
    }
}
// the seven patch K-chunks of this lane at local step tt: p = 4 cc + q  (table index 5 tt + p; the chunk beyond the patch multiplies zeros)
__device__ __forceinline__ void m16_patch(const float* flrow, int tt, int q, float (&pv)[7]) {
    const float* s = flrow + 5 * tt + q;
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) pv[cc] = s[4 * cc];
    pv[6] = q == 0 ? s[24] : 0.0f;
}
struct M16Cell { f32x4 s0, s1; };      // (i | f), (g | o)
__device__ __forceinline__ M16Cell m16_gates(TabPtr tl, const float (&pv)[7], const f32x4& h, bool odd) {
    f32x4 p0 = as_f32x4(tab_ld(tl, (M16::BIAS + 0) * 64)), p1 = as_f32x4(tab_ld(tl, (M16::BIAS + 1) * 64));
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
        const f32x4 w0 = as_f32x4(tab_ld(tl, (M16::AT + g2) * 64)), w1 = as_f32x4(tab_ld(tl, (M16::AT + 2 + g2) * 64));
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g2 + e < 7) { p0 = mfma4(w0[e], pv[4 * g2 + e], p0); p1 = mfma4(w1[e], pv[4 * g2 + e], p1); }
    }
    p0 = m16_mv(tl, M16::WHH + 0, h, p0);
    p1 = m16_mv(tl, M16::WHH + 1, h, p1);
    M16Cell c;
    c.s0 = sigmoid4_prescaled(p0);
    const f32x4 sg = sigmoid4_prescaled(p1), th = tanh4_precise(p1);
    ODPD_EACH4 c.s1[i] = odd ? sg[i] : th[i];
    return c;
}
// c' = f c + i g, h' = o tanh(c') on the odd quads (the even ones hold 0)
__device__ __forceinline__ void m16_update(const M16Cell& g, bool odd, f32x4& c, f32x4& h) {
    const f32x4 ig = m16_swap(mul4(g.s0, g.s1));
    f32x4 cn;
    ODPD_EACH4 cn[i] = odd ? __builtin_fmaf(g.s0[i], c[i], ig[i]) : 0.0f;
    const f32x4 tc = tanh4_precise(cn);
    ODPD_EACH4 h[i] = odd ? g.s1[i] * tc[i] : 0.0f;
    c = cn;
}

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void mcl16_fwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride, kWave = 16 * kMclRowF + 2 * 16 * kChunkPad;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const bool odd = q & 1;
    const MclLayout L = mcl_layout(a.H);
    float* tab = smem;
    const MclComp K = mcl_comp(tab + s16_tab_floats(M16::NG), L.C);
    float* ubase = K.bfc + 2;                      // the parameters while composing, the per-wave regions afterwards
    ubase += (4 - ((ubase - smem) & 3)) & 3;
    stage_params(ubase, a.params, L.P);
    mcl_compose(K, ubase, L);
    m16_build_table(tab, K, ubase, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const float bf0 = m16_uni(K.bfc[0]), bf1 = m16_uni(K.bfc[1]);
    float* fl = ubase + (size_t)wave * kWave;
    float2* ys = reinterpret_cast<float2*>(fl + 16 * kMclRowF);
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * 2 * 64 + lane : nullptr;   // [ckpt][h | c][lane]
        f32x4 h = {0.f, 0.f, 0.f, 0.f}, c = h;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            m16_stage_feat(fl, a.x, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const TabPtr tk = opaque(tl);
                float pv[7];
                m16_patch(fl + n * kMclRowF, tt, q, pv);
                const M16Cell g = m16_gates(tk, pv, h, odd);
                m16_update(g, odd, c, h);
                const f32x4 w0 = as_f32x4(tab_ld(tk, (M16::WFC + 0) * 64)), w1 = as_f32x4(tab_ld(tk, (M16::WFC + 1) * 64));
                float s0 = 0.0f, s1 = 0.0f;
                ODPD_EACH4 { s0 = __builtin_fmaf(w0[i], h[i], s0); s1 = __builtin_fmaf(w1[i], h[i], s1); }
                const float y0 = quad_sum(s0) + bf0, y1 = quad_sum(s1) + bf1;
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (ck != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    ck[((size_t)(t1 / S) * 2) * 64] = make_float4(h[0], h[1], h[2], h[3]);
                    ck[((size_t)(t1 / S) * 2 + 1) * 64] = make_float4(c[0], c[1], c[2], c[3]);
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
struct M16Grad {
    f32x4 da[2][2], dwhh[2], db[2];        // dA (gate tile x patch-column tile), dW_hh (gate tile), db (gate tile)
    f32x4 dwfc[2];
    float dbfc[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        da[0][0] = da[0][1] = da[1][0] = da[1][1] = dwhh[0] = dwhh[1] = db[0] = db[1] = dwfc[0] = dwfc[1] = z4;
        dbfc[0] = dbfc[1] = 0.f;
    }
};

template <bool NW, bool DX, bool FULL>
__device__ __forceinline__ void m16_bwd_block(TabPtr tl0, bool odd, M16Grad& G, const float* fl, const float2* dys, float* dfl, float* tiles,
                                              int n, int q, int tloc, int nstep, f32x4 h, f32x4 c, f32x4& dh, f32x4& dc) {
    constexpr int S = kCkptStride;
    f32x4 s0_s[S], s1_s[S], cp_s[S], cn_s[S], hp_s[S];
    const float* flrow = fl + n * kMclRowF;
#pragma unroll
    for (int si = 0; si < S; ++si) {
        if (FULL || si < nstep) {
            float pv[7];
            m16_patch(flrow, tloc + si, q, pv);
            const M16Cell g = m16_gates(opaque(tl0), pv, h, odd);
            s0_s[si] = g.s0; s1_s[si] = g.s1; cp_s[si] = c; hp_s[si] = h;
            m16_update(g, odd, c, h);
            cn_s[si] = c;
        }
    }
    auto tile = [tiles](int qty) { return tiles + qty * kTileFloats; };     // 0 D0 1 D1 | 2 h_prev
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            const TabPtr tl = opaque(tl0);
            const int tt = tloc + si;
            const float2 dyv = dys[n * kChunkPad + tt];
            const f32x4 s0 = s0_s[si], s1 = s1_s[si], cp = cp_s[si], tc = tanh4_precise(cn_s[si]);
            const f32x4 w0 = as_f32x4(tab_ld(tl, (M16::WFC + 0) * 64)), w1 = as_f32x4(tab_ld(tl, (M16::WFC + 1) * 64));
            if constexpr (NW) {
                ODPD_EACH4 {
                    const float hn = s1[i] * tc[i];                        // h' (odd quads; the W_fc columns of the even ones are dropped)
                    G.dwfc[0][i] = __builtin_fmaf(dyv.x, hn, G.dwfc[0][i]);
                    G.dwfc[1][i] = __builtin_fmaf(dyv.y, hn, G.dwfc[1][i]);
                }
                G.dbfc[0] += q == 0 ? dyv.x : 0.0f; G.dbfc[1] += q == 0 ? dyv.y : 0.0f;
            }
            // odd quads: through h' = o tanh(c'), c' = f c + i g;  d(i g) goes over to the even quads
            f32x4 dct, dfo0, dfo1;
            ODPD_EACH4 {
                const float dht = __builtin_fmaf(dyv.x, w0[i], __builtin_fmaf(dyv.y, w1[i], dh[i]));
                dct[i] = odd ? __builtin_fmaf(dht * s1[i], __builtin_fmaf(-tc[i], tc[i], 1.0f), dc[i]) : 0.0f;
                dfo0[i] = dct[i] * cp[i] * s0[i] * (1.0f - s0[i]);                 // f
                dfo1[i] = dht * tc[i] * s1[i] * (1.0f - s1[i]);                    // o
                dc[i] = dct[i] * s0[i];
            }
            const f32x4 dig = m16_swap(dct);
            f32x4 D0, D1;
            ODPD_EACH4 {
                D0[i] = odd ? dfo0[i] : dig[i] * s1[i] * s0[i] * (1.0f - s0[i]);                       // i
                D1[i] = odd ? dfo1[i] : dig[i] * s0[i] * __builtin_fmaf(-s1[i], s1[i], 1.0f);          // g
            }
            if constexpr (NW) { G.db[0] = add4(G.db[0], D0); G.db[1] = add4(G.db[1], D1); }
            dh = m16_mv(tl, M16::WHHT + 0, D0, z4);
            dh = m16_mv(tl, M16::WHHT + 1, D1, dh);
            if constexpr (DX) {
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    f32x4 dp = m16_mv(tl, M16::ATT + 2 * ot + 0, D0, z4);
                    dp = m16_mv(tl, M16::ATT + 2 * ot + 1, D1, dp);
                    float* d = dfl + n * kMclRowF + 5 * tt + 16 * ot + 4 * q;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ot == 0 || 16 + 4 * q + e < kMclP) d[e] += dp[e];       // (ot 1: q 0, 1 whole, q 2 its first element)
                }
            }
            if constexpr (NW) {
                wave_lds_fence();
                tile_put(tile(0), n, q, D0); tile_put(tile(1), n, q, D1); tile_put(tile(2), n, q, hp_s[si]);
                wave_lds_fence();
                float dT[2][4], sT[4];
                tile_get(tile(0), n, q, dT[0]); tile_get(tile(1), n, q, dT[1]); tile_get(tile(2), n, q, sT);
                // dA: this lane is patch column n of the first column tile and 16 + n (< 25) of the second, for the sequences 4q + c
                const bool col1 = 16 + n < kMclP;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float* xr = fl + (4 * q + c4) * kMclRowF + 5 * tt + n;
                    const float x0 = xr[0], x1 = col1 ? xr[16] : 0.0f;
                    G.da[0][0] = mfma4(dT[0][c4], x0, G.da[0][0]); G.da[0][1] = mfma4(dT[0][c4], x1, G.da[0][1]);
                    G.da[1][0] = mfma4(dT[1][c4], x0, G.da[1][0]); G.da[1][1] = mfma4(dT[1][c4], x1, G.da[1][1]);
                    G.dwhh[0] = mfma4(dT[0][c4], sT[c4], G.dwhh[0]); G.dwhh[1] = mfma4(dT[1][c4], sT[c4], G.dwhh[1]);
                }
            }
        }
    }
}

// the wave's accumulators as dA (32 x 25), db (32), dW_hh (32 x 8), dW_fc (2 x 8), db_fc (2) in reference row order
__device__ __forceinline__ void m16_write_raw(float* raw, M16Grad& G, int lane, int n, int q) {
    float* rA = raw; float* rb = rA + kMclG * kMclP; float* rhh = rb + kMclG; float* rfc = rhh + kMclG * kMclH; float* rbf = rfc + 2 * kMclH;
    const int un = mcl_k_unit(n >> 2, n & 3), uq = mcl_k_unit(q, 0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int td = 0; td < 2; ++td) {
            const int g = mcl_gate_row(td, 4 * q + rr);
            rA[g * kMclP + n] = G.da[td][0][rr];
            if (16 + n < kMclP) rA[g * kMclP + 16 + n] = G.da[td][1][rr];
            if (un >= 0) rhh[g * kMclH + un] = G.dwhh[td][rr];
            const float b = row_sum16(G.db[td][rr]);
            if (n == 0) rb[g] = b;
        }
        const float f0 = row_sum16(G.dwfc[0][rr]), f1 = row_sum16(G.dwfc[1][rr]);
        if (n == 0 && uq >= 0) { rfc[uq + rr] = f0; rfc[kMclH + uq + rr] = f1; }
    }
    const float d0 = row_sum16(G.dbfc[0]), d1 = row_sum16(G.dbfc[1]);      // accumulated on the q == 0 lanes only
    if (lane == 0) { rbf[0] = d0; rbf[1] = d1; }
}
// chain rule through A = W_ih M(w_1, w_1d, w_2), b = W_ih z0 + b_ih + b_hh, W_fc = W_2 W_1, b_fc = W_2 b_1 + b_2: one entry of the row of partials.
// raw: the workgroup's dA / db / dW_hh / dW_fc / db_fc; Gz (5C x 26) = W_ih^T [dA | db] = the sums of dz (x) P and of dz over the samples
__device__ __forceinline__ float m16_param_grad(int i, const float* raw, const float* Gz, const MclComp& K, const float* pl, const MclLayout& L) {
    const int C = L.C, Z = 5 * C;
    const float* rA = raw; const float* rb = rA + kMclG * kMclP; const float* rhh = rb + kMclG; const float* rfc = rhh + kMclG * kMclH; const float* rbf = rfc + 2 * kMclH;
    auto gz = [&](int c, int m, int f, int mm) -> float {       // Gz[(c, m)][(f, mm)], 0 outside the patch / the map
        return (c >= 0 && c < C && m >= 0 && m < 5 && f >= 0 && f < 5 && mm >= 0 && mm < 5) ? Gz[(c * 5 + m) * kMclPP + 5 * mm + f] : 0.0f;
    };
    auto g0 = [&](int c, int m) -> float { return (c >= 0 && c < C && m >= 0 && m < 5) ? Gz[(c * 5 + m) * kMclPP + kMclP] : 0.0f; };
    float acc = 0.0f;
    if (i < L.o_b1) {                               // conv2d_1 weight [c2][df][dmm]
        const int c2 = i / 9, df = (i % 9) / 3, dmm = i % 3;
        for (int ch = 0; ch < 5; ++ch)
            for (int m2 = 0; m2 < 5; ++m2)
                for (int dc = 0; dc < 3; ++dc)
                    for (int dm = 0; dm < 3; ++dm)
                        acc = __builtin_fmaf(pl[L.o_w2 + ch * 9 + dc * 3 + dm], gz(c2 - dc + 1, m2 - dm + 1, ch + df - 1, m2 + dmm - 1), acc);
    } else if (i < L.o_w1d) {                       // conv2d_1 bias
        const int c2 = i - L.o_b1;
        for (int ch = 0; ch < 5; ++ch)
            for (int m2 = 0; m2 < 5; ++m2)
                for (int dc = 0; dc < 3; ++dc)
                    for (int dm = 0; dm < 3; ++dm) acc = __builtin_fmaf(pl[L.o_w2 + ch * 9 + dc * 3 + dm], g0(c2 - dc + 1, m2 - dm + 1), acc);
    } else if (i < L.o_b1d) {                       // conv1d weight [oc][dmm]: lands at (channel 5 + oc % 5, row oc / 5), reads feature oc / C
        const int oc = (i - L.o_w1d) / 3, dmm = (i - L.o_w1d) % 3;
        for (int m2 = 0; m2 < 5; ++m2)
            for (int dc = 0; dc < 3; ++dc)
                for (int dm = 0; dm < 3; ++dm)
                    acc = __builtin_fmaf(pl[L.o_w2 + (5 + oc % 5) * 9 + dc * 3 + dm], gz(oc / 5 - dc + 1, m2 - dm + 1, oc / C, m2 + dmm - 1), acc);
    } else if (i < L.o_w2) {                        // conv1d bias
        const int oc = i - L.o_b1d;
        for (int m2 = 0; m2 < 5; ++m2)
            for (int dc = 0; dc < 3; ++dc)
                for (int dm = 0; dm < 3; ++dm) acc = __builtin_fmaf(pl[L.o_w2 + (5 + oc % 5) * 9 + dc * 3 + dm], g0(oc / 5 - dc + 1, m2 - dm + 1), acc);
    } else if (i < L.o_b2) {                        // conv2d_2 weight [ch][dc][dm]: sum over z of dz (x) its input map
        const int ch = (i - L.o_w2) / 9, dc = ((i - L.o_w2) % 9) / 3, dm = (i - L.o_w2) % 3;
        for (int c = 0; c < C; ++c)
            for (int m = 0; m < 5; ++m) {
                const int cc = c + dc - 1, m2 = m + dm - 1;
                if (cc < 0 || cc >= C || m2 < 0 || m2 >= 5) continue;
                if (ch < 5) {
                    acc = __builtin_fmaf(g0(c, m), pl[L.o_b1 + cc], acc);
                    for (int df = 0; df < 3; ++df)
                        for (int dmm = 0; dmm < 3; ++dmm) acc = __builtin_fmaf(gz(c, m, ch + df - 1, m2 + dmm - 1), pl[L.o_w1 + cc * 9 + df * 3 + dmm], acc);
                } else {
                    const int oc = 5 * cc + ch - 5;
                    acc = __builtin_fmaf(g0(c, m), pl[L.o_b1d + oc], acc);
                    for (int dmm = 0; dmm < 3; ++dmm) acc = __builtin_fmaf(gz(c, m, oc / C, m2 + dmm - 1), pl[L.o_w1d + oc * 3 + dmm], acc);
                }
            }
    } else if (i < L.o_wih) {                       // conv2d_2 bias
        for (int zi = 0; zi < Z; ++zi) acc += Gz[zi * kMclPP + kMclP];
    } else if (i < L.o_whh) {                       // W_ih [g][zi] = dA M^T + db z0
        const int g = (i - L.o_wih) / Z, zi = (i - L.o_wih) % Z;
        for (int p = 0; p < kMclP; ++p) acc = __builtin_fmaf(rA[g * kMclP + p], K.M[zi * kMclPP + p], acc);
        acc = __builtin_fmaf(rb[g], K.M[zi * kMclPP + kMclP], acc);
    } else if (i < L.o_bih) acc = rhh[i - L.o_whh];
    else if (i < L.o_bhh) acc = rb[i - L.o_bih];
    else if (i < L.o_wf1) acc = rb[i - L.o_bhh];
    else if (i < L.o_bf1) {                         // fc_out [k][j] = W_2^T dW_fc
        const int k = (i - L.o_wf1) / kMclH, j = (i - L.o_wf1) % kMclH;
        acc = pl[L.o_wf2 + k] * rfc[j] + pl[L.o_wf2 + 16 + k] * rfc[kMclH + j];
    } else if (i < L.o_wf2) {                       // fc_out bias = W_2^T db_fc
        const int k = i - L.o_bf1;
        acc = pl[L.o_wf2 + k] * rbf[0] + pl[L.o_wf2 + 16 + k] * rbf[1];
    } else if (i < L.o_bf2) {                       // fc_out_2 [o][k] = dW_fc W_1^T + db_fc b_1
        const int o = (i - L.o_wf2) / 16, k = (i - L.o_wf2) % 16;
        acc = rbf[o] * pl[L.o_bf1 + k];
        for (int j = 0; j < kMclH; ++j) acc = __builtin_fmaf(rfc[o * kMclH + j], pl[L.o_wf1 + k * kMclH + j], acc);
    } else acc = rbf[i - L.o_bf2];
    return acc;
}

template <bool NW, bool DX>
__global__ __launch_bounds__(256, 1) void mcl16_bwd_kernel(SeqArgs a) {
    constexpr int S = kCkptStride;
    constexpr int kWave = (DX ? 2 : 1) * 16 * kMclRowF + 2 * 16 * kChunkPad + (NW ? M16::kTiles * kTileFloats : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const bool odd = q & 1;
    const MclLayout L = mcl_layout(a.H);
    float* tab = smem;
    const MclComp K = mcl_comp(tab + s16_tab_floats(M16::NG), L.C);
    float* ubase = K.bfc + 2;
    ubase += (4 - ((ubase - smem) & 3)) & 3;
    stage_params(ubase, a.params, L.P);
    mcl_compose(K, ubase, L);
    m16_build_table(tab, K, ubase, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float* fl = ubase + (size_t)wave * kWave;
    float2* dys = reinterpret_cast<float2*>(fl + 16 * kMclRowF);
    float* dfl = reinterpret_cast<float*>(dys + 16 * kChunkPad);
    float* tiles = dfl + (DX ? 16 * kMclRowF : 0);
    M16Grad G;
    G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * 2 * 64 + lane;
        f32x4 dh = {0.f, 0.f, 0.f, 0.f}, dc = dh;
        int cur_chunk = -1;
        // feature gradients of the finished chunk -> dL/dx of its samples: d a = d f2 + 2 a d f3 + 3 a^2 d f4, a = |x|
        auto flush_dx = [&](int pt0) {
            const int len = min(kChunk, a.T - pt0);
            float2* g2 = reinterpret_cast<float2*>(a.dx);
#pragma unroll
            for (int j = 0; j < 16 * kChunk / 64; ++j) {
                const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
                const float* f = fl + m * kMclRowF + 5 * (tt + kMclHalo);
                const float* d = dfl + m * kMclRowF + 5 * (tt + kMclHalo);
                const float da = __builtin_fmaf(3.0f * f[3], d[4], __builtin_fmaf(2.0f * f[2], d[3], d[2])) * fast_rcp(f[2]);
                if (tt < len && b0 + m < a.B) g2[(size_t)(b0 + m) * a.T + pt0 + tt] = make_float2(__builtin_fmaf(da, f[0], d[0]), __builtin_fmaf(da, f[1], d[1]));
            }
        };
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    wave_lds_fence();
                    if (cur_chunk >= 0) flush_dx(cur_chunk * kChunk);
                    // hand-over: what the finished chunk put before its own t0 (20 floats per sequence) belongs to the end of the earlier one
                    float carry[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int e = lane + 64 * j, m = e / 20, i = e % 20;
                        carry[j] = cur_chunk < 0 ? 0.0f : dfl[m * kMclRowF + i];
                    }
                    wave_lds_fence();
                    for (int e = lane; e < 16 * kMclRowF; e += 64) dfl[e] = 0.0f;
                    wave_lds_fence();
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const int e = lane + 64 * j, m = e / 20, i = e % 20;
                        dfl[m * kMclRowF + 5 * kChunk + i] = carry[j];
                    }
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                m16_stage_feat(fl, a.x, b0, a.B, a.T, t0, len, lane);
                stage_in<16>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 h0 = blk ? as_f32x4(ck[((size_t)blk * 2) * 64]) : z4, c0 = blk ? as_f32x4(ck[((size_t)blk * 2 + 1) * 64]) : z4;
            if (nstep == S) m16_bwd_block<NW, DX, true>(tl, odd, G, fl, dys, dfl, tiles, n, q, tb - t0, nstep, h0, c0, dh, dc);
            else m16_bwd_block<NW, DX, false>(tl, odd, G, fl, dys, dfl, tiles, n, q, tb - t0, nstep, h0, c0, dh, dc);
        }
        if constexpr (DX) {
            wave_lds_fence();
            flush_dx(0);
            // the circular window: what chunk 0 put before the frame start belongs to the frame's LAST four samples (already written)
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            {
                const int m = lane >> 2, i = lane & 3;
                const float* f = fl + m * kMclRowF + 5 * i;
                const float* d = dfl + m * kMclRowF + 5 * i;
                const float da = __builtin_fmaf(3.0f * f[3], d[4], __builtin_fmaf(2.0f * f[2], d[3], d[2])) * fast_rcp(f[2]);
                if (b0 + m < a.B) {
                    float2* g2 = reinterpret_cast<float2*>(a.dx) + (size_t)(b0 + m) * a.T + a.T - kMclHalo + i;
                    float2 v = *g2;
                    v.x += __builtin_fmaf(da, f[0], d[0]); v.y += __builtin_fmaf(da, f[1], d[1]);
                    *g2 = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            wave_lds_fence();
        }
    }
    if constexpr (NW) {
        const int P4 = L.P + kLossCols, Z = 5 * L.C;
        __syncthreads();
        float* raw = ubase;                                     // [wave][kMclRaw], then the parameters again, then Gz
        float* pl = raw + nwb * kMclRaw;
        float* Gz = pl + pad4(L.P);
        m16_write_raw(raw + wave * kMclRaw, G, lane, n, q);
        for (int i = threadIdx.x; i < L.P; i += blockDim.x) pl[i] = a.params[i];
        __syncthreads();
        for (int i = threadIdx.x; i < kMclRaw; i += blockDim.x) {
            float v = raw[i];
            for (int wv = 1; wv < nwb; ++wv) v += raw[wv * kMclRaw + i];
            raw[i] = v;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < Z * kMclPP; i += blockDim.x) {
            const int zi = i / kMclPP, pp = i % kMclPP;
            float acc = 0.0f;
            for (int g = 0; g < kMclG; ++g) acc = __builtin_fmaf(pl[L.o_wih + g * Z + zi], pp < kMclP ? raw[g * kMclP + pp] : raw[kMclG * kMclP + g], acc);
            Gz[i] = acc;
        }
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) prow[i] = i < L.P ? m16_param_grad(i, raw, Gz, K, pl, L) : 0.0f;
    }
}

LaunchShape m16_shape(int ngroups, int waves) {
    LaunchShape ls;
    ls.waves = waves;
    const int need = (ngroups + waves - 1) / waves, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
size_t m16_front_floats(int C) { return (size_t)s16_tab_floats(M16::NG) + mcl_comp_floats(C) + 4; }
template <bool NW, bool DX>
int m16_launch_bwd(hipStream_t st, const SeqArgs& a, int P, int C) {
    const LaunchShape ls = m16_shape(a.ngroups, 4);
    size_t body = (size_t)ls.waves * ((DX ? 2 : 1) * 16 * kMclRowF + 2 * 16 * kChunkPad + (NW ? M16::kTiles * kTileFloats : 0));
    const size_t tail = (size_t)ls.waves * kMclRaw + pad4(P) + 5 * C * kMclPP;
    if (body < (size_t)pad4(P)) body = pad4(P);
    if (NW && body < tail) body = tail;
    const size_t lds = (m16_front_floats(C) + body) * sizeof(float);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = mcl16_bwd_kernel<NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}


// -------------------------------------------------------------------------------------------------
// Fused train kernel for the reference's own batch sizes (train_funcs.py:28-48; a frame gets a CU of its own): ONE sequence per eight-wave
// workgroup.  Wave 0 runs the LSTM(8) recurrences gate-parallel (row k of the wave = gate k — i | f | g | o —, one rotated dot product per
// step and orientation, gates / h / c / tanh c of the frame parked in LDS, the step's W_hh gradient as one 4-block MFMA); everything that
// does not depend on h is spread over the 512 threads with thread = time step: the feature table, gates_in = A vec(P) + b of every step,
// the read-out with loss and dL/dy, and — on the matrix pipe, an eighth of the frame per wave — dA | db = sum_t d_gates(t) (x) [P(t), 1].
// The prologue composes A, b, W_fc (mcl_compose), the epilogue takes dA, db, dW_hh, dW_fc, db_fc back to the gradients of the
// convolutions, W_ih, the biases and the two linear layers (m16_param_grad), all four waves at work.  Weight gradients only (the
// frozen-PA role stays on the S16 kernels).  Taken while the frame's state fits the CU's LDS.
// -------------------------------------------------------------------------------------------------
constexpr int kMgpP16 = 17, kMgpP32 = 33;
constexpr int kMgpWaves = 8, kMgpThreads = 64 * kMgpWaves;     // waves of the train kernel's workgroup (wave 0: the recurrences; all: everything else)
struct MgpBuf { int ft, gin, gts, hist, cpk, tpk, dyb, dump, total; };
__host__ __device__ inline MgpBuf mgp_buf(int T, int C) {
    MgpBuf b; int o = 0;
    b.ft = o; o += 5 * (T + kMclHalo) + 3;   // [(T + 4)][5]: row i <-> time i - 4 (the frame's last four samples in front); patch of step t = 25 floats from row t
    b.gin = o; o += kMgpP32 * T;             // [T][33]: gates_in, gate k at 8 k + unit; overwritten by d_gates in the backward steps
    b.gts = o; o += kMgpP32 * T;             // the four gates of step t
    b.hist = o; o += kMgpP16 * (T + 1);      // entry t + 1 = h(t), entry 0 = 0
    b.cpk = o; o += kMgpP16 * (T + 1);       // c likewise
    b.tpk = o; o += kMgpP16 * T;             // tanh c(t)
    b.dyb = o; o += 2 * T + 2;               // float2 [T]
    b.dump = o; o += 256;
    const int tail = kMgpWaves * kMclRaw + 5 * C * kMclPP + 8;      // the epilogue's [wave][raw] and Gz share the frame buffers
    b.total = o > tail ? o : tail;
    return b;
}
__global__ __launch_bounds__(kMgpThreads) void mcl_gp_train_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, role = lane >> 4, cu = col & 7;
    const MclLayout L = mcl_layout(a.H);
    const int T = a.T, C = L.C, Z = 5 * C;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    const MclComp K = mcl_comp(pl + pad4(L.P), C);
    mcl_compose(K, pl, L);
    float* buf = pl + pad4(L.P) + pad4(mcl_comp_floats(C) + kMgpWaves);
    const MgpBuf O = mgp_buf(T, C);
    float *ft = buf + O.ft, *gin = buf + O.gin, *gts = buf + O.gts, *hist = buf + O.hist, *cpk = buf + O.cpk, *tpk = buf + O.tpk;
    float2* dyb = reinterpret_cast<float2*>(buf + ((O.dyb + 1) & ~1));
    float* dump = buf + O.dump;
    float* loss4 = K.bfc + 2;                               // (kMgpWaves floats of padding after the composed operands)
    // wave 0: the row's recurrent block W_hh[gate][unit][:] and its transpose, rotated for this lane
    float wF[16], wT[16];
    {
        const int dir = rot_dir(col);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15;
            const bool ok = col < kMclH && m < kMclH;
            wF[k] = ok ? pl[L.o_whh + (role * kMclH + col) * kMclH + m] : 0.0f;
            wT[k] = ok ? pl[L.o_whh + (role * kMclH + m) * kMclH + col] : 0.0f;
        }
    }
    const bool vo = col < kMclH, is_g = role == 2;
    const float wfc0 = vo ? K.wfc[col] : 0.0f, wfc1 = vo ? K.wfc[kMclH + col] : 0.0f;
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    const int dmp = (int)(dump - smem) + lane;
    // per-step stores of the forward pass: every valid lane its own gate; row 1 h, row 2 c, row 3 tanh c
    const int pg0 = vo ? (int)(gts - smem) + role * kMclH + col : dmp, pg_step = vo ? kMgpP32 : 0;
    const int ps0 = role == 1 ? (int)(hist - smem) + kMgpP16 + col : role == 2 ? (int)(cpk - smem) + kMgpP16 + col : role == 3 ? (int)(tpk - smem) + col : dmp;
    const int ps_step = role == 0 ? 0 : kMgpP16;
    const int dg0 = vo ? (int)(gin - smem) + role * kMclH + col : dmp, dg_step = vo ? kMgpP32 : 0;

    f32x16 acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc1[i] = 0.0f;
    f32x4 dA[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dA[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dwfc[2][kMclH], dbfc0 = 0.0f, dbfc1 = 0.0f, loss_acc = 0.0f;
#pragma unroll
    for (int j = 0; j < kMclH; ++j) { dwfc[0][j] = 0.0f; dwfc[1][j] = 0.0f; }

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        __syncthreads();
        // ---- the feature table (circular window: mcldnn.py:115-118) ----
        for (int i = tid; i < T + kMclHalo; i += kMgpThreads) {
            const int t = i - kMclHalo;
            const float2 xv = xg[t < 0 ? t + T : t];
            const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), am = __builtin_amdgcn_sqrtf(a2);
            float* d = ft + 5 * i;
            d[0] = xv.x; d[1] = xv.y; d[2] = am; d[3] = a2; d[4] = a2 * am;
        }
        if (tid < 16) { hist[tid] = 0.0f; cpk[tid] = 0.0f; }
        __syncthreads();
        // ---- gates_in of every step; thread = time step ----
        for (int t = tid; t < T; t += kMgpThreads) {
            float pv[kMclP];
#pragma unroll
            for (int p = 0; p < kMclP; ++p) pv[p] = ft[5 * t + p];
            for (int g = 0; g < kMclG; ++g) {
                const float* ar = K.A + g * kMclPP;
                float acc = ar[kMclP];
#pragma unroll
                for (int p = 0; p < kMclP; ++p) acc = __builtin_fmaf(ar[p], pv[p], acc);
                gin[t * kMgpP32 + g] = acc;
            }
        }
        __syncthreads();
        // ---- forward recurrence (wave 0) ----
        if (wave == 0) {
            float h = 0.0f, c = 0.0f;
            int pg = pg0, ps = ps0;
            for (int t = 0; t < T; ++t) {
                const float acc = rotdot(gin[t * kMgpP32 + role * kMclH + cu], wF, h);
                const float sg = sigmoidf_(acc), th = tanhf_(acc);
                const float v = is_g ? th : sg;
                float g4[4];
                gather_rows(v, g4);
                c = __builtin_fmaf(g4[1], c, g4[0] * g4[2]);
                const float tc = tanhf_(c);
                h = g4[3] * tc;
                smem[pg] = v;
                smem[ps] = role == 1 ? h : role == 2 ? c : tc;
                pg += pg_step; ps += ps_step;
            }
        }
        __syncthreads();
        // ---- read-out, loss and dL/dy of every step; thread = time step ----
        for (int t = tid; t < T; t += kMgpThreads) {
            const float* hv = hist + (t + 1) * kMgpP16;
            float y0 = K.bfc[0], y1 = K.bfc[1];
#pragma unroll
            for (int j = 0; j < kMclH; ++j) { y0 = __builtin_fmaf(K.wfc[j], hv[j], y0); y1 = __builtin_fmaf(K.wfc[kMclH + j], hv[j], y1); }
            const float2 tv = tg[t];
            float dy0, dy1;
            s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
            dbfc0 += dy0; dbfc1 += dy1;
            dyb[t] = make_float2(dy0, dy1);
#pragma unroll
            for (int j = 0; j < kMclH; ++j) { dwfc[0][j] = __builtin_fmaf(dy0, hv[j], dwfc[0][j]); dwfc[1][j] = __builtin_fmaf(dy1, hv[j], dwfc[1][j]); }
        }
        __syncthreads();
        // ---- backward recurrence (wave 0) ----
        if (wave == 0) {
            float dh = 0.0f, dc = 0.0f;
            int dg = dg0 + (T - 1) * dg_step;
            for (int t = T - 1; t >= 0; --t) {
                const float* gr = gts + t * kMgpP32 + cu;
                const float gi = gr[0], gf = gr[kMclH], gg = gr[2 * kMclH], go = gr[3 * kMclH];
                const float hp = hist[t * kMgpP16 + col], cp = cpk[t * kMgpP16 + col], tc = tpk[t * kMgpP16 + col];
                const float2 dyv = dyb[t];
                const float dht = __builtin_fmaf(dyv.y, wfc1, __builtin_fmaf(dyv.x, wfc0, dh));
                const float dct = __builtin_fmaf(dht * go, __builtin_fmaf(-tc, tc, 1.0f), dc);
                dc = dct * gf;
                // the row's own pre-activation gradient: d_i = dc g i (1 - i), d_f = dc c(t-1) f (1 - f), d_g = dc i (1 - g^2), d_o = dh tanh c o (1 - o)
                const float own = role == 0 ? gi : role == 1 ? gf : role == 2 ? gg : go;
                const float mul = role == 0 ? gg : role == 1 ? cp : role == 2 ? gi : tc;
                const float up = (role == 3 ? dht : dct) * mul;
                const float d_row = vo ? up * (is_g ? __builtin_fmaf(-own, own, 1.0f) : own * (1.0f - own)) : 0.0f;
                float part = rotdot(0.0f, wT, d_row);
                part = sum_rows4(part);
                dh = part;
                smem[dg] = d_row;
                dg -= dg_step;
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_row, hp, acc1, 0, 0, 0);
            }
        }
        __syncthreads();
        // ---- dA | db: sum over time of d_gates (x) [patch, 1] on the matrix pipe, a quarter of the 4-step slices per wave ----
        for (int t4 = 4 * wave; t4 < T; t4 += 4 * kMgpWaves) {
            const int t = t4 + role;
            const bool ok = t < T;
            const int tc_ = ok ? t : 0;
            const float a0 = ok ? gin[tc_ * kMgpP32 + col] : 0.0f, a1 = ok ? gin[tc_ * kMgpP32 + 16 + col] : 0.0f;
            const float b0 = ok ? ft[5 * tc_ + col] : 0.0f;
            const float b1 = !ok ? 0.0f : (col < kMclP - 16 ? ft[5 * tc_ + 16 + col] : (col == kMclP - 16 ? 1.0f : 0.0f));
            dA[0][0] = mfma4(a0, b0, dA[0][0]); dA[0][1] = mfma4(a0, b1, dA[0][1]);
            dA[1][0] = mfma4(a1, b0, dA[1][0]); dA[1][1] = mfma4(a1, b1, dA[1][1]);
        }
    }
    // ---- epilogue: the waves' raw sums, then the chain rule back to the parameters (as mcl16_bwd_kernel) ----
    __syncthreads();
    float* raw = buf;                                       // [wave][kMclRaw], then Gz
    float* Gz = raw + kMgpWaves * kMclRaw;
    {
        float* rw = raw + wave * kMclRaw;
        float* rA = rw; float* rb = rA + kMclG * kMclP; float* rhh = rb + kMclG; float* rfc = rhh + kMclG * kMclH; float* rbf = rfc + 2 * kMclH;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int g = 16 * mt + 4 * role + rr, p = 16 * nt + col;
                    if (p < kMclP) rA[g * kMclP + p] = dA[mt][nt][rr];
                    else if (p == kMclP) rb[g] = dA[mt][nt][rr];
                }
        // 4-block MFMA: block k = gate k; register 4 k + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = 4 * role + rr;
                if (i < kMclH && col < kMclH) rhh[(k * kMclH + i) * kMclH + col] = wave == 0 ? acc1[4 * k + rr] : 0.0f;
            }
        float lp = loss_acc, s0 = dbfc0, s1 = dbfc1;
        for (int o = 32; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
#pragma unroll
        for (int j = 0; j < kMclH; ++j) {
            float v0 = dwfc[0][j], v1 = dwfc[1][j];
            for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); }
            if (lane == 0) { rfc[j] = v0; rfc[kMclH + j] = v1; }
        }
        if (lane == 0) { rbf[0] = s0; rbf[1] = s1; loss4[wave] = lp; }
    }
    __syncthreads();
    for (int i = tid; i < kMclRaw; i += kMgpThreads) {
        float v = raw[i];
#pragma unroll
        for (int w = 1; w < kMgpWaves; ++w) v += raw[w * kMclRaw + i];
        raw[i] = v;
    }
    __syncthreads();
    for (int i = tid; i < Z * kMclPP; i += kMgpThreads) {
        const int zi = i / kMclPP, pp = i % kMclPP;
        float acc = 0.0f;
        for (int g = 0; g < kMclG; ++g) acc = __builtin_fmaf(pl[L.o_wih + g * Z + zi], pp < kMclP ? raw[g * kMclP + pp] : raw[kMclG * kMclP + g], acc);
        Gz[i] = acc;
    }
    __syncthreads();
    const int P4 = L.P + kLossCols;
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    float loss_total = 0.0f;
#pragma unroll
    for (int w = 0; w < kMgpWaves; ++w) loss_total += loss4[w];
    for (int i = tid; i < P4; i += kMgpThreads) prow[i] = i < L.P ? m16_param_grad(i, raw, Gz, K, pl, L) : (i == L.P ? loss_total : 0.0f);
}

// Evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): ONE sequence per four-wave workgroup, the forward
// half of mcl_gp_train_kernel in chunks of kMevChunk steps (feature table, gates_in and the read-out with thread = time step, the recurrence on wave 0);
// no checkpoints.
constexpr int kMevChunk = 256;
constexpr int kMevFloats = 5 * (kMevChunk + kMclHalo) + 3 + kMgpP32 * kMevChunk + kMgpP16 * kMevChunk + 64;
__global__ __launch_bounds__(256) void mcl_gp_eval_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int Tc = kMevChunk;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, role = lane >> 4, cu = col & 7;
    const MclLayout L = mcl_layout(a.H);
    const int T = a.T, C = L.C;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    const MclComp K = mcl_comp(pl + pad4(L.P), C);
    mcl_compose(K, pl, L);
    float* ft = pl + pad4(L.P) + pad4(mcl_comp_floats(C) + 4);     // [(Tc + 4)][5]: row i <-> time t0 - 4 + i
    float* gin = ft + 5 * (Tc + kMclHalo) + 3;                      // [Tc][33]
    float* hist = gin + kMgpP32 * Tc;                               // [Tc][17]: h(t0 + tt)
    float* dump = hist + kMgpP16 * Tc;
    float wF[16];
    {
        const int dir = rot_dir(col);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (col + dir * k) & 15;
            wF[k] = (col < kMclH && m < kMclH) ? pl[L.o_whh + (role * kMclH + col) * kMclH + m] : 0.0f;
        }
    }
    const bool is_g = role == 2;
    const int pk0 = role == 0 ? (int)(hist - smem) + col : (int)(dump - smem) + lane, pk_step = role == 0 ? kMgpP16 : 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float h = 0.0f, c = 0.0f;
        for (int t0 = 0; t0 < T; t0 += Tc) {
            const int len = min(Tc, T - t0);
            __syncthreads();
            for (int i = tid; i < len + kMclHalo; i += 256) {
                const int t = t0 - kMclHalo + i;
                const float2 xv = xg[t < 0 ? t + T : t];
                const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), am = __builtin_amdgcn_sqrtf(a2);
                float* d = ft + 5 * i;
                d[0] = xv.x; d[1] = xv.y; d[2] = am; d[3] = a2; d[4] = a2 * am;
            }
            __syncthreads();
            for (int tt = tid; tt < len; tt += 256) {
                float pv[kMclP];
#pragma unroll
                for (int p = 0; p < kMclP; ++p) pv[p] = ft[5 * tt + p];
                for (int g = 0; g < kMclG; ++g) {
                    const float* ar = K.A + g * kMclPP;
                    float acc = ar[kMclP];
#pragma unroll
                    for (int p = 0; p < kMclP; ++p) acc = __builtin_fmaf(ar[p], pv[p], acc);
                    gin[tt * kMgpP32 + g] = acc;
                }
            }
            __syncthreads();
            if (wave == 0) {
                int pk = pk0;
                for (int tt = 0; tt < len; ++tt) {
                    const float acc = rotdot(gin[tt * kMgpP32 + role * kMclH + cu], wF, h);
                    const float sg = sigmoidf_(acc), th = tanhf_(acc);
                    float g4[4];
                    gather_rows(is_g ? th : sg, g4);
                    c = __builtin_fmaf(g4[1], c, g4[0] * g4[2]);
                    h = g4[3] * tanhf_(c);
                    smem[pk] = h;
                    pk += pk_step;
                }
            }
            __syncthreads();
            for (int tt = tid; tt < len; tt += 256) {
                const float* hv = hist + tt * kMgpP16;
                float y0 = K.bfc[0], y1 = K.bfc[1];
#pragma unroll
                for (int j = 0; j < kMclH; ++j) { y0 = __builtin_fmaf(K.wfc[j], hv[j], y0); y1 = __builtin_fmaf(K.wfc[kMclH + j], hv[j], y1); }
                yg[t0 + tt] = make_float2(y0, y1);
            }
        }
    }
}

static size_t mcl_gp_lds_bytes(int C, int T) {
    return ((size_t)pad4(mcl_layout(C).P) + pad4(mcl_comp_floats(C) + kMgpWaves) + mgp_buf(T, C).total) * sizeof(float);
}
static int mcl_gp_blocks_per_cu(int C, int T) {
    const size_t lds = mcl_gp_lds_bytes(C, T);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 2 ? n : 2;
}

}  // namespace

bool mcldnn_ok(const odpd_model_t* m) { return m->hidden >= 1 && m->hidden <= kMclMaxC; }
int64_t mcldnn_param_count(const odpd_model_t* m) {
    return mcldnn_ok(m) ? (int64_t)mcl_layout(m->hidden).P : (int64_t)ODPD_EUNSUPPORTED;
}
int mcldnn_rows(const odpd_model_t* m, int B) {
    if (!mcldnn_ok(m)) return ODPD_EUNSUPPORTED;
    return m16_shape((B + 15) / 16, 4).grid;
}
int64_t mcldnn_ckpt_floats(const odpd_model_t* m, int B, int T) {
    if (!mcldnn_ok(m)) return ODPD_EUNSUPPORTED;
    return (int64_t)((B + 15) / 16) * num_ckpt(T) * 2 * 256;
}
// the fused train kernel of the reference's batch sizes: one sequence per four-wave workgroup, the frame's state in LDS
bool mcldnn_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (!mcldnn_ok(m) || T < kMclHalo) return false;
    const int per_cu = mcl_gp_blocks_per_cu(m->hidden, T);
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && per_cu > 0;
    // up to four rounds of workgroups (measured: profiles/r03/gp_train_bench_f4.txt): the alternative is the forward / loss / backward chain of the S16 kernels
    return (long)B <= 4L * device_cus() * per_cu;
}
int mcldnn_gp_rows(const odpd_model_t* m, int B, int T) {
    const long cap = (long)device_cus() * mcl_gp_blocks_per_cu(m->hidden, T);
    return B < cap ? B : (int)cap;
}
int mcldnn_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const size_t lds = mcl_gp_lds_bytes(m->hidden, a.T);
    if (int e = allow_big_lds(mcl_gp_train_kernel, lds)) return e;
    hipLaunchKernelGGL(mcl_gp_train_kernel, dim3(mcldnn_gp_rows(m, a.B, a.T)), dim3(kMgpThreads), lds, st, a);
    return (int)hipGetLastError();
}
// mode 1 forward, 2 backward
int mcldnn_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    if (!mcldnn_ok(m)) return ODPD_EUNSUPPORTED;
    if (a0.T < kMclHalo) return ODPD_EINVAL;        // the circular window takes the frame's own last four samples (mcldnn.py:115-118)
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int C = m->hidden, P = mcl_layout(C).P;
    if (mode == 1 && !a.ckpt && a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0) {
        // sequences that each get a CU of their own (inference: no checkpoints)
        const size_t lds = ((size_t)pad4(P) + pad4(mcl_comp_floats(C) + 4) + kMevFloats) * sizeof(float);
        if (int e = allow_big_lds(mcl_gp_eval_kernel, lds)) return e;
        hipLaunchKernelGGL(mcl_gp_eval_kernel, dim3(a.B), dim3(256), lds, st, a);
        return (int)hipGetLastError();
    }
    if (mode == 1) {
        const LaunchShape ls = m16_shape(a.ngroups, a.ngroups <= 4 * device_cus() ? 4 : 8);
        size_t body = (size_t)ls.waves * (16 * kMclRowF + 2 * 16 * kChunkPad);
        if (body < (size_t)pad4(P)) body = pad4(P);
        const size_t lds = (m16_front_floats(C) + body) * sizeof(float);
        auto k = mcl16_fwd_kernel;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    }
    if (!a.ckpt && a.nck > 1) return ODPD_EINVAL;
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (!nw && !dx) return ODPD_EINVAL;
    if (nw && dx) return m16_launch_bwd<true, true>(st, a, P, C);
    if (nw) return m16_launch_bwd<true, false>(st, a, P, C);
    return m16_launch_bwd<false, true>(st, a, P, C);
}

}  // namespace odpd
