// qgru_family.hip — quantisation-aware QGRU (reference: quant/__init__.py:20-37 -> quant/quant_envs.py:138-306 applied
// to backbones/qgru.py / qgru_amp1.py; cell quant/modules/gru.py:43-59; quantisers quant/qmodules/quantizers.py:15-97).
//
// Fake-quant arithmetic on power-of-two grids, restated so that the integer-grid quantities are BIT-EXACT:
//   s = 2^round(log2|scale|);  q(x) = rint(clamp(x/s, Qn, Qp)) * s   (clamp before round, round half to even)
//   x_t = (sum_i q_a(feat)_i q_w(W_x)_ki) + b_x      — the products/sums are exact integers on the grid for 8 bits;
//   h_t = (sum_m q_a(h)_m  q_w(W_h)_km)  + b_h        the fp32 bias is added once (this is what F.linear does)
//   r = q_s(sig(q_add(x_r + h_r))), z likewise, n = q_t(tanh(q_add(x_n + q_mul(r h_n)))),
//   h' = q_add(q_mul(z h) + q_mul((1-z) n)),  y = fc_out(q_a(h')) (+ 16-bit output quantiser in eval mode).
// The inputs of sigmoid/tanh live on the add-quantiser grid: for <= 8 activation bits they are looked up in two 2^bits
// tables built in LDS at kernel start with double-precision math (no transcendental in the loop, identical on every
// lane/launch); for wider grids they are evaluated in double per element.  This file is compiled with FP contraction
// off: a fused multiply-add would change roundings the reference does not have.
// Backward = straight-through estimator (gradient passes where x/s lies inside [Qn,Qp]); the 13 scale parameters get an
// exact 0 gradient.  One 16-lane row per sequence (H <= 16); dL/dx = the quantised W_x^T behind the straight-through mask of
// the x2h activation quantiser and the feature Jacobian (QAT model as the frozen PA / x.requires_grad).
#include "odpd_seq.h"

#pragma clang fp contract(off)

namespace odpd {

struct QgruLayout {
    int H, o_wx, o_bx, o_sxw, o_sxa, o_sxo, o_wh, o_bh, o_shw, o_sha, o_sho, o_ssig, o_stanh, o_sadd, o_smul, o_wo, o_bo,
        o_sow, o_soa, o_soo, P;
};
__host__ __device__ inline QgruLayout qgru_layout(int H) {
    QgruLayout L; L.H = H; int o = 0;
    L.o_wx = o; o += 3 * H * 4; L.o_bx = o; o += 3 * H; L.o_sxw = o++; L.o_sxa = o++; L.o_sxo = o++;
    L.o_wh = o; o += 3 * H * H; L.o_bh = o; o += 3 * H; L.o_shw = o++; L.o_sha = o++; L.o_sho = o++;
    L.o_ssig = o++; L.o_stanh = o++; L.o_sadd = o++; L.o_smul = o++;
    L.o_wo = o; o += 2 * H; L.o_bo = o; o += 2; L.o_sow = o++; L.o_soa = o++; L.o_soo = o++;
    L.P = o;
    return L;
}
constexpr int kQTabFloats = 6 * 4 * 64 * 4;   // q_w(W_h) rows r,z,n + transposes

__device__ __forceinline__ float pow2_scale(float scale) { return exp2f(rintf(log2f(fabsf(scale)))); }
struct Quant { float s, inv, qn, qp; };
__device__ __forceinline__ Quant make_quant(float scale, int bits) {
    Quant q; q.s = pow2_scale(scale); q.inv = 1.0f / q.s; q.qn = -(float)(1 << (bits - 1)); q.qp = (float)((1 << (bits - 1)) - 1);
    return q;
}
// clamp as one v_med3_f32; the straight-through pass mask "Qn <= x/s <= Qp" is "the clamp left x/s unchanged": one compare
__device__ __forceinline__ float qapply(float x, const Quant& q) {
    const float v = x * q.inv;
    return rintf(__builtin_amdgcn_fmed3f(v, q.qn, q.qp)) * q.s;
}
__device__ __forceinline__ float qpass(float x, const Quant& q) {
    const float v = x * q.inv;
    return __builtin_amdgcn_fmed3f(v, q.qn, q.qp) == v ? 1.0f : 0.0f;
}

template <bool WITH_T>
__device__ __forceinline__ void fill_qgru_tabs(float* tab, const float* pl, const QgruLayout& L, const Quant& qw, int lane,
                                               int wave, int nwb) {
    const int H = L.H, col = lane & 15, o = col, dir = rot_dir(col);
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int idx = wave; idx < 6 * 4; idx += nwb) {
        const int tr = idx >> 2, q = idx & 3;
        const bool transposed = tr >= 3;
        if (!WITH_T && transposed) continue;
        const int g = transposed ? tr - 3 : tr;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = (col + dir * (4 * q + e)) & 15;
            const bool ok = o < H && m < H;
            v[e] = ok ? qapply(pl[L.o_wh + (transposed ? (g * H + m) * H + o : (g * H + o) * H + m)], qw) : 0.0f;
        }
        t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
// sigmoid / tanh tables over the add-quantiser grid (index = integer grid value - Qn), double-precision evaluation
__device__ __forceinline__ void fill_luts(float* lut, const Quant& qadd, int bits) {
    const int n = 1 << bits;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)((float)(i + (int)qadd.qn) * qadd.s);
        lut[i] = (float)(1.0 / (1.0 + exp(-x)));
        lut[n + i] = (float)tanh(x);
    }
}

struct QgruW {
    float wx[3][4], bx[3], bh[3], wo[2], bo[2];
    Quant qxa, qha, qoa, qsig, qtanh, qadd, qmul, qout;
};
template <bool AMP1>
__device__ __forceinline__ void qgru_feat(float2 xv, float (&f)[4]) {
    const float a2 = xv.x * xv.x + xv.y * xv.y;
    f[0] = xv.x; f[1] = xv.y;
    if constexpr (AMP1) { const float a = sqrtf(a2); f[2] = a; f[3] = a * a * a; }   // qgru_amp1.py:63-70
    else { f[2] = a2; f[3] = a2 * a2; }                                             // qgru.py:61-66
}

// one forward step; everything the straight-through backward needs comes out through `sv`
struct QStep { float hp, hq, z, n, nf, htn, r, rf, zf, hnew; unsigned mask; float fsel; };
enum { M_PH = 1, M_AR = 2, M_AZ = 4, M_R = 8, M_Z = 16, M_M1 = 32, M_AN = 64, M_N = 128, M_M2 = 256, M_M3 = 512, M_AH = 1024 };

template <bool LUT>
__device__ __forceinline__ void gate_fn(float a, const QgruW& w, const float* lut, int nlut, float& sg, float& th) {
    if constexpr (LUT) {
        const int idx = (int)(a * w.qadd.inv) - (int)w.qadd.qn;
        sg = lut[idx]; th = lut[nlut + idx];
    } else {
        sg = (float)(1.0 / (1.0 + exp(-(double)a))); th = (float)tanh((double)a);
    }
}

template <bool AMP1, bool LUT>
__device__ __forceinline__ float qgru_cell_fwd(const QgruW& w, const float (&whq)[3][16], const float* lut, int nlut,
                                               const float (&f)[4], int col, float h, QStep& sv) {
    float fq[4], xs0 = 0.f, xs1 = 0.f, xs2 = 0.f;
    sv.fsel = (col == 4) ? 1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fq[i] = qapply(f[i], w.qxa);
        sv.fsel = (col == i) ? fq[i] : sv.fsel;
        xs0 = __builtin_fmaf(w.wx[0][i], fq[i], xs0);   // exact on the grid (fma == mul+add here)
        xs1 = __builtin_fmaf(w.wx[1][i], fq[i], xs1);
        xs2 = __builtin_fmaf(w.wx[2][i], fq[i], xs2);
    }
    unsigned mk = 0;
    const float hq = qapply(h, w.qha);
    mk |= qpass(h, w.qha) != 0.0f ? M_PH : 0;
    float hs0 = 0.f, hs1 = 0.f, hs2 = 0.f;
    rotdot3(hs0, hs1, hs2, whq[0], whq[1], whq[2], hq);
    const float xr = xs0 + w.bx[0], xz = xs1 + w.bx[1], xn = xs2 + w.bx[2];
    const float hr = hs0 + w.bh[0], hz = hs1 + w.bh[1], hn = hs2 + w.bh[2];
    const float vr = xr + hr, vz = xz + hz;
    const float ar = qapply(vr, w.qadd), az = qapply(vz, w.qadd);
    mk |= qpass(vr, w.qadd) != 0.0f ? M_AR : 0; mk |= qpass(vz, w.qadd) != 0.0f ? M_AZ : 0;
    float rf, zf, dmy;
    gate_fn<LUT>(ar, w, lut, nlut, rf, dmy);
    gate_fn<LUT>(az, w, lut, nlut, zf, dmy);
    const float r = qapply(rf, w.qsig), z = qapply(zf, w.qsig);
    mk |= qpass(rf, w.qsig) != 0.0f ? M_R : 0; mk |= qpass(zf, w.qsig) != 0.0f ? M_Z : 0;
    const float pm1 = r * hn, m1 = qapply(pm1, w.qmul);
    mk |= qpass(pm1, w.qmul) != 0.0f ? M_M1 : 0;
    const float vn = xn + m1, an = qapply(vn, w.qadd);
    mk |= qpass(vn, w.qadd) != 0.0f ? M_AN : 0;
    float nf;
    gate_fn<LUT>(an, w, lut, nlut, dmy, nf);
    const float n = qapply(nf, w.qtanh);
    mk |= qpass(nf, w.qtanh) != 0.0f ? M_N : 0;
    const float pm2 = z * h, pm3 = (1.0f - z) * n;
    const float m2 = qapply(pm2, w.qmul), m3 = qapply(pm3, w.qmul);
    mk |= qpass(pm2, w.qmul) != 0.0f ? M_M2 : 0; mk |= qpass(pm3, w.qmul) != 0.0f ? M_M3 : 0;
    const float vh = m2 + m3, hnew = qapply(vh, w.qadd);
    mk |= qpass(vh, w.qadd) != 0.0f ? M_AH : 0;
    sv.hp = h; sv.hq = hq; sv.z = z; sv.n = n; sv.nf = nf; sv.htn = hn; sv.r = r; sv.rf = rf; sv.zf = zf; sv.hnew = hnew;
    sv.mask = mk;
    return hnew;
}

__device__ __forceinline__ void load_qgru_w(QgruW& w, const float* pl, const QgruLayout& L, int col, int bits_w, int bits_a) {
    const int H = L.H, o = col;
    const bool vo = o < H;
    const Quant qxw = make_quant(pl[L.o_sxw], bits_w), qow = make_quant(pl[L.o_sow], bits_w);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) w.wx[g][i] = vo ? qapply(pl[L.o_wx + (g * H + o) * 4 + i], qxw) : 0.0f;
        w.bx[g] = vo ? pl[L.o_bx + g * H + o] : 0.0f;
        w.bh[g] = vo ? pl[L.o_bh + g * H + o] : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) { w.wo[c] = vo ? qapply(pl[L.o_wo + c * H + o], qow) : 0.0f; w.bo[c] = pl[L.o_bo + c]; }
    w.qxa = make_quant(pl[L.o_sxa], bits_a); w.qha = make_quant(pl[L.o_sha], bits_a); w.qoa = make_quant(pl[L.o_soa], bits_a);
    w.qsig = make_quant(pl[L.o_ssig], bits_a); w.qtanh = make_quant(pl[L.o_stanh], bits_a);
    w.qadd = make_quant(pl[L.o_sadd], bits_a); w.qmul = make_quant(pl[L.o_smul], bits_a);
    w.qout = make_quant(pl[L.o_soo], 16);
}

// common prologue: params, tables, luts.  returns pointers
template <bool WITH_T, bool LUT>
__device__ __forceinline__ void qgru_prologue(const SeqArgs& a, float* smem, const QgruLayout& L, const LaneId& id, QgruW& w,
                                              TabPtr& tlane, const float*& lut, int& nlut, float*& wave_base,
                                              int wave_floats, int bits_w, int bits_a) {
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    load_qgru_w(w, pl, L, id.col, bits_w, bits_a);
    fill_qgru_tabs<WITH_T>(tab, pl, L, make_quant(pl[L.o_shw], bits_w), id.lane, id.wave, id.nwb);
    float* lutw = tab + kQTabFloats;
    nlut = LUT ? (1 << bits_a) : 0;
    if constexpr (LUT) fill_luts(lutw, w.qadd, bits_a);
    __syncthreads();
    tlane = to_tab(reinterpret_cast<const float4*>(tab) + id.lane);
    lut = lutw;
    wave_base = lutw + 2 * nlut + (size_t)id.wave * wave_floats;
}

template <bool AMP1, bool LUT>
__global__ __launch_bounds__(kMaxThreads) void qgru_fwd_kernel(SeqArgs a, int bits_w, int bits_a, int eval_mode) {
    constexpr int SPW = 4, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<1>();
    const int lane = id.lane, col = id.col, s = id.s;
    const QgruLayout L = qgru_layout(a.H);
    QgruW w; TabPtr tlane; const float* lut; int nlut; float* wb;
    qgru_prologue<false, LUT>(a, smem, L, id, w, tlane, lut, nlut, wb, 2 * 2 * SPW * kChunkPad, bits_w, bits_a);
    float2* xs = reinterpret_cast<float2*>(wb);
    float2* ys = xs + SPW * kChunkPad;
    float whq[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_rot(whq[g], tlane + g * 4 * 64);
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float h = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                float f[4];
                qgru_feat<AMP1>(xs[s * kChunkPad + tt], f);
                QStep sv;
                h = qgru_cell_fwd<AMP1, LUT>(w, whq, lut, nlut, f, col, h, sv);
                const float ho = qapply(h, w.qoa);
                float y0 = row_sum16(w.wo[0] * ho) + w.bo[0], y1 = row_sum16(w.wo[1] * ho) + w.bo[1];
                if (eval_mode) { y0 = qapply(y0, w.qout); y1 = qapply(y1, w.qout); }
                if (col == 0) ys[s * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (a.ckpt != nullptr && (t1 % S) == 0 && t1 < a.T) a.ckpt[((size_t)grp * a.nck + t1 / S) * 64 + lane] = h;
            }
            wave_lds_fence();
            stage_out<SPW>(ys, a.y, b0, a.B, a.T, t0, len, lane);
            wave_lds_fence();
        }
    }
}

struct QgruGrad {
    f32x4 tih[3], thh[3];
    float dbhn, dwo[2], dbo[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 3; ++g) { tih[g] = z4; thh[g] = z4; }
        dbhn = dwo[0] = dwo[1] = dbo[0] = dbo[1] = 0.f;
    }
};

template <bool AMP1, bool LUT, bool FULL, bool NW, bool DX>
__device__ __forceinline__ void qgru_bwd_block(const SeqArgs& a, const QgruW& w, TabPtr tlane, const float* lut, int nlut,
                                               QgruGrad& G, const LaneId& id, const float2* xs, const float2* dys, float2* dxs,
                                               int tloc, int nstep, float h, float& dh) {
    constexpr int S = kCkptStride;
    const int col = id.col, s = id.s;
    QStep sv[S];
    tlane = opaque(tlane);
    {
        float whq[3][16];
#pragma unroll
        for (int g = 0; g < 3; ++g) load_rot(whq[g], tlane + g * 4 * 64);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            if (FULL || i < nstep) {
                float f[4];
                qgru_feat<AMP1>(xs[s * kChunkPad + tloc + i], f);
                h = qgru_cell_fwd<AMP1, LUT>(w, whq, lut, nlut, f, col, h, sv[i]);
            }
        }
    }
    tlane = opaque(tlane);
    float whT[3][16];
#pragma unroll
    for (int g = 0; g < 3; ++g) load_rot(whT[g], tlane + (3 + g) * 4 * 64);
#pragma unroll
    for (int i = S - 1; i >= 0; --i) {
        if (FULL || i < nstep) {
            const QStep& v = sv[i];
            const float2 dyv = dys[s * kChunkPad + tloc + i];
            const unsigned mk = v.mask;
#define QM(bit) ((mk & (bit)) ? 1.0f : 0.0f)
            const float ho = qapply(v.hnew, w.qoa), pho = qpass(v.hnew, w.qoa);
            if constexpr (NW) {
                G.dwo[0] += dyv.x * ho; G.dwo[1] += dyv.y * ho;
                G.dbo[0] += dyv.x; G.dbo[1] += dyv.y;
            }
            const float dhn = dh + (dyv.x * w.wo[0] + dyv.y * w.wo[1]) * pho;
            const float g = dhn * QM(M_AH);
            const float dm2 = g * QM(M_M2), dm3 = g * QM(M_M3);
            const float dz = dm2 * v.hp - dm3 * v.n;
            const float dn = dm3 * (1.0f - v.z);
            const float dan = dn * QM(M_N) * (1.0f - v.nf * v.nf) * QM(M_AN);
            const float dm1 = dan * QM(M_M1);
            const float dr = dm1 * v.htn, dhtn = dm1 * v.r;
            const float dar = dr * QM(M_R) * v.rf * (1.0f - v.rf) * QM(M_AR);
            const float daz = dz * QM(M_Z) * v.zf * (1.0f - v.zf) * QM(M_AZ);
            if constexpr (NW) {
                G.dbhn += dhtn;
                G.tih[0] = mfma4(dar, v.fsel, G.tih[0]); G.tih[1] = mfma4(daz, v.fsel, G.tih[1]); G.tih[2] = mfma4(dan, v.fsel, G.tih[2]);
                G.thh[0] = mfma4(dar, v.hq, G.thh[0]); G.thh[1] = mfma4(daz, v.hq, G.thh[1]); G.thh[2] = mfma4(dhtn, v.hq, G.thh[2]);
            }
            if constexpr (DX) {
                // dL/dfeat_i = pass(feat_i) sum_o q_w(W_x)[g][o][i] d_g[o]  (straight-through over the x2h activation quantiser)
                const float2 xv = xs[s * kChunkPad + tloc + i];
                float f[4], df[4];
                qgru_feat<AMP1>(xv, f);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    df[j] = row_sum16(w.wx[0][j] * dar + w.wx[1][j] * daz + w.wx[2][j] * dan) * qpass(f[j], w.qxa);
                float dI, dQ;
                feat_bwd<AMP1 ? FEAT_A4 : FEAT_Q4>(xv.x, xv.y, df, dI, dQ);
                if (col == 0) dxs[s * kChunkPad + tloc + i] = make_float2(dI, dQ);
            }
            float d0 = 0.f, d1 = 0.f, d2 = 0.f;
            rotdot3x(d0, d1, d2, whT[0], whT[1], whT[2], dar, daz, dhtn);
            dh = dm2 * v.z + ((d0 + d1) + d2) * QM(M_PH);
#undef QM
        }
    }
}

__device__ __forceinline__ void qgru_write_partials(float* prow, const float* pl, const QgruLayout& L, QgruGrad& G, int lane,
                                                    int col, int bits_w) {
    const int H = L.H, o = col, seq = lane >> 4, g4 = lane >> 4, c = lane & 15;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
    // the scale parameters: exact zero gradient
    if (lane == 0) {
        const int sc[13] = {L.o_sxw, L.o_sxa, L.o_sxo, L.o_shw, L.o_sha, L.o_sho, L.o_ssig, L.o_stanh, L.o_sadd, L.o_smul,
                            L.o_sow, L.o_soa, L.o_soo};
#pragma unroll
        for (int k = 0; k < 13; ++k) prow[sc[k]] = 0.0f;
    }
    const Quant qxw = make_quant(pl[L.o_sxw], bits_w), qhw = make_quant(pl[L.o_shw], bits_w), qow = make_quant(pl[L.o_sow], bits_w);
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * g4 + rr;
            if (i < H) {
                if (c < 4) { const int k = L.o_wx + (g * H + i) * 4 + c; prow[k] = G.tih[g][rr] * qpass(pl[k], qxw); }
                else if (c == 4) { prow[L.o_bx + g * H + i] = G.tih[g][rr]; if (g < 2) prow[L.o_bh + g * H + i] = G.tih[g][rr]; }
                if (c < H) { const int k = L.o_wh + (g * H + i) * H + c; prow[k] = G.thh[g][rr] * qpass(pl[k], qhw); }
            }
        }
    const float bn = across_seqs<1>(G.dbhn), w0 = across_seqs<1>(G.dwo[0]), w1 = across_seqs<1>(G.dwo[1]);
    const float b0 = across_seqs<1>(G.dbo[0]), b1 = across_seqs<1>(G.dbo[1]);
    if (seq == 0 && o < H) {
        prow[L.o_bh + 2 * H + o] = bn;
        prow[L.o_wo + o] = w0 * qpass(pl[L.o_wo + o], qow);
        prow[L.o_wo + H + o] = w1 * qpass(pl[L.o_wo + H + o], qow);
    }
    if (lane == 0) { prow[L.o_bo] = b0; prow[L.o_bo + 1] = b1; }
}

template <bool AMP1, bool LUT, bool NW, bool DX>
// weight gradients OR dL/dx alone fit 256 registers: two four-wave workgroups per CU (two waves per SIMD); both together: one
__global__ __launch_bounds__(kMaxThreads / 2, (NW && DX) ? 1 : 2) void qgru_bwd_kernel(SeqArgs a, int bits_w, int bits_a) {
    constexpr int SPW = 4, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<1>();
    const int lane = id.lane, col = id.col;
    const QgruLayout L = qgru_layout(a.H);
    QgruW w; TabPtr tlane; const float* lut; int nlut; float* wb;
    qgru_prologue<true, LUT>(a, smem, L, id, w, tlane, lut, nlut, wb, 3 * 2 * SPW * kChunkPad, bits_w, bits_a);
    float2* xs = reinterpret_cast<float2*>(wb);
    float2* dys = xs + SPW * kChunkPad;
    float2* dxs = dys + SPW * kChunkPad;
    QgruGrad G;
    G.zero();
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float dh = 0.0f;
        int cur_chunk = -1;
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
                stage_in<SPW>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            const float h0 = blk ? a.ckpt[((size_t)grp * a.nck + blk) * 64 + lane] : 0.0f;
            if (nstep == S) qgru_bwd_block<AMP1, LUT, true, NW, DX>(a, w, tlane, lut, nlut, G, id, xs, dys, dxs, tb - t0, nstep, h0, dh);
            else qgru_bwd_block<AMP1, LUT, false, NW, DX>(a, w, tlane, lut, nlut, G, id, xs, dys, dxs, tb - t0, nstep, h0, dh);
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
    if constexpr (!NW) return;
    // the partial rows are built in LDS over the params/tables: keep a private copy of what write_partials reads
    const int P4 = L.P + kLossCols;
    __syncthreads();
    float* rows = smem + pad4(L.P);        // params stay at smem[0..P): only the tables/luts/staging area is reused
    qgru_write_partials(rows + id.wave * P4, smem, L, G, lane, col, bits_w);
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = rows[i];
        for (int wv = 1; wv < id.nwb; ++wv) v += rows[wv * P4 + i];
        prow[i] = v;
    }
}

static size_t qgru_lds_bytes(int P, int waves, int bits_a, bool lut, bool reduce) {
    size_t n = ((size_t)pad4(P) + kQTabFloats + (lut ? 2 * (1 << bits_a) : 0) + (size_t)waves * 3 * 2 * 4 * kChunkPad) * sizeof(float);
    const size_t need = ((size_t)pad4(P) + (size_t)waves * (P + kLossCols)) * sizeof(float);
    if (reduce && n < need) n = need;
    return n;
}
// the grid (= rows of partials) is the same for every flavour of the backward: up to two workgroups per CU
static LaunchShape qgru_bwd_shape(int ngroups) { return persistent_shape(ngroups, 8, 4); }
static bool qat_ok(const odpd_model_t* m) {
    return m->hidden <= 16 && m->bits_w >= 2 && m->bits_w <= 16 && m->bits_a >= 2 && m->bits_a <= 16;
}

template <bool AMP1, bool LUT>
static int qgru_launch_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int P) {
    const LaunchShape ls = persistent_shape(a.ngroups, 8);
    const size_t lds = qgru_lds_bytes(P, ls.waves, m->bits_a, LUT, false);
    auto k = qgru_fwd_kernel<AMP1, LUT>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a, m->bits_w, m->bits_a, (m->flags & ODPD_FLAG_EVAL) ? 1 : 0);
    return (int)hipGetLastError();
}
template <bool AMP1, bool LUT>
static int qgru_launch_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int P) {
    const LaunchShape ls = qgru_bwd_shape(a.ngroups);
    const size_t lds = qgru_lds_bytes(P, ls.waves, m->bits_a, LUT, true);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a, m->bits_w, m->bits_a);
        return (int)hipGetLastError();
    };
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return launch(qgru_bwd_kernel<AMP1, LUT, true, true>);
    if (nw) return launch(qgru_bwd_kernel<AMP1, LUT, true, false>);
    return launch(qgru_bwd_kernel<AMP1, LUT, false, true>);
}

int qgru_family_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!qat_ok(m)) return ODPD_EUNSUPPORTED;
    const int P = qgru_layout(m->hidden).P;
    const bool amp1 = m->backbone == ODPD_QGRU_AMP1, lut = m->bits_a <= 8;
    if (amp1) return lut ? qgru_launch_fwd<true, true>(st, m, a, P) : qgru_launch_fwd<true, false>(st, m, a, P);
    return lut ? qgru_launch_fwd<false, true>(st, m, a, P) : qgru_launch_fwd<false, false>(st, m, a, P);
}
int qgru_family_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!qat_ok(m)) return ODPD_EUNSUPPORTED;
    if (a.partials == nullptr && a.dx == nullptr) return ODPD_EINVAL;
    const int P = qgru_layout(m->hidden).P;
    const bool amp1 = m->backbone == ODPD_QGRU_AMP1, lut = m->bits_a <= 8;
    if (amp1) return lut ? qgru_launch_bwd<true, true>(st, m, a, P) : qgru_launch_bwd<true, false>(st, m, a, P);
    return lut ? qgru_launch_bwd<false, true>(st, m, a, P) : qgru_launch_bwd<false, false>(st, m, a, P);
}
int qgru_family_rows(const odpd_model_t* m, int B) {
    if (!qat_ok(m)) return ODPD_EUNSUPPORTED;
    return qgru_bwd_shape(num_groups(B, 1)).grid;
}
int64_t qgru_param_count(const odpd_model_t* m) { return qgru_layout(m->hidden).P; }

}  // namespace odpd
