// gru_s16x.hip — the 16-sequences-per-wave kernels of the GRU family (backbones/{gru,dgru,qgru,qgru_amp1}.py) for hidden 17 .. 24 — the
// reference's default PA size is 23 — on the bf16 MATRIX pipe.  Kernels (all on one checkpoint layout):
//   gru16x_lossdx_kernel   r05  frozen-PA step: forward + loss + dL/du (modules/train_funcs.py:33-39 behind models.py:163-176)
//   gru16x_train_kernel    r06  fused train step: forward + loss + BPTT with parameter gradients (train_funcs.py:28-48)
//   gru16x_fwd_kernel      r06  forward (y, optional checkpoints): odpd_backbone_fwd
//   gru16x_bwd_kernel      r06  backward from dL/dy: parameter gradients and / or dL/dx: odpd_backbone_bwd
// The text below describes the arithmetic and the lane mapping on the frozen-PA step, where they were introduced; the weight gradient of the
// trained kernels has its own header further down ("r06: the fused TRAIN step").
//
// Arithmetic: the matrix pipe with
// three-way operand splits ("bf16x3"): every fp32 operand v is carried as three bf16 terms v = v1 + v2 + v3 (round-to-nearest residuals,
// exact), and a product a.b is the six term products of weight 2^-16 and above (a1b3, a3b1, a2b2, a1b2, a2b1, a1b1) accumulated in
// fp32 by v_mfma_f32_16x16x32_bf16: what is dropped is below 2^-23 |a||b| — the size of one fp32 rounding of the product.
//
// Why (profiles/r05/frozen_pa.md): the exact-fp32 v_mfma_f32_16x16x4_f32 of gru_s16n.hip issues on the VALU lanes at the vector
// rate (32 cycles per 16x16x4 tile, nothing overlaps with it) and needs 16-unit M tiles: hidden 23 ran 174 of them per 16-sequence step
// (5 568 issue cycles of a ~9 000-cycle step, 2.2x the algorithmic flops).  The bf16 instruction does a 16 x 16 x 32 tile in 16 cycles on
// the matrix pipe, BESIDE the VALU, and its K = 32 is exactly [24 hidden units | 8 feature slots]: one instruction per M tile and term
// product.  A frozen model has no weight gradient, so every A operand is a constant that is split once per launch; only the B operands
// (h, the features, d(hid), d(gates)) are split at run time, 5.5 VALU instructions per value.
//
// Lane mapping: a wave holds 16 sequences, lane l = (n = l & 15 sequence, q = l >> 4); the lane owns U = 6 hidden units 6q .. 6q+5
// (units >= H are padding whose weights are zero) and TWO of the eight feature slots [feat_0 .. feat_{F-1}, 1, 0 ..] (slots 2q, 2q+1;
// the constant-1 slot carries every bias).  The instruction's operand layout (lane l holds A[l & 15][8 (l >> 4) + i], B[8 (l >> 4) + i]
// [l & 15], D[4 (l >> 4) + r][l & 15]) then reads the lane's OWN eight values [h_0 .. h_5, f_0, f_1] as its B operand — no cross-lane
// movement — and delivers four result rows per M tile to the lane, so the M rows are permuted at table-build time such that quad q
// receives the gates of ITS units: the 4 x 6 gate slots (r, z, W_hn h, W_in x) of a quad fill SIX tiles exactly (the exact-fp32 kernel
// pads 3 x 23 rows to 3 x 32); relu(fc_hid h) of the previous step rides along as two more tiles on the same B operand.
// The backward product packs K the same way: the lane's 24 values [d r_pre | d z_pre | d(W_hn h) | d n_pre] are three K = 32 operands;
// the two output tiles hold dL/dh(t-1) (six slots per quad) and, in the two free slots per quad, the feature gradient that dL/du needs.
//
// BPTT: h checkpoints every S steps in the HBM workspace ([16-sequence task][checkpoint][float4 x 64 | float2 x 64]), block recompute into
// registers.  Two waves per SIMD (256 registers).  Parity: tests/test_gru_s16x_gpu.py (against the oracle and against gru_s16n.hip).
//
// Build: MFMA results in VGPRs wherever the register allocator can afford it.  The train kernel needs more than 256 registers, and the
// compiler's default for a function that touches AGPRs at all is the AGPR form for EVERY matrix instruction: each of the ~350 results per
// two-step block that a vector instruction consumes (gates, transposed operands) then costs a v_accvgpr_read — 17 % of the block's vector
// instructions.  With the VGPR form the accumulators of the weight gradient live in AGPRs through copies the allocator places itself
// (1 966 -> 1 830 instructions per block, no scratch); the frozen kernel (no AGPRs) compiles to the same code either way.
// odpd-build-flags: -mllvm -amdgpu-mfma-vgpr-form
#include "odpd_x3.h"

namespace odpd {

// acc[t] += A(group grp0 + 3 t .. + 2: the three terms of tile t) . B, smallest products first
// PF (the lone-wave train kernel): the next tile's three operands are requested before the current tile's six products are issued — a wave
// that is alone on its SIMD has nobody to cover the LDS latency of a load issued where it is needed
template <int NT, bool PF = false>
__device__ __forceinline__ void mm6(TabPtr tl, int grp0, const Split3& B, f32x4 (&acc)[NT]) {
    if constexpr (PF) {
        u32x4 a1 = tabx_ld(tl, grp0 * 64), a2 = tabx_ld(tl, (grp0 + 1) * 64), a3 = tabx_ld(tl, (grp0 + 2) * 64);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            u32x4 n1 = a1, n2 = a2, n3 = a3;
            if (t + 1 < NT) { n1 = tabx_ld(tl, (grp0 + 3 * t + 3) * 64); n2 = tabx_ld(tl, (grp0 + 3 * t + 4) * 64); n3 = tabx_ld(tl, (grp0 + 3 * t + 5) * 64); }
            f32x4 c = acc[t];
            c = mfma32(a1, B.t[2], c);
            c = mfma32(a3, B.t[0], c);
            c = mfma32(a2, B.t[1], c);
            c = mfma32(a1, B.t[1], c);
            c = mfma32(a2, B.t[0], c);
            c = mfma32(a1, B.t[0], c);
            acc[t] = c;
            a1 = n1; a2 = n2; a3 = n3;
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const u32x4 a1 = tabx_ld(tl, (grp0 + 3 * t) * 64), a2 = tabx_ld(tl, (grp0 + 3 * t + 1) * 64), a3 = tabx_ld(tl, (grp0 + 3 * t + 2) * 64);
        f32x4 c = acc[t];
        c = mfma32(a1, B.t[2], c);
        c = mfma32(a3, B.t[0], c);
        c = mfma32(a2, B.t[1], c);
        c = mfma32(a1, B.t[1], c);
        c = mfma32(a2, B.t[0], c);
        c = mfma32(a1, B.t[0], c);
        acc[t] = c;
    }
}

// The same with the A operands already in registers (forward loop: the six cell tiles stay pinned — a ds_read_b128 costs the SIMD ~16
// cycles that nothing overlaps, tools/ubench/mfma_lds.hip), two tiles at a time with their products interleaved
template <int NT>
__device__ __forceinline__ void mm6r(const u32x4 (&A)[NT][3], const Split3& B, f32x4 (&acc)[NT]) {
    static_assert(NT % 2 == 0, "tiles go in pairs");
#pragma unroll
    for (int t = 0; t < NT; t += 2) {
        f32x4 c = acc[t], d = acc[t + 1];
        c = mfma32(A[t][0], B.t[2], c); d = mfma32(A[t + 1][0], B.t[2], d);
        c = mfma32(A[t][2], B.t[0], c); d = mfma32(A[t + 1][2], B.t[0], d);
        c = mfma32(A[t][1], B.t[1], c); d = mfma32(A[t + 1][1], B.t[1], d);
        c = mfma32(A[t][0], B.t[1], c); d = mfma32(A[t + 1][0], B.t[1], d);
        c = mfma32(A[t][1], B.t[0], c); d = mfma32(A[t + 1][1], B.t[0], d);
        c = mfma32(A[t][0], B.t[0], c); d = mfma32(A[t + 1][0], B.t[0], d);
        acc[t] = c; acc[t + 1] = d;
    }
}

// ---- operand tables ------------------------------------------------------------------------------------------------------------------
// groups of 64 lanes x 16 B.  bf16 groups hold A[row m = lane & 15][k = 8 (lane >> 4) + i], i < 8, as one term of the split.
template <bool DG, int U>
struct S16X {
    static_assert(U == 6, "tile bookkeeping below is written for six units per lane");
    static constexpr int NFS = 8 - U;                   // feature slots per lane
    static constexpr int NTF = U;                       // cell tiles: slots [r | z | W_hn h | W_in x] x U per quad = 4 U rows per quad
    static constexpr int NTH = DG ? (U + 3) / 4 : 0;    // fc_hid tiles (slot j of the quad = unit U q + j)
    static constexpr int NTO = (U + NFS + 3) / 4;       // backward output tiles: U slots dL/dh(t-1), then NFS slots feature gradient
    static constexpr int NM = U / 2;                    // K = 32 operands of the backward product (4 U values per lane)
    static constexpr int FW = 0;                        // 3 (tile) + term, tiles [cell | fc_hid]
    static constexpr int HT = FW + 3 * (NTF + NTH);     // 3 (tile) + term: fc_hid^T
    static constexpr int BW = HT + 3 * NTH;             // 3 (tile * NM + c) + term
    static constexpr int VW = BW + 3 * NTO * NM;        // fp32 groups: fc_out weights of the lane's units, then of its feature slots
    static constexpr int NVW = (2 * U + 3) / 4 + 1;
    static constexpr int NG = VW + NVW;
    static constexpr int NQ = (U + 3) / 4;              // float4 per lane and checkpoint
};

// value of one A element.  kind 0: cell / fc_hid tiles (grp = tile), 1: fc_hid^T, 2: backward (grp = tile * NM + c)
template <int FM, bool DG, int U>
__device__ __forceinline__ float s16x_weight(const float* pl, const GruLayout& L, int kind, int grp, int m, int kq, int i) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F;
    const int H = L.H, mq = m >> 2, mr = m & 3;
    if (kind == 0) {
        const bool hid = grp >= T::NTF;
        const int s = hid ? 4 * (grp - T::NTF) + mr : 4 * grp + mr;            // slot of quad mq
        const int gate = hid ? 4 : s / U, j = hid ? s : s % U;
        if (j >= U) return 0.0f;
        const int u = U * mq + j;
        if (u >= H) return 0.0f;
        const float sc = gate < 2 ? kNegLog2e : 1.0f;
        if (i < U) {                                                           // K slot = hidden unit U kq + i
            const int k = U * kq + i;
            if (k >= H || gate == 3) return 0.0f;
            if (gate == 4) return pl[L.o_w_hid + u * H + k];
            return sc * pl[L.o_w_hh + (gate * H + u) * H + k];
        }
        const int fsl = T::NFS * kq + (i - U);                                  // K slot = feature slot
        if (fsl > F) return 0.0f;
        if (gate == 4) return fsl == F ? pl[L.o_b_hid + u] : 0.0f;
        if (gate == 2) return fsl == F ? pl[L.o_b_hh + 2 * H + u] : 0.0f;
        const int g = gate == 3 ? 2 : gate;
        if (fsl < F) return sc * pl[L.o_w_ih + (g * H + u) * F + fsl];
        return gate < 2 ? sc * (pl[L.o_b_ih + g * H + u] + pl[L.o_b_hh + g * H + u]) : pl[L.o_b_ih + 2 * H + u];
    }
    if (kind == 1) {                                                            // dht[k_out] += sum_u fc_hid[u][k_out] dhid[u]
        const int j = 4 * grp + mr, ko = U * mq + j, u = U * kq + i;
        if (j >= U || i >= U || ko >= H || u >= H) return 0.0f;
        return pl[L.o_w_hid + u * H + ko];
    }
    const int tile = grp / T::NM, c = grp % T::NM;
    const int s = 4 * tile + mr;                                                // output slot of quad mq: < U dL/dh, then feature slots
    const int e = 8 * c + i, ge = e / U, u = U * kq + e % U;                    // K element: value e of the lane's [drp | dzp | dgh | dnp]
    if (u >= H) return 0.0f;
    if (s < U) {
        const int ko = U * mq + s;
        if (ko >= H || ge == 3) return 0.0f;
        return pl[L.o_w_hh + (ge * H + u) * H + ko];
    }
    const int fsl = T::NFS * mq + (s - U);
    if (s >= U + T::NFS || fsl >= F || ge == 2) return 0.0f;
    return pl[L.o_w_ih + ((ge == 3 ? 2 : ge) * H + u) * F + fsl];
}
template <int FM, bool DG, int U>
__device__ __forceinline__ void s16x_fill_table(float* tab, const float* pl, const GruLayout& L, int lane, int wave, int nwb) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F;
    u32x4* t4 = reinterpret_cast<u32x4*>(tab);
    const int m = lane & 15, kq = lane >> 4, H = L.H, OW = DG ? H + 6 : H;
    constexpr int nA = T::VW / 3;                                               // split operands (one per tile / tile x chunk)
    for (int op = wave; op < nA; op += nwb) {
        const int kind = 3 * op < T::HT ? 0 : (3 * op < T::BW ? 1 : 2);
        const int grp = op - (kind == 0 ? 0 : (kind == 1 ? T::HT / 3 : T::BW / 3));
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = s16x_weight<FM, DG, U>(pl, L, kind, grp, m, kq, i);
        const Split3 s = split8(v);
#pragma unroll
        for (int t = 0; t < 3; ++t) t4[(3 * op + t) * 64 + lane] = s.t[t];
    }
    // fp32 groups: [w_out[0][U q + j], j < U | w_out[1][..]] packed, then {wf[0][0], wf[0][1], wf[1][0], wf[1][1]} of the lane's feature slots
    for (int g = wave; g < T::NVW; g += nwb) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (g < T::NVW - 1) {
                const int idx = 4 * g + e, cc = idx / U, u = U * kq + idx % U;
                v[e] = (idx < 2 * U && u < H) ? pl[L.o_w_out + cc * OW + u] : 0.0f;
            } else {
                const int cc = e >> 1, fsl = T::NFS * kq + (e & 1);
                v[e] = (DG && fsl < F) ? pl[L.o_w_out + cc * OW + H + fsl] : (fsl == F ? pl[L.o_b_out + cc] : 0.0f);
            }
        }
        reinterpret_cast<float4*>(tab)[(T::VW + g) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
}

// the lane's feature slots: slot NFS q + e of [feat_0 .. feat_{F-1}, 1, 0, ..] (one-hot arithmetic, no selects)
template <int FM, int U>
__device__ __forceinline__ void s16x_feats(float I, float Q, const float (&oh)[4], float (&fs)[8 - U]) {
    constexpr int F = S16Cfg<FM>::F, NFS = 8 - U;
    float f[F];
    feat_fwd<FM>(I, Q, f);
#pragma unroll
    for (int e = 0; e < NFS; ++e) {
        float acc = 0.0f;
        bool first = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int sl = NFS * k + e;
            if (sl < F) { acc = first ? oh[k] * f[sl < F ? sl : 0] : __builtin_fmaf(oh[k], f[sl < F ? sl : 0], acc); first = false; }
            else if (sl == F) { acc = first ? oh[k] : acc + oh[k]; first = false; }
        }
        fs[e] = acc;
    }
}

__device__ __forceinline__ float sig_ps(float v) { return fast_rcp(__builtin_amdgcn_exp2f(v) + 1.0f); }      // v = -log2(e) * pre-activation
template <int FM>
__device__ __forceinline__ float tanh_x(float x) {      // tanh4 / tanh4_rel of odpd_s16.h, one element
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f;
    const float t = __builtin_fmaf(fast_rcp(e), -2.0f, 1.0f);
    if constexpr (FM == FEAT_DGRU6) return t;
    else {
        const float x2 = x * x;
        float p = __builtin_fmaf(x2, 0.021869488536155203f, -0.053968253968253971f);
        p = __builtin_fmaf(x2, p, 0.13333333333333333f);
        p = __builtin_fmaf(x2, p, -0.33333333333333333f);
        p = __builtin_fmaf(x2 * x, p, x);
        const float w = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fabsf(x), -0x1p100f, 0.3f * 0x1p100f), 0.0f, 1.0f);
        return __builtin_fmaf(w, p - t, t);
    }
}

// B operand of a cell step: [h_0 .. h_{U-1}, f_0 .. f_{NFS-1}]
template <int U>
__device__ __forceinline__ Split3 s16x_operand(const float (&h)[U], const float (&fs)[8 - U]) {
    float v[8];
#pragma unroll
    for (int i = 0; i < U; ++i) v[i] = h[i];
#pragma unroll
    for (int i = U; i < 8; ++i) v[i] = fs[i - U];
    return split8(v);
}
// gates of one step from the tile results; h <- h(t).  WITH_HID: the fc_hid tiles ride on the same operand (they are the hidden
// layer of the operand's h, i.e. of the PREVIOUS step)
template <int FM, bool DG, int U, bool WITH_HID, bool PF = false>
__device__ __forceinline__ void s16x_cell(TabPtr tl, const Split3& B, float (&h)[U], float (&r)[U], float (&z)[U], float (&n)[U], float (&nh)[U],
                                          float (&hid_prev)[U]) {
    using T = S16X<DG, U>;
    constexpr int NT = T::NTF + (WITH_HID ? T::NTH : 0);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mm6<NT, PF>(tl, T::FW, B, acc);
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int sr = j, sz = U + j, sh = 2 * U + j, si = 3 * U + j;
        r[j] = sig_ps(acc[sr / 4][sr % 4]);
        z[j] = sig_ps(acc[sz / 4][sz % 4]);
        nh[j] = acc[sh / 4][sh % 4];
        n[j] = tanh_x<FM>(__builtin_fmaf(r[j], nh[j], acc[si / 4][si % 4]));
        h[j] = __builtin_fmaf(z[j], h[j] - n[j], n[j]);
        if constexpr (WITH_HID) hid_prev[j] = acc[T::NTF + j / 4][j % 4];
    }
}
template <bool DG, int U, bool PF = false>
__device__ __forceinline__ void s16x_hid(TabPtr tl, const Split3& B, float (&hid)[U]) {
    using T = S16X<DG, U>;
    f32x4 acc[T::NTH];
#pragma unroll
    for (int t = 0; t < T::NTH; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mm6<T::NTH, PF>(tl, T::FW + 3 * T::NTF, B, acc);
#pragma unroll
    for (int j = 0; j < U; ++j) hid[j] = acc[j / 4][j % 4];
}

// one block of <= S steps of the backward pass: recompute from the checkpoint h0, then loss, dL/dy and BPTT down to dL/du
// FUSED = false (r06, the split backward of a model without parameter gradients): `ts` holds dL/dy instead of the target, no y, no loss
template <int FM, bool DG, int U, int S, bool FULL, bool FUSED = true>
__device__ __forceinline__ void s16x_block(const SeqArgs& a, TabPtr tl0, const float (&oh)[4], const float2* xs, const float2* ts, float2* dxs,
                                           int n, int q, int tloc, int nstep, bool valid, const float (&h0)[U], float (&dh)[U], float& loss_acc) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F, NFS = T::NFS;
    float h[U], hp_s[S][U], r_s[S][U], z_s[S][U], n_s[S][U], nh_s[S][U], hid_s[S][U], fs_s[S][NFS];
#pragma unroll
    for (int j = 0; j < U; ++j) h[j] = h0[j];
    TabPtr tl = opaque(tl0);
#pragma unroll
    for (int st = 0; st < S; ++st) {
        if (FULL || st < nstep) {
            const float2 xv = xs[n * kChunkPad + tloc + st];
            s16x_feats<FM, U>(xv.x, xv.y, oh, fs_s[st]);
#pragma unroll
            for (int j = 0; j < U; ++j) hp_s[st][j] = h[j];
            const Split3 B = s16x_operand<U>(h, fs_s[st]);
            if constexpr (DG && FULL) {
                if (st > 0) s16x_cell<FM, DG, U, true>(tl, B, h, r_s[st], z_s[st], n_s[st], nh_s[st], hid_s[st > 0 ? st - 1 : 0]);
                else s16x_cell<FM, DG, U, false>(tl, B, h, r_s[st], z_s[st], n_s[st], nh_s[st], hid_s[0]);
            } else {
                s16x_cell<FM, DG, U, false>(tl, B, h, r_s[st], z_s[st], n_s[st], nh_s[st], hid_s[st]);
                if constexpr (DG) {      // (ragged tail block: the hidden layer of every step on an operand of its own)
                    const Split3 B2 = s16x_operand<U>(h, fs_s[st]);
                    s16x_hid<DG, U>(tl, B2, hid_s[st]);
                }
            }
        }
    }
    if constexpr (DG && FULL) {
        const Split3 B2 = s16x_operand<U>(h, fs_s[S - 1]);
        s16x_hid<DG, U>(tl, B2, hid_s[S - 1]);
    }
    tl = opaque(tl0);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, valid ? a.inv_count : 0.0f, valid && q == 0);
#pragma unroll
    for (int st = S - 1; st >= 0; --st) {
        if (FULL || st < nstep) {
            // ---- head: y, loss, dL/dy ----
            float w0[U], w1[U];
            {
                float wv[4 * (T::NVW - 1)];
#pragma unroll
                for (int g = 0; g < T::NVW - 1; ++g) {
                    const float4 v = tab_ld(tl, (T::VW + g) * 64);
                    wv[4 * g] = v.x; wv[4 * g + 1] = v.y; wv[4 * g + 2] = v.z; wv[4 * g + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < U; ++j) { w0[j] = wv[j]; w1[j] = wv[U + j]; }
            }
            const float4 wf = tab_ld(tl, (T::VW + T::NVW - 1) * 64);
            float act[U];
            if constexpr (DG) {
#pragma unroll
                for (int j = 0; j < U; ++j) act[j] = relu_(hid_s[st][j]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) act[j] = __builtin_fmaf(z_s[st][j], hp_s[st][j] - n_s[st][j], n_s[st][j]);      // h(t)
            }
            const float2 tv = ts[n * kChunkPad + tloc + st];
            float dy0 = tv.x, dy1 = tv.y;
            if constexpr (FUSED) {
                float p0 = wf.x * fs_s[st][0], p1 = wf.z * fs_s[st][0];
                if constexpr (NFS > 1) { p0 = __builtin_fmaf(wf.y, fs_s[st][1], p0); p1 = __builtin_fmaf(wf.w, fs_s[st][1], p1); }
#pragma unroll
                for (int j = 0; j < U; ++j) { p0 = __builtin_fmaf(w0[j], act[j], p0); p1 = __builtin_fmaf(w1[j], act[j], p1); }
                const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
            }
            // ---- dL/dh(t) ----
            float dht[U];
            if constexpr (DG) {
                float v[8];
#pragma unroll
                for (int j = 0; j < U; ++j) v[j] = __builtin_fmaf(dy0, w0[j], w1[j] * dy1) * relu_gate(hid_s[st][j]);
#pragma unroll
                for (int j = U; j < 8; ++j) v[j] = 0.0f;
                const Split3 Bd = split8(v);
                f32x4 acc[T::NTH];
#pragma unroll
                for (int t = 0; t < T::NTH; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] = 4 * t + e < U ? dh[4 * t + e] : 0.0f;
                mm6<T::NTH>(tl, T::HT, Bd, acc);
#pragma unroll
                for (int j = 0; j < U; ++j) dht[j] = acc[j / 4][j % 4];
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) dht[j] = dh[j] + __builtin_fmaf(dy0, w0[j], w1[j] * dy1);
            }
            // ---- gate derivatives: v = [d r_pre | d z_pre | d(W_hn h) | d n_pre] ----
            float v[4 * U];
            f32x4 acc[T::NTO];
#pragma unroll
            for (int t = 0; t < T::NTO; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const float r = r_s[st][j], z = z_s[st][j], nn = n_s[st][j];
                const float dn = dht[j] * (1.0f - z);
                const float dnp = dn * __builtin_fmaf(-nn, nn, 1.0f);
                const float dgh = dnp * r;
                v[j] = dgh * nh_s[st][j] * (1.0f - r);
                v[U + j] = (hp_s[st][j] - nn) * z * dn;
                v[2 * U + j] = dgh;
                v[3 * U + j] = dnp;
                acc[j / 4][j % 4] = dht[j] * z;
            }
#pragma unroll
            for (int c = 0; c < T::NM; ++c) {
                float vc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vc[i] = v[8 * c + i];
                const Split3 Bg = split8(vc);
                // tile t of chunk c: groups BW + 3 (t * NM + c)
#pragma unroll
                for (int t = 0; t < T::NTO; ++t) {
                    f32x4 one[1] = {acc[t]};
                    mm6<1>(tl, T::BW + 3 * (t * T::NM + c), Bg, one);
                    acc[t] = one[0];
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) dh[j] = acc[j / 4][j % 4];
            // ---- dL/du: feature gradient of the lane's slots, gathered over the quads ----
            float dfs[NFS];
#pragma unroll
            for (int e = 0; e < NFS; ++e) {
                dfs[e] = acc[(U + e) / 4][(U + e) % 4];
                if constexpr (DG) dfs[e] = __builtin_fmaf(dy0, e ? wf.y : wf.x, __builtin_fmaf(dy1, e ? wf.w : wf.z, dfs[e]));
            }
            float df[F];
            {
                float g0[4], g1[4];
                gather_rows(dfs[0], g0);
                if constexpr (F > 1 && NFS > 1) gather_rows(dfs[1], g1);
#pragma unroll
                for (int f = 0; f < F; ++f) df[f] = (f % NFS) ? g1[f / NFS] : g0[f / NFS];
            }
            const float2 xv = xs[n * kChunkPad + tloc + st];
            float dI, dQ;
            feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
            if (q == 0) dxs[n * kChunkPad + tloc + st] = make_float2(dI, dQ);
        }
    }
}

constexpr int kS16xCkptFloats = 384;      // floats per (16-sequence task, checkpoint): 64 lanes x (float4 + float2)
// ---- forward pass of one 16-sequence task: h checkpoints only; the cell tiles stay in registers ----
template <int FM, bool DG, int U, int S, bool BATCHED = false>
__device__ __forceinline__ void s16x_forward_pass(const SeqArgs& a, TabPtr tl, const float (&oh)[4], float2* xs, float4* ck4, float2* ck2, int b0, int n,
                                                  int lane) {
    using T = S16X<DG, U>;
    constexpr int NFS = T::NFS;
    u32x4 A[T::NTF][3];
    {
        TabPtr tp = opaque(tl);
#pragma unroll
        for (int t = 0; t < T::NTF; ++t)
#pragma unroll
            for (int k = 0; k < 3; ++k) A[t][k] = tabx_ld(tp, (T::FW + 3 * t + k) * 64);
    }
    float h[U];
#pragma unroll
    for (int j = 0; j < U; ++j) h[j] = 0.0f;
    for (int t0 = 0; t0 < a.T; t0 += kChunk) {
        const int len = min(kChunk, a.T - t0);
        wave_lds_fence();
        stage_in<16, BATCHED>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
        wave_lds_fence();
        for (int tt = 0; tt < len; ++tt) {
            const float2 xv = xs[n * kChunkPad + tt];
            float fs[NFS];
            s16x_feats<FM, U>(xv.x, xv.y, oh, fs);
            const Split3 B = s16x_operand<U>(h, fs);
            f32x4 acc[T::NTF];
#pragma unroll
            for (int t = 0; t < T::NTF; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            mm6r<T::NTF>(A, B, acc);
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int sr = j, sz = U + j, sh = 2 * U + j, si = 3 * U + j;
                const float r = sig_ps(acc[sr / 4][sr % 4]), z = sig_ps(acc[sz / 4][sz % 4]);
                const float nn = tanh_x<FM>(__builtin_fmaf(r, acc[sh / 4][sh % 4], acc[si / 4][si % 4]));
                h[j] = __builtin_fmaf(z, h[j] - nn, nn);
            }
            const int t1 = t0 + tt + 1;
            if ((t1 % S) == 0 && t1 < a.T) {
                const size_t o = (size_t)(t1 / S) * kS16xCkptFloats;
                ck4[o / 4] = make_float4(h[0], h[1], h[2], h[3]);
                ck2[o / 2] = make_float2(h[4], h[5]);
            }
        }
    }
}

template <int FM, bool DG, int U, int S>
__global__ __launch_bounds__(512, 1) void gru16x_lossdx_kernel(SeqArgs a) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F, NFS = T::NFS, NQ = T::NQ;
    constexpr int kWave = 3 * 2 * 16 * kChunkPad;
    static_assert(kChunk % S == 0, "a block never straddles two staged chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* tab = smem;
    float* pl = tab + s16_tab_floats(T::NG);
    stage_params(pl, a.params, L.P);
    s16x_fill_table<FM, DG, U>(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + 16 * kChunkPad;
    float2* dxs = ts + 16 * kChunkPad;
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * nwb, nblk = (a.T + S - 1) / S;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        // checkpoints: [block][float4 x 64 lanes (units 0..3) | float2 x 64 lanes (units 4, 5)] = 384 floats (r05 stored two float4: a
        // quarter of the checkpoint traffic was padding)
        static_assert(U == 6 && NQ == 2, "checkpoint layout below: six units per lane = one float4 + one float2");
        float* ckg = a.ckpt + (size_t)grp * nblk * kS16xCkptFloats;
        float4* ck4 = reinterpret_cast<float4*>(ckg) + lane;
        float2* ck2 = reinterpret_cast<float2*>(ckg + 256) + lane;
        s16x_forward_pass<FM, DG, U, S>(a, tl, oh, xs, ck4, ck2, b0, n, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        {
            // ---- backward ----
            float dh[U];
#pragma unroll
            for (int j = 0; j < U; ++j) dh[j] = 0.0f;
            int cur_chunk = -1;
            for (int blk = nblk - 1; blk >= 0; --blk) {
                const int tb = blk * S, nstep = min(S, a.T - tb);
                const int chunk = tb / kChunk, t0 = chunk * kChunk;
                float h0[U];
                {
                    const size_t o = (size_t)blk * kS16xCkptFloats;
                    const float4 v = blk ? ck4[o / 4] : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float2 w = blk ? ck2[o / 2] : make_float2(0.f, 0.f);
                    h0[0] = v.x; h0[1] = v.y; h0[2] = v.z; h0[3] = v.w; h0[4] = w.x; h0[5] = w.y;
                }
                if (chunk != cur_chunk) {
                    if (cur_chunk >= 0) {
                        const int pt0 = cur_chunk * kChunk;
                        wave_lds_fence();
                        stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    }
                    wave_lds_fence();
                    const int len = min(kChunk, a.T - t0);
                    stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                    stage_in<16>(ts, a.target, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                    wave_lds_fence();
                    cur_chunk = chunk;
                }
                if (nstep == S) s16x_block<FM, DG, U, S, true>(a, tl, oh, xs, ts, dxs, n, q, tb - t0, nstep, valid, h0, dh, loss_acc);
                else s16x_block<FM, DG, U, S, false>(a, tl, oh, xs, ts, dxs, n, q, tb - t0, nstep, valid, h0, dh, loss_acc);
            }
            if (cur_chunk >= 0) {
                const int pt0 = cur_chunk * kChunk;
                wave_lds_fence();
                stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                wave_lds_fence();
            }
        }
    }
    // the loss partial of the workgroup (one row of kLossCols per workgroup)
    const float lp = row_sum16(loss_acc);          // accumulated on the q == 0 lanes
    __syncthreads();
    if (lane == 0) smem[wave] = lp;
    __syncthreads();
    if (threadIdx.x < kLossCols) {
        float v = 0.0f;
        if (threadIdx.x == 0)
            for (int wv = 0; wv < nwb; ++wv) v += smem[wv];
        a.partials[(size_t)blockIdx.x * kLossCols + threadIdx.x] = v;
    }
}

// =====================================================================================================================================
// r06: the fused TRAIN step (forward + loss + BPTT with weight gradients: train_pa of the 17 .. 24-unit GRU family, e.g. the DGRU H23 PA
// every train_dpd run of the reference trains first — bash_scripts/OpenDPDv2.sh:39-52, backbones/dgru.py:59-74) on the same pipe.
//
// Forward, recompute, dL/dh chain: exactly the frozen kernel's (the weights are constants within a launch: split once per launch, i.e.
// once per optimiser step).  New: the WEIGHT GRADIENT.  The forward table [6 cell tiles | 2 fc_hid tiles] x [K = 24 hidden + 8 feature
// slots] is one matrix, so its gradient is one GEMM over (sequence, time):
//      d tab[tile T][row 4 q + r][k = 8 kq + i]  =  sum_{n, t}  G_{4T + r}(n, q, t) . V_i(n, kq, t)
// with G = [d r_pre | d z_pre | d(W_hn h) | d n_pre] (the lane's 24 gate derivatives, the B operand of the dL/dh product) and V = [h(t-1) |
// feature slots] (the lane's B operand of the cell) — every W_ih / W_hh / bias gradient is an entry of it (bias = the constant-1 slot), and
// fc_hid's rows ride as two more M tiles on the operand [h(t) | .] one step later, as in the forward.  On the matrix pipe the contraction index
// must sit on K = (quad, element) while the chain holds it on lane & 15 (the sequence), so both operands are TRANSPOSED first — by the matrix
// pipe itself: with the data as the A operand (A[m = sequence][k = the lane's eight values]) and a 0 / 1 selection matrix as B, D[sequence]
// [row] comes back on lane (row, Q) with the four sequences 4 Q .. 4 Q + 3 in its registers: the layout of an A / B operand of the
// contraction over sequences.  A term of the three-way split is a bf16, 1.0 x it summed with zeros in fp32 is exact, and it converts back
// to bf16 exactly: the transposition is lossless and costs no LDS traffic (a ds_read_b128 is 16 SIMD cycles nothing overlaps;
// profiles/r05/frozen_pa.md) and no LDS capacity (the split G of one 16-sequence step alone is 11.5 KB).  K = 32 = 16 sequences x the
// block's two time steps.  Both operands are run-time values: six term products per tile pair, smallest first, fp32 accumulation.
// Per 16-sequence step: 33 transposing + 48 contraction MFMAs on top of the chain's 138; accumulators: 16 tiles = 64 registers -> one
// wave per SIMD (four waves per workgroup, like gru16n_kernel's train flavour).
// =====================================================================================================================================
struct S16XSel { u32x4 g[2], v[2]; };
// g[h]: rows 4 q + r of an M tile <- elements 4 h + r of quad q;   v[j]: columns 8 (kq - 2 j) + i of N tile j <- element i of quad kq
__device__ __forceinline__ S16XSel s16x_sel(int lane) {
    const int nn = lane & 15, kq = lane >> 4;
    S16XSel e;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int pos = 4 * h + (nn & 3);
        const bool on = kq == (nn >> 2);
#pragma unroll
        for (int w = 0; w < 4; ++w) e.g[h][w] = (on && (pos >> 1) == w) ? (0x3F80u << (16 * (pos & 1))) : 0u;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pos = nn & 7;
        const bool on = kq == 2 * j + (nn >> 3);
#pragma unroll
        for (int w = 0; w < 4; ++w) e.v[j][w] = (on && (pos >> 1) == w) ? (0x3F80u << (16 * (pos & 1))) : 0u;
    }
    return e;
}
// one transposing product: the lane's eight bf16 (one term of a split) -> two packed words = the selected row's four sequences 4 Q .. 4 Q + 3
__device__ __forceinline__ void s16x_tr(const u32x4& x, const u32x4& sel, unsigned (&w)[2]) {
    const f32x4 d = mfma32(x, sel, f32x4{0.f, 0.f, 0.f, 0.f});
    w[0] = pk_bf16(d[0], d[1]);      // (exact: every d is a bf16 value)
    w[1] = pk_bf16(d[2], d[3]);
}
// the lane's operand [h | features] of one step as the contraction's B operand: [N tile][term][word]
__device__ __forceinline__ void s16x_tr_v(const Split3& B, const S16XSel& E, unsigned (&out)[2][3][2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) s16x_tr(B.t[p], E.v[j], out[j][p]);
}
// acc += A . B with both operands split (six term products of weight >= 2^-16, smallest first)
__device__ __forceinline__ f32x4 mm6w(const u32x4 (&A)[3], const u32x4 (&B)[3], f32x4 c) {
    c = mfma32(A[0], B[2], c);
    c = mfma32(A[2], B[0], c);
    c = mfma32(A[1], B[1], c);
    c = mfma32(A[0], B[1], c);
    c = mfma32(A[1], B[0], c);
    c = mfma32(A[0], B[0], c);
    return c;
}
template <bool DG, int U>
struct S16XGrad {
    f32x4 cell[S16X<DG, U>::NTF][2];                  // [M tile][N tile]: lane (nn, Q) register R = d tab[tile][row 4 Q + R][k = 16 j + nn]
    f32x4 hid[DG ? S16X<DG, U>::NTH : 1][2];
    float dwo[2][U], dwf[2][8 - U];                   // fc_out: the lane's units / its feature slots (bias = the constant-1 slot)
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < S16X<DG, U>::NTF; ++t) { cell[t][0] = z4; cell[t][1] = z4; }
#pragma unroll
        for (int t = 0; t < (DG ? S16X<DG, U>::NTH : 1); ++t) { hid[t][0] = z4; hid[t][1] = z4; }
#pragma unroll
        for (int j = 0; j < U; ++j) dwo[0][j] = dwo[1][j] = 0.0f;
#pragma unroll
        for (int e = 0; e < 8 - U; ++e) dwf[0][e] = dwf[1][e] = 0.0f;
    }
};

// one block of <= S = 2 steps of the train step's backward pass (s16x_block without dL/du, with the weight gradient)
// FUSED = false (the split backward): `ts` holds dL/dy; DX: dL/dx as well (the frozen block's feature-gradient path), written to `dxs`
template <int FM, bool DG, int U, int S, bool FULL, bool FUSED = true, bool DX = false>
__device__ __forceinline__ void s16x_train_block(const SeqArgs& a, TabPtr tl0, const float (&oh)[4], const S16XSel& E, S16XGrad<DG, U>& G,
                                                 const float2* xs, const float2* ts, int n, int q, int tloc, int nstep, bool valid,
                                                 const float (&h0)[U], float (&dh)[U], float& loss_acc, float2* dxs = nullptr) {
    using T = S16X<DG, U>;
    constexpr int NFS = T::NFS, F = S16Cfg<FM>::F;
    static_assert(S == 2, "K = 32 of the weight-gradient contraction = 16 sequences x the block's two steps");
    float h[U], hp_s[S][U], r_s[S][U], z_s[S][U], n_s[S][U], nh_s[S][U], hid_s[S][U], fs_s[S][NFS];
    unsigned vc[S][2][3][2], vh[S][2][3][2];          // transposed operands: [step][N tile][term][word] of the cell's / fc_hid's V
    unsigned gt[S][T::NTF][3][2], ht[S][DG ? T::NTH : 1][3][2];      // transposed gate / hidden-layer derivatives: [step][M tile][term][word]
    if constexpr (!FULL) {      // steps the ragged tail block does not run contribute 0 x 0 (never 0 x an uninitialised NaN pattern)
#pragma unroll
        for (int st = 0; st < S; ++st)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int w = 0; w < 2; ++w) { vc[st][j][p][w] = 0u; vh[st][j][p][w] = 0u; }
#pragma unroll
        for (int st = 0; st < S; ++st)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
#pragma unroll
                    for (int t = 0; t < T::NTF; ++t) gt[st][t][p][w] = 0u;
#pragma unroll
                    for (int t = 0; t < (DG ? T::NTH : 1); ++t) ht[st][t][p][w] = 0u;
                }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) h[j] = h0[j];
    TabPtr tl = opaque(tl0);
#pragma unroll
    for (int st = 0; st < S; ++st) {
        if (FULL || st < nstep) {
            const float2 xv = xs[n * kChunkPad + tloc + st];
            s16x_feats<FM, U>(xv.x, xv.y, oh, fs_s[st]);
#pragma unroll
            for (int j = 0; j < U; ++j) hp_s[st][j] = h[j];
            const Split3 B = s16x_operand<U>(h, fs_s[st]);
            s16x_tr_v(B, E, vc[st]);
            if constexpr (DG && FULL) {
                if (st > 0) s16x_cell<FM, DG, U, true, true>(tl, B, h, r_s[st], z_s[st], n_s[st], nh_s[st], hid_s[st > 0 ? st - 1 : 0]);
                else s16x_cell<FM, DG, U, false, true>(tl, B, h, r_s[st], z_s[st], n_s[st], nh_s[st], hid_s[0]);
            } else {
                s16x_cell<FM, DG, U, false, true>(tl, B, h, r_s[st], z_s[st], n_s[st], nh_s[st], hid_s[st]);
                if constexpr (DG) {      // (ragged tail block: the hidden layer of every step on an operand of its own)
                    const Split3 B2 = s16x_operand<U>(h, fs_s[st]);
                    s16x_hid<DG, U, true>(tl, B2, hid_s[st]);
                    s16x_tr_v(B2, E, vh[st]);
                }
            }
        }
    }
    if constexpr (DG && FULL) {
        const Split3 B2 = s16x_operand<U>(h, fs_s[S - 1]);
        s16x_hid<DG, U, true>(tl, B2, hid_s[S - 1]);
        s16x_tr_v(B2, E, vh[S - 1]);
        // fc_hid of step st rode on the cell operand of step st + 1: [h(st) | features of st + 1] (the constant-1 slot is what its bias needs)
#pragma unroll
        for (int st = 0; st + 1 < S; ++st)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) { vh[st][j][p][0] = vc[st + 1][j][p][0]; vh[st][j][p][1] = vc[st + 1][j][p][1]; }
    }
    tl = opaque(tl0);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, valid ? a.inv_count : 0.0f, valid && q == 0);
#pragma unroll
    for (int st = S - 1; st >= 0; --st) {
        if (FULL || st < nstep) {
            // ---- head: y, loss, dL/dy, fc_out's gradient ----
            float w0[U], w1[U];
            {
                float wv[4 * (T::NVW - 1)];
#pragma unroll
                for (int g = 0; g < T::NVW - 1; ++g) {
                    const float4 v = tab_ld(tl, (T::VW + g) * 64);
                    wv[4 * g] = v.x; wv[4 * g + 1] = v.y; wv[4 * g + 2] = v.z; wv[4 * g + 3] = v.w;
                }
#pragma unroll
                for (int j = 0; j < U; ++j) { w0[j] = wv[j]; w1[j] = wv[U + j]; }
            }
            const float4 wf = tab_ld(tl, (T::VW + T::NVW - 1) * 64);
            float act[U];
            if constexpr (DG) {
#pragma unroll
                for (int j = 0; j < U; ++j) act[j] = relu_(hid_s[st][j]);
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) act[j] = __builtin_fmaf(z_s[st][j], hp_s[st][j] - n_s[st][j], n_s[st][j]);      // h(t)
            }
            const float2 tv = ts[n * kChunkPad + tloc + st];
            float dy0 = tv.x, dy1 = tv.y;
            if constexpr (FUSED) {
                float p0 = wf.x * fs_s[st][0], p1 = wf.z * fs_s[st][0];
                if constexpr (NFS > 1) { p0 = __builtin_fmaf(wf.y, fs_s[st][1], p0); p1 = __builtin_fmaf(wf.w, fs_s[st][1], p1); }
#pragma unroll
                for (int j = 0; j < U; ++j) { p0 = __builtin_fmaf(w0[j], act[j], p0); p1 = __builtin_fmaf(w1[j], act[j], p1); }
                const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
            }
#pragma unroll
            for (int j = 0; j < U; ++j) { G.dwo[0][j] = __builtin_fmaf(dy0, act[j], G.dwo[0][j]); G.dwo[1][j] = __builtin_fmaf(dy1, act[j], G.dwo[1][j]); }
#pragma unroll
            for (int e = 0; e < NFS; ++e) { G.dwf[0][e] = __builtin_fmaf(dy0, fs_s[st][e], G.dwf[0][e]); G.dwf[1][e] = __builtin_fmaf(dy1, fs_s[st][e], G.dwf[1][e]); }
            // ---- dL/dh(t) ----
            float dht[U];
            if constexpr (DG) {
                float v[8];
#pragma unroll
                for (int j = 0; j < U; ++j) v[j] = __builtin_fmaf(dy0, w0[j], w1[j] * dy1) * relu_gate(hid_s[st][j]);
#pragma unroll
                for (int j = U; j < 8; ++j) v[j] = 0.0f;
                const Split3 Bd = split8(v);
                f32x4 acc[T::NTH];
#pragma unroll
                for (int t = 0; t < T::NTH; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] = 4 * t + e < U ? dh[4 * t + e] : 0.0f;
                mm6<T::NTH, true>(tl, T::HT, Bd, acc);
#pragma unroll
                for (int j = 0; j < U; ++j) dht[j] = acc[j / 4][j % 4];
                // d(hid) as rows of the weight-gradient contraction: tile t' <- elements 4 t' .. 4 t' + 3
#pragma unroll
                for (int t = 0; t < T::NTH; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p) s16x_tr(Bd.t[p], E.g[t], ht[st][t][p]);
                if (st == 0) {      // both steps' rows are there: fc_hid's share of the block's weight gradient (K = 32 = (sequence 4 Q + (i & 3), step i >> 2))
#pragma unroll
                    for (int t = 0; t < T::NTH; ++t) {
                        u32x4 A[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) A[p] = u32x4{ht[0][t][p][0], ht[0][t][p][1], ht[1][t][p][0], ht[1][t][p][1]};
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            u32x4 Bw[3];
#pragma unroll
                            for (int p = 0; p < 3; ++p) Bw[p] = u32x4{vh[0][j][p][0], vh[0][j][p][1], vh[1][j][p][0], vh[1][j][p][1]};
                            G.hid[t][j] = mm6w(A, Bw, G.hid[t][j]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < U; ++j) dht[j] = dh[j] + __builtin_fmaf(dy0, w0[j], w1[j] * dy1);
            }
            // ---- gate derivatives: v = [d r_pre | d z_pre | d(W_hn h) | d n_pre] ----
            float v[4 * U];
            f32x4 acc[T::NTO];
#pragma unroll
            for (int t = 0; t < T::NTO; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const float r = r_s[st][j], z = z_s[st][j], nn = n_s[st][j];
                const float dn = dht[j] * (1.0f - z);
                const float dnp = dn * __builtin_fmaf(-nn, nn, 1.0f);
                const float dgh = dnp * r;
                v[j] = dgh * nh_s[st][j] * (1.0f - r);
                v[U + j] = (hp_s[st][j] - nn) * z * dn;
                v[2 * U + j] = dgh;
                v[3 * U + j] = dnp;
                acc[j / 4][j % 4] = dht[j] * z;
            }
#pragma unroll
            for (int c = 0; c < T::NM; ++c) {
                float vcx[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vcx[i] = v[8 * c + i];
                const Split3 Bg = split8(vcx);
#pragma unroll
                for (int t = 0; t < T::NTO; ++t) {
                    f32x4 one[1] = {acc[t]};
                    mm6<1>(tl, T::BW + 3 * (t * T::NM + c), Bg, one);
                    acc[t] = one[0];
                }
                // the chunk's eight values = the rows of M tiles 2 c (elements 0 .. 3) and 2 c + 1 (elements 4 .. 7)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int p = 0; p < 3; ++p) s16x_tr(Bg.t[p], E.g[hh], gt[st][2 * c + hh][p]);
                if (st == 0) {      // the two tiles' share of the block's weight gradient, issued here so that it runs beside the next chunk's split
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const int t = 2 * c + hh;
                        u32x4 A[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) A[p] = u32x4{gt[0][t][p][0], gt[0][t][p][1], gt[1][t][p][0], gt[1][t][p][1]};
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            u32x4 Bw[3];
#pragma unroll
                            for (int p = 0; p < 3; ++p) Bw[p] = u32x4{vc[0][j][p][0], vc[0][j][p][1], vc[1][j][p][0], vc[1][j][p][1]};
                            G.cell[t][j] = mm6w(A, Bw, G.cell[t][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) dh[j] = acc[j / 4][j % 4];
            if constexpr (DX) {      // dL/dx: the feature gradient of the lane's slots (two free rows of the second output tile), gathered over the quads
                float dfs[NFS];
#pragma unroll
                for (int e = 0; e < NFS; ++e) {
                    dfs[e] = acc[(U + e) / 4][(U + e) % 4];
                    if constexpr (DG) dfs[e] = __builtin_fmaf(dy0, e ? wf.y : wf.x, __builtin_fmaf(dy1, e ? wf.w : wf.z, dfs[e]));
                }
                float df[F];
                {
                    float g0[4], g1[4];
                    gather_rows(dfs[0], g0);
                    if constexpr (F > 1 && NFS > 1) gather_rows(dfs[1], g1);
#pragma unroll
                    for (int f = 0; f < F; ++f) df[f] = (f % NFS) ? g1[f / NFS] : g0[f / NFS];
                }
                const float2 xv = xs[n * kChunkPad + tloc + st];
                float dI, dQ;
                feat_bwd<FM>(xv.x, xv.y, df, dI, dQ);
                if (q == 0) dxs[n * kChunkPad + tloc + st] = make_float2(dI, dQ);
            }
        }
    }
}

// the wave's accumulators -> one row of P + kLossCols partial gradients (every parameter has exactly one writer)
template <int FM, bool DG, int U>
__device__ __forceinline__ void s16x_write_row(float* prow, const GruLayout& L, S16XGrad<DG, U>& G, int n, int q, float loss_acc) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F, NFS = T::NFS;
    const int H = L.H, OW = DG ? H + 6 : H;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int kq = 2 * j + (n >> 3), i = n & 7;
        const int ku = U * kq + i, fsl = NFS * kq + (i - U);      // the column: hidden unit ku (i < U) or feature slot fsl
#pragma unroll
        for (int t = 0; t < T::NTF; ++t)
#pragma unroll
            for (int R = 0; R < 4; ++R) {
                constexpr int dummy = 0; (void)dummy;
                const int s = 4 * t + R, gate = s / U, u = U * q + s % U;
                const float val = G.cell[t][j][R];
                if (u < H) {
                    if (i < U) {
                        if (gate != 3 && ku < H) prow[L.o_w_hh + (gate * H + u) * H + ku] = val;
                    } else if (gate == 2) {
                        if (fsl == F) prow[L.o_b_hh + 2 * H + u] = val;
                    } else {
                        const int g = gate == 3 ? 2 : gate;
                        if (fsl < F) prow[L.o_w_ih + (g * H + u) * F + fsl] = val;
                        else if (fsl == F) {
                            prow[L.o_b_ih + g * H + u] = val;
                            if (gate < 2) prow[L.o_b_hh + g * H + u] = val;
                        }
                    }
                }
            }
        if constexpr (DG) {
#pragma unroll
            for (int t = 0; t < T::NTH; ++t)
#pragma unroll
                for (int R = 0; R < 4; ++R) {
                    const int jj = 4 * t + R, u = U * q + jj;
                    const float val = G.hid[t][j][R];
                    if (jj < U && u < H) {
                        if (i < U) { if (ku < H) prow[L.o_w_hid + u * H + ku] = val; }
                        else if (fsl == F) prow[L.o_b_hid + u] = val;
                    }
                }
        }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const float v = row_sum16(G.dwo[cc][j]);
            if (n == 0 && U * q + j < H) prow[L.o_w_out + cc * OW + U * q + j] = v;
        }
#pragma unroll
        for (int e = 0; e < NFS; ++e) {
            const float v = row_sum16(G.dwf[cc][e]);
            const int slot = NFS * q + e;
            if (n == 0) {
                if (DG && slot < F) prow[L.o_w_out + cc * OW + H + slot] = v;
                else if (slot == F) prow[L.o_b_out + cc] = v;
            }
        }
    }
    const float lp = row_sum16(loss_acc);          // accumulated on the q == 0 lanes
    if (n == 0 && q == 0) {
        prow[L.P] = lp;
        prow[L.P + 1] = 0.f; prow[L.P + 2] = 0.f; prow[L.P + 3] = 0.f;
    }
}

template <int FM, bool DG, int U, int S>
__global__ __launch_bounds__(256, 1) void gru16x_train_kernel(SeqArgs a) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F;
    constexpr int kWave = 2 * 2 * 16 * kChunkPad;
    static_assert(kChunk % S == 0, "a block never straddles two staged chunks");
    static_assert(U == 6, "checkpoint layout: six units per lane = one float4 + one float2");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* tab = smem;
    float* pl = tab + s16_tab_floats(T::NG);
    stage_params(pl, a.params, L.P);
    s16x_fill_table<FM, DG, U>(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    const S16XSel E = s16x_sel(lane);
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + 16 * kChunkPad;
    S16XGrad<DG, U> G;
    G.zero();
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * nwb, nblk = (a.T + S - 1) / S;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        float* ckg = a.ckpt + (size_t)grp * nblk * kS16xCkptFloats;
        float4* ck4 = reinterpret_cast<float4*>(ckg) + lane;
        float2* ck2 = reinterpret_cast<float2*>(ckg + 256) + lane;
        s16x_forward_pass<FM, DG, U, S, true>(a, tl, oh, xs, ck4, ck2, b0, n, lane);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        float dh[U];
#pragma unroll
        for (int j = 0; j < U; ++j) dh[j] = 0.0f;
        int cur_chunk = -1;
        // the checkpoint of a block is requested ONE BLOCK AHEAD: this wave is alone on its SIMD, a load issued where its value is needed
        // would expose the whole HBM / Infinity-Cache latency (1 - 2 us of a ~7 us block) — counters before: 14 % of the wave's cycles in s_waitcnt
        float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 nw = make_float2(0.f, 0.f);
        if (nblk > 1) {
            const size_t o = (size_t)(nblk - 1) * kS16xCkptFloats;
            nv = ck4[o / 4]; nw = ck2[o / 2];
        }
        for (int blk = nblk - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            float h0[U];
            h0[0] = nv.x; h0[1] = nv.y; h0[2] = nv.z; h0[3] = nv.w; h0[4] = nw.x; h0[5] = nw.y;
            if (blk > 1) {
                const size_t o = (size_t)(blk - 1) * kS16xCkptFloats;
                nv = ck4[o / 4]; nw = ck2[o / 2];
            } else {
                nv = make_float4(0.f, 0.f, 0.f, 0.f); nw = make_float2(0.f, 0.f);
            }
            if (chunk != cur_chunk) {
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                stage_in<16, true>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                stage_in<16, true>(ts, a.target, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                wave_lds_fence();
                cur_chunk = chunk;
            }
            if (nstep == S) s16x_train_block<FM, DG, U, S, true>(a, tl, oh, E, G, xs, ts, n, q, tb - t0, nstep, valid, h0, dh, loss_acc);
            else s16x_train_block<FM, DG, U, S, false>(a, tl, oh, E, G, xs, ts, n, q, tb - t0, nstep, valid, h0, dh, loss_acc);
        }
    }
    // one row per wave in LDS (the tables are dead), summed into the workgroup's row of partials
    const int P4 = L.P + kLossCols;
    __syncthreads();
    s16x_write_row<FM, DG, U>(smem + wave * P4, L, G, n, q, loss_acc);
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = smem[i];
        for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}

// =====================================================================================================================================
// r06: the SPLIT entry points of these models (odpd_backbone_fwd / odpd_backbone_bwd at 16-sequences-per-wave batch sizes: a trained DPD of
// 17 .. 24 units inside the chained train_dpd step, the autograd path, inference on large batches) on the same pipe, so that a checkpoint
// written by one kernel of the family is read by another of the SAME family: forward (y, optional checkpoints in this file's layout),
// backward from dL/dy with parameter gradients (and optionally dL/dx), backward from dL/dy with dL/dx only.
// =====================================================================================================================================
template <int FM, bool DG, int U, int S>
__global__ __launch_bounds__(512, 1) void gru16x_fwd_kernel(SeqArgs a) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F, NFS = T::NFS;
    constexpr int kWave = 2 * 2 * 16 * kChunkPad;
    static_assert(U == 6, "checkpoint layout: six units per lane = one float4 + one float2");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* tab = smem;
    float* pl = tab + s16_tab_floats(T::NG);
    stage_params(pl, a.params, L.P);
    s16x_fill_table<FM, DG, U>(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * kChunkPad;
    // fc_out of the lane's units and feature slots: constants of the launch
    float w0[U], w1[U];
    {
        float wv[4 * (T::NVW - 1)];
#pragma unroll
        for (int g = 0; g < T::NVW - 1; ++g) {
            const float4 v = tab_ld(tl, (T::VW + g) * 64);
            wv[4 * g] = v.x; wv[4 * g + 1] = v.y; wv[4 * g + 2] = v.z; wv[4 * g + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) { w0[j] = wv[j]; w1[j] = wv[U + j]; }
    }
    const float4 wf = tab_ld(tl, (T::VW + T::NVW - 1) * 64);
    const int nwaves = gridDim.x * nwb, nblk = (a.T + S - 1) / S;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        float* ckg = a.ckpt ? a.ckpt + (size_t)grp * nblk * kS16xCkptFloats : nullptr;
        float4* ck4 = reinterpret_cast<float4*>(ckg) + lane;
        float2* ck2 = reinterpret_cast<float2*>(ckg + 256) + lane;
        u32x4 A[T::NTF][3];
        {
            TabPtr tp = opaque(tl);
#pragma unroll
            for (int t = 0; t < T::NTF; ++t)
#pragma unroll
                for (int k = 0; k < 3; ++k) A[t][k] = tabx_ld(tp, (T::FW + 3 * t + k) * 64);
        }
        float h[U];
#pragma unroll
        for (int j = 0; j < U; ++j) h[j] = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            wave_lds_fence();
            stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xs[n * kChunkPad + tt];
                float fs[NFS];
                s16x_feats<FM, U>(xv.x, xv.y, oh, fs);
                const Split3 B = s16x_operand<U>(h, fs);
                f32x4 acc[T::NTF];
#pragma unroll
                for (int t = 0; t < T::NTF; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                mm6r<T::NTF>(A, B, acc);
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int sr = j, sz = U + j, sh = 2 * U + j, si = 3 * U + j;
                    const float r = sig_ps(acc[sr / 4][sr % 4]), z = sig_ps(acc[sz / 4][sz % 4]);
                    const float nn = tanh_x<FM>(__builtin_fmaf(r, acc[sh / 4][sh % 4], acc[si / 4][si % 4]));
                    h[j] = __builtin_fmaf(z, h[j] - nn, nn);
                }
                // head of this step: the hidden layer on an operand of its own ([h(t) | the step's feature slots])
                float act[U];
                if constexpr (DG) {
                    const Split3 B2 = s16x_operand<U>(h, fs);
                    float hid[U];
                    s16x_hid<DG, U>(opaque(tl), B2, hid);
#pragma unroll
                    for (int j = 0; j < U; ++j) act[j] = relu_(hid[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < U; ++j) act[j] = h[j];
                }
                float p0 = wf.x * fs[0], p1 = wf.z * fs[0];
                if constexpr (NFS > 1) { p0 = __builtin_fmaf(wf.y, fs[1], p0); p1 = __builtin_fmaf(wf.w, fs[1], p1); }
#pragma unroll
                for (int j = 0; j < U; ++j) { p0 = __builtin_fmaf(w0[j], act[j], p0); p1 = __builtin_fmaf(w1[j], act[j], p1); }
                const float y0 = quad_sum(p0), y1 = quad_sum(p1);
                if (q == 0) ys[n * kChunkPad + tt] = make_float2(y0, y1);
                const int t1 = t0 + tt + 1;
                if (ckg != nullptr && (t1 % S) == 0 && t1 < a.T) {
                    const size_t o = (size_t)(t1 / S) * kS16xCkptFloats;
                    ck4[o / 4] = make_float4(h[0], h[1], h[2], h[3]);
                    ck2[o / 2] = make_float2(h[4], h[5]);
                }
            }
            wave_lds_fence();
            stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
        }
    }
}

// backward from dL/dy.  NW: parameter gradients (one wave per SIMD, rows of partials as the train kernel); DX: dL/dx
template <int FM, bool DG, int U, int S, bool NW, bool DX>
__global__ __launch_bounds__(NW ? 256 : 512, 1) void gru16x_bwd_kernel(SeqArgs a) {
    using T = S16X<DG, U>;
    constexpr int F = S16Cfg<FM>::F;
    constexpr int kWave = (DX ? 3 : 2) * 2 * 16 * kChunkPad;
    static_assert(NW || DX, "nothing to compute");
    static_assert(kChunk % S == 0, "a block never straddles two staged chunks");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    float* tab = smem;
    float* pl = tab + s16_tab_floats(T::NG);
    stage_params(pl, a.params, L.P);
    s16x_fill_table<FM, DG, U>(tab, pl, L, lane, wave, nwb);
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ts = xs + 16 * kChunkPad;                     // dL/dy
    float2* dxs = DX ? ts + 16 * kChunkPad : nullptr;
    S16XSel E;
    S16XGrad<DG, U> G;
    if constexpr (NW) { E = s16x_sel(lane); G.zero(); }
    float loss_acc = 0.0f;
    const int nwaves = gridDim.x * nwb, nblk = (a.T + S - 1) / S;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        const float* ckg = a.ckpt + (size_t)grp * nblk * kS16xCkptFloats;
        const float4* ck4 = reinterpret_cast<const float4*>(ckg) + lane;
        const float2* ck2 = reinterpret_cast<const float2*>(ckg + 256) + lane;
        float dh[U];
#pragma unroll
        for (int j = 0; j < U; ++j) dh[j] = 0.0f;
        int cur_chunk = -1;
        for (int blk = nblk - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kChunk, t0 = chunk * kChunk;
            float h0[U];
            {
                const size_t o = (size_t)blk * kS16xCkptFloats;
                const float4 v = blk ? ck4[o / 4] : make_float4(0.f, 0.f, 0.f, 0.f);
                const float2 w = blk ? ck2[o / 2] : make_float2(0.f, 0.f);
                h0[0] = v.x; h0[1] = v.y; h0[2] = v.z; h0[3] = v.w; h0[4] = w.x; h0[5] = w.y;
            }
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        const int pt0 = cur_chunk * kChunk;
                        wave_lds_fence();
                        stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                    }
                }
                wave_lds_fence();
                const int len = min(kChunk, a.T - t0);
                stage_in<16>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f), a.frame_idx, a.frame_stride, a.frames_bf16 != 0);
                stage_in<16>(ts, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            if constexpr (NW) {
                if (nstep == S) s16x_train_block<FM, DG, U, S, true, false, DX>(a, tl, oh, E, G, xs, ts, n, q, tb - t0, nstep, valid, h0, dh, loss_acc, dxs);
                else s16x_train_block<FM, DG, U, S, false, false, DX>(a, tl, oh, E, G, xs, ts, n, q, tb - t0, nstep, valid, h0, dh, loss_acc, dxs);
            } else {
                if (nstep == S) s16x_block<FM, DG, U, S, true, false>(a, tl, oh, xs, ts, dxs, n, q, tb - t0, nstep, valid, h0, dh, loss_acc);
                else s16x_block<FM, DG, U, S, false, false>(a, tl, oh, xs, ts, dxs, n, q, tb - t0, nstep, valid, h0, dh, loss_acc);
            }
        }
        if constexpr (DX) {
            if (cur_chunk >= 0) {
                const int pt0 = cur_chunk * kChunk;
                wave_lds_fence();
                stage_out<16>(dxs, a.dx, b0, a.B, a.T, pt0, min(kChunk, a.T - pt0), lane);
                wave_lds_fence();
            }
        }
    }
    if constexpr (NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        s16x_write_row<FM, DG, U>(smem + wave * P4, L, G, n, q, 0.0f);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
constexpr int kS16xStride = 2;      // checkpoint stride of this kernel (its own: the workspace is private to the launch)
static bool s16x_cfg(const odpd_model_t* m, int& FM, bool& DG) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; DG = false; return true;
    case ODPD_DGRU: FM = FEAT_DGRU6; DG = true; return true;
    case ODPD_QGRU: FM = FEAT_Q4; DG = false; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; DG = false; return true;
    default: return false;
    }
}
bool gru_s16x_ok(const odpd_model_t* m) {
    int FM; bool DG;
    return s16x_cfg(m, FM, DG) && m->hidden >= 17 && m->hidden <= 24 && m->bits_w == 0 && !(m->flags & ODPD_FLAG_TWO_LAYERS) &&
           tuning().s16x != 0;
}
int64_t gru_s16x_ckpt_floats(const odpd_model_t* m, int B, int T) {
    (void)m;
    return (int64_t)((B + 15) / 16) * ((T + kS16xStride - 1) / kS16xStride) * kS16xCkptFloats;
}
template <int FM, bool DG>
static int launch_s16x(hipStream_t st, const SeqArgs& a, int P, int grid) {
    using T = S16X<DG, 6>;
    constexpr int kWave = 3 * 2 * 16 * kChunkPad;
    size_t body = (size_t)8 * kWave;
    if (body < (size_t)pad4(P)) body = pad4(P);
    const size_t lds = ((size_t)s16_tab_floats(T::NG) + body) * sizeof(float);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = gru16x_lossdx_kernel<FM, DG, 6, kS16xStride>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
    return (int)hipGetLastError();
}
bool gru_s16x_train_ok(const odpd_model_t* m) { return gru_s16x_ok(m) && tuning().s16x_train != 0; }
template <int FM, bool DG>
static int launch_s16x_train(hipStream_t st, const SeqArgs& a, int P, int grid) {
    using T = S16X<DG, 6>;
    constexpr int kWave = 2 * 2 * 16 * kChunkPad, kWaves = 4;
    size_t body = (size_t)kWaves * kWave;
    if (body < (size_t)pad4(P)) body = pad4(P);
    size_t lds = ((size_t)s16_tab_floats(T::NG) + body) * sizeof(float);
    if (lds < reduce_scratch_bytes(P, kWaves)) lds = reduce_scratch_bytes(P, kWaves);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto k = gru16x_train_kernel<FM, DG, 6, kS16xStride>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * kWaves), lds, st, a);
    return (int)hipGetLastError();
}
// fused train step (forward + loss + BPTT with weight gradients); one row of partials per workgroup, grid = gru_s16n_rows(m, B): the same rows
// (and row layout) as gru16n_kernel's train flavour, so the reduction / optimiser launches behind it are unchanged
int gru_s16x_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int grid) {
    int FM; bool DG;
    if (!s16x_cfg(m, FM, DG) || !gru_s16x_train_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a0.ckpt || !a0.partials || !a0.target) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    if (FM == FEAT_RAW2) return launch_s16x_train<FEAT_RAW2, false>(st, a, P, grid);
    if (FM == FEAT_DGRU6) return launch_s16x_train<FEAT_DGRU6, true>(st, a, P, grid);
    if (FM == FEAT_Q4) return launch_s16x_train<FEAT_Q4, false>(st, a, P, grid);
    return launch_s16x_train<FEAT_A4, false>(st, a, P, grid);
}
// ---- split entry points (forward / backward from dL/dy) ----
template <int FM, bool DG>
static int launch_s16x_fwd(hipStream_t st, const SeqArgs& a, int P) {
    using T = S16X<DG, 6>;
    constexpr int kWave = 2 * 2 * 16 * kChunkPad, kWaves = 8;
    size_t body = (size_t)kWaves * kWave;
    if (body < (size_t)pad4(P)) body = pad4(P);
    const size_t lds = ((size_t)s16_tab_floats(T::NG) + body) * sizeof(float);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    const int need = (a.ngroups + kWaves - 1) / kWaves, cap = device_cus();
    auto k = gru16x_fwd_kernel<FM, DG, 6, kS16xStride>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(need < cap ? need : cap), dim3(64 * kWaves), lds, st, a);
    return (int)hipGetLastError();
}
int gru_s16x_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0) {
    int FM; bool DG;
    if (!s16x_cfg(m, FM, DG) || !gru_s16x_train_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a0.y) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    if (FM == FEAT_RAW2) return launch_s16x_fwd<FEAT_RAW2, false>(st, a, P);
    if (FM == FEAT_DGRU6) return launch_s16x_fwd<FEAT_DGRU6, true>(st, a, P);
    if (FM == FEAT_Q4) return launch_s16x_fwd<FEAT_Q4, false>(st, a, P);
    return launch_s16x_fwd<FEAT_A4, false>(st, a, P);
}
template <int FM, bool DG, bool NW, bool DX>
static int launch_s16x_bwd(hipStream_t st, const SeqArgs& a, int P, int rows) {
    using T = S16X<DG, 6>;
    constexpr int kWave = (DX ? 3 : 2) * 2 * 16 * kChunkPad, kWaves = NW ? 4 : 8;
    size_t body = (size_t)kWaves * kWave;
    if (body < (size_t)pad4(P)) body = pad4(P);
    size_t lds = ((size_t)s16_tab_floats(T::NG) + body) * sizeof(float);
    if (NW && lds < reduce_scratch_bytes(P, kWaves)) lds = reduce_scratch_bytes(P, kWaves);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    const int need = (a.ngroups + kWaves - 1) / kWaves, cap = device_cus();
    const int grid = NW ? rows : (need < cap ? need : cap);      // (with parameter gradients: one row per workgroup, the host's row count)
    auto k = gru16x_bwd_kernel<FM, DG, 6, kS16xStride, NW, DX>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * kWaves), lds, st, a);
    return (int)hipGetLastError();
}
template <int FM, bool DG>
static int launch_s16x_bwd_mode(hipStream_t st, const SeqArgs& a, int P, int rows) {
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return launch_s16x_bwd<FM, DG, true, true>(st, a, P, rows);
    if (nw) return launch_s16x_bwd<FM, DG, true, false>(st, a, P, rows);
    if (dx) return launch_s16x_bwd<FM, DG, false, true>(st, a, P, rows);
    return ODPD_EINVAL;
}
// rows = gru_s16n_rows(m, B): the partial rows the host reduces (the exact kernels' count)
int gru_s16x_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int rows) {
    int FM; bool DG;
    if (!s16x_cfg(m, FM, DG) || !gru_s16x_train_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a0.ckpt || !a0.dy) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    if (FM == FEAT_RAW2) return launch_s16x_bwd_mode<FEAT_RAW2, false>(st, a, P, rows);
    if (FM == FEAT_DGRU6) return launch_s16x_bwd_mode<FEAT_DGRU6, true>(st, a, P, rows);
    if (FM == FEAT_Q4) return launch_s16x_bwd_mode<FEAT_Q4, false>(st, a, P, rows);
    return launch_s16x_bwd_mode<FEAT_A4, false>(st, a, P, rows);
}
// frozen-PA loss step; the grid (= loss rows the host reduces) is gru_s16n_rows(m, B), the same as the exact-fp32 kernel's
int gru_s16x_lossdx(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int grid) {
    int FM; bool DG;
    if (!s16x_cfg(m, FM, DG) || !gru_s16x_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a0.ckpt || !a0.dx || !a0.partials || !a0.target) return ODPD_EINVAL;
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const int P = gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P;
    if (FM == FEAT_RAW2) return launch_s16x<FEAT_RAW2, false>(st, a, P, grid);
    if (FM == FEAT_DGRU6) return launch_s16x<FEAT_DGRU6, true>(st, a, P, grid);
    if (FM == FEAT_Q4) return launch_s16x<FEAT_Q4, false>(st, a, P, grid);
    return launch_s16x<FEAT_A4, false>(st, a, P, grid);
}

}  // namespace odpd
