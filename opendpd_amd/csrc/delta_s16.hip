// delta_s16.hip — S16 split kernels (forward, backward for parameter gradients) of the delta-network GRU backbones
//   deltagru          backbones/deltagru.py:10-276           feat = [I,Q,a,a^3,sin,cos]; biases = initial accumulators
//   deltagru_tcnskip  backbones/deltagru_tcnskip.py:11-304   feat = [I,Q,a,a^3,I_next,Q_next]; bias-free; + TCN skip
//   deltajanet        backbones/deltajanet.py:11-274         (JAN) feat as deltagru; TWO gates f, g, both sigmoids, both fed by the
//                                                            input and the state deltas: h = (1-f) g + f h.  On the kernels' gate
//                                                            slots (r, z, n) it is z = f, n = g with r == 1, the n accumulator taking
//                                                            the state-delta product directly, sigmoid instead of tanh; the r slot
//                                                            does not exist (its table groups are zero and its MFMAs are not issued)
// for batches large enough to fill the chip with 16-sequence wavefronts (the DPD side of the train_dpd cascade at large
// batch, BASELINE config 3).  Arithmetic as in delta_family.hip (thresholded deltas, accumulators `dm`, memories x_p /
// h_p, device sparsity counters, carried accumulator gradients in the backward pass); mapping as in gru_s16.hip: lane
// (n = sequence, q = unit quad) owning units 16kt + 4q + i of NT tiles (hidden <= 16 NT; NT = 2 also lifts the
// 16-unit limit of the row-rotated delta kernels: hidden 17..32 runs here at every batch size), mat-vecs on the exact-fp32
// MFMA with operands streamed from an LDS table — the masked deltas are the B operands, the accumulators are the
// in-place C/D operands (r, z rows stored pre-multiplied by -log2(e)).  Feature deltas live on the feature slots (slot
// 4c+q on lane q of chunk c).
// Checkpoint per block of kCkptStride steps (r04: 3 float4 per lane at one unit tile instead of 7): h and h_p at the block's start
// (float4 per lane and unit tile) + ONE float4 = (x_p slot 0, x_p slot 1, mask word 0, mask word 1): the block's own threshold
// decisions, one bit each.  The four accumulators are NOT stored.  A masked delta moves its memory by exactly what it adds to the
// accumulator's operand, so the sums telescope: dm(t) = dm(0) + W_x x_p(t) + W_h h_p(t) — the backward pass rebuilds them with one
// step's worth of MFMAs per block and REPLAYS the forward pass's decisions from the mask bits, so a rounding-level difference of
// the rebuilt accumulators (summation order: ~1e-6 relative after 200 steps) can never flip a threshold between the two passes.
#include "odpd_s16.h"

namespace odpd {

namespace d16 {
constexpr int kHalo = 16;                                   // TCN taps at t-16, t, t+16
constexpr int kStride = kChunk + 2 * kHalo + 1;             // float2 per sequence row of the staged x
}  // namespace d16
// table groups and sizes for NT tiles of 16 hidden units (hidden <= 16 NT)
template <int NT>
struct D16 {
    static constexpr int IH = 0;                       // g*NT + mt           : (chunk 0, chunk 1) W_ig[16mt+m][4e+q]
    static constexpr int HH = IH + 3 * NT;             // (g*NT + mt)*NT + kt : W_hg[16mt+m][16kt+4q+e]
    static constexpr int HHT = HH + 3 * NT * NT;       // (g*NT + mt)*NT + kt : W_hg[16kt+4q+e][16mt+m]
    static constexpr int WOUT = HHT + 3 * NT * NT;     // cc*NT + mt          : fc_out[cc][16mt+4q+e]
    static constexpr int DM0 = WOUT + 2 * NT;          // j*NT + mt           : initial accumulators r, z, n, nh
    static constexpr int NG = DM0 + 4 * NT;
    static constexpr int IHT = NG;                     // g*NT + kt : W_ig[16kt+4q+e][slot(m)], slot(m) = 4 (m & 3) + (m >> 2)  (dL/dx)
    static constexpr int NG_DX = IHT + 3 * NT;
    static constexpr int kCk = 2 * NT + 1;             // float4 per lane per checkpoint: h, h_p per unit tile + (x_p0, x_p1, masks)
    // steps per checkpoint block.  Two unit tiles (hidden 17..32): two steps — the backward's per-step saved state x four steps x two
    // tiles does not fit the register file even at one wave per SIMD (r03: 430 .. 630 B of scratch per lane), and since r04 a checkpoint
    // is five float4, so halving the block costs 80 B per sequence and step of HBM traffic, not 208
    static constexpr int S = NT == 1 ? kCkptStride : 2;
    static constexpr int kTiles = 5 * NT + 1;          // gr gz gn gnh dhm per unit tile + feature-delta tile
};

// JAN: gate slot g (0 r, 1 z, 2 n) holds parameter gate g - 1 (f, g); slot 0 is empty; the n slot is a sigmoid gate as well
template <bool TRES, int NT, bool JAN = false>
__device__ __forceinline__ float4 d16_entry(const float* pl, const DeltaLayout& L, int grp, int m, int q) {
    using T = D16<NT>;
    const int H = L.H;
    auto pg = [](int g) { return JAN ? g - 1 : g; };
    auto has = [](int g) { return !JAN || g > 0; };
    auto sc = [](int g) { return (g < 2 || JAN) ? kNegLog2e : 1.0f; };
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (grp < T::HH) {
            const int g = grp / NT, o = 16 * (grp % NT) + m, slot = 4 * e + q;
            v[e] = (has(g) && e < 2 && slot < 6 && o < H) ? pl[L.o_w_ih + (pg(g) * H + o) * 6 + slot] * sc(g) : 0.0f;
        } else if (grp < T::HHT) {
            const int r = grp - T::HH, g = r / (NT * NT), o = 16 * ((r / NT) % NT) + m, k = 16 * (r % NT) + 4 * q + e;
            v[e] = (has(g) && o < H && k < H) ? pl[L.o_w_hh + (pg(g) * H + o) * H + k] * sc(g) : 0.0f;
        } else if (grp < T::WOUT) {
            const int r = grp - T::HHT, g = r / (NT * NT), i = 16 * ((r / NT) % NT) + m, k = 16 * (r % NT) + 4 * q + e;
            v[e] = (has(g) && i < H && k < H) ? pl[L.o_w_hh + (pg(g) * H + k) * H + i] : 0.0f;
        } else if (grp < T::DM0) {
            const int r = grp - T::WOUT, k = 16 * (r % NT) + 4 * q + e;
            v[e] = k < H ? pl[L.o_w_out + (r / NT) * H + k] : 0.0f;
        } else if (grp >= T::IHT) {
            // transposed input weights with the output rows permuted so that D row 4 q' + i = slot 4 i + q': the MFMA result
            // of lane (n, q) element c IS the gradient of the lane's own feature slot 4 c + q
            const int r = grp - T::IHT, g = r / NT, k = 16 * (r % NT) + 4 * q + e, slot = 4 * (m & 3) + (m >> 2);
            v[e] = (has(g) && slot < 6 && k < H) ? pl[L.o_w_ih + (pg(g) * H + k) * 6 + slot] : 0.0f;
        } else {
            const int r = grp - T::DM0, j = r / NT, k = 16 * (r % NT) + 4 * q + e;
            float b = 0.0f;
            if (JAN) {       // dm_f, dm_g start at b_ih + b_hh (deltajanet.py:162-166)
                if (k < H && (j == 1 || j == 2)) b = (pl[L.o_b_ih + (j - 1) * H + k] + pl[L.o_b_hh + (j - 1) * H + k]) * kNegLog2e;
            } else if (!TRES && k < H) {
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
template <int NT>
struct D16State { f32x4 h[NT], hp[NT], dmr[NT], dmz[NT], dmn[NT], dmnh[NT]; float xp[2]; };

template <bool TRES>
__device__ __forceinline__ void d16_slots(float2 xv, float2 xn, const float (&oh)[4], float (&fs)[2]) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), a = __builtin_amdgcn_sqrtf(a2);
    float f4, f5;
    if constexpr (TRES) { f4 = xn.x; f5 = xn.y; }
    else { const float ia = fast_rcp(a); f4 = xv.y * ia; f5 = xv.x * ia; }
    fs[0] = __builtin_fmaf(oh[0], xv.x, __builtin_fmaf(oh[1], xv.y, __builtin_fmaf(oh[2], a, oh[3] * (a2 * a))));
    fs[1] = __builtin_fmaf(oh[0], f4, oh[1] * f5);
}

// threshold decisions of one block of kCkptStride steps, one bit each, as the forward pass took them (checkpoint float4 .z / .w):
// word kt, bit 4 s + i: hidden unit 16 kt + 4 q + i of step s was kept;  word 0, bit 16 + 2 s + c: feature slot c of step s.
// The forward pass builds the words in float arithmetic (24 bits: exact), the backward pass converts them once per block.
__device__ __forceinline__ bool d16_bit(unsigned w, int pos) { return ((w >> pos) & 1u) != 0u; }

// the accumulators take one set of masked deltas: dm += W_x dx + W_h dh  (JAN: both gates' state parts go to dm itself)
template <int NT, bool JAN>
__device__ __forceinline__ void d16_accumulate(TabPtr tl, const float (&dxm)[2], const f32x4 (&dhm)[NT], D16State<NT>& st) {
    using T = D16<NT>;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const float4 wr = tab_ld(tl, (T::IH + 0 * NT + mt) * 64), wz = tab_ld(tl, (T::IH + 1 * NT + mt) * 64),
                     wn = tab_ld(tl, (T::IH + 2 * NT + mt) * 64);
        if constexpr (!JAN) { st.dmr[mt] = mfma4(wr.x, dxm[0], st.dmr[mt]); st.dmr[mt] = mfma4(wr.y, dxm[1], st.dmr[mt]); }
        st.dmz[mt] = mfma4(wz.x, dxm[0], st.dmz[mt]); st.dmz[mt] = mfma4(wz.y, dxm[1], st.dmz[mt]);
        st.dmn[mt] = mfma4(wn.x, dxm[0], st.dmn[mt]); st.dmn[mt] = mfma4(wn.y, dxm[1], st.dmn[mt]);
    }
    if constexpr (!JAN) s16n_matvec<NT>(tl, T::HH + 0 * NT * NT, dhm, st.dmr);
    s16n_matvec<NT>(tl, T::HH + 1 * NT * NT, dhm, st.dmz);
    if constexpr (JAN) s16n_matvec<NT>(tl, T::HH + 2 * NT * NT, dhm, st.dmn);       // dm = (dx W_ih^T + dm) + dh W_hh^T for both gates
    else s16n_matvec<NT>(tl, T::HH + 2 * NT * NT, dhm, st.dmnh);
}

// one forward step.  slot_ok[c]: the lane's slot of chunk c is a real feature; unit_ok[kt][i]: a real hidden unit.
// REPLAY (backward recompute): the keep / drop decisions are not taken but read from the block's mask words, step `sidx` of the block
template <bool TRES, int NT, bool JAN = false, bool REPLAY = false>
__device__ __forceinline__ void d16_cell_fwd(TabPtr tl, const float (&fs)[2], float thx, float thh, const bool (&slot_ok)[2],
                                             const f32x4 (&unit_ok)[NT], D16State<NT>& st, f32x4 (&hprev)[NT], f32x4 (&dhm)[NT],
                                             f32x4 (&mh)[NT], f32x4 (&r)[NT], f32x4 (&z)[NT], f32x4 (&n)[NT], float (&dxm)[2],
                                             float& zx, float& zh, float (&mx)[2], const unsigned* mw = nullptr, int sidx = 0) {
    using T = D16<NT>;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float d = fs[c] - st.xp[c];
        // masked_fill(|d| < th, 0)  (deltagru.py:179-183); x_p follows where the delta was kept (|d| >= th: the same predicate)
        const bool keep = REPLAY ? d16_bit(mw[0], 16 + 2 * sidx + c) : !(__builtin_fabsf(d) < thx);
        dxm[c] = keep ? d : 0.0f;
        mx[c] = keep ? 1.0f : 0.0f;
        st.xp[c] = keep ? fs[c] : st.xp[c];
        if constexpr (!REPLAY) zx += (slot_ok[c] && dxm[c] == 0.0f) ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
        ODPD_EACH4 {
            const float d = st.h[kt][i] - st.hp[kt][i];
            const bool keep = REPLAY ? d16_bit(mw[kt], 4 * sidx + i) : !(__builtin_fabsf(d) < thh);
            dhm[kt][i] = keep ? d : 0.0f;
            mh[kt][i] = keep ? 1.0f : 0.0f;
            st.hp[kt][i] = keep ? st.h[kt][i] : st.hp[kt][i];
            if constexpr (!REPLAY) zh += (unit_ok[kt][i] != 0.0f && dhm[kt][i] == 0.0f) ? 1.0f : 0.0f;
        }
    d16_accumulate<NT, JAN>(tl, dxm, dhm, st);
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        z[mt] = sigmoid4_prescaled(st.dmz[mt]);
        if constexpr (JAN) {
            r[mt] = splat4(1.0f);
            n[mt] = sigmoid4_prescaled(st.dmn[mt]);                                  // deltajanet.py:247: the candidate is a sigmoid
        } else {
            r[mt] = sigmoid4_prescaled(st.dmr[mt]);
            n[mt] = tanh4_precise(fma4(r[mt], st.dmnh[mt], st.dmn[mt]));
        }
        hprev[mt] = st.h[mt];
        st.h[mt] = fma4(z[mt], sub4(st.h[mt], n[mt]), n[mt]);
    }
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

template <int CH = kChunk>
__device__ __forceinline__ void d16_stage_x(float2* lds, const float* g, int b0, int B, int T, int t0, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int PER = CH + 2 * d16::kHalo, TOT = 16 * PER, N = (TOT + 63) / 64, STR = CH + 2 * d16::kHalo + 1;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        if (e < TOT) {
            const int m = e / PER, pos = e % PER, tg = t0 - d16::kHalo + pos;
            float2 v = make_float2(0.0f, 0.0f);          // outside the frame: the conv's zero padding
            if (tg >= 0 && tg < T) v = (b0 + m < B) ? g2[(size_t)(b0 + m) * T + tg] : make_float2(0.5f, 0.5f);
            lds[m * STR + pos] = v;
        }
    }
}
// stage_in / stage_out of odpd_seq.h for a kernel-local chunk length CH ([16 sequences][CH + 1] float2)
template <int CH>
__device__ __forceinline__ void d16_stage_in(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane, float2 fill) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int N = 16 * CH / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH;
        float2 v = fill;      // (an `if`, not `?:`: with a run-time `fill` the compiler selects between two ADDRESSES — see stage_in_ch, odpd_seq.h)
        if (tt < len && b0 + m < B) v = g2[(size_t)(b0 + m) * T + t0 + tt];
        lds[m * (CH + 1) + tt] = v;
    }
}
template <int CH>
__device__ __forceinline__ void d16_stage_out(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
    constexpr int N = 16 * CH / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH;
        if (tt < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + tt] = lds[m * (CH + 1) + tt];
    }
}
// r06, the forward kernels of one unit tile: four waves per SIMD.  The staged x carries no halo — CH + 1 samples per sequence, the step's own and
// the next (torch.roll) — and the chunks are 16 steps, so a wave's LDS share falls 12.5 -> 4.4 KB and TWO eight-wave workgroups fit a CU (the
// kernels are capped at 128 registers for that).  deltagru_tcnskip (BASELINE config 3's DPD): the TCN skip leaves the step loop.  It has no
// state, so it is evaluated where the chunk is staged (lane = (sequence, step) there: the 44 instructions once per sample, its taps x[t - 16],
// x[t], x[t + 16] read from global memory — L2 hits of the stream the kernel stages anyway) and parked in the output tile; the step adds to it.
template <int CH>
__device__ __forceinline__ void d16_stage_x1(float2* lds, const float* g, int b0, int B, int T, int t0, int lane) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int PER = CH + 1, TOT = 16 * PER, N = (TOT + 63) / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        if (e < TOT) {
            const int m = e / PER, pos = e % PER, tg = t0 + pos;
            lds[m * (CH + 2) + pos] = (tg < T && b0 + m < B) ? g2[(size_t)(b0 + m) * T + tg] : make_float2(0.5f, 0.5f);
        }
    }
}
// the TCN skip of the chunk's samples, parked where the step loop will add the recurrent output: lane = (sequence, step), taps from global memory,
// issued with the chunk's staging loads (their latency is waited for once, together)
template <int CH>
__device__ __forceinline__ void d16_park_skip(float2* ys, const float* x, const D16Scalars<true>& sc, int b0, int B, int T, int t0, int len, int lane) {
    const float2* x2 = reinterpret_cast<const float2*>(x);
    constexpr int N = 16 * CH / 64;
    const float2 zero = make_float2(0.0f, 0.0f);          // outside the frame: the conv's zero padding
    float2 xm[N], xc[N], xq[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH, t = t0 + tt;
        const bool ok = tt < len && b0 + m < B;
        const float2* row = x2 + (size_t)(b0 + m) * T;
        xm[j] = (ok && t >= d16::kHalo) ? row[t - d16::kHalo] : zero;
        xc[j] = ok ? row[t] : zero;
        xq[j] = (ok && t + d16::kHalo < T) ? row[t + d16::kHalo] : zero;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH;
        float s1[3], s2[2];
        d16_tcn<true>(sc, xm[j], xc[j], xq[j], s1, s2);
        ys[m * (CH + 1) + tt] = make_float2(hardswishf_(s2[0]), hardswishf_(s2[1]));
    }
}
// staging geometry of the forward kernel: chunk length, float2 per staged x row, offset of step 0 in it, floats of LDS per wave
template <bool TRES, int NT> struct D16Fwd {
    static constexpr bool LEAN = NT == 1, SKIPOUT = TRES && LEAN;
    static constexpr int CH = LEAN ? 16 : kChunk, XROW = LEAN ? CH + 2 : d16::kStride, XOFF = LEAN ? 0 : d16::kHalo, YROW = CH + 1;
    static constexpr int kWave = 2 * 16 * XROW + 2 * 16 * YROW;
    static constexpr int kWavesPerSimd = LEAN ? 4 : 1;
};

// Chunk length and workgroup size of the backward kernel.  The weight-gradient-only kernel of one unit tile (the trained DPD of train_dpd,
// BASELINE config 3) runs EIGHT waves per workgroup — two per SIMD — on 16-step chunks (the per-wave LDS region then fits eight times next to
// the operand table): a lone wave issues one VALU instruction per ~4.7 cycles, two sharing a SIMD one per ~2.3.  r04 measured that shape at
// -4 % with 472 B of scratch per lane; r05 took the TCN skip's weight gradient out of the kernel (tres_skip_wgrad_kernel: it has no state in
// it and is time-parallel) — 24 accumulators and their arithmetic less — which leaves 248 B of scratch and 1.65 -> 1.34 ms at 65 536 x 200
// (profiles/r05/delta16.md; two-step blocks would spill less and are SLOWER: 1.40 ms).  The other variants keep four waves and 32 steps.
template <int NT, bool DX> constexpr int d16_bwd_ch() { return (NT == 1 && !DX) ? 16 : kChunk; }

template <int NT>
__device__ __forceinline__ void d16_init_state(TabPtr tl, D16State<NT>& st) {
    using T = D16<NT>;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        st.h[mt] = z4; st.hp[mt] = z4;
        st.dmr[mt] = as_f32x4(tab_ld(tl, (T::DM0 + 0 * NT + mt) * 64)); st.dmz[mt] = as_f32x4(tab_ld(tl, (T::DM0 + 1 * NT + mt) * 64));
        st.dmn[mt] = as_f32x4(tab_ld(tl, (T::DM0 + 2 * NT + mt) * 64)); st.dmnh[mt] = as_f32x4(tab_ld(tl, (T::DM0 + 3 * NT + mt) * 64));
    }
    st.xp[0] = st.xp[1] = 0.0f;
}
__device__ __forceinline__ float4 d16_f4(const f32x4& v) { return make_float4(v[0], v[1], v[2], v[3]); }

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <bool TRES, int NT, bool JAN = false>
__global__ __launch_bounds__(NT == 1 ? 512 : 256, (D16Fwd<TRES, NT>::kWavesPerSimd)) void delta16_fwd_kernel(SeqArgs a) {
    using T = D16<NT>;
    using G = D16Fwd<TRES, NT>;
    constexpr int S = D16<NT>::S;
    constexpr int kWave = G::kWave, CH = G::CH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const DeltaLayout L = delta_layout(a.H, TRES, JAN ? 2 : 3);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    {
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int grp = wave; grp < T::NG; grp += nwb) t4[grp * 64 + lane] = d16_entry<TRES, NT, JAN>(pl, L, grp, n, q);
        __syncthreads();
    }
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    D16Scalars<TRES> sc;
    sc.load(pl, L);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    const bool slot_ok[2] = {true, q < 2};
    f32x4 unit_ok[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) ODPD_EACH4 unit_ok[kt][i] = (16 * kt + 4 * q + i < a.H) ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(T::NG) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* ys = xs + 16 * G::XROW;
    const float2* xr = xs + n * G::XROW + G::XOFF;
    float zx = 0.0f, zh = 0.0f;
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        float4* ck = a.ckpt ? reinterpret_cast<float4*>(a.ckpt) + (size_t)grp * a.nck * T::kCk * 64 + lane : nullptr;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[(size_t)(b0 + n) * a.T] : make_float2(0.5f, 0.5f);
        D16State<NT> st;
        d16_init_state<NT>(tl, st);
        float zxs = 0.0f, zhs = 0.0f;
        // mask words of the running block (float arithmetic on integers < 2^24: exact), x_p at its start, 16^s and 2^(16 + 2 s) of step s
        float mwf[NT], xps[2] = {st.xp[0], st.xp[1]}, scale = 1.0f, scale_x = 65536.0f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) mwf[kt] = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += CH) {
            const int len = min(CH, a.T - t0);
            wave_lds_fence();
            if constexpr (G::SKIPOUT) d16_park_skip<CH>(ys, a.x, sc, b0, a.B, a.T, t0, len, lane);
            if constexpr (G::LEAN) d16_stage_x1<CH>(xs, a.x, b0, a.B, a.T, t0, lane);
            else d16_stage_x(xs, a.x, b0, a.B, a.T, t0, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const float2 xv = xr[tt];
                const float2 xn = (t0 + tt + 1 < a.T) ? xr[tt + 1] : x0;      // torch.roll(x, -1): the last step sees sample 0
                float fs[2], dxm[2], mx[2];
                f32x4 hprev[NT], dhm[NT], mh[NT], r[NT], z[NT], nn[NT];
                d16_slots<TRES>(xv, xn, oh, fs);
                d16_cell_fwd<TRES, NT, JAN>(opaque(tl), fs, a.thx, a.thh, slot_ok, unit_ok, st, hprev, dhm, mh, r, z, nn, dxm, zxs, zhs, mx);
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < NT; ++mt) {
                    const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
                    ODPD_EACH4 { p0 = __builtin_fmaf(w0[i], st.h[mt][i], p0); p1 = __builtin_fmaf(w1[i], st.h[mt][i], p1); }
                }
                const float y0 = quad_sum(p0) + sc.bout[0], y1 = quad_sum(p1) + sc.bout[1];
                if constexpr (G::SKIPOUT) {          // (same sum as before: y + HS(s2))
                    if (q == 0) { const float2 sk = ys[n * G::YROW + tt]; ys[n * G::YROW + tt] = make_float2(y0 + sk.x, y1 + sk.y); }
                } else if (q == 0) ys[n * G::YROW + tt] = make_float2(y0, y1);
                if constexpr (TRES && !G::SKIPOUT) {
                    // the TCN skip has no state: instead of all four quads of a sequence evaluating the same 44 instructions every step
                    // (r01..r03), quad q evaluates step q of each block of four steps and adds it to the parked output
                    if ((tt & 3) == 3 || tt == len - 1) {
                        const int ts = (tt & ~3) + q;
                        wave_lds_fence();
                        if (ts <= tt) {
                            float s1[3], s2[2];
                            d16_tcn<TRES>(sc, xr[ts - d16::kHalo], xr[ts], xr[ts + d16::kHalo], s1, s2);
                            float2 yv = ys[n * kChunkPad + ts];
                            yv.x += hardswishf_(s2[0]); yv.y += hardswishf_(s2[1]);
                            ys[n * kChunkPad + ts] = yv;
                        }
                        wave_lds_fence();
                    }
                }
                const int t1 = t0 + tt + 1;
                if (ck != nullptr) {
                    // this step's decisions into the block's mask words: unit bits 4 s + i (weight 16^s 2^i), slot bits 16 + 2 s + c
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt) {
                        const float nib = __builtin_fmaf(mh[kt][3], 8.0f, __builtin_fmaf(mh[kt][2], 4.0f, __builtin_fmaf(mh[kt][1], 2.0f, mh[kt][0])));
                        mwf[kt] = __builtin_fmaf(nib, scale, mwf[kt]);
                    }
                    mwf[0] = __builtin_fmaf(__builtin_fmaf(mx[1], 2.0f, mx[0]), scale_x, mwf[0]);
                    scale *= 16.0f; scale_x *= 4.0f;
                    if ((t1 % S) == 0 || t1 == a.T) {          // the block ends: its masks + x_p at its start; the next block's start state
                        ck[(size_t)((t1 - 1) / S) * T::kCk * 64 + 2 * NT * 64] = make_float4(xps[0], xps[1], mwf[0], NT > 1 ? mwf[NT - 1] : 0.0f);
                        if (t1 < a.T) {
                            float4* c = ck + (size_t)(t1 / S) * T::kCk * 64;
#pragma unroll
                            for (int kt = 0; kt < NT; ++kt) { c[(0 * NT + kt) * 64] = d16_f4(st.h[kt]); c[(1 * NT + kt) * 64] = d16_f4(st.hp[kt]); }
                        }
                        xps[0] = st.xp[0]; xps[1] = st.xp[1]; scale = 1.0f; scale_x = 65536.0f;
#pragma unroll
                        for (int kt = 0; kt < NT; ++kt) mwf[kt] = 0.0f;
                    }
                }
            }
            wave_lds_fence();
            if constexpr (G::LEAN) d16_stage_out<CH>(ys, a.y, b0, a.B, a.T, t0, len, lane);
            else stage_out<16>(ys, a.y, b0, a.B, a.T, t0, len, lane);
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
template <bool TRES, int NT>
struct D16Grad {
    f32x4 thh[3][NT][NT], tih[3][NT];
    f32x4 dwout[2][NT], db[4][NT];
    float dbout[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            dwout[0][a] = dwout[1][a] = z4;
#pragma unroll
            for (int j = 0; j < 4; ++j) db[j][a] = z4;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                tih[g][a] = z4;
#pragma unroll
                for (int b = 0; b < NT; ++b) thh[g][a][b] = z4;
            }
        }
        dbout[0] = dbout[1] = 0.f;
    }
};
template <int NT>
struct D16Carry { f32x4 gh[NT], ghp[NT], gr[NT], gz[NT], gn[NT], gnh[NT]; float gxp[2], wrap[2]; };

template <bool TRES, int NT, bool FULL, bool DX, bool JAN = false>
__device__ __forceinline__ void d16_bwd_block(const SeqArgs& a, TabPtr tl0, const D16Scalars<TRES>& sc, const float (&oh)[4],
                                              D16Grad<TRES, NT>& G, const float2* xr, const float2* dys, float2* dxs, float* tiles,
                                              float2 x0, int n, int q, int tglob, int tloc, int nstep, int chunk_len, float* dxrow,
                                              D16State<NT> st, D16Carry<NT>& C, const unsigned (&mw)[NT]) {
    using T = D16<NT>;
    constexpr int S = D16<NT>::S, CH = d16_bwd_ch<NT, DX>();
    const bool slot_ok[2] = {true, q < 2};
    f32x4 all_units[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) all_units[kt] = f32x4{1.f, 1.f, 1.f, 1.f};
    // saved per step: the gates, the n-gate's state accumulator and the masked input deltas.  NOT saved (r04; they were 16 more registers
    // per step): the masks — read back from the block's decision bits where they are used —, and h(t-1) / the masked state delta, which
    // follow from the block's start state and the saved gates with a few VALU operations (hprev / dhm below)
    f32x4 r_s[S][NT], z_s[S][NT], n_s[S][NT], nh_s[S][NT];
    float dxm_s[S][2];
    f32x4 h0[NT], hp0[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) { h0[kt] = st.h[kt]; hp0[kt] = st.hp[kt]; }
    TabPtr tl = opaque(tl0);
    {
        float zx = 0.f, zh = 0.f;
#pragma unroll
        for (int si = 0; si < S; ++si) {
            if (FULL || si < nstep) {
                const float2 xv = xr[tloc + si];
                const float2 xn = (tglob + si + 1 < a.T) ? xr[tloc + si + 1] : x0;
                float fs[2];
                d16_slots<TRES>(xv, xn, oh, fs);
                f32x4 hprev_[NT], dhm_[NT], mh_[NT];
                float mx_[2];
                d16_cell_fwd<TRES, NT, JAN, true>(tl, fs, a.thx, a.thh, slot_ok, all_units, st, hprev_, dhm_, mh_, r_s[si],
                                                  z_s[si], n_s[si], dxm_s[si], zx, zh, mx_, mw, si);
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) nh_s[si][kt] = st.dmnh[kt];
            }
        }
    }
    tl = opaque(tl0);
    auto tile = [tiles](int qty, int kt) { return tiles + (qty * NT + kt) * kTileFloats; };   // qty: 0 gr 1 gz 2 gn 3 gnh 4 dhm
    float* t_f = tiles + 5 * NT * kTileFloats;
    const f32x4 one = splat4(1.0f);
    // (TRes: the TCN skip's weight gradient has no state in it — tres_skip_wgrad_kernel, time-parallel, writes it into rows of its own)
#pragma unroll
    for (int si = S - 1; si >= 0; --si) {
        if (FULL || si < nstep) {
            const int tt = tloc + si;
            const float2 dyv = dys[n * (CH + 1) + tt];
            G.dbout[0] += q == 0 ? dyv.x : 0.0f;
            G.dbout[1] += q == 0 ? dyv.y : 0.0f;
            __builtin_amdgcn_sched_barrier(0);      // keep every step's work together (the block is fully unrolled: measured -1.8 %)
            // h(t-1), h_p before the step and the masked state delta of step si: the forward recurrences h <- z (h - n) + n and
            // h_p <- kept ? h : h_p replayed over the block's earlier steps (same operations on the same values: the same bits)
            f32x4 hprev[NT], dhm[NT], mhk[NT];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                f32x4 hh = h0[kt], pp = hp0[kt];
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    if (k < si) {
                        ODPD_EACH4 pp[i] = d16_bit(mw[kt], 4 * k + i) ? hh[i] : pp[i];
                        hh = fma4(z_s[k][kt], sub4(hh, n_s[k][kt]), n_s[k][kt]);
                    }
                }
                hprev[kt] = hh;
                ODPD_EACH4 {
                    const bool keep = d16_bit(mw[kt], 4 * si + i);
                    dhm[kt][i] = keep ? hh[i] - pp[i] : 0.0f;
                    mhk[kt][i] = keep ? 1.0f : 0.0f;
                }
            }
            f32x4 ghprev[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const f32x4 w0 = as_f32x4(tab_ld(tl, (T::WOUT + mt) * 64)), w1 = as_f32x4(tab_ld(tl, (T::WOUT + NT + mt) * 64));
                const f32x4 r = r_s[si][mt], z = z_s[si][mt], nn = n_s[si][mt];
                const f32x4 ht = fma4(z, sub4(hprev[mt], nn), nn);
                const f32x4 gh = add4(C.gh[mt], fma4(splat4(dyv.x), w0, mul4(w1, splat4(dyv.y))));
                G.dwout[0][mt] = fma4(splat4(dyv.x), ht, G.dwout[0][mt]);
                G.dwout[1][mt] = fma4(splat4(dyv.y), ht, G.dwout[1][mt]);
                const f32x4 dn = mul4(gh, sub4(one, z)), dz = mul4(gh, sub4(hprev[mt], nn));
                ghprev[mt] = mul4(gh, z);
                f32x4 omn2;
                ODPD_EACH4 omn2[i] = JAN ? nn[i] * (1.0f - nn[i]) : __builtin_fmaf(-nn[i], nn[i], 1.0f);     // sigmoid' | tanh'
                const f32x4 dpre = mul4(dn, omn2);
                C.gn[mt] = add4(C.gn[mt], dpre);
                if constexpr (JAN) {
                    C.gnh[mt] = C.gn[mt];            // the state delta feeds the g accumulator itself: one carried gradient
                } else {
                    C.gnh[mt] = fma4(dpre, r, C.gnh[mt]);
                    C.gr[mt] = fma4(mul4(dpre, nh_s[si][mt]), mul4(r, sub4(one, r)), C.gr[mt]);
                }
                C.gz[mt] = fma4(dz, mul4(z, sub4(one, z)), C.gz[mt]);
            }
            // data gradient to the masked dh: W_hh^T [G_r, G_z, G_nh]
            f32x4 ddh[NT];
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) ddh[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!JAN) s16n_matvec<NT>(tl, T::HHT + 0 * NT * NT, C.gr, ddh);
            s16n_matvec<NT>(tl, T::HHT + 1 * NT * NT, C.gz, ddh);
            s16n_matvec<NT>(tl, T::HHT + 2 * NT * NT, C.gnh, ddh);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                const f32x4 mk = mhk[mt];
                C.gh[mt] = fma4(mk, add4(ddh[mt], C.ghp[mt]), ghprev[mt]);
                ODPD_EACH4 C.ghp[mt][i] = __builtin_fmaf(-mk[i], ddh[mt][i], (1.0f - mk[i]) * C.ghp[mt][i]);
            }
            if constexpr (DX) {
                // dL/d(masked feature delta) = W_ih^T [G_r, G_z, G_n], landing on the lane's own feature slots; then the same
                // keep / carry logic as for h (x_p <- x where kept):  dL/dfeat = m (g + G_xp),  G_xp <- -m g + (1 - m) G_xp
                f32x4 ds = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    const f32x4 wr = as_f32x4(tab_ld(tl, (T::IHT + 0 * NT + kt) * 64)), wz = as_f32x4(tab_ld(tl, (T::IHT + 1 * NT + kt) * 64)),
                                wn = as_f32x4(tab_ld(tl, (T::IHT + 2 * NT + kt) * 64));
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if constexpr (!JAN) ds = mfma4(wr[c], C.gr[kt][c], ds);
                        ds = mfma4(wz[c], C.gz[kt][c], ds); ds = mfma4(wn[c], C.gn[kt][c], ds);
                    }
                }
                float dfs[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float m = d16_bit(mw[0], 16 + 2 * si + c) ? 1.0f : 0.0f, g = ds[c];
                    dfs[c] = m * (g + C.gxp[c]);
                    C.gxp[c] = __builtin_fmaf(-m, g, (1.0f - m) * C.gxp[c]);
                }
                // feature Jacobian: every lane contributes its two slots (4 c + q), the sequence's four lanes are summed
                const float2 xv = xr[tt];
                float dI, dQ, nI = 0.0f, nQ = 0.0f;
                if constexpr (TRES) {      // [I, Q, |x|, |x|^3, I_next, Q_next]: slots 4, 5 belong to sample t + 1
                    const float df[4] = {oh[0] * dfs[0], oh[1] * dfs[0], oh[2] * dfs[0], oh[3] * dfs[0]};
                    feat_bwd<FEAT_A4>(xv.x, xv.y, df, dI, dQ);
                    nI = quad_sum(oh[0] * dfs[1]); nQ = quad_sum(oh[1] * dfs[1]);
                } else {                   // [I, Q, |x|, |x|^3, sin, cos]
                    const float df[6] = {oh[0] * dfs[0], oh[1] * dfs[0], oh[2] * dfs[0], oh[3] * dfs[0], oh[0] * dfs[1], oh[1] * dfs[1]};
                    feat_bwd<FEAT_DGRU6>(xv.x, xv.y, df, dI, dQ);
                }
                dI = quad_sum(dI); dQ = quad_sum(dQ);
                if (q == 0) {
                    dxs[n * (CH + 1) + tt] = make_float2(dI, dQ);
                    if constexpr (TRES) {
                        const int t1 = tglob + si + 1;
                        if (t1 >= a.T) { C.wrap[0] = nI; C.wrap[1] = nQ; }                 // torch.roll: the last step's "next" is sample 0
                        else if (tt + 1 < chunk_len) { dxs[n * (CH + 1) + tt + 1].x += nI; dxs[n * (CH + 1) + tt + 1].y += nQ; }
                        else if (dxrow != nullptr) {   // sample t + 1 lives in the chunk this wave flushed before: add at L2
                            __threadfence();
                            atomicAdd(dxrow + 2 * t1, nI);
                            atomicAdd(dxrow + 2 * t1 + 1, nQ);
                        }
                    }
                }
            }
            // weight gradients: dW_ih += G_dm^T (x) dx_masked, dW_hh += [G_r, G_z, G_nh]^T (x) dh_masked
            wave_lds_fence();
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if constexpr (!JAN) tile_put(tile(0, kt), n, q, C.gr[kt]);
                tile_put(tile(1, kt), n, q, C.gz[kt]);
                tile_put(tile(2, kt), n, q, C.gn[kt]);
                tile_put(tile(3, kt), n, q, C.gnh[kt]);
                tile_put(tile(4, kt), n, q, dhm[kt]);
            }
            t_f[n * kTilePitch + q] = dxm_s[si][0];
            t_f[n * kTilePitch + 4 + q] = dxm_s[si][1];      // slots 6, 7 are zero deltas (the lanes' features are 0 there)
            wave_lds_fence();
            float fT[4], hT[NT][4];
            tile_get(t_f, n, q, fT);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) tile_get(tile(4, kt), n, q, hT[kt]);
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) {
                float rT[4] = {0.f, 0.f, 0.f, 0.f}, zT[4], nT[4], gT[4];
                if constexpr (!JAN) tile_get(tile(0, mt), n, q, rT);
                tile_get(tile(1, mt), n, q, zT);
                tile_get(tile(2, mt), n, q, nT); tile_get(tile(3, mt), n, q, gT);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (!JAN) G.tih[0][mt] = mfma4(rT[c], fT[c], G.tih[0][mt]);
                    G.tih[1][mt] = mfma4(zT[c], fT[c], G.tih[1][mt]);
                    G.tih[2][mt] = mfma4(nT[c], fT[c], G.tih[2][mt]);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        if constexpr (!JAN) G.thh[0][mt][nt] = mfma4(rT[c], hT[nt][c], G.thh[0][mt][nt]);
                        G.thh[1][mt][nt] = mfma4(zT[c], hT[nt][c], G.thh[1][mt][nt]);
                        G.thh[2][mt][nt] = mfma4(gT[c], hT[nt][c], G.thh[2][mt][nt]);
                    }
                }
            }
        }
    }
}

template <bool TRES, int NT, bool JAN = false>
__device__ __forceinline__ void d16_write_row(float* prow, const DeltaLayout& L, D16Grad<TRES, NT>& G, int lane, int n, int q) {
    const int H = L.H;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int u = 16 * mt + 4 * q + rr;
            if (u < H) {
#pragma unroll
                for (int g = JAN ? 1 : 0; g < 3; ++g) {
                    const int pgt = JAN ? g - 1 : g;         // parameter gate of slot g
                    if (n < 6) prow[L.o_w_ih + (pgt * H + u) * 6 + n] = G.tih[g][mt][rr];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        if (16 * nt + n < H) prow[L.o_w_hh + (pgt * H + u) * H + 16 * nt + n] = G.thh[g][mt][nt][rr];
                }
            }
            const float w0 = row_sum16(G.dwout[0][mt][rr]), w1 = row_sum16(G.dwout[1][mt][rr]);
            float db[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) db[j] = row_sum16(G.db[j][mt][rr]);
            if (n == 0 && u < H) {
                prow[L.o_w_out + u] = w0; prow[L.o_w_out + H + u] = w1;
                if constexpr (JAN) {
                    prow[L.o_b_ih + u] = db[1]; prow[L.o_b_hh + u] = db[1];
                    prow[L.o_b_ih + H + u] = db[2]; prow[L.o_b_hh + H + u] = db[2];
                } else if constexpr (!TRES) {
                    prow[L.o_b_ih + u] = db[0]; prow[L.o_b_hh + u] = db[0];
                    prow[L.o_b_ih + H + u] = db[1]; prow[L.o_b_hh + H + u] = db[1];
                    prow[L.o_b_ih + 2 * H + u] = db[2]; prow[L.o_b_hh + 2 * H + u] = db[3];
                }
            }
        }
    if constexpr (TRES) {
        if (lane < 24) prow[L.o_tcn0 + lane] = 0.0f;      // tcn.0.weight (18) and tcn.2.weight (6) are contiguous: their gradient arrives in the rows of tres_skip_wgrad_kernel
    } else {
        const float b0 = row_sum16(G.dbout[0]), b1 = row_sum16(G.dbout[1]);
        if (n == 0 && q == 0) { prow[L.o_b_out] = b0; prow[L.o_b_out + 1] = b1; }
    }
}

template <bool TRES, int NT, bool DX, bool JAN = false>
__global__ __launch_bounds__((NT == 1 && !DX) ? 512 : 256, 1) void delta16_bwd_kernel(SeqArgs a) {
    using T = D16<NT>;
    constexpr int S = D16<NT>::S, kD16BwdCh = d16_bwd_ch<NT, DX>();
    // (the staged x carries no halo since r06: the TCN skip's gradients are time-parallel kernels of their own — CH + 1 samples per sequence)
    constexpr int kWave = 2 * 16 * (kD16BwdCh + 2) + (DX ? 2 : 1) * 2 * 16 * (kD16BwdCh + 1) + T::kTiles * kTileFloats;
    constexpr int kGroups = DX ? T::NG_DX : T::NG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwb = blockDim.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const DeltaLayout L = delta_layout(a.H, TRES, JAN ? 2 : 3);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    {
        float4* t4 = reinterpret_cast<float4*>(tab);
        for (int grp = wave; grp < kGroups; grp += nwb) t4[grp * 64 + lane] = d16_entry<TRES, NT, JAN>(pl, L, grp, n, q);
        __syncthreads();
    }
    const TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    D16Scalars<TRES> sc;
    sc.load(pl, L);
    float oh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) oh[e] = q == e ? 1.0f : 0.0f;
    float* wbase = tab + s16_tab_floats(kGroups) + (size_t)wave * kWave;
    float2* xs = reinterpret_cast<float2*>(wbase);
    float2* dys = xs + 16 * (kD16BwdCh + 2);
    float2* dxs = dys + 16 * (kD16BwdCh + 1);                   // DX only
    float* tiles = reinterpret_cast<float*>(dys + (DX ? 2 : 1) * 16 * (kD16BwdCh + 1));
    for (int i = lane; i < kTileFloats; i += 64) tiles[5 * NT * kTileFloats + i] = 0.0f;
    const float2* xr = xs + n * (kD16BwdCh + 2);
    D16Grad<TRES, NT> G;
    G.zero();
    const int nwaves = gridDim.x * nwb;
    for (int grp = blockIdx.x * nwb + wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * 16;
        const bool valid = b0 + n < a.B;
        const float4* ck = reinterpret_cast<const float4*>(a.ckpt) + (size_t)grp * a.nck * T::kCk * 64 + lane;
        const float2 x0 = valid ? reinterpret_cast<const float2*>(a.x)[(size_t)(b0 + n) * a.T] : make_float2(0.5f, 0.5f);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        D16Carry<NT> C;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) { C.gh[kt] = z4; C.ghp[kt] = z4; C.gr[kt] = z4; C.gz[kt] = z4; C.gn[kt] = z4; C.gnh[kt] = z4; }
        C.gxp[0] = C.gxp[1] = C.wrap[0] = C.wrap[1] = 0.0f;
        float* dxrow = (DX && valid) ? a.dx + (size_t)(b0 + n) * a.T * 2 : nullptr;
        int cur_chunk = -1, cur_len = 0;
        for (int blk = a.nck - 1; blk >= 0; --blk) {
            const int tb = blk * S, nstep = min(S, a.T - tb);
            const int chunk = tb / kD16BwdCh, t0 = chunk * kD16BwdCh;
            if (chunk != cur_chunk) {
                if constexpr (DX) {
                    if (cur_chunk >= 0) {
                        wave_lds_fence();
                        d16_stage_out<kD16BwdCh>(dxs, a.dx, b0, a.B, a.T, cur_chunk * kD16BwdCh, cur_len, lane);
                    }
                }
                wave_lds_fence();
                const int len = min(kD16BwdCh, a.T - t0);
                cur_len = len;
                d16_stage_x1<kD16BwdCh>(xs, a.x, b0, a.B, a.T, t0, lane);
                d16_stage_in<kD16BwdCh>(dys, a.dy, b0, a.B, a.T, t0, len, lane, make_float2(0.0f, 0.0f));
                wave_lds_fence();
                cur_chunk = chunk;
            }
            D16State<NT> st;
            d16_init_state<NT>(tl, st);                       // h = h_p = x_p = 0, accumulators = their initial values (the biases)
            const float4* c = ck + (size_t)blk * T::kCk * 64;
            const float4 aux = c[2 * NT * 64];                // (x_p0, x_p1, mask words) of this block
            unsigned mw[NT];
            mw[0] = (unsigned)aux.z;
            if constexpr (NT > 1) mw[NT - 1] = (unsigned)aux.w;
            if (blk) {
                // state at the block's start: h, h_p, x_p from the checkpoint; the accumulators rebuilt from the telescoped sums of the
                // masked deltas, dm = dm(0) + W_x x_p + W_h h_p (one step's MFMAs; x_p / h_p start at 0)
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) { st.h[kt] = as_f32x4(c[(0 * NT + kt) * 64]); st.hp[kt] = as_f32x4(c[(1 * NT + kt) * 64]); }
                st.xp[0] = aux.x; st.xp[1] = aux.y;
                d16_accumulate<NT, JAN>(opaque(tl), st.xp, st.hp, st);
            }
            if (nstep == S) d16_bwd_block<TRES, NT, true, DX, JAN>(a, tl, sc, oh, G, xr, dys, dxs, tiles, x0, n, q, tb, tb - t0, nstep, cur_len, dxrow, st, C, mw);
            else d16_bwd_block<TRES, NT, false, DX, JAN>(a, tl, sc, oh, G, xr, dys, dxs, tiles, x0, n, q, tb, tb - t0, nstep, cur_len, dxrow, st, C, mw);
        }
        if constexpr (DX) {
            wave_lds_fence();
            if (q == 0) { dxs[n * (kD16BwdCh + 1)].x += C.wrap[0]; dxs[n * (kD16BwdCh + 1)].y += C.wrap[1]; }   // roll(x, -1): step T-1 saw sample 0
            wave_lds_fence();
            d16_stage_out<kD16BwdCh>(dxs, a.dx, b0, a.B, a.T, 0, cur_len, lane);
            wave_lds_fence();
        }
        // gradient w.r.t. the initial accumulators = bias gradients (deltagru.py:165-170)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            G.db[0][kt] = add4(G.db[0][kt], C.gr[kt]); G.db[1][kt] = add4(G.db[1][kt], C.gz[kt]);
            G.db[2][kt] = add4(G.db[2][kt], C.gn[kt]); G.db[3][kt] = add4(G.db[3][kt], C.gnh[kt]);
        }
    }
    const int P4 = L.P + kLossCols;
    __syncthreads();
    d16_write_row<TRES, NT, JAN>(smem + wave * P4, L, G, lane, n, q);
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = smem[i];
        for (int wv = 1; wv < nwb; ++wv) v += smem[wv * P4 + i];
        prow[i] = v;
    }
}

// dL/dx through the TRes skip path  skip = HS(conv2(HS(conv1(x)))), conv1: k3, dilation 16, zero padding (time-parallel):
// dx[ch][t] += sum_k sum_c w1[c][ch][k] d1[c][t - 16 (k - 1)],  d1 = HS'(s1) (w2^T d2),  d2 = HS'(s2) dy
__global__ __launch_bounds__(256) void tres_skip_dx_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ params, float* __restrict__ dx, int B, int T, int H) {
    const DeltaLayout L = delta_layout(H, true);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * T) return;
    const int b = (int)(idx / T), t = (int)(idx % T);
    const float2* x2 = reinterpret_cast<const float2*>(x) + (size_t)b * T;
    const float2* d2y = reinterpret_cast<const float2*>(dy) + (size_t)b * T;
    const float* w1 = params + L.o_tcn0;
    const float* w2 = params + L.o_tcn2;
    auto at = [&](int p) { return (p >= 0 && p < T) ? x2[p] : make_float2(0.0f, 0.0f); };
    float gI = 0.0f, gQ = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int p = t - 16 * (k - 1);
        if (p < 0 || p >= T) continue;
        const float2 xm = at(p - 16), xc = at(p), xq = at(p + 16), dyp = d2y[p];
        float s1[3], s2[2] = {0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s1[c] = w1[c * 6] * xm.x + w1[c * 6 + 1] * xc.x + w1[c * 6 + 2] * xq.x + w1[c * 6 + 3] * xm.y + w1[c * 6 + 4] * xc.y + w1[c * 6 + 5] * xq.y;
            const float hs = hardswishf_(s1[c]);
            s2[0] = __builtin_fmaf(w2[c], hs, s2[0]); s2[1] = __builtin_fmaf(w2[3 + c], hs, s2[1]);
        }
        const float e0 = dyp.x * d16_hsg(s2[0]), e1 = dyp.y * d16_hsg(s2[1]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d1 = __builtin_fmaf(e0, w2[c], e1 * w2[3 + c]) * d16_hsg(s1[c]);
            gI = __builtin_fmaf(w1[c * 6 + k], d1, gI);
            gQ = __builtin_fmaf(w1[c * 6 + 3 + k], d1, gQ);
        }
    }
    float2* o = reinterpret_cast<float2*>(dx) + (size_t)b * T + t;
    const float2 cur = *o;
    *o = make_float2(cur.x + gI, cur.y + gQ);
}

// Weight gradient of the TRes skip path (tcn.0.weight: 3 x 2 x 3, tcn.2.weight: 2 x 3), time-parallel: one (sequence, step) per thread and
// grid stride, 24 accumulators per thread, a fixed-order workgroup reduction, ONE full-width partial row per workgroup (zero outside the
// 24 TCN columns) behind the rows of delta16_bwd_kernel — odpd_reduce_partials sums them with the rest.
constexpr int kTcnRows = 256, kTcnThreads = 1024;      // one workgroup per CU, sixteen waves each: ~50 samples per thread at 65 536 x 200
__global__ __launch_bounds__(kTcnThreads) void tres_skip_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ params, float* __restrict__ rows, int B, int T, int H) {
    __shared__ float red[kTcnThreads / 64][24];
    const DeltaLayout L = delta_layout(H, true);
    const float* w1 = params + L.o_tcn0;
    const float* w2 = params + L.o_tcn2;
    float acc[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) acc[i] = 0.0f;
    const long total = (long)B * T;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / T), t = (int)(idx % T);
        const float2* x2 = reinterpret_cast<const float2*>(x) + (size_t)b * T;
        const float2 dyv = reinterpret_cast<const float2*>(dy)[idx];
        const float2 z2 = make_float2(0.0f, 0.0f);
        const float2 xm = t >= 16 ? x2[t - 16] : z2, xc = x2[t], xq = t + 16 < T ? x2[t + 16] : z2;
        float s1[3], hs[3], s2[2] = {0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = w1[c * 6] * xm.x;
            v = __builtin_fmaf(w1[c * 6 + 1], xc.x, v); v = __builtin_fmaf(w1[c * 6 + 2], xq.x, v);
            v = __builtin_fmaf(w1[c * 6 + 3], xm.y, v); v = __builtin_fmaf(w1[c * 6 + 4], xc.y, v);
            s1[c] = __builtin_fmaf(w1[c * 6 + 5], xq.y, v);
            hs[c] = hardswishf_(s1[c]);
        }
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            float v = w2[o * 3] * hs[0];
            v = __builtin_fmaf(w2[o * 3 + 1], hs[1], v);
            s2[o] = __builtin_fmaf(w2[o * 3 + 2], hs[2], v);
        }
        const float d2[2] = {dyv.x * d16_hsg(s2[0]), dyv.y * d16_hsg(s2[1])};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            acc[18 + c] = __builtin_fmaf(d2[0], hs[c], acc[18 + c]);
            acc[21 + c] = __builtin_fmaf(d2[1], hs[c], acc[21 + c]);
            const float d1 = __builtin_fmaf(d2[0], w2[c], d2[1] * w2[3 + c]) * d16_hsg(s1[c]);
            acc[c * 6 + 0] = __builtin_fmaf(d1, xm.x, acc[c * 6 + 0]); acc[c * 6 + 1] = __builtin_fmaf(d1, xc.x, acc[c * 6 + 1]);
            acc[c * 6 + 2] = __builtin_fmaf(d1, xq.x, acc[c * 6 + 2]); acc[c * 6 + 3] = __builtin_fmaf(d1, xm.y, acc[c * 6 + 3]);
            acc[c * 6 + 4] = __builtin_fmaf(d1, xc.y, acc[c * 6 + 4]); acc[c * 6 + 5] = __builtin_fmaf(d1, xq.y, acc[c * 6 + 5]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        float v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);      // fixed tree: bit-repeatable
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    const int P4 = L.P + kLossCols;
    float* prow = rows + (size_t)blockIdx.x * P4;
    for (int i = threadIdx.x; i < P4; i += blockDim.x) {
        float v = 0.0f;
        const int j = i - L.o_tcn0;      // (tcn.0.weight and tcn.2.weight are adjacent in the parameter buffer)
        if (j >= 0 && j < 24) {
#pragma unroll
            for (int w = 0; w < kTcnThreads / 64; ++w) v += red[w][j];      // fixed order
        }
        prow[i] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
// hidden <= 16: from the batch size that fills the chip; hidden 17..32: always (the row-rotated delta kernels stop at 16)
bool delta_uses_s16(const odpd_model_t* m, int B) {
    if (m->backbone == ODPD_DELTAJANET) return m->hidden <= 32;               // deltajanet lives in these kernels only
    if ((m->backbone != ODPD_DELTAGRU && m->backbone != ODPD_TRES_DELTAGRU) || m->hidden > 32) return false;
    if (m->hidden > 16 || (m->flags & ODPD_FLAG_NEED_DX)) return true;      // dL/dx lives in these kernels only
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 16L * 4 * device_cus();
    return B >= min_batch;
}
// `per_cu`: workgroups of eight waves a CU holds (2 = the four-waves-per-SIMD forward of deltagru_tcnskip)
static LaunchShape d16_fwd_shape(int ngroups, int nt, int per_cu = 1) {
    LaunchShape ls;
    const int cus = device_cus();
    ls.waves = (nt > 1 || ngroups <= 4 * cus) ? 4 : 8;
    const int need = (ngroups + ls.waves - 1) / ls.waves, cap = (ls.waves == 8 ? per_cu : 1) * cus;
    ls.grid = need < cap ? need : cap;
    return ls;
}
static int d16_tiles(int H) { return (H + 15) / 16; }
static size_t d16_bwd_lds(int P, int nt, int waves, bool dx) {
    const int groups = nt == 1 ? (dx ? D16<1>::NG_DX : D16<1>::NG) : (dx ? D16<2>::NG_DX : D16<2>::NG);
    const int tiles = nt == 1 ? D16<1>::kTiles : D16<2>::kTiles;
    const int ch = (nt == 1 && !dx) ? d16_bwd_ch<1, false>() : kChunk;
    size_t lds = ((size_t)pad4(P) + s16_tab_floats(groups) +
                  (size_t)waves * (2 * 16 * (ch + 2) + (dx ? 2 : 1) * 2 * 16 * (ch + 1) + tiles * kTileFloats)) * sizeof(float);
    if (lds < reduce_scratch_bytes(P, waves)) lds = reduce_scratch_bytes(P, waves);
    return lds;
}
// waves per block: as many (<= 4) as the LDS budget holds next to the staged parameters and the operand table
static LaunchShape d16_bwd_shape(const odpd_model_t* m, int ngroups) {
    LaunchShape ls;
    const int P = delta_layout(m->hidden, m->backbone == ODPD_TRES_DELTAGRU, m->backbone == ODPD_DELTAJANET ? 2 : 3).P, nt = d16_tiles(m->hidden);
    const bool dx = (m->flags & ODPD_FLAG_NEED_DX) != 0;      // the shape (= rows of partials) is fixed by the model, not by the call
    ls.waves = (nt == 1 && !dx) ? 8 : 4;
    while (ls.waves > 1 && d16_bwd_lds(P, nt, ls.waves, dx) > kMaxLds) --ls.waves;
    const int need = (ngroups + ls.waves - 1) / ls.waves, cus = device_cus();
    ls.grid = need < cus ? need : cus;
    return ls;
}
// rows of partials: one per workgroup of the backward kernel; TRes: + the rows of the skip path's weight-gradient kernel
int delta_s16_rows(const odpd_model_t* m, int B) {
    return d16_bwd_shape(m, (B + 15) / 16).grid + (m->backbone == ODPD_TRES_DELTAGRU ? kTcnRows : 0);
}
int64_t delta_s16_ckpt_floats(const odpd_model_t* m, int B, int T) {
    const int nt = d16_tiles(m->hidden), S = nt == 1 ? D16<1>::S : D16<2>::S;
    return (int64_t)((B + 15) / 16) * ((T + S - 1) / S) * (2 * nt + 1) * 256;
}
template <bool TRES, int NT, bool JAN = false>
static int d16_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int P, int mode) {
    using T = D16<NT>;
    SeqArgs a = a0;
    a.nck = (a.T + T::S - 1) / T::S;                          // checkpoint blocks of this tile count (the generic count is for four steps)
    if (mode == 1) {
        const LaunchShape ls = d16_fwd_shape(a.ngroups, NT, D16Fwd<TRES, NT>::LEAN ? 2 : 1);
        const size_t lds = ((size_t)pad4(P) + s16_tab_floats(T::NG) + (size_t)ls.waves * D16Fwd<TRES, NT>::kWave) * sizeof(float);
        auto k = delta16_fwd_kernel<TRES, NT, JAN>;
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    }
    // the weight gradients ride along in every backward launch: partials are required (a frozen model passes a scratch buffer)
    if (a.partials == nullptr) return ODPD_EINVAL;
    if (a.ckpt == nullptr) return ODPD_EINVAL;                // every block's threshold decisions live there, also for T <= kCkptStride
    const bool dxf = (m->flags & ODPD_FLAG_NEED_DX) != 0;
    if (a.dx != nullptr && !dxf) return ODPD_EINVAL;          // the forward must have run with ODPD_FLAG_NEED_DX as well
    const LaunchShape ls = d16_bwd_shape(m, a.ngroups);
    const size_t lds = d16_bwd_lds(P, NT, ls.waves, dxf);
    if (lds > kMaxLds) return ODPD_EUNSUPPORTED;
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    };
    if (int e = a.dx == nullptr ? launch(delta16_bwd_kernel<TRES, NT, false, JAN>) : launch(delta16_bwd_kernel<TRES, NT, true, JAN>)) return e;
    if (TRES) {
        hipLaunchKernelGGL(tres_skip_wgrad_kernel, dim3(kTcnRows), dim3(kTcnThreads), 0, st, a.x, a.dy, a.params,
                           a.partials + (size_t)ls.grid * (P + kLossCols), a.B, a.T, a.H);
        if (a.dx != nullptr) {
            const long n = (long)a.B * a.T;
            hipLaunchKernelGGL(tres_skip_dx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.x, a.dy, a.params, a.dx, a.B, a.T, a.H);
        }
    }
    return (int)hipGetLastError();
}
int delta_s16_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a0, int mode) {
    SeqArgs a = a0;
    a.ngroups = (a.B + 15) / 16;
    const bool tres = m->backbone == ODPD_TRES_DELTAGRU;
    if (m->backbone == ODPD_DELTAJANET) {
        const int Pj = delta_layout(m->hidden, false, 2).P, ntj = d16_tiles(m->hidden);
        if (ntj == 1) return d16_launch<false, 1, true>(st, m, a, Pj, mode);
        if (ntj == 2) return d16_launch<false, 2, true>(st, m, a, Pj, mode);
        return ODPD_EUNSUPPORTED;
    }
    const int P = delta_layout(m->hidden, tres).P, nt = d16_tiles(m->hidden);
    if (nt == 1) return tres ? d16_launch<true, 1>(st, m, a, P, mode) : d16_launch<false, 1>(st, m, a, P, mode);
    if (nt == 2) return tres ? d16_launch<true, 2>(st, m, a, P, mode) : d16_launch<false, 2>(st, m, a, P, mode);
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
