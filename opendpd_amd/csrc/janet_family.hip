// janet_family.hip — persistent-RNN kernels for the PGJANET backbone (backbones/pgjanet.py:5-84).
// Per step (pgjanet.py:33-72):  amp = |x|, (cos,sin) = (I,Q)/amp  [the reference takes cos/sin of atan2(Q,I)],
//   a = tanh(W_a [h,amp] + b), p1 = tanh(W_p1 [h,cos] + b), p2 = tanh(W_p2 [h,sin] + b),
//   u = a p1 p2 (1-a)(1-p1)(1-p2),  f = s(W_f [h,u] + b), g = tanh(W_g [h,u] + b),  h' = f h + (1-f) g,  y = W_o h' + b.
// One 16-lane row per sequence (H <= 16).  The seven HxH blocks (W_a,W_p1,W_p2,W_f,W_g acting on h; W_f,W_g acting on u)
// and their transposes live in LDS rotated-quad tables and are streamed per use (ds_read_b128 + 4 DPP FMAs),
// so no weight matrix is pinned in registers.  BPTT: checkpoint of h every kCkptStride steps + block recompute;
// weight gradients of the HxH blocks by exact-fp32 MFMA; dL/dx (frozen PA of a cascade) through the three scalar
// input columns and the polar features.
#include "odpd_seq.h"
#include "odpd_s16.h"

namespace odpd {

// table rows: 0 a_h, 1 p1_h, 2 p2_h, 3 f_h, 4 g_h, 5 f_u, 6 g_u; +7 = transposed
constexpr int kJRows = 14, kJTabFloats = kJRows * 4 * 64 * 4;

template <bool WITH_T>
__device__ __forceinline__ void fill_janet_tabs(float* tab, const float* pl, const JanetLayout& L, int lane, int wave, int nwb) {
    const int H = L.H, col = lane & 15, o = col, dir = rot_dir(col);
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int idx = wave; idx < kJRows * 4; idx += nwb) {
        const int tr = idx >> 2, q = idx & 3;
        const bool transposed = tr >= 7;
        if (!WITH_T && transposed) continue;
        const int r = transposed ? tr - 7 : tr;
        // block r: base offset, row stride, column offset inside the reference weight
        const int base = r == 0 ? L.o_wa : r == 1 ? L.o_wp1 : r == 2 ? L.o_wp2 : (r == 3 || r == 5) ? L.o_wf : L.o_wg;
        const int ld = r < 3 ? H + 1 : 2 * H, coff = r >= 5 ? H : 0;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = (col + dir * (4 * q + e)) & 15;
            const bool ok = o < H && m < H;
            v[e] = ok ? pl[base + (transposed ? m * ld + coff + o : o * ld + coff + m)] : 0.0f;
        }
        t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
}

struct JanetW { float sa, sp1, sp2, ba, bp1, bp2, bf, bg, wo[2], bo[2]; };   // scalar-input columns, biases, head
__device__ __forceinline__ void load_janet_w(JanetW& w, const float* pl, const JanetLayout& L, int col) {
    const int H = L.H, o = col;
    const bool vo = o < H;
    w.sa = vo ? pl[L.o_wa + o * (H + 1) + H] : 0.f; w.sp1 = vo ? pl[L.o_wp1 + o * (H + 1) + H] : 0.f;
    w.sp2 = vo ? pl[L.o_wp2 + o * (H + 1) + H] : 0.f;
    w.ba = vo ? pl[L.o_ba + o] : 0.f; w.bp1 = vo ? pl[L.o_bp1 + o] : 0.f; w.bp2 = vo ? pl[L.o_bp2 + o] : 0.f;
    w.bf = vo ? pl[L.o_bf + o] : 0.f; w.bg = vo ? pl[L.o_bg + o] : 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        w.wo[c] = vo ? pl[L.o_wo + c * H + o] : 0.f;
        w.bo[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, pl[L.o_bo + c])));
    }
}

__device__ __forceinline__ void janet_inputs(float2 xv, float& amp, float& ct, float& st) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y);
    amp = __builtin_amdgcn_sqrtf(a2);
    const float ia = fast_rcp(amp);
    ct = xv.x * ia; st = xv.y * ia;
}

// forward step; padded lanes (o >= H) have all-zero weights: a=p1=p2=0 -> u=0, f=0.5, g=0 -> h stays 0
__device__ __forceinline__ void janet_cell_fwd(const JanetW& w, TabPtr tlane, float amp, float ct, float st, float& h,
                                               float& an, float& p1, float& p2, float& u, float& f, float& g) {
    an = tanhf_(tab_rotdot<1>(__builtin_fmaf(w.sa, amp, w.ba), tlane, 0, h));
    p1 = tanhf_(tab_rotdot<1>(__builtin_fmaf(w.sp1, ct, w.bp1), tlane, 1, h));
    p2 = tanhf_(tab_rotdot<1>(__builtin_fmaf(w.sp2, st, w.bp2), tlane, 2, h));
    u = (an * p1 * p2) * ((1.0f - an) * (1.0f - p1) * (1.0f - p2));
    float pf = tab_rotdot<1>(w.bf, tlane, 3, h), pg = tab_rotdot<1>(w.bg, tlane, 4, h);
    pf = tab_rotdot<1>(pf, tlane, 5, u); pg = tab_rotdot<1>(pg, tlane, 6, u);
    f = sigmoidf_(pf); g = tanhf_(pg);
    h = __builtin_fmaf(f, h - g, g);   // f h + (1 - f) g
}

// -------------------------------------------------------------------------------------------------
template <int dummy = 0>
__global__ __launch_bounds__(kMaxThreads) void janet_fwd_kernel(SeqArgs a) {
    constexpr int SPW = 4, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<1>();
    const int lane = id.lane, col = id.col, s = id.s;
    const JanetLayout L = janet_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_janet_tabs<false>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + kJTabFloats) + id.wave * (2 * SPW * kChunkPad);
    float2* ys = xs + SPW * kChunkPad;
    JanetW w;
    load_janet_w(w, pl, L, col);
    const int nwaves = gridDim.x * id.nwb;
    for (int grp = blockIdx.x * id.nwb + id.wave; grp < a.ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        float h = 0.0f;
        for (int t0 = 0; t0 < a.T; t0 += kChunk) {
            const int len = min(kChunk, a.T - t0);
            stage_in<SPW>(xs, a.x, b0, a.B, a.T, t0, len, lane, make_float2(0.5f, 0.5f));
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                float amp, ct, st, an, p1, p2, u, f, g;
                janet_inputs(xs[s * kChunkPad + tt], amp, ct, st);
                janet_cell_fwd(w, opaque(tlane), amp, ct, st, h, an, p1, p2, u, f, g);
                const float y0 = row_sum16(w.wo[0] * h) + w.bo[0], y1 = row_sum16(w.wo[1] * h) + w.bo[1];
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

// -------------------------------------------------------------------------------------------------
// evaluation kernel (net_eval / run_dpd on a few very long sequences, train_funcs.py:57-90): ONE sequence per wave, the seven H x H
// products of a step in two rounds of one register-resident rotated dot product per 16-lane row instead of seven streamed from LDS.
// Round A on h: rows a | p1 | p2 | f (its h half); round B: row 0 continues f on u, row 1 g on u, row 2 g on h.  Two three-swap gathers
// hand every row (a, p1, p2, f_h) and then (f, g_u, g_h); h' is updated redundantly on all rows.  Same arithmetic per element as janet_cell_fwd apart from g's pre-activation, summed as (b + W_gh h) + W_gu u.
// The step keeps only the recurrence: |x|, cos, sin of a 64-step chunk are computed with lane = time step and parked in LDS, a step
// parks h, and fc_out of the chunk follows, one time step per lane.
// -------------------------------------------------------------------------------------------------
constexpr int kJEvalHistStride = 64 + 4;
template <bool CK>      // CK: also writes the BPTT checkpoints (the forward of the split train path)
__global__ __launch_bounds__(64) void janet_eval_kernel(SeqArgs a) {
    constexpr int EC = kEvalChunk, HS = kJEvalHistStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;
    const JanetLayout L = janet_layout(a.H);
    const int H = L.H;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_janet_tabs<false>(tab, pl, L, lane, 0, 1);
    float* ftab = tab + kJTabFloats;                   // [EC][4]: |x|, cos, sin of time t0 + i
    float* hist = ftab + EC * 4;                       // [EC][HS]: h of time t0 + i, every lane's copy
    float* hw = hist + EC * HS;                        // fc_out [2][16], zero padded
    if (lane < 32) hw[lane] = (lane & 15) < H ? pl[L.o_wo + (lane >> 4) * H + (lane & 15)] : 0.0f;
    wave_lds_fence();
    TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const bool vo = col < H;
    float wa[16], wb[16];
    load_rot(wa, tl + role * 4 * 64);                                           // a_h | p1_h | p2_h | f_h
    load_rot(wb, tl + (role == 0 ? 5 : role == 1 ? 6 : 4) * 4 * 64);            // f_u | g_u | g_h | -
    if (role == 3) {
#pragma unroll
        for (int k = 0; k < 16; ++k) wb[k] = 0.0f;
    }
    const int o_w = role == 0 ? L.o_wa : role == 1 ? L.o_wp1 : L.o_wp2, o_b = role == 0 ? L.o_ba : role == 1 ? L.o_bp1 : role == 2 ? L.o_bp2 : L.o_bf;
    const float sw = (vo && role < 3) ? pl[o_w + col * (H + 1) + H] : 0.0f;    // the row's scalar-input column
    const float ba = vo ? pl[o_b + col] : 0.0f;
    const float bg = (vo && role == 2) ? pl[L.o_bg + col] : 0.0f;
    const int fsel = role < 3 ? role : 2;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        float h = 0.0f;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * a.T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * a.T;
        // the samples of a chunk are fetched while the previous one is stepped (lane = time step)
        float2 raw = lane < a.T ? xg[lane] : make_float2(0.5f, 0.5f);
        for (int t0 = 0; t0 < a.T; t0 += EC) {
            const int len = min(EC, a.T - t0);
            {
                float amp, ct, st;
                janet_inputs(raw, amp, ct, st);
                wave_lds_fence();
                reinterpret_cast<float4*>(ftab)[lane] = make_float4(amp, ct, st, 0.0f);
                wave_lds_fence();
            }
            raw = t0 + EC + lane < a.T ? xg[t0 + EC + lane] : make_float2(0.5f, 0.5f);
            for (int tt = 0; tt < len; ++tt) {
                const float sc = ftab[tt * 4 + fsel];                           // the row's scalar input: |x| | cos | sin
                const float pa = rotdot(__builtin_fmaf(sw, sc, ba), wa, h);
                float g4[4];
                gather_rows(role == 3 ? pa : tanhf_(pa), g4);
                const float an = g4[0], p1 = g4[1], p2 = g4[2];
                const float u = (an * p1 * p2) * ((1.0f - an) * (1.0f - p1) * (1.0f - p2));
                const float pb = rotdot(role == 0 ? g4[3] : bg, wb, role < 2 ? u : h);
                gather_rows(pb, g4);
                const float f = sigmoidf_(g4[0]), g = tanhf_(g4[2] + g4[1]);
                h = __builtin_fmaf(f, h - g, g);
                hist[tt * HS + lane] = h;
                if constexpr (CK) {                  // BPTT checkpoints in the layout of the row-rotated backward (lane = 16 s + col)
                    const int t1 = t0 + tt + 1;
                    if ((t1 % kCkptStride) == 0 && t1 < a.T && role == 0)
                        a.ckpt[((size_t)(b >> 2) * a.nck + t1 / kCkptStride) * 64 + 16 * (b & 3) + col] = h;
                }
            }
            wave_lds_fence();
            // fc_out of the chunk, lane = time step
            if (lane < len) {
                const float4* hv4 = reinterpret_cast<const float4*>(hist + lane * HS);
                const float4* hw4 = reinterpret_cast<const float4*>(hw);
                float y0 = pl[L.o_bo], y1 = pl[L.o_bo + 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 hv = hv4[q], w0 = hw4[q], w1 = hw4[4 + q];
                    y0 = __builtin_fmaf(w0.x, hv.x, y0); y0 = __builtin_fmaf(w0.y, hv.y, y0); y0 = __builtin_fmaf(w0.z, hv.z, y0); y0 = __builtin_fmaf(w0.w, hv.w, y0);
                    y1 = __builtin_fmaf(w1.x, hv.x, y1); y1 = __builtin_fmaf(w1.y, hv.y, y1); y1 = __builtin_fmaf(w1.z, hv.z, y1); y1 = __builtin_fmaf(w1.w, hv.w, y1);
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
}

// -------------------------------------------------------------------------------------------------
// Gate-parallel fused train kernel for the reference's own batch sizes (train_funcs.py:28-48; a wave is alone on its SIMD there): ONE
// sequence per wave (one wave per workgroup), only the recurrence in the step loops.
//   forward   as janet_eval_kernel (two rounds of one rotated dot product per row); h(t), (a, p1, p2, f) and (g, u) of every step are parked in LDS;
//   head      fc_out, the loss and dL/dy of all T steps with lane = time step;
//   backward  two rounds with the transposed weights: rows (W_fu^T d_f | W_gu^T d_g | W_gh^T d_g | W_fh^T d_f) — the first pair sums to dL/du, the
//             second to h(t-1)'s share — then rows (W_a^T d_a | W_p1^T d_p1 | W_p2^T d_p2 | -); the weight gradients of a step are TWO 4-block
//             MFMAs (v_mfma_f32_16x16x1_4b_f32): (d_f | d_g | d_g | d_f) x (u | u | h(t-1) | h(t-1)) and (d_a | d_p1 | d_p2 | 0) x h(t-1).
// One partial-gradient row per workgroup.  Taken while the frame's parked state fits the CU's LDS share.
// -------------------------------------------------------------------------------------------------
__host__ __device__ inline int janet_gp_buffer_floats(int T) {
    const int Tp = (T + 63) & ~63;
    const int buf = Tp * 4 + (Tp + 1) * 16 + Tp * 64 + Tp * 32 + Tp * 2 + 256 + 32;
    return buf > kJTabFloats ? buf : kJTabFloats;
}
__global__ __launch_bounds__(64) void janet_gp_train_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, role = lane >> 4;
    const JanetLayout L = janet_layout(a.H);
    const int H = L.H, T = a.T, Tp = (T + 63) & ~63;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_janet_tabs<true>(tab, pl, L, lane, 0, 1);
    TabPtr tl = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    const bool vo = col < H;
    // forward: round A a_h | p1_h | p2_h | f_h, round B f_u | g_u | g_h | -; backward (transposed): round B' f_u | g_u | g_h | f_h, round A' a_h | p1_h | p2_h | -
    float wa[16], wb[16], wtb[16], wta[16];
    load_rot(wa, tl + role * 4 * 64);
    load_rot(wb, tl + (role == 0 ? 5 : role == 1 ? 6 : 4) * 4 * 64);
    load_rot(wtb, tl + (7 + (role == 0 ? 5 : role == 1 ? 6 : role == 2 ? 4 : 3)) * 4 * 64);
    load_rot(wta, tl + (7 + (role < 3 ? role : 0)) * 4 * 64);
    if (role == 3) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { wb[k] = 0.0f; wta[k] = 0.0f; }
    }
    const int o_w = role == 0 ? L.o_wa : role == 1 ? L.o_wp1 : L.o_wp2, o_b = role == 0 ? L.o_ba : role == 1 ? L.o_bp1 : role == 2 ? L.o_bp2 : L.o_bf;
    const float sw = (vo && role < 3) ? pl[o_w + col * (H + 1) + H] : 0.0f;    // the row's scalar-input column
    const float ba = vo ? pl[o_b + col] : 0.0f;
    const float bg = (vo && role == 2) ? pl[L.o_bg + col] : 0.0f;
    const float wo0 = vo ? pl[L.o_wo + col] : 0.0f, wo1 = vo ? pl[L.o_wo + H + col] : 0.0f;
    const float bo0 = pl[L.o_bo], bo1 = pl[L.o_bo + 1];
    const int fsel = role < 3 ? role : 2;
    wave_lds_fence();
    // per-time buffers over the tables
    float* ftab = tab;                                  // [Tp][4]   |x|, cos, sin of step t
    float* hist = ftab + Tp * 4;                        // [Tp + 1][16]   entry t + 1 = h(t), entry 0 = 0
    float* gpk = hist + (Tp + 1) * 16;                  // [Tp][16][4]   a, p1, p2, f of step t
    float* gu = gpk + Tp * 64;                          // [Tp][16][2]   g, u of step t
    float* dyb = gu + Tp * 32;                          // [Tp][2]   dL/dy(t)
    float* dump = dyb + Tp * 2;                         // [256]
    float* hw = dump + 256;                             // fc_out [2][16], zero padded
    if (lane < 32) hw[lane] = (lane & 15) < H ? pl[L.o_wo + (lane >> 4) * H + (lane & 15)] : 0.0f;
    if (lane < 16) hist[lane] = 0.0f;
    const RowMasks rm = row_masks();
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    // per-step stores of the forward pass: row 0 parks (a, p1, p2, f), row 1 (g, u), row 2 h(t); the other rows hit the dump
    const int dmp = (int)(dump - smem);
    const int p4_0 = role == 0 ? (int)(gpk - smem) + 4 * col : dmp + 4 * lane, p4_step = role == 0 ? 64 : 0;
    const int p2_0 = role == 1 ? (int)(gu - smem) + 2 * col : dmp + 2 * lane, p2_step = role == 1 ? 32 : 0;
    const int p1_0 = role == 2 ? (int)(hist - smem) + 16 + col : dmp + lane, p1_step = role == 2 ? 16 : 0;

    f32x16 acc1, acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc1[i] = 0.0f; acc2[i] = 0.0f; }
    float dsc = 0.0f, db1 = 0.0f, dbg = 0.0f, dwo0 = 0.0f, dwo1 = 0.0f, dbo0 = 0.0f, dbo1 = 0.0f, loss_acc = 0.0f;

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        // ---- forward ----
        {
            float h = 0.0f;
            int q4 = p4_0, q2 = p2_0, q1 = p1_0;
            float2 raw = lane < T ? xg[lane] : make_float2(0.5f, 0.5f);
            for (int t0 = 0; t0 < T; t0 += kEvalChunk) {
                const int len = min(kEvalChunk, T - t0);
                {
                    float amp, ct, st;
                    janet_inputs(raw, amp, ct, st);
                    wave_lds_fence();
                    reinterpret_cast<float4*>(ftab)[t0 + lane] = make_float4(amp, ct, st, 0.0f);
                    wave_lds_fence();
                }
                raw = t0 + kEvalChunk + lane < T ? xg[t0 + kEvalChunk + lane] : make_float2(0.5f, 0.5f);
                for (int tt = 0; tt < len; ++tt) {
                    const float sc = ftab[(t0 + tt) * 4 + fsel];                  // the row's scalar input: |x| | cos | sin
                    const float pa = rotdot(__builtin_fmaf(sw, sc, ba), wa, h);
                    float g4[4];
                    gather_rows(role == 3 ? pa : tanhf_(pa), g4);
                    const float an = g4[0], p1 = g4[1], p2 = g4[2];
                    const float u = (an * p1 * p2) * ((1.0f - an) * (1.0f - p1) * (1.0f - p2));
                    const float pb = rotdot(role == 0 ? g4[3] : bg, wb, role < 2 ? u : h);
                    gather_rows(pb, g4);
                    const float f = sigmoidf_(g4[0]), g = tanhf_(g4[2] + g4[1]);
                    h = __builtin_fmaf(f, h - g, g);
                    *reinterpret_cast<float4*>(smem + q4) = make_float4(an, p1, p2, f);
                    *reinterpret_cast<float2*>(smem + q2) = make_float2(g, u);
                    smem[q1] = h;
                    q4 += p4_step; q2 += p2_step; q1 += p1_step;
                }
            }
            wave_lds_fence();
        }
        // ---- fc_out, loss and dL/dy of every step, lane = time step ----
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            if (t < T) {
                const float4* hv4 = reinterpret_cast<const float4*>(hist + (t + 1) * 16);
                const float4* hw4 = reinterpret_cast<const float4*>(hw);
                float y0 = bo0, y1 = bo1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 hv = hv4[q], w0 = hw4[q], w1 = hw4[4 + q];
                    y0 = __builtin_fmaf(w0.x, hv.x, y0); y0 = __builtin_fmaf(w0.y, hv.y, y0); y0 = __builtin_fmaf(w0.z, hv.z, y0); y0 = __builtin_fmaf(w0.w, hv.w, y0);
                    y1 = __builtin_fmaf(w1.x, hv.x, y1); y1 = __builtin_fmaf(w1.y, hv.y, y1); y1 = __builtin_fmaf(w1.z, hv.z, y1); y1 = __builtin_fmaf(w1.w, hv.w, y1);
                }
                const float2 tv = tg[t];
                float dy0, dy1;
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
                dbo0 += dy0; dbo1 += dy1;
                *reinterpret_cast<float2*>(dyb + 2 * t) = make_float2(dy0, dy1);
            }
        }
        wave_lds_fence();
        // ---- backward ----
        {
            float carry = 0.0f;
            for (int t = T - 1; t >= 0; --t) {
                const float hp = hist[t * 16 + col], ht = hist[(t + 1) * 16 + col];
                const float4 ga = reinterpret_cast<const float4*>(gpk)[t * 16 + col];
                const float2 gg = reinterpret_cast<const float2*>(gu)[t * 16 + col];
                const float2 dyv = *reinterpret_cast<const float2*>(dyb + 2 * t);
                const float sc = ftab[t * 4 + fsel];
                const float an = ga.x, p1 = ga.y, p2 = ga.z, f = ga.w, g = gg.x, u = gg.y;
                const float dht = carry + __builtin_fmaf(dyv.x, wo0, dyv.y * wo1);
                dwo0 = __builtin_fmaf(dyv.x, ht, dwo0); dwo1 = __builtin_fmaf(dyv.y, ht, dwo1);
                const float dfp = (dht * (hp - g)) * (f * (1.0f - f));
                const float dgp = (dht * (1.0f - f)) * __builtin_fmaf(-g, g, 1.0f);
                // round B': rows 0 / 1 sum to dL/du, rows 2 / 3 to h(t-1)'s share through W_gh, W_fh
                const float dsel = vsel(rm.m[0] | rm.m[3], dfp, dgp);
                float pb = rotdot(0.0f, wtb, dsel);
                const RowDup pb2 = dup16(pb);
                const HalfDup ps = dup32(pb2.even + pb2.odd);                      // rows 0, 1 sum to du | rows 2, 3 to the h share
                const float du = ps.lo, dhb = ps.hi;
                const float Aa = an * (1.0f - an), Ab = p1 * (1.0f - p1), Ac = p2 * (1.0f - p2);
                const float dap = (du * (1.0f - 2.0f * an) * Ab * Ac) * __builtin_fmaf(-an, an, 1.0f);
                const float dbp = (du * Aa * (1.0f - 2.0f * p1) * Ac) * __builtin_fmaf(-p1, p1, 1.0f);
                const float dcp = (du * Aa * Ab * (1.0f - 2.0f * p2)) * __builtin_fmaf(-p2, p2, 1.0f);
                const float d_a = vsel(rm.m[0], dap, vsel(rm.m[1], dbp, vsel(rm.m[2], dcp, 0.0f)));
                float pa = rotdot(0.0f, wta, d_a);
                pa = sum_rows4(pa);
                carry = __builtin_fmaf(dht, f, dhb) + pa;
                // weight gradients
                acc1 = __builtin_amdgcn_mfma_f32_16x16x1f32(dsel, vsel(rm.m[0] | rm.m[1], u, hp), acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(d_a, hp, acc2, 0, 0, 0);
                dsc = __builtin_fmaf(d_a, sc, dsc);                               // rows 0..2: the scalar-input column of W_a / W_p1 / W_p2
                db1 += vsel(rm.m[3], dfp, d_a);                                    // rows 0..2: b_a / b_p1 / b_p2, row 3: b_f
                dbg += dgp;
            }
        }
        wave_lds_fence();
    }
    // ---- the workgroup's row of partial gradients (every entry written) ----
    float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
    float lp = loss_acc, t0s = dbo0, t1s = dbo1;
    for (int o = 32; o > 0; o >>= 1) { lp += __shfl_xor(lp, o); t0s += __shfl_xor(t0s, o); t1s += __shfl_xor(t1s, o); }
    if (vo) {
        if (role < 3) { prow[o_w + col * (H + 1) + H] = dsc; prow[o_b + col] = db1; }
        else prow[L.o_bf + col] = db1;
        if (role == 0) { prow[L.o_bg + col] = dbg; prow[L.o_wo + col] = dwo0; prow[L.o_wo + H + col] = dwo1; }
    }
    if (lane == 0) {
        prow[L.o_bo] = t0s; prow[L.o_bo + 1] = t1s;
        prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
    }
    // MFMA 1 blocks: W_f[:, H + j] | W_g[:, H + j] | W_g[:, j] | W_f[:, j]; MFMA 2 blocks: W_a | W_p1 | W_p2 | -
    // register 4 blk + rr of lane l = entry (4 (l / 16) + rr, l % 16) of the block
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * role + rr;
            if (i < H && col < H) {
                const int b1 = (blk == 0 || blk == 3) ? L.o_wf : L.o_wg, coff = blk < 2 ? H : 0;
                prow[b1 + i * 2 * H + coff + col] = acc1[4 * blk + rr];
                if (blk < 3) prow[(blk == 0 ? L.o_wa : blk == 1 ? L.o_wp1 : L.o_wp2) + i * (H + 1) + col] = acc2[4 * blk + rr];
            }
        }
}

struct JanetGrad {
    f32x4 t[7];                 // dW blocks in table order (a_h, p1_h, p2_h, f_h, g_h, f_u, g_u)
    float ds[3], db[5];         // scalar-input columns (amp, cos, sin) and biases (a, p1, p2, f, g)
    float dwo[2], dbo[2];
    __device__ __forceinline__ void zero() {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 7; ++i) t[i] = z4;
        ds[0] = ds[1] = ds[2] = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) db[i] = 0.f;
        dwo[0] = dwo[1] = dbo[0] = dbo[1] = 0.f;
    }
};

template <bool NW, bool DX, bool FULL>
__device__ __forceinline__ void janet_bwd_block(const SeqArgs& a, const JanetW& w, TabPtr tlane, JanetGrad& G,
                                                const LaneId& id, const float2* xs, const float2* dys, float2* dxs, int tloc,
                                                int nstep, float h, float& dh) {
    constexpr int S = kCkptStride;
    const int s = id.s;
    float hp_s[S], an_s[S], p1_s[S], p2_s[S], u_s[S], f_s[S], g_s[S];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        if (FULL || i < nstep) {
            float amp, ct, st;
            janet_inputs(xs[s * kChunkPad + tloc + i], amp, ct, st);
            hp_s[i] = h;
            janet_cell_fwd(w, opaque(tlane), amp, ct, st, h, an_s[i], p1_s[i], p2_s[i], u_s[i], f_s[i], g_s[i]);
        }
    }
#pragma unroll
    for (int i = S - 1; i >= 0; --i) {
        if (FULL || i < nstep) {
            const int tt = tloc + i;
            const float2 dyv = dys[s * kChunkPad + tt];
            float amp, ct, st;
            janet_inputs(xs[s * kChunkPad + tt], amp, ct, st);
            const float hp = hp_s[i], f = f_s[i], g = g_s[i], u = u_s[i];
            const float ht = __builtin_fmaf(f, hp - g, g);
            const float dht = dh + __builtin_fmaf(dyv.x, w.wo[0], dyv.y * w.wo[1]);
            const float dfp = (dht * (hp - g)) * (f * (1.0f - f));
            const float dgp = (dht * (1.0f - f)) * __builtin_fmaf(-g, g, 1.0f);
            if constexpr (NW) {
                G.dwo[0] = __builtin_fmaf(dyv.x, ht, G.dwo[0]); G.dwo[1] = __builtin_fmaf(dyv.y, ht, G.dwo[1]);
                G.dbo[0] += dyv.x; G.dbo[1] += dyv.y;
                G.db[3] += dfp; G.db[4] += dgp;
                G.t[3] = mfma4(dfp, hp, G.t[3]); G.t[4] = mfma4(dgp, hp, G.t[4]);
                G.t[5] = mfma4(dfp, u, G.t[5]); G.t[6] = mfma4(dgp, u, G.t[6]);
            }
            TabPtr tl = opaque(tlane);
            float dhp = tab_rotdot<1>(dht * f, tl, 7 + 3, dfp);
            dhp = tab_rotdot<1>(dhp, tl, 7 + 4, dgp);
            float du = tab_rotdot<1>(0.0f, tl, 7 + 5, dfp);
            du = tab_rotdot<1>(du, tl, 7 + 6, dgp);
            const float an = an_s[i], p1 = p1_s[i], p2 = p2_s[i];
            const float Aa = an * (1.0f - an), Ab = p1 * (1.0f - p1), Ac = p2 * (1.0f - p2);
            const float dap = (du * (1.0f - 2.0f * an) * Ab * Ac) * __builtin_fmaf(-an, an, 1.0f);
            const float dbp = (du * Aa * (1.0f - 2.0f * p1) * Ac) * __builtin_fmaf(-p1, p1, 1.0f);
            const float dcp = (du * Aa * Ab * (1.0f - 2.0f * p2)) * __builtin_fmaf(-p2, p2, 1.0f);
            if constexpr (NW) {
                G.db[0] += dap; G.db[1] += dbp; G.db[2] += dcp;
                G.ds[0] = __builtin_fmaf(dap, amp, G.ds[0]); G.ds[1] = __builtin_fmaf(dbp, ct, G.ds[1]);
                G.ds[2] = __builtin_fmaf(dcp, st, G.ds[2]);
                G.t[0] = mfma4(dap, hp, G.t[0]); G.t[1] = mfma4(dbp, hp, G.t[1]); G.t[2] = mfma4(dcp, hp, G.t[2]);
            }
            if constexpr (DX) {
                const float2 gx = polar_sample_bwd(amp, ct, st, row_sum16(w.sa * dap), row_sum16(w.sp1 * dbp),
                                                   row_sum16(w.sp2 * dcp));
                if (id.col == 0) dxs[s * kChunkPad + tt] = gx;
            }
            dhp = tab_rotdot<1>(dhp, tl, 7 + 0, dap);
            dhp = tab_rotdot<1>(dhp, tl, 7 + 1, dbp);
            dhp = tab_rotdot<1>(dhp, tl, 7 + 2, dcp);
            dh = dhp;
        }
    }
}

__device__ __forceinline__ void janet_write_partials(float* prow, const JanetLayout& L, JanetGrad& G, int lane, int col) {
    const int H = L.H, o = col, seq = lane >> 4, g4 = lane >> 4, c = lane & 15;
    for (int i = lane; i < kLossCols; i += 64) prow[L.P + i] = 0.f;
    const int base[7] = {L.o_wa, L.o_wp1, L.o_wp2, L.o_wf, L.o_wg, L.o_wf, L.o_wg};
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const int ld = r < 3 ? H + 1 : 2 * H, coff = r >= 5 ? H : 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int i = 4 * g4 + rr;
            if (i < H && c < H) prow[base[r] + i * ld + coff + c] = G.t[r][rr];
        }
    }
    float ds[3], db[5];
#pragma unroll
    for (int k = 0; k < 3; ++k) ds[k] = across_seqs<1>(G.ds[k]);
#pragma unroll
    for (int k = 0; k < 5; ++k) db[k] = across_seqs<1>(G.db[k]);
    const float w0 = across_seqs<1>(G.dwo[0]), w1 = across_seqs<1>(G.dwo[1]);
    const float b0 = across_seqs<1>(G.dbo[0]), b1 = across_seqs<1>(G.dbo[1]);
    if (seq == 0 && o < H) {
        prow[L.o_wa + o * (H + 1) + H] = ds[0]; prow[L.o_wp1 + o * (H + 1) + H] = ds[1]; prow[L.o_wp2 + o * (H + 1) + H] = ds[2];
        prow[L.o_ba + o] = db[0]; prow[L.o_bp1 + o] = db[1]; prow[L.o_bp2 + o] = db[2];
        prow[L.o_bf + o] = db[3]; prow[L.o_bg + o] = db[4];
        prow[L.o_wo + o] = w0; prow[L.o_wo + H + o] = w1;
    }
    if (lane == 0) { prow[L.o_bo] = b0; prow[L.o_bo + 1] = b1; }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(kMaxThreads, 2) void janet_bwd_kernel(SeqArgs a) {
    constexpr int SPW = 4, S = kCkptStride;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const LaneId id = lane_id<1>();
    const int lane = id.lane, col = id.col;
    const JanetLayout L = janet_layout(a.H);
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* tab = smem + pad4(L.P);
    fill_janet_tabs<true>(tab, pl, L, lane, id.wave, id.nwb);
    TabPtr tlane = to_tab(reinterpret_cast<const float4*>(tab) + lane);
    float2* xs = reinterpret_cast<float2*>(tab + kJTabFloats) + id.wave * (3 * SPW * kChunkPad);
    float2* dys = xs + SPW * kChunkPad;
    float2* dxs = dys + SPW * kChunkPad;
    JanetW w;
    load_janet_w(w, pl, L, col);
    JanetGrad G;
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
            if (nstep == S) janet_bwd_block<NW, DX, true>(a, w, tlane, G, id, xs, dys, dxs, tb - t0, nstep, h0, dh);
            else janet_bwd_block<NW, DX, false>(a, w, tlane, G, id, xs, dys, dxs, tb - t0, nstep, h0, dh);
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
    if constexpr (NW) {
        const int P4 = L.P + kLossCols;
        __syncthreads();
        janet_write_partials(smem + id.wave * P4, L, G, lane, col);
        __syncthreads();
        float* prow = a.partials + (size_t)blockIdx.x * P4;
        for (int i = threadIdx.x; i < P4; i += blockDim.x) {
            float v = smem[i];
            for (int wv = 1; wv < id.nwb; ++wv) v += smem[wv * P4 + i];
            prow[i] = v;
        }
    }
}

static size_t janet_lds_bytes(int P, int waves, bool reduce) {
    size_t n = ((size_t)pad4(P) + kJTabFloats + (size_t)waves * 3 * (2 * 4 * kChunkPad)) * sizeof(float);
    if (reduce && n < reduce_scratch_bytes(P, waves)) n = reduce_scratch_bytes(P, waves);
    return n;
}
static LaunchShape janet_bwd_shape(int ngroups) { return persistent_shape(ngroups, 8, 8); }

// the gate-parallel fused train kernel: one sequence per single-wave workgroup, the frame's parked state in LDS
static size_t janet_gp_lds_bytes(int P, int T) { return ((size_t)pad4(P) + janet_gp_buffer_floats(T)) * sizeof(float); }
static int janet_gp_blocks_per_cu(int P, int T) {
    const size_t lds = janet_gp_lds_bytes(P, T);
    const int n = lds > kMaxLds ? 0 : (int)(kMaxLds / lds);
    return n < 4 ? n : 4;
}
bool janet_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (m->backbone != ODPD_PGJANET || m->hidden > 16 || janet_uses_s16(m, B)) return false;
    const int P = janet_layout(m->hidden).P;
    const long max_batch = tuning().gp_max_batch;
    if (max_batch >= 0) return B <= max_batch && janet_gp_blocks_per_cu(P, T) > 0;
    // up to two rounds of workgroups: the alternative is the forward / loss / backward chain of the row-rotated kernels
    return (long)B <= 2L * device_cus() * janet_gp_blocks_per_cu(P, T);
}
int janet_gp_rows(const odpd_model_t* m, int B, int T) {
    const int P = janet_layout(m->hidden).P;
    const long cap = (long)device_cus() * (kMaxLds / janet_gp_lds_bytes(P, T));
    return B < cap ? B : (int)cap;
}
int janet_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const int P = janet_layout(m->hidden).P;
    const size_t lds = janet_gp_lds_bytes(P, a.T);
    if (int e = allow_big_lds(janet_gp_train_kernel, lds)) return e;
    hipLaunchKernelGGL(janet_gp_train_kernel, dim3(janet_gp_rows(m, a.B, a.T)), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
int janet_family_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    if (janet_uses_s16(m, a.B)) return janet_s16_launch(st, m, a, 1);
    const int P = janet_layout(m->hidden).P;
    if (a.B <= 2 * device_cus() && tuning().s16_min_batch != 0 && tuning().gp_max_batch != 0) {     // sequences that each get a SIMD of their own (inference / checkpoint-writing forward)
        const size_t lds = ((size_t)pad4(P) + kJTabFloats + kEvalChunk * 4 + kEvalChunk * kJEvalHistStride + 32) * sizeof(float);
        auto launch = [&](auto k) {
            if (int e = allow_big_lds(k, lds)) return e;
            hipLaunchKernelGGL(k, dim3(a.B), dim3(64), lds, st, a);
            return (int)hipGetLastError();
        };
        return a.ckpt ? launch(janet_eval_kernel<true>) : launch(janet_eval_kernel<false>);
    }
    const LaunchShape ls = persistent_shape(a.ngroups, 16);
    const size_t lds = janet_lds_bytes(P, ls.waves, false);
    auto k = janet_fwd_kernel<0>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
    return (int)hipGetLastError();
}
int janet_family_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    if (janet_uses_s16(m, a.B)) return janet_s16_launch(st, m, a, 2);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (!nw && !dx) return ODPD_EINVAL;
    const int P = janet_layout(m->hidden).P;
    const LaunchShape ls = janet_bwd_shape(a.ngroups);
    const size_t lds = janet_lds_bytes(P, ls.waves, nw);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(ls.grid), dim3(64 * ls.waves), lds, st, a);
        return (int)hipGetLastError();
    };
    if (nw && dx) return launch(janet_bwd_kernel<true, true>);
    if (nw) return launch(janet_bwd_kernel<true, false>);
    return launch(janet_bwd_kernel<false, true>);
}
int janet_family_rows(const odpd_model_t* m, int B) {
    if (m->hidden > 16) return ODPD_EUNSUPPORTED;
    if (janet_uses_s16(m, B)) return janet_s16_rows(m, B);
    return janet_bwd_shape(num_groups(B, 1)).grid;
}

}  // namespace odpd
