// lstm_layers2.hip — the plain LSTM backbone with TWO stacked layers (nn.LSTM num_layers = 2: backbones/lstm.py:17-21; `--PA_num_layers 2`), hidden
// <= 32: the arrangement of gru_layers2.hip — layer 1 on lanes 0 .. 31, layer 2 on lanes 32 .. 63, time-skewed by one step, one block matrix
// [[W_hh0, 0], [W_ih1, W_hh1]] per gate against the broadcast vector [h1(s-1) | h2(s-2)] — with lstm_wide.hip's step arithmetic (four gates i, f,
// g, o whose input and hidden parts simply add; (h, c) per lane; gate o's row in a padded LDS copy in the forward pass).
// Per-tick records (i, f, g, o, c, h of both layers) in HBM: B x (T + 1) x 6 x 64 floats.
#include "odpd_seq.h"

namespace odpd {
namespace {
constexpr int kLC = 64, kLS = 65, kLNS = 6;
constexpr int kLHs = ((kLC + 1) * kLS + 3) & ~3;

struct Lstm2Layout { int H, o_w_ih0, o_w_hh0, o_b_ih0, o_b_hh0, o_w_ih1, o_w_hh1, o_b_ih1, o_b_hh1, o_w_out, o_b_out, P; };
__host__ __device__ inline Lstm2Layout lstm2_layout(int H) {      // named_parameters() of nn.LSTM(2 -> H, num_layers 2) + fc_out
    Lstm2Layout L;
    L.H = H;
    int o = 0;
    L.o_w_ih0 = o; o += 4 * H * 2; L.o_w_hh0 = o; o += 4 * H * H; L.o_b_ih0 = o; o += 4 * H; L.o_b_hh0 = o; o += 4 * H;
    L.o_w_ih1 = o; o += 4 * H * H; L.o_w_hh1 = o; o += 4 * H * H; L.o_b_ih1 = o; o += 4 * H; L.o_b_hh1 = o; o += 4 * H;
    L.o_w_out = o; o += 2 * H; L.o_b_out = o; o += 2;
    L.P = o;
    return L;
}
// entry (row lane j, column k) of gate g's block matrix [[W_hh0, 0], [W_ih1, W_hh1]] (32-unit blocks, zero padded); -1: structural zero
__host__ __device__ inline int lstm2_super_index(const Lstm2Layout& L, int g, int j, int k) {
    const int H = L.H, ju = j & 31, ku = k & 31;
    if (ju >= H || ku >= H) return -1;
    if (j < 32) return k < 32 ? L.o_w_hh0 + (g * H + ju) * H + ku : -1;
    return k < 32 ? L.o_w_ih1 + (g * H + ju) * H + ku : L.o_w_hh1 + (g * H + ju) * H + ku;
}
__host__ __device__ inline int lstm2_fwd_floats(int P) { return pad4(P) + kLC * 2 + 64 + kLC * kLS + 64 * kLS; }
__host__ __device__ inline int lstm2_bwd_floats(int P) { return pad4(P) + 4 * 64 * 64 + kLC * 2 + kLC * 2 + kLC * 2 + 4 * 64 + kLHs; }

template <bool SAVE>
__global__ __launch_bounds__(64) void lstm2_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31;
    const bool l2 = lane >= 32;
    const Lstm2Layout L = lstm2_layout(a.H);
    const int H = L.H, T = a.T, NT = T + 1;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* xb = smem + pad4(L.P);              // [64][2]: I, Q of the chunk's steps
    float* hb = xb + kLC * 2;                  // [64]: the state, for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: h of the chunk's steps
    float* wop = hist + kLC * kLS;             // [64][65]: gate o's block-matrix rows
    const bool vo = ju < H;
    for (int i = lane; i < 64 * kLS; i += 64) {
        const int j = i / kLS, k = i % kLS;
        const int idx = k < 64 ? lstm2_super_index(L, 3, j, k) : -1;
        wop[i] = idx >= 0 ? pl[idx] : 0.0f;
    }
    float whh[3][64], wih[4][2], bg[4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 64; ++k) { const int idx = lstm2_super_index(L, g, lane, k); whh[g][k] = idx >= 0 ? pl[idx] : 0.0f; }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        wih[g][0] = (vo && !l2) ? pl[L.o_w_ih0 + (g * H + ju) * 2] : 0.0f;
        wih[g][1] = (vo && !l2) ? pl[L.o_w_ih0 + (g * H + ju) * 2 + 1] : 0.0f;
        bg[g] = vo ? pl[(l2 ? L.o_b_ih1 : L.o_b_ih0) + g * H + ju] + pl[(l2 ? L.o_b_hh1 : L.o_b_hh0) + g * H + ju] : 0.0f;
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * NT * kLNS * 64 : nullptr;
        float h = 0.0f, c = 0.0f;
        for (int t0 = 0; t0 < NT; t0 += kLC) {
            const int len = min(kLC, NT - t0);
            wave_lds_fence();
            reinterpret_cast<float2*>(xb)[lane] = t0 + lane < T ? xg[t0 + lane] : make_float2(0.0f, 0.0f);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                hb[lane] = h;
                wave_lds_fence();
                const float2 xv = reinterpret_cast<const float2*>(xb)[tt];
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) pre[g] = __builtin_fmaf(wih[g][1], xv.y, __builtin_fmaf(wih[g][0], xv.x, bg[g]));
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
                const float* wo = wop + lane * kLS;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 hv = hb4[q];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        pre[g] = __builtin_fmaf(whh[g][4 * q], hv.x, pre[g]); pre[g] = __builtin_fmaf(whh[g][4 * q + 1], hv.y, pre[g]);
                        pre[g] = __builtin_fmaf(whh[g][4 * q + 2], hv.z, pre[g]); pre[g] = __builtin_fmaf(whh[g][4 * q + 3], hv.w, pre[g]);
                    }
                    pre[3] = __builtin_fmaf(wo[4 * q], hv.x, pre[3]); pre[3] = __builtin_fmaf(wo[4 * q + 1], hv.y, pre[3]);
                    pre[3] = __builtin_fmaf(wo[4 * q + 2], hv.z, pre[3]); pre[3] = __builtin_fmaf(wo[4 * q + 3], hv.w, pre[3]);
                }
                const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), go = sigmoidf_(pre[3]);
                const bool active = vo && (l2 ? t0 + tt >= 1 : t0 + tt < T);      // (an idle layer's state passes through the tick unchanged)
                const float cn = active ? __builtin_fmaf(gf, c, gi * gg) : c;
                const float hn = active ? go * tanhf_(cn) : h;
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * kLNS * 64 + lane;
                    s[0] = gi; s[64] = gf; s[128] = gg; s[192] = go; s[256] = cn; s[320] = hn;
                }
                c = cn; h = hn;
                hist[tt * kLS + lane] = h;
                wave_lds_fence();
            }
            if (lane < len && t0 + lane >= 1) {      // outputs of the chunk's ticks, lane = tick: y(s - 1) = fc_out(h2(s - 1))
                const float* hr = hist + lane * kLS + 32;
                float y0 = pl[L.o_b_out], y1 = pl[L.o_b_out + 1];
                for (int j = 0; j < H; ++j) {
                    const float hv = hr[j];
                    y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + H + j], hv, y1);
                }
                yg[t0 + lane - 1] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
}

template <bool NW, bool DX>
__global__ __launch_bounds__(64) void lstm2_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31, col = lane & 15, quad = lane >> 4;
    const bool l2 = lane >= 32;
    const Lstm2Layout L = lstm2_layout(a.H);
    const int H = L.H, T = a.T, NT = T + 1, NC = (NT + kLC - 1) / kLC;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* wsup = smem + pad4(L.P);            // [4][64][64]: the gates' block matrices (row j, column k), zero padded
    for (int i = lane; i < 4 * 64 * 64; i += 64) {
        const int idx = lstm2_super_index(L, i >> 12, (i >> 6) & 63, i & 63);
        wsup[i] = idx >= 0 ? pl[idx] : 0.0f;
    }
    float* xb = wsup + 4 * 64 * 64;            // [64][2]  I, Q of the chunk's ticks
    float* dxb = xb + kLC * 2;                 // [64][2]  dL/dx of the chunk's steps
    float* dyb = dxb + kLC * 2;                // [64][2]  dL/dy of the chunk's steps
    float* dgb = dyb + kLC * 2;                // [4][64]  the step's gate gradients, for the broadcast reads
    float* hs = dgb + 4 * 64;                  // [65][65] row i = h(t0 - 1 + i)
    const bool vo = ju < H;
    float wih[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        wih[g][0] = (vo && !l2) ? pl[L.o_w_ih0 + (g * H + ju) * 2] : 0.0f;
        wih[g][1] = (vo && !l2) ? pl[L.o_w_ih0 + (g * H + ju) * 2 + 1] : 0.0f;
    }
    const float wo0 = (vo && l2) ? pl[L.o_w_out + ju] : 0.0f, wo1 = (vo && l2) ? pl[L.o_w_out + H + ju] : 0.0f;
    f32x16 acc[4][4];                          // dW_hh: gate g, the state rotated by 16 r lanes
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    float dwih[4][2], dbs[4] = {0.f, 0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) { dwih[g][0] = 0.0f; dwih[g][1] = 0.0f; }
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * NT * kLNS * 64;
        float dh = 0.0f, dc = 0.0f;
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kLC, len = min(kLC, NT - t0);
            wave_lds_fence();
            reinterpret_cast<float2*>(xb)[lane] = t0 + lane < T ? xg[t0 + lane] : make_float2(0.0f, 0.0f);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len && t0 + lane >= 1) dyv = dyg[t0 + lane - 1];      // tick s carries dL/dy(s - 1)
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            hs[lane] = t0 > 0 ? sv[(size_t)(t0 - 1) * kLNS * 64 + 320 + lane] : 0.0f;
            for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * kLS + lane] = sv[(size_t)(t0 + tt) * kLNS * 64 + 320 + lane];
            wave_lds_fence();
            // the chunk's steps in reverse, lane = unit (the next step's record is in flight while this one is worked on)
            float in_, fn_, gn_, on_, cn_, cpn_;
            {
                const float* s = sv + (size_t)(t0 + len - 1) * kLNS * 64 + lane;
                in_ = s[0]; fn_ = s[64]; gn_ = s[128]; on_ = s[192]; cn_ = s[256];
                cpn_ = t0 + len - 1 > 0 ? s[256 - kLNS * 64] : 0.0f;
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float gi = in_, gf = fn_, gg = gn_, go = on_, ct = cn_, cp = cpn_;
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * kLNS * 64 + lane;
                    in_ = s[0]; fn_ = s[64]; gn_ = s[128]; on_ = s[192]; cn_ = s[256];
                    cpn_ = t0 + tt - 1 > 0 ? s[256 - kLNS * 64] : 0.0f;
                }
                const bool active = vo && (l2 ? t0 + tt >= 1 : t0 + tt < T);
                const float hp = hs[tt * kLS + lane], ht = hs[(tt + 1) * kLS + lane];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                const float dht = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, dh));      // (layer 1 lanes: wo = 0)
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, ht, dwo0); dwo1 = __builtin_fmaf(d.y, ht, dwo1); }
                const float tc = tanhf_(ct);
                const float dct = __builtin_fmaf(dht * go, __builtin_fmaf(-tc, tc, 1.0f), dc);      // dL/dc(t)
                const float dpi = active ? (dct * gg) * (gi * (1.0f - gi)) : 0.0f;
                const float dpf = active ? (dct * cp) * (gf * (1.0f - gf)) : 0.0f;
                const float dpg = active ? (dct * gi) * __builtin_fmaf(-gg, gg, 1.0f) : 0.0f;
                const float dpo = active ? (dht * tc) * (go * (1.0f - go)) : 0.0f;
                dc = active ? dct * gf : (vo ? dc : 0.0f);              // an idle layer's (h, c) pass through the tick unchanged
                dgb[lane] = dpi; dgb[64 + lane] = dpf; dgb[128 + lane] = dpg; dgb[192 + lane] = dpo;
                wave_lds_fence();
                float dhn = active ? 0.0f : dht;
                {
                    const float* w0 = wsup + lane;
                    for (int j4 = 0; j4 < 64; j4 += 4) {
                        const float4 a0 = *reinterpret_cast<const float4*>(dgb + j4), a1 = *reinterpret_cast<const float4*>(dgb + 64 + j4),
                                     a2 = *reinterpret_cast<const float4*>(dgb + 128 + j4), a3 = *reinterpret_cast<const float4*>(dgb + 192 + j4);
                        const float v0[4] = {a0.x, a0.y, a0.z, a0.w}, v1[4] = {a1.x, a1.y, a1.z, a1.w}, v2[4] = {a2.x, a2.y, a2.z, a2.w},
                                    v3[4] = {a3.x, a3.y, a3.z, a3.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* wr = w0 + (j4 + e) * 64;
                            dhn = __builtin_fmaf(v0[e], wr[0], dhn); dhn = __builtin_fmaf(v1[e], wr[4096], dhn);
                            dhn = __builtin_fmaf(v2[e], wr[8192], dhn); dhn = __builtin_fmaf(v3[e], wr[12288], dhn);
                        }
                    }
                }
                dh = vo ? dhn : 0.0f;
                const float2 xv = reinterpret_cast<const float2*>(xb)[tt];
                if constexpr (NW) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float hpr = rr == 0 ? hp : __shfl(hp, (lane + 16 * rr) & 63);
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpi, hpr, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpf, hpr, acc[1][rr], 0, 0, 0);
                        acc[2][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpg, hpr, acc[2][rr], 0, 0, 0);
                        acc[3][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dpo, hpr, acc[3][rr], 0, 0, 0);
                    }
                    const float dp[4] = {dpi, dpf, dpg, dpo};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        dwih[g][0] = __builtin_fmaf(dp[g], xv.x, dwih[g][0]); dwih[g][1] = __builtin_fmaf(dp[g], xv.y, dwih[g][1]);
                        dbs[g] += dp[g];
                    }
                }
                if constexpr (DX) {
                    float d0 = __builtin_fmaf(dpi, wih[0][0], __builtin_fmaf(dpf, wih[1][0], __builtin_fmaf(dpg, wih[2][0], dpo * wih[3][0])));
                    float d1 = __builtin_fmaf(dpi, wih[0][1], __builtin_fmaf(dpf, wih[1][1], __builtin_fmaf(dpg, wih[2][1], dpo * wih[3][1])));
                    for (int o = 32; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); }
                    if (lane == 0) reinterpret_cast<float2*>(dxb)[tt] = make_float2(d0, d1);
                }
                wave_lds_fence();
            }
            if constexpr (DX) {
                wave_lds_fence();
                if (lane < len && t0 + lane < T) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t0 + lane] = reinterpret_cast<const float2*>(dxb)[lane];
            }
        }
        wave_lds_fence();
    }
    if constexpr (NW) {
        float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
        for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.0f;
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_fence();
        for (int o = 32; o > 0; o >>= 1) { tb0 += __shfl_xor(tb0, o); tb1 += __shfl_xor(tb1, o); }
        if (lane == 0) { prow[L.o_b_out] = tb0; prow[L.o_b_out + 1] = tb1; }
        if (vo) {
            if (l2) { prow[L.o_w_out + ju] = dwo0; prow[L.o_w_out + H + ju] = dwo1; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!l2) { prow[L.o_w_ih0 + (g * H + ju) * 2] = dwih[g][0]; prow[L.o_w_ih0 + (g * H + ju) * 2 + 1] = dwih[g][1]; }
                prow[(l2 ? L.o_b_ih1 : L.o_b_ih0) + g * H + ju] = dbs[g]; prow[(l2 ? L.o_b_hh1 : L.o_b_hh0) + g * H + ju] = dbs[g];
            }
        }
        // MFMA block bb of (gate g, rotation rr): register 4 bb + i of lane l = entry (row 4 (l / 16) + i, column l % 16) of the block
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = lstm2_super_index(L, g, 16 * bb + 4 * quad + i, 16 * ((bb + rr) & 3) + col);
                        if (idx >= 0) prow[idx] = acc[g][rr][4 * bb + i];
                    }
    }
}

template <typename K>
int lstm2_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// float lstm with two recurrent layers (ODPD_FLAG_TWO_LAYERS) of <= 32 hidden units
bool lstm2_ok(const odpd_model_t* m) {
    return (m->flags & ODPD_FLAG_TWO_LAYERS) && m->backbone == ODPD_LSTM && m->bits_w == 0 && m->hidden >= 1 && m->hidden <= 32;
}
int64_t lstm2_param_count(const odpd_model_t* m) { return lstm2_layout(m->hidden).P; }
int64_t lstm2_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * (T + 1) * kLNS * 64; }
int lstm2_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int lstm2_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!lstm2_ok(m)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)lstm2_fwd_floats(lstm2_layout(m->hidden).P) * sizeof(float);
    const int grid = lstm2_rows(m, a.B);
    return a.ckpt ? lstm2_launch(st, lstm2_fwd_kernel<true>, grid, lds, a) : lstm2_launch(st, lstm2_fwd_kernel<false>, grid, lds, a);
}
int lstm2_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!lstm2_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)lstm2_bwd_floats(lstm2_layout(m->hidden).P) * sizeof(float);
    const int grid = lstm2_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw && dx) return lstm2_launch(st, lstm2_bwd_kernel<true, true>, grid, lds, a);
    if (nw) return lstm2_launch(st, lstm2_bwd_kernel<true, false>, grid, lds, a);
    return lstm2_launch(st, lstm2_bwd_kernel<false, true>, grid, lds, a);
}

}  // namespace odpd
