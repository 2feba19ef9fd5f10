// gru_layers2.hip — gru / qgru / qgru_amp1 with TWO stacked recurrent layers (`num_layers 2` of nn.GRU: backbones/gru.py:17-21, qgru.py:22-27;
// arguments.py `--PA_num_layers` / `--DPD_num_layers`), hidden <= 32: both layers in ONE wave, time-skewed by one step.
//   lanes 0 .. 31 = the units of layer 1, lanes 32 .. 63 = the units of layer 2; at tick s layer 1 takes step s and layer 2 step s - 1, so the
//   state vector broadcast through LDS at that tick, [h1(s-1) | h2(s-2)], is exactly what BOTH layers need: layer 1 multiplies its W_hh rows
//   with the first half, layer 2 its weight_ih_l1 rows with the first half (its input h1(s-1)) and its weight_hh_l1 rows with the second.
//   A lane therefore holds three 64-wide rows of a block "super-matrix" [[W_hh0, 0], [W_ih1, W_hh1]] per gate, and the kernels are those of
//   gru_wide.hip with T + 1 ticks — plus the one thing the block form does not hide: the n gate keeps its input part and its hidden part apart
//   (n = tanh(gi_n + r gh_n)), so the two halves of that row are accumulated separately (forward) and take different gate gradients
//   (backward: d_n on input-part columns, r d_n on hidden-part columns).
//   backward  reverse ticks with the same skew: the one transposed mat-vec of a tick hands layer 2's W_ih1^T d(gates) to layer 1 as dL/dh1 of
//             the SAME step and both layers' W_hh^T d(gates) to the previous step; the outer-product operand [h1(s-1) | h2(s-2)] is common to
//             all rows, so dW of both layers accumulates as one rotated 4-block MFMA update per gate and rotation.
// Per-tick records (r, z, n, W_hn h + b_hn, h of both layers) in HBM: B x (T + 1) x 5 x 64 floats.
#include "odpd_seq.h"

namespace odpd {
namespace {
constexpr int k2C = 64, k2S = 65, k2NS = 5;
constexpr int k2Hs = ((k2C + 1) * k2S + 3) & ~3;

struct Gru2Layout { int H, F, o_w_ih0, o_w_hh0, o_b_ih0, o_b_hh0, o_w_ih1, o_w_hh1, o_b_ih1, o_b_hh1, o_w_out, o_b_out, P; };
__host__ __device__ inline Gru2Layout gru2_layout(int H, int F) {      // named_parameters() of nn.GRU(F -> H, num_layers 2) + fc_out
    Gru2Layout L;
    L.H = H; L.F = F;
    int o = 0;
    L.o_w_ih0 = o; o += 3 * H * F; L.o_w_hh0 = o; o += 3 * H * H; L.o_b_ih0 = o; o += 3 * H; L.o_b_hh0 = o; o += 3 * H;
    L.o_w_ih1 = o; o += 3 * H * H; L.o_w_hh1 = o; o += 3 * H * H; L.o_b_ih1 = o; o += 3 * H; L.o_b_hh1 = o; o += 3 * H;
    L.o_w_out = o; o += 2 * H; L.o_b_out = o; o += 2;
    L.P = o;
    return L;
}
// entry (row lane j, column k) of gate g's block matrix [[W_hh0, 0], [W_ih1, W_hh1]] (32-unit blocks, zero padded); -1: structural zero
__host__ __device__ inline int gru2_super_index(const Gru2Layout& L, int g, int j, int k) {
    const int H = L.H, ju = j & 31, ku = k & 31;
    if (ju >= H || ku >= H) return -1;
    if (j < 32) return k < 32 ? L.o_w_hh0 + (g * H + ju) * H + ku : -1;
    return k < 32 ? L.o_w_ih1 + (g * H + ju) * H + ku : L.o_w_hh1 + (g * H + ju) * H + ku;
}
__host__ __device__ inline int gru2_fwd_floats(int P) { return pad4(P) + k2C * 8 + 64 + k2C * k2S; }
__host__ __device__ inline int gru2_bwd_floats(int P) { return pad4(P) + 3 * 64 * 64 + k2C * 8 + k2C * 2 + k2C * 2 + 4 * 64 + k2Hs; }

template <int FM>
__device__ __forceinline__ void gru2_stage_features(float* ftab, const float2* xg, int s0, int T, int lane) {
    constexpr int F = FeatDim<FM>::F;
    const int t = s0 + lane;
    const float2 xv = t < T ? xg[t] : make_float2(0.5f, 0.5f);
    float f[F], o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    feat_fwd<FM>(xv.x, xv.y, f);
#pragma unroll
    for (int i = 0; i < F; ++i) o[i] = f[i];
    reinterpret_cast<float4*>(ftab)[2 * lane] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<float4*>(ftab)[2 * lane + 1] = make_float4(o[4], o[5], o[6], o[7]);
}

template <int FM, bool SAVE>
__global__ __launch_bounds__(64) void gru2_fwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31;
    const bool l2 = lane >= 32;
    const Gru2Layout L = gru2_layout(a.H, F);
    const int H = L.H, T = a.T, NT = T + 1;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [64][8]: features of the chunk's ticks (layer 1's inputs)
    float* hb = ftab + k2C * 8;                // [64]: [h1 | h2], for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: the state after each tick of the chunk
    const bool vo = ju < H;
    float wsm[3][64], wih[3][F], bi[3], bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int k = 0; k < 64; ++k) { const int idx = gru2_super_index(L, g, lane, k); wsm[g][k] = idx >= 0 ? pl[idx] : 0.0f; }
#pragma unroll
        for (int i = 0; i < F; ++i) wih[g][i] = (vo && !l2) ? pl[L.o_w_ih0 + (g * H + ju) * F + i] : 0.0f;
        bi[g] = vo ? pl[(l2 ? L.o_b_ih1 : L.o_b_ih0) + g * H + ju] : 0.0f;
        bh[g] = vo ? pl[(l2 ? L.o_b_hh1 : L.o_b_hh0) + g * H + ju] : 0.0f;
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * NT * k2NS * 64 : nullptr;
        float h = 0.0f;
        for (int s0 = 0; s0 < NT; s0 += k2C) {
            const int len = min(k2C, NT - s0);
            wave_lds_fence();
            gru2_stage_features<FM>(ftab, xg, s0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                const int s = s0 + tt;
                hb[lane] = h;
                wave_lds_fence();
                float ga[3] = {0.f, 0.f, 0.f}, gb[3] = {0.f, 0.f, 0.f}, gf[3] = {0.f, 0.f, 0.f};      // first / second half of the state; features
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 hv = hb4[q];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        float& acc = q < 8 ? ga[g] : gb[g];
                        acc = __builtin_fmaf(wsm[g][4 * q], hv.x, acc); acc = __builtin_fmaf(wsm[g][4 * q + 1], hv.y, acc);
                        acc = __builtin_fmaf(wsm[g][4 * q + 2], hv.z, acc); acc = __builtin_fmaf(wsm[g][4 * q + 3], hv.w, acc);
                    }
                }
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 1];
                const float fe[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < F; ++i) gf[g] = __builtin_fmaf(wih[g][i], fe[i], gf[g]);
                // layer 1: input part = features, hidden part = first half; layer 2: input part = first half (h1), hidden part = second half
                float gi[3], gh[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) { gi[g] = bi[g] + (l2 ? ga[g] : gf[g]); gh[g] = bh[g] + (l2 ? gb[g] : ga[g]); }
                const float r = sigmoidf_(gi[0] + gh[0]), z = sigmoidf_(gi[1] + gh[1]);
                const float n = tanhf_(__builtin_fmaf(r, gh[2], gi[2]));
                const bool active = vo && (l2 ? s >= 1 : s < T);
                const float hn = active ? __builtin_fmaf(z, h - n, n) : h;
                if constexpr (SAVE) {
                    float* rec = sv + (size_t)s * k2NS * 64 + lane;
                    rec[0] = r; rec[64] = z; rec[128] = n; rec[192] = gh[2]; rec[256] = hn;
                }
                h = hn;
                hist[tt * k2S + lane] = h;
                wave_lds_fence();
            }
            // outputs of the chunk's ticks, lane = tick: y(s - 1) = fc_out(h2(s - 1))
            if (lane < len && s0 + lane >= 1) {
                const float* hr = hist + lane * k2S + 32;
                float y0 = pl[L.o_b_out], y1 = pl[L.o_b_out + 1];
                for (int j = 0; j < H; ++j) {
                    const float hv = hr[j];
                    y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + H + j], hv, y1);
                }
                yg[s0 + lane - 1] = make_float2(y0, y1);
            }
        }
        wave_lds_fence();
    }
}

template <int FM, bool NW, bool DX>
__global__ __launch_bounds__(64) void gru2_bwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31, col = lane & 15, quad = lane >> 4;
    const bool l2 = lane >= 32;
    const Gru2Layout L = gru2_layout(a.H, F);
    const int H = L.H, T = a.T, NT = T + 1, NC = (NT + k2C - 1) / k2C;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* wsup = smem + pad4(L.P);            // [3][64][64]: the gates' block matrices (row j, column k), zero padded
    float* ftab = wsup + 3 * 64 * 64;          // [64][8]  features of the chunk's ticks
    float* dxb = ftab + k2C * 8;               // [64][2]  dL/dx of the chunk's ticks
    float* dyb = dxb + k2C * 2;                // [64][2]  dL/dy(s - 1) at tick s
    float* dgb = dyb + k2C * 2;                // [4][64]  d_r, d_z, n-gate gradient for first-half columns, for second-half columns
    float* hs = dgb + 4 * 64;                  // [65][65] row i = the state after tick s0 - 1 + i
    const bool vo = ju < H;
    for (int i = lane; i < 3 * 64 * 64; i += 64) {
        const int g = i >> 12, j = (i >> 6) & 63, k = i & 63;
        const int idx = gru2_super_index(L, g, j, k);
        wsup[i] = idx >= 0 ? pl[idx] : 0.0f;
    }
    float wih[3][F];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < F; ++i) wih[g][i] = (vo && !l2) ? pl[L.o_w_ih0 + (g * H + ju) * F + i] : 0.0f;
    const float wo0 = (vo && l2) ? pl[L.o_w_out + ju] : 0.0f, wo1 = (vo && l2) ? pl[L.o_w_out + H + ju] : 0.0f;
    f32x16 acc[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    float dwih[3][F], dbs[4] = {0.f, 0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < F; ++i) dwih[g][i] = 0.0f;
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * NT * k2NS * 64;
        float dh = 0.0f;
        for (int c = NC - 1; c >= 0; --c) {
            const int s0 = c * k2C, len = min(k2C, NT - s0);
            wave_lds_fence();
            gru2_stage_features<FM>(ftab, xg, s0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len && s0 + lane >= 1) dyv = dyg[s0 + lane - 1];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            hs[lane] = s0 > 0 ? sv[(size_t)(s0 - 1) * k2NS * 64 + 256 + lane] : 0.0f;
            for (int tt = 0; tt < len; ++tt) hs[(tt + 1) * k2S + lane] = sv[(size_t)(s0 + tt) * k2NS * 64 + 256 + lane];
            wave_lds_fence();
            float rn, zn, nn, gn_;
            {
                const float* rec = sv + (size_t)(s0 + len - 1) * k2NS * 64 + lane;
                rn = rec[0]; zn = rec[64]; nn = rec[128]; gn_ = rec[192];
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const int s = s0 + tt;
                const float r = rn, z = zn, n = nn, ghn = gn_;
                if (tt > 0) {
                    const float* rec = sv + (size_t)(s - 1) * k2NS * 64 + lane;
                    rn = rec[0]; zn = rec[64]; nn = rec[128]; gn_ = rec[192];
                }
                const bool active = vo && (l2 ? s >= 1 : s < T);
                const float hp = hs[tt * k2S + lane], ht = hs[(tt + 1) * k2S + lane];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                const float dht = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, dh));      // (layer 1 lanes: wo = 0)
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, ht, dwo0); dwo1 = __builtin_fmaf(d.y, ht, dwo1); }
                const float dn = dht * (1.0f - z), dz = dht * (hp - n);
                const float dnp = active ? dn * __builtin_fmaf(-n, n, 1.0f) : 0.0f;
                const float drp = (dnp * ghn) * (r * (1.0f - r));
                const float dzp = active ? dz * (z * (1.0f - z)) : 0.0f;
                const float dghn = dnp * r;
                // n gate: input-part columns take d_n, hidden-part columns r d_n.  First-half columns are the hidden part of layer-1 rows and
                // the input part of layer-2 rows; second-half columns are the hidden part of layer-2 rows.
                const float gA = l2 ? dnp : dghn, gB = dghn;
                dgb[lane] = drp; dgb[64 + lane] = dzp; dgb[128 + lane] = gA; dgb[192 + lane] = gB;
                wave_lds_fence();
                float dhn = active ? dht * z : dht;                 // an idle layer's state passes through the tick unchanged
                {
                    const float* w0 = wsup + lane;
                    const float* gn4 = dgb + (l2 ? 192 : 128);
                    for (int j4 = 0; j4 < 64; j4 += 4) {
                        const float4 gr = *reinterpret_cast<const float4*>(dgb + j4), gz = *reinterpret_cast<const float4*>(dgb + 64 + j4),
                                     gn = *reinterpret_cast<const float4*>(gn4 + j4);
                        const float grv[4] = {gr.x, gr.y, gr.z, gr.w}, gzv[4] = {gz.x, gz.y, gz.z, gz.w}, gnv[4] = {gn.x, gn.y, gn.z, gn.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float* wr = w0 + (j4 + e) * 64;
                            dhn = __builtin_fmaf(grv[e], wr[0], dhn); dhn = __builtin_fmaf(gzv[e], wr[4096], dhn); dhn = __builtin_fmaf(gnv[e], wr[8192], dhn);
                        }
                    }
                }
                dh = vo ? dhn : 0.0f;
                if constexpr (NW) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float hpr = rr == 0 ? hp : __shfl(hp, (lane + 16 * rr) & 63);
                        const float gnv = (((quad + rr) & 3) >= 2) ? gB : gA;      // this lane's row block against column block (quad + rr) % 4
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(drp, hpr, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dzp, hpr, acc[1][rr], 0, 0, 0);
                        acc[2][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(gnv, hpr, acc[2][rr], 0, 0, 0);
                    }
                    dbs[0] += drp; dbs[1] += dzp; dbs[2] += dnp; dbs[3] += dghn;
                }
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 1];
                const float fe[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                if constexpr (NW) {
#pragma unroll
                    for (int i = 0; i < F; ++i) {
                        dwih[0][i] = __builtin_fmaf(drp, fe[i], dwih[0][i]); dwih[1][i] = __builtin_fmaf(dzp, fe[i], dwih[1][i]);
                        dwih[2][i] = __builtin_fmaf(dnp, fe[i], dwih[2][i]);
                    }
                }
                if constexpr (DX) {
                    float df[F];
#pragma unroll
                    for (int i = 0; i < F; ++i) {
                        float v = __builtin_fmaf(drp, wih[0][i], __builtin_fmaf(dzp, wih[1][i], dnp * wih[2][i]));      // (layer 2 lanes: wih = 0)
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        df[i] = v;
                    }
                    float dI, dQ;
                    feat_bwd<FM>(fe[0], fe[1], df, dI, dQ);
                    if (lane == 0) reinterpret_cast<float2*>(dxb)[tt] = make_float2(dI, dQ);
                }
                wave_lds_fence();
            }
            if constexpr (DX) {
                wave_lds_fence();
                if (lane < len && s0 + lane < T) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + s0 + lane] = reinterpret_cast<const float2*>(dxb)[lane];
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
            for (int g = 0; g < 3; ++g) {
                if (!l2) {
#pragma unroll
                    for (int i = 0; i < F; ++i) prow[L.o_w_ih0 + (g * H + ju) * F + i] = dwih[g][i];
                }
                prow[(l2 ? L.o_b_ih1 : L.o_b_ih0) + g * H + ju] = dbs[g];
                prow[(l2 ? L.o_b_hh1 : L.o_b_hh0) + g * H + ju] = g < 2 ? dbs[g] : dbs[3];
            }
        }
        // MFMA block bb of (gate g, rotation rr): register 4 bb + i of lane l = entry (row 4 (l / 16) + i, column l % 16) of the block
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = gru2_super_index(L, g, 16 * bb + 4 * quad + i, 16 * ((bb + rr) & 3) + col);
                        if (idx >= 0) prow[idx] = acc[g][rr][4 * bb + i];
                    }
    }
}

bool gru2_cfg(const odpd_model_t* m, int& FM) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; return true;
    case ODPD_QGRU: FM = FEAT_Q4; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; return true;
    default: return false;
    }
}
template <typename K>
int gru2_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// float gru / qgru / qgru_amp1 with two recurrent layers (ODPD_FLAG_TWO_LAYERS) of <= 32 hidden units
bool gru2_ok(const odpd_model_t* m) {
    int FM;
    return (m->flags & ODPD_FLAG_TWO_LAYERS) && m->bits_w == 0 && m->hidden >= 1 && m->hidden <= 32 && gru2_cfg(m, FM);
}
int64_t gru2_param_count(const odpd_model_t* m) {
    int FM;
    if (!gru2_cfg(m, FM)) return ODPD_EUNSUPPORTED;
    return gru2_layout(m->hidden, FM == FEAT_RAW2 ? 2 : 4).P;
}
int64_t gru2_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * (T + 1) * k2NS * 64; }
int gru2_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int gru2_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM;
    if (!gru2_ok(m) || !gru2_cfg(m, FM)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)gru2_fwd_floats((int)gru2_param_count(m)) * sizeof(float);
    const int grid = gru2_rows(m, a.B);
#define ODPD_GRU2_FWD(FM_) \
    if (FM == FM_) return a.ckpt ? gru2_launch(st, gru2_fwd_kernel<FM_, true>, grid, lds, a) : gru2_launch(st, gru2_fwd_kernel<FM_, false>, grid, lds, a);
    ODPD_GRU2_FWD(FEAT_RAW2) ODPD_GRU2_FWD(FEAT_Q4) ODPD_GRU2_FWD(FEAT_A4)
#undef ODPD_GRU2_FWD
    return ODPD_EUNSUPPORTED;
}
int gru2_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM;
    if (!gru2_ok(m) || !gru2_cfg(m, FM)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)gru2_bwd_floats((int)gru2_param_count(m)) * sizeof(float);
    const int grid = gru2_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
#define ODPD_GRU2_BWD(FM_)                                                                       \
    if (FM == FM_) {                                                                             \
        if (nw && dx) return gru2_launch(st, gru2_bwd_kernel<FM_, true, true>, grid, lds, a);    \
        if (nw) return gru2_launch(st, gru2_bwd_kernel<FM_, true, false>, grid, lds, a);         \
        return gru2_launch(st, gru2_bwd_kernel<FM_, false, true>, grid, lds, a);                 \
    }
    ODPD_GRU2_BWD(FEAT_RAW2) ODPD_GRU2_BWD(FEAT_Q4) ODPD_GRU2_BWD(FEAT_A4)
#undef ODPD_GRU2_BWD
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
