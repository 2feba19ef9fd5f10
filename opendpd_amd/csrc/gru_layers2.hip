// gru_layers2.hip — gru / dgru / qgru / qgru_amp1 with TWO stacked recurrent layers (`num_layers 2` of nn.GRU: backbones/gru.py:17-21, dgru.py:22-27;
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
//   dgru      its head (relu(fc_hid h2), fc_out over [hid, features]) runs with lane = tick on the chunk, reading the features of time s - 1 at tick
//             s; in the backward pass the head's share of dL/dx(s - 1) is stored at tick s and layer 1 adds its share at tick s - 1.
// Per-tick records (r, z, n, W_hn h + b_hn, h of both layers; dgru: + the fc_hid pre-activation) in HBM: B x (T + 1) x 5 (6) x 64 floats.
#include "odpd_seq.h"

namespace odpd {
namespace {
constexpr int k2C = 64, k2S = 65, k2NS = 5;
constexpr int k2Hs = ((k2C + 1) * k2S + 3) & ~3;

struct Gru2Layout { int H, F, OW, o_w_ih0, o_w_hh0, o_b_ih0, o_b_hh0, o_w_ih1, o_w_hh1, o_b_ih1, o_b_hh1, o_w_out, o_b_out, o_w_hid, o_b_hid, P; };
// named_parameters() of nn.GRU(F -> H, num_layers 2) + fc_out (dgru: over [relu(fc_hid h), features], then fc_hid: dgru.py:22-32)
__host__ __device__ inline Gru2Layout gru2_layout(int H, int F, int dgru = 0) {
    Gru2Layout L;
    L.H = H; L.F = F; L.OW = dgru ? H + 6 : H;
    int o = 0;
    L.o_w_ih0 = o; o += 3 * H * F; L.o_w_hh0 = o; o += 3 * H * H; L.o_b_ih0 = o; o += 3 * H; L.o_b_hh0 = o; o += 3 * H;
    L.o_w_ih1 = o; o += 3 * H * H; L.o_w_hh1 = o; o += 3 * H * H; L.o_b_ih1 = o; o += 3 * H; L.o_b_hh1 = o; o += 3 * H;
    L.o_w_out = o; o += 2 * L.OW; L.o_b_out = o; o += 2;
    L.o_w_hid = o; o += dgru ? H * H : 0; L.o_b_hid = o; o += dgru ? H : 0;
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
constexpr int k2X = 33;          // row stride of the dgru head's [tick][unit <= 32] arrays
__host__ __device__ inline int gru2_fwd_floats(int P, bool dg) { return pad4(P) + (k2C + 1) * 8 + 64 + k2C * k2S + (dg ? k2C * k2X + 32 * 32 : 0); }
__host__ __device__ inline int gru2_bwd_floats(int P, bool dg) {
    return pad4(P) + 3 * 64 * 64 + (k2C + 1) * 8 + k2C * 2 + k2C * 2 + 4 * 64 + k2Hs + (dg ? 2 * k2C * k2X + 32 * 32 : 0);
}

// features of the chunk's ticks: row 1 + i = time s0 + i (layer 1's input at that tick), row 0 = time s0 - 1 (the dgru head of tick s0 reads it)
template <int FM>
__device__ __forceinline__ void gru2_stage_features(float* ftab, const float2* xg, int s0, int T, int lane) {
    constexpr int F = FeatDim<FM>::F;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && lane != 0) break;
        const int t = pass == 0 ? s0 + lane : s0 - 1, row = pass == 0 ? lane + 1 : 0;
        const float2 xv = (t >= 0 && t < T) ? xg[t] : make_float2(0.5f, 0.5f);
        float f[F], o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        feat_fwd<FM>(xv.x, xv.y, f);
#pragma unroll
        for (int i = 0; i < F; ++i) o[i] = f[i];
        reinterpret_cast<float4*>(ftab)[2 * row] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4*>(ftab)[2 * row + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

template <int FM, bool DG, bool SAVE>
__global__ __launch_bounds__(64) void gru2_fwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, NS = DG ? 6 : k2NS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31;
    const bool l2 = lane >= 32;
    const Gru2Layout L = gru2_layout(a.H, F, DG);
    const int H = L.H, T = a.T, NT = T + 1, OW = L.OW;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [65][8]: features (gru2_stage_features)
    float* hb = ftab + (k2C + 1) * 8;          // [64]: [h1 | h2], for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: the state after each tick of the chunk
    float* hist2 = hist + k2C * k2S;           // DG: [64][33] fc_hid pre-activations of the chunk's ticks
    float* whp = hist2 + k2C * k2X;            // DG: fc_hid rows, zero padded to 32 columns (64 x 33 floats before it: 16-byte aligned)
    const bool vo = ju < H;
    if constexpr (DG) {
        for (int i = lane; i < 32 * 32; i += 64) whp[i] = ((i >> 5) < H && (i & 31) < H) ? pl[L.o_w_hid + (i >> 5) * H + (i & 31)] : 0.0f;
    }
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
        float* sv = SAVE ? a.ckpt + (size_t)b * NT * NS * 64 : nullptr;
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
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt + 2], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 3];
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
                    float* rec = sv + (size_t)s * NS * 64 + lane;
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
                if constexpr (!DG) {
                    for (int j = 0; j < H; ++j) {
                        const float hv = hr[j];
                        y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + OW + j], hv, y1);
                    }
                } else {      // out = relu(fc_hid(h2)); y = fc_out(cat(out, features of time s - 1)) (dgru.py:71-73)
                    float hrow[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) hrow[k] = hr[k];
                    for (int j = 0; j < H; ++j) {
                        float acc = pl[L.o_b_hid + j];
                        const float4* w4 = reinterpret_cast<const float4*>(whp + j * 32);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 w = w4[q];
                            acc = __builtin_fmaf(w.x, hrow[4 * q], acc); acc = __builtin_fmaf(w.y, hrow[4 * q + 1], acc);
                            acc = __builtin_fmaf(w.z, hrow[4 * q + 2], acc); acc = __builtin_fmaf(w.w, hrow[4 * q + 3], acc);
                        }
                        hist2[lane * k2X + j] = acc;
                        const float o = __builtin_fmaxf(acc, 0.0f);
                        y0 = __builtin_fmaf(pl[L.o_w_out + j], o, y0); y1 = __builtin_fmaf(pl[L.o_w_out + OW + j], o, y1);
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const float fv = ftab[lane * 8 + i];            // row `lane` = time s0 + lane - 1
                        y0 = __builtin_fmaf(pl[L.o_w_out + H + i], fv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + OW + H + i], fv, y1);
                    }
                }
                yg[s0 + lane - 1] = make_float2(y0, y1);
            }
            if constexpr (DG && SAVE) {      // the fc_hid pre-activations join the tick's record (slot 5, layer-2 lanes)
                wave_lds_fence();
                if (l2)
                    for (int tt = 0; tt < len; ++tt) sv[(size_t)(s0 + tt) * NS * 64 + 320 + lane] = (ju < H && s0 + tt >= 1) ? hist2[tt * k2X + ju] : 0.0f;
            }
        }
        wave_lds_fence();
    }
}

template <int FM, bool DG, bool NW, bool DX>
__global__ __launch_bounds__(64) void gru2_bwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, NS = DG ? 6 : k2NS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, ju = lane & 31, col = lane & 15, quad = lane >> 4;
    const bool l2 = lane >= 32;
    const Gru2Layout L = gru2_layout(a.H, F, DG);
    const int H = L.H, T = a.T, NT = T + 1, NC = (NT + k2C - 1) / k2C, OW = L.OW;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* wsup = smem + pad4(L.P);            // [3][64][64]: the gates' block matrices (row j, column k), zero padded
    float* ftab = wsup + 3 * 64 * 64;          // [65][8]  features (gru2_stage_features)
    float* dxb = ftab + (k2C + 1) * 8;         // [64][2]  dL/dx of the chunk's ticks (layer 1's share)
    float* dyb = dxb + k2C * 2;                // [64][2]  dL/dy(s - 1) at tick s
    float* dgb = dyb + k2C * 2;                // [4][64]  d_r, d_z, n-gate gradient for first-half columns, for second-half columns
    float* hs = dgb + 4 * 64;                  // [65][65] row i = the state after tick s0 - 1 + i
    float* x1 = hs + k2Hs;                     // DG: [64][33] relu(fc_hid) of the tick, then fc_hid^T dL/dhid
    float* dhid = x1 + k2C * k2X;              // DG: [64][33] dL/d(fc_hid pre-activation)
    float* whp = dhid + k2C * k2X;             // DG: fc_hid rows, zero padded to 32 columns
    const bool vo = ju < H;
    if constexpr (DG) {
        for (int i = lane; i < 32 * 32; i += 64) whp[i] = ((i >> 5) < H && (i & 31) < H) ? pl[L.o_w_hid + (i >> 5) * H + (i & 31)] : 0.0f;
    }
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
    const float wo0 = (!DG && vo && l2) ? pl[L.o_w_out + ju] : 0.0f, wo1 = (!DG && vo && l2) ? pl[L.o_w_out + OW + ju] : 0.0f;
    f32x4 ahid[2][2];                          // DG: dW_hid tiles (unit block jb, unit block kb)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) ahid[jb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbhid = 0.0f, tf[12];                // DG: fc_hid bias (layer-2 lanes); per tick lane: fc_out's feature columns
#pragma unroll
    for (int i = 0; i < 12; ++i) tf[i] = 0.0f;
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
        const float* sv = a.ckpt + (size_t)b * NT * NS * 64;
        float dh = 0.0f;
        for (int c = NC - 1; c >= 0; --c) {
            const int s0 = c * k2C, len = min(k2C, NT - s0);
            wave_lds_fence();
            gru2_stage_features<FM>(ftab, xg, s0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len && s0 + lane >= 1) dyv = dyg[s0 + lane - 1];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            hs[lane] = s0 > 0 ? sv[(size_t)(s0 - 1) * NS * 64 + 256 + lane] : 0.0f;
            for (int tt = 0; tt < k2C; ++tt) {
                hs[(tt + 1) * k2S + lane] = tt < len ? sv[(size_t)(s0 + tt) * NS * 64 + 256 + lane] : 0.0f;
                if constexpr (DG) { if (l2) x1[tt * k2X + ju] = tt < len ? sv[(size_t)(s0 + tt) * NS * 64 + 320 + lane] : 0.0f; }
            }
            wave_lds_fence();
            if constexpr (DG) {
                // (i) lane = tick s (time s - 1): dL/dhid = relu'(pre) (fc_out^T dL/dy); relu(pre) kept for fc_out's weight gradient; fc_out's feature
                //     columns; the head's share of dL/dx(s - 1) goes straight to memory (layer 1 adds its share when its tick s - 1 comes)
                {
                    float fe[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) fe[i] = ftab[lane * 8 + i];            // row `lane` = time s0 + lane - 1
                    for (int j = 0; j < 32; ++j) {
                        const float pre = x1[lane * k2X + j];
                        const float d = (j < H && pre > 0.0f) ? __builtin_fmaf(dyv.x, pl[L.o_w_out + j], dyv.y * pl[L.o_w_out + OW + j]) : 0.0f;
                        dhid[lane * k2X + j] = d;
                        x1[lane * k2X + j] = __builtin_fmaxf(pre, 0.0f);
                    }
                    if constexpr (NW) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) { tf[i] = __builtin_fmaf(dyv.x, fe[i], tf[i]); tf[6 + i] = __builtin_fmaf(dyv.y, fe[i], tf[6 + i]); }
                    }
                    if constexpr (DX) {
                        float df[6];
#pragma unroll
                        for (int i = 0; i < 6; ++i) df[i] = __builtin_fmaf(dyv.x, pl[L.o_w_out + H + i], dyv.y * pl[L.o_w_out + OW + H + i]);
                        float dI, dQ;
                        feat_bwd<FM>(fe[0], fe[1], df, dI, dQ);
                        if (lane < len && s0 + lane >= 1) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + s0 + lane - 1] = make_float2(dI, dQ);
                    }
                }
                wave_lds_fence();
                // (ii) lane = unit of layer 2: fc_out's hid columns, fc_hid's bias; dW_hid += sum over the ticks of dL/dhid (x) h2, 16 x 16 x 4 tiles
                if constexpr (NW) {
                    if (l2) {
                        for (int tt = 0; tt < len; ++tt) {
                            const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                            const float o = x1[tt * k2X + ju];
                            dwo0 = __builtin_fmaf(d.x, o, dwo0); dwo1 = __builtin_fmaf(d.y, o, dwo1);
                            dbhid += dhid[tt * k2X + ju];
                        }
                    }
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb) {
                            f32x4 t = ahid[jb][kb];
                            for (int t4 = 0; t4 < k2C / 4; ++t4) {
                                const float av = dhid[(4 * t4 + quad) * k2X + 16 * jb + col];
                                const float bv = hs[(4 * t4 + quad + 1) * k2S + 32 + 16 * kb + col];
                                t = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, t, 0, 0, 0);
                            }
                            ahid[jb][kb] = t;
                        }
                }
                wave_lds_fence();
                // (iii) lane = tick: fc_hid^T dL/dhid -> the head's dL/dh2 of the tick, into x1
                {
                    float dhh[32];
#pragma unroll
                    for (int k = 0; k < 32; ++k) dhh[k] = 0.0f;
                    for (int j = 0; j < H; ++j) {
                        const float d = dhid[lane * k2X + j];
                        const float4* w4 = reinterpret_cast<const float4*>(whp + j * 32);
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 w = w4[q];
                            dhh[4 * q] = __builtin_fmaf(d, w.x, dhh[4 * q]); dhh[4 * q + 1] = __builtin_fmaf(d, w.y, dhh[4 * q + 1]);
                            dhh[4 * q + 2] = __builtin_fmaf(d, w.z, dhh[4 * q + 2]); dhh[4 * q + 3] = __builtin_fmaf(d, w.w, dhh[4 * q + 3]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 32; ++k) x1[lane * k2X + k] = dhh[k];
                }
                wave_lds_fence();
            }
            float rn, zn, nn, gn_;
            {
                const float* rec = sv + (size_t)(s0 + len - 1) * NS * 64 + lane;
                rn = rec[0]; zn = rec[64]; nn = rec[128]; gn_ = rec[192];
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const int s = s0 + tt;
                const float r = rn, z = zn, n = nn, ghn = gn_;
                if (tt > 0) {
                    const float* rec = sv + (size_t)(s - 1) * NS * 64 + lane;
                    rn = rec[0]; zn = rec[64]; nn = rec[128]; gn_ = rec[192];
                }
                const bool active = vo && (l2 ? s >= 1 : s < T);
                const float hp = hs[tt * k2S + lane], ht = hs[(tt + 1) * k2S + lane];
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                float dht = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, dh));      // (layer 1 lanes, dgru: wo = 0)
                if constexpr (DG) { if (l2) dht += x1[tt * k2X + ju]; }
                if constexpr (NW && !DG) { dwo0 = __builtin_fmaf(d.x, ht, dwo0); dwo1 = __builtin_fmaf(d.y, ht, dwo1); }
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
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt + 2], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 3];
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
                if (lane < len && s0 + lane < T) {
                    float2* dst = reinterpret_cast<float2*>(a.dx) + (size_t)b * T + s0 + lane;
                    float2 v = reinterpret_cast<const float2*>(dxb)[lane];
                    if constexpr (DG) {      // + the head's share, stored by this wave at tick s0 + lane + 1 (an earlier phase of this or the later chunk)
                        __builtin_amdgcn_s_waitcnt(0);
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
                        const volatile float* hd = reinterpret_cast<const volatile float*>(dst);
                        v.x += hd[0]; v.y += hd[1];
                    }
                    *dst = v;
                }
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
        if constexpr (DG) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                for (int o = 32; o > 0; o >>= 1) tf[i] += __shfl_xor(tf[i], o);
        }
        if (lane == 0) {
            prow[L.o_b_out] = tb0; prow[L.o_b_out + 1] = tb1;
            if constexpr (DG) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { prow[L.o_w_out + H + i] = tf[i]; prow[L.o_w_out + OW + H + i] = tf[6 + i]; }
            }
        }
        if constexpr (DG) {
            if (vo && l2) prow[L.o_b_hid + ju] = dbhid;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int jr = 16 * jb + 4 * quad + i, kc = 16 * kb + col;
                        if (jr < H && kc < H) prow[L.o_w_hid + jr * H + kc] = ahid[jb][kb][i];
                    }
        }
        if (vo) {
            if (l2) { prow[L.o_w_out + ju] = dwo0; prow[L.o_w_out + OW + ju] = dwo1; }
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
    case ODPD_DGRU: FM = FEAT_DGRU6; return true;
    case ODPD_QGRU: FM = FEAT_Q4; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; return true;
    default: return false;
    }
}
int gru2_P(const odpd_model_t* m, int FM) { return gru2_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), FM == FEAT_DGRU6).P; }
template <typename K>
int gru2_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// float gru / dgru / qgru / qgru_amp1 with two recurrent layers (ODPD_FLAG_TWO_LAYERS) of <= 32 hidden units
bool gru2_ok(const odpd_model_t* m) {
    int FM;
    return (m->flags & ODPD_FLAG_TWO_LAYERS) && m->bits_w == 0 && m->hidden >= 1 && m->hidden <= 32 && gru2_cfg(m, FM);
}
int64_t gru2_param_count(const odpd_model_t* m) {
    int FM;
    if (!gru2_cfg(m, FM)) return ODPD_EUNSUPPORTED;
    return gru2_P(m, FM);
}
int64_t gru2_ckpt_floats(const odpd_model_t* m, int B, int T) { return (int64_t)B * (T + 1) * (m->backbone == ODPD_DGRU ? 6 : k2NS) * 64; }
int gru2_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int gru2_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM;
    if (!gru2_ok(m) || !gru2_cfg(m, FM)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)gru2_fwd_floats(gru2_P(m, FM), FM == FEAT_DGRU6) * sizeof(float);
    const int grid = gru2_rows(m, a.B);
#define ODPD_GRU2_FWD(FM_, DG_) \
    if (FM == FM_) return a.ckpt ? gru2_launch(st, gru2_fwd_kernel<FM_, DG_, true>, grid, lds, a) : gru2_launch(st, gru2_fwd_kernel<FM_, DG_, false>, grid, lds, a);
    ODPD_GRU2_FWD(FEAT_RAW2, false) ODPD_GRU2_FWD(FEAT_DGRU6, true) ODPD_GRU2_FWD(FEAT_Q4, false) ODPD_GRU2_FWD(FEAT_A4, false)
#undef ODPD_GRU2_FWD
    return ODPD_EUNSUPPORTED;
}
int gru2_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM;
    if (!gru2_ok(m) || !gru2_cfg(m, FM)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)gru2_bwd_floats(gru2_P(m, FM), FM == FEAT_DGRU6) * sizeof(float);
    const int grid = gru2_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
#define ODPD_GRU2_BWD(FM_, DG_)                                                                       \
    if (FM == FM_) {                                                                                  \
        if (nw && dx) return gru2_launch(st, gru2_bwd_kernel<FM_, DG_, true, true>, grid, lds, a);    \
        if (nw) return gru2_launch(st, gru2_bwd_kernel<FM_, DG_, true, false>, grid, lds, a);         \
        return gru2_launch(st, gru2_bwd_kernel<FM_, DG_, false, true>, grid, lds, a);                 \
    }
    ODPD_GRU2_BWD(FEAT_RAW2, false) ODPD_GRU2_BWD(FEAT_DGRU6, true) ODPD_GRU2_BWD(FEAT_Q4, false) ODPD_GRU2_BWD(FEAT_A4, false)
#undef ODPD_GRU2_BWD
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
