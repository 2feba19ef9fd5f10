// rvtdcnn_q.hip — `--quant` on rvtdcnn (reference quant/quant_envs.py:145-148, 285-306 on backbones/rvtdcnn.py:9-62): Conv2d becomes an
// INT_Conv2D (quant/qmodules/quant_layers.py:10-45: weight and activation quantisers; its weight scale starts at 2 mean|w| / sqrt(Qp) and is
// rounded to a power of two like every scale), fc_hid and fc_out become INT_Linear (:48-85; fc_out applies the 16-bit output quantiser in
// eval mode), the functional tanh calls stay float.  Parameter layout = the quantised model's named_parameters():
//     Conv2d.weight (27), Conv2d.bias (3), s_w, s_a | fc_hid.weight (H x 36), fc_hid.bias (H), s_w, s_a, s_out | fc_out.weight (2 x H), fc_out.bias (2), s_w, s_a, s_out
//
// Mapping: ONE LANE PER SAMPLE (as csrc/rvtdcnn.hip), but a plain kernel: the workgroup stages the parameters in LDS and quantises the three
// weight tensors IN PLACE there (every later read is a broadcast of q_w(W)); a sample's patch, conv outputs and their quantised values live in
// registers; the fc_hid rows are visited twice (forward, then backward with the row's activation recomputed) so that nothing per-unit is kept.
// Weight gradients of fc_hid / fc_out are contractions over SAMPLES: each wave parks its 64 samples' columns [dL/dpre_u | q_a(z) | dy | q_a(hid)]
// in a private LDS tile and accumulates the 16 x 16 output tiles with v_mfma_f32_16x16x4_f32 (exact fp32; the sample index is K — the operand
// layout of csrc/rvtdcnn.hip); the 27 + 3 convolution gradients and the fc_out bias are per-lane accumulators reduced once; every wave
// deposits its sums in its own LDS row and the four rows are added in fixed order at the end (deterministic).  The weight quantisers' pass masks are applied at write-out from the global (unquantised) weights, the eight
// scale columns stay 0 (round() inside the quantiser: quantizers.py:56-65).  dL/dx in gather form: the lane of sample s repeats the forward
// and backward of the four samples whose patch holds s (rows 3 .. 0) and keeps its own row of each patch gradient — no atomics, no exchange.
#include <type_traits>
#include <utility>

#include "odpd_s16.h"
#include "odpd_quant.h"

#pragma clang fp contract(off)

namespace odpd {
namespace {

constexpr int kQT = 256, kQZ = 36;
template <class F, int... I>
__device__ __forceinline__ void rvq_static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void rvq_static_for(F&& f) { rvq_static_for_impl(f, std::make_integer_sequence<int, N>{}); }
struct RvqLayout { int H, oK, okb, oqc, owh, obh, oqh, owo, obo, oqo, P; };
__host__ __device__ inline RvqLayout rvq_layout(int H) {
    RvqLayout L; L.H = H; int o = 0;
    L.oK = o; o += 27; L.okb = o; o += 3; L.oqc = o; o += 2;
    L.owh = o; o += kQZ * H; L.obh = o; o += H; L.oqh = o; o += 3;
    L.owo = o; o += 2 * H; L.obo = o; o += 2; L.oqo = o; o += 3;
    L.P = o;
    return L;
}
struct RvqQ { q16::Quant ca, ha, oa, out; };      // the three activation quantisers, fc_out's output quantiser
// per-wave LDS tile of MODE 1 (sample = lane is the fast index; stride 68 = 4 mod 32 floats: the MFMA operand reads of lanes (element n,
// sample 4c + q) land on banks 4n + q + const): dhp[u][s] (32 units), hq[u][s], zq[k][s] (36), dy[c][s]
constexpr int kQCol = 68, kQoDh = 0, kQoHq = 32 * kQCol, kQoZ = 64 * kQCol, kQoDy = kQoZ + kQZ * kQCol, kQTile = kQoDy + 2 * 64;

__device__ __forceinline__ float rvq_wsum(float v) {
    v = row_sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ size_t rvq_base(const SeqArgs& a, int b) {
    return a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * a.T;
}
// the 4 x 5 patch of sample t: row w = features [I, Q, a, a^2, a^3] of sample t - 3 + w, the frame's own last samples in front
// (rvtdcnn.py:41-53); `in` = q_a(features), `pin` = the activation quantiser's pass mask
__device__ __forceinline__ void rvq_patch(const float2* x2, size_t base, int t, int T, const q16::Quant& qa, float (&in)[4][5], float (&pin)[4][5]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int s = t - 3 + w;
        s += s < 0 ? T : 0;
        const float2 xv = x2[base + s];
        const float a2 = xv.x * xv.x + xv.y * xv.y, am = sqrtf(a2);
        const float f[5] = {xv.x, xv.y, am, a2, am * am * am};
#pragma unroll
        for (int j = 0; j < 5; ++j) { in[w][j] = q16::qapply(f[j], qa); pin[w][j] = q16::qpass(f[j], qa); }
    }
}
// z = tanh(conv(q_a(patch), q_w(K)) + b) (grid sums first, then the float bias: F.conv2d on quantised operands), zq = q_a(z)
__device__ __forceinline__ void rvq_conv(const float* pl, const RvqLayout& L, const q16::Quant& qh, const float (&in)[4][5], float (&z)[kQZ], float (&zq)[kQZ]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float k[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) k[i] = pl[L.oK + c * 9 + i];
        const float kb = pl[L.okb + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj)
                        if (r + dw - 1 >= 0 && r + dw - 1 <= 3) acc = __builtin_fmaf(k[dw * 3 + dj], in[r + dw - 1][j + dj], acc);
                const int idx = (c * 4 + r) * 3 + j;
                z[idx] = tanhf_(acc + kb);
                zq[idx] = q16::qapply(z[idx], qh);
            }
    }
}
// pre-activation of fc_hid's unit u on the quantised conv outputs (row u of q_w(W_hid) as nine broadcast float4 reads)
__device__ __forceinline__ float rvq_hid_pre(const float* pl, const RvqLayout& L, int u, const float (&zq)[kQZ], float (&w)[kQZ]) {
    const float4* wr = reinterpret_cast<const float4*>(pl + L.owh + u * kQZ);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const float4 v = wr[q];
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        a0 = __builtin_fmaf(v.x, zq[4 * q], a0); a1 = __builtin_fmaf(v.y, zq[4 * q + 1], a1);
        a0 = __builtin_fmaf(v.z, zq[4 * q + 2], a0); a1 = __builtin_fmaf(v.w, zq[4 * q + 3], a1);
    }
    return (a0 + a1) + pl[L.obh + u];
}
// forward of one sample from its quantised conv outputs: y (before the output quantiser)
__device__ __forceinline__ void rvq_head(const float* pl, const RvqLayout& L, const RvqQ& Q, const float (&zq)[kQZ], float& y0, float& y1) {
    float p0 = 0.0f, p1 = 0.0f, w[kQZ];
    for (int u = 0; u < L.H; ++u) {
        const float hq = q16::qapply(tanhf_(rvq_hid_pre(pl, L, u, zq, w)), Q.oa);
        p0 = __builtin_fmaf(pl[L.owo + u], hq, p0); p1 = __builtin_fmaf(pl[L.owo + L.H + u], hq, p1);
    }
    y0 = p0 + pl[L.obo]; y1 = p1 + pl[L.obo + 1];
}
// dL/dz of one sample (through fc_out, the tanh of fc_hid and the quantisers' pass masks); NW: the unit's columns go to the wave's tile `row`
template <bool NW>
__device__ __forceinline__ void rvq_back_rows(const float* pl, const RvqLayout& L, const RvqQ& Q, const float (&zq)[kQZ], float d0, float d1, float own,
                                              float* row, int lane, float (&dz)[kQZ]) {
#pragma unroll
    for (int k = 0; k < kQZ; ++k) dz[k] = 0.0f;
    float w[kQZ];
    for (int u = 0; u < L.H; ++u) {
        const float hid = tanhf_(rvq_hid_pre(pl, L, u, zq, w));
        const float hq = q16::qapply(hid, Q.oa), ph = q16::qpass(hid, Q.oa);
        const float dh = ((pl[L.owo + u] * d0 + pl[L.owo + L.H + u] * d1) * ph) * (1.0f - hid * hid);
#pragma unroll
        for (int k = 0; k < kQZ; ++k) dz[k] = __builtin_fmaf(w[k], dh, dz[k]);
        if constexpr (NW) {      // `row` = the wave's tile: this unit's columns (d0 / d1 carry `own`, so idle samples contribute zeros)
            row[kQoDh + u * kQCol + lane] = own * dh;
            row[kQoHq + u * kQCol + lane] = hq;
        }
    }
}

// MODE 0: forward.  MODE 1: weight gradients (FUSED: a.target holds the target, the loss and dL/dy are formed here; frames may be windows
// of resident streams).  MODE 2: dL/dx.
template <int MODE, bool FUSED>
__global__ __launch_bounds__(kQT) void rvq_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RvqLayout L = rvq_layout(a.H);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P4 = pad4(L.P + kLossCols);
    float* pl = smem;
    float* row = smem + pad4(L.P) + wave * P4;          // MODE 1: this wave's gradient row
    float* tile = smem + pad4(L.P) + 4 * P4 + wave * kQTile;      // ... and its operand tile
    for (int i = tid; i < L.P; i += kQT) pl[i] = a.params[i];
    if constexpr (MODE == 1) {
        for (int i = tid; i < 4 * P4; i += kQT) smem[pad4(L.P) + i] = 0.0f;
        for (int i = tid; i < 4 * kQTile; i += kQT) smem[pad4(L.P) + 4 * P4 + i] = 0.0f;      // (units >= H are never written: zero operands)
    }
    __syncthreads();
    RvqQ Q;
    Q.ca = q16::make_quant(pl[L.oqc + 1], a.bits_a); Q.ha = q16::make_quant(pl[L.oqh + 1], a.bits_a);
    Q.oa = q16::make_quant(pl[L.oqo + 1], a.bits_a); Q.out = q16::make_quant(pl[L.oqo + 2], 16);
    {
        const q16::Quant qc = q16::make_quant(pl[L.oqc], a.bits_w), qh = q16::make_quant(pl[L.oqh], a.bits_w), qo = q16::make_quant(pl[L.oqo], a.bits_w);
        __syncthreads();
        for (int i = tid; i < 27; i += kQT) pl[L.oK + i] = q16::qapply(pl[L.oK + i], qc);
        for (int i = tid; i < kQZ * L.H; i += kQT) pl[L.owh + i] = q16::qapply(pl[L.owh + i], qh);
        for (int i = tid; i < 2 * L.H; i += kQT) pl[L.owo + i] = q16::qapply(pl[L.owo + i], qo);
        __syncthreads();
    }
    const float2* x2 = reinterpret_cast<const float2*>(a.x);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    float dK[27], dkb[3] = {0.f, 0.f, 0.f}, dbo[2] = {0.f, 0.f}, loss_acc = 0.0f;
    f32x4 Dz[2][3], Dh[2];                             // MODE 1: dW_hid | db_hid tiles (units x (36 z columns + the constant 1)), dW_out tiles
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        Dh[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) Dz[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int n = lane & 15, q = lane >> 4;
    // B-operand selectors of the third z tile (columns 32 + n): z[32 .. 35], then the constant 1 (bias gradient), then nothing
    const int zk2 = n < 4 ? 32 + n : 35;
    const float zmul2 = n < 4 ? 1.0f : 0.0f, zadd2 = n == 4 ? 1.0f : 0.0f, ymask = n < 2 ? 1.0f : 0.0f;
#pragma unroll
    for (int i = 0; i < 27; ++i) dK[i] = 0.0f;
    const long long N = (long long)a.B * a.T;
    const long long npass = (N + kQT - 1) / kQT;
    for (long long pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const long long s = pass * kQT + tid;
        const bool active = s < N;
        const int b = active ? (int)(s / a.T) : 0, t = active ? (int)(s - (long long)b * a.T) : 0;
        const size_t base = rvq_base(a, b);
        if constexpr (MODE == 0) {
            float in[4][5], pin[4][5], z[kQZ], zq[kQZ], y0, y1;
            rvq_patch(x2, base, t, a.T, Q.ca, in, pin);
            rvq_conv(pl, L, Q.ha, in, z, zq);
            rvq_head(pl, L, Q, zq, y0, y1);
            if (a.eval_out) { y0 = q16::qapply(y0, Q.out); y1 = q16::qapply(y1, Q.out); }
            if (active) reinterpret_cast<float2*>(a.y)[s] = make_float2(y0, y1);
        } else if constexpr (MODE == 1) {
            float in[4][5], pin[4][5], z[kQZ], zq[kQZ], dz[kQZ], d0, d1;
            rvq_patch(x2, base, t, a.T, Q.ca, in, pin);
            rvq_conv(pl, L, Q.ha, in, z, zq);
            const float own = active ? 1.0f : 0.0f;
            if constexpr (FUSED) {
                float y0, y1, l = 0.0f;
                rvq_head(pl, L, Q, zq, y0, y1);
                const float2 tv = reinterpret_cast<const float2*>(a.target)[base + t];
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, d0, d1, l);
                loss_acc += own * l;
            } else {
                const float2 dv = reinterpret_cast<const float2*>(a.dy)[base + t];
                d0 = dv.x; d1 = dv.y;
            }
            d0 *= own; d1 *= own;
            dbo[0] += d0; dbo[1] += d1;
            wave_lds_fence();                                  // the previous pass's MFMA operand reads are done
#pragma unroll
            for (int k = 0; k < kQZ; ++k) tile[kQoZ + k * kQCol + lane] = zq[k];
            tile[kQoDy + lane] = d0; tile[kQoDy + 64 + lane] = d1;
            rvq_back_rows<true>(pl, L, Q, zq, d0, d1, own, tile, lane, dz);
            wave_lds_fence();
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {                     // K = the wave's 64 samples, four per MFMA
                const int sm = 4 * c + q;
                const float ay = ymask * tile[kQoDy + (n & 1) * 64 + sm];
                float ad[2], hv[2], bz[3];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) { ad[mt] = tile[kQoDh + (16 * mt + n) * kQCol + sm]; hv[mt] = tile[kQoHq + (16 * mt + n) * kQCol + sm]; }
                bz[0] = tile[kQoZ + n * kQCol + sm];
                bz[1] = tile[kQoZ + (16 + n) * kQCol + sm];
                bz[2] = __builtin_fmaf(tile[kQoZ + zk2 * kQCol + sm], zmul2, zadd2);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) Dz[mt][nt] = mfma4(ad[mt], bz[nt], Dz[mt][nt]);
                    Dh[mt] = mfma4(ay, hv[mt], Dh[mt]);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int idx = (c * 4 + r) * 3 + j;
                        const float dc = (dz[idx] * q16::qpass(z[idx], Q.ha)) * (1.0f - z[idx] * z[idx]);
                        dkb[c] += dc;
#pragma unroll
                        for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                            for (int dj = 0; dj < 3; ++dj)
                                if (r + dw - 1 >= 0 && r + dw - 1 <= 3) dK[(c * 3 + dw) * 3 + dj] = __builtin_fmaf(dc, in[r + dw - 1][j + dj], dK[(c * 3 + dw) * 3 + dj]);
                    }
        } else {
            // sample s sits in row 3 - i of the patch of sample t + i (i = 0 .. 3, circular)
            float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            const float2* dy2 = reinterpret_cast<const float2*>(a.dy);
            rvq_static_for<4>([&](auto ic) {
                constexpr int i = decltype(ic)::value, wr = 3 - i;
                int tt = t + i;
                tt -= tt >= a.T ? a.T : 0;
                float in[4][5], pin[4][5], z[kQZ], zq[kQZ], dz[kQZ];
                rvq_patch(x2, base, tt, a.T, Q.ca, in, pin);
                rvq_conv(pl, L, Q.ha, in, z, zq);
                const float2 dv = dy2[base + tt];
                rvq_back_rows<false>(pl, L, Q, zq, dv.x, dv.y, 0.0f, nullptr, lane, dz);
                float din[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int idx = (c * 4 + r) * 3 + j;
                            const float dc = (dz[idx] * q16::qpass(z[idx], Q.ha)) * (1.0f - z[idx] * z[idx]);
#pragma unroll
                            for (int dj = 0; dj < 3; ++dj) {
                                const int dw = wr - r + 1;          // the kernel row that reads patch row wr for output row r
                                if (dw >= 0 && dw <= 2) din[j + dj] = __builtin_fmaf(pl[L.oK + (c * 3 + dw) * 3 + dj], dc, din[j + dj]);
                            }
                        }
#pragma unroll
                for (int f = 0; f < 5; ++f) g[f] = __builtin_fmaf(din[f], pin[wr][f], g[f]);
            });
            if (active) {      // features [I, Q, a, a^2, a^3]: da/dI = I / a, da^2/dI = 2 I, da^3/dI = 3 a I
                const float2 xv = x2[base + t];
                const float am = sqrtf(xv.x * xv.x + xv.y * xv.y);
                const float ga = g[2] / am + 2.0f * g[3] + 3.0f * am * g[4];
                reinterpret_cast<float2*>(a.dx)[s] = make_float2(g[0] + ga * xv.x, g[1] + ga * xv.y);
            }
        }
    }
    if constexpr (MODE == 1) {
        // element i of a tile on lane (n, q): row 4 q + i, column n.  Dz[mt][nt]: unit 16 mt + 4 q + i x z column 16 nt + n (column 36: the bias);
        // Dh[mt]: output channel 4 q + i x unit 16 mt + n
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = 16 * mt + 4 * q + i;
                if (u < L.H) {
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) {
                        const int k = 16 * nt + n;
                        if (k < kQZ) row[L.owh + u * kQZ + k] = Dz[mt][nt][i];
                        else if (k == kQZ) row[L.obh + u] = Dz[mt][nt][i];
                    }
                }
                const int c = 4 * q + i, uo = 16 * mt + n;
                if (c < 2 && uo < L.H) row[L.owo + c * L.H + uo] = Dh[mt][i];
            }
#pragma unroll
        for (int i = 0; i < 27; ++i) { const float v = rvq_wsum(dK[i]); if (lane == 0) row[L.oK + i] = v; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { const float v = rvq_wsum(dkb[i]); if (lane == 0) row[L.okb + i] = v; }
        const float v0 = rvq_wsum(dbo[0]), v1 = rvq_wsum(dbo[1]), lp = rvq_wsum(loss_acc);
        if (lane == 0) { row[L.obo] = v0; row[L.obo + 1] = v1; row[L.P] = lp; }
        __syncthreads();
        // the four waves' rows in fixed order; the weight quantisers' pass masks from the unquantised weights; scale columns: exact 0
        const q16::Quant qc = q16::make_quant(a.params[L.oqc], a.bits_w), qh = q16::make_quant(a.params[L.oqh], a.bits_w),
                         qo = q16::make_quant(a.params[L.oqo], a.bits_w);
        const float* r0 = smem + pad4(L.P);
        float* out = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
        for (int i = tid; i < L.P + kLossCols; i += kQT) {
            float v = (r0[i] + r0[P4 + i]) + (r0[2 * P4 + i] + r0[3 * P4 + i]);
            if (i >= L.oK && i < L.oK + 27) v *= q16::qpass(a.params[i], qc);
            else if (i >= L.owh && i < L.obh) v *= q16::qpass(a.params[i], qh);
            else if (i >= L.owo && i < L.obo) v *= q16::qpass(a.params[i], qo);
            else if ((i >= L.oqc && i < L.owh) || (i >= L.oqh && i < L.owo) || (i >= L.oqo && i < L.P)) v = 0.0f;
            out[i] = v;
        }
    }
}

inline int rvq_grid(int B, int T) {
    const long long np = ((long long)B * T + kQT - 1) / kQT;
    const long long cap = 2LL * device_cus();
    return (int)(np < cap ? (np < 1 ? 1 : np) : cap);
}
template <typename K> int rvq_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kQT), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

bool rvtdcnn_q_ok(const odpd_model_t* m, int T) {
    return m->backbone == ODPD_RVTDCNN && m->bits_w > 0 && m->bits_w <= 16 && m->bits_a > 0 && m->bits_a <= 16 && m->hidden >= 1 && m->hidden <= 32 && T >= 3;
}
int64_t rvtdcnn_q_param_count(const odpd_model_t* m) { return rvq_layout(m->hidden).P; }
int rvtdcnn_q_rows(const odpd_model_t*, int B, int T) { return rvq_grid(B, T); }
int rvtdcnn_q_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!rvtdcnn_q_ok(m, a.T)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)pad4(rvq_layout(m->hidden).P) * sizeof(float);
    return rvq_launch(st, rvq_kernel<0, false>, rvq_grid(a.B, a.T), lds, a);
}
// dy -> partials and / or dx; `fused`: a.target instead of a.dy, loss in column P of the partial rows
int rvtdcnn_q_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, bool fused) {
    if (!rvtdcnn_q_ok(m, a.T)) return ODPD_EUNSUPPORTED;
    const RvqLayout L = rvq_layout(m->hidden);
    const int grid = rvq_grid(a.B, a.T);
    if (a.partials != nullptr) {
        const size_t lds = (size_t)(pad4(L.P) + 4 * pad4(L.P + kLossCols) + 4 * kQTile) * sizeof(float);
        if (int e = fused ? rvq_launch(st, rvq_kernel<1, true>, grid, lds, a) : rvq_launch(st, rvq_kernel<1, false>, grid, lds, a)) return e;
    }
    if (a.dx != nullptr && !fused) return rvq_launch(st, rvq_kernel<2, false>, grid, (size_t)pad4(L.P) * sizeof(float), a);
    return 0;
}

}  // namespace odpd
