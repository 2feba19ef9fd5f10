// gru_wide.hip — gru / dgru / qgru / qgru_amp1 with 33 .. 64 hidden units (one layer; reference: backbones/gru.py:4-48, dgru.py:9-74,
// qgru.py:9-71, qgru_amp1.py:9-76 — `hidden_size` is a free argument of all four, arguments.py:49-60): ONE sequence per single-wave workgroup,
// LANE = HIDDEN UNIT.  The 16-wide tile kernels (gru_family.hip, gru_s16*.hip) stop at two unit tiles; here a unit's three W_hh rows (3 x 64
// registers) stay with its lane, the state is broadcast through LDS once per step (16 x ds_read_b128 of one address), and what does not depend
// on the recurrence runs with lane = time step on 64-step chunks (features; heads: fc_out, or dgru's fc_hid + relu + fc_out over [hid, features]).
//   forward   r, z, n, W_hn h + b_hn and h of every step (dgru: + the fc_hid pre-activation) go to a per-sequence record in HBM (the `ckpt`
//             buffer: B x T x NS x 64 floats) when the backward pass is going to need them;
//   backward  chunks in reverse; per step the lane of unit k forms dL/dh(t-1)[k] from the step's gate gradients (broadcast through LDS) and
//             column k of W_hh (read from the staged parameters: consecutive lanes, consecutive addresses); dW_hh accumulates as outer
//             products on the 4-block MFMA (v_mfma_f32_16x16x1_4b_f32: block b = units 16b .., the state rotated by 0 / 16 / 32 / 48 lanes
//             supplies the four column blocks), dW_ih and the biases on the VALU; dgru: the head's gradients of a chunk are formed with
//             lane = time step (dL/dhid, fc_hid^T dL/dhid -> dL/dh, the feature columns of fc_out), dW_hid as 16 x 16 x 4 MFMA tiles over the
//             chunk's (dL/dhid, h) rows in LDS.  One row of partial gradients per workgroup (every entry written).
// These kernels serve the shapes the tile kernels do not; they are built for correctness and a sane step time (LDS-broadcast bound, about
// 0.2 .. 0.4 us per time step and sequence), not for the roofline.
#include "odpd_seq.h"

namespace odpd {
namespace {
constexpr int kWC = 64;          // time steps per chunk
constexpr int kWS = 65;          // row stride of the per-chunk [time][unit] arrays (lane = unit and lane = time accesses both conflict-free)
constexpr int kWHs = ((kWC + 1) * kWS + 3) & ~3;      // floats of the [65][65] state array, padded so that what follows stays 16-byte aligned

template <int FM>
__device__ __forceinline__ void wide_stage_features(float* ftab, const float2* xg, int t0, int T, int lane) {
    constexpr int F = FeatDim<FM>::F;
    const int t = t0 + lane;
    const float2 xv = t < T ? xg[t] : make_float2(0.5f, 0.5f);
    float f[F], o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    feat_fwd<FM>(xv.x, xv.y, f);
#pragma unroll
    for (int i = 0; i < F; ++i) o[i] = f[i];
    reinterpret_cast<float4*>(ftab)[2 * lane] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<float4*>(ftab)[2 * lane + 1] = make_float4(o[4], o[5], o[6], o[7]);
}
__host__ __device__ inline int wide_fwd_floats(int P, bool dg, int H) {
    return pad4(P) + kWC * 8 + 64 + kWC * kWS + (dg ? kWC * kWS + H * 64 : 0);
}
__host__ __device__ inline int wide_bwd_floats(int P, bool dg, int H) {
    return pad4(P) + kWC * 8 + kWC * 8 + kWC * 2 + 4 * 64 + kWHs + (dg ? 2 * kWC * kWS + H * 64 : 0);
}

template <int FM, bool DG, bool SAVE>
__global__ __launch_bounds__(64) void wide_gru_fwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, NS = DG ? 6 : 5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const GruLayout L = gru_layout(a.H, F, DG);
    const int H = L.H, T = a.T, OW = DG ? H + 6 : H;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [64][8]: features of the chunk's steps
    float* hb = ftab + kWC * 8;                // [64]: the state, for the broadcast reads
    float* hist = hb + 64;                     // [64][65]: h of the chunk's steps
    float* hist2 = hist + kWC * kWS;           // DG: [64][65] fc_hid pre-activations
    float* whp = hist2 + kWC * kWS;            // DG: fc_hid rows, zero padded to 64 columns
    const bool vo = lane < H;
    if constexpr (DG) {
        for (int i = lane; i < H * 64; i += 64) whp[i] = (i & 63) < H ? pl[L.o_w_hid + (i >> 6) * H + (i & 63)] : 0.0f;
    }
    float whh[3][64], wih[3][F], bi[3], bh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int k = 0; k < 64; ++k) whh[g][k] = (vo && k < H) ? pl[L.o_w_hh + (g * H + lane) * H + k] : 0.0f;
#pragma unroll
        for (int i = 0; i < F; ++i) wih[g][i] = vo ? pl[L.o_w_ih + (g * H + lane) * F + i] : 0.0f;
        bi[g] = vo ? pl[L.o_b_ih + g * H + lane] : 0.0f;
        bh[g] = vo ? pl[L.o_b_hh + g * H + lane] : 0.0f;
    }
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * NS * 64 : nullptr;
        float h = 0.0f;
        for (int t0 = 0; t0 < T; t0 += kWC) {
            const int len = min(kWC, T - t0);
            wave_lds_fence();
            wide_stage_features<FM>(ftab, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                hb[lane] = h;
                wave_lds_fence();
                float gh[3] = {bh[0], bh[1], bh[2]}, gi[3] = {bi[0], bi[1], bi[2]};
                const float4* hb4 = reinterpret_cast<const float4*>(hb);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float4 hv = hb4[q];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        gh[g] = __builtin_fmaf(whh[g][4 * q], hv.x, gh[g]); gh[g] = __builtin_fmaf(whh[g][4 * q + 1], hv.y, gh[g]);
                        gh[g] = __builtin_fmaf(whh[g][4 * q + 2], hv.z, gh[g]); gh[g] = __builtin_fmaf(whh[g][4 * q + 3], hv.w, gh[g]);
                    }
                }
                const float4 f0 = reinterpret_cast<const float4*>(ftab)[2 * tt], f1 = reinterpret_cast<const float4*>(ftab)[2 * tt + 1];
                const float fe[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int i = 0; i < F; ++i) gi[g] = __builtin_fmaf(wih[g][i], fe[i], gi[g]);
                const float r = sigmoidf_(gi[0] + gh[0]), z = sigmoidf_(gi[1] + gh[1]);
                const float n = tanhf_(__builtin_fmaf(r, gh[2], gi[2]));
                const float hn = vo ? __builtin_fmaf(z, h - n, n) : 0.0f;            // (1 - z) n + z h
                if constexpr (SAVE) {
                    float* s = sv + (size_t)(t0 + tt) * NS * 64 + lane;
                    s[0] = r; s[64] = z; s[128] = n; s[192] = gh[2]; s[256] = hn;
                }
                h = hn;
                hist[tt * kWS + lane] = h;
                wave_lds_fence();
            }
            // the chunk's outputs, lane = time step
            if (lane < len) {
                const float* hr = hist + lane * kWS;
                float y0 = pl[L.o_b_out], y1 = pl[L.o_b_out + 1];
                if constexpr (!DG) {
                    for (int j = 0; j < H; ++j) {
                        const float hv = hr[j];
                        y0 = __builtin_fmaf(pl[L.o_w_out + j], hv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + OW + j], hv, y1);
                    }
                } else {
                    float hrow[64];
#pragma unroll
                    for (int k = 0; k < 64; ++k) hrow[k] = hr[k];
                    for (int j = 0; j < H; ++j) {             // out = relu(fc_hid(h)) (dgru.py:71)
                        float acc = pl[L.o_b_hid + j];
                        const float4* w4 = reinterpret_cast<const float4*>(whp + j * 64);
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float4 w = w4[q];
                            acc = __builtin_fmaf(w.x, hrow[4 * q], acc); acc = __builtin_fmaf(w.y, hrow[4 * q + 1], acc);
                            acc = __builtin_fmaf(w.z, hrow[4 * q + 2], acc); acc = __builtin_fmaf(w.w, hrow[4 * q + 3], acc);
                        }
                        hist2[lane * kWS + j] = acc;
                        const float o = __builtin_fmaxf(acc, 0.0f);
                        y0 = __builtin_fmaf(pl[L.o_w_out + j], o, y0); y1 = __builtin_fmaf(pl[L.o_w_out + OW + j], o, y1);
                    }
#pragma unroll
                    for (int i = 0; i < 6; ++i) {             // y = fc_out(cat(out, features)) (dgru.py:72-73)
                        const float fv = ftab[lane * 8 + i];
                        y0 = __builtin_fmaf(pl[L.o_w_out + H + i], fv, y0); y1 = __builtin_fmaf(pl[L.o_w_out + OW + H + i], fv, y1);
                    }
                }
                yg[t0 + lane] = make_float2(y0, y1);
            }
            if constexpr (DG && SAVE) {
                wave_lds_fence();
                for (int tt = 0; tt < len; ++tt) sv[(size_t)(t0 + tt) * NS * 64 + 320 + lane] = hist2[tt * kWS + lane];
            }
        }
        wave_lds_fence();
    }
}

template <int FM, bool DG, bool NW, bool DX>
__global__ __launch_bounds__(64) void wide_gru_bwd_kernel(SeqArgs a) {
    constexpr int F = FeatDim<FM>::F, NS = DG ? 6 : 5;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, col = lane & 15, quad = lane >> 4;
    const GruLayout L = gru_layout(a.H, F, DG);
    const int H = L.H, T = a.T, OW = DG ? H + 6 : H, NC = (T + kWC - 1) / kWC;
    float* pl = smem;
    stage_params(pl, a.params, L.P);
    float* ftab = smem + pad4(L.P);            // [64][8]  features of the chunk's steps
    float* dfh = ftab + kWC * 8;               // [64][8]  DG + DX: the head's share of dL/d(features); then dL/dx of the chunk's steps
    float* dyb = dfh + kWC * 8;                // [64][2]  dL/dy of the chunk's steps
    float* dgb = dyb + kWC * 2;                // [4][64]  the step's gate gradients (d_r, d_z, d_hn), for the broadcast reads
    float* hs = dgb + 4 * 64;                  // [65][65] row i = h(t0 - 1 + i)
    float* x1 = hs + kWHs;                     // DG: [64][65] relu(fc_hid), then fc_hid^T dL/dhid
    float* dhid = x1 + kWC * kWS;              // DG: [64][65] dL/d(fc_hid pre-activation)
    float* whp = dhid + kWC * kWS;             // DG: fc_hid rows, zero padded to 64 columns
    const bool vo = lane < H;
    if constexpr (DG) {
        for (int i = lane; i < H * 64; i += 64) whp[i] = (i & 63) < H ? pl[L.o_w_hid + (i >> 6) * H + (i & 63)] : 0.0f;
    }
    float wih[3][F];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < F; ++i) wih[g][i] = vo ? pl[L.o_w_ih + (g * H + lane) * F + i] : 0.0f;
    const float wo0 = (!DG && vo) ? pl[L.o_w_out + lane] : 0.0f, wo1 = (!DG && vo) ? pl[L.o_w_out + OW + lane] : 0.0f;
    // accumulators: per unit (lane)
    f32x16 acc[3][4];                          // dW_hh: gate g, the state rotated by 16 r lanes
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][r][i] = 0.0f;
    f32x4 ahid[4][4];                          // DG: dW_hid tiles (unit block jb, unit block kb)
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) ahid[jb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dwih[3][F], dbs[4] = {0.f, 0.f, 0.f, 0.f};      // dW_ih rows; sums of d_r, d_z, d_n, d_hn (the six bias gradients)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < F; ++i) dwih[g][i] = 0.0f;
    float dwo0 = 0.0f, dwo1 = 0.0f, dbhid = 0.0f;        // fc_out columns of the unit (dgru: of its hid), fc_hid bias
    float tacc[14];                                       // per time lane: fc_out bias (2), dgru: fc_out's feature columns (12)
#pragma unroll
    for (int i = 0; i < 14; ++i) tacc[i] = 0.0f;
    wave_lds_fence();

    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * NS * 64;
        float dh = 0.0f;
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kWC, len = min(kWC, T - t0);
            wave_lds_fence();
            // ---- stage the chunk: features and dL/dy (lane = time step), h(t0 - 1 .. t0 + len - 1) and the fc_hid pre-activations (lane = unit)
            wide_stage_features<FM>(ftab, xg, t0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            *reinterpret_cast<float2*>(dyb + 2 * lane) = dyv;
            hs[lane] = t0 > 0 ? sv[(size_t)(t0 - 1) * NS * 64 + 256 + lane] : 0.0f;
            for (int tt = 0; tt < kWC; ++tt) {
                hs[(tt + 1) * kWS + lane] = tt < len ? sv[(size_t)(t0 + tt) * NS * 64 + 256 + lane] : 0.0f;
                if constexpr (DG) x1[tt * kWS + lane] = tt < len ? sv[(size_t)(t0 + tt) * NS * 64 + 320 + lane] : 0.0f;
            }
            wave_lds_fence();
            if constexpr (NW) { tacc[0] += dyv.x; tacc[1] += dyv.y; }
            if constexpr (DG) {
                // (i) lane = time step: dL/dhid = relu'(pre) (fc_out^T dL/dy), relu(pre) kept for fc_out's weight gradient
                {
                    float fe[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) fe[i] = ftab[lane * 8 + i];
                    for (int j = 0; j < H; ++j) {
                        const float pre = x1[lane * kWS + j];
                        const float d = pre > 0.0f ? __builtin_fmaf(dyv.x, pl[L.o_w_out + j], dyv.y * pl[L.o_w_out + OW + j]) : 0.0f;
                        dhid[lane * kWS + j] = d;
                        x1[lane * kWS + j] = __builtin_fmaxf(pre, 0.0f);
                    }
                    for (int j = H; j < 64; ++j) { dhid[lane * kWS + j] = 0.0f; x1[lane * kWS + j] = 0.0f; }
                    if constexpr (NW) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) { tacc[2 + i] = __builtin_fmaf(dyv.x, fe[i], tacc[2 + i]); tacc[8 + i] = __builtin_fmaf(dyv.y, fe[i], tacc[8 + i]); }
                    }
                    if constexpr (DX) {
                        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 6; ++i) o[i] = __builtin_fmaf(dyv.x, pl[L.o_w_out + H + i], dyv.y * pl[L.o_w_out + OW + H + i]);
                        reinterpret_cast<float4*>(dfh)[2 * lane] = make_float4(o[0], o[1], o[2], o[3]);
                        reinterpret_cast<float4*>(dfh)[2 * lane + 1] = make_float4(o[4], o[5], o[6], o[7]);
                    }
                }
                wave_lds_fence();
                // (ii) lane = unit: fc_out's hid columns, fc_hid's bias
                if constexpr (NW) {
                    for (int tt = 0; tt < len; ++tt) {
                        const float2 d = *reinterpret_cast<const float2*>(dyb + 2 * tt);
                        const float o = x1[tt * kWS + lane];
                        dwo0 = __builtin_fmaf(d.x, o, dwo0); dwo1 = __builtin_fmaf(d.y, o, dwo1);
                        dbhid += dhid[tt * kWS + lane];
                    }
                    // dW_hid += sum over the chunk's steps of dL/dhid(t) (x) h(t): 16 x 16 x 4 tiles, K = four time steps
#pragma unroll
                    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) {
                            f32x4 t = ahid[jb][kb];
                            for (int t4 = 0; t4 < kWC / 4; ++t4) {
                                const float av = dhid[(4 * t4 + quad) * kWS + 16 * jb + col];
                                const float bv = hs[(4 * t4 + quad + 1) * kWS + 16 * kb + col];
                                t = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, t, 0, 0, 0);
                            }
                            ahid[jb][kb] = t;
                        }
                }
                wave_lds_fence();
                // (iii) lane = time step: fc_hid^T dL/dhid -> the head's dL/dh(t), into x1
                {
                    float dhh[64];
#pragma unroll
                    for (int k = 0; k < 64; ++k) dhh[k] = 0.0f;
                    for (int j = 0; j < H; ++j) {
                        const float d = dhid[lane * kWS + j];
                        const float4* w4 = reinterpret_cast<const float4*>(whp + j * 64);
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const float4 w = w4[q];
                            dhh[4 * q] = __builtin_fmaf(d, w.x, dhh[4 * q]); dhh[4 * q + 1] = __builtin_fmaf(d, w.y, dhh[4 * q + 1]);
                            dhh[4 * q + 2] = __builtin_fmaf(d, w.z, dhh[4 * q + 2]); dhh[4 * q + 3] = __builtin_fmaf(d, w.w, dhh[4 * q + 3]);
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 64; ++k) x1[lane * kWS + k] = dhh[k];
                }
                wave_lds_fence();
            }
            // ---- the chunk's steps in reverse, lane = unit (the next step's record is in flight while this one is worked on) ----
            float rn, zn, nn, gn_;
            {
                const float* s = sv + (size_t)(t0 + len - 1) * NS * 64 + lane;
                rn = s[0]; zn = s[64]; nn = s[128]; gn_ = s[192];
            }
            for (int tt = len - 1; tt >= 0; --tt) {
                const float r = rn, z = zn, n = nn, ghn = gn_;
                if (tt > 0) {
                    const float* s = sv + (size_t)(t0 + tt - 1) * NS * 64 + lane;
                    rn = s[0]; zn = s[64]; nn = s[128]; gn_ = s[192];
                }
                const float hp = hs[tt * kWS + lane], ht = hs[(tt + 1) * kWS + lane];
                const float2 d = *reinterpret_cast<const float2*>(dyb + 2 * tt);
                float dht = dh;
                if constexpr (DG) dht += x1[tt * kWS + lane];
                else {
                    dht = __builtin_fmaf(d.x, wo0, __builtin_fmaf(d.y, wo1, dht));
                    if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, ht, dwo0); dwo1 = __builtin_fmaf(d.y, ht, dwo1); }
                }
                // cell backward: h = (1 - z) n + z h(t-1)
                const float dn = dht * (1.0f - z), dz = dht * (hp - n);
                const float dnp = vo ? dn * __builtin_fmaf(-n, n, 1.0f) : 0.0f;
                const float drp = (dnp * ghn) * (r * (1.0f - r));
                const float dzp = vo ? dz * (z * (1.0f - z)) : 0.0f;
                const float dghn = dnp * r;
                dgb[lane] = drp; dgb[64 + lane] = dzp; dgb[128 + lane] = dghn;
                wave_lds_fence();
                // dL/dh(t-1)[k] = dL/dh(t)[k] z[k] + sum_j (d_r[j] W_hr[j][k] + d_z[j] W_hz[j][k] + d_hn[j] W_hn[j][k])
                float dhn = dht * z;
                {
                    const float* w0 = pl + L.o_w_hh + lane;
                    const int HH = H * H;
                    const int kk = vo ? 0 : -lane;           // (lanes beyond H read column 0: finite values, result discarded)
                    for (int j4 = 0; j4 < H; j4 += 4) {
                        const float4 gr = *reinterpret_cast<const float4*>(dgb + j4), gz = *reinterpret_cast<const float4*>(dgb + 64 + j4),
                                     gn = *reinterpret_cast<const float4*>(dgb + 128 + j4);
                        const float grv[4] = {gr.x, gr.y, gr.z, gr.w}, gzv[4] = {gz.x, gz.y, gz.z, gz.w}, gnv[4] = {gn.x, gn.y, gn.z, gn.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int j = min(j4 + e, H - 1);      // (rows beyond H: gate gradients are zero there)
                            const float* wr = w0 + j * H + kk;
                            dhn = __builtin_fmaf(grv[e], wr[0], dhn); dhn = __builtin_fmaf(gzv[e], wr[HH], dhn); dhn = __builtin_fmaf(gnv[e], wr[2 * HH], dhn);
                        }
                    }
                }
                dh = vo ? dhn : 0.0f;
                if constexpr (NW) {
                    // dW_hh: block b of rotation r = units 16 b .. (rows) x units 16 ((b + r) % 4) .. (columns)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float hpr = rr == 0 ? hp : __shfl(hp, (lane + 16 * rr) & 63);
                        acc[0][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(drp, hpr, acc[0][rr], 0, 0, 0);
                        acc[1][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dzp, hpr, acc[1][rr], 0, 0, 0);
                        acc[2][rr] = __builtin_amdgcn_mfma_f32_16x16x1f32(dghn, hpr, acc[2][rr], 0, 0, 0);
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
                        float v = __builtin_fmaf(drp, wih[0][i], __builtin_fmaf(dzp, wih[1][i], dnp * wih[2][i]));
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                        df[i] = v;
                        if constexpr (DG) df[i] += dfh[tt * 8 + i];
                    }
                    float dI, dQ;
                    feat_bwd<FM>(fe[0], fe[1], df, dI, dQ);
                    if (lane == 0) { dfh[tt * 8 + 6] = dI; dfh[tt * 8 + 7] = dQ; }
                }
                wave_lds_fence();
            }
            if constexpr (DX) {
                wave_lds_fence();
                if (lane < len) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t0 + lane] = make_float2(dfh[lane * 8 + 6], dfh[lane * 8 + 7]);
            }
        }
        wave_lds_fence();
    }
    if constexpr (NW) {
        // ---- the workgroup's row of partial gradients (every entry written) ----
        float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
        for (int i = lane; i < L.P + kLossCols; i += 64) prow[i] = 0.0f;
        __builtin_amdgcn_s_waitcnt(0);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 14; ++i)
            for (int o = 32; o > 0; o >>= 1) tacc[i] += __shfl_xor(tacc[i], o);
        if (lane == 0) {
            prow[L.o_b_out] = tacc[0]; prow[L.o_b_out + 1] = tacc[1];
            if constexpr (DG) {
#pragma unroll
                for (int i = 0; i < 6; ++i) { prow[L.o_w_out + H + i] = tacc[2 + i]; prow[L.o_w_out + OW + H + i] = tacc[8 + i]; }
            }
        }
        if (vo) {
            prow[L.o_w_out + lane] = dwo0; prow[L.o_w_out + OW + lane] = dwo1;
            if constexpr (DG) prow[L.o_b_hid + lane] = dbhid;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int i = 0; i < F; ++i) prow[L.o_w_ih + (g * H + lane) * F + i] = dwih[g][i];
                prow[L.o_b_ih + g * H + lane] = dbs[g];
                prow[L.o_b_hh + g * H + lane] = g < 2 ? dbs[g] : dbs[3];
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
                        const int ju = 16 * bb + 4 * quad + i, ku = 16 * ((bb + rr) & 3) + col;
                        if (ju < H && ku < H) prow[L.o_w_hh + (g * H + ju) * H + ku] = acc[g][rr][4 * bb + i];
                    }
        if constexpr (DG) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int ju = 16 * jb + 4 * quad + i, ku = 16 * kb + col;
                        if (ju < H && ku < H) prow[L.o_w_hid + ju * H + ku] = ahid[jb][kb][i];
                    }
        }
    }
}

bool wide_cfg(const odpd_model_t* m, int& FM, bool& DG) {
    switch (m->backbone) {
    case ODPD_GRU: FM = FEAT_RAW2; DG = false; return true;
    case ODPD_DGRU: FM = FEAT_DGRU6; DG = true; return true;
    case ODPD_QGRU: FM = FEAT_Q4; DG = false; return true;
    case ODPD_QGRU_AMP1: FM = FEAT_A4; DG = false; return true;
    default: return false;
    }
}
int wide_P(const odpd_model_t* m, int FM, bool DG) { return gru_layout(m->hidden, FM == FEAT_RAW2 ? 2 : (FM == FEAT_DGRU6 ? 6 : 4), DG).P; }
template <typename K>
int wide_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// float gru / dgru / qgru / qgru_amp1 of 33 .. 64 hidden units
bool gru_wide_ok(const odpd_model_t* m) {
    int FM; bool DG;
    return m->bits_w == 0 && m->hidden > 32 && m->hidden <= 64 && wide_cfg(m, FM, DG);
}
int64_t gru_wide_ckpt_floats(const odpd_model_t* m, int B, int T) { return (int64_t)B * T * (m->backbone == ODPD_DGRU ? 6 : 5) * 64; }
int gru_wide_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int gru_wide_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM; bool DG;
    if (!gru_wide_ok(m) || !wide_cfg(m, FM, DG)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)wide_fwd_floats(wide_P(m, FM, DG), DG, m->hidden) * sizeof(float);
    const int grid = gru_wide_rows(m, a.B);
#define ODPD_WIDE_FWD(FM_, DG_) \
    if (FM == FM_) return a.ckpt ? wide_launch(st, wide_gru_fwd_kernel<FM_, DG_, true>, grid, lds, a) : wide_launch(st, wide_gru_fwd_kernel<FM_, DG_, false>, grid, lds, a);
    ODPD_WIDE_FWD(FEAT_RAW2, false) ODPD_WIDE_FWD(FEAT_DGRU6, true) ODPD_WIDE_FWD(FEAT_Q4, false) ODPD_WIDE_FWD(FEAT_A4, false)
#undef ODPD_WIDE_FWD
    return ODPD_EUNSUPPORTED;
}
int gru_wide_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    int FM; bool DG;
    if (!gru_wide_ok(m) || !wide_cfg(m, FM, DG)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)wide_bwd_floats(wide_P(m, FM, DG), DG, m->hidden) * sizeof(float);
    const int grid = gru_wide_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
#define ODPD_WIDE_BWD(FM_, DG_)                                                                              \
    if (FM == FM_) {                                                                                         \
        if (nw && dx) return wide_launch(st, wide_gru_bwd_kernel<FM_, DG_, true, true>, grid, lds, a);       \
        if (nw) return wide_launch(st, wide_gru_bwd_kernel<FM_, DG_, true, false>, grid, lds, a);            \
        return wide_launch(st, wide_gru_bwd_kernel<FM_, DG_, false, true>, grid, lds, a);                    \
    }
    ODPD_WIDE_BWD(FEAT_RAW2, false) ODPD_WIDE_BWD(FEAT_DGRU6, true) ODPD_WIDE_BWD(FEAT_Q4, false) ODPD_WIDE_BWD(FEAT_A4, false)
#undef ODPD_WIDE_BWD
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
