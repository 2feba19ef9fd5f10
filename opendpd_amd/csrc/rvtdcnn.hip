// rvtdcnn.hip — real-valued time-delay CNN backbone (reference backbones/rvtdcnn.py:9-62, built by models.py:80-81 with
// fc_hid_size = hidden_size; window 4, 3 conv channels -> 39 H + 32 parameters, 1007 at H = 25).
//
// Per sample t a 4 x 5 patch: row w = features [I, Q, a, a^2, a^3] of sample t-3+w, the frame's own LAST three samples
// standing in front of it (rvtdcnn.py:51-53: circular).  Conv2d(1->3, k3, padding (1,0)) -> tanh -> 36 values (channel, row,
// column) -> Linear(36 -> H) -> tanh -> Linear(H -> 2).  Not recurrent: time is parallel.
//
// Mapping: ONE LANE PER SAMPLE, everything of a sample in registers (patch 20, conv outputs 36, hidden <= 32).  Every weight
// is wave-uniform, so the weights never occupy VGPRs or LDS: they are read through the constant address space (s_load) in their
// native order and enter v_fmac as the scalar operand (the GMP pattern, csrc/gmp.hip).  The fp32 matrix instruction runs at
// the vector rate on gfx950, so the per-sample mat-vecs gain nothing from it; it is used where it removes data movement — the
// weight gradients, which are contractions over SAMPLES:
//     dW_hid[u][k] = sum_s dhp_s[u] z_s[k],  db_hid[u] = sum_s dhp_s[u],  dW_out[c][u] = sum_s dy_s[c] hid_s[u]
// Each wave bounces its samples' rows [dhp | z | 1 | dy | hid] through a private LDS tile, 16 samples at a time, and accumulates
// the 16 x 16 output tiles with v_mfma_f32_16x16x4_f32 (exact fp32; the sample index is K).  The 27 + 3 convolution gradients
// and the fc_out bias gradient are per-lane accumulators reduced once at the end.  One partials row per workgroup.
// dL/dx: the patch gradient of sample t belongs to samples t-3 .. t: the threads of a workgroup cover a frame chunk plus a
// 3-sample halo and exchange the 4 x 5 patch gradients through LDS (gather form, fixed order).
#include "odpd_s16.h"

namespace odpd {
namespace {

constexpr int kRvThreads = 256;
constexpr int kRvZ = 36;                 // conv outputs per sample = fc_hid in_features (rvtdcnn.py:17)
typedef const __attribute__((address_space(4))) float* WPtr;
// a fresh name for the weight pointer: keeps the scalar loads of one weight row next to their use instead of hoisted out of
// the sample loop (the 36 H weights of fc_hid would otherwise be live in SGPRs at once and spill)
__device__ __forceinline__ WPtr rv_fresh(WPtr w) { asm volatile("" : "+s"(w)); return w; }

struct RvLayout { int H, o_wh, o_bh, o_wo, o_bo, P; };
__host__ __device__ inline RvLayout rv_layout(int H) {
    RvLayout L; L.H = H;
    L.o_wh = 30; L.o_bh = 30 + kRvZ * H; L.o_wo = L.o_bh + H; L.o_bo = L.o_wo + 2 * H; L.P = L.o_bo + 2;
    return L;
}
// Per-wave LDS columns (sample = lane is the fast index; stride 68 = 4 mod 32 floats: the MFMA operand reads of lanes
// (element m, sample 4c+k) land on banks 4m + k + const, 2-way at most): hid[u][s] (written by the forward row loop, read back by
// the backward row loop AND as MFMA operands), z[k][s], dy[c][s].
constexpr int kRvCol = 68;
template <int HT> struct RvLds {
    static constexpr int oHid = 0, oZ = 16 * HT * kRvCol, oDy = oZ + kRvZ * kRvCol, wave_floats = oDy + 2 * 64;
};

// tanh with relative accuracy near 0 (the conv pre-activations scale with the signal amplitude): polynomial below 0.3 blended
// arithmetically (no compare / select: v_cmp + v_cndmask through vcc costs ~21 cycles, profiles/r01/ubench_issue_costs.md)
__device__ __forceinline__ float rv_tanh(float x) {
    const float x2 = x * x;
    float p = __builtin_fmaf(x2, 0.021869488536155203f, -0.053968253968253971f);
    p = __builtin_fmaf(x2, p, 0.13333333333333333f);
    p = __builtin_fmaf(x2, p, -0.33333333333333333f);
    p = __builtin_fmaf(x2 * x, p, x);
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f;
    const float t = __builtin_fmaf(fast_rcp(e), -2.0f, 1.0f);
    const float w = __builtin_amdgcn_fmed3f(__builtin_fmaf(__builtin_fabsf(x), -0x1p100f, 0.3f * 0x1p100f), 0.0f, 1.0f);
    return __builtin_fmaf(w, p - t, t);
}

// start of frame b in float2 units: (B,T,2) tensor row, or a window of a resident stream (SeqArgs::frame_idx)
__device__ __forceinline__ size_t rv_base(const SeqArgs& a, int b) {
    return a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * a.T;
}
__device__ __forceinline__ void rv_patch(const float2* x2, size_t base, int t, int T, float (&in)[4][5], float2 (&xv)[4]) {
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int s = t - 3 + w;
        s += s < 0 ? T : 0;                                          // rvtdcnn.py:51-52: the frame's last samples in front
        xv[w] = x2[base + s];
        const float a2 = __builtin_fmaf(xv[w].x, xv[w].x, xv[w].y * xv[w].y);
        const float am = __builtin_amdgcn_sqrtf(a2);
        in[w][0] = xv[w].x; in[w][1] = xv[w].y; in[w][2] = am; in[w][3] = a2; in[w][4] = am * am * am;   // rvtdcnn.py:41-46
    }
}

// z = tanh(Conv2d(patch)) (rvtdcnn.py:57-58); the 30 conv parameters are scalar operands, loaded here and dead afterwards
__device__ __forceinline__ void rv_conv(WPtr w0, const float (&in)[4][5], float (&z)[kRvZ]) {
    WPtr w = rv_fresh(w0);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float acc = w[27 + c];
#pragma unroll
                for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj)
                        if (r + dw - 1 >= 0 && r + dw - 1 <= 3)               // zero padding of the window rows
                            acc = __builtin_fmaf(w[(c * 3 + dw) * 3 + dj], in[r + dw - 1][j + dj], acc);
                z[(c * 4 + r) * 3 + j] = rv_tanh(acc);
            }
}

// one row of fc_hid with what goes with unit u: its bias and the two fc_out weights — 39 scalars, loaded one row AHEAD of their use
// (the row loops are run-time loops: the next row's s_loads are in flight while the current row's 36 FMAs issue)
struct RvRowW { float w[kRvZ], b, o0, o1; };
typedef float rv_f16 __attribute__((ext_vector_type(16), aligned(4)));
typedef float rv_f4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void rv_load_row(WPtr w, const RvLayout& L, int u, RvRowW& r) {
    // 36 consecutive weights as s_load_dwordx16 x 2 + s_load_dwordx4 (scalar loads need dword alignment only; left to itself the
    // compiler issues 36 single-dword loads with an address computation each when the row index is a run-time value)
    WPtr wr = w + 30 + u * kRvZ;
    const rv_f16 v0 = *reinterpret_cast<const __attribute__((address_space(4))) rv_f16*>(wr);
    const rv_f16 v1 = *reinterpret_cast<const __attribute__((address_space(4))) rv_f16*>(wr + 16);
    const rv_f4 v2 = *reinterpret_cast<const __attribute__((address_space(4))) rv_f4*>(wr + 32);
#pragma unroll
    for (int k = 0; k < 16; ++k) { r.w[k] = v0[k]; r.w[16 + k] = v1[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) r.w[32 + k] = v2[k];
    r.b = w[L.o_bh + u]; r.o0 = w[L.o_wo + u]; r.o1 = w[L.o_wo + L.H + u];
}
// hid_u = tanh(b_u + W_u . z) with four partial sums (a single accumulator is a 36-deep dependent chain)
__device__ __forceinline__ float rv_hid(const RvRowW& r, const float (&z)[kRvZ]) {
    float a0 = r.b, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
    for (int k = 0; k < kRvZ; k += 4) {
        a0 = __builtin_fmaf(r.w[k], z[k], a0); a1 = __builtin_fmaf(r.w[k + 1], z[k + 1], a1);
        a2 = __builtin_fmaf(r.w[k + 2], z[k + 2], a2); a3 = __builtin_fmaf(r.w[k + 3], z[k + 3], a3);
    }
    return rv_tanh((a0 + a1) + (a2 + a3));
}

// Row loop with the scalar loads one row ahead.  Scalar loads return out of order, so any use of one waits for ALL outstanding
// ones (s_waitcnt lgkmcnt(0)): a stage is therefore [wait for the current row] [issue the next row's loads] [36 FMAs of the
// current row], with scheduling barriers so that the loads are not sunk below the arithmetic they are meant to overlap.  Two
// register sets ping-pong (no copies); body(row, u) sees rows 0 .. H-1 in order.
template <class Body>
__device__ __forceinline__ void rv_rows(WPtr w, const RvLayout& L, Body&& body) {
    RvRowW ra, rb;
    rv_load_row(w, L, 0, ra);
    for (int u = 0; u < L.H; u += 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rv_load_row(w, L, min(u + 1, L.H - 1), rb);
        __builtin_amdgcn_sched_barrier(0);
        body(ra, u);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rv_load_row(w, L, min(u + 2, L.H - 1), ra);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < L.H) body(rb, u + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(kRvThreads, 6) void rv_fwd_kernel(SeqArgs a) {
    const RvLayout L = rv_layout(a.H);
    const WPtr w = (WPtr)a.params;
    const float2* x2 = reinterpret_cast<const float2*>(a.x);
    float2* y2 = reinterpret_cast<float2*>(a.y);
    const long long N = (long long)a.B * a.T;
    for (long long i = (long long)blockIdx.x * kRvThreads + threadIdx.x; i < N; i += (long long)gridDim.x * kRvThreads) {
        const int b = (int)(i / a.T), t = (int)(i - (long long)b * a.T);
        float in[4][5], z[kRvZ];
        float2 xv[4];
        rv_patch(x2, (size_t)b * a.T, t, a.T, in, xv);
        rv_conv(w, in, z);
        float y0 = w[L.o_bo], y1 = w[L.o_bo + 1];
        rv_rows(w, L, [&](const RvRowW& r, int) {
            const float h = rv_hid(r, z);
            y0 = __builtin_fmaf(r.o0, h, y0);
            y1 = __builtin_fmaf(r.o1, h, y1);
        });
        y2[i] = make_float2(y0, y1);
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------
// thread -> sample.  !DX: flat over (frame, t), every thread owns a sample.  DX: a workgroup pass serves G groups of RH = R + 3
// threads; group g = frame chunk [c0, c0 + len) plus the 3 samples after it (circularly): the halo threads compute their patch
// gradients for the chunk's last samples but own nothing (no output, no weight-gradient contribution).
struct RvGeom { int R, RH, G, nchunk, nitems, npass; };
static RvGeom rv_geom(int B, int T, bool dx) {
    RvGeom g;
    if (!dx) {
        g.R = g.RH = kRvThreads; g.G = 1; g.nchunk = 1; g.nitems = 0;
        g.npass = (int)(((long long)B * T + kRvThreads - 1) / kRvThreads);
        return g;
    }
    g.R = T < kRvThreads - 3 ? T : kRvThreads - 3;
    g.RH = g.R + 3;
    g.G = kRvThreads / g.RH;
    g.nchunk = (T + g.R - 1) / g.R;
    g.nitems = B * g.nchunk;
    g.npass = (g.nitems + g.G - 1) / g.G;
    return g;
}

// FUSED: a.target holds the target, loss and dL/dy are formed here (train step);  else a.dy holds dL/dy
// NW: weight-gradient partials (one row per workgroup)      DX: dL/dx
template <int HT, bool FUSED, bool NW, bool DX>
__global__ __launch_bounds__(kRvThreads, 2) void rv_bwd_kernel(SeqArgs a, RvGeom g) {
    using Lds = RvLds<HT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RvLayout L = rv_layout(a.H);
    const WPtr w0 = (WPtr)a.params;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    float* wl = smem + (size_t)wave * Lds::wave_floats;
    float* hidc = wl + Lds::oHid + lane;          // [u * kRvCol]
    float* zc = wl + Lds::oZ + lane;              // [k * kRvCol]
    float* dyc = wl + Lds::oDy;                   // [c * 64 + sample]
    float* din_x = smem + Lds::oZ;                // DX: patch gradients [thread][21], over the z columns once the MFMAs have read them
    const float2* x2 = reinterpret_cast<const float2*>(a.x);
    const float2* d2 = reinterpret_cast<const float2*>(FUSED ? a.target : a.dy);
    float2* dx2 = reinterpret_cast<float2*>(a.dx);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);

    f32x4 Dz[HT][3], Dh[HT];
    float dK[27], dkb[3], dbo[2] = {0.f, 0.f}, loss_acc = 0.0f;
    float wo0[HT], wo1[HT];                       // fc_out weights of the units this lane feeds to the MFMA (element n of tile mt)
#pragma unroll
    for (int i = 0; i < 27; ++i) dK[i] = 0.0f;
    dkb[0] = dkb[1] = dkb[2] = 0.0f;
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
        Dh[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) Dz[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int u = 16 * mt + n;
        wo0[mt] = u < L.H ? a.params[L.o_wo + u] : 0.0f;
        wo1[mt] = u < L.H ? a.params[L.o_wo + L.H + u] : 0.0f;
    }
    // B-operand selectors of the third z tile (columns 32 + n): z[32..35], then the constant 1 (bias gradient), then nothing
    const int zk2 = n < 4 ? 32 + n : 35;
    const float zmul2 = n < 4 ? 1.0f : 0.0f, zadd2 = n == 4 ? 1.0f : 0.0f, ymask = n < 2 ? 1.0f : 0.0f;
    if constexpr (NW) {       // units H .. 16 HT - 1 are never written by the row loops: their hid columns must read as numbers
        for (int u = L.H; u < 16 * HT; ++u) hidc[u * kRvCol] = 0.0f;
    }
    const long long N = (long long)a.B * a.T;

    for (int pass = blockIdx.x; pass < g.npass; pass += gridDim.x) {
        // ---- which sample ----
        bool active, owner;
        int b = 0, t = 0;
        if constexpr (DX) {
            const int grp = tid / g.RH, i = tid - grp * g.RH, item = pass * g.G + grp;
            const int c = item % g.nchunk, c0 = c * g.R, len = min(g.R, a.T - c0);
            b = item / g.nchunk;
            active = grp < g.G && item < g.nitems && i < len + 3;
            owner = active && i < len;
            t = c0 + i;
            t -= t >= a.T ? a.T : 0;
            if (!active) { b = 0; t = 0; }
        } else {
            const long long s = (long long)pass * kRvThreads + tid;
            active = owner = s < N;
            if (active) { b = (int)(s / a.T); t = (int)(s - (long long)b * a.T); }
        }
        float z[kRvZ], in[4][5], dy0, dy1;
        float2 xv[4];
        const size_t base = rv_base(a, b);
        rv_patch(x2, base, t, a.T, in, xv);
        rv_conv(w0, in, z);
        // ---- forward row loop: hid_u to its LDS column, y on the fly ----
        if constexpr (DX) __syncthreads();             // the previous pass's patch-gradient gathers (over the z columns) are done
        else wave_lds_fence();                         // ... MFMA operand reads
        if constexpr (NW && DX) {     // the patch-gradient exchange also ran over the padded hid columns: make them numbers again
            for (int u = L.H; u < 16 * HT; ++u) hidc[u * kRvCol] = 0.0f;
        }
        {
            float y0 = w0[L.o_bo], y1 = w0[L.o_bo + 1];
            rv_rows(w0, L, [&](const RvRowW& r, int u) {
                const float h = rv_hid(r, z);
                hidc[u * kRvCol] = h;
                y0 = __builtin_fmaf(r.o0, h, y0);
                y1 = __builtin_fmaf(r.o1, h, y1);
            });
            const float2 dv = d2[base + t];
            if constexpr (FUSED) {
                float l = 0.0f;
                s16_loss(lossc, y0 - dv.x, y1 - dv.y, dy0, dy1, l);
                loss_acc += owner ? l : 0.0f;
            } else {
                dy0 = dv.x; dy1 = dv.y;
            }
            if (!active) { dy0 = 0.0f; dy1 = 0.0f; }       // an idle thread recomputes sample (0,0) and contributes zeros
        }
        // ---- weight gradients of fc_hid / fc_out: contraction over the wave's 64 samples on the MFMA, operands straight from
        //      the LDS columns; dhp = (W_out^T dy) (1 - hid^2) is formed in the operand layout (element n of a tile, sample 4c+q)
        if constexpr (NW) {
            const float own = owner ? 1.0f : 0.0f;
            dbo[0] = __builtin_fmaf(own, dy0, dbo[0]); dbo[1] = __builtin_fmaf(own, dy1, dbo[1]);
#pragma unroll
            for (int k = 0; k < kRvZ; ++k) zc[k * kRvCol] = z[k];
            dyc[lane] = own * dy0; dyc[64 + lane] = own * dy1;     // halo / idle samples contribute zero rows
            wave_lds_fence();
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const int s = 4 * c + q;
                const float e0 = dyc[s], e1 = dyc[64 + s];
                const float ay = ymask * dyc[(n & 1) * 64 + s];
                float hv[HT], ad[HT], bz[3];
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) {
                    hv[mt] = wl[Lds::oHid + (16 * mt + n) * kRvCol + s];
                    ad[mt] = __builtin_fmaf(wo0[mt], e0, wo1[mt] * e1) * __builtin_fmaf(-hv[mt], hv[mt], 1.0f);
                }
                bz[0] = wl[Lds::oZ + n * kRvCol + s];
                bz[1] = wl[Lds::oZ + (16 + n) * kRvCol + s];
                bz[2] = __builtin_fmaf(wl[Lds::oZ + zk2 * kRvCol + s], zmul2, zadd2);
#pragma unroll
                for (int mt = 0; mt < HT; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) Dz[mt][nt] = mfma4(ad[mt], bz[nt], Dz[mt][nt]);
                    Dh[mt] = mfma4(ay, hv[mt], Dh[mt]);
                }
            }
        }
        // ---- backward row loop: dL/dz = W_hid^T dhp (36 independent accumulators), through tanh ----
        float dc[kRvZ];
#pragma unroll
        for (int k = 0; k < kRvZ; ++k) dc[k] = 0.0f;
        rv_rows(w0, L, [&](const RvRowW& r, int u) {
            const float h = hidc[u * kRvCol];
            const float dhp = __builtin_fmaf(r.o0, dy0, r.o1 * dy1) * __builtin_fmaf(-h, h, 1.0f);
#pragma unroll
            for (int k = 0; k < kRvZ; ++k) dc[k] = __builtin_fmaf(r.w[k], dhp, dc[k]);
        });
#pragma unroll
        for (int k = 0; k < kRvZ; ++k) dc[k] *= __builtin_fmaf(-z[k], z[k], 1.0f);
        // ---- convolution: weight gradients (per-lane accumulators) and patch gradient ----
        if constexpr (NW) {
            if (owner) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float d = dc[(c * 4 + r) * 3 + j];
                            dkb[c] += d;
#pragma unroll
                            for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                                for (int dj = 0; dj < 3; ++dj)
                                    if (r + dw - 1 >= 0 && r + dw - 1 <= 3)
                                        dK[(c * 3 + dw) * 3 + dj] = __builtin_fmaf(d, in[r + dw - 1][j + dj], dK[(c * 3 + dw) * 3 + dj]);
                        }
            }
        }
        if constexpr (DX) {
            float din[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int f = 0; f < 5; ++f) din[r][f] = 0.0f;
            {
                WPtr w = rv_fresh(w0);
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
#pragma unroll
                            for (int dw = 0; dw < 3; ++dw)
#pragma unroll
                                for (int dj = 0; dj < 3; ++dj)
                                    if (r + dw - 1 >= 0 && r + dw - 1 <= 3)
                                        din[r + dw - 1][j + dj] = __builtin_fmaf(w[(c * 3 + dw) * 3 + dj], dc[(c * 4 + r) * 3 + j], din[r + dw - 1][j + dj]);
            }
            __syncthreads();                                   // every wave's MFMAs have read the z columns
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int f = 0; f < 5; ++f) din_x[tid * 21 + r * 5 + f] = din[r][f];
            __syncthreads();
            if (owner) {
                // window row r of sample t+3-r is sample t (thread tid + 3 - r of the same group)
                float df[5];
#pragma unroll
                for (int f = 0; f < 5; ++f) {
                    float v = din_x[(tid + 3) * 21 + f];
#pragma unroll
                    for (int r = 1; r < 4; ++r) v += din_x[(tid + 3 - r) * 21 + r * 5 + f];
                    df[f] = v;
                }
                // features [I, Q, a, a^2, a^3]: da/dI = I/a, da^2/dI = 2 I, da^3/dI = 3 a I
                const float I = xv[3].x, Q = xv[3].y, am = in[3][2];
                const float ga = __builtin_fmaf(df[2], fast_rcp(am), __builtin_fmaf(3.0f * am, df[4], 2.0f * df[3]));
                dx2[(size_t)b * a.T + t] = make_float2(__builtin_fmaf(ga, I, df[0]), __builtin_fmaf(ga, Q, df[1]));
            }
        }
    }
    if constexpr (!NW) return;
    // ---- one row of partial gradients per workgroup (fixed summation order) ----
    const int P4 = L.P + kLossCols;
    __syncthreads();
    float* prow = smem + (size_t)wave * P4;
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = 16 * mt + 4 * q + i;                 // Dz[mt][nt][i] of lane (n,q) = D[u][k = 16 nt + n]
            if (u < L.H) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    const int k = 16 * nt + n;
                    if (k < kRvZ) prow[L.o_wh + u * kRvZ + k] = Dz[mt][nt][i];
                    else if (k == kRvZ) prow[L.o_bh + u] = Dz[mt][nt][i];
                }
            }
            const int c = 4 * q + i, uo = 16 * mt + n;         // Dh[mt][i] of lane (n,q) = dW_out[c][uo]
            if (c < 2 && uo < L.H) prow[L.o_wo + c * L.H + uo] = Dh[mt][i];
        }
    }
    auto wave_sum = [](float v) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        return v;
    };
#pragma unroll
    for (int i = 0; i < 27; ++i) { const float v = wave_sum(dK[i]); if (lane == 0) prow[i] = v; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float v = wave_sum(dkb[i]); if (lane == 0) prow[27 + i] = v; }
    {
        const float v0 = wave_sum(dbo[0]), v1 = wave_sum(dbo[1]), lp = wave_sum(loss_acc);
        if (lane == 0) {
            prow[L.o_bo] = v0; prow[L.o_bo + 1] = v1;
            prow[L.P] = lp; prow[L.P + 1] = 0.0f; prow[L.P + 2] = 0.0f; prow[L.P + 3] = 0.0f;
        }
    }
    __syncthreads();
    float* out = a.partials + (size_t)blockIdx.x * P4;
    for (int i = tid; i < P4; i += kRvThreads) out[i] = (smem[i] + smem[P4 + i]) + (smem[2 * P4 + i] + smem[3 * P4 + i]);
}

template <int HT> size_t rv_lds_bytes(int P, bool nw, bool dx) {
    size_t n = (size_t)(kRvThreads / 64) * RvLds<HT>::wave_floats;
    const size_t ex = dx ? (size_t)RvLds<HT>::oZ + (size_t)(kRvThreads + 3) * 21 : 0;     // patch-gradient exchange over the z columns
    const size_t rows = nw ? (size_t)(kRvThreads / 64) * (P + kLossCols) : 0;
    n = n > ex ? n : ex;
    return (n > rows ? n : rows) * sizeof(float);
}
inline bool rv_ok(const odpd_model_t* m, int T) { return m->hidden >= 1 && m->hidden <= 32 && T >= 3; }
inline int rv_grid(int npass, int blocks_per_cu = 2) {
    const int cap = device_cus() * blocks_per_cu;
    return npass < cap ? (npass < 1 ? 1 : npass) : cap;
}

template <int HT, bool FUSED>
int rv_launch_bwd(hipStream_t st, const SeqArgs& a, bool nw, bool dx) {
    const RvGeom g = rv_geom(a.B, a.T, dx);
    const int P = rv_layout(a.H).P;
    const size_t lds = rv_lds_bytes<HT>(P, nw, dx);
    const int grid = nw ? rvtdcnn_rows_for(a.B, a.T, dx) : rv_grid(g.npass);
    auto go = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(grid), dim3(kRvThreads), lds, st, a, g);
        return (int)hipGetLastError();
    };
    if (nw && dx) return go(rv_bwd_kernel<HT, FUSED, true, true>);
    if (nw) return go(rv_bwd_kernel<HT, FUSED, true, false>);
    return go(rv_bwd_kernel<HT, FUSED, false, true>);
}

}  // namespace

// rows of partials = workgroups of the backward / fused launch for this shape (dx: the chunk + halo thread layout)
int rvtdcnn_rows_for(int B, int T, bool dx) { return rv_grid(rv_geom(B, T, dx).npass); }

// bits_w > 0: the quantised model (INT_Conv2D + INT_Linear layers) on its own kernels, csrc/rvtdcnn_q.hip
int rvtdcnn_rows(const odpd_model_t* m, int B, int T) {
    if (m->bits_w > 0) return rvtdcnn_q_ok(m, T) ? rvtdcnn_q_rows(m, B, T) : ODPD_EUNSUPPORTED;
    if (!rv_ok(m, T)) return ODPD_EUNSUPPORTED;
    // the split backward may be asked for dL/dx as well: size for the larger of the two grids
    const int r0 = rvtdcnn_rows_for(B, T, false), r1 = rvtdcnn_rows_for(B, T, true);
    return r0 > r1 ? r0 : r1;
}

int rvtdcnn_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->bits_w > 0) return rvtdcnn_q_fwd(st, m, a);
    if (!rv_ok(m, a.T)) return ODPD_EUNSUPPORTED;
    const long long N = (long long)a.B * a.T;
    const int grid = rv_grid((int)((N + kRvThreads - 1) / kRvThreads), 6);       // 59 VGPRs: six waves per SIMD hide the scalar loads
    hipLaunchKernelGGL(rv_fwd_kernel, dim3(grid), dim3(kRvThreads), 0, st, a);
    return (int)hipGetLastError();
}

// dy -> partials (weight gradients) and / or dx.  A partials buffer sized by rvtdcnn_rows() may have more rows than this launch's
// grid writes: the unused rows are zeroed so that odpd_reduce_partials can sum all of them.
int rvtdcnn_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (a.partials == nullptr && a.dx == nullptr) return ODPD_EINVAL;
    if (m->bits_w > 0) return rvtdcnn_q_bwd(st, m, a, false);
    if (!rv_ok(m, a.T)) return ODPD_EUNSUPPORTED;
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
    if (nw) {
        const int used = rvtdcnn_rows_for(a.B, a.T, dx), all = rvtdcnn_rows(m, a.B, a.T);
        const size_t P4 = rv_layout(a.H).P + kLossCols;
        if (all > used) ODPD_CHECK_HIP(hipMemsetAsync(a.partials + (size_t)used * P4, 0, (size_t)(all - used) * P4 * sizeof(float), st));
    }
    return m->hidden <= 16 ? rv_launch_bwd<1, false>(st, a, nw, dx) : rv_launch_bwd<2, false>(st, a, nw, dx);
}

// fused train step: forward + loss + weight gradients in one launch; frames may be addressed inside resident streams
int rvtdcnn_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!a.partials || !a.target) return ODPD_EINVAL;
    if (m->bits_w > 0) return rvtdcnn_q_bwd(st, m, a, true);
    if (!rv_ok(m, a.T)) return ODPD_EUNSUPPORTED;
    return m->hidden <= 16 ? rv_launch_bwd<1, true>(st, a, true, false) : rv_launch_bwd<2, true>(st, a, true, false);
}
int rvtdcnn_train_rows(const odpd_model_t* m, int B, int T) {
    if (m->bits_w > 0) return rvtdcnn_q_ok(m, T) ? rvtdcnn_q_rows(m, B, T) : ODPD_EUNSUPPORTED;
    if (!rv_ok(m, T)) return ODPD_EUNSUPPORTED;
    return rvtdcnn_rows_for(B, T, false);
}

}  // namespace odpd
