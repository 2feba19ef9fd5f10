// gmp.hip — generalised memory polynomial backbone (reference backbones/gmp.py:5-50, built as GMP() by models.py:26-28:
// memory_length M = 11, degree 5 -> 495 real weights on complex terms).
//
// The reference writes y[t] = sum_m u[t+m] (w0[m] + sum_{d,i} w[d,i,m] amp[t+i+m]^d) over a signal delayed by M-1 and an
// envelope delayed by 2(M-1) (gmp.py:26-33: zero history).  In sample coordinates, with r = M-1-m, k = M-1-i and
// P_d[s] = |x[s]|^d (zero before the frame start), this is
//     y[t]      = sum_r x[t-r] * Ec[t-r, r],      Ec[s, r] = w0[r] + sum_{d,k} w[d,k,r] P_d[s-k]
// and everything a sample s contributes to is a function of ITS OWN look-back envelope (k = 0..10) and look-ahead output
// gradient (r = 0..10):
//     dL/dw[d,k,r] = sum_s G[s,r] P_d[s-k],        G[s,r] = Re(conj(x[s]) dy[s+r])          (dL/dw0[r] = sum_s G[s,r])
//     dL/dx[s]     = sum_r dy[s+r] Ec[s,r] + (x[s]/|x[s]|) sum_d d |x[s]|^(d-1) sum_{k,r} w[d,k,r] G[s+k,r]
// Not recurrent.  A workgroup stages whole frames (time chunks of <= 512 samples with a 20-sample halo on both sides for
// long records) in LDS: the envelope powers |x|, |x|^2, |x|^3, |x|^4 as four arrays in RESIDUE-MAJOR order (entry e at plane
// e mod 4, slot e / 4), x (+ dy / target) as float2 in natural order; the next region's global loads are in flight while
// the current one is computed.  Forward: a lane owns 4 consecutive samples, so its 24-entry envelope window is read once
// (one power at a time, compile-time plane / slot offsets, conflict-free) and serves 4 x 484 FMAs at 3 waves per SIMD.  The 484 envelope weights
// are wave-uniform: they are read through the constant address space in their native (d, i, m) order — s_load_dwordx8/x2/x1
// per 11-weight row, two rows ahead of their use — and enter v_fmac as the scalar operand: no LDS or VGPR traffic for them.
// The weight gradient is the (11 x 45) contraction G^T [P | 1] over samples: v_mfma_f32_16x16x4_f32, three 16x16 tiles per
// four samples (exact fp32), operands fetched straight from the LDS arrays.
#include <utility>

#include "odpd_host.h"
#include "odpd_s16.h"

namespace odpd {
namespace {

template <class F, int... I>
__device__ __forceinline__ void gmp_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void gmp_for(F&& f) { gmp_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int kM = 11;                       // memory_length
constexpr int kNP = 4;                       // envelope powers 1 .. degree-1
constexpr int kHalo = 2 * (kM - 1);          // look-back of the forward (envelope of the oldest tap) = look-ahead of dy in dL/dx
constexpr int kGmpP = kM * (1 + kNP * kM);   // 495
constexpr int kThreads = 256;
constexpr int kGmpMaxChunk = 512;
constexpr int kGradCols = 48;                // 44 (d,k) columns + the ones column (dL/dw0) padded to three MFMA tiles
constexpr int kR = 4;                        // consecutive samples per lane in the forward
constexpr int kEPT = 8;                      // staged entries per thread at most (a region holds <= kEPT * kThreads entries)
typedef const __attribute__((address_space(4))) float* WPtr;
// a fresh name for the weight pointer: keeps the scalar loads of one weight row together instead of hoisted out of the
// sample loop (495 live SGPRs would spill)
__device__ __forceinline__ WPtr gmp_fresh(WPtr w) { asm volatile("" : "+s"(w)); return w; }
// the same, ordered after the arithmetic that produced `e` (the scheduler would otherwise collect all fresh names, and with
// them all loads, at the top of the unrolled block)
__device__ __forceinline__ WPtr gmp_fresh_after(WPtr w, float (&e)[11]) {
    asm volatile("" : "+s"(w), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(e[8]),
                 "+v"(e[9]), "+v"(e[10]));
    return w;
}

__device__ __forceinline__ WPtr gmp_fresh_after4(WPtr w, float (&e)[4]) {
    asm volatile("" : "+s"(w), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
    return w;
}

// flat index (gmp.py:41-44: [u[t+m]] then d-major, i, m) of w[d,k,r] / w0[r] in sample coordinates
__host__ __device__ constexpr int gmp_w(int d, int k, int r) { return kM + ((d * kM + (kM - 1 - k)) * kM + (kM - 1 - r)); }
__host__ __device__ constexpr int gmp_w0(int r) { return kM - 1 - r; }

// items = (frame, time chunk); a workgroup works on `NI` items at a time (one "region" of LDS), E entries each (multiple of 4:
// every item starts on plane 0); Q = slots per plane, Q mod 16 = 4 so that the four planes start 16 banks apart
struct GmpGeom { int TC, nchunk, nitems, NI, E, nregions, Q; };

struct GmpLds {
    float* P;          // P[d][plane][slot]: |x|^(d+1) of entry e at d * 4Q + (e & 3) * Q + (e >> 2)
    float2* X; float2* DY;
    int Q;
    __device__ __forceinline__ GmpLds(float* smem, const GmpGeom& g, bool with_dy) {
        P = smem;
        X = reinterpret_cast<float2*>(P + 16 * (size_t)g.Q);
        DY = with_dy ? X + (size_t)g.NI * g.E : nullptr;
        Q = g.Q;
    }
    __device__ __forceinline__ int pos(int e) const { return (e & 3) * Q + (e >> 2); }
    __device__ __forceinline__ float4 powers(int e) const {
        const float* p = P + pos(e);
        return make_float4(p[0], p[4 * Q], p[8 * Q], p[12 * Q]);
    }
};

// ---- staging: global loads of a region (issued early, consumed one region later) and their deposit in LDS ----------------
// SRC 0: x;  1: x and dy;  2: x and target of the fused train step (frames possibly addressed inside resident streams,
// SeqArgs::frame_idx) — the second stream lands in the DY array
struct GmpFetch { float2 x[kEPT], d[kEPT]; };
template <int SRC>
__device__ __forceinline__ void gmp_fetch(const SeqArgs& a, const GmpGeom& g, int reg, GmpFetch& f) {
    const float2* x2 = reinterpret_cast<const float2*>(a.x);
    const float2* d2 = reinterpret_cast<const float2*>(SRC == 2 ? a.target : a.dy);
    const int total = g.NI * g.E;
#pragma unroll
    for (int u = 0; u < kEPT; ++u) {
        const int e = (int)threadIdx.x + u * kThreads;
        const int it = e / g.E, eo = e - it * g.E;
        const int item = reg * g.NI + it;
        int b = item, c = 0;
        if (g.nchunk > 1) { b = item / g.nchunk; c = item - b * g.nchunk; }
        const int t = c * g.TC + eo - kHalo;
        // zero history (gmp.py:26-27,33) and nothing after the frame
        const bool in = e < total && item < g.nitems && t >= 0 && t < a.T;
        f.x[u] = make_float2(0.0f, 0.0f); f.d[u] = make_float2(0.0f, 0.0f);
        if (in) {
            const size_t at = (SRC == 2 && a.frame_idx != nullptr) ? (size_t)a.frame_idx[b] * a.frame_stride + t : (size_t)b * a.T + t;
            f.x[u] = x2[at];
            if constexpr (SRC != 0) f.d[u] = d2[at];
        }
    }
}
template <int SRC>
__device__ __forceinline__ void gmp_deposit(const GmpGeom& g, const GmpLds& s, const GmpFetch& f) {
    const int total = g.NI * g.E;
#pragma unroll
    for (int u = 0; u < kEPT; ++u) {
        const int e = (int)threadIdx.x + u * kThreads;
        if (e < total) {
            const float2 xv = f.x[u];
            const float am = __builtin_amdgcn_sqrtf(__builtin_fmaf(xv.x, xv.x, xv.y * xv.y));
            const float a2 = am * am;
            float* p = s.P + s.pos(e);
            p[0] = am; p[4 * s.Q] = a2; p[8 * s.Q] = a2 * am; p[12 * s.Q] = a2 * a2;
            s.X[e] = xv;
            if constexpr (SRC != 0) s.DY[e] = f.d[u];
        }
    }
}
// region loop shared by the kernels: body(reg) runs with the region in LDS and (PREFETCH) the next region's loads in flight;
// the fused train kernel trades the 32 prefetch registers for a third resident workgroup per CU
template <int SRC, bool PREFETCH, class Body>
__device__ __forceinline__ void gmp_regions(const SeqArgs& a, const GmpGeom& g, const GmpLds& s, Body&& body) {
    GmpFetch f;
    int reg = blockIdx.x;
    if (PREFETCH && reg < g.nregions) gmp_fetch<SRC>(a, g, reg, f);
    for (; reg < g.nregions; reg += gridDim.x) {
        if constexpr (!PREFETCH) gmp_fetch<SRC>(a, g, reg, f);
        __syncthreads();
        gmp_deposit<SRC>(g, s, f);
        __syncthreads();
        if (PREFETCH && reg + (int)gridDim.x < g.nregions) gmp_fetch<SRC>(a, g, reg + gridDim.x, f);
        body(reg);
    }
}

// ---- forward: a lane owns the kR consecutive samples off .. off + kR - 1 of an item (off multiple of 4) ---------------------
struct GmpTile {
    int e0;            // natural LDS entry of the first sample (multiple of 4)
    size_t g0;         // (b, t) offset of the first sample in the tensors
    int nval, nown;    // samples of the tile inside the frame / inside the item's own chunk (the fused step also walks the
                       // kM-1 samples after a chunk: their dy is needed by the chunk's weight gradient)
};
__device__ __forceinline__ GmpTile gmp_locate_tile(const SeqArgs& a, const GmpGeom& g, int reg, int ti, int ntile, int span) {
    GmpTile L;
    const int it = ti / ntile, off = kR * (ti - it * ntile);
    const int item = reg * g.NI + it;
    int b = item, c = 0;
    if (g.nchunk > 1) { b = item / g.nchunk; c = item - b * g.nchunk; }
    const int t0 = c * g.TC + off;
    const bool ok = it < g.NI && item < g.nitems;
    L.nval = ok ? max(0, min(kR, min(a.T - t0, span - off))) : 0;
    L.nown = ok ? max(0, min(L.nval, g.TC - off)) : 0;
    L.e0 = L.nval > 0 ? it * g.E + off + kHalo : kHalo;
    L.g0 = L.nval > 0 ? (size_t)b * a.T + t0 : 0;
    return L;
}

// Packed fp32 (v_pk_fma_f32: two FMAs per lane and instruction; same flops per cycle as plain FMAs on gfx950, but half the
// instruction stream — measured 129 -> 111 us at 32768 x 200): the accumulators of samples (0,1) and (2,3) form register pairs;
// the envelope window is kept twice, as even- and odd-aligned pairs
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ WPtr gmp_fresh_after2(WPtr w, v2f (&e)[11]) {
    asm volatile("" : "+s"(w), "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]), "+v"(e[8]),
                 "+v"(e[9]), "+v"(e[10]));
    return w;
}
__device__ __forceinline__ void gmp_tile_outputs(const GmpLds& s, WPtr w0, int e0, float (&yr)[kR], float (&yi)[kR]) {
    static_assert(kR == 4, "two sample pairs per lane");
    WPtr w = gmp_fresh(w0);
    v2f ec[2][kM];                             // [pair][r] = (Ec of sample 2 pair, Ec of sample 2 pair + 1)
#pragma unroll
    for (int up = 0; up < 2; ++up)
#pragma unroll
        for (int r = 0; r < kM; ++r) { const float v = w[gmp_w0(r)]; ec[up][r] = v2f{v, v}; }
    // window entry m = natural entry e0 - kHalo + m, m = 0 .. kHalo + kR - 1: plane m mod 4 (e0, kHalo multiples of 4); one power
    // at a time is held in registers (compile-time plane / slot offsets).
    // Sample u needs Ec[t-r, r] = w0[r] + sum_{d,k} w[d,k,r] P_d[t-r-k] -> window entry kHalo + u - (r + k).
    // One segment = the 11 weights w[d,k,0..10] (contiguous in the reference's order); the scalar loads run two segments ahead
    constexpr int kWin = kHalo + kR, kSegs = kNP * kM;
    v2f pa[kWin / 2], pb[kWin / 2 - 1];        // (pw[2j], pw[2j+1]) and (pw[2j+1], pw[2j+2])
    const float* pq = s.P + ((e0 - kHalo) >> 2);
    float wc[kM], wn[kM], wnn[kM];
#pragma unroll
    for (int r = 0; r < kM; ++r) { wc[r] = w[gmp_w(0, kM - 1, r)]; wn[r] = w[gmp_w(0, kM - 2, r)]; }
    gmp_for<kSegs>([&](auto sc) {
        constexpr int seg = decltype(sc)::value, d = seg / kM, k = kM - 1 - seg % kM;
        if constexpr (seg % kM == 0) {
            auto at = [&](int m) { return pq[(4 * d + (m & 3)) * s.Q + (m >> 2)]; };
#pragma unroll
            for (int j = 0; j < kWin / 2; ++j) pa[j] = v2f{at(2 * j), at(2 * j + 1)};
#pragma unroll
            for (int j = 0; j < kWin / 2 - 1; ++j) pb[j] = v2f{at(2 * j + 1), at(2 * j + 2)};
        }
        if constexpr (seg + 2 < kSegs) {
            constexpr int dn = (seg + 2) / kM, kn = kM - 1 - (seg + 2) % kM;
            w = gmp_fresh_after2(w, ec[0]);
            w = gmp_fresh_after2(w, ec[1]);
#pragma unroll
            for (int r = 0; r < kM; ++r) wnn[r] = w[gmp_w(dn, kn, r)];
        }
#pragma unroll
        for (int r = kM - 1; r >= 0; --r)
#pragma unroll
            for (int up = 0; up < 2; ++up) {
                const int m = kHalo + 2 * up - r - k;
                const v2f p = (m & 1) ? pb[m >> 1] : pa[m >> 1];
                ec[up][r] = __builtin_elementwise_fma(v2f{wc[r], wc[r]}, p, ec[up][r]);
            }
#pragma unroll
        for (int r = 0; r < kM; ++r) { wc[r] = wn[r]; wn[r] = wnn[r]; }
    });
    v2f xw[kM - 1 + kR];                       // x[t0 - 10 .. t0 + 3] as (I, Q) pairs
#pragma unroll
    for (int v = 0; v < kM - 1 + kR; ++v) { const float2 t = s.X[e0 - (kM - 1) + v]; xw[v] = v2f{t.x, t.y}; }
#pragma unroll
    for (int u = 0; u < kR; ++u) {
        v2f y = {0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < kM; ++r) {
            const float e = ec[u >> 1][r][u & 1];
            y = __builtin_elementwise_fma(v2f{e, e}, xw[kM - 1 + u - r], y);
        }
        yr[u] = y[0]; yi[u] = y[1];
    }
}

__global__ __launch_bounds__(kThreads, 2) void gmp_fwd_kernel(SeqArgs a, GmpGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GmpLds s(smem, g, false);
    const WPtr w = (WPtr)a.params;
    const int ntile = (g.TC + kR - 1) / kR, tiles = g.NI * ntile;
    float2* y2 = reinterpret_cast<float2*>(a.y);
    gmp_regions<0, true>(a, g, s, [&](int reg) {
        for (int ti = threadIdx.x; ti < tiles; ti += kThreads) {
            const GmpTile L = gmp_locate_tile(a, g, reg, ti, ntile, g.TC);
            float yr[kR], yi[kR];
            gmp_tile_outputs(s, w, L.e0, yr, yi);
#pragma unroll
            for (int u = 0; u < kR; ++u)
                if (u < L.nval) y2[L.g0 + u] = make_float2(yr[u], yi[u]);
        }
    });
}

// dL/dW: every wave owns three 16 x 16 accumulator tiles [r][(d,k) | ones]; one partials row per workgroup.
// FUSED: the whole train step of a region (x, target -> forward -> loss and dy in LDS -> weight gradient): x and target are
// read once, nothing else touches HBM but the partials row (loss partial in column P).
template <bool FUSED>
__global__ __launch_bounds__(kThreads, 3) void gmp_wgrad_kernel(SeqArgs a, GmpGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GmpLds s(smem, g, true);
    const WPtr w = (WPtr)a.params;
    const int span = g.TC + (g.nchunk > 1 ? kM - 1 : 0), ntile = (span + kR - 1) / kR, tiles = g.NI * ntile;
    float loss_acc = 0.0f;
    const float* pf = s.P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    // B operand of column c = 16 ct + n: P_d[s-k] for c = 11 d + k < 44, the constant 1 for c = 44 (dL/dw0[r] = sum_s G[s,r]);
    // columns 45..47 and rows r = 11..15 of the tiles are never read back, so they may hold anything finite.
    // Sample q of a group sits at natural entry eb + 4 grp + q (eb multiple of 4): P_d[s-k] is at plane (q-k) mod 4 of power d,
    // slot eb/4 + grp + floor((q-k)/4) -> per-lane float offset boff, then one float per group
    int boff[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
        const int c = 16 * ct + n, d = c < kNP * kM ? c / kM : 0, k = c < kNP * kM ? c % kM : 0;
        const int rel = q - k;                                        // -10 .. 3
        boff[ct] = (4 * d + ((rel + 12) & 3)) * s.Q + ((rel + 12) >> 2) - 3;
    }
    const float bmul2 = 32 + n < kNP * kM ? 1.0f : 0.0f, badd2 = 32 + n == kNP * kM ? 1.0f : 0.0f;
    f32x4 acc[3];
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    gmp_regions<FUSED ? 2 : 1, !FUSED>(a, g, s, [&](int reg) {
        if constexpr (FUSED) {
            // forward of the region; dy = dLoss/dy replaces the staged target in LDS, the loss of the chunk's own samples
            // accumulates per lane
            for (int ti = threadIdx.x; ti < tiles; ti += kThreads) {
                const GmpTile L = gmp_locate_tile(a, g, reg, ti, ntile, span);
                float yr[kR], yi[kR];
                gmp_tile_outputs(s, w, L.e0, yr, yi);
#pragma unroll
                for (int u = 0; u < kR; ++u) {
                    const float2 tv = s.DY[L.e0 + u];
                    const S16Loss lc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, u < L.nval ? a.inv_count : 0.0f, u < L.nown);
                    float d0, d1;
                    s16_loss(lc, yr[u] - tv.x, yi[u] - tv.y, d0, d1, loss_acc);
                    if (u < L.nval) s.DY[L.e0 + u] = make_float2(d0, d1);
                }
            }
            __syncthreads();
        }
        for (int it = 0; it < g.NI; ++it) {
            const int item = reg * g.NI + it;
            if (item >= g.nitems) break;
            const int c = item % g.nchunk;
            const int len = min(g.TC, a.T - c * g.TC);
            // a group = 4 consecutive samples (K of the MFMA); wave w takes groups w, w + 4, ...; four groups in flight so that
            // their LDS reads (immediate offsets from per-iteration bases) go first; no masks until the ragged tail
            constexpr int GU = 4, kWaves = kThreads / 64;
            const int eb = it * g.E + kHalo, nfull = len >> 2;
            const float2* xp = s.X + eb + q;
            const float2* dp = s.DY + eb + q + n;
            const float* p0 = pf + (eb >> 2) + boff[0];
            const float* p1 = pf + (eb >> 2) + boff[1];
            const float* p2 = pf + (eb >> 2) + boff[2];
            int grp = wave;
            for (; grp + (GU - 1) * kWaves < nfull; grp += GU * kWaves) {
                float gv[GU], bv[GU][3];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int gi = grp + u * kWaves, o = 4 * gi;
                    const float2 xv = xp[o], dv = dp[o];
                    gv[u] = __builtin_fmaf(xv.x, dv.x, xv.y * dv.y);
                    bv[u][0] = p0[gi]; bv[u][1] = p1[gi];
                    bv[u][2] = __builtin_fmaf(p2[gi], bmul2, badd2);
                }
#pragma unroll
                for (int u = 0; u < GU; ++u)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) acc[ct] = mfma4(gv[u], bv[u][ct], acc[ct]);
            }
            for (; 4 * grp < len; grp += kWaves) {       // samples past the chunk's end are staged (halo) but do not count
                const int o = 4 * grp;
                const float okf = __builtin_amdgcn_fmed3f((float)(len - o - q), 0.0f, 1.0f);
                const float2 xv = xp[o], dv = dp[o];
                const float gvt = __builtin_fmaf(xv.x, dv.x, xv.y * dv.y) * okf;
                acc[0] = mfma4(gvt, p0[grp], acc[0]);
                acc[1] = mfma4(gvt, p1[grp], acc[1]);
                acc[2] = mfma4(gvt, __builtin_fmaf(p2[grp], bmul2, badd2), acc[2]);
            }
        }
    });
    // acc[ct][v] of lane (n, q) = D[r = 4q + v][col = 16 ct + n]
    __syncthreads();
    float* red = smem + wave * 16 * kGradCols;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[(4 * q + v) * kGradCols + 16 * ct + n] = acc[ct][v];
    float* lred = smem + (kThreads / 64) * 16 * kGradCols;
    if constexpr (FUSED) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) loss_acc += __shfl_xor(loss_acc, m);
        if (lane == 0) lred[wave] = loss_acc;
    }
    __syncthreads();
    float* prow = a.partials + (size_t)blockIdx.x * (kGmpP + kLossCols);
    for (int i = threadIdx.x; i < kGmpP + kLossCols; i += kThreads) {
        float v = 0.0f;
        if (i < kGmpP) {
            int r, c;
            if (i < kM) { r = kM - 1 - i; c = kNP * kM; }
            else {
                const int f = i - kM, d = f / (kM * kM), ii = (f / kM) % kM, m = f % kM;
                r = kM - 1 - m; c = d * kM + (kM - 1 - ii);
            }
            const float* p = smem + r * kGradCols + c;
            v = (p[0] + p[16 * kGradCols]) + (p[2 * 16 * kGradCols] + p[3 * 16 * kGradCols]);
        } else if (FUSED && i == kGmpP) {
            v = (lred[0] + lred[1]) + (lred[2] + lred[3]);
        }
        prow[i] = v;
    }
}

// dL/dx: one lane per sample
__global__ __launch_bounds__(kThreads, 3) void gmp_dx_kernel(SeqArgs a, GmpGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GmpLds s(smem, g, true);
    const WPtr w0 = (WPtr)a.params;
    const int nsamp = g.NI * g.TC;
    float2* dx2 = reinterpret_cast<float2*>(a.dx);
    gmp_regions<1, true>(a, g, s, [&](int reg) {
        for (int idx = threadIdx.x; idx < nsamp; idx += kThreads) {
            const int it = idx / g.TC, off = idx - it * g.TC;
            const int item = reg * g.NI + it;
            int b = item, c = 0;
            if (g.nchunk > 1) { b = item / g.nchunk; c = item - b * g.nchunk; }
            const int t = c * g.TC + off;
            const bool ok = item < g.nitems && t < a.T;
            const int e = ok ? it * g.E + off + kHalo : kHalo;
            WPtr w = gmp_fresh(w0);
            float ec[kM];
#pragma unroll
            for (int r = 0; r < kM; ++r) ec[r] = w[gmp_w0(r)];
#pragma unroll
            for (int k = 0; k < kM; ++k) {
                const float4 p = s.powers(e - k);
                w = gmp_fresh_after(w, ec);
#pragma unroll
                for (int r = 0; r < kM; ++r) {
                    ec[r] = __builtin_fmaf(w[gmp_w(0, k, r)], p.x, ec[r]);
                    ec[r] = __builtin_fmaf(w[gmp_w(1, k, r)], p.y, ec[r]);
                    ec[r] = __builtin_fmaf(w[gmp_w(2, k, r)], p.z, ec[r]);
                    ec[r] = __builtin_fmaf(w[gmp_w(3, k, r)], p.w, ec[r]);
                }
            }
            float2 dv[kHalo + 1];
#pragma unroll
            for (int j = 0; j <= kHalo; ++j) dv[j] = s.DY[e + j];
            float dr = 0.0f, di = 0.0f;                   // through the complex factor u[t+m] of every term
#pragma unroll
            for (int r = 0; r < kM; ++r) {
                dr = __builtin_fmaf(dv[r].x, ec[r], dr);
                di = __builtin_fmaf(dv[r].y, ec[r], di);
            }
            float dp[kNP] = {0.f, 0.f, 0.f, 0.f};        // dL/dP_d[s]: the terms whose envelope sample is s
#pragma unroll
            for (int k = 0; k < kM; ++k) {
                const float2 xk = s.X[e + k];
                w = gmp_fresh_after4(w, dp);
#pragma unroll
                for (int r = 0; r < kM; ++r) {
                    const float gv = __builtin_fmaf(xk.x, dv[k + r].x, xk.y * dv[k + r].y);
                    dp[0] = __builtin_fmaf(w[gmp_w(0, k, r)], gv, dp[0]);
                    dp[1] = __builtin_fmaf(w[gmp_w(1, k, r)], gv, dp[1]);
                    dp[2] = __builtin_fmaf(w[gmp_w(2, k, r)], gv, dp[2]);
                    dp[3] = __builtin_fmaf(w[gmp_w(3, k, r)], gv, dp[3]);
                }
            }
            const float4 p0 = s.powers(e);
            const float2 xs = s.X[e];
            float damp = dp[0];
            damp = __builtin_fmaf(2.0f * p0.x, dp[1], damp);
            damp = __builtin_fmaf(3.0f * p0.y, dp[2], damp);
            damp = __builtin_fmaf(4.0f * p0.z, dp[3], damp);
            const float ia = p0.x > 0.0f ? fast_rcp(p0.x) : 0.0f;      // d|x|/dx = x/|x|, 0 at the origin (torch.abs)
            damp *= ia;
            if (ok) dx2[(size_t)b * a.T + t] = make_float2(__builtin_fmaf(damp, xs.x, dr), __builtin_fmaf(damp, xs.y, di));
        }
    });
}

// aux_bytes: LDS per entry besides the float4 envelope powers (float2 x, + float2 dy / target)
GmpGeom gmp_geom(int B, int T, int aux_bytes) {
    GmpGeom g;
    g.TC = T <= kGmpMaxChunk ? T : kGmpMaxChunk;
    g.nchunk = (T + g.TC - 1) / g.TC;
    g.nitems = B * g.nchunk;
    g.E = (g.TC + 2 * kHalo + 3) / 4 * 4;
    const int budget = 48 * 1024;                       // three workgroups per CU
    int ni_max = budget / (g.E * (16 + aux_bytes) + 64);
    if (ni_max > kEPT * kThreads / g.E) ni_max = kEPT * kThreads / g.E;
    const int spread = g.nitems / (3 * device_cus());   // keep at least three regions per CU when the batch allows
    if (ni_max > spread) ni_max = spread;
    if (ni_max > g.nitems) ni_max = g.nitems;
    if (ni_max < 1) ni_max = 1;
    // forward lanes map to the NI * ceil(TC / 4) sample tiles of a region in passes of 256: fewest idle lanes wins
    const int ntile = (g.TC + kR - 1) / kR;
    int best = ni_max;
    double best_waste = 1e9;
    for (int ni = ni_max; ni >= (ni_max + 1) / 2; --ni) {
        const long n = (long)ni * ntile, padded = (n + kThreads - 1) / kThreads * kThreads;
        const double waste = (double)padded / (double)n;
        if (waste < best_waste - 1e-9) { best_waste = waste; best = ni; }
    }
    g.NI = best;
    g.nregions = (g.nitems + g.NI - 1) / g.NI;
    const int slots = g.NI * g.E / 4;
    g.Q = slots + ((4 - slots % 16) + 16) % 16;         // Q mod 16 = 4
    return g;
}
int gmp_grid(const GmpGeom& g) {
    const int cap = 3 * device_cus();
    return g.nregions < cap ? g.nregions : cap;
}
size_t gmp_lds(const GmpGeom& g, int aux_bytes) {
    const size_t stage = (size_t)4 * g.Q * 16 + (size_t)g.NI * g.E * aux_bytes;
    const size_t red = (size_t)((kThreads / 64) * 16 * kGradCols + 4) * sizeof(float);
    return stage > red ? stage : red;
}
bool gmp_ok(const odpd_model_t* m) { return m->hidden == kM; }

}  // namespace

int gmp_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!gmp_ok(m)) return ODPD_EUNSUPPORTED;
    const GmpGeom g = gmp_geom(a.B, a.T, 8);
    hipLaunchKernelGGL(gmp_fwd_kernel, dim3(gmp_grid(g)), dim3(kThreads), gmp_lds(g, 8), st, a, g);
    return (int)hipGetLastError();
}
int gmp_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!gmp_ok(m)) return ODPD_EUNSUPPORTED;
    if (a.partials == nullptr && a.dx == nullptr) return ODPD_EINVAL;
    const GmpGeom g = gmp_geom(a.B, a.T, 16);
    if (a.partials != nullptr) {
        hipLaunchKernelGGL(gmp_wgrad_kernel<false>, dim3(gmp_grid(g)), dim3(kThreads), gmp_lds(g, 16), st, a, g);
        if (int e = (int)hipGetLastError()) return e;
    }
    if (a.dx != nullptr) hipLaunchKernelGGL(gmp_dx_kernel, dim3(gmp_grid(g)), dim3(kThreads), gmp_lds(g, 16), st, a, g);
    return (int)hipGetLastError();
}
// fused train step: x, target (optionally framed) -> partials rows [dL/dW | loss partial]
int gmp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!gmp_ok(m)) return ODPD_EUNSUPPORTED;
    const GmpGeom g = gmp_geom(a.B, a.T, 16);
    hipLaunchKernelGGL(gmp_wgrad_kernel<true>, dim3(gmp_grid(g)), dim3(kThreads), gmp_lds(g, 16), st, a, g);
    return (int)hipGetLastError();
}
int gmp_rows(const odpd_model_t* m, int B, int T) {
    if (!gmp_ok(m)) return ODPD_EUNSUPPORTED;
    return gmp_grid(gmp_geom(B, T, 16));
}

}  // namespace odpd
