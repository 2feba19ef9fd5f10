// tcnn.hip — kernels for the TCNN backbone (backbones/tcnn.py:5-97):
//   feat = [I,Q,a,a^3,sin,cos] -> Conv1d(6->C,k1,bias) -> Hardswish -> 4 x [depthwise Conv1d(C,C,k5,dil d,pad 2d,no bias)
//   -> Hardswish], d = 1,2,4,8 -> Conv1d(C->2,k1,no bias);  y = net + [I,Q]                       (tcnn.py:82-96)
// The network is separable per channel except for the last 1x1 sum, and non-recurrent: time is the parallel axis.
//   forward : one workgroup per (sequence, time tile); thread = time step (tile + 32-step halo each side, receptive
//             field 30); channels are looped INSIDE, two LDS line buffers per stage ping-pong, y accumulates in registers.
//   backward: one workgroup per (channel, slice of the batch); for every sequence of the slice the five stages of that
//             channel are recomputed into LDS, back-propagated, and the channel's 29 parameter gradients accumulate in
//             per-thread registers; one block reduction at the end -> columns of partial row `slice`.
//   dL/dx   : (frozen PA of a cascade) the forward tiling with a 64-step halo: dx[t] needs dL/d pre of the last stage at
//             t +- 30, whose activations need x at +- 30 around them.  Channels are looped inside the workgroup, so the
//             cross-channel sum sum_c gp0_c[t] W0[c][:] stays in six registers per thread: deterministic, no atomics.
// Zero padding semantics: every stage's activation is 0 outside [0,T) (PyTorch pads each conv input).
#include "odpd_seq.h"

namespace odpd {

struct TcnnLayout { int C, o_w0, o_b0, o_dw[4], o_w5, P; };
__host__ __device__ inline TcnnLayout tcnn_layout(int C) {
    TcnnLayout L; L.C = C; int o = 0;
    L.o_w0 = o; o += 6 * C; L.o_b0 = o; o += C;
    for (int l = 0; l < 4; ++l) { L.o_dw[l] = o; o += 5 * C; }
    L.o_w5 = o; o += 2 * C;
    L.P = o;
    return L;
}
constexpr int kTHalo = 32;   // >= receptive-field radius 2*(1+2+4+8) = 30

struct TcnnTile { int nthreads, tile, ntiles; };
// forward: up to 16 waves per workgroup; backward (116 gradient accumulators per thread): up to 8 waves = 256 VGPRs
inline TcnnTile tcnn_tiling(int T, int max_waves = 16) {
    TcnnTile t;
    const int cap = 64 * max_waves - 2 * kTHalo;
    int want = (T < cap ? T : cap) + 2 * kTHalo;
    int nw = (want + 63) / 64; if (nw > max_waves) nw = max_waves;
    t.nthreads = 64 * nw; t.tile = t.nthreads - 2 * kTHalo; t.ntiles = (T + t.tile - 1) / t.tile;
    return t;
}

__device__ __forceinline__ float hsg(float v) { return v < -3.0f ? 0.0f : (v <= 3.0f ? __builtin_fmaf(v, 1.0f / 3.0f, 0.5f) : 1.0f); }
__device__ __forceinline__ float ldz(const float* buf, int i, int n) { return (i >= 0 && i < n) ? buf[i] : 0.0f; }

__device__ __forceinline__ void tcnn_feat(float2 xv, bool in, float (&f)[6]) {
    if (!in) { f[0] = f[1] = f[2] = f[3] = f[4] = f[5] = 0.0f; return; }
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), a = __builtin_amdgcn_sqrtf(a2), ia = fast_rcp(a);
    f[0] = xv.x; f[1] = xv.y; f[2] = a; f[3] = a2 * a; f[4] = xv.y * ia; f[5] = xv.x * ia;
}

// Channels are processed kCG at a time: one workgroup barrier per stage serves kCG channels (the stage buffers
// ping-pong, so the write of stage l never races with the reads of stage l-1), and x / dy / the features are loaded
// and formed once per group instead of once per channel.
constexpr int kCG = 1;

// grid = (ntiles, B); block = nthreads; LDS = 2 * kCG * nthreads floats
__global__ __launch_bounds__(1024) void tcnn_fwd_kernel(SeqArgs a, int tile) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockDim.x, pos = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * tile, t = t0 - kTHalo + pos;
    const bool in = t >= 0 && t < a.T;
    const TcnnLayout L = tcnn_layout(a.H);
    const float* __restrict__ p = a.params;
    const float2 xv = in ? reinterpret_cast<const float2*>(a.x)[(size_t)b * a.T + t] : make_float2(0.f, 0.f);
    float f[6];
    tcnn_feat(xv, in, f);
    float y0 = 0.0f, y1 = 0.0f;
    int flip = 0;
    for (int c0 = 0; c0 < L.C; c0 += kCG) {
        float act[kCG];
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const int c = min(c0 + j, L.C - 1);
            float v = p[L.o_b0 + c];
#pragma unroll
            for (int i = 0; i < 6; ++i) v = __builtin_fmaf(p[L.o_w0 + c * 6 + i], f[i], v);
            act[j] = in ? hardswishf_(v) : 0.0f;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int d = 1 << l;
            float* buf = smem + (flip ^= 1) * kCG * n;
#pragma unroll
            for (int j = 0; j < kCG; ++j) buf[j * n + pos] = act[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kCG; ++j) {
                const int c = min(c0 + j, L.C - 1);
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k) s = __builtin_fmaf(p[L.o_dw[l] + c * 5 + k], ldz(buf + j * n, pos + d * (k - 2), n), s);
                act[j] = in ? hardswishf_(s) : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            if (c0 + j < L.C) {
                y0 = __builtin_fmaf(p[L.o_w5 + c0 + j], act[j], y0);
                y1 = __builtin_fmaf(p[L.o_w5 + L.C + c0 + j], act[j], y1);
            }
        }
    }
    if (in && pos >= kTHalo && pos < kTHalo + tile)
        reinterpret_cast<float2*>(a.y)[(size_t)b * a.T + t] = make_float2(y0 + xv.x, y1 + xv.y);
}

// grid = (ceil(C / kCG), nslices); block = nthreads; LDS = 6 * kCG * nthreads floats (4 stage inputs kept for the
// weight gradients + 2 ping-pong exchange buffers), re-used as reduction scratch at the end
__global__ __launch_bounds__(512) void tcnn_bwd_kernel(SeqArgs a, int tile, int ntiles, int nslices) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockDim.x, pos = threadIdx.x, c0 = blockIdx.x * kCG, slice = blockIdx.y;
    const TcnnLayout L = tcnn_layout(a.H);
    const float* __restrict__ p = a.params;
    float* act = smem;                    // act[(l * kCG + j) * n + pos], l = 0..3: inputs of the depthwise stages
    float* xch = smem + 4 * kCG * n;      // two exchange buffers of kCG * n
    float gacc[kCG][29];
#pragma unroll
    for (int j = 0; j < kCG; ++j)
#pragma unroll
        for (int i = 0; i < 29; ++i) gacc[j][i] = 0.0f;

    const int nwork = a.B * ntiles;
    for (int wk = slice; wk < nwork; wk += nslices) {
        const int b = wk / ntiles, t0 = (wk % ntiles) * tile, t = t0 - kTHalo + pos;
        const bool in = t >= 0 && t < a.T, own = in && pos >= kTHalo && pos < kTHalo + tile;
        const float2 xv = in ? reinterpret_cast<const float2*>(a.x)[(size_t)b * a.T + t] : make_float2(0.f, 0.f);
        const float2 dyv = own ? reinterpret_cast<const float2*>(a.dy)[(size_t)b * a.T + t] : make_float2(0.f, 0.f);
        float f[6];
        tcnn_feat(xv, in, f);
        float pre[kCG][5], cur[kCG];
        __syncthreads();   // the previous work item is done with every buffer
        // forward of the group's channels; act_l stays in LDS, the thread keeps its own pre-activations
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const int c = min(c0 + j, L.C - 1);
            float v = p[L.o_b0 + c];
#pragma unroll
            for (int i = 0; i < 6; ++i) v = __builtin_fmaf(p[L.o_w0 + c * 6 + i], f[i], v);
            pre[j][0] = v;
            cur[j] = in ? hardswishf_(v) : 0.0f;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int d = 1 << l;
#pragma unroll
            for (int j = 0; j < kCG; ++j) act[(l * kCG + j) * n + pos] = cur[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kCG; ++j) {
                const int c = min(c0 + j, L.C - 1);
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    s = __builtin_fmaf(p[L.o_dw[l] + c * 5 + k], ldz(act + (l * kCG + j) * n, pos + d * (k - 2), n), s);
                pre[j][l + 1] = s;
                cur[j] = in ? hardswishf_(s) : 0.0f;
            }
        }
        // backward.  g = dL/d act_l ; gp = dL/d pre_l (masked outside the frame)
        float g[kCG];
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const int c = min(c0 + j, L.C - 1);
            gacc[j][27] = __builtin_fmaf(dyv.x, cur[j], gacc[j][27]);
            gacc[j][28] = __builtin_fmaf(dyv.y, cur[j], gacc[j][28]);
            g[j] = __builtin_fmaf(dyv.x, p[L.o_w5 + c], dyv.y * p[L.o_w5 + L.C + c]);
        }
#pragma unroll
        for (int l = 3; l >= 0; --l) {
            const int d = 1 << l;
            float* ex = xch + (l & 1) * kCG * n;
#pragma unroll
            for (int j = 0; j < kCG; ++j) {
                const float gp = in ? g[j] * hsg(pre[j][l + 1]) : 0.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k)
                    gacc[j][7 + l * 5 + k] = __builtin_fmaf(gp, ldz(act + (l * kCG + j) * n, pos + d * (k - 2), n), gacc[j][7 + l * 5 + k]);
                ex[j * n + pos] = gp;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kCG; ++j) {
                const int c = min(c0 + j, L.C - 1);
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k) s = __builtin_fmaf(p[L.o_dw[l] + c * 5 + k], ldz(ex + j * n, pos - d * (k - 2), n), s);
                g[j] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const float gp0 = in ? g[j] * hsg(pre[j][0]) : 0.0f;
            gacc[j][6] += gp0;
#pragma unroll
            for (int i = 0; i < 6; ++i) gacc[j][i] = __builtin_fmaf(gp0, f[i], gacc[j][i]);
        }
    }
    // block reduction of the 29 accumulators of every channel of the group -> partial row `slice`
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int j = 0; j < kCG; ++j)
#pragma unroll
        for (int i = 0; i < 29; ++i) {
            float s = gacc[j][i];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if (lane == 0) smem[(wave * kCG + j) * 32 + i] = s;
        }
    __syncthreads();
    if (threadIdx.x < kCG * 32) {
        const int j = threadIdx.x >> 5, i = threadIdx.x & 31, c = c0 + j;
        if (i < 29 && c < L.C) {
            float s = 0.0f;
            for (int wv = 0; wv < nw; ++wv) s += smem[(wv * kCG + j) * 32 + i];
            int col;
            if (i < 6) col = L.o_w0 + c * 6 + i;
            else if (i == 6) col = L.o_b0 + c;
            else if (i < 27) col = L.o_dw[(i - 7) / 5] + c * 5 + (i - 7) % 5;
            else col = L.o_w5 + (i - 27) * L.C + c;
            a.partials[(size_t)slice * (L.P + kLossCols) + col] = s;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < kLossCols) a.partials[(size_t)slice * (L.P + kLossCols) + L.P + threadIdx.x] = 0.0f;
}

// grid = (ntiles, B); block = nthreads (tile + 64-step halo each side); LDS = 2 * kCG * nthreads floats
constexpr int kTHaloDx = 64;
__global__ __launch_bounds__(1024) void tcnn_dx_kernel(SeqArgs a, int tile) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = blockDim.x, pos = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * tile, t = t0 - kTHaloDx + pos;
    const bool in = t >= 0 && t < a.T;
    const TcnnLayout L = tcnn_layout(a.H);
    const float* __restrict__ p = a.params;
    const float2 xv = in ? reinterpret_cast<const float2*>(a.x)[(size_t)b * a.T + t] : make_float2(0.f, 0.f);
    const float2 dyv = in ? reinterpret_cast<const float2*>(a.dy)[(size_t)b * a.T + t] : make_float2(0.f, 0.f);
    float f[6], df[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tcnn_feat(xv, in, f);
    int flip = 0;
    for (int c0 = 0; c0 < L.C; c0 += kCG) {
        // forward of the group; the thread keeps its own pre-activations of the five stages
        float pre[kCG][5], cur[kCG];
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const int c = min(c0 + j, L.C - 1);
            float v = p[L.o_b0 + c];
#pragma unroll
            for (int i = 0; i < 6; ++i) v = __builtin_fmaf(p[L.o_w0 + c * 6 + i], f[i], v);
            pre[j][0] = v;
            cur[j] = in ? hardswishf_(v) : 0.0f;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const int d = 1 << l;
            float* buf = smem + (flip ^= 1) * kCG * n;
#pragma unroll
            for (int j = 0; j < kCG; ++j) buf[j * n + pos] = cur[j];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kCG; ++j) {
                const int c = min(c0 + j, L.C - 1);
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k) s = __builtin_fmaf(p[L.o_dw[l] + c * 5 + k], ldz(buf + j * n, pos + d * (k - 2), n), s);
                pre[j][l + 1] = s;
                cur[j] = in ? hardswishf_(s) : 0.0f;
            }
        }
        // backward of the group down to the features
        float g[kCG];
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const int c = min(c0 + j, L.C - 1);
            g[j] = c0 + j < L.C ? __builtin_fmaf(dyv.x, p[L.o_w5 + c], dyv.y * p[L.o_w5 + L.C + c]) : 0.0f;
        }
#pragma unroll
        for (int l = 3; l >= 0; --l) {
            const int d = 1 << l;
            float* buf = smem + (flip ^= 1) * kCG * n;
#pragma unroll
            for (int j = 0; j < kCG; ++j) buf[j * n + pos] = in ? g[j] * hsg(pre[j][l + 1]) : 0.0f;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kCG; ++j) {
                const int c = min(c0 + j, L.C - 1);
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 5; ++k) s = __builtin_fmaf(p[L.o_dw[l] + c * 5 + k], ldz(buf + j * n, pos - d * (k - 2), n), s);
                g[j] = s;
            }
        }
#pragma unroll
        for (int j = 0; j < kCG; ++j) {
            const int c = min(c0 + j, L.C - 1);
            const float gp0 = in ? g[j] * hsg(pre[j][0]) : 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) df[i] = __builtin_fmaf(gp0, p[L.o_w0 + c * 6 + i], df[i]);
        }
    }
    if (in && pos >= kTHaloDx && pos < kTHaloDx + tile) {
        float dI, dQ;
        feat_bwd<FEAT_DGRU6>(xv.x, xv.y, df, dI, dQ);
        reinterpret_cast<float2*>(a.dx)[(size_t)b * a.T + t] = make_float2(dI + dyv.x, dQ + dyv.y);   // + residual path
    }
}
struct TcnnTileDx { int nthreads, tile, ntiles; };
inline TcnnTileDx tcnn_tiling_dx(int T) {
    TcnnTileDx t;
    int want = (T < 896 ? T : 896) + 2 * kTHaloDx;
    int nw = (want + 63) / 64; if (nw > 16) nw = 16;
    t.nthreads = 64 * nw; t.tile = t.nthreads - 2 * kTHaloDx; t.ntiles = (T + t.tile - 1) / t.tile;
    return t;
}

static int tcnn_slices(int B, int ntiles, int C) {
    const int nwork = B * ntiles, ngrp = (C + kCG - 1) / kCG;
    int want = (8 * device_cus() + ngrp - 1) / ngrp;     // ~8 blocks per CU overall
    if (want < 1) want = 1;
    return nwork < want ? nwork : want;
}

int tcnn_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->hidden > 64) return ODPD_EUNSUPPORTED;
    const TcnnTile tl = tcnn_tiling(a.T);
    hipLaunchKernelGGL(tcnn_fwd_kernel, dim3(tl.ntiles, a.B), dim3(tl.nthreads), 2 * kCG * tl.nthreads * sizeof(float), st, a, tl.tile);
    return (int)hipGetLastError();
}
int tcnn_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (m->hidden > 64) return ODPD_EUNSUPPORTED;
    if (a.partials == nullptr && a.dx == nullptr) return ODPD_EINVAL;
    if (a.partials != nullptr) {
        const TcnnTile tl = tcnn_tiling(a.T, 8);
        const int ns = tcnn_slices(a.B, tl.ntiles, m->hidden);
        size_t lds = (size_t)6 * kCG * tl.nthreads * sizeof(float);
        if (lds < (size_t)16 * kCG * 32 * sizeof(float)) lds = (size_t)16 * kCG * 32 * sizeof(float);
        auto kb = tcnn_bwd_kernel;
        if (int e = allow_big_lds(kb, lds)) return e;
        hipLaunchKernelGGL(kb, dim3((m->hidden + kCG - 1) / kCG, ns), dim3(tl.nthreads), lds, st, a, tl.tile, tl.ntiles, ns);
        if (int e = (int)hipGetLastError()) return e;
    }
    if (a.dx != nullptr) {
        const TcnnTileDx td = tcnn_tiling_dx(a.T);
        hipLaunchKernelGGL(tcnn_dx_kernel, dim3(td.ntiles, a.B), dim3(td.nthreads), 2 * kCG * td.nthreads * sizeof(float), st, a, td.tile);
    }
    return (int)hipGetLastError();
}
int tcnn_rows(const odpd_model_t* m, int B, int T) {
    if (m->hidden > 64) return ODPD_EUNSUPPORTED;
    return tcnn_slices(B, tcnn_tiling(T, 8).ntiles, m->hidden);
}

}  // namespace odpd
