// odpd_seq.h — device/host plumbing shared by the persistent sequence kernels (GRU, LSTM, delta, JANET
// families): lane identification, LDS staging of (B,T,2) streams, rotated-quad weight-table access,
// launch-shape helpers.
#pragma once
#include "odpd_host.h"

namespace odpd {

// direction of row_ror measured with the instruction itself: lane col receives lane (col + dir*k) & 15
__device__ __forceinline__ int rot_dir(int col) { return dpp_ror_i<1>(col) == ((col + 15) & 15) ? -1 : 1; }

// Returns the same pointer through an empty asm: the compiler can no longer prove that the table
// loads of successive blocks read the same addresses, so it cannot hoist them out of the block loop
// (which would pin W_hh and W_hh^T in registers at the same time again).
// Weight-table pointers carry the LDS address space: the loads are ds_read_b128, not flat_load_dwordx4 (a generic
// pointer into LDS is resolved at run time by the flat unit: longer latency, and it ties up both vmcnt and lgkmcnt).
// (HIP's float4 is a class and cannot be copied out of a non-generic address space: the tables are read as the
// builtin 4-float vector and converted.)
typedef const __attribute__((address_space(3))) f32x4* TabPtr;
__device__ __forceinline__ TabPtr to_tab(const float4* p) { return (TabPtr)reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float4 tab_ld(TabPtr p, int i) {
    const f32x4 v = p[i];
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ TabPtr opaque(TabPtr p) {
    asm volatile("" : "+v"(p));
    return p;
}

// pull one table row (16 rotated weights) / a gate triple into registers
__device__ __forceinline__ void load_rot(float (&w)[16], TabPtr trow) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = tab_ld(trow, q * 64);
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
    }
}
template <int R>
__device__ __forceinline__ void load_rot3(float (&w)[3][R][16], TabPtr tlane, int first_row) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int rb = 0; rb < R; ++rb) load_rot(w[g][rb], tlane + (first_row + g * R + rb) * 4 * 64);
}

// acc + (table rows first_row.. of this lane) . h  — fc_hid pre-activation or its transpose product
template <int R>
__device__ __forceinline__ float tab_rotdot(float acc, TabPtr tlane, int first_row, float h) {
    TabPtr t0 = tlane + first_row * 4 * 64;
    float v = rotdot_quads(acc, [t0](int q) { return tab_ld(t0, q * 64); }, h);
    if constexpr (R == 2) {
        TabPtr t1 = t0 + 4 * 64;
        v = rotdot_quads(v, [t1](int q) { return tab_ld(t1, q * 64); }, swap16(h));
    }
    return v;
}

// -------------------------------------------------------------------------------------------------
// LDS staging of (B,T,2) streams: one chunk = kChunk steps of the wave's SPW sequences,
// LDS layout [seq][kChunkPad] float2
// -------------------------------------------------------------------------------------------------
// one (I, Q) sample of a stream: fp32 pair, or (bf16 storage, BASELINE configs[1]) one 32-bit word = bf16 I | bf16 Q << 16, widened
// exactly — the arithmetic stays fp32 on the values the stream holds
__device__ __forceinline__ float2 ld_iq(const float* g, size_t i, bool bf16) {
    if (bf16) {
        const unsigned w = reinterpret_cast<const unsigned*>(g)[i];
        return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
    }
    return reinterpret_cast<const float2*>(g)[i];
}
// `fidx` (nullable): frames are windows of a resident stream, row b starts at sample fidx[b] * fstride; `bf16`: its sample format
// BATCHED: the loads of one call in two passes (see below) — 2 N more live registers at the call site, so the register-capped two-wave kernels
// keep the element-at-a-time form, whose exposed latency their partner wave covers
template <int SPW, bool BATCHED = false>
__device__ __forceinline__ void stage_in(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane,
                                         float2 fill, const long long* fidx = nullptr, int fstride = 0, bool bf16 = false) {
    constexpr int N = SPW * kChunk / 64;   // float2 per lane
    static_assert(N >= 1 && (SPW * kChunk) % 64 == 0, "chunk must tile the wave");
#ifndef ODPD_NO_BF16_FRAMES     // (timing experiments: the library without the bf16 branch, tools/exp_time.py)
    if (bf16) {      // (its own wave-uniform loop: the fp32 path keeps the code — and the register allocation — it had)
        const unsigned* g1 = reinterpret_cast<const unsigned*>(g);
#pragma unroll 1
        for (int j = 0; j < N; ++j) {
            const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
            float2 v = fill;
            if (tt < len && b0 + m < B) {
                const unsigned w = g1[(size_t)fidx[b0 + m] * fstride + t0 + tt];
                v = make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
            }
            lds[m * kChunkPad + tt] = v;
        }
        return;
    }
#endif
    const float2* g2 = reinterpret_cast<const float2*>(g);
    if constexpr (!BATCHED) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
            float2 v = fill;
            if (tt < len && b0 + m < B) {
                const size_t row = fidx ? (size_t)fidx[b0 + m] * fstride : (size_t)(b0 + m) * T;
                v = g2[row + t0 + tt];
            }
            lds[m * kChunkPad + tt] = v;
        }
        return;
    }
    // Two passes (r06): all frame-start indices first, then all samples, then the LDS stores.  Written as one loop, the compiler emits per
    // element "load the index, s_waitcnt vmcnt(0), load the sample, s_waitcnt vmcnt(0)": 2 N dependent round trips per staged stream where
    // two suffice — exposed in full on a wave that is alone on its SIMD (gru16x_train_kernel), half hidden with a partner wave.
    size_t row[N];
    if (fidx) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int m = (lane + 64 * j) / kChunk;
            row[j] = (size_t)fidx[min(b0 + m, B - 1)] * fstride;      // (clamped: an unconditional load; rows beyond the batch are not used)
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = (size_t)(b0 + (lane + 64 * j) / kChunk) * T;
    }
    float2 v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
        v[j] = fill;
        if (tt < len && b0 + m < B) v[j] = g2[row[j] + t0 + tt];
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j;
        lds[(e / kChunk) * kChunkPad + e % kChunk] = v[j];
    }
}
template <int SPW>
__device__ __forceinline__ void stage_out(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
    constexpr int N = SPW * kChunk / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / kChunk, tt = e % kChunk;
        if (tt < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + tt] = lds[m * kChunkPad + tt];
    }
}

// the same two for a kernel-local chunk length CH (LDS layout [16 sequences][CH + 1] float2): forward kernels whose registers allow four waves per
// SIMD stage 16-step chunks, so that two eight-wave workgroups fit a CU's LDS (r06; tools/occupancy_audit.py)
template <int CH>
__device__ __forceinline__ void stage_in_ch(float2* lds, const float* g, int b0, int B, int T, int t0, int len, int lane, float2 fill) {
    const float2* g2 = reinterpret_cast<const float2*>(g);
    constexpr int N = 16 * CH / 64;
    static_assert(N >= 1 && (16 * CH) % 64 == 0, "chunk must tile the wave");
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH;
        float2 v = fill;      // (not a `?:` on the load: the compiler then selects between the ADDRESS and a stack slot holding `fill`, a flat load)
        if (tt < len && b0 + m < B) v = g2[(size_t)(b0 + m) * T + t0 + tt];
        lds[m * (CH + 1) + tt] = v;
    }
}
template <int CH>
__device__ __forceinline__ void stage_out_ch(const float2* lds, float* g, int b0, int B, int T, int t0, int len, int lane) {
    float2* g2 = reinterpret_cast<float2*>(g);
    constexpr int N = 16 * CH / 64;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int e = lane + 64 * j, m = e / CH, tt = e % CH;
        if (tt < len && b0 + m < B) g2[(size_t)(b0 + m) * T + t0 + tt] = lds[m * (CH + 1) + tt];
    }
}

__device__ __forceinline__ void stage_params(float* pl, const float* params, int P) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) pl[i] = params[i];
    __syncthreads();
}

// common kernel prologue: identifies the lane
struct LaneId { int lane, wave, nwb, col, row, s; };
template <int R>
__device__ __forceinline__ LaneId lane_id() {
    LaneId id;
    id.lane = threadIdx.x & 63; id.wave = threadIdx.x >> 6; id.nwb = blockDim.x >> 6;
    id.col = id.lane & 15; id.row = (id.lane >> 4) & (R - 1); id.s = id.lane / (16 * R);
    return id;
}
__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }


// ---- host side ---------------------------------------------------------------------------------
constexpr size_t kMaxLds = 160 * 1024;
// $ODPD_AUDIT_LDS=1: every launch that goes through allow_big_lds reports its dynamic LDS size on stderr (rocprofv3's kernel trace records only the
// static part) — tools/occupancy_audit.py joins the lines with the trace to find launches whose LDS allocation, not their registers, caps the
// waves per SIMD
void audit_lds(const void* kernel, size_t lds);
template <typename K>
static inline int allow_big_lds(K kernel, size_t lds) {
    audit_lds(reinterpret_cast<const void*>(kernel), lds);
    if (lds <= 64 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kMaxLds);
}
static inline size_t reduce_scratch_bytes(int P, int waves) { return (size_t)waves * (P + kLossCols) * sizeof(float); }

}  // namespace odpd
