// odpd_device.h — gfx950 device-side building blocks shared by the OpenDPD recurrent kernels.
//
// Lane mapping used by every recurrent backbone ("row-rotated persistent RNN"):
//   * a 64-lane wavefront holds SPW = 4/R sequences; one sequence owns R DPP rows of 16 lanes;
//   * lane (row q, col j) of a sequence owns hidden unit o = 16*q + j  (o >= H lanes are padding
//     whose weights are zero, so their state stays exactly 0);
//   * a matvec  acc_o = sum_m W[o][m] * h[m]  is done WITHOUT moving h through memory: the lane
//     keeps its weight row pre-rotated, w_rot[k] = W[o][src_k(j)], and accumulates
//     acc += w_rot[k] * row_ror_k(h) for k = 0..15 — one v_fmac_f32_dpp per term;
//   * for R == 2 the other row's 16 values arrive with one permlane/swizzle swap per step.
// Weight-gradient outer products use the exact-fp32 MFMA v_mfma_f32_16x16x4_f32, whose A/B
// fragment layout (lane l <-> [l&15][l>>4]) coincides with this lane mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef ODPD_DPP_ASM
#define ODPD_DPP_ASM 1   // 1: v_fmac_f32_dpp via inline asm; 0: compiler-managed v_mov_dpp + fma
#endif

namespace odpd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));   // accumulator of the 4-block MFMA (v_mfma_f32_16x16x1_4b_f32)

constexpr int kWave = 64;
constexpr int kCkptStride = 4;   // S: recurrent state is checkpointed every S steps for BPTT
constexpr int kChunk = 32;       // time steps staged in LDS per chunk (multiple of kCkptStride)
constexpr int kEvalChunk = 64;   // the evaluation kernels (one sequence per wave) stage one time step per lane
constexpr int kChunkPad = kChunk + 1;  // float2 row stride in LDS (breaks the 2-way bank conflict)
constexpr int kMaxWavesPerBlock = 8;   // waves per workgroup is a launch-time choice (1,2,4,8)
constexpr int kMaxThreads = kWave * kMaxWavesPerBlock;

// ---------------------------------------------------------------------------------------------
// cross-lane primitives
// ---------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ int dpp_ror_i(int v) {
    if constexpr (K == 0) return v;
    else return __builtin_amdgcn_mov_dpp(v, 0x120 + K, 0xf, 0xf, true);  // row_ror:K
}
template <int K>
__device__ __forceinline__ float dpp_ror(float v) {
    return __builtin_bit_cast(float, dpp_ror_i<K>(__builtin_bit_cast(int, v)));
}

// swap the two 16-lane rows inside each 32-lane half: lane i <-> lane i^16
__device__ __forceinline__ float swap16(float v) {
    // ds_swizzle BitMode: and_mask=0x1f, or_mask=0, xor_mask=0x10
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

// sum over the 16 lanes of a DPP row; every lane of the row receives the total
__device__ __forceinline__ float row_sum16(float v) {
    v += dpp_ror<8>(v);
    v += dpp_ror<4>(v);
    v += dpp_ror<2>(v);
    v += dpp_ror<1>(v);
    return v;
}
__device__ __forceinline__ float xor32(float v) {       // lane i <-> lane i ^ 32
    const int iv = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);      // r[0] = (lo, lo), r[1] = (hi, hi)
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor16(float v) {       // lane i <-> lane i ^ 16
    const int iv = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);      // r[0] = rows (0, 0, 2, 2), r[1] = rows (1, 1, 3, 3)
    return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
}
// Both results of swapping a value's rows with themselves, for steps whose consumers sit on known rows (no select needed):
// dup16(v).even = v of rows (0, 0, 2, 2), .odd = rows (1, 1, 3, 3); dup32(v).lo = rows (0, 1, 0, 1), .hi = rows (2, 3, 2, 3)
struct RowDup { float even, odd; };
struct HalfDup { float lo, hi; };
__device__ __forceinline__ RowDup dup16(float v) {
    const int iv = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    const int a = r[0], b = r[1];
    return RowDup{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
__device__ __forceinline__ HalfDup dup32(float v) {
    const int iv = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
    const int a = r[0], b = r[1];
    return HalfDup{__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)};
}
// sum over the wave's four rows, every lane receives it (the additions of `v += xor16(v); v += xor32(v)`, without the two selects)
__device__ __forceinline__ float sum_rows4(float v) {
    const RowDup p = dup16(v);
    const HalfDup q = dup32(p.even + p.odd);
    return q.lo + q.hi;
}
// Row select without control flow: a ?: on the row index can come out as exec-mask branches inside a step loop, and an inline-asm
// v_cndmask hides its VGPR write from the compiler's MFMA hazard handling (a following v_mfma read the stale operand).  So: per-lane
// all-ones / zero masks, made opaque once at kernel start, and a bitwise blend (one v_bfi_b32).
struct RowMasks { int m[4]; };
__device__ __forceinline__ RowMasks row_masks() {
    RowMasks r;
    const int role = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.m[k] = role == k ? -1 : 0;
    asm volatile("" : "+v"(r.m[0]), "+v"(r.m[1]), "+v"(r.m[2]), "+v"(r.m[3]));
    return r;
}
__device__ __forceinline__ float vsel(int m, float t, float f) {      // m = all ones: t, m = 0: f
    return __builtin_bit_cast(float, (m & __builtin_bit_cast(int, t)) | (~m & __builtin_bit_cast(int, f)));
}
// every row of the wave receives all four rows' values: g[k] = v of row k, same column (three cross-row swaps)
__device__ __forceinline__ void gather_rows(float v, float (&g)[4]) {
    const int iv = __builtin_bit_cast(int, v);
    const auto h = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);        // h[0] = rows (0, 1, 0, 1), h[1] = rows (2, 3, 2, 3)
    const int h0 = h[0], h1 = h[1];
    const auto lo = __builtin_amdgcn_permlane16_swap(h0, h0, false, false);       // (0, 0, 0, 0) | (1, 1, 1, 1)
    const auto hi = __builtin_amdgcn_permlane16_swap(h1, h1, false, false);       // (2, 2, 2, 2) | (3, 3, 3, 3)
    // elements through scalar copies: __builtin_bit_cast applied to a vector-element lvalue reads element 0 whatever the index (clang 20)
    const int g0 = lo[0], g1 = lo[1], g2 = hi[0], g3 = hi[1];
    g[0] = __builtin_bit_cast(float, g0); g[1] = __builtin_bit_cast(float, g1);
    g[2] = __builtin_bit_cast(float, g2); g[3] = __builtin_bit_cast(float, g3);
}
// sum over the R rows of one sequence (R = 1 or 2)
template <int R>
__device__ __forceinline__ float seq_sum(float v) {
    v = row_sum16(v);
    if constexpr (R == 2) v += swap16(v);
    return v;
}
// sum the same (row, col) position over the SPW sequences of the wave (every lane gets the total)
template <int R>
__device__ __forceinline__ float across_seqs(float v) {
    if constexpr (R == 1) v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// ---- rotated dot products ----------------------------------------------------------------------
// The accumulators of these multi-instruction groups are EARLY-CLOBBER operands ("+&v"): without it the register allocator may give an
// input that holds the same value at entry (a weight the compiler proved to be the constant 0 and an accumulator that starts at 0) the
// accumulator's own register, and a later instruction of the group then reads the running sum as its weight.
// hipcc (ROCm 7.2) folds v_mov_dpp into v_add/v_mul but not into v_fmac, so the DPP FMAs are
// written as inline asm.  The compiler does not model hazards inside asm: a DPP read of a VGPR
// written by the previous VALU needs 2 wait states (gfx9 hazard table), hence every asm group
// opens with `s_nop 1` (covers a fresh h as well as any register copy the allocator may insert
// between two groups).  VALU->VALU accumulator dependencies inside a group are interlocked by HW.
#define ODPD_DPPF(ACC, H, W, K) \
    "v_fmac_f32_dpp %" #ACC ", %" #H ", %" #W " row_ror:" #K " row_mask:0xf bank_mask:0xf\n\t"
#define ODPD_F3(K, W0, W1, W2) ODPD_DPPF(0, 3, W0, K) ODPD_DPPF(1, 3, W1, K) ODPD_DPPF(2, 3, W2, K)
#define ODPD_ROT3_GROUP(K1, K2, K3, K4, K5)                                                        \
    asm("s_nop 1\n\t" ODPD_F3(K1, 4, 5, 6) ODPD_F3(K2, 7, 8, 9) ODPD_F3(K3, 10, 11, 12)            \
            ODPD_F3(K4, 13, 14, 15) ODPD_F3(K5, 16, 17, 18)                                        \
        : "+&v"(a0), "+&v"(a1), "+&v"(a2)                                                             \
        : "v"(h), "v"(w0[K1]), "v"(w1[K1]), "v"(w2[K1]), "v"(w0[K2]), "v"(w1[K2]), "v"(w2[K2]),    \
          "v"(w0[K3]), "v"(w1[K3]), "v"(w2[K3]), "v"(w0[K4]), "v"(w1[K4]), "v"(w2[K4]),            \
          "v"(w0[K5]), "v"(w1[K5]), "v"(w2[K5]))

// Three interleaved rotated dot products sharing the same h (gate triples):
//   a_g += sum_k w_g[k] * row_ror_k(h),  g = 0,1,2 — three independent accumulator chains in flight.
__device__ __forceinline__ void rotdot3(float& a0, float& a1, float& a2, const float (&w0)[16],
                                        const float (&w1)[16], const float (&w2)[16], float h) {
    a0 = __builtin_fmaf(w0[0], h, a0);
    a1 = __builtin_fmaf(w1[0], h, a1);
    a2 = __builtin_fmaf(w2[0], h, a2);
#if ODPD_DPP_ASM
    ODPD_ROT3_GROUP(1, 2, 3, 4, 5);
    ODPD_ROT3_GROUP(6, 7, 8, 9, 10);
    ODPD_ROT3_GROUP(11, 12, 13, 14, 15);
#else
#define ODPD_R3(K) { float hk = dpp_ror<K>(h); a0 = __builtin_fmaf(w0[K], hk, a0); \
                     a1 = __builtin_fmaf(w1[K], hk, a1); a2 = __builtin_fmaf(w2[K], hk, a2); }
    ODPD_R3(1) ODPD_R3(2) ODPD_R3(3) ODPD_R3(4) ODPD_R3(5) ODPD_R3(6) ODPD_R3(7) ODPD_R3(8)
    ODPD_R3(9) ODPD_R3(10) ODPD_R3(11) ODPD_R3(12) ODPD_R3(13) ODPD_R3(14) ODPD_R3(15)
#undef ODPD_R3
#endif
}

// One rotated dot product: returns acc + sum_k w[k] * row_ror_k(h)  (two accumulator chains: even/odd k)
__device__ __forceinline__ float rotdot(float acc, const float (&w)[16], float h) {
    float a0 = __builtin_fmaf(w[0], h, acc), a1 = 0.0f;
#if ODPD_DPP_ASM
    asm("s_nop 1\n\t"
        ODPD_DPPF(1, 2, 3, 1) ODPD_DPPF(0, 2, 4, 2) ODPD_DPPF(1, 2, 5, 3) ODPD_DPPF(0, 2, 6, 4)
        ODPD_DPPF(1, 2, 7, 5) ODPD_DPPF(0, 2, 8, 6) ODPD_DPPF(1, 2, 9, 7) ODPD_DPPF(0, 2, 10, 8)
        ODPD_DPPF(1, 2, 11, 9) ODPD_DPPF(0, 2, 12, 10) ODPD_DPPF(1, 2, 13, 11) ODPD_DPPF(0, 2, 14, 12)
        ODPD_DPPF(1, 2, 15, 13) ODPD_DPPF(0, 2, 16, 14) ODPD_DPPF(1, 2, 17, 15)
        : "+&v"(a0), "+&v"(a1)
        : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]),
          "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
#else
#define ODPD_R1(K, A) A = __builtin_fmaf(w[K], dpp_ror<K>(h), A);
    ODPD_R1(1, a1) ODPD_R1(2, a0) ODPD_R1(3, a1) ODPD_R1(4, a0) ODPD_R1(5, a1) ODPD_R1(6, a0) ODPD_R1(7, a1)
    ODPD_R1(8, a0) ODPD_R1(9, a1) ODPD_R1(10, a0) ODPD_R1(11, a1) ODPD_R1(12, a0) ODPD_R1(13, a1)
    ODPD_R1(14, a0) ODPD_R1(15, a1)
#undef ODPD_R1
#endif
    return a0 + a1;
}

// ... over rotations 0..7 only (a block whose 8 units are held twice: odpd_gru.h, fill_gru_tabs<.., HALF>)
__device__ __forceinline__ float rotdot8(float acc, const float (&w)[16], float h) {
    float a0 = __builtin_fmaf(w[0], h, acc), a1 = 0.0f;
#if ODPD_DPP_ASM
    asm("s_nop 1\n\t"
        ODPD_DPPF(1, 2, 3, 1) ODPD_DPPF(0, 2, 4, 2) ODPD_DPPF(1, 2, 5, 3) ODPD_DPPF(0, 2, 6, 4)
        ODPD_DPPF(1, 2, 7, 5) ODPD_DPPF(0, 2, 8, 6) ODPD_DPPF(1, 2, 9, 7)
        : "+&v"(a0), "+&v"(a1)
        : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
#else
#define ODPD_R1(K, A) A = __builtin_fmaf(w[K], dpp_ror<K>(h), A);
    ODPD_R1(1, a1) ODPD_R1(2, a0) ODPD_R1(3, a1) ODPD_R1(4, a0) ODPD_R1(5, a1) ODPD_R1(6, a0) ODPD_R1(7, a1)
#undef ODPD_R1
#endif
    return a0 + a1;
}

// ... as ONE accumulator chain in the order k = 0 .. 15 (the order of rotdot3): a lone wave is issue bound, a dependent v_fmac_dpp issues
// back to back, and the second chain costs a v_mov and a v_add per dot product (r05: -4 % on the evaluation pass)
__device__ __forceinline__ float rotdot1(float acc, const float (&w)[16], float h) {
    float c0 = __builtin_fmaf(w[0], h, acc);
#if ODPD_DPP_ASM
    asm("s_nop 1\n\t"
        ODPD_DPPF(0, 1, 2, 1) ODPD_DPPF(0, 1, 3, 2) ODPD_DPPF(0, 1, 4, 3) ODPD_DPPF(0, 1, 5, 4)
        ODPD_DPPF(0, 1, 6, 5) ODPD_DPPF(0, 1, 7, 6) ODPD_DPPF(0, 1, 8, 7) ODPD_DPPF(0, 1, 9, 8)
        ODPD_DPPF(0, 1, 10, 9) ODPD_DPPF(0, 1, 11, 10) ODPD_DPPF(0, 1, 12, 11) ODPD_DPPF(0, 1, 13, 12)
        ODPD_DPPF(0, 1, 14, 13) ODPD_DPPF(0, 1, 15, 14) ODPD_DPPF(0, 1, 16, 15)
        : "+&v"(c0)
        : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(w[8]),
          "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
#else
#define ODPD_R1(K) c0 = __builtin_fmaf(w[K], dpp_ror<K>(h), c0);
    ODPD_R1(1) ODPD_R1(2) ODPD_R1(3) ODPD_R1(4) ODPD_R1(5) ODPD_R1(6) ODPD_R1(7) ODPD_R1(8)
    ODPD_R1(9) ODPD_R1(10) ODPD_R1(11) ODPD_R1(12) ODPD_R1(13) ODPD_R1(14) ODPD_R1(15)
#undef ODPD_R1
#endif
    return c0;
}
__device__ __forceinline__ float rotdot8_1(float acc, const float (&w)[16], float h) {
    float c0 = __builtin_fmaf(w[0], h, acc);
#if ODPD_DPP_ASM
    asm("s_nop 1\n\t"
        ODPD_DPPF(0, 1, 2, 1) ODPD_DPPF(0, 1, 3, 2) ODPD_DPPF(0, 1, 4, 3) ODPD_DPPF(0, 1, 5, 4)
        ODPD_DPPF(0, 1, 6, 5) ODPD_DPPF(0, 1, 7, 6) ODPD_DPPF(0, 1, 8, 7)
        : "+&v"(c0)
        : "v"(h), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
#else
#define ODPD_R1(K) c0 = __builtin_fmaf(w[K], dpp_ror<K>(h), c0);
    ODPD_R1(1) ODPD_R1(2) ODPD_R1(3) ODPD_R1(4) ODPD_R1(5) ODPD_R1(6) ODPD_R1(7)
#undef ODPD_R1
#endif
    return c0;
}

// Rotated dot product with the 16 weights delivered as four quads (e.g. ds_read_b128 from an LDS
// master copy: only 4-8 weight registers are live at a time).  Two accumulator chains.
#define ODPD_ROTQ_GROUP(K0, K1, K2, K3, Q)                                                          \
    asm("s_nop 1\n\t" ODPD_DPPF(0, 2, 3, K0) ODPD_DPPF(1, 2, 4, K1) ODPD_DPPF(0, 2, 5, K2)          \
            ODPD_DPPF(1, 2, 6, K3)                                                                  \
        : "+&v"(a0), "+&v"(a1) : "v"(h), "v"(Q.x), "v"(Q.y), "v"(Q.z), "v"(Q.w))
template <typename QuadSrc>
__device__ __forceinline__ float rotdot_quads(float acc, QuadSrc quad, float h) {
    float a0, a1;
    {
        const float4 q = quad(0);
        a0 = __builtin_fmaf(q.x, h, acc);
        a1 = 0.0f;
#if ODPD_DPP_ASM
        asm("s_nop 1\n\t" ODPD_DPPF(1, 2, 3, 1) ODPD_DPPF(0, 2, 4, 2) ODPD_DPPF(1, 2, 5, 3)
            : "+&v"(a0), "+&v"(a1) : "v"(h), "v"(q.y), "v"(q.z), "v"(q.w));
#else
        a1 = __builtin_fmaf(q.y, dpp_ror<1>(h), a1);
        a0 = __builtin_fmaf(q.z, dpp_ror<2>(h), a0);
        a1 = __builtin_fmaf(q.w, dpp_ror<3>(h), a1);
#endif
    }
#if ODPD_DPP_ASM
    { const float4 q = quad(1); ODPD_ROTQ_GROUP(4, 5, 6, 7, q); }
    { const float4 q = quad(2); ODPD_ROTQ_GROUP(8, 9, 10, 11, q); }
    { const float4 q = quad(3); ODPD_ROTQ_GROUP(12, 13, 14, 15, q); }
#else
#define ODPD_RQ(Q, K) { const float4 q = quad(Q); a0 = __builtin_fmaf(q.x, dpp_ror<K>(h), a0); \
        a1 = __builtin_fmaf(q.y, dpp_ror<K + 1>(h), a1); a0 = __builtin_fmaf(q.z, dpp_ror<K + 2>(h), a0); \
        a1 = __builtin_fmaf(q.w, dpp_ror<K + 3>(h), a1); }
    ODPD_RQ(1, 4) ODPD_RQ(2, 8) ODPD_RQ(3, 12)
#undef ODPD_RQ
#endif
    return a0 + a1;
}

// Three rotated dot products with three different inputs (transposed mat-vec of a gate triple):
//   a_g += sum_k w_g[k] * row_ror_k(h_g)
#define ODPD_F3X(K, W0, W1, W2) ODPD_DPPF(0, 3, W0, K) ODPD_DPPF(1, 4, W1, K) ODPD_DPPF(2, 5, W2, K)
#define ODPD_ROT3X_GROUP(K1, K2, K3, K4, K5)                                                       \
    asm("s_nop 1\n\t" ODPD_F3X(K1, 6, 7, 8) ODPD_F3X(K2, 9, 10, 11) ODPD_F3X(K3, 12, 13, 14)       \
            ODPD_F3X(K4, 15, 16, 17) ODPD_F3X(K5, 18, 19, 20)                                      \
        : "+&v"(a0), "+&v"(a1), "+&v"(a2)                                                             \
        : "v"(h0), "v"(h1), "v"(h2), "v"(w0[K1]), "v"(w1[K1]), "v"(w2[K1]), "v"(w0[K2]),           \
          "v"(w1[K2]), "v"(w2[K2]), "v"(w0[K3]), "v"(w1[K3]), "v"(w2[K3]), "v"(w0[K4]),            \
          "v"(w1[K4]), "v"(w2[K4]), "v"(w0[K5]), "v"(w1[K5]), "v"(w2[K5]))
__device__ __forceinline__ void rotdot3x(float& a0, float& a1, float& a2, const float (&w0)[16],
                                         const float (&w1)[16], const float (&w2)[16], float h0,
                                         float h1, float h2) {
    a0 = __builtin_fmaf(w0[0], h0, a0);
    a1 = __builtin_fmaf(w1[0], h1, a1);
    a2 = __builtin_fmaf(w2[0], h2, a2);
#if ODPD_DPP_ASM
    ODPD_ROT3X_GROUP(1, 2, 3, 4, 5);
    ODPD_ROT3X_GROUP(6, 7, 8, 9, 10);
    ODPD_ROT3X_GROUP(11, 12, 13, 14, 15);
#else
#define ODPD_R3X(K) { a0 = __builtin_fmaf(w0[K], dpp_ror<K>(h0), a0); a1 = __builtin_fmaf(w1[K], dpp_ror<K>(h1), a1); \
                      a2 = __builtin_fmaf(w2[K], dpp_ror<K>(h2), a2); }
    ODPD_R3X(1) ODPD_R3X(2) ODPD_R3X(3) ODPD_R3X(4) ODPD_R3X(5) ODPD_R3X(6) ODPD_R3X(7) ODPD_R3X(8)
    ODPD_R3X(9) ODPD_R3X(10) ODPD_R3X(11) ODPD_R3X(12) ODPD_R3X(13) ODPD_R3X(14) ODPD_R3X(15)
#undef ODPD_R3X
#endif
}

// source column of the value lane `col` receives under row_ror:K (self-calibrating: measured with
// the same instruction, so the weight pre-rotation never depends on the documented direction)
template <int K = 0>
__device__ __forceinline__ void rot_sources(int (&src)[16], int col) {
    src[K] = dpp_ror_i<K>(col);
    if constexpr (K < 15) rot_sources<K + 1>(src, col);
}

// ---------------------------------------------------------------------------------------------
// activations (fp32; v_exp_f32 / v_rcp_f32 based, ~1e-7 abs error)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float sigmoidf_(float v) {
    float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * v);
    return fast_rcp(1.0f + e);
}
__device__ __forceinline__ float tanhf_(float v) {
    float a = __builtin_fabsf(v);
    // |v| < 0.3: odd Taylor/Padé-free polynomial (rel err < 6e-8); else 1 - 2/(e^{2a}+1)
    float v2 = v * v;
    float p = __builtin_fmaf(v2, 0.021869488536155203f, -0.053968253968253971f);
    p = __builtin_fmaf(v2, p, 0.13333333333333333f);
    p = __builtin_fmaf(v2, p, -0.33333333333333333f);
    p = __builtin_fmaf(v2 * v, p, v);
    float e = __builtin_amdgcn_exp2f(2.8853900817779268f * a);
    float t = 1.0f - 2.0f * fast_rcp(e + 1.0f);
    t = __builtin_copysignf(t, v);
    return a < 0.3f ? p : t;
}
__device__ __forceinline__ float hardswishf_(float v) {
    float r = __builtin_fminf(__builtin_fmaxf(v + 3.0f, 0.0f), 6.0f);
    return v * r * (1.0f / 6.0f);
}

// ---------------------------------------------------------------------------------------------
// I/Q feature extraction (reference: dgru.py:61-68, deltagru.py:61-73, tcnn.py:84-91,
// qgru.py:61-66, qgru_amp1.py:63-70) and its backward
// ---------------------------------------------------------------------------------------------
enum FeatMode { FEAT_RAW2 = 0, FEAT_DGRU6 = 1, FEAT_Q4 = 2, FEAT_A4 = 3 };
template <int FM> struct FeatDim { static constexpr int F = (FM == FEAT_RAW2) ? 2 : (FM == FEAT_DGRU6 ? 6 : 4); };

template <int FM>
__device__ __forceinline__ void feat_fwd(float I, float Q, float (&f)[FeatDim<FM>::F]) {
    f[0] = I; f[1] = Q;
    if constexpr (FM == FEAT_DGRU6) {
        float a2 = __builtin_fmaf(I, I, Q * Q);
        float a = __builtin_amdgcn_sqrtf(a2);
        float ia = fast_rcp(a);
        f[2] = a; f[3] = a2 * a; f[4] = Q * ia; f[5] = I * ia;
    } else if constexpr (FM == FEAT_Q4) {
        float a2 = __builtin_fmaf(I, I, Q * Q);
        f[2] = a2; f[3] = a2 * a2;
    } else if constexpr (FM == FEAT_A4) {
        float a2 = __builtin_fmaf(I, I, Q * Q);
        float a = __builtin_amdgcn_sqrtf(a2);
        f[2] = a; f[3] = a2 * a;
    }
}
template <int FM>
__device__ __forceinline__ void feat_bwd(float I, float Q, const float (&df)[FeatDim<FM>::F], float& dI, float& dQ) {
    float gi = df[0], gq = df[1];
    if constexpr (FM == FEAT_DGRU6) {
        float a2 = __builtin_fmaf(I, I, Q * Q);
        float a = __builtin_amdgcn_sqrtf(a2);
        float ia = fast_rcp(a), ia2 = fast_rcp(a2);
        float da = df[2] + 3.0f * a2 * df[3] - (Q * ia2) * df[4] - (I * ia2) * df[5];
        gi += df[5] * ia + da * I * ia;
        gq += df[4] * ia + da * Q * ia;
    } else if constexpr (FM == FEAT_Q4) {
        float a2 = __builtin_fmaf(I, I, Q * Q);
        float da2 = df[2] + 2.0f * a2 * df[3];
        gi += 2.0f * I * da2; gq += 2.0f * Q * da2;
    } else if constexpr (FM == FEAT_A4) {
        float a2 = __builtin_fmaf(I, I, Q * Q);
        float a = __builtin_amdgcn_sqrtf(a2);
        float ia = fast_rcp(a);
        float da = df[2] + 3.0f * a2 * df[3];
        gi += da * I * ia; gq += da * Q * ia;
    }
    dI = gi; dQ = gq;
}
// gradient of the polar input features of one sample: a = |x|, cos = I/a, sin = Q/a (vdlstm.py:66-76,
// pgjanet.py:38-44 via atan2) -> (dL/dI, dL/dQ) given dL/d(a, cos, sin)
__device__ __forceinline__ float2 polar_sample_bwd(float a_, float cw, float sw, float ga, float gc, float gs) {
    const float ia = fast_rcp(a_);
    const float cross = __builtin_fmaf(gc, sw, -gs * cw);      // gc*sin - gs*cos
    return make_float2(__builtin_fmaf(ga, cw, cross * (sw * ia)), __builtin_fmaf(ga, sw, -cross * (cw * ia)));
}
// per-lane selector: lane col j gets f[j] for j < F, `one` for j == F, 0 otherwise
template <int F>
__device__ __forceinline__ float feat_select(const float (&f)[F], int col, float one) {
    float v = (col == F) ? one : 0.0f;
#pragma unroll
    for (int i = 0; i < F; ++i) v = (col == i) ? f[i] : v;
    return v;
}

// wave-level LDS hand-off: lanes of ONE wave exchange data through LDS without a workgroup barrier
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// exact-fp32 MFMA rank-4 update: D[i][j] += sum_k A[i][k] B[k][j], lane l feeds A[l&15][l>>4], B[l>>4][l&15]
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
#ifdef ODPD_X_NOMFMA      // removal experiment (timing only): four vector FMAs keep every accumulator element live and data dependent (with fewer the
    c[0] = __builtin_fmaf(a, b, c[0]); c[1] = __builtin_fmaf(a, b, c[1]);      // compiler deletes whole gate chains); the matrix instruction itself is gone:
    c[2] = __builtin_fmaf(a, b, c[2]); c[3] = __builtin_fmaf(a, b, c[3]);      // ~16 issue cycles instead of ~32
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

}  // namespace odpd
