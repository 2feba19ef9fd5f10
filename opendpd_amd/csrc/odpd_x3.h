// odpd_x3.h — "bf16x3": fp32 operands on the bf16 matrix pipe.  Every fp32 value v is carried as three bf16 terms v = v1 + v2 + v3
// (round-to-nearest residuals: the subtractions are exact and the three 8-bit significands cover the 24), a product a.b is the six term
// products of weight 2^-16 and above (a1 b3, a3 b1, a2 b2, a1 b2, a2 b1, a1 b1) accumulated smallest first in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16: what is dropped is below 2^-23 |a b|, the size of one fp32 rounding of the product.  Shared by gru_s16x.hip
// (GRU family, hidden 17 .. 24: frozen-PA step and train step).  (r06 also tried it on the recurrent chain of the K-packed LSTM / VDLSTM kernel — 60 bf16 instructions
// for 56 exact-fp32 ones plus the splits: 1.157 -> 1.143 ms, inside the noise; not kept: profiles/r06/cfg4_removal.md.)
#pragma once
#include "odpd_s16.h"

namespace odpd {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- bf16x3 splits -------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even): a in the low half
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (a, b) -> three packed bf16 pairs; the residual subtractions are exact (a bf16 term has at most 8 of the operand's 24 bits)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = pk_bf16(a, b);
    const float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(p2 << 16), sb = rb - __uint_as_float(p2 & 0xffff0000u);
    p3 = pk_bf16(sa, sb);
}
struct Split3 { u32x4 t[3]; };      // one K = 32 B operand: the lane's eight values as three bf16x8 terms
__device__ __forceinline__ Split3 split8(const float (&v)[8]) {
    Split3 s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned a, b, c;
        split_pair(v[2 * p], v[2 * p + 1], a, b, c);
        s.t[0][p] = a; s.t[1][p] = b; s.t[2][p] = c;
    }
    return s;
}
__device__ __forceinline__ f32x4 mfma32(const u32x4& a, const u32x4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 tabx_ld(TabPtr p, int i) { return __builtin_bit_cast(u32x4, p[i]); }
}  // namespace odpd
