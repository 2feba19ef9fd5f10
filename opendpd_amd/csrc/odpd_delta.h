// odpd_delta.h — pieces shared by the delta-backbone translation units (delta_family.hip, gru_cascade.hip): LDS weight tables, features.
#pragma once
#include "odpd_s16.h"

namespace odpd {

__device__ __forceinline__ float wave_sum_(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
constexpr int kDHalo = 16;                                  // TCN taps at t-16, t, t+16
constexpr int kDStride = kChunk + 2 * kDHalo + 1;           // float2 per sequence row
constexpr int kDTabFloats = 6 * 4 * 64 * 4;                 // W_hh (3 rows) + W_hh^T (3 rows)
constexpr int kDState = 7;                                  // checkpoint: h, h_p, dm_r, dm_z, dm_n, dm_nh, x_p[col]

template <bool WITH_T>
__device__ __forceinline__ void fill_delta_tabs(float* tab, const float* pl, const DeltaLayout& L, int lane, int wave, int nwb) {
    const int H = L.H, col = lane & 15, o = col, dir = rot_dir(col);
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int idx = wave; idx < 6 * 4; idx += nwb) {
        const int tr = idx >> 2, q = idx & 3;
        const bool transposed = tr >= 3;
        if (!WITH_T && transposed) continue;
        const int g = transposed ? tr - 3 : tr;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = (col + dir * (4 * q + e)) & 15;
            const bool ok = o < H && m < H && g < L.G;
            v[e] = ok ? pl[L.o_w_hh + g * H * H + (transposed ? m * H + o : o * H + m)] : 0.0f;
        }
        t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
}
__device__ __forceinline__ float hswish_grad_(float v) { return v < -3.0f ? 0.0f : (v <= 3.0f ? __builtin_fmaf(v, 1.0f / 3.0f, 0.5f) : 1.0f); }

template <bool TRES>
__device__ __forceinline__ void delta_feat(float2 xv, float2 xn, float (&f)[6]) {
    const float a2 = __builtin_fmaf(xv.x, xv.x, xv.y * xv.y), a = __builtin_amdgcn_sqrtf(a2);
    f[0] = xv.x; f[1] = xv.y; f[2] = a; f[3] = a2 * a;
    if constexpr (TRES) { f[4] = xn.x; f[5] = xn.y; }
    else { const float ia = fast_rcp(a); f[4] = xv.y * ia; f[5] = xv.x * ia; }
}

}  // namespace odpd
