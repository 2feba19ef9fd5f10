// odpd_gru.h — pieces shared by the GRU-family translation units (gru_family.hip, gru_cascade.hip): the LDS rotated-quad weight tables.
#pragma once
#include "odpd_s16.h"

namespace odpd {

// -------------------------------------------------------------------------------------------------
// LDS rotated-quad weight tables
// -------------------------------------------------------------------------------------------------
template <int R, bool DG>
struct GruTabs {
    static constexpr int kHH = 0;            // rows g*R + rb           : W_hg[o][16*blk + src]
    static constexpr int kHHT = 3 * R;       // rows 3R + g*R + rb      : W_hg[16*blk + src][o]
    static constexpr int kHID = 6 * R;       // rows 6R + rb            : fc_hid[o][16*blk + src]
    static constexpr int kHIDT = 7 * R;      // rows 7R + rb            : fc_hid[16*blk + src][o]
    static constexpr int kRows = DG ? 8 * R : 6 * R;
    static constexpr int kFloats = kRows * 4 * 64 * 4;
};

// Cooperative fill (all waves of the block; ends with __syncthreads()).
// HALF (R = 2, hidden 17..24): the second 16-lane block holds its <= 8 units twice (lane c and lane c ^ 8 both carry unit 16 + c % 8), so that
// a rotated dot product over that block needs rotations 0..7 only — whichever 8 consecutive positions a lane sees, they are the 8 units.
template <int R, bool DG, bool WITH_T, bool HALF = false>
__device__ __forceinline__ void fill_gru_tabs(float* tab, const float* pl, const GruLayout& L, int lane, int wave, int nwb) {
    using T = GruTabs<R, DG>;
    static_assert(!HALF || R == 2, "the half-block layout is for two-block models");
    const int H = L.H, col = lane & 15, row = (lane >> 4) & (R - 1), dir = rot_dir(col);
    const int o = 16 * row + ((HALF && row == 1) ? (col & 7) : col);
    float4* t4 = reinterpret_cast<float4*>(tab);
    for (int idx = wave; idx < T::kRows * 4; idx += nwb) {
        const int tr = idx >> 2, q = idx & 3;
        const bool transposed = (tr >= T::kHHT && tr < T::kHID) || tr >= T::kHIDT;
        if (!WITH_T && transposed) continue;
        const bool hid = tr >= T::kHID;
        const int local = hid ? (tr - (transposed ? T::kHIDT : T::kHID)) : (tr - (transposed ? T::kHHT : T::kHH));
        const int g = hid ? 0 : local / R, rb = hid ? local : local % R;
        const int base = hid ? L.o_w_hid : L.o_w_hh + g * H * H;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int mb = (row + rb) % R, mp = (col + dir * (4 * q + e)) & 15;
            const int m = 16 * mb + ((HALF && mb == 1) ? (mp & 7) : mp);
            const bool ok = o < H && m < H;
            v[e] = ok ? pl[base + (transposed ? m * H + o : o * H + m)] : 0.0f;
        }
        t4[idx * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
}

}  // namespace odpd
