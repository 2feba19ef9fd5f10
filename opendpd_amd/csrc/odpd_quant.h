// odpd_quant.h — INT_Quantizer (quant/qmodules/quantizers.py:15-79) as device helpers: power-of-two scale, clamp + round, pass mask.  Shared by
// the quantisation-aware cells (odpd_qat.h) and by the kernels of backbones in which the surgery swaps only nn.Linear heads (lstm_family.hip).
// Nothing here is contraction-sensitive (each function is one multiply, a clamp, a round, one multiply).
#pragma once

namespace odpd {
namespace q16 {

__device__ __forceinline__ float pow2_scale(float scale) { return exp2f(rintf(log2f(fabsf(scale)))); }
struct Quant { float s, inv, qn, qp; };
__device__ __forceinline__ Quant make_quant(float scale, int bits) {
    Quant q; q.s = pow2_scale(scale); q.inv = 1.0f / q.s; q.qn = -(float)(1 << (bits - 1)); q.qp = (float)((1 << (bits - 1)) - 1);
    return q;
}
// clamp as one v_med3_f32; the straight-through pass mask "Qn <= x/s <= Qp" is "the clamp left x/s unchanged"
__device__ __forceinline__ float qapply(float x, const Quant& q) {
    const float v = x * q.inv;
    return rintf(__builtin_amdgcn_fmed3f(v, q.qn, q.qp)) * q.s;
}
__device__ __forceinline__ float qgrid(float x, const Quant& q) { return rintf(__builtin_amdgcn_fmed3f(x * q.inv, q.qn, q.qp)); }      // q(x) / s: the grid index
__device__ __forceinline__ bool qpassb(float x, const Quant& q) {
    const float v = x * q.inv;
    return __builtin_amdgcn_fmed3f(v, q.qn, q.qp) == v;
}
__device__ __forceinline__ float qpass(float x, const Quant& q) { return qpassb(x, q) ? 1.0f : 0.0f; }
}  // namespace q16
}  // namespace odpd
