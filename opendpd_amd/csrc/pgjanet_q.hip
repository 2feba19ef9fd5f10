// pgjanet_q.hip — `--quant` on pgjanet (reference quant/quant_envs.py:145-148, 285-306 on backbones/pgjanet.py:5-84): the cell's six nn.Linear —
// W_a, W_p1, W_p2 on [h, |x| / cos / sin], W_f, W_g on [h, u], W_o on h — become INT_Linear (quant/qmodules/quant_layers.py:48-85): each
// quantises ITS input on an activation grid of its own and its weight on a weight grid (three scale parameters behind each layer's bias); the
// tanh / sigmoid calls are functional and stay float; no module is named fc_out, so the 16-bit output quantiser never runs.
//
// A plain kernel pair next to the float ones (janet_family.hip, janet_s16.hip, janet_wide.hip): ONE sequence per single-wave workgroup, LANE =
// HIDDEN UNIT (hidden <= 32).  The parameters are staged in LDS with the six weight matrices quantised IN PLACE; per step the lane of unit k
// writes q_l(h_k) for the five gate layers (and q_l(u_k) for W_f, W_g) to LDS, the lane of unit j runs its rows against those broadcasts (row
// reads: lanes stride H + 1 / 2 H floats apart; transposed reads in the backward pass: consecutive lanes, consecutive addresses).  The
// forward pass records (a_n, p1, p2, u, f, g, h') per step in HBM (`ckpt`: B x T x 8 x 64 floats) when a backward pass follows.  Backward:
// reverse steps, the gate gradients broadcast through LDS, the transposed mat-vecs with every layer's activation pass mask, the weight
// gradients accumulated in the registers of the lane that owns the rows (deposited in a second LDS copy of the parameter layout at the end), the weight
// quantisers' pass masks applied at write-out from the unquantised weights, the 18 scale columns exact zeros.
#include "odpd_seq.h"
#include "odpd_quant.h"

#pragma clang fp contract(off)

namespace odpd {
namespace {
constexpr int kPC = 64, kPS = 33, kPNS = 8;
struct PgqLayout { int H, ow[6], ob[6], oq[6], P; };      // 0 W_a  1 W_p1  2 W_p2  3 W_f  4 W_g  5 W_o
__host__ __device__ inline int pgq_nin(int l, int H) { return l < 3 ? H + 1 : (l < 5 ? 2 * H : H); }
__host__ __device__ inline PgqLayout pgq_layout(int H) {
    PgqLayout L; L.H = H; int o = 0;
    for (int l = 0; l < 6; ++l) {
        const int nout = l < 5 ? H : 2;
        L.ow[l] = o; o += nout * pgq_nin(l, H); L.ob[l] = o; o += nout; L.oq[l] = o; o += 3;
    }
    L.P = o;
    return L;
}
__host__ __device__ inline int pgq_fwd_floats(int P) { return pad4(P) + kPC * 4 + 8 * 32 + kPC * kPS; }
__host__ __device__ inline int pgq_bwd_floats(int P) { return 2 * pad4(P) + kPC * 4 + kPC * 2 + kPC * 2 + 16 * 32 + 4; }

// |x|, cos(theta), sin(theta) of the chunk's steps, lane = time step (theta = atan2(Q, I): cos = I / |x|, sin = Q / |x|)
__device__ __forceinline__ void pgq_stage_inputs(float* ftab, const float2* xg, int t0, int T, int lane) {
    const float2 xv = t0 + lane < T ? xg[t0 + lane] : make_float2(0.6f, 0.8f);
    const float am = sqrtf(xv.x * xv.x + xv.y * xv.y);
    reinterpret_cast<float4*>(ftab)[lane] = make_float4(am, xv.x / am, xv.y / am, 0.0f);
}
struct PgqQ { q16::Quant a[6]; };
// stage the parameters, form the activation quantisers, quantise the six weight matrices in the staged copy
__device__ __forceinline__ void pgq_setup(float* pl, const SeqArgs& a, const PgqLayout& L, PgqQ& Q, int lane) {
    stage_params(pl, a.params, L.P);
    wave_lds_fence();
#pragma unroll
    for (int l = 0; l < 6; ++l) Q.a[l] = q16::make_quant(pl[L.oq[l] + 1], a.bits_a);
    q16::Quant qw[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) qw[l] = q16::make_quant(pl[L.oq[l]], a.bits_w);
    wave_lds_fence();
#pragma unroll
    for (int l = 0; l < 6; ++l) {
        const int n = (l < 5 ? L.H : 2) * pgq_nin(l, L.H);
        for (int i = lane; i < n; i += 64) pl[L.ow[l] + i] = q16::qapply(pl[L.ow[l] + i], qw[l]);
    }
    wave_lds_fence();
}

// Forward.  HP = padded unit count (16: hidden <= 16, 32: hidden 17 .. 32).  The wave's two halves split a unit's five gate rows, all of them in
// REGISTERS: lane j < 32 holds the a_n / p1 / p2 rows of unit j, lane 32 + j its f / g rows (h part and u part) and the state h_j.  A step is two
// rounds of row-times-broadcast-vector dot products — the vectors are read as ds_read_b128 broadcasts, each half from its own layers' slots —
//   round A: lower half W_a q_0(h), W_p1 q_1(h), W_p2 q_2(h) (+ the scalar input's column) -> a_n, p1, p2 -> u -> q_3(u), q_4(u) to LDS;
//            upper half, in the same instructions, the h parts W_f[:, :H] q_3(h), W_g[:, :H] q_4(h)
//   round B: upper half continues its two sums with the u parts -> f, g -> h'
// with every sum in the order of the reference's mat-vec (k ascending, h part before u part; the padded columns hold zeros and add nothing), so
// the results are those of the first version of this kernel (r04: both operands of every FMA read from LDS, 1.4 .. 2.6 ms per 256 x 200 step) bit for bit.
template <int HP, bool SAVE>
__global__ __launch_bounds__(64) void pgq_fwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const PgqLayout L = pgq_layout(a.H);
    const int H = L.H, T = a.T, H1 = H + 1, H2 = 2 * H;
    float* pl = smem;
    float* ftab = smem + pad4(L.P);            // [64][4]: |x|, cos, sin of the chunk's steps
    float* vq = ftab + kPC * 4;                // [8][32]: q_l(h) for l = 0 .. 4, q_3(u), q_4(u)
    float* hist = vq + 8 * 32;                 // [64][33]: h of the chunk's steps
    PgqQ Q;
    pgq_setup(pl, a, L, Q, lane);
    const bool up = lane >= 32;
    const int j0 = lane & 31;
    const bool vo = j0 < H;
    const int j = vo ? j0 : 0;
    float w[3][HP], wu[2][HP], wsc[3], bias[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int lay = up ? 3 + r : r;                       // (upper half, r = 2: no third row — zeros)
        const bool row = vo && (!up || r < 2);
#pragma unroll
        for (int k = 0; k < HP; ++k) w[r][k] = (row && k < H) ? pl[L.ow[row ? lay : 0] + j * (up ? H2 : H1) + k] : 0.0f;
        wsc[r] = (vo && !up) ? pl[L.ow[r] + j * H1 + H] : 0.0f;
        bias[r] = row ? pl[L.ob[row ? lay : 0] + j] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < HP; ++k) wu[r][k] = (vo && up && k < H) ? pl[L.ow[3 + r] + j * H2 + H + k] : 0.0f;
    for (int i = lane; i < 8 * 32; i += 64) vq[i] = 0.0f;
    const float* va = vq + (up ? 3 * 32 : 0);                  // round A: row r against slot (up ? 3 : 0) + r
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        float* sv = SAVE ? a.ckpt + (size_t)b * T * kPNS * 64 : nullptr;
        float h = 0.0f;                                        // upper half: the state of unit j0
        for (int t0 = 0; t0 < T; t0 += kPC) {
            const int len = min(kPC, T - t0);
            wave_lds_fence();
            pgq_stage_inputs(ftab, xg, t0, T, lane);
            wave_lds_fence();
            for (int tt = 0; tt < len; ++tt) {
                if (up) {
#pragma unroll
                    for (int l = 0; l < 5; ++l) vq[l * 32 + j0] = vo ? q16::qapply(h, Q.a[l]) : 0.0f;
                }
                wave_lds_fence();
                const float4 in = reinterpret_cast<const float4*>(ftab)[tt];
                const float sc[3] = {in.x, in.y, in.z};
                float acc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int q4 = 0; q4 < HP / 4; ++q4)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float4 v = *reinterpret_cast<const float4*>(va + r * 32 + 4 * q4);
                        acc[r] = __builtin_fmaf(w[r][4 * q4], v.x, acc[r]); acc[r] = __builtin_fmaf(w[r][4 * q4 + 1], v.y, acc[r]);
                        acc[r] = __builtin_fmaf(w[r][4 * q4 + 2], v.z, acc[r]); acc[r] = __builtin_fmaf(w[r][4 * q4 + 3], v.w, acc[r]);
                    }
                // lower half: the three input gates and u
                float gate[3];
#pragma unroll
                for (int l = 0; l < 3; ++l) gate[l] = tanhf_(__builtin_fmaf(wsc[l], q16::qapply(sc[l], Q.a[l]), acc[l]) + bias[l]);
                const float an = gate[0], p1 = gate[1], p2 = gate[2];
                const float u = (an * p1 * p2) * ((1.0f - an) * (1.0f - p1) * (1.0f - p2));
                if (!up) {
                    vq[5 * 32 + j0] = vo ? q16::qapply(u, Q.a[3]) : 0.0f;
                    vq[6 * 32 + j0] = vo ? q16::qapply(u, Q.a[4]) : 0.0f;
                }
                wave_lds_fence();
                // upper half: the u parts behind the h parts
#pragma unroll
                for (int q4 = 0; q4 < HP / 4; ++q4)
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const float4 v = *reinterpret_cast<const float4*>(vq + (5 + r) * 32 + 4 * q4);
                        acc[r] = __builtin_fmaf(wu[r][4 * q4], v.x, acc[r]); acc[r] = __builtin_fmaf(wu[r][4 * q4 + 1], v.y, acc[r]);
                        acc[r] = __builtin_fmaf(wu[r][4 * q4 + 2], v.z, acc[r]); acc[r] = __builtin_fmaf(wu[r][4 * q4 + 3], v.w, acc[r]);
                    }
                const float f = sigmoidf_(acc[0] + bias[0]), g = tanhf_(acc[1] + bias[1]);
                const float hn = (vo && up) ? f * h + (1.0f - f) * g : 0.0f;
                if constexpr (SAVE) {      // record of step t: [a_n | p1 | p2 | u | f | g | h'][unit]
                    float* s = sv + (size_t)(t0 + tt) * kPNS * 64 + j0;
                    if (!up) { s[0] = an; s[64] = p1; s[128] = p2; s[192] = u; }
                    else { s[256] = f; s[320] = g; s[384] = hn; }
                }
                h = hn;
                if (up) hist[tt * kPS + j0] = h;
                wave_lds_fence();
            }
            if (lane < len) {      // the chunk's outputs, lane = time step: W_o on q_5(h)
                const float* hr = hist + lane * kPS;
                float y0 = 0.0f, y1 = 0.0f;
#pragma unroll 8
                for (int k = 0; k < H; ++k) {
                    const float hv = q16::qapply(hr[k], Q.a[5]);
                    y0 = __builtin_fmaf(pl[L.ow[5] + k], hv, y0); y1 = __builtin_fmaf(pl[L.ow[5] + H + k], hv, y1);
                }
                yg[t0 + lane] = make_float2(y0 + pl[L.ob[5]], y1 + pl[L.ob[5] + 1]);
            }
        }
        wave_lds_fence();
    }
}

// Backward, the same split.  Lane j < 32: column j of W_a / W_p1 / W_p2 (h part) and the gradient rows of unit j of those layers; lane 32 + j:
// column j of the h and u parts of W_f / W_g, their gradient rows of unit j, W_o's column, and dL/dh_j — 2 x 4 x HP registers per lane.  Per step:
// (1) upper half: read-out, d_f, d_g -> LDS; W_f^T d_f, W_g^T d_g (h and u parts: four sums) -> dL/dh(t-1)'s first terms and dL/du; (2) dL/du crosses
// to the lower half (v_permlane32_swap), which forms d_a, d_p1, d_p2 -> LDS and the three transposed sums, which cross back; (3) the outer products
// d (x) q_l(input) into the row accumulators, both halves in the same instructions.  Every sum in the order of the first version: same bits.
template <int HP, bool NW, bool DX>
__global__ __launch_bounds__(64) void pgq_bwd_kernel(SeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const PgqLayout L = pgq_layout(a.H);
    const int H = L.H, T = a.T, NC = (T + kPC - 1) / kPC, H1 = H + 1, H2 = 2 * H;
    float* pl = smem;
    float* gw = smem + pad4(L.P);              // weight gradients in the parameter layout (deposited at the end)
    float* ftab = gw + pad4(L.P);              // [64][4]  |x|, cos, sin of the chunk's steps
    float* dxb = ftab + kPC * 4;               // [64][2]  dL/dx of the chunk's steps
    float* dyb = dxb + kPC * 2;                // [64][2]  dL/dy of the chunk's steps
    float* vb = dyb + kPC * 2;                 // [16][32] 0..4 q_l(h(t-1)), 5 q_3(u), 6 q_4(u), 7 d_f, 8 d_g, 9 d_a, 10 d_p1, 11 d_p2
    PgqQ Q;
    pgq_setup(pl, a, L, Q, lane);
    for (int i = lane; i < pad4(L.P); i += 64) gw[i] = 0.0f;
    for (int i = lane; i < 16 * 32; i += 64) vb[i] = 0.0f;
    const bool up = lane >= 32;
    const int j0 = lane & 31;
    const bool vo = j0 < H;
    const int j = vo ? j0 : 0;
    // transposed columns: lower c = 0..2: W_{a,p1,p2}[r][j] (c = 3: zeros); upper c = 0, 1: W_{f,g}[r][j], c = 2, 3: W_{f,g}[r][H + j]
    float cT[4][HP];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool col = vo && (up || c < 3);
        const int lay = up ? 3 + (c & 1) : (c < 3 ? c : 0);
        const int off = L.ow[lay] + j + ((up && c >= 2) ? H : 0), pitch = up ? H2 : H1;
#pragma unroll
        for (int r = 0; r < HP; ++r) cT[c][r] = (col && r < H) ? pl[off + r * pitch] : 0.0f;
    }
    float wsc[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) wsc[l] = (vo && !up) ? pl[L.ow[l] + j * H1 + H] : 0.0f;
    const float wo0 = (vo && up) ? pl[L.ow[5] + j] : 0.0f, wo1 = (vo && up) ? pl[L.ow[5] + H + j] : 0.0f;
    // gradient rows of unit j: lower c = 0..2: W_{a,p1,p2} (h part; gs: the scalar input's column); upper c = 0, 1: W_{f,g} h part, c = 2, 3: u part
    float gr[4][HP], gs[3] = {0.f, 0.f, 0.f}, dbx[3] = {0.f, 0.f, 0.f}, dwo0 = 0.0f, dwo1 = 0.0f, tb0 = 0.0f, tb1 = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < HP; ++k) gr[c][k] = 0.0f;
    const float* vn = vb + (up ? 3 * 32 : 0);  // outer products: accumulator c against slot (up ? 3 : 0) + c
    wave_lds_fence();
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        const float2* dyg = reinterpret_cast<const float2*>(a.dy) + (size_t)b * T;
        const float* sv = a.ckpt + (size_t)b * T * kPNS * 64;
        float dh = 0.0f;                                       // upper half: dL/dh of unit j0
        // the records of a step are fetched a step ahead (they come from HBM / the Infinity Cache: ~0.5 us that nothing in the step could cover)
        const int o0 = up ? 192 : 0, o1 = up ? 256 : 64, o2 = up ? 320 : 128;
        float n0, n1, n2, nht, nhp;
        {
            const float* s = sv + (size_t)(T - 1) * kPNS * 64 + j0;
            n0 = s[o0]; n1 = s[o1]; n2 = s[o2]; nht = s[384]; nhp = T > 1 ? s[384 - kPNS * 64] : 0.0f;
        }
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kPC, len = min(kPC, T - t0);
            wave_lds_fence();
            pgq_stage_inputs(ftab, xg, t0, T, lane);
            float2 dyv = make_float2(0.0f, 0.0f);
            if (lane < len) dyv = dyg[t0 + lane];
            reinterpret_cast<float2*>(dyb)[lane] = dyv;
            if constexpr (NW) { tb0 += dyv.x; tb1 += dyv.y; }
            wave_lds_fence();
            for (int tt = len - 1; tt >= 0; --tt) {
                const int t = t0 + tt;
                // lower half: a_n, p1, p2; upper half: u, f, g, h(t), h(t-1)
                const float r0 = n0, r1 = n1, r2 = n2, ht = nht, hp = nhp;
                if (t > 0) {
                    const float* s = sv + (size_t)(t - 1) * kPNS * 64 + j0;
                    n0 = s[o0]; n1 = s[o1]; n2 = s[o2]; nht = hp; nhp = t > 1 ? s[384 - kPNS * 64] : 0.0f;
                }
                const float an = r0, p1 = r1, p2 = r2, u = r0, f = r1, g = r2;
                const float4 in = reinterpret_cast<const float4*>(ftab)[tt];
                const float sc[3] = {in.x, in.y, in.z};
                // the step's quantised layer inputs (what the forward pass multiplied with), the read-out, d_f and d_g: upper half
                const float2 d = reinterpret_cast<const float2*>(dyb)[tt];
                const float hoq = q16::qapply(ht, Q.a[5]);
                dh = __builtin_fmaf(q16::qpass(ht, Q.a[5]), d.x * wo0 + d.y * wo1, dh);
                if constexpr (NW) { dwo0 = __builtin_fmaf(d.x, hoq, dwo0); dwo1 = __builtin_fmaf(d.y, hoq, dwo1); }
                const float dfp = (vo && up) ? (dh * (hp - g)) * (f * (1.0f - f)) : 0.0f;
                const float dgp = (vo && up) ? (dh * (1.0f - f)) * (1.0f - g * g) : 0.0f;
                float dhp = dh * f;
                if (up) {
#pragma unroll
                    for (int l = 0; l < 5; ++l) vb[l * 32 + j0] = vo ? q16::qapply(hp, Q.a[l]) : 0.0f;
                    vb[5 * 32 + j0] = vo ? q16::qapply(u, Q.a[3]) : 0.0f;
                    vb[6 * 32 + j0] = vo ? q16::qapply(u, Q.a[4]) : 0.0f;
                    vb[7 * 32 + j0] = dfp; vb[8 * 32 + j0] = dgp;
                }
                wave_lds_fence();
                // (1) W_f^T d_f, W_g^T d_g: the h parts (-> dL/dh(t-1)) and the u parts (-> dL/du), each through its layer's activation mask
                float du = 0.0f;
                {
                    float ah[2] = {0.f, 0.f}, au[2] = {0.f, 0.f};
#pragma unroll
                    for (int q4 = 0; q4 < HP / 4; ++q4) {
                        const float4 vf = *reinterpret_cast<const float4*>(vb + 7 * 32 + 4 * q4), vg = *reinterpret_cast<const float4*>(vb + 8 * 32 + 4 * q4);
                        const float df4[4] = {vf.x, vf.y, vf.z, vf.w}, dg4[4] = {vg.x, vg.y, vg.z, vg.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            ah[0] = __builtin_fmaf(cT[0][4 * q4 + e], df4[e], ah[0]); au[0] = __builtin_fmaf(cT[2][4 * q4 + e], df4[e], au[0]);
                            ah[1] = __builtin_fmaf(cT[1][4 * q4 + e], dg4[e], ah[1]); au[1] = __builtin_fmaf(cT[3][4 * q4 + e], dg4[e], au[1]);
                        }
                    }
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        dhp = __builtin_fmaf(q16::qpass(hp, Q.a[3 + l]), ah[l], dhp);
                        du = __builtin_fmaf(q16::qpass(u, Q.a[3 + l]), au[l], du);
                    }
                }
                // (2) dL/du to the lower half; u = A(a) A(p1) A(p2), A(v) = v (1 - v), A'(v) = 1 - 2 v; through the tanh of the three input gates
                const float dul = dup32(du).hi;
                const float Aa = an * (1.0f - an), Ab = p1 * (1.0f - p1), Ac = p2 * (1.0f - p2);
                float dpre[3];
                dpre[0] = (vo && !up) ? ((dul * (1.0f - 2.0f * an)) * (Ab * Ac)) * (1.0f - an * an) : 0.0f;
                dpre[1] = (vo && !up) ? ((dul * (1.0f - 2.0f * p1)) * (Aa * Ac)) * (1.0f - p1 * p1) : 0.0f;
                dpre[2] = (vo && !up) ? ((dul * (1.0f - 2.0f * p2)) * (Aa * Ab)) * (1.0f - p2 * p2) : 0.0f;
                if (!up) {
#pragma unroll
                    for (int l = 0; l < 3; ++l) vb[(9 + l) * 32 + j0] = dpre[l];
                }
                wave_lds_fence();
                float dsc[3] = {0.f, 0.f, 0.f};
                {
                    float ah[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int q4 = 0; q4 < HP / 4; ++q4)
#pragma unroll
                        for (int l = 0; l < 3; ++l) {
                            const float4 v = *reinterpret_cast<const float4*>(vb + (9 + l) * 32 + 4 * q4);
                            ah[l] = __builtin_fmaf(cT[l][4 * q4], v.x, ah[l]); ah[l] = __builtin_fmaf(cT[l][4 * q4 + 1], v.y, ah[l]);
                            ah[l] = __builtin_fmaf(cT[l][4 * q4 + 2], v.z, ah[l]); ah[l] = __builtin_fmaf(cT[l][4 * q4 + 3], v.w, ah[l]);
                        }
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        dhp = __builtin_fmaf(q16::qpass(hp, Q.a[l]), dup32(ah[l]).lo, dhp);      // (the lower half's sum, on the upper half)
                        if constexpr (DX) {      // the scalar input's column: sum over the units
                            float v = wsc[l] * dpre[l];
                            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                            dsc[l] = v * q16::qpass(sc[l], Q.a[l]);
                        }
                    }
                }
                if constexpr (NW) {      // (3) the rows of unit j: d (x) q_l(input) (the d's are 0 on lanes without a unit)
                    const float dsel[4] = {up ? dfp : dpre[0], up ? dgp : dpre[1], up ? dfp : dpre[2], up ? dgp : 0.0f};
#pragma unroll
                    for (int q4 = 0; q4 < HP / 4; ++q4)
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4) {
                            const float4 v = *reinterpret_cast<const float4*>(vn + c4 * 32 + 4 * q4);
                            gr[c4][4 * q4] = __builtin_fmaf(dsel[c4], v.x, gr[c4][4 * q4]); gr[c4][4 * q4 + 1] = __builtin_fmaf(dsel[c4], v.y, gr[c4][4 * q4 + 1]);
                            gr[c4][4 * q4 + 2] = __builtin_fmaf(dsel[c4], v.z, gr[c4][4 * q4 + 2]); gr[c4][4 * q4 + 3] = __builtin_fmaf(dsel[c4], v.w, gr[c4][4 * q4 + 3]);
                        }
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        gs[l] = __builtin_fmaf(dpre[l], q16::qapply(sc[l], Q.a[l]), gs[l]);
                        dbx[l] += dsel[l];
                    }
                }
                if constexpr (DX) {
                    if (lane == 0) {      // theta = atan2(Q, I): dtheta = -sin dcos + cos dsin; dtheta/dI = -Q / a^2, dtheta/dQ = I / a^2
                        const float am = in.x, ct = in.y, st = in.z, I = ct * am, Qv = st * am, a2 = am * am;
                        const float dth = -st * dsc[1] + ct * dsc[2];
                        reinterpret_cast<float2*>(dxb)[tt] = make_float2(dsc[0] * I / am - dth * Qv / a2, dsc[0] * Qv / am + dth * I / a2);
                    }
                }
                dh = (vo && up) ? dhp : 0.0f;
                wave_lds_fence();
            }
            if constexpr (DX) {
                wave_lds_fence();
                if (lane < len) reinterpret_cast<float2*>(a.dx)[(size_t)b * T + t0 + lane] = reinterpret_cast<const float2*>(dxb)[lane];
            }
        }
        wave_lds_fence();
    }
    if constexpr (NW) {
        float* prow = a.partials + (size_t)blockIdx.x * (L.P + kLossCols);
        for (int o = 32; o > 0; o >>= 1) { tb0 += __shfl_xor(tb0, o); tb1 += __shfl_xor(tb1, o); }
        if (vo && !up) {
#pragma unroll
            for (int l = 0; l < 3; ++l) {
#pragma unroll
                for (int k = 0; k < HP; ++k) if (k < H) gw[L.ow[l] + j * H1 + k] = gr[l][k];
                gw[L.ow[l] + j * H1 + H] = gs[l];
                gw[L.ob[l] + j] = dbx[l];
            }
        }
        if (vo && up) {
#pragma unroll
            for (int l = 0; l < 2; ++l) {
#pragma unroll
                for (int k = 0; k < HP; ++k) if (k < H) { gw[L.ow[3 + l] + j * H2 + k] = gr[l][k]; gw[L.ow[3 + l] + j * H2 + H + k] = gr[2 + l][k]; }
                gw[L.ob[3 + l] + j] = dbx[l];
            }
            gw[L.ow[5] + j] = dwo0; gw[L.ow[5] + H + j] = dwo1;
        }
        if (lane == 0) { gw[L.ob[5]] = tb0; gw[L.ob[5] + 1] = tb1; }
        wave_lds_fence();
        // weight quantisers' pass masks from the unquantised weights; scale columns exact zeros
        q16::Quant qw[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) qw[l] = q16::make_quant(a.params[L.oq[l]], a.bits_w);
        for (int i = lane; i < L.P + kLossCols; i += 64) {
            float v = i < L.P ? gw[i] : 0.0f;
#pragma unroll
            for (int l = 0; l < 6; ++l) {
                if (i >= L.ow[l] && i < L.ob[l]) v *= q16::qpass(a.params[i], qw[l]);
                if (i >= L.oq[l] && i < L.oq[l] + 3) v = 0.0f;
            }
            prow[i] = v;
        }
    }
}

template <typename K> int pgq_launch(hipStream_t st, K k, int grid, size_t lds, const SeqArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

bool pgjanet_q_ok(const odpd_model_t* m) {
    return m->backbone == ODPD_PGJANET && m->bits_w > 0 && m->bits_w <= 16 && m->bits_a > 0 && m->bits_a <= 16 && m->hidden >= 1 && m->hidden <= 32 &&
           !(m->flags & ODPD_FLAG_TWO_LAYERS);
}
int64_t pgjanet_q_param_count(const odpd_model_t* m) { return pgq_layout(m->hidden).P; }
int64_t pgjanet_q_ckpt_floats(const odpd_model_t*, int B, int T) { return (int64_t)B * T * kPNS * 64; }
int pgjanet_q_rows(const odpd_model_t*, int B) { const int cap = 4 * device_cus(); return B < cap ? B : cap; }
int pgjanet_q_fwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!pgjanet_q_ok(m)) return ODPD_EUNSUPPORTED;
    const size_t lds = (size_t)pgq_fwd_floats(pgq_layout(m->hidden).P) * sizeof(float);
    const int grid = pgjanet_q_rows(m, a.B);
    if (m->hidden <= 16) return a.ckpt ? pgq_launch(st, pgq_fwd_kernel<16, true>, grid, lds, a) : pgq_launch(st, pgq_fwd_kernel<16, false>, grid, lds, a);
    return a.ckpt ? pgq_launch(st, pgq_fwd_kernel<32, true>, grid, lds, a) : pgq_launch(st, pgq_fwd_kernel<32, false>, grid, lds, a);
}
int pgjanet_q_bwd(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!pgjanet_q_ok(m)) return ODPD_EUNSUPPORTED;
    if (!a.ckpt) return ODPD_EINVAL;
    const size_t lds = (size_t)pgq_bwd_floats(pgq_layout(m->hidden).P) * sizeof(float);
    const int grid = pgjanet_q_rows(m, a.B);
    const bool nw = a.partials != nullptr, dx = a.dx != nullptr;
#define ODPD_PGQ_BWD(HP_)                                                                           \
    {                                                                                              \
        if (nw && dx) return pgq_launch(st, pgq_bwd_kernel<HP_, true, true>, grid, lds, a);        \
        if (nw) return pgq_launch(st, pgq_bwd_kernel<HP_, true, false>, grid, lds, a);             \
        return pgq_launch(st, pgq_bwd_kernel<HP_, false, true>, grid, lds, a);                     \
    }
    if (m->hidden <= 16) ODPD_PGQ_BWD(16)
    ODPD_PGQ_BWD(32)
#undef ODPD_PGQ_BWD
}

}  // namespace odpd
