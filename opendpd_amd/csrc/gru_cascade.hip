// gru_cascade.hip — train_dpd (steps/train_dpd.py:60-63, models.py:163-176: y = PA(DPD(x)), PA frozen) at the reference's own batch
// sizes (64 .. 256 frames of 50 / 200 samples, train_funcs.py:28-48) as ONE launch: the DPD and the frozen PA of a frame run as the two
// waves of a workgroup, on SIMDs of their own, and hand the frame over through LDS 32 steps (kCascChunk) at a time.
//
// At these batches a sequence has a SIMD to itself and the T-serial dependency chains are the whole cost (gru_family.hip,
// gru_gp_train_kernel).  Chained launches pay DPD forward + PA forward + PA backward + DPD forward again + DPD backward; here
//   forward    DPD wave: chunk k (features, recurrence, fc_out with lane = time step -> u and the PA's features of u in LDS)
//              PA wave:  chunk k - 1 (recurrence)                                              — one workgroup barrier per chunk
//   loss       PA wave: fc_out, loss and dL/dy of all T steps, lane = time step
//   backward   PA wave:  chunk c (recurrence, parking the pre-activation gradients; then dL/du of the chunk, lane = time step -> LDS)
//              DPD wave: the chunk above it (recurrence + weight gradients as 4-block MFMAs)  — one barrier per chunk
// so the DPD's chains hide behind the PA's but for one chunk at either end, the DPD's forward runs once, and u / dL/du never leave LDS.
// The per-step arithmetic is gru_gp_train_kernel's (same gate-parallel mapping: rows r | n | head | z of a wave, one rotated dot
// product per step and orientation); DPD and PA of <= 32 units (two 16-unit blocks above 16; PAs of 17..24 on the half-block layout).
// One partial-gradient row per workgroup: (P_dpd + kLossCols), column P_dpd = the loss partial sum.
// delta_cascade_kernel / lstm_cascade_kernel: the same workgroup with a deltagru / TRes-DeltaGRU (odpd_deltaseq.h) or a plain LSTM
// (odpd_lstm.h) of <= 16 units as the DPD.
#include "odpd_gpseq.h"
#include "odpd_deltaseq.h"
#include "odpd_lstm.h"

namespace odpd {

// The frozen PA's wave of a cascade workgroup.  One workgroup barrier per hand-off (the DPD wave executes the same number): forward
// chunk k - 1 while the DPD wave produces chunk k; loss; backward chunk c + dL/du of the chunk into the DPD wave's dL/dy buffer.
template <typename P>
__device__ __forceinline__ void casc_pa_wave(const CascArgs& a, float* smem, float* rp, float2* pa_ubuf, float2* dpd_dyb, float* xch) {
    const int lane = threadIdx.x & 63, T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    P e;
    e.setup(smem, rp, a.pa_params, a.Hp, T);
    __syncthreads();
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    float2* pa_dyb = reinterpret_cast<float2*>(e.dyb);
    float loss_acc = 0.0f;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        e.fwd_begin();
        for (int k = 0; k <= NC; ++k) {
            if (k >= 1) {
                const int t0 = (k - 1) * kCascChunk;
                e.fwd_steps(t0, min(kCascChunk, T - t0));
            }
            __syncthreads();
        }
        e.store_act(T - 1);
        wave_lds_fence();
        for (int t0 = 0; t0 < T; t0 += kCascChunk)
            e.head_chunk(t0, min(kCascChunk, T - t0), [&](int t, float y0, float y1) {
                const float2 tv = tg[t];
                float dy0, dy1;
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, dy0, dy1, loss_acc);
                pa_dyb[t] = make_float2(dy0, dy1);
            });
        wave_lds_fence();
        e.bwd_begin();
        for (int k = 0; k <= NC; ++k) {
            if (k < NC) {
                const int c = NC - 1 - k, t0 = c * kCascChunk, len = min(kCascChunk, T - t0);
                e.bwd_steps(t0 + len - 1, t0, t0);
                wave_lds_fence();
                e.dx_chunk(t0, len, pa_ubuf, dpd_dyb);
                wave_lds_fence();
            }
            __syncthreads();
        }
    }
    float lp = loss_acc;
    for (int o = 32; o > 0; o >>= 1) lp += __shfl_down(lp, o);
    if (lane == 0) xch[0] = lp;
    __syncthreads();
}

// PA variants: PV = 0 hidden <= 16 | 1 hidden 17..24 (two blocks, the second held twice) | 2 hidden 25..32
// NBD: unit blocks of the DPD (1: hidden <= 16, 2: hidden 17..32 — e.g. the qgru H20 / H30 of quant_qgru_dpd_regr.sh's float stage)
template <int NBD, int FMD, bool DGD, int PV, int FMP, bool DGP>
__global__ __launch_bounds__(128) void gru_cascade_kernel(CascArgs a) {
    using D = GpSeq<NBD, FMD, DGD, true>;
    using P = GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    const int Pd = gru_layout(a.Hd, D::F, DGD).P, Pp = gru_layout(a.Hp, P::F, DGP).P;
    float* rd = smem;                                   // the DPD wave's region
    float* rp = rd + D::region_floats(T, Pd);           // the PA wave's
    float* xch = rp + P::region_floats(T, Pp);          // [4]: the loss on its way to the partial row
    float* pa_ftab = rp + pad4(Pp);
    float2* pa_ubuf = reinterpret_cast<float2*>(rp + pad4(Pp) + P::off_ubuf(T));
    float2* dpd_dyb = reinterpret_cast<float2*>(rd + pad4(Pd) + D::off_dyb(T));
    if (wave == 0) {
        // ---------------- the DPD (trained) ----------------
        D e;
        e.setup(smem, rd, a.dpd_params, a.Hd, T);
        __syncthreads();
        for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
            const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
            const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
            e.fwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k < NC) {
                    const int t0 = k * kCascChunk, len = min(kCascChunk, T - t0);
                    if (lane < len) { const float2 raw = xg[t0 + lane]; D::write_feat(e.ftab, t0 + lane, raw.x, raw.y); }
                    wave_lds_fence();
                    e.fwd_steps(t0, len);
                    e.store_act(t0 + len - 1);
                    wave_lds_fence();
                    e.head_chunk(t0, len, [&](int t, float y0, float y1) {
                        pa_ubuf[t] = make_float2(y0, y1);
                        P::write_feat(pa_ftab, t, y0, y1);
                    });
                }
                __syncthreads();
            }
            for (int k = 0; k <= NC; ++k) {
                if (k >= 1) {
                    const int c = NC - k;
                    if (k == 1) e.bwd_begin();
                    e.bwd_steps(min(T - 1, c * kCascChunk + kCascChunk), c > 0 ? c * kCascChunk + 1 : 0, 0);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        e.write_partials(a.partials + (size_t)blockIdx.x * (Pd + kLossCols), xch[0]);
    } else {
        // ---------------- the PA (frozen) ----------------
        casc_pa_wave<P>(a, smem, rp, pa_ubuf, dpd_dyb, xch);
    }
}

// The same workgroup with a delta backbone as the DPD (deltagru / TRes-DeltaGRU — BASELINE config 3, the OpenDPDv2 model — hidden <= 16):
// DeltaSeq's forward chunk produces u (fc_out + TCN skip) and the PA's features; its backward chunk first runs the chunk's forward steps
// again from the cell state kept at the chunk start (odpd_deltaseq.h), then back-propagates with the dL/du the PA wave left in LDS.
template <bool TRES, int PV, int FMP, bool DGP>
__global__ __launch_bounds__(128) void delta_cascade_kernel(CascArgs a) {
    using D = DeltaSeq<TRES>;
    using P = GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>;
    static_assert(D::C == kCascChunk, "one hand-off granularity");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    const int Pd = delta_layout(a.Hd, TRES).P, Pp = gru_layout(a.Hp, P::F, DGP).P;
    float* rd = smem;
    float* rp = rd + D::region_floats(T, Pd);
    float* xch = rp + P::region_floats(T, Pp);
    float* pa_ftab = rp + pad4(Pp);
    float2* pa_ubuf = reinterpret_cast<float2*>(rp + pad4(Pp) + P::off_ubuf(T));
    float2* dpd_dyb = reinterpret_cast<float2*>(rd + pad4(Pd) + D::off_dyb(T));
    if (wave == 0) {
        D e;
        e.setup(smem, rd, a.dpd_params, a.Hd, T, a.thx, a.thh);
        __syncthreads();
        for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
            const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
            const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
            e.fwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k < NC) {
                    const int t0 = k * kCascChunk;
                    e.fwd_chunk(k, t0, min(kCascChunk, T - t0), xg, [&](int t, float y0, float y1) {
                        pa_ubuf[t] = make_float2(y0, y1);
                        P::write_feat(pa_ftab, t, y0, y1);
                    });
                }
                __syncthreads();
            }
            e.bwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k >= 1) {
                    const int c = NC - k, t0 = c * kCascChunk;
                    e.bwd_chunk(c, t0, min(kCascChunk, T - t0), xg);
                }
                __syncthreads();
            }
            e.bwd_end();
        }
        __syncthreads();
        e.write_partials(a.partials + (size_t)blockIdx.x * (Pd + kLossCols), xch[0]);
        e.add_stats(a.stats, a.B);
    } else {
        casc_pa_wave<P>(a, smem, rp, pa_ubuf, dpd_dyb, xch);
    }
}

// ... and with a plain LSTM (backbones/lstm.py) as the DPD (LstmSeq, odpd_lstm.h; hidden <= 16)
template <int PV, int FMP, bool DGP>
__global__ __launch_bounds__(128) void lstm_cascade_kernel(CascArgs a) {
    using D = LstmSeq;
    using P = GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    const int Pd = lstm_layout(a.Hd, 0).P, Pp = gru_layout(a.Hp, P::F, DGP).P;
    float* rd = smem;
    float* rp = rd + D::region_floats(T, Pd);
    float* xch = rp + P::region_floats(T, Pp);
    float* pa_ftab = rp + pad4(Pp);
    float2* pa_ubuf = reinterpret_cast<float2*>(rp + pad4(Pp) + P::off_ubuf(T));
    float2* dpd_dyb = reinterpret_cast<float2*>(rd + pad4(Pd) + D::off_dyb(T));
    if (wave == 0) {
        D e;
        e.setup(smem, rd, a.dpd_params, a.Hd, T);
        __syncthreads();
        for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
            const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
            const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
            e.fwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k < NC) {
                    const int t0 = k * kCascChunk;
                    e.fwd_chunk(t0, min(kCascChunk, T - t0), xg, [&](int t, float y0, float y1) {
                        pa_ubuf[t] = make_float2(y0, y1);
                        P::write_feat(pa_ftab, t, y0, y1);
                    });
                }
                __syncthreads();
            }
            e.bwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k >= 1) {
                    const int c = NC - k, t0 = c * kCascChunk;
                    e.bwd_steps(min(T, t0 + kCascChunk) - 1, t0);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        e.write_partials(a.partials + (size_t)blockIdx.x * (Pd + kLossCols), xch[0]);
    } else {
        casc_pa_wave<P>(a, smem, rp, pa_ubuf, dpd_dyb, xch);
    }
}

}  // namespace odpd
// ... and with a quantisation-aware GRUCell model as the DPD (BASELINE config 5: QGRU INT8 train_dpd; QatSeq, odpd_qatseq.h).
// (Everything from here on is compiled with FP contraction off — odpd_qat.h; the kernels above are not affected.)
#include "odpd_qatseq.h"
namespace odpd {
template <int MK, bool LUT, int PV, int FMP, bool DGP>
__global__ __launch_bounds__(128) void qat_cascade_kernel(CascArgs a) {
    using D = q16::QatSeq<MK, LUT>;
    using P = GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>;
    static_assert(D::C == kCascChunk, "one hand-off granularity");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    const int Pd = q16::qat_layout(MK, a.Hd).P, Pp = gru_layout(a.Hp, P::F, DGP).P;
    float* rd = smem;
    float* rp = rd + D::region_floats(T, Pd);
    float* xch = rp + P::region_floats(T, Pp);
    float* pa_ftab = rp + pad4(Pp);
    float2* pa_ubuf = reinterpret_cast<float2*>(rp + pad4(Pp) + P::off_ubuf(T));
    float2* dpd_dyb = reinterpret_cast<float2*>(rd + D::off_dyb_region(T, Pd));
    if (wave == 0) {
        D e;
        e.setup(smem, rd, a.dpd_params, a.Hd, T, T, a.bits_w, a.bits_a);
        __syncthreads();
        for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
            const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
            const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
            e.fwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k < NC) {
                    const int t0 = k * kCascChunk;
                    e.fwd_chunk(k, t0, min(kCascChunk, T - t0), xg, [&](int t, float y0, float y1) {
                        pa_ubuf[t] = make_float2(y0, y1);
                        P::write_feat(pa_ftab, t, y0, y1);
                    });
                }
                __syncthreads();
            }
            e.bwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k >= 1) {
                    const int c = NC - k, t0 = c * kCascChunk;
                    e.bwd_chunk(c, t0, min(kCascChunk, T - t0));
                }
                __syncthreads();
            }
        }
        __syncthreads();
        e.write_partials(a.partials + (size_t)blockIdx.x * (Pd + kLossCols), xch[0]);
    } else {
        casc_pa_wave<P>(a, smem, rp, pa_ubuf, dpd_dyb, xch);
    }
}

// ... and the quantised TRes-DeltaGRU (the OpenDPDv2 QAT stage; QatDeltaSeq)
template <bool LUT, int PV, int FMP, bool DGP>
__global__ __launch_bounds__(128) void qat_delta_cascade_kernel(CascArgs a) {
    using D = q16::QatDeltaSeq<LUT>;
    using P = GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>;
    static_assert(D::C == kCascChunk, "one hand-off granularity");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    const int Pd = q16::qat_layout(q16::K_TRES, a.Hd).P, Pp = gru_layout(a.Hp, P::F, DGP).P;
    float* rd = smem;
    float* rp = rd + D::region_floats(T, Pd);
    float* xch = rp + P::region_floats(T, Pp);
    float* pa_ftab = rp + pad4(Pp);
    float2* pa_ubuf = reinterpret_cast<float2*>(rp + pad4(Pp) + P::off_ubuf(T));
    float2* dpd_dyb = reinterpret_cast<float2*>(rd + D::off_dyb_region(T, Pd));
    if (wave == 0) {
        D e;
        e.setup(smem, rd, a.dpd_params, a.Hd, T, T, a.bits_w, a.bits_a, a.thx, a.thh);
        __syncthreads();
        for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
            const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
            const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
            e.fwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k < NC) {
                    const int t0 = k * kCascChunk;
                    e.fwd_chunk(k, t0, min(kCascChunk, T - t0), xg, [&](int t, float y0, float y1) {
                        pa_ubuf[t] = make_float2(y0, y1);
                        P::write_feat(pa_ftab, t, y0, y1);
                    });
                }
                __syncthreads();
            }
            e.bwd_begin();
            for (int k = 0; k <= NC; ++k) {
                if (k >= 1) {
                    const int c = NC - k, t0 = c * kCascChunk;
                    e.bwd_chunk(c, t0, min(kCascChunk, T - t0), xg);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        e.write_partials(a.partials + (size_t)blockIdx.x * (Pd + kLossCols), xch[0]);
        e.add_stats(a.stats, a.B);
    } else {
        casc_pa_wave<P>(a, smem, rp, pa_ubuf, dpd_dyb, xch);
    }
}

// Evaluation passes of the quantised models (net_eval / run_dpd shapes: a few long sequences, torch.no_grad()) on the same engines: ONE
// sequence per single-wave workgroup, forward chunks only on a one-chunk ring of buffers, fc_out's 16-bit output quantiser in eval mode.
template <typename E, bool TRES>
__global__ __launch_bounds__(64) void qat_eval_kernel(SeqArgs a, int bits_w, int bits_a, int eval_mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T;
    E e;
    if constexpr (TRES) e.setup(smem, smem, a.params, a.H, T, kCascChunk, bits_w, bits_a, a.thx, a.thh);
    else e.setup(smem, smem, a.params, a.H, T, kCascChunk, bits_w, bits_a);
    e.ring = true; e.eval_out = eval_mode != 0;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const float2* xg = reinterpret_cast<const float2*>(a.x) + (size_t)b * T;
        float2* yg = reinterpret_cast<float2*>(a.y) + (size_t)b * T;
        e.fwd_begin();
        for (int t0 = 0; t0 < T; t0 += kCascChunk)
            e.fwd_chunk(0, t0, min(kCascChunk, T - t0), xg, [&](int t, float y0, float y1) { yg[t] = make_float2(y0, y1); });
    }
    if constexpr (TRES) e.add_stats(a.stats, a.B);
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {
constexpr int kDpdDelta = 100, kDpdTres = 101, kDpdLstm = 102;
constexpr int kDpdQat = 200;       // + 2 * kind (q16::K_GRU / K_Q4 / K_A4) + (LUT gates: <= 8 bits)
constexpr int kDpdQatTres = 300;   // + (LUT gates)      // CascCfg::fmd of the delta DPDs (the GRU-family ones carry their feature mode)
struct CascCfg { int fmd, fmp, pv, nbd, Pd, Pp; bool dgd, dgp; };
bool casc_model(const odpd_model_t* m, int& fm, bool& dg) {
    if (m->bits_w > 0) return false;
    switch (m->backbone) {
    case ODPD_GRU: fm = FEAT_RAW2; dg = false; return true;
    case ODPD_DGRU: fm = FEAT_DGRU6; dg = true; return true;
    case ODPD_QGRU: fm = FEAT_Q4; dg = false; return true;
    case ODPD_QGRU_AMP1: fm = FEAT_A4; dg = false; return true;
    default: return false;
    }
}
int feat_dim(int fm) { return fm == FEAT_RAW2 ? 2 : fm == FEAT_DGRU6 ? 6 : 4; }
bool casc_cfg(const odpd_model_t* dpd, const odpd_model_t* pa, CascCfg& c) {
    if (!casc_model(pa, c.fmp, c.dgp)) return false;
    const bool delta = dpd->bits_w == 0 && (dpd->backbone == ODPD_DELTAGRU || dpd->backbone == ODPD_TRES_DELTAGRU);
    const bool lstm = dpd->bits_w == 0 && dpd->backbone == ODPD_LSTM;
    const bool qat = dpd->bits_w > 0 && dpd->bits_a > 0 && (dpd->backbone == ODPD_GRU || dpd->backbone == ODPD_QGRU || dpd->backbone == ODPD_QGRU_AMP1);
    const int qkind = dpd->backbone == ODPD_GRU ? q16::K_GRU : dpd->backbone == ODPD_QGRU ? q16::K_Q4 : q16::K_A4;
    const bool qtres = dpd->bits_w > 0 && dpd->bits_a > 0 && dpd->backbone == ODPD_TRES_DELTAGRU;
    if (delta) { c.fmd = dpd->backbone == ODPD_TRES_DELTAGRU ? kDpdTres : kDpdDelta; c.dgd = false; }
    else if (lstm) { c.fmd = kDpdLstm; c.dgd = false; }
    else if (qat) { c.fmd = kDpdQat + 2 * qkind + ((dpd->bits_w <= 8 && dpd->bits_a <= 8) ? 1 : 0); c.dgd = false; }
    else if (qtres) { c.fmd = kDpdQatTres + ((dpd->bits_w <= 8 && dpd->bits_a <= 8) ? 1 : 0); c.dgd = false; }
    else if (!casc_model(dpd, c.fmd, c.dgd)) return false;
    if (dpd->hidden < 1 || dpd->hidden > ((delta || lstm || qat || qtres) ? 16 : 32) || pa->hidden < 1 || pa->hidden > 32) return false;
    c.nbd = dpd->hidden > 16 ? 2 : 1;
    if (c.fmp != FEAT_RAW2 && c.fmp != FEAT_DGRU6) return false;      // PAs of the reference's scripts: gru, dgru
    c.pv = pa->hidden > 24 ? 2 : pa->hidden > 16 ? 1 : 0;
    c.Pd = delta ? delta_layout(dpd->hidden, c.fmd == kDpdTres).P : lstm ? lstm_layout(dpd->hidden, 0).P : qat ? q16::qat_layout(qkind, dpd->hidden).P : qtres ? q16::qat_layout(q16::K_TRES, dpd->hidden).P
                                                                     : gru_layout(dpd->hidden, feat_dim(c.fmd), c.dgd).P;
    c.Pp = gru_layout(pa->hidden, feat_dim(c.fmp), c.dgp).P;
    return true;
}
// the DPD engine of a CascCfg::fmd
template <int NBD, int FMD, bool DGD> struct DpdEngine { using type = GpSeq<NBD, FMD, DGD, true>; };
template <> struct DpdEngine<1, kDpdDelta, false> { using type = DeltaSeq<false>; };
template <> struct DpdEngine<1, kDpdTres, false> { using type = DeltaSeq<true>; };
template <> struct DpdEngine<1, kDpdLstm, false> { using type = LstmSeq; };
#define ODPD_QAT_ENGINE(MK_) \
    template <> struct DpdEngine<1, kDpdQat + 2 * MK_, false> { using type = q16::QatSeq<MK_, false>; };      \
    template <> struct DpdEngine<1, kDpdQat + 2 * MK_ + 1, false> { using type = q16::QatSeq<MK_, true>; };
ODPD_QAT_ENGINE(q16::K_GRU) ODPD_QAT_ENGINE(q16::K_Q4) ODPD_QAT_ENGINE(q16::K_A4)
#undef ODPD_QAT_ENGINE
template <> struct DpdEngine<1, kDpdQatTres, false> { using type = q16::QatDeltaSeq<false>; };
template <> struct DpdEngine<1, kDpdQatTres + 1, false> { using type = q16::QatDeltaSeq<true>; };
template <int NBD, int FMD, bool DGD, int PV, int FMP, bool DGP>
size_t casc_lds(int T, int Pd, int Pp) {
    return ((size_t)DpdEngine<NBD, FMD, DGD>::type::region_floats(T, Pd) + GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>::region_floats(T, Pp) + 4) *
           sizeof(float);
}
#define ODPD_CASC_PA(NBD_, FMD_, DGD_, CALL)                                                            \
    if (c.nbd == NBD_ && c.fmd == FMD_) {                                                               \
        if (c.pv == 0 && c.fmp == FEAT_RAW2) return CALL(NBD_, FMD_, DGD_, 0, FEAT_RAW2, false);        \
        if (c.pv == 1 && c.fmp == FEAT_RAW2) return CALL(NBD_, FMD_, DGD_, 1, FEAT_RAW2, false);        \
        if (c.pv == 2 && c.fmp == FEAT_RAW2) return CALL(NBD_, FMD_, DGD_, 2, FEAT_RAW2, false);        \
        if (c.pv == 0 && c.fmp == FEAT_DGRU6) return CALL(NBD_, FMD_, DGD_, 0, FEAT_DGRU6, true);       \
        if (c.pv == 1 && c.fmp == FEAT_DGRU6) return CALL(NBD_, FMD_, DGD_, 1, FEAT_DGRU6, true);       \
        if (c.pv == 2 && c.fmp == FEAT_DGRU6) return CALL(NBD_, FMD_, DGD_, 2, FEAT_DGRU6, true);       \
    }
#define ODPD_CASC_ALL(CALL)                    \
    ODPD_CASC_PA(1, FEAT_RAW2, false, CALL)    \
    ODPD_CASC_PA(1, FEAT_DGRU6, true, CALL)    \
    ODPD_CASC_PA(1, FEAT_Q4, false, CALL)      \
    ODPD_CASC_PA(1, FEAT_A4, false, CALL)      \
    ODPD_CASC_PA(2, FEAT_RAW2, false, CALL)    \
    ODPD_CASC_PA(2, FEAT_DGRU6, true, CALL)    \
    ODPD_CASC_PA(2, FEAT_Q4, false, CALL)      \
    ODPD_CASC_PA(2, FEAT_A4, false, CALL)      \
    ODPD_CASC_PA(1, kDpdDelta, false, CALL)    \
    ODPD_CASC_PA(1, kDpdTres, false, CALL)     \
    ODPD_CASC_PA(1, kDpdLstm, false, CALL)     \
    ODPD_CASC_PA(1, kDpdQat + 2 * q16::K_GRU, false, CALL) ODPD_CASC_PA(1, kDpdQat + 2 * q16::K_GRU + 1, false, CALL) \
    ODPD_CASC_PA(1, kDpdQat + 2 * q16::K_Q4, false, CALL)  ODPD_CASC_PA(1, kDpdQat + 2 * q16::K_Q4 + 1, false, CALL)  \
    ODPD_CASC_PA(1, kDpdQat + 2 * q16::K_A4, false, CALL)  ODPD_CASC_PA(1, kDpdQat + 2 * q16::K_A4 + 1, false, CALL)  \
    ODPD_CASC_PA(1, kDpdQatTres, false, CALL)              ODPD_CASC_PA(1, kDpdQatTres + 1, false, CALL)

size_t casc_lds_bytes(const CascCfg& c, int T) {
#define ODPD_CASC_LDS(NBD_, FMD_, DGD_, PV_, FMP_, DGP_) casc_lds<NBD_, FMD_, DGD_, PV_, FMP_, DGP_>(T, c.Pd, c.Pp)
    ODPD_CASC_ALL(ODPD_CASC_LDS)
#undef ODPD_CASC_LDS
    return 0;
}
int casc_grid(const CascCfg& c, int B, int T) {
    const size_t lds = casc_lds_bytes(c, T);
    if (lds == 0 || lds > kMaxLds) return 0;
    const int per_cu = (int)(kMaxLds / lds);
    const long cap = (long)device_cus() * (per_cu < 2 ? per_cu : 2);        // a workgroup = two waves, on SIMDs of their own
    return B <= cap ? B : 0;                                               // every frame resident at once
}
template <int NBD, int FMD, bool DGD, int PV, int FMP, bool DGP>
int casc_launch(hipStream_t st, const CascArgs& a, const CascCfg& c) {
    const size_t lds = casc_lds<NBD, FMD, DGD, PV, FMP, DGP>(a.T, c.Pd, c.Pp);
    auto launch = [&](auto k) {
        if (int e = allow_big_lds(k, lds)) return e;
        hipLaunchKernelGGL(k, dim3(casc_grid(c, a.B, a.T)), dim3(128), lds, st, a);
        return (int)hipGetLastError();
    };
    if constexpr (FMD == kDpdDelta || FMD == kDpdTres) return launch(delta_cascade_kernel<FMD == kDpdTres, PV, FMP, DGP>);
    else if constexpr (FMD == kDpdLstm) return launch(lstm_cascade_kernel<PV, FMP, DGP>);
    else if constexpr (FMD >= kDpdQatTres) return launch(qat_delta_cascade_kernel<FMD == kDpdQatTres + 1, PV, FMP, DGP>);
    else if constexpr (FMD >= kDpdQat) return launch(qat_cascade_kernel<(FMD - kDpdQat) / 2, ((FMD - kDpdQat) & 1) != 0, PV, FMP, DGP>);
    else return launch(gru_cascade_kernel<NBD, FMD, DGD, PV, FMP, DGP>);
}
}  // namespace

// rows of partials (> 0) if the pair of models and the batch shape are served by the one-launch cascade step, else ODPD_EUNSUPPORTED
int gru_cascade_rows(const odpd_model_t* dpd, const odpd_model_t* pa, int B, int T) {
    CascCfg c;
    if (!casc_cfg(dpd, pa, c) || tuning().gp_max_batch == 0 || !tuning().cascade_one_launch) return ODPD_EUNSUPPORTED;
    const int g = casc_grid(c, B, T);
    return g > 0 ? g : (int)ODPD_EUNSUPPORTED;
}
int gru_cascade_train(hipStream_t st, const odpd_model_t* dpd, const odpd_model_t* pa, const CascArgs& a) {
    CascCfg c;
    if (!casc_cfg(dpd, pa, c) || casc_grid(c, a.B, a.T) <= 0) return ODPD_EUNSUPPORTED;
#define ODPD_CASC_LAUNCH(NBD_, FMD_, DGD_, PV_, FMP_, DGP_) casc_launch<NBD_, FMD_, DGD_, PV_, FMP_, DGP_>(st, a, c)
    ODPD_CASC_ALL(ODPD_CASC_LAUNCH)
#undef ODPD_CASC_LAUNCH
    return ODPD_EUNSUPPORTED;
}


// evaluation passes of the quantised models on the one-sequence-per-wave engines: no checkpoints asked for, every sequence on a SIMD of its own
bool qat_uses_gp_eval(const odpd_model_t* m, int B, bool want_ckpt) {
    if (want_ckpt || m->bits_w <= 0 || m->bits_a <= 0 || m->hidden < 1 || m->hidden > 16 || tuning().gp_max_batch == 0 || tuning().s16_min_batch == 0) return false;
    if (m->backbone != ODPD_GRU && m->backbone != ODPD_QGRU && m->backbone != ODPD_QGRU_AMP1 && m->backbone != ODPD_TRES_DELTAGRU) return false;
    return B <= 2 * device_cus();
}
namespace {
template <typename E, bool TRES>
int qat_eval_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a, int P) {
    const size_t lds = (size_t)E::region_floats(kCascChunk, P) * sizeof(float);
    auto k = qat_eval_kernel<E, TRES>;
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(64), lds, st, a, (int)m->bits_w, (int)m->bits_a, (m->flags & ODPD_FLAG_EVAL) ? 1 : 0);
    return (int)hipGetLastError();
}
}  // namespace
int qat_gp_eval(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    const bool lut = m->bits_w <= 8 && m->bits_a <= 8;
    if (m->backbone == ODPD_TRES_DELTAGRU) {
        const int P = q16::qat_layout(q16::K_TRES, m->hidden).P;
        return lut ? qat_eval_launch<q16::QatDeltaSeq<true>, true>(st, m, a, P) : qat_eval_launch<q16::QatDeltaSeq<false>, true>(st, m, a, P);
    }
#define ODPD_QAT_EVAL(BB_, MK_)                                                                                                     \
    if (m->backbone == BB_) {                                                                                                       \
        const int P = q16::qat_layout(MK_, m->hidden).P;                                                                           \
        return lut ? qat_eval_launch<q16::QatSeq<MK_, true>, false>(st, m, a, P) : qat_eval_launch<q16::QatSeq<MK_, false>, false>(st, m, a, P); \
    }
    ODPD_QAT_EVAL(ODPD_GRU, q16::K_GRU) ODPD_QAT_EVAL(ODPD_QGRU, q16::K_Q4) ODPD_QAT_EVAL(ODPD_QGRU_AMP1, q16::K_A4)
#undef ODPD_QAT_EVAL
    return ODPD_EUNSUPPORTED;
}

}  // namespace odpd
