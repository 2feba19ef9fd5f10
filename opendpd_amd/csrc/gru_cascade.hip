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

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {
constexpr int kDpdDelta = 100, kDpdTres = 101, kDpdLstm = 102;
constexpr int kDpdQat = 200;       // a quantised DPD (qat_cascade.hip)      // CascCfg::fmd of the delta DPDs (the GRU-family ones carry their feature mode)
struct CascCfg { int fmd, fmp, pv, nbd, Pd, Pp; bool dgd, dgp; const odpd_model_t* qdpd; };
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
    if ((dpd->flags | pa->flags) & ODPD_FLAG_TWO_LAYERS) return false;      // one-layer parameter layouts only (two layers: the chained launches)
    if (!casc_model(pa, c.fmp, c.dgp)) return false;
    const bool delta = dpd->bits_w == 0 && (dpd->backbone == ODPD_DELTAGRU || dpd->backbone == ODPD_TRES_DELTAGRU);
    const bool lstm = dpd->bits_w == 0 && dpd->backbone == ODPD_LSTM;
    const bool qat = dpd->bits_w > 0 && dpd->bits_a > 0 &&
                     (dpd->backbone == ODPD_GRU || dpd->backbone == ODPD_QGRU || dpd->backbone == ODPD_QGRU_AMP1 || dpd->backbone == ODPD_DGRU);
    const bool qtres = dpd->bits_w > 0 && dpd->bits_a > 0 && dpd->backbone == ODPD_TRES_DELTAGRU;
    if (delta) { c.fmd = dpd->backbone == ODPD_TRES_DELTAGRU ? kDpdTres : kDpdDelta; c.dgd = false; }
    else if (lstm) { c.fmd = kDpdLstm; c.dgd = false; }
    else if (qat || qtres) { c.fmd = kDpdQat; c.dgd = false; c.qdpd = dpd; }      // quantised DPDs: kernels and selection live in qat_cascade.hip
    else if (!casc_model(dpd, c.fmd, c.dgd)) return false;
    if (dpd->hidden < 1 || dpd->hidden > ((delta || lstm || qtres) ? 16 : 32) || pa->hidden < 1 || pa->hidden > 32) return false;
    c.nbd = dpd->hidden > 16 ? 2 : 1;
    if (c.fmp != FEAT_RAW2 && c.fmp != FEAT_DGRU6) return false;      // PAs of the reference's scripts: gru, dgru
    c.pv = pa->hidden > 24 ? 2 : pa->hidden > 16 ? 1 : 0;
    c.Pd = delta ? delta_layout(dpd->hidden, c.fmd == kDpdTres).P : lstm ? lstm_layout(dpd->hidden, 0).P : (qat || qtres) ? (int)qat_s16_param_count(dpd)
                                                                     : gru_layout(dpd->hidden, feat_dim(c.fmd), c.dgd).P;
    c.Pp = gru_layout(pa->hidden, feat_dim(c.fmp), c.dgp).P;
    return true;
}
// the DPD engine of a CascCfg::fmd
template <int NBD, int FMD, bool DGD> struct DpdEngine { using type = GpSeq<NBD, FMD, DGD, true>; };
template <> struct DpdEngine<1, kDpdDelta, false> { using type = DeltaSeq<false>; };
template <> struct DpdEngine<1, kDpdTres, false> { using type = DeltaSeq<true>; };
template <> struct DpdEngine<1, kDpdLstm, false> { using type = LstmSeq; };
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
    ODPD_CASC_PA(1, kDpdLstm, false, CALL)

size_t casc_lds_bytes(const CascCfg& c, int T) {
    if (c.fmd == kDpdQat) return qat_casc_lds_bytes(c.qdpd, c.pv, c.fmp == FEAT_DGRU6, T, c.Pp);
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
    if (c.fmd == kDpdQat) return qat_casc_launch(st, dpd, c.pv, c.fmp == FEAT_DGRU6, casc_grid(c, a.B, a.T), a, c.Pp);
#define ODPD_CASC_LAUNCH(NBD_, FMD_, DGD_, PV_, FMP_, DGP_) casc_launch<NBD_, FMD_, DGD_, PV_, FMP_, DGP_>(st, a, c)
    ODPD_CASC_ALL(ODPD_CASC_LAUNCH)
#undef ODPD_CASC_LAUNCH
    return ODPD_EUNSUPPORTED;
}


}  // namespace odpd
