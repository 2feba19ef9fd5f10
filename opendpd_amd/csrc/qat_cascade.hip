// qat_cascade.hip — the quantised models on the one-sequence-per-wave engines (odpd_qatseq.h): as the trained DPD of a train_dpd step in
// front of a frozen gru / dgru PA, in ONE launch (the workgroup of gru_cascade.hip: DPD wave + PA wave per frame, LDS hand-off every 32
// steps) — BASELINE config 5 (quantisation-aware QGRU INT8) and the OpenDPDv2 QAT stage (quantised TRes-DeltaGRU) — and their evaluation
// passes.  Compiled with FP contraction off from odpd_qat.h on (the PA wave's GpSeq code comes before it and keeps its FMAs).
#include "odpd_gpseq.h"
#include "odpd_qatseq.h"

namespace odpd {

// DPD = quantisation-aware GRUCell model (QatSeq); NBD unit blocks
template <int MK, bool LUT, int NBD, int PV, int FMP, bool DGP>
__global__ __launch_bounds__(128) void qat_cascade_kernel(CascArgs a) {
    using D = q16::QatSeq<MK, LUT, NBD>;
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


// The whole train step of a quantised model trained ALONE (train_pa --quant; Base_GRUQuantEnv models behind net_train, train_funcs.py:28-48) at
// the reference's batch sizes: ONE frame per single-wave workgroup on the same engines — forward chunks (y into the loss: residual, loss
// term, dL/dy parked in the engine's dL/du buffer, lane = time step), then the backward chunks in reverse, one row of partial gradients
// per workgroup with the loss sum in column P.  Frames are read in place (tensors or resident streams).
template <typename E, bool TRES>
__global__ __launch_bounds__(64) void qat_gp_train_kernel(SeqArgs a, int bits_w, int bits_a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, NC = (T + kCascChunk - 1) / kCascChunk;
    E e;
    if constexpr (TRES) e.setup(smem, smem, a.params, a.H, T, T, bits_w, bits_a, a.thx, a.thh);
    else e.setup(smem, smem, a.params, a.H, T, T, bits_w, bits_a);
    const S16Loss lossc = s16_loss_setup(a.loss_kind == ODPD_LOSS_L2, a.inv_count, true);
    float2* dyb = reinterpret_cast<float2*>(e.dyb);
    float loss_acc = 0.0f;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        const size_t base = a.frame_idx ? (size_t)a.frame_idx[b] * a.frame_stride : (size_t)b * T;
        const float2* xg = reinterpret_cast<const float2*>(a.x) + base;
        const float2* tg = reinterpret_cast<const float2*>(a.target) + base;
        e.fwd_begin();
        for (int k = 0; k < NC; ++k) {
            const int t0 = k * kCascChunk;
            e.fwd_chunk(k, t0, min(kCascChunk, T - t0), xg, [&](int t, float y0, float y1) {
                const float2 tv = tg[t];
                float d0, d1;
                s16_loss(lossc, y0 - tv.x, y1 - tv.y, d0, d1, loss_acc);
                dyb[t] = make_float2(d0, d1);
            });
        }
        e.bwd_begin();
        for (int c = NC - 1; c >= 0; --c) {
            const int t0 = c * kCascChunk;
            if constexpr (TRES) e.bwd_chunk(c, t0, min(kCascChunk, T - t0), xg);
            else e.bwd_chunk(c, t0, min(kCascChunk, T - t0));
        }
    }
    for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o);
    e.write_partials(a.partials + (size_t)blockIdx.x * (e.L.P + kLossCols), loss_acc);
    if constexpr (TRES) e.add_stats(a.stats, a.B);
}


// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
namespace {
struct QSel { int mk, nbd; bool lut, tres; };
QSel qsel(const odpd_model_t* m) {
    QSel s;
    s.tres = m->backbone == ODPD_TRES_DELTAGRU;
    s.mk = s.tres ? q16::K_TRES : m->backbone == ODPD_GRU ? q16::K_GRU : m->backbone == ODPD_QGRU ? q16::K_Q4 : m->backbone == ODPD_DGRU ? q16::K_DGRU : q16::K_A4;
    s.nbd = m->hidden > 16 ? 2 : 1;
    s.lut = m->bits_w <= 8 && m->bits_a <= 8;
    return s;
}
template <int PV, int FMP, bool DGP> using PaEngine = GpSeq<PV == 0 ? 1 : 2, FMP, DGP, false, PV == 1>;
template <typename D> size_t lds_of(int pv, bool dgp, int T, int Pd, int Pp) {
#define ODPD_QAT_PA(PV_, FMP_, DGP_) if (pv == PV_ && dgp == DGP_) return ((size_t)D::region_floats(T, Pd) + PaEngine<PV_, FMP_, DGP_>::region_floats(T, Pp) + 4) * sizeof(float);
    ODPD_QAT_PA(0, FEAT_RAW2, false) ODPD_QAT_PA(1, FEAT_RAW2, false) ODPD_QAT_PA(2, FEAT_RAW2, false)
    ODPD_QAT_PA(0, FEAT_DGRU6, true) ODPD_QAT_PA(1, FEAT_DGRU6, true) ODPD_QAT_PA(2, FEAT_DGRU6, true)
#undef ODPD_QAT_PA
    return 0;
}
template <typename K> int launch_k(hipStream_t st, K k, int grid, size_t lds, const CascArgs& a) {
    if (int e = allow_big_lds(k, lds)) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(128), lds, st, a);
    return (int)hipGetLastError();
}
}  // namespace

namespace {
template <int MK> size_t lds_kind(const QSel& s, int pv, bool dgp, int T, int Pd, int Pp) {
    if (s.nbd == 1) return s.lut ? lds_of<q16::QatSeq<MK, true, 1>>(pv, dgp, T, Pd, Pp) : lds_of<q16::QatSeq<MK, false, 1>>(pv, dgp, T, Pd, Pp);
    return s.lut ? lds_of<q16::QatSeq<MK, true, 2>>(pv, dgp, T, Pd, Pp) : lds_of<q16::QatSeq<MK, false, 2>>(pv, dgp, T, Pd, Pp);
}
}  // namespace
size_t qat_casc_lds_bytes(const odpd_model_t* dpd, int pv, bool dgp, int T, int Pp) {
    const QSel s = qsel(dpd);
    if (s.tres && dpd->hidden > 16) return 0;
    const int Pd = q16::qat_layout(s.mk, dpd->hidden).P;
    if (s.tres) return s.lut ? lds_of<q16::QatDeltaSeq<true>>(pv, dgp, T, Pd, Pp) : lds_of<q16::QatDeltaSeq<false>>(pv, dgp, T, Pd, Pp);
    if (s.mk == q16::K_GRU) return lds_kind<q16::K_GRU>(s, pv, dgp, T, Pd, Pp);
    if (s.mk == q16::K_Q4) return lds_kind<q16::K_Q4>(s, pv, dgp, T, Pd, Pp);
    if (s.mk == q16::K_DGRU) return lds_kind<q16::K_DGRU>(s, pv, dgp, T, Pd, Pp);
    return lds_kind<q16::K_A4>(s, pv, dgp, T, Pd, Pp);
}
int qat_casc_launch(hipStream_t st, const odpd_model_t* dpd, int pv, bool dgp, int grid, const CascArgs& a, int Pp) {
    const QSel s = qsel(dpd);
    const size_t lds = qat_casc_lds_bytes(dpd, pv, dgp, a.T, Pp);
    if (lds == 0 || grid <= 0) return ODPD_EUNSUPPORTED;
#define ODPD_QAT_PA(MK_, LUT_, NBD_, PV_, FMP_, DGP_) if (pv == PV_ && dgp == DGP_) return launch_k(st, qat_cascade_kernel<MK_, LUT_, NBD_, PV_, FMP_, DGP_>, grid, lds, a);
#define ODPD_QAT_ALLPA(MK_, LUT_, NBD_)                                                                                                   \
    ODPD_QAT_PA(MK_, LUT_, NBD_, 0, FEAT_RAW2, false) ODPD_QAT_PA(MK_, LUT_, NBD_, 1, FEAT_RAW2, false) ODPD_QAT_PA(MK_, LUT_, NBD_, 2, FEAT_RAW2, false) \
    ODPD_QAT_PA(MK_, LUT_, NBD_, 0, FEAT_DGRU6, true) ODPD_QAT_PA(MK_, LUT_, NBD_, 1, FEAT_DGRU6, true) ODPD_QAT_PA(MK_, LUT_, NBD_, 2, FEAT_DGRU6, true)
#define ODPD_QAT_KIND(MK_)                                                                           \
    if (!s.tres && s.mk == MK_) {                                                                    \
        if (s.nbd == 1) { if (s.lut) { ODPD_QAT_ALLPA(MK_, true, 1) } else { ODPD_QAT_ALLPA(MK_, false, 1) } } \
        else { if (s.lut) { ODPD_QAT_ALLPA(MK_, true, 2) } else { ODPD_QAT_ALLPA(MK_, false, 2) } }  \
    }
    ODPD_QAT_KIND(q16::K_GRU) ODPD_QAT_KIND(q16::K_Q4) ODPD_QAT_KIND(q16::K_A4) ODPD_QAT_KIND(q16::K_DGRU)
#undef ODPD_QAT_KIND
#undef ODPD_QAT_ALLPA
#undef ODPD_QAT_PA
    if (s.tres) {
#define ODPD_QAT_PA(LUT_, PV_, FMP_, DGP_) if (pv == PV_ && dgp == DGP_) return launch_k(st, qat_delta_cascade_kernel<LUT_, PV_, FMP_, DGP_>, grid, lds, a);
#define ODPD_QAT_ALLPA(LUT_)                                                                                          \
    ODPD_QAT_PA(LUT_, 0, FEAT_RAW2, false) ODPD_QAT_PA(LUT_, 1, FEAT_RAW2, false) ODPD_QAT_PA(LUT_, 2, FEAT_RAW2, false) \
    ODPD_QAT_PA(LUT_, 0, FEAT_DGRU6, true) ODPD_QAT_PA(LUT_, 1, FEAT_DGRU6, true) ODPD_QAT_PA(LUT_, 2, FEAT_DGRU6, true)
        if (s.lut) { ODPD_QAT_ALLPA(true) } else { ODPD_QAT_ALLPA(false) }
#undef ODPD_QAT_ALLPA
#undef ODPD_QAT_PA
    }
    return ODPD_EUNSUPPORTED;
}

// evaluation passes of the quantised models on the one-sequence-per-wave engines: no checkpoints asked for, every sequence on a SIMD of its own
bool qat_uses_gp_eval(const odpd_model_t* m, int B, bool want_ckpt) {
    if (want_ckpt || m->bits_w <= 0 || m->bits_a <= 0 || m->hidden < 1 || m->hidden > (m->backbone == ODPD_TRES_DELTAGRU ? 16 : 32) || tuning().gp_max_batch == 0 || tuning().s16_min_batch == 0) return false;
    if (m->backbone != ODPD_GRU && m->backbone != ODPD_QGRU && m->backbone != ODPD_QGRU_AMP1 && m->backbone != ODPD_TRES_DELTAGRU && m->backbone != ODPD_DGRU)
        return false;
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
        if (m->hidden > 16)                                                                                                        \
            return lut ? qat_eval_launch<q16::QatSeq<MK_, true, 2>, false>(st, m, a, P) : qat_eval_launch<q16::QatSeq<MK_, false, 2>, false>(st, m, a, P); \
        return lut ? qat_eval_launch<q16::QatSeq<MK_, true, 1>, false>(st, m, a, P) : qat_eval_launch<q16::QatSeq<MK_, false, 1>, false>(st, m, a, P); \
    }
    ODPD_QAT_EVAL(ODPD_GRU, q16::K_GRU) ODPD_QAT_EVAL(ODPD_QGRU, q16::K_Q4) ODPD_QAT_EVAL(ODPD_QGRU_AMP1, q16::K_A4) ODPD_QAT_EVAL(ODPD_DGRU, q16::K_DGRU)
#undef ODPD_QAT_EVAL
    return ODPD_EUNSUPPORTED;
}


// the train step of a quantised model trained alone, at the reference's batch sizes: every frame on a SIMD of its own (qat_gp_train_kernel)
namespace {
template <typename E, bool TRES> struct QatEngine { using type = E; static constexpr bool tres = TRES; };
// f(QatEngine<engine, TRES>{}, parameter count) for the engine of a quantised model
template <typename Fn>
int64_t qat_with_engine(const odpd_model_t* m, Fn&& f) {
    const bool lut = m->bits_w <= 8 && m->bits_a <= 8;
    if (m->backbone == ODPD_TRES_DELTAGRU) {
        const int P = q16::qat_layout(q16::K_TRES, m->hidden).P;
        return lut ? f(QatEngine<q16::QatDeltaSeq<true>, true>{}, P) : f(QatEngine<q16::QatDeltaSeq<false>, true>{}, P);
    }
#define ODPD_QAT_ENGINE(BB_, MK_)                                                                                                    \
    if (m->backbone == BB_) {                                                                                                        \
        const int P = q16::qat_layout(MK_, m->hidden).P;                                                                            \
        if (m->hidden > 16)                                                                                                         \
            return lut ? f(QatEngine<q16::QatSeq<MK_, true, 2>, false>{}, P) : f(QatEngine<q16::QatSeq<MK_, false, 2>, false>{}, P); \
        return lut ? f(QatEngine<q16::QatSeq<MK_, true, 1>, false>{}, P) : f(QatEngine<q16::QatSeq<MK_, false, 1>, false>{}, P);     \
    }
    ODPD_QAT_ENGINE(ODPD_GRU, q16::K_GRU) ODPD_QAT_ENGINE(ODPD_QGRU, q16::K_Q4) ODPD_QAT_ENGINE(ODPD_QGRU_AMP1, q16::K_A4) ODPD_QAT_ENGINE(ODPD_DGRU, q16::K_DGRU)
#undef ODPD_QAT_ENGINE
    return ODPD_EUNSUPPORTED;
}
}  // namespace
bool qat_train_uses_gp(const odpd_model_t* m, int B, int T) {
    if (!qat_uses_gp_eval(m, B, false) || (m->flags & (ODPD_FLAG_NEED_DX | ODPD_FLAG_EVAL))) return false;
    const int64_t lds = qat_with_engine(m, [&](auto eng, int P) { return (int64_t)decltype(eng)::type::region_floats(T, P) * (int64_t)sizeof(float); });
    return lds > 0 && lds <= (int64_t)kMaxLds;
}
int qat_gp_train_rows(const odpd_model_t*, int B, int) { return B; }
// a.stats: the sparsity counters of the quantised TRes-DeltaGRU's forward pass (nullable)
int qat_gp_train(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (!qat_train_uses_gp(m, a.B, a.T)) return ODPD_EUNSUPPORTED;
    return (int)qat_with_engine(m, [&](auto eng, int P) {
        using Eng = decltype(eng);
        const size_t lds = (size_t)Eng::type::region_floats(a.T, P) * sizeof(float);
        auto k = qat_gp_train_kernel<typename Eng::type, Eng::tres>;
        if (int e = allow_big_lds(k, lds)) return (int64_t)e;
        hipLaunchKernelGGL(k, dim3(qat_gp_train_rows(m, a.B, a.T)), dim3(64), lds, st, a, (int)m->bits_w, (int)m->bits_a);
        return (int64_t)hipGetLastError();
    });
}

}  // namespace odpd
