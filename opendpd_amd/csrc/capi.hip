// capi.hip — the extern "C" surface declared in include/opendpd_hip.h: argument validation and
// dispatch to the backbone families.
#include <dlfcn.h>

#include <mutex>
#include <set>
#include <utility>

#include "odpd_host.h"
#include "odpd_xchg.h"

using namespace odpd;

namespace {
enum Family { FAM_NONE = 0, FAM_GRU, FAM_LSTM, FAM_DELTA, FAM_JANET, FAM_TCNN, FAM_QAT, FAM_GMP, FAM_RVTDCNN, FAM_DVR, FAM_BOJ, FAM_APN, FAM_MCL, FAM_GRU2 };
inline Family family_of(int bb);
// a quantisation-aware model: bits_w > 0 on one of the backbones the reference's surgery turns into a quantised cell
// (quant/quant_envs.py:114-130, 290-306: gru, dgru, qgru, qgru_amp1 through the GRU swap; deltagru_tcnskip through its op modules)
inline Family family_of(const odpd_model_t* m) {
    // two recurrent layers: a family of its own (gru_layers2.hip: forward / backward) — every entry point written for the one-layer kernels
    // tests `family_of(m) == FAM_...` and so answers ODPD_EUNSUPPORTED for these descriptors instead of misreading their parameter buffer
    if (m->flags & ODPD_FLAG_TWO_LAYERS) return FAM_GRU2;
    if (m->bits_w > 0 && (m->backbone == ODPD_QGRU || m->backbone == ODPD_QGRU_AMP1 || m->backbone == ODPD_GRU || m->backbone == ODPD_DGRU ||
                          m->backbone == ODPD_TRES_DELTAGRU))
        return FAM_QAT;
    return family_of((int)m->backbone);
}
// which QAT kernels serve a batch: the row-rotated ones (qgru_family.hip: qgru / qgru_amp1, hidden <= 16, 4 sequences per wave) while
// the batch leaves the chip mostly empty, the 16-sequences-per-wave ones (qat_s16.hip) otherwise and for every other kind / hidden size
inline bool qat_uses_s16(const odpd_model_t* m, int B) {
    const bool rot_ok = (m->backbone == ODPD_QGRU || m->backbone == ODPD_QGRU_AMP1) && m->hidden <= 16;
    if (!rot_ok) return true;
    long min_batch = tuning().s16_min_batch;
    if (min_batch < 0) min_batch = 16L * 3 * device_cus();
    return B >= min_batch;
}
inline Family family_of(int bb) {
    switch (bb) {
    case ODPD_GRU: case ODPD_DGRU: case ODPD_QGRU: case ODPD_QGRU_AMP1: return FAM_GRU;
    case ODPD_LSTM: case ODPD_VDLSTM: return FAM_LSTM;
    case ODPD_DELTAGRU: case ODPD_TRES_DELTAGRU: case ODPD_DELTAJANET: return FAM_DELTA;
    case ODPD_PGJANET: return FAM_JANET;
    case ODPD_TCNN: case ODPD_NEURALTX: return FAM_TCNN;
    case ODPD_GMP: return FAM_GMP;
    case ODPD_RVTDCNN: return FAM_RVTDCNN;
    case ODPD_DVRJANET: return FAM_DVR;
    case ODPD_BOJANET: return FAM_BOJ;
    case ODPD_APNRRU: return FAM_APN;
    case ODPD_MCLDNN: return FAM_MCL;
    default: return FAM_NONE;
    }
}
inline int feat_dim(int bb) {
    switch (bb) {
    case ODPD_GRU: case ODPD_LSTM: return 2;
    case ODPD_DGRU: case ODPD_DELTAGRU: case ODPD_TRES_DELTAGRU: case ODPD_DELTAJANET: case ODPD_TCNN: return 6;
    case ODPD_QGRU: case ODPD_QGRU_AMP1: case ODPD_VDLSTM: return 4;
    default: return 0;
    }
}
// models served by the lane-per-unit kernels (gru_wide.hip, lstm_wide.hip, vdlstm_wide.hip, delta_wide.hip: 33 .. 64 hidden units; janet_wide.hip: pgjanet 17 .. 32): forward / backward
// only — the fused entry points answer ODPD_EUNSUPPORTED for them and the caller chains forward, loss, backward
inline bool lane_per_unit_model(const odpd_model_t* m) {
    return gru_wide_ok(m) || lstm_wide_ok(m) || vdlstm_wide_ok(m) || delta_wide_ok(m) || pgjanet_wide_ok(m) || deltajanet_wide_ok(m) || pgjanet_q_ok(m);
}
// bits_w > 0 selects a quantised model only where one exists (include/opendpd_hip.h, odpd_model_t::bits_w); on every other backbone
// the descriptor is refused outright — no entry point may answer it with the float kernels on a float parameter layout
inline bool quant_desc_ok(const odpd_model_t* m) {
    if (m->bits_w <= 0 || m->backbone == ODPD_DVRJANET) return true;      // (dvrjanet: bits_w carries num_dvr_units)
    switch (m->backbone) {
    case ODPD_GRU: case ODPD_DGRU: case ODPD_QGRU: case ODPD_QGRU_AMP1: case ODPD_TRES_DELTAGRU:      // the surgery's quantised cells
    case ODPD_LSTM: case ODPD_VDLSTM: case ODPD_DELTAJANET: case ODPD_NEURALTX:                      // float core, INT_Linear heads
    case ODPD_RVTDCNN:                                                                               // INT_Conv2D + INT_Linear layers (rvtdcnn_q.hip)
    case ODPD_PGJANET:                                                                               // six INT_Linear (pgjanet_q.hip)
        return m->bits_a > 0;
    default: return false;
    }
}
inline bool model_ok(const odpd_model_t* m) {
    return m && m->backbone >= 0 && m->backbone < ODPD_BACKBONE_COUNT && m->hidden > 0 && quant_desc_ok(m);
}
inline SeqArgs make_args(const odpd_model_t* m, int B, int T) {
    SeqArgs a{};
    a.B = B; a.T = T; a.H = m->hidden;
    a.ngroups = num_groups(B, rows_per_seq(m->hidden) ? rows_per_seq(m->hidden) : 1);
    a.nck = num_ckpt(T);
    a.thx = m->thx; a.thh = m->thh;
    a.bits_w = m->bits_w; a.bits_a = m->bits_a; a.eval_out = (m->flags & ODPD_FLAG_EVAL) ? 1 : 0;
    return a;
}
}  // namespace

#include <stdlib.h>
#include <string.h>
#include <vector>
odpd::Tuning& odpd::tuning() {
    static Tuning t = [] {
        Tuning v;
        const char* e = getenv("ODPD_S16_MIN_BATCH");
        v.s16_min_batch = e ? atol(e) : -1;   // -1 = built-in crossover
        e = getenv("ODPD_S16_OCCUPANCY");
        v.s16_occupancy = e ? atoi(e) : 0;    // 0 = by batch size
        e = getenv("ODPD_GP_MAX_BATCH");
        v.gp_max_batch = e ? atol(e) : -1;    // -1 = built-in crossover, 0 = never the gate-parallel fused train kernel
        e = getenv("ODPD_CASCADE_ONE_LAUNCH");
        v.cascade_one_launch = e ? atoi(e) : 1;   // 0 = train_dpd steps always as chained launches
        e = getenv("ODPD_XCHG_FUSED");
        v.xchg_fused = e ? atoi(e) : 1;           // 0 = the one-shot exchange as its own launch instead of the optimiser kernel's prologue
        e = getenv("ODPD_S16X");
        v.s16x = e ? atoi(e) : 1;                 // 0 = frozen-PA step of hidden 17 .. 24 on the exact-fp32 kernel (gru_s16n.hip) instead of the bf16x3 one
        e = getenv("ODPD_S16X_TRAIN");
        v.s16x_train = e ? atoi(e) : 1;           // 0 = fused train step of hidden 17 .. 24 on the exact-fp32 kernel (gru_s16n.hip) instead of the bf16x3 one
        e = getenv("ODPD_LSTM_PACK");
        v.lstm_pack = e ? atoi(e) : 1;            // 0 = lstm16_train_kernel without K-packed input slots (hidden <= 13)
        e = getenv("ODPD_QAT_U3");
        v.qat_u3 = e ? atoi(e) : 1;               // 0 = quantisation-aware GRUCell kinds of hidden <= 12 with four unit slots per lane instead of three (qat_s16.hip)
        return v;
    }();
    return t;
}
// ODPD_AUDIT_LDS (odpd_seq.h): one stderr line per distinct (kernel, dynamic LDS bytes)
void odpd::audit_lds(const void* kernel, size_t lds) {
    static const bool on = getenv("ODPD_AUDIT_LDS") != nullptr;
    if (!on) return;
    static std::mutex mu;
    static std::set<std::pair<const void*, size_t>> seen;
    std::lock_guard<std::mutex> lock(mu);
    if (!seen.insert({kernel, lds}).second) return;
    Dl_info info;
    hipFuncAttributes at;
    const bool named = dladdr(kernel, &info) != 0 && info.dli_sname != nullptr;
    const int regs = hipFuncGetAttributes(&at, kernel) == hipSuccess ? at.numRegs : -1;
    fprintf(stderr, "[odpd-lds] %s lds=%zu regs=%d\n", named ? info.dli_sname : "?", lds, regs);
}
static int64_t g_tuning_generation = 0;
extern "C" int odpd_set_tuning(const char* key, int64_t value) {
    if (!key) return ODPD_EINVAL;
    if (!strcmp(key, "s16_min_batch")) { tuning().s16_min_batch = (long)value; ++g_tuning_generation; return 0; }
    if (!strcmp(key, "s16_occupancy")) { tuning().s16_occupancy = (int)value; ++g_tuning_generation; return 0; }
    if (!strcmp(key, "gp_max_batch")) { tuning().gp_max_batch = (long)value; ++g_tuning_generation; return 0; }
    if (!strcmp(key, "cascade_one_launch")) { tuning().cascade_one_launch = (int)value; ++g_tuning_generation; return 0; }
    if (!strcmp(key, "xchg_fused")) { tuning().xchg_fused = (int)value; return 0; }      // (no buffer depends on it)
    if (!strcmp(key, "s16x")) { tuning().s16x = (int)value; ++g_tuning_generation; return 0; }
    if (!strcmp(key, "s16x_train")) { tuning().s16x_train = (int)value; ++g_tuning_generation; return 0; }
    if (!strcmp(key, "lstm_pack")) { tuning().lstm_pack = (int)value; return 0; }      // (same buffers either way)
    if (!strcmp(key, "qat_u3")) { tuning().qat_u3 = (int)value; return 0; }            // (same buffers either way)
    return ODPD_EINVAL;
}
extern "C" int64_t odpd_tuning_generation(void) { return g_tuning_generation; }

// ---- odpd_probe_issue_ns: vector-issue speed of this GPU ---------------------------------------------------------------
__global__ __launch_bounds__(256) void probe_issue_kernel(float* out, int iters, float b, float c) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (float)(threadIdx.x + j) * 1e-6f;
    const float x = (float)threadIdx.x * c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = __builtin_fmaf(x, b, a[j]);          // 8 independent accumulate chains x 8 = 64 v_fmac_f32
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    if (s == 123.456f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;              // (keeps the chains alive; never true for these operands)
}
extern "C" int odpd_probe_issue_ns(void* stream, int iters, double* ns_per_wave_instr) {
    if (iters < 1 || ns_per_wave_instr == nullptr) return ODPD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* out = nullptr;
    const int blocks = device_cus() * 4;             // 4 workgroups of 4 waves per CU = four waves per SIMD
    if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return ODPD_EINVAL;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.0f;
    for (int pass = 0; pass < 2; ++pass) {
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(probe_issue_kernel, dim3(blocks), dim3(256), 0, st, out, iters, 0.999f, 1e-3f);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(out);
    *ns_per_wave_instr = (double)ms * 1e6 / ((double)iters * 64.0 * 4.0);
    return (int)hipGetLastError();
}

extern "C" int odpd_abi_version(void) { return 13; }   // 13: + odpd_probe_issue_ns; 12: + odpd_sweep_* / odpd_train_epoch_sweep / odpd_backbone_fwd_sweep (K runs of one model shape in lockstep: one launch per step carries them all); 11: + ODPD_FLAG_TWO_LAYERS (gru / qgru / qgru_amp1 with two recurrent layers), hidden 33 .. 64 for the GRU family and lstm, bits_w > 0 on lstm / vdlstm (INT_Linear heads); 10: + odpd_xchg_* (one-shot gradient exchange over peer-mapped slots), odpd_clip_optim_step_dp, odpd_comm_kind / _errors; 9: + odpd_cascade_rows, odpd_cascade_fwd_bwd (train_dpd step body in one launch); 8: + odpd_comm_*, odpd_shard_range, odpd_train_epoch_dp (RCCL inside the native step path); 7: quantised gru / dgru / deltagru_tcnskip descriptors (bits_w > 0), QAT hidden <= 32; 3: + odpd_clip_adamw_step_masked, odpd_tuning_generation, backbones 11..13; 4: backbones 14..17; 5: + odpd_framed_train_supported_shape; 6: + odpd_train_epoch_split
extern "C" const char* odpd_built_arch(void) { return "gfx950"; }

extern "C" int64_t odpd_param_count(const odpd_model_t* m) {
    if (!model_ok(m)) return ODPD_EINVAL;
    if (family_of(m) == FAM_GRU2) return gru2_ok(m) ? gru2_param_count(m) : lstm2_ok(m) ? lstm2_param_count(m) : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_QAT) return qat_s16_param_count(m);
    const int64_t H = m->hidden, F = feat_dim(m->backbone);
    switch (m->backbone) {
    case ODPD_GRU: case ODPD_QGRU: case ODPD_QGRU_AMP1: return 3 * H * F + 3 * H * H + 6 * H + 2 * H + 2;
    case ODPD_DGRU: return 3 * H * F + 3 * H * H + 6 * H + 2 * (H + 6) + 2 + H * H + H;
    // bits_w > 0 on lstm / vdlstm: their nn.Linear heads are INT_Linear (+ three scale parameters each; quant_envs.py:40-60)
    case ODPD_LSTM: return 4 * H * F + 4 * H * H + 8 * H + 2 * H + 2 + (m->bits_w > 0 ? 3 : 0);
    case ODPD_VDLSTM: return 4 * H * 4 + 4 * H * H + 8 * H + 2 * (4 * H + 4) + 2 * 8 + 2 + (m->bits_w > 0 ? 9 : 0);
    case ODPD_DELTAGRU: return 3 * H * 6 + 3 * H * H + 6 * H + 2 * H + 2;
    case ODPD_TRES_DELTAGRU: return 3 * H * 6 + 3 * H * H + 2 * H + 18 + 6;
    case ODPD_DELTAJANET: return 2 * H * 6 + 2 * H * H + 4 * H + 2 * H + 2 + (m->bits_w > 0 ? 3 : 0);      // two gates (deltajanet.py:96-111) + fc_out (bits_w > 0: INT_Linear, + three scales)
    case ODPD_TCNN: return 6 * H + H + 4 * 5 * H + 2 * H;
    case ODPD_PGJANET: return m->bits_w > 0 ? (pgjanet_q_ok(m) ? pgjanet_q_param_count(m) : (int64_t)ODPD_EUNSUPPORTED)
                                            : 3 * (H * (H + 1) + H) + 2 * (H * 2 * H + H) + 2 * H + 2;
    case ODPD_NEURALTX: return H <= 64 ? 27 * H + 14 + (m->bits_w > 0 ? 3 : 0) : (int64_t)ODPD_EUNSUPPORTED;   // two 5-tap FIRs, 4->C (+bias), 4 x depthwise k5, C->2, IQ_match 2x2 (bits_w > 0: INT_Linear, + three scales)
    case ODPD_GMP: return H == 11 ? H * (1 + 4 * H) : (int64_t)ODPD_EUNSUPPORTED;   // memory_length 11, degree 5 (models.py:26-28)
    case ODPD_DVRJANET: return dvrjanet_param_count(m);   // K + 7H^2 + 7H + 2, K = bits_w (dvrjanet.py:11-30, 47-52)
    case ODPD_MCLDNN: return mcldnn_param_count(m);       // 190C + 589 (mcldnn.py:21-27)
    case ODPD_APNRRU: return apnrru_param_count(m);       // 343 + 70H (apnrru.py:13-19, 45-53)
    case ODPD_BOJANET: return bojanet_param_count(m);     // 2H^2 + 28H + 194 (bojanet.py:15-26)
    case ODPD_RVTDCNN: return H <= 32 ? (m->bits_w > 0 ? rvtdcnn_q_param_count(m) : 39 * H + 32) : (int64_t)ODPD_EUNSUPPORTED;  // conv 27+3, fc_hid 36H+H, fc_out 2H+2 (rvtdcnn.py:19-33)
    default: return ODPD_EUNSUPPORTED;
    }
}

extern "C" int64_t odpd_ckpt_floats(const odpd_model_t* m, int B, int T) {
    if (!model_ok(m) || B <= 0 || T <= 0) return ODPD_EINVAL;
    if (family_of(m) == FAM_TCNN || family_of(m) == FAM_GMP || family_of(m) == FAM_RVTDCNN) return 0;   // not recurrent: nothing to checkpoint
    if (family_of(m) == FAM_DVR) return dvrjanet_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_BOJ) return bojanet_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_APN) return apnrru_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_MCL) return mcldnn_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_QAT && qat_uses_s16(m, B)) return qat_s16_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_GRU2) return gru2_ok(m) ? gru2_ckpt_floats(m, B, T) : lstm2_ok(m) ? lstm2_ckpt_floats(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_GRU && gru_wide_ok(m)) return gru_wide_ckpt_floats(m, B, T);      // 33 .. 64 units: the per-step records of gru_wide.hip
    if (family_of(m) == FAM_LSTM && lstm_wide_ok(m)) return lstm_wide_ckpt_floats(m, B, T);    // ... of lstm_wide.hip
    if (family_of(m) == FAM_LSTM && vdlstm_wide_ok(m)) return vdlstm_wide_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_DELTA && delta_wide_ok(m)) return delta_wide_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_DELTA && deltajanet_wide_ok(m)) return deltajanet_wide_ckpt_floats(m, B, T);
    if (family_of(m) == FAM_JANET && m->bits_w > 0) return pgjanet_q_ok(m) ? pgjanet_q_ckpt_floats(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_JANET && pgjanet_wide_ok(m)) return pgjanet_wide_ckpt_floats(m, B, T);
    const int R = rows_per_seq(m->hidden);
    if (!R) return ODPD_EUNSUPPORTED;
    switch (family_of(m)) {
    case FAM_GRU:   // [4-sequence group][ckpt][64 lanes], or [16-sequence task][ckpt][64 lanes][4] for the S16 kernels
        if (gru_uses_s16n(m, B)) {      // (this buffer is also the workspace of odpd_frozen_loss_dx: gru_s16x.hip checkpoints every two steps)
            const int64_t own = gru_s16n_ckpt_floats(m, B, T), x = gru_s16x_ok(m) ? gru_s16x_ckpt_floats(m, B, T) : 0;
            return own > x ? own : x;
        }
        return gru_split_uses_s16(m, B) ? (int64_t)((B + 15) / 16) * num_ckpt(T) * 256 : (int64_t)num_groups(B, R) * num_ckpt(T) * 64;
    case FAM_LSTM: return (int64_t)num_groups(B, R) * num_ckpt(T) * 128;   // h and c
    case FAM_DELTA:
        if (delta_uses_s16(m, B)) return delta_s16_ckpt_floats(m, B, T);
        return R == 1 ? (int64_t)num_groups(B, 1) * num_ckpt(T) * 7 * 64 : (int64_t)ODPD_EUNSUPPORTED;
    case FAM_JANET:
        if (janet_uses_s16(m, B)) return janet_s16_ckpt_floats(m, B, T);
        return R == 1 ? (int64_t)num_groups(B, 1) * num_ckpt(T) * 64 : (int64_t)ODPD_EUNSUPPORTED;
    case FAM_QAT: return R == 1 ? (int64_t)num_groups(B, 1) * num_ckpt(T) * 64 : (int64_t)ODPD_EUNSUPPORTED;
    default: return ODPD_EUNSUPPORTED;
    }
}

extern "C" int64_t odpd_partial_rows(const odpd_model_t* m, int B, int T, int fused) {
    if (!model_ok(m) || B <= 0 || T <= 0) return ODPD_EINVAL;
    switch (family_of(m)) {
    case FAM_GRU2: return fused ? (int64_t)ODPD_EUNSUPPORTED : gru2_ok(m) ? (int64_t)gru2_rows(m, B) : lstm2_ok(m) ? (int64_t)lstm2_rows(m, B) : (int64_t)ODPD_EUNSUPPORTED;
    case FAM_GRU:
        if (gru_wide_ok(m)) return fused ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)gru_wide_rows(m, B);
        return gru_family_rows(m, B, fused ? 1 : 0, T);
    case FAM_LSTM:
        if (lstm_wide_ok(m)) return fused ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)lstm_wide_rows(m, B);
        if (vdlstm_wide_ok(m)) return fused ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)vdlstm_wide_rows(m, B);
        if (fused) return lstm_train_uses_s16(m, B) ? (int64_t)lstm_s16_rows(m, B)
                                                    : lstm_train_uses_gp(m, B, T) ? (int64_t)lstm_gp_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return lstm_family_rows(m, B);
    case FAM_DELTA:
        if (delta_wide_ok(m)) return fused ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)delta_wide_rows(m, B);
        if (deltajanet_wide_ok(m)) return fused ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)deltajanet_wide_rows(m, B);
        if (fused) return delta_train_uses_gp(m, B, T) ? (int64_t)delta_gp_train_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return delta_family_rows(m, B, T);
    case FAM_JANET:
        if (m->bits_w > 0) return (fused || !pgjanet_q_ok(m)) ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)pgjanet_q_rows(m, B);
        if (pgjanet_wide_ok(m)) return fused ? (int64_t)ODPD_EUNSUPPORTED : (int64_t)pgjanet_wide_rows(m, B);
        if (fused) return janet_train_uses_gp(m, B, T) ? (int64_t)janet_gp_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return janet_family_rows(m, B);
    case FAM_DVR:
        if (fused) return dvrjanet_train_uses_gp(m, B, T) ? (int64_t)dvrjanet_gp_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return (int64_t)dvrjanet_rows(m, B);
    case FAM_BOJ:
        if (fused) return bojanet_train_uses_gp(m, B, T) ? (int64_t)bojanet_gp_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return (int64_t)bojanet_rows(m, B);
    case FAM_APN:
        if (fused) return apnrru_train_uses_gp(m, B, T) ? (int64_t)apnrru_gp_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return (int64_t)apnrru_rows(m, B);
    case FAM_MCL:
        if (fused) return mcldnn_train_uses_gp(m, B, T) ? (int64_t)mcldnn_gp_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
        return (int64_t)mcldnn_rows(m, B);
    case FAM_TCNN: return fused ? (int64_t)ODPD_EUNSUPPORTED : tcnn_rows(m, B, T);
    case FAM_GMP: return gmp_rows(m, B, T);
    case FAM_RVTDCNN: return fused ? rvtdcnn_train_rows(m, B, T) : rvtdcnn_rows(m, B, T);
    case FAM_QAT:      // fused: forward-with-checkpoints + backward-with-loss launches of the 16-sequences-per-wave kernels (same rows as their backward)
        if (fused && qat_train_uses_gp(m, B, T)) return (int64_t)qat_gp_train_rows(m, B, T);      // (one frame per wave: the reference's batch sizes)
        if (fused) return qat_uses_s16(m, B) ? (int64_t)qat_s16_rows(m, B) : (int64_t)ODPD_EUNSUPPORTED;
        return qat_uses_s16(m, B) ? (int64_t)qat_s16_rows(m, B) : (int64_t)qgru_family_rows(m, B);
    default: return ODPD_EUNSUPPORTED;
    }
}

extern "C" int64_t odpd_train_workspace_floats(const odpd_model_t* m, int B, int T) {
    if (!model_ok(m) || B <= 0 || T <= 0) return ODPD_EINVAL;
    if (lane_per_unit_model(m)) return ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_LSTM)
        return lstm_train_uses_s16(m, B) ? lstm_s16_workspace_floats(m, B, T) : lstm_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_JANET) return janet_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_DELTA) return delta_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;   // (`workspace` then carries the counters)
    if (family_of(m) == FAM_BOJ) return bojanet_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_APN) return apnrru_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_DVR) return dvrjanet_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_MCL) return mcldnn_train_uses_gp(m, B, T) ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_GMP || family_of(m) == FAM_RVTDCNN) return odpd_param_count(m) > 0 ? 0 : (int64_t)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_QAT) {      // (one frame per wave: no scratch — `workspace` then carries the quantised TRes-DeltaGRU's counters)
        if (qat_train_uses_gp(m, B, T)) return 0;
        return qat_uses_s16(m, B) ? qat_s16_ckpt_floats(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
    }
    if (family_of(m) != FAM_GRU) return ODPD_EUNSUPPORTED;
    if (gru_uses_s16n(m, B)) {      // (the larger of the two layouts: the "s16x_train" knob may change between sizing and launch only with a new generation, but both fit)
        const int64_t own = gru_s16n_ckpt_floats(m, B, T), x = gru_s16x_ok(m) ? gru_s16x_ckpt_floats(m, B, T) : 0;
        return own > x ? own : x;
    }
    return gru_train_uses_s16(m, B, T) ? gru_s16_workspace_floats(m, B, T) : 0;
}

extern "C" int odpd_backbone_fwd(void* stream, const odpd_model_t* m, int B, int T, const float* params,
                                 const float* x, float* y, float* ckpt, double* stats) {
    if (!model_ok(m) || !params || !x || !y || B <= 0 || T <= 0) return ODPD_EINVAL;
    SeqArgs a = make_args(m, B, T);
    a.params = params; a.x = x; a.y = y; a.ckpt = ckpt; a.stats = stats;
    switch (family_of(m)) {
    case FAM_GRU2: return lstm2_ok(m) ? lstm2_fwd((hipStream_t)stream, m, a) : gru2_fwd((hipStream_t)stream, m, a);
    case FAM_GRU: return gru_wide_ok(m) ? gru_wide_fwd((hipStream_t)stream, m, a) : gru_family_fwd((hipStream_t)stream, m, a);
    case FAM_LSTM:
        if (vdlstm_wide_ok(m)) return vdlstm_wide_fwd((hipStream_t)stream, m, a);
        return lstm_wide_ok(m) ? lstm_wide_fwd((hipStream_t)stream, m, a) : lstm_family_fwd((hipStream_t)stream, m, a);
    case FAM_DELTA:
        if (deltajanet_wide_ok(m)) return deltajanet_wide_fwd((hipStream_t)stream, m, a);
        return delta_wide_ok(m) ? delta_wide_fwd((hipStream_t)stream, m, a) : delta_family_fwd((hipStream_t)stream, m, a);
    case FAM_JANET:
        if (m->bits_w > 0) return pgjanet_q_fwd((hipStream_t)stream, m, a);      // `--quant`: six INT_Linear (pgjanet_q.hip)
        return pgjanet_wide_ok(m) ? pgjanet_wide_fwd((hipStream_t)stream, m, a) : janet_family_fwd((hipStream_t)stream, m, a);
    case FAM_DVR: return dvrjanet_launch((hipStream_t)stream, m, a, 1);
    case FAM_BOJ: return bojanet_launch((hipStream_t)stream, m, a, 1);
    case FAM_APN: return apnrru_launch((hipStream_t)stream, m, a, 1);
    case FAM_MCL: return mcldnn_launch((hipStream_t)stream, m, a, 1);
    case FAM_TCNN: return tcnn_fwd((hipStream_t)stream, m, a);
    case FAM_GMP: return gmp_fwd((hipStream_t)stream, m, a);
    case FAM_RVTDCNN: return rvtdcnn_fwd((hipStream_t)stream, m, a);
    case FAM_QAT:
        if (qat_uses_gp_eval(m, B, ckpt != nullptr)) return qat_gp_eval((hipStream_t)stream, m, a);
        return qat_uses_s16(m, B) ? qat_s16_launch((hipStream_t)stream, m, a, 1) : qgru_family_fwd((hipStream_t)stream, m, a);
    default: return ODPD_EUNSUPPORTED;
    }
}

extern "C" int odpd_backbone_bwd(void* stream, const odpd_model_t* m, int B, int T, const float* params,
                                 const float* x, const float* dy, const float* ckpt, float* partials, float* dx) {
    if (!model_ok(m) || !params || !x || !dy || B <= 0 || T <= 0 || (!partials && !dx)) return ODPD_EINVAL;
    SeqArgs a = make_args(m, B, T);
    a.params = params; a.x = x; a.dy = dy; a.ckpt = const_cast<float*>(ckpt); a.partials = partials; a.dx = dx;
    switch (family_of(m)) {
    case FAM_GRU2: return lstm2_ok(m) ? lstm2_bwd((hipStream_t)stream, m, a) : gru2_bwd((hipStream_t)stream, m, a);
    case FAM_GRU:
        if (gru_wide_ok(m)) return gru_wide_bwd((hipStream_t)stream, m, a);
        if (!ckpt && a.nck > 1) return ODPD_EINVAL;
        return gru_family_bwd((hipStream_t)stream, m, a);
    case FAM_LSTM:
        if (lstm_wide_ok(m)) return lstm_wide_bwd((hipStream_t)stream, m, a);
        if (vdlstm_wide_ok(m)) return vdlstm_wide_bwd((hipStream_t)stream, m, a);
        if (!ckpt && a.nck > 1) return ODPD_EINVAL;
        return lstm_family_bwd((hipStream_t)stream, m, a);
    case FAM_DELTA:
        if (delta_wide_ok(m)) return delta_wide_bwd((hipStream_t)stream, m, a);
        if (deltajanet_wide_ok(m)) return deltajanet_wide_bwd((hipStream_t)stream, m, a);
        if (!ckpt && a.nck > 1) return ODPD_EINVAL;
        return delta_family_bwd((hipStream_t)stream, m, a);
    case FAM_JANET:
        if (m->bits_w > 0) return pgjanet_q_bwd((hipStream_t)stream, m, a);
        if (pgjanet_wide_ok(m)) return pgjanet_wide_bwd((hipStream_t)stream, m, a);
        if (!ckpt && a.nck > 1) return ODPD_EINVAL;
        return janet_family_bwd((hipStream_t)stream, m, a);
    case FAM_DVR: return dvrjanet_launch((hipStream_t)stream, m, a, 2);
    case FAM_BOJ: return bojanet_launch((hipStream_t)stream, m, a, 2);
    case FAM_APN: return apnrru_launch((hipStream_t)stream, m, a, 2);
    case FAM_MCL: return mcldnn_launch((hipStream_t)stream, m, a, 2);
    case FAM_TCNN: return tcnn_bwd((hipStream_t)stream, m, a);
    case FAM_GMP: return gmp_bwd((hipStream_t)stream, m, a);
    case FAM_RVTDCNN: return rvtdcnn_bwd((hipStream_t)stream, m, a);
    case FAM_QAT:
        if (qat_uses_s16(m, B)) return qat_s16_launch((hipStream_t)stream, m, a, 2);      // (checks its own checkpoint count)
        if (!ckpt && a.nck > 1) return ODPD_EINVAL;
        return qgru_family_bwd((hipStream_t)stream, m, a);
    default: return ODPD_EUNSUPPORTED;
    }
}

extern "C" int odpd_train_fwd_bwd(void* stream, const odpd_model_t* m, int loss_kind, int B, int T, int64_t count,
                                  const float* params, const float* x, const float* target, float* partials,
                                  float* workspace) {
    if (!model_ok(m) || !params || !x || !target || !partials || B <= 0 || T <= 0 || count <= 0) return ODPD_EINVAL;
    if (lane_per_unit_model(m)) return ODPD_EUNSUPPORTED;
    SeqArgs a = make_args(m, B, T);
    a.params = params; a.x = x; a.target = target; a.partials = partials;
    a.inv_count = (float)(1.0 / (double)count); a.loss_kind = loss_kind; a.ckpt = workspace;
    switch (family_of(m)) {
    case FAM_GRU:
        if (gru_uses_s16n(m, B))
            return gru_s16x_train_ok(m) ? gru_s16x_train((hipStream_t)stream, m, a, gru_s16n_rows(m, B)) : gru_s16n_launch((hipStream_t)stream, m, a, 0);
        return gru_train_uses_s16(m, B, T) ? gru_s16_train((hipStream_t)stream, m, a)
                                           : gru_family_train((hipStream_t)stream, m, a);
    case FAM_LSTM:
        return lstm_train_uses_s16(m, B) ? lstm_s16_train((hipStream_t)stream, m, a)
                                         : lstm_train_uses_gp(m, B, T) ? lstm_gp_train((hipStream_t)stream, m, a) : (int)ODPD_EUNSUPPORTED;
    case FAM_JANET: return janet_train_uses_gp(m, B, T) ? janet_gp_train((hipStream_t)stream, m, a) : (int)ODPD_EUNSUPPORTED;
    case FAM_DELTA:      // delta backbones: `workspace` = the four sparsity counters of the step's forward pass (double[4], may be NULL)
        a.stats = reinterpret_cast<double*>(workspace); a.ckpt = nullptr;
        return delta_gp_train((hipStream_t)stream, m, a);
    case FAM_BOJ: return bojanet_train_uses_gp(m, B, T) ? bojanet_gp_train((hipStream_t)stream, m, a) : (int)ODPD_EUNSUPPORTED;
    case FAM_APN: return apnrru_train_uses_gp(m, B, T) ? apnrru_gp_train((hipStream_t)stream, m, a) : (int)ODPD_EUNSUPPORTED;
    case FAM_DVR: return dvrjanet_train_uses_gp(m, B, T) ? dvrjanet_gp_train((hipStream_t)stream, m, a) : (int)ODPD_EUNSUPPORTED;
    case FAM_MCL: return mcldnn_train_uses_gp(m, B, T) ? mcldnn_gp_train((hipStream_t)stream, m, a) : (int)ODPD_EUNSUPPORTED;
    case FAM_GMP: return gmp_train((hipStream_t)stream, m, a);
    case FAM_RVTDCNN: return rvtdcnn_train((hipStream_t)stream, m, a);
    case FAM_QAT:
        if (qat_train_uses_gp(m, B, T)) {
            a.stats = m->backbone == ODPD_TRES_DELTAGRU ? reinterpret_cast<double*>(workspace) : nullptr; a.ckpt = nullptr;
            return qat_gp_train((hipStream_t)stream, m, a);
        }
        return qat_uses_s16(m, B) ? qat_s16_launch((hipStream_t)stream, m, a, 0) : (int)ODPD_EUNSUPPORTED;
    default: return ODPD_EUNSUPPORTED;
    }
}

// Frozen model in front of the loss (the PA of train_dpd): forward + loss + dL/du in ONE launch.  Available for the GRU family
// (row-rotated kernels with LDS-resident BPTT state at small batches, S16 / S16N at large ones); everywhere else
// ODPD_EUNSUPPORTED (rows < 0) and the caller chains odpd_backbone_fwd, odpd_loss_fwd_bwd, odpd_backbone_bwd.
extern "C" int64_t odpd_frozen_loss_rows(const odpd_model_t* m, int B, int T) {
    if (!model_ok(m) || B <= 0 || T <= 0) return ODPD_EINVAL;
    if (family_of(m) != FAM_GRU || lane_per_unit_model(m)) return ODPD_EUNSUPPORTED;
    return gru_family_lossdx_rows(m, B, T);
}
extern "C" int odpd_frozen_loss_dx(void* stream, const odpd_model_t* m, int loss_kind, int B, int T, int64_t count,
                                   const float* params, const float* u, const float* target, float* du, float* loss_rows,
                                   float* workspace) {
    if (!model_ok(m) || !params || !u || !target || !du || !loss_rows || !workspace || B <= 0 || T <= 0 || count <= 0) return ODPD_EINVAL;
    if (family_of(m) != FAM_GRU || lane_per_unit_model(m)) return ODPD_EUNSUPPORTED;
    SeqArgs a = make_args(m, B, T);
    a.params = params; a.x = u; a.target = target; a.dx = du; a.partials = loss_rows;
    a.inv_count = (float)(1.0 / (double)count); a.loss_kind = loss_kind; a.ckpt = workspace;
    return gru_family_lossdx((hipStream_t)stream, m, a);
}

// DPDs of the one-launch cascade step: the float GRU family, the float delta-GRU backbones and the plain LSTM (gru_cascade.hip)
static bool cascade_dpd_family(const odpd_model_t* m) {
    // (family_of answers FAM_GRU2 for ODPD_FLAG_TWO_LAYERS descriptors — the one-launch step only knows one-layer parameter layouts, so a
    // two-layer lstm must not slip in through its backbone id: ADVICE r04)
    return family_of(m) == FAM_GRU || (family_of(m) == FAM_DELTA && m->backbone != ODPD_DELTAJANET) ||
           (family_of(m) == FAM_LSTM && m->backbone == ODPD_LSTM) ||
           family_of(m) == FAM_QAT;      // (every quantised kind: gru, dgru, qgru, qgru_amp1, deltagru_tcnskip)
}
extern "C" int64_t odpd_cascade_rows(const odpd_model_t* dpd, const odpd_model_t* pa, int B, int T) {
    if (!model_ok(dpd) || !model_ok(pa) || B <= 0 || T <= 0) return ODPD_EINVAL;
    if (!cascade_dpd_family(dpd) || family_of(pa) != FAM_GRU) return ODPD_EUNSUPPORTED;
    // a quantised DPD whose module is in eval() (ODPD_FLAG_EVAL: the 16-bit output quantiser of fc_out is active): the one-launch step
    // only carries the train-mode arithmetic, the chained launches honour the flag — so the step goes to them
    if (family_of(dpd) == FAM_QAT && (dpd->flags & ODPD_FLAG_EVAL)) return ODPD_EUNSUPPORTED;
    return gru_cascade_rows(dpd, pa, B, T);
}
extern "C" int odpd_cascade_fwd_bwd(void* stream, const odpd_model_t* dpd, const odpd_model_t* pa, int loss_kind, int B, int T, int64_t count,
                                    const float* dpd_params, const float* pa_params, const float* x, const float* target,
                                    const int64_t* frame_idx, int frame_stride, float* partials, double* dpd_stats) {
    if (!model_ok(dpd) || !model_ok(pa) || !dpd_params || !pa_params || !x || !target || !partials || B <= 0 || T <= 0 || count <= 0)
        return ODPD_EINVAL;
    if (!cascade_dpd_family(dpd) || family_of(pa) != FAM_GRU) return ODPD_EUNSUPPORTED;
    CascArgs a{};
    a.thx = dpd->thx; a.thh = dpd->thh; a.stats = dpd_stats; a.bits_w = dpd->bits_w; a.bits_a = dpd->bits_a;
    a.dpd_params = dpd_params; a.pa_params = pa_params; a.x = x; a.target = target; a.partials = partials;
    a.frame_idx = reinterpret_cast<const long long*>(frame_idx); a.frame_stride = frame_stride;
    a.inv_count = (float)(1.0 / (double)count); a.loss_kind = loss_kind; a.B = B; a.T = T; a.Hd = dpd->hidden; a.Hp = pa->hidden;
    return gru_cascade_train((hipStream_t)stream, dpd, pa, a);
}

namespace {
// backbones whose fused train kernel addresses frames inside resident streams (SeqArgs::frame_idx)
inline bool framed_train_ok(const odpd_model_t* m) {
    return (!lane_per_unit_model(m) && family_of(m) == FAM_GRU) || family_of(m) == FAM_GMP || family_of(m) == FAM_RVTDCNN;
}
// ... for this batch shape: also the one-sequence-per-wave fused kernels of lstm / vdlstm / pgjanet (their large-batch kernels take tensors)
inline bool framed_train_ok_shape(const odpd_model_t* m, int B, int T) {
    if (lane_per_unit_model(m)) return false;
    if (framed_train_ok(m)) return true;
    if (family_of(m) == FAM_QAT) return qat_train_uses_gp(m, B, T) || qat_uses_s16(m, B);
    if (family_of(m) == FAM_LSTM) return !lstm_train_uses_s16(m, B) && lstm_train_uses_gp(m, B, T);
    if (family_of(m) == FAM_JANET) return janet_train_uses_gp(m, B, T);
    if (family_of(m) == FAM_DELTA) return delta_train_uses_gp(m, B, T);
    if (family_of(m) == FAM_BOJ) return bojanet_train_uses_gp(m, B, T);
    if (family_of(m) == FAM_APN) return apnrru_train_uses_gp(m, B, T);
    if (family_of(m) == FAM_DVR) return dvrjanet_train_uses_gp(m, B, T);
    if (family_of(m) == FAM_MCL) return mcldnn_train_uses_gp(m, B, T);
    return false;
}
inline int framed_train_launch(hipStream_t st, const odpd_model_t* m, const SeqArgs& a) {
    if (family_of(m) == FAM_QAT) {
        if (qat_train_uses_gp(m, a.B, a.T)) {      // (the `workspace` pointer of the framed entry points: the counters, as for the delta backbones)
            SeqArgs b = a;
            b.stats = m->backbone == ODPD_TRES_DELTAGRU ? reinterpret_cast<double*>(a.ckpt) : nullptr; b.ckpt = nullptr;
            return qat_gp_train(st, m, b);
        }
        return framed_train_ok_shape(m, a.B, a.T) ? qat_s16_launch(st, m, a, 0) : (int)ODPD_EUNSUPPORTED;
    }
    if (family_of(m) == FAM_GMP) return gmp_train(st, m, a);
    if (family_of(m) == FAM_RVTDCNN) return rvtdcnn_train(st, m, a);
    if (family_of(m) == FAM_LSTM) return framed_train_ok_shape(m, a.B, a.T) ? lstm_gp_train(st, m, a) : (int)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_JANET) return framed_train_ok_shape(m, a.B, a.T) ? janet_gp_train(st, m, a) : (int)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_DELTA) {     // the `workspace` pointer of the framed entry points carries the sparsity counters here
        SeqArgs b = a;
        b.stats = reinterpret_cast<double*>(a.ckpt); b.ckpt = nullptr;
        return delta_gp_train(st, m, b);
    }
    if (family_of(m) == FAM_BOJ) return framed_train_ok_shape(m, a.B, a.T) ? bojanet_gp_train(st, m, a) : (int)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_APN) return framed_train_ok_shape(m, a.B, a.T) ? apnrru_gp_train(st, m, a) : (int)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_DVR) return framed_train_ok_shape(m, a.B, a.T) ? dvrjanet_gp_train(st, m, a) : (int)ODPD_EUNSUPPORTED;
    if (family_of(m) == FAM_MCL) return framed_train_ok_shape(m, a.B, a.T) ? mcldnn_gp_train(st, m, a) : (int)ODPD_EUNSUPPORTED;
    const bool s16n = gru_uses_s16n(m, a.B), s16 = !s16n && gru_train_uses_s16(m, a.B, a.T);
    if ((s16 || s16n) && !a.ckpt) return ODPD_EINVAL;
    if (s16n) return gru_s16x_train_ok(m) ? gru_s16x_train(st, m, a, gru_s16n_rows(m, a.B)) : gru_s16n_launch(st, m, a, 0);
    return s16 ? gru_s16_train(st, m, a) : gru_family_train(st, m, a);
}
// bf16 sample storage is read by the float GRU family's fused train kernels (stage_in / ld_iq); every other path wants fp32 streams
inline bool frames_format_ok(const odpd_model_t* m, const odpd_frames_t* fr) {
    if (fr->sample_format == ODPD_SAMPLES_F32) return true;
    return fr->sample_format == ODPD_SAMPLES_BF16 && family_of(m) == FAM_GRU;
}
}  // namespace
extern "C" int odpd_framed_train_supported(const odpd_model_t* m) { return model_ok(m) && framed_train_ok(m) ? 1 : 0; }
extern "C" int odpd_framed_train_supported_shape(const odpd_model_t* m, int B, int T) {
    return model_ok(m) && B > 0 && T > 0 && framed_train_ok_shape(m, B, T) ? 1 : 0;
}

// odpd_train_fwd_bwd on frames addressed inside resident streams (no materialised (B,T,2) tensors): batch = the B frames
// fr->order[first .. first+B).  The caller finishes the step as usual (reduce, all-reduce when sharded, clip + AdamW).
extern "C" int odpd_train_fwd_bwd_framed(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr,
                                         int64_t first, int B, int64_t count, const float* params, float* partials,
                                         float* workspace) {
    if (!model_ok(m) || !fr || !fr->x_stream || !fr->y_stream || !fr->order || fr->frame_length <= 0 || fr->stride <= 0 ||
        first < 0 || B <= 0 || first + B > fr->n_frames || count <= 0 || !params || !partials)
        return ODPD_EINVAL;
    const int T = fr->frame_length;
    if (!framed_train_ok_shape(m, B, T) || !frames_format_ok(m, fr)) return ODPD_EUNSUPPORTED;
    SeqArgs a = make_args(m, B, T);
    a.params = params; a.x = fr->x_stream; a.target = fr->y_stream; a.partials = partials; a.ckpt = workspace;
    a.frame_idx = (const long long*)(fr->order + first); a.frame_stride = fr->stride; a.frames_bf16 = fr->sample_format == ODPD_SAMPLES_BF16;
    a.inv_count = (float)(1.0 / (double)count); a.loss_kind = loss_kind;
    return framed_train_launch((hipStream_t)stream, m, a);
}

// Native epoch loop (replaces the Python `for batch in loader` of net_train, train_funcs.py:28-48, for a single
// backbone with a fused kernel): every step = fused fwd+loss+bwd on frames addressed inside the resident streams,
// reduction, clip + AdamW; three launches per step issued back to back from C++, no host synchronisation, no
// gather kernels, no Python between steps.  losses_out[i] = mean loss of batch i (device).
// opt_kind < 0: AdamW with the given hyper-parameters; otherwise an enum odpd_optimizer kind with project.py's own
extern "C" void odpd_shard_range(int64_t n, int rank, int world, int64_t* lo, int64_t* hi) {
    const int64_t base = n / world, rem = n % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}
// the tail of every data-parallel step: all-reduce of grad[0 .. P+4) over `comm` (NULL = one process), then clip + optimiser
// (opt_kind < 0: AdamW with the given hyper-parameters, else an enum odpd_optimizer kind).  A one-shot communicator's exchange runs as
// the optimiser kernel's prologue (one launch for both); RCCL's all-reduce is enqueued in front of it.
static int dp_clip_step(hipStream_t st, void* comm, int opt_kind, int64_t P, float* params, float* grad, float* state1, float* state2,
                        int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm, float* norm_out,
                        float* loss_out, float inv_count, const unsigned char* skip) {
    XchgDev xd;
    const XchgDev* xp = nullptr;
    if (comm) {
        if (tuning().xchg_fused && comm_next_xchg(comm, P + kLossCols, &xd)) xp = &xd;
        else if (int rc = comm_allreduce(st, comm, grad, P + kLossCols)) return rc;
    }
    const int rc = opt_kind < 0 ? launch_clip_adamw(st, P, params, grad, state1, state2, step, lr, beta1, beta2, eps, weight_decay, max_norm, norm_out,
                                                    loss_out, inv_count, skip, xp)
                                : launch_clip_optim(st, opt_kind, P, params, grad, state1, state2, step, lr, max_norm, norm_out, loss_out, inv_count, skip, xp);
    if (rc != 0 && xp) comm_xchg_rollback(comm);      // the launch that was to carry the exchange failed: stay in step with the peers
    return rc;
}
extern "C" int odpd_clip_optim_step_dp(void* stream, void* comm, int opt_kind, int64_t P, float* params, float* grad, float* state1, float* state2,
                                       int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                                       float* norm_out, const unsigned char* skip) {
    if (opt_kind > ODPD_OPT_RMSPROP) return ODPD_EINVAL;
    return dp_clip_step((hipStream_t)stream, comm, opt_kind < 0 ? -1 : opt_kind, P, params, grad, state1, state2, step, lr, beta1, beta2, eps,
                        weight_decay, max_norm, norm_out, nullptr, 0.0f, skip);
}
// comm != NULL: every global batch sharded over the communicator's ranks, one all-reduce of grad[0 .. P+4) per step
static int train_epoch_impl(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, float* params, float* grad,
                            float* state1, float* state2, int64_t first_step, int opt_kind, double lr, double beta1, double beta2, double eps,
                            double weight_decay, double max_norm, float* partials, float* workspace, float* losses_out, void* comm = nullptr) {
    if (!model_ok(m) || !fr || !fr->x_stream || !fr->y_stream || !fr->order || fr->n_frames <= 0 || fr->frame_length <= 0 ||
        fr->stride <= 0 || batch <= 0 || !params || !grad || !state1 || !state2 || !partials || !losses_out || first_step <= 0)
        return ODPD_EINVAL;
    const int T = fr->frame_length;
    if (!frames_format_ok(m, fr)) return ODPD_EUNSUPPORTED;
    const int rank = comm ? comm_rank(comm) : 0, world = comm ? comm_world(comm) : 1;
    {   // every (shard of a) batch of the epoch — the full ones and the tail — must have a frame-reading fused kernel
        const int64_t full = fr->n_frames < batch ? fr->n_frames : batch, tail = fr->n_frames % batch;
        for (int64_t gb : {full, tail}) {
            if (!gb) continue;
            int64_t lo, hi;
            odpd_shard_range(gb, rank, world, &lo, &hi);
            if (hi > lo && !framed_train_ok_shape(m, (int)(hi - lo), T)) return ODPD_EUNSUPPORTED;
        }
    }
    const int64_t P = odpd_param_count(m);
    hipStream_t st = (hipStream_t)stream;
    int64_t step = first_step;
    for (int64_t f0 = 0, i = 0; f0 < fr->n_frames; f0 += batch, ++i, ++step) {
        const int64_t GB = (fr->n_frames - f0) < batch ? (fr->n_frames - f0) : batch;      // the global batch
        int64_t lo, hi;
        odpd_shard_range(GB, rank, world, &lo, &hi);
        const int B = (int)(hi - lo);                                                       // this rank's frames of it
        const int64_t count = GB * T * 2;                                                   // loss mean over the GLOBAL batch
        const float inv_count = (float)(1.0 / (double)count);
        int rc;
        if (B > 0) {
            const int64_t rows = odpd_partial_rows(m, B, T, 1);
            if (rows <= 0) return rows < 0 ? (int)rows : ODPD_EUNSUPPORTED;
            SeqArgs a = make_args(m, B, T);
            a.params = params; a.x = fr->x_stream; a.target = fr->y_stream; a.partials = partials; a.ckpt = workspace;
            a.frame_idx = (const long long*)(fr->order + f0 + lo); a.frame_stride = fr->stride;
            a.frames_bf16 = fr->sample_format == ODPD_SAMPLES_BF16;
            a.inv_count = inv_count; a.loss_kind = loss_kind;
            rc = framed_train_launch(st, m, a);
            if (rc) return rc;
            rc = odpd_reduce_partials(stream, rows, P, partials, grad, 0);
            if (rc) return rc;
        } else {      // an empty shard of a short last batch: zeros into the sum
            rc = (int)hipMemsetAsync(grad, 0, (size_t)(P + kLossCols) * sizeof(float), st);
            if (rc) return rc;
        }
        rc = dp_clip_step(st, comm, opt_kind, P, params, grad, state1, state2, step, lr, beta1, beta2, eps, weight_decay, max_norm, nullptr,
                          losses_out + i, inv_count, nullptr);
        if (rc) return rc;
    }
    return 0;
}
// ---- native epoch loop of train_dpd at the reference's batch sizes: the one-launch cascade step (gru_cascade.hip) on frames read in place ----
// comm != NULL: every global batch sharded over the communicator's ranks (as train_epoch_impl), one all-reduce of grad[0 .. P+4) per step
extern "C" int odpd_train_epoch_cascade(void* stream, void* comm, const odpd_model_t* dpd, const odpd_model_t* pa, int loss_kind,
                                        const odpd_frames_t* fr, int batch, int opt_kind, float* dpd_params, const float* pa_params, float* grad,
                                        float* state1, float* state2, int64_t first_step, double lr, double beta1, double beta2, double eps,
                                        double weight_decay, double max_norm, const unsigned char* skip, float* partials, double* dpd_stats,
                                        float* losses_out) {
    if (!model_ok(dpd) || !model_ok(pa) || !fr || !fr->x_stream || !fr->y_stream || !fr->order || fr->n_frames <= 0 || fr->frame_length <= 0 ||
        fr->stride <= 0 || batch <= 0 || !dpd_params || !pa_params || !grad || !state1 || !state2 || !partials || !losses_out || first_step <= 0 ||
        opt_kind > ODPD_OPT_RMSPROP)
        return ODPD_EINVAL;
    if (fr->sample_format != ODPD_SAMPLES_F32) return ODPD_EUNSUPPORTED;
    const int T = fr->frame_length;
    const int rank = comm ? comm_rank(comm) : 0, world = comm ? comm_world(comm) : 1;
    const int64_t full = fr->n_frames < batch ? fr->n_frames : batch, tail = fr->n_frames % batch;
    for (int64_t gb : {full, tail}) {
        if (!gb) continue;
        int64_t lo, hi;
        odpd_shard_range(gb, rank, world, &lo, &hi);
        if (hi > lo && odpd_cascade_rows(dpd, pa, (int)(hi - lo), T) <= 0) return ODPD_EUNSUPPORTED;
    }
    const int64_t P = odpd_param_count(dpd);
    hipStream_t st = (hipStream_t)stream;
    int64_t step = first_step;
    for (int64_t f0 = 0, i = 0; f0 < fr->n_frames; f0 += batch, ++i, ++step) {
        const int64_t GB = (fr->n_frames - f0) < batch ? (fr->n_frames - f0) : batch;      // the global batch
        int64_t lo, hi;
        odpd_shard_range(GB, rank, world, &lo, &hi);
        const int B = (int)(hi - lo);                                                       // this rank's frames of it
        const int64_t count = GB * T * 2;                                                   // loss mean over the GLOBAL batch
        const float inv_count = (float)(1.0 / (double)count);
        int rc;
        if (B > 0) {
            rc = odpd_cascade_fwd_bwd(stream, dpd, pa, loss_kind, B, T, count, dpd_params, pa_params, fr->x_stream, fr->y_stream,
                                      fr->order + f0 + lo, fr->stride, partials, dpd_stats);
            if (rc) return rc;
            rc = odpd_reduce_partials(stream, odpd_cascade_rows(dpd, pa, B, T), P, partials, grad, 0);
            if (rc) return rc;
        } else {      // an empty shard of a short last batch: zeros into the sum
            rc = (int)hipMemsetAsync(grad, 0, (size_t)(P + kLossCols) * sizeof(float), st);
            if (rc) return rc;
        }
        rc = dp_clip_step(st, comm, opt_kind, P, dpd_params, grad, state1, state2, step, lr, beta1, beta2, eps, weight_decay, max_norm, nullptr,
                          losses_out + i, inv_count, skip);
        if (rc) return rc;
    }
    return 0;
}
// ---- native epoch loop for backbones without a fused train kernel at this shape (the split chain of train_funcs.py:33-44) ----------
namespace {
// frames order[f0 .. f0 + B) of the two streams gathered into (B,T,2) tensors (what IQFrameDataset + the DataLoader's collate build)
__global__ void gather_frames_kernel(const float2* __restrict__ xs, const float2* __restrict__ ys, const long long* __restrict__ order,
                                     int stride, int B, int T, float2* __restrict__ xo, float2* __restrict__ yo) {
    const long long n = (long long)B * T;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / T), t = (int)(i % T);
        const long long src = order[b] * stride + t;
        xo[i] = xs[src]; yo[i] = ys[src];
    }
}
// grad[P] = mean loss * count: the loss travels with the gradient (column P) exactly as in the Python-driven step
__global__ void loss_sum_kernel(float* grad_p, const float* loss_mean, float count) { grad_p[0] = loss_mean[0] * count; }
}  // namespace
extern "C" int odpd_train_epoch_split(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, int opt_kind,
                                      float* params, float* grad, float* state1, float* state2, int64_t first_step, double lr, double beta1,
                                      double beta2, double eps, double weight_decay, double max_norm, const unsigned char* skip,
                                      float* xbuf, float* tbuf, float* ybuf, float* dybuf, float* ckpt, float* partials, float* loss_scratch,
                                      double* stats, float* losses_out) {
    if (!model_ok(m) || !fr || !fr->x_stream || !fr->y_stream || !fr->order || fr->n_frames <= 0 || fr->frame_length <= 0 ||
        fr->stride <= 0 || batch <= 0 || !params || !grad || !state1 || !state2 || !xbuf || !tbuf || !ybuf || !dybuf || !partials ||
        !loss_scratch || !losses_out || first_step <= 0 || opt_kind > ODPD_OPT_RMSPROP)
        return ODPD_EINVAL;
    if (fr->sample_format != ODPD_SAMPLES_F32) return ODPD_EUNSUPPORTED;
    const int T = fr->frame_length;
    const int64_t P = odpd_param_count(m);
    if (P <= 0) return P < 0 ? (int)P : ODPD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int64_t step = first_step;
    for (int64_t f0 = 0, i = 0; f0 < fr->n_frames; f0 += batch, ++i, ++step) {
        const int B = (int)((fr->n_frames - f0) < batch ? (fr->n_frames - f0) : batch);
        const int64_t rows = odpd_partial_rows(m, B, T, 0), n = (int64_t)B * T * 2;
        if (rows <= 0) return rows < 0 ? (int)rows : ODPD_EUNSUPPORTED;
        const long long work = (long long)B * T;
        const int grid = (int)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048);
        hipLaunchKernelGGL(gather_frames_kernel, dim3(grid), dim3(256), 0, st, (const float2*)fr->x_stream, (const float2*)fr->y_stream,
                           (const long long*)(fr->order + f0), fr->stride, B, T, (float2*)xbuf, (float2*)tbuf);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
        rc = odpd_backbone_fwd(stream, m, B, T, params, xbuf, ybuf, ckpt, stats);
        if (rc) return rc;
        rc = odpd_loss_fwd_bwd(stream, loss_kind, n, n, ybuf, tbuf, dybuf, loss_scratch);
        if (rc) return rc;
        rc = odpd_backbone_bwd(stream, m, B, T, params, xbuf, dybuf, ckpt, partials, nullptr);
        if (rc) return rc;
        rc = odpd_reduce_partials(stream, rows, P, partials, grad, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(1), 0, st, grad + P, loss_scratch, (float)n);
        const float inv = (float)(1.0 / (double)n);
        rc = opt_kind < 0 ? launch_clip_adamw(st, P, params, grad, state1, state2, step, lr, beta1, beta2, eps, weight_decay, max_norm, nullptr,
                                              losses_out + i, inv, skip)
                          : launch_clip_optim(st, opt_kind, P, params, grad, state1, state2, step, lr, max_norm, nullptr, losses_out + i, inv, skip);
        if (rc) return rc;
    }
    return 0;
}
// ---- lockstep sweeps: K independent runs of ONE model shape (the seeds of bash_scripts/train_all_pa.sh:26-57) advance together — per step
// one fused train launch of K x grid workgroups, one row reduction, one clip + AdamW launch; each run bit-identical to its solo epoch ----
extern "C" int64_t odpd_sweep_scratch_bytes(int K, int64_t n_steps) {
    if (K <= 0 || n_steps < 0) return ODPD_EINVAL;
    return (int64_t)K * (int64_t)sizeof(SweepRun) + (int64_t)K * n_steps * (int64_t)sizeof(float) + 256;
}
static int upload_sweep(hipStream_t st, int K, const odpd_sweep_run_t* runs, double weight_decay, void* scratch, SweepRun** dev_out) {
    std::vector<SweepRun> h((size_t)K);
    for (int k = 0; k < K; ++k) {
        const odpd_sweep_run_t& r = runs[k];
        SweepRun& d = h[(size_t)k];
        d.params = r.params; d.grad = r.grad; d.state1 = r.exp_avg; d.state2 = r.exp_avg_sq; d.partials = r.partials; d.losses = r.losses_out;
        d.y = r.y; d.workspace = r.workspace; d.order = (const long long*)r.order; d.decay = (float)(1.0 - r.lr * weight_decay); d.pad = 0.0f;
    }
    *dev_out = reinterpret_cast<SweepRun*>(scratch);
    return (int)hipMemcpyAsync(scratch, h.data(), (size_t)K * sizeof(SweepRun), hipMemcpyHostToDevice, st);      // (pageable source: staged before the call returns)
}
extern "C" int odpd_sweep_train_supported(const odpd_model_t* m, int B, int T) {
    return model_ok(m) && family_of(m) == FAM_GRU && !lane_per_unit_model(m) && gru_sweep_train_ok(m, B, T) ? 1 : 0;
}
extern "C" int odpd_sweep_fwd_supported(const odpd_model_t* m, int B, int T) {
    return model_ok(m) && family_of(m) == FAM_GRU && !lane_per_unit_model(m) && gru_sweep_eval_ok(m, B, T) ? 1 : 0;
}
extern "C" int odpd_sweep_s16_supported(const odpd_model_t* m) {
    return model_ok(m) && family_of(m) == FAM_GRU && !lane_per_unit_model(m) && gru_s16_sweep_ok(m) ? 1 : 0;
}
extern "C" int64_t odpd_sweep_partial_rows(const odpd_model_t* m, int B, int T, int flags) {
    if (!model_ok(m) || B <= 0 || T <= 0 || family_of(m) != FAM_GRU || lane_per_unit_model(m)) return ODPD_EUNSUPPORTED;
    if (flags & ODPD_SWEEP_S16) return gru_s16_sweep_ok(m) ? gru_s16_rows(m, B) : (int64_t)ODPD_EUNSUPPORTED;
    return gru_sweep_train_ok(m, B, T) ? gru_sweep_train_rows(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
}
extern "C" int64_t odpd_sweep_workspace_floats(const odpd_model_t* m, int B, int T, int flags) {
    if (!model_ok(m) || B <= 0 || T <= 0) return ODPD_EINVAL;
    if (!(flags & ODPD_SWEEP_S16)) return 0;
    return odpd_sweep_s16_supported(m) ? gru_s16_workspace_floats(m, B, T) : (int64_t)ODPD_EUNSUPPORTED;
}
extern "C" int odpd_train_epoch_sweep(void* stream, const odpd_model_t* m, int K, const odpd_sweep_run_t* runs, int loss_kind, const odpd_frames_t* fr,
                                      int batch, int64_t first_step, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                                      int flags, void* scratch) {
    if (!model_ok(m) || K <= 0 || !runs || !fr || !fr->x_stream || !fr->y_stream || fr->n_frames <= 0 || fr->frame_length <= 0 || fr->stride <= 0 ||
        batch <= 0 || first_step <= 0 || !scratch)
        return ODPD_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!runs[k].params || !runs[k].grad || !runs[k].exp_avg || !runs[k].exp_avg_sq || !runs[k].partials || !runs[k].losses_out || !runs[k].order)
            return ODPD_EINVAL;
    const int T = fr->frame_length;
    const int64_t n = fr->n_frames, n_steps = (n + batch - 1) / batch, tail = n - (n_steps - 1) * batch;
    const bool s16 = (flags & ODPD_SWEEP_S16) != 0;
    if (family_of(m) != FAM_GRU || lane_per_unit_model(m) || !frames_format_ok(m, fr)) return ODPD_EUNSUPPORTED;
    if (s16) {
        if (!gru_s16_sweep_ok(m)) return ODPD_EUNSUPPORTED;
        for (int k = 0; k < K; ++k)
            if (!runs[k].workspace) return ODPD_EINVAL;
    } else if (!gru_sweep_train_ok(m, (int)(n < batch ? n : batch), T) || !gru_sweep_train_ok(m, (int)tail, T)) {
        return ODPD_EUNSUPPORTED;
    }
    const int64_t P = odpd_param_count(m);
    hipStream_t st = (hipStream_t)stream;
    SweepRun* dev = nullptr;
    if (int rc = upload_sweep(st, K, runs, weight_decay, scratch, &dev)) return rc;
    // per-step, per-run step sizes lr_k / (1 - beta1^step): double arithmetic rounded once, exactly launch_clip_adamw's
    float* ss_dev = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + (((size_t)K * sizeof(SweepRun) + 255) & ~(size_t)255));
    std::vector<float> ss((size_t)(K * n_steps));
    for (int64_t i = 0; i < n_steps; ++i) {
        const double bc1 = 1.0 - pow(beta1, (double)(first_step + i));
        for (int k = 0; k < K; ++k) ss[(size_t)(i * K + k)] = (float)(runs[k].lr / bc1);
    }
    ODPD_CHECK_HIP(hipMemcpyAsync(ss_dev, ss.data(), ss.size() * sizeof(float), hipMemcpyHostToDevice, st));
    for (int64_t f0 = 0, i = 0; f0 < n; f0 += batch, ++i) {
        const int B = (int)((n - f0) < batch ? (n - f0) : batch);
        SeqArgs a = make_args(m, B, T);
        a.x = fr->x_stream; a.target = fr->y_stream; a.frame_stride = fr->stride; a.frames_bf16 = fr->sample_format == ODPD_SAMPLES_BF16;
        const float inv_count = (float)(1.0 / (double)((int64_t)B * T * 2));
        a.inv_count = inv_count; a.loss_kind = loss_kind;
        if (s16) a.nck = num_ckpt(T);
        if (int rc = s16 ? gru_s16_sweep_train(st, m, a, dev, K, (long long)f0) : gru_sweep_train(st, m, a, dev, K, (long long)f0)) return rc;
        if (int rc = launch_reduce_sweep(st, dev, K, s16 ? gru_s16_rows(m, B) : gru_sweep_train_rows(m, B, T), P)) return rc;
        if (int rc = launch_clip_adamw_sweep(st, dev, K, P, ss_dev + i * K, first_step + i, i, beta1, beta2, eps, max_norm, inv_count)) return rc;
    }
    return 0;
}
extern "C" int odpd_backbone_fwd_sweep(void* stream, const odpd_model_t* m, int K, const odpd_sweep_run_t* runs, int B, int T, const float* x,
                                       void* scratch) {
    if (!model_ok(m) || K <= 0 || !runs || !x || B <= 0 || T <= 0 || !scratch) return ODPD_EINVAL;
    for (int k = 0; k < K; ++k)
        if (!runs[k].params || !runs[k].y) return ODPD_EINVAL;
    if (family_of(m) != FAM_GRU || lane_per_unit_model(m) || !gru_sweep_eval_ok(m, B, T)) return ODPD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    SweepRun* dev = nullptr;
    if (int rc = upload_sweep(st, K, runs, 0.0, scratch, &dev)) return rc;
    SeqArgs a = make_args(m, B, T);
    a.x = x;
    return gru_sweep_eval(st, m, a, dev, K);
}

extern "C" int odpd_train_epoch(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch,
                                float* params, float* grad, float* exp_avg, float* exp_avg_sq, int64_t first_step, double lr,
                                double beta1, double beta2, double eps, double weight_decay, double max_norm,
                                float* partials, float* workspace, float* losses_out) {
    return train_epoch_impl(stream, m, loss_kind, fr, batch, params, grad, exp_avg, exp_avg_sq, first_step, -1, lr, beta1, beta2, eps, weight_decay,
                            max_norm, partials, workspace, losses_out);
}
extern "C" int odpd_train_epoch_dp(void* stream, void* comm, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, int opt_kind,
                                   float* params, float* grad, float* state1, float* state2, int64_t first_step, double lr, double beta1,
                                   double beta2, double eps, double weight_decay, double max_norm, float* partials, float* workspace,
                                   float* losses_out) {
    if (!comm || opt_kind > ODPD_OPT_RMSPROP) return ODPD_EINVAL;
    return train_epoch_impl(stream, m, loss_kind, fr, batch, params, grad, state1, state2, first_step, opt_kind < 0 ? -1 : opt_kind, lr, beta1, beta2,
                            eps, weight_decay, max_norm, partials, workspace, losses_out, comm);
}
extern "C" int odpd_train_epoch_opt(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, int opt_kind,
                                    float* params, float* grad, float* state1, float* state2, int64_t first_step, double lr,
                                    double max_norm, float* partials, float* workspace, float* losses_out) {
    if (opt_kind < ODPD_OPT_ADAMW || opt_kind > ODPD_OPT_RMSPROP) return ODPD_EINVAL;
    return train_epoch_impl(stream, m, loss_kind, fr, batch, params, grad, state1, state2, first_step, opt_kind, lr, 0, 0, 0, 0, max_norm, partials,
                            workspace, losses_out);
}
