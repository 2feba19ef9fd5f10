// odpd_host.h — host-side helpers shared by the launchers (argument checks, grid sizing, layouts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/opendpd_hip.h"
#include "odpd_device.h"

namespace odpd {

#define ODPD_CHECK_HIP(expr)                         \
    do {                                             \
        hipError_t _e = (expr);                      \
        if (_e != hipSuccess) return (int)_e;        \
    } while (0)

constexpr int kLossCols = 4;  // extra columns of a partials row: [P]=loss partial sum, [P+1..3] reserved

inline int rows_per_seq(int H) { return H <= 16 ? 1 : (H <= 32 ? 2 : 0); }  // R (0 = unsupported)
inline int seqs_per_wave(int R) { return 4 / R; }
inline int num_groups(int B, int R) { int spw = seqs_per_wave(R); return (B + spw - 1) / spw; }
__host__ __device__ inline int num_ckpt_hd(int T) { return (T + kCkptStride - 1) / kCkptStride; }
inline int num_ckpt(int T) { return num_ckpt_hd(T); }

// number of CUs of the current device (cached)
inline int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// Launch shape of a persistent sequence kernel: `ngroups` wave-tasks are spread over workgroups of
// `waves` wavefronts (1,2,4 or 8: as few as keeps every CU busy, so small batches use many CUs with
// one wave each, large batches share the per-block LDS weight tables between 8 waves); the grid is
// capped at `waves_per_cu` resident waves per CU and each wave loops over its share of the tasks.
struct LaunchShape { int grid, waves; };
inline LaunchShape persistent_shape(int ngroups, int waves_per_cu, int max_waves_per_block = kMaxWavesPerBlock) {
    const int cus = device_cus();
    int waves = 1;
    while (waves < max_waves_per_block && waves < waves_per_cu && ngroups > waves * cus) waves *= 2;
    int need = (ngroups + waves - 1) / waves;
    int cap = cus * (waves_per_cu / waves > 0 ? waves_per_cu / waves : 1);
    LaunchShape ls;
    ls.waves = waves;
    ls.grid = need < cap ? need : cap;
    if (ls.grid < 1) ls.grid = 1;
    return ls;
}

// run-time tuning knobs (odpd_set_tuning; initialised from $ODPD_S16_MIN_BATCH / $ODPD_S16_OCCUPANCY)
void audit_lds(const void* kernel, size_t lds);      // $ODPD_AUDIT_LDS: see odpd_seq.h (defined in capi.hip)
struct Tuning { long s16_min_batch; int s16_occupancy; long gp_max_batch; int cascade_one_launch; int xchg_fused; int s16x; int lstm_pack; int s16x_train; int qat_u3; };
Tuning& tuning();

// ---- parameter layouts (flattened named_parameters() order of the reference modules) -------------
struct GruLayout {
    int F, H, dgru;
    int o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_out, o_b_out, o_w_hid, o_b_hid, P;
};
__host__ __device__ inline GruLayout gru_layout(int H, int F, int dgru) {
    GruLayout L;
    L.F = F; L.H = H; L.dgru = dgru;
    int o = 0;
    L.o_w_ih = o; o += 3 * H * F;
    L.o_w_hh = o; o += 3 * H * H;
    L.o_b_ih = o; o += 3 * H;
    L.o_b_hh = o; o += 3 * H;
    L.o_w_out = o; o += 2 * (dgru ? H + 6 : H);
    L.o_b_out = o; o += 2;
    L.o_w_hid = o; o += dgru ? H * H : 0;
    L.o_b_hid = o; o += dgru ? H : 0;
    L.P = o;
    return L;
}

// qh: the nn.Linear heads are INT_Linear (`--quant` on lstm / vdlstm: the surgery swaps only those, quant/quant_envs.py:40-60) — three scale
// parameters (weight_quantizer, act_quantizer, out_quantizer) behind each head's weight and bias, the named_parameters() order
struct LstmLayout {
    int H, F, vd;
    int o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_l1, o_b_l1, o_w_l2, o_b_l2, o_w_out, o_b_out, o_q_l1, o_q_l2, o_q_out, P;
};
__host__ __device__ inline LstmLayout lstm_layout(int H, int vd, int qh = 0) {
    LstmLayout L;
    L.H = H; L.vd = vd; L.F = vd ? 4 : 2;
    L.o_q_l1 = L.o_q_l2 = L.o_q_out = 0;
    int o = 0;
    L.o_w_ih = o; o += 4 * H * L.F;
    L.o_w_hh = o; o += 4 * H * H;
    L.o_b_ih = o; o += 4 * H;
    L.o_b_hh = o; o += 4 * H;
    L.o_w_l1 = L.o_b_l1 = L.o_w_l2 = L.o_b_l2 = 0;
    if (vd) {
        L.o_w_l1 = o; o += 4 * H; L.o_b_l1 = o; o += 4; if (qh) { L.o_q_l1 = o; o += 3; }
        L.o_w_l2 = o; o += 4 * H; L.o_b_l2 = o; o += 4; if (qh) { L.o_q_l2 = o; o += 3; }
        L.o_w_out = o; o += 16; L.o_b_out = o; o += 2;
    } else {
        L.o_w_out = o; o += 2 * H; L.o_b_out = o; o += 2;
    }
    if (qh) { L.o_q_out = o; o += 3; }
    L.P = o;
    return L;
}

struct DeltaLayout { int H, tres, G, o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_out, o_b_out, o_tcn0, o_tcn2, P; };
// gates: 3 (r, z, n: deltagru / TRes-DeltaGRU) or 2 (f, g: deltajanet)
__host__ __device__ inline DeltaLayout delta_layout(int H, int tres, int gates = 3) {
    DeltaLayout L;
    L.H = H; L.tres = tres; L.G = gates;
    int o = 0;
    L.o_w_ih = o; o += gates * H * 6;
    L.o_w_hh = o; o += gates * H * H;
    L.o_b_ih = L.o_b_hh = L.o_b_out = L.o_tcn0 = L.o_tcn2 = 0;
    if (!tres) { L.o_b_ih = o; o += gates * H; L.o_b_hh = o; o += gates * H; }
    L.o_w_out = o; o += 2 * H;
    if (!tres) { L.o_b_out = o; o += 2; }
    else { L.o_tcn0 = o; o += 18; L.o_tcn2 = o; o += 6; }
    L.P = o;
    return L;
}

// kernel arguments shared by the sequence kernels
struct SeqArgs {
    const float* params;
    const float* x;       // (B,T,2)
    float* y;             // (B,T,2) forward output
    float* ckpt;          // BPTT checkpoints (nullable in forward = inference)
    const float* dy;      // (B,T,2)
    float* partials;      // (rows, P + kLossCols), nullable
    float* dx;            // (B,T,2), nullable
    const float* target;  // fused train step
    double* stats;        // delta sparsity counters (nullable)
    // frames addressed as windows of a resident stream (IQFrameDataset, data_collector.py:239-247): when
    // frame_idx != NULL, sequence b of x / target starts at sample frame_idx[b] * frame_stride of the (N,2) stream
    const long long* frame_idx;
    int frame_stride;
    int frames_bf16;      // frame_idx != NULL: the streams hold bf16 (I, Q) pairs, 4 bytes per sample (odpd_frames_t::sample_format)
    float inv_count;      // 1 / global element count (fused loss)
    float thx, thh;
    int loss_kind;
    int B, T, H, ngroups, nck;
    int bits_w, bits_a, eval_out;      // quantised heads (lstm): grid widths; eval_out: ODPD_FLAG_EVAL (fc_out's 16-bit output quantiser)
};

// one run of a lockstep sweep (K independent runs of one model shape advancing together: odpd_train_epoch_sweep, odpd_backbone_fwd_sweep);
// the table lives in device memory, a sweep kernel's workgroup picks its entry by blockIdx.x / G
struct SweepRun {
    float* params;              // flat parameters of the run
    float* grad;                // P + kLossCols
    float* state1;              // AdamW exp_avg
    float* state2;              // AdamW exp_avg_sq
    float* partials;            // (rows, P + kLossCols)
    float* losses;              // per-step mean losses of the epoch
    float* y;                   // forward output of the evaluation sweep
    float* workspace;           // BPTT checkpoint workspace of the run (16-sequences-per-wave sweeps)
    const long long* order;     // the run's epoch order (frame indices)
    float decay;                // AdamW: 1 - lr * weight_decay at the run's current learning rate
    float pad;
};

// comm.hip: communicator of the data-parallel step (one-shot exchange over peer-mapped slots, or RCCL)
struct XchgDev;                                                          // odpd_xchg.h
int comm_allreduce(hipStream_t st, void* comm, float* buf, int64_t n);     // in-place sum of n floats over the ranks, on the stream
int comm_rank(void* comm);
int comm_world(void* comm);
// one-shot communicators: the NEXT exchange as a kernel argument, for a kernel that folds it into its prologue (optim.hip) — false
// for an RCCL communicator or n > kXchgMaxFloats (the caller then enqueues comm_allreduce).  Advances the sequence: call once per step.
bool comm_next_xchg(void* comm, int64_t n, XchgDev* out);
void comm_xchg_rollback(void* comm);      // ... undone: the kernel that was to carry the exchange could not be launched

// family entry points (defined in the family .hip files)
int gru_family_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_family_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_family_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_family_lossdx(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);        // frozen PA: forward + loss + dL/dx in one launch
int gru_family_lossdx_rows(const odpd_model_t* m, int B, int T);
int gru_family_rows(const odpd_model_t* m, int B, int which /*0 bwd, 1 fused*/, int T);
// lockstep sweeps on the one-sequence-per-wave kernels (gru_family.hip): a.params / partials / frame_idx / y come from the table
bool gru_sweep_train_ok(const odpd_model_t* m, int B, int T);
int gru_sweep_train_rows(const odpd_model_t* m, int B, int T);
int gru_sweep_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, const SweepRun* runs, int K, long long first);
bool gru_sweep_eval_ok(const odpd_model_t* m, int B, int T);
int gru_sweep_eval(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, const SweepRun* runs, int K);
// ... on the 16-sequences-per-wave train kernel whatever the batch size (gru_s16.hip, hidden <= 16): the throughput mode of a sweep
bool gru_s16_sweep_ok(const odpd_model_t* m);
int gru_s16_sweep_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, const SweepRun* runs, int K, long long first);
// optim.hip: row reduction and clip + AdamW of K runs in one launch each (step_sizes: device, one per run: lr_k / (1 - beta1^step))
int launch_reduce_sweep(hipStream_t st, const SweepRun* runs, int K, int64_t rows, int64_t P);
int launch_clip_adamw_sweep(hipStream_t st, const SweepRun* runs, int K, int64_t P, const float* step_sizes, int64_t step, int64_t loss_index, double beta1,
                            double beta2, double eps, double max_norm, float inv_count);
// gru_wide.hip: float gru / dgru / qgru / qgru_amp1 of 33 .. 64 hidden units (one sequence per wave, lane = unit; per-step records in `ckpt`)
bool gru_wide_ok(const odpd_model_t* m);
int64_t gru_wide_ckpt_floats(const odpd_model_t* m, int B, int T);
int gru_wide_rows(const odpd_model_t* m, int B);
int gru_wide_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_wide_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// gru_layers2.hip: gru / qgru / qgru_amp1 with two recurrent layers (ODPD_FLAG_TWO_LAYERS), both layers in one wave, time-skewed
bool gru2_ok(const odpd_model_t* m);
int64_t gru2_param_count(const odpd_model_t* m);
int64_t gru2_ckpt_floats(const odpd_model_t* m, int B, int T);
int gru2_rows(const odpd_model_t* m, int B);
int gru2_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru2_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// lstm_layers2.hip: lstm with two recurrent layers (same arrangement)
bool lstm2_ok(const odpd_model_t* m);
int64_t lstm2_param_count(const odpd_model_t* m);
int64_t lstm2_ckpt_floats(const odpd_model_t* m, int B, int T);
int lstm2_rows(const odpd_model_t* m, int B);
int lstm2_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm2_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// lstm_wide.hip: float lstm of 33 .. 64 hidden units (same mapping)
bool lstm_wide_ok(const odpd_model_t* m);
int64_t lstm_wide_ckpt_floats(const odpd_model_t* m, int B, int T);
int lstm_wide_rows(const odpd_model_t* m, int B);
int lstm_wide_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm_wide_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// vdlstm_wide.hip: float vdlstm of 33 .. 64 hidden units (same mapping; window inputs, lambda heads)
bool vdlstm_wide_ok(const odpd_model_t* m);
int64_t vdlstm_wide_ckpt_floats(const odpd_model_t* m, int B, int T);
int vdlstm_wide_rows(const odpd_model_t* m, int B);
int vdlstm_wide_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int vdlstm_wide_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// delta_wide.hip: float deltagru / deltagru_tcnskip of 33 .. 64 hidden units (same mapping around the delta cell)
bool delta_wide_ok(const odpd_model_t* m);
int64_t delta_wide_ckpt_floats(const odpd_model_t* m, int B, int T);
int delta_wide_rows(const odpd_model_t* m, int B);
int delta_wide_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int delta_wide_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// deltajanet_wide.hip: deltajanet of 33 .. 64 hidden units
bool deltajanet_wide_ok(const odpd_model_t* m);
int64_t deltajanet_wide_ckpt_floats(const odpd_model_t* m, int B, int T);
int deltajanet_wide_rows(const odpd_model_t* m, int B);
int deltajanet_wide_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int deltajanet_wide_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// janet_wide.hip: pgjanet of 17 .. 32 hidden units (lane = hidden unit, the wave's halves sharing a unit's gates)
bool pgjanet_wide_ok(const odpd_model_t* m);
int64_t pgjanet_wide_ckpt_floats(const odpd_model_t* m, int B, int T);
int pgjanet_wide_rows(const odpd_model_t* m, int B);
int pgjanet_wide_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int pgjanet_wide_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// train_dpd at the reference's batch sizes as one launch (gru_cascade.hip): DPD wave + frozen-PA wave per frame
struct CascArgs {
    const float* dpd_params;
    const float* pa_params;
    const float* x;            // (B,T,2), or the stream the frames index into
    const float* target;
    float* partials;           // (rows, P_dpd + kLossCols)
    const long long* frame_idx;
    int frame_stride;
    float inv_count;
    float thx, thh;            // delta DPD: its thresholds
    double* stats;             // delta DPD: sparsity counters of the forward passes (nullable)
    int bits_w, bits_a;        // quantised DPD
    int loss_kind, B, T, Hd, Hp;
};
int gru_cascade_rows(const odpd_model_t* dpd, const odpd_model_t* pa, int B, int T);
// quantised DPDs of the one-launch cascade step (qat_cascade.hip): LDS bytes of the workgroup (0 = not served), launch
size_t qat_casc_lds_bytes(const odpd_model_t* dpd, int pv, bool dgp, int T, int Pp);
int qat_casc_launch(hipStream_t s, const odpd_model_t* dpd, int pv, bool dgp, int grid, const CascArgs& a, int Pp);
// evaluation passes of the quantised models, one sequence per wave (qat_cascade.hip)
bool qat_uses_gp_eval(const odpd_model_t* m, int B, bool want_ckpt);
int qat_gp_eval(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
bool qat_train_uses_gp(const odpd_model_t* m, int B, int T);      // train_pa --quant at the reference's batch sizes: one launch, one frame per wave
int qat_gp_train_rows(const odpd_model_t* m, int B, int T);
int qat_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_cascade_train(hipStream_t s, const odpd_model_t* dpd, const odpd_model_t* pa, const CascArgs& a);
// 16-sequences-per-wave fused kernel (gru_s16.hip) and the rule that selects it
bool gru_train_uses_s16(const odpd_model_t* m, int B, int T);
int gru_s16_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_s16_lossdx(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);   // frozen PA: forward + loss + dL/dx, loss rows in partials
int gru_s16_rows(const odpd_model_t* m, int B);
bool gru_split_uses_s16(const odpd_model_t* m, int B);
int gru_s16_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_s16_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_s16_bwd_rows(const odpd_model_t* m, int B);
// hidden 17..32 (gru_s16n.hip): mode 0 fused train (ckpt = workspace), 1 forward, 2 backward
bool gru_uses_s16n(const odpd_model_t* m, int B);
int gru_s16n_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int gru_s16n_rows(const odpd_model_t* m, int B);
int64_t gru_s16n_ckpt_floats(const odpd_model_t* m, int B, int T);
// hidden 17..24, frozen-PA loss step on the bf16 matrix pipe with three-way operand splits (gru_s16x.hip); grid = gru_s16n_rows
bool gru_s16x_ok(const odpd_model_t* m);
int gru_s16x_lossdx(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int grid);
int64_t gru_s16x_ckpt_floats(const odpd_model_t* m, int B, int T);
// ... and the fused train step of those models on the same pipe (r06; "s16x_train" knob), rows = gru_s16n_rows, workspace = gru_s16x_ckpt_floats
bool gru_s16x_train_ok(const odpd_model_t* m);
int gru_s16x_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int grid);
// ... and their split forward / backward (from dL/dy) entry points: one checkpoint layout for the whole family (gru_s16x_ckpt_floats)
int gru_s16x_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gru_s16x_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int rows);
// optim.hip: clip + AdamW launch that also records loss = grad[P] * inv_count into loss_out (nullable)
int launch_clip_adamw(hipStream_t st, int64_t P, float* params, float* grad, float* exp_avg, float* exp_avg_sq, int64_t step,
                      double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm, float* norm_out,
                      float* loss_out, float inv_count, const unsigned char* skip = nullptr, const XchgDev* xchg = nullptr);
// (xchg != NULL: the kernel first all-reduces grad[0 .. P+4) over the ranks through the one-shot exchange)
// the same for the optimiser kinds of enum odpd_optimizer (project.py:274-297's hyper-parameters)
int launch_clip_optim(hipStream_t st, int kind, int64_t P, float* params, float* grad, float* state1, float* state2, int64_t step, double lr,
                      double max_norm, float* norm_out, float* loss_out, float inv_count, const unsigned char* skip = nullptr,
                      const XchgDev* xchg = nullptr);
int64_t gru_s16_workspace_floats(const odpd_model_t* m, int B, int T);
// fused train step of the delta backbones at the reference's batch sizes (delta_family.hip: delta_gp_bwd_kernel<.., FUSED>)
bool delta_train_uses_gp(const odpd_model_t* m, int B, int T);
int delta_gp_train_rows(const odpd_model_t* m, int B, int T);
int delta_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm_family_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm_family_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm_family_rows(const odpd_model_t* m, int B);
// 16-sequences-per-wave fused train kernel of lstm / vdlstm (lstm_s16.hip)
bool lstm_train_uses_s16(const odpd_model_t* m, int B);
// gate-parallel fused train kernel of lstm / vdlstm at the reference's batch sizes (lstm_family.hip)
bool lstm_train_uses_gp(const odpd_model_t* m, int B, int T);
int lstm_gp_rows(const odpd_model_t* m, int B, int T);
int lstm_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm_s16_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int lstm_s16_rows(const odpd_model_t* m, int B);
int64_t lstm_s16_workspace_floats(const odpd_model_t* m, int B, int T);
int delta_family_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int delta_family_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int delta_family_rows(const odpd_model_t* m, int B, int T);
// 16-sequences-per-wave split kernels of the delta backbones (delta_s16.hip): mode 1 forward, 2 backward
bool delta_uses_s16(const odpd_model_t* m, int B);
int delta_s16_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int delta_s16_rows(const odpd_model_t* m, int B);
int64_t delta_s16_ckpt_floats(const odpd_model_t* m, int B, int T);
// parameter layout of PGJANET (pgjanet.py:14-31)
struct JanetLayout { int H, o_wa, o_ba, o_wp1, o_bp1, o_wp2, o_bp2, o_wf, o_bf, o_wg, o_bg, o_wo, o_bo, P; };
__host__ __device__ inline JanetLayout janet_layout(int H) {
    JanetLayout L; L.H = H; int o = 0;
    L.o_wa = o; o += H * (H + 1); L.o_ba = o; o += H;
    L.o_wp1 = o; o += H * (H + 1); L.o_bp1 = o; o += H;
    L.o_wp2 = o; o += H * (H + 1); L.o_bp2 = o; o += H;
    L.o_wf = o; o += 2 * H * H; L.o_bf = o; o += H;
    L.o_wg = o; o += 2 * H * H; L.o_bg = o; o += H;
    L.o_wo = o; o += 2 * H; L.o_bo = o; o += 2;
    L.P = o;
    return L;
}
int janet_family_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int janet_family_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int janet_family_rows(const odpd_model_t* m, int B);
// gate-parallel fused train kernel of PGJANET at the reference's batch sizes (janet_family.hip)
bool janet_train_uses_gp(const odpd_model_t* m, int B, int T);
int janet_gp_rows(const odpd_model_t* m, int B, int T);
int janet_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// 16-sequences-per-wave split kernels of PGJANET (janet_s16.hip): mode 1 forward, 2 backward
bool janet_uses_s16(const odpd_model_t* m, int B);
int janet_s16_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int janet_s16_rows(const odpd_model_t* m, int B);
int64_t janet_s16_ckpt_floats(const odpd_model_t* m, int B, int T);
// dvrjanet_s16.hip (hidden <= 16, num_dvr_units = bits_w <= 8): mode 1 forward, 2 backward
int dvrjanet_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int dvrjanet_rows(const odpd_model_t* m, int B);
int64_t dvrjanet_param_count(const odpd_model_t* m);
int64_t dvrjanet_ckpt_floats(const odpd_model_t* m, int B, int T);
bool dvrjanet_train_uses_gp(const odpd_model_t* m, int B, int T);     // (as bojanet's)
int dvrjanet_gp_rows(const odpd_model_t* m, int B, int T);
int dvrjanet_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// mcldnn.hip (hidden = channels <= 16): mode 1 forward, 2 backward
int mcldnn_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int mcldnn_rows(const odpd_model_t* m, int B);
int64_t mcldnn_param_count(const odpd_model_t* m);
int64_t mcldnn_ckpt_floats(const odpd_model_t* m, int B, int T);
bool mcldnn_train_uses_gp(const odpd_model_t* m, int B, int T);       // (as bojanet's; one sequence per four-wave workgroup)
int mcldnn_gp_rows(const odpd_model_t* m, int B, int T);
int mcldnn_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// apnrru_s16.hip (hidden <= 14): mode 1 forward, 2 backward
int apnrru_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int apnrru_rows(const odpd_model_t* m, int B);
int64_t apnrru_param_count(const odpd_model_t* m);
int64_t apnrru_ckpt_floats(const odpd_model_t* m, int B, int T);
bool apnrru_train_uses_gp(const odpd_model_t* m, int B, int T);       // (as bojanet's)
int apnrru_gp_rows(const odpd_model_t* m, int B, int T);
int apnrru_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// bojanet_s16.hip (hidden <= 16): mode 1 forward, 2 backward
int bojanet_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int bojanet_rows(const odpd_model_t* m, int B);
int64_t bojanet_param_count(const odpd_model_t* m);
int64_t bojanet_ckpt_floats(const odpd_model_t* m, int B, int T);
// ... its gate-parallel fused train kernel (one sequence per wave: the reference's own batch sizes)
bool bojanet_train_uses_gp(const odpd_model_t* m, int B, int T);
int bojanet_gp_rows(const odpd_model_t* m, int B, int T);
int bojanet_gp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int tcnn_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int tcnn_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int tcnn_rows(const odpd_model_t* m, int B, int T);
// gmp.hip (hidden = memory_length 11, degree 5)
int gmp_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gmp_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int gmp_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);   // fused fwd + loss + dL/dW (frames addressable in streams)
int gmp_rows(const odpd_model_t* m, int B, int T);                      // same grid for the split backward and the fused step
// rvtdcnn.hip (hidden = fc_hid_size <= 32)
int rvtdcnn_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int rvtdcnn_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int rvtdcnn_train(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);  // fused fwd + loss + dL/dW (frames addressable in streams)
int rvtdcnn_rows(const odpd_model_t* m, int B, int T);                       // partials rows of the split backward (with or without dL/dx)
int rvtdcnn_train_rows(const odpd_model_t* m, int B, int T);                 // ... of the fused step
int rvtdcnn_rows_for(int B, int T, bool dx);
// the quantised pgjanet (bits_w > 0; csrc/pgjanet_q.hip): forward / backward, hidden <= 32
bool pgjanet_q_ok(const odpd_model_t* m);
int64_t pgjanet_q_param_count(const odpd_model_t* m);
int64_t pgjanet_q_ckpt_floats(const odpd_model_t* m, int B, int T);
int pgjanet_q_rows(const odpd_model_t* m, int B);
int pgjanet_q_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int pgjanet_q_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
// the quantised rvtdcnn (bits_w > 0; csrc/rvtdcnn_q.hip)
bool rvtdcnn_q_ok(const odpd_model_t* m, int T);
int64_t rvtdcnn_q_param_count(const odpd_model_t* m);
int rvtdcnn_q_rows(const odpd_model_t* m, int B, int T);
int rvtdcnn_q_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int rvtdcnn_q_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, bool fused);
int qgru_family_fwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int qgru_family_bwd(hipStream_t s, const odpd_model_t* m, const SeqArgs& a);
int qgru_family_rows(const odpd_model_t* m, int B);
int64_t qgru_param_count(const odpd_model_t* m);
// quantised gru / dgru / qgru / qgru_amp1 / deltagru_tcnskip on the 16-sequences-per-wave mapping, hidden <= 32 (qat_s16.hip): mode 1 forward, 2 backward
bool qat_s16_supported(const odpd_model_t* m);
int qat_s16_launch(hipStream_t s, const odpd_model_t* m, const SeqArgs& a, int mode);
int qat_s16_rows(const odpd_model_t* m, int B);
int64_t qat_s16_param_count(const odpd_model_t* m);
int64_t qat_s16_ckpt_floats(const odpd_model_t* m, int B, int T);

}  // namespace odpd
