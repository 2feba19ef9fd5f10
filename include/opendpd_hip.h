/*
 * opendpd_hip.h — C ABI of the MI355X-native OpenDPD hot path (libopendpd_hip.so).
 *
 * The reference (lab-emi/OpenDPD) has no FFI: its operator boundary is the Python duck type
 * `backbone.forward(x:(B,T,2) fp32, h_0) -> (B,T,2)` behind `models.CoreModel` (models.py:10-160),
 * chained by `CascadedModel.forward` (models.py:173-176) and driven by `net_train` / `net_eval`
 * (modules/train_funcs.py:16-90).  This header is the C boundary a binding for that path would
 * call: plain device pointers + sizes, a HIP stream handle, int return codes.  No torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 unless stated) valid on the stream's device;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous;
 *   - `params` is the backbone's parameters flattened in the reference's `named_parameters()` order
 *     (e.g. DGRU: rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0, fc_out.weight,
 *     fc_out.bias, fc_hid.weight, fc_hid.bias — backbones/dgru.py:22-32); `odpd_param_count` gives P;
 *   - x, y, dy, dx are (B,T,2) row-major, last dim = (I,Q)  (models.py:150-160);
 *   - return 0 on success, a negative ODPD_E* code on bad arguments, a positive hipError_t otherwise.
 */
#ifndef OPENDPD_HIP_H
#define OPENDPD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* backbone ids — the strings of the reference registry `models.CoreModel` (models.py:26-141) */
enum odpd_backbone {
    ODPD_GRU = 0,        /* backbones/gru.py:4-48 */
    ODPD_DGRU = 1,       /* backbones/dgru.py:9-74 */
    ODPD_QGRU = 2,       /* backbones/qgru.py:9-71        (float path; features I,Q,a^2,a^4) */
    ODPD_QGRU_AMP1 = 3,  /* backbones/qgru_amp1.py:9-76   (float path; features I,Q,a,a^3) */
    ODPD_LSTM = 4,       /* backbones/lstm.py:4-48 */
    ODPD_VDLSTM = 5,     /* backbones/vdlstm.py:5-111 */
    ODPD_DELTAGRU = 6,   /* backbones/deltagru.py:10-276 */
    ODPD_TRES_DELTAGRU = 7, /* backbones/deltagru_tcnskip.py:11-304 ('deltagru_tcnskip') */
    ODPD_TCNN = 8,       /* backbones/tcnn.py:5-97 */
    ODPD_PGJANET = 9,    /* backbones/pgjanet.py:5-84 */
    ODPD_GMP = 10,       /* backbones/gmp.py:5-50 (hidden = memory_length; degree 5 as built by models.py:26-28) */
    ODPD_RVTDCNN = 11,   /* backbones/rvtdcnn.py:9-62 (hidden = fc_hid_size, models.py:80-81; window 4, 3 conv channels) */
    ODPD_NEURALTX = 12,  /* backbones/neuraltx.py:5-137 (hidden = hidden_channels; complex 5-tap FIR + the TCNN stack on 4 features) */
    ODPD_DELTAJANET = 13, /* backbones/deltajanet.py:11-274 (two-gate delta cell; the wrapper fixes both thresholds at 0) */
    ODPD_DVRJANET = 14,  /* backbones/dvrjanet.py:5-112 (num_dvr_units rides in odpd_model_t::bits_w) */
    ODPD_BOJANET = 15,   /* backbones/bojanet.py:5-138 (16-tap complex FIR bank, vector demodulator, JANET cell, phase re-rotation; hidden <= 16 — the reference's own forward stops at 18) */
    ODPD_APNRRU = 16,    /* backbones/apnrru.py:5-152 (3-filter FIR bank + raw sample, phase-normalised RRU cell on a complex state; hidden <= 14) */
    ODPD_MCLDNN = 17,    /* backbones/mcldnn.py:9-134 (two conv branches on a 5x5 feature patch merged by a third, LSTM(5C -> 8), two linear layers; hidden = C) */
    ODPD_BACKBONE_COUNT = 18
};

enum odpd_error {
    ODPD_OK = 0,
    ODPD_EINVAL = -1,      /* bad argument (null pointer, B<=0, ...) */
    ODPD_EUNSUPPORTED = -2, /* backbone / hidden size not supported by the kernels */
    ODPD_ECOMM = -3         /* the collective library (RCCL) is missing or returned an error */
};

/* Model descriptor: what `CoreModel.__init__` receives (models.py:11). */
typedef struct odpd_model {
    int32_t backbone; /* enum odpd_backbone */
    int32_t hidden;   /* hidden_size (channels for tcnn; memory_length for gmp): <= 32 — float gru / dgru / qgru / qgru_amp1 / lstm / vdlstm / deltagru /
                         deltagru_tcnskip / deltajanet: <= 64 (33 .. 64: forward / backward only, the fused entry points answer ODPD_EUNSUPPORTED) — (pgjanet: <= 32, its 17 .. 32 likewise;
                         tcnn: <= 64; gmp: 11; rvtdcnn: fc_hid_size), else ODPD_EUNSUPPORTED */
    float thx;        /* delta threshold on inputs  (deltagru*, models.py:11) */
    float thh;        /* delta threshold on hidden state */
    int32_t bits_w;   /* QAT weight bits (0 = float model) — quant/quant_envs.py:145: > 0 on gru, dgru, qgru, qgru_amp1 or deltagru_tcnskip selects the
                         model the reference's surgery makes of it (quant_envs.py:114-130, 290-306: GRU of GRUCells / quantised delta layer,
                         INT_Linear heads; `params` then follows THAT model's named_parameters(), quantiser scales included);
                         > 0 on lstm / vdlstm: the surgery finds only their nn.Linear heads to swap (quant_envs.py:40-60) — float nn.LSTM core,
                         fc_out (vdlstm: fc_lambda_1, fc_lambda_2, fc_out) as INT_Linear, three scale parameters behind each head's weight and bias;
                         > 0 on deltajanet (hidden <= 64): likewise — the cell's gates are nn.Parameter tensors (deltajanet.py:100-113), fc_out becomes
                         INT_Linear, three scales behind fc_out.bias; > 0 on neuraltx: IQ_match (the one nn.Linear; the Conv1d layers are not in the
                         surgery's layer map, quant_envs.py:145-148) becomes a bias-free INT_Linear, three scales behind IQ_match.weight, no
                         output quantiser in either mode (no module is named fc_out);
                         dvrjanet: num_dvr_units (models.py:119);
                         > 0 on rvtdcnn: Conv2d as INT_Conv2D (two scales behind its bias), fc_hid / fc_out as INT_Linear (three scales each);
                         > 0 on pgjanet (hidden <= 32): its six nn.Linear as INT_Linear, three scales behind each layer's bias;
                         > 0 on any other backbone (no quantised kernels: apnrru, bojanet, mcldnn, deltagru; nothing to
                         quantise: gmp, tcnn): every entry point answers ODPD_EINVAL — never the float kernels */
    int32_t bits_a;   /* QAT activation bits */
    int32_t flags;    /* ODPD_FLAG_* */
} odpd_model_t;

/* eval-mode forward of a quantised model: fc_out's 16-bit output quantiser is active only when the reference
 * module is not in training mode (quant/qmodules/quant_layers.py:77-80) */
#define ODPD_FLAG_EVAL 1
/* delta backbones: the caller will ask odpd_backbone_bwd for dL/dx (frozen PA of a cascade, x.requires_grad): selects the kernels
 * that provide it for forward, checkpoint sizing and backward alike; backward then also needs a `partials` buffer */
#define ODPD_FLAG_NEED_DX 2
/* two stacked recurrent layers (nn.GRU / nn.LSTM num_layers = 2: backbones/gru.py:17-21, lstm.py:17-21, `--PA_num_layers 2`): gru / dgru /
 * qgru / qgru_amp1 / lstm of <= 32 hidden units;
 * `params` follows named_parameters() of the two-layer module (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0, weight_ih_l1 (3H x H),
 * weight_hh_l1, bias_ih_l1, bias_hh_l1, fc_out.weight, fc_out.bias[, dgru: fc_hid.weight, fc_hid.bias]).  Forward / backward only (odpd_backbone_fwd / _bwd); every fused entry point
 * answers ODPD_EUNSUPPORTED and the caller chains forward, loss, backward. */
#define ODPD_FLAG_TWO_LAYERS 4

/* loss kinds — project.py:262-272 */
enum odpd_loss { ODPD_LOSS_L2 = 0, ODPD_LOSS_L1 = 1 };

/* ---- sizes ------------------------------------------------------------------------------- */
/* number of parameters P of the backbone (== utils/util.py:8-15 count_net_params), <0 on error */
int64_t odpd_param_count(const odpd_model_t* m);
/* floats of recurrent-state checkpoints `odpd_*_fwd` writes for BPTT (0 for non-recurrent) */
int64_t odpd_ckpt_floats(const odpd_model_t* m, int B, int T);
/* rows of per-workgroup gradient partials that odpd_backbone_bwd (fused = 0) or odpd_train_fwd_bwd
 * (fused = 1) write for a (B,T,2) batch: partials is (rows, P+4).  ODPD_EUNSUPPORTED for fused = 1
 * when the backbone has no fused kernel for this shape (use the split forward/backward calls then). */
int64_t odpd_partial_rows(const odpd_model_t* m, int B, int T, int fused);
/* floats of scratch `odpd_train_fwd_bwd` needs behind `workspace` for a (B,T,2) batch (0 = none: the
 * small-batch kernel keeps its BPTT checkpoints in LDS; the large-batch kernel spills them to HBM). */
int64_t odpd_train_workspace_floats(const odpd_model_t* m, int B, int T);
/* Kernel-selection knobs (not needed for correct results; buffers must be re-sized with the size queries
 * above after a change).  "s16_min_batch": smallest batch served by the 16-sequences-per-wave fused GRU
 * kernel (-1 = built-in crossover, 0 = always when supported); "s16_occupancy": 1 or 2 waves per SIMD (0 = chosen by batch size);
 * "gp_max_batch": largest batch served by the one-sequence-per-wave fused train kernels of the GRU / LSTM families (-1 = built-in: while every
 * sequence gets a SIMD of its own and the frame's BPTT state fits the CU's LDS share; 0 = never, which also keeps the one-sequence-per-wave
 * forward kernels off); "cascade_one_launch": 0 = odpd_cascade_rows answers ODPD_EUNSUPPORTED (train_dpd steps as chained launches),
 * 1 = built-in choice; "xchg_fused": 0 = the one-shot gradient exchange as a launch of its own instead of the optimiser kernel's prologue;
 * "s16x": 0 = the frozen-PA step of 17 .. 24 hidden units on the exact-fp32 kernel instead of the bf16-split one (same results to fp32
 * rounding; CHANGES odpd_ckpt_floats of those models: the two kernels lay their checkpoints out differently); "s16x_train" (r06): 0 = the
 * fused TRAIN step of those models (odpd_train_fwd_bwd / the framed epoch loops at 16-sequences-per-wave batch sizes) on the exact-fp32 kernel
 * instead of the bf16-split one (same results to fp32 rounding; odpd_train_workspace_floats answers the larger of the two layouts); "lstm_pack": 0 = the fused
 * lstm / vdlstm train kernel of <= 13 hidden units without K-packed input slots (same results to fp32 rounding); "qat_u3": 0 = the
 * quantisation-aware GRUCell models (gru / dgru / qgru / qgru_amp1 with ODPD_FLAG_QUANT) of <= 12 hidden units with four unit slots per lane
 * instead of three (identical results on 8-bit grids, summation order of the wider ones unchanged); "xchg_fused", "lstm_pack" and "qat_u3"
 * change no buffer size (and leave odpd_tuning_generation alone); every successful call on one of the others bumps it.
 * Each knob's initial value comes from the environment variable of its name in capitals with the prefix ODPD_ (ODPD_S16_MIN_BATCH,
 * ODPD_QAT_U3, ...).  Two more variables are read: ODPD_XCHG_TIMEOUT_MS (peer wait of the one-shot gradient exchange) and ODPD_AUDIT_LDS
 * (diagnostic: every launch reports its dynamic LDS size on stderr once, tools/occupancy_audit.py). */
int odpd_set_tuning(const char* key, int64_t value);
/* Counter bumped by every successful odpd_set_tuning: buffers sized by the queries above are valid for the generation they were
 * sized in (the row count / workspace layout of a (B,T) shape depends on the knobs). */
int64_t odpd_tuning_generation(void);
/* library/ABI version, and the gfx arch string the code objects were built for */
int odpd_abi_version(void);
const char* odpd_built_arch(void);
/* Speed of THIS GPU as the recurrent kernels see it (no reference counterpart; bench.py reports it beside its timings so that box-to-box
 * differences of the issue-bound kernels can be told from code changes): runs `iters` x 64 v_fmac_f32 (eight independent accumulate chains) per
 * wave at four waves per SIMD on every CU, twice (the first pass warms the clocks), and returns the second pass's nanoseconds per wave
 * instruction per SIMD (MI355X: 1.74 = four cycles at 2.3 GHz; the accumulate reads a scalar operand, which issues at the four-cycle rate of
 * DPP / compare / max instructions — the all-VGPR v_fmac_f32 / v_mul_f32 forms issue in ~1.1 - 1.2 ns, transcendentals in 3.45:
 * profiles/r06/ubench_issue_costs.txt).  Synchronises the stream. */
int odpd_probe_issue_ns(void* stream, int iters, double* ns_per_wave_instr);

/* ---- backbone forward / backward (replaces backbone.forward + autograd BPTT) ------------- */
/* y = backbone(x).  ckpt may be NULL (inference, net_eval train_funcs.py:57-90).
 * stats (nullable, 4 doubles: dx_zeros, dx_numel, dh_zeros, dh_numel) is ACCUMULATED for the
 * delta backbones (deltagru.py:241-247). */
int odpd_backbone_fwd(void* stream, const odpd_model_t* m, int B, int T, const float* params,
                      const float* x, float* y, float* ckpt, double* stats);
/* Given dy = dL/dy: writes per-workgroup partial parameter gradients to `partials`
 * ((odpd_partial_rows, P+4), OVERWRITTEN; nullable when only dx is wanted = frozen PA, models.py:169-171)
 * and, if dx != NULL, dL/dx (B,T,2) OVERWRITTEN (needed when the backbone is the PA of a cascade). */
int odpd_backbone_bwd(void* stream, const odpd_model_t* m, int B, int T, const float* params,
                      const float* x, const float* dy, const float* ckpt, float* partials, float* dx);
/* Deterministic second-stage reduction: grad[p] = sum_rows partials[row][p]  (grad OVERWRITTEN or
 * ACCUMULATED if accumulate != 0).  Columns P..P+3 of `partials` carry loss partial sums; their
 * reduction lands in grad[P..P+3] (so `grad` holds P+4 floats). */
int odpd_reduce_partials(void* stream, int64_t rows, int64_t P, const float* partials, float* grad,
                         int accumulate);

/* ---- loss (replaces nn.MSELoss / nn.L1Loss + its backward, project.py:262-272) ------------ */
/* loss_out[0] = mean over n elements; dy = dLoss/dy (same shape as y, nullable). n = B*T*2.
 * loss_out must point to 1+256 floats: [0] result, [1..256] scratch for per-workgroup sums.
 * `count` is the GLOBAL element count used for the mean (== n on one GPU; sum over ranks when the
 * batch is sharded, so that summing rank gradients gives the global-batch gradient). */
int odpd_loss_fwd_bwd(void* stream, int kind, int64_t n, int64_t count, const float* y,
                      const float* target, float* dy, float* loss_out);

/* ---- fused train step pieces (replaces train_funcs.py:33-44) ------------------------------ */
/* One fused launch: forward + loss + backward for a single backbone on (B,T,2) frames.
 * Writes `partials` ((rows,P+4); column P holds the un-normalised loss partial sum). No y is
 * written; BPTT state stays in LDS or goes to `workspace` (odpd_train_workspace_floats floats, may be
 * NULL when that is 0).  `count` as in odpd_loss_fwd_bwd.
 * Available (odpd_partial_rows(m, B, T, 1) > 0): GRU family at every batch; lstm / vdlstm at the reference's batch sizes (one sequence
 * per wave, hidden <= 16) and at large batches (16 sequences per wave); pgjanet, bojanet, apnrru, dvrjanet and mcldnn at the reference's
 * batch sizes (one frame per workgroup while the frame's state fits a CU's LDS: frames up to ~230 .. 270 samples); gmp; rvtdcnn;
 * the quantised GRU-cell / delta models at large batches.  Delta backbones (deltagru, deltagru_tcnskip, deltajanet; fused at the
 * reference's batch sizes, odpd_partial_rows(.., 1) > 0): `workspace` is not scratch but the four sparsity counters of the step's forward
 * pass, double[4] as odpd_backbone_fwd's `stats` (may be NULL); the same holds for the `workspace` of the framed / epoch entry points. */
int odpd_train_fwd_bwd(void* stream, const odpd_model_t* m, int loss_kind, int B, int T,
                       int64_t count, const float* params, const float* x, const float* target,
                       float* partials, float* workspace);
/* Frames addressed as windows of resident I/Q streams — what IQFrameDataset materialises with np.stack
 * (data_collector.py:239-247): frame f = samples [f*stride, f*stride + frame_length) of the (N,2) fp32 streams. */
/* Frozen model in front of the loss — the PA of train_dpd (models.py:169-176, train_funcs.py:35-39): forward + loss + dL/du in
 * one launch.  `loss_rows` is (rows, 4) with rows = odpd_frozen_loss_rows (column 0 = un-normalised loss partial sum; reduce
 * with odpd_reduce_partials(rows, 0, ...)); `workspace` holds odpd_ckpt_floats(m, B, T) floats.  ODPD_EUNSUPPORTED (rows < 0)
 * where no such kernel serves the model / batch: chain odpd_backbone_fwd, odpd_loss_fwd_bwd, odpd_backbone_bwd instead. */
int64_t odpd_frozen_loss_rows(const odpd_model_t* m, int B, int T);
int odpd_frozen_loss_dx(void* stream, const odpd_model_t* m, int loss_kind, int B, int T, int64_t count,
                        const float* params, const float* u, const float* target, float* du, float* loss_rows,
                        float* workspace);

/* The whole train_dpd step body — y = PA(DPD(x)) with the PA frozen, loss, dL/d(DPD parameters) — in ONE launch, for the reference's own
 * batch sizes (64 .. 256 frames, train_funcs.py:28-48): the DPD and the PA of a frame run as the two waves of a workgroup and hand the
 * frame over through LDS (csrc/gru_cascade.hip).  Served: float gru / dgru / qgru / qgru_amp1 DPD of hidden <= 32 or deltagru /
 * deltagru_tcnskip / lstm DPD of hidden <= 16, or a quantised (bits_w > 0) gru / qgru / qgru_amp1 DPD of hidden <= 32 / deltagru_tcnskip DPD of hidden <= 16, in front of a float gru / dgru PA of hidden <= 32, batches whose frames are all resident at once
 * (odpd_cascade_rows > 0); otherwise ODPD_EUNSUPPORTED: chain odpd_backbone_fwd, odpd_frozen_loss_dx, odpd_backbone_bwd.
 * `partials` is (rows, P_dpd + 4), column P_dpd = un-normalised loss partial sum; `frame_idx` (device, may be NULL) addresses x /
 * target as windows of resident streams: frame b starts at sample frame_idx[b] * frame_stride; `dpd_stats` (may be NULL): the four
 * sparsity counters of a delta DPD's forward passes, as odpd_backbone_fwd's `stats`. */
int64_t odpd_cascade_rows(const odpd_model_t* dpd, const odpd_model_t* pa, int B, int T);
int odpd_cascade_fwd_bwd(void* stream, const odpd_model_t* dpd, const odpd_model_t* pa, int loss_kind, int B, int T, int64_t count,
                         const float* dpd_params, const float* pa_params, const float* x, const float* target,
                         const int64_t* frame_idx, int frame_stride, float* partials, double* dpd_stats);

enum odpd_sample_format {
    ODPD_SAMPLES_F32 = 0,   /* (I, Q) as two fp32: 8 bytes per sample — what the reference's datasets hold (data_collector.py:239-247) */
    ODPD_SAMPLES_BF16 = 1   /* (I, Q) as two bf16 in one 32-bit word (I in the low half): 4 bytes per sample.  BASELINE configs[1]'s "bf16"
                               frame storage: the kernels widen each value exactly and compute in fp32 — results equal the fp32 path run on
                               the bf16-rounded streams; against the unrounded data the inputs carry a relative rounding of 2^-9.
                               Opt-in; served by the GRU-family fused train kernels (ODPD_EUNSUPPORTED elsewhere). */
};
typedef struct odpd_frames {
    const float* x_stream;  /* (N,2) device: model input stream (fp32 pairs, or N 32-bit words when sample_format = ODPD_SAMPLES_BF16) */
    const float* y_stream;  /* (N,2) device: target stream, same format */
    const int64_t* order;   /* device, n_frames frame indices in visiting order (the DataLoader's epoch permutation) */
    int64_t n_frames;
    int32_t frame_length;
    int32_t stride;
    int32_t sample_format;  /* enum odpd_sample_format */
    int32_t reserved;       /* 0 */
} odpd_frames_t;
/* 1 when the model's fused train kernel can address frames inside resident streams (GRU family, GMP), else 0: the two entry
 * points below return ODPD_EUNSUPPORTED for the others (materialise the batch and call odpd_train_fwd_bwd). */
int odpd_framed_train_supported(const odpd_model_t* m);
/* the same question for one batch shape: also 1 for lstm / vdlstm / pgjanet where their one-sequence-per-wave fused kernels serve (B, T) */
int odpd_framed_train_supported_shape(const odpd_model_t* m, int B, int T);
/* odpd_train_fwd_bwd with the batch given as frames order[first .. first+B) of resident streams (no materialised
 * (B,T,2) tensors).  Same outputs; the caller finishes the step (odpd_reduce_partials, all-reduce, odpd_clip_adamw_step). */
int odpd_train_fwd_bwd_framed(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int64_t first,
                              int B, int64_t count, const float* params, float* partials, float* workspace);
/* One training epoch of a single backbone with odpd_framed_train_supported: for every batch of `batch`
 * frames of fr->order (last one may be smaller) the step of train_funcs.py:33-44 — fused fwd+loss+bwd reading the
 * frames straight from the streams, reduction, clip + AdamW with step index first_step + i — issued from C++ with
 * no host synchronisation.  losses_out (device, ceil(n_frames/batch) floats) receives the per-batch mean losses
 * (train_funcs.py:48-50 averages them).  partials / workspace sized for B = batch by odpd_partial_rows(.., 1) /
 * odpd_train_workspace_floats.  Single process only (no all-reduce between reduction and optimiser step). */
int odpd_train_epoch(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch,
                     float* params, float* grad, float* exp_avg, float* exp_avg_sq, int64_t first_step, double lr,
                     double beta1, double beta2, double eps, double weight_decay, double max_norm, float* partials,
                     float* workspace, float* losses_out);
/* ---- lockstep sweeps: K independent runs of ONE model shape advancing together (the seeds x backbones loops of the reference's
 * bash_scripts/train_all_pa.sh:26-57 and train_all_dpd.sh start one process per run; at the reference's batch sizes one run fills 256 of the
 * chip's >= 8 192 wave slots).  Per step ONE fused train launch carries all K runs (run k owns workgroups [k G, (k + 1) G) and sees exactly the
 * launch it would have had alone), one row reduction and one clip + AdamW launch follow: every run is bit-identical to its solo
 * odpd_train_epoch.  All pointers of a run are device pointers; `runs` itself is a HOST array (copied into `scratch`, device memory of
 * odpd_sweep_scratch_bytes(K, steps of the epoch) bytes, which must stay untouched until the stream has drained).
 * Served: the float GRU family where odpd_sweep_train_supported / odpd_sweep_fwd_supported say so (the one-sequence-per-wave kernels of the
 * reference's own batch sizes); otherwise ODPD_EUNSUPPORTED and the caller loops over odpd_train_epoch. */
typedef struct {
    float* params;            /* P floats */
    float* grad;              /* P + 4 */
    float* exp_avg;           /* P */
    float* exp_avg_sq;        /* P */
    float* partials;          /* (odpd_partial_rows(m, batch, T, 1), P + 4) */
    float* losses_out;        /* ceil(n_frames / batch): mean loss of every step */
    float* y;                 /* odpd_backbone_fwd_sweep: (B, T, 2) output of this run (else unused) */
    float* workspace;         /* ODPD_SWEEP_S16: odpd_sweep_workspace_floats floats of BPTT checkpoint scratch of this run (else unused) */
    const int64_t* order;     /* the run's epoch order: n_frames frame indices (fr->order is ignored) */
    double lr;                /* the run's learning rate of this epoch (the plateau schedulers of the runs are independent) */
} odpd_sweep_run_t;
/* flags of odpd_train_epoch_sweep.  Default (0): every run on the kernel its solo step uses at this batch size — the one-sequence-per-wave kernel
 * whose frame state lives in LDS (bit-identical to the solo run; ~110 KB of LDS per frame, so one CU hosts one frame and K runs cost K solo
 * epochs).  ODPD_SWEEP_S16: the 16-sequences-per-wave MFMA kernel whatever the batch size (hidden <= 16) — K x ceil(B / 16) waves fill the
 * chip; every run then equals its solo run with that kernel forced (odpd_set_tuning("s16_min_batch", 0)) bit for bit, and the default
 * solo run to float tolerance (another summation order). */
#define ODPD_SWEEP_S16 1
int64_t odpd_sweep_scratch_bytes(int K, int64_t n_steps);
int odpd_sweep_s16_supported(const odpd_model_t* m);
int64_t odpd_sweep_partial_rows(const odpd_model_t* m, int B, int T, int flags);      /* rows of a run's `partials` */
int64_t odpd_sweep_workspace_floats(const odpd_model_t* m, int B, int T, int flags);  /* floats of a run's `workspace` (0: none needed) */
int odpd_sweep_train_supported(const odpd_model_t* m, int B, int T);
int odpd_sweep_fwd_supported(const odpd_model_t* m, int B, int T);
/* one epoch of K runs: as odpd_train_epoch (AdamW with the given hyper-parameters, clip at max_norm), step index first_step + i */
int odpd_train_epoch_sweep(void* stream, const odpd_model_t* m, int K, const odpd_sweep_run_t* runs, int loss_kind, const odpd_frames_t* fr,
                           int batch, int64_t first_step, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                           int flags, void* scratch);
/* the evaluation pass of K models of one shape on the same (B, T, 2) sequences x (net_eval, train_funcs.py:57-90): runs[k].y = model_k(x) */
int odpd_backbone_fwd_sweep(void* stream, const odpd_model_t* m, int K, const odpd_sweep_run_t* runs, int B, int T, const float* x,
                            void* scratch);

/* clip_grad_norm_(max_norm) (0 = no clipping) + AdamW step over P parameters
 * (torch.optim.AdamW defaults project.py:283: betas .9/.999, eps 1e-8, weight_decay 0.01).
 * grad is scaled in place like clip_grad_norm_ does.  `step` is the 1-based step index.
 * Hyper-parameters are doubles (Python floats in the reference): 1-beta2 etc. are formed in double
 * and rounded to fp32 once, exactly like torch/optim/adam.py does.
 * norm_out (nullable) receives the pre-clip total norm. */
int odpd_clip_adamw_step(void* stream, int64_t P, float* params, float* grad, float* exp_avg,
                         float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2,
                         double eps, double weight_decay, double max_norm, float* norm_out);
/* Same step with a per-parameter skip mask (device, P bytes, nullable = no mask): skip[i] != 0 marks a parameter whose .grad is
 * None in the reference — torch.optim.AdamW leaves it and its state untouched (no weight decay either) and clip_grad_norm_ does
 * not count it.  Used by the QAT models: the 16-bit out_quantizer scales never enter the train-mode graph
 * (quant/qmodules/quant_layers.py:77-80). */
int odpd_clip_adamw_step_masked(void* stream, int64_t P, float* params, float* grad, float* exp_avg,
                                float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2,
                                double eps, double weight_decay, double max_norm, float* norm_out,
                                const unsigned char* skip);

/* The other optimisers of project.py:274-297, fused with clip_grad_norm_ the same way, with the hyper-parameters the reference
 * constructs them with: ADAMW (betas .9/.999, eps 1e-8, weight_decay 0.01 — odpd_clip_adamw_step_masked), ADAM (the same without the
 * decay), SGD (momentum 0.9: state1 = momentum buffer), RMSPROP (alpha 0.99, eps 1e-8: state2 = square average).  Both state
 * buffers are P floats, zero before the first step.  (adabound, the fifth --opt_type choice, imports a package the reference does not
 * ship: unavailable there and here.) */
enum odpd_optimizer { ODPD_OPT_ADAMW = 0, ODPD_OPT_ADAM = 1, ODPD_OPT_SGD = 2, ODPD_OPT_RMSPROP = 3 };
int odpd_clip_optim_step(void* stream, int kind, int64_t P, float* params, float* grad, float* state1, float* state2,
                         int64_t step, double lr, double max_norm, float* norm_out, const unsigned char* skip);
/* odpd_train_epoch stepping with one of these kinds */
int odpd_train_epoch_opt(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, int opt_kind,
                         float* params, float* grad, float* state1, float* state2, int64_t first_step, double lr, double max_norm,
                         float* partials, float* workspace, float* losses_out);
/* A whole train_dpd epoch of one-launch cascade steps from C++ (net_train's loop, train_funcs.py:28-48, for a CascadedModel with a frozen
 * PA): per step odpd_cascade_fwd_bwd on the frames `order[f0 .. f0 + B)` read in place, row reduction, clip + optimiser step
 * (opt_kind < 0: AdamW with the given betas / eps / weight_decay; else ODPD_OPT_*), mean loss of step i -> losses_out[i].
 * ODPD_EUNSUPPORTED unless odpd_cascade_rows > 0 for the epoch's full batch and its tail. `partials`: max over those of
 * (rows, P_dpd + 4).  `comm` (odpd_comm_init; may be NULL = one process): as odpd_train_epoch_dp — every global batch sharded over the
 * communicator's ranks (odpd_shard_range), the loss mean over the GLOBAL batch, one RCCL all-reduce of P_dpd + 4 floats per step.
 * `skip` (may be NULL): one byte per DPD parameter, non-zero = left out of the norm and of the update, as odpd_clip_adamw_step_masked
 * (the quantised models' 16-bit output-quantiser scales, whose .grad is None in the reference). */
int odpd_train_epoch_cascade(void* stream, void* comm, const odpd_model_t* dpd, const odpd_model_t* pa, int loss_kind, const odpd_frames_t* fr,
                             int batch, int opt_kind, float* dpd_params, const float* pa_params, float* grad, float* state1,
                             float* state2, int64_t first_step, double lr, double beta1, double beta2, double eps, double weight_decay,
                             double max_norm, const unsigned char* skip, float* partials, double* dpd_stats, float* losses_out);


/* The epoch loop for a backbone WITHOUT a fused train kernel at this batch shape: per step the frames are gathered into (B,T,2)
 * buffers, then odpd_backbone_fwd, odpd_loss_fwd_bwd, odpd_backbone_bwd, odpd_reduce_partials and the optimiser (opt_kind < 0: AdamW
 * with the given hyper-parameters, else an enum odpd_optimizer kind; `skip` as in odpd_clip_adamw_step_masked, nullable) run back to
 * back from C++ — the same launches as the caller-driven chain, without the caller between them.  Buffers (device): xbuf, tbuf, ybuf,
 * dybuf = batch*T*2 floats; ckpt = odpd_ckpt_floats; partials = odpd_partial_rows(m, B, T, 0) rows (largest of the full and the
 * tail batch); loss_scratch = 1 + 256 floats; stats = the delta backbones' counters (nullable); losses_out[i] = mean loss of batch i. */
int odpd_train_epoch_split(void* stream, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, int opt_kind,
                           float* params, float* grad, float* state1, float* state2, int64_t first_step, double lr, double beta1,
                           double beta2, double eps, double weight_decay, double max_norm, const unsigned char* skip,
                           float* xbuf, float* tbuf, float* ybuf, float* dybuf, float* ckpt, float* partials, float* loss_scratch,
                           double* stats, float* losses_out);

/* ---- data parallel: the batch of frames sharded over the GPUs of a node, one process per GPU -------------------
 * The reference is single-device; this is the boundary a multi-GPU binding of train_funcs.py:16-54 would call.  Every rank holds a
 * replica of the parameters and of the optimiser state, takes the contiguous shard [lo, hi) = shard of each global batch (sizes differ
 * by at most one, odpd_shard_range), normalises its loss gradient by the GLOBAL element count, and ONE in-place all-reduce (sum) of
 * the P + 4 floats behind `grad` (gradient + loss partial sum) over xGMI per optimiser step — RCCL, or the one-shot exchange below — makes every rank apply the same
 * clip + optimiser update.  librccl is loaded on first use (dlopen); ODPD_ECOMM when it is missing or fails. */
/* rank 0: 128 bytes identifying a new communicator; hand them to every rank (any host channel), then all call odpd_comm_init */
int odpd_comm_unique_id(void* id128);
/* collective over `world` processes, each with its GPU current (hipSetDevice): *comm_out = opaque handle.  world == 1 is valid */
int odpd_comm_init(const void* id128, int world, int rank, void** comm_out);
int odpd_comm_destroy(void* comm);
/* in-place sum over the ranks of n fp32 values, enqueued on `stream` (asynchronous like every other call here) */
int odpd_comm_allreduce_sum(void* stream, void* comm, float* buf, int64_t n);
/* ---- the one-shot exchange: a communicator WITHOUT RCCL for the step's ~4 KB vector (one-shot direct / LL over ring) ----
 * Every rank owns slots (2 parities x world rows x 8192 words) that the peers map; an all-reduce = every rank stores its vector as
 * (sequence, fp32) 8-byte words into its row of every peer's slots and sums the rows arriving in its own, in rank order (replicas
 * stay bit-identical).  Transports: shm_name == NULL — uncached device memory shared through hipIpc (peer HBM over xGMI; also two
 * processes on one GPU); shm_name != NULL — one POSIX shared-memory segment of that name, registered with HIP, for GPUs without peer
 * access.  world <= 8, vectors <= 8192 floats per exchange.
 *   odpd_xchg_create : local part.  *handle64_out = 64 bytes to hand to every rank (hipIpc transport; any host channel).
 *   odpd_xchg_connect: after EVERY rank created: `handles` = world x 64 bytes in rank order (ignored for shared memory).
 *   odpd_xchg_unlink : after EVERY rank connected (shared-memory transport): drops the segment's name.
 * The handle then serves odpd_comm_allreduce_sum, odpd_train_epoch_dp, odpd_train_epoch_cascade, odpd_clip_optim_step_dp and
 * odpd_comm_destroy like an RCCL one.  A peer that does not arrive within $ODPD_XCHG_TIMEOUT_MS (600 000: ten minutes, a collective's allowance under torch's NCCL
 * watchdog) poisons the sum with NaN and
 * counts in odpd_comm_errors — a lost rank ends in NaN losses, never in a hung GPU. */
int odpd_xchg_create(int world, int rank, const char* shm_name, void** comm_out, void* handle64_out);
int odpd_xchg_connect(void* comm, const void* handles);
int odpd_xchg_unlink(void* comm);
/* how long an exchange waits for a peer's row before it gives up (ms <= 0: back to $ODPD_XCHG_TIMEOUT_MS / ten minutes); the
 * communicator's builder uses a few seconds for its self-test, where the ranks have just met */
int odpd_comm_set_timeout_ms(void* comm, int64_t ms);
/* 0 = RCCL, 1 = one-shot exchange over hipIpc device memory, 2 = one-shot exchange over host shared memory */
int odpd_comm_kind(void* comm);
/* exchanges of this rank that timed out so far (synchronises the device); 0 for RCCL communicators */
int odpd_comm_errors(void* comm);
/* The tail of a data-parallel step in one call: all-reduce of grad[0 .. P+4) over `comm` (NULL = no collective), then
 * clip_grad_norm_(max_norm) + the optimiser step as odpd_clip_adamw_step_masked (opt_kind < 0: AdamW with these hyper-parameters)
 * or odpd_clip_optim_step (an enum odpd_optimizer kind).  With a one-shot communicator the exchange is the optimiser kernel's
 * prologue — ONE launch for collective + clip + update; with RCCL the all-reduce is enqueued in front of it. */
int odpd_clip_optim_step_dp(void* stream, void* comm, int opt_kind, int64_t P, float* params, float* grad, float* state1, float* state2,
                            int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                            float* norm_out, const unsigned char* skip);
/* contiguous shard [*lo, *hi) of n items for `rank` of `world` (the split of train_funcs.py's batch across ranks) */
void odpd_shard_range(int64_t n, int rank, int world, int64_t* lo, int64_t* hi);
/* odpd_train_epoch / odpd_train_epoch_opt (opt_kind < 0: AdamW with the given hyper-parameters) with every GLOBAL batch of `batch`
 * frames of fr->order sharded over the ranks of `comm`: per step the fused fwd+loss+bwd kernel on this rank's frames (a rank whose
 * shard of a short last batch is empty contributes zeros), the partial-row reduction, the all-reduce of grad[0 .. P+4) on the same
 * stream, the clip + optimiser step — issued back to back from C++, no host synchronisation, no Python between steps.  Every rank
 * passes the same fr->order (same seeded permutation), batch and hyper-parameters and its own replica buffers.  partials /
 * workspace sized for the largest shard (odpd_partial_rows / odpd_train_workspace_floats at B = ceil(batch / world)).
 * losses_out[i] = mean loss of GLOBAL batch i (identical on every rank). */
int odpd_train_epoch_dp(void* stream, void* comm, const odpd_model_t* m, int loss_kind, const odpd_frames_t* fr, int batch, int opt_kind,
                        float* params, float* grad, float* state1, float* state2, int64_t first_step, double lr, double beta1,
                        double beta2, double eps, double weight_decay, double max_norm, float* partials, float* workspace,
                        float* losses_out);

#ifdef __cplusplus
}
#endif
#endif /* OPENDPD_HIP_H */
