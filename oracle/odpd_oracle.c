/*
 * odpd_oracle.c — CPU restatement of the OpenDPD hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: a plain-C, loop-level restatement of what the reference computes
 * on its CPU path (PyTorch ATen ops composed by backbones/*.py and modules/train_funcs.py).  Only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.  The product path
 * (opendpd_amd/) never imports it and fails loudly when the HIP library is missing.
 *
 * Pinning: every function here is checked against golden vectors produced by running the reference
 * itself (oracle/gen_golden.py -> tests/golden/*.npz) in tests/test_oracle_golden.py.
 *
 * Build: see oracle/Makefile.  `-DODPD_REAL=double` builds the same code in fp64 (used by the tests
 * to decide which of two fp32 results is closer to the exact value).
 *
 * Layouts follow include/opendpd_hip.h: x,y,dy,dx are (B,T,2); params are flattened in the
 * reference's named_parameters() order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/opendpd_hip.h"

#ifndef ODPD_REAL
#define ODPD_REAL float
#endif
typedef ODPD_REAL real;

#define MAXH 64
#define MAXF 8

static inline real sigm(real v) { return (real)1 / ((real)1 + (real)exp(-(double)v)); }
static inline real tanhr(real v) { return (real)tanh((double)v); }

/* ------------------------------------------------------------------------------------------ */
/* parameter layout                                                                             */
/* ------------------------------------------------------------------------------------------ */
static int feat_dim(int bb) {
    switch (bb) {
    case ODPD_GRU: case ODPD_LSTM: return 2;
    case ODPD_DGRU: case ODPD_DELTAGRU: case ODPD_TRES_DELTAGRU: case ODPD_TCNN: return 6;
    case ODPD_QGRU: case ODPD_QGRU_AMP1: case ODPD_VDLSTM: return 4;
    default: return 0;
    }
}

int64_t oracle_param_count(const odpd_model_t* m) {
    int64_t H = m->hidden, F = feat_dim(m->backbone);
    switch (m->backbone) {
    case ODPD_GRU: case ODPD_QGRU: case ODPD_QGRU_AMP1:
        return 3 * H * F + 3 * H * H + 6 * H + 2 * H + 2;
    case ODPD_DGRU:
        return 3 * H * F + 3 * H * H + 6 * H + 2 * (H + 6) + 2 + H * H + H;
    case ODPD_LSTM:
        return 4 * H * F + 4 * H * H + 8 * H + 2 * H + 2;
    case ODPD_VDLSTM:
        return 4 * H * 4 + 4 * H * H + 8 * H + 2 * (4 * H + 4) + 2 * 8 + 2;
    case ODPD_DELTAGRU:
        return 3 * H * 6 + 3 * H * H + 6 * H + 2 * H + 2;
    case ODPD_TRES_DELTAGRU:
        return 3 * H * 6 + 3 * H * H + 2 * H + 18 + 6;
    case ODPD_TCNN:
        return 6 * H + H + 4 * 5 * H + 2 * H;
    case ODPD_PGJANET:
        return 3 * (H * (H + 1) + H) + 2 * (H * 2 * H + H) + 2 * H + 2;
    default: return -1;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* I/Q feature extraction (K1) and its backward                                                */
/* ------------------------------------------------------------------------------------------ */
/* dgru.py:61-68 / deltagru.py:61-73 / tcnn.py:84-91: [I,Q,a,a^3,sin=Q/a,cos=I/a]
 * qgru.py:61-66: [I,Q,a^2,a^4];  qgru_amp1.py:63-70: [I,Q,a,a^3];  gru/lstm: [I,Q] */
static void feat_fwd(int bb, real I, real Q, real* f) {
    real a2 = I * I + Q * Q;
    f[0] = I; f[1] = Q;
    switch (bb) {
    case ODPD_DGRU: case ODPD_DELTAGRU: case ODPD_TCNN: {
        real a = (real)sqrt((double)a2);
        f[2] = a; f[3] = a * a * a; f[4] = Q / a; f[5] = I / a;
    } break;
    case ODPD_QGRU: f[2] = a2; f[3] = a2 * a2; break;
    case ODPD_QGRU_AMP1: { real a = (real)sqrt((double)a2); f[2] = a; f[3] = a * a * a; } break;
    default: break;
    }
}
/* given df (dL/dfeat) returns dI,dQ */
static void feat_bwd(int bb, real I, real Q, const real* df, real* dI, real* dQ) {
    real a2 = I * I + Q * Q;
    real gi = df[0], gq = df[1];
    switch (bb) {
    case ODPD_DGRU: case ODPD_DELTAGRU: case ODPD_TCNN: {
        real a = (real)sqrt((double)a2);
        /* sin = Q/a, cos = I/a */
        real da = df[2] + (real)3 * a * a * df[3] - (Q / a2) * df[4] - (I / a2) * df[5];
        gi += df[5] / a + da * I / a;
        gq += df[4] / a + da * Q / a;
    } break;
    case ODPD_QGRU: {
        real da2 = df[2] + (real)2 * a2 * df[3];
        gi += (real)2 * I * da2; gq += (real)2 * Q * da2;
    } break;
    case ODPD_QGRU_AMP1: {
        real a = (real)sqrt((double)a2);
        real da = df[2] + (real)3 * a * a * df[3];
        gi += da * I / a; gq += da * Q / a;
    } break;
    default: break;
    }
    *dI = gi; *dQ = gq;
}

/* ------------------------------------------------------------------------------------------ */
/* GRU family: gru.py, dgru.py, qgru.py, qgru_amp1.py  (nn.GRU cell, gate order r,z,n)         */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int H, F, dgru;
    const real *w_ih, *w_hh, *b_ih, *b_hh, *w_out, *b_out, *w_hid, *b_hid;
    int64_t o_w_ih, o_w_hh, o_b_ih, o_b_hh, o_w_out, o_b_out, o_w_hid, o_b_hid;
} gru_params_t;

static void gru_layout(const odpd_model_t* m, const real* p, gru_params_t* g) {
    int64_t H = m->hidden, F = feat_dim(m->backbone), o = 0;
    g->H = (int)H; g->F = (int)F; g->dgru = (m->backbone == ODPD_DGRU);
    g->o_w_ih = o; o += 3 * H * F;
    g->o_w_hh = o; o += 3 * H * H;
    g->o_b_ih = o; o += 3 * H;
    g->o_b_hh = o; o += 3 * H;
    int64_t OW = g->dgru ? H + 6 : H;
    g->o_w_out = o; o += 2 * OW;
    g->o_b_out = o; o += 2;
    g->o_w_hid = o; if (g->dgru) o += H * H;
    g->o_b_hid = o; if (g->dgru) o += H;
    g->w_ih = p + g->o_w_ih; g->w_hh = p + g->o_w_hh; g->b_ih = p + g->o_b_ih; g->b_hh = p + g->o_b_hh;
    g->w_out = p + g->o_w_out; g->b_out = p + g->o_b_out; g->w_hid = p + g->o_w_hid; g->b_hid = p + g->o_b_hid;
}

typedef struct { real f[MAXF], hp[MAXH], r[MAXH], z[MAXH], n[MAXH], ghn[MAXH], hid[MAXH], h[MAXH]; } gru_step_t;

/* one sequence forward; if S != NULL every step's activations are kept for the backward pass */
static void gru_seq_fwd(const odpd_model_t* m, const gru_params_t* g, int T, const real* x, real* y, gru_step_t* S) {
    int H = g->H, F = g->F;
    real h[MAXH] = {0}, hn[MAXH];
    gru_step_t tmp;
    for (int t = 0; t < T; ++t) {
        gru_step_t* s = S ? &S[t] : &tmp;
        feat_fwd(m->backbone, x[2 * t], x[2 * t + 1], s->f);
        for (int j = 0; j < H; ++j) {
            real gi[3], gh[3];
            for (int k = 0; k < 3; ++k) {
                real a = g->b_ih[k * H + j], b = g->b_hh[k * H + j];
                for (int i = 0; i < F; ++i) a += g->w_ih[(k * H + j) * F + i] * s->f[i];
                for (int i = 0; i < H; ++i) b += g->w_hh[(k * H + j) * H + i] * h[i];
                gi[k] = a; gh[k] = b;
            }
            s->hp[j] = h[j];
            s->r[j] = sigm(gi[0] + gh[0]);
            s->z[j] = sigm(gi[1] + gh[1]);
            s->ghn[j] = gh[2];
            s->n[j] = tanhr(gi[2] + s->r[j] * gh[2]);
            hn[j] = ((real)1 - s->z[j]) * s->n[j] + s->z[j] * h[j];
        }
        for (int j = 0; j < H; ++j) { h[j] = hn[j]; s->h[j] = hn[j]; }
        /* output head */
        if (g->dgru) {  /* dgru.py:71-73: out = relu(fc_hid(h)); y = fc_out(cat(out, feat)) */
            for (int j = 0; j < H; ++j) {
                real a = g->b_hid[j];
                for (int i = 0; i < H; ++i) a += g->w_hid[j * H + i] * h[i];
                s->hid[j] = a;  /* pre-activation */
            }
            for (int c = 0; c < 2; ++c) {
                real a = g->b_out[c];
                for (int j = 0; j < H; ++j) a += g->w_out[c * (H + 6) + j] * (s->hid[j] > 0 ? s->hid[j] : (real)0);
                for (int i = 0; i < 6; ++i) a += g->w_out[c * (H + 6) + H + i] * s->f[i];
                y[2 * t + c] = a;
            }
        } else {        /* gru.py:46-47 */
            for (int c = 0; c < 2; ++c) {
                real a = g->b_out[c];
                for (int j = 0; j < H; ++j) a += g->w_out[c * H + j] * h[j];
                y[2 * t + c] = a;
            }
        }
    }
}

/* one sequence backward; dp (P reals) is accumulated; dx (T,2) overwritten if not NULL */
static void gru_seq_bwd(const odpd_model_t* m, const gru_params_t* g, int T, const real* x, const real* dy,
                        const gru_step_t* S, real* dp, real* dx) {
    int H = g->H, F = g->F;
    real dh[MAXH] = {0};
    for (int t = T - 1; t >= 0; --t) {
        const gru_step_t* s = &S[t];
        real df[MAXF] = {0};
        real dht[MAXH];
        for (int j = 0; j < H; ++j) dht[j] = dh[j];
        if (g->dgru) {
            real dhid[MAXH];
            for (int j = 0; j < H; ++j) dhid[j] = 0;
            for (int c = 0; c < 2; ++c) {
                real d = dy[2 * t + c];
                dp[g->o_b_out + c] += d;
                for (int j = 0; j < H; ++j) {
                    real a = s->hid[j] > 0 ? s->hid[j] : (real)0;
                    dp[g->o_w_out + c * (H + 6) + j] += d * a;
                    dhid[j] += d * g->w_out[c * (H + 6) + j];
                }
                for (int i = 0; i < 6; ++i) {
                    dp[g->o_w_out + c * (H + 6) + H + i] += d * s->f[i];
                    df[i] += d * g->w_out[c * (H + 6) + H + i];
                }
            }
            for (int j = 0; j < H; ++j) {
                real d = s->hid[j] > 0 ? dhid[j] : (real)0;
                dp[g->o_b_hid + j] += d;
                for (int i = 0; i < H; ++i) {
                    dp[g->o_w_hid + j * H + i] += d * s->h[i];
                    dht[i] += d * g->w_hid[j * H + i];
                }
            }
        } else {
            for (int c = 0; c < 2; ++c) {
                real d = dy[2 * t + c];
                dp[g->o_b_out + c] += d;
                for (int j = 0; j < H; ++j) {
                    dp[g->o_w_out + c * H + j] += d * s->h[j];
                    dht[j] += d * g->w_out[c * H + j];
                }
            }
        }
        /* cell backward: h = (1-z) n + z hp */
        real dhp[MAXH];
        for (int j = 0; j < H; ++j) dhp[j] = dht[j] * s->z[j];
        for (int j = 0; j < H; ++j) {
            real dn = dht[j] * ((real)1 - s->z[j]);
            real dz = dht[j] * (s->hp[j] - s->n[j]);
            real dnp = dn * ((real)1 - s->n[j] * s->n[j]);
            real dr = dnp * s->ghn[j];
            real dghn = dnp * s->r[j];
            real drp = dr * s->r[j] * ((real)1 - s->r[j]);
            real dzp = dz * s->z[j] * ((real)1 - s->z[j]);
            real dgi[3] = {drp, dzp, dnp}, dgh[3] = {drp, dzp, dghn};
            for (int k = 0; k < 3; ++k) {
                dp[g->o_b_ih + k * H + j] += dgi[k];
                dp[g->o_b_hh + k * H + j] += dgh[k];
                for (int i = 0; i < F; ++i) {
                    dp[g->o_w_ih + (k * H + j) * F + i] += dgi[k] * s->f[i];
                    df[i] += dgi[k] * g->w_ih[(k * H + j) * F + i];
                }
                for (int i = 0; i < H; ++i) {
                    dp[g->o_w_hh + (k * H + j) * H + i] += dgh[k] * s->hp[i];
                    dhp[i] += dgh[k] * g->w_hh[(k * H + j) * H + i];
                }
            }
        }
        for (int j = 0; j < H; ++j) dh[j] = dhp[j];
        if (dx) feat_bwd(m->backbone, x[2 * t], x[2 * t + 1], df, &dx[2 * t], &dx[2 * t + 1]);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* dispatch                                                                                     */
/* ------------------------------------------------------------------------------------------ */
static int is_gru_family(int bb) { return bb == ODPD_GRU || bb == ODPD_DGRU || bb == ODPD_QGRU || bb == ODPD_QGRU_AMP1; }

int oracle_backbone_fwd(const odpd_model_t* m, int B, int T, const real* params, const real* x, real* y, double* stats) {
    if (!m || !params || !x || !y || B <= 0 || T <= 0 || m->hidden > MAXH) return ODPD_EINVAL;
    (void)stats;
    if (is_gru_family(m->backbone)) {
        gru_params_t g; gru_layout(m, params, &g);
#pragma omp parallel for schedule(static)
        for (int b = 0; b < B; ++b) gru_seq_fwd(m, &g, T, x + (int64_t)b * T * 2, y + (int64_t)b * T * 2, NULL);
        return 0;
    }
    return ODPD_EUNSUPPORTED;
}

/* dparams (P) is OVERWRITTEN with the sum over the batch; dx nullable. y (nullable) also returned. */
int oracle_backbone_bwd(const odpd_model_t* m, int B, int T, const real* params, const real* x, const real* dy,
                        real* dparams, real* dx) {
    if (!m || !params || !x || !dy || !dparams || B <= 0 || T <= 0 || m->hidden > MAXH) return ODPD_EINVAL;
    int64_t P = oracle_param_count(m);
    if (P < 0) return ODPD_EUNSUPPORTED;
    memset(dparams, 0, sizeof(real) * P);
    if (is_gru_family(m->backbone)) {
        gru_params_t g; gru_layout(m, params, &g);
#pragma omp parallel
        {
            real* dp = (real*)calloc(P, sizeof(real));
            gru_step_t* S = (gru_step_t*)malloc(sizeof(gru_step_t) * T);
            real* ytmp = (real*)malloc(sizeof(real) * 2 * T);
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b) {
                const real* xb = x + (int64_t)b * T * 2;
                gru_seq_fwd(m, &g, T, xb, ytmp, S);
                gru_seq_bwd(m, &g, T, xb, dy + (int64_t)b * T * 2, S, dp, dx ? dx + (int64_t)b * T * 2 : NULL);
            }
#pragma omp critical
            for (int64_t i = 0; i < P; ++i) dparams[i] += dp[i];
            free(dp); free(S); free(ytmp);
        }
        return 0;
    }
    return ODPD_EUNSUPPORTED;
}

/* ------------------------------------------------------------------------------------------ */
/* loss: nn.MSELoss() / nn.L1Loss() mean reduction (project.py:262-272) + backward             */
/* ------------------------------------------------------------------------------------------ */
double oracle_loss_fwd_bwd(int kind, int64_t n, int64_t count, const real* y, const real* target, real* dy) {
    double acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        real d = y[i] - target[i];
        if (kind == ODPD_LOSS_L2) {
            acc += (double)d * (double)d;
            if (dy) dy[i] = (real)2 * d / (real)count;
        } else {
            acc += fabs((double)d);
            if (dy) dy[i] = (d > 0 ? (real)1 : (d < 0 ? (real)-1 : (real)0)) / (real)count;
        }
    }
    return acc / (double)count;
}

/* ------------------------------------------------------------------------------------------ */
/* clip_grad_norm_ (train_funcs.py:41-42) + torch.optim.AdamW single-tensor step (project.py:283)*/
/* tensor_sizes: the P parameters split into n_tensors tensors (clip_grad_norm_ takes the 2-norm */
/* of per-tensor 2-norms).  grad is scaled in place like clip_grad_norm_.                        */
/* ------------------------------------------------------------------------------------------ */
double oracle_clip_adamw_step(int64_t P, real* params, real* grad, real* exp_avg, real* exp_avg_sq, int64_t step,
                              double lr, double beta1, double beta2, double eps, double weight_decay,
                              double max_norm, const int64_t* tensor_sizes, int n_tensors) {
    double total = 0;
    if (tensor_sizes && n_tensors > 0) {
        int64_t o = 0;
        for (int k = 0; k < n_tensors; ++k) {
            real s = 0;
            for (int64_t i = 0; i < tensor_sizes[k]; ++i) s += grad[o + i] * grad[o + i];
            real nk = (real)sqrt((double)s);
            total += (double)nk * (double)nk;
            o += tensor_sizes[k];
        }
    } else {
        for (int64_t i = 0; i < P; ++i) total += (double)grad[i] * (double)grad[i];
    }
    real total_norm = (real)sqrt(total);
    if (max_norm > 0) {
        real coef = (real)max_norm / (total_norm + (real)1e-6);
        if (coef > (real)1) coef = (real)1;
        for (int64_t i = 0; i < P; ++i) grad[i] *= coef;
    }
    double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    real step_size = (real)(lr / bc1), bc2s = (real)sqrt(bc2);
    real decay = (real)(1.0 - lr * weight_decay), w1 = (real)(1.0 - beta1), b2 = (real)beta2, w2 = (real)(1.0 - beta2);
    for (int64_t i = 0; i < P; ++i) {
        real g = grad[i];
        params[i] *= decay;
        exp_avg[i] += (g - exp_avg[i]) * w1;               /* lerp_ */
        exp_avg_sq[i] = exp_avg_sq[i] * b2 + w2 * g * g;   /* mul_ + addcmul_ */
        real denom = (real)sqrt((double)exp_avg_sq[i]) / bc2s + (real)eps;
        params[i] -= step_size * (exp_avg[i] / denom);     /* addcdiv_ */
    }
    return (double)total_norm;
}

/* One whole train step (train_funcs.py:33-44) on the CPU; returns the loss. Used by tests and as
 * bench.py's `cpu_baseline` ("port").  scratch: y,dy (B*T*2 each), grad (P). */
double oracle_train_step(const odpd_model_t* m, int loss_kind, int B, int T, real* params, const real* x,
                         const real* target, real* exp_avg, real* exp_avg_sq, int64_t step, double lr,
                         double max_norm, real* y, real* dy, real* grad) {
    int64_t n = (int64_t)B * T * 2, P = oracle_param_count(m);
    if (oracle_backbone_fwd(m, B, T, params, x, y, NULL)) return NAN;
    double loss = oracle_loss_fwd_bwd(loss_kind, n, n, y, target, dy);
    if (oracle_backbone_bwd(m, B, T, params, x, dy, grad, NULL)) return NAN;
    oracle_clip_adamw_step(P, params, grad, exp_avg, exp_avg_sq, step, lr, 0.9, 0.999, 1e-8, 0.01, max_norm, NULL, 0);
    return loss;
}

int oracle_real_bytes(void) { return (int)sizeof(real); }
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
